R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 1 2; do timeout 1500 python -m pytest tests/test_gpu_glue.py tests/test_smpl_prior.py tests/test_gpu_iteration.py tests/test_gpu_dataset_train.py tests/test_shapegen.py -q 2>&1 | tail -2; done
