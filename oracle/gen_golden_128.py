"""Golden fixture of BASELINE config 3's sampling regime (64 coarse + 64 importance samples per ray, four up-sampling steps of 16) by
RUNNING THE REFERENCE'S OWN CODE.  Run in the build container only (needs /root/reference):   python oracle/gen_golden_128.py
TEST INFRASTRUCTURE ONLY.

tests/golden/neus_full_128.npz: the full-size nets of neus_full.npz (the weights are read from that fixture and are NOT stored again),
96 rays of a 32 x 32 view (hits, grazing rays and misses), injected jitter, a per-ray grey background ([R, 1], the silhouette-mode form
of renderer.py:277-281) -- per-step intermediates of the reference's up_sample / cat_z_vals at n = 64, 80, 96, 112 with m = 16
(renderer.py:133-193, 39-69), the outputs of NeuSRenderer.render at S = 128 (renderer.py:195-300) and the reference's autograd
gradients of gen_golden.scalar_loss for every parameter, including the double backward of SDFNetwork.gradient (fields.py:96-107).
The loss coefficients are the SAME on every ray (run_case(coherent=True): one sign per term, as the losses of main.py:489-534 have), so the
1e-2 gate of SURVEY 8d applies to EVERY tensor without the carve-out the random-coefficient fixtures need (profiles/r05_gradient_noise.md).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.gen_golden import GOLD, extract_functions, run_case  # noqa: E402


def main():
    R = ref_loader.load_reference()
    AG = R.ref_ag
    ds_fns = extract_functions(os.path.join(AG, "models", "dataset.py"), ["gen_rays_pose", "near_far_from_sphere"], "SMPL_Dataset")
    ut_fns = extract_functions(os.path.join(AG, "models", "utils.py"), ["norm_np_arr", "lookat"])
    for f in ut_fns.values():
        f.__globals__.update(ut_fns)
    z = np.load(os.path.join(GOLD, "neus_full.npz"))
    sd = {p: {k[len(p) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(p + ".")} for p in ("sdf", "col", "var")}
    sdf = R.SDFNetwork(**ref_loader.FULL_SDF)
    col = R.RenderingNetwork(**ref_loader.FULL_COLOR)
    var = R.SingleVarianceNetwork(0.3)
    sdf.load_state_dict(sd["sdf"]); col.load_state_dict(sd["col"]); var.load_state_dict(sd["var"])
    eye = np.array([-0.5, 0.3, 1.2], dtype=np.float32)
    pose = torch.from_numpy(ut_fns["lookat"](eye, np.zeros(3, np.float32), np.array([0, 1, 0]))).float()
    fake_self = type("DS", (), {})()
    fake_self.W = fake_self.H = 32
    f32 = 0.5 * 32 / np.tan(np.pi / 6)
    fake_self.K = torch.from_numpy(np.array([[f32, 0, 16.0], [0, f32, 16.0], [0, 0, 1]]))
    o, v = ds_fns["gen_rays_pose"](fake_self, pose, 1)
    ro, rd = o.reshape(-1, 3).float().contiguous(), v.reshape(-1, 3).float().contiguous()
    near, far = ds_fns["near_far_from_sphere"](fake_self, ro, rd)
    ren = R.NeuSRenderer(None, sdf, var, col, 64, 64, 0, 4, 0.0, True)
    ws = ren.render(ro, rd, near, far, perturb_overwrite=0)["weight_sum"].detach().reshape(-1)
    hit = torch.nonzero(ws > 0.6).reshape(-1)
    edge = torch.nonzero((ws > 0.05) & (ws <= 0.6)).reshape(-1)
    miss = torch.nonzero(ws <= 0.05).reshape(-1)
    g = torch.Generator().manual_seed(21)
    sel = torch.cat([hit[torch.randperm(len(hit), generator=g)[:56]], edge[torch.randperm(len(edge), generator=g)[:24]],
                     miss[torch.randperm(len(miss), generator=g)[:16]]])
    sel = torch.sort(sel[:96])[0]
    print("128 spp: hit %d edge %d miss %d -> %d rays" % (len(hit), len(edge), len(miss), len(sel)))
    NR = len(sel)
    jitter = torch.rand(NR, 1, generator=torch.Generator().manual_seed(22))
    bg = torch.rand(NR, 1, generator=torch.Generator().manual_seed(23))
    rec = run_case(R, sdf, col, var, 64, 64, 4, ro[sel], rd[sel], near[sel], far[sel], jitter, bg, 0.6, seed=24, coherent=True)
    rec["ray_index"] = sel.numpy()
    assert rec["z_final"].shape == (NR, 128) and rec["up3_z_in"].shape == (NR, 112) and rec["up0_new_z"].shape == (NR, 16)
    np.savez_compressed(os.path.join(GOLD, "neus_full_128.npz"), **rec)
    print("neus_full_128.npz", os.path.getsize(os.path.join(GOLD, "neus_full_128.npz")), "loss", float(rec["loss"]))


if __name__ == "__main__":
    main()
