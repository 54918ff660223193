"""BASELINE config 4 (N independent prompts, one avatar per GPU, no collective): thin wrapper of avatarclip_amd/replicas.py.
    python scripts/run_replicas.py --confs confs/a.conf confs/b.conf --gpus 0,1 -- --allow_standins"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd.replicas import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
