"""The role-specialised backward (csrc/avc_bwd_ring.hip, include/avc_ring.h) is an EXPERIMENT that lives in libavc_ring.so, not in the
product's libavc.so.  Its parity cases (tests/ring_cases.py: ring path == panel path on both nets, ragged sizes, slot reuse, 1-3
consumers per product) run in a process of their own that loads that library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_ring_cases_in_their_own_process_with_the_experimental_library():
    from avatarclip_amd import build
    path = build.build(ring=True)
    assert os.path.basename(path) == "libavc_ring.so"
    env = dict(os.environ, AVC_LIB_NAME="libavc_ring.so")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "ring_cases.py"), "-x", "-q", "-m", "gpu", "-o",
                        "python_files=ring_cases.py", "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]
