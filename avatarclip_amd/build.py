"""In-tree build of libavc.so (hipcc, gfx950).  The .so is git-ignored but travels with the gpurun snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, os.environ.get("AVC_LIB_NAME", "libavc.so"))
# csrc/avc_bwd_ring.hip (the role-specialised backward of round 4, measured slower: profiles/r04_ring_handoff.md) is NOT part of
# libavc.so; `build(ring=True)` / `python -m avatarclip_amd.build --ring` / AVC_WITH_RING=1 links it, with everything else, into
# libavc_ring.so (include/avc_ring.h), which AVC_LIB_NAME=libavc_ring.so AVC_BWD_RING=1 selects.
RING_SOURCE = "avc_bwd_ring.hip"
RING_LIB = os.path.join(HERE, "libavc_ring.so")
SOURCES = ["avc_core.hip", "avc_mlp_fwd.hip", "avc_mlp_bwd.hip", "avc_wgrad.hip", "avc_rays.hip", "avc_vit.hip", "avc_vit_attn.hip", "avc_vit_gemm.hip", "avc_mcubes.hip", "avc_raster.hip", "avc_params.hip", "avc_glue.hip"]
HEADERS = ["avc_common.h", "avc_stage.h", "avc_mlp.h", "avc_bwd_body.h", "avc_wgrad_body.h", "avc_offsets_gen.h", os.path.join("..", "..", "include", "avc.h"),
           os.path.join("..", "..", "include", "avc_ring.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"] + os.environ.get("AVC_EXTRA_FLAGS", "").split()
# avc_mlp_fwd.hip holds the one kernel built for ONE wavefront per SIMD on the 512-entry unified register file (mlp_sdf2_kernel, 256-thread
# workgroups): hipcc then picks the AGPR form of the MFMAs (accumulators in the accumulation half, one v_accvgpr_read per element in front of
# every epilogue).  The VGPR form keeps the accumulators where the epilogues read them; the kernel puts the ACTIVATIONS into AGPRs itself.
# Every other kernel of that file has 512+ threads per workgroup and uses no AGPRs either way (identical code with and without the flag).
SOURCE_FLAGS = {"avc_mlp_fwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _gen_offsets():
    """csrc/avc_offsets_gen.h = the packed-blob offsets of packing.py as compile-time constants (scripts/gen_offsets.py)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_offsets", os.path.join(os.path.dirname(HERE), "scripts", "gen_offsets.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()


def build(force: bool = False, verbose: bool = False, ring: bool = False) -> str:
    """libavc.so (or $AVC_LIB_NAME); ring=True: libavc_ring.so = the same objects + csrc/avc_bwd_ring.hip"""
    _gen_offsets()
    # AVC_LIB_NAME=libavc_ring.so makes LIB the ring library's path: a plain build() must then not relink it without its ring object
    ring = ring or os.environ.get("AVC_WITH_RING", "0") == "1" or os.path.basename(LIB) == os.path.basename(RING_LIB)
    target = RING_LIB if ring else LIB
    srcs = [s for s in SOURCES + ([RING_SOURCE] if ring else []) if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", os.environ.get("AVC_OBJ_SUFFIX", "") + ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc()] + FLAGS + SOURCE_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr[-4000:]))

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(target, objs):
        tmp = "%s.%d.tmp" % (target, os.getpid())       # concurrent builders: nobody ever maps a half-written library
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs)
        os.replace(tmp, target)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, ring="--ring" in sys.argv))
