"""MFMA / VALU overlap micro-benchmark (development aid): python scripts/ubench/run.py"""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libubench.so"))
lib.ub_run.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(1024, device="cuda")
st = torch.cuda.current_stream().cuda_stream
iters = 4000
names = {0: "MFMA only (16 dependent 32x32x16 per iter)", 1: "v_fma only", 2: "MFMA + v_fma", 3: "v_exp only", 4: "MFMA + v_exp"}
for waves in (4, 8):            # waves per workgroup: 4 = one per SIMD, 8 = two per SIMD
    for nv in (2, 4, 8):
        for mode in (0, 1, 2, 3, 4):
            lib.ub_run(mode, nv, waves, 256, 10, out.data_ptr(), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lib.ub_run(mode, nv, waves, 256, iters, out.data_ptr(), st); e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            cyc = ms * 1e-3 * 2.4e9 / (iters * 16)
            print("waves/WG %d  ops-per-MFMA-slot %d  %-44s %8.3f ms  = %6.1f cycles per slot (at 2.4 GHz)" % (waves, nv, names[mode], ms, cyc), flush=True)
