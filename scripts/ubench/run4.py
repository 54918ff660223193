"""Round-6 micro-benchmark: MFMA + VALU / transcendental fillers in SHADER CYCLES (s_memtime), 1 / 2 / 3 wavefronts per SIMD.
    python scripts/ubench/run4.py      (builds libubench4.so beside itself when missing)"""
import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libubench4.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "ubench4.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "ubench4.hip")])
lib = ctypes.CDLL(so)
lib.ub4.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
names = {0: "MFMA only (2 acc)", 1: "4 v_fma only", 2: "MFMA + 4 v_fma", 3: "8 v_fma only", 4: "MFMA + 8 v_fma", 5: "2 v_exp(+mul) only", 6: "MFMA + 2 v_exp",
         7: "4 v_exp only", 8: "MFMA + 4 v_exp", 9: "5 v_fma + 3 v_exp only", 10: "MFMA + 5 v_fma + 3 v_exp (2 acc)", 11: "MFMA + 5 v_fma + 3 v_exp (1 acc)",
         12: "MFMA only (1 acc: dependent chain)", 13: "MFMA + 8 v_fma (4 acc)", 14: "12 v_fma only", 15: "MFMA + 12 v_fma", 16: "2 v_fma only", 17: "MFMA + 2 v_fma"}
iters = 2000
print("# shader cycles per slot PER SIMD (s_memtime; 32 = the matrix pipe saturated), and the wall-clock view of the same launch; 256 workgroups, one per CU")
for waves_per_simd in (1, 2, 3):
    waves = 4 * waves_per_simd
    out = torch.zeros(256 * waves, dtype=torch.int64, device="cuda")
    for v in (0, 12, 16, 17, 1, 2, 3, 4, 14, 15, 13, 5, 6, 7, 8, 9, 10, 11):
        assert lib.ub4(v, waves, 256, iters, out.data_ptr(), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.ub4(v, waves, 256, iters, out.data_ptr(), st); e1.record(); torch.cuda.synchronize()
        cyc = out.double().mean().item() / (iters * 16) / waves_per_simd
        ms = e0.elapsed_time(e1)
        ghz = out.double().mean().item() / (ms * 1e-3) / 1e9
        print("waves/SIMD %d  %-36s %7.1f cycles per slot and SIMD   (%.3f ms, %.2f GHz effective)" % (waves_per_simd, names[v], cyc, ms, ghz), flush=True)
