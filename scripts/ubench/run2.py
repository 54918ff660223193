"""Round-2 micro-benchmarks (development aid): python scripts/ubench/run2.py  -> profiles/r02_ubench_*.txt
(1) MFMA with independent accumulators + VALU fillers, (2) fp32 atomic-add merge of per-slab weight-gradient tiles."""
import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libubench2.so"))
lib.ub2_mfma.argtypes = [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p]
lib.ub2_atomic.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
out = torch.zeros(1024, device="cuda")


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


if "mfma" in sys.argv or len(sys.argv) == 1:
    iters = 3000
    print("# cycles per MFMA slot at 2.4 GHz (32 = matrix pipe saturated); 256 workgroups")
    for waves in (4, 8):
        for trans in (0, 1):
            for nv in (0, 2, 4, 6, 8, 12):
                row = []
                for nacc in (1, 2, 4):
                    ms = timed(lambda: lib.ub2_mfma(nacc, nv, trans, waves, 256, iters, out.data_ptr(), st))
                    # per SIMD: waves/4 waves each issuing iters*16 slots
                    row.append(ms * 1e-3 * 2.4e9 / (iters * 16) / (waves / 4))
                print("waves/WG %d  %-5s fillers/MFMA %2d :  1 acc %6.1f   2 acc %6.1f   4 acc %6.1f  cycles per MFMA (per SIMD)"
                      % (waves, "v_exp" if trans else "v_fma", nv, *row), flush=True)

if "atomic" in sys.argv or len(sys.argv) == 1:
    stride = 400 * 1024          # floats per accumulation buffer (1.6 MB: every dense weight of the full nets)
    tile = 64 * 1024             # floats per merge (one 256x256 fp32 tile)
    passes = 64
    xcc = torch.zeros(16, dtype=torch.int32, device="cuda")
    for mode, name in ((0, "atomic add, workgroup scope, per-XCD buffer"), (1, "atomic add, agent scope, per-XCD buffer"),
                       (2, "plain read-modify-write, private per-workgroup buffer")):
        for grid in (256, 512):
            nbuf = 16 if mode < 2 else grid
            buf = torch.zeros(nbuf * stride, device="cuda")
            xcc.zero_()
            ms = timed(lambda: lib.ub2_atomic(mode, grid, passes, tile, buf.data_ptr(), stride, xcc.data_ptr(), st))
            gb = grid * passes * tile * 4 / 1e9
            total = buf.double().sum().item()
            expect = 2.0 * grid * passes * tile      # two launches (warm-up + timed)
            print("%-58s grid %4d: %8.3f ms  %8.1f GB/s of merged tiles (%.2f Gadd/s)  sum check %s  xcc histogram %s"
                  % (name, grid, ms, gb / (ms * 1e-3), gb / 4 / (ms * 1e-3), "OK" if abs(total - expect) < 1e-3 * expect else "BAD %g vs %g" % (total, expect),
                     xcc.tolist() if mode < 2 else "-"), flush=True)

if "copy" in sys.argv:
    # counter calibration (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): 4 GiB read + 4 GiB written per launch
    lib.ub2_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    ntiles = 1 << 21
    src = torch.zeros(ntiles * 512, dtype=torch.int32, device="cuda"); dst = torch.empty_like(src)
    for nt in (0, 1):
        ms = timed(lambda: lib.ub2_copy(src.data_ptr(), dst.data_ptr(), ntiles, nt, 2048, st))
        print("copy of %d tiles x 2 KiB (nt=%d): %.3f ms, %.0f GB/s read + the same written" % (ntiles, nt, ms, ntiles * 2048 / ms / 1e6), flush=True)
    for nt, what in ((2, "write only"), (3, "read only")):
        ms = timed(lambda: lib.ub2_copy(src.data_ptr(), dst.data_ptr(), ntiles, nt, 2048, st))
        print("%s, %d tiles x 2 KiB: %.3f ms, %.0f GB/s" % (what, ntiles, ms, ntiles * 2048 / ms / 1e6), flush=True)
