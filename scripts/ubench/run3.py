"""Round-3 micro-benchmark (development aid): python scripts/ubench/run3.py  -> profiles/r03_ubench3.txt
Producer -> consumer hand-off of 2-KiB operand tiles through a ring in the XCD's L2 (scripts/ubench/ubench3.hip), beside the plain
streaming write / read of the same tiles.  Under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` the per-kernel counters say whether
the ring bytes reach HBM (every ring size is its own kernel instantiation)."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libubench3.so"))
P = ctypes.c_void_p
lib.ub3_ring.argtypes = [ctypes.c_int, ctypes.c_int, P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
lib.ub3_stream.argtypes = [ctypes.c_int, P, ctypes.c_long, ctypes.c_int, P]
st = torch.cuda.current_stream().cuda_stream
dev = "cuda"


def timed(fn, reset):
    reset(); fn(); torch.cuda.synchronize()
    reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


GRID, PAIRS, STEPS = 256, 16, 4096          # 256 workgroups = one per CU, 32 per XCD = 16 producer / consumer pairs per XCD
SLOT_BYTES = 8 * 2048                       # one step = 8 tiles of 2 KiB (one per wavefront)
print("# %d workgroups of 8 wavefronts, %d pairs per XCD, %d steps of %d KiB per pair; hand-off bytes per launch %.2f GB"
      % (GRID, PAIRS, STEPS, SLOT_BYTES // 1024, 8 * PAIRS * STEPS * SLOT_BYTES / 1e9))
for mode, slots in ((0, 4), (0, 8), (0, 64), (1, 8)):
    data = torch.zeros(16 * PAIRS * slots * SLOT_BYTES // 4, dtype=torch.int32, device=dev)
    prod = torch.zeros(16 * PAIRS * 32, dtype=torch.int32, device=dev)
    cons = torch.zeros_like(prod)
    ticket = torch.zeros(16, dtype=torch.int32, device=dev)
    err = torch.zeros(4, dtype=torch.int32, device=dev)

    def reset():
        prod.zero_(); cons.zero_(); ticket.zero_()
    ms = timed(lambda: lib.ub3_ring(mode, GRID, data.data_ptr(), prod.data_ptr(), cons.data_ptr(), ticket.data_ptr(), err.data_ptr(),
                                    PAIRS, slots, STEPS, st), reset)
    nb = 8 * PAIRS * STEPS * SLOT_BYTES
    e = err.tolist()
    print("ring mode %d (%s)  slots %2d = %5.2f MB per XCD : %8.3f ms  %7.1f GB/s handed off (%.1f GB/s per pair)   mismatches %d  time-outs %d  unpaired %d  tickets %s"
          % (mode, "same-XCD, no write-back, sc1 loads" if mode == 0 else "agent-scope release / acquire", slots, PAIRS * slots * SLOT_BYTES / 1e6,
             ms, nb / ms / 1e6, nb / ms / 1e6 / (8 * PAIRS), e[0], e[1], e[2], ticket.tolist()[:8]), flush=True)
# the same tiles streamed to / from a 17-GB buffer (what the operand panels do today)
tiles_per_wg = 4096 * 8
buf = torch.zeros(GRID * tiles_per_wg * 512, dtype=torch.int32, device=dev)
for write, what in ((1, "streaming write"), (0, "streaming read")):
    ms = timed(lambda: lib.ub3_stream(GRID, buf.data_ptr(), tiles_per_wg, write, st), lambda: None)
    print("%-16s of %.1f GB in 2-KiB tiles: %8.3f ms  %7.1f GB/s" % (what, GRID * tiles_per_wg * 2048 / 1e9, ms, GRID * tiles_per_wg * 2048 / ms / 1e6), flush=True)

# ---- does the Infinity Cache (256 MB, memory side) keep freshly WRITTEN tiles for a reader that comes right after?  write X then read X
# (candidate hits) against write X then read a cold Y of the same size, over sizes around the cache capacity.  If the second kernel is
# clearly faster on X below ~256 MB, a slab-sized producer -> consumer pipeline could take its G tiles from the cache; if not, only
# the L2 rings above can.
import sys
if "mall" in sys.argv:
  for wmode, wname in ((1, "nt stores"), (2, "normal-policy stores")):
    print("# write X (%s) -> read X | read cold Y; MB, us per read launch, GB/s" % wname)
    for mb in (8, 32, 64, 128, 192, 256, 384, 768):
        tiles_per_wg = mb * (1 << 20) // 2048 // GRID
        n = GRID * tiles_per_wg * 512
        X = torch.zeros(n, dtype=torch.int32, device=dev)
        Y = torch.zeros(n, dtype=torch.int32, device=dev)
        big = torch.zeros(1 << 28, dtype=torch.int32, device=dev)      # 1 GiB: flushes the caches between repetitions
        res = {}
        for name, src in (("same", X), ("cold", Y)):
            ts = []
            for rep in range(6):
                big.add_(1)                                             # evict
                lib.ub3_stream(GRID, X.data_ptr(), tiles_per_wg, wmode, st)  # producer writes X
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); lib.ub3_stream(GRID, src.data_ptr(), tiles_per_wg, 0, st); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res[name] = sorted(ts)[len(ts) // 2]
        b = GRID * tiles_per_wg * 2048
        print("%4d MB  read-after-write %8.1f us (%6.0f GB/s)   cold read %8.1f us (%6.0f GB/s)"
              % (mb, res["same"], b / res["same"] / 1e3, res["cold"], b / res["cold"] / 1e3), flush=True)
        del X, Y, big
