"""weight-gradient kernel per pair shape (development aid): each (ta, tb, type_a, type_b) of the real pair list alone over 4 Mi
points of panels with the real region strides (F region 89 tiles, G region 83 tiles per block): python scripts/wg_layout_probe.py"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import lib as L
lib = L.load()
dev = torch.device("cuda")
nblk = 1 << 17     # 4 Mi points
st = torch.cuda.current_stream().cuda_stream
nsplit = 256
ftiles, gtiles = 89, 83
fpanels = torch.zeros(nblk * ftiles * 1024, dtype=torch.int16, device=dev)
gpanels = torch.zeros(nblk * gtiles * 1024, dtype=torch.int16, device=dev)
out = torch.empty(nsplit, 4 * 72 * 1024, device=dev)
bo = torch.empty(nsplit, 1024, device=dev)
tot = 0.0
for (ta, tb, tya, tyb, count) in ((8, 2, 1, 0, 2), (8, 2, 0, 1, 1), (8, 8, 1, 0, 3), (8, 8, 0, 1, 2), (7, 8, 1, 0, 1), (7, 8, 0, 1, 1), (8, 7, 1, 0, 1),
                                  (1, 7, 1, 0, 1), (1, 2, 1, 0, 1), (1, 7, 1, 1, 1), (1, 2, 1, 1, 1), (8, 9, 1, 0, 1), (1, 8, 1, 0, 1)):
    pairs = np.array([[0, ta, 20, tb, 0, -1, tya, tyb]], dtype=np.int32)
    def run():
        rc = lib.avc_weight_grad_all(fpanels.data_ptr(), ftiles, gpanels.data_ptr(), gtiles, 1, pairs.ctypes.data, nblk, out.data_ptr(), bo.data_ptr(), nsplit, out.stride(0), bo.stride(0), st)
        assert rc == 0
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    gb = nblk * (ta + tb) * 2048 / 1e9
    tot += ms * count
    print("ta %d tb %d types %d/%d  x%d: %.3f ms for %.2f GB of tiles: %.0f GB/s" % (ta, tb, tya, tyb, count, ms, gb, gb / ms * 1e3), flush=True)
print("sum over the real pair list: %.3f ms" % tot)
# cost of the bias row sums (16 cvt + 16 add per A tile and block on the waves with wb = 0): the same 8 x 8 pair with and without
for boff in (-1, 0):
    pairs = np.array([[0, 8, 20, 8, 0, boff, 1, 0]], dtype=np.int32)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    print("8 x 8 pair (bf16 x f16), bias sums %s: %.3f ms" % ("on" if boff >= 0 else "off", e0.elapsed_time(e1)))
# the same 8 x 8 pair on random data (zeros above): does the data matter (toggle power, clocks)?
fpanels.copy_(torch.randint(-2000, 2000, fpanels.shape, dtype=torch.int16, device=dev))
gpanels.copy_(torch.randint(-2000, 2000, gpanels.shape, dtype=torch.int16, device=dev))
pairs = np.array([[0, 8, 20, 8, 0, -1, 1, 0]], dtype=np.int32)
for rep in range(2):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    print("8 x 8 pair on random data: %.3f ms" % e0.elapsed_time(e1))
