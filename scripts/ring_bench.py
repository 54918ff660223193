"""Round-4 development aid: the backward pair (backward sweeps + weight-gradient products) through the G region vs the role-specialised
launch with the L2 ring (csrc/avc_bwd_ring.hip), event-timed, with the ring's own counters.
    python scripts/ring_bench.py [npts] [mode ...]      mode = plain | ring:<cpt>:<slots>
Under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE the per-kernel counters say how many of the handed-off bytes reached HBM.
Since round 5 the ring kernel is not part of libavc.so: python -m avatarclip_amd.build --ring; AVC_LIB_NAME=libavc_ring.so python scripts/ring_bench.py ..."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import fields, renderer
from avatarclip_amd.engine import Engine
dev = torch.device("cuda"); torch.manual_seed(0)
sdf = fields.SDFNetwork(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6).to(dev)
col = fields.RenderingNetwork(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2, extra_color=True).to(dev)
var = fields.SingleVarianceNetwork(0.3).to(dev)
ren = renderer.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0, True)
eng = ren.engine; pk = eng.pack(ren.flat_params())
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
modes = sys.argv[2:] or ["plain", "ring:2:6"]
reps = int(os.environ.get("RING_REPS", "3"))
R = npts // 64
ro = torch.randn(R, 3, device=dev) * 0.1; rd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(R, 64, device=dev) * 2, dim=-1)[0].contiguous()
dsdf = torch.randn(R, 64, device=dev); dn = torch.randn(R, 64, 3, device=dev) * 0.1; drgb = torch.randn(R, 64, 6, device=dev) * 0.1
# experiment: the operand panels in memory of another kind (PANEL_ALLOC = uncached | finegrained): pure streaming data that an L2
# cannot help -- does keeping it out of the L2 leave the ring (and the packed weights) alone?
kind = os.environ.get("PANEL_ALLOC", "default")
if kind != "default":
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]

    class RawBuf:
        is_cuda = True

        def __init__(self, nbytes):
            p = ctypes.c_void_p()
            rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, {"uncached": 3, "finegrained": 1}[kind])
            assert rc == 0 and p.value, ("hipExtMallocWithFlags", rc)
            self.p, self.n = p.value, nbytes

        def data_ptr(self): return self.p
        def is_contiguous(self): return True
        def numel(self): return self.n
        def element_size(self): return 1
    nb = (npts + 31) // 32
    fbuf, gbuf = RawBuf((nb + 1) * eng.fwd_tiles * 2048), RawBuf((nb + 1) * eng.grad_tiles * 2048)
    mbuf = torch.empty((nb + 1) * eng.mask_u16, dtype=torch.int16, device=dev)
    eng._bufs_f = lambda n: (fbuf, mbuf)
    eng._bufs_g = lambda n: gbuf
    eng.plan = lambda R_, S_: (R_, R_)
    print("# operand panels in %s memory" % kind)
_, _, rgbf = eng.points_fwd_train(pk, ro, rd, z, 2 / 32)
torch.cuda.synchronize()
g_ref = None
for mode in modes:
    if mode == "plain":
        Engine.RING = False
    else:
        _, cpt, slots = mode.split(":")
        Engine.RING, Engine.RING_CPT, Engine.RING_SLOTS = True, int(cpt), int(slots)
    Engine.RING_CHECK = False
    for rep in range(2):
        Engine.PROFILE = rep == 1
        Engine.prof_events = []
        for _ in range(reps if rep else 1):
            g = eng.points_bwd(pk, ro, rd, z, 2 / 32, dsdf, dn, drgb, rgbf, panels_valid=True)
        torch.cuda.synchronize()
    acc = {}
    for name, n, e0, e1 in Engine.prof_events:
        acc.setdefault(name, []).append(e0.elapsed_time(e1))
    nslab = len(acc[next(iter(acc))]) // reps
    line = "%-10s npts %d (%d slab(s))  " % (mode, npts, nslab) + "  ".join("%s %.3f ms" % (k.replace("avc_", ""), sum(v) / reps) for k, v in acc.items())
    line += "  pair %.3f ms" % (sum(sum(v) for v in acc.values()) / reps)
    if mode != "plain":
        Engine.RING_CHECK = True          # one more pass with the synchronising check: error word + counters of the LAST slab's launch
        Engine.PROFILE = False
        g = eng.points_bwd(pk, ro, rd, z, 2 / 32, dsdf, dn, drgb, rgbf, panels_valid=True)
        st = eng.ring_stats
        pw, cw = int(st[7]), int(st[5])
        line += ("\n           last launch: %d producer / %d consumer workgroups, %d iterations, %d units; per producer iteration: slot wait %.2f us, "
                 "hand-off %.2f us (both layers), %.1f polls; per consumer: idle %.1f us of the launch"
                 % (pw, cw, int(st[6]), int(st[4]), 0.01 * int(st[0]) / max(int(st[6]), 1), 0.01 * int(st[1]) / max(int(st[6]), 1),
                    int(st[2]) / max(int(st[6]), 1), 0.01 * int(st[3]) / max(cw, 1)))
    if g_ref is None:
        g_ref = g.clone()
    else:
        line += "\n           max rel diff vs first mode: %.2e" % float((g - g_ref).abs().max() / g_ref.abs().max())
    print(line, flush=True)
