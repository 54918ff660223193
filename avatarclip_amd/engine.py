"""Device-side driver of the fused gfx950 kernels: parameter packing, kernel launches through the C ABI
(include/avc.h via ctypes) and the autograd.Function that makes render_core differentiable.

There is deliberately no eager/CPU fallback in this file: without libavc.so or without a GPU the calls raise.
"""
import ctypes
import os
import weakref

import numpy as np
import torch

from . import lib as L
from . import packing as PK


class _DevLayout:
    """device copies of the index tables of packing.Layout."""

    _cache = {}

    def __init__(self, lay: PK.Layout, device):
        t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.lay = lay
        self.idx16 = t(lay.idx16)
        self.scale16 = t(lay.scale16)
        self.idx32 = t(lay.idx32)
        self.scale32 = t(lay.scale32)
        self.un_src = t(lay.un_src)
        self.un_tgt = t(lay.un_tgt)
        self.un_scale = t(lay.un_scale)
        self.ub_src = t(lay.ub_src)
        self.ub_tgt = t(lay.ub_tgt)
        # the same maps as a per-parameter list (CSR) over [gout | gbias]: what avc_weight_grad_unpack gathers with
        tgt = np.concatenate([lay.un_tgt, lay.ub_tgt]).astype(np.int64)
        src = np.concatenate([lay.un_src, np.asarray(lay.ub_src, np.int64) + lay.gout_size]).astype(np.int64)
        scl = np.concatenate([lay.un_scale, np.ones(len(lay.ub_tgt), np.float32)]).astype(np.float32)
        order = np.argsort(tgt, kind="stable")
        off = np.zeros(lay.nparam + 1, np.int64)
        np.cumsum(np.bincount(tgt, minlength=lay.nparam), out=off[1:])
        self.csr_off, self.csr_src, self.csr_scale = t(off.astype(np.int32)), t(src[order].astype(np.int32)), t(scl[order])
        # column sums of gbar_hs / gbar_h0 (second-order term of row 0 of the last SDF layer, reduced inside the backward kernel) -> the
        # dense gradient: every target parameter exactly once
        self.cs_src, self.cs_tgt, self.cs_scale = t(lay.cs_src), t(lay.cs_tgt), t(lay.cs_scale)
        self._offsets_arr = (ctypes.c_int * PK.OFF_COUNT)(*[int(v) for v in lay.offsets])
        self.offsets = ctypes.cast(self._offsets_arr, ctypes.c_void_p)

    @classmethod
    def get(cls, spec, device):
        key = (spec.H, spec.NMID, spec.NCMID, str(device))
        if key not in cls._cache:
            cls._cache[key] = cls(PK.layout_for(spec), device)
        return cls._cache[key]


FUSED_PACK = os.environ.get("AVC_FUSED_PACK", "1") != "0"      # Packed / the coarse sample depths as one launch each (0: torch ops)


class Packed:
    """Packed parameter blobs of one optimisation step."""

    def __init__(self, dl: _DevLayout, flatP: torch.Tensor):
        self.dl = dl
        if flatP.is_cuda and FUSED_PACK:     # one launch (csrc/avc_params.hip); below: the torch statement of the same gathers
            f = flatP.detach().float().contiguous()
            n16, n32 = dl.idx16.numel(), dl.idx32.numel()
            self.w_f16 = torch.empty(n16, dtype=torch.float16, device=f.device)
            self.w_bf16 = torch.empty(n16, dtype=torch.bfloat16, device=f.device)
            self.tab = torch.empty(n32, dtype=torch.float32, device=f.device)
            L.check(L.load().avc_pack_params(L.ptr(f), f.numel(), L.ptr(dl.idx16), L.ptr(dl.scale16), n16, L.ptr(dl.idx32), L.ptr(dl.scale32), n32,
                                             L.ptr(self.w_f16), L.ptr(self.w_bf16), L.ptr(self.tab), L.stream()), "avc_pack_params")
            return
        pz = torch.cat([flatP.detach().float(), flatP.new_zeros(1)])
        w = pz[dl.idx16] * dl.scale16
        self.w_f16 = w.to(torch.float16).contiguous()
        self.w_bf16 = w.to(torch.bfloat16).contiguous()
        self.tab = (pz[dl.idx32] * dl.scale32).contiguous()


def flatten_dense_torch(sdf_net, col_net, spec: PK.NetSpec) -> torch.Tensor:
    """Flat differentiable vector of every dense weight in packing.param_shapes order as plain torch expressions (weight norm
    by `fields.dense_weight`): the host-logic statement of flatten_dense, used by the CPU tests and as the parity reference of
    the fused kernel."""
    parts = []
    for W, b in sdf_net.dense():
        parts += [W.reshape(-1), b.reshape(-1)]
    if col_net is not None:
        for W, b in col_net.dense():
            parts += [W.reshape(-1), b.reshape(-1)]
    else:
        lay = PK.layout_for(spec)
        n_sdf = sum(int(np.prod(s)) for n, s in lay.shapes if n.startswith("sdf."))
        parts.append(parts[0].new_zeros(lay.nparam - n_sdf))
    return torch.cat(parts)


def _dense_layers(sdf_net, col_net, lay):
    """[(linear module, flat offset of W, flat offset of b)] in packing.param_shapes order; the stacked 6 x H colour head is two
    linears (lin_last rows 0..2, extra_lin rows 3..5) writing into the same W / b blocks"""
    out = []
    n_sdf = sdf_net.num_layers - 1
    for l in range(n_sdf):
        out.append((getattr(sdf_net, "lin%d" % l), lay.pbase["sdf.W%d" % l], lay.pbase["sdf.b%d" % l]))
    if col_net is not None:
        nh = col_net.num_layers - 2
        for l in range(nh):
            out.append((getattr(col_net, "lin%d" % l), lay.pbase["col.W%d" % l], lay.pbase["col.b%d" % l]))
        last = getattr(col_net, "lin%d" % nh)
        H = last.weight_v.shape[1] if hasattr(last, "weight_v") else last.weight.shape[1]
        out.append((last, lay.pbase["col.Wh"], lay.pbase["col.bh"]))
        if col_net.extra_color:
            out.append((col_net.extra_lin, lay.pbase["col.Wh"] + 3 * H, lay.pbase["col.bh"] + 3))
    return out


class DenseParamsFn(torch.autograd.Function):
    """every dense weight (weight norm resolved: W = g v / ||v||_row, fields.py:65-66) and bias in ONE flat vector, one launch
    forward and one backward (csrc/avc_params.hip) instead of ~4 + ~10 torch kernels per weight-normed linear"""

    @staticmethod
    def forward(ctx, meta, *tensors):
        lay, offs = meta
        n = len(offs)
        dev = tensors[0].device
        flat = torch.zeros(lay.nparam, device=dev, dtype=torch.float32)
        vs, gs, bs = tensors[0::3], tensors[1::3], tensors[2::3]
        PA = ctypes.c_void_p * n
        arr = lambda ts: PA(*[(t.data_ptr() if t.numel() else None) for t in ts])
        ctx.rows = (ctypes.c_int * n)(*[v.shape[0] for v in vs])
        ctx.cols = (ctypes.c_int * n)(*[v.shape[1] for v in vs])
        ctx.w_off = (ctypes.c_long * n)(*[o[0] for o in offs])
        ctx.b_off = (ctypes.c_long * n)(*[o[1] for o in offs])
        ctx.n = n
        lib = L.load()
        assert all(t.is_contiguous() and t.dtype == torch.float32 for t in tensors)
        L.check(lib.avc_dense_params_fwd(n, arr(vs), arr(gs), arr(bs), ctx.rows, ctx.cols, ctx.w_off, ctx.b_off, L.ptr(flat), L.stream()),
                "avc_dense_params_fwd")
        ctx.save_for_backward(*tensors)
        return flat

    @staticmethod
    def backward(ctx, dflat):
        tensors = ctx.saved_tensors
        n = ctx.n
        vs, gs, bs = tensors[0::3], tensors[1::3], tensors[2::3]
        dflat = dflat.contiguous().float()
        outs = [torch.empty_like(t) for t in tensors]
        PA = ctypes.c_void_p * n
        arr = lambda ts: PA(*[(t.data_ptr() if t.numel() else None) for t in ts])
        L.check(L.load().avc_dense_params_bwd(n, arr(vs), arr(gs), arr(outs[0::3]), arr(outs[1::3]), arr(outs[2::3]), ctx.rows, ctx.cols,
                                              ctx.w_off, ctx.b_off, L.ptr(dflat), L.stream()), "avc_dense_params_bwd")
        return (None,) + tuple(o if t.numel() else None for o, t in zip(outs, tensors))


def flatten_dense(sdf_net, col_net, spec: PK.NetSpec) -> torch.Tensor:
    """Flat differentiable vector of every dense weight in packing.param_shapes order (the fused kernel on the device; modules on
    the CPU -- host-logic tests only -- go through the torch statement)."""
    p0 = next(sdf_net.parameters())
    if p0.device.type != "cuda":
        return flatten_dense_torch(sdf_net, col_net, spec)
    lay = PK.layout_for(spec)
    tensors, offs = [], []
    empty = p0.new_zeros(0)
    for lin, w_off, b_off in _dense_layers(sdf_net, col_net, lay):
        if hasattr(lin, "weight_g"):
            tensors += [lin.weight_v, lin.weight_g, lin.bias if lin.bias is not None else empty]
        else:
            tensors += [lin.weight, empty, lin.bias if lin.bias is not None else empty]
        offs.append((int(w_off), int(b_off)))
    return DenseParamsFn.apply((lay, offs), *tensors)


def plan_panels(R, S, budget, blk_f, blk_g, slab_blocks, min_slab_blocks):
    """How a ray set [R rays x S samples] is cut for the training path (pure arithmetic; Engine.plan supplies the budget).
    blk_f / blk_g = bytes per 32-point block of the F region (+ masks) / of the G region.  Returns (rays per chunk, rays per
    slab), both multiples of 32 rays (block-aligned for every S) unless they cover the whole ray set:
      * the slab is `slab_blocks` blocks unless the whole ray set's F panels + one slab exceed the budget -- then it is halved down
        to `min_slab_blocks` (the whole ray set in one chunk is worth smaller slabs);
      * if even that does not fit, the ray set is cut into chunks (F panels of a chunk + one slab <= budget) and the backward
        re-runs the training forward per chunk."""
    nb = lambda rays: (rays * S + 31) // 32 + 1              # + the sink block
    rays_of = lambda blocks: min(R, max(32, blocks * 32 // S // 32 * 32))
    blocks = slab_blocks
    slab = rays_of(blocks)
    while nb(R) * blk_f + nb(slab) * blk_g > budget and blocks > min_slab_blocks:
        blocks //= 2
        slab = rays_of(blocks)
    if nb(R) * blk_f + nb(slab) * blk_g <= budget:
        return R, slab
    if nb(slab) * (blk_f + blk_g) > budget:                  # not even one slab with its own F panels: shrink the slab further
        slab = max(32, (budget // (blk_f + blk_g) - 1) * 32 // S // 32 * 32)
    chunk = max(slab, ((budget - nb(slab) * blk_g) // blk_f - 1) * 32 // S // 32 * 32)
    return min(chunk, R), min(slab, R)


class Engine:
    _by_net = weakref.WeakKeyDictionary()
    PROFILE = False               # bench.py: record (name, points, start_event, end_event) per kernel launch group
    prof_events = []
    WG_BLOCKS_PER_SPLIT = int(os.environ.get("AVC_WG_BLOCKS_PER_SPLIT", "256"))     # split-K of the weight-gradient launch: blocks per workgroup (nsplit = blocks / this, 1..256).  A
                                  # 512^2 x 64 spp slab (262144 blocks) has its 256 splits either way; smaller point sets want the finer deal -- at 224^2 (100352
                                  # blocks) 256 splits x 17 pairs instead of 98 x 17 workgroups over 256 CUs: 8.52 -> 8.08 ms (profiles/r03_ab_kernels.txt)
    WG_MAX_SPLITS = int(os.environ.get("AVC_WG_MAX_SPLITS", "256"))   # (<= 256: the size of the split buffers)
    FUSED_WG_TAIL = os.environ.get("AVC_FUSED_WG_TAIL", "1") != "0"    # split sums + un-packing of the dense gradient as 1 + 1 launches
    MAX_FWD_WAVES = 2048          # persistent grid of avc_render_points_fwd: 256 CUs x one 8-wave workgroup
    MAX_BWD_WAVES = 2048          # 256 CUs x one 8-wave workgroup
    # Operand panels (csrc/avc_mlp.h: PanelLayout).  F region: 89 tiles = 5.6 KiB per point (full nets), written by the training
    # forward for every block of a CHUNK of rays and kept until the backward pass; G region: 83 tiles = 5.2 KiB per point, one SLAB
    # of at most SLAB_BLOCKS 32-point blocks, rewritten slab by slab by the backward pass (backward kernel + weight-gradient kernel
    # per slab).  Every slab boundary costs the weight-gradient launch a tail (profiles/r03_slab_sweep.txt, 512^2 x 64 spp: 1 / 2 / 4
    # / 8 slabs = 39.9 / 40.5 / 41.7 / 44.2 ms), so the slab is as large as the memory allows up to SLAB_BLOCKS and is halved
    # (down to MIN_SLAB_BLOCKS) before the ray set is cut: 512^2 x 64 spp = 87 GiB of F panels + ONE 84-GiB slab (round 4; two
    # slabs before: 121.3 -> 120.6 ms per step, profiles/r04_ab_kernels.txt); 512^2 x 128 spp
    # (BASELINE config 3 per GPU) = 174 GiB + 23-GiB slabs, still ONE chunk on a 288-GB MI355X.  Only when even that does not fit
    # the budget -- min(AVC_PANEL_GIB, 80 % of the free HBM) -- the ray set is cut into chunks and the backward re-runs the
    # training forward chunk by chunk.
    # Role-specialised backward (csrc/avc_bwd_ring.hip): the abar tiles of the middle SDF layers are handed from the backward sweeps
    # to accumulator-owning consumer workgroups of the same XCD through an L2-resident ring instead of through the G region.
    # RING_CPT = consumer workgroups per (XCD, product), RING_SLOTS = ring slots (128 KiB each for the full nets) per (XCD, product).
    # the sdf-bias gradient (= sum of d_sdf over all points, a heavily cancelling sum: the eikonal term pulls both ways) from the fp32
    # cotangent tensor instead of from the hi + lo bf16 tile: ONE reduction per backward.  The tile path measured 0.03 % on the 512^2
    # fixture, but a cancelling sum's relative error grows as the sum approaches zero (late in training, when eikonal and mask terms
    # balance), and the reference accumulates this gradient in fp32 -- so the override is ON; AVC_SDF_BIAS_FP32=0 for the A/B.
    SDF_BIAS_FP32 = os.environ.get("AVC_SDF_BIAS_FP32", "1") != "0"
    RING = os.environ.get("AVC_BWD_RING", "0") != "0"
    RING_CPT = int(os.environ.get("AVC_RING_CPT", "2"))
    RING_SLOTS = int(os.environ.get("AVC_RING_SLOTS", "6"))
    RING_CHECK = os.environ.get("AVC_RING_CHECK", "0") != "0"      # synchronise after every ring launch and raise on its error word
    PANEL_BYTES_BUDGET = int(os.environ.get("AVC_PANEL_GIB", "224")) << 30
    SLAB_BLOCKS = int(os.environ.get("AVC_SLAB_BLOCKS", str(512 * 1024)))
    MIN_SLAB_BLOCKS = 32 * 1024

    def __init__(self, spec: PK.NetSpec, device):
        if device.type != "cuda":
            raise RuntimeError("avatarclip_amd runs its hot path on an MI355X; got device %s (no CPU fallback)" % device)
        self.spec = spec
        self.device = device
        self.lib = L.load()
        self.dl = _DevLayout.get(spec, device)
        self.net = spec.net_id
        assert self.lib.avc_num_offsets() == PK.OFF_COUNT
        self.fwd_tiles = self.lib.avc_fwd_panel_tiles(self.net)      # tiles per block of the F region / of the G region
        self.grad_tiles = self.lib.avc_grad_panel_tiles(self.net)
        assert (self.fwd_tiles, self.grad_tiles) == (self.dl.lay.panel["FTILES"], self.dl.lay.panel["GTILES"]), \
            "panel layout mismatch between packing.py and csrc/avc_mlp.h"
        assert self.lib.avc_bwd_colsum_floats(self.net) == self.dl.lay.cs_size, "column-sum layout mismatch between packing.py and csrc/avc_mlp.h"
        self._colsum = None           # [wavefronts of a backward launch][cs_size]: the backward kernel's per-wavefront column sums
        self.mask_u16 = self.lib.avc_mask_u16_per_block(self.net)
        self.fwd_scr_bytes = self.lib.avc_fwd_scratch_bytes_per_wave(self.net)
        self._fwd_scratch = None
        self._fpanels = None          # F region + masks of the current chunk
        self._gpanels = None          # G region of one slab
        self._partials = None
        self._bpartials = None
        self._masks = None
        self._pairs_host = PK.region_local_pairs(self.dl.lay)   # (launch order = layout order; widest-first measured the same: 40.5 ms)
        self._sdf_bias0 = int(self.dl.lay.pbase["sdf.b%d" % (spec.NMID + 2)])   # flat index of bias[0] of the last SDF layer
        self._packed_key = None
        self._packed = None
        self._panel_owner = None      # token of the RenderCoreFn.forward whose operand panels the buffers hold
        self._ring = None             # buffers of the role-specialised backward (lazily)
        self.ring_stats = None        # RING_CHECK: the u64 counters of the last ring launch (include/avc.h)

    @classmethod
    def for_networks(cls, sdf_net, col_net):
        eng = cls._by_net.get(sdf_net)
        if eng is None:
            col_conf = col_net.conf if col_net is not None else dict(
                d_hidden=sdf_net.conf["d_hidden"], n_layers=2 if sdf_net.conf["d_hidden"] == 256 else 1,
                mode="no_view_dir", multires_view=0)
            spec = PK.spec_from_conf(sdf_net.conf, col_conf)
            eng = cls(spec, next(sdf_net.parameters()).device)
            cls._by_net[sdf_net] = eng
        return eng

    class _Timed:
        def __init__(self, name, npts):
            self.name, self.npts = name, npts

        def __enter__(self):
            if Engine.PROFILE:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *a):
            if Engine.PROFILE:
                self.e1.record()
                Engine.prof_events.append((self.name, self.npts, self.e0, self.e1))
            return False

    # ------------------------------------------------------------------ packing
    def pack(self, flatP: torch.Tensor) -> Packed:
        # (render() packs for the up-sampling passes and RenderCoreFn packs the same vector again: one launch instead of two)
        last = getattr(self, "_last_pack", None)
        if last is not None and last[0]() is flatP and last[1] == flatP._version:
            return last[2]
        pk = Packed(self.dl, flatP)
        self._last_pack = (weakref.ref(flatP), flatP._version, pk)     # (weak: the vector carries last iteration's autograd graph)
        return pk

    def colsum_scratch(self):
        """the per-workgroup sums + ticket word of avc_colsum (zeroed once; calls are stream-ordered, one stream per engine)"""
        if getattr(self, "_colsum_scratch", None) is None:
            self._colsum_scratch = torch.zeros(self.lib.avc_colsum_scratch_bytes(), dtype=torch.uint8, device=self.device)
        return self._colsum_scratch

    def _grow(self, attr, nbytes, dtype=torch.uint8):
        buf = getattr(self, attr)
        esz = torch.empty(0, dtype=dtype).element_size()
        n = nbytes // esz
        if buf is None or buf.numel() < n:
            # whole 2-MiB units: the caching allocator rounds a large request up to that and keeps the rest of the segment as a free block
            # -- the next 1-2 MB tensor anybody allocates (a long-lived workspace, say) lands in it and pins the whole multi-GiB segment
            # in torch's cache after this buffer is gone (seen as 89 GiB "reserved" after the 512^2 runner of bench.py was dropped: the
            # 128-spp leg then planned with 70 GiB less and cut its ray set into chunks)
            n = ((n * esz + (1 << 21) - 1) >> 21 << 21) // esz
            buf = None                   # release the old buffer BEFORE the larger one is allocated (peak = need, not have + need):
            setattr(self, attr, None)    # neither the attribute nor this local may keep it alive across empty_cache()
            torch.cuda.empty_cache()
            # zero-filled once (~20 ms per 100 GiB, only when a buffer grows): wavefronts past the last block read the sink block of the F
            # region and multiply their (zero) cotangents with what they find there -- the forward has always written it by then, but no
            # byte of a panel buffer may ever be an uninitialised NaN pattern
            buf = torch.zeros(n, dtype=dtype, device=self.device)
            setattr(self, attr, buf)
        return buf

    def _bufs_f(self, nblk_chunk):
        """F panels + ReLU masks for `nblk_chunk` 32-point blocks (+ 1 sink block for the wavefronts past the end)"""
        return (self._grow("_fpanels", (nblk_chunk + 1) * self.fwd_tiles * 2048),
                self._grow("_masks", (nblk_chunk + 1) * self.mask_u16 * 2, torch.int16))

    def _bufs_g(self, nblk_slab):
        return self._grow("_gpanels", (nblk_slab + 1) * self.grad_tiles * 2048)

    def _ring_setup(self):
        """buffers + pair tables of the role-specialised backward: (ctl, payload, partial, bias_partial, pb_tiles, pairs of the
        remaining weight-gradient launch, [(out_off, bias_off)] of the ring products)"""
        if not L.has_ring():
            raise RuntimeError("AVC_BWD_RING=1 needs libavc_ring.so (python -m avatarclip_amd.build --ring; AVC_LIB_NAME=libavc_ring.so): the "
                               "role-specialised backward is an experiment that is not part of libavc.so (include/avc_ring.h)")
        key = (self.RING_CPT, self.RING_SLOTS)
        if self._ring is not None and self._ring["key"] == key:
            return self._ring
        lay, lib = self.dl.lay, self.lib
        ht = self.spec.H // 32
        ntypes = lib.avc_bwd_ring_types(self.net)
        assert ntypes == self.spec.NMID
        ring_rows, pb, offs = [], [], []
        for m in range(ntypes):
            row = [i for i, pr in enumerate(lay.pairs) if pr[0] == lay.panel["ABM"] + m * ht and pr[6] == 1]
            assert len(row) == 1
            pr = lay.pairs[row[0]]
            assert pr[1] == ht and pr[3] == ht and pr[7] == 0 and pr[5] >= 0, "ring products are HT x HT with an F-region B operand and a bias"
            ring_rows.append(row[0]); pb.append(int(pr[2])); offs.append((int(pr[4]), int(pr[5])))
        rest = np.ascontiguousarray(np.delete(self._pairs_host, ring_rows, axis=0))
        nrow = 8 * self.RING_CPT
        dev = self.device
        self._ring = dict(
            key=key, ntypes=ntypes, ht=ht, offs=offs, rest=rest,
            ctl=torch.zeros(lib.avc_bwd_ring_ctl_bytes() // 4, dtype=torch.int32, device=dev),
            payload=torch.empty(lib.avc_bwd_ring_payload_bytes(self.net, ntypes, self.RING_SLOTS), dtype=torch.uint8, device=dev),
            partial=torch.zeros(ntypes, nrow, ht * ht * 1024, dtype=torch.float32, device=dev),
            bias=torch.zeros(ntypes, nrow, ht * 32, dtype=torch.float32, device=dev),
            pb=(ctypes.c_int * ntypes)(*pb),
            err=torch.zeros((), dtype=torch.int32, device=dev),
            grid=torch.cuda.get_device_properties(dev).multi_processor_count)
        return self._ring

    def _held_bytes(self):
        return sum(b.numel() * b.element_size() for b in (self._fpanels, self._gpanels, self._masks) if b is not None)

    def plan(self, R, S):
        """(rays per chunk, rays per slab): a chunk's F panels + masks and one slab's G panels fit the budget (plan_panels)"""
        key = (R, S, self.PANEL_BYTES_BUDGET, self.SLAB_BLOCKS)
        if getattr(self, "_plan_key", None) == key:
            chunk, slab = self._plan_val
            if chunk < R or (self._fpanels is not None and self._fpanels.numel() >= ((R * S + 31) // 32 + 1) * self.fwd_tiles * 2048):
                return self._plan_val  # (no driver query on the hot path: hipMemGetInfo synchronises with the device)
        # a ray set that the buffers at hand already cover in one chunk and one slab (the silhouette mode: a different, data-dependent
        # R every iteration, all far below the first allocation) needs no driver query either
        nb_all = (R * S + 31) // 32 + 1
        if (self._fpanels is not None and self._gpanels is not None and nb_all <= self.SLAB_BLOCKS
                and self._fpanels.numel() >= nb_all * self.fwd_tiles * 2048 and self._gpanels.numel() >= nb_all * self.grad_tiles * 2048
                and self._held_bytes() <= self.PANEL_BYTES_BUDGET):
            return R, R
        # 80 % of what is free once the current buffers are given back (they are released before larger ones are allocated)
        budget = max(min(self.PANEL_BYTES_BUDGET, (torch.cuda.mem_get_info(self.device)[0] + self._held_bytes()) * 8 // 10), 1 << 26)
        self._plan_key = key
        self._plan_val = plan_panels(R, S, budget, self.fwd_tiles * 2048 + self.mask_u16 * 2, self.grad_tiles * 2048,
                                     self.SLAB_BLOCKS, self.MIN_SLAB_BLOCKS)
        return self._plan_val

    def rays_per_chunk(self, R, S):
        return self.plan(R, S)[0]

    # ------------------------------------------------------------------ forward launches
    def sdf_rays(self, pk: Packed, rays_o, rays_d, z, sdf_out=None, slot=None, ld_out=0):
        R, S = z.shape
        if sdf_out is None:
            sdf_out = torch.empty(R, S, device=self.device, dtype=torch.float32)
        with Engine._Timed("avc_sdf_forward", R * S):
            L.check(self.lib.avc_sdf_forward(self.net, None, L.ptr(rays_o), L.ptr(rays_d), L.ptr(z), S, z.stride(0), R * S,
                                             L.ptr(pk.w_f16), L.ptr(pk.tab), self.dl.offsets, L.ptr(sdf_out),
                                             L.ptr(slot), ld_out, L.stream()), "avc_sdf_forward")
        return sdf_out

    def sdf_pts(self, pk: Packed, pts):
        pts = pts.contiguous().float()
        out = torch.empty(pts.shape[0], 1, device=self.device, dtype=torch.float32)
        L.check(self.lib.avc_sdf_forward(self.net, L.ptr(pts), None, None, None, 1, 1, pts.shape[0], L.ptr(pk.w_f16),
                                         L.ptr(pk.tab), self.dl.offsets, L.ptr(out), None, 0, L.stream()),
                "avc_sdf_forward")
        return out

    def upsample_step(self, rays_o, rays_d, z, sdf, m, inv_s):
        R, n = z.shape
        z_out = torch.empty(R, n + m, device=self.device, dtype=torch.float32)
        sdf_out = torch.empty(R, n + m, device=self.device, dtype=torch.float32)
        z_new = torch.empty(R, m, device=self.device, dtype=torch.float32)
        slot = torch.empty(R, m, device=self.device, dtype=torch.int32)
        # AVC_UPSAMPLE_GROUP=0: one wavefront per ray for every n (A/B partner and cross-check of the grouped kernels), read HERE per call
        lanes = 64 if os.environ.get("AVC_UPSAMPLE_GROUP", "1") == "0" else 0
        L.check(self.lib.avc_upsample_step_lanes(L.ptr(rays_o), L.ptr(rays_d), L.ptr(z), L.ptr(sdf), R, n, m, float(inv_s),
                                                 L.ptr(z_out), L.ptr(sdf_out), L.ptr(z_new), L.ptr(slot), lanes, L.stream()),
                "avc_upsample_step")
        return z_out, sdf_out, z_new, slot

    def points_fwd(self, pk: Packed, rays_o, rays_d, z, sample_dist):
        R, S = z.shape
        N = R * S
        sdf = torch.empty(R, S, device=self.device, dtype=torch.float32)
        nrm = torch.empty(R, S, 3, device=self.device, dtype=torch.float32)
        rgb = torch.empty(R, S, 6, device=self.device, dtype=torch.float32)
        if self._fwd_scratch is None:
            self._fwd_scratch = torch.empty(self.MAX_FWD_WAVES * self.fwd_scr_bytes, dtype=torch.uint8, device=self.device)
        with Engine._Timed("avc_render_points_fwd", N):
            L.check(self.lib.avc_render_points_fwd(self.net, None, L.ptr(rays_o), L.ptr(rays_d), L.ptr(z), S, z.stride(0),
                                                   float(sample_dist), N, L.ptr(pk.w_f16), L.ptr(pk.tab), self.dl.offsets,
                                                   L.ptr(sdf), L.ptr(nrm), L.ptr(rgb), self.MAX_FWD_WAVES, L.ptr(self._fwd_scratch),
                                                   L.stream()), "avc_render_points_fwd")
        return sdf, nrm, rgb

    def points_fwd_train(self, pk: Packed, rays_o, rays_d, z, sample_dist, r0=0, r1=None, out=None):
        """the differentiable forward of rays r0:r1: outputs + operand panels / masks of those rays' blocks (engine buffers)"""
        R, S = z.shape
        r1 = R if r1 is None else r1
        if out is None:
            out = (torch.empty(R, S, device=self.device, dtype=torch.float32), torch.empty(R, S, 3, device=self.device, dtype=torch.float32),
                   torch.empty(R, S, 6, device=self.device, dtype=torch.float32))
        sdf, nrm, rgb = out
        npts = (r1 - r0) * S
        panels, masks = self._bufs_f((npts + 31) // 32)
        esz = 4
        with Engine._Timed("avc_render_points_fwd_train", npts):
            L.check(self.lib.avc_render_points_fwd_train(
                self.net, None, rays_o.data_ptr() + r0 * 3 * esz, rays_d.data_ptr() + r0 * 3 * esz,
                z.data_ptr() + r0 * z.stride(0) * esz, S, z.stride(0), float(sample_dist), npts, L.ptr(pk.w_f16), L.ptr(pk.tab),
                self.dl.offsets, sdf.data_ptr() + r0 * S * esz, nrm.data_ptr() + r0 * S * 3 * esz, rgb.data_ptr() + r0 * S * 6 * esz,
                self.MAX_FWD_WAVES, L.ptr(panels), L.ptr(masks), L.stream()), "avc_render_points_fwd_train")
        return out

    def composite_fwd(self, sdf, nrm, rgb, z, rays_o, rays_d, inv_s, sample_dist, cos_anneal, bg, bg_mode):
        """-> color, extra, weights, cdf, mid_z, inside, eik, wstat [2,R] = (sum_i w_i, max_i w_i), nsum [R,3] = sum_i w_i n_i"""
        R, S = z.shape
        dev, f32 = self.device, torch.float32
        color = torch.empty(R, 3, device=dev, dtype=f32)
        extra = torch.empty(R, 3, device=dev, dtype=f32)
        weights = torch.empty(R, S, device=dev, dtype=f32)
        cdf = torch.empty(R, S, device=dev, dtype=f32)
        mid_z = torch.empty(R, S, device=dev, dtype=f32)
        inside = torch.empty(R, S, device=dev, dtype=f32)
        eik = torch.empty(R, 2, device=dev, dtype=f32)
        wstat = torch.empty(2, R, device=dev, dtype=f32)
        nsum = torch.empty(R, 3, device=dev, dtype=f32)
        L.check(self.lib.avc_composite_fwd(L.ptr(sdf), L.ptr(nrm), L.ptr(rgb), L.ptr(z), L.ptr(rays_o), L.ptr(rays_d), R, S,
                                           L.ptr(inv_s), float(sample_dist), float(cos_anneal), L.ptr(bg), bg_mode,
                                           L.ptr(color), L.ptr(extra), L.ptr(weights), L.ptr(cdf), L.ptr(mid_z),
                                           L.ptr(inside), L.ptr(eik), L.ptr(wstat), L.ptr(nsum), L.stream()), "avc_composite_fwd")
        return color, extra, weights, cdf, mid_z, inside, eik, wstat, nsum

    def composite_bwd(self, sdf, nrm, rgb, z, rays_o, rays_d, inv_s, sample_dist, cos_anneal, bg, bg_mode, d_color,
                      d_extra, d_w, d_n_up, eik_scale, d_wsum=None, d_nsum=None):
        R, S = z.shape
        dev, f32 = self.device, torch.float32
        d_sdf = torch.empty(R, S, device=dev, dtype=f32)
        d_n = torch.empty(R, S, 3, device=dev, dtype=f32)
        d_rgb = torch.empty(R, S, 6, device=dev, dtype=f32)
        d_inv = torch.empty(R, device=dev, dtype=f32)
        L.check(self.lib.avc_composite_bwd(L.ptr(sdf), L.ptr(nrm), L.ptr(rgb), L.ptr(z), L.ptr(rays_o), L.ptr(rays_d), R, S,
                                           L.ptr(inv_s), float(sample_dist), float(cos_anneal), L.ptr(bg), bg_mode,
                                           L.ptr(d_color), L.ptr(d_extra), L.ptr(d_w), L.ptr(d_n_up), L.ptr(d_wsum), L.ptr(d_nsum),
                                           L.ptr(eik_scale), L.ptr(d_sdf), L.ptr(d_n), L.ptr(d_rgb), L.ptr(d_inv), L.stream()),
                "avc_composite_bwd")
        return d_sdf, d_n, d_rgb, d_inv

    # ------------------------------------------------------------------ backward of the point MLP
    def points_bwd(self, pk: Packed, rays_o, rays_d, z, sample_dist, d_sdf, d_n, d_rgb, rgb, panels_valid=False):
        """returns the flat dense gradient [nparam] (fp32).  `rgb` = the forward's colours.  panels_valid: the F region still
        holds the forward-type operand panels of exactly this ray set (one chunk, nothing rendered since) -- otherwise the
        training forward is re-run per chunk.  The backward itself walks every chunk in slabs: backward kernel (writes the
        slab's gradient-type panels) + weight-gradient kernel (contracts both regions over the slab)."""
        lay = self.dl.lay
        R, S = z.shape
        rays_per_chunk, rays_per_slab = self.plan(R, S)
        if rays_per_chunk < R:
            panels_valid = False
        rg = self._ring_setup() if self.RING else None
        # the split sums and the way back to the dense vector: one launch per slab + one at the end (avc_weight_grad_reduce / _unpack);
        # the ring experiment patches columns of the sums, so it keeps the torch statement of the same arithmetic
        fused_tail = rg is None and self.FUSED_WG_TAIL and lay.gout_size % 4 == 0 and lay.gbias_size % 4 == 0
        if fused_tail:
            gacc = torch.empty(lay.gout_size + lay.gbias_size, device=self.device, dtype=torch.float32)
            nslab = 0
        else:
            gout = torch.zeros(lay.gout_size, device=self.device, dtype=torch.float32)
            gbias = torch.zeros(max(lay.gbias_size, 1), device=self.device, dtype=torch.float32)
        if self._partials is None:
            self._partials = torch.empty(256, lay.gout_size, device=self.device, dtype=torch.float32)
            self._bpartials = torch.empty(256, max(lay.gbias_size, 1), device=self.device, dtype=torch.float32)
        st = L.stream()
        esz = 4
        scratch_out = None
        if rg is not None:
            rg["err"].zero_()
        pairs_host = rg["rest"] if rg is not None else self._pairs_host
        cs_rows = int(self.lib.avc_bwd_colsum_rows(min(R, rays_per_slab) * S, self.MAX_BWD_WAVES)) if rg is None else 8 * rg["grid"]
        if self._colsum is None or self._colsum.shape[0] < cs_rows:
            self._colsum = torch.zeros(cs_rows, lay.cs_size, device=self.device, dtype=torch.float32)
        cs_total = None               # sum over wavefronts and slabs of the column sums
        for c0 in range(0, R, rays_per_chunk):
            c1 = min(R, c0 + rays_per_chunk)
            if not panels_valid:
                if scratch_out is None:   # outputs of the re-run are not needed (identical to the first pass)
                    scratch_out = (torch.empty(R, S, device=self.device), torch.empty(R, S, 3, device=self.device),
                                   torch.empty(R, S, 6, device=self.device))
                self.points_fwd_train(pk, rays_o, rays_d, z, sample_dist, c0, c1, out=scratch_out)
            fpanels, masks = self._bufs_f(((c1 - c0) * S + 31) // 32)
            for s0 in range(c0, c1, rays_per_slab):
                s1 = min(c1, s0 + rays_per_slab)
                npts = (s1 - s0) * S
                nblk = (npts + 31) // 32
                assert ((s0 - c0) * S) % 32 == 0, "slabs start on a 32-point block of their chunk"
                fblk0 = (s0 - c0) * S // 32          # the slab's first block in the chunk's F region
                fptr = fpanels.data_ptr() + fblk0 * self.fwd_tiles * 2048
                mptr = masks.data_ptr() + fblk0 * self.mask_u16 * 2
                gpanels = self._bufs_g(nblk)
                bwd_args = (self.net, None, rays_o.data_ptr() + s0 * 3 * esz, rays_d.data_ptr() + s0 * 3 * esz,
                            z.data_ptr() + s0 * z.stride(0) * esz, S, z.stride(0), float(sample_dist), npts, L.ptr(pk.w_bf16), L.ptr(pk.tab),
                            self.dl.offsets, d_sdf.data_ptr() + s0 * S * esz, d_n.data_ptr() + s0 * S * 3 * esz,
                            d_rgb.data_ptr() + s0 * S * 6 * esz, rgb.data_ptr() + s0 * S * 6 * esz, fptr, L.ptr(gpanels), mptr, L.ptr(self._colsum))
                if rg is None:
                    rows = int(self.lib.avc_bwd_colsum_rows(npts, self.MAX_BWD_WAVES))     # every row of the launch's grid is written
                    with Engine._Timed("avc_render_points_bwd", npts):
                        L.check(self.lib.avc_render_points_bwd(*bwd_args, self.MAX_BWD_WAVES, st), "avc_render_points_bwd")
                else:
                    rows = 8 * rg["grid"]
                    self._colsum[:rows].zero_()          # consumer workgroups write no row
                    rg["partial"].zero_()
                    rg["bias"].zero_()
                    with Engine._Timed("avc_render_points_bwd_ring", npts):
                        L.check(self.lib.avc_render_points_bwd_ring(
                            *bwd_args, L.ptr(rg["ctl"]), L.ptr(rg["payload"]), L.ptr(rg["partial"]), L.ptr(rg["bias"]), rg["pb"],
                            rg["ntypes"], self.RING_CPT, self.RING_SLOTS, rg["grid"], st), "avc_render_points_bwd_ring")
                    rg["err"] += rg["ctl"][64]
                    if self.RING_CHECK:
                        torch.cuda.synchronize()
                        self.ring_stats = rg["ctl"][96:112].cpu().numpy().view(np.uint64).copy()
                        if int(rg["ctl"][64]) != 0:
                            raise RuntimeError("avc_render_points_bwd_ring: %d spin time-out(s), first site %d, counters %s"
                                               % (int(rg["ctl"][64]), int(rg["ctl"][65]), self.ring_stats))
                cs_slab = self._colsum[:rows].sum(0)      # (torch's reduction: a fixed tree, deterministic)
                cs_total = cs_slab if cs_total is None else cs_total + cs_slab
                ns = max(1, min(self.WG_MAX_SPLITS, nblk // self.WG_BLOCKS_PER_SPLIT, nblk))
                with Engine._Timed("avc_weight_grad(all pairs)", npts):
                    L.check(self.lib.avc_weight_grad_all(fptr, self.fwd_tiles, L.ptr(gpanels), self.grad_tiles, len(pairs_host),
                                                         pairs_host.ctypes.data, nblk, L.ptr(self._partials),
                                                         L.ptr(self._bpartials), ns, self._partials.stride(0),
                                                         self._bpartials.stride(0), st), "avc_weight_grad_all")
                    if fused_tail:
                        L.check(self.lib.avc_weight_grad_reduce(L.ptr(self._partials), L.ptr(self._bpartials), ns, self._partials.stride(0),
                                                                self._bpartials.stride(0), lay.gout_size, lay.gbias_size, L.ptr(gacc),
                                                                int(nslab > 0), st), "avc_weight_grad_reduce")
                        nslab += 1
                        continue
                    po, pb_ = self._partials[:ns].sum(0), self._bpartials[:ns].sum(0)
                    if rg is not None:   # the ring products' columns of the split buffers are not written by this launch (stale): take the consumers' sums
                        for m, (o_off, b_off) in enumerate(rg["offs"]):
                            po[o_off:o_off + rg["partial"].shape[2]] = rg["partial"][m].sum(0)
                            pb_[b_off:b_off + rg["bias"].shape[2]] = rg["bias"][m].sum(0)
                    gout += po
                    gbias += pb_
        self._panel_owner = None
        if fused_tail:
            grad = torch.empty(lay.nparam, device=self.device, dtype=torch.float32)
            if nslab == 0:
                grad.zero_()
            else:
                L.check(self.lib.avc_weight_grad_unpack(L.ptr(gacc), L.ptr(self.dl.csr_off), L.ptr(self.dl.csr_src), L.ptr(self.dl.csr_scale),
                                                        lay.nparam, L.ptr(grad), st), "avc_weight_grad_unpack")
        else:
            grad = torch.zeros(lay.nparam, device=self.device, dtype=torch.float32)
            grad.index_add_(0, self.dl.un_tgt, gout[self.dl.un_src] * self.dl.un_scale)
            if lay.gbias_size:
                grad.index_add_(0, self.dl.ub_tgt, gbias[self.dl.ub_src])
        if cs_total is not None:      # second-order term of row 0 of the last SDF layer: sum_points [gbar_hs ; gbar_h0] / sqrt2 (unique targets)
            grad.index_add_(0, self.dl.cs_tgt, cs_total[self.dl.cs_src] * self.dl.cs_scale)
        # (d loss / d (sdf bias) = sum of d_sdf over all points cancels heavily -- the eikonal term pulls both ways.  The tile carries d_sdf
        # as hi + lo bf16 (packing.py / avc_bwd_body.h) and gets it to 0.03 %; the default takes the sum from the fp32 cotangent instead,
        # AVC_SDF_BIAS_FP32=0 leaves the products' value.  sdf = lin_out / scale (fields.py:93) with scale == 1: packing.spec_from_conf
        # refuses every other value, so no division is needed here.)
        if self.SDF_BIAS_FP32:
            grad[self._sdf_bias0] = d_sdf.sum()
        if rg is not None:   # a hand-off that timed out leaves the products incomplete: make that loud without a host round trip
            grad = torch.where(rg["err"] != 0, torch.full_like(grad, float("nan")), grad)
        return grad


class RenderCoreFn(torch.autograd.Function):
    """render_core (renderer.py:195-300) on fixed z_vals: fused point MLP + compositing, differentiable wrt the
    flat dense parameter vector and inv_s."""

    @staticmethod
    def forward(ctx, flatP, inv_s, eng, rays_o, rays_d, z_vals, sample_dist, cos_anneal, bg, bg_mode):
        pk = eng.pack(flatP)
        inv_s_d = inv_s.detach().float().contiguous()
        R, S = z_vals.shape
        needs_grad = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])   # (grad mode itself is off inside forward)
        ctx.panel_token = None
        if needs_grad and eng.plan(R, S)[0] >= R:
            # the whole ray set's operand panels fit: the forward runs once and leaves them for the backward pass
            sdf, nrm, rgb = eng.points_fwd_train(pk, rays_o, rays_d, z_vals, sample_dist)
            ctx.panel_token = eng._panel_owner = object()
        else:
            sdf, nrm, rgb = eng.points_fwd(pk, rays_o, rays_d, z_vals, sample_dist)
        color, extra, weights, cdf, mid_z, inside, eik, wstat, nsum = eng.composite_fwd(
            sdf, nrm, rgb, z_vals, rays_o, rays_d, inv_s_d, sample_dist, cos_anneal, bg, bg_mode)
        wsum, wmax = wstat[0].reshape(R, 1), wstat[1].reshape(R, 1)       # (planar: views, no copies)
        eo = torch.empty(2, device=eik.device, dtype=torch.float32)        # renderer.py:283-285 in one launch: (gradient_error, its denominator)
        L.check(eng.lib.avc_colsum(L.ptr(eik), R, 2, 1, L.ptr(eo), L.ptr(eng.colsum_scratch()), L.stream()), "avc_colsum")
        gerr, eik_den = eo[0], eo[1]
        ctx.eng, ctx.pk = eng, pk
        ctx.consts = (sample_dist, cos_anneal, bg_mode)
        ctx.save_for_backward(rays_o, rays_d, z_vals, sdf, nrm, rgb, inv_s_d, eik_den, bg if bg is not None else inv_s_d)
        ctx.has_bg = bg is not None
        ctx.mark_non_differentiable(cdf, mid_z, inside, sdf, wmax)
        # outputs the loss does not touch (the per-sample weights and normals when the shading takes wsum / nsum) hand `None` to the
        # backward instead of zero tensors: [R,S] + [R,S,3] fp32 of fills and of reads in composite_bwd per step otherwise
        ctx.set_materialize_grads(False)
        # wsum = sum_i w_i, wmax = max_i w_i, nsum = sum_i w_i n_i: the per-ray reductions render() and the shading of main.py:428
        # take of the weights, out of the compositing kernel's own registers (and differentiable through its reverse scan)
        return color, extra, weights, nrm, gerr, cdf, mid_z, inside, sdf, wsum, wmax, nsum

    @staticmethod
    def backward(ctx, d_color, d_extra, d_weights, d_nrm, d_gerr, d_cdf=None, d_midz=None, d_inside=None, d_sdf_out=None,
                 d_wsum=None, d_wmax=None, d_nsum=None):
        eng, pk = ctx.eng, ctx.pk
        rays_o, rays_d, z_vals, sdf, nrm, rgb, inv_s_d, eik_den, bg = ctx.saved_tensors
        if not ctx.has_bg:
            bg = None
        sample_dist, cos_anneal, bg_mode = ctx.consts
        R, S = z_vals.shape
        zeros = lambda *s: torch.zeros(*s, device=z_vals.device, dtype=torch.float32)
        d_color = d_color.contiguous().float() if d_color is not None else zeros(R, 3)
        d_extra = d_extra.contiguous().float() if d_extra is not None else zeros(R, 3)
        d_weights = d_weights.contiguous().float() if d_weights is not None else None
        d_nrm = d_nrm.contiguous().float() if d_nrm is not None else None
        d_wsum = d_wsum.contiguous().float().reshape(R) if d_wsum is not None else None
        d_nsum = d_nsum.contiguous().float() if d_nsum is not None else None
        d_gerr = d_gerr if d_gerr is not None else zeros(())
        eik_scale = (d_gerr.float() / eik_den).reshape(1).contiguous()
        d_sdf, d_n, d_rgb, d_inv = eng.composite_bwd(sdf, nrm, rgb, z_vals, rays_o, rays_d, inv_s_d, sample_dist,
                                                     cos_anneal, bg, bg_mode, d_color, d_extra, d_weights, d_nrm, eik_scale,
                                                     d_wsum, d_nsum)
        valid = ctx.panel_token is not None and eng._panel_owner is ctx.panel_token
        grad = eng.points_bwd(pk, rays_o, rays_d, z_vals, sample_dist, d_sdf, d_n, d_rgb, rgb, panels_valid=valid)
        d_inv_s = torch.empty(1, device=d_inv.device, dtype=torch.float32)
        L.check(eng.lib.avc_colsum(L.ptr(d_inv), R, 1, 0, L.ptr(d_inv_s), L.ptr(eng.colsum_scratch()), L.stream()), "avc_colsum")
        return grad, d_inv_s, None, None, None, None, None, None, None, None
