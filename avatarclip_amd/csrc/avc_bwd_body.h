// The three sweeps of the point-MLP backward for one workgroup iteration (8 wavefronts x 32 points), shared by the plain backward
// kernel (avc_mlp_bwd.hip) and the role-specialised one (avc_bwd_ring.hip).  Mathematics: SURVEY.md A.1/A.2 (fields.py:96-107
// double backward; autograd at main.py:537), proven against torch.autograd in tests/test_analytic.py (oracle/analytic.py).
//
// NOTHING of the forward pass is recomputed: the forward kernel (avc_render_points_fwd_train) left h_l, g_a,l, the ReLU masks and
// the colours in the block's operand panels (csrc/avc_mlp.h: PanelLayout, F region).  The sweeps -- colour backward (phase D),
// second-order sweep (i) (phase E), reverse sweep (ii) (phase F) -- run on bf16 operands with fp32 accumulation, read sigma's
// argument / g_a / gbar_h back from the panels as fragments (no transposition) and write the gradient-type operands of the
// weight-gradient products (gbar_h, abar, delta, ybar) to the G region of the current slab.  The second-order term abar' is not
// stored: the reverse sweep rebuilds it from the gbar_h, g_a and h tiles (abar' = gbar_h g_a beta (1-s)/s).
//
// `Ring` policy: NoRing = every gradient-type tile goes to the G region.  A ring policy (avc_bwd_ring.hip) takes the abar tiles of
// the middle SDF layers instead -- they are pure hand-off tiles (written here, read only by the weight-gradient product
// abar_m (x) h_in) -- and passes them to a consumer workgroup of the same XCD through an L2-resident ring.
#pragma once
#include "avc_mlp.h"
#ifndef BWD_G
#define BWD_G 4   // tiles per staged group (LDS = 2 * G * 16 KiB + table: one 8-wave workgroup per CU)
#endif
#ifndef BWD_WPB
#define BWD_WPB 8   // wavefronts per workgroup: every staged weight tile is shared by 256 points (LDS-DMA fill rate is the scarce resource)
#endif
// cache policy of the tile loads: NT = streamed past the caches.  Measured per 4 Mi points (profiles/r03_ab_kernels.txt):
//   AVC_BWD_E_NT    the h tiles the second-order sweep reads (they are read AGAIN by the reverse sweep ~6 layer steps later):
//                   normal policy 10.21 ms vs nt 10.43 -> 0
//   AVC_BWD_RR_NT   the tiles this kernel wrote itself (normal-policy stores) and reads back (gbar_h, ybar[1:]): nt loads 10.43 vs
//                   normal 10.59 (both switches off: 11.12) -> 1
#ifndef AVC_BWD_E_NT
#define AVC_BWD_E_NT 0
#endif
#ifndef AVC_BWD_RR_NT
#define AVC_BWD_RR_NT 1
#endif
// Round 5 (the kernel's self-re-reads, 70 tiles per block).  Timing ablations -- results are garbage --
//   AVC_ABL_BWD_NOEH    the second-order sweep does not load its h tiles (an opaque constant instead): what ANY scheme that removes the
//                       first of the two h reads (31 tiles) could gain at most
//   AVC_ABL_BWD_NORR    the reverse sweep does not re-read the tiles this kernel wrote itself (gbar_h 31, ybar[1:] 8 tiles)
//   AVC_ABL_BWD_RECOMP  NOEH + the price of recomputing h inside the second-order sweep from the staged W_l fragments, priced LOW: a
//                       second MFMA chain per tile on the same A fragments (one LDS read feeds two MFMAs) + the 16 softplus per lane
//                       and tile, but NOT the second input array (64 VGPRs) nor the f16 weight set a real version needs
// and one real variant (parity-valid):
//   AVC_BWD_KEEP_GBS    turn-around residency: gbar_hs (the last tiles the second-order sweep produces, the first the reverse sweep
//                       consumes) stays in registers across the turn instead of being re-read (7 tiles); = 2: h_s as well (14 tiles)
#ifndef AVC_BWD_KEEP_GBS
#define AVC_BWD_KEEP_GBS 1
#endif
template <typename V>
__device__ __forceinline__ FragPair<V> abl_const_pair() {
  FragPair<V> d;
#pragma unroll
  for (int j = 0; j < 8; ++j) { d.a0[j] = (typename MF<V>::S)0.75f; d.a1[j] = (typename MF<V>::S)1.25f; }
  asm volatile("" : "+v"(d.a0), "+v"(d.a1));
  return d;
}

template <typename P> __device__ __forceinline__ P launder(P p) {
  asm volatile("" : "+s"(p));
  return p;
}
template <typename V>
__device__ __forceinline__ V zero_frag() {
  V z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (typename MF<V>::S)0.f;
  return z;
}
// abar' = gbar_a g_h sp''(h) with gbar_a = gbar_h / s and g_h sp'' = g_a beta (1 - s): everything on the right is a tile
// of the panels.  s -> 0 makes both gbar_h and g_a vanish; the guard keeps 0/0 out.
__device__ __forceinline__ float second_term(float gbar_h, float g_a, float s) {
  const float r = s > 1e-30f ? __builtin_amdgcn_rcpf(s) : 0.f;
  return gbar_h * g_a * (AVC_BETA * (1.f - s) * r);
}
struct PF3 { h8 h0, h1; b8 b0, b1; h8 g0, g1; };   // h, gbar_h, g_a tiles of one layer, loaded one MFMA chain ahead of their epilogue

struct BwdArgs {
  PointSrc ps;
  long npts;
  const b8* Wb0;
  const float* T0;
  const float* d_sdf;
  const float* d_normal;
  const float* d_rgb;
  const float* rgb_fwd;
  const char* fpanels;
  char* gpanels;
  const unsigned short* masks;
};

// The inputs the FIRST MFMA chain and the first epilogues of a block wait for: delta_o (from d_rgb and the forward's colours) and the
// ReLU masks.  AVC_BWD_PIPE_IN=1: the persistent kernel requests them for its NEXT block under the last layer of the current one
// (loop-carried, 20 VGPRs) instead of at the top of the block, where all eight wavefronts sit out one exposed HBM round trip.
#ifndef AVC_BWD_PIPE_IN
#define AVC_BWD_PIPE_IN 1   // (profiles/r05_ab_kernels.txt: 9.81 -> 9.48 ms per 4 Mi points)
#endif
template <class N> struct BlkIn { b8 dof; unsigned m1[N::HT], m2[N::HT]; };
template <class N>
__device__ __forceinline__ void load_blk_in(const BwdArgs& a, long blk, long nblk, int lane, BlkIn<N>& bi) {
  typedef PanelLayout<N> L;
  const int h = lane >> 5, p = lane & 31;
  long i = blk * 32 + p;
  const float vmask = i < a.npts ? 1.f : 0.f;
  if (i >= a.npts) i = a.npts - 1;
  // delta_o = d_rgb * rgb (1 - rgb) with the colours of the forward pass; half 0: outputs 0..3, half 1: outputs 4,5
  bi.dof = zero_frag<b8>();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ch = h ? 4 + r : r;
    const float c = (ch < 6) ? a.rgb_fwd[6 * i + (ch < 6 ? ch : 0)] : 0.f;
    const float dr = (ch < 6) ? a.d_rgb[6 * i + (ch < 6 ? ch : 0)] * vmask : 0.f;
    bi.dof[r] = (__bf16)(dr * c * (1.f - c));
  }
  // ReLU masks of r1 / r2 (16 bits per tile and lane, written by the forward kernel: accumulator register r at bit relu_mask_bit(r))
  const AVC_GLOBAL unsigned short* mk = as_global(a.masks) + (blk < nblk ? blk : nblk) * (long)L::MASK_U16 + lane;
#pragma unroll
  for (int t = 0; t < N::HT; ++t) {
    bi.m1[t] = mk[t * 64];
    bi.m2[t] = (N::NCMID == 1) ? mk[(N::HT + t) * 64] : 0u;
  }
}

struct NoRing {
  static constexpr bool on = false;
  template <class N> __device__ __forceinline__ void handoff(int, const b8 (&)[N::HK], long, int, int) const {}
};

// one workgroup iteration: blocks blk0 .. blk0 + BWD_WPB - 1 (wave wv owns block blk0 + wv)
// PIPE: `bi` holds this block's inputs on entry and the inputs of block blk0_next + wv on return (plain backward kernel only: the
// role-specialised kernel claims its blocks dynamically and does not know the next one)
template <class N, bool PIPE = false, class R>
__device__ __forceinline__ void bwd_sweeps(StageT<BWD_G>& sg, const BwdArgs& a, lds_tab_t Tl, long blk0, long nblk, int lane0,
                                           int wv, R& ring, BlkIn<N>* bip = nullptr, long blk0_next = 0) {
  typedef PanelLayout<N> L;
  constexpr AvcOffsets o = Off<N>::value;
  const PointSrc& ps = a.ps;
  const long npts = a.npts;
  const b8* Wb = launder(a.Wb0);
  // per-iteration copies of the loop invariants: otherwise everything derived from them is hoisted out of the loop and spilled
  lds_tab_t T = Tl;
  asm volatile("" : "+s"(T));
  int lane = lane0;
  asm volatile("" : "+v"(lane));
  const int h = lane >> 5, p = lane & 31;
  sg.lane = lane;
  const long blk = blk0 + wv;
  // wavefronts past the end walk the tile sequence for the barriers and write to the sink block (index nblk) of the G region
  // (what they read from block nblk of the F region -- the next slab's first block or the forward's sink -- is discarded)
  const long bsel = blk < nblk ? blk : nblk;
  const PanelPtr ftiles = panel_ptr(const_cast<char*>(a.fpanels) + bsel * (long)L::P_TILES * 2048, lane);   // forward-type operands: read only
  const PanelPtr tiles = panel_ptr(a.gpanels + bsel * (long)L::G_TILES * 2048, lane);                       // gradient-type operands of this slab
  long i = blk * 32 + p;
  const bool valid = i < npts;
  if (!valid) i = npts - 1;
  const float vmask = valid ? 1.f : 0.f;
  float x[3];
  fetch_point(ps, i, x);
  // ------------------------------------------------------------------ phase D: colour backward (bf16)
  // delta_o = d_rgb * rgb (1 - rgb) with the colours of the forward pass; half 0: outputs 0..3, half 1: outputs 4,5
  float nbar[3];
  {
    BlkIn<N> bi_local;
    if constexpr (!PIPE) load_blk_in<N>(a, blk, nblk, lane, bi_local);
    BlkIn<N>& bi = PIPE ? *bip : bi_local;
    b8 dof[1];
    dof[0] = bi.dof;
    tile_store<false>(tiles, L::G_DO, dof[0], zero_frag<b8>());
    unsigned m1[N::HT], m2[N::HT];
#pragma unroll
    for (int t = 0; t < N::HT; ++t) { m1[t] = bi.m1[t]; m2[t] = bi.m2[t]; }
#define AVC_RELU_BWD(OUT, MSK, PT)                                                                         \
  AVC_EPI(const unsigned bits = MSK[t];                                                                      \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                    \
            OUT[2 * t][j] = (__bf16)(((bits >> relu_mask_bit(j)) & 1u) ? acc[j] : 0.f);                      \
            OUT[2 * t + 1][j] = (__bf16)(((bits >> relu_mask_bit(8 + j)) & 1u) ? acc[8 + j] : 0.f); }        \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                  \
          tile_store<false>(tiles, (PT) + t, OUT[2 * t], OUT[2 * t + 1]);)
    b8 dl[N::HK];
    b8 d1[N::HK];
    if constexpr (N::NCMID == 1) {
      layer_s<b8, 1, N::HT>(sg, Wb, o.v[OFF_CHT], nxt<N, OFF_CM0T>(sg, Wb, o), dof, AVC_RELU_BWD(dl, m2, L::G_D2));
      layer_s<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_CM0T], nxt<N, OFF_C0T>(sg, Wb, o), dl, AVC_RELU_BWD(d1, m1, L::G_D1));
    } else {
      layer_s<b8, 1, N::HT>(sg, Wb, o.v[OFF_CHT], nxt<N, OFF_C0T>(sg, Wb, o), dof, AVC_RELU_BWD(d1, m1, L::G_D1));
    }
    // d r0 = C0^T delta1: HT feature tiles (ybar[1:], kept for the reverse sweep), then the [x,n] tile (rows 3,4,5 = d n)
    float dn_acc[3] = {0.f, 0.f, 0.f};
    layer_s<b8, N::HK, N::HT + 1>(sg, Wb, o.v[OFF_C0T], nxt<N, OFF_W0G>(sg, Wb, o), d1, AVC_EPI(
      if (t < N::HT) {
        b8 f0, f1;
        _Pragma("unroll") for (int j = 0; j < 8; ++j) { f0[j] = (__bf16)acc[j]; f1[j] = (__bf16)acc[8 + j]; }
        pin2(f0, f1);
        tile_store<true>(tiles, L::G_DFEAT + (t < N::HT ? t : 0), f0, f1);
      } else {
        dn_acc[0] = acc[3]; dn_acc[1] = acc[0]; dn_acc[2] = acc[1];
      }
    ));
    {
      // row 3 -> (h0,r3), row 4 -> (h1,r0), row 5 -> (h1,r1)
      const float a3 = dn_acc[0], a0 = dn_acc[1], a1 = dn_acc[2];
      const float o3 = __shfl_xor(a3, 32), o0 = __shfl_xor(a0, 32), o1 = __shfl_xor(a1, 32);
      nbar[0] = a.d_normal[3 * i + 0] * vmask + (h ? o3 : a3);
      nbar[1] = a.d_normal[3 * i + 1] * vmask + (h ? a0 : o0);
      nbar[2] = a.d_normal[3 * i + 2] * vmask + (h ? a1 : o1);
    }
  }
  const float dsdf = a.d_sdf[i] * vmask;
  const float dsdfS = dsdf * AVC_S;   // OFF_WL0_ACC holds W_last[0,:]/(S sqrt2): undo S for the gradient use
  {   // operand tiles with one live feature (slot (half 0, j = 0) = feature 0; d_sdf: two): d_sdf and the constant 1 (row 0 of the last layer)
    b8 fs = zero_frag<b8>(), fo = zero_frag<b8>();
    if (h == 0) {   // d_sdf split hi + lo over two slots (both map to row 0, packing.py): 16 bits of mantissa for the one cotangent whose sums cancel heavily
      fs[0] = (__bf16)dsdf;
      fs[1] = (__bf16)(dsdf - (float)fs[0]);
      fo[0] = (__bf16)vmask;
    }
    tile_store<false>(tiles, L::G_SDF, fs, zero_frag<b8>());
    tile_store<false>(tiles, L::G_ONE, fo, zero_frag<b8>());
  }
  // ------------------------------------------------------------------ phase E: second-order sweep (i) (bf16)
  b8 gbs[N::SK];        // gbar_hs: with AVC_BWD_KEEP_GBS it stays in registers for the first layer of the reverse sweep
  b8 dfeat[N::HK];      // ybar[1:]: the input of the reverse sweep (written by phase D, read back here)
  auto load_dfeat = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < N::HT; ++t) {
#ifdef AVC_ABL_BWD_NORR
      const FragPair<b8> d = abl_const_pair<b8>();
#else
      const FragPair<b8> d = tile_load<(AVC_BWD_RR_NT != 0), b8>(tiles, L::G_DFEAT + t);
#endif
      dfeat[2 * t] = d.a0;
      dfeat[2 * t + 1] = d.a1;
    }
  };
#if AVC_BWD_KEEP_GBS >= 2
  h8 hs_keep[N::SK];    // ... and so does h_s
#endif
  {
    b8 gb0[3];
    {
      PE pe4;
      pe_compute(x, h, pe4);
#pragma unroll
      for (int q = 0; q < 24; ++q) gb0[q >> 3][q & 7] = (__bf16)(pe4.d[q] * nbar[q % 3]);
    }
    tile_store<false>(tiles, L::G_GB0, gb0[0], gb0[1]);
    tile_store<false>(tiles, L::G_GB0 + 1, gb0[2], zero_frag<b8>());
    // gbar_a = W gbar_h(in); gbar_h(out) = gbar_a * sigma(h_out)
#if defined(AVC_ABL_BWD_NOEH) || defined(AVC_ABL_BWD_RECOMP)
#define AVC_E_LOADH(PH) abl_const_pair<h8>()
#else
#define AVC_E_LOADH(PH) tile_load<(AVC_BWD_E_NT != 0), h8>(ftiles, (PH) + t)
#endif
#define AVC_SECOND_(OUT, PH, PT, KEEPH)                                                                      \
  AVC_PRE(return AVC_E_LOADH(PH);),                                                                          \
  AVC_EPID(FragPair<h8>, _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                     \
            OUT[2 * t][j] = (__bf16)(acc[j] * sig_from_h((float)d.a0[j]));                                   \
            OUT[2 * t + 1][j] = (__bf16)(acc[8 + j] * sig_from_h((float)d.a1[j])); }                         \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                  \
          KEEPH                                                                                              \
          tile_store<true>(tiles, (PT) + t, OUT[2 * t], OUT[2 * t + 1]);)
#define AVC_SECOND(OUT, PH, PT) AVC_SECOND_(OUT, PH, PT, )
#if AVC_BWD_KEEP_GBS >= 2
#define AVC_SECOND_S(OUT, PH, PT) AVC_SECOND_(OUT, PH, PT, hs_keep[2 * t] = d.a0; hs_keep[2 * t + 1] = d.a1;)
#else
#define AVC_SECOND_S(OUT, PH, PT) AVC_SECOND_(OUT, PH, PT, )
#endif
    b8 gb1[N::HK];
    layer_sqd<b8, 3, N::HT>(sg, Wb, o.v[OFF_W0G], nxt<N, OFF_WM0>(sg, Wb, o), gb0, AVC_SECOND(gb1, L::P_H1, L::G_GBH1));
    b8 gbm[N::HK];
    if constexpr (N::NMID == 2) {
      layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wb, o), gb1, AVC_SECOND(gbm, L::P_HM, L::G_GBHM));
      b8 gbm1[N::HK];
      layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wb, o), gbm,
                                  AVC_SECOND(gbm1, L::P_HM + N::HT, L::G_GBHM + N::HT));
      layer_sqd<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WS], nxt<N, OFF_WLT>(sg, Wb, o), gbm1, AVC_SECOND_S(gbs, L::P_HS, L::G_GBHS));
    } else {
      layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wb, o), gb1, AVC_SECOND(gbm, L::P_HM, L::G_GBHM));
      layer_sqd<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WS], nxt<N, OFF_WLT>(sg, Wb, o), gbm, AVC_SECOND_S(gbs, L::P_HS, L::G_GBHS));
    }
  }
  // ------------------------------------------------------------------ phase F: reverse sweep (ii) (bf16)
  {
    b8 as_[N::SK];
    load_dfeat();
#ifdef AVC_ABL_BWD_NORR
#define AVC_F_LOADB(PB) abl_const_pair<b8>()
#else
#define AVC_F_LOADB(PB) tile_load<(AVC_BWD_RR_NT != 0), b8>(tiles, (PB) + t)
#endif
#define AVC_LOAD3(PH, PB, PG)                                                                               \
  AVC_PRE(PF3 d; { const FragPair<h8> a_ = tile_load<true, h8>(ftiles, (PH) + t); d.h0 = a_.a0; d.h1 = a_.a1; } \
          { const FragPair<b8> a_ = AVC_F_LOADB(PB); d.b0 = a_.a0; d.b1 = a_.a1; } \
          { const FragPair<h8> a_ = tile_load<true, h8>(ftiles, (PG) + t); d.g0 = a_.a0; d.g1 = a_.a1; } return d;)
  // the same for the first layer of the reverse sweep: gbar_hs (and h_s) straight from the registers of the second-order sweep when kept
#if AVC_BWD_KEEP_GBS >= 2
#define AVC_LOAD3_S(PH, PB, PG)                                                                             \
  AVC_PRE(PF3 d; d.h0 = hs_keep[2 * t]; d.h1 = hs_keep[2 * t + 1]; d.b0 = gbs[2 * t]; d.b1 = gbs[2 * t + 1];  \
          { const FragPair<h8> a_ = tile_load<true, h8>(ftiles, (PG) + t); d.g0 = a_.a0; d.g1 = a_.a1; } return d;)
#elif AVC_BWD_KEEP_GBS == 1
#define AVC_LOAD3_S(PH, PB, PG)                                                                             \
  AVC_PRE(PF3 d; { const FragPair<h8> a_ = tile_load<true, h8>(ftiles, (PH) + t); d.h0 = a_.a0; d.h1 = a_.a1; } \
          d.b0 = gbs[2 * t]; d.b1 = gbs[2 * t + 1];                                                          \
          { const FragPair<h8> a_ = tile_load<true, h8>(ftiles, (PG) + t); d.g0 = a_.a0; d.g1 = a_.a1; } return d;)
#else
#define AVC_LOAD3_S(PH, PB, PG) AVC_LOAD3(PH, PB, PG)
#endif
    // ubar[:SKIP]/sqrt2 = (W_last[1:,:]^T dfeat + W_last[0,:] d_sdf)/sqrt2 ; 1/sqrt2 is folded into both packs
    layer_sq<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WLT], nxt<N, OFF_WST>(sg, Wb, o), dfeat,
      AVC_LOAD3_S(L::P_HS, L::G_GBHS, L::P_GAS), AVC_EPID(PF3,
      float wa[16];
      load16(T + o.v[OFF_WL0_ACC], t, h, wa);
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {
        const float s0 = sig_from_h((float)d.h0[j]), s1 = sig_from_h((float)d.h1[j]);
        as_[2 * t][j] = (__bf16)(second_term((float)d.b0[j], (float)d.g0[j], s0) + (acc[j] + wa[j] * dsdfS) * s0);
        as_[2 * t + 1][j] = (__bf16)(second_term((float)d.b1[j], (float)d.g1[j], s1) + (acc[8 + j] + wa[8 + j] * dsdfS) * s1);
      }
      pin2(as_[2 * t], as_[2 * t + 1]);
      tile_store<false>(tiles, L::G_ABS + t, as_[2 * t], as_[2 * t + 1]);
    ));
    // hbar(prev) = W^T abar(cur); abar(prev) = abar'(prev) + hbar * sigma(h_prev).  RINGED: the tile does not go to the G region
    // (the ring policy hands the whole activation to a consumer workgroup after the layer)
#define AVC_REVERSE(OUT, PH, PB, PG, PT, RINGED)                                                            \
  AVC_LOAD3(PH, PB, PG),                                                                                     \
  AVC_EPID(PF3, _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                              \
            const float s0 = sig_from_h((float)d.h0[j]), s1 = sig_from_h((float)d.h1[j]);                    \
            OUT[2 * t][j] = (__bf16)(second_term((float)d.b0[j], (float)d.g0[j], s0) + acc[j] * s0);         \
            OUT[2 * t + 1][j] = (__bf16)(second_term((float)d.b1[j], (float)d.g1[j], s1) + acc[8 + j] * s1); } \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                  \
          if constexpr (!(RINGED)) tile_store<false>(tiles, (PT) + t, OUT[2 * t], OUT[2 * t + 1]);)
    b8 am[N::HK];
    b8 am0[N::HK];
    const Next first = nxt<N, OFF_CHT>(sg, a.Wb0, o);   // prefetch the first tile of the next block iteration
    // ... and, in the persistent kernel, the next block's delta_o and masks (issued after the first group barrier of the last layer)
#define AVC_F_LASTHOOK AVC_HOOK(if constexpr (PIPE) load_blk_in<N>(a, blk0_next + wv, nblk, lane, *bip);)
    if constexpr (N::NMID == 2) {
      layer_sq<b8, N::SK, N::HT>(sg, Wb, o.v[OFF_WST], nxt<N, OFF_WM1T>(sg, Wb, o), as_,
                                 AVC_REVERSE(am, L::P_HM + N::HT, L::G_GBHM + N::HT, L::P_GAM + N::HT, L::G_ABM + N::HT, R::on));
      if constexpr (R::on) ring.template handoff<N>(1, am, blk0, lane, wv);
      layer_sq<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM1T], nxt<N, OFF_WM0T>(sg, Wb, o), am,
                                 AVC_REVERSE(am0, L::P_HM, L::G_GBHM, L::P_GAM, L::G_ABM, R::on));
      if constexpr (R::on) ring.template handoff<N>(0, am0, blk0, lane, wv);
      layer_sq<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0T], first, am0, AVC_REVERSE(am, L::P_H1, L::G_GBH1, L::P_GA1, L::G_AB1, false), AVC_F_LASTHOOK);
    } else {
      layer_sq<b8, N::SK, N::HT>(sg, Wb, o.v[OFF_WST], nxt<N, OFF_WM0T>(sg, Wb, o), as_,
                                 AVC_REVERSE(am, L::P_HM, L::G_GBHM, L::P_GAM, L::G_ABM, R::on));
      if constexpr (R::on) ring.template handoff<N>(0, am, blk0, lane, wv);
      layer_sq<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0T], first, am, AVC_REVERSE(am0, L::P_H1, L::G_GBH1, L::P_GA1, L::G_AB1, false), AVC_F_LASTHOOK);
    }
  }
#undef AVC_RELU_BWD
#undef AVC_SECOND
#undef AVC_LOAD3
#undef AVC_LOAD3_S
#undef AVC_F_LOADB
#undef AVC_E_LOADH
#undef AVC_SECOND_
#undef AVC_SECOND_S
#undef AVC_F_LASTHOOK
#undef AVC_REVERSE
}
