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
// (gbar_hs -- the last tiles the second-order sweep produces, the first the reverse sweep consumes -- ALWAYS stays in registers across
//  the turn: it has no panel since round 5, see col_sums below; keeping h_s as well spilled 78 registers and lost 5 %)
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
  float* colsum;   // [wavefronts of the launch][PanelLayout::CS_FLOATS]: per-wavefront column sums over its points of gbar_hs and gbar_h0
};

// ---- Column sums over the points.  The second-order term of row 0 of the last SDF layer is dW_last[0, :] += sum_points gbar_u,
// gbar_u = [gbar_hs ; gbar_h0] / sqrt2: a sum, not a product.  Rounds 2-4 stored gbar_hs and a constant-one tile and let
// the weight-gradient kernel contract "1 (x) [gbar_hs | gbar_h0]" -- 8 tiles written and 10 read per block for 39 + SKIP numbers.  The
// values are in this kernel's registers: 32 points on the 32 lanes of a half-wave, so the sum is a lane reduction (transpose_sum below),
// added to the wavefront's own slot in LDS (single writer, plain read + write: deterministic) and written out once at the end of the
// launch; the host adds the rows up and scatters them into the dense gradient (packing.Layout.cs_*).  fp32 throughout -- the product
// had rounded gbar to bf16 first.
#define AVC_CS_DUMMY 64   // floats behind a slot that take the writes of the non-writing lanes (distinct addresses: no branch, no conflict)
template <class N> struct ColSum {
  static constexpr int SLOT_FLOATS = PanelLayout<N>::CS_FLOATS + AVC_CS_DUMMY;
  static constexpr int LDS_BYTES = BWD_WPB * SLOT_FLOATS * 4;
};
typedef AVC_LDS float* cs_slot_t;
// value of the lane's xor-1 / xor-2 partner (DPP quad permutes: no LDS traffic)
__device__ __forceinline__ float quad_xor1(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
}
// Transposing reduction: NV = 8 or 16 values per lane -> ONE per lane, the sum over the half-wave's 32 lanes of value (lane & (NV - 1)).
// Every butterfly step halves the number of values: of the pair (a, b) the lane keeps the one its own bit selects and gives the other to
// its partner, who keeps exactly that one -- NV - 1 exchanges instead of NV x 5.  Steps 1, 2 are DPP quad permutes, the others go
// through the LDS crossbar (ds_bpermute).  (LDS float ATOMICS, the first version of this, cost ~900 cycles per instruction on gfx950:
// 136 of them per block made the kernel 45 % slower, profiles/r05_ab_kernels.txt.)
template <int NV>
__device__ __forceinline__ float transpose_sum(const float (&v)[NV], int lane) {
  static_assert(NV == 8 || NV == 16, "8 or 16 values per lane");
  float w[NV / 2];
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
  for (int k = 0; k < NV / 2; ++k) w[k] = (b0 ? v[2 * k + 1] : v[2 * k]) + quad_xor1(b0 ? v[2 * k] : v[2 * k + 1]);
  float x[NV / 4];
#pragma unroll
  for (int k = 0; k < NV / 4; ++k) x[k] = (b1 ? w[2 * k + 1] : w[2 * k]) + quad_xor2(b1 ? w[2 * k] : w[2 * k + 1]);
  float y[NV / 8];
#pragma unroll
  for (int k = 0; k < NV / 8; ++k) y[k] = (b2 ? x[2 * k + 1] : x[2 * k]) + __shfl_xor(b2 ? x[2 * k] : x[2 * k + 1], 4);
  float z;
  if constexpr (NV == 16) z = (b3 ? y[1] : y[0]) + __shfl_xor(b3 ? y[0] : y[1], 8);
  else z = y[0] + __shfl_xor(y[0], 8);
  return z + __shfl_xor(z, 16);
}
// add the block's column sums of NV values per lane to slot[base + (lane & (NV - 1))]: the lanes of the half-wave's first NV lanes
// write (their half's index is folded into `base` by the caller), every other lane read-modify-writes its own dummy word.  Plain LDS
// read + write: the slot has a single writer, this wavefront.
template <class N, int NV>
__device__ __forceinline__ void col_sums(cs_slot_t slot, int lane, int base, const float (&v)[NV]) {
#ifdef AVC_ABL_CS_NODPP
  const float total = v[0];
#else
  const float total = transpose_sum<NV>(v, lane);
#endif
  const bool writer = (lane & 31) < NV;
  cs_slot_t dst = writer ? slot + base + (lane & (NV - 1)) : slot + PanelLayout<N>::CS_FLOATS + (lane & 63);
#ifdef AVC_ABL_CS_NOADD
  asm volatile("" :: "v"(total), "v"(dst));
#else
  *dst = *dst + total;
#endif
}
template <class N>
__device__ __forceinline__ cs_slot_t cs_init(char* lds_base, int wv, int lane) {
  cs_slot_t slot = (cs_slot_t)lds_base + wv * ColSum<N>::SLOT_FLOATS;
  for (int k = lane; k < ColSum<N>::SLOT_FLOATS; k += 64) slot[k] = 0.f;
  return slot;
}
// after the last block: the slot goes to row `row` of the launch's colsum buffer
template <class N>
__device__ __forceinline__ void cs_flush(cs_slot_t slot, float* out, long row, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int k = lane; k < PanelLayout<N>::CS_FLOATS; k += 64) out[row * PanelLayout<N>::CS_FLOATS + k] = slot[k];
}

// The inputs the FIRST MFMA chain and the first epilogues of a block wait for: delta_o (from d_rgb and the forward's colours) and the
// ReLU masks.  AVC_BWD_PIPE_IN=1: the persistent kernel requests them for its NEXT block under the last layer of the current one
// (loop-carried, 20 VGPRs) instead of at the top of the block, where all eight wavefronts sit out one exposed HBM round trip.
#ifndef AVC_BWD_PIPE_IN
#define AVC_BWD_PIPE_IN 1   // (profiles/r05_ab_kernels.txt: 9.81 -> 9.48 ms per 4 Mi points)
#endif
template <class N> struct BlkIn { b8 dof; unsigned m1[N::HT], m2[N::HT]; float dn[3], dsdf; };   // (+ the cotangents d_normal, d_sdf: used mid-block, behind barriers no load can be hoisted over)
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
#pragma unroll
  for (int c = 0; c < 3; ++c) bi.dn[c] = a.d_normal[3 * i + c] * vmask;
  bi.dsdf = a.d_sdf[i] * vmask;
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
                                           int wv, R& ring, cs_slot_t cs, BlkIn<N>* bip = nullptr, long blk0_next = 0) {
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
  float dsdf_in;
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
    const float dn_in[3] = {bi.dn[0], bi.dn[1], bi.dn[2]};
    dsdf_in = bi.dsdf;
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
      // (lanes past the last point: exact zeros by SELECTION, not by 0 * x -- everything the second-order sweep and the unmasked column
      // sums of gbar_h0 / gbar_hs derive from nbar is then zero whatever the colour sweep left in those lanes)
      nbar[0] = valid ? dn_in[0] + (h ? o3 : a3) : 0.f;
      nbar[1] = valid ? dn_in[1] + (h ? a0 : o0) : 0.f;
      nbar[2] = valid ? dn_in[2] + (h ? a1 : o1) : 0.f;
    }
  }
  const float dsdf = dsdf_in;
  const float dsdfS = dsdf * AVC_S;   // OFF_WL0_ACC holds W_last[0,:]/(S sqrt2): undo S for the gradient use
  {   // operand tile with two live features: d_sdf (row 0 of the last layer), slots (half 0, j = 0, 1)
    b8 fs = zero_frag<b8>();
    if (h == 0) {   // d_sdf split hi + lo over two slots (both map to row 0, packing.py): 16 bits of mantissa for the one cotangent whose sums cancel heavily
      fs[0] = (__bf16)dsdf;
      fs[1] = (__bf16)(dsdf - (float)fs[0]);
    }
    tile_store<false>(tiles, L::G_SDF, fs, zero_frag<b8>());
  }
  // ------------------------------------------------------------------ phase E: second-order sweep (i) (bf16)
  b8 gbs[N::SK];        // gbar_hs: stays in registers for the first layer of the reverse sweep (no panel)
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
  {
    b8 gb0[3];
    {
      PE pe4;
      pe_compute(x, h, pe4);
      float g0[24];
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        g0[q] = pe4.d[q] * nbar[q % 3];
        gb0[q >> 3][q & 7] = (__bf16)g0[q];
      }
      // column sums of gbar_h0 over the block's points: [fragment][half][8 slots] behind the ST gbar_hs tiles of the slot
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = g0[8 * s + j];
        col_sums<N, 8>(cs, lane, N::ST * 32 + (s * 2 + h) * 8, v);
      }
    }
    tile_store<false>(tiles, L::G_GB0, gb0[0], gb0[1]);
    tile_store<false>(tiles, L::G_GB0 + 1, gb0[2], zero_frag<b8>());
    // gbar_a = W gbar_h(in); gbar_h(out) = gbar_a * sigma(h_out)
#if defined(AVC_ABL_BWD_NOEH) || defined(AVC_ABL_BWD_RECOMP)
#define AVC_E_LOADH(PH) abl_const_pair<h8>()
#else
#define AVC_E_LOADH(PH) tile_load<(AVC_BWD_E_NT != 0), h8>(ftiles, (PH) + t)
#endif
#define AVC_SECOND(OUT, PH, PT)                                                                              \
  AVC_PRE(return AVC_E_LOADH(PH);),                                                                          \
  AVC_EPID(FragPair<h8>, _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                     \
            OUT[2 * t][j] = (__bf16)(acc[j] * sig_from_h((float)d.a0[j]));                                   \
            OUT[2 * t + 1][j] = (__bf16)(acc[8 + j] * sig_from_h((float)d.a1[j])); }                         \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                  \
          tile_store<true>(tiles, (PT) + t, OUT[2 * t], OUT[2 * t + 1]);)
    // the skip layer's gbar_hs: no panel -- the fragments stay in `gbs`, the column sums over the points go to the wavefront's slot
#define AVC_SECOND_S(OUT, PH)                                                                                \
  AVC_PRE(return AVC_E_LOADH(PH);),                                                                          \
  AVC_EPID(FragPair<h8>, float v[16];                                                                        \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                    \
            v[j] = acc[j] * sig_from_h((float)d.a0[j]);                                                      \
            v[8 + j] = acc[8 + j] * sig_from_h((float)d.a1[j]);                                              \
            OUT[2 * t][j] = (__bf16)v[j];                                                                    \
            OUT[2 * t + 1][j] = (__bf16)v[8 + j]; }                                                          \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                  \
          col_sums<N, 16>(cs, lane, (t * 2 + h) * 16, v);)
    b8 gb1[N::HK];
    layer_sqd<b8, 3, N::HT>(sg, Wb, o.v[OFF_W0G], nxt<N, OFF_WM0>(sg, Wb, o), gb0, AVC_SECOND(gb1, L::P_H1, L::G_GBH1));
    b8 gbm[N::HK];
    if constexpr (N::NMID == 2) {
      layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wb, o), gb1, AVC_SECOND(gbm, L::P_HM, L::G_GBHM));
      b8 gbm1[N::HK];
      layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wb, o), gbm,
                                  AVC_SECOND(gbm1, L::P_HM + N::HT, L::G_GBHM + N::HT));
      layer_sqd<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WS], nxt<N, OFF_WLT>(sg, Wb, o), gbm1, AVC_SECOND_S(gbs, L::P_HS));
    } else {
      layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wb, o), gb1, AVC_SECOND(gbm, L::P_HM, L::G_GBHM));
      layer_sqd<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WS], nxt<N, OFF_WLT>(sg, Wb, o), gbm, AVC_SECOND_S(gbs, L::P_HS));
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
  // the first layer of the reverse sweep takes gbar_hs from the registers of the second-order sweep (it has no panel)
#define AVC_LOAD3_S(PH, PG)                                                                                 \
  AVC_PRE(PF3 d; { const FragPair<h8> a_ = tile_load<true, h8>(ftiles, (PH) + t); d.h0 = a_.a0; d.h1 = a_.a1; } \
          d.b0 = gbs[2 * t]; d.b1 = gbs[2 * t + 1];                                                          \
          { const FragPair<h8> a_ = tile_load<true, h8>(ftiles, (PG) + t); d.g0 = a_.a0; d.g1 = a_.a1; } return d;)
    // ubar[:SKIP]/sqrt2 = (W_last[1:,:]^T dfeat + W_last[0,:] d_sdf)/sqrt2 ; 1/sqrt2 is folded into both packs
    layer_sq<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WLT], nxt<N, OFF_WST>(sg, Wb, o), dfeat,
      AVC_LOAD3_S(L::P_HS, L::P_GAS), AVC_EPID(PF3,
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
#undef AVC_SECOND_S
#undef AVC_F_LASTHOOK
#undef AVC_REVERSE
}
