// Forward kernels of the SDF + colour MLP (gfx950, f16 MFMA, fp32 accumulate).
//   avc_sdf_forward       SDF only (row 0 of the last layer) -- the no-grad evaluations of the hierarchical sampler
//                         (renderer.py:337-338,187) and of extract_fields (renderer.py:10-25)
//   avc_render_points_fwd sdf + normal (d sdf/dx) + 6 colour channels per sample point (renderer.py:221-232)
// two output tiles per MFMA stream (independent accumulator chains) in the layers of THIS file: forward 6.21 -> 6.02 ms, training
// forward 8.20 -> 7.90 ms per 4 Mi points; the backward kernel loses 4 % with it (profiles/r03_ab_kernels.txt) and keeps one chain
#ifndef AVC_PAIR
#define AVC_PAIR 1
#endif
#include "avc_mlp.h"
#ifndef FWD_WPB
#define FWD_WPB 8   // one 8-wave workgroup per CU shares every staged weight tile (LDS-DMA fill rate is the scarce resource)
#endif
#ifndef SDF_WPB
#define SDF_WPB 12   // wavefronts per workgroup of the SDF-only kernel: the only kernel of the engine that fits 168 VGPRs (16 spilled), i.e. 3 waves per SIMD and 384 points per staged weight tile (8 -> 12: -6 %)
#endif
#ifndef FWD_G
#define FWD_G 4
#endif
#ifndef AVC_SPREAD_STORES
#define AVC_SPREAD_STORES 1   // 1: the tile stores of a layer's input are dealt to ALL weight-group intervals of the layer, 0: all after the first barrier
#endif
#if AVC_SPREAD_STORES
#define AVC_STORE_HOOK(KEEP, NT, PT, ARR) AVC_HOOKG(tiles_store_part<KEEP, NT>(tiles, PT, ARR, grp_, ngrp_);)
#else
#define AVC_STORE_HOOK(KEEP, NT, PT, ARR) AVC_HOOK(tiles_store<KEEP, NT>(tiles, PT, ARR);)
#endif
// timing ablations of the training forward (results are garbage): which of its extra stores cost what
#ifdef AVC_ABL_NOMASK
constexpr bool ABL_NOMASK = true;
#else
constexpr bool ABL_NOMASK = false;
#endif
#ifdef AVC_ABL_NORSTORE
constexpr bool ABL_NORSTORE = true;
#else
constexpr bool ABL_NORSTORE = false;
#endif
#ifdef AVC_ABL_NOGASTORE
constexpr bool ABL_NOGASTORE = true;
#else
constexpr bool ABL_NOGASTORE = false;
#endif
#ifdef AVC_ABL_NOMISC
constexpr bool ABL_NOMISC = true;
#else
constexpr bool ABL_NOMISC = false;
#endif
#include "../../include/avc.h"
#include <stdlib.h>
#ifndef AVC_SDF_PPW_DEFAULT
#define AVC_SDF_PPW_DEFAULT 32   // points per wavefront of avc_sdf_forward: 32 (mlp_sdf_kernel) | 64 (mlp_sdf2_kernel)
#endif

template <class N>
__global__ __launch_bounds__(64 * SDF_WPB) void mlp_sdf_kernel(PointSrc ps, long npts, const h8* __restrict__ Wf,
                                                               const float* __restrict__ T,
                                                               float* __restrict__ sdf_out, const int* __restrict__ slot,
                                                               int ld_out) {
  constexpr AvcOffsets o = Off<N>::value;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef StageT<FWD_G> ST;
  avc_static_wave_priority();
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5;
  const int p = lane & 31;
  // one wavefront per 32-point block; the wavefronts of a workgroup share every weight tile through LDS, so no wave
  // may leave early: blocks past the end are clamped to the last point and only their stores are suppressed.
  const long blk = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  long i = blk * 32 + p;
  const bool valid = i < npts;
  if (!valid) i = npts - 1;
  ST sg = stage_init<ST::G>(lds);
  stage_issue(sg, nxt<N, OFF_W0>(sg, Wf, o), 0);
  const lds_tab_t Tl = tab_to_lds(lds + ST::LDS_BYTES, T, o.v[OFF_TAB_END]);
  __syncthreads();
  float x0[3];
  fetch_point(ps, i, x0);
  long oi = i;
  if (slot) {
    const long ray = i / ps.S;
    oi = ray * ld_out + slot[i];
  }
  const float sdfv = sdf_only<N>(sg, Wf, Tl, o, h, x0);
  if (valid && h == 0) sdf_out[oi] = sdfv;
}


// ---- the same with two 32-point groups per wavefront (sdf_only2, avc_mlp.h): 4-wave workgroups, one wavefront per SIMD on the
// ---- 512-entry unified register file; every staged weight tile serves 256 points and every LDS A fragment two MFMAs
#ifndef SDF2_WPB
#define SDF2_WPB 4
#endif
template <class N>
__global__ __launch_bounds__(64 * SDF2_WPB) void mlp_sdf2_kernel(PointSrc ps, long npts, const h8* __restrict__ Wf,
                                                                 const float* __restrict__ T, float* __restrict__ sdf_out,
                                                                 const int* __restrict__ slot, int ld_out) {
  constexpr AvcOffsets o = Off<N>::value;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef StageT<FWD_G> ST;
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5;
  const int p = lane & 31;
  const long blk = 2 * ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));   // this wavefront's blocks: blk, blk + 1
  ST sg = stage_init<ST::G>(lds);
  stage_issue(sg, nxt<N, OFF_W0>(sg, Wf, o), 0);
  const lds_tab_t Tl = tab_to_lds(lds + ST::LDS_BYTES, T, o.v[OFF_TAB_END]);
  __syncthreads();
  float x[2][3];
  long oi[2];
  bool valid[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    long i = (blk + q) * 32 + p;
    valid[q] = i < npts;
    if (!valid[q]) i = npts - 1;
    fetch_point(ps, i, x[q]);
    oi[q] = i;
    if (slot) oi[q] = (i / ps.S) * ld_out + slot[i];
  }
  float sdfv[2];
  sdf_only2<N>(sg, Wf, Tl, o, h, x, sdfv);
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (valid[q] && h == 0) sdf_out[oi[q]] = sdfv[q];
}

// ---------------------------------------------------------------------------------------------------------------
// avc_render_points_fwd / avc_render_points_fwd_train: sdf + normal + colour.  The normal sweep needs sigma(h_l) of every
// trunk layer in REVERSE order; keeping h1..h_s of 32 points in registers is 248 VGPRs of state on top of the working set
// (the first version of this kernel paid for it with 331 spilled registers).  The H-wide activations therefore leave the
// registers as 2-KiB tiles in fragment layout (one coalesced 1-KiB store per k-step) and are read back tile by tile in the
// epilogues of the sweep; h_s never leaves the registers (it is consumed at once by g_a,s and by the feature layer).
//   TRAIN = false: the tiles go to a per-wavefront slot that is reused for every block (2048 slots x 64 KiB = 134 MB).
//   TRAIN = true : the tiles go to the block's operand panels (PanelLayout) together with everything else the backward pass and
//                  the weight-gradient products need from the forward pass -- PE values, h_s, g_a of every layer, the feature
//                  vector, [x,n], r1, r2 and the ReLU masks -- so that the backward kernel recomputes NOTHING of the forward
//                  (renderer.py:221-232 runs once per iteration, as in the reference).  The extra stores are streaming
//                  (non-temporal) and ride under a kernel that is bound by its matrix / vector work.
// Persistent workgroups.  Plain (temporal) accesses for the tiles that are re-read within the same block.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ FragPair<h8> fwd_abl_pair(int t) {   // (AVC_ABL_FWD_NORR: an opaque constant in place of a tile read back)
  FragPair<h8> d;
  _Pragma("unroll") for (int j = 0; j < 8; ++j) { d.a0[j] = (_Float16)(0.25f + 0.01f * t); d.a1[j] = (_Float16)(0.5f - 0.01f * t); }
  asm volatile("" : "+v"(d.a0), "+v"(d.a1));
  return d;
}
template <typename P> __device__ __forceinline__ P launder_ptr(P p) {
  asm volatile("" : "+s"(p));
  return p;
}

template <class N, bool TRAIN>
__global__ __launch_bounds__(64 * FWD_WPB) void mlp_render_kernel(PointSrc ps, long npts, const h8* __restrict__ Wf0,
                                                                  const float* __restrict__ T0,
                                                                  float* __restrict__ sdf_out, float* __restrict__ normal_out,
                                                                  float* __restrict__ rgb_out, char* __restrict__ store,
                                                                  unsigned short* __restrict__ masks) {
  constexpr AvcOffsets o = Off<N>::value;
  typedef typename std::conditional<TRAIN, PanelLayout<N>, ScratchLayout<N>>::type L;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef StageT<FWD_G> ST;
  avc_static_wave_priority();
  const int lane0 = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const long nblk = (npts + 31) >> 5;
  const long wslot = (long)blockIdx.x * FWD_WPB + wv;
  char* slot0 = store + wslot * (long)L::P_TILES * 2048;   // TRAIN = false: this wave's private slot
  ST sg = stage_init<FWD_G>(lds);
  stage_issue(sg, nxt<N, OFF_W0>(sg, Wf0, o), 0);
  const lds_tab_t Tl = tab_to_lds(lds + ST::LDS_BYTES, T0, o.v[OFF_TAB_END]);
  __syncthreads();
  for (long blk0 = (long)blockIdx.x * FWD_WPB; blk0 < nblk; blk0 += (long)gridDim.x * FWD_WPB) {
    // opaque per iteration: otherwise LICM hoists the loop-invariant addresses (weight tiles, ~25 per-lane table addresses) out of
    // the persistent loop and spills them at its top
    const h8* Wf = launder_ptr(Wf0);
    lds_tab_t T = Tl;
    asm volatile("" : "+s"(T));
    // ... and every lane-derived constant (fragment / table addresses, PE frequencies): recomputed per block from an opaque lane id
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int h = lane >> 5, p = lane & 31;
    sg.lane = lane;
    const long blk = blk0 + wv;
    // TRAIN: block nblk of the panel / mask buffers is a sink for the wavefronts past the end (they walk the tile sequence for the barriers)
    const PanelPtr tiles = panel_ptr(TRAIN ? store + (blk < nblk ? blk : nblk) * (long)L::P_TILES * 2048 : slot0, lane);
    AVC_GLOBAL unsigned short* mk = as_global(masks) + (TRAIN ? (blk < nblk ? blk : nblk) * (long)PanelLayout<N>::MASK_U16 + lane : 0);
    long i = blk * 32 + p;
    const bool valid = i < npts;
    if (!valid) i = npts - 1;
    float x[3];
    fetch_point(ps, i, x);
    // ---------------------------------------------------------------- trunk
    float sdf;
    h8 g_s[N::SK];
    {
      h8 hs[N::SK];
      float part = 0.f;
      PE pe;
      pe_compute(x, h, pe);
      {
        const auto wpe = T + o.v[OFF_WL0_PE] + h * 24;
#pragma unroll
        for (int q = 0; q < 24; ++q) part += wpe[q] * pe.v[q];
        asm volatile("" : "+v"(part));   // done HERE: hipcc otherwise sinks the 24 fmacs to the first use of `part`, three layers
                                         // down, and keeps (spills, then reloads one by one) their 48 operands until then
      }
      h8 pef[3];
      pe_to_frags_f16(pe, x, h, pef);
      // every tile store below is issued by the hook of the NEXT layer (right after its first group barrier, see avc_mlp.h)
      // (the bias rows enter through the accumulators: TabBias, avc_stage.h)
#define AVC_F_BIAS(OFFB) TabBias{T + o.v[OFFB], h}
#define AVC_F_ACT(OUT)                                                                        \
  AVC_EPI(float a[16];                                                                        \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r]);           \
          acc_to_frags(a, OUT[2 * t], OUT[2 * t + 1]);)
#define AVC_F_LAST()                                                                          \
  AVC_EPI(float b[16], a[16];                                                                 \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r]);           \
          load16(T + o.v[OFF_WL0_ACC], t, h, b);                                              \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) part += b[r] * a[r];                 \
          acc_to_frags(a, hs[2 * t], hs[2 * t + 1]);                                          \
          float w0[8], w1[8];                                                                 \
          load8(T + o.v[OFF_WL0_FRAG], 2 * t, h, w0); load8(T + o.v[OFF_WL0_FRAG], 2 * t + 1, h, w1); \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                     \
            g_s[2 * t][j] = (_Float16)(w0[j] * sig_from_h(a[j]));                             \
            g_s[2 * t + 1][j] = (_Float16)(w1[j] * sig_from_h(a[8 + j])); }                   \
          pin2(g_s[2 * t], g_s[2 * t + 1]);)
      {
        h8 h1[N::HK];
        layer_s<h8, 3, N::HT>(sg, Wf, o.v[OFF_W0], nxt<N, OFF_WM0>(sg, Wf, o), pef, AVC_F_ACT(h1), AVC_HOOK(
          if constexpr (TRAIN) {
            h8 zf;
            _Pragma("unroll") for (int j = 0; j < 8; ++j) zf[j] = (_Float16)0.f;
            tile_store<false>(tiles, L::P_H0, pef[0], pef[1]);
            tile_store<false>(tiles, L::P_H0 + 1, pef[2], zf);
          }), AVC_F_BIAS(OFF_B0));
        h8 hm0[N::HK];
        if constexpr (N::NMID == 2) {
          layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wf, o), h1, AVC_F_ACT(hm0),
                                    AVC_STORE_HOOK(true, N::HT, L::P_H1, h1), AVC_F_BIAS(OFF_BM0));
          h8 hm1[N::HK];
          layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wf, o), hm0, AVC_F_ACT(hm1),
                                    AVC_STORE_HOOK(true, N::HT, L::P_HM, hm0), AVC_F_BIAS(OFF_BM1));
          layer_s<h8, N::HK, N::ST>(sg, Wf, o.v[OFF_WS], nxt<N, OFF_WL>(sg, Wf, o), hm1, AVC_F_LAST(),
                                    AVC_STORE_HOOK(true, N::HT, L::P_HM + N::HT, hm1), AVC_F_BIAS(OFF_BS));
        } else {
          layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wf, o), h1, AVC_F_ACT(hm0),
                                    AVC_STORE_HOOK(true, N::HT, L::P_H1, h1), AVC_F_BIAS(OFF_BM0));
          layer_s<h8, N::HK, N::ST>(sg, Wf, o.v[OFF_WS], nxt<N, OFF_WL>(sg, Wf, o), hm0, AVC_F_LAST(),
                                    AVC_STORE_HOOK(true, N::HT, L::P_HM, hm0), AVC_F_BIAS(OFF_BS));
        }
      }
      sdf = xhalf_sum(part) + T[o.v[OFF_BL0]];
      // feature = rows 1..H of the last layer on u = [h_s ; pe]/sqrt2 (1/sqrt2 folded into the packed weights)
      layer2_s<h8, N::SK, 3, N::HT>(sg, Wf, o.v[OFF_WL], nxt<N, OFF_WST>(sg, Wf, o), hs, pef, AVC_EPI(
        float a[16];
        _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = acc[r];
        h8 f0, f1;
        acc_to_frags(a, f0, f1);
        tile_store<true>(tiles, L::P_FEAT + t, f0, f1);
      ), AVC_HOOK(
        if constexpr (TRAIN && !ABL_NOMISC) {
          tiles_store<false, N::ST>(tiles, L::P_HS, hs);
          tiles_store<false, N::ST>(tiles, L::P_GAS, g_s);
        }), AVC_F_BIAS(OFF_BL));
    }
    // ---------------------------------------------------------------- normal sweep: g_h(prev) = W^T g_a ; g_a(prev) = g_h sigma(h_prev)
    float n[3];
    {
#ifdef AVC_ABL_FWD_NORR   // timing ablation only (garbage results): the normal sweep WITHOUT its re-reads of the parked h tiles = the upper bound of
                          // any scheme that keeps h_1 .. h_s resident through the sweep
#define AVC_F_HLOAD(PH) fwd_abl_pair(t)
#else
#define AVC_F_HLOAD(PH) tile_load<false, h8>(tiles, (PH) + t)
#endif
#define AVC_F_NSTEP(OUT, PH)                                                                              \
  AVC_PRE(return AVC_F_HLOAD(PH);),                                                                         \
  AVC_EPID(FragPair<h8>, _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                    \
            OUT[2 * t][j] = (_Float16)(acc[j] * sig_from_h((float)d.a0[j]));                                \
            OUT[2 * t + 1][j] = (_Float16)(acc[8 + j] * sig_from_h((float)d.a1[j])); }                      \
          pin2(OUT[2 * t], OUT[2 * t + 1]);)
#if AVC_SPREAD_STORES
#define AVC_F_GSTORE(PG, G) AVC_HOOKG(if constexpr (TRAIN && !ABL_NOGASTORE) tiles_store_part<false, N::HT>(tiles, PG, G, grp_, ngrp_);)
#else
#define AVC_F_GSTORE(PG, G) AVC_HOOK(if constexpr (TRAIN && !ABL_NOGASTORE) tiles_store<false, N::HT>(tiles, PG, G);)
#endif
      h8 g[N::HK];
      h8 g2[N::HK];
      if constexpr (N::NMID == 2) {
        layer_sq<h8, N::SK, N::HT>(sg, Wf, o.v[OFF_WST], nxt<N, OFF_WM1T>(sg, Wf, o), g_s, AVC_F_NSTEP(g, L::P_HM + N::HT));
        layer_sq<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1T], nxt<N, OFF_WM0T>(sg, Wf, o), g, AVC_F_NSTEP(g2, L::P_HM),
                                   AVC_F_GSTORE(L::P_GAM + N::HT, g));
        layer_sq<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0T], nxt<N, OFF_W0T>(sg, Wf, o), g2, AVC_F_NSTEP(g, L::P_H1),
                                   AVC_F_GSTORE(L::P_GAM, g2));
      } else {
        layer_sq<h8, N::SK, N::HT>(sg, Wf, o.v[OFF_WST], nxt<N, OFF_WM0T>(sg, Wf, o), g_s, AVC_F_NSTEP(g2, L::P_HM));
        layer_sq<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0T], nxt<N, OFF_W0T>(sg, Wf, o), g2, AVC_F_NSTEP(g, L::P_H1),
                                   AVC_F_GSTORE(L::P_GAM, g2));
      }
      float part[3] = {0.f, 0.f, 0.f};
      const auto wpe = T + o.v[OFF_WL0_PE] + h * 24;
      // the PE derivatives are RE-computed here from an opaque copy of x: hipcc otherwise merges this call with the one at the top
      // of the block and keeps its 48 values alive across the whole trunk (59 spill stores + 35 reloads per block)
      float x2[3] = {x[0], x[1], x[2]};
      asm volatile("" : "+v"(x2[0]), "+v"(x2[1]), "+v"(x2[2]));
      PE pe2;
      pe_compute(x2, h, pe2);
      layer_s<h8, N::HK, 2>(sg, Wf, o.v[OFF_W0T], nxt<N, OFF_C0>(sg, Wf, o), g, AVC_EPI(
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {
          const int q = 16 * t + r;
          if (q < 24) part[q % 3] += pe2.d[q] * (acc[r] + wpe[q]);
        }
      ), AVC_F_GSTORE(L::P_GA1, g));
#pragma unroll
      for (int c = 0; c < 3; ++c) n[c] = xhalf_sum(part[c]);
    }
    // ---------------------------------------------------------------- colour MLP (fields.py:154-185)
    float rgb[4];
    {
      h8 xn[1];
#pragma unroll
      for (int j = 0; j < 8; ++j) xn[0][j] = (_Float16)0.f;
      if (h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { xn[0][c] = (_Float16)x[c]; xn[0][3 + c] = (_Float16)n[c]; }
      }
      h8 feat[N::HK];
#pragma unroll
      for (int t = 0; t < N::HT; ++t) {
        const FragPair<h8> d = tile_load<false, h8>(tiles, L::P_FEAT + t);
        feat[2 * t] = d.a0;
        feat[2 * t + 1] = d.a1;
      }
      // ReLU layers; TRAIN: the activations go out as weight-gradient operands and their > 0 bits (16 per tile and lane, relu_frags) as the
      // masks of the backward pass -- both from the hook of the next layer
#define AVC_F_RELU(OUT, ML)                                                                   \
  AVC_EPI(const unsigned bits = relu_frags<TRAIN>(acc, OUT[2 * t], OUT[2 * t + 1]);            \
          if constexpr (TRAIN && !ABL_NOMASK) mk[((ML) * N::HT + t) * 64] = (unsigned short)bits;)
#if AVC_SPREAD_STORES
#define AVC_F_RSTORE(PT, R)                                                                   \
  AVC_HOOKG(if constexpr (TRAIN && !ABL_NORSTORE) tiles_store_part<false, N::HT>(tiles, PT, R, grp_, ngrp_);)
#else
#define AVC_F_RSTORE(PT, R)                                                                   \
  AVC_HOOK(if constexpr (TRAIN && !ABL_NORSTORE) tiles_store<false, N::HT>(tiles, PT, R);)
#endif
      h8 r1[N::HK];
      h8 r2[N::HK];
      if constexpr (N::NCMID == 1) {
        layer2_s<h8, N::HK, 1, N::HT>(sg, Wf, o.v[OFF_C0], nxt<N, OFF_CM0>(sg, Wf, o), feat, xn, AVC_F_RELU(r1, 0), AVC_HOOK(
          if constexpr (TRAIN) {
            h8 zf;
            _Pragma("unroll") for (int j = 0; j < 8; ++j) zf[j] = (_Float16)0.f;
            tile_store<false>(tiles, L::P_XN, xn[0], zf);
          }), AVC_F_BIAS(OFF_CB0));
        layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_CM0], nxt<N, OFF_CH>(sg, Wf, o), r1, AVC_F_RELU(r2, 1),
                                  AVC_F_RSTORE(L::P_R1, r1), AVC_F_BIAS(OFF_CBM0));
      } else {
        layer2_s<h8, N::HK, 1, N::HT>(sg, Wf, o.v[OFF_C0], nxt<N, OFF_CH>(sg, Wf, o), feat, xn, AVC_F_RELU(r1, 0), AVC_HOOK(
          if constexpr (TRAIN) {
            h8 zf;
            _Pragma("unroll") for (int j = 0; j < 8; ++j) zf[j] = (_Float16)0.f;
            tile_store<false>(tiles, L::P_XN, xn[0], zf);
          }), AVC_F_BIAS(OFF_CB0));
#pragma unroll
        for (int s = 0; s < N::HK; ++s) r2[s] = r1[s];
      }
      // the first tile of the next block iteration is prefetched under the head layer
      layer_s<h8, N::HK, 1>(sg, Wf, o.v[OFF_CH], nxt<N, OFF_W0>(sg, Wf0, o), r2, AVC_EPI(
        float b[16];
        load16(T + o.v[OFF_CBH], 0, h, b);
        _Pragma("unroll") for (int r = 0; r < 4; ++r) rgb[r] = sigmoidf_(acc[r] + b[r]);
      ), AVC_F_RSTORE((N::NCMID == 1 ? L::P_R2 : L::P_R1), r2));
    }
    if (valid) {
      if (h == 0) {
        sdf_out[i] = sdf;
        normal_out[3 * i + 0] = n[0]; normal_out[3 * i + 1] = n[1]; normal_out[3 * i + 2] = n[2];
        rgb_out[6 * i + 0] = rgb[0]; rgb_out[6 * i + 1] = rgb[1]; rgb_out[6 * i + 2] = rgb[2]; rgb_out[6 * i + 3] = rgb[3];
      } else {
        rgb_out[6 * i + 4] = rgb[0]; rgb_out[6 * i + 5] = rgb[1];
      }
    }
  }
}

extern "C" long avc_fwd_scratch_bytes_per_wave(int net) {
  return (long)(net == AVC_NET_FULL ? ScratchLayout<NetFull>::P_TILES : ScratchLayout<NetSmall>::P_TILES) * 2048;
}
extern "C" int avc_fwd_panel_tiles(int net) {
  return net == AVC_NET_FULL ? PanelLayout<NetFull>::P_TILES : PanelLayout<NetSmall>::P_TILES;
}
extern "C" int avc_grad_panel_tiles(int net) {
  return net == AVC_NET_FULL ? PanelLayout<NetFull>::G_TILES : PanelLayout<NetSmall>::G_TILES;
}
extern "C" int avc_mask_u16_per_block(int net) {
  return net == AVC_NET_FULL ? PanelLayout<NetFull>::MASK_U16 : PanelLayout<NetSmall>::MASK_U16;
}

static int grid_for(long npts, int waves_per_block, int max_blocks) {
  long nblk = (npts + 31) / 32;
  long g = (nblk + waves_per_block - 1) / waves_per_block;
  if (g > max_blocks) g = max_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

static int check_tab(const int* offs) {
  if (offs[OFF_TAB_END] * 4 > AVC_TAB_LDS_BYTES) { avc_set_error("fp32 table does not fit its LDS window"); return 1; }
  return 0;
}

static int launch_sdf(int net, PointSrc ps, long npts, const void* wf, const float* tab, const int* offs,
                      float* sdf_out, const int* slot, int ld_out, void* stream) {
  if (npts <= 0) return 0;
  if (check_tab(offs)) return 1;
  if (!(net == AVC_NET_FULL ? offsets_match<NetFull>(offs) : offsets_match<NetSmall>(offs))) {
    avc_set_error("packed-blob offsets differ from the compiled-in table (regenerate csrc/avc_offsets_gen.h)");
    return 1;
  }
  hipStream_t s = (hipStream_t)stream;
  const int lds_bytes = StageT<FWD_G>::LDS_BYTES + AVC_TAB_LDS_BYTES;
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen)) {
    hipFuncSetAttribute((const void*)mlp_sdf_kernel<NetFull>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipFuncSetAttribute((const void*)mlp_sdf_kernel<NetSmall>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipFuncSetAttribute((const void*)mlp_sdf2_kernel<NetFull>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipFuncSetAttribute((const void*)mlp_sdf2_kernel<NetSmall>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  }
  // AVC_SDF_POINTS_PER_WAVE=64: the two-group kernel (A/B partner: profiles/r06_ab_kernels.txt); read once per process
  static const int ppw = [] { const char* e = getenv("AVC_SDF_POINTS_PER_WAVE"); return e ? atoi(e) : AVC_SDF_PPW_DEFAULT; }();
  if (ppw == 64) {
    const int grid2 = grid_for(npts, 2 * SDF2_WPB, 0x7fffffff);
    if (net == AVC_NET_FULL)
      hipLaunchKernelGGL((mlp_sdf2_kernel<NetFull>), dim3(grid2), dim3(64 * SDF2_WPB), lds_bytes, s, ps, npts, (const h8*)wf, tab, sdf_out, slot, ld_out);
    else if (net == AVC_NET_SMALL)
      hipLaunchKernelGGL((mlp_sdf2_kernel<NetSmall>), dim3(grid2), dim3(64 * SDF2_WPB), lds_bytes, s, ps, npts, (const h8*)wf, tab, sdf_out, slot, ld_out);
    else { avc_set_error("unknown net id"); return 1; }
    return avc_check_launch("avc_sdf_forward");
  }
  const int wpb = SDF_WPB;   // wavefronts per workgroup
  const int grid = grid_for(npts, wpb, 0x7fffffff);
  if (net == AVC_NET_FULL)
    hipLaunchKernelGGL((mlp_sdf_kernel<NetFull>), dim3(grid), dim3(64 * wpb), lds_bytes, s, ps, npts, (const h8*)wf, tab,
                       sdf_out, slot, ld_out);
  else if (net == AVC_NET_SMALL)
    hipLaunchKernelGGL((mlp_sdf_kernel<NetSmall>), dim3(grid), dim3(64 * wpb), lds_bytes, s, ps, npts, (const h8*)wf, tab,
                       sdf_out, slot, ld_out);
  else {
    avc_set_error("unknown net id");
    return 1;
  }
  return avc_check_launch("avc_sdf_forward");
}

extern "C" int avc_sdf_forward(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                               int S, int ldz, long npts, const void* wf16, const float* tab, const int* offs,
                               float* sdf_out, const int* slot, int ld_out, void* stream) {
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, 0, 0.f};
  return launch_sdf(net, ps, npts, wf16, tab, offs, sdf_out, slot, ld_out, stream);
}

template <bool TRAIN>
static int launch_render(int net, PointSrc ps, long npts, const void* wf16, const float* tab, const int* offs, float* sdf_out,
                         float* normal_out, float* rgb_out, long max_waves, void* store, void* masks, void* stream,
                         const char* what) {
  if (check_tab(offs)) return 1;
  if (!(net == AVC_NET_FULL ? offsets_match<NetFull>(offs) : offsets_match<NetSmall>(offs))) {
    avc_set_error("packed-blob offsets differ from the compiled-in table (regenerate csrc/avc_offsets_gen.h)");
    return 1;
  }
  long maxg = max_waves / FWD_WPB;
  if (maxg < 1) maxg = 1;
  const int grid = grid_for(npts, FWD_WPB, (int)(maxg < 0x7fffffff ? maxg : 0x7fffffff));
  const int lds_bytes = StageT<FWD_G>::LDS_BYTES + AVC_TAB_LDS_BYTES;
  hipStream_t s = (hipStream_t)stream;
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen)) {
    (void)hipFuncSetAttribute((const void*)mlp_render_kernel<NetFull, TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void*)mlp_render_kernel<NetSmall, TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  }
  if (net == AVC_NET_FULL)
    hipLaunchKernelGGL((mlp_render_kernel<NetFull, TRAIN>), dim3(grid), dim3(64 * FWD_WPB), lds_bytes, s, ps, npts, (const h8*)wf16, tab,
                       sdf_out, normal_out, rgb_out, (char*)store, (unsigned short*)masks);
  else if (net == AVC_NET_SMALL)
    hipLaunchKernelGGL((mlp_render_kernel<NetSmall, TRAIN>), dim3(grid), dim3(64 * FWD_WPB), lds_bytes, s, ps, npts, (const h8*)wf16, tab,
                       sdf_out, normal_out, rgb_out, (char*)store, (unsigned short*)masks);
  else {
    avc_set_error("unknown net id");
    return 1;
  }
  return avc_check_launch(what);
}

extern "C" int avc_render_points_fwd(int net, const float* pts, const float* rays_o, const float* rays_d,
                                     const float* z, int S, int ldz, float sample_dist, long npts, const void* wf16,
                                     const float* tab, const int* offs, float* sdf_out, float* normal_out,
                                     float* rgb_out, long max_waves, void* scratch, void* stream) {
  if (npts <= 0) return 0;
  if (!scratch) { avc_set_error("avc_render_points_fwd: scratch == NULL"); return 1; }
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, pts ? 0 : 1, sample_dist};
  return launch_render<false>(net, ps, npts, wf16, tab, offs, sdf_out, normal_out, rgb_out, max_waves, scratch, nullptr, stream,
                              "avc_render_points_fwd");
}

extern "C" int avc_render_points_fwd_train(int net, const float* pts, const float* rays_o, const float* rays_d,
                                           const float* z, int S, int ldz, float sample_dist, long npts, const void* wf16,
                                           const float* tab, const int* offs, float* sdf_out, float* normal_out,
                                           float* rgb_out, long max_waves, void* panels, void* masks, void* stream) {
  if (npts <= 0) return 0;
  if (!panels || !masks) { avc_set_error("avc_render_points_fwd_train: panels / masks == NULL"); return 1; }
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, pts ? 0 : 1, sample_dist};
  return launch_render<true>(net, ps, npts, wf16, tab, offs, sdf_out, normal_out, rgb_out, max_waves, panels, masks, stream,
                             "avc_render_points_fwd_train");
}
