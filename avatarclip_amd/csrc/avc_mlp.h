// Register-resident SDF + colour MLP forward for one wavefront (32 points).  See avc_common.h for the layout.
// Follows AvatarGen/AppearanceGen/models/fields.py:72-107 (SDFNetwork.forward/.gradient) and :154-185
// (RenderingNetwork.forward, mode 'no_view_dir', extra_color) of the reference.
#pragma once
#include "avc_common.h"
#include "avc_stage.h"

template <int H_, int NMID_, int NCMID_>
struct NetT {
  static constexpr int H = H_;          // hidden width: 256 (confs/examples) or 128 (confs/examples_small)
  static constexpr int NMID = NMID_;    // HxH middle SDF layers: 2 | 1
  static constexpr int NCMID = NCMID_;  // HxH middle colour layers: 1 | 0
  static constexpr int HT = H / 32;     // 32-row output tiles of an H-wide layer
  static constexpr int HK = H / 16;     // 16-deep k-steps of an H-wide input
  static constexpr int SKIP = H - 39;   // width of the layer feeding the skip concat (fields.py:36-39)
  static constexpr int ST = (SKIP + 31) / 32;
  static constexpr int SK = 2 * ST;
};
typedef NetT<256, 2, 1> NetFull;
typedef NetT<128, 1, 0> NetSmall;

struct PointSrc {
  const float* pts;      // [N,3] or nullptr -> ray mode
  const float* rays_o;   // [R,3]
  const float* rays_d;   // [R,3]
  const float* z;        // [R,ldz]
  int S;                 // samples per ray in this launch
  int ldz;
  int midpoint;          // 1: evaluate at section mid-points z + dist/2 (renderer.py:210-215)
  float sample_dist;
};

__device__ __forceinline__ void fetch_point(const PointSrc& ps, long i, float (&x)[3]) {
  if (ps.pts) {
    x[0] = ps.pts[3 * i]; x[1] = ps.pts[3 * i + 1]; x[2] = ps.pts[3 * i + 2];
    return;
  }
  const long ray = i / ps.S;
  const int s = (int)(i - ray * ps.S);
  const float* zr = ps.z + ray * ps.ldz;
  float t = zr[s];
  if (ps.midpoint) {
    const float dist = (s + 1 < ps.S) ? (zr[s + 1] - t) : ps.sample_dist;
    t = t + dist * 0.5f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) x[c] = ps.rays_o[3 * ray + c] + ps.rays_d[3 * ray + c] * t;
}

template <typename V, int KS>
__device__ __forceinline__ const V* tptr(const V* blob, int off, int t, int lane) {
  return blob + (off >> 3) + (long)(t * KS) * 64 + lane;
}


// ---- tile geometry of every packed weight (k-steps per tile, tiles) ----
template <class N, int OFF> struct TileInfo;
#define AVC_TI(OFF, KS_, NT_) template <class N> struct TileInfo<N, OFF> { static constexpr int KS = KS_, NT = NT_; };
AVC_TI(OFF_W0, 3, N::HT)          AVC_TI(OFF_WM0, N::HK, N::HT)   AVC_TI(OFF_WM1, N::HK, N::HT)  AVC_TI(OFF_WS, N::HK, N::ST)
AVC_TI(OFF_WL, N::SK + 3, N::HT)  AVC_TI(OFF_W0T, N::HK, 2)       AVC_TI(OFF_WM0T, N::HK, N::HT) AVC_TI(OFF_WM1T, N::HK, N::HT)
AVC_TI(OFF_WST, N::SK, N::HT)     AVC_TI(OFF_WLT, N::HK, N::ST)   AVC_TI(OFF_C0, N::HK + 1, N::HT) AVC_TI(OFF_CM0, N::HK, N::HT)
AVC_TI(OFF_CH, N::HK, 1)          AVC_TI(OFF_C0T, N::HK, N::HT + 1) AVC_TI(OFF_CM0T, N::HK, N::HT) AVC_TI(OFF_CHT, 1, N::HT)
AVC_TI(OFF_W0G, 3, N::HT)

// descriptor of the FIRST group of packed weight OFF (what the layer before it prefetches)
template <class N, int OFF, class ST>
__device__ __forceinline__ Next nxt(const ST&, const void* blob, const AvcOffsets& o) {
  typedef TileInfo<N, OFF> TI;
  Next n;
  n.ptr = reinterpret_cast<const char*>(blob) + (long)o.v[OFF] * 2;
  n.chunks = TI::KS * (TI::NT < ST::G ? TI::NT : ST::G);
  return n;
}
__device__ __forceinline__ Next no_next() { Next n; n.ptr = nullptr; n.chunks = 0; return n; }

// ---- staged, software-pipelined layer: per group one barrier + the DMA of the next group; inside a group tile t's MFMAs
// ---- are issued before the epilogue of tile t-1, so the VALU / transcendental / store work of one tile hides under the
// ---- matrix pipe of the next (same basic block, no barrier between).
#define AVC_EPI(...) [&](int t, const facc& acc) __attribute__((always_inline)) { __VA_ARGS__ }

template <typename V, int KS, int NT, class ST, typename Epi>
__device__ __forceinline__ void layer_s(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&in)[KS],
                                        Epi&& epi) {
  constexpr int G = ST::G;
  constexpr int NG = (NT + G - 1) / G;
  facc prev;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    __syncthreads();   // group g has landed (hipcc drains vmcnt before the barrier); the other buffer is free
    if (g + 1 < NG) {
      Next n;
      n.ptr = blob + (offw >> 3) + (long)((g + 1) * G * KS) * 64;
      n.chunks = KS * ((NT - (g + 1) * G) < G ? (NT - (g + 1) * G) : G);
      stage_issue(st, n, st.par ^ 1);
    } else {
      stage_issue(st, after, st.par ^ 1);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int t = g * G + j;
      if (t < NT) {
        facc acc = tile_mma<V, KS>(st, j, in);
        if (t > 0) epi(t - 1, prev);
        prev = acc;
        __builtin_amdgcn_sched_barrier(0);   // keep epilogues from being sunk past later tiles
      }
    }
    st.par ^= 1;
  }
  epi(NT - 1, prev);
  __builtin_amdgcn_sched_barrier(0);
}
template <typename V, int KA, int KB, int NT, class ST, typename Epi>
__device__ __forceinline__ void layer2_s(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&ina)[KA],
                                         const V (&inb)[KB], Epi&& epi) {
  constexpr int G = ST::G;
  constexpr int KS = KA + KB;
  constexpr int NG = (NT + G - 1) / G;
  facc prev;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    __syncthreads();
    if (g + 1 < NG) {
      Next n;
      n.ptr = blob + (offw >> 3) + (long)((g + 1) * G * KS) * 64;
      n.chunks = KS * ((NT - (g + 1) * G) < G ? (NT - (g + 1) * G) : G);
      stage_issue(st, n, st.par ^ 1);
    } else {
      stage_issue(st, after, st.par ^ 1);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int t = g * G + j;
      if (t < NT) {
        facc acc = tile_mma2<V, KA, KB>(st, j, ina, inb);
        if (t > 0) epi(t - 1, prev);
        prev = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    st.par ^= 1;
  }
  epi(NT - 1, prev);
  __builtin_amdgcn_sched_barrier(0);
}

// Forward state of one wave (kept in registers for the reverse sweeps).
template <class N>
struct FwdState {
  float x[3];
  PE pe;
  h8 pef[3];
  h8 h1[N::HK];
  h8 hm[N::NMID][N::HK];
  h8 hs[N::SK];
  float sdf;
};

// SDF trunk: layer0 .. skip layer, plus the fp32 sdf dot product (row 0 of the last layer).
// Precondition: tile 0 of OFF_W0 has been issued (stage_issue).  gnext/KSN: the tile consumed after the trunk.
template <class N, class ST>
__device__ __forceinline__ void sdf_trunk(ST& sg, const h8* __restrict__ Wf, const float* __restrict__ T, const AvcOffsets& o,
                                          int h, FwdState<N>& st, const Next& gnext) {
  pe_compute(st.x, h, st.pe);
  pe_to_frags_f16(st.pe, st.x, h, st.pef);
  layer_s<h8, 3, N::HT>(sg, Wf, o.v[OFF_W0], nxt<N, OFF_WM0>(sg, Wf, o), st.pef, AVC_EPI(
    float b[16], a[16];
    load16(T + o.v[OFF_B0], t, h, b);
    _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);
    acc_to_frags(a, st.h1[2 * t], st.h1[2 * t + 1]);
  ));
  if constexpr (N::NMID == 2) {
    layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wf, o), st.h1, AVC_EPI(
      float b[16], a[16];
      load16(T + o.v[OFF_BM0], t, h, b);
      _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);
      acc_to_frags(a, st.hm[0][2 * t], st.hm[0][2 * t + 1]);
    ));
    layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wf, o), st.hm[0], AVC_EPI(
      float b[16], a[16];
      load16(T + o.v[OFF_BM1], t, h, b);
      _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);
      acc_to_frags(a, st.hm[1][2 * t], st.hm[1][2 * t + 1]);
    ));
  } else {
    layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wf, o), st.h1, AVC_EPI(
      float b[16], a[16];
      load16(T + o.v[OFF_BM0], t, h, b);
      _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);
      acc_to_frags(a, st.hm[0][2 * t], st.hm[0][2 * t + 1]);
    ));
  }
  float part = 0.f;
  layer_s<h8, N::HK, N::ST>(sg, Wf, o.v[OFF_WS], gnext, st.hm[N::NMID - 1], AVC_EPI(
    float b[16], a[16];
    load16(T + o.v[OFF_BS], t, h, b);
    _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);
    load16(T + o.v[OFF_WL0_ACC], t, h, b);
    _Pragma("unroll") for (int r = 0; r < 16; ++r) part += b[r] * a[r];
    acc_to_frags(a, st.hs[2 * t], st.hs[2 * t + 1]);
  ));
  {
    const float* wpe = T + o.v[OFF_WL0_PE] + h * 24;
#pragma unroll
    for (int q = 0; q < 24; ++q) part += wpe[q] * st.pe.v[q];
  }
  st.sdf = xhalf_sum(part) + T[o.v[OFF_BL0]];
}

// SDF value only (avc_sdf_forward): same layers as sdf_trunk but every activation array dies as soon as the next layer
// has consumed it, which keeps the kernel at two wavefronts per SIMD.
template <class N, class ST>
__device__ __forceinline__ float sdf_only(ST& sg, const h8* __restrict__ Wf, const float* __restrict__ T, const AvcOffsets& o,
                                          int h, const float (&x)[3]) {
  PE pe;
  pe_compute(x, h, pe);
  float part = 0.f;
  {
    const float* wpe = T + o.v[OFF_WL0_PE] + h * 24;
#pragma unroll
    for (int q = 0; q < 24; ++q) part += wpe[q] * pe.v[q];
  }
  h8 hlast[N::HK];
  {
    h8 h1[N::HK];
    {
      h8 pef[3];
      pe_to_frags_f16(pe, x, h, pef);
      layer_s<h8, 3, N::HT>(sg, Wf, o.v[OFF_W0], nxt<N, OFF_WM0>(sg, Wf, o), pef, AVC_EPI(
        float b[16], a[16];
        load16(T + o.v[OFF_B0], t, h, b);
        _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);
        acc_to_frags(a, h1[2 * t], h1[2 * t + 1]);
      ));
    }
    if constexpr (N::NMID == 2) {
      h8 hm0[N::HK];
      layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wf, o), h1, AVC_EPI(
        float b[16], a[16];
        load16(T + o.v[OFF_BM0], t, h, b);
        _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);
        acc_to_frags(a, hm0[2 * t], hm0[2 * t + 1]);
      ));
      layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wf, o), hm0, AVC_EPI(
        float b[16], a[16];
        load16(T + o.v[OFF_BM1], t, h, b);
        _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);
        acc_to_frags(a, hlast[2 * t], hlast[2 * t + 1]);
      ));
    } else {
      layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wf, o), h1, AVC_EPI(
        float b[16], a[16];
        load16(T + o.v[OFF_BM0], t, h, b);
        _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);
        acc_to_frags(a, hlast[2 * t], hlast[2 * t + 1]);
      ));
    }
  }
  layer_s<h8, N::HK, N::ST>(sg, Wf, o.v[OFF_WS], no_next(), hlast, AVC_EPI(
    float b[16], w[16];
    load16(T + o.v[OFF_BS], t, h, b);
    load16(T + o.v[OFF_WL0_ACC], t, h, w);
    _Pragma("unroll") for (int r = 0; r < 16; ++r) part += w[r] * softplus2(acc[r] + b[r]);
  ));
  return xhalf_sum(part) + T[o.v[OFF_BL0]];
}

// feature = rows 1..H of the last layer (u = [h_skip ; pe]/sqrt2 folded into the packed weights)
template <class N, class ST>
__device__ __forceinline__ void sdf_feature(ST& sg, const h8* __restrict__ Wf, const float* __restrict__ T, const AvcOffsets& o,
                                            int h, const FwdState<N>& st, h8 (&feat)[N::HK], const Next& gnext) {
  layer2_s<h8, N::SK, 3, N::HT>(sg, Wf, o.v[OFF_WL], gnext, st.hs, st.pef, AVC_EPI(
    float b[16], a[16];
    load16(T + o.v[OFF_BL], t, h, b);
    _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = acc[r] + b[r];
    acc_to_frags(a, feat[2 * t], feat[2 * t + 1]);
  ));
}

// Normal n = d sdf / d x by the reverse sweep.
template <class N, class ST>
__device__ __forceinline__ void sdf_normal(ST& sg, const h8* __restrict__ Wf, const float* __restrict__ T, const AvcOffsets& o,
                                           int h, const FwdState<N>& st, float (&n)[3], const Next& gnext) {
  float w8[8];
  h8 g_in_s[N::SK];
#pragma unroll
  for (int s = 0; s < N::SK; ++s) {
    load8(T + o.v[OFF_WL0_FRAG], s, h, w8);
#pragma unroll
    for (int j = 0; j < 8; ++j) g_in_s[s][j] = (_Float16)(w8[j] * sig_from_h((float)st.hs[s][j]));
  }
  h8 g[N::HK];
  layer_s<h8, N::SK, N::HT>(sg, Wf, o.v[OFF_WST], (N::NMID == 2 ? nxt<N, OFF_WM1T>(sg, Wf, o) : nxt<N, OFF_WM0T>(sg, Wf, o)),
                                   g_in_s, AVC_EPI(
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {
      g[2 * t][j] = (_Float16)(acc[j] * sig_from_h((float)st.hm[N::NMID - 1][2 * t][j]));
      g[2 * t + 1][j] = (_Float16)(acc[8 + j] * sig_from_h((float)st.hm[N::NMID - 1][2 * t + 1][j]));
    }
    pin2(g[2 * t], g[2 * t + 1]);
  ));
  h8 g2[N::HK];
  if constexpr (N::NMID == 2) {
    layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1T], nxt<N, OFF_WM0T>(sg, Wf, o), g, AVC_EPI(
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {
        g2[2 * t][j] = (_Float16)(acc[j] * sig_from_h((float)st.hm[0][2 * t][j]));
        g2[2 * t + 1][j] = (_Float16)(acc[8 + j] * sig_from_h((float)st.hm[0][2 * t + 1][j]));
      }
      pin2(g2[2 * t], g2[2 * t + 1]);
    ));
    layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0T], nxt<N, OFF_W0T>(sg, Wf, o), g2, AVC_EPI(
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {
        g[2 * t][j] = (_Float16)(acc[j] * sig_from_h((float)st.h1[2 * t][j]));
        g[2 * t + 1][j] = (_Float16)(acc[8 + j] * sig_from_h((float)st.h1[2 * t + 1][j]));
      }
      pin2(g[2 * t], g[2 * t + 1]);
    ));
  } else {
    layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0T], nxt<N, OFF_W0T>(sg, Wf, o), g, AVC_EPI(
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {
        g2[2 * t][j] = (_Float16)(acc[j] * sig_from_h((float)st.h1[2 * t][j]));
        g2[2 * t + 1][j] = (_Float16)(acc[8 + j] * sig_from_h((float)st.h1[2 * t + 1][j]));
      }
      pin2(g2[2 * t], g2[2 * t + 1]);
    ));
#pragma unroll
    for (int s = 0; s < N::HK; ++s) g[s] = g2[s];
  }
  float part[3] = {0.f, 0.f, 0.f};
  const float* wpe = T + o.v[OFF_WL0_PE] + h * 24;
  layer_s<h8, N::HK, 2>(sg, Wf, o.v[OFF_W0T], gnext, g, AVC_EPI(
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {
      const int q = 16 * t + r;
      if (q < 24) part[q % 3] += st.pe.d[q] * (acc[r] + wpe[q]);
    }
  ));
#pragma unroll
  for (int c = 0; c < 3; ++c) n[c] = xhalf_sum(part[c]);
}

// colour MLP: r0 = [x, n, feature] -> ... -> sigmoid([rgb_prior ; rgb_clip])  (6 outputs: half 0 holds 0..3, half 1 holds 4,5)
template <class N, class ST>
__device__ __forceinline__ void color_forward(ST& sg, const h8* __restrict__ Wf, const float* __restrict__ T, const AvcOffsets& o,
                                              int h, const float (&x)[3], const float (&n)[3], const h8 (&feat)[N::HK],
                                              float (&rgb)[4]) {
  h8 xn[1];
#pragma unroll
  for (int j = 0; j < 8; ++j) xn[0][j] = (_Float16)0.f;
  if (h == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { xn[0][c] = (_Float16)x[c]; xn[0][3 + c] = (_Float16)n[c]; }
  }
  h8 r1[N::HK];
  layer2_s<h8, N::HK, 1, N::HT>(sg, Wf, o.v[OFF_C0], (N::NCMID == 1 ? nxt<N, OFF_CM0>(sg, Wf, o) : nxt<N, OFF_CH>(sg, Wf, o)),
                                       feat, xn, AVC_EPI(
    float b[16], a[16];
    load16(T + o.v[OFF_CB0], t, h, b);
    _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = fmaxf(acc[r] + b[r], 0.f);
    acc_to_frags(a, r1[2 * t], r1[2 * t + 1]);
  ));
  h8 r2[N::HK];
  if constexpr (N::NCMID == 1) {
    layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_CM0], nxt<N, OFF_CH>(sg, Wf, o), r1, AVC_EPI(
      float b[16], a[16];
      load16(T + o.v[OFF_CBM0], t, h, b);
      _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = fmaxf(acc[r] + b[r], 0.f);
      acc_to_frags(a, r2[2 * t], r2[2 * t + 1]);
    ));
  } else {
#pragma unroll
    for (int s = 0; s < N::HK; ++s) r2[s] = r1[s];
  }
  layer_s<h8, N::HK, 1>(sg, Wf, o.v[OFF_CH], no_next(), r2, AVC_EPI(
    float b[16];
    load16(T + o.v[OFF_CBH], 0, h, b);
    _Pragma("unroll") for (int r = 0; r < 4; ++r) rgb[r] = sigmoidf_(acc[r] + b[r]);
  ));
}
