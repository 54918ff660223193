// Register-resident SDF + colour MLP forward for one wavefront (32 points).  See avc_common.h for the layout.
// Follows AvatarGen/AppearanceGen/models/fields.py:72-107 (SDFNetwork.forward/.gradient) and :154-185
// (RenderingNetwork.forward, mode 'no_view_dir', extra_color) of the reference.
#pragma once
#include "avc_common.h"

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

// out = act(W in + b) for an H-wide (NT tiles) layer; ACT 0 none, 1 softplus(beta=100), 2 relu
template <typename V, int KS, int NT, int ACT>
__device__ __forceinline__ void dense_layer(const V* __restrict__ blob, int offw, const float* __restrict__ bias,
                                            int lane, int h, const V (&in)[KS], V (&out)[2 * NT]) {
  float b[16], a[16];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    facc acc = tile_gemm<V, KS>(tptr<V, KS>(blob, offw, t, lane), in);
    load16(bias, t, h, b);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[r] + b[r];
      a[r] = ACT == 1 ? softplus100(v) : (ACT == 2 ? fmaxf(v, 0.f) : v);
    }
    acc_to_frags(a, out[2 * t], out[2 * t + 1]);
  }
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
template <class N>
__device__ __forceinline__ void sdf_trunk(const h8* __restrict__ Wf, const float* __restrict__ T, const AvcOffsets& o,
                                          int lane, int h, FwdState<N>& st) {
  pe_compute(st.x, h, st.pe);
  pe_to_frags_f16(st.pe, st.x, h, st.pef);
  float b[16], a[16];
  dense_layer<h8, 3, N::HT, 1>(Wf, o.v[OFF_W0], T + o.v[OFF_B0], lane, h, st.pef, st.h1);
  dense_layer<h8, N::HK, N::HT, 1>(Wf, o.v[OFF_WM0], T + o.v[OFF_BM0], lane, h, st.h1, st.hm[0]);
  if constexpr (N::NMID == 2)
    dense_layer<h8, N::HK, N::HT, 1>(Wf, o.v[OFF_WM1], T + o.v[OFF_BM1], lane, h, st.hm[0], st.hm[1]);
  // skip layer H -> SKIP, with the fp32 sdf dot product folded into its epilogue
  float part = 0.f;
#pragma unroll
  for (int t = 0; t < N::ST; ++t) {
    facc acc = tile_gemm<h8, N::HK>(tptr<h8, N::HK>(Wf, o.v[OFF_WS], t, lane), st.hm[N::NMID - 1]);
    load16(T + o.v[OFF_BS], t, h, b);
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = softplus100(acc[r] + b[r]);
    load16(T + o.v[OFF_WL0_ACC], t, h, b);
#pragma unroll
    for (int r = 0; r < 16; ++r) part += b[r] * a[r];
    acc_to_frags(a, st.hs[2 * t], st.hs[2 * t + 1]);
  }
  {
    const float* wpe = T + o.v[OFF_WL0_PE] + h * 24;
#pragma unroll
    for (int q = 0; q < 24; ++q) part += wpe[q] * st.pe.v[q];
  }
  st.sdf = xhalf_sum(part) + T[o.v[OFF_BL0]];
}

// feature = rows 1..H of the last layer (u = [h_skip ; pe]/sqrt2 folded into the packed weights)
template <class N>
__device__ __forceinline__ void sdf_feature(const h8* __restrict__ Wf, const float* __restrict__ T, const AvcOffsets& o,
                                            int lane, int h, const FwdState<N>& st, h8 (&feat)[N::HK]) {
  float b[16], a[16];
#pragma unroll
  for (int t = 0; t < N::HT; ++t) {
    facc acc = tile_gemm2<h8, N::SK, 3>(tptr<h8, N::SK + 3>(Wf, o.v[OFF_WL], t, lane), st.hs, st.pef);
    load16(T + o.v[OFF_BL], t, h, b);
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = acc[r] + b[r];
    acc_to_frags(a, feat[2 * t], feat[2 * t + 1]);
  }
}

// Normal n = d sdf / d x by the reverse sweep.  When G != nullptr-like (KEEP), the per-layer
// g_h (gradient wrt the post-activation h_l) are also returned for the double-backward.
template <class N, typename V, bool KEEP>
struct NormalSweep {
  V ga_s[N::SK];               // g_a of the skip layer output
  V ga_m[N::NMID][N::HK];      // g_a of middle layer outputs (index m -> layer m+1's output h_{m+2})
  V ga_1[N::HK];               // g_a of layer0's output
};

template <class N>
__device__ __forceinline__ void sdf_normal(const h8* __restrict__ Wf, const float* __restrict__ T, const AvcOffsets& o,
                                           int lane, int h, const FwdState<N>& st, float (&n)[3]) {
  float w8[8];
  h8 g_in_s[N::SK];
#pragma unroll
  for (int s = 0; s < N::SK; ++s) {
    load8(T + o.v[OFF_WL0_FRAG], s, h, w8);
#pragma unroll
    for (int j = 0; j < 8; ++j) g_in_s[s][j] = (_Float16)(w8[j] * sig_from_h((float)st.hs[s][j]));
  }
  h8 g[N::HK];
  // through the skip layer (transposed): rows = H features of h_{last middle}
#pragma unroll
  for (int t = 0; t < N::HT; ++t) {
    facc acc = tile_gemm<h8, N::SK>(tptr<h8, N::SK>(Wf, o.v[OFF_WST], t, lane), g_in_s);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      g[2 * t][j] = (_Float16)(acc[j] * sig_from_h((float)st.hm[N::NMID - 1][2 * t][j]));
      g[2 * t + 1][j] = (_Float16)(acc[8 + j] * sig_from_h((float)st.hm[N::NMID - 1][2 * t + 1][j]));
    }
  }
  // through the middle layers in reverse
#pragma unroll
  for (int m = N::NMID - 1; m >= 0; --m) {
    const int offw = (m == 0) ? o.v[OFF_WM0T] : o.v[OFF_WM1T];
    h8 g2[N::HK];
#pragma unroll
    for (int t = 0; t < N::HT; ++t) {
      facc acc = tile_gemm<h8, N::HK>(tptr<h8, N::HK>(Wf, offw, t, lane), g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float s0 = (m == 0) ? (float)st.h1[2 * t][j] : (float)st.hm[m > 0 ? m - 1 : 0][2 * t][j];
        const float s1 = (m == 0) ? (float)st.h1[2 * t + 1][j] : (float)st.hm[m > 0 ? m - 1 : 0][2 * t + 1][j];
        g2[2 * t][j] = (_Float16)(acc[j] * sig_from_h(s0));
        g2[2 * t + 1][j] = (_Float16)(acc[8 + j] * sig_from_h(s1));
      }
    }
#pragma unroll
    for (int s = 0; s < N::HK; ++s) g[s] = g2[s];
  }
  // through layer 0 (transposed): rows = pe slots, two tiles of 16 slots per half
  float part[3] = {0.f, 0.f, 0.f};
  const float* wpe = T + o.v[OFF_WL0_PE] + h * 24;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    facc acc = tile_gemm<h8, N::HK>(tptr<h8, N::HK>(Wf, o.v[OFF_W0T], t, lane), g);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = 16 * t + r;
      if (q < 24) part[q % 3] += st.pe.d[q] * (acc[r] + wpe[q]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) n[c] = xhalf_sum(part[c]);
}

// colour MLP: r0 = [x, n, feature] -> ... -> sigmoid([rgb_prior ; rgb_clip])  (6 outputs: half 0 holds 0..3, half 1 holds 4,5)
template <class N>
__device__ __forceinline__ void color_forward(const h8* __restrict__ Wf, const float* __restrict__ T, const AvcOffsets& o,
                                              int lane, int h, const float (&x)[3], const float (&n)[3],
                                              const h8 (&feat)[N::HK], float (&rgb)[4]) {
  h8 xn[1];
#pragma unroll
  for (int j = 0; j < 8; ++j) xn[0][j] = (_Float16)0.f;
  if (h == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { xn[0][c] = (_Float16)x[c]; xn[0][3 + c] = (_Float16)n[c]; }
  }
  float b[16], a[16];
  h8 r1[N::HK];
#pragma unroll
  for (int t = 0; t < N::HT; ++t) {
    facc acc = tile_gemm2<h8, N::HK, 1>(tptr<h8, N::HK + 1>(Wf, o.v[OFF_C0], t, lane), feat, xn);
    load16(T + o.v[OFF_CB0], t, h, b);
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(acc[r] + b[r], 0.f);
    acc_to_frags(a, r1[2 * t], r1[2 * t + 1]);
  }
#pragma unroll
  for (int m = 0; m < N::NCMID; ++m) {
    h8 r2[N::HK];
#pragma unroll
    for (int t = 0; t < N::HT; ++t) {
      facc acc = tile_gemm<h8, N::HK>(tptr<h8, N::HK>(Wf, o.v[OFF_CM0], t, lane), r1);
      load16(T + o.v[OFF_CBM0], t, h, b);
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = fmaxf(acc[r] + b[r], 0.f);
      acc_to_frags(a, r2[2 * t], r2[2 * t + 1]);
    }
#pragma unroll
    for (int s = 0; s < N::HK; ++s) r1[s] = r2[s];
  }
  facc acc = tile_gemm<h8, N::HK>(tptr<h8, N::HK>(Wf, o.v[OFF_CH], 0, lane), r1);
  load16(T + o.v[OFF_CBH], 0, h, b);
#pragma unroll
  for (int r = 0; r < 4; ++r) rgb[r] = sigmoidf_(acc[r] + b[r]);
}
