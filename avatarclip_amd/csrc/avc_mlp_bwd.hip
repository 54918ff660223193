// Backward of the fused SDF + colour MLP wrt every dense weight, incl. the double backward through
// SDFNetwork.gradient (fields.py:96-107; autograd at main.py:537).  Mathematics: SURVEY.md A.1/A.2, proven
// against torch.autograd in tests/test_analytic.py (oracle/analytic.py: mlp_backward).
//
//   avc_render_points_bwd : one wavefront per 32 points.  Recomputes the forward (f16), runs the normal sweep,
//        the colour backward, the second-order sweep (i) and the reverse sweep (ii) (bf16 operands, fp32 acc),
//        and writes every operand of every weight-gradient product as a TRANSPOSED bf16 panel
//        (feature-major: lane = feature, 16 points per lane) -- the transposition runs on the matrix core
//        (two MFMAs against a 0/1 selection fragment per 32x32 block), not through LDS.
//   avc_weight_grad       : dW[a,b] += sum_points A[p,a] B[p,b], K = points, straight from the panels.
//
// Activations that a later phase needs again are parked in a per-wavefront scratch slot (frag layout, L2/MALL
// resident because the slot is reused for every block the wave processes).
#include "avc_mlp.h"
#include "../../include/avc.h"

// ---------------------------------------------------------------------------------------------
// panel / scratch bookkeeping (mirrored by packing.py: panel_layout / scratch_layout)
// ---------------------------------------------------------------------------------------------
template <class N>
struct BwdLayout {
  static constexpr int HT = N::HT, ST = N::ST, NM = N::NMID, NC = N::NCMID;
  // panel tile offsets (in 32-feature tiles) inside one 32-point block
  static constexpr int P_H0 = 0;                    // pe values (2 tiles)
  static constexpr int P_GB0 = P_H0 + 2;            // gbar_h0 (2)
  static constexpr int P_H1 = P_GB0 + 2;            // h1
  static constexpr int P_HM = P_H1 + HT;            // hm[NM]
  static constexpr int P_HS = P_HM + NM * HT;       // hs (ST)
  static constexpr int P_GBH1 = P_HS + ST;          // gbar_h1
  static constexpr int P_GBHM = P_GBH1 + HT;        // gbar_hm[NM]
  static constexpr int P_GBHS = P_GBHM + NM * HT;   // gbar_hs (ST)
  static constexpr int P_GA1 = P_GBHS + ST;         // g_a1
  static constexpr int P_GAM = P_GA1 + HT;          // g_am[NM]
  static constexpr int P_GAS = P_GAM + NM * HT;     // g_as (ST)
  static constexpr int P_AB1 = P_GAS + ST;          // abar_1
  static constexpr int P_ABM = P_AB1 + HT;          // abar_m[NM]
  static constexpr int P_ABS = P_ABM + NM * HT;     // abar_s (ST)
  static constexpr int P_DFEAT = P_ABS + ST;        // ybar[1:] (HT)
  static constexpr int P_SDF = P_DFEAT + HT;        // feature 0 = d_sdf (1)
  static constexpr int P_ONE = P_SDF + 1;           // feature 0 = 1 (1)
  static constexpr int P_FEAT = P_ONE + 1;          // feature (HT)
  static constexpr int P_XN = P_FEAT + HT;          // [x, n] (1)
  static constexpr int P_R1 = P_XN + 1;             // r1 (HT)
  static constexpr int P_R2 = P_R1 + HT;            // r2 (HT, only NC==1)
  static constexpr int P_D1 = P_R2 + NC * HT;       // delta1 (HT)
  static constexpr int P_D2 = P_D1 + HT;            // delta2 (HT, only NC==1)
  static constexpr int P_DO = P_D2 + NC * HT;       // delta_o (1)
  static constexpr int P_TILES = P_DO + 1;
  // scratch k-step offsets inside one wavefront slot (16-byte chunks x 64 lanes per k-step)
  static constexpr int S_H1 = 0;
  static constexpr int S_HM = S_H1 + N::HK;
  static constexpr int S_HS = S_HM + NM * N::HK;
  static constexpr int S_Q1 = S_HS + N::SK;
  static constexpr int S_QM = S_Q1 + N::HK;
  static constexpr int S_QS = S_QM + NM * N::HK;
  static constexpr int S_AP1 = S_QS + N::SK;
  static constexpr int S_APM = S_AP1 + N::HK;
  static constexpr int S_R1 = S_APM + NM * N::HK;
  static constexpr int S_R2 = S_R1 + N::HK;
  static constexpr int S_KSTEPS = S_R2 + NC * N::HK;
};

extern "C" int avc_bwd_panel_tiles(int net) {
  return net == AVC_NET_FULL ? BwdLayout<NetFull>::P_TILES : BwdLayout<NetSmall>::P_TILES;
}
extern "C" long avc_bwd_scratch_bytes_per_wave(int net) {
  return (long)(net == AVC_NET_FULL ? BwdLayout<NetFull>::S_KSTEPS : BwdLayout<NetSmall>::S_KSTEPS) * 64 * 16;
}

template <typename P> __device__ __forceinline__ P launder(P p) {
  asm volatile("" : "+s"(p));
  return p;
}

// selection fragments of the MFMA transposition: lane (n,h) of k-step-half e: 1 where feature slot (h,j) == n
template <typename V>
__device__ __forceinline__ void make_sel(int lane, V& e0, V& e1) {
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f = 8 * (j >> 2) + 4 * h + (j & 3);
    e0[j] = (typename MF<V>::S)(n == f ? 1.f : 0.f);
    e1[j] = (typename MF<V>::S)(n == 16 + f ? 1.f : 0.f);
  }
}

// transpose the two k-steps (f0,f1) of a 32-feature tile to feature-major and store it as bf16 panel tile
template <typename V>
__device__ __forceinline__ void panel_store(b8* __restrict__ panel_blk, int tile, int lane, const V& f0, const V& f1,
                                            const V& e0, const V& e1) {
  facc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = MF<V>::mma(f0, e0, acc);
  acc = MF<V>::mma(f1, e1, acc);
  b8 k0, k1;
#pragma unroll
  for (int j = 0; j < 8; ++j) { k0[j] = (__bf16)acc[j]; k1[j] = (__bf16)acc[8 + j]; }
  b8* dst = panel_blk + (long)tile * 128 + lane;
  dst[0] = k0;
  dst[64] = k1;
}
template <typename V>
__device__ __forceinline__ V zero_frag() {
  V z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (typename MF<V>::S)0.f;
  return z;
}

template <typename V> __device__ __forceinline__ void scr_store(V* scr, int ks, int lane, const V& v) { scr[ks * 64 + lane] = v; }
template <typename V> __device__ __forceinline__ V scr_load(const V* scr, int ks, int lane) { return scr[ks * 64 + lane]; }

// f16 forward layer that also parks its output in scratch and in a panel
template <class N, int KS, int NT>
__device__ __forceinline__ void fwd_layer_keep(const h8* __restrict__ Wf, int offw, const float* __restrict__ bias, int lane,
                                               int h, const h8 (&in)[KS], h8 (&out)[2 * NT], h8* scr, int scr_ks,
                                               b8* panel_blk, int ptile, const h8& e0, const h8& e1) {
  float b[16], a[16];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    facc acc = tile_gemm<h8, KS>(tptr<h8, KS>(Wf, offw, t, lane), in);
    load16(bias, t, h, b);
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = softplus100(acc[r] + b[r]);
    acc_to_frags(a, out[2 * t], out[2 * t + 1]);
    scr_store(scr, scr_ks + 2 * t, lane, out[2 * t]);
    scr_store(scr, scr_ks + 2 * t + 1, lane, out[2 * t + 1]);
    panel_store<h8>(panel_blk, ptile + t, lane, out[2 * t], out[2 * t + 1], e0, e1);
  }
}

// one step of the normal sweep: g_h(prev) = W^T g_a(cur); then g_a(prev) = g_h ⊙ σ(h_prev), q = g_h ⊙ sp''(h_prev)
template <class N, int KS, int NT>
__device__ __forceinline__ void normal_step(const h8* __restrict__ Wf, int offw, int lane, const h8 (&gin)[KS],
                                            h8 (&gout)[2 * NT], h8* scr, int scr_h, int scr_q, b8* panel_blk, int ptile_ga,
                                            const h8& e0, const h8& e1) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    facc acc = tile_gemm<h8, KS>(tptr<h8, KS>(Wf, offw, t, lane), gin);
    const h8 hv0 = scr_load(scr, scr_h + 2 * t, lane), hv1 = scr_load(scr, scr_h + 2 * t + 1, lane);
    h8 q0, q1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s0 = sig_from_h((float)hv0[j]), s1 = sig_from_h((float)hv1[j]);
      gout[2 * t][j] = (_Float16)(acc[j] * s0);
      gout[2 * t + 1][j] = (_Float16)(acc[8 + j] * s1);
      q0[j] = (_Float16)(acc[j] * AVC_BETA * s0 * (1.f - s0) * (1.f / 64.f));   // scaled: keeps beta*g in f16 range
      q1[j] = (_Float16)(acc[8 + j] * AVC_BETA * s1 * (1.f - s1) * (1.f / 64.f));
    }
    scr_store(scr, scr_q + 2 * t, lane, q0);
    scr_store(scr, scr_q + 2 * t + 1, lane, q1);
    panel_store<h8>(panel_blk, ptile_ga + t, lane, gout[2 * t], gout[2 * t + 1], e0, e1);
  }
}

// one layer of the second-order sweep (i): gbar_a = W gbar_h(in); abar' = gbar_a ⊙ q ; gbar_h(out) = gbar_a ⊙ σ(h_out)
template <class N, int KS, int NT, bool KEEP_AP_REGS>
__device__ __forceinline__ void second_step(const b8* __restrict__ Wb, int offw, int lane, const b8 (&gin)[KS],
                                            b8 (&gout)[2 * NT], h8* scr, int scr_h, int scr_q, int scr_ap, b8 (&ap)[2 * NT],
                                            b8* panel_blk, int ptile_gbh, const b8& e0, const b8& e1) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    facc acc = tile_gemm<b8, KS>(tptr<b8, KS>(Wb, offw, t, lane), gin);
    const h8 hv0 = scr_load(scr, scr_h + 2 * t, lane), hv1 = scr_load(scr, scr_h + 2 * t + 1, lane);
    const h8 q0 = scr_load(scr, scr_q + 2 * t, lane), q1 = scr_load(scr, scr_q + 2 * t + 1, lane);
    b8 a0, a1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gout[2 * t][j] = (__bf16)(acc[j] * sig_from_h((float)hv0[j]));
      gout[2 * t + 1][j] = (__bf16)(acc[8 + j] * sig_from_h((float)hv1[j]));
      a0[j] = (__bf16)(acc[j] * (float)q0[j] * 64.f);
      a1[j] = (__bf16)(acc[8 + j] * (float)q1[j] * 64.f);
    }
    if (KEEP_AP_REGS) { ap[2 * t] = a0; ap[2 * t + 1] = a1; }
    else {
      scr_store(reinterpret_cast<b8*>(scr), scr_ap + 2 * t, lane, a0);
      scr_store(reinterpret_cast<b8*>(scr), scr_ap + 2 * t + 1, lane, a1);
    }
    panel_store<b8>(panel_blk, ptile_gbh + t, lane, gout[2 * t], gout[2 * t + 1], e0, e1);
  }
}

// one step of the reverse sweep (ii): hbar(prev) = W^T abar(cur); abar(prev) = abar'(prev) + hbar ⊙ σ(h_prev)
template <class N, int KS, int NT>
__device__ __forceinline__ void reverse_step(const b8* __restrict__ Wb, int offw, int lane, const b8 (&ain)[KS],
                                             b8 (&aout)[2 * NT], h8* scr, int scr_h, int scr_ap, b8* panel_blk, int ptile_ab,
                                             const b8& e0, const b8& e1) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    facc acc = tile_gemm<b8, KS>(tptr<b8, KS>(Wb, offw, t, lane), ain);
    const h8 hv0 = scr_load(scr, scr_h + 2 * t, lane), hv1 = scr_load(scr, scr_h + 2 * t + 1, lane);
    const b8 p0 = scr_load(reinterpret_cast<const b8*>(scr), scr_ap + 2 * t, lane);
    const b8 p1 = scr_load(reinterpret_cast<const b8*>(scr), scr_ap + 2 * t + 1, lane);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      aout[2 * t][j] = (__bf16)((float)p0[j] + acc[j] * sig_from_h((float)hv0[j]));
      aout[2 * t + 1][j] = (__bf16)((float)p1[j] + acc[8 + j] * sig_from_h((float)hv1[j]));
    }
    panel_store<b8>(panel_blk, ptile_ab + t, lane, aout[2 * t], aout[2 * t + 1], e0, e1);
  }
}

template <class N>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(PointSrc ps, long npts, const h8* __restrict__ Wf0,
                                                      const b8* __restrict__ Wb0, const float* __restrict__ T0, AvcOffsets o,
                                                      const float* __restrict__ d_sdf, const float* __restrict__ d_normal,
                                                      const float* __restrict__ d_rgb, b8* __restrict__ panels,
                                                      char* __restrict__ scratch) {
  typedef BwdLayout<N> L;
  const int lane = threadIdx.x & 63, h = lane >> 5, p = lane & 31;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  const long nblk = (npts + 31) >> 5;
  h8* scr = reinterpret_cast<h8*>(scratch + wave * (long)L::S_KSTEPS * 64 * 16);
  h8 e0h, e1h; b8 e0b, e1b;
  make_sel<h8>(lane, e0h, e1h);
  make_sel<b8>(lane, e0b, e1b);

  for (long blk = wave; blk < nblk; blk += nwaves) {
    // opaque per-iteration copies of the parameter pointers: keeps LICM from hoisting ~3000 weight loads
    const h8* Wf = launder(Wf0);
    const b8* Wb = launder(Wb0);
    const float* T = launder(T0);
    b8* pblk = panels + blk * (long)L::P_TILES * 128;
    long i = blk * 32 + p;
    const bool valid = i < npts;
    if (!valid) i = npts - 1;
    const float vmask = valid ? 1.f : 0.f;

    // ------------------------------------------------------------------ phase A: forward recompute (f16)
    float x[3];
    fetch_point(ps, i, x);
    PE pe;
    pe_compute(x, h, pe);
    h8 pef[3];
    pe_to_frags_f16(pe, x, h, pef);
    panel_store<h8>(pblk, L::P_H0, lane, pef[0], pef[1], e0h, e1h);
    panel_store<h8>(pblk, L::P_H0 + 1, lane, pef[2], zero_frag<h8>(), e0h, e1h);
    h8 hs[N::SK];
    {
      h8 h1[N::HK];
      fwd_layer_keep<N, 3, N::HT>(Wf, o.v[OFF_W0], T + o.v[OFF_B0], lane, h, pef, h1, scr, L::S_H1, pblk, L::P_H1, e0h, e1h);
      h8 hm0[N::HK];
      fwd_layer_keep<N, N::HK, N::HT>(Wf, o.v[OFF_WM0], T + o.v[OFF_BM0], lane, h, h1, hm0, scr, L::S_HM, pblk, L::P_HM, e0h, e1h);
      if constexpr (N::NMID == 2) {
        h8 hm1[N::HK];
        fwd_layer_keep<N, N::HK, N::HT>(Wf, o.v[OFF_WM1], T + o.v[OFF_BM1], lane, h, hm0, hm1, scr, L::S_HM + N::HK, pblk,
                                        L::P_HM + N::HT, e0h, e1h);
        fwd_layer_keep<N, N::HK, N::ST>(Wf, o.v[OFF_WS], T + o.v[OFF_BS], lane, h, hm1, hs, scr, L::S_HS, pblk, L::P_HS, e0h, e1h);
      } else {
        fwd_layer_keep<N, N::HK, N::ST>(Wf, o.v[OFF_WS], T + o.v[OFF_BS], lane, h, hm0, hs, scr, L::S_HS, pblk, L::P_HS, e0h, e1h);
      }
    }
    // ------------------------------------------------------------------ phase B: normal sweep (f16)
    float n[3];
    {
      h8 g_s[N::SK];
      float w8[8];
#pragma unroll
      for (int s = 0; s < N::SK; ++s) {
        load8(T + o.v[OFF_WL0_FRAG], s, h, w8);
        h8 q;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float sg = sig_from_h((float)hs[s][j]);
          g_s[s][j] = (_Float16)(w8[j] * sg);
          q[j] = (_Float16)(w8[j] * AVC_BETA * sg * (1.f - sg) * (1.f / 64.f));
        }
        scr_store(scr, L::S_QS + s, lane, q);
      }
#pragma unroll
      for (int t = 0; t < N::ST; ++t) panel_store<h8>(pblk, L::P_GAS + t, lane, g_s[2 * t], g_s[2 * t + 1], e0h, e1h);
      h8 g[N::HK];
      normal_step<N, N::SK, N::HT>(Wf, o.v[OFF_WST], lane, g_s, g, scr, L::S_HM + (N::NMID - 1) * N::HK,
                                   L::S_QM + (N::NMID - 1) * N::HK, pblk, L::P_GAM + (N::NMID - 1) * N::HT, e0h, e1h);
      if constexpr (N::NMID == 2) {
        h8 g2[N::HK];
        normal_step<N, N::HK, N::HT>(Wf, o.v[OFF_WM1T], lane, g, g2, scr, L::S_HM, L::S_QM, pblk, L::P_GAM, e0h, e1h);
        normal_step<N, N::HK, N::HT>(Wf, o.v[OFF_WM0T], lane, g2, g, scr, L::S_H1, L::S_Q1, pblk, L::P_GA1, e0h, e1h);
      } else {
        h8 g2[N::HK];
        normal_step<N, N::HK, N::HT>(Wf, o.v[OFF_WM0T], lane, g, g2, scr, L::S_H1, L::S_Q1, pblk, L::P_GA1, e0h, e1h);
#pragma unroll
        for (int s = 0; s < N::HK; ++s) g[s] = g2[s];
      }
      float part[3] = {0.f, 0.f, 0.f};
      const float* wpe = T + o.v[OFF_WL0_PE] + h * 24;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        facc acc = tile_gemm<h8, N::HK>(tptr<h8, N::HK>(Wf, o.v[OFF_W0T], t, lane), g);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = 16 * t + r;
          if (q < 24) part[q % 3] += pe.d[q] * (acc[r] + wpe[q]);
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) n[c] = xhalf_sum(part[c]);
    }
    // ------------------------------------------------------------------ phase C: colour forward (f16)
    float delta_o[4];   // half 0: outputs 0..3, half 1: outputs 4,5 (delta = d_rgb * rgb (1-rgb))
    {
      h8 feat[N::HK];
      {
        float b[16], a[16];
#pragma unroll
        for (int t = 0; t < N::HT; ++t) {
          facc acc = tile_gemm2<h8, N::SK, 3>(tptr<h8, N::SK + 3>(Wf, o.v[OFF_WL], t, lane), hs, pef);
          load16(T + o.v[OFF_BL], t, h, b);
#pragma unroll
          for (int r = 0; r < 16; ++r) a[r] = acc[r] + b[r];
          acc_to_frags(a, feat[2 * t], feat[2 * t + 1]);
          panel_store<h8>(pblk, L::P_FEAT + t, lane, feat[2 * t], feat[2 * t + 1], e0h, e1h);
        }
      }
      h8 xn[1];
      xn[0] = zero_frag<h8>();
      if (h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { xn[0][c] = (_Float16)x[c]; xn[0][3 + c] = (_Float16)n[c]; }
      }
      panel_store<h8>(pblk, L::P_XN, lane, xn[0], zero_frag<h8>(), e0h, e1h);
      float b[16], a[16];
      h8 r1[N::HK];
#pragma unroll
      for (int t = 0; t < N::HT; ++t) {
        facc acc = tile_gemm2<h8, N::HK, 1>(tptr<h8, N::HK + 1>(Wf, o.v[OFF_C0], t, lane), feat, xn);
        load16(T + o.v[OFF_CB0], t, h, b);
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = fmaxf(acc[r] + b[r], 0.f);
        acc_to_frags(a, r1[2 * t], r1[2 * t + 1]);
        scr_store(scr, L::S_R1 + 2 * t, lane, r1[2 * t]);
        scr_store(scr, L::S_R1 + 2 * t + 1, lane, r1[2 * t + 1]);
        panel_store<h8>(pblk, L::P_R1 + t, lane, r1[2 * t], r1[2 * t + 1], e0h, e1h);
      }
      facc acco;
      if constexpr (N::NCMID == 1) {
        h8 r2[N::HK];
#pragma unroll
        for (int t = 0; t < N::HT; ++t) {
          facc acc = tile_gemm<h8, N::HK>(tptr<h8, N::HK>(Wf, o.v[OFF_CM0], t, lane), r1);
          load16(T + o.v[OFF_CBM0], t, h, b);
#pragma unroll
          for (int r = 0; r < 16; ++r) a[r] = fmaxf(acc[r] + b[r], 0.f);
          acc_to_frags(a, r2[2 * t], r2[2 * t + 1]);
          scr_store(scr, L::S_R2 + 2 * t, lane, r2[2 * t]);
          scr_store(scr, L::S_R2 + 2 * t + 1, lane, r2[2 * t + 1]);
          panel_store<h8>(pblk, L::P_R2 + t, lane, r2[2 * t], r2[2 * t + 1], e0h, e1h);
        }
        acco = tile_gemm<h8, N::HK>(tptr<h8, N::HK>(Wf, o.v[OFF_CH], 0, lane), r2);
      } else {
        acco = tile_gemm<h8, N::HK>(tptr<h8, N::HK>(Wf, o.v[OFF_CH], 0, lane), r1);
      }
      load16(T + o.v[OFF_CBH], 0, h, b);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rgb = sigmoidf_(acco[r] + b[r]);
        const int ch = h ? 4 + r : r;
        const float dr = (ch < 6) ? d_rgb[6 * i + (ch < 6 ? ch : 0)] * vmask : 0.f;
        delta_o[r] = dr * rgb * (1.f - rgb);
      }
    }
    // ------------------------------------------------------------------ phase D: colour backward (bf16)
    float nbar[3];
    b8 dfeat[N::HK];
    {
      b8 dof[1];
      dof[0] = zero_frag<b8>();
#pragma unroll
      for (int r = 0; r < 4; ++r) dof[0][r] = (__bf16)delta_o[r];
      panel_store<b8>(pblk, L::P_DO, lane, dof[0], zero_frag<b8>(), e0b, e1b);
      b8 dl[N::HK];   // delta of the last hidden colour layer
      {
        const int scr_r = (N::NCMID == 1) ? L::S_R2 : L::S_R1;
        const int pt = (N::NCMID == 1) ? L::P_D2 : L::P_D1;
#pragma unroll
        for (int t = 0; t < N::HT; ++t) {
          facc acc = tile_gemm<b8, 1>(tptr<b8, 1>(Wb, o.v[OFF_CHT], t, lane), dof);
          const h8 rv0 = scr_load(scr, scr_r + 2 * t, lane), rv1 = scr_load(scr, scr_r + 2 * t + 1, lane);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            dl[2 * t][j] = (__bf16)((float)rv0[j] > 0.f ? acc[j] : 0.f);
            dl[2 * t + 1][j] = (__bf16)((float)rv1[j] > 0.f ? acc[8 + j] : 0.f);
          }
          panel_store<b8>(pblk, pt + t, lane, dl[2 * t], dl[2 * t + 1], e0b, e1b);
        }
      }
      if constexpr (N::NCMID == 1) {
        b8 d1[N::HK];
#pragma unroll
        for (int t = 0; t < N::HT; ++t) {
          facc acc = tile_gemm<b8, N::HK>(tptr<b8, N::HK>(Wb, o.v[OFF_CM0T], t, lane), dl);
          const h8 rv0 = scr_load(scr, L::S_R1 + 2 * t, lane), rv1 = scr_load(scr, L::S_R1 + 2 * t + 1, lane);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            d1[2 * t][j] = (__bf16)((float)rv0[j] > 0.f ? acc[j] : 0.f);
            d1[2 * t + 1][j] = (__bf16)((float)rv1[j] > 0.f ? acc[8 + j] : 0.f);
          }
          panel_store<b8>(pblk, L::P_D1 + t, lane, d1[2 * t], d1[2 * t + 1], e0b, e1b);
        }
#pragma unroll
        for (int s = 0; s < N::HK; ++s) dl[s] = d1[s];
      }
      // d r0 = C0^T delta1: rows = feature (HT tiles) then the [x,n] tile
#pragma unroll
      for (int t = 0; t < N::HT; ++t) {
        facc acc = tile_gemm<b8, N::HK>(tptr<b8, N::HK>(Wb, o.v[OFF_C0T], t, lane), dl);
#pragma unroll
        for (int j = 0; j < 8; ++j) { dfeat[2 * t][j] = (__bf16)acc[j]; dfeat[2 * t + 1][j] = (__bf16)acc[8 + j]; }
        panel_store<b8>(pblk, L::P_DFEAT + t, lane, dfeat[2 * t], dfeat[2 * t + 1], e0b, e1b);
      }
      {
        facc acc = tile_gemm<b8, N::HK>(tptr<b8, N::HK>(Wb, o.v[OFF_C0T], N::HT, lane), dl);
        // rows 3,4,5 = d n : row 3 -> (h0,r3), row 4 -> (h1,r0), row 5 -> (h1,r1)
        const float a3 = acc[3], a0 = acc[0], a1 = acc[1];
        const float o3 = __shfl_xor(a3, 32), o0 = __shfl_xor(a0, 32), o1 = __shfl_xor(a1, 32);
        const float dn0 = h ? o3 : a3;
        const float dn1 = h ? a0 : o0;
        const float dn2 = h ? a1 : o1;
        nbar[0] = d_normal[3 * i + 0] * vmask + dn0;
        nbar[1] = d_normal[3 * i + 1] * vmask + dn1;
        nbar[2] = d_normal[3 * i + 2] * vmask + dn2;
      }
    }
    const float dsdf = d_sdf[i] * vmask;
    // A-panels with a single live feature: d_sdf and the constant 1 (row 0 of the last layer)
    {
      const int nf = lane & 31;
      b8 k0 = zero_frag<b8>(), k1 = zero_frag<b8>(), o0 = zero_frag<b8>(), o1 = zero_frag<b8>();
      if (nf == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pt = (r & 3) + 8 * (r >> 2) + 4 * h;
          const long ii = blk * 32 + pt;
          const float v = ii < npts ? d_sdf[ii] : 0.f;
          const float one = ii < npts ? 1.f : 0.f;
          if (r < 8) { k0[r] = (__bf16)v; o0[r] = (__bf16)one; } else { k1[r - 8] = (__bf16)v; o1[r - 8] = (__bf16)one; }
        }
      }
      b8* d1 = pblk + (long)L::P_SDF * 128 + lane; d1[0] = k0; d1[64] = k1;
      b8* d2 = pblk + (long)L::P_ONE * 128 + lane; d2[0] = o0; d2[64] = o1;
    }
    // ------------------------------------------------------------------ phase E: second-order sweep (i) (bf16)
    b8 aps[N::SK];   // abar'_s stays in registers into phase F
    {
      b8 gb0[3];
#pragma unroll
      for (int q = 0; q < 24; ++q) gb0[q >> 3][q & 7] = (__bf16)(pe.d[q] * nbar[q % 3]);
      panel_store<b8>(pblk, L::P_GB0, lane, gb0[0], gb0[1], e0b, e1b);
      panel_store<b8>(pblk, L::P_GB0 + 1, lane, gb0[2], zero_frag<b8>(), e0b, e1b);
      b8 dummy[2 * N::HT];
      b8 gb1[N::HK];
      second_step<N, 3, N::HT, false>(Wb, o.v[OFF_W0], lane, gb0, gb1, scr, L::S_H1, L::S_Q1, L::S_AP1, dummy, pblk, L::P_GBH1, e0b, e1b);
      b8 gbm[N::HK];
      second_step<N, N::HK, N::HT, false>(Wb, o.v[OFF_WM0], lane, gb1, gbm, scr, L::S_HM, L::S_QM, L::S_APM, dummy, pblk, L::P_GBHM, e0b, e1b);
      b8 gbs[N::SK];
      if constexpr (N::NMID == 2) {
        b8 gbm1[N::HK];
        second_step<N, N::HK, N::HT, false>(Wb, o.v[OFF_WM1], lane, gbm, gbm1, scr, L::S_HM + N::HK, L::S_QM + N::HK,
                                            L::S_APM + N::HK, dummy, pblk, L::P_GBHM + N::HT, e0b, e1b);
        second_step<N, N::HK, N::ST, true>(Wb, o.v[OFF_WS], lane, gbm1, gbs, scr, L::S_HS, L::S_QS, 0, aps, pblk, L::P_GBHS, e0b, e1b);
      } else {
        second_step<N, N::HK, N::ST, true>(Wb, o.v[OFF_WS], lane, gbm, gbs, scr, L::S_HS, L::S_QS, 0, aps, pblk, L::P_GBHS, e0b, e1b);
      }
    }
    // ------------------------------------------------------------------ phase F: reverse sweep (ii) (bf16)
    {
      b8 as_[N::SK];
      float wa[16];
#pragma unroll
      for (int t = 0; t < N::ST; ++t) {
        // ubar[:SKIP]/sqrt2 = (W_last[1:,:]^T dfeat + W_last[0,:] d_sdf)/sqrt2 ; 1/sqrt2 is folded into both packs
        facc acc = tile_gemm<b8, N::HK>(tptr<b8, N::HK>(Wb, o.v[OFF_WLT], t, lane), dfeat);
        load16(T + o.v[OFF_WL0_ACC], t, h, wa);
        const h8 hv0 = scr_load(scr, L::S_HS + 2 * t, lane), hv1 = scr_load(scr, L::S_HS + 2 * t + 1, lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          as_[2 * t][j] = (__bf16)((float)aps[2 * t][j] + (acc[j] + wa[j] * dsdf) * sig_from_h((float)hv0[j]));
          as_[2 * t + 1][j] = (__bf16)((float)aps[2 * t + 1][j] + (acc[8 + j] + wa[8 + j] * dsdf) * sig_from_h((float)hv1[j]));
        }
        panel_store<b8>(pblk, L::P_ABS + t, lane, as_[2 * t], as_[2 * t + 1], e0b, e1b);
      }
      b8 am[N::HK];
      reverse_step<N, N::SK, N::HT>(Wb, o.v[OFF_WST], lane, as_, am, scr, L::S_HM + (N::NMID - 1) * N::HK,
                                    L::S_APM + (N::NMID - 1) * N::HK, pblk, L::P_ABM + (N::NMID - 1) * N::HT, e0b, e1b);
      if constexpr (N::NMID == 2) {
        b8 am0[N::HK];
        reverse_step<N, N::HK, N::HT>(Wb, o.v[OFF_WM1T], lane, am, am0, scr, L::S_HM, L::S_APM, pblk, L::P_ABM, e0b, e1b);
        b8 a1[N::HK];
        reverse_step<N, N::HK, N::HT>(Wb, o.v[OFF_WM0T], lane, am0, a1, scr, L::S_H1, L::S_AP1, pblk, L::P_AB1, e0b, e1b);
      } else {
        b8 a1[N::HK];
        reverse_step<N, N::HK, N::HT>(Wb, o.v[OFF_WM0T], lane, am, a1, scr, L::S_H1, L::S_AP1, pblk, L::P_AB1, e0b, e1b);
      }
    }
  }
}

extern "C" int avc_render_points_bwd(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                                     int S, int ldz, float sample_dist, long npts, const void* wf16, const void* wbf16,
                                     const float* tab, const int* offs, const float* d_sdf, const float* d_normal,
                                     const float* d_rgb, void* panels, long max_waves, float* scratch, void* stream) {
  if (npts <= 0) return 0;
  AvcOffsets o;
  for (int k = 0; k < OFF_COUNT; ++k) o.v[k] = offs[k];
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, pts ? 0 : 1, sample_dist};
  const long nblk = (npts + 31) / 32;
  long nw = nblk < max_waves ? nblk : max_waves;
  int grid = (int)((nw + 3) / 4);
  if (grid < 1) grid = 1;
  if ((long)grid * 4 > max_waves && max_waves >= 4) grid = (int)(max_waves / 4);
  hipStream_t s = (hipStream_t)stream;
  if (net == AVC_NET_FULL)
    hipLaunchKernelGGL((mlp_bwd_kernel<NetFull>), dim3(grid), dim3(256), 0, s, ps, npts, (const h8*)wf16, (const b8*)wbf16,
                       tab, o, d_sdf, d_normal, d_rgb, (b8*)panels, (char*)scratch);
  else if (net == AVC_NET_SMALL)
    hipLaunchKernelGGL((mlp_bwd_kernel<NetSmall>), dim3(grid), dim3(256), 0, s, ps, npts, (const h8*)wf16, (const b8*)wbf16,
                       tab, o, d_sdf, d_normal, d_rgb, (b8*)panels, (char*)scratch);
  else { avc_set_error("unknown net id"); return 1; }
  return avc_check_launch("avc_render_points_bwd");
}

// ---------------------------------------------------------------------------------------------
// weight-gradient GEMM: out[ta][tb] (+)= sum_blocks sum_kappa  A[blk][ta][kappa] x B[blk][tb][kappa]
// grid: (ta, ksplit).  Each wavefront owns one A tile-row and TBW consecutive B tiles; the 4 waves of a block
// take different B tile groups.  Partial sums are combined with fp32 atomics (ksplit x few-hundred KB: negligible).
// ---------------------------------------------------------------------------------------------
template <int TBW>
__global__ __launch_bounds__(256) void weight_grad_kernel(const b8* __restrict__ panels, int ptiles, int pa, int ta_n, int pb,
                                                          int tb_n, long nblk, float* __restrict__ out,
                                                          float* __restrict__ bias_out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ta = blockIdx.x;
  const int nsplit = gridDim.y, split = blockIdx.y;
  const int ngroups = (tb_n + TBW - 1) / TBW;
  const long b0 = nblk * split / nsplit, b1 = nblk * (split + 1) / nsplit;
  for (int grp = wv + 4 * blockIdx.z; grp < ngroups; grp += 4 * gridDim.z) {
    facc acc[TBW];
#pragma unroll
    for (int q = 0; q < TBW; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float bsum = 0.f;
    for (long blk = b0; blk < b1; ++blk) {
      const b8* base = panels + blk * (long)ptiles * 128 + lane;
      const b8 a0 = base[(long)(pa + ta) * 128], a1 = base[(long)(pa + ta) * 128 + 64];
      if (bias_out && grp == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += (float)a0[j] + (float)a1[j];
      }
#pragma unroll
      for (int q = 0; q < TBW; ++q) {
        const int tb = grp * TBW + q;
        if (tb < tb_n) {
          const b8 v0 = base[(long)(pb + tb) * 128], v1 = base[(long)(pb + tb) * 128 + 64];
          acc[q] = MF<b8>::mma(a0, v0, acc[q]);
          acc[q] = MF<b8>::mma(a1, v1, acc[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < TBW; ++q) {
      const int tb = grp * TBW + q;
      if (tb < tb_n) {
        float* dst = out + ((long)(ta * tb_n + tb) * 64 + lane) * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(dst + r, acc[q][r]);
      }
    }
    if (bias_out && grp == 0) {
      bsum = xhalf_sum(bsum);
      if (lane < 32) atomicAdd(bias_out + ta * 32 + lane, bsum);
    }
  }
}

extern "C" int avc_weight_grad(const void* panels, int ptiles, int pa, int ta, int pb, int tb, long nblk, float* out,
                               float* bias_out, int nsplit, void* stream) {
  if (nblk <= 0 || ta <= 0 || tb <= 0) return 0;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > nblk) nsplit = (int)nblk;
  const int ngroups = (tb + 1) / 2;
  dim3 grid(ta, nsplit, (ngroups + 3) / 4);
  hipLaunchKernelGGL((weight_grad_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, (const b8*)panels, ptiles, pa, ta, pb,
                     tb, nblk, out, bias_out);
  return avc_check_launch("avc_weight_grad");
}
