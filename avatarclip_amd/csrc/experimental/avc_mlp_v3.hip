// Fused point kernels on the v3 engine (rolled tile loops, scratch-resident activations, LDS-staged weights):
//   mlp_v3_kernel<N, false>  = avc_render_points_fwd : sdf + normal (d sdf/dx) + 6 colour channels per sample point
//                              (render_core, renderer.py:221-232; fields.py:72-107,154-185)
//   mlp_v3_kernel<N, true>   = avc_render_points_bwd : recompute of the above + colour backward + second-order sweep
//                              + reverse sweep (autograd of main.py:537 incl. the double backward of fields.py:96-107);
//                              writes the transposed bf16 operand panels of every weight-gradient product.
//   weight_grad_kernel       = avc_weight_grad       : dW partials = sum_points A^T B straight from the panels.
// Mathematics: SURVEY.md A.1/A.2 == oracle/analytic.py (mlp_forward / mlp_backward); index arithmetic mirrored and
// verified on CPU by tests/wave_emulator.py.
#include "avc_mlp_v3.h"
#include "../../include/avc.h"

// Workgroup shape.  The weight stream is the scarce resource: every workgroup pulls the whole packed weight set
// (1.2 MB forward, 2.7 MB backward) from L2 into LDS for each block of (32 x waves) points, and the measured LDS-DMA fill rate
// of a CU is only ~25-50 GB/s (kbench: a lone wave needs ~2.5k cycles per 17-KiB tile, 5x its MFMA time).  So ONE
// 8-wave workgroup per CU (2 waves/SIMD) shares each staged tile between 256 points, with groups of 4 tiles (136 KiB LDS).
#ifndef V3_WPB
#define V3_WPB 8
#endif
#ifndef V3_G
#define V3_G 4
#endif

template <class N>
struct BwdLayout {
  static constexpr int HT = N::HT, ST = N::ST, NM = N::NMID, NC = N::NCMID;
  // panel tile offsets (32-feature tiles) inside one 32-point block -- mirrored by packing.py
  static constexpr int P_H0 = 0;
  static constexpr int P_GB0 = P_H0 + 2;
  static constexpr int P_H1 = P_GB0 + 2;
  static constexpr int P_HM = P_H1 + HT;
  static constexpr int P_HS = P_HM + NM * HT;
  static constexpr int P_GBH1 = P_HS + ST;
  static constexpr int P_GBHM = P_GBH1 + HT;
  static constexpr int P_GBHS = P_GBHM + NM * HT;
  static constexpr int P_GA1 = P_GBHS + ST;
  static constexpr int P_GAM = P_GA1 + HT;
  static constexpr int P_GAS = P_GAM + NM * HT;
  static constexpr int P_AB1 = P_GAS + ST;
  static constexpr int P_ABM = P_AB1 + HT;
  static constexpr int P_ABS = P_ABM + NM * HT;
  static constexpr int P_DFEAT = P_ABS + ST;
  static constexpr int P_SDF = P_DFEAT + HT;
  static constexpr int P_ONE = P_SDF + 1;
  static constexpr int P_FEAT = P_ONE + 1;
  static constexpr int P_XN = P_FEAT + HT;
  static constexpr int P_R1 = P_XN + 1;
  static constexpr int P_R2 = P_R1 + HT;
  static constexpr int P_D1 = P_R2 + NC * HT;
  static constexpr int P_D2 = P_D1 + HT;
  static constexpr int P_DO = P_D2 + NC * HT;
  static constexpr int P_TILES = P_DO + 1;
  // scratch k-step offsets inside one wavefront slot (1 KiB per k-step)
  static constexpr int S_H1 = 0;
  static constexpr int S_HM = S_H1 + N::HK;
  static constexpr int S_HS = S_HM + NM * N::HK;
  static constexpr int S_Q1 = S_HS + N::SK;
  static constexpr int S_QM = S_Q1 + N::HK;
  static constexpr int S_QS = S_QM + NM * N::HK;
  static constexpr int S_AP1 = S_QS + N::SK;
  static constexpr int S_APM = S_AP1 + N::HK;
  static constexpr int S_APS = S_APM + NM * N::HK;
  static constexpr int S_R1 = S_APS + N::SK;
  static constexpr int S_R2 = S_R1 + N::HK;
  static constexpr int S_FEAT = S_R2 + N::HK;
  static constexpr int S_DFEAT = S_FEAT + N::HK;
  static constexpr int S_T0 = S_DFEAT + N::HK;     // generic ping-pong buffers (sweep chains)
  static constexpr int S_T1 = S_T0 + N::HK;
  static constexpr int S_AS = S_T1 + N::HK;
  static constexpr int S_KSTEPS = S_AS + N::SK;
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
template <typename V>
__device__ __forceinline__ V zero_frag() {
  V z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (typename MF<V>::S)0.f;
  return z;
}
// transpose the two k-steps (f0,f1) of a 32-feature tile to feature-major on the matrix core and store it as a bf16 panel tile
template <typename V>
__device__ __forceinline__ void pstore(b8* __restrict__ panel_blk, bool live, int tile, int lane, const V& f0, const V& f1,
                                       const V& e0, const V& e1) {
  facc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = MF<V>::mma(f0, e0, acc);
  acc = MF<V>::mma(f1, e1, acc);
  b8 k0, k1;
#pragma unroll
  for (int j = 0; j < 8; ++j) { k0[j] = (__bf16)acc[j]; k1[j] = (__bf16)acc[8 + j]; }
  if (live) {
    b8* dst = panel_blk + (long)tile * 128 + lane;
    dst[0] = k0;
    dst[64] = k1;
  }
}

struct PFF { h8 h0, h1; b8 a0, a1; float wa[16]; };

template <class N, bool BWD>
__global__ __launch_bounds__(64 * V3_WPB) void mlp_v3_kernel(PointSrc ps, long npts, const h8* __restrict__ Wf0,
                                                        const b8* __restrict__ Wb0, const float* __restrict__ T0, AvcOffsets o,
                                                        float* __restrict__ sdf_out, float* __restrict__ normal_out,
                                                        float* __restrict__ rgb_out, const float* __restrict__ d_sdf,
                                                        const float* __restrict__ d_normal, const float* __restrict__ d_rgb,
                                                        b8* __restrict__ panels, char* __restrict__ scratch) {
  typedef BwdLayout<N> L;
  typedef StageT<V3_G> ST;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, h = lane >> 5, p = lane & 31;
  const int wv = threadIdx.x >> 6;
  const long nblk = (npts + 31) >> 5;
  const long wslot = (long)blockIdx.x * V3_WPB + wv;
  char* scr0 = scratch + wslot * (long)L::S_KSTEPS * 64 * 16 + lane * 16;   // this lane's 16-B column of the wave's slot
  h8 e0h, e1h; b8 e0b, e1b;
  if (BWD) { make_sel<h8>(lane, e0h, e1h); make_sel<b8>(lane, e0b, e1b); }
  ST sg = stage_init<V3_G>(lds);
  stage_issue(sg, nxt<N, OFF_W0>(sg, Wf0, o), 0);

  // every wavefront of a workgroup runs the same number of iterations (workgroup-uniform loop bound)
#pragma unroll 1
  for (long blk0 = (long)blockIdx.x * V3_WPB; blk0 < nblk; blk0 += (long)gridDim.x * V3_WPB) {
    // opaque per-iteration copies: keeps LICM from hoisting the (loop-invariant) table loads and address arithmetic
    const h8* Wf = launder(Wf0);
    const b8* Wb = launder(Wb0);
    const float* T = launder(T0);
    asm volatile("" : "+v"(scr0));
    h8* scrh = reinterpret_cast<h8*>(scr0);
    b8* scrb = reinterpret_cast<b8*>(scr0);
    const long blk = blk0 + wv;
#ifdef AVC_X_NOPANEL
    const bool live = false;   // timing experiment only
#else
    const bool live = blk < nblk;
#endif
    b8* pblk = BWD ? panels + (live ? blk : 0) * (long)L::P_TILES * 128 : nullptr;
    long i = blk * 32 + p;
    const bool valid = i < npts;
    if (!valid) i = npts - 1;
    const float vmask = valid ? 1.f : 0.f;
    float x[3];
    fetch_point(ps, i, x);

#define PRE_BIAS(OFFB) AVC_PRE(PF16 q_; load16(T + o.v[OFFB], t, h, q_.b); return q_;)
#define PRE_H(SH) AVC_PRE(PF2<h8> q_; q_.a = scr_ld(scrh, (SH) + 2 * t); q_.b = scr_ld(scrh, (SH) + 2 * t + 1); return q_;)
    // ------------------------------------------------------------------ phase A: SDF trunk (f16)
    float sdfv;
    {
      {
        PE pe;
        pe_compute(x, h, pe);
        h8 pef[3];
        pe_to_frags_f16(pe, x, h, pef);
        if (BWD) {
          pstore<h8>(pblk, live, L::P_H0, lane, pef[0], pef[1], e0h, e1h);
          pstore<h8>(pblk, live, L::P_H0 + 1, lane, pef[2], zero_frag<h8>(), e0h, e1h);
        }
#define EPI_SOFTPLUS(SOUT, POUT)                                                                       \
  AVC_EPI3(PF16, float a[16];                                                                          \
           _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + pf.b[r]);          \
           h8 f0, f1; frags_from(a, f0, f1);                                                           \
           scr_st(scrh, (SOUT) + 2 * t, f0); scr_st(scrh, (SOUT) + 2 * t + 1, f1);                     \
           if (BWD) pstore<h8>(pblk, live, (POUT) + t, lane, f0, f1, e0h, e1h);)
        layer_r<h8, 3, N::HT>(sg, Wf, o.v[OFF_W0], nxt<N, OFF_WM0>(sg, Wf, o), pef, PRE_BIAS(OFF_B0),
                              EPI_SOFTPLUS(L::S_H1, L::P_H1));
      }
      {
        h8 in[N::HK];
#pragma unroll
        for (int s = 0; s < N::HK; ++s) in[s] = scr_ld(scrh, L::S_H1 + s);
        if constexpr (N::NMID == 2) {
          layer_r<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wf, o), in, PRE_BIAS(OFF_BM0),
                                    EPI_SOFTPLUS(L::S_HM, L::P_HM));
#pragma unroll
          for (int s = 0; s < N::HK; ++s) in[s] = scr_ld(scrh, L::S_HM + s);
          layer_r<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wf, o), in, PRE_BIAS(OFF_BM1),
                                    EPI_SOFTPLUS(L::S_HM + N::HK, L::P_HM + N::HT));
        } else {
          layer_r<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wf, o), in, PRE_BIAS(OFF_BM0),
                                    EPI_SOFTPLUS(L::S_HM, L::P_HM));
        }
#pragma unroll
        for (int s = 0; s < N::HK; ++s) in[s] = scr_ld(scrh, L::S_HM + (N::NMID - 1) * N::HK + s);
        float part = 0.f;
        layer_r<h8, N::HK, N::ST>(sg, Wf, o.v[OFF_WS], nxt<N, OFF_WST>(sg, Wf, o), in,
          AVC_PRE(PF32 q_; load16(T + o.v[OFF_BS], t, h, q_.b); load16(T + o.v[OFF_WL0_ACC], t, h, q_.w); return q_;),
          AVC_EPI3(PF32, float a[16];
                   _Pragma("unroll") for (int r = 0; r < 16; ++r) { a[r] = softplus2(acc[r] + pf.b[r]); part += pf.w[r] * a[r]; }
                   h8 f0, f1; frags_from(a, f0, f1);
                   scr_st(scrh, L::S_HS + 2 * t, f0); scr_st(scrh, L::S_HS + 2 * t + 1, f1);
                   if (BWD) pstore<h8>(pblk, live, L::P_HS + t, lane, f0, f1, e0h, e1h);));
        {
          PE pe;
          pe_compute(x, h, pe);
          const float* wpe = T + o.v[OFF_WL0_PE] + h * 24;
#pragma unroll
          for (int q = 0; q < 24; ++q) part += wpe[q] * pe.v[q];
        }
        sdfv = xhalf_sum(part) + T[o.v[OFF_BL0]];
      }
    }
    // ------------------------------------------------------------------ phase B: normal sweep (f16)
    float n[3];
    {
      {
        h8 gs[N::SK];
        float w8[8];
#pragma unroll
        for (int s = 0; s < N::SK; ++s) {
          load8(T + o.v[OFF_WL0_FRAG], s, h, w8);
          const h8 hsv = scr_ld(scrh, L::S_HS + s);
          h8 q;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float sg_ = sig_from_h((float)hsv[j]);
            gs[s][j] = (_Float16)(w8[j] * sg_);
            q[j] = (_Float16)(w8[j] * AVC_BETA * sg_ * (1.f - sg_) * (1.f / 64.f));
          }
          if (BWD) scr_st(scrh, L::S_QS + s, q);
        }
        if (BWD) {
#pragma unroll
          for (int t = 0; t < N::ST; ++t) pstore<h8>(pblk, live, L::P_GAS + t, lane, gs[2 * t], gs[2 * t + 1], e0h, e1h);
        }
        // g_h(prev) = W^T g_a ; g_a(prev) = g_h * sigma(h_prev) ; q = g_h * sp''(h_prev) / 64
#define EPI_NSTEP(SOUT, SQ, PT)                                                                              \
  AVC_EPI3(PF2<h8>, h8 f0, f1, q0, q1;                                                                       \
           _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                   \
             const float s0 = sig_from_h((float)pf.a[j]), s1 = sig_from_h((float)pf.b[j]);                   \
             f0[j] = (_Float16)(acc[j] * s0); f1[j] = (_Float16)(acc[8 + j] * s1);                           \
             if (BWD) { q0[j] = (_Float16)(acc[j] * AVC_BETA * s0 * (1.f - s0) * (1.f / 64.f));              \
                        q1[j] = (_Float16)(acc[8 + j] * AVC_BETA * s1 * (1.f - s1) * (1.f / 64.f)); } }      \
           scr_st(scrh, (SOUT) + 2 * t, f0); scr_st(scrh, (SOUT) + 2 * t + 1, f1);                           \
           if (BWD) { scr_st(scrh, (SQ) + 2 * t, q0); scr_st(scrh, (SQ) + 2 * t + 1, q1);                    \
                      pstore<h8>(pblk, live, (PT) + t, lane, f0, f1, e0h, e1h); })
        if constexpr (N::NMID == 2) {
          layer_r<h8, N::SK, N::HT>(sg, Wf, o.v[OFF_WST], nxt<N, OFF_WM1T>(sg, Wf, o), gs, PRE_H(L::S_HM + N::HK),
                                    EPI_NSTEP(L::S_T0, L::S_QM + N::HK, L::P_GAM + N::HT));
        } else {
          layer_r<h8, N::SK, N::HT>(sg, Wf, o.v[OFF_WST], nxt<N, OFF_WM0T>(sg, Wf, o), gs, PRE_H(L::S_HM),
                                    EPI_NSTEP(L::S_T0, L::S_QM, L::P_GAM));
        }
      }
      h8 g[N::HK];
#pragma unroll
      for (int s = 0; s < N::HK; ++s) g[s] = scr_ld(scrh, L::S_T0 + s);
      if constexpr (N::NMID == 2) {
        layer_r<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1T], nxt<N, OFF_WM0T>(sg, Wf, o), g, PRE_H(L::S_HM),
                                  EPI_NSTEP(L::S_T1, L::S_QM, L::P_GAM));
#pragma unroll
        for (int s = 0; s < N::HK; ++s) g[s] = scr_ld(scrh, L::S_T1 + s);
        layer_r<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0T], nxt<N, OFF_W0T>(sg, Wf, o), g, PRE_H(L::S_H1),
                                  EPI_NSTEP(L::S_T0, L::S_Q1, L::P_GA1));
#pragma unroll
        for (int s = 0; s < N::HK; ++s) g[s] = scr_ld(scrh, L::S_T0 + s);
      } else {
        layer_r<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0T], nxt<N, OFF_W0T>(sg, Wf, o), g, PRE_H(L::S_H1),
                                  EPI_NSTEP(L::S_T1, L::S_Q1, L::P_GA1));
#pragma unroll
        for (int s = 0; s < N::HK; ++s) g[s] = scr_ld(scrh, L::S_T1 + s);
      }
      // through layer 0 (transposed): rows = pe slots, two tiles (static tile index: the PE derivative table is in registers)
      float part[3] = {0.f, 0.f, 0.f};
      const float* wpe = T + o.v[OFF_WL0_PE] + h * 24;
      PE pe;
      pe_compute(x, h, pe);
      layer_s<h8, N::HK, 2>(sg, Wf, o.v[OFF_W0T], nxt<N, OFF_WL>(sg, Wf, o), g, AVC_EPI(
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {
          const int q = 16 * t + r;
          if (q < 24) part[q % 3] += pe.d[q] * (acc[r] + wpe[q]);
        }
      ));
#pragma unroll
      for (int c = 0; c < 3; ++c) n[c] = xhalf_sum(part[c]);
    }
    // ------------------------------------------------------------------ phase C: feature + colour forward (f16)
    float rgbv[4];      // sigmoid outputs: half 0 -> channels 0..3, half 1 -> channels 4,5
    float delta_o[4];   // BWD: d_rgb * rgb (1-rgb)
    {
      {
        h8 in[N::SK + 3];
#pragma unroll
        for (int s = 0; s < N::SK; ++s) in[s] = scr_ld(scrh, L::S_HS + s);
        {
          PE pe;
          pe_compute(x, h, pe);
          h8 pef[3];
          pe_to_frags_f16(pe, x, h, pef);
          in[N::SK] = pef[0]; in[N::SK + 1] = pef[1]; in[N::SK + 2] = pef[2];
        }
        layer_r<h8, N::SK + 3, N::HT>(sg, Wf, o.v[OFF_WL], nxt<N, OFF_C0>(sg, Wf, o), in, PRE_BIAS(OFF_BL),
          AVC_EPI3(PF16, float a[16];
                   _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = acc[r] + pf.b[r];
                   h8 f0, f1; frags_from(a, f0, f1);
                   scr_st(scrh, L::S_FEAT + 2 * t, f0); scr_st(scrh, L::S_FEAT + 2 * t + 1, f1);
                   if (BWD) pstore<h8>(pblk, live, L::P_FEAT + t, lane, f0, f1, e0h, e1h);));
      }
#define EPI_RELU(SOUT, POUT)                                                                           \
  AVC_EPI3(PF16, float a[16];                                                                          \
           _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = fmaxf(acc[r] + pf.b[r], 0.f);         \
           h8 f0, f1; frags_from(a, f0, f1);                                                           \
           scr_st(scrh, (SOUT) + 2 * t, f0); scr_st(scrh, (SOUT) + 2 * t + 1, f1);                     \
           if (BWD) pstore<h8>(pblk, live, (POUT) + t, lane, f0, f1, e0h, e1h);)
      {
        h8 in[N::HK + 1];
#pragma unroll
        for (int s = 0; s < N::HK; ++s) in[s] = scr_ld(scrh, L::S_FEAT + s);
        in[N::HK] = zero_frag<h8>();
        if (h == 0) {
#pragma unroll
          for (int c = 0; c < 3; ++c) { in[N::HK][c] = (_Float16)x[c]; in[N::HK][3 + c] = (_Float16)n[c]; }
        }
        if (BWD) pstore<h8>(pblk, live, L::P_XN, lane, in[N::HK], zero_frag<h8>(), e0h, e1h);
        layer_r<h8, N::HK + 1, N::HT>(sg, Wf, o.v[OFF_C0], (N::NCMID == 1 ? nxt<N, OFF_CM0>(sg, Wf, o) : nxt<N, OFF_CH>(sg, Wf, o)),
                                      in, PRE_BIAS(OFF_CB0), EPI_RELU(L::S_R1, L::P_R1));
      }
      h8 r[N::HK];
#pragma unroll
      for (int s = 0; s < N::HK; ++s) r[s] = scr_ld(scrh, L::S_R1 + s);
      if constexpr (N::NCMID == 1) {
        layer_r<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_CM0], nxt<N, OFF_CH>(sg, Wf, o), r, PRE_BIAS(OFF_CBM0),
                                  EPI_RELU(L::S_R2, L::P_R2));
#pragma unroll
        for (int s = 0; s < N::HK; ++s) r[s] = scr_ld(scrh, L::S_R2 + s);
      }
      // heads: one tile (6 live rows)
      layer_s<h8, N::HK, 1>(sg, Wf, o.v[OFF_CH], (BWD ? nxt<N, OFF_CHT>(sg, Wb, o) : nxt<N, OFF_W0>(sg, Wf0, o)), r, AVC_EPI(
        float b[16];
        load16(T + o.v[OFF_CBH], 0, h, b);
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {
          const float v = sigmoidf_(acc[k] + b[k]);
          rgbv[k] = v;
          if (BWD) {
            const int ch = h ? 4 + k : k;
            const float dr = (ch < 6) ? d_rgb[6 * i + (ch < 6 ? ch : 0)] * vmask : 0.f;
            delta_o[k] = dr * v * (1.f - v);
          }
        }
      ));
    }
    if (!BWD) {
      if (valid) {
        if (h == 0) {
          sdf_out[i] = sdfv;
          normal_out[3 * i + 0] = n[0]; normal_out[3 * i + 1] = n[1]; normal_out[3 * i + 2] = n[2];
          rgb_out[6 * i + 0] = rgbv[0]; rgb_out[6 * i + 1] = rgbv[1]; rgb_out[6 * i + 2] = rgbv[2]; rgb_out[6 * i + 3] = rgbv[3];
        } else {
          rgb_out[6 * i + 4] = rgbv[0]; rgb_out[6 * i + 5] = rgbv[1];
        }
      }
      continue;
    }
    // ------------------------------------------------------------------ phase D: colour backward (bf16)
    float nbar[3];
    {
      {
        b8 dof[1];
        dof[0] = zero_frag<b8>();
#pragma unroll
        for (int k = 0; k < 4; ++k) dof[0][k] = (__bf16)delta_o[k];
        pstore<b8>(pblk, live, L::P_DO, lane, dof[0], zero_frag<b8>(), e0b, e1b);
#define EPI_RELU_BWD(SOUT, PT)                                                                          \
  AVC_EPI3(PF2<h8>, b8 f0, f1;                                                                          \
           _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                              \
             f0[j] = (__bf16)((float)pf.a[j] > 0.f ? acc[j] : 0.f);                                     \
             f1[j] = (__bf16)((float)pf.b[j] > 0.f ? acc[8 + j] : 0.f); }                               \
           scr_st(scrb, (SOUT) + 2 * t, f0); scr_st(scrb, (SOUT) + 2 * t + 1, f1);                      \
           pstore<b8>(pblk, live, (PT) + t, lane, f0, f1, e0b, e1b);)
        if constexpr (N::NCMID == 1) {
          layer_r<b8, 1, N::HT>(sg, Wb, o.v[OFF_CHT], nxt<N, OFF_CM0T>(sg, Wb, o), dof, PRE_H(L::S_R2), EPI_RELU_BWD(L::S_T0, L::P_D2));
        } else {
          layer_r<b8, 1, N::HT>(sg, Wb, o.v[OFF_CHT], nxt<N, OFF_C0T>(sg, Wb, o), dof, PRE_H(L::S_R1), EPI_RELU_BWD(L::S_T1, L::P_D1));
        }
      }
      b8 d[N::HK];
      if constexpr (N::NCMID == 1) {
#pragma unroll
        for (int s = 0; s < N::HK; ++s) d[s] = scr_ld(scrb, L::S_T0 + s);
        layer_r<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_CM0T], nxt<N, OFF_C0T>(sg, Wb, o), d, PRE_H(L::S_R1), EPI_RELU_BWD(L::S_T1, L::P_D1));
      }
#pragma unroll
      for (int s = 0; s < N::HK; ++s) d[s] = scr_ld(scrb, L::S_T1 + s);
      // d r0 = C0^T delta1: HT feature tiles (-> ybar[1:]), then the [x,n] tile (rows 3,4,5 = d n)
      float dn_acc[3] = {0.f, 0.f, 0.f};
      layer_r<b8, N::HK, N::HT + 1>(sg, Wb, o.v[OFF_C0T], nxt<N, OFF_W0G>(sg, Wb, o), d, AVC_PRE(return PFNone();),
        AVC_EPI3(PFNone,
          if (t < N::HT) {
            b8 f0, f1;
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { f0[j] = (__bf16)acc[j]; f1[j] = (__bf16)acc[8 + j]; }
            scr_st(scrb, L::S_DFEAT + 2 * t, f0); scr_st(scrb, L::S_DFEAT + 2 * t + 1, f1);
            pstore<b8>(pblk, live, L::P_DFEAT + t, lane, f0, f1, e0b, e1b);
          } else {
            dn_acc[0] = acc[3]; dn_acc[1] = acc[0]; dn_acc[2] = acc[1];
          }));
      {
        // row 3 -> (h0,r3), row 4 -> (h1,r0), row 5 -> (h1,r1)
        const float a3 = dn_acc[0], a0 = dn_acc[1], a1 = dn_acc[2];
        const float o3 = __shfl_xor(a3, 32), o0 = __shfl_xor(a0, 32), o1 = __shfl_xor(a1, 32);
        nbar[0] = d_normal[3 * i + 0] * vmask + (h ? o3 : a3);
        nbar[1] = d_normal[3 * i + 1] * vmask + (h ? a0 : o0);
        nbar[2] = d_normal[3 * i + 2] * vmask + (h ? a1 : o1);
      }
    }
    const float dsdfS = d_sdf[i] * vmask * AVC_S;   // OFF_WL0_ACC holds W_last[0,:]/(S sqrt2): undo S for the gradient use
    if (live) {   // A-panels with a single live feature: d_sdf and the constant 1 (row 0 of the last layer)
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
      b8* dd1 = pblk + (long)L::P_SDF * 128 + lane; dd1[0] = k0; dd1[64] = k1;
      b8* dd2 = pblk + (long)L::P_ONE * 128 + lane; dd2[0] = o0; dd2[64] = o1;
    }
    // ------------------------------------------------------------------ phase E: second-order sweep (i) (bf16)
    {
#define PRE_HQ(SH, SQ) AVC_PRE(PF4<h8, h8> q_; q_.h0 = scr_ld(scrh, (SH) + 2 * t); q_.h1 = scr_ld(scrh, (SH) + 2 * t + 1); \
                               q_.q0 = scr_ld(scrh, (SQ) + 2 * t); q_.q1 = scr_ld(scrh, (SQ) + 2 * t + 1); return q_;)
      // gbar_a = W gbar_h(in); abar' = gbar_a * q * 64 ; gbar_h(out) = gbar_a * sigma(h_out)
#define EPI_SECOND(SOUT, SAP, PT)                                                                            \
  AVC_EPI3(PF4<h8 COMMA h8>, b8 f0, f1, a0, a1;                                                              \
           _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                   \
             f0[j] = (__bf16)(acc[j] * sig_from_h((float)pf.h0[j]));                                         \
             f1[j] = (__bf16)(acc[8 + j] * sig_from_h((float)pf.h1[j]));                                     \
             a0[j] = (__bf16)(acc[j] * (float)pf.q0[j] * 64.f); a1[j] = (__bf16)(acc[8 + j] * (float)pf.q1[j] * 64.f); } \
           if ((SOUT) >= 0) { scr_st(scrb, (SOUT) + 2 * t, f0); scr_st(scrb, (SOUT) + 2 * t + 1, f1); }      \
           scr_st(scrb, (SAP) + 2 * t, a0); scr_st(scrb, (SAP) + 2 * t + 1, a1);                             \
           pstore<b8>(pblk, live, (PT) + t, lane, f0, f1, e0b, e1b);)
#define COMMA ,
      {
        b8 gb0[3];
        {
          PE pe;
          pe_compute(x, h, pe);
#pragma unroll
          for (int q = 0; q < 24; ++q) gb0[q >> 3][q & 7] = (__bf16)(pe.d[q] * nbar[q % 3]);
        }
        pstore<b8>(pblk, live, L::P_GB0, lane, gb0[0], gb0[1], e0b, e1b);
        pstore<b8>(pblk, live, L::P_GB0 + 1, lane, gb0[2], zero_frag<b8>(), e0b, e1b);
        layer_r<b8, 3, N::HT>(sg, Wb, o.v[OFF_W0G], nxt<N, OFF_WM0>(sg, Wb, o), gb0, PRE_HQ(L::S_H1, L::S_Q1),
                              EPI_SECOND(L::S_T0, L::S_AP1, L::P_GBH1));
      }
      b8 gb[N::HK];
#pragma unroll
      for (int s = 0; s < N::HK; ++s) gb[s] = scr_ld(scrb, L::S_T0 + s);
      if constexpr (N::NMID == 2) {
        layer_r<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wb, o), gb, PRE_HQ(L::S_HM, L::S_QM),
                                  EPI_SECOND(L::S_T1, L::S_APM, L::P_GBHM));
#pragma unroll
        for (int s = 0; s < N::HK; ++s) gb[s] = scr_ld(scrb, L::S_T1 + s);
        layer_r<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wb, o), gb, PRE_HQ(L::S_HM + N::HK, L::S_QM + N::HK),
                                  EPI_SECOND(L::S_T0, L::S_APM + N::HK, L::P_GBHM + N::HT));
#pragma unroll
        for (int s = 0; s < N::HK; ++s) gb[s] = scr_ld(scrb, L::S_T0 + s);
      } else {
        layer_r<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wb, o), gb, PRE_HQ(L::S_HM, L::S_QM),
                                  EPI_SECOND(L::S_T1, L::S_APM, L::P_GBHM));
#pragma unroll
        for (int s = 0; s < N::HK; ++s) gb[s] = scr_ld(scrb, L::S_T1 + s);
      }
      layer_r<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WS], nxt<N, OFF_WLT>(sg, Wb, o), gb, PRE_HQ(L::S_HS, L::S_QS),
                                EPI_SECOND(-1, L::S_APS, L::P_GBHS));
    }
    // ------------------------------------------------------------------ phase F: reverse sweep (ii) (bf16)
    {
      {
        b8 dfeat[N::HK];
#pragma unroll
        for (int s = 0; s < N::HK; ++s) dfeat[s] = scr_ld(scrb, L::S_DFEAT + s);
        // ubar[:SKIP]/sqrt2 = (W_last[1:,:]^T dfeat + W_last[0,:] d_sdf)/sqrt2 ; abar_s = abar'_s + ubar * sigma(h_s)
        layer_r<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WLT], nxt<N, OFF_WST>(sg, Wb, o), dfeat,
          AVC_PRE(PFF q_; q_.h0 = scr_ld(scrh, L::S_HS + 2 * t); q_.h1 = scr_ld(scrh, L::S_HS + 2 * t + 1);
                  q_.a0 = scr_ld(scrb, L::S_APS + 2 * t); q_.a1 = scr_ld(scrb, L::S_APS + 2 * t + 1);
                  load16(T + o.v[OFF_WL0_ACC], t, h, q_.wa); return q_;),
          AVC_EPI3(PFF, b8 f0, f1;
                   _Pragma("unroll") for (int j = 0; j < 8; ++j) {
                     f0[j] = (__bf16)((float)pf.a0[j] + (acc[j] + pf.wa[j] * dsdfS) * sig_from_h((float)pf.h0[j]));
                     f1[j] = (__bf16)((float)pf.a1[j] + (acc[8 + j] + pf.wa[8 + j] * dsdfS) * sig_from_h((float)pf.h1[j])); }
                   scr_st(scrb, L::S_AS + 2 * t, f0); scr_st(scrb, L::S_AS + 2 * t + 1, f1);
                   pstore<b8>(pblk, live, L::P_ABS + t, lane, f0, f1, e0b, e1b);));
      }
#define PRE_HA(SH, SAP) AVC_PRE(PF4<h8, b8> q_; q_.h0 = scr_ld(scrh, (SH) + 2 * t); q_.h1 = scr_ld(scrh, (SH) + 2 * t + 1); \
                                q_.q0 = scr_ld(scrb, (SAP) + 2 * t); q_.q1 = scr_ld(scrb, (SAP) + 2 * t + 1); return q_;)
      // hbar(prev) = W^T abar(cur); abar(prev) = abar'(prev) + hbar * sigma(h_prev)
#define EPI_REVERSE(SOUT, PT)                                                                                \
  AVC_EPI3(PF4<h8 COMMA b8>, b8 f0, f1;                                                                      \
           _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                   \
             f0[j] = (__bf16)((float)pf.q0[j] + acc[j] * sig_from_h((float)pf.h0[j]));                       \
             f1[j] = (__bf16)((float)pf.q1[j] + acc[8 + j] * sig_from_h((float)pf.h1[j])); }                 \
           if ((SOUT) >= 0) { scr_st(scrb, (SOUT) + 2 * t, f0); scr_st(scrb, (SOUT) + 2 * t + 1, f1); }      \
           pstore<b8>(pblk, live, (PT) + t, lane, f0, f1, e0b, e1b);)
      const Next first = nxt<N, OFF_W0>(sg, Wf0, o);   // first group of the next block iteration
      {
        b8 as_[N::SK];
#pragma unroll
        for (int s = 0; s < N::SK; ++s) as_[s] = scr_ld(scrb, L::S_AS + s);
        if constexpr (N::NMID == 2) {
          layer_r<b8, N::SK, N::HT>(sg, Wb, o.v[OFF_WST], nxt<N, OFF_WM1T>(sg, Wb, o), as_,
                                    PRE_HA(L::S_HM + N::HK, L::S_APM + N::HK), EPI_REVERSE(L::S_T0, L::P_ABM + N::HT));
        } else {
          layer_r<b8, N::SK, N::HT>(sg, Wb, o.v[OFF_WST], nxt<N, OFF_WM0T>(sg, Wb, o), as_, PRE_HA(L::S_HM, L::S_APM),
                                    EPI_REVERSE(L::S_T0, L::P_ABM));
        }
      }
      b8 am[N::HK];
#pragma unroll
      for (int s = 0; s < N::HK; ++s) am[s] = scr_ld(scrb, L::S_T0 + s);
      if constexpr (N::NMID == 2) {
        layer_r<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM1T], nxt<N, OFF_WM0T>(sg, Wb, o), am, PRE_HA(L::S_HM, L::S_APM),
                                  EPI_REVERSE(L::S_T1, L::P_ABM));
#pragma unroll
        for (int s = 0; s < N::HK; ++s) am[s] = scr_ld(scrb, L::S_T1 + s);
      }
      layer_r<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0T], first, am, PRE_HA(L::S_H1, L::S_AP1), EPI_REVERSE(-1, L::P_AB1));
    }
  }
}

static int v3_grid(long npts, long max_waves) {
  const long nblk = (npts + 31) / 32;
  long ngroups = (nblk + V3_WPB - 1) / V3_WPB;
  long maxg = max_waves / V3_WPB;
  if (maxg < 1) maxg = 1;
  long g = ngroups < maxg ? ngroups : maxg;
  return (int)(g < 1 ? 1 : g);
}

template <bool BWD>
static int v3_launch(int net, PointSrc ps, long npts, const void* wf16, const void* wbf16, const float* tab, const int* offs,
                     float* sdf_out, float* normal_out, float* rgb_out, const float* d_sdf, const float* d_normal,
                     const float* d_rgb, void* panels, long max_waves, void* scratch, void* stream) {
  if (npts <= 0) return 0;
  AvcOffsets o;
  for (int k = 0; k < OFF_COUNT; ++k) o.v[k] = offs[k];
  const int grid = v3_grid(npts, max_waves);
  hipStream_t s = (hipStream_t)stream;
  const int lds_bytes = StageT<V3_G>::LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)mlp_v3_kernel<NetFull, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipFuncSetAttribute((const void*)mlp_v3_kernel<NetSmall, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set = true;
  }
  if (net == AVC_NET_FULL)
    hipLaunchKernelGGL((mlp_v3_kernel<NetFull, BWD>), dim3(grid), dim3(64 * V3_WPB), lds_bytes, s, ps, npts, (const h8*)wf16,
                       (const b8*)wbf16, tab, o, sdf_out, normal_out, rgb_out, d_sdf, d_normal, d_rgb, (b8*)panels, (char*)scratch);
  else if (net == AVC_NET_SMALL)
    hipLaunchKernelGGL((mlp_v3_kernel<NetSmall, BWD>), dim3(grid), dim3(64 * V3_WPB), lds_bytes, s, ps, npts, (const h8*)wf16,
                       (const b8*)wbf16, tab, o, sdf_out, normal_out, rgb_out, d_sdf, d_normal, d_rgb, (b8*)panels, (char*)scratch);
  else { avc_set_error("unknown net id"); return 1; }
  return avc_check_launch(BWD ? "avc_render_points_bwd" : "avc_render_points_fwd");
}

extern "C" int avc_render_points_fwd(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                                     int S, int ldz, float sample_dist, long npts, const void* wf16, const float* tab,
                                     const int* offs, float* sdf_out, float* normal_out, float* rgb_out, long max_waves,
                                     void* scratch, void* stream) {
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, pts ? 0 : 1, sample_dist};
  return v3_launch<false>(net, ps, npts, wf16, nullptr, tab, offs, sdf_out, normal_out, rgb_out, nullptr, nullptr, nullptr,
                          nullptr, max_waves, scratch, stream);
}

extern "C" int avc_render_points_bwd(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                                     int S, int ldz, float sample_dist, long npts, const void* wf16, const void* wbf16,
                                     const float* tab, const int* offs, const float* d_sdf, const float* d_normal,
                                     const float* d_rgb, void* panels, long max_waves, float* scratch, void* stream) {
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, pts ? 0 : 1, sample_dist};
  return v3_launch<true>(net, ps, npts, wf16, wbf16, tab, offs, nullptr, nullptr, nullptr, d_sdf, d_normal, d_rgb, panels,
                         max_waves, scratch, stream);
}
