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
// The panels double as the activation store of the sweeps: a phase that needs h, g_a, r or ybar again reads the
// panel tile back and un-transposes it with the same two selection MFMAs (the transposition is an involution), so
// each activation crosses HBM once as a panel instead of once as a panel and once as a scratch copy (PMC: the
// scratch copies were 18 % of the kernel's HBM bytes and the kernel is HBM-bound).  The second-order term abar' is not
// stored at all: the reverse sweep rebuilds it from the gbar_h, g_a and h panels (abar' = gbar_h g_a beta (1-s)/s).
// ReLU masks travel from the colour forward to the colour backward as 16 bits per tile in registers.
#include "avc_mlp.h"
#ifndef BWD_WAVES_PER_EU
#define BWD_WAVES_PER_EU 2   // 2 waves/SIMD (256 VGPRs): measured 20 % faster than 1 wave x 512 registers
#endif
#ifndef BWD_G
#define BWD_G 4   // tiles per staged group (LDS = 2 * G * 17 KiB = 136 KiB: one 8-wave workgroup per CU)
#endif
#ifndef BWD_WPB
#define BWD_WPB 8   // wavefronts per workgroup: every staged weight tile is shared by 256 points (LDS-DMA fill rate is the scarce resource)
#endif
#define BWD_MASK_BYTES (2 * 8 * 64 * 2)   // per wavefront: 2 layers x <= 8 tiles x 64 lanes x 16 bits
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
};

extern "C" int avc_bwd_panel_tiles(int net) {
  return net == AVC_NET_FULL ? BwdLayout<NetFull>::P_TILES : BwdLayout<NetSmall>::P_TILES;
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
  AVC_NT_STORE(k0, &dst[0]);
  AVC_NT_STORE(k1, &dst[64]);
}
template <typename V>
__device__ __forceinline__ V zero_frag() {
  V z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (typename MF<V>::S)0.f;
  return z;
}


// All phases run on the staged engine (avc_stage.h / layer_s): every weight tile is copied once per workgroup into
// LDS, the epilogue of tile t-1 (activation, panel transposition, scratch parking) is issued under the MFMAs of tile t.
// `live` = this wavefront owns a real 32-point block (waves past the end still walk the tile sequence for the barriers).

// KEEP = the tile is read back by a later sweep of the same block: normal cache policy (it may still be in L2);
// otherwise the tile is only read by the weight-gradient kernel, much later: non-temporal, does not displace the weights.
#ifndef PANEL_KEEP
#define PANEL_KEEP true
#endif
template <typename V, bool KEEP>
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
    if (KEEP) {
      dst[0] = k0;
      dst[64] = k1;
    } else {
      AVC_NT_STORE(k0, &dst[0]);
      AVC_NT_STORE(k1, &dst[64]);
    }
  }
}

// read a panel tile back into accumulator layout (lane = point, reg r <-> feature row (r&3)+8(r>>2)+4h, i.e. regs 0..7 =
// the slots of k-step 2t, regs 8..15 = the slots of k-step 2t+1): the feature-major tile is the A operand, the same 0/1
// selection fragments pick the point column.  Values come back exactly as stored (bf16).
__device__ __forceinline__ facc punpack(const b8* __restrict__ panel_blk, int tile, int lane, const b8& e0, const b8& e1) {
  const b8* src = panel_blk + (long)tile * 128 + lane;
  const b8 k0 = AVC_NT_LOAD(&src[0]), k1 = AVC_NT_LOAD(&src[64]);
  facc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = MF<b8>::mma(k0, e0, acc);
  acc = MF<b8>::mma(k1, e1, acc);
  return acc;
}
// abar' = gbar_a g_h sp''(h) with gbar_a = gbar_h / s and g_h sp'' = g_a beta (1 - s): everything on the right is a panel
// the wave has already written.  s -> 0 makes both gbar_h and g_a vanish; the guard keeps 0/0 out.
__device__ __forceinline__ float second_term(float gbar_h, float g_a, float s) {
  const float r = s > 1e-30f ? __builtin_amdgcn_rcpf(s) : 0.f;
  return gbar_h * g_a * (AVC_BETA * (1.f - s) * r);
}
// split form for layer_sq: the raw loads ...
struct PF1 { b8 a0, a1; };
struct PF3 { b8 a0, a1, b0, b1, c0, c1; };
__device__ __forceinline__ PF1 pfetch1(const b8* __restrict__ panel_blk, int tile, int lane) {
  const b8* src = panel_blk + (long)tile * 128 + lane;
  PF1 d;
  d.a0 = AVC_NT_LOAD(&src[0]);
  d.a1 = AVC_NT_LOAD(&src[64]);
  return d;
}
__device__ __forceinline__ PF3 pfetch3(const b8* __restrict__ panel_blk, int ta, int tb, int tc, int lane) {
  const b8* sa = panel_blk + (long)ta * 128 + lane;
  const b8* sb = panel_blk + (long)tb * 128 + lane;
  const b8* sc = panel_blk + (long)tc * 128 + lane;
  PF3 d;
  d.a0 = AVC_NT_LOAD(&sa[0]); d.a1 = AVC_NT_LOAD(&sa[64]);
  d.b0 = AVC_NT_LOAD(&sb[0]); d.b1 = AVC_NT_LOAD(&sb[64]);
  d.c0 = AVC_NT_LOAD(&sc[0]); d.c1 = AVC_NT_LOAD(&sc[64]);
  return d;
}
// ... and the un-transposition at the point of use
__device__ __forceinline__ facc ptrans(const b8& k0, const b8& k1, const b8& e0, const b8& e1) {
  facc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = MF<b8>::mma(k0, e0, acc);
  acc = MF<b8>::mma(k1, e1, acc);
  return acc;
}
template <typename V>
__device__ __forceinline__ void punpack_frags(const b8* __restrict__ panel_blk, int tile, int lane, const b8& e0, const b8& e1,
                                              V& f0, V& f1) {
  const facc a = punpack(panel_blk, tile, lane, e0, e1);
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = a[r];
  acc_to_frags(v, f0, f1);
}

template <class N>
__global__ __launch_bounds__(64 * BWD_WPB) void mlp_bwd_kernel(PointSrc ps, long npts, const h8* __restrict__ Wf0,
                                                      const b8* __restrict__ Wb0, const float* __restrict__ T0, AvcOffsets o,
                                                      const float* __restrict__ d_sdf, const float* __restrict__ d_normal,
                                                      const float* __restrict__ d_rgb, b8* __restrict__ panels) {
  typedef BwdLayout<N> L;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef StageT<BWD_G> ST;
  const int lane = threadIdx.x & 63, h = lane >> 5, p = lane & 31;
  const int wv = threadIdx.x >> 6;
  const long nblk = (npts + 31) >> 5;
  h8 e0h, e1h; b8 e0b, e1b;
  make_sel<h8>(lane, e0h, e1h);
  make_sel<b8>(lane, e0b, e1b);
  ST sg = stage_init<BWD_G>(lds);
  stage_issue(sg, nxt<N, OFF_W0>(sg, Wf0, o), 0);
  // the fp32 table lives in LDS: a global load in an epilogue would queue behind the LDS-DMA of the next weight group
  const lds_tab_t T = tab_to_lds(lds + ST::LDS_BYTES + BWD_WPB * BWD_MASK_BYTES, T0, o.v[OFF_TAB_END]);
  __syncthreads();

  // every wavefront of a workgroup runs the same number of iterations (workgroup-uniform loop bound)
  for (long blk0 = (long)blockIdx.x * BWD_WPB; blk0 < nblk; blk0 += (long)gridDim.x * BWD_WPB) {
    const h8* Wf = launder(Wf0);
    const b8* Wb = launder(Wb0);
    const long blk = blk0 + wv;
    const bool live = blk < nblk;
    b8* pblk = panels + (live ? blk : 0) * (long)L::P_TILES * 128;
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
    pstore<h8, false>(pblk, live, L::P_H0, lane, pef[0], pef[1], e0h, e1h);
    pstore<h8, false>(pblk, live, L::P_H0 + 1, lane, pef[2], zero_frag<h8>(), e0h, e1h);
    // Register discipline: nothing but x, n, nbar, d_sdf survives a phase.  Every activation goes out as a panel and is
    // read back (punpack) / re-computed (positional encoding) where it is needed again; this keeps each phase at
    // "input + output + accumulators" and leaves registers for pipelining the LDS operand reads.
    h8 g_s[N::SK];   // g_a of the skip layer = W_last[0,:] * sigma(h_s): the seed of the normal sweep (phase B)
    {
#define AVC_FWD_KEEP(OFFB, OUT, PT)                                                     \
  AVC_EPI(float b[16], a[16]; load16(T + o.v[OFFB], t, h, b);                                \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);  \
          acc_to_frags(a, OUT[2 * t], OUT[2 * t + 1]);                                        \
          pstore<h8, PANEL_KEEP>(pblk, live, (PT) + t, lane, OUT[2 * t], OUT[2 * t + 1], e0h, e1h);)
      // last trunk layer: h_s goes out as a panel only; the registers keep g_a,s
#define AVC_FWD_LAST(OFFB, PT, PG)                                                            \
  AVC_EPI(float b[16], a[16]; load16(T + o.v[OFFB], t, h, b);                                \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r] + b[r]);  \
          h8 hs0, hs1; acc_to_frags(a, hs0, hs1);                                             \
          pstore<h8, PANEL_KEEP>(pblk, live, (PT) + t, lane, hs0, hs1, e0h, e1h);                         \
          float w0[8], w1[8];                                                                 \
          load8(T + o.v[OFF_WL0_FRAG], 2 * t, h, w0); load8(T + o.v[OFF_WL0_FRAG], 2 * t + 1, h, w1); \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                     \
            g_s[2 * t][j] = (_Float16)(w0[j] * sig_from_h(a[j]));                             \
            g_s[2 * t + 1][j] = (_Float16)(w1[j] * sig_from_h(a[8 + j])); }                   \
          pin2(g_s[2 * t], g_s[2 * t + 1]);                                                   \
          pstore<h8, PANEL_KEEP>(pblk, live, (PG) + t, lane, g_s[2 * t], g_s[2 * t + 1], e0h, e1h);)
      h8 h1[N::HK];
      layer_s<h8, 3, N::HT>(sg, Wf, o.v[OFF_W0], nxt<N, OFF_WM0>(sg, Wf, o), pef,
                                   AVC_FWD_KEEP(OFF_B0, h1, L::P_H1));
      h8 hm0[N::HK];
      if constexpr (N::NMID == 2) {
        layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wf, o), h1,
                                         AVC_FWD_KEEP(OFF_BM0, hm0, L::P_HM));
        h8 hm1[N::HK];
        layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wf, o), hm0,
                                         AVC_FWD_KEEP(OFF_BM1, hm1, L::P_HM + N::HT));
        layer_s<h8, N::HK, N::ST>(sg, Wf, o.v[OFF_WS], nxt<N, OFF_WST>(sg, Wf, o), hm1,
                                         AVC_FWD_LAST(OFF_BS, L::P_HS, L::P_GAS));
      } else {
        layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wf, o), h1,
                                         AVC_FWD_KEEP(OFF_BM0, hm0, L::P_HM));
        layer_s<h8, N::HK, N::ST>(sg, Wf, o.v[OFF_WS], nxt<N, OFF_WST>(sg, Wf, o), hm0,
                                         AVC_FWD_LAST(OFF_BS, L::P_HS, L::P_GAS));
      }
    }
    // ------------------------------------------------------------------ phase B: normal sweep (f16)
    float n[3];
    {
      // g_h(prev) = W^T g_a ; g_a(prev) = g_h * sigma(h_prev)
#define AVC_NSTEP(OUT, PH, PT)                                                                            \
  AVC_PRE(return pfetch1(pblk, (PH) + t, lane);),                                                           \
  AVC_EPID(PF1, const facc hv = ptrans(d.a0, d.a1, e0b, e1b);                                               \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                   \
            OUT[2 * t][j] = (_Float16)(acc[j] * sig_from_h(hv[j]));                                         \
            OUT[2 * t + 1][j] = (_Float16)(acc[8 + j] * sig_from_h(hv[8 + j])); }                           \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                 \
          pstore<h8, PANEL_KEEP>(pblk, live, (PT) + t, lane, OUT[2 * t], OUT[2 * t + 1], e0h, e1h);)
      h8 g[N::HK];
      h8 g2[N::HK];
      if constexpr (N::NMID == 2) {
        layer_sqd<h8, N::SK, N::HT>(sg, Wf, o.v[OFF_WST], nxt<N, OFF_WM1T>(sg, Wf, o), g_s,
                                         AVC_NSTEP(g, L::P_HM + N::HT, L::P_GAM + N::HT));
        layer_sqd<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1T], nxt<N, OFF_WM0T>(sg, Wf, o), g,
                                         AVC_NSTEP(g2, L::P_HM, L::P_GAM));
        layer_sqd<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0T], nxt<N, OFF_W0T>(sg, Wf, o), g2,
                                         AVC_NSTEP(g, L::P_H1, L::P_GA1));
      } else {
        layer_sqd<h8, N::SK, N::HT>(sg, Wf, o.v[OFF_WST], nxt<N, OFF_WM0T>(sg, Wf, o), g_s,
                                         AVC_NSTEP(g2, L::P_HM, L::P_GAM));
        layer_sqd<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0T], nxt<N, OFF_W0T>(sg, Wf, o), g2,
                                         AVC_NSTEP(g, L::P_H1, L::P_GA1));
      }
      float part[3] = {0.f, 0.f, 0.f};
      const auto wpe = T + o.v[OFF_WL0_PE] + h * 24;
      PE pe2;
      pe_compute(x, h, pe2);
      layer_s<h8, N::HK, 2>(sg, Wf, o.v[OFF_W0T], nxt<N, OFF_WL>(sg, Wf, o), g, AVC_EPI(
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {
          const int q = 16 * t + r;
          if (q < 24) part[q % 3] += pe2.d[q] * (acc[r] + wpe[q]);
        }
      ));
#pragma unroll
      for (int c = 0; c < 3; ++c) n[c] = xhalf_sum(part[c]);
    }
    // ------------------------------------------------------------------ phase C: colour forward (f16)
    float delta_o[4];   // half 0: outputs 0..3, half 1: outputs 4,5 (delta = d_rgb * rgb (1-rgb))
    // ReLU masks of r1 / r2: 16 bits per tile and lane (bit r = accumulator reg r), parked in a wave-private corner of LDS
    // (registers are the scarce resource between the colour forward and its backward)
    unsigned short* m1 = reinterpret_cast<unsigned short*>(lds + ST::LDS_BYTES + wv * BWD_MASK_BYTES) + lane;
    unsigned short* m2 = m1 + N::HT * 64;
    {
      h8 feat[N::HK];
      {
        h8 hs[N::SK];
#pragma unroll
        for (int t = 0; t < N::ST; ++t) punpack_frags<h8>(pblk, L::P_HS + t, lane, e0b, e1b, hs[2 * t], hs[2 * t + 1]);
        PE pe3;
        pe_compute(x, h, pe3);
        h8 pef3[3];
        pe_to_frags_f16(pe3, x, h, pef3);
        layer2_s<h8, N::SK, 3, N::HT>(sg, Wf, o.v[OFF_WL], nxt<N, OFF_C0>(sg, Wf, o), hs, pef3, AVC_EPI(
        float b[16], a[16];
        load16(T + o.v[OFF_BL], t, h, b);
        _Pragma("unroll") for (int r = 0; r < 16; ++r) a[r] = acc[r] + b[r];
        acc_to_frags(a, feat[2 * t], feat[2 * t + 1]);
        pstore<h8, false>(pblk, live, L::P_FEAT + t, lane, feat[2 * t], feat[2 * t + 1], e0h, e1h);
        ));
      }
      h8 xn[1];
      xn[0] = zero_frag<h8>();
      if (h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { xn[0][c] = (_Float16)x[c]; xn[0][3 + c] = (_Float16)n[c]; }
      }
      pstore<h8, false>(pblk, live, L::P_XN, lane, xn[0], zero_frag<h8>(), e0h, e1h);
#define AVC_RELU_KEEP(OFFB, OUT, MSK, PT)                                                    \
  AVC_EPI(float b[16], a[16]; load16(T + o.v[OFFB], t, h, b);                                \
          unsigned bits = 0u;                                                                 \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                    \
            a[r] = fmaxf(acc[r] + b[r], 0.f); bits |= (a[r] > 0.f ? 1u : 0u) << r; }          \
          MSK[t * 64] = (unsigned short)bits;                                                 \
          acc_to_frags(a, OUT[2 * t], OUT[2 * t + 1]);                                        \
          pstore<h8, false>(pblk, live, (PT) + t, lane, OUT[2 * t], OUT[2 * t + 1], e0h, e1h);)
      h8 r1[N::HK];
      h8 r2[N::HK];
      if constexpr (N::NCMID == 1) {
        layer2_s<h8, N::HK, 1, N::HT>(sg, Wf, o.v[OFF_C0], nxt<N, OFF_CM0>(sg, Wf, o), feat, xn,
                                             AVC_RELU_KEEP(OFF_CB0, r1, m1, L::P_R1));
        layer_s<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_CM0], nxt<N, OFF_CH>(sg, Wf, o), r1,
                                         AVC_RELU_KEEP(OFF_CBM0, r2, m2, L::P_R2));
      } else {
        layer2_s<h8, N::HK, 1, N::HT>(sg, Wf, o.v[OFF_C0], nxt<N, OFF_CH>(sg, Wf, o), feat, xn,
                                             AVC_RELU_KEEP(OFF_CB0, r1, m1, L::P_R1));
#pragma unroll
        for (int s = 0; s < N::HK; ++s) r2[s] = r1[s];
      }
      layer_s<h8, N::HK, 1>(sg, Wf, o.v[OFF_CH], nxt<N, OFF_CHT>(sg, Wb, o), r2, AVC_EPI(
        float b[16];
        load16(T + o.v[OFF_CBH], 0, h, b);
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {
          const float rgb = sigmoidf_(acc[r] + b[r]);
          const int ch = h ? 4 + r : r;
          const float dr = (ch < 6) ? d_rgb[6 * i + (ch < 6 ? ch : 0)] * vmask : 0.f;
          delta_o[r] = dr * rgb * (1.f - rgb);
        }
      ));
    }
    // ------------------------------------------------------------------ phase D: colour backward (bf16)
    float nbar[3];
    {
      b8 dfeat[N::HK];
      b8 dof[1];
      dof[0] = zero_frag<b8>();
#pragma unroll
      for (int r = 0; r < 4; ++r) dof[0][r] = (__bf16)delta_o[r];
      pstore<b8, false>(pblk, live, L::P_DO, lane, dof[0], zero_frag<b8>(), e0b, e1b);
#define AVC_RELU_BWD(OUT, MSK, PT)                                                                         \
  AVC_EPI(const unsigned bits = MSK[t * 64];                                                                 \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                    \
            OUT[2 * t][j] = (__bf16)(((bits >> j) & 1u) ? acc[j] : 0.f);                                     \
            OUT[2 * t + 1][j] = (__bf16)(((bits >> (8 + j)) & 1u) ? acc[8 + j] : 0.f); }                     \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                  \
          pstore<b8, false>(pblk, live, (PT) + t, lane, OUT[2 * t], OUT[2 * t + 1], e0b, e1b);)
      b8 dl[N::HK];
      b8 d1[N::HK];
      if constexpr (N::NCMID == 1) {
        layer_s<b8, 1, N::HT>(sg, Wb, o.v[OFF_CHT], nxt<N, OFF_CM0T>(sg, Wb, o), dof,
                                     AVC_RELU_BWD(dl, m2, L::P_D2));
        layer_s<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_CM0T], nxt<N, OFF_C0T>(sg, Wb, o), dl,
                                         AVC_RELU_BWD(d1, m1, L::P_D1));
      } else {
        layer_s<b8, 1, N::HT>(sg, Wb, o.v[OFF_CHT], nxt<N, OFF_C0T>(sg, Wb, o), dof,
                                     AVC_RELU_BWD(d1, m1, L::P_D1));
      }
      // d r0 = C0^T delta1: HT feature tiles, then the [x,n] tile (rows 3,4,5 = d n)
      float dn_acc[3] = {0.f, 0.f, 0.f};
      layer_s<b8, N::HK, N::HT + 1>(sg, Wb, o.v[OFF_C0T], nxt<N, OFF_W0G>(sg, Wb, o), d1, AVC_EPI(
        if (t < N::HT) {
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {
            dfeat[2 * (t < N::HT ? t : 0)][j] = (__bf16)acc[j];
            dfeat[2 * (t < N::HT ? t : 0) + 1][j] = (__bf16)acc[8 + j];
          }
          pin2(dfeat[2 * (t < N::HT ? t : 0)], dfeat[2 * (t < N::HT ? t : 0) + 1]);
          pstore<b8, PANEL_KEEP>(pblk, live, L::P_DFEAT + t, lane, dfeat[2 * (t < N::HT ? t : 0)], dfeat[2 * (t < N::HT ? t : 0) + 1], e0b, e1b);
        } else {
          dn_acc[0] = acc[3]; dn_acc[1] = acc[0]; dn_acc[2] = acc[1];
        }
      ));
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
      b8* dd1 = pblk + (long)L::P_SDF * 128 + lane; AVC_NT_STORE(k0, &dd1[0]); AVC_NT_STORE(k1, &dd1[64]);
      b8* dd2 = pblk + (long)L::P_ONE * 128 + lane; AVC_NT_STORE(o0, &dd2[0]); AVC_NT_STORE(o1, &dd2[64]);
    }
    // ------------------------------------------------------------------ phase E: second-order sweep (i) (bf16)
    {
      b8 gb0[3];
      {
        PE pe4;
        pe_compute(x, h, pe4);
#pragma unroll
        for (int q = 0; q < 24; ++q) gb0[q >> 3][q & 7] = (__bf16)(pe4.d[q] * nbar[q % 3]);
      }
      pstore<b8, false>(pblk, live, L::P_GB0, lane, gb0[0], gb0[1], e0b, e1b);
      pstore<b8, false>(pblk, live, L::P_GB0 + 1, lane, gb0[2], zero_frag<b8>(), e0b, e1b);
      // gbar_a = W gbar_h(in); gbar_h(out) = gbar_a * sigma(h_out)
#define AVC_SECOND(OUT, PH, PT)                                                                             \
  AVC_PRE(return pfetch1(pblk, (PH) + t, lane);),                                                            \
  AVC_EPID(PF1, const facc hv = ptrans(d.a0, d.a1, e0b, e1b);                                                \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                    \
            OUT[2 * t][j] = (__bf16)(acc[j] * sig_from_h(hv[j]));                                            \
            OUT[2 * t + 1][j] = (__bf16)(acc[8 + j] * sig_from_h(hv[8 + j])); }                              \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                  \
          pstore<b8, PANEL_KEEP>(pblk, live, (PT) + t, lane, OUT[2 * t], OUT[2 * t + 1], e0b, e1b);)
      b8 gb1[N::HK];
      layer_sqd<b8, 3, N::HT>(sg, Wb, o.v[OFF_W0G], nxt<N, OFF_WM0>(sg, Wb, o), gb0,
                                   AVC_SECOND(gb1, L::P_H1, L::P_GBH1));
      b8 gbm[N::HK];
      b8 gbs[N::SK];
      if constexpr (N::NMID == 2) {
        layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wb, o), gb1,
                                         AVC_SECOND(gbm, L::P_HM, L::P_GBHM));
        b8 gbm1[N::HK];
        layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wb, o), gbm,
                                         AVC_SECOND(gbm1, L::P_HM + N::HT, L::P_GBHM + N::HT));
        layer_sqd<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WS], nxt<N, OFF_WLT>(sg, Wb, o), gbm1,
                                         AVC_SECOND(gbs, L::P_HS, L::P_GBHS));
      } else {
        layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wb, o), gb1,
                                         AVC_SECOND(gbm, L::P_HM, L::P_GBHM));
        layer_sqd<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WS], nxt<N, OFF_WLT>(sg, Wb, o), gbm,
                                         AVC_SECOND(gbs, L::P_HS, L::P_GBHS));
      }
    }
    // ------------------------------------------------------------------ phase F: reverse sweep (ii) (bf16)
    {
      b8 as_[N::SK];
      b8 dfeat[N::HK];
#pragma unroll
      for (int t = 0; t < N::HT; ++t) punpack_frags<b8>(pblk, L::P_DFEAT + t, lane, e0b, e1b, dfeat[2 * t], dfeat[2 * t + 1]);
      // ubar[:SKIP]/sqrt2 = (W_last[1:,:]^T dfeat + W_last[0,:] d_sdf)/sqrt2 ; 1/sqrt2 is folded into both packs
      layer_sq<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WLT], nxt<N, OFF_WST>(sg, Wb, o), dfeat,
        AVC_PRE(return pfetch3(pblk, L::P_HS + t, L::P_GBHS + t, L::P_GAS + t, lane);), AVC_EPID(PF3,
        float wa[16];
        load16(T + o.v[OFF_WL0_ACC], t, h, wa);
        const facc hv = ptrans(d.a0, d.a1, e0b, e1b);
        const facc bv = ptrans(d.b0, d.b1, e0b, e1b);
        const facc gv = ptrans(d.c0, d.c1, e0b, e1b);
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {
          const float s0 = sig_from_h(hv[j]), s1 = sig_from_h(hv[8 + j]);
          as_[2 * t][j] = (__bf16)(second_term(bv[j], gv[j], s0) + (acc[j] + wa[j] * dsdfS) * s0);
          as_[2 * t + 1][j] = (__bf16)(second_term(bv[8 + j], gv[8 + j], s1) + (acc[8 + j] + wa[8 + j] * dsdfS) * s1);
        }
        pin2(as_[2 * t], as_[2 * t + 1]);
        pstore<b8, false>(pblk, live, L::P_ABS + t, lane, as_[2 * t], as_[2 * t + 1], e0b, e1b);
      ));
      // hbar(prev) = W^T abar(cur); abar(prev) = abar'(prev) + hbar * sigma(h_prev)
#define AVC_REVERSE(OUT, PH, PB, PG, PT)                                                                    \
  AVC_PRE(return pfetch3(pblk, (PH) + t, (PB) + t, (PG) + t, lane);),                                        \
  AVC_EPID(PF3, const facc hv = ptrans(d.a0, d.a1, e0b, e1b);                                                \
          const facc bv = ptrans(d.b0, d.b1, e0b, e1b);                                                      \
          const facc gv = ptrans(d.c0, d.c1, e0b, e1b);                                                      \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                    \
            const float s0 = sig_from_h(hv[j]), s1 = sig_from_h(hv[8 + j]);                                  \
            OUT[2 * t][j] = (__bf16)(second_term(bv[j], gv[j], s0) + acc[j] * s0);                           \
            OUT[2 * t + 1][j] = (__bf16)(second_term(bv[8 + j], gv[8 + j], s1) + acc[8 + j] * s1); }         \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                  \
          pstore<b8, false>(pblk, live, (PT) + t, lane, OUT[2 * t], OUT[2 * t + 1], e0b, e1b);)
      b8 am[N::HK];
      b8 am0[N::HK];
      const Next first = nxt<N, OFF_W0>(sg, Wf0, o);   // prefetch the first tile of the next block iteration
      if constexpr (N::NMID == 2) {
        layer_sq<b8, N::SK, N::HT>(sg, Wb, o.v[OFF_WST], nxt<N, OFF_WM1T>(sg, Wb, o), as_,
                                         AVC_REVERSE(am, L::P_HM + N::HT, L::P_GBHM + N::HT, L::P_GAM + N::HT, L::P_ABM + N::HT));
        layer_sq<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM1T], nxt<N, OFF_WM0T>(sg, Wb, o), am,
                                         AVC_REVERSE(am0, L::P_HM, L::P_GBHM, L::P_GAM, L::P_ABM));
        layer_sq<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0T], first, am0, AVC_REVERSE(am, L::P_H1, L::P_GBH1, L::P_GA1, L::P_AB1));
      } else {
        layer_sq<b8, N::SK, N::HT>(sg, Wb, o.v[OFF_WST], nxt<N, OFF_WM0T>(sg, Wb, o), as_,
                                         AVC_REVERSE(am, L::P_HM, L::P_GBHM, L::P_GAM, L::P_ABM));
        layer_sq<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0T], first, am, AVC_REVERSE(am0, L::P_H1, L::P_GBH1, L::P_GA1, L::P_AB1));
      }
    }
  }
}

extern "C" int avc_render_points_bwd(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                                     int S, int ldz, float sample_dist, long npts, const void* wf16, const void* wbf16,
                                     const float* tab, const int* offs, const float* d_sdf, const float* d_normal,
                                     const float* d_rgb, void* panels, long max_waves, void* stream) {
  if (npts <= 0) return 0;
  AvcOffsets o;
  for (int k = 0; k < OFF_COUNT; ++k) o.v[k] = offs[k];
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, pts ? 0 : 1, sample_dist};
  const long nblk = (npts + 31) / 32;
  long ngroups = (nblk + BWD_WPB - 1) / BWD_WPB;
  long maxg = max_waves / BWD_WPB;
  if (maxg < 1) maxg = 1;
  int grid = (int)(ngroups < maxg ? ngroups : maxg);
  if (grid < 1) grid = 1;
  hipStream_t s = (hipStream_t)stream;
  const int lds_bytes = StageT<BWD_G>::LDS_BYTES + BWD_WPB * BWD_MASK_BYTES + AVC_TAB_LDS_BYTES;
  if (offs[OFF_TAB_END] * 4 > AVC_TAB_LDS_BYTES) { avc_set_error("fp32 table does not fit its LDS window"); return 1; }
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)mlp_bwd_kernel<NetFull>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipFuncSetAttribute((const void*)mlp_bwd_kernel<NetSmall>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set = true;
  }
  if (net == AVC_NET_FULL)
    hipLaunchKernelGGL((mlp_bwd_kernel<NetFull>), dim3(grid), dim3(64 * BWD_WPB), lds_bytes, s, ps, npts, (const h8*)wf16, (const b8*)wbf16,
                       tab, o, d_sdf, d_normal, d_rgb, (b8*)panels);
  else if (net == AVC_NET_SMALL)
    hipLaunchKernelGGL((mlp_bwd_kernel<NetSmall>), dim3(grid), dim3(64 * BWD_WPB), lds_bytes, s, ps, npts, (const h8*)wf16, (const b8*)wbf16,
                       tab, o, d_sdf, d_normal, d_rgb, (b8*)panels);
  else { avc_set_error("unknown net id"); return 1; }
  return avc_check_launch("avc_render_points_bwd");
}

