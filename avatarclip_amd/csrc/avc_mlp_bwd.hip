// Backward of the fused SDF + colour MLP wrt every dense weight, incl. the double backward through
// SDFNetwork.gradient (fields.py:96-107; autograd at main.py:537).  Mathematics: SURVEY.md A.1/A.2, proven
// against torch.autograd in tests/test_analytic.py (oracle/analytic.py: mlp_backward).
//
//   avc_render_points_bwd : one wavefront per 32 points.  NOTHING of the forward pass is recomputed: the forward kernel
//        (avc_render_points_fwd_train) left h_l, g_a,l, the ReLU masks and the colours in the block's operand panels
//        (csrc/avc_mlp.h: PanelLayout, F region; this kernel writes the G region of the current slab).  This kernel runs the colour backward (phase D), the second-order sweep (i) (phase E)
//        and the reverse sweep (ii) (phase F) on bf16 operands with fp32 accumulation, reads sigma's argument / g_a / gbar_h
//        back from the panels as fragments (no transposition) and writes the gradient-type operands of the weight-gradient
//        products (gbar_h, abar, delta, ybar) next to them.  The second-order term abar' is not stored: the reverse sweep
//        rebuilds it from the gbar_h, g_a and h tiles (abar' = gbar_h g_a beta (1-s)/s).
//   avc_weight_grad (csrc/avc_wgrad.hip): dW[a,b] += sum_points A[p,a] B[p,b], K = points, straight from the panels.
//
// Round-1 version of this file recomputed the forward and the normal sweep here (25 layer sweeps per block, 24.5 KiB/point of
// HBM traffic incl. transposed panel writes): 13 layer sweeps and ~14 KiB/point now.
#include "avc_mlp.h"
#ifndef BWD_G
#define BWD_G 4   // tiles per staged group (LDS = 2 * G * 16 KiB + table: one 8-wave workgroup per CU)
#endif
#ifndef BWD_WPB
#define BWD_WPB 8   // wavefronts per workgroup: every staged weight tile is shared by 256 points (LDS-DMA fill rate is the scarce resource)
#endif
#include "../../include/avc.h"
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

template <class N>
__global__ __launch_bounds__(64 * BWD_WPB) void mlp_bwd_kernel(PointSrc ps, long npts, const b8* __restrict__ Wb0,
                                                               const float* __restrict__ T0,
                                                               const float* __restrict__ d_sdf, const float* __restrict__ d_normal,
                                                               const float* __restrict__ d_rgb, const float* __restrict__ rgb_fwd,
                                                               const char* __restrict__ fpanels, char* __restrict__ gpanels,
                                                               const unsigned short* __restrict__ masks) {
  typedef PanelLayout<N> L;
  constexpr AvcOffsets o = Off<N>::value;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef StageT<BWD_G> ST;
  avc_static_wave_priority();
  const int lane0 = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const long nblk = (npts + 31) >> 5;
  ST sg = stage_init<BWD_G>(lds);
  stage_issue(sg, nxt<N, OFF_CHT>(sg, Wb0, o), 0);
  // the fp32 table lives in LDS: a global load in an epilogue would queue behind the LDS-DMA of the next weight group
  const lds_tab_t Tl = tab_to_lds(lds + ST::LDS_BYTES, T0, o.v[OFF_TAB_END]);
  __syncthreads();

  // every wavefront of a workgroup runs the same number of iterations (workgroup-uniform loop bound)
  for (long blk0 = (long)blockIdx.x * BWD_WPB; blk0 < nblk; blk0 += (long)gridDim.x * BWD_WPB) {
    const b8* Wb = launder(Wb0);
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
    const PanelPtr ftiles = panel_ptr(const_cast<char*>(fpanels) + bsel * (long)L::P_TILES * 2048, lane);   // forward-type operands: read only
    const PanelPtr tiles = panel_ptr(gpanels + bsel * (long)L::G_TILES * 2048, lane);                       // gradient-type operands of this slab
    const AVC_GLOBAL unsigned short* mk = as_global(masks) + (blk < nblk ? blk : nblk) * (long)L::MASK_U16 + lane;
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
      b8 dof[1];
      dof[0] = zero_frag<b8>();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = h ? 4 + r : r;
        const float c = (ch < 6) ? rgb_fwd[6 * i + (ch < 6 ? ch : 0)] : 0.f;
        const float dr = (ch < 6) ? d_rgb[6 * i + (ch < 6 ? ch : 0)] * vmask : 0.f;
        dof[0][r] = (__bf16)(dr * c * (1.f - c));
      }
      tile_store<false>(tiles, L::G_DO, dof[0], zero_frag<b8>());
      // ReLU masks of r1 / r2 (16 bits per tile and lane, written by the forward kernel: accumulator register r at bit relu_mask_bit(r))
      unsigned m1[N::HT], m2[N::HT];
#pragma unroll
      for (int t = 0; t < N::HT; ++t) {
        m1[t] = mk[t * 64];
        m2[t] = (N::NCMID == 1) ? mk[(N::HT + t) * 64] : 0u;
      }
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
        nbar[0] = d_normal[3 * i + 0] * vmask + (h ? o3 : a3);
        nbar[1] = d_normal[3 * i + 1] * vmask + (h ? a0 : o0);
        nbar[2] = d_normal[3 * i + 2] * vmask + (h ? a1 : o1);
      }
    }
    const float dsdf = d_sdf[i] * vmask;
    const float dsdfS = dsdf * AVC_S;   // OFF_WL0_ACC holds W_last[0,:]/(S sqrt2): undo S for the gradient use
    {   // operand tiles with a single live feature (slot (half 0, j = 0) = feature 0): d_sdf and the constant 1 (row 0 of the last layer)
      b8 fs = zero_frag<b8>(), fo = zero_frag<b8>();
      if (h == 0) { fs[0] = (__bf16)dsdf; fo[0] = (__bf16)vmask; }
      tile_store<false>(tiles, L::G_SDF, fs, zero_frag<b8>());
      tile_store<false>(tiles, L::G_ONE, fo, zero_frag<b8>());
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
      tile_store<false>(tiles, L::G_GB0, gb0[0], gb0[1]);
      tile_store<false>(tiles, L::G_GB0 + 1, gb0[2], zero_frag<b8>());
      // gbar_a = W gbar_h(in); gbar_h(out) = gbar_a * sigma(h_out)
#define AVC_SECOND(OUT, PH, PT)                                                                             \
  AVC_PRE(return tile_load<(AVC_BWD_E_NT != 0), h8>(ftiles, (PH) + t);),                                     \
  AVC_EPID(FragPair<h8>, _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                     \
            OUT[2 * t][j] = (__bf16)(acc[j] * sig_from_h((float)d.a0[j]));                                   \
            OUT[2 * t + 1][j] = (__bf16)(acc[8 + j] * sig_from_h((float)d.a1[j])); }                         \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                  \
          tile_store<true>(tiles, (PT) + t, OUT[2 * t], OUT[2 * t + 1]);)
      b8 gb1[N::HK];
      layer_sqd<b8, 3, N::HT>(sg, Wb, o.v[OFF_W0G], nxt<N, OFF_WM0>(sg, Wb, o), gb0, AVC_SECOND(gb1, L::P_H1, L::G_GBH1));
      b8 gbm[N::HK];
      b8 gbs[N::SK];
      if constexpr (N::NMID == 2) {
        layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wb, o), gb1, AVC_SECOND(gbm, L::P_HM, L::G_GBHM));
        b8 gbm1[N::HK];
        layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wb, o), gbm,
                                    AVC_SECOND(gbm1, L::P_HM + N::HT, L::G_GBHM + N::HT));
        layer_sqd<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WS], nxt<N, OFF_WLT>(sg, Wb, o), gbm1, AVC_SECOND(gbs, L::P_HS, L::G_GBHS));
      } else {
        layer_sqd<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wb, o), gb1, AVC_SECOND(gbm, L::P_HM, L::G_GBHM));
        layer_sqd<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WS], nxt<N, OFF_WLT>(sg, Wb, o), gbm, AVC_SECOND(gbs, L::P_HS, L::G_GBHS));
      }
    }
    // ------------------------------------------------------------------ phase F: reverse sweep (ii) (bf16)
    {
      b8 as_[N::SK];
      b8 dfeat[N::HK];
#pragma unroll
      for (int t = 0; t < N::HT; ++t) {
        const FragPair<b8> d = tile_load<(AVC_BWD_RR_NT != 0), b8>(tiles, L::G_DFEAT + t);
        dfeat[2 * t] = d.a0;
        dfeat[2 * t + 1] = d.a1;
      }
#define AVC_LOAD3(PH, PB, PG)                                                                               \
  AVC_PRE(PF3 d; { const FragPair<h8> a = tile_load<true, h8>(ftiles, (PH) + t); d.h0 = a.a0; d.h1 = a.a1; } \
          { const FragPair<b8> a = tile_load<(AVC_BWD_RR_NT != 0), b8>(tiles, (PB) + t); d.b0 = a.a0; d.b1 = a.a1; } \
          { const FragPair<h8> a = tile_load<true, h8>(ftiles, (PG) + t); d.g0 = a.a0; d.g1 = a.a1; } return d;)
      // ubar[:SKIP]/sqrt2 = (W_last[1:,:]^T dfeat + W_last[0,:] d_sdf)/sqrt2 ; 1/sqrt2 is folded into both packs
      layer_sq<b8, N::HK, N::ST>(sg, Wb, o.v[OFF_WLT], nxt<N, OFF_WST>(sg, Wb, o), dfeat,
        AVC_LOAD3(L::P_HS, L::G_GBHS, L::P_GAS), AVC_EPID(PF3,
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
      // hbar(prev) = W^T abar(cur); abar(prev) = abar'(prev) + hbar * sigma(h_prev)
#define AVC_REVERSE(OUT, PH, PB, PG, PT)                                                                    \
  AVC_LOAD3(PH, PB, PG),                                                                                     \
  AVC_EPID(PF3, _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                              \
            const float s0 = sig_from_h((float)d.h0[j]), s1 = sig_from_h((float)d.h1[j]);                    \
            OUT[2 * t][j] = (__bf16)(second_term((float)d.b0[j], (float)d.g0[j], s0) + acc[j] * s0);         \
            OUT[2 * t + 1][j] = (__bf16)(second_term((float)d.b1[j], (float)d.g1[j], s1) + acc[8 + j] * s1); } \
          pin2(OUT[2 * t], OUT[2 * t + 1]);                                                                  \
          tile_store<false>(tiles, (PT) + t, OUT[2 * t], OUT[2 * t + 1]);)
      b8 am[N::HK];
      b8 am0[N::HK];
      const Next first = nxt<N, OFF_CHT>(sg, Wb0, o);   // prefetch the first tile of the next block iteration
      if constexpr (N::NMID == 2) {
        layer_sq<b8, N::SK, N::HT>(sg, Wb, o.v[OFF_WST], nxt<N, OFF_WM1T>(sg, Wb, o), as_,
                                   AVC_REVERSE(am, L::P_HM + N::HT, L::G_GBHM + N::HT, L::P_GAM + N::HT, L::G_ABM + N::HT));
        layer_sq<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM1T], nxt<N, OFF_WM0T>(sg, Wb, o), am,
                                   AVC_REVERSE(am0, L::P_HM, L::G_GBHM, L::P_GAM, L::G_ABM));
        layer_sq<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0T], first, am0, AVC_REVERSE(am, L::P_H1, L::G_GBH1, L::P_GA1, L::G_AB1));
      } else {
        layer_sq<b8, N::SK, N::HT>(sg, Wb, o.v[OFF_WST], nxt<N, OFF_WM0T>(sg, Wb, o), as_,
                                   AVC_REVERSE(am, L::P_HM, L::G_GBHM, L::P_GAM, L::G_ABM));
        layer_sq<b8, N::HK, N::HT>(sg, Wb, o.v[OFF_WM0T], first, am, AVC_REVERSE(am0, L::P_H1, L::G_GBH1, L::P_GA1, L::G_AB1));
      }
    }
  }
}

extern "C" int avc_render_points_bwd(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                                     int S, int ldz, float sample_dist, long npts, const void* wbf16, const float* tab,
                                     const int* offs, const float* d_sdf, const float* d_normal, const float* d_rgb,
                                     const float* rgb_fwd, const void* fpanels, void* gpanels, const void* masks, long max_waves,
                                     void* stream) {
  if (npts <= 0) return 0;
  if (!fpanels || !gpanels || !masks || !rgb_fwd) { avc_set_error("avc_render_points_bwd: fpanels / gpanels / masks / rgb_fwd == NULL"); return 1; }
  if (!(net == AVC_NET_FULL ? offsets_match<NetFull>(offs) : offsets_match<NetSmall>(offs))) {
    avc_set_error("packed-blob offsets differ from the compiled-in table (regenerate csrc/avc_offsets_gen.h)");
    return 1;
  }
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, pts ? 0 : 1, sample_dist};
  const long nblk = (npts + 31) / 32;
  long ngroups = (nblk + BWD_WPB - 1) / BWD_WPB;
  long maxg = max_waves / BWD_WPB;
  if (maxg < 1) maxg = 1;
  int grid = (int)(ngroups < maxg ? ngroups : maxg);
  if (grid < 1) grid = 1;
  hipStream_t s = (hipStream_t)stream;
  const int lds_bytes = StageT<BWD_G>::LDS_BYTES + AVC_TAB_LDS_BYTES;
  if (offs[OFF_TAB_END] * 4 > AVC_TAB_LDS_BYTES) { avc_set_error("fp32 table does not fit its LDS window"); return 1; }
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen)) {
    (void)hipFuncSetAttribute((const void*)mlp_bwd_kernel<NetFull>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void*)mlp_bwd_kernel<NetSmall>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  }
  if (net == AVC_NET_FULL)
    hipLaunchKernelGGL((mlp_bwd_kernel<NetFull>), dim3(grid), dim3(64 * BWD_WPB), lds_bytes, s, ps, npts, (const b8*)wbf16, tab, d_sdf,
                       d_normal, d_rgb, rgb_fwd, (const char*)fpanels, (char*)gpanels, (const unsigned short*)masks);
  else if (net == AVC_NET_SMALL)
    hipLaunchKernelGGL((mlp_bwd_kernel<NetSmall>), dim3(grid), dim3(64 * BWD_WPB), lds_bytes, s, ps, npts, (const b8*)wbf16, tab, d_sdf,
                       d_normal, d_rgb, rgb_fwd, (const char*)fpanels, (char*)gpanels, (const unsigned short*)masks);
  else { avc_set_error("unknown net id"); return 1; }
  return avc_check_launch("avc_render_points_bwd");
}
