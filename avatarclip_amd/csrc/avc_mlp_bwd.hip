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
#include "avc_bwd_body.h"
#include "../../include/avc.h"

template <class N>
__global__ __launch_bounds__(64 * BWD_WPB) void mlp_bwd_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef StageT<BWD_G> ST;
  constexpr AvcOffsets o = Off<N>::value;
  avc_static_wave_priority();
  const int lane0 = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const long nblk = (a.npts + 31) >> 5;
  ST sg = stage_init<BWD_G>(lds);
  stage_issue(sg, nxt<N, OFF_CHT>(sg, a.Wb0, o), 0);
  // the fp32 table lives in LDS: a global load in an epilogue would queue behind the LDS-DMA of the next weight group
  const lds_tab_t Tl = tab_to_lds(lds + ST::LDS_BYTES, a.T0, o.v[OFF_TAB_END]);
  const cs_slot_t cs = cs_init<N>(lds + ST::LDS_BYTES + AVC_TAB_LDS_BYTES, wv, lane0);   // this wavefront's column-sum slot (csrc/avc_bwd_body.h)
  __syncthreads();
  NoRing ring;
  // every wavefront of a workgroup runs the same number of iterations (workgroup-uniform loop bound)
#if AVC_BWD_PIPE_IN
  BlkIn<N> bi;     // the block's first inputs, requested one block ahead (csrc/avc_bwd_body.h)
  load_blk_in<N>(a, (long)blockIdx.x * BWD_WPB + wv, nblk, lane0, bi);
  for (long blk0 = (long)blockIdx.x * BWD_WPB; blk0 < nblk; blk0 += (long)gridDim.x * BWD_WPB)
    bwd_sweeps<N, true>(sg, a, Tl, blk0, nblk, lane0, wv, ring, cs, &bi, blk0 + (long)gridDim.x * BWD_WPB);
#else
  for (long blk0 = (long)blockIdx.x * BWD_WPB; blk0 < nblk; blk0 += (long)gridDim.x * BWD_WPB)
    bwd_sweeps<N>(sg, a, Tl, blk0, nblk, lane0, wv, ring, cs);
#endif
  cs_flush<N>(cs, a.colsum, (long)blockIdx.x * BWD_WPB + wv, lane0);
}

// rows x floats of the colsum buffer a launch with `max_waves` fills: one row per wavefront of the (persistent) grid
extern "C" int avc_bwd_colsum_floats(int net) { return net == AVC_NET_FULL ? PanelLayout<NetFull>::CS_FLOATS : PanelLayout<NetSmall>::CS_FLOATS; }
extern "C" long avc_bwd_colsum_rows(long npts, long max_waves) {
  const long nblk = (npts + 31) / 32;
  long ngroups = (nblk + BWD_WPB - 1) / BWD_WPB, maxg = max_waves / BWD_WPB;
  if (maxg < 1) maxg = 1;
  long grid = ngroups < maxg ? ngroups : maxg;
  if (grid < 1) grid = 1;
  return grid * BWD_WPB;
}

extern "C" int avc_render_points_bwd(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                                     int S, int ldz, float sample_dist, long npts, const void* wbf16, const float* tab,
                                     const int* offs, const float* d_sdf, const float* d_normal, const float* d_rgb,
                                     const float* rgb_fwd, const void* fpanels, void* gpanels, const void* masks, float* colsum,
                                     long max_waves, void* stream) {
  if (npts <= 0) return 0;
  if (!fpanels || !gpanels || !masks || !rgb_fwd || !colsum) { avc_set_error("avc_render_points_bwd: fpanels / gpanels / masks / rgb_fwd / colsum == NULL"); return 1; }
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
  const int lds_bytes = StageT<BWD_G>::LDS_BYTES + AVC_TAB_LDS_BYTES + ColSum<NetFull>::LDS_BYTES;
  if (offs[OFF_TAB_END] * 4 > AVC_TAB_LDS_BYTES) { avc_set_error("fp32 table does not fit its LDS window"); return 1; }
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen)) {
    (void)hipFuncSetAttribute((const void*)mlp_bwd_kernel<NetFull>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void*)mlp_bwd_kernel<NetSmall>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  }
  const BwdArgs args{ps, npts, (const b8*)wbf16, tab, d_sdf, d_normal, d_rgb, rgb_fwd, (const char*)fpanels, (char*)gpanels,
                     (const unsigned short*)masks, colsum};
  if (net == AVC_NET_FULL)
    hipLaunchKernelGGL((mlp_bwd_kernel<NetFull>), dim3(grid), dim3(64 * BWD_WPB), lds_bytes, s, args);
  else if (net == AVC_NET_SMALL)
    hipLaunchKernelGGL((mlp_bwd_kernel<NetSmall>), dim3(grid), dim3(64 * BWD_WPB), lds_bytes, s, args);
  else { avc_set_error("unknown net id"); return 1; }
  return avc_check_launch("avc_render_points_bwd");
}
