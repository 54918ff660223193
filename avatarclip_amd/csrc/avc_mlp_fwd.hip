// Forward kernels of the SDF + colour MLP (gfx950, f16 MFMA, fp32 accumulate).
//   avc_sdf_forward       SDF only (row 0 of the last layer) -- the no-grad evaluations of the hierarchical sampler
//                         (renderer.py:337-338,187) and of extract_fields (renderer.py:10-25)
//   avc_render_points_fwd sdf + normal (d sdf/dx) + 6 colour channels per sample point (renderer.py:221-232)
#include "avc_mlp.h"
#ifndef FWD_WPB
#define FWD_WPB 8   // one 8-wave workgroup per CU shares every staged weight tile (LDS-DMA fill rate is the scarce resource)
#endif
#ifndef FWD_G
#define FWD_G 4
#endif
#include "../../include/avc.h"

template <class N, int MODE>
__global__ __launch_bounds__(64 * FWD_WPB) void mlp_fwd_kernel(PointSrc ps, long npts, const h8* __restrict__ Wf,
                                                                       const float* __restrict__ T, AvcOffsets o,
                                                                       float* __restrict__ sdf_out, const int* __restrict__ slot,
                                                                       int ld_out, float* __restrict__ normal_out,
                                                                       float* __restrict__ rgb_out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef StageT<FWD_G> ST;
  const int lane = threadIdx.x & 63;
  const int h = lane >> 5;
  const int p = lane & 31;
  // one wavefront per 32-point block; the wavefronts of a workgroup share every weight tile through LDS, so no wave
  // may leave early: blocks past the end are clamped to the last point and only their stores are suppressed.
  // (No persistent loop: LICM would hoist the loop-invariant weight traffic and spill.)
  const long blk = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  long i = blk * 32 + p;
  const bool valid = i < npts;
  if (!valid) i = npts - 1;
  ST sg = stage_init<ST::G>(lds);
  stage_issue(sg, nxt<N, OFF_W0>(sg, Wf, o), 0);
  float x0[3];
  fetch_point(ps, i, x0);
  long oi = i;
  if (slot) {
    const long ray = i / ps.S;
    oi = ray * ld_out + slot[i];
  }
  if (MODE == 0) {
    const float sdfv = sdf_only<N>(sg, Wf, T, o, h, x0);
    if (valid && h == 0) sdf_out[oi] = sdfv;
    return;
  }
  FwdState<N> st;
  st.x[0] = x0[0]; st.x[1] = x0[1]; st.x[2] = x0[2];
  sdf_trunk<N>(sg, Wf, T, o, h, st, nxt<N, OFF_WL>(sg, Wf, o));
  h8 feat[N::HK];
  sdf_feature<N>(sg, Wf, T, o, h, st, feat, nxt<N, OFF_WST>(sg, Wf, o));
  float n[3];
  sdf_normal<N>(sg, Wf, T, o, h, st, n, nxt<N, OFF_C0>(sg, Wf, o));
  float rgb[4];
  color_forward<N>(sg, Wf, T, o, h, st.x, n, feat, rgb);
  if (valid) {
    if (h == 0) {
      sdf_out[oi] = st.sdf;
      normal_out[3 * oi + 0] = n[0]; normal_out[3 * oi + 1] = n[1]; normal_out[3 * oi + 2] = n[2];
      rgb_out[6 * oi + 0] = rgb[0]; rgb_out[6 * oi + 1] = rgb[1]; rgb_out[6 * oi + 2] = rgb[2]; rgb_out[6 * oi + 3] = rgb[3];
    } else {
      rgb_out[6 * oi + 4] = rgb[0]; rgb_out[6 * oi + 5] = rgb[1];
    }
  }
}

static int grid_for(long npts, int waves_per_block, int max_blocks) {
  long nblk = (npts + 31) / 32;
  long g = (nblk + waves_per_block - 1) / waves_per_block;
  if (g > max_blocks) g = max_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

template <int MODE>
static int launch_fwd(int net, PointSrc ps, long npts, const void* wf, const float* tab, const int* offs,
                      float* sdf_out, const int* slot, int ld_out, float* normal_out, float* rgb_out, void* stream) {
  if (npts <= 0) return 0;
  AvcOffsets o;
  for (int k = 0; k < OFF_COUNT; ++k) o.v[k] = offs[k];
  hipStream_t s = (hipStream_t)stream;
  const int wpb = FWD_WPB;   // wavefronts per workgroup
  const int grid = grid_for(npts, wpb, 0x7fffffff);
  const int lds_bytes = StageT<FWD_G>::LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)mlp_fwd_kernel<NetFull, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipFuncSetAttribute((const void*)mlp_fwd_kernel<NetSmall, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set = true;
  }
  if (net == AVC_NET_FULL)
    hipLaunchKernelGGL((mlp_fwd_kernel<NetFull, MODE>), dim3(grid), dim3(64 * wpb), lds_bytes, s, ps, npts, (const h8*)wf, tab, o,
                       sdf_out, slot, ld_out, normal_out, rgb_out);
  else if (net == AVC_NET_SMALL)
    hipLaunchKernelGGL((mlp_fwd_kernel<NetSmall, MODE>), dim3(grid), dim3(64 * wpb), lds_bytes, s, ps, npts, (const h8*)wf, tab, o,
                       sdf_out, slot, ld_out, normal_out, rgb_out);
  else {
    avc_set_error("unknown net id");
    return 1;
  }
  return avc_check_launch(MODE ? "avc_render_points_fwd" : "avc_sdf_forward");
}

extern "C" int avc_sdf_forward(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                               int S, int ldz, long npts, const void* wf16, const float* tab, const int* offs,
                               float* sdf_out, const int* slot, int ld_out, void* stream) {
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, 0, 0.f};
  return launch_fwd<0>(net, ps, npts, wf16, tab, offs, sdf_out, slot, ld_out, nullptr, nullptr, stream);
}

extern "C" int avc_render_points_fwd(int net, const float* pts, const float* rays_o, const float* rays_d,
                                     const float* z, int S, int ldz, float sample_dist, long npts, const void* wf16,
                                     const float* tab, const int* offs, float* sdf_out, float* normal_out,
                                     float* rgb_out, void* stream) {
  PointSrc ps{pts, rays_o, rays_d, z, S, ldz, pts ? 0 : 1, sample_dist};
  return launch_fwd<1>(net, ps, npts, wf16, tab, offs, sdf_out, nullptr, 0, normal_out, rgb_out, stream);
}
