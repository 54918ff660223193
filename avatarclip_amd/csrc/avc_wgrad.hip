// Weight-gradient GEMM of the NeuS MLP backward (main.py:537): dW = sum_points A^T B from the bf16 operand panels
// written by avc_render_points_bwd (csrc/avc_mlp_bwd.hip).
#include "avc_common.h"
#include "../../include/avc.h"

// ---------------------------------------------------------------------------------------------
// weight-gradient GEMM: partial[split][ta][tb] = sum over the split's 32-point blocks, sum_kappa A[blk][ta][kappa] x B[blk][tb][kappa]
// One 8-wave workgroup per K-split.  Per block the (ta+tb) panel tiles are copied ONCE into LDS by global->LDS DMA
// (double buffered, the copy of block b+1 runs under the MFMAs of block b); wave w owns A tile w and all tb B tiles
// (<= 9 accumulators).  Partials are written with plain stores and summed on the host side of the ABI (no atomics).
// HBM-bound by construction: 2 KiB per tile per block is read exactly once.
// ---------------------------------------------------------------------------------------------
#define WG_TB_MAX 9
#define WG_BUF_BYTES (17 * 2048)

__device__ __forceinline__ void weight_grad_body(char* lds, const b8* __restrict__ panels, int ptiles, int pa, int ta_n, int pb,
                                                 int tb_n, long nblk, float* __restrict__ partial,
                                                 float* __restrict__ bias_partial, int out_elems, int bias_elems) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int split = blockIdx.x, nsplit = gridDim.x;   // (the pair index, if any, is blockIdx.y)
  const long b0 = nblk * split / nsplit, b1 = nblk * (split + 1) / nsplit;
  const int nchunk = (ta_n + tb_n) * 2;
  auto issue = [&](long blk, int buf) {
    const char* base = reinterpret_cast<const char*>(panels + blk * (long)ptiles * 128);
    for (int c = wv; c < nchunk; c += 8) {
      const int tix = c >> 1;
      const int tile = tix < ta_n ? pa + tix : pb + (tix - ta_n);
      const char* g = base + ((long)tile * 128 + (c & 1) * 64 + lane) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(lds + buf * WG_BUF_BYTES + c * 1024), 16, 0, 0);
    }
  };
  facc acc[WG_TB_MAX];
#pragma unroll
  for (int q = 0; q < WG_TB_MAX; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float bsum = 0.f;
  const bool own = wv < ta_n;
  if (b0 < b1) issue(b0, 0);
  int par = 0;
  for (long blk = b0; blk < b1; ++blk) {
    __syncthreads();
    if (blk + 1 < b1) issue(blk + 1, par ^ 1);
    if (own) {
      const b8* L = reinterpret_cast<const b8*>(lds + par * WG_BUF_BYTES) + lane;
      const b8 a0 = L[(wv * 2) * 64], a1 = L[(wv * 2 + 1) * 64];
      if (bias_partial) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += (float)a0[j] + (float)a1[j];
      }
#pragma unroll
      for (int q = 0; q < WG_TB_MAX; ++q) {
        if (q < tb_n) {
          const b8 v0 = L[((ta_n + q) * 2) * 64], v1 = L[((ta_n + q) * 2 + 1) * 64];
          acc[q] = MF<b8>::mma(a0, v0, acc[q]);
          acc[q] = MF<b8>::mma(a1, v1, acc[q]);
        }
      }
    }
    par ^= 1;
  }
  if (own) {
    float* dst = partial + (long)split * out_elems;
#pragma unroll
    for (int q = 0; q < WG_TB_MAX; ++q) {
      if (q < tb_n) {
        f4* d4 = reinterpret_cast<f4*>(dst + ((long)(wv * tb_n + q) * 64 + lane) * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f4 v; v[0] = acc[q][4 * k]; v[1] = acc[q][4 * k + 1]; v[2] = acc[q][4 * k + 2]; v[3] = acc[q][4 * k + 3];
          d4[k] = v;
        }
      }
    }
    if (bias_partial) {
      bsum = xhalf_sum(bsum);
      if (lane < 32) bias_partial[(long)split * bias_elems + wv * 32 + lane] = bsum;
    }
  }
}

__global__ __launch_bounds__(512) void weight_grad_kernel(const b8* __restrict__ panels, int ptiles, int pa, int ta_n, int pb,
                                                          int tb_n, long nblk, float* __restrict__ partial,
                                                          float* __restrict__ bias_partial, int out_elems, int bias_elems) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  weight_grad_body(lds, panels, ptiles, pa, ta_n, pb, tb_n, nblk, partial, bias_partial, out_elems, bias_elems);
}

// every product of one backward pass in ONE launch: blockIdx.y = pair, blockIdx.x = K-split.  Workgroups are dispatched
// x-fastest, so the tail of one pair's splits overlaps the head of the next pair's instead of draining the chip 17 times.
#define WG_MAX_PAIRS 24
struct WgPairs { int v[WG_MAX_PAIRS][6]; };   // pa, ta, pb, tb, out_off, bias_off (-1 = no bias)
__global__ __launch_bounds__(512) void weight_grad_all_kernel(const b8* __restrict__ panels, int ptiles, WgPairs pp, long nblk,
                                                              float* __restrict__ partial, float* __restrict__ bias_partial,
                                                              int out_elems, int bias_elems) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int* d = pp.v[blockIdx.y];
  weight_grad_body(lds, panels, ptiles, d[0], d[1], d[2], d[3], nblk, partial + d[4], d[5] >= 0 ? bias_partial + d[5] : nullptr,
                   out_elems, bias_elems);
}

extern "C" int avc_weight_grad_all(const void* panels, int ptiles, int npairs, const int* pairs, long nblk, float* partial,
                                   float* bias_partial, int nsplit, int out_stride, int bias_stride, void* stream) {
  if (nblk <= 0 || npairs <= 0) return 0;
  if (npairs > WG_MAX_PAIRS) { avc_set_error("avc_weight_grad_all: too many pairs"); return 1; }
  WgPairs pp;
  for (int i = 0; i < npairs; ++i) {
    for (int k = 0; k < 6; ++k) pp.v[i][k] = pairs[i * 6 + k];
    if (pp.v[i][1] < 1 || pp.v[i][1] > 8 || pp.v[i][3] < 1 || pp.v[i][3] > WG_TB_MAX) {
      avc_set_error("avc_weight_grad_all: 1 <= ta <= 8, 1 <= tb <= 9");
      return 1;
    }
  }
  if (nsplit < 1) nsplit = 1;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)weight_grad_all_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_BUF_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(weight_grad_all_kernel, dim3(nsplit, npairs), dim3(512), 2 * WG_BUF_BYTES, (hipStream_t)stream, (const b8*)panels,
                     ptiles, pp, nblk, partial, bias_partial, out_stride, bias_stride);
  return avc_check_launch("avc_weight_grad_all");
}

extern "C" int avc_weight_grad(const void* panels, int ptiles, int pa, int ta, int pb, int tb, long nblk, float* partial,
                               float* bias_partial, int nsplit, int out_stride, int bias_stride, void* stream) {
  if (nblk <= 0 || ta <= 0 || tb <= 0) return 0;
  if (ta > 8 || tb > WG_TB_MAX) { avc_set_error("avc_weight_grad: ta <= 8, tb <= 9"); return 1; }
  if (nsplit < 1) nsplit = 1;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)weight_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_BUF_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(weight_grad_kernel, dim3(nsplit), dim3(512), 2 * WG_BUF_BYTES, (hipStream_t)stream, (const b8*)panels, ptiles, pa, ta, pb,
                     tb, nblk, partial, bias_partial, out_stride, bias_stride);
  return avc_check_launch("avc_weight_grad");
}
