// Weight-gradient GEMM of the NeuS MLP backward (main.py:537): dW = sum_points A^T B from the operand panels written by
// avc_render_points_fwd_train (forward-type operands, f16) and avc_render_points_bwd (gradient-type operands, bf16).
#include "avc_common.h"
#include "../../include/avc.h"

// ---------------------------------------------------------------------------------------------
// partial[split][ta][tb] = sum over the split's 32-point blocks, sum_kappa A[blk][ta][kappa] x B[blk][tb][kappa]
// One 8-wave workgroup per (K-split, pair).  Per block the (ta+tb) panel tiles are copied ONCE into LDS by global->LDS DMA
// (double buffered, the copy of block b+1 runs under the work on block b).  The tiles arrive in the producers' FRAGMENT layout
// (lane = point, 8 features per k-step); the contraction over points needs them feature-major (lane = feature, 16 points per
// lane).  The transposition runs on the matrix core: two MFMAs against a 0/1 selection fragment turn a 32 x 32 tile from
// lane = point to lane = feature (exact); the waves share the (ta+tb) tiles of a block, write the bf16 results to a second
// LDS area, and after one more barrier wave w contracts A tile w with all tb B tiles (<= 9 accumulators).
// (Round 1 transposed in the producers, twice per re-read tile; here it costs ~2 MFMAs per tile and block in a kernel whose
// matrix pipe idles behind HBM.)  Partials are written with plain stores and summed on the host side of the ABI (no atomics).
// HBM-bound by construction: 2 KiB per tile per block is read exactly once per pair it takes part in.
// ---------------------------------------------------------------------------------------------
#define WG_TB_MAX 9
#define WG_TILES_MAX 17
#define WG_BUF_BYTES (WG_TILES_MAX * 2048)

template <typename V>
__device__ __forceinline__ void make_sel(int lane, V& e0, V& e1) {
  // selection fragments: lane (n,h) of k-step-half e: 1 where feature slot (h,j) == n
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f = 8 * (j >> 2) + 4 * h + (j & 3);
    e0[j] = (typename MF<V>::S)(n == f ? 1.f : 0.f);
    e1[j] = (typename MF<V>::S)(n == 16 + f ? 1.f : 0.f);
  }
}

__device__ __forceinline__ void weight_grad_body(char* lds, const b8* __restrict__ panels, int ptiles, int pa, int ta_n, int pb,
                                                 int tb_n, int type_a, int type_b, long nblk, float* __restrict__ partial,
                                                 float* __restrict__ bias_partial, int out_elems, int bias_elems) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int split = blockIdx.x, nsplit = gridDim.x;   // (the pair index is blockIdx.y)
  const long b0 = nblk * split / nsplit, b1 = nblk * (split + 1) / nsplit;
  const int ntile = ta_n + tb_n;
  const int nchunk = ntile * 2;
  char* raw = lds;                               // 2 x WG_BUF_BYTES: fragment-layout tiles as they arrive
  char* tr = lds + 2 * WG_BUF_BYTES;             // WG_BUF_BYTES: transposed bf16 tiles of the current block
  auto issue = [&](long blk, int buf) {
    const char* base = reinterpret_cast<const char*>(panels + blk * (long)ptiles * 128);
    for (int c = wv; c < nchunk; c += 8) {
      const int tix = c >> 1;
      const int tile = tix < ta_n ? pa + tix : pb + (tix - ta_n);
      const char* g = base + ((long)tile * 128 + (c & 1) * 64 + lane) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(raw + buf * WG_BUF_BYTES + c * 1024), 16, 0, 0);
    }
  };
  h8 e0h, e1h;
  b8 e0b, e1b;
  make_sel<h8>(lane, e0h, e1h);
  make_sel<b8>(lane, e0b, e1b);
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
    __syncthreads();   // block `blk` has landed (the barrier drains the DMA); every wave is done with the previous block's `tr`
    if (blk + 1 < b1) issue(blk + 1, par ^ 1);
    // transposition: the (ta+tb) tiles of the block are dealt to the 8 waves
    for (int tix = wv; tix < ntile; tix += 8) {
      const b8* src = reinterpret_cast<const b8*>(raw + par * WG_BUF_BYTES) + (tix * 2) * 64 + lane;
      const b8 f0 = src[0], f1 = src[64];
      facc t;
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = 0.f;
      if ((tix < ta_n ? type_a : type_b) == 0) {
        t = MF<h8>::mma(__builtin_bit_cast(h8, f0), e0h, t);
        t = MF<h8>::mma(__builtin_bit_cast(h8, f1), e1h, t);
      } else {
        t = MF<b8>::mma(f0, e0b, t);
        t = MF<b8>::mma(f1, e1b, t);
      }
      b8 k0, k1;
#pragma unroll
      for (int j = 0; j < 8; ++j) { k0[j] = (__bf16)t[j]; k1[j] = (__bf16)t[8 + j]; }
      b8* dst = reinterpret_cast<b8*>(tr) + (tix * 2) * 64 + lane;
      dst[0] = k0;
      dst[64] = k1;
    }
    // the transposed tiles are visible to every wave; a raw barrier, so that the DMA of the next block stays in flight across it
    // (__syncthreads() would drain vmcnt first and serialise the copy with the contraction below)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (own) {
      const b8* L = reinterpret_cast<const b8*>(tr) + lane;
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

// every product of one backward pass in ONE launch: blockIdx.y = pair, blockIdx.x = K-split.  Workgroups are dispatched
// x-fastest, so the tail of one pair's splits overlaps the head of the next pair's instead of draining the chip 17 times.
#define WG_MAX_PAIRS 24
struct WgPairs { int v[WG_MAX_PAIRS][8]; };   // pa, ta, pb, tb, out_off, bias_off (-1 = no bias), type_a, type_b
__global__ __launch_bounds__(512) void weight_grad_all_kernel(const b8* __restrict__ panels, int ptiles, WgPairs pp, long nblk,
                                                              float* __restrict__ partial, float* __restrict__ bias_partial,
                                                              int out_elems, int bias_elems) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int* d = pp.v[blockIdx.y];
  weight_grad_body(lds, panels, ptiles, d[0], d[1], d[2], d[3], d[6], d[7], nblk, partial + d[4],
                   d[5] >= 0 ? bias_partial + d[5] : nullptr, out_elems, bias_elems);
}

extern "C" int avc_weight_grad_all(const void* panels, int ptiles, int npairs, const int* pairs, long nblk, float* partial,
                                   float* bias_partial, int nsplit, int out_stride, int bias_stride, void* stream) {
  if (nblk <= 0 || npairs <= 0) return 0;
  if (npairs > WG_MAX_PAIRS) { avc_set_error("avc_weight_grad_all: too many pairs"); return 1; }
  WgPairs pp;
  for (int i = 0; i < npairs; ++i) {
    for (int k = 0; k < 8; ++k) pp.v[i][k] = pairs[i * 8 + k];
    if (pp.v[i][1] < 1 || pp.v[i][1] > 8 || pp.v[i][3] < 1 || pp.v[i][3] > WG_TB_MAX) {
      avc_set_error("avc_weight_grad_all: 1 <= ta <= 8, 1 <= tb <= 9");
      return 1;
    }
  }
  if (nsplit < 1) nsplit = 1;
  const int lds_bytes = 3 * WG_BUF_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)weight_grad_all_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(weight_grad_all_kernel, dim3(nsplit, npairs), dim3(512), lds_bytes, (hipStream_t)stream, (const b8*)panels,
                     ptiles, pp, nblk, partial, bias_partial, out_stride, bias_stride);
  return avc_check_launch("avc_weight_grad_all");
}
