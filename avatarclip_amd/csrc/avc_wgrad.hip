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
#ifndef WG_DEPTH
#define WG_DEPTH 4   // blocks in the LDS ring (>= 3): one being contracted, one being transposed, WG_DEPTH - 2 copies in flight
#endif

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
// wait until at most `n` of this wave's vector-memory operations are outstanding (n is wave-uniform, <= 15 here)
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // any other count: conservative
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
  const int my_chunks = (nchunk - wv + 7) >> 3;   // DMA instructions this wave issues per block
  auto issue = [&](long blk, int slot) {
    const char* base = reinterpret_cast<const char*>(panels + blk * (long)ptiles * 128);
    for (int c = wv; c < nchunk; c += 8) {
      const int tix = c >> 1;
      const int tile = tix < ta_n ? pa + tix : pb + (tix - ta_n);
      const char* g = base + ((long)tile * 128 + (c & 1) * 64 + lane) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(lds + slot * WG_BUF_BYTES + c * 1024), 16, 0, 0);
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
  // transposition IN PLACE of the block in ring slot `sl`: its (ta+tb) tiles are dealt to the 8 waves; a tile is read and
  // rewritten by one wave only
  auto transpose = [&](int sl) {
    char* buf = lds + sl * WG_BUF_BYTES;
    for (int tix = wv; tix < ntile; tix += 8) {
      b8* tp = reinterpret_cast<b8*>(buf) + (tix * 2) * 64 + lane;
      const b8 f0 = tp[0], f1 = tp[64];
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
      tp[0] = k0;
      tp[64] = k1;
    }
  };
  // Ring of WG_DEPTH block buffers.  Software pipeline with ONE barrier per block: in the interval of block b the waves contract
  // block b (transposed during the previous interval) and transpose block b+1 (its copy has landed), while the copies of blocks
  // b+2 .. b+WG_DEPTH-1 are in flight.  Raw barriers + counted vmcnt throughout: __syncthreads() would drain the copies.
  for (int d = 0; d < WG_DEPTH - 1; ++d)
    if (b0 + d < b1) issue(b0 + d, d);
  if (b0 < b1) {
    long younger = b1 - 1 - b0;
    if (younger > WG_DEPTH - 2) younger = WG_DEPTH - 2;
    wait_vmcnt((int)younger * my_chunks);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    transpose(0);
  }
  int slot = 0;
  for (long blk = b0; blk < b1; ++blk) {
    // this wave's chunks of block blk+1 have landed when at most the chunks of the younger blocks in flight are outstanding
    // (LDS-DMA completes in issue order)
    long younger = b1 - 2 - blk;
    if (younger > WG_DEPTH - 3) younger = WG_DEPTH - 3;
    if (younger < 0) younger = 0;
    wait_vmcnt((int)younger * my_chunks);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my transposed tiles of block blk are written, my reads of block blk-1 are done
    __builtin_amdgcn_s_barrier();                        // -> block blk is transposed, block blk+1 has landed, the slot of blk-1 is free
    __builtin_amdgcn_sched_barrier(0);
    if (blk + WG_DEPTH - 1 < b1) issue(blk + WG_DEPTH - 1, (slot + WG_DEPTH - 1) % WG_DEPTH);
    if (own) {
      const b8* L = reinterpret_cast<const b8*>(lds + slot * WG_BUF_BYTES) + lane;
      const b8 a0 = L[(wv * 2) * 64], a1 = L[(wv * 2 + 1) * 64];
      if (bias_partial) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += (float)a0[j] + (float)a1[j];
      }
      // the B fragments are requested in batches of WG_BATCH ahead of their MFMAs (hipcc otherwise emits read -> wait -> MFMA
      // per fragment: 20 exposed LDS round trips per block); two batches keep the kernel clear of spills
#ifndef WG_BATCH
#define WG_BATCH 5
#endif
#pragma unroll
      for (int q0 = 0; q0 < WG_TB_MAX; q0 += WG_BATCH) {
        b8 bv[WG_BATCH][2];
#pragma unroll
        for (int k = 0; k < WG_BATCH; ++k) {
          if (q0 + k < WG_TB_MAX && q0 + k < tb_n) {
            bv[k][0] = L[((ta_n + q0 + k) * 2) * 64];
            bv[k][1] = L[((ta_n + q0 + k) * 2 + 1) * 64];
          }
        }
#pragma unroll
        for (int k = 0; k < WG_BATCH; ++k) asm volatile("" : "+v"(bv[k][0]), "+v"(bv[k][1]));
#pragma unroll
        for (int k = 0; k < WG_BATCH; ++k) {
          if (q0 + k < WG_TB_MAX && q0 + k < tb_n) {
            acc[q0 + k] = MF<b8>::mma(a0, bv[k][0], acc[q0 + k]);
            acc[q0 + k] = MF<b8>::mma(a1, bv[k][1], acc[q0 + k]);
          }
        }
      }
    }
    if (blk + 1 < b1) transpose((slot + 1) % WG_DEPTH);
    slot = (slot + 1) % WG_DEPTH;
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
  const int lds_bytes = WG_DEPTH * WG_BUF_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)weight_grad_all_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(weight_grad_all_kernel, dim3(nsplit, npairs), dim3(512), lds_bytes, (hipStream_t)stream, (const b8*)panels,
                     ptiles, pp, nblk, partial, bias_partial, out_stride, bias_stride);
  return avc_check_launch("avc_weight_grad_all");
}
