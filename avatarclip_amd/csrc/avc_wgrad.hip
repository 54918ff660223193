// Weight-gradient GEMM of the NeuS MLP backward (main.py:537): dW = sum_points A^T B from the operand panels written by
// avc_render_points_fwd_train (forward-type operands, f16) and avc_render_points_bwd (gradient-type operands, bf16).
#include "avc_common.h"
#include "../../include/avc.h"

// ---------------------------------------------------------------------------------------------
// partial[split][ta][tb] = sum over the split's 32-point blocks, sum_kappa A[blk][ta][kappa] x B[blk][tb][kappa]
// One 8-wave workgroup per (K-split, pair).  Per block the (ta+tb) panel tiles are copied ONCE into an LDS ring by
// global->LDS DMA (WG_DEPTH - 1 blocks in flight under the work on the current one).  The tiles arrive in the producers'
// FRAGMENT layout (lane = point, 8 features per k-step); the contraction over points needs them feature-major (lane =
// feature, 8 points per lane).  gfx950's LDS transpose read does that on the way into the registers: ds_read_b64_tr_b16
// hands lane i of a 16-lane group element (i & 3) of the 8 bytes addressed by lane 4 j + (i >> 2) of the group, for j = 0..3,
// i.e. the column i of a [4 points][16 features] block.  The DMA therefore deals the 16-byte chunks of a tile (chunk (s,h,p)
// = features 16 s + 8 (j >> 2) + 4 h + (j & 3) of point p) to the LDS position (p >> 2) * 16 + (2 s + h) * 4 + (p & 3): the 32
// lanes of a half-wave then read 256 contiguous bytes (no bank conflict) and every DMA instruction still reads whole
// 256-byte runs of the panel.  (Round 1 transposed in the producers; the first round-2 kernel transposed on the matrix core
// in a separate pass through LDS: 68 KiB of extra LDS traffic and one more pipeline stage per block.)
// The 8 waves split the ta x tb output tiles 4 x 2 (2 A tiles x 4 B tiles per wave: 6 tile reads for 8 products) when both
// sides are wide, 8 x 1 or 1 x 8 otherwise.  One operand of every pair is f16 (forward-type), the other bf16: the f16 side is
// converted after the read.  Partials are written with plain stores and summed on the host side of the ABI (no atomics).
// HBM-bound by construction: 2 KiB per tile per block is read exactly once per pair it takes part in.
// ---------------------------------------------------------------------------------------------
#include "avc_wgrad_body.h"

// NI x NK output tiles per wave: A tiles wa + WA i, B tiles wb + WB k
template <int NI, int NK>
__device__ __forceinline__ void weight_grad_body(char* lds, const WgRegions& rg, int pa, int ta_n, int pb,
                                                 int tb_n, int type_a, int type_b, long nblk, float* __restrict__ partial,
                                                 float* __restrict__ bias_partial, int out_elems, int bias_elems, int WA) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int WB = 8 / WA;
  const int wa = wv % WA, wb = wv / WA;
  const int split = blockIdx.x, nsplit = gridDim.x;   // (the pair index is blockIdx.y)
  const long b0 = nblk * split / nsplit, b1 = nblk * (split + 1) / nsplit;
  const int ntile = ta_n + tb_n;
  const int nchunk = ntile * 2;
  const int my_chunks = (nchunk - wv + 7) >> 3;   // DMA instructions this wave issues per block
  constexpr int slot_bytes = WG_BUF_BYTES, depth = WG_DEPTH;
  // source chunk of this lane in DMA instruction c of a tile: LDS position 64 c + lane = (p >> 2) * 16 + (2 s + h) * 4 + (p & 3)
  const int src_lo = ((lane >> 2) & 3) * 32 + 4 * (lane >> 4) + (lane & 3);   // (2 s + h) * 32 + p with p = 4 (lane >> 4) + (lane & 3)
  // operand A lives in region type_a, operand B in region type_b (0 = F region: f16 forward-type tiles, 1 = G region: bf16)
  const char* const base_a = (type_a ? rg.base[1] : rg.base[0]) + (long)pa * 2048;
  const char* const base_b = (type_b ? rg.base[1] : rg.base[0]) + (long)pb * 2048;
  const long stride_a = type_a ? rg.stride[1] : rg.stride[0], stride_b = type_b ? rg.stride[1] : rg.stride[0];
  auto issue = [&](long blk, int slot) {
    for (int c = wv; c < nchunk; c += 8) {
      const int tix = c >> 1;
      const char* tb_ = tix < ta_n ? base_a + blk * stride_a + (long)tix * 2048 : base_b + blk * stride_b + (long)(tix - ta_n) * 2048;
      const char* g = tb_ + (src_lo + (c & 1) * 16) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(lds + slot * slot_bytes + c * 1024), 16, 0, WG_DMA_AUX);
    }
  };
  // this lane's byte offset inside a tile for the transpose reads: group g = lane >> 4 = 2 hh + s, i = lane & 15 supplies
  // the 8 bytes (half c >> 1 of chunk (s, h' = c & 1, p = p0 + r)), r = i >> 2, c = i & 3;  p0 = 16 t + 8 hh + 4 u
  const int lane_off = (((lane >> 5) * 2) * 16 + (((lane >> 4) & 1) * 2 + (lane & 1)) * 4 + ((lane & 15) >> 2)) * 16 + 8 * ((lane & 3) >> 1);
  facc acc[NI * NK];
#pragma unroll
  for (int q = 0; q < NI * NK; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float bsum[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) bsum[i] = 0.f;
  // Ring of `depth` block buffers, ONE barrier per block: it publishes block blk (every wave has waited for its own chunks) and
  // frees the slot of block blk - 1, which the next copy then overwrites.  Raw barriers + counted vmcnt throughout:
  // __syncthreads() would drain the copies in flight.
  for (int d = 0; d < depth - 1; ++d)
    if (b0 + d < b1) issue(b0 + d, d);
  int slot = 0;
  for (long blk = b0; blk < b1; ++blk) {
    // LDS-DMA completes in issue order: block blk has landed when at most the chunks of the younger blocks are outstanding
    long younger = b1 - 1 - blk;
    if (younger > depth - 2) younger = depth - 2;
    wait_vmcnt((int)younger * my_chunks);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of block blk - 1 are done
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (blk + depth - 1 < b1) { int ns = slot + depth - 1; if (ns >= depth) ns -= depth; issue(blk + depth - 1, ns); }
    lds_char* buf = (lds_char*)(lds + slot * slot_bytes) + lane_off;
    b8 a[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ta = wa + WA * i;
      if (ta < ta_n) {
        a[i][0] = tr_frag(buf + ta * 2048, 0);
        a[i][1] = tr_frag(buf + ta * 2048, 1);
      }
    }
    b8 bv[NK][2];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int tb = wb + WB * k;
      if (tb < tb_n) {
        bv[k][0] = tr_frag(buf + (ta_n + tb) * 2048, 0);
        bv[k][1] = tr_frag(buf + (ta_n + tb) * 2048, 1);
      }
    }
    if (type_a == 0) {
#pragma unroll
      for (int i = 0; i < NI; ++i) { a[i][0] = f16_to_bf16(a[i][0]); a[i][1] = f16_to_bf16(a[i][1]); }
    }
    if (type_b == 0) {
#pragma unroll
      for (int k = 0; k < NK; ++k) { bv[k][0] = f16_to_bf16(bv[k][0]); bv[k][1] = f16_to_bf16(bv[k][1]); }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (wa + WA * i < ta_n) {
        if (bias_partial && wb == i) {   // the bias row sums of A tile i are the job of the wave column wb = i (balanced across waves)
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[i] += (float)a[i][0][j] + (float)a[i][1][j];
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          if (wb + WB * k < tb_n) {
#ifdef AVC_ABL_WG_NOMFMA   // timing ablation (garbage results): the operands are streamed, transposed and converted but never contracted
            asm volatile("" :: "v"(a[i][0]), "v"(bv[k][0]), "v"(a[i][1]), "v"(bv[k][1]));
#else
            acc[i * NK + k] = MF<b8>::mma(a[i][0], bv[k][0], acc[i * NK + k]);
            acc[i * NK + k] = MF<b8>::mma(a[i][1], bv[k][1], acc[i * NK + k]);
#endif
          }
        }
      }
    }
    if (++slot == depth) slot = 0;
  }
  float* dst = partial + (long)split * out_elems;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ta = wa + WA * i;
    if (ta >= ta_n) continue;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int tb = wb + WB * k;
      if (tb >= tb_n) continue;
      f4* d4 = reinterpret_cast<f4*>(dst + ((long)(ta * tb_n + tb) * 64 + lane) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f4 v;
        v[0] = acc[i * NK + k][4 * q]; v[1] = acc[i * NK + k][4 * q + 1]; v[2] = acc[i * NK + k][4 * q + 2]; v[3] = acc[i * NK + k][4 * q + 3];
        d4[q] = v;
      }
    }
    if (bias_partial && wb == i) {
      const float s = xhalf_sum(bsum[i]);
      if (lane < 32) bias_partial[(long)split * bias_elems + ta * 32 + lane] = s;
    }
  }
}

// The merged last-layer product: A = [ybar[1:] (8 tiles) | d_sdf row (1 tile)], B = [hs (7) | pe (2)], 9 x 9 output tiles, so that
// [hs | pe] is read from HBM once for the 256 feature rows AND the sdf row.  Waves 4 x 2 over the 8 x 9 part as in the generic
// body (A tiles wa, wa + 4; B tiles wb, wb + 2, ..: 10 products for wb = 0, 8 for wb = 1); the nine products of the ninth A tile go to
// the waves that have the matching B tile in registers anyway: wave (wa, 1) takes B tile 2 wa + 1 and -- one more LDS read -- 2 wa,
// wave (0, 0) takes B tile 8: 12 accumulators at most.  B fragments are streamed one tile ahead of their MFMAs (all five at once,
// as in the generic body, do not fit beside 12 accumulators).
__device__ __forceinline__ void weight_grad_body_9x9(char* lds, const WgRegions& rg, int pa, int pb, int type_a,
                                                     int type_b, long nblk, float* __restrict__ partial,
                                                     float* __restrict__ bias_partial, int out_elems, int bias_elems) {
  constexpr int TA = 9, TB = 9, NT = TA + TB;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wa = wv & 3, wb = wv >> 2;
  const int split = blockIdx.x, nsplit = gridDim.x;
  const long b0 = nblk * split / nsplit, b1 = nblk * (split + 1) / nsplit;
  constexpr int nchunk = NT * 2;
  const int my_chunks = (nchunk - wv + 7) >> 3;
  const int src_lo = ((lane >> 2) & 3) * 32 + 4 * (lane >> 4) + (lane & 3);
  const char* const base_a = (type_a ? rg.base[1] : rg.base[0]) + (long)pa * 2048;
  const char* const base_b = (type_b ? rg.base[1] : rg.base[0]) + (long)pb * 2048;
  const long stride_a = type_a ? rg.stride[1] : rg.stride[0], stride_b = type_b ? rg.stride[1] : rg.stride[0];
  auto issue = [&](long blk, int slot) {
    for (int c = wv; c < nchunk; c += 8) {
      const int tix = c >> 1;
      const char* tb_ = tix < TA ? base_a + blk * stride_a + (long)tix * 2048 : base_b + blk * stride_b + (long)(tix - TA) * 2048;
      const char* g = tb_ + (src_lo + (c & 1) * 16) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(lds + slot * WG_BUF_BYTES + c * 1024), 16, 0, WG_DMA_AUX);
    }
  };
  const int lane_off = (((lane >> 5) * 2) * 16 + (((lane >> 4) & 1) * 2 + (lane & 1)) * 4 + ((lane & 15) >> 2)) * 16 + 8 * ((lane & 3) >> 1);
  facc acc[10], ex[2];
#pragma unroll
  for (int q = 0; q < 10; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) ex[q][r] = 0.f;
  float bsum[2] = {0.f, 0.f}, xsum = 0.f;
  const bool has_x = wb == 1 || wa == 0;          // waves that hold products of the ninth A tile
  const int xk = wb == 1 ? wa : 4;                 // the regular B slot k whose tile also meets the ninth A tile
  for (int d = 0; d < WG_DEPTH - 1; ++d)
    if (b0 + d < b1) issue(b0 + d, d);
  int slot = 0;
  for (long blk = b0; blk < b1; ++blk) {
    long younger = b1 - 1 - blk;
    if (younger > WG_DEPTH - 2) younger = WG_DEPTH - 2;
    wait_vmcnt((int)younger * my_chunks);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (blk + WG_DEPTH - 1 < b1) issue(blk + WG_DEPTH - 1, (slot + WG_DEPTH - 1) % WG_DEPTH);
    lds_char* buf = (lds_char*)(lds + slot * WG_BUF_BYTES) + lane_off;
    auto load_b = [&](int tb, b8& f0, b8& f1) {
      f0 = tr_frag(buf + (TA + tb) * 2048, 0);
      f1 = tr_frag(buf + (TA + tb) * 2048, 1);
    };
    b8 a[2][2], sa[2], c0, c1, n0, n1;
    load_b(wb, c0, c1);
    n0 = c0; n1 = c1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i][0] = tr_frag(buf + (wa + 4 * i) * 2048, 0);
      a[i][1] = tr_frag(buf + (wa + 4 * i) * 2048, 1);
    }
    if (has_x) { sa[0] = tr_frag(buf + 8 * 2048, 0); sa[1] = tr_frag(buf + 8 * 2048, 1); }
    if (type_a == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { a[i][0] = f16_to_bf16(a[i][0]); a[i][1] = f16_to_bf16(a[i][1]); }
      if (has_x) { sa[0] = f16_to_bf16(sa[0]); sa[1] = f16_to_bf16(sa[1]); }
    }
    if (bias_partial) {   // wave (wa, wb): row sums of A tile wa + 4 wb
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (wb == i) {
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[i] += (float)a[i][0][j] + (float)a[i][1][j];
        }
    }
    if (bias_partial && wv == 4) {   // wave (0, 1): row sums of the ninth A tile
#pragma unroll
      for (int j = 0; j < 8; ++j) xsum += (float)sa[0][j] + (float)sa[1][j];
    }
    // regular slots k = 0..4 (B tile wb + 2 k), then -- waves with wb = 1 -- the extra B tile 2 wa; each tile is requested before
    // the MFMAs of the previous one
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int tb_next = (k + 1 < 5) ? wb + 2 * (k + 1) : 2 * wa;
      const bool next_ok = (k + 1 < 5) ? (tb_next < TB) : (k + 1 == 5 && wb == 1);
      if (next_ok) load_b(tb_next, n0, n1);
      asm volatile("" : "+v"(n0), "+v"(n1));
      const bool cur_ok = (k < 5) ? (wb + 2 * k < TB) : (wb == 1);
      if (cur_ok) {
        if (type_b == 0) { c0 = f16_to_bf16(c0); c1 = f16_to_bf16(c1); }
        if (k < 5) {
          acc[k] = MF<b8>::mma(a[0][0], c0, acc[k]);
          acc[k] = MF<b8>::mma(a[0][1], c1, acc[k]);
          acc[5 + k] = MF<b8>::mma(a[1][0], c0, acc[5 + k]);
          acc[5 + k] = MF<b8>::mma(a[1][1], c1, acc[5 + k]);
          if (has_x && k == xk) {
            ex[0] = MF<b8>::mma(sa[0], c0, ex[0]);
            ex[0] = MF<b8>::mma(sa[1], c1, ex[0]);
          }
        } else {
          ex[1] = MF<b8>::mma(sa[0], c0, ex[1]);
          ex[1] = MF<b8>::mma(sa[1], c1, ex[1]);
        }
      }
      c0 = n0; c1 = n1;
    }
    slot = (slot + 1) % WG_DEPTH;
  }
  float* dst = partial + (long)split * out_elems;
  auto store_tile = [&](int ta, int tb, const facc& v16) {
    f4* d4 = reinterpret_cast<f4*>(dst + ((long)(ta * TB + tb) * 64 + lane) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f4 v;
      v[0] = v16[4 * q]; v[1] = v16[4 * q + 1]; v[2] = v16[4 * q + 2]; v[3] = v16[4 * q + 3];
      d4[q] = v;
    }
  };
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (wb + 2 * k < TB) {
      store_tile(wa, wb + 2 * k, acc[k]);
      store_tile(wa + 4, wb + 2 * k, acc[5 + k]);
    }
  }
  if (has_x) store_tile(8, wb + 2 * xk, ex[0]);
  if (wb == 1) store_tile(8, 2 * wa, ex[1]);
  if (bias_partial) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (wb == i) {
        const float sx = xhalf_sum(bsum[i]);
        if (lane < 32) bias_partial[(long)split * bias_elems + (wa + 4 * i) * 32 + lane] = sx;
      }
    }
    if (wv == 4) {
      const float sx = xhalf_sum(xsum);
      if (lane < 32) bias_partial[(long)split * bias_elems + 8 * 32 + lane] = sx;
    }
  }
}

// every product of one backward pass in ONE launch: blockIdx.y = pair, blockIdx.x = K-split.  Workgroups are dispatched
// x-fastest, so the tail of one pair's splits overlaps the head of the next pair's instead of draining the chip 17 times.
#define WG_MAX_PAIRS 24
struct WgPairs { int v[WG_MAX_PAIRS][8]; };   // pa, ta, pb, tb, out_off, bias_off (-1 = no bias), type_a, type_b
__global__ __launch_bounds__(512) void weight_grad_all_kernel(WgRegions rg, WgPairs pp, long nblk,
                                                              float* __restrict__ partial, float* __restrict__ bias_partial,
                                                              int out_elems, int bias_elems) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int* d = pp.v[blockIdx.y];
  float* bp = d[5] >= 0 ? bias_partial + d[5] : nullptr;
  // wide x wide: 2 x 4 (or 4 x 2) tiles per wave, the f16 operand on the side with fewer tiles per wave (it is converted to bf16
  // after the read); narrow pairs: 8 x 1 or 1 x 8 waves
#define WG_ARGS lds, rg, d[0], d[1], d[2], d[3], d[6], d[7], nblk, partial + d[4], bp, out_elems, bias_elems
  if (d[1] > 1 && d[3] > 2 && d[3] <= 8) {
    if (d[7] == 0) weight_grad_body<4, 2>(WG_ARGS, 2);
    else weight_grad_body<2, 4>(WG_ARGS, 4);
  } else if (d[1] == 9 && d[3] == 9) {
    weight_grad_body_9x9(lds, rg, d[0], d[2], d[6], d[7], nblk, partial + d[4], bp, out_elems, bias_elems);
  } else if (d[1] > 1 && d[3] > 8) {
    weight_grad_body<2, 5>(WG_ARGS, 4);
  } else if (d[1] > 1) {
    weight_grad_body<1, 2>(WG_ARGS, 8);      // tb <= 2: one A tile x both B tiles per wave
  } else {
    weight_grad_body<1, 2>(WG_ARGS, 1);      // a single A tile: 1 x 8 waves, one (or two) B tiles each
  }
#undef WG_ARGS
}

extern "C" int avc_weight_grad_all(const void* fpanels, int ftiles, const void* gpanels, int gtiles, int npairs, const int* pairs,
                                   long nblk, float* partial, float* bias_partial, int nsplit, int out_stride, int bias_stride,
                                   void* stream) {
  if (nblk <= 0 || npairs <= 0) return 0;
  if (!fpanels || !gpanels) { avc_set_error("avc_weight_grad_all: fpanels / gpanels == NULL"); return 1; }
  if (npairs > WG_MAX_PAIRS) { avc_set_error("avc_weight_grad_all: too many pairs"); return 1; }
  WgPairs pp;
  for (int i = 0; i < npairs; ++i) {
    for (int k = 0; k < 8; ++k) pp.v[i][k] = pairs[i * 8 + k];
    const bool merged = pp.v[i][1] == 9 && pp.v[i][3] == 9;
    if (!merged && (pp.v[i][1] < 1 || pp.v[i][1] > 8 || pp.v[i][3] < 1 || pp.v[i][3] > WG_TB_MAX)) {
      avc_set_error("avc_weight_grad_all: 1 <= ta <= 8, 1 <= tb <= 9 (or the 9 x 9 product)");
      return 1;
    }
  }
  if (nsplit < 1) nsplit = 1;
  const int lds_bytes = WG_DEPTH * WG_BUF_BYTES;
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen)) {
    (void)hipFuncSetAttribute((const void*)weight_grad_all_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  }
  WgRegions rg;
  rg.base[0] = (const char*)fpanels; rg.stride[0] = (long)ftiles * 2048;
  rg.base[1] = (const char*)gpanels; rg.stride[1] = (long)gtiles * 2048;
  hipLaunchKernelGGL(weight_grad_all_kernel, dim3(nsplit, npairs), dim3(512), lds_bytes, (hipStream_t)stream, rg, pp, nblk, partial,
                     bias_partial, out_stride, bias_stride);
  return avc_check_launch("avc_weight_grad_all");
}

// ---- split sums and the way back to the dense parameter vector (what the caller of avc_weight_grad_all does next) ----
// acc[j] (+)= sum_s partial[s][j] for the gout_size weight-product floats followed by the gbias_size bias floats, splits added in
// order (deterministic), 4 floats per thread: ns x 1.7 MB of streaming reads, one launch instead of two torch reductions + two adds
// per slab.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bpartial, int ns,
                                                           long stride, long bstride, int gout4, int gbias4, float* __restrict__ acc,
                                                           int accumulate) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= gout4 + gbias4) return;
  const f4* src = j < gout4 ? reinterpret_cast<const f4*>(partial) + j : reinterpret_cast<const f4*>(bpartial) + (j - gout4);
  const long st4 = (j < gout4 ? stride : bstride) >> 2;
  f4 a = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= ns; s += 8) {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (long)(s + u) * st4);
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  for (; s < ns; ++s) a += __builtin_nontemporal_load(src + (long)s * st4);
  f4* out = reinterpret_cast<f4*>(acc) + j;
  if (accumulate) a += *out;
  *out = a;
}
// grad[t] = sum_{k in [off[t], off[t+1])} acc[src[k]] * scale[k]: the tile layout of the products -> the dense parameter vector (a
// parameter that several tile entries map to -- the hi + lo slots of the merged last-layer product -- adds them in list order)
__global__ __launch_bounds__(256) void wgrad_unpack_kernel(const float* __restrict__ acc, const int* __restrict__ off, const int* __restrict__ src,
                                                           const float* __restrict__ scale, int nparam, float* __restrict__ grad) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nparam) return;
  float g = 0.f;
  for (int k = off[t]; k < off[t + 1]; ++k) g += acc[src[k]] * scale[k];
  grad[t] = g;
}
extern "C" int avc_weight_grad_reduce(const float* partial, const float* bias_partial, int nsplit, int out_stride, int bias_stride,
                                      int gout_size, int gbias_size, float* acc, int accumulate, void* stream) {
  if ((gout_size & 3) || (gbias_size & 3) || (out_stride & 3) || (bias_stride & 3)) {
    avc_set_error("avc_weight_grad_reduce: sizes and strides must be multiples of 4 floats");
    return 1;
  }
  if (nsplit < 1 || !partial || !acc || (gbias_size && !bias_partial)) { avc_set_error("avc_weight_grad_reduce: bad arguments"); return 1; }
  const int n4 = (gout_size + gbias_size) / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, bias_partial, nsplit,
                     (long)out_stride, (long)bias_stride, gout_size / 4, gbias_size / 4, acc, accumulate);
  return avc_check_launch("avc_weight_grad_reduce");
}
extern "C" int avc_weight_grad_unpack(const float* acc, const int* off, const int* src, const float* scale, int nparam, float* grad,
                                      void* stream) {
  if (nparam <= 0) return 0;
  if (!acc || !off || !src || !scale || !grad) { avc_set_error("avc_weight_grad_unpack: NULL buffer"); return 1; }
  hipLaunchKernelGGL(wgrad_unpack_kernel, dim3((nparam + 255) / 256), dim3(256), 0, (hipStream_t)stream, acc, off, src, scale, nparam, grad);
  return avc_check_launch("avc_weight_grad_unpack");
}
