// CLIP ViT-B/32 image-encoder kernels (perceptor.encode_image, main.py:512; OpenAI clip/model.py VisionTransformer).
//   avc_vit_linear   Y[M,N] = act(X[M,K] W[N,K]^T + b) (+ residual)   -- bf16 MFMA 32x32x16, fp32 accumulate.
//                    M = 50..100 tokens: the GEMM is weight-streaming / latency bound, so W is pre-packed in B-operand
//                    fragment order and X is packed into A-operand fragments by a pre-pass (one coalesced 16-B load per
//                    lane per k-step for both), the K range is dealt to the 8 wavefronts of a workgroup and reduced
//                    through LDS, one workgroup per 32 output columns.
//                    The same kernel computes dX = dY W with the pre-packed W^T (weights are frozen: no dW).
//   avc_vit_attention_fwd / _bwd   12-head attention over 50 tokens, one workgroup per (image, head), fp32 in LDS.
#include "avc_common.h"
#include "../../include/avc.h"

#define VIT_MAX_MT 4   // up to 128 rows (tokens): the per-iteration latency path (one row tile per workgroup); more rows: groups of 4
#define VIT_WAVES 8    // wavefronts per workgroup = K-split factor

// X[M,K] fp32 -> bf16 A-operand fragments [m-tile][k-step][lane][8] (lane (i,h) holds X[32m+i][16s+8h+j]): the GEMM then
// reads its activations with the same coalesced 16-B-per-lane loads as its weights.
// With `pre` (the pre-activation of a QuickGELU layer, same shape as X) the rows are multiplied by gelu'(pre) on the way: the
// backward of the activation is folded into the packing pass of the GEMM that follows it (dX = (dY * gelu'(pre)) W).
__global__ __launch_bounds__(64) void vit_pack_x_kernel(const float* __restrict__ X, const float* __restrict__ pre,
                                                        b8* __restrict__ xs, int M, int K) {
  const int s = blockIdx.x, m = blockIdx.y, KS = K >> 4;
  const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
  const int row = 32 * m + n;
  b8 xf;
  if (row < M) {
    const f4* xp = reinterpret_cast<const f4*>(X + (long)row * K + 16 * s + 8 * h);
    f4 a = xp[0], b = xp[1];
    if (pre) {
      const f4* pp = reinterpret_cast<const f4*>(pre + (long)row * K + 16 * s + 8 * h);
      const f4 pa = pp[0], pb = pp[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sa = sigmoidf_(1.702f * pa[j]), sb = sigmoidf_(1.702f * pb[j]);
        a[j] *= sa + 1.702f * pa[j] * sa * (1.f - sa);
        b[j] *= sb + 1.702f * pb[j] * sb * (1.f - sb);
      }
    }
    xf[0] = (__bf16)a[0]; xf[1] = (__bf16)a[1]; xf[2] = (__bf16)a[2]; xf[3] = (__bf16)a[3];
    xf[4] = (__bf16)b[0]; xf[5] = (__bf16)b[1]; xf[6] = (__bf16)b[2]; xf[7] = (__bf16)b[3];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) xf[j] = (__bf16)0.f;
  }
  xs[((long)m * KS + s) * 64 + lane] = xf;
}

// One workgroup per 32 output columns; the K range is dealt round-robin to 8 wavefronts (k-step s goes to wave s % 8), each
// wave keeps 6 k-steps of loads in flight (the GEMM is a latency chain otherwise: M <= 128 rows give the matrix core nothing
// to hide a round trip behind).  Every wave parks its partial tiles in LDS ([wave][m-tile][row][lane]: conflict-free both
// ways), ONE barrier, then the 16 MT accumulator rows are dealt to the 8 waves: each sums the 8 partials of its rows in a fixed
// order (deterministic) and runs the epilogue (bias, QuickGELU, residual, store) for them -- a tree reduction with the whole
// epilogue on wave 0 cost three barrier pairs and 64 serial load / store pairs on one wave.
// NB = k-steps a wave loads as ONE batch before their MFMAs: the per-iteration calls leave most of the chip idle (96-384 workgroups),
// so the only way to shorten a K = 3072 linear (24 k-steps per wave) is to have all of its loads in flight at once instead of four
// rounds of six -- 14.7 -> see profiles/r04_ab_kernels.txt
template <int MT, int NB = 6>
__global__ __launch_bounds__(64 * VIT_WAVES) void vit_linear_kernel(const b8* __restrict__ Xs, const b8* __restrict__ Wp,
                                                                    const float* __restrict__ bias, const float* __restrict__ res,
                                                                    float* __restrict__ Y, float* __restrict__ Ypre, int M, int N,
                                                                    int K, int act, __bf16* __restrict__ Ys = nullptr,
                                                                    const float* __restrict__ gpre = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [VIT_WAVES][MT][16][64]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int t = blockIdx.x;           // output column tile
  const int KS = K >> 4;              // k-steps of 16
  // blockIdx.y = group of MT 32-row tiles of the rows.  Per-iteration calls (M <= 128 rows): MT = 1, one workgroup per (column tile,
  // row tile) -- it then reads a quarter of the packed activations (the K = 3072 linears re-read 768 KB of them per workgroup from
  // L2 otherwise: 35 us) and the four workgroups that share a weight tile sit on one XCD (block id = x + gridDim.x * y, gridDim.x a
  // multiple of 8), i.e. share its L2.  Batched scoring (hundreds of images): MT = 4, a weight fragment feeds four MFMAs.
  const int mb = blockIdx.y;
  Xs += (long)mb * MT * KS * 64;
  const int row0 = 32 * MT * mb;
  facc acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const b8* wp = Wp + ((long)t * KS) * 64 + lane;
  const b8* xp = Xs + lane;
  for (int s0 = wv; s0 < KS; s0 += VIT_WAVES * NB) {
    b8 w[NB], x[MT][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int s = s0 + j * VIT_WAVES;
      if (s < KS) {
        w[j] = wp[(long)s * 64];
#pragma unroll
        for (int m = 0; m < MT; ++m) x[m][j] = xp[((long)m * KS + s) * 64];
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (s0 + j * VIT_WAVES < KS) {
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = MF<b8>::mma(x[m][j], w[j], acc[m]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wv * MT + m) * 16 + r) * 64 + lane] = acc[m][r];
  __syncthreads();
  const int col = 32 * t + n;
  const float bv = bias ? bias[col] : 0.f;
  constexpr int PER_WAVE = MT * 16 / VIT_WAVES;   // accumulator rows per wave: 2 MT
#pragma unroll
  for (int q = 0; q < PER_WAVE; ++q) {
    const int item = wv * PER_WAVE + q, m = item >> 4, r = item & 15;
    const int row = row0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < VIT_WAVES; ++w) v += red[((w * MT + m) * 16 + r) * 64 + lane];
    if (row < M) {
      v += bv;
      const long o = (long)row * N + col;
      if (act == 1) {
        if (Ypre) Ypre[o] = v;
        v = v * sigmoidf_(1.702f * v);
      } else if (act == 2) {          // backward of a QuickGELU layer: this product is the gradient of its OUTPUT
        const float p = gpre[o], sg = sigmoidf_(1.702f * p);
        v *= sg + 1.702f * p * sg * (1.f - sg);
      }
      if (res) v += res[o];
      if (Y) Y[o] = v;
    } else {
      v = 0.f;
    }
    // the per-iteration training pipeline (avc_vit_linear_small): the result leaves (also) as the packed bf16 operand of the linear
    // behind it -- element (row, col) is slot col & 7 of lane (row & 31, (col >> 3) & 1) of k-step col >> 4 of row tile row >> 5; the
    // rows of the last tile past M are written as zeros
    if (Ys) Ys[((((long)(row >> 5)) * (N >> 4) + (col >> 4)) * 64 + (row & 31) + 32 * ((col >> 3) & 1)) * 8 + (col & 7)] = (__bf16)v;
  }
}

// Batched calls (hundreds of images: ShapeGen codebook search, pose retrieval -- row f-4): a plain tiled GEMM.  One 4-wavefront
// workgroup per 128 x 128 output block, every wavefront a 64 x 64 quarter of it (2 x 2 MFMA tiles: each fragment it loads feeds two
// MFMAs, 1 KiB of operands per MFMA; the two wavefronts that share a row / column pair meet in the CU's L1), the whole K range per
// wavefront (no split-K, no LDS), fragments prefetched VIT_GEMM_AHEAD k-steps ahead.  Both operands arrive pre-packed in fragment
// order, so every load is one coalesced 16 B per lane.
#ifndef VIT_GEMM_AHEAD
#define VIT_GEMM_AHEAD 3
#endif
__global__ __launch_bounds__(256) void vit_gemm_kernel(const b8* __restrict__ Xs, const b8* __restrict__ Wp, const float* __restrict__ bias,
                                                       const float* __restrict__ res, float* __restrict__ Y, float* __restrict__ Ypre,
                                                       int M, int N, int K, int act) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int KS = K >> 4;
  const int mt0 = 4 * blockIdx.y + 2 * (wv & 1), nt0 = 4 * blockIdx.x + 2 * (wv >> 1);
  const b8* xa = Xs + (long)mt0 * KS * 64 + lane;
  const b8* wb = Wp + (long)nt0 * KS * 64 + lane;
  const long xstep = (long)KS * 64;     // next row / column tile
  facc acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  b8 fx[VIT_GEMM_AHEAD][2], fw[VIT_GEMM_AHEAD][2];
#pragma unroll
  for (int d = 0; d < VIT_GEMM_AHEAD; ++d) {
    if (d < KS) {
      fx[d][0] = xa[(long)d * 64]; fx[d][1] = xa[xstep + (long)d * 64];
      fw[d][0] = wb[(long)d * 64]; fw[d][1] = wb[xstep + (long)d * 64];
    }
  }
  for (int s0 = 0; s0 < KS; s0 += VIT_GEMM_AHEAD) {
#pragma unroll
    for (int d = 0; d < VIT_GEMM_AHEAD; ++d) {
      const int s = s0 + d;
      if (s < KS) {
        const b8 x0 = fx[d][0], x1 = fx[d][1], w0 = fw[d][0], w1 = fw[d][1];
        const int sn = s + VIT_GEMM_AHEAD;
        if (sn < KS) {
          fx[d][0] = xa[(long)sn * 64]; fx[d][1] = xa[xstep + (long)sn * 64];
          fw[d][0] = wb[(long)sn * 64]; fw[d][1] = wb[xstep + (long)sn * 64];
        }
        acc[0][0] = MF<b8>::mma(x0, w0, acc[0][0]);
        acc[0][1] = MF<b8>::mma(x0, w1, acc[0][1]);
        acc[1][0] = MF<b8>::mma(x1, w0, acc[1][0]);
        acc[1][1] = MF<b8>::mma(x1, w1, acc[1][1]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = 32 * (nt0 + j) + n;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * (mt0 + i) + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) {
          float v = acc[i][j][r] + bv;
          const long o = (long)row * N + col;
          if (act == 1) {
            if (Ypre) Ypre[o] = v;
            v = v * sigmoidf_(1.702f * v);
          }
          if (res) v += res[o];
          Y[o] = v;
        }
      }
    }
  }
}

// batched calls: the LDS-staged GEMM of avc_vit_gemm.hip (128 x 128 blocks)
bool avc_vit_gemm_lds(const void* xs, const void* wp, const float* bias, const float* res, float* y, float* y_pre, void* ys, int M, int N,
                      int K, int act, int mt_packed, void* stream);
#ifndef VIT_GEMM_LDS
#define VIT_GEMM_LDS 1   // 0: the direct-from-L1 128 x 128 kernel above (the A/B of profiles/r03_score_bench.txt)
#endif

extern "C" long avc_vit_workspace_bytes(int M, int K) {
  const long mt = ((M + 127) / 128) * 4;   // whole groups of 4 row tiles (the rows past M are packed as zeros)
  return mt * (K / 16) * 1024L;
}

// one row tile per workgroup (M <= 128): the batch of loads a wave keeps in flight covers its whole share of K up to K = 3072
static void vit_launch_small(dim3 grid, int lds, hipStream_t s, const b8* xs, const b8* wp, const float* bias, const float* res, float* y,
                             float* y_pre, int M, int N, int K, int act, __bf16* ys, const float* gpre) {
  const int per_wave = ((K >> 4) + VIT_WAVES - 1) / VIT_WAVES;
  const dim3 block(64 * VIT_WAVES);
  if (per_wave <= 6) hipLaunchKernelGGL((vit_linear_kernel<1, 6>), grid, block, lds, s, xs, wp, bias, res, y, y_pre, M, N, K, act, ys, gpre);
  else if (per_wave <= 12) hipLaunchKernelGGL((vit_linear_kernel<1, 12>), grid, block, lds, s, xs, wp, bias, res, y, y_pre, M, N, K, act, ys, gpre);
  else if (per_wave <= 18) hipLaunchKernelGGL((vit_linear_kernel<1, 18>), grid, block, lds, s, xs, wp, bias, res, y, y_pre, M, N, K, act, ys, gpre);
  else hipLaunchKernelGGL((vit_linear_kernel<1, 24>), grid, block, lds, s, xs, wp, bias, res, y, y_pre, M, N, K, act, ys, gpre);
}

static int vit_linear_impl(const float* x, const float* x_gelu_pre, const void* w_packed, const float* bias, const float* residual,
                           float* y, float* y_pre, int M, int N, int K, int act, void* workspace, void* stream) {
  if (M <= 0) return 0;
  if ((N & 31) || (K & 15)) {
    avc_set_error("avc_vit_linear: need N % 32 == 0, K % 16 == 0");
    return 1;
  }
  if (!workspace) { avc_set_error("avc_vit_linear: workspace == NULL (avc_vit_workspace_bytes)"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  const int mt = (M + 31) / 32;
  const bool batched = mt > VIT_MAX_MT;                 // more than 128 rows: groups of 4 row tiles per workgroup
  const int mt_packed = batched ? ((mt + 3) / 4) * 4 : mt;
  b8* xs = (b8*)workspace;
  hipLaunchKernelGGL(vit_pack_x_kernel, dim3(K / 16, mt_packed), dim3(64), 0, s, x, x_gelu_pre, xs, M, K);
  const dim3 grid(N / 32, batched ? mt_packed / 4 : mt), block(64 * VIT_WAVES);
  const b8* wp = (const b8*)w_packed;
  const int lds = VIT_WAVES * (batched ? 4 : 1) * 4096;
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen)) {
    (void)hipFuncSetAttribute((const void*)vit_linear_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, VIT_WAVES * 4 * 4096);
  }
#ifndef VIT_BATCHED_GEMM
#define VIT_BATCHED_GEMM 1   // 0: batched calls through the split-K latency kernel with 4 row tiles per workgroup (13.5 k images/s at B = 512)
#endif
  if (batched && VIT_BATCHED_GEMM && VIT_GEMM_LDS && avc_vit_gemm_lds(xs, wp, bias, residual, y, y_pre, nullptr, M, N, K, act, mt_packed, stream)) {
  } else if (batched && VIT_BATCHED_GEMM && (N & 127) == 0) {
    hipLaunchKernelGGL(vit_gemm_kernel, dim3(N / 128, mt_packed / 4), dim3(256), 0, s, xs, wp, bias, residual, y, y_pre, M, N, K, act);
  } else if (batched) {
    hipLaunchKernelGGL((vit_linear_kernel<4>), grid, block, lds, s, xs, wp, bias, residual, y, y_pre, M, N, K, act);
  } else {
    vit_launch_small(grid, lds, s, xs, wp, bias, residual, y, y_pre, M, N, K, act, nullptr, nullptr);
  }
  return avc_check_launch("avc_vit_linear");
}

extern "C" int avc_vit_linear(const float* x, const void* w_packed, const float* bias, const float* residual, float* y,
                              float* y_pre, int M, int N, int K, int act, void* workspace, void* stream) {
  return vit_linear_impl(x, nullptr, w_packed, bias, residual, y, y_pre, M, N, K, act, workspace, stream);
}
extern "C" int avc_vit_linear_bwd_gelu(const float* dy, const float* pre, const void* wt_packed, float* dx, int M, int N, int K,
                                       void* workspace, void* stream) {
  return vit_linear_impl(dy, pre, wt_packed, nullptr, nullptr, dx, nullptr, M, N, K, 0, workspace, stream);
}

// ---- the batched no-grad pipeline (scoring): activations stay packed bf16 between the kernels ----
// LayerNorm (fp32 statistics, eps inside the square root like torch) of x[M,K] straight into the packed operand of the next linear.
// One workgroup per 32-row tile: a wavefront normalises 8 rows into an LDS image of the tile, then the fragments go out whole.
#define LNP_LD (768 + 8)
__global__ __launch_bounds__(256) void vit_ln_pack_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, b8* __restrict__ xs, int M) {
  __shared__ __attribute__((aligned(16))) __bf16 tile[32][LNP_LD];
  constexpr int K = 768, KS = K / 16;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, mt = blockIdx.x;
  f4 g[3], bt[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    g[c] = *reinterpret_cast<const f4*>(gamma + 4 * (lane + 64 * c));
    bt[c] = *reinterpret_cast<const f4*>(beta + 4 * (lane + 64 * c));
  }
  for (int q = 0; q < 8; ++q) {
    const int i = 8 * wv + q, row = 32 * mt + i;
    f4 v[3];
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      v[c] = row < M ? *reinterpret_cast<const f4*>(X + (long)row * K + 4 * (lane + 64 * c)) : f4{0.f, 0.f, 0.f, 0.f};
      s1 += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s1 += __shfl_xor(s1, d);
    const float mean = s1 * (1.f / K);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int u = 0; u < 4; ++u) { const float dlt = v[c][u] - mean; s2 += dlt * dlt; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s2 += __shfl_xor(s2, d);
    const float rstd = rsqrtf(s2 * (1.f / K) + eps);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
      bf4 o;
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] = (__bf16)((v[c][u] - mean) * rstd * g[c][u] + bt[c][u]);
      *reinterpret_cast<bf4*>(&tile[i][4 * (lane + 64 * c)]) = o;
    }
  }
  __syncthreads();
  const int n = lane & 31, h = lane >> 5;
  for (int s = wv; s < KS; s += 4) xs[((long)mt * KS + s) * 64 + lane] = *reinterpret_cast<const b8*>(&tile[n][16 * s + 8 * h]);
}
// Up to 128 rows (the per-iteration calls) the tile kernel above is four workgroups walking 8 rows per wavefront one after the
// other -- a latency chain.  Here: one wavefront per row, all of a row's loads in flight at once, and no LDS: a lane holds 4
// consecutive columns of its row = one 8-byte half of a packed 16-byte chunk (column c of row r is slot c & 7 of lane (r & 31,
// (c >> 3) & 1) of k-step c >> 4 of row tile r >> 5).  Rows past M are not touched (the caller's buffer holds zeros there).
__device__ __forceinline__ void ln_row_stats(f4 (&v)[3], float eps, float& rstd) {   // on return v = x - mean
  constexpr int K = 768;
  float s1 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) s1 += v[c][0] + v[c][1] + v[c][2] + v[c][3];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s1 += __shfl_xor(s1, d);
  const float mean = s1 * (1.f / K);
  float s2 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[c][u] -= mean; s2 += v[c][u] * v[c][u]; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s2 += __shfl_xor(s2, d);
  rstd = rsqrtf(s2 * (1.f / K) + eps);
}
__device__ __forceinline__ void put_packed4(b8* xs, int row, int col, f4 v) {   // 4 consecutive columns from col (col % 4 == 0), K = 768
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  bf4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  char* p = reinterpret_cast<char*>(xs + (((long)(row >> 5)) * 48 + (col >> 4)) * 64 + (row & 31) + 32 * ((col >> 3) & 1)) + 2 * (col & 7);
  *reinterpret_cast<bf4*>(p) = o;
}
__global__ __launch_bounds__(256) void vit_ln_pack_rows_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, b8* __restrict__ xs, int M) {
  constexpr int K = 768;
  const int lane = threadIdx.x & 63, row = 4 * blockIdx.x + (threadIdx.x >> 6);
  if (row >= M) return;
  f4 v[3], g[3], bt[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    v[c] = *reinterpret_cast<const f4*>(X + (long)row * K + 4 * (lane + 64 * c));
    g[c] = *reinterpret_cast<const f4*>(gamma + 4 * (lane + 64 * c));
    bt[c] = *reinterpret_cast<const f4*>(beta + 4 * (lane + 64 * c));
  }
  float rstd;
  ln_row_stats(v, eps, rstd);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = v[c][u] * rstd * g[c][u] + bt[c][u];
    put_packed4(xs, row, 4 * (lane + 64 * c), o);
  }
}
extern "C" int avc_vit_ln_pack(const float* x, const float* gamma, const float* beta, float eps, int M, int K, void* xs_packed,
                               void* stream) {
  if (K != 768) { avc_set_error("avc_vit_ln_pack: built for the ViT-B/32 width 768"); return 1; }
  if (M <= 0) return 0;
  if (M <= 32 * VIT_MAX_MT) {
    hipLaunchKernelGGL(vit_ln_pack_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, (b8*)xs_packed, M);
    return avc_check_launch("avc_vit_ln_pack");
  }
  const int mt = ((M + 127) / 128) * 4;          // the row-tile groups of avc_vit_workspace_bytes (rows past M: LayerNorm of zeros = beta)
  hipLaunchKernelGGL(vit_ln_pack_kernel, dim3(mt), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, (b8*)xs_packed, M);
  return avc_check_launch("avc_vit_ln_pack");
}
// y = act(xs W^T + b) (+ residual) from an already packed operand (avc_vit_ln_pack, avc_vit_attention_fwd_packed or a previous
// call's ys_packed); ys_packed != NULL: the result leaves as the packed operand of the next linear instead of fp32 rows
extern "C" int avc_vit_linear_packed(const void* xs_packed, const void* w_packed, const float* bias, const float* residual, float* y,
                                     void* ys_packed, int M, int N, int K, int act, void* stream) {
  if (M <= 0) return 0;
  const int mt_packed = ((M + 127) / 128) * 4;
  if (!avc_vit_gemm_lds(xs_packed, w_packed, bias, residual, y, nullptr, ys_packed, M, N, K, act, mt_packed, stream)) {
    avc_set_error("avc_vit_linear_packed: shape not covered (N % 128, K % 32, no residual / activation mix with a packed output)");
    return 1;
  }
  return avc_check_launch("avc_vit_linear_packed");
}

// ---- the per-iteration training pipeline (1-2 images WITH a gradient to the pixels, M <= 128 rows): the same packed hand-offs for
// the latency kernel, forward and backward (clip_vit.BlocksFn).  Per block 7 launches forward (ln_pack, qkv, attention, out +
// residual, ln_pack, fc -> pre + packed QuickGELU, proj + residual) and 8 backward instead of 11 + 14 torch-autograd nodes with a
// packing launch in front of every linear, torch LayerNorm kernels and AccumulateGrad adds between them.
extern "C" int avc_vit_pack(const float* x, const float* gelu_pre, void* xs_packed, int M, int K, void* stream) {
  if (M <= 0) return 0;
  if (K & 15) { avc_set_error("avc_vit_pack: need K % 16 == 0"); return 1; }
  const int mt = (M + 31) / 32;
  hipLaunchKernelGGL(vit_pack_x_kernel, dim3(K / 16, mt), dim3(64), 0, (hipStream_t)stream, x, gelu_pre, (b8*)xs_packed, M, K);
  return avc_check_launch("avc_vit_pack");
}
// y = f(xs W^T + b) (+ residual), M <= 128 rows, from a packed operand.  act 0: identity; 1: QuickGELU (y_pre, if given, receives
// the pre-activation); 2: times QuickGELU'(gelu_pre[M,N]) (the backward of a QuickGELU layer folded into the product that yields the
// gradient of its output).  y (fp32 rows) and ys_packed (the operand of the next linear) are both optional.
extern "C" int avc_vit_linear_small(const void* xs_packed, const void* w_packed, const float* bias, const float* residual,
                                    const float* gelu_pre, float* y, float* y_pre, void* ys_packed, int M, int N, int K, int act,
                                    void* stream) {
  if (M <= 0) return 0;
  if (M > 32 * VIT_MAX_MT || (N & 31) || (K & 15)) { avc_set_error("avc_vit_linear_small: need M <= 128, N % 32 == 0, K % 16 == 0"); return 1; }
  if ((act == 2) != (gelu_pre != nullptr) || (!y && !ys_packed)) { avc_set_error("avc_vit_linear_small: act 2 <=> gelu_pre; y or ys_packed"); return 1; }
  const int mt = (M + 31) / 32;
  vit_launch_small(dim3(N / 32, mt), VIT_WAVES * 4096, (hipStream_t)stream, (const b8*)xs_packed, (const b8*)w_packed, bias, residual, y, y_pre,
                   M, N, K, act, (__bf16*)ys_packed, gelu_pre);
  return avc_check_launch("avc_vit_linear_small");
}
// backward of LayerNorm over the last dimension (768) + the residual branch's gradient: dx = LN'(x; gamma)^T dy (+ res), written as
// fp32 rows and (xs_packed != NULL) as the packed operand of the transposed linear that consumes it.  Statistics recomputed from x
// in fp32 (two-pass, like vit_ln_pack_kernel); one wavefront per row (vit_ln_pack_rows_kernel); rows past M of xs are not touched.
__global__ __launch_bounds__(256) void vit_ln_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                         const float* __restrict__ gamma, float eps, const float* __restrict__ res,
                                                         float* __restrict__ dX, b8* __restrict__ xs, int M) {
  constexpr int K = 768;
  const int lane = threadIdx.x & 63, row = 4 * blockIdx.x + (threadIdx.x >> 6);
  if (row >= M) return;
  f4 v[3], gy[3], rs[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const long o = (long)row * K + 4 * (lane + 64 * c);
    v[c] = *reinterpret_cast<const f4*>(X + o);
    gy[c] = *reinterpret_cast<const f4*>(dY + o) * *reinterpret_cast<const f4*>(gamma + 4 * (lane + 64 * c));
    rs[c] = res ? *reinterpret_cast<const f4*>(res + o) : f4{0.f, 0.f, 0.f, 0.f};
  }
  float rstd;
  ln_row_stats(v, eps, rstd);
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[c][u] *= rstd; c1 += gy[c][u]; c2 += gy[c][u] * v[c][u]; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { c1 += __shfl_xor(c1, d); c2 += __shfl_xor(c2, d); }
  c1 *= 1.f / K; c2 *= 1.f / K;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f4 dx;
#pragma unroll
    for (int u = 0; u < 4; ++u) dx[u] = rstd * (gy[c][u] - c1 - v[c][u] * c2) + rs[c][u];
    *reinterpret_cast<f4*>(dX + (long)row * K + 4 * (lane + 64 * c)) = dx;
    if (xs) put_packed4(xs, row, 4 * (lane + 64 * c), dx);
  }
}
extern "C" int avc_vit_ln_bwd(const float* dy, const float* x, const float* gamma, float eps, const float* residual_grad, float* dx,
                              void* xs_packed, int M, int K, void* stream) {
  if (K != 768) { avc_set_error("avc_vit_ln_bwd: built for the ViT-B/32 width 768"); return 1; }
  if (M <= 0) return 0;
  hipLaunchKernelGGL(vit_ln_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, eps, residual_grad, dx,
                     (b8*)xs_packed, M);
  return avc_check_launch("avc_vit_ln_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// attention: qkv [B,T,3*W] (q | k | v, head hd at column hd*64), out [B,T,W].  One 4-wave workgroup per (b, head).
// ---------------------------------------------------------------------------------------------------------
#define AT_T 50
#define AT_D 64
#define AT_LD 65   // +1 padding: thread i reads row i -> conflict-free

// AT_PARTS wavefronts per (b, head): wave `part` owns the keys j = part (mod AT_PARTS) for the scores and the AT_COLS output
// columns AT_COLS part .. for P V; every LDS read of K / V is a broadcast (a wave shares j), P rows are stride-51 (conflict-free).
// Only B x heads = 24 workgroups exist per call, so the kernels are latency chains: 8 wavefronts instead of 4 halve every
// per-thread loop (forward 26 -> see profiles/r03_ab_kernels.txt, backward 56 ->).
#ifndef AT_PARTS
#define AT_PARTS 8
#endif
#define AT_COLS (AT_D / AT_PARTS)
__device__ __forceinline__ float at_max_parts(const float (*r)[64], int i) {
  float m = r[0][i];
#pragma unroll
  for (int p = 1; p < AT_PARTS; ++p) m = fmaxf(m, r[p][i]);
  return m;
}
__device__ __forceinline__ float at_sum_parts(const float (*r)[64], int i) {
  float m = r[0][i];
#pragma unroll
  for (int p = 1; p < AT_PARTS; ++p) m += r[p][i];
  return m;
}
__global__ __launch_bounds__(64 * AT_PARTS) void vit_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int Wd,
                                                                     int heads, float scale) {
  __shared__ float Ks[AT_T][AT_LD], Vs[AT_T][AT_LD], Ps[AT_T][AT_T + 1], red[2][AT_PARTS][64];
  const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
  const int i = threadIdx.x & 63, part = threadIdx.x >> 6;
  const float* base = qkv + (long)b * AT_T * 3 * Wd + hd * AT_D;
  for (int e = threadIdx.x; e < AT_T * AT_D; e += 64 * AT_PARTS) {
    const int r = e / AT_D, c = e % AT_D;
    Ks[r][c] = base[(long)r * 3 * Wd + Wd + c];
    Vs[r][c] = base[(long)r * 3 * Wd + 2 * Wd + c];
  }
  __syncthreads();
  const bool live = i < AT_T;
  float mx = -1e30f;
  if (live) {
    float q[AT_D];
#pragma unroll
    for (int c = 0; c < AT_D; ++c) q[c] = base[(long)i * 3 * Wd + c] * scale;
    for (int j = part; j < AT_T; j += AT_PARTS) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < AT_D; ++c) s += q[c] * Ks[j][c];
      Ps[i][j] = s;
      mx = fmaxf(mx, s);
    }
  }
  red[0][part][i] = mx;
  __syncthreads();
  float sum = 0.f;
  if (live) {
    mx = at_max_parts(red[0], i);
    for (int j = part; j < AT_T; j += AT_PARTS) { const float e = __expf(Ps[i][j] - mx); Ps[i][j] = e; sum += e; }
  }
  red[1][part][i] = sum;
  __syncthreads();
  if (!live) return;
  const float inv = 1.f / at_sum_parts(red[1], i);
  float o[AT_COLS];
#pragma unroll
  for (int c = 0; c < AT_COLS; ++c) o[c] = 0.f;
  for (int j = 0; j < AT_T; ++j) {
    const float pj = Ps[i][j] * inv;
#pragma unroll
    for (int c = 0; c < AT_COLS; ++c) o[c] += pj * Vs[j][AT_COLS * part + c];
  }
  float* op = out + ((long)b * AT_T + i) * Wd + hd * AT_D + AT_COLS * part;
#pragma unroll
  for (int c = 0; c < AT_COLS; ++c) op[c] = o[c];
}

// Text tower (clip/model.py encode_text: 77 tokens, 8 heads of 64, causal mask): forward only -- prompts are encoded once per
// run (main.py:273-288).  One workgroup per (sequence, head), thread i = query row i, K/V rows in LDS, online softmax.
#define TA_TMAX 128
__global__ __launch_bounds__(TA_TMAX) void text_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T,
                                                                int Wd, int heads, float scale, int causal) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  float (*Ks)[AT_LD] = reinterpret_cast<float (*)[AT_LD]>(tsm);
  float (*Vs)[AT_LD] = Ks + T;
  const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
  const int i = threadIdx.x;
  const float* base = qkv + (long)b * T * 3 * Wd + hd * AT_D;
  for (int e = threadIdx.x; e < T * AT_D; e += TA_TMAX) {
    const int r = e / AT_D, c = e % AT_D;
    Ks[r][c] = base[(long)r * 3 * Wd + Wd + c];
    Vs[r][c] = base[(long)r * 3 * Wd + 2 * Wd + c];
  }
  __syncthreads();
  if (i >= T) return;
  float q[AT_D], o[AT_D];
#pragma unroll
  for (int c = 0; c < AT_D; ++c) { q[c] = base[(long)i * 3 * Wd + c] * scale; o[c] = 0.f; }
  float mx = -1e30f, sum = 0.f;
  const int jend = causal ? i + 1 : T;
  for (int j = 0; j < jend; ++j) {
    float sc = 0.f;
#pragma unroll
    for (int c = 0; c < AT_D; ++c) sc += q[c] * Ks[j][c];
    const float mn = fmaxf(mx, sc);
    const float corr = __expf(mx - mn), pj = __expf(sc - mn);
    sum = sum * corr + pj;
#pragma unroll
    for (int c = 0; c < AT_D; ++c) o[c] = o[c] * corr + pj * Vs[j][c];
    mx = mn;
  }
  const float inv = 1.f / sum;
  float* op = out + ((long)b * T + i) * Wd + hd * AT_D;
#pragma unroll
  for (int c = 0; c < AT_D; ++c) op[c] = o[c] * inv;
}

__global__ __launch_bounds__(64 * AT_PARTS) void vit_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                     float* __restrict__ dqkv, int Wd, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float (*Ks)[AT_LD] = reinterpret_cast<float (*)[AT_LD]>(sm);
  float (*Vs)[AT_LD] = Ks + AT_T;
  float (*Qs)[AT_LD] = Vs + AT_T;
  float (*Ds)[AT_LD] = Qs + AT_T;                       // dO
  float (*Ps)[AT_T + 1] = reinterpret_cast<float (*)[AT_T + 1]>(Ds + AT_T);   // P
  float (*Ss)[AT_T + 1] = Ps + AT_T;                    // dS (already scaled)
  float (*red)[AT_PARTS][64] = reinterpret_cast<float (*)[AT_PARTS][64]>(Ss + AT_T);   // [3][parts][64]: max, sum, sum p d
  const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
  const int i = threadIdx.x & 63, part = threadIdx.x >> 6;
  const bool live = i < AT_T;
  const float* base = qkv + (long)b * AT_T * 3 * Wd + hd * AT_D;
  const float* dob = dout + (long)b * AT_T * Wd + hd * AT_D;
  for (int e = threadIdx.x; e < AT_T * AT_D; e += 64 * AT_PARTS) {
    const int r = e / AT_D, c = e % AT_D;
    Qs[r][c] = base[(long)r * 3 * Wd + c];
    Ks[r][c] = base[(long)r * 3 * Wd + Wd + c];
    Vs[r][c] = base[(long)r * 3 * Wd + 2 * Wd + c];
    Ds[r][c] = dob[(long)r * Wd + c];
  }
  __syncthreads();
  // scores and dP = dO V^T for the keys of this wave
  float mx = -1e30f;
  if (live) {
    for (int j = part; j < AT_T; j += AT_PARTS) {
      float s = 0.f, d = 0.f;
#pragma unroll
      for (int c = 0; c < AT_D; ++c) { s += Qs[i][c] * Ks[j][c]; d += Ds[i][c] * Vs[j][c]; }
      s *= scale;
      Ps[i][j] = s;
      Ss[i][j] = d;
      mx = fmaxf(mx, s);
    }
  }
  red[0][part][i] = mx;
  __syncthreads();
  float sum = 0.f;
  if (live) {
    mx = at_max_parts(red[0], i);
    for (int j = part; j < AT_T; j += AT_PARTS) { const float e = __expf(Ps[i][j] - mx); Ps[i][j] = e; sum += e; }
  }
  red[1][part][i] = sum;
  __syncthreads();
  float dsum = 0.f;
  if (live) {
    const float inv = 1.f / at_sum_parts(red[1], i);
    for (int j = part; j < AT_T; j += AT_PARTS) {
      const float pj = Ps[i][j] * inv;
      Ps[i][j] = pj;
      dsum += pj * Ss[i][j];
    }
  }
  red[2][part][i] = dsum;
  __syncthreads();
  if (live) {
    dsum = at_sum_parts(red[2], i);
    for (int j = part; j < AT_T; j += AT_PARTS) Ss[i][j] = Ps[i][j] * (Ss[i][j] - dsum) * scale;
  }
  __syncthreads();
  if (!live) return;
  // dQ row i, dK / dV row j = i: this wave's AT_COLS columns
  const int c0 = AT_COLS * part;
  float dq[AT_COLS], dk[AT_COLS], dv[AT_COLS];
#pragma unroll
  for (int c = 0; c < AT_COLS; ++c) { dq[c] = 0.f; dk[c] = 0.f; dv[c] = 0.f; }
  for (int r = 0; r < AT_T; ++r) {
    const float ds_q = Ss[i][r];
    const float ds_k = Ss[r][i], pr = Ps[r][i];
#pragma unroll
    for (int c = 0; c < AT_COLS; ++c) {
      dq[c] += ds_q * Ks[r][c0 + c];
      dk[c] += ds_k * Qs[r][c0 + c];
      dv[c] += pr * Ds[r][c0 + c];
    }
  }
  float* dqp = dqkv + ((long)b * AT_T + i) * 3 * Wd + hd * AT_D + c0;
#pragma unroll
  for (int c = 0; c < AT_COLS; ++c) { dqp[c] = dq[c]; dqp[Wd + c] = dk[c]; dqp[2 * Wd + c] = dv[c]; }
}

// matrix-core attention (avc_vit_attn.hip); VIT_ATTN_MFMA=0 keeps the fp32 VALU kernels above (the A/B of profiles/r03_ab_kernels.txt)
#ifndef VIT_ATTN_MFMA
#define VIT_ATTN_MFMA 1
#endif
int avc_attn_fwd_mfma(const float* qkv, float* out, int B, int width, int heads, void* stream);
int avc_attn_bwd_mfma(const float* qkv, const float* dout, float* dqkv, int B, int width, int heads, void* stream);
int avc_attn_fwd_mfma_packed(const float* qkv, void* out_packed, int B, int width, int heads, void* stream);
int avc_attn_bwd_mfma_packed(const float* qkv, const float* dout, void* dqkv_packed, int B, int width, int heads, void* stream);
extern "C" int avc_vit_attention_bwd_packed(const float* qkv, const float* dout, void* dqkv_packed, int B, int T, int width, int heads,
                                            void* stream) {
  if (T != AT_T || width != heads * AT_D) { avc_set_error("avc_vit_attention: built for 50 tokens, head dim 64"); return 1; }
  return avc_attn_bwd_mfma_packed(qkv, dout, dqkv_packed, B, width, heads, stream);
}
extern "C" int avc_vit_attention_fwd_packed(const float* qkv, void* out_packed, int B, int T, int width, int heads, void* stream) {
  if (T != AT_T || width != heads * AT_D) { avc_set_error("avc_vit_attention: built for 50 tokens, head dim 64"); return 1; }
  return avc_attn_fwd_mfma_packed(qkv, out_packed, B, width, heads, stream);
}

extern "C" int avc_vit_attention_fwd(const float* qkv, float* out, int B, int T, int width, int heads, void* stream) {
  if (T != AT_T || width != heads * AT_D) { avc_set_error("avc_vit_attention: built for 50 tokens, head dim 64"); return 1; }
  if (VIT_ATTN_MFMA) return avc_attn_fwd_mfma(qkv, out, B, width, heads, stream);
  hipLaunchKernelGGL(vit_attn_fwd_kernel, dim3(B * heads), dim3(64 * AT_PARTS), 0, (hipStream_t)stream, qkv, out, width, heads, 0.125f);
  return avc_check_launch("avc_vit_attention_fwd");
}
extern "C" int avc_text_attention_fwd(const float* qkv, float* out, int B, int T, int width, int heads, int causal, void* stream) {
  if (T < 1 || T > TA_TMAX || width != heads * AT_D) { avc_set_error("avc_text_attention_fwd: 1 <= T <= 128, head dim 64"); return 1; }
  if (B <= 0) return 0;
  const size_t lds = 2 * (size_t)T * AT_LD * sizeof(float);
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen)) {
    hipFuncSetAttribute((const void*)text_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TA_TMAX * AT_LD * 4);
  }
  hipLaunchKernelGGL(text_attn_fwd_kernel, dim3(B * heads), dim3(TA_TMAX), lds, (hipStream_t)stream, qkv, out, T, width, heads,
                     0.125f, causal);
  return avc_check_launch("avc_text_attention_fwd");
}
extern "C" int avc_vit_attention_bwd(const float* qkv, const float* dout, float* dqkv, int B, int T, int width, int heads,
                                     void* stream) {
  if (T != AT_T || width != heads * AT_D) { avc_set_error("avc_vit_attention: built for 50 tokens, head dim 64"); return 1; }
  if (VIT_ATTN_MFMA) return avc_attn_bwd_mfma(qkv, dout, dqkv, B, width, heads, stream);
  const size_t lds = (4 * AT_T * AT_LD + 2 * AT_T * (AT_T + 1) + 3 * AT_PARTS * 64) * sizeof(float);
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen)) {
    hipFuncSetAttribute((const void*)vit_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(vit_attn_bwd_kernel, dim3(B * heads), dim3(64 * AT_PARTS), lds, (hipStream_t)stream, qkv, dout, dqkv, width, heads,
                     0.125f);
  return avc_check_launch("avc_vit_attention_bwd");
}
