// Multi-head self-attention of CLIP's ResidualAttentionBlock (OpenAI clip/model.py; perceptor.encode_image, main.py:512,524) on
// the matrix core: 50 tokens (padded to 64), head dim 64, one 4-wavefront workgroup per (image, head), bf16 operands with fp32
// accumulation (the class the reference's fp16 CLIP runs in), softmax statistics in fp32.
//
// Everything is computed TRANSPOSED, S^T = K Q^T, so that a wavefront's accumulator tile (v_mfma_f32_32x32x16_bf16 C layout: lane
// = column, 16 rows in registers, the other 16 rows in lane ^ 32) holds, for ITS query (column), all keys of the tile in registers:
// the softmax reductions over the keys are register loops plus one lane-32 exchange, no LDS.  As in the MLP engine
// (avc_common.h), the accumulator tile -- after the softmax and a cvt to bf16 -- IS the B operand of the next product
// (O^T = V^T P^T, dQ^T = K^T dS^T): its k-slot (s, h, j) carries key kappa(s,h,j) = 32 (s >> 1) + 16 (s & 1) + 8 (j >> 2) + 4 h + (j & 3),
// and the A operand (V^T, K^T from LDS, stored [d][key]) is read with the same permutation (two 8-byte reads per fragment).
// The products that contract over the QUERIES (dV = P^T dO, dK = dS^T Q) need P^T / dS^T as A operands, lane = key row: those two
// tiles go through LDS once ([key][query], bf16).
#include "avc_common.h"
#include "../../include/avc.h"

#define ATM_T 50
#define ATM_TP 64     // padded tokens
#define ATM_D 64
#define ATM_LD 72     // row stride of the bf16 LDS matrices (144 B: 16-byte aligned rows, 36-bank skew)

typedef __bf16 bf;
typedef short s4v __attribute__((ext_vector_type(4)));

struct AtmMats {        // bf16 matrices of one (image, head) in LDS
  bf Q[ATM_TP][ATM_LD];     // [token][d], pre-scaled by 1/sqrt(d)
  bf K[ATM_TP][ATM_LD];
  bf VT[ATM_D][ATM_LD];     // [d][token]
};
struct AtmMatsBwd {
  bf Q[ATM_TP][ATM_LD], K[ATM_TP][ATM_LD], V[ATM_TP][ATM_LD], dO[ATM_TP][ATM_LD];   // [token][d]
  bf KT[ATM_D][ATM_LD], QT[ATM_D][ATM_LD], dOT[ATM_D][ATM_LD];                      // [d][token]
  bf PT[ATM_TP][ATM_LD], dST[ATM_TP][ATM_LD];                                       // [key][query]
};

// natural fragment: row (or column) `r`, k-slots 16 s + 8 h + 0..7 of a [.][k] row-major matrix
__device__ __forceinline__ b8 frag_nat(const bf (*m)[ATM_LD], int r, int s, int h) {
  return *reinterpret_cast<const b8*>(&m[r][16 * s + 8 * h]);
}
// permuted fragment: k-slot j of k-step s carries column kappa(s,h,j) (see the header): 4 + 4 contiguous entries
__device__ __forceinline__ b8 frag_perm(const bf (*m)[ATM_LD], int r, int s, int h) {
  const int k0 = 32 * (s >> 1) + 16 * (s & 1) + 4 * h;
  struct { s4v lo, hi; } v;
  v.lo = *reinterpret_cast<const s4v*>(&m[r][k0]);
  v.hi = *reinterpret_cast<const s4v*>(&m[r][k0 + 8]);
  return __builtin_bit_cast(b8, v);
}
__device__ __forceinline__ void put4(bf* p, f4 v) {
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  bf4 o = {(bf)v[0], (bf)v[1], (bf)v[2], (bf)v[3]};
  *reinterpret_cast<bf4*>(p) = o;
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// S^T tiles of query tile `ti` (columns) against both key tiles, then the softmax over the keys: on return p[tj][r] = P[i][j] for
// query i = 32 ti + (lane & 31), key j = 32 tj + acc_row(r, h)
__device__ __forceinline__ void scores_softmax(const bf (*Q)[ATM_LD], const bf (*K)[ATM_LD], int ti, int lane, facc (&p)[2]) {
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
    for (int r = 0; r < 16; ++r) p[tj][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) p[tj] = MF<b8>::mma(frag_nat(K, 32 * tj + n, s, h), frag_nat(Q, 32 * ti + n, s, h), p[tj]);
  }
  float mx = -1e30f;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (32 * tj + acc_row(r, h) >= ATM_T) p[tj][r] = -1e30f;   // padded keys
      mx = fmaxf(mx, p[tj][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 16; ++r) { p[tj][r] = __expf(p[tj][r] - mx); sum += p[tj][r]; }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.f / sum;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 16; ++r) p[tj][r] *= inv;
}
// accumulator tiles [key tile][16] (x scale) -> the four B-operand k-steps over the keys
__device__ __forceinline__ void acc_to_b(const facc (&a)[2], float scale, b8 (&f)[4]) {
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[2 * tj][j] = (bf)(a[tj][j] * scale); f[2 * tj + 1][j] = (bf)(a[tj][8 + j] * scale); }
}

// element (row, col) of a packed bf16 operand with KS k-steps per row tile = slot col & 7 of lane (row & 31, (col >> 3) & 1) of k-step
// col >> 4 of row tile row >> 5
__device__ __forceinline__ void put_packed_elem(bf* xs, int KS, long row, int col, float v) {
  xs[(((row >> 5) * KS + (col >> 4)) * 64 + (row & 31) + 32 * ((col >> 3) & 1)) * 8 + (col & 7)] = (bf)v;
}
// PACKED: `out` is the packed bf16 operand of the out-projection ([row tile][k-step][lane (row, half)][8 columns], rows = b * 50 +
// token) instead of fp32 [B,T,W] -- the batched scoring pipeline (avc_vit_attention_fwd_packed)
template <bool PACKED>
__global__ __launch_bounds__(256) void vit_attn_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int Wd, int heads,
                                                                float scale) {
  __shared__ __attribute__((aligned(16))) AtmMats m;
  const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* base = qkv + (long)b * ATM_T * 3 * Wd + hd * ATM_D;
  for (int e = threadIdx.x; e < ATM_TP * (ATM_D / 4); e += 256) {
    const int r = e / (ATM_D / 4), c = 4 * (e % (ATM_D / 4));
    f4 q = {0.f, 0.f, 0.f, 0.f}, k = q, v = q;
    if (r < ATM_T) {
      const float* rp = base + (long)r * 3 * Wd + c;
      q = *reinterpret_cast<const f4*>(rp) * scale; k = *reinterpret_cast<const f4*>(rp + Wd); v = *reinterpret_cast<const f4*>(rp + 2 * Wd);
    }
    put4(&m.Q[r][c], q); put4(&m.K[r][c], k);
#pragma unroll
    for (int u = 0; u < 4; ++u) m.VT[c + u][r] = (bf)v[u];
  }
  __syncthreads();
  if (wv >= 2) return;              // one wavefront per 32-query tile
  const int ti = wv, n = lane & 31, h = lane >> 5;
  facc p[2];
  scores_softmax(m.Q, m.K, ti, lane, p);
  b8 pf[4];
  acc_to_b(p, 1.f, pf);
  const int i = 32 * ti + n;
#pragma unroll
  for (int td = 0; td < 2; ++td) {   // O^T tile: rows d, column = this lane's query
    facc o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) o = MF<b8>::mma(frag_perm(m.VT, 32 * td + n, s, h), pf[s], o);
    if (PACKED) {
      // 2-byte stores straight from the accumulator registers (a version that assembled whole 16-byte chunks through a lane-pair
      // exchange indexed the accumulator by the lane's half -- a dynamic register index -- and took 9.1 instead of 5.3 us)
      if (i < ATM_T) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          put_packed_elem(reinterpret_cast<bf*>(out), Wd >> 4, (long)b * ATM_T + i, hd * ATM_D + 32 * td + acc_row(r, h), o[r]);
      }
    } else if (i < ATM_T) {
      float* op = out + ((long)b * ATM_T + i) * Wd + hd * ATM_D + 32 * td;
#pragma unroll
      for (int r = 0; r < 16; ++r) op[acc_row(r, h)] = o[r];
    }
  }
}

// PACKED: dqkv leaves as the packed operand of the transposed in-projection (K = 3 W) instead of fp32 rows: the per-iteration training
// pipeline (clip_vit.BlocksFn)
template <bool PACKED>
__global__ __launch_bounds__(256) void vit_attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                float* __restrict__ dqkv, int Wd, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  AtmMatsBwd& m = *reinterpret_cast<AtmMatsBwd*>(smem);
  const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = lane & 31, h = lane >> 5;
  const float* base = qkv + (long)b * ATM_T * 3 * Wd + hd * ATM_D;
  const float* dob = dout + (long)b * ATM_T * Wd + hd * ATM_D;
  for (int e = threadIdx.x; e < ATM_TP * (ATM_D / 4); e += 256) {
    const int r = e / (ATM_D / 4), c = 4 * (e % (ATM_D / 4));
    f4 q = {0.f, 0.f, 0.f, 0.f}, k = q, v = q, d = q;
    if (r < ATM_T) {
      const float* rp = base + (long)r * 3 * Wd + c;
      q = *reinterpret_cast<const f4*>(rp) * scale; k = *reinterpret_cast<const f4*>(rp + Wd); v = *reinterpret_cast<const f4*>(rp + 2 * Wd);
      d = *reinterpret_cast<const f4*>(dob + (long)r * Wd + c);
    }
    put4(&m.Q[r][c], q); put4(&m.K[r][c], k); put4(&m.V[r][c], v); put4(&m.dO[r][c], d);
#pragma unroll
    for (int u = 0; u < 4; ++u) { m.QT[c + u][r] = (bf)q[u]; m.KT[c + u][r] = (bf)k[u]; m.dOT[c + u][r] = (bf)d[u]; }
  }
  __syncthreads();
  if (wv < 2) {
    // ---- per query tile: P^T, dP^T = V dO^T, dS^T = P^T (dP^T - sum_j P dP), dQ^T = K^T dS^T (x scale)
    const int ti = wv;
    facc p[2], dp[2];
    scores_softmax(m.Q, m.K, ti, lane, p);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[tj][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) dp[tj] = MF<b8>::mma(frag_nat(m.V, 32 * tj + n, s, h), frag_nat(m.dO, 32 * ti + n, s, h), dp[tj]);
    }
    float dsum = 0.f;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) dsum += p[tj][r] * dp[tj][r];
    dsum += __shfl_xor(dsum, 32);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[tj][r] = p[tj][r] * (dp[tj][r] - dsum);      // dS^T (w.r.t. the scaled scores)
    b8 dsf[4];
    acc_to_b(dp, scale, dsf);                       // dQ = scale * dS K
    const int i = 32 * ti + n;
#pragma unroll
    for (int td = 0; td < 2; ++td) {
      facc o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) o = MF<b8>::mma(frag_perm(m.KT, 32 * td + n, s, h), dsf[s], o);
      if (i < ATM_T) {
        if (PACKED) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            put_packed_elem(reinterpret_cast<bf*>(dqkv), (3 * Wd) >> 4, (long)b * ATM_T + i, hd * ATM_D + 32 * td + acc_row(r, h), o[r]);
        } else {
          float* qp = dqkv + ((long)b * ATM_T + i) * 3 * Wd + hd * ATM_D + 32 * td;
#pragma unroll
          for (int r = 0; r < 16; ++r) qp[acc_row(r, h)] = o[r];
        }
      }
    }
    // P^T and dS^T to LDS as [key][query] for the products that contract over the queries
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * tj + acc_row(r, h);
        m.PT[j][i] = (bf)p[tj][r];
        m.dST[j][i] = (bf)dp[tj][r];
      }
  }
  __syncthreads();
  // ---- dV = P^T dO, dK = dS^T Q_scaled: 4 + 4 tiles [key tile][d tile], two of each per wavefront; contraction over all 64 queries
  const int tj = wv & 1, td = wv >> 1;
  facc dv, dk;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    dv = MF<b8>::mma(frag_nat(m.PT, 32 * tj + n, s, h), frag_nat(m.dOT, 32 * td + n, s, h), dv);
    dk = MF<b8>::mma(frag_nat(m.dST, 32 * tj + n, s, h), frag_nat(m.QT, 32 * td + n, s, h), dk);
  }
  // C layout: lane = column d = 32 td + n, rows = keys 32 tj + acc_row(r, h)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = 32 * tj + acc_row(r, h);
    if (j < ATM_T) {
      if (PACKED) {
        const int col = hd * ATM_D + 32 * td + n;
        put_packed_elem(reinterpret_cast<bf*>(dqkv), (3 * Wd) >> 4, (long)b * ATM_T + j, Wd + col, dk[r]);
        put_packed_elem(reinterpret_cast<bf*>(dqkv), (3 * Wd) >> 4, (long)b * ATM_T + j, 2 * Wd + col, dv[r]);
      } else {
        float* kp = dqkv + ((long)b * ATM_T + j) * 3 * Wd + hd * ATM_D + 32 * td + n;
        kp[Wd] = dk[r];
        kp[2 * Wd] = dv[r];
      }
    }
  }
}

int avc_attn_fwd_mfma(const float* qkv, float* out, int B, int width, int heads, void* stream) {
  hipLaunchKernelGGL(vit_attn_fwd_mfma_kernel<false>, dim3(B * heads), dim3(256), 0, (hipStream_t)stream, qkv, out, width, heads, 0.125f);
  return avc_check_launch("avc_vit_attention_fwd");
}
int avc_attn_fwd_mfma_packed(const float* qkv, void* out_packed, int B, int width, int heads, void* stream) {
  hipLaunchKernelGGL(vit_attn_fwd_mfma_kernel<true>, dim3(B * heads), dim3(256), 0, (hipStream_t)stream, qkv, (float*)out_packed, width, heads,
                     0.125f);
  return avc_check_launch("avc_vit_attention_fwd_packed");
}
int avc_attn_bwd_mfma(const float* qkv, const float* dout, float* dqkv, int B, int width, int heads, void* stream) {
  const int lds = (int)sizeof(AtmMatsBwd);
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen))
    (void)hipFuncSetAttribute((const void*)vit_attn_bwd_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(vit_attn_bwd_mfma_kernel<false>, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, qkv, dout, dqkv, width, heads, 0.125f);
  return avc_check_launch("avc_vit_attention_bwd");
}
int avc_attn_bwd_mfma_packed(const float* qkv, const float* dout, void* dqkv_packed, int B, int width, int heads, void* stream) {
  const int lds = (int)sizeof(AtmMatsBwd);
  static unsigned long long attr_seen = 0;
  if (avc_first_use_on_device(attr_seen))
    (void)hipFuncSetAttribute((const void*)vit_attn_bwd_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(vit_attn_bwd_mfma_kernel<true>, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, qkv, dout, (float*)dqkv_packed, width,
                     heads, 0.125f);
  return avc_check_launch("avc_vit_attention_bwd_packed");
}
