// Common device helpers for the AvatarCLIP/NeuS gfx950 kernels.
//
// MLP engine: one wavefront owns 32 sample points and keeps every activation in registers.
// All layer products are computed TRANSPOSED (H^T = W * X^T) with v_mfma_f32_32x32x16_{f16,bf16}:
//     A operand = packed weights   (lane l: row 32t+(l&31), 8 k-slots of half h=l>>5)
//     B operand = activations      (lane l: point p=l&31,  8 k-slots of half h=l>>5)
//     C/D       = 32 features x 32 points, lane l holds point p=l&31 and the 16 feature rows
//                 row(r,h) = (r&3) + 8*(r>>2) + 4*h                      (MI355X C/D layout)
// so the accumulator of out-tile t becomes, after the activation and a cvt to 16 bit, exactly the B operand
// of k-steps 2t (regs 0..7) and 2t+1 (regs 8..15) of the next layer -- no LDS, no cross-lane traffic.
// The k-slot <-> feature permutation this induces,
//     feature(s,h,j) = 32*(s>>1) + 16*(s&1) + 8*(j>>2) + 4*h + (j&3),
// is baked into the host-side weight packing (avatarclip_amd/packing.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float facc __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define AVC_BETA 100.0f
#define AVC_INV_BETA 0.01f
// Base-2 softplus units.  With S = beta*log2(e) the kernels carry  t = S*a  (pre-activation) and  H = S*h  (activation):
//   H = log2(1 + 2^t) = max(t,0) + log2(1 + 2^-|t|),   sigma(beta a) = 1 - 2^-H,
// i.e. one v_exp_f32 + one v_log_f32 and 3 plain VALU ops per element, no range fix-ups (arguments are in the safe
// range by construction).  S is folded into the packed layer-0 weights and the biases, 1/S into the last layer
// (packing.py); hidden HxH weights are unchanged because W (S h) = S (W h).
#define AVC_S 144.26950408889634f

// ---- offsets of the packed parameter blobs (element units of the blob's dtype); mirrored by packing.py,
// ---- which parses this enum.  *_T = transposed weight (rows = in-features).
enum AvcOff {
  // 16-bit packs (same slot numbering for the f16 and the bf16 blob)
  OFF_W0 = 0, OFF_WM0, OFF_WM1, OFF_WS, OFF_WL,          // SDF forward: layer0, middle0, middle1, skip, last(rows 1..H)
  OFF_W0T, OFF_WM0T, OFF_WM1T, OFF_WST, OFF_WLT,         // SDF transposed (WLT: rows = [skip feats | pe slots], K = H feature rows 1..H)
  OFF_C0, OFF_CM0, OFF_CH,                               // colour forward: layer0 (K = feat + [x,n]), middle, heads(6 rows)
  OFF_C0T, OFF_CM0T, OFF_CHT,                            // colour transposed
  OFF_W0G,                                               // layer 0, forward orientation, UNscaled (second-order sweep)
  // fp32 tables
  OFF_B0, OFF_BM0, OFF_BM1, OFF_BS, OFF_BL, OFF_BL0,     // packed biases (acc order), BL0 = scalar sdf bias
  OFF_WL0_ACC, OFF_WL0_FRAG, OFF_WL0_PE,                 // row 0 of the last layer / sqrt2 in acc order, frag order, pe-slot order
  OFF_CB0, OFF_CBM0, OFF_CBH,
  OFF_TAB_END,                                           // length of the fp32 table (floats)
  OFF_COUNT
};
struct AvcOffsets { int v[OFF_COUNT]; };

// ---------------------------------------------------------------------------------------------
template <typename V> struct MF;
template <> struct MF<h8> {
  typedef _Float16 S;
  static __device__ __forceinline__ facc mma(h8 a, h8 b, facc c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct MF<b8> {
  typedef __bf16 S;
  static __device__ __forceinline__ facc mma(b8 a, b8 b, facc c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// One 32-feature output tile: acc = sum_s Wp[s] * in[s].  wp points at this lane's 16-B chunk of k-step 0
// of the tile; consecutive k-steps are 64 chunks apart.
template <typename V, int KS>
__device__ __forceinline__ facc tile_gemm(const V* __restrict__ wp, const V (&in)[KS]) {
  facc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = MF<V>::mma(wp[s * 64], in[s], acc);
  __builtin_amdgcn_sched_barrier(0);
  return acc;
}
// same, but the K dimension is split over two register arrays (skip connection / concatenated inputs)
template <typename V, int KA, int KB>
__device__ __forceinline__ facc tile_gemm2(const V* __restrict__ wp, const V (&ina)[KA], const V (&inb)[KB]) {
  facc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < KA; ++s) acc = MF<V>::mma(wp[s * 64], ina[s], acc);
#pragma unroll
  for (int s = 0; s < KB; ++s) acc = MF<V>::mma(wp[(KA + s) * 64], inb[s], acc);
  __builtin_amdgcn_sched_barrier(0);
  return acc;
}

// Pointers that went through a register-laundering asm lose their address space: hipcc then emits FLAT loads/stores, which
// count on vmcnt AND lgkmcnt (every "wait for my LDS operands" also waits for them) and are ordered against LDS traffic.
// Everything the kernels touch through such pointers is global memory -- say so.
#define AVC_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ const AVC_GLOBAL T* as_global(const T* p) {
  return (const AVC_GLOBAL T*)(p);
}
template <typename T> __device__ __forceinline__ AVC_GLOBAL T* as_global(T* p) { return (AVC_GLOBAL T*)(p); }

// Streaming traffic (parked activations, weight-gradient panels): written once, read once or a few times much later, far larger
// than the 4 MB L2 of an XCD.  Non-temporal accesses keep it from evicting the packed weights every workgroup re-reads.
#ifndef AVC_NO_NT_STORE
#define AVC_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define AVC_NT_STORE(v, p) (*(p) = (v))
#endif
#ifndef AVC_NO_NT_LOAD
#define AVC_NT_LOAD(p) __builtin_nontemporal_load((p))
#else
#define AVC_NT_LOAD(p) (*(p))
#endif

// packed per-tile fp32 table [tile][half][16] -> this lane's 16 values
__device__ __forceinline__ void load16(const float* __restrict__ tab, int t, int h, float (&out)[16]) {
  const AVC_GLOBAL f4* p = as_global(reinterpret_cast<const f4*>(tab + (t * 2 + h) * 16));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f4 v = p[q];
    out[4 * q + 0] = v[0]; out[4 * q + 1] = v[1]; out[4 * q + 2] = v[2]; out[4 * q + 3] = v[3];
  }
}
// the same from the copy of the table the forward kernels keep in LDS (tab_to_lds): a global load in an epilogue sits behind
// the LDS-DMA of the next weight group in the in-order vmcnt queue and stalls the wave until that whole group has landed
// (ablation: 22-30 % of the forward kernels' time); ds_read has its own counter and ~100 cycles of latency
#define AVC_LDS __attribute__((address_space(3)))
typedef const AVC_LDS float* lds_tab_t;
__device__ __forceinline__ void load16(lds_tab_t tab, int t, int h, float (&out)[16]) {
  const AVC_LDS f4* p = reinterpret_cast<const AVC_LDS f4*>(tab + (t * 2 + h) * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f4 v = p[q];
    out[4 * q + 0] = v[0]; out[4 * q + 1] = v[1]; out[4 * q + 2] = v[2]; out[4 * q + 3] = v[3];
  }
}
__device__ __forceinline__ void load8(lds_tab_t tab, int s, int h, float (&out)[8]) {
  const AVC_LDS f4* p = reinterpret_cast<const AVC_LDS f4*>(tab + (s * 2 + h) * 8);
  f4 a = p[0], b = p[1];
  out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = a[3];
  out[4] = b[0]; out[5] = b[1]; out[6] = b[2]; out[7] = b[3];
}
#define AVC_TAB_LDS_BYTES 10240   // fp32 table of the full net: 2292 floats
// copy the fp32 table into LDS (all threads of the workgroup; the caller synchronises before the first use)
__device__ __forceinline__ lds_tab_t tab_to_lds(char* lds_dst, const float* __restrict__ tab, int n) {
  AVC_LDS float* d = (AVC_LDS float*)lds_dst;
  for (int i = threadIdx.x; i < n; i += blockDim.x) d[i] = tab[i];
  return (lds_tab_t)d;
}
// frag-order fp32 table [kstep][half][8] -> this lane's 8 values
__device__ __forceinline__ void load8(const float* __restrict__ tab, int s, int h, float (&out)[8]) {
  const AVC_GLOBAL f4* p = as_global(reinterpret_cast<const f4*>(tab + (s * 2 + h) * 8));
  f4 a = p[0], b = p[1];
  out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = a[3];
  out[4] = b[0]; out[5] = b[1]; out[6] = b[2]; out[7] = b[3];
}

// max(t, 0) as ONE instruction for a value that comes straight out of an MFMA: fmaxf() there makes hipcc emit a second v_max as
// canonicalisation (the bias add used to provide it for free); v_med3_f32(t, 0, 3e38) needs none.  (NOT inline asm: the hazard
// recogniser does not look inside an asm, and a VALU read of an MFMA result without the wait states it inserts returns the
// previous contents of the first result registers -- seen as 12 wrong columns in the last tile of a layer.)
#ifndef AVC_SOFTPLUS_DIRECT
#define AVC_SOFTPLUS_DIRECT 1
#endif
__device__ __forceinline__ float relu_raw(float t) { return __builtin_amdgcn_fmed3f(t, 0.f, 3.0e38f); }   // (a finite bound: +inf folds back to fmaxnum)
__device__ __forceinline__ float softplus2(float t) {
  // H = S * Softplus_beta100(t / S)  (nn.Softplus(beta=100), fields.py:68) in base-2 units; raw v_exp_f32 / v_log_f32
#ifdef AVC_ABL_CHEAPACT   // timing ablation only (what the transcendentals cost)
  return fmaxf(t, 0.f);
#endif
#if AVC_SOFTPLUS_DIRECT
  // H = log2(1 + 2^t) as written -- v_exp, v_add, v_log -- plus ONE v_med3 that repairs the only range where that fails: 2^t
  // overflows for t >= 128, the logarithm returns +inf, and the median of (+inf, t, 128) is t (= H to fp32 precision there); below,
  // t <= L <= max(t, 0) + 1 <= 128 makes L the median.  2 plain + 2 transcendental instructions per element; the split form below
  // (max(t, 0) + log2(1 + 2^-|t|)) needs 3 + 2, and plain VALU issue is what these kernels are short of.
  // Very negative t: 2^t underflows to 0, L = 0 (true value 2^t log2 e < 2^-126).  Absolute error vs the split form <= 1 ulp of fp32.
  const float L = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(t));
  return __builtin_amdgcn_fmed3f(L, t, 128.f);
#else
  const float e = __builtin_amdgcn_exp2f(-fabsf(t));
  return relu_raw(t) + __builtin_amdgcn_logf(1.f + e);
#endif
}
// softplus2 of a whole accumulator tile.  AVC_SOFTPLUS_PK=1 (experiment): the "1 + 2^t" adds on register PAIRS
// (v_pk_add_f32) -- 8 instead of 16 plain adds per tile; MI355X_MICROARCH.md prices a packed-f32 op beside MFMAs at ~13 cycles MORE than
// the two scalar ops it replaces, which is what profiles/r05_ab_kernels.txt measures.
#ifndef AVC_SOFTPLUS_PK
#define AVC_SOFTPLUS_PK 0
#endif
typedef float f2v __attribute__((ext_vector_type(2)));
template <typename A>
__device__ __forceinline__ void softplus2_tile(const A& acc, float (&a)[16]) {
#if AVC_SOFTPLUS_PK && AVC_SOFTPLUS_DIRECT && !defined(AVC_ABL_CHEAPACT)
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    f2v e = {__builtin_amdgcn_exp2f(acc[r]), __builtin_amdgcn_exp2f(acc[r + 1])};
    const f2v one = {1.f, 1.f};
    e = e + one;
    asm("" : "+v"(e));   // keep the pair together: the add stays ONE packed instruction
    a[r] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(e[0]), acc[r], 128.f);
    a[r + 1] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(e[1]), acc[r + 1], 128.f);
  }
#else
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = softplus2(acc[r]);
#endif
}
// The same activation evaluated AFTER the f16 conversion the next layer's operand needs anyway (experiment, AVC_SDF_F16_ACT; VERDICT r3
// item 6): v_cvt_pk_f16_f32, v_exp_f16, v_pk_add_f16, v_log_f16 and the overflow repair as v_pk_max_f16 / v_pk_min_f16 -- H = min(log2(1 +
// 2^t), max(t, 16)): 2^t overflows f16 from t = 16 on, the logarithm then returns +inf and max(t, 16) = t is the minimum; below, L <= 16 <=
// max(t, 16).  2.0 plain VALU instructions per element instead of 2.5; the price is a second rounding (of t) in front of the operand's.
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
template <typename A>
__device__ __forceinline__ void softplus_frags_f16(const A& acc, h8& f0, h8& f1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { f0[j] = (_Float16)acc[j]; f1[j] = (_Float16)acc[8 + j]; }
  const h8 one = {1, 1, 1, 1, 1, 1, 1, 1}, cap = {16, 16, 16, 16, 16, 16, 16, 16};
  h8 l0 = __builtin_elementwise_log2(__builtin_elementwise_exp2(f0) + one);
  h8 l1 = __builtin_elementwise_log2(__builtin_elementwise_exp2(f1) + one);
  f0 = __builtin_elementwise_min(l0, __builtin_elementwise_max(f0, cap));
  f1 = __builtin_elementwise_min(l1, __builtin_elementwise_max(f1, cap));
  asm volatile("" : "+v"(f0), "+v"(f1));
}
#ifndef AVC_SDF_F16_ACT
#define AVC_SDF_F16_ACT 0
#endif
// sigma(beta a) recovered from H = S * softplus(a):  1 - 2^-H
__device__ __forceinline__ float sig_from_h(float H) { return 1.f - __builtin_amdgcn_exp2f(-H); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename V> __device__ __forceinline__ void set8(V& f, int j, float v) { f[j] = (typename MF<V>::S)v; }
template <typename V> __device__ __forceinline__ float get8(const V& f, int j) { return (float)f[j]; }

// Pin freshly produced fragments at this program point: LLVM otherwise SINKS the whole (pure) epilogue of a tile down to
// its first use in the next layer, keeping every fp32 accumulator of the layer alive (hundreds of spilled VGPRs).
template <typename V>
__device__ __forceinline__ void pin2(V& f0, V& f1) {
  asm volatile("" : "+v"(f0), "+v"(f1));
}
// accumulator tile (already activated, fp32) -> the two B-operand k-steps it feeds
template <typename V>
__device__ __forceinline__ void acc_to_frags(const float (&a)[16], V& f0, V& f1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { set8(f0, j, a[j]); set8(f1, j, a[8 + j]); }
  pin2(f0, f1);
}

// ReLU of an accumulator tile straight into its two f16 fragments: convert first (v_cvt_pk_f16_f32), then take the maximum with 0
// on the 16-bit patterns as SIGNED INTEGERS (v_pk_max_i16: a negative float, -0 included, is a negative integer; a positive one is
// itself) -- 8 + 8 packed instructions per tile instead of 16 v_med3 + 8 conversions; the same values (the conversion is monotonic
// and keeps the sign).  Returns the ReLU mask of the tile for the backward pass when WANT_BITS: bit p = element 2 p of the
// accumulator registers is > 0, bit 8 + p = element 2 p + 1 (p = 0..7) -- i.e. register r sits at bit relu_mask_bit(r) -- gathered
// with one v_pk_min_u16 (pattern -> 0 / 1 per half) and one v_lshl_or per register pair.  (A positive value below the smallest f16
// subnormal, 6e-8, counts as 0 here: its activation IS 0 in the operand the next layer multiplies.)
__host__ __device__ constexpr int relu_mask_bit(int r) { return (r >> 1) + 8 * (r & 1); }
template <bool WANT_BITS, typename A>
__device__ __forceinline__ unsigned relu_frags(const A& acc, h8& f0, h8& f1) {
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int j = 0; j < 8; ++j) { f0[j] = (_Float16)acc[j]; f1[j] = (_Float16)acc[8 + j]; }
  pin2(f0, f1);                                    // (the conversions as v_cvt_pk_f16_f32 pairs, like acc_to_frags)
  u4v w[2] = {__builtin_bit_cast(u4v, f0), __builtin_bit_cast(u4v, f1)};
  unsigned m = 0u;
  // (inline asm: hipcc rewrites the generic forms -- min(x, 1) becomes two 16-bit compares + selects, the packed maximum a
  // compare + select per half.  Its operands are conversion results, not MFMA results: no hazard the assembler has to see.)
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    unsigned x = w[p >> 2][p & 3];
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(x) : "v"(x));
    w[p >> 2][p & 3] = x;
    if (WANT_BITS) {
      unsigned q;
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(q) : "v"(x), "s"(0x00010001u));
      if (p == 0) m = q;
      else asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(m) : "v"(q), "n"(p), "v"(m));   // bit p: element 2 p, bit 16 + p: element 2 p + 1
    }
  }
  f0 = __builtin_bit_cast(h8, w[0]);
  f1 = __builtin_bit_cast(h8, w[1]);
  pin2(f0, f1);
  return WANT_BITS ? ((m & 0xffu) | (m >> 8)) : 0u;   // -> bits 0..7 | 8..15 (the callers store 16 bits)
}

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }

// Positional-encoding slots (host mirror: packing.pe_slot_table):
//   half 0: q 0..2 = x_c ; q 3+6k+c = sin(2^k x_c), 6+6k+c = cos(2^k x_c) for k=0..2 ; q 21..23 = x_lo_c
//   half 1: q 6k'+c = sin(2^(3+k') x_c), 6k'+3+c = cos(...)  for k'=0..2 ; q 18..23 = 0
// vals[q] = fp32 value, dcoef[q] = d vals[q] / d x_c (c = slot's coordinate), coord implicit (c = (q % 3) pattern).
struct PE {
  float v[24];   // feature value
  float d[24];   // derivative of the feature wrt its coordinate
};
__device__ __forceinline__ void pe_compute(const float (&x)[3], int h, PE& pe) {
  // static register indexing only (a lane-dependent index would send the arrays to scratch)
  const float f0 = h ? 8.f : 1.f;
  float sn[3][3], cs[3][3], fr[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    fr[k] = f0 * (float)(1 << k);
#pragma unroll
    for (int c = 0; c < 3; ++c) __sincosf(x[c] * fr[k], &sn[k][c], &cs[k][c]);
  }
#pragma unroll
  for (int q = 0; q < 24; ++q) {
    // half-0 view of slot q
    float v0 = 0.f, d0 = 0.f;
    if (q < 3) { v0 = x[q]; d0 = 1.f; }
    else if (q < 21) {
      const int k = (q - 3) / 6, rem = (q - 3) % 6;
      if (rem < 3) { v0 = sn[k][rem]; d0 = fr[k] * cs[k][rem]; }
      else { v0 = cs[k][rem - 3]; d0 = -fr[k] * sn[k][rem - 3]; }
    }
    // half-1 view
    float v1 = 0.f, d1 = 0.f;
    if (q < 18) {
      const int k = q / 6, rem = q % 6;
      if (rem < 3) { v1 = sn[k][rem]; d1 = fr[k] * cs[k][rem]; }
      else { v1 = cs[k][rem - 3]; d1 = -fr[k] * sn[k][rem - 3]; }
    }
    pe.v[q] = h ? v1 : v0;
    pe.d[q] = h ? d1 : d0;
  }
}
// PE values -> three f16 k-steps (with the hi/lo split of x: slot c holds fp16(x), slot 21+c the residual)
__device__ __forceinline__ void pe_to_frags_f16(const PE& pe, const float (&x)[3], int h, h8 (&f)[3]) {
#pragma unroll
  for (int q = 0; q < 24; ++q) f[q >> 3][q & 7] = (_Float16)pe.v[q];
  if (h == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      _Float16 hi = (_Float16)x[c];
      f[0][c] = hi;
      f[2][5 + c] = (_Float16)(x[c] - (float)hi);
    }
  }
}

static inline int avc_div_up(int a, int b) { return (a + b - 1) / b; }
// hipFuncSetAttribute applies to the CURRENT device only: the launchers raise their dynamic-LDS limit once per device
// (`seen` = the launcher's own bit set of device ordinals)
static inline bool avc_first_use_on_device(unsigned long long& seen) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;
  if ((seen >> d) & 1ull) return false;
  seen |= 1ull << d;
  return true;
}

// error plumbing for the C ABI
extern "C" const char* avc_last_error();
void avc_set_error(const char* msg);
int avc_check_launch(const char* what);
