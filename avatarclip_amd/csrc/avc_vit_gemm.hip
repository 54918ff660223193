// Batched linears of the CLIP image encoder (hundreds of images per call: ShapeGen codebook search `ShapeGen/main.py:104-128`,
// pose retrieval `AvatarAnimate/models/pose_generation.py:79-110`): Y[M,N] = act(X W^T + b) (+ residual)
// as an LDS-staged bf16 GEMM on the matrix core.
//
// One 4-wavefront workgroup per (64 TM) x (64 TN) output block, every wavefront a TM x TN grid of 32 x 32 MFMA tiles.  Both operands
// arrive PRE-PACKED in fragment order ([tile][k-step][lane][8 bf16], vit_pack_x_kernel / pack_weight), so staging is a straight
// copy: global -> LDS DMA of whole 1-KiB fragments (global_load_lds_dwordx4), conflict-free ds_read_b128 on the other side, no
// swizzle.  NST ring slots of KB k-steps each are in flight; ONE barrier per slot orders both hazards (the slot to be
// refilled is the one everybody finished reading before the barrier); the fragments of k-step s + 1 are requested before the MFMAs
// of k-step s.
//
// Block shape, measured (B = 512 images, 12 layers, profiles/r03_score_bench.txt): these GEMMs write fp32 activations (4 bytes per
// 2 K FLOP: 190-380 FLOP/B, below the matrix core's ridge) and every workgroup of a launch reaches its store-only epilogue at the
// same time, so what pays is MORE, SMALLER workgroups per CU whose epilogues overlap the others' main loops -- not fewer LDS bytes
// per MFMA: 256 x 256 blocks (one workgroup per CU) 16.8 ms per encoder pass, 256 x 128 (two per CU) 14.9, 128 x 128 (three per CU)
// 12.7; ring depth / slot size within 3 %.  The direct-from-L1 kernel this replaces: 20.3; the same linears through hipBLASLt: 17.1.
#include "avc_common.h"
#include "../../include/avc.h"

#ifndef G2_TM
#define G2_TM 2          // 32-row MFMA tiles per wavefront (the workgroup is 2 x 2 wavefronts: 128 x 128 outputs)
#define G2_TN 2
#define G2_KB 2          // 16-deep k-steps per ring slot
#define G2_STAGES 3      // ring slots: 3 x 16 KiB
#define G2_OCC 3         // workgroups per CU the register budget is held to
#endif
// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  G2_XCD_REMAP = 1 renumbers them so that every XCD works on
// a CONTIGUOUS range of output blocks in row-major order: the column blocks of one row block of X then share one L2 instead of
// pulling that row block into all eight (12.8 -> 12.6 ms).
#ifndef G2_XCD_REMAP
#define G2_XCD_REMAP 1
#endif

typedef __attribute__((address_space(3))) char g2_lds_char;

template <int NWAIT>
__device__ __forceinline__ void g2_wait_vm() {
  static_assert(NWAIT >= 0 && NWAIT < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
}

template <int TM, int TN, int KB, int NST, int OCC, bool ACT, bool PRE, bool RES, bool PACK>
__global__ __launch_bounds__(256, OCC) void vit_gemm_lds_kernel(const b8* __restrict__ Xs, const b8* __restrict__ Wp,
                                                           const float* __restrict__ bias, const float* __restrict__ res,
                                                           float* __restrict__ Y, float* __restrict__ Ypre, b8* __restrict__ Ys, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int AT = 2 * TM, BT = 2 * TN, FR = AT + BT;         // row tiles, column tiles, fragments per k-step of the block
  constexpr int CHUNKS = FR * KB;                             // 1-KiB DMA chunks per ring slot
  constexpr int PW = CHUNKS / 4;                                 // ... per wavefront
  constexpr int SLOT = CHUNKS * 1024;
  static_assert(CHUNKS % 4 == 0, "chunks are dealt to 4 wavefronts");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int wm = wv & 1, wn = wv >> 1;
  const int KS = K >> 4, NIT = KS / KB;
#if G2_XCD_REMAP
  const int ncol = gridDim.x, total = gridDim.x * gridDim.y, L = blockIdx.x + ncol * blockIdx.y;
  const int xcd = L & 7, q8 = total >> 3, r8 = total & 7;
  const int T = xcd * q8 + (xcd < r8 ? xcd : r8) + (L >> 3);
  const long mt_base = (long)(T / ncol) * AT, nt_base = (long)(T % ncol) * BT;
#else
  const long mt_base = (long)blockIdx.y * AT, nt_base = (long)blockIdx.x * BT;
#endif

  // chunk c of a slot: fragment (tile = c / KB, k-step = c % KB); tiles 0 .. AT-1 are rows of X, the rest columns of W
  auto issue = [&](int it, int slot) {
#pragma unroll
    for (int q = 0; q < PW; ++q) {
      const int c = wv + 4 * q;
      const int tile = c / KB, ks = c % KB;
      const b8* src = (tile < AT ? Xs + ((mt_base + tile) * KS + (long)it * KB + ks) * 64
                                 : Wp + ((nt_base + tile - AT) * KS + (long)it * KB + ks) * 64) + lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + slot * SLOT + c * 1024), 16, 0, 0);
    }
  };
  facc acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int d = 0; d < NST - 1; ++d)
    if (d < NIT) issue(d, d);
  int slot = 0;
  for (int it = 0; it < NIT; ++it) {
    // my chunks of slot `it` have landed when at most the chunks of the younger slots are outstanding
    if (it + NST - 1 <= NIT) g2_wait_vm<(NST - 2) * PW>(); else g2_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (it + NST - 1 < NIT) { int ns = slot + NST - 1; if (ns >= NST) ns -= NST; issue(it + NST - 1, ns); }
    const g2_lds_char* base = (const g2_lds_char*)(lds + slot * SLOT) + lane * 16;
    // the fragments of k-step ks + 1 are requested BEFORE the MFMAs of k-step ks (hipcc would sink every read next to its first use:
    // one LDS round trip per MFMA row)
    b8 a[2][TM], b[2][TN];
    auto fetch = [&](int ks, int p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[p][i] = *reinterpret_cast<const __attribute__((address_space(3))) b8*>(base + ((wm * TM + i) * KB + ks) * 1024);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[p][j] = *reinterpret_cast<const __attribute__((address_space(3))) b8*>(base + ((AT + wn * TN + j) * KB + ks) * 1024);
    };
    fetch(0, 0);
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) {
      if (ks + 1 < KB) fetch(ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MF<b8>::mma(a[ks & 1][i], b[ks & 1][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++slot == NST) slot = 0;
  }
  if (PACK) {
    // the output as the NEXT linear's packed bf16 operand ([row tile][k-step][lane (row, half)][8 columns]) instead of fp32 rows:
    // every wavefront turns its tiles round through its own 2-KiB corner of the (now idle) ring -- accumulator layout (lane =
    // column, registers = rows) in, 16 bytes of one row out -- and stores whole 1-KiB fragments
    __syncthreads();                                     // everybody is done reading the ring
    __bf16* T = reinterpret_cast<__bf16*>(lds) + wv * (32 * 40);        // [32 rows][32 columns], row stride 40 (80 B: 16-B aligned, skewed)
    const int KSo = N >> 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long nt = nt_base + wn * TN + j;
      const float bv = bias ? bias[32 * (int)nt + n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const long mt = mt_base + wm * TM + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bv;
          if (ACT) v = v * sigmoidf_(1.702f * v);
          T[((r & 3) + 8 * (r >> 2) + 4 * h) * 40 + n] = (__bf16)v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < 2; ++e) {                    // the two k-steps of the next layer this 32-column tile covers
          const b8 f = *reinterpret_cast<const b8*>(&T[n * 40 + 16 * e + 8 * h]);
          Ys[((mt * KSo) + 2 * nt + e) * 64 + lane] = f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = 32 * (int)(nt_base + wn * TN + j) + n;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = 32 * (int)(mt_base + wm * TM + i) + 4 * h;
      if (row0 - 4 * h >= M) continue;              // a row tile of padding (wave-uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        if (row < M) {
          float v = acc[i][j][r] + bv;
          const long o = (long)row * N + col;
          if (ACT) {
            if (PRE) Ypre[o] = v;
            v = v * sigmoidf_(1.702f * v);
          }
          if (RES) v += res[o];
          Y[o] = v;
        }
      }
    }
  }
}

template <int TM, int TN, int KB, int NST, int OCC>
static void g2_launch(const b8* xs, const b8* wp, const float* bias, const float* res, float* y, float* y_pre, b8* ys, int M, int N, int K,
                      int act, int mt_packed, hipStream_t s) {
  constexpr int lds = (2 * TM + 2 * TN) * KB * 1024 * NST;
  static_assert(lds >= 4 * 32 * 40 * 2, "the pack-out epilogue borrows 2.5 KiB per wavefront");
  static unsigned long long attr_seen = 0;
#define G2_K(A, P, R, Q) vit_gemm_lds_kernel<TM, TN, KB, NST, OCC, A, P, R, Q>
  if (avc_first_use_on_device(attr_seen)) {
    (void)hipFuncSetAttribute((const void*)G2_K(false, false, false, false), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)G2_K(false, false, true, false), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)G2_K(true, false, false, false), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)G2_K(true, true, false, false), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)G2_K(true, false, false, true), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)G2_K(false, false, false, true), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  const dim3 grid(N / (64 * TN), mt_packed / (2 * TM)), block(256);
#define G2_GO(A, P, R, Q) hipLaunchKernelGGL((G2_K(A, P, R, Q)), grid, block, lds, s, xs, wp, bias, res, y, y_pre, ys, M, N, K)
  if (ys && act) G2_GO(true, false, false, true);
  else if (ys) G2_GO(false, false, false, true);
  else if (act && y_pre) G2_GO(true, true, false, false);
  else if (act) G2_GO(true, false, false, false);
  else if (res) G2_GO(false, false, true, false);
  else G2_GO(false, false, false, false);
#undef G2_GO
#undef G2_K
}
// the caller (avc_vit.hip) has packed whole groups of 4 row tiles; false = shape not covered (N, K not multiples of the block)
// ys != NULL: the output goes out as the packed bf16 operand of the next linear (no fp32 y, no residual, no y_pre)
bool avc_vit_gemm_lds(const void* xs, const void* wp, const float* bias, const float* res, float* y, float* y_pre, void* ys, int M, int N,
                      int K, int act, int mt_packed, void* stream) {
  if ((K % (16 * G2_KB)) || (mt_packed % (2 * G2_TM)) || (N % (64 * G2_TN)) || (act && res) || (ys && (res || y_pre))) return false;
  g2_launch<G2_TM, G2_TN, G2_KB, G2_STAGES, G2_OCC>((const b8*)xs, (const b8*)wp, bias, res, y, y_pre, (b8*)ys, M, N, K, act, mt_packed,
                                                    (hipStream_t)stream);
  return true;
}
