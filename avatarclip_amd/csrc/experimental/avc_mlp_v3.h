// v3 engine: rolled tile loops + scratch-resident activations.
//
// Measured on the fully unrolled v2 kernels (SQ_WAIT_ANY 76 % of wave cycles, 250 spilled VGPRs, ds_read -> s_waitcnt ->
// mfma serialised because no register was free for operand prefetch, ~300 KB of straight-line code per kernel):
// the register file, not the matrix pipe, was the limiter.  v3 keeps in registers only the INPUT fragments of the
// layer being computed; every layer output goes to the wavefront's scratch slot (16 B per lane per k-step, L2/MALL
// resident because the slot is reused for every 32-point block) and is re-loaded as the next layer's input.  That
// removes the `out[]` register arrays, so the loop over output tiles can be a real loop (run-time tile index only
// addresses memory), the kernel shrinks ~8x, and there is room to (a) prefetch the LDS weight operands and (b) issue
// the epilogue's own loads (bias tables, sigma sources) one tile ahead, under the MFMAs.
#pragma once
#include "avc_mlp.h"

#ifdef AVC_X_NOSCR   // timing experiment only: scratch traffic removed (results are garbage)
template <typename V> __device__ __forceinline__ void scr_st(V* scr, int ks, const V& v) { asm volatile("" :: "v"(v)); }
template <typename V> __device__ __forceinline__ V scr_ld(const V* scr, int ks) { V v; for (int j = 0; j < 8; ++j) v[j] = (typename MF<V>::S)(0.01f * ks); return v; }
#else
template <typename V> __device__ __forceinline__ void scr_st(V* scr, int ks, const V& v) { scr[ks * 64] = v; }
template <typename V> __device__ __forceinline__ V scr_ld(const V* scr, int ks) { return scr[ks * 64]; }
#endif

template <typename V, int KS, class ST>
__device__ __forceinline__ facc tile_mma_rt(const ST& st, int j, const V (&in)[KS]) {
  const V* a = reinterpret_cast<const V*>(st.lds + st.par * ST::BUF_BYTES + j * (KS * 1024)) + st.lane;
  facc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  return mma_chain_lds<V, KS>(a, in, acc);
}

// Rolled, software-pipelined layer.  pre(t) issues the loads the epilogue of tile t will need (returned by value, kept in
// registers across one tile of MFMAs); epi(t, acc, pf) consumes them.  Iteration t: [group barrier + DMA of the next
// group] -> pre(t) -> MFMAs(t) -> epi(t-1).
template <typename V, int KS, int NT, class ST, typename Pre, typename Epi>
__device__ __forceinline__ void layer_r(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&in)[KS],
                                        Pre&& pre, Epi&& epi) {
  constexpr int G = ST::G;
  typedef decltype(pre(0)) PF;
  facc prev;
  PF pf_prev;
  const char* wbase = reinterpret_cast<const char*>(blob + (offw >> 3));
#pragma unroll 1
  for (int t = 0; t <= NT; ++t) {
    facc acc;
    PF pf;
    if (t < NT) {
      const int j = t % G;
      if (j == 0) {
        __syncthreads();   // this group has landed; the other buffer is free
        const int tn = t + G;
        if (tn < NT) {
          Next n;
          n.ptr = wbase + (long)tn * (KS * 1024);
          n.chunks = KS * ((NT - tn) < G ? (NT - tn) : G);
          stage_issue(st, n, st.par ^ 1);
        } else {
          stage_issue(st, after, st.par ^ 1);
        }
      }
      pf = pre(t);
      acc = tile_mma_rt<V, KS>(st, j, in);
      if (j == G - 1 || t == NT - 1) st.par ^= 1;
    }
    if (t > 0) epi(t - 1, prev, pf_prev);
    prev = acc;
    pf_prev = pf;
  }
}

struct PFNone {};
struct PF16 { float b[16]; };
struct PF32 { float b[16]; float w[16]; };
template <typename V> struct PF2 { V a, b; };
template <typename V, typename U> struct PF4 { V h0, h1; U q0, q1; };

#define AVC_PRE(...) [&](int t) __attribute__((always_inline)) { __VA_ARGS__ }
#define AVC_EPI3(PFT, ...) [&](int t, const facc& acc, const PFT& pf) __attribute__((always_inline)) { __VA_ARGS__ }

// cvt an accumulator tile (fp32 values a[16]) to the two k-step fragments it feeds
template <typename V>
__device__ __forceinline__ void frags_from(const float (&a)[16], V& f0, V& f1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { set8(f0, j, a[j]); set8(f1, j, a[8 + j]); }
}
