// Building blocks of the SDF + colour MLP kernels: one wavefront owns 32 points, every layer is a staged sequence of
// 32-row weight tiles (avc_stage.h).  See avc_common.h for the fragment layout.
// Follows AvatarGen/AppearanceGen/models/fields.py:72-107 (SDFNetwork.forward/.gradient) and :154-185
// (RenderingNetwork.forward, mode 'no_view_dir', extra_color) of the reference.
#pragma once
#include "avc_common.h"
#include "avc_stage.h"

template <int H_, int NMID_, int NCMID_>
struct NetT {
  static constexpr int H = H_;          // hidden width: 256 (confs/examples) or 128 (confs/examples_small)
  static constexpr int NMID = NMID_;    // HxH middle SDF layers: 2 | 1
  static constexpr int NCMID = NCMID_;  // HxH middle colour layers: 1 | 0
  static constexpr int HT = H / 32;     // 32-row output tiles of an H-wide layer
  static constexpr int HK = H / 16;     // 16-deep k-steps of an H-wide input
  static constexpr int SKIP = H - 39;   // width of the layer feeding the skip concat (fields.py:36-39)
  static constexpr int ST = (SKIP + 31) / 32;
  static constexpr int SK = 2 * ST;
};
typedef NetT<256, 2, 1> NetFull;
typedef NetT<128, 1, 0> NetSmall;
#include "avc_offsets_gen.h"
// the launchers verify the table that came through the C ABI against the compiled-in one
template <class N> static inline bool offsets_match(const int* offs) {
  for (int k = 0; k < OFF_COUNT; ++k)
    if (offs[k] != Off<N>::value.v[k]) return false;
  return true;
}

// Static issue priority per wavefront (s_setprio once, before the main loop).  The wavefronts w, w + 4 (, w + 8) of a workgroup share a
// SIMD and run the same instruction sequence between the same barriers; between equals the arbiter prefers the OLDER wave, i.e. the
// later-dispatched ones lose every contested VALU slot.  AVC_WAVE_PRIO: 0 = leave it to age, 1 = the younger half of the workgroup
// at priority 1, 2 = priority w / 4, 3 = the older half at priority 1.  (Measured: profiles/r03_ab_kernels.txt.)
#ifndef AVC_WAVE_PRIO
#define AVC_WAVE_PRIO 0
#endif
__device__ __forceinline__ void avc_static_wave_priority() {
#if AVC_WAVE_PRIO == 1
  if ((threadIdx.x >> 6) >= (blockDim.x >> 7)) __builtin_amdgcn_s_setprio(1);
#elif AVC_WAVE_PRIO == 2
  const int q = threadIdx.x >> 8;
  if (q == 1) __builtin_amdgcn_s_setprio(1);
  if (q >= 2) __builtin_amdgcn_s_setprio(2);
#elif AVC_WAVE_PRIO == 3
  if ((threadIdx.x >> 6) < (blockDim.x >> 7)) __builtin_amdgcn_s_setprio(1);
#endif
}

struct PointSrc {
  const float* pts;      // [N,3] or nullptr -> ray mode
  const float* rays_o;   // [R,3]
  const float* rays_d;   // [R,3]
  const float* z;        // [R,ldz]
  int S;                 // samples per ray in this launch
  int ldz;
  int midpoint;          // 1: evaluate at section mid-points z + dist/2 (renderer.py:210-215)
  float sample_dist;
};

__device__ __forceinline__ void fetch_point(const PointSrc& ps, long i, float (&x)[3]) {
  if (ps.pts) {
    x[0] = ps.pts[3 * i]; x[1] = ps.pts[3 * i + 1]; x[2] = ps.pts[3 * i + 2];
    return;
  }
  const long ray = i / ps.S;
  const int s = (int)(i - ray * ps.S);
  const float* zr = ps.z + ray * ps.ldz;
  float t = zr[s];
  if (ps.midpoint) {
    const float dist = (s + 1 < ps.S) ? (zr[s + 1] - t) : ps.sample_dist;
    t = t + dist * 0.5f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) x[c] = ps.rays_o[3 * ray + c] + ps.rays_d[3 * ray + c] * t;
}

template <typename V, int KS>
__device__ __forceinline__ const V* tptr(const V* blob, int off, int t, int lane) {
  return blob + (off >> 3) + (long)(t * KS) * 64 + lane;
}


// ---- tile geometry of every packed weight (k-steps per tile, tiles) ----
template <class N, int OFF> struct TileInfo;
#define AVC_TI(OFF, KS_, NT_) template <class N> struct TileInfo<N, OFF> { static constexpr int KS = KS_, NT = NT_; };
AVC_TI(OFF_W0, 3, N::HT)          AVC_TI(OFF_WM0, N::HK, N::HT)   AVC_TI(OFF_WM1, N::HK, N::HT)  AVC_TI(OFF_WS, N::HK, N::ST)
AVC_TI(OFF_WL, N::SK + 3, N::HT)  AVC_TI(OFF_W0T, N::HK, 2)       AVC_TI(OFF_WM0T, N::HK, N::HT) AVC_TI(OFF_WM1T, N::HK, N::HT)
AVC_TI(OFF_WST, N::SK, N::HT)     AVC_TI(OFF_WLT, N::HK, N::ST)   AVC_TI(OFF_C0, N::HK + 1, N::HT) AVC_TI(OFF_CM0, N::HK, N::HT)
AVC_TI(OFF_CH, N::HK, 1)          AVC_TI(OFF_C0T, N::HK, N::HT + 1) AVC_TI(OFF_CM0T, N::HK, N::HT) AVC_TI(OFF_CHT, 1, N::HT)
AVC_TI(OFF_W0G, 3, N::HT)

// descriptor of the FIRST group of packed weight OFF (what the layer before it prefetches)
template <class N, int OFF, class ST>
__device__ __forceinline__ Next nxt(const ST&, const void* blob, const AvcOffsets& o) {
  typedef TileInfo<N, OFF> TI;
  Next n;
  n.ptr = reinterpret_cast<const char*>(blob) + (long)o.v[OFF] * 2;
  constexpr int G = ST::template group<TI::KS>();
  n.chunks = TI::KS * (TI::NT < G ? TI::NT : G);
  return n;
}
__device__ __forceinline__ Next no_next() { Next n; n.ptr = nullptr; n.chunks = 0; return n; }

// ---- staged, software-pipelined layer: per group one barrier + the DMA of the next group; inside a group tile t's MFMAs
// ---- are issued before the epilogue of tile t-1, so the VALU / transcendental / store work of one tile hides under the
// ---- matrix pipe of the next (same basic block, no barrier between).
// The two wavefronts of a SIMD (waves w and w+4 of the workgroup) leave every group barrier together and would run their MFMA
// chains at the same time and their VALU epilogues at the same time (matrix pipe idle during the epilogues).  Holding the
// second wave back by about half a tile step lets one wave's epilogue run under the other's MFMA chain.
#ifdef AVC_ABL_NOSYNC   // timing ablation only
#define AVC_SYNC() do {} while (0)
#elif defined(AVC_EXP_SYNC)   // timing experiment only (results are garbage): group barrier that leaves AVC_EXP_SYNC VMEM ops in flight
#define AVC_STR2(x) #x
#define AVC_STR(x) AVC_STR2(x)
#define AVC_SYNC() do { asm volatile("s_waitcnt vmcnt(" AVC_STR(AVC_EXP_SYNC) ") lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#else
#define AVC_SYNC() __syncthreads()
#endif
#ifndef AVC_DEPHASE
#define AVC_DEPHASE 0   // s_sleep units (64 cycles); 0 = off
#endif
template <class ST>
__device__ __forceinline__ void dephase(const ST& st) {
  if (AVC_DEPHASE > 0) {
    if (st.wave >= (st.nw >> 1)) __builtin_amdgcn_s_sleep(AVC_DEPHASE);
  }
}
#define AVC_EPI(...) [&](int t, const facc& acc) __attribute__((always_inline)) { __VA_ARGS__ }
// Optional hook run once per layer right AFTER the barrier of its first weight group: the place for streaming stores of the
// previous layer's output tiles (still live as this layer's input).  hipcc drains vmcnt(0) before every group barrier while an
// LDS-DMA is pending, and stores count on vmcnt: a store issued in an epilogue right before a barrier stalls the whole
// workgroup for an HBM write round trip; issued right after one it has a group of MFMA work (~2 us) to complete under.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
#define AVC_HOOK(...) [&]() __attribute__((always_inline)) { __VA_ARGS__ }
// a hook that spreads its stores over the layer's weight groups: called after EVERY group barrier with (g, number of groups) and
// stores the g-th share of its tiles (tiles_store_part) -- bursts of 4 instead of 8 tile stores per barrier interval
#define AVC_HOOKG(...) [&](int grp_, int ngrp_) __attribute__((always_inline)) { __VA_ARGS__ }
// AVC_STORE_PER_TILE=1 (round 5): a spreading hook is called once per output-TILE step with (t, NT) instead of once per weight group
// with (g, NG): its tile stores go out one tile (2 KiB per wavefront, 16 KiB per workgroup) at a time, each under ~500 cycles of
// MFMA work, instead of in bursts of 4 tiles behind a barrier (64 KiB per workgroup against a store path of 64 B/clk: the waves then sit
// on VMEM back-pressure).  One-shot hooks (AVC_HOOK) still run once, after the first barrier.
#ifndef AVC_STORE_PER_TILE
#define AVC_STORE_PER_TILE 1   // (profiles/r05_ab_kernels.txt: training forward 7.51 -> 7.45 ms, plain forward 5.88 -> 5.82 ms per 4 Mi points)
#endif
template <typename Hook>
__device__ __forceinline__ void hook_call(Hook&& hook, int g, int ng) {
  if constexpr (std::is_invocable_v<Hook, int, int>) { if (!AVC_STORE_PER_TILE) hook(g, ng); }
  else if (g == 0) hook();
}
template <typename Hook>
__device__ __forceinline__ void hook_tile(Hook&& hook, int t, int nt) {
  if constexpr (std::is_invocable_v<Hook, int, int>) { if (AVC_STORE_PER_TILE) hook(t, nt); }
}

#ifndef AVC_PAIR
#define AVC_PAIR 0   // 1: two output tiles per MFMA stream (independent accumulators), 0: one dependent chain per tile
#endif
#ifndef AVC_PAIR_SQ
#define AVC_PAIR_SQ AVC_PAIR   // the same for the layers whose epilogues read panel tiles back (layer_sq: normal sweep of the forward)
#endif
#ifndef AVC_PAIR_L2
#define AVC_PAIR_L2 AVC_PAIR   // ... and for the layers with a split K dimension (layer2_s: feature layer, first colour layer)
#endif
// PAIRED = two output tiles per MFMA stream.  Measured per 4 Mi points (profiles/r03_ab_kernels.txt): forward 6.16 -> 6.01 ms and
// training forward 8.25 -> 7.84 ms with it (and 26 / 36 spilled registers become 0 / 12); the SDF-only kernel loses 6 % (its 168
// registers leave no room for the second accumulator: 16 -> 28 spills) and the backward kernel 3 %: those keep one chain.
template <bool PAIRED, typename V, int KS, int NT, class ST, typename Epi, typename Hook = NoHook, class Bias = NoBias>
__device__ __forceinline__ void layer_sp(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&in)[KS],
                                         Epi&& epi, Hook&& hook = NoHook{}, const Bias& bias = NoBias{}) {
  constexpr int G = ST::template group<KS>();
  constexpr int NG = (NT + G - 1) / G;
  facc prev0, prev1;
  int tp = -1, np = 0;   // first tile / number of tiles whose epilogue is pending (compile-time after unrolling)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    AVC_SYNC();   // group g has landed (hipcc drains vmcnt before the barrier); the other buffer is free
    if (g + 1 < NG) {
      Next n;
      n.ptr = blob + (offw >> 3) + (long)((g + 1) * G * KS) * 64;
      n.chunks = KS * ((NT - (g + 1) * G) < G ? (NT - (g + 1) * G) : G);
      stage_issue(st, n, st.par ^ 1);
    } else {
      stage_issue(st, after, st.par ^ 1);
    }
    hook_call(hook, g, NG);
    dephase(st);
#pragma unroll
    for (int j = 0; j < G; j += (PAIRED ? 2 : 1)) {
      const int t = g * G + j;
      if (t < NT) {
        const bool two = PAIRED && (j + 1 < G) && (t + 1 < NT);
        hook_tile(hook, t, NT);
        if (two) hook_tile(hook, t + 1, NT);
        facc a0, a1;
        if (two) tile_mma_pair<V, KS>(st, j, in, a0, a1, bias, t);
        else a0 = tile_mma<V, KS>(st, j, in, bias, t);
        if (np > 0) {
          epi(tp, prev0);
          if (np > 1) epi(tp + 1, prev1);
          if (two) { interleave_mfma_valu<KS>(); interleave_mfma_valu<KS>(); } else interleave_mfma_valu<KS>();
        }
        prev0 = a0;
        if (two) prev1 = a1;
        tp = t; np = two ? 2 : 1;
        __builtin_amdgcn_sched_barrier(0);   // keep epilogues from being sunk past later tiles
      }
    }
    st.par ^= 1;
  }
  epi(tp, prev0);
  if (np > 1) epi(tp + 1, prev1);
  __builtin_amdgcn_sched_barrier(0);
}
template <typename V, int KS, int NT, class ST, typename Epi, typename Hook = NoHook, class Bias = NoBias>
__device__ __forceinline__ void layer_s(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&in)[KS],
                                        Epi&& epi, Hook&& hook = NoHook{}, const Bias& bias = NoBias{}) {
  layer_sp<(AVC_PAIR != 0), V, KS, NT>(st, blob, offw, after, in, epi, hook, bias);
}
template <typename V, int KS, int NT, class ST, typename Epi, typename Hook = NoHook, class Bias = NoBias>
__device__ __forceinline__ void layer_s1(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&in)[KS],
                                         Epi&& epi, Hook&& hook = NoHook{}, const Bias& bias = NoBias{}) {
  layer_sp<false, V, KS, NT>(st, blob, offw, after, in, epi, hook, bias);   // always one accumulator chain
}
// ---- the same layer with the epilogue's global loads issued BEFORE the MFMA chain they trail: pre(t) returns the raw loads
// ---- (panel tiles read back by the backward sweeps), epi(t, acc, data) consumes them after the chain of tile t+1.  The
// ---- backward kernel keeps only ~16 KiB of reads in flight per CU when every epilogue loads and immediately waits; one
// ---- chain (~600 cycles) of head start costs no extra live set beyond the loaded registers themselves.
#ifndef AVC_DEEP_PF1
#define AVC_DEEP_PF1 0
#endif
#define AVC_PRE(...) [&](int t) __attribute__((always_inline)) { __VA_ARGS__ }
#define AVC_EPID(DT, ...) [&](int t, const facc& acc, const DT& d) __attribute__((always_inline)) { __VA_ARGS__ }
// DEEP = false: pre(t-1) goes out before the chain of tile t (one chain of head start, one load set live);
// DEEP = true:  pre(t) goes out before the chain of tile t (two chains + one epilogue of head start, two sets live) --
//               measured slower for the three-array loads of the reverse sweep (register pressure), see DESIGN.md 5.
template <bool AVC_PRE_DEEP, bool PAIRED, bool DUAL, typename V, int KS, int NT, class ST, typename Pre, typename Epi, typename Hook>
__device__ __forceinline__ void layer_sq_(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&in)[KS],
                                         Pre&& pre, Epi&& epi, Hook&& hook) {
  constexpr int G = ST::template group<KS>();
  constexpr int NG = (NT + G - 1) / G;
  if constexpr (PAIRED && !AVC_PRE_DEEP) {
    // two output tiles per MFMA stream: the loads of the previous pair's epilogues go out before the chains of the current pair
    facc prev0, prev1;
    decltype(pre(0)) d0{}, d1{};
    int tp = -1, np = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      AVC_SYNC();
      if (g + 1 < NG) {
        Next n;
        n.ptr = blob + (offw >> 3) + (long)((g + 1) * G * KS) * 64;
        n.chunks = KS * ((NT - (g + 1) * G) < G ? (NT - (g + 1) * G) : G);
        stage_issue(st, n, st.par ^ 1);
      } else {
        stage_issue(st, after, st.par ^ 1);
      }
      hook_call(hook, g, NG);
#pragma unroll
      for (int j = 0; j < G; j += 2) {
        const int t = g * G + j;
        if (t < NT) {
          const bool two = (j + 1 < G) && (t + 1 < NT);
          hook_tile(hook, t, NT);
          if (two) hook_tile(hook, t + 1, NT);
          if (np > 0) {
            d0 = pre(tp);
            if (np > 1) d1 = pre(tp + 1);
            __builtin_amdgcn_sched_barrier(0);   // the loads go out before the chains
          }
          facc a0, a1;
          if (two) tile_mma_pair<V, KS>(st, j, in, a0, a1, NoBias{}, t);
          else a0 = tile_mma<V, KS>(st, j, in);
          if (np > 0) {
            epi(tp, prev0, d0);
            if (np > 1) epi(tp + 1, prev1, d1);
            if (two) { interleave_mfma_valu<KS>(); interleave_mfma_valu<KS>(); } else interleave_mfma_valu<KS>();
          }
          prev0 = a0;
          if (two) prev1 = a1;
          tp = t; np = two ? 2 : 1;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      st.par ^= 1;
    }
    d0 = pre(tp);
    if (np > 1) d1 = pre(tp + 1);
    epi(tp, prev0, d0);
    if (np > 1) epi(tp + 1, prev1, d1);
    __builtin_amdgcn_sched_barrier(0);
    return;
  }
  facc prev, prev2;
  decltype(pre(0)) dprev{}, dcur{};
  // DUAL (timing ablation AVC_ABL_BWD_RECOMP): every tile runs a second chain on the same A fragments and its epilogue pays 16 softplus
  auto fold2 = [&](const facc& a, const facc& a2) __attribute__((always_inline)) {
    facc r = a;
    if constexpr (DUAL) {
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] += 1e-38f * softplus2(a2[q]);
    }
    return r;
  };
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    AVC_SYNC();
    if (g + 1 < NG) {
      Next n;
      n.ptr = blob + (offw >> 3) + (long)((g + 1) * G * KS) * 64;
      n.chunks = KS * ((NT - (g + 1) * G) < G ? (NT - (g + 1) * G) : G);
      stage_issue(st, n, st.par ^ 1);
    } else {
      stage_issue(st, after, st.par ^ 1);
    }
    hook_call(hook, g, NG);
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int t = g * G + j;
      if (t < NT) {
        hook_tile(hook, t, NT);
        if (AVC_PRE_DEEP) {
          dcur = pre(t);
          __builtin_amdgcn_sched_barrier(0);   // the loads go out before the chain
        } else if (t > 0) {
          dprev = pre(t - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        facc acc, acc2;
        if constexpr (DUAL) {
          const V* a = reinterpret_cast<const V*>(st.lds + st.par * ST::BUF_BYTES + j * KS * 1024) + st.lane;
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 1.f; }
          asm volatile("" : "+v"(acc2));
          acc = mma_chain_lds_dual<V, KS>(a, in, acc, acc2);
        } else {
          acc = tile_mma<V, KS>(st, j, in);
        }
        if (t > 0) { epi(t - 1, fold2(prev, prev2), dprev); interleave_mfma_valu<KS>(); }
        prev = acc;
        if constexpr (DUAL) prev2 = acc2;
        if (AVC_PRE_DEEP) dprev = dcur;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    st.par ^= 1;
  }
  if (!AVC_PRE_DEEP) dprev = pre(NT - 1);
  epi(NT - 1, fold2(prev, prev2), dprev);
  __builtin_amdgcn_sched_barrier(0);
}
template <typename V, int KS, int NT, class ST, typename Pre, typename Epi, typename Hook = NoHook>
__device__ __forceinline__ void layer_sq(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&in)[KS],
                                         Pre&& pre, Epi&& epi, Hook&& hook = NoHook{}) {
  layer_sq_<false, (AVC_PAIR_SQ != 0), false, V, KS, NT>(st, blob, offw, after, in, pre, epi, hook);
}
template <typename V, int KS, int NT, class ST, typename Pre, typename Epi, typename Hook = NoHook>
__device__ __forceinline__ void layer_sqd(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&in)[KS],
                                          Pre&& pre, Epi&& epi, Hook&& hook = NoHook{}) {
#ifdef AVC_ABL_BWD_RECOMP
  layer_sq_<AVC_DEEP_PF1 != 0, false, true, V, KS, NT>(st, blob, offw, after, in, pre, epi, hook);
#else
  layer_sq_<AVC_DEEP_PF1 != 0, (AVC_PAIR_SQ != 0), false, V, KS, NT>(st, blob, offw, after, in, pre, epi, hook);
#endif
}
template <typename V, int KA, int KB, int NT, class ST, typename Epi, typename Hook = NoHook, class Bias = NoBias>
__device__ __forceinline__ void layer2_s(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&ina)[KA],
                                         const V (&inb)[KB], Epi&& epi, Hook&& hook = NoHook{}, const Bias& bias = NoBias{}) {
  constexpr int KS = KA + KB;
  constexpr int G = ST::template group<KS>();
  constexpr int NG = (NT + G - 1) / G;
  constexpr bool PAIRED = AVC_PAIR_L2 != 0;
  facc prev0, prev1;
  int tp = -1, np = 0;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    AVC_SYNC();
    if (g + 1 < NG) {
      Next n;
      n.ptr = blob + (offw >> 3) + (long)((g + 1) * G * KS) * 64;
      n.chunks = KS * ((NT - (g + 1) * G) < G ? (NT - (g + 1) * G) : G);
      stage_issue(st, n, st.par ^ 1);
    } else {
      stage_issue(st, after, st.par ^ 1);
    }
    hook_call(hook, g, NG);
    dephase(st);
#pragma unroll
    for (int j = 0; j < G; j += (PAIRED ? 2 : 1)) {
      const int t = g * G + j;
      if (t < NT) {
        const bool two = PAIRED && (j + 1 < G) && (t + 1 < NT);
        hook_tile(hook, t, NT);
        if (two) hook_tile(hook, t + 1, NT);
        facc a0, a1;
        if (two) tile_mma2_pair<V, KA, KB>(st, j, ina, inb, a0, a1, bias, t);
        else a0 = tile_mma2<V, KA, KB>(st, j, ina, inb, bias, t);
        if (np > 0) {
          epi(tp, prev0);
          if (np > 1) epi(tp + 1, prev1);
          if (two) { interleave_mfma_valu<KS>(); interleave_mfma_valu<KS>(); } else interleave_mfma_valu<KS>();
        }
        prev0 = a0;
        if (two) prev1 = a1;
        tp = t; np = two ? 2 : 1;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    st.par ^= 1;
  }
  epi(tp, prev0);
  if (np > 1) epi(tp + 1, prev1);
  __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------------
// Operand panels of the training path (mirrored by packing.py: build_layout).  Every operand of every weight-gradient
// product is stored ONCE, in the layout the producing wavefront holds it in: per 32-point block and 32-feature tile the two
// B-operand fragments [k-step e][lane][8 x 16 bit] (lane = point + 32 * half), 2 KiB per tile.  Two REGIONS with their own
// block stride:
//   F region (P_* indices, P_TILES tiles per block): forward-type operands (h, g_a, feature, r, [x,n], PE), f16, written by
//            the forward kernel for every block of the ray set and kept until the backward pass;
//   G region (G_* indices, G_TILES tiles per block): gradient-type operands (gbar_h, abar, delta, ybar), bf16, written by the
//            backward kernel.  The backward pass walks the ray set in SLABS (backward kernel + weight-gradient kernel per
//            slab), so the G region holds one slab and is reused: the gradient-type half of the operands never exists for
//            more than a slab at a time (512^2 x 64 spp: 87 GiB of F panels + 23 GiB of G panels instead of 177 GiB).
// The sweeps of both kernels read tiles back as they are (no transposition), the weight-gradient kernel transposes them with
// the LDS transpose read when it loads them (avc_wgrad.hip).
// ---------------------------------------------------------------------------------------------------------------
template <class N>
struct PanelLayout {
  static constexpr int HT = N::HT, ST = N::ST, NM = N::NMID, NC = N::NCMID;
  // ---- F region.  (hs, pe) are adjacent: the last layer's input is [hs | pe], so its weight-gradient product reads them as
  // ONE run of ST + 2 tiles; (feature, [x,n]) likewise for the first colour layer
  static constexpr int P_H1 = 0;                    // h1
  static constexpr int P_HM = P_H1 + HT;            // hm[NM]
  static constexpr int P_HS = P_HM + NM * HT;       // hs (ST)
  static constexpr int P_H0 = P_HS + ST;            // pe values (2 tiles)
  static constexpr int P_GA1 = P_H0 + 2;            // g_a1
  static constexpr int P_GAM = P_GA1 + HT;          // g_am[NM]
  static constexpr int P_GAS = P_GAM + NM * HT;     // g_as (ST)
  static constexpr int P_FEAT = P_GAS + ST;         // feature (HT)
  static constexpr int P_XN = P_FEAT + HT;          // [x, n] (1)
  static constexpr int P_R1 = P_XN + 1;             // r1 (HT)
  static constexpr int P_R2 = P_R1 + HT;            // r2 (HT, only NC==1)
  static constexpr int P_TILES = P_R2 + NC * HT;    // tiles per block of the F region
  // ---- G region.  (ybar[1:], d_sdf) are adjacent for the same reason.  gbar_hs has NO panel: its only consumers are the first layer of
  // the reverse sweep (which takes it from the registers of the second-order sweep) and the column sum over the points that is the
  // second-order term of row 0 of the last layer (reduced in the backward kernel: col_sums in csrc/avc_bwd_body.h) -- rounds 2-4 wrote it,
  // and a constant-one tile, for a "1 (x) [gbar_hs | gbar_h0]" product in the weight-gradient kernel: 8 tiles written + 10 read per block
  static constexpr int G_GBH1 = 0;                  // gbar_h1
  static constexpr int G_GBHM = G_GBH1 + HT;        // gbar_hm[NM]
  static constexpr int G_GB0 = G_GBHM + NM * HT;    // gbar_h0 (2)
  static constexpr int G_AB1 = G_GB0 + 2;           // abar_1
  static constexpr int G_ABM = G_AB1 + HT;          // abar_m[NM]
  static constexpr int G_ABS = G_ABM + NM * HT;     // abar_s (ST)
  static constexpr int G_DFEAT = G_ABS + ST;        // ybar[1:] (HT)
  static constexpr int G_SDF = G_DFEAT + HT;        // feature 0 = d_sdf (1)
  static constexpr int G_D1 = G_SDF + 1;            // delta1 (HT)
  static constexpr int G_D2 = G_D1 + HT;            // delta2 (HT, only NC==1)
  static constexpr int G_DO = G_D2 + NC * HT;       // delta_o (1)
  static constexpr int G_TILES = G_DO + 1;          // tiles per block of the G region
  static constexpr int MASK_U16 = 2 * HT * 64;      // ReLU masks of r1 / r2 per 32-point block: [layer][tile][lane] x 16 bits
  static constexpr int CS_FLOATS = ST * 32 + 48;    // column sums per wavefront: [ST][2 halves][16 acc registers] of gbar_hs, [3 frags][2][8 slots] of gbar_h0
};
// the no-grad forward parks only what its own normal sweep reads back (h1, hm, feature), in a per-wavefront slot it reuses
template <class N>
struct ScratchLayout {
  static constexpr int HT = N::HT, NM = N::NMID;
  static constexpr int P_H1 = 0, P_HM = HT, P_FEAT = HT + NM * HT, P_TILES = P_FEAT + HT;
  static constexpr int P_H0 = 0, P_HS = 0, P_GA1 = 0, P_GAM = 0, P_GAS = 0, P_XN = 0, P_R1 = 0, P_R2 = 0;   // never written
};

// Panel addressing: `ub` = wave-uniform base of the block's panels (SGPR pair), `off` = lane * 16 (ONE VGPR shared by every
// access); the tile base is formed on the scalar unit and kept opaque, so that hipcc emits the saddr form
// (global_store v_off, v_data, s[base]) instead of one 64-bit VGPR address pair per tile (those pairs were being
// computed early and spilled: +130 spilled registers in the training forward kernel).
struct PanelPtr {
  AVC_GLOBAL char* ub;
  unsigned off;
};
__device__ __forceinline__ PanelPtr panel_ptr(char* base, int lane) {
  const unsigned long long v = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  PanelPtr p;
  p.ub = (AVC_GLOBAL char*)(((unsigned long long)hi << 32) | lo);
  p.off = (unsigned)lane * 16u;
  return p;
}
template <typename V> __device__ __forceinline__ AVC_GLOBAL V* tile_addr(const PanelPtr& pp, int tile) {
  AVC_GLOBAL char* tb = pp.ub + (long)tile * 2048;
  asm("" : "+s"(tb));
  return (AVC_GLOBAL V*)(tb + pp.off);
}
// KEEP = read back soon by the same kernel (normal cache policy), otherwise streamed past the caches (consumed by a later launch)
// (No predicate: wavefronts past the end of the point set write to the SINK block that follows the last real block of the
// panel buffer -- a branch around every store splits the scheduling regions of the epilogues and costs ~110 spilled registers.)
template <bool KEEP, typename V>
__device__ __forceinline__ void tile_store(const PanelPtr& pp, int tile, const V& f0, const V& f1) {
  AVC_GLOBAL V* p = tile_addr<V>(pp, tile);
#ifdef AVC_KEEP_NT   // timing experiment: the re-read tiles streamed past the caches as well
  constexpr bool keep = false;
#else
  constexpr bool keep = KEEP;
#endif
  if (keep) {
    p[0] = f0;
    p[64] = f1;
  } else {
    AVC_NT_STORE(f0, &p[0]);
    AVC_NT_STORE(f1, &p[64]);
  }
}
// all tiles of one activation (fragment array `f`, 2 per tile) in one go -- what the hooks above issue
template <bool KEEP, int NT, typename V, int KS>
__device__ __forceinline__ void tiles_store(const PanelPtr& pp, int tile0, const V (&f)[KS]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) tile_store<KEEP>(pp, tile0 + t, f[2 * t], f[2 * t + 1]);
}
// the g-th of ng shares of the tiles of one activation (g, ng compile-time after unrolling: the tile test folds)
template <bool KEEP, int NT, typename V, int KS>
__device__ __forceinline__ void tiles_store_part(const PanelPtr& pp, int tile0, const V (&f)[KS], int g, int ng) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t * ng / NT == g) tile_store<KEEP>(pp, tile0 + t, f[2 * t], f[2 * t + 1]);
}
template <typename V> struct FragPair { V a0, a1; };     // the two k-step fragments of a panel tile
template <bool NT, typename V>
__device__ __forceinline__ FragPair<V> tile_load(const PanelPtr& pp, int tile) {
  const AVC_GLOBAL V* p = tile_addr<V>(pp, tile);
  FragPair<V> d;
  if (NT) { d.a0 = AVC_NT_LOAD(&p[0]); d.a1 = AVC_NT_LOAD(&p[64]); }
  else { d.a0 = p[0]; d.a1 = p[64]; }
  return d;
}

// SDF value only (avc_sdf_forward): same layers as sdf_trunk but every activation array dies as soon as the next layer
// has consumed it, which keeps the kernel at two wavefronts per SIMD.
template <class N, class ST, typename TP>
__device__ __forceinline__ float sdf_only(ST& sg, const h8* __restrict__ Wf, TP T, const AvcOffsets& o,
                                          int h, const float (&x)[3]) {
  PE pe;
  pe_compute(x, h, pe);
  float part = 0.f;
  {
    const auto wpe = T + o.v[OFF_WL0_PE] + h * 24;
#pragma unroll
    for (int q = 0; q < 24; ++q) part += wpe[q] * pe.v[q];
    asm volatile("" : "+v"(part));   // done HERE (hipcc otherwise sinks the fmacs to the first use of `part` and keeps their operands alive)
  }
  h8 hlast[N::HK];
  {
    h8 h1[N::HK];
    {
      h8 pef[3];
      pe_to_frags_f16(pe, x, h, pef);
      layer_s1<h8, 3, N::HT>(sg, Wf, o.v[OFF_W0], nxt<N, OFF_WM0>(sg, Wf, o), pef, AVC_EPI(
        if constexpr (AVC_SDF_F16_ACT != 0) { softplus_frags_f16(acc, h1[2 * t], h1[2 * t + 1]); } else {
        float a[16];
        softplus2_tile(acc, a);
        acc_to_frags(a, h1[2 * t], h1[2 * t + 1]); }
      ), NoHook{}, TabBias{T + o.v[OFF_B0], h});
    }
    if constexpr (N::NMID == 2) {
      h8 hm0[N::HK];
      layer_s1<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wf, o), h1, AVC_EPI(
        if constexpr (AVC_SDF_F16_ACT != 0) { softplus_frags_f16(acc, hm0[2 * t], hm0[2 * t + 1]); } else {
        float a[16];
        softplus2_tile(acc, a);
        acc_to_frags(a, hm0[2 * t], hm0[2 * t + 1]); }
      ), NoHook{}, TabBias{T + o.v[OFF_BM0], h});
      layer_s1<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wf, o), hm0, AVC_EPI(
        if constexpr (AVC_SDF_F16_ACT != 0) { softplus_frags_f16(acc, hlast[2 * t], hlast[2 * t + 1]); } else {
        float a[16];
        softplus2_tile(acc, a);
        acc_to_frags(a, hlast[2 * t], hlast[2 * t + 1]); }
      ), NoHook{}, TabBias{T + o.v[OFF_BM1], h});
    } else {
      layer_s1<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wf, o), h1, AVC_EPI(
        if constexpr (AVC_SDF_F16_ACT != 0) { softplus_frags_f16(acc, hlast[2 * t], hlast[2 * t + 1]); } else {
        float a[16];
        softplus2_tile(acc, a);
        acc_to_frags(a, hlast[2 * t], hlast[2 * t + 1]); }
      ), NoHook{}, TabBias{T + o.v[OFF_BM0], h});
    }
  }
  layer_s1<h8, N::HK, N::ST>(sg, Wf, o.v[OFF_WS], no_next(), hlast, AVC_EPI(
    float w[16];
    load16(T + o.v[OFF_WL0_ACC], t, h, w);
    _Pragma("unroll") for (int r = 0; r < 16; ++r) part += w[r] * softplus2(acc[r]);
  ), NoHook{}, TabBias{T + o.v[OFF_BS], h});
  return xhalf_sum(part) + T[o.v[OFF_BL0]];
}

// ---------------------------------------------------------------------------------------------------------------
// Two 32-point groups per wavefront (round 6): the structural terms of the one-group engine -- one LDS A fragment
// per MFMA, one group barrier per 4 weight tiles per 32 points, 392 KiB of weight DMA per workgroup round -- are per WAVEFRONT, not per
// point.  A wavefront that owns 64 points feeds every A fragment to TWO MFMAs (independent accumulators, so a filler between them
// never breaks a back-to-back accumulate path), at twice the register state: ~380 live registers, i.e. one wavefront per SIMD on the
// 512-entry unified file (MFMA takes its B operands from either half).
// ---------------------------------------------------------------------------------------------------------------
#define AVC_EPI2(...) [&](int t, int q, const facc& acc) __attribute__((always_inline)) { __VA_ARGS__ }
#ifndef AVC_LDS_AHEAD_M2
#define AVC_LDS_AHEAD_M2 8   // A fragments in flight ahead of the MFMA pairs
#endif
// the epilogues of two groups (2 x 16 elements: exp, add, log, med3 + 8 conversions = ~150 instructions, 64 of them transcendental)
// dealt over the 2 KS MFMAs of the next tile: with ONE wavefront per SIMD about five single-issue instructions fit the 32-cycle shadow of
// an MFMA (MI355X_MICROARCH.md, per-instruction constants), which is what a 256-wide layer offers (32 MFMAs per tile step)
#ifndef AVC_M2_TRANS_PER_MFMA
#define AVC_M2_TRANS_PER_MFMA 2
#endif
#ifndef AVC_M2_VALU_PER_MFMA
#define AVC_M2_VALU_PER_MFMA 3
#endif
template <int NM>
__device__ __forceinline__ void interleave_m2() {
  constexpr int scale = NM >= 32 ? 1 : (32 + NM - 1) / NM;   // short chains (layer 0: 6 MFMAs) take the whole epilogue between them
#pragma unroll
  for (int k = 0; k < NM; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x400, AVC_M2_TRANS_PER_MFMA * scale, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, AVC_M2_VALU_PER_MFMA * scale, 0);
  }
}
template <typename V, int KS, class ST, class B>
__device__ __forceinline__ void tile_mma_m2(const ST& st, int j, const V (&in0)[KS], const V (&in1)[KS], facc& acc0, facc& acc1,
                                            const B& bias, int t) {
  const V* a_lds = reinterpret_cast<const V*>(st.lds + st.par * ST::BUF_BYTES + j * KS * 1024) + st.lane;
  if constexpr (B::on) {   // the tile's bias row enters through both accumulators (read twice: an LDS read is cheaper than 16 moves)
    float b0[16], b1[16];
    load16(bias.tab, t, bias.h, b0);
    load16(bias.tab, t, bias.h, b1);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = b0[r]; acc1[r] = b1[r]; }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  }
  V a[KS];
#pragma unroll
  for (int s = 0; s < KS && s < AVC_LDS_AHEAD_M2; ++s) a[s] = a_lds[s * 64];
#pragma unroll
  for (int s = 0; s < KS; s += 2) {
    if (s + 1 < KS) asm("" : "+v"(a[s]), "+v"(a[s + 1]));
#pragma unroll
    for (int k = s + AVC_LDS_AHEAD_M2; k < s + AVC_LDS_AHEAD_M2 + 2 && k < KS; ++k) a[k] = a_lds[k * 64];
#pragma unroll
    for (int k = s; k < s + 2 && k < KS; ++k) {
      acc0 = MF<V>::mma(a[k], in0[k], acc0);
      acc1 = MF<V>::mma(a[k], in1[k], acc1);
    }
  }
}
template <typename V, int KS, int NT, class ST, typename Epi, class Bias = NoBias>
__device__ __forceinline__ void layer_m2(ST& st, const V* __restrict__ blob, int offw, const Next& after, const V (&in0)[KS],
                                         const V (&in1)[KS], Epi&& epi, const Bias& bias = NoBias{}) {
  constexpr int G = ST::template group<KS>();
  constexpr int NG = (NT + G - 1) / G;
  facc prev0, prev1;
  int tp = -1;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    AVC_SYNC();
    if (g + 1 < NG) {
      Next n;
      n.ptr = blob + (offw >> 3) + (long)((g + 1) * G * KS) * 64;
      n.chunks = KS * ((NT - (g + 1) * G) < G ? (NT - (g + 1) * G) : G);
      stage_issue(st, n, st.par ^ 1);
    } else {
      stage_issue(st, after, st.par ^ 1);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int t = g * G + j;
      if (t < NT) {
        facc a0, a1;
        tile_mma_m2<V, KS>(st, j, in0, in1, a0, a1, bias, t);
        if (tp >= 0) {
          epi(tp, 0, prev0);
          epi(tp, 1, prev1);
          interleave_m2<2 * KS>();
        }
        prev0 = a0;
        prev1 = a1;
        tp = t;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    st.par ^= 1;
  }
  epi(tp, 0, prev0);
  epi(tp, 1, prev1);
  __builtin_amdgcn_sched_barrier(0);
}

// SDF value of two 32-point groups (x[q] = this lane's point of group q): the layers of sdf_only, every A fragment used twice
template <class N, class ST, typename TP>
__device__ __forceinline__ void sdf_only2(ST& sg, const h8* __restrict__ Wf, TP T, const AvcOffsets& o, int h, const float (&x)[2][3],
                                          float (&out)[2]) {
  float part[2] = {0.f, 0.f};
  h8 pef[2][3];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    PE pe;
    pe_compute(x[q], h, pe);
    const auto wpe = T + o.v[OFF_WL0_PE] + h * 24;
#pragma unroll
    for (int k = 0; k < 24; ++k) part[q] += wpe[k] * pe.v[k];
    asm volatile("" : "+v"(part[q]));
    pe_to_frags_f16(pe, x[q], h, pef[q]);
  }
  // the activations live in the ACCUMULATION half of the unified register file from the moment they exist ("a" constraint: 8 v_accvgpr_write
  // per tile and group); the MFMAs take them from there as B operands, and the 256 architectural VGPRs stay free for the accumulators, the
  // A fragments and the epilogues' temporaries -- left to itself hipcc keeps ~180 of the 256 activation registers in VGPRs and then serialises
  // every softplus through one temporary (exp -> add -> log -> med3, dependent, with nobody else on the SIMD to fill the latencies)
#define AVC_M2_ACT(OUT) AVC_EPI2(float a[16]; softplus2_tile(acc, a);                                                          \
                                 h8 f0, f1;                                                                                   \
                                 _Pragma("unroll") for (int j = 0; j < 8; ++j) { f0[j] = (_Float16)a[j]; f1[j] = (_Float16)a[8 + j]; } \
                                 asm volatile("" : "+a"(f0), "+a"(f1));                                                       \
                                 OUT[q][2 * t] = f0; OUT[q][2 * t + 1] = f1;)
  h8 hlast[2][N::HK];
  {
    h8 h1[2][N::HK];
    layer_m2<h8, 3, N::HT>(sg, Wf, o.v[OFF_W0], nxt<N, OFF_WM0>(sg, Wf, o), pef[0], pef[1], AVC_M2_ACT(h1), TabBias{T + o.v[OFF_B0], h});
    if constexpr (N::NMID == 2) {
      h8 hm0[2][N::HK];
      layer_m2<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WM1>(sg, Wf, o), h1[0], h1[1], AVC_M2_ACT(hm0), TabBias{T + o.v[OFF_BM0], h});
      layer_m2<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM1], nxt<N, OFF_WS>(sg, Wf, o), hm0[0], hm0[1], AVC_M2_ACT(hlast), TabBias{T + o.v[OFF_BM1], h});
    } else {
      layer_m2<h8, N::HK, N::HT>(sg, Wf, o.v[OFF_WM0], nxt<N, OFF_WS>(sg, Wf, o), h1[0], h1[1], AVC_M2_ACT(hlast), TabBias{T + o.v[OFF_BM0], h});
    }
  }
  layer_m2<h8, N::HK, N::ST>(sg, Wf, o.v[OFF_WS], no_next(), hlast[0], hlast[1], AVC_EPI2(
    float w[16];
    load16(T + o.v[OFF_WL0_ACC], t, h, w);
    _Pragma("unroll") for (int r = 0; r < 16; ++r) part[q] += w[r] * softplus2(acc[r]);
  ), TabBias{T + o.v[OFF_BS], h});
#pragma unroll
  for (int q = 0; q < 2; ++q) out[q] = xhalf_sum(part[q]) + T[o.v[OFF_BL0]];
}
