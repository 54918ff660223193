// Workgroup-level weight staging for the register-resident MLP engine.
//
// Every wavefront of a workgroup walks the same static sequence of 32-row weight tiles.  Instead of each wave
// streaming its own copy of every 1-KiB fragment from L2 (what caps the un-staged kernels at ~17 % of MFMA peak),
// the workgroup copies the tiles ONCE into LDS with direct global->LDS DMA (global_load_lds_dwordx4; the packed
// fragment order is exactly the lane-linear image that instruction writes) and every wave reads its A operands
// with conflict-free ds_read_b128.
//
// Granularity: a GROUP of up to G consecutive tiles of one layer (G = 4: half a 256-wide layer, 64-68 KiB).  Two LDS
// buffers: group g+1 is in flight while group g feeds the MFMAs; ONE __syncthreads per group orders both the RAW
// (DMA landed) and the WAR (buffer free) hazard.  Coarse groups matter: hipcc drains vmcnt(0) before a barrier while
// an LDS-DMA is pending, which also waits for every panel/scratch store of the epilogues -- per-tile barriers cost an
// L2 round trip per tile (measured: SQ_WAIT_ANY 78 % of wave cycles in the backward kernel), per-group barriers a
// quarter of that, and a 64 KiB copy has half a layer of MFMA work to hide under.
#pragma once
#include "avc_common.h"

#define STAGE_TILE_BYTES (16 * 1024)   // a 256-wide tile: 16 k-steps x 1 KiB; the 17-k-step layers (K = skip features + PE slots,
                                        // H + [x,n]) go in groups of G-1 tiles so that a buffer is G x 16 KiB (LDS also holds the
                                        // fp32 table and, in the backward kernel, the ReLU masks)

struct Next {          // the group that follows in the static tile sequence
  const void* ptr;     // nullptr = nothing follows
  int chunks;          // 1-KiB chunks to copy
};

template <int G_>
struct StageT {
  static constexpr int G = G_;
  static constexpr int BUF_BYTES = G_ * STAGE_TILE_BYTES;
  static constexpr int LDS_BYTES = 2 * G_ * STAGE_TILE_BYTES;
  // tiles per group of a layer whose tiles have KS k-steps
  template <int KS> static constexpr int group() { return (KS * 1024 * G_ <= BUF_BYTES) ? G_ : BUF_BYTES / (KS * 1024); }
  char* lds;   // LDS_BYTES, 16-byte aligned
  int par;     // buffer holding the group that is consumed next
  int wave;    // wave index in the workgroup (SGPR)
  int lane;
  int nw;      // waves per workgroup
  int turn;    // which of the wavefronts of a SIMD issues the next group's DMA (AVC_DMA_TURNS)
};

template <int G>
__device__ __forceinline__ StageT<G> stage_init(char* lds) {
  StageT<G> st;
  st.lds = lds;
  st.par = 0;
  st.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  st.lane = threadIdx.x & 63;
  st.nw = blockDim.x >> 6;
  st.turn = 0;
  return st;
}

// Whose turn it is to issue (round 6).  Every wavefront of the workgroup used to issue its share of the next group's DMA right after
// the group barrier, in front of its first MFMA: a global_load_lds costs the issuing wave 60-185 cycles (MI355X_MICROARCH.md, "LDS-DMA
// piece issue cost"), all wavefronts of a SIMD pay it at the same moment and nobody feeds the matrix pipe meanwhile -- with the DMA
// removed the SDF kernel runs 19.5 % faster, with the barriers removed 3 % (profiles/r06_ab_kernels.txt: most of the "waits" of DESIGN.md
// section 5).  AVC_DMA_TURNS=1: of the wavefronts that share a SIMD (w, w + 4, w + 8) only ONE issues per group -- the four of a turn,
// one per SIMD, copy the whole group -- and the turn rotates with every group; the others go straight to their MFMA chains and have the
// SIMD to themselves while the issuing wave is busy, which then catches up while they wait at the next barrier.  No new control flow:
// the chunk loop stays where it was, only its first index and stride change (a wave out of turn starts at `chunks`).
#ifndef AVC_DMA_TURNS
#define AVC_DMA_TURNS 1
#endif
#ifndef AVC_DMA_ROT
#define AVC_DMA_ROT 0
#endif
template <class ST>
__device__ __forceinline__ void stage_issue(ST& st, const Next& nx, int buf, bool rotate = true) {
  if (!nx.ptr) return;
#ifdef AVC_ABL_NODMA   // timing ablation only (results are garbage)
  return;
#endif
  const char* g = reinterpret_cast<const char*>(nx.ptr);
  char* dst = st.lds + buf * ST::BUF_BYTES;
  int first = st.wave, stride = st.nw;
#if AVC_DMA_TURNS
  if (rotate) {
    const int nsub = st.nw >> 2;                  // wavefronts per SIMD
    first = ((st.wave >> 2) == st.turn) ? (st.wave & 3) : nx.chunks;
    stride = 4;
    const int nt = st.turn + 1;
    st.turn = nt >= nsub ? 0 : nt;
  }
#endif
#ifdef AVC_ABL_HALFDMA   // timing ablation only (garbage results): every second chunk -- is the cost proportional to the bytes?
  stride *= 2;
#endif
  for (int c = first; c < nx.chunks; c += stride) {
#if AVC_DMA_ROT          // experiment: workgroups walk the group's chunks from different starting points (same data, same LDS image)
    int cs = c + (int)(blockIdx.x * 5u) % nx.chunks;
    cs = cs >= nx.chunks ? cs - nx.chunks : cs;
#else
    const int cs = c;
#endif
#ifdef AVC_ABL_DMA_SAMESRC   // timing ablation only (garbage results): every chunk from the SAME 1 KiB of the blob -- L2 / fabric side or LDS / issue side?
    const int src = 0;
#else
    const int src = cs;
#endif
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + src * 1024 + st.lane * 16),
                                     (__attribute__((address_space(3))) void*)(dst + cs * 1024), 16, 0, 0);
  }
}


// hipcc sinks every ds_read_b128 next to the MFMA that consumes it (ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma, one LDS
// round trip per MFMA, seen in the ISA of every kernel built on this engine).  pinN() makes a batch of fragments opaque at a
// program point, so the reads of the whole batch are issued back to back and the MFMA chain then runs at the matrix pipe's
// own rate; the second half's reads are already in flight while the first half's MFMAs execute.
// (not volatile: a volatile asm is ordered against every memory operation and every other volatile asm, which would pin the
// epilogue of the previous tile -- its loads, stores and its own pin2 -- behind the last batch of the MFMA chain)
template <typename V>
__device__ __forceinline__ void pin4(V& a, V& b, V& c, V& d) { asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }

// Interleave request for one tile step: the scheduling region holds the MFMA chain of tile t and the (independent) epilogue
// of tile t-1.  Left alone, hipcc emits the 16 dependent MFMAs back to back (the wave stalls ~32 cycles on each) and then
// ~150 VALU instructions with the matrix pipe idle; both wavefronts of a SIMD run in lockstep between the group barriers,
// so MFMA time and VALU time ADD (measured: forward kernel = MFMA 32 us + VALU 32 us + LDS 31 us + DMA 24 us per round).
// One MFMA followed by a slice of VALU work, KS times, lets the epilogue run in the shadow of the chain.
#ifndef AVC_VALU_PER_MFMA
#define AVC_VALU_PER_MFMA 10
#endif
#ifndef AVC_TRANS_PER_MFMA
#define AVC_TRANS_PER_MFMA 0   // > 0: ask for that many transcendentals right behind every MFMA (they issue under the matrix pipe for free up to ~2 per MFMA: profiles/r02_ubench2.txt), the plain VALU slice after them
#endif
template <int KS>
__device__ __forceinline__ void interleave_mfma_valu() {
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // 1 MFMA
    if (AVC_TRANS_PER_MFMA > 0) __builtin_amdgcn_sched_group_barrier(0x400, AVC_TRANS_PER_MFMA, 0);   // transcendentals
    __builtin_amdgcn_sched_group_barrier(0x002, AVC_VALU_PER_MFMA, 0);   // then a slice of VALU
  }
}
#ifndef AVC_LDS_AHEAD
#define AVC_LDS_AHEAD 8   // A fragments in flight ahead of the MFMA chain (x4 VGPRs each)
#endif
template <typename V, int KS>
__device__ __forceinline__ facc mma_chain_lds(const V* __restrict__ a_lds /* lane's chunk of k-step 0 */, const V (&in)[KS], facc acc) {
  // rolling prefetch: AVC_LDS_AHEAD fragments are requested up front, every group of 4 MFMAs is preceded by the requests
  // of the group AVC_LDS_AHEAD further on.  (All KS at once costs 4*KS VGPRs -- 64 for a 256-wide layer -- and pushed the
  // sweeps over the 256-register budget: ~140 spilled registers in every kernel built on this engine.)
  V a[KS];
#pragma unroll
  for (int s = 0; s < KS && s < AVC_LDS_AHEAD; ++s) a[s] = a_lds[s * 64];
#pragma unroll
  for (int s = 0; s < KS; s += 4) {
    if (s + 3 < KS) pin4(a[s], a[s + 1], a[s + 2], a[s + 3]);
#pragma unroll
    for (int k = s + AVC_LDS_AHEAD; k < s + AVC_LDS_AHEAD + 4 && k < KS; ++k) a[k] = a_lds[k * 64];
#pragma unroll
    for (int k = s; k < s + 4 && k < KS; ++k) acc = MF<V>::mma(a[k], in[k], acc);
  }
  return acc;
}


// timing ablation (AVC_ABL_BWD_RECOMP, csrc/avc_bwd_body.h): a second accumulator chain on the SAME A fragments -- one LDS read feeds
// two MFMAs -- which is what recomputing h_l inside the second-order sweep would add to its MFMA stream at the very least
template <typename V, int KS>
__device__ __forceinline__ facc mma_chain_lds_dual(const V* __restrict__ a_lds, const V (&in)[KS], facc acc, facc& acc2) {
  V a[KS];
#pragma unroll
  for (int s = 0; s < KS && s < AVC_LDS_AHEAD; ++s) a[s] = a_lds[s * 64];
#pragma unroll
  for (int s = 0; s < KS; s += 4) {
    if (s + 3 < KS) pin4(a[s], a[s + 1], a[s + 2], a[s + 3]);
#pragma unroll
    for (int k = s + AVC_LDS_AHEAD; k < s + AVC_LDS_AHEAD + 4 && k < KS; ++k) a[k] = a_lds[k * 64];
#pragma unroll
    for (int k = s; k < s + 4 && k < KS; ++k) {
      acc = MF<V>::mma(a[k], in[k], acc);
      acc2 = MF<V>::mma(a[k], in[k], acc2);
    }
  }
  return acc;
}

// where the accumulator of an output tile starts: zero, or the tile's bias row from the fp32 table in LDS (see tile_mma below)
struct NoBias { static constexpr bool on = false; };
struct TabBias { static constexpr bool on = true; lds_tab_t tab; int h; };

// ---- two output tiles at once: the MFMA stream alternates between two INDEPENDENT accumulators (both tiles share every B
// ---- operand).  A filler issued between two MFMAs on the SAME accumulator breaks the back-to-back accumulate path of the
// ---- matrix pipe (~+43 cycles, MI355X_MICROARCH.md "per-instruction cycle constants"), which is why a single dependent chain
// ---- cannot hide the epilogue of the previous tile; between MFMAs on different accumulators a filler costs its issue slot.
#ifndef AVC_LDS_AHEAD2
#define AVC_LDS_AHEAD2 4   // A fragments in flight per tile of the pair
#endif
// the MFMAs of two tiles over one input array: a0 / a1 = this lane's chunk of k-step 0 of either tile
template <typename V, int KS>
__device__ __forceinline__ void pair_chain(const V* __restrict__ a0, const V* __restrict__ a1, const V (&in)[KS], facc& acc0, facc& acc1) {
  V fa[KS], fb[KS];
#pragma unroll
  for (int s = 0; s < KS && s < AVC_LDS_AHEAD2; ++s) { fa[s] = a0[s * 64]; fb[s] = a1[s * 64]; }
#pragma unroll
  for (int s = 0; s < KS; s += 2) {
    if (s + 1 < KS) pin4(fa[s], fb[s], fa[s + 1], fb[s + 1]);
#pragma unroll
    for (int k = s + AVC_LDS_AHEAD2; k < s + AVC_LDS_AHEAD2 + 2 && k < KS; ++k) { fa[k] = a0[k * 64]; fb[k] = a1[k * 64]; }
#pragma unroll
    for (int k = s; k < s + 2 && k < KS; ++k) {
      acc0 = MF<V>::mma(fa[k], in[k], acc0);
      acc1 = MF<V>::mma(fb[k], in[k], acc1);
    }
  }
}
template <class B>
__device__ __forceinline__ void pair_init(facc& acc0, facc& acc1, const B& bias, int t) {
  if constexpr (B::on) {   // the bias rows of tiles t, t + 1 enter through the accumulators (see TabBias)
    float b0[16], b1[16];
    load16(bias.tab, t, bias.h, b0);
    load16(bias.tab, t + 1, bias.h, b1);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = b0[r]; acc1[r] = b1[r]; }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  }
}
template <typename V, int KS, class ST, class B>
__device__ __forceinline__ void tile_mma_pair(const ST& st, int j, const V (&in)[KS], facc& acc0, facc& acc1, const B& bias, int t) {
  const V* a0 = reinterpret_cast<const V*>(st.lds + st.par * ST::BUF_BYTES + j * KS * 1024) + st.lane;
  pair_init(acc0, acc1, bias, t);
  pair_chain<V, KS>(a0, a0 + KS * 64, in, acc0, acc1);
}
// the same with the K dimension split over two register arrays (tiles of KA + KB k-steps)
template <typename V, int KA, int KB, class ST, class B>
__device__ __forceinline__ void tile_mma2_pair(const ST& st, int j, const V (&ina)[KA], const V (&inb)[KB], facc& acc0, facc& acc1,
                                               const B& bias, int t) {
  const V* a0 = reinterpret_cast<const V*>(st.lds + st.par * ST::BUF_BYTES + j * (KA + KB) * 1024) + st.lane;
  const V* a1 = a0 + (KA + KB) * 64;
  pair_init(acc0, acc1, bias, t);
  pair_chain<V, KA>(a0, a1, ina, acc0, acc1);
  pair_chain<V, KB>(a0 + KA * 64, a1 + KA * 64, inb, acc0, acc1);
}

// Where the accumulator of an output tile starts: zero, or the tile's bias row read from the fp32 table in LDS straight into
// the accumulator registers (the C operand of the first MFMA) -- one VALU add per output element less in the epilogue, on an
// engine whose VALU time adds to its MFMA time.
// MFMAs of tile j of the current group (KS k-steps per tile) against the register-resident B operands
template <typename V, int KS, class ST, class B = NoBias>
__device__ __forceinline__ facc tile_mma(const ST& st, int j, const V (&in)[KS], const B& bias = NoBias{}, int t = 0) {
  const V* a = reinterpret_cast<const V*>(st.lds + st.par * ST::BUF_BYTES + j * KS * 1024) + st.lane;
  facc acc;
  if constexpr (B::on) {
    float b[16];
    load16(bias.tab, t, bias.h, b);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  }
  return mma_chain_lds<V, KS>(a, in, acc);
}
template <typename V, int KA, int KB, class ST, class B = NoBias>
__device__ __forceinline__ facc tile_mma2(const ST& st, int j, const V (&ina)[KA], const V (&inb)[KB], const B& bias = NoBias{}, int t = 0) {
  const V* a = reinterpret_cast<const V*>(st.lds + st.par * ST::BUF_BYTES + j * (KA + KB) * 1024) + st.lane;
  facc acc;
  if constexpr (B::on) {
    float b[16];
    load16(bias.tab, t, bias.h, b);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  }
  acc = mma_chain_lds<V, KA>(a, ina, acc);
  return mma_chain_lds<V, KB>(a + KA * 64, inb, acc);
}
