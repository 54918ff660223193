// Shared pieces of the weight-gradient contraction (avc_wgrad.hip, avc_bwd_ring.hip): LDS ring geometry, counted vmcnt waits,
// the LDS transpose read that turns a fragment-layout tile (lane = point) into the feature-major MFMA operand, f16 -> bf16.
#pragma once
#include "avc_common.h"
#define WG_TB_MAX 9
#define WG_TILES_MAX 18
#define WG_BUF_BYTES (WG_TILES_MAX * 2048)
#ifndef WG_DEPTH
#define WG_DEPTH 4   // blocks in the LDS ring: one being contracted, WG_DEPTH - 1 copies in flight
#endif
// cache policy of the panel -> LDS copies (aux operand of global_load_lds): 2 = nt -- every tile is read exactly once by exactly one
// workgroup.  (Measured at 4 Mi points, profiles/r03_ab_kernels.txt: nt 10.15 ms vs default 10.22; a ring as deep as the LDS allows per
// pair -- 8 slots for the 9/10-tile products, 5 for the 15/16-tile ones -- 10.22 vs 10.23: the kernel does not lack bytes in flight.)
#ifndef WG_DMA_AUX
#define WG_DMA_AUX 2
#endif

// the two operand regions (csrc/avc_mlp.h: PanelLayout): [0] = F region (forward-type, f16), [1] = G region (gradient-type, bf16)
struct WgRegions { const char* base[2]; long stride[2]; };   // byte address of block 0 of the slab, bytes per block

typedef short vs4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) char lds_char;

// wait until at most `n` of this wave's vector-memory operations are outstanding (n is wave-uniform; the count is an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define WG_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    WG_W(0) WG_W(1) WG_W(2) WG_W(3) WG_W(4) WG_W(5) WG_W(6) WG_W(7) WG_W(8) WG_W(9) WG_W(10) WG_W(11) WG_W(12) WG_W(13) WG_W(14) WG_W(15)
    WG_W(16) WG_W(17) WG_W(18) WG_W(19) WG_W(20) WG_W(21) WG_W(22) WG_W(23) WG_W(24) WG_W(25) WG_W(26) WG_W(27) WG_W(28) WG_W(29) WG_W(30)
    WG_W(31) WG_W(32) WG_W(33) WG_W(34) WG_W(35) WG_W(36)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // any other count: conservative
  }
#undef WG_W
}

// operand fragment of k-step t (points 16 t .. 16 t + 15) of the tile at LDS address `tile` (+ this lane's offset): lane
// (feature m = lane & 31, hh = lane >> 5) receives points 16 t + 8 hh + 0..7
__device__ __forceinline__ b8 tr_frag(lds_char* p, int t) {
  struct { vs4 lo, hi; } r;
  r.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) vs4*>(p + t * 1024));
  r.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) vs4*>(p + t * 1024 + 256));
  return __builtin_bit_cast(b8, r);
}
__device__ __forceinline__ b8 f16_to_bf16(b8 v) {
  const h8 x = __builtin_bit_cast(h8, v);
  b8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)(float)x[j];
  return o;
}

