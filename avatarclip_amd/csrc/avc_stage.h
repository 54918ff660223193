// Workgroup-level weight staging for the register-resident MLP engine.
//
// Every wavefront of a workgroup walks the same static sequence of 32-row weight tiles.  Instead of each wave
// streaming its own copy of every 1-KiB fragment from L2 (what caps the un-staged kernels at ~17 % of MFMA peak),
// the workgroup copies each tile ONCE into LDS with direct global->LDS DMA (global_load_lds_dwordx4; the packed
// fragment order is exactly the lane-linear image that instruction writes) and every wave reads its A operands
// with conflict-free ds_read_b128.  Two LDS buffers: tile t+1 is in flight while tile t feeds the MFMAs; one
// __syncthreads per tile orders both the RAW (DMA landed) and the WAR (buffer free) hazard.
#pragma once
#include "avc_common.h"

#define STAGE_BUF_BYTES (18 * 1024)   // >= 17 k-steps x 1 KiB (largest tile: K = skip features + PE slots / H + [x,n])
#define STAGE_LDS_BYTES (2 * STAGE_BUF_BYTES)

struct Stage {
  char* lds;   // STAGE_LDS_BYTES, 16-byte aligned
  int par;     // buffer holding the tile that is consumed next
  int wave;    // wave index in the workgroup (SGPR)
  int lane;
  int nw;      // waves per workgroup
};

__device__ __forceinline__ Stage stage_init(char* lds) {
  Stage st;
  st.lds = lds;
  st.par = 0;
  st.wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  st.lane = threadIdx.x & 63;
  st.nw = blockDim.x >> 6;
  return st;
}

template <typename V, int KS>
__device__ __forceinline__ void stage_issue(const Stage& st, const void* __restrict__ gtile_, int buf) {
  // gtile: first 16-B chunk of the tile (lane 0, k-step 0); chunk c of the tile is 64 lanes x 16 B = 1 KiB
  // (V only documents the element type: f16 and bf16 tiles have the same byte image)
  const char* gtile = reinterpret_cast<const char*>(gtile_);
  char* dst = st.lds + buf * STAGE_BUF_BYTES;
  for (int c = st.wave; c < KS; c += st.nw) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gtile + c * 1024 + st.lane * 16),
                                     (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
  }
}

// Consume the staged tile (KS k-steps) against the register-resident B operands `in`, after issuing the copy of the
// next tile (KSN k-steps at gnext; nullptr = nothing follows).
template <typename V, int KS, int KSN>
__device__ __forceinline__ facc tile_gemm_s(Stage& st, const void* __restrict__ gnext, const V (&in)[KS]) {
  __syncthreads();   // tile in buffer `par` has landed (hipcc drains vmcnt before the barrier); buffer par^1 is free
  if (gnext) stage_issue<V, KSN>(st, gnext, st.par ^ 1);
  const V* a = reinterpret_cast<const V*>(st.lds + st.par * STAGE_BUF_BYTES) + st.lane;
  facc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) acc = MF<V>::mma(a[s * 64], in[s], acc);
  st.par ^= 1;
  return acc;
}
template <typename V, int KA, int KB, int KSN>
__device__ __forceinline__ facc tile_gemm2_s(Stage& st, const void* __restrict__ gnext, const V (&ina)[KA], const V (&inb)[KB]) {
  __syncthreads();
  if (gnext) stage_issue<V, KSN>(st, gnext, st.par ^ 1);
  const V* a = reinterpret_cast<const V*>(st.lds + st.par * STAGE_BUF_BYTES) + st.lane;
  facc acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < KA; ++s) acc = MF<V>::mma(a[s * 64], ina[s], acc);
#pragma unroll
  for (int s = 0; s < KB; ++s) acc = MF<V>::mma(a[(KA + s) * 64], inb[s], acc);
  st.par ^= 1;
  return acc;
}

// global address of tile t of a packed weight (KS k-steps per tile), lane-0 chunk
template <typename V, int KS>
__device__ __forceinline__ const V* gtile(const V* blob, int off, int t) {
  return blob + (off >> 3) + (long)(t * KS) * 64;
}
