// Micro-benchmarks, round 2 (development aid; results in profiles/r02_ubench_*.txt):
//  (1) ub2_mfma: MFMA chains over NACC INDEPENDENT accumulators (round-robin) with NV plain VALU / transcendental fillers per
//      MFMA slot -- the round-1 benchmark used ONE dependent chain, where every filler breaks the back-to-back accumulate path.
//  (2) ub2_atomic: fp32 atomic adds of a 256-KiB tile (64 K floats per workgroup and pass) into a per-XCD buffer selected by
//      the hardware XCC id, workgroup-scope (executed in the XCD's L2) vs agent-scope (memory side) -- the price of merging
//      per-slab weight-gradient tiles on chip instead of shipping operands through HBM.
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float facc __attribute__((ext_vector_type(16)));

template <int NACC, int NV, int TRANS>
__global__ void ub2_mfma_kernel(float* out, int iters) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * j); }
  facc acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = 1.0f + 0.01f * (threadIdx.x + q);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k % NACC], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        if (TRANS) v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7]) * 0.5f;
        else v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  for (int q = 0; q < 8; ++q) s += v[q];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

#define L3(NACC, NV, TR) hipLaunchKernelGGL((ub2_mfma_kernel<NACC, NV, TR>), dim3(grid), dim3(64 * waves), 0, (hipStream_t)stream, out, iters)
#define L2(NACC, NV) do { if (trans) L3(NACC, NV, 1); else L3(NACC, NV, 0); } while (0)
#define L1(NACC) do { switch (nv) { case 0: L2(NACC, 0); break; case 2: L2(NACC, 2); break; case 4: L2(NACC, 4); break; \
                                   case 6: L2(NACC, 6); break; case 8: L2(NACC, 8); break; case 12: L2(NACC, 12); break; default: return -1; } } while (0)
extern "C" int ub2_mfma(int nacc, int nv, int trans, int waves, int grid, int iters, float* out, void* stream) {
  switch (nacc) { case 1: L1(1); break; case 2: L1(2); break; case 4: L1(4); break; default: return -1; }
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int xcc_id() {
  // HW_REG_XCC_ID = 20, bits [3:0]; s_getreg_b32 simm16 = (size-1) << 11 | offset << 6 | id
  return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;
}

template <int SCOPE /* 0 workgroup, 1 agent */>
__global__ __launch_bounds__(512) void ub2_atomic_kernel(float* buf, long buf_stride, int passes, int tile_floats, int* xcc_count) {
  const int x = xcc_id();
  if (threadIdx.x == 0) atomicAdd(&xcc_count[x], 1);
  float* dst = buf + (long)x * buf_stride;
  for (int p = 0; p < passes; ++p) {
    // each pass adds one 256-KiB tile; consecutive workgroups start at different offsets of the 1.6 MB accumulation buffer
    const long base = ((long)(blockIdx.x + p) * tile_floats) % buf_stride;
    for (int i = threadIdx.x; i < tile_floats; i += blockDim.x) {
      long k = base + i;
      if (k >= buf_stride) k -= buf_stride;
      if (SCOPE == 0) __hip_atomic_fetch_add(dst + k, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(dst + k, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// plain read-modify-write of a PRIVATE per-workgroup buffer (no atomics): the HBM/MALL alternative
__global__ __launch_bounds__(512) void ub2_rmw_kernel(float* buf, long buf_stride, int passes, int tile_floats) {
  float4* dst = reinterpret_cast<float4*>(buf + (long)blockIdx.x * buf_stride);
  const int n4 = tile_floats / 4, s4 = (int)(buf_stride / 4);
  for (int p = 0; p < passes; ++p) {
    const int base = (p * n4) % s4;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      int k = base + i;
      if (k >= s4) k -= s4;
      float4 v = dst[k];
      v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
      dst[k] = v;
    }
  }
}
extern "C" int ub2_atomic(int mode, int grid, int passes, int tile_floats, float* buf, long buf_stride, int* xcc_count, void* stream) {
  if (mode == 0) hipLaunchKernelGGL(ub2_atomic_kernel<0>, dim3(grid), dim3(512), 0, (hipStream_t)stream, buf, buf_stride, passes, tile_floats, xcc_count);
  else if (mode == 1) hipLaunchKernelGGL(ub2_atomic_kernel<1>, dim3(grid), dim3(512), 0, (hipStream_t)stream, buf, buf_stride, passes, tile_floats, xcc_count);
  else hipLaunchKernelGGL(ub2_rmw_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, buf, buf_stride, passes, tile_floats);
  return (int)hipGetLastError();
}

// ---- (3) counter calibration: a copy of KNOWN size with the panel access pattern (one wavefront moves 2 KiB tiles, 16 B per
// lane and instruction), so FETCH_SIZE / WRITE_SIZE of rocprofv3 can be checked against bytes that are certain.
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void ub2_copy_kernel(const v4i* __restrict__ src, v4i* __restrict__ dst, long ntiles, int nt) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 8 + (threadIdx.x >> 6), nw = (long)gridDim.x * 8;
  for (long t = wave; t < ntiles; t += nw) {
    const v4i* s = src + t * 128 + lane;
    v4i* d = dst + t * 128 + lane;
    v4i a, b;
    if (nt == 2) {            // write only
      a.x = (int)t; a.y = lane; a.z = 0; a.w = 1; b = a;
      __builtin_nontemporal_store(a, d); __builtin_nontemporal_store(b, d + 64);
      continue;
    }
    if (nt == 3) {            // read only (the sum keeps the loads alive; one lane in a million writes)
      a = __builtin_nontemporal_load(s); b = __builtin_nontemporal_load(s + 64);
      if ((a.x ^ b.y) == 0x7fffffff) d[0] = a;
      continue;
    }
    if (nt) { a = __builtin_nontemporal_load(s); b = __builtin_nontemporal_load(s + 64); }
    else { a = s[0]; b = s[64]; }
    a.x += 1; b.y += 1;
    if (nt) { __builtin_nontemporal_store(a, d); __builtin_nontemporal_store(b, d + 64); }
    else { d[0] = a; d[64] = b; }
  }
}
extern "C" int ub2_copy(const void* src, void* dst, long ntiles, int nt, int grid, void* stream) {
  hipLaunchKernelGGL(ub2_copy_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, (const v4i*)src, (v4i*)dst, ntiles, nt);
  return (int)hipGetLastError();
}
