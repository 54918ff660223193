// Micro-benchmark, round 6 (development aid; results in profiles/r06_ubench4_cycles.txt): does VALU / transcendental work issued beside a chain of
// v_mfma_f32_32x32x16_f16 cost SIMD time, measured in SHADER CYCLES (s_memtime inside the kernel) instead of wall time -- rounds 1 and 2 timed these
// loops with events, i.e. through a clock that moves with the instruction mix (DVFS: 1.9-2.3 GHz), and DESIGN.md section 5 rests on them.
//   slot = one MFMA (round-robin over NACC independent accumulators) + NV independent v_fma_f32 + NT independent v_exp_f32, scheduling barrier per slot.
//   W wavefronts per SIMD run the same loop (4 W waves per workgroup, one workgroup per CU, 256 workgroups).
//   out[wave] = cycles for ITERS x 16 slots.
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float facc __attribute__((ext_vector_type(16)));

template <int MFMA, int NACC, int NV, int NT>
__global__ void ub4_kernel(unsigned long long* out, int iters) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * j); }
  facc acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float v[12], t[6];
  for (int q = 0; q < 12; ++q) v[q] = 1.0f + 0.01f * (threadIdx.x + q);
  for (int q = 0; q < 6; ++q) t[q] = 0.5f + 0.001f * (threadIdx.x + q);
  // warm-up (instruction cache, clocks)
  for (int it = 0; it < 8; ++it)
    for (int k = 0; k < 16; ++k)
      if (MFMA) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k % NACC], 0, 0, 0);
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (MFMA) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k % NACC], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q % 12] = __builtin_fmaf(v[q % 12], 1.0001f, 0.5f);
#pragma unroll
      for (int q = 0; q < NT; ++q) t[q % 6] = __builtin_amdgcn_exp2f(t[q % 6]) * 0.25f + 0.0f;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  for (int q = 0; q < 12; ++q) s += v[q];
  for (int q = 0; q < 6; ++q) s += t[q];
  if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0 + (s == 12345.678f ? 1 : 0);
}

#define L(MF, NA, NV, NT) hipLaunchKernelGGL((ub4_kernel<MF, NA, NV, NT>), dim3(grid), dim3(64 * waves), 0, (hipStream_t)stream, out, iters)
#define CASE(id, MF, NA, NV, NT) case id: L(MF, NA, NV, NT); break;
extern "C" int ub4(int variant, int waves, int grid, int iters, unsigned long long* out, void* stream) {
  switch (variant) {
    CASE(0, 1, 2, 0, 0)   CASE(1, 0, 2, 4, 0)   CASE(2, 1, 2, 4, 0)   CASE(3, 0, 2, 8, 0)   CASE(4, 1, 2, 8, 0)
    CASE(5, 0, 2, 0, 2)   CASE(6, 1, 2, 0, 2)   CASE(7, 0, 2, 0, 4)   CASE(8, 1, 2, 0, 4)
    CASE(9, 0, 2, 5, 3)   CASE(10, 1, 2, 5, 3)  CASE(11, 1, 1, 5, 3)  CASE(12, 1, 1, 0, 0)  CASE(13, 1, 4, 8, 0)
    CASE(14, 0, 2, 12, 0) CASE(15, 1, 2, 12, 0) CASE(16, 0, 2, 2, 0)  CASE(17, 1, 2, 2, 0)
    default: return -1;
  }
  return (int)hipGetLastError();
}
