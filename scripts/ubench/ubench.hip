// Micro-benchmark: can one wavefront overlap MFMA execution with independent VALU / transcendental work on gfx950?
// mode 0: 16 dependent MFMAs per iteration; mode 1: VALU only (16 x NV v_fma); mode 2: MFMA + NV v_fma after each MFMA;
// mode 3: trans only (16 x NT v_exp); mode 4: MFMA + NT v_exp after each MFMA.  One workgroup per CU slot, W waves/WG.
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float facc __attribute__((ext_vector_type(16)));

template <int MODE, int NV>
__global__ void ub_kernel(float* out, int iters) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * j); }
  facc acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float v[8];
  for (int q = 0; q < 8; ++q) v[q] = 1.0f + 0.01f * (threadIdx.x + q);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (MODE == 0 || MODE == 2 || MODE == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
      }
      if (MODE == 3 || MODE == 4) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7]) * 0.5f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int q = 0; q < 8; ++q) s += v[q];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

#define LAUNCH(MODE, NV) hipLaunchKernelGGL((ub_kernel<MODE, NV>), dim3(grid), dim3(64 * waves), 0, (hipStream_t)stream, out, iters)
extern "C" int ub_run(int mode, int nv, int waves, int grid, int iters, float* out, void* stream) {
  if (nv == 8) {
    switch (mode) { case 0: LAUNCH(0, 8); break; case 1: LAUNCH(1, 8); break; case 2: LAUNCH(2, 8); break; case 3: LAUNCH(3, 8); break; case 4: LAUNCH(4, 8); break; }
  } else if (nv == 2) {
    switch (mode) { case 0: LAUNCH(0, 2); break; case 1: LAUNCH(1, 2); break; case 2: LAUNCH(2, 2); break; case 3: LAUNCH(3, 2); break; case 4: LAUNCH(4, 2); break; }
  } else {
    switch (mode) { case 0: LAUNCH(0, 4); break; case 1: LAUNCH(1, 4); break; case 2: LAUNCH(2, 4); break; case 3: LAUNCH(3, 4); break; case 4: LAUNCH(4, 4); break; }
  }
  return (int)hipGetLastError();
}
