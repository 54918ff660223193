// Error plumbing + version for the C ABI (include/avc.h).
#include "avc_common.h"
#include "../../include/avc.h"
#include <string.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" const char* avc_last_error(void) { return g_err; }
extern "C" int avc_version(void) { return AVC_ABI_VERSION; }
extern "C" int avc_num_offsets(void) { return OFF_COUNT; }

void avc_set_error(const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

int avc_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

// ---- hardware-layout probe used by tests/test_gpu_kernels.py: D = A x B for one wavefront, operands and result
// ---- passed as raw per-lane fragments, so the test can check the assumed gfx950 operand / accumulator maps.
__global__ void probe_mfma_kernel(const h8* a, const h8* b, float* d, const b8* ab, const b8* bb, float* db) {
  const int lane = threadIdx.x;
  facc acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = MF<h8>::mma(a[lane], b[lane], acc);
  for (int r = 0; r < 16; ++r) d[lane * 16 + r] = acc[r];
  facc acc2;
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  acc2 = MF<b8>::mma(ab[lane], bb[lane], acc2);
  for (int r = 0; r < 16; ++r) db[lane * 16 + r] = acc2[r];
}
extern "C" int avc_probe_mfma(const void* a_f16, const void* b_f16, float* d, const void* a_bf16, const void* b_bf16,
                              float* d_bf, void* stream) {
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const h8*)a_f16, (const h8*)b_f16, d,
                     (const b8*)a_bf16, (const b8*)b_bf16, d_bf);
  return avc_check_launch("avc_probe_mfma");
}
