// Dense-parameter assembly of one optimisation step: the weight-norm reparametrisation of every linear, W = g * v / ||v||_row
// (nn.utils.weight_norm(dim=0), reference fields.py:65-66,139-143), written straight into the flat dense vector the packing reads
// (avatarclip_amd/packing.param_shapes order), and its backward
//     dg_r = (dW_r . v_r) / ||v_r||,    dv_r = g_r / ||v_r|| * (dW_r - (dW_r . v_r / ||v_r||^2) v_r),    db = dflat slice.
// One launch each way instead of ~4 torch kernels per layer forward and ~10 backward (9 weight-normed linears in the full nets:
// ~150 launches of 4-5 us per step, profiles/r03_step_census.txt).  One wavefront per weight row; rows are 39..262 wide.
#include "avc_common.h"
#include "../../include/avc.h"

#define AVC_WN_MAX 16
struct WnDesc {
  const float* v[AVC_WN_MAX];    // weight_v [rows, cols] (or the plain weight when g == nullptr)
  const float* g[AVC_WN_MAX];    // weight_g [rows, 1] or nullptr
  const float* b[AVC_WN_MAX];    // bias [rows] or nullptr
  float* dv[AVC_WN_MAX];         // backward outputs (same shapes)
  float* dg[AVC_WN_MAX];
  float* db[AVC_WN_MAX];
  int rows[AVC_WN_MAX], cols[AVC_WN_MAX];
  long w_off[AVC_WN_MAX], b_off[AVC_WN_MAX];   // offsets of W (row-major) and of the bias in the flat vector (floats)
  int row_start[AVC_WN_MAX + 1];               // prefix sum of rows
  int n;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int find_layer(const WnDesc& d, int row) {
  int l = 0;
  while (l + 1 < d.n && row >= d.row_start[l + 1]) ++l;
  return l;
}

__global__ __launch_bounds__(256) void wn_fwd_kernel(WnDesc d, float* __restrict__ flat) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= d.row_start[d.n]) return;
  const int l = find_layer(d, row), r = row - d.row_start[l], C = d.cols[l];
  const float* v = d.v[l] + (long)r * C;
  float* w = flat + d.w_off[l] + (long)r * C;
  float scale = 1.f;
  if (d.g[l]) {
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) ss += v[c] * v[c];
    scale = d.g[l][r] / sqrtf(wave_sum(ss));
  }
  for (int c = lane; c < C; c += 64) w[c] = v[c] * scale;
  if (lane == 0 && d.b[l]) flat[d.b_off[l] + r] = d.b[l][r];
}

__global__ __launch_bounds__(256) void wn_bwd_kernel(WnDesc d, const float* __restrict__ dflat) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= d.row_start[d.n]) return;
  const int l = find_layer(d, row), r = row - d.row_start[l], C = d.cols[l];
  const float* v = d.v[l] + (long)r * C;
  const float* dw = dflat + d.w_off[l] + (long)r * C;
  float* dv = d.dv[l] + (long)r * C;
  if (d.g[l]) {
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < C; c += 64) { ss += v[c] * v[c]; dot += dw[c] * v[c]; }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float inv = 1.f / sqrtf(ss), g = d.g[l][r];
    const float a = g * inv, bcoef = g * dot * inv / ss;
    for (int c = lane; c < C; c += 64) dv[c] = a * dw[c] - bcoef * v[c];
    if (lane == 0) d.dg[l][r] = dot * inv;
  } else {
    for (int c = lane; c < C; c += 64) dv[c] = dw[c];
  }
  if (lane == 0 && d.db[l]) d.db[l][r] = dflat[d.b_off[l] + r];
}

static int fill_desc(WnDesc& d, int n, const void* const* v, const void* const* g, const void* const* b, void* const* dv,
                     void* const* dg, void* const* db, const int* rows, const int* cols, const long* w_off, const long* b_off) {
  if (n < 1 || n > AVC_WN_MAX) { avc_set_error("avc_dense_params: 1 <= layers <= 16"); return 1; }
  d.n = n;
  d.row_start[0] = 0;
  for (int i = 0; i < n; ++i) {
    d.v[i] = (const float*)v[i]; d.g[i] = (const float*)g[i]; d.b[i] = (const float*)b[i];
    d.dv[i] = dv ? (float*)dv[i] : nullptr; d.dg[i] = dg ? (float*)dg[i] : nullptr; d.db[i] = db ? (float*)db[i] : nullptr;
    d.rows[i] = rows[i]; d.cols[i] = cols[i]; d.w_off[i] = w_off[i]; d.b_off[i] = b_off[i];
    d.row_start[i + 1] = d.row_start[i] + rows[i];
  }
  return 0;
}

extern "C" int avc_dense_params_fwd(int n, const void* const* v, const void* const* g, const void* const* b, const int* rows,
                                    const int* cols, const long* w_off, const long* b_off, float* flat, void* stream) {
  WnDesc d;
  if (fill_desc(d, n, v, g, b, nullptr, nullptr, nullptr, rows, cols, w_off, b_off)) return 1;
  const int total = d.row_start[n];
  hipLaunchKernelGGL(wn_fwd_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, d, flat);
  return avc_check_launch("avc_dense_params_fwd");
}
extern "C" int avc_dense_params_bwd(int n, const void* const* v, const void* const* g, void* const* dv, void* const* dg,
                                    void* const* db, const int* rows, const int* cols, const long* w_off, const long* b_off,
                                    const float* dflat, void* stream) {
  WnDesc d;
  if (fill_desc(d, n, v, g, (const void* const*)db, dv, dg, db, rows, cols, w_off, b_off)) return 1;
  const int total = d.row_start[n];
  hipLaunchKernelGGL(wn_bwd_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, d, dflat);
  return avc_check_launch("avc_dense_params_bwd");
}

// The packed parameter blobs of one optimisation step (engine.Packed; packing.py: pure index gathers of the flat dense vector):
//   w16[i] = flat[idx16[i]] * scale16[i] as f16 AND as bf16 (forward / gradient sweeps), tab[i] = flat[idx32[i]] * scale32[i];
// index nparam = the appended zero.  One launch instead of cat, 2 gathers, 2 multiplies, 2 conversions.
__global__ __launch_bounds__(256) void pack_params_kernel(const float* __restrict__ flat, int nparam, const long* __restrict__ idx16,
                                                          const float* __restrict__ scale16, int n16, const long* __restrict__ idx32,
                                                          const float* __restrict__ scale32, int n32, _Float16* __restrict__ w_f16,
                                                          __bf16* __restrict__ w_bf16, float* __restrict__ tab) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) {
    const long k = idx16[i];
    float w = (k < nparam ? flat[k] : 0.f) * scale16[i];
    // the product is rounded to fp32 first, like the torch multiply it replaces: left to itself the compiler selects a mixed-precision
    // multiply + conversion (v_fma_mixlo_f16) that rounds once and differs in the last f16 bit where the fp32 product is a tie
    asm volatile("" : "+v"(w));
    w_f16[i] = (_Float16)w;
    w_bf16[i] = (__bf16)w;
  } else if (i < n16 + n32) {
    const int j = i - n16;
    const long k = idx32[j];
    tab[j] = (k < nparam ? flat[k] : 0.f) * scale32[j];
  }
}
extern "C" int avc_pack_params(const float* flat, int nparam, const long* idx16, const float* scale16, int n16, const long* idx32,
                               const float* scale32, int n32, void* w_f16, void* w_bf16, float* tab, void* stream) {
  if (!flat || !idx16 || !scale16 || !idx32 || !scale32 || !w_f16 || !w_bf16 || !tab) { avc_set_error("avc_pack_params: NULL buffer"); return 1; }
  hipLaunchKernelGGL(pack_params_kernel, dim3((n16 + n32 + 255) / 256), dim3(256), 0, (hipStream_t)stream, flat, nparam, idx16, scale16, n16,
                     idx32, scale32, n32, (_Float16*)w_f16, (__bf16*)w_bf16, tab);
  return avc_check_launch("avc_pack_params");
}
