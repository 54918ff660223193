// Per-ray kernels: hierarchical up-sampling and NeuS compositing (forward + reverse).  One wavefront per ray,
// the ray's samples staged in LDS, wave scans for the cumulative product / sum.
//   avc_upsample_step   renderer.py:133-177 (up_sample) + :39-69 (sample_pdf, det=True) + :179-193 (cat_z_vals merge)
//   avc_composite_fwd   renderer.py:234-286 (alpha, transmittance, colours, eikonal partials)
//   avc_composite_bwd   its reverse
#include <cstdlib>
#include "avc_common.h"
#include "../../include/avc.h"

#define MAXS 256          // max samples per ray handled by these kernels
#define RPB 4             // rays (wavefronts) per 256-thread block
// Every wavefront works on its own ray in its own LDS rows: the phases of a ray need ordering only WITHIN the wavefront.  LDS
// operations of one wavefront complete in issue order, so draining its LDS counter (a compiler barrier as well) is all the
// synchronisation there is -- a workgroup barrier here made the four rays of a block wait for each other nine times per step.
#define WAVE_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ float wave_incl_scan_mul(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(v, d);
    if (lane >= d) v *= o;
  }
  return v;
}
__device__ __forceinline__ float wave_incl_scan_add(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// exclusive cumprod over n values in LDS buf (in place): out[i] = prod_{j<i} buf[j].  Each lane owns a
// contiguous chunk of `per` values.
__device__ __forceinline__ void excl_cumprod(float* buf, int n, int per, int lane) {
  float loc = 1.f;
  const int b = lane * per;
  for (int k = 0; k < per; ++k) if (b + k < n) loc *= buf[b + k];
  float inc = wave_incl_scan_mul(loc, lane);
  float run = __shfl_up(inc, 1);
  if (lane == 0) run = 1.f;
  for (int k = 0; k < per; ++k) {
    if (b + k < n) { const float t = buf[b + k]; buf[b + k] = run; run *= t; }
  }
}
// inclusive cumsum in place
__device__ __forceinline__ void incl_cumsum(float* buf, int n, int per, int lane) {
  float loc = 0.f;
  const int b = lane * per;
  for (int k = 0; k < per; ++k) if (b + k < n) loc += buf[b + k];
  float inc = wave_incl_scan_add(loc, lane);
  float run = __shfl_up(inc, 1);
  if (lane == 0) run = 0.f;
  for (int k = 0; k < per; ++k) {
    if (b + k < n) { run += buf[b + k]; buf[b + k] = run; }
  }
}

__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ z_in, const float* __restrict__ sdf_in,
                                                       int R, int n, int m, float inv_s, float* __restrict__ z_out,
                                                       float* __restrict__ sdf_out, float* __restrict__ z_new,
                                                       int* __restrict__ slot_new) {
  __shared__ float sz[RPB][MAXS], ss[RPB][MAXS], sa[RPB][MAXS], sc[RPB][MAXS], sn[RPB][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // (wavefronts past the last ray redo ray R - 1 without storing)
  const bool active = blockIdx.x * RPB + w < R;
  const int ray = active ? blockIdx.x * RPB + w : R - 1;
  float* Z = sz[w]; float* Sd = ss[w]; float* A = sa[w]; float* C = sc[w]; float* NZ = sn[w];
  const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
  const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
  for (int i = lane; i < n; i += 64) { Z[i] = z_in[(long)ray * n + i]; Sd[i] = sdf_in[(long)ray * n + i]; }
  WAVE_SYNC();
  const int nm1 = n - 1;
  // raw cos of every section (renderer.py:143)
  for (int i = lane; i < nm1; i += 64) C[i] = (Sd[i + 1] - Sd[i]) / (Z[i + 1] - Z[i] + 1e-5f);
  WAVE_SYNC();
  for (int i = lane; i < nm1; i += 64) {
    const float z0 = Z[i], z1 = Z[i + 1];
    const float px0 = ox + dx * z0, py0 = oy + dy * z0, pz0 = oz + dz * z0;
    const float px1 = ox + dx * z1, py1 = oy + dy * z1, pz1 = oz + dz * z1;
    const float r0 = sqrtf(px0 * px0 + py0 * py0 + pz0 * pz0), r1 = sqrtf(px1 * px1 + py1 * py1 + pz1 * pz1);
    const float inside = (r0 < 1.0f || r1 < 1.0f) ? 1.f : 0.f;
    const float prev = (i == 0) ? 0.f : C[i - 1];
    float cv = fminf(prev, C[i]);
    cv = fminf(fmaxf(cv, -1e3f), 0.f) * inside;
    const float mid = (Sd[i] + Sd[i + 1]) * 0.5f;
    const float dist = z1 - z0;
    const float pe = mid - cv * dist * 0.5f, ne = mid + cv * dist * 0.5f;
    const float pc = sigmoidf_(pe * inv_s), nc = sigmoidf_(ne * inv_s);
    A[i] = (pc - nc + 1e-5f) / (pc + 1e-5f);
  }
  WAVE_SYNC();
  // transmittance: exclusive cumprod of (1 - alpha + 1e-7)  -> reuse C
  for (int i = lane; i < nm1; i += 64) C[i] = 1.f - A[i] + 1e-7f;
  WAVE_SYNC();
  const int per = (nm1 + 63) / 64;
  excl_cumprod(C, nm1, per, lane);
  WAVE_SYNC();
  // weights + 1e-5, pdf, cdf (sample_pdf, renderer.py:42-45)
  float loc = 0.f;
  for (int i = lane; i < nm1; i += 64) { const float wv = A[i] * C[i] + 1e-5f; A[i] = wv; loc += wv; }
  const float tot = wave_sum(loc);
  WAVE_SYNC();
  for (int i = lane; i < nm1; i += 64) A[i] = A[i] / tot;
  WAVE_SYNC();
  incl_cumsum(A, nm1, per, lane);
  WAVE_SYNC();
  // cdf = [0, A[0..nm1-1]] has n entries; invert at the m deterministic u's (torch.linspace semantics)
  if (lane < m) {
    const float start = 0.5f / m, end = 1.f - 0.5f / m;
    const float step = (m > 1) ? (end - start) / (float)(m - 1) : 0.f;
    const float u = (lane < m / 2) ? start + step * lane : end - step * (m - 1 - lane);
    // searchsorted(cdf, u, right=True): number of cdf entries <= u
    int lo = 0, hi = n;
    while (lo < hi) {
      const int midi = (lo + hi) >> 1;
      const float cv = (midi == 0) ? 0.f : A[midi - 1];
      if (cv <= u) lo = midi + 1; else hi = midi;
    }
    const int below = max(lo - 1, 0), above = min(lo, n - 1);
    const float cb = (below == 0) ? 0.f : A[below - 1];
    const float ca = (above == 0) ? 0.f : A[above - 1];
    float denom = ca - cb;
    if (denom < 1e-5f) denom = 1.f;
    const float t = (u - cb) / denom;
    NZ[lane] = Z[below] + t * (Z[above] - Z[below]);
  }
  WAVE_SYNC();
  // merge (both lists ascending; ties keep the old sample first, like a stable sort of cat([z, new_z]))
  float* zo = z_out + (long)ray * (n + m);
  float* so = sdf_out + (long)ray * (n + m);
  for (int i = lane; i < n; i += 64) {
    const float zv = Z[i];
    int lo = 0, hi = m;  // count new < zv
    while (lo < hi) { const int k = (lo + hi) >> 1; if (NZ[k] < zv) lo = k + 1; else hi = k; }
    if (active) { zo[i + lo] = zv; so[i + lo] = Sd[i]; }
  }
  if (lane < m) {
    const float zv = NZ[lane];
    int lo = 0, hi = n;  // count old <= zv
    while (lo < hi) { const int k = (lo + hi) >> 1; if (Z[k] <= zv) lo = k + 1; else hi = k; }
    if (active) {
      zo[lane + lo] = zv;
      so[lane + lo] = 0.f;
      z_new[(long)ray * m + lane] = zv;
      slot_new[(long)ray * m + lane] = lane + lo;
    }
  }
}

// The same step with SEVERAL rays per wavefront: a group of GL lanes per ray (GL = 16 for n + m <= 64, the reference's 64-spp confs:
// n = 32 .. 56, m = 8 -- four rays per wavefront; GL = 32 for n + m <= 128, the 128-spp conf: two), a lane owns every GL-th sample in
// the elementwise phases and a contiguous chunk of <= 4 in the scans.  With one ray per wavefront most of a step's ~600 instructions
// are scan / search / synchronisation overhead executed for 32-56 live lanes (8 in the inversion); a group pays that overhead once for
// its ray and the wavefront carries four of them (profiles/r04_ab_kernels.txt).
template <int GL>
__device__ __forceinline__ float grp_incl_scan_mul(float v, int gl) {
#pragma unroll
  for (int d = 1; d < GL; d <<= 1) { const float o = __shfl_up(v, d, GL); if (gl >= d) v *= o; }
  return v;
}
template <int GL>
__device__ __forceinline__ float grp_incl_scan_add(float v, int gl) {
#pragma unroll
  for (int d = 1; d < GL; d <<= 1) { const float o = __shfl_up(v, d, GL); if (gl >= d) v += o; }
  return v;
}
template <int GL>
__device__ __forceinline__ float grp_sum(float v) {
#pragma unroll
  for (int d = GL / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
template <int GL>
__global__ __launch_bounds__(256) void upsample_grp_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const float* __restrict__ z_in, const float* __restrict__ sdf_in,
                                                         int R, int n, int m, float inv_s, float* __restrict__ z_out,
                                                         float* __restrict__ sdf_out, float* __restrict__ z_new,
                                                         int* __restrict__ slot_new) {
  constexpr int US = 4 * GL, RPBG = 256 / GL;      // samples a group can hold, rays per 256-thread block
  __shared__ float sz[RPBG][US], ss[RPBG][US], sa[RPBG][US], sc[RPBG][US], sn[RPBG][16];
  const int lane = threadIdx.x & 63, gl = lane & (GL - 1);
  const int slot = (threadIdx.x >> 6) * (64 / GL) + lane / GL;
  // (groups past the last ray redo ray R - 1 without storing)
  const bool active = (long)blockIdx.x * RPBG + slot < R;
  const int ray = active ? blockIdx.x * RPBG + slot : R - 1;
  float* Z = sz[slot]; float* Sd = ss[slot]; float* A = sa[slot]; float* C = sc[slot]; float* NZ = sn[slot];
  const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
  const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
  for (int i = gl; i < n; i += GL) { Z[i] = z_in[(long)ray * n + i]; Sd[i] = sdf_in[(long)ray * n + i]; }
  WAVE_SYNC();
  const int nm1 = n - 1;
  for (int i = gl; i < nm1; i += GL) C[i] = (Sd[i + 1] - Sd[i]) / (Z[i + 1] - Z[i] + 1e-5f);
  WAVE_SYNC();
  for (int i = gl; i < nm1; i += GL) {
    const float z0 = Z[i], z1 = Z[i + 1];
    const float px0 = ox + dx * z0, py0 = oy + dy * z0, pz0 = oz + dz * z0;
    const float px1 = ox + dx * z1, py1 = oy + dy * z1, pz1 = oz + dz * z1;
    const float r0 = sqrtf(px0 * px0 + py0 * py0 + pz0 * pz0), r1 = sqrtf(px1 * px1 + py1 * py1 + pz1 * pz1);
    const float inside = (r0 < 1.0f || r1 < 1.0f) ? 1.f : 0.f;
    const float prev = (i == 0) ? 0.f : C[i - 1];
    float cv = fminf(prev, C[i]);
    cv = fminf(fmaxf(cv, -1e3f), 0.f) * inside;
    const float mid = (Sd[i] + Sd[i + 1]) * 0.5f;
    const float dist = z1 - z0;
    const float pe = mid - cv * dist * 0.5f, ne = mid + cv * dist * 0.5f;
    const float pc = sigmoidf_(pe * inv_s), nc = sigmoidf_(ne * inv_s);
    A[i] = (pc - nc + 1e-5f) / (pc + 1e-5f);
  }
  WAVE_SYNC();
  // transmittance: exclusive cumprod of (1 - alpha + 1e-7) -> C; a lane owns the `per` consecutive entries from gl * per
  const int per = (nm1 + GL - 1) / GL;
  const int b = gl * per;
  {
    float loc = 1.f;
    for (int k = 0; k < per; ++k) if (b + k < nm1) loc *= 1.f - A[b + k] + 1e-7f;
    const float inc = grp_incl_scan_mul<GL>(loc, gl);
    float run = __shfl_up(inc, 1, GL);
    if (gl == 0) run = 1.f;
    for (int k = 0; k < per; ++k)
      if (b + k < nm1) { C[b + k] = run; run *= 1.f - A[b + k] + 1e-7f; }
  }
  WAVE_SYNC();
  // weights + 1e-5, pdf, cdf (sample_pdf, renderer.py:42-45)
  float loc = 0.f;
  for (int i = gl; i < nm1; i += GL) { const float wv = A[i] * C[i] + 1e-5f; A[i] = wv; loc += wv; }
  const float tot = grp_sum<GL>(loc);
  WAVE_SYNC();
  {
    float s = 0.f;
    for (int k = 0; k < per; ++k) if (b + k < nm1) s += A[b + k] / tot;
    const float inc = grp_incl_scan_add<GL>(s, gl);
    float run = __shfl_up(inc, 1, GL);
    if (gl == 0) run = 0.f;
    for (int k = 0; k < per; ++k)
      if (b + k < nm1) { run += A[b + k] / tot; A[b + k] = run; }
  }
  WAVE_SYNC();
  // cdf = [0, A[0..nm1-1]] has n entries; invert at the m deterministic u's (torch.linspace semantics)
  if (gl < m) {
    const float start = 0.5f / m, end = 1.f - 0.5f / m;
    const float step = (m > 1) ? (end - start) / (float)(m - 1) : 0.f;
    const float u = (gl < m / 2) ? start + step * gl : end - step * (m - 1 - gl);
    int lo = 0, hi = n;
    while (lo < hi) {
      const int midi = (lo + hi) >> 1;
      const float cv = (midi == 0) ? 0.f : A[midi - 1];
      if (cv <= u) lo = midi + 1; else hi = midi;
    }
    const int below = max(lo - 1, 0), above = min(lo, n - 1);
    const float cb = (below == 0) ? 0.f : A[below - 1];
    const float ca = (above == 0) ? 0.f : A[above - 1];
    float denom = ca - cb;
    if (denom < 1e-5f) denom = 1.f;
    const float t = (u - cb) / denom;
    NZ[gl] = Z[below] + t * (Z[above] - Z[below]);
  }
  WAVE_SYNC();
  // merge (both lists ascending; ties keep the old sample first, like a stable sort of cat([z, new_z]))
  float* zo = z_out + (long)ray * (n + m);
  float* so = sdf_out + (long)ray * (n + m);
  for (int i = gl; i < n; i += GL) {
    const float zv = Z[i];
    int lo = 0, hi = m;  // count new < zv
    while (lo < hi) { const int k = (lo + hi) >> 1; if (NZ[k] < zv) lo = k + 1; else hi = k; }
    if (active) { zo[i + lo] = zv; so[i + lo] = Sd[i]; }
  }
  if (gl < m) {
    const float zv = NZ[gl];
    int lo = 0, hi = n;  // count old <= zv
    while (lo < hi) { const int k = (lo + hi) >> 1; if (Z[k] <= zv) lo = k + 1; else hi = k; }
    if (active) {
      zo[gl + lo] = zv;
      so[gl + lo] = 0.f;
      z_new[(long)ray * m + gl] = zv;
      slot_new[(long)ray * m + gl] = gl + lo;
    }
  }
}

// lanes_per_ray: 0 = chosen from (n, m) -- 16 lanes per ray for n + m <= 64, 32 for n + m <= 128, a whole wavefront otherwise; 64 = one
// wavefront per ray for every n (the A/B partner and the cross-check of the tests).  The switch is an argument, not library state.
extern "C" int avc_upsample_step_lanes(const float* rays_o, const float* rays_d, const float* z_in, const float* sdf_in,
                                       int R, int n, int m, float inv_s, float* z_out, float* sdf_out, float* z_new,
                                       int* slot_new, int lanes_per_ray, void* stream) {
  if (n > MAXS || m > 64 || n + m > MAXS || n < 2) { avc_set_error("avc_upsample_step: need 2 <= n, n+m <= 256, m <= 64"); return 1; }
  if (lanes_per_ray != 0 && lanes_per_ray != 64) { avc_set_error("avc_upsample_step_lanes: lanes_per_ray is 0 (automatic) or 64"); return 1; }
  if (R <= 0) return 0;
  const bool grouped = lanes_per_ray == 0 && m <= 16;
  if (grouped && n + m <= 64)
    hipLaunchKernelGGL(upsample_grp_kernel<16>, dim3((R + 15) / 16), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, z_in, sdf_in, R, n, m,
                       inv_s, z_out, sdf_out, z_new, slot_new);
  else if (grouped && n + m <= 128)
    hipLaunchKernelGGL(upsample_grp_kernel<32>, dim3((R + 7) / 8), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, z_in, sdf_in, R, n, m,
                       inv_s, z_out, sdf_out, z_new, slot_new);
  else
    hipLaunchKernelGGL(upsample_kernel, dim3((R + RPB - 1) / RPB), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, z_in,
                       sdf_in, R, n, m, inv_s, z_out, sdf_out, z_new, slot_new);
  return avc_check_launch("avc_upsample_step");
}
extern "C" int avc_upsample_step(const float* rays_o, const float* rays_d, const float* z_in, const float* sdf_in,
                                 int R, int n, int m, float inv_s, float* z_out, float* sdf_out, float* z_new,
                                 int* slot_new, void* stream) {
  return avc_upsample_step_lanes(rays_o, rays_d, z_in, sdf_in, R, n, m, inv_s, z_out, sdf_out, z_new, slot_new, 0, stream);
}

// ------------------------------------------------------------------------------------------------------
// compositing
// ------------------------------------------------------------------------------------------------------
struct SampleVals {   // per-sample quantities of renderer.py:237-254
  float dist, c, ic, e_prev, e_next, P, Q, a_raw, alpha;
};
__device__ __forceinline__ SampleVals sample_eval(float sdf, float nx, float ny, float nz, float dx, float dy, float dz,
                                                  float dist, float inv_s, float car) {
  SampleVals v;
  v.dist = dist;
  v.c = dx * nx + dy * ny + dz * nz;
  v.ic = -(fmaxf(-v.c * 0.5f + 0.5f, 0.f) * (1.f - car) + fmaxf(-v.c, 0.f) * car);
  v.e_next = sdf + v.ic * dist * 0.5f;
  v.e_prev = sdf - v.ic * dist * 0.5f;
  v.P = sigmoidf_(v.e_prev * inv_s);
  v.Q = sigmoidf_(v.e_next * inv_s);
  v.a_raw = (v.P - v.Q + 1e-5f) / (v.P + 1e-5f);
  v.alpha = fminf(fmaxf(v.a_raw, 0.f), 1.f);
  return v;
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float* __restrict__ sdf, const float* __restrict__ normal, const float* __restrict__ rgb,
    const float* __restrict__ z, const float* __restrict__ rays_o, const float* __restrict__ rays_d, int R, int S,
    const float* __restrict__ inv_s_p, float sample_dist, float car, const float* __restrict__ bg, int bg_mode,
    float* __restrict__ color, float* __restrict__ extra, float* __restrict__ weights, float* __restrict__ cdf,
    float* __restrict__ mid_z, float* __restrict__ inside, float* __restrict__ eik, float* __restrict__ wstat,
    float* __restrict__ nsum) {
  __shared__ float sT[RPB][MAXS], sA[RPB][MAXS];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // (wavefronts past the last ray redo ray R - 1 without storing)
  const bool active = blockIdx.x * RPB + w < R;
  const int ray = active ? blockIdx.x * RPB + w : R - 1;
  float* T = sT[w]; float* A = sA[w];
  const float inv_s = inv_s_p[0];
  const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
  const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
  const long base = (long)ray * S;
  float e_num = 0.f, e_den = 0.f;
  for (int i = lane; i < S; i += 64) {
    const float zv = z[base + i];
    const float dist = (i + 1 < S) ? z[base + i + 1] - zv : sample_dist;
    const float mz = zv + dist * 0.5f;
    const float px = ox + dx * mz, py = oy + dy * mz, pz = oz + dz * mz;
    const float pn = sqrtf(px * px + py * py + pz * pz);
    const float nx = normal[3 * (base + i)], ny = normal[3 * (base + i) + 1], nz = normal[3 * (base + i) + 2];
    const SampleVals v = sample_eval(sdf[base + i], nx, ny, nz, dx, dy, dz, dist, inv_s, car);
    A[i] = v.alpha;
    T[i] = 1.f - v.alpha + 1e-7f;
    if (active) {
      cdf[base + i] = v.P;
      mid_z[base + i] = mz;
      inside[base + i] = pn < 1.0f ? 1.f : 0.f;
    }
    if (pn < 1.2f) {
      const float gn = sqrtf(nx * nx + ny * ny + nz * nz);
      e_num += (gn - 1.f) * (gn - 1.f);
      e_den += 1.f;
    }
  }
  WAVE_SYNC();
  excl_cumprod(T, S, (S + 63) / 64, lane);
  WAVE_SYNC();
  float c0 = 0, c1 = 0, c2 = 0, x0 = 0, x1 = 0, x2 = 0, ws = 0, wm = 0, n0 = 0, n1 = 0, n2 = 0;
  for (int i = lane; i < S; i += 64) {
    const float wv = A[i] * T[i];
    if (active) weights[base + i] = wv;
    const float* r = rgb + 6 * (base + i);
    c0 += wv * r[0]; c1 += wv * r[1]; c2 += wv * r[2];
    x0 += wv * r[3]; x1 += wv * r[4]; x2 += wv * r[5];
    ws += wv;
    wm = fmaxf(wm, wv);
    if (nsum) {   // sum_i w_i n_i: the shading normal of main.py:428 before its normalisation
      const float* nn = normal + 3 * (base + i);
      n0 += wv * nn[0]; n1 += wv * nn[1]; n2 += wv * nn[2];
    }
  }
  c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
  x0 = wave_sum(x0); x1 = wave_sum(x1); x2 = wave_sum(x2);
  ws = wave_sum(ws); e_num = wave_sum(e_num); e_den = wave_sum(e_den);
  if (wstat) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
  }
  if (nsum) { n0 = wave_sum(n0); n1 = wave_sum(n1); n2 = wave_sum(n2); }
  if (lane == 0 && active) {
    if (wstat) { wstat[ray] = ws; wstat[(long)R + ray] = wm; }    // planar [2][R]: both rows are the [R,1] tensors render() returns
    if (nsum) { nsum[3 * ray] = n0; nsum[3 * ray + 1] = n1; nsum[3 * ray + 2] = n2; }
    float b0 = 0, b1 = 0, b2 = 0;
    if (bg_mode == 1) { b0 = bg[0]; b1 = bg[1]; b2 = bg[2]; }
    else if (bg_mode == 2) { b0 = b1 = b2 = bg[ray]; }
    color[3 * ray] = c0; color[3 * ray + 1] = c1; color[3 * ray + 2] = c2;
    extra[3 * ray] = x0 + b0 * (1.f - ws); extra[3 * ray + 1] = x1 + b1 * (1.f - ws); extra[3 * ray + 2] = x2 + b2 * (1.f - ws);
    eik[2 * ray] = e_num; eik[2 * ray + 1] = e_den;
  }
}

extern "C" int avc_composite_fwd(const float* sdf, const float* normal, const float* rgb, const float* z,
                                 const float* rays_o, const float* rays_d, int R, int S, const float* inv_s,
                                 float sample_dist, float cos_anneal, const float* bg, int bg_mode, float* color,
                                 float* extra, float* weights, float* cdf, float* mid_z, float* inside, float* eik,
                                 float* wstat, float* nsum, void* stream) {
  if (S > MAXS || S < 1) { avc_set_error("avc_composite_fwd: 1 <= S <= 256"); return 1; }
  if (R <= 0) return 0;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((R + RPB - 1) / RPB), dim3(256), 0, (hipStream_t)stream, sdf, normal, rgb,
                     z, rays_o, rays_d, R, S, inv_s, sample_dist, cos_anneal, bg, bg_mode, color, extra, weights, cdf,
                     mid_z, inside, eik, wstat, nsum);
  return avc_check_launch("avc_composite_fwd");
}

// reverse scan: Ssuf_{i} = sum_{j>i} wbar_j alpha_j prod_{i<k<j} t_k  via the affine recurrence
// S_{i-1} = wbar_i alpha_i + t_i S_i.  Composition of affine maps x -> a + t x is associative, so a
// wave-level suffix scan over (a, t) pairs gives every S_i.
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float* __restrict__ sdf, const float* __restrict__ normal, const float* __restrict__ rgb,
    const float* __restrict__ z, const float* __restrict__ rays_o, const float* __restrict__ rays_d, int R, int S,
    const float* __restrict__ inv_s_p, float sample_dist, float car, const float* __restrict__ bg, int bg_mode,
    const float* __restrict__ d_color, const float* __restrict__ d_extra, const float* __restrict__ d_weights,
    const float* __restrict__ d_normal_up, const float* __restrict__ d_wsum, const float* __restrict__ d_nsum,
    const float* __restrict__ eik_scale_p, float* __restrict__ d_sdf,
    float* __restrict__ d_normal, float* __restrict__ d_rgb, float* __restrict__ d_inv_s) {
  __shared__ float sT[RPB][MAXS], sA[RPB][MAXS], sW[RPB][MAXS], sS[RPB][MAXS];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // (wavefronts past the last ray redo ray R - 1 without storing)
  const bool active = blockIdx.x * RPB + w < R;
  const int ray = active ? blockIdx.x * RPB + w : R - 1;
  float* T = sT[w]; float* A = sA[w]; float* WB = sW[w]; float* SS = sS[w];
  const float inv_s = inv_s_p[0];
  const float eik_scale = eik_scale_p[0];
  const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
  const float dx = rays_d[3 * ray], dy = rays_d[3 * ray + 1], dz = rays_d[3 * ray + 2];
  const long base = (long)ray * S;
  float b0 = 0, b1 = 0, b2 = 0;
  if (bg_mode == 1) { b0 = bg[0]; b1 = bg[1]; b2 = bg[2]; }
  else if (bg_mode == 2) { b0 = b1 = b2 = bg[ray]; }
  const float dc0 = d_color[3 * ray], dc1 = d_color[3 * ray + 1], dc2 = d_color[3 * ray + 2];
  const float de0 = d_extra[3 * ray], de1 = d_extra[3 * ray + 1], de2 = d_extra[3 * ray + 2];
  // upstream gradients of the per-ray reductions of the weights (sum_i w_i and sum_i w_i n_i: avc_composite_fwd's wstat / nsum)
  const float dws = d_wsum ? d_wsum[ray] : 0.f;
  const float dn0 = d_nsum ? d_nsum[3 * ray] : 0.f, dn1 = d_nsum ? d_nsum[3 * ray + 1] : 0.f, dn2 = d_nsum ? d_nsum[3 * ray + 2] : 0.f;
  for (int i = lane; i < S; i += 64) {
    const float zv = z[base + i];
    const float dist = (i + 1 < S) ? z[base + i + 1] - zv : sample_dist;
    const float nx = normal[3 * (base + i)], ny = normal[3 * (base + i) + 1], nz = normal[3 * (base + i) + 2];
    const SampleVals v = sample_eval(sdf[base + i], nx, ny, nz, dx, dy, dz, dist, inv_s, car);
    A[i] = v.alpha;
    T[i] = 1.f - v.alpha + 1e-7f;
    const float* r = rgb + 6 * (base + i);
    WB[i] = dc0 * r[0] + dc1 * r[1] + dc2 * r[2] + de0 * (r[3] - b0) + de1 * (r[4] - b1) + de2 * (r[5] - b2) +
            (d_weights ? d_weights[base + i] : 0.f) + dws + dn0 * nx + dn1 * ny + dn2 * nz;
  }
  WAVE_SYNC();
  const int per = (S + 63) / 64;
  // suffix scan of affine maps.  Lane owns chunk [b, b+per); local composite (a_loc, t_loc) maps S_{end} -> S_{b-1}
  {
    const int b = lane * per;
    float a_loc = 0.f, t_loc = 1.f;
    for (int k = per - 1; k >= 0; --k) {
      const int i = b + k;
      if (i < S) { a_loc = WB[i] * A[i] + T[i] * a_loc; t_loc = T[i] * t_loc; }
    }
    // the true map of the chunk applied to incoming x is a_loc' + t_loc x, where a_loc was built with x = 0: ok
    // inclusive suffix scan across lanes: (a,t)_lane := compose(chunk_lane, (a,t)_{lane+1})
    float a = a_loc, t = t_loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float a2 = __shfl_down(a, d), t2 = __shfl_down(t, d);
      if (lane + d < 64) { a = a + t * a2; t = t * t2; }
    }
    // value entering this chunk from the right = S at index (b+per-1) = result of all chunks to the right applied to 0
    float run = __shfl_down(a, 1);
    if (lane == 63) run = 0.f;
    for (int k = per - 1; k >= 0; --k) {
      const int i = b + k;
      if (i < S) { SS[i] = run; run = WB[i] * A[i] + T[i] * run; }
    }
  }
  WAVE_SYNC();
  excl_cumprod(T, S, per, lane);   // T now holds the transmittance
  WAVE_SYNC();
  float dinv = 0.f;
  for (int i = lane; i < S; i += 64) {
    const float zv = z[base + i];
    const float dist = (i + 1 < S) ? z[base + i + 1] - zv : sample_dist;
    const float mz = zv + dist * 0.5f;
    const float px = ox + dx * mz, py = oy + dy * mz, pz = oz + dz * mz;
    const float pn = sqrtf(px * px + py * py + pz * pz);
    const float nx = normal[3 * (base + i)], ny = normal[3 * (base + i) + 1], nz = normal[3 * (base + i) + 2];
    const SampleVals v = sample_eval(sdf[base + i], nx, ny, nz, dx, dy, dz, dist, inv_s, car);
    const float Ti = T[i];
    const float wv = v.alpha * Ti;
    if (!active) continue;
    float* dr = d_rgb + 6 * (base + i);
    dr[0] = wv * dc0; dr[1] = wv * dc1; dr[2] = wv * dc2; dr[3] = wv * de0; dr[4] = wv * de1; dr[5] = wv * de2;
    const float abar = Ti * (WB[i] - SS[i]);
    const float d_araw = (v.a_raw >= 0.f && v.a_raw <= 1.f) ? abar : 0.f;
    const float pp = v.P + 1e-5f;
    const float dP = d_araw * v.Q / (pp * pp);
    const float dQ = -d_araw / pp;
    const float gP = dP * v.P * (1.f - v.P), gQ = dQ * v.Q * (1.f - v.Q);
    const float de_prev = gP * inv_s, de_next = gQ * inv_s;
    dinv += gP * v.e_prev + gQ * v.e_next;
    d_sdf[base + i] = de_prev + de_next;
    const float d_ic = (de_next - de_prev) * dist * 0.5f;
    const float d_c = d_ic * (0.5f * (1.f - car) * ((-0.5f * v.c + 0.5f) > 0.f ? 1.f : 0.f) + car * ((-v.c) > 0.f ? 1.f : 0.f));
    float gx = d_c * dx, gy = d_c * dy, gz = d_c * dz;
    if (d_normal_up) { gx += d_normal_up[3 * (base + i)]; gy += d_normal_up[3 * (base + i) + 1]; gz += d_normal_up[3 * (base + i) + 2]; }
    gx += wv * dn0; gy += wv * dn1; gz += wv * dn2;
    if (pn < 1.2f) {
      const float gn = sqrtf(nx * nx + ny * ny + nz * nz);
      const float k = eik_scale * 2.f * (gn - 1.f) / gn;
      gx += k * nx; gy += k * ny; gz += k * nz;
    }
    d_normal[3 * (base + i)] = gx; d_normal[3 * (base + i) + 1] = gy; d_normal[3 * (base + i) + 2] = gz;
  }
  dinv = wave_sum(dinv);
  if (lane == 0 && active) d_inv_s[ray] = dinv;
}

extern "C" int avc_composite_bwd(const float* sdf, const float* normal, const float* rgb, const float* z,
                                 const float* rays_o, const float* rays_d, int R, int S, const float* inv_s,
                                 float sample_dist, float cos_anneal, const float* bg, int bg_mode,
                                 const float* d_color, const float* d_extra, const float* d_weights,
                                 const float* d_normal_up, const float* d_wsum, const float* d_nsum, const float* eik_scale,
                                 float* d_sdf, float* d_normal, float* d_rgb, float* d_inv_s, void* stream) {
  if (S > MAXS || S < 1) { avc_set_error("avc_composite_bwd: 1 <= S <= 256"); return 1; }
  if (R <= 0) return 0;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((R + RPB - 1) / RPB), dim3(256), 0, (hipStream_t)stream, sdf, normal, rgb,
                     z, rays_o, rays_d, R, S, inv_s, sample_dist, cos_anneal, bg, bg_mode, d_color, d_extra, d_weights,
                     d_normal_up, d_wsum, d_nsum, eik_scale, d_sdf, d_normal, d_rgb, d_inv_s);
  return avc_check_launch("avc_composite_bwd");
}

// Column sums of x [R,C] (C <= 4) in a fixed order -- up to CS_BLOCKS workgroups each add a contiguous range of rows (thread t: rows
// t, t + 1024, ..; lanes; waves), the one that finishes last (ticket) adds the per-workgroup sums in workgroup order -- the scalar
// reductions around the compositing kernels without a torch reduction + its bookkeeping launches each:
//   mode 0: out[c] = sum_r x[r][c]
//   mode 1 (renderer.py:283-285, the eikonal term from avc_composite_fwd's eik [R,2]): out[1] = sum x[:,1] + 1e-5, out[0] = sum x[:,0] / out[1]
// scratch: CS_BLOCKS x 4 floats + one zero-initialised word that the kernel leaves zero.
#define CS_BLOCKS 64
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, long R, int C, int mode, float* __restrict__ out,
                                                      float* __restrict__ part, unsigned* __restrict__ ticket) {
  __shared__ float red[4][16];
  __shared__ bool last;
  const long per = (R + gridDim.x - 1) / gridDim.x, r0 = per * blockIdx.x, r1 = min(R, r0 + per);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (long r = r0 + threadIdx.x; r < r1; r += 1024)
    for (int c = 0; c < C; ++c) acc[c] += x[r * C + c];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c = 0; c < C; ++c) {
    float v = acc[c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) red[c][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int c = 0; c < C; ++c) {
      float v = 0.f;
      for (int w = 0; w < 16; ++w) v += red[c][w];
      part[4 * blockIdx.x + c] = v;
    }
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last || threadIdx.x != 0) return;
  __threadfence();
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < (int)gridDim.x; ++b)
    for (int c = 0; c < C; ++c) s[c] += part[4 * b + c];
  if (mode == 1) { const float den = s[1] + 1e-5f; out[1] = den; out[0] = s[0] / den; }
  else for (int c = 0; c < C; ++c) out[c] = s[c];
  *ticket = 0u;
}
extern "C" long avc_colsum_scratch_bytes() { return (CS_BLOCKS * 4 + 1) * 4; }
extern "C" int avc_colsum(const float* x, long R, int C, int mode, float* out, void* scratch, void* stream) {
  if (C < 1 || C > 4 || (mode == 1 && C != 2) || R < 0 || (!x && R > 0) || !out || !scratch) {    // (R == 0: the sums of nothing, x may be NULL)
    avc_set_error("avc_colsum: 1 <= C <= 4 (mode 1: C == 2), buffers");
    return 1;
  }
  const int nb = (int)max(1L, min((long)CS_BLOCKS, (R + 4095) / 4096));
  float* part = (float*)scratch;
  hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, x, R, C, mode, out, part, (unsigned*)(part + CS_BLOCKS * 4));
  return avc_check_launch("avc_colsum");
}
// inv_s = exp(10 variance).clip(1e-6, 1e6) (fields.py:275-276 + renderer.py:234) and its backward: g == NULL: out[0] = inv_s, out[1] = 1 / inv_s;
// else out[0] = g[0] * d inv_s / d variance (10 exp(10 v) inside the clip range, ends included, else 0)
__global__ void inv_s_kernel(const float* __restrict__ variance, const float* __restrict__ g, float* __restrict__ out) {
  const float e = expf(variance[0] * 10.0f);
  if (!g) { const float v = fminf(fmaxf(e, 1e-6f), 1e6f); out[0] = v; out[1] = 1.0f / v; }      // (out[1]: the logged s_val, renderer.py:288)
  else out[0] = (e >= 1e-6f && e <= 1e6f) ? g[0] * (e * 10.0f) : 0.f;
}
extern "C" int avc_inv_s(const float* variance, const float* g, float* out, void* stream) {
  if (!variance || !out) { avc_set_error("avc_inv_s: NULL buffer"); return 1; }
  hipLaunchKernelGGL(inv_s_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, variance, g, out);
  return avc_check_launch("avc_inv_s");
}
