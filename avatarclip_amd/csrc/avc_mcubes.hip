// Iso-surface extraction of Runner.validate_mesh (main.py:850-919 -> renderer.py:28-36 -> mcubes.marching_cubes):
// marching cubes over the resolution^3 field u = -sdf with welded vertices (one vertex per sign-changing grid edge),
// entirely on the device.  HBM-bound integer/float streaming work: one thread per grid point, x fastest... the field is
// [nx][ny][nz] row-major as the reference builds it (renderer.py:16-24), so consecutive threads walk z.
//   pass 1 (avc_mc_classify): per grid point the 3 edge flags it owns (+x, +y, +z) and the triangle count of its cell
//   host: two exclusive scans (torch.cumsum)
//   pass 2 (avc_mc_emit):     vertices at the linear zero crossings, triangles through the generated case table
//                              (avatarclip_amd/mc_tables.py), in the order oracle/mcubes_oracle.py documents.
#include "avc_common.h"
#include "../../include/avc.h"

__device__ __forceinline__ int mc_case(const float* __restrict__ u, long p, int ny, int nz, float iso) {
  const long sy = nz, sx = (long)ny * nz;
  int m = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const long q = p + (c & 1) * sx + ((c >> 1) & 1) * sy + ((c >> 2) & 1);
    m |= (u[q] > iso ? 1 : 0) << c;
  }
  return m;
}

__global__ __launch_bounds__(256) void mc_classify_kernel(const float* __restrict__ u, int nx, int ny, int nz, float iso,
                                                          const int* __restrict__ ntri_table, int* __restrict__ vflag,
                                                          int* __restrict__ ccount) {
  const long n = (long)nx * ny * nz;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int k = (int)(p % nz), j = (int)((p / nz) % ny), i = (int)(p / ((long)ny * nz));
  const bool in0 = u[p] > iso;
  const long sy = nz, sx = (long)ny * nz;
  vflag[3 * p + 0] = (i + 1 < nx && (u[p + sx] > iso) != in0) ? 1 : 0;
  vflag[3 * p + 1] = (j + 1 < ny && (u[p + sy] > iso) != in0) ? 1 : 0;
  vflag[3 * p + 2] = (k + 1 < nz && (u[p + 1] > iso) != in0) ? 1 : 0;
  int cnt = 0;
  if (i + 1 < nx && j + 1 < ny && k + 1 < nz) cnt = ntri_table[mc_case(u, p, ny, nz, iso)];
  ccount[p] = cnt;
}

__global__ __launch_bounds__(256) void mc_emit_kernel(const float* __restrict__ u, int nx, int ny, int nz, float iso,
                                                      const int* __restrict__ vflag, const int* __restrict__ vid,
                                                      const int* __restrict__ ccount, const int* __restrict__ coff,
                                                      const signed char* __restrict__ tri_table,
                                                      const int* __restrict__ edge_table, float* __restrict__ verts,
                                                      int* __restrict__ tris) {
  const long n = (long)nx * ny * nz;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int k = (int)(p % nz), j = (int)((p / nz) % ny), i = (int)(p / ((long)ny * nz));
  const long sy = nz, sx = (long)ny * nz;
  const float u0 = u[p];
  const long strides[3] = {sx, sy, 1};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (vflag[3 * p + a]) {
      const float u1 = u[p + strides[a]];
      const float t = (iso - u0) / (u1 - u0);   // mcubes: linear interpolation along the edge
      float pos[3] = {(float)i, (float)j, (float)k};
      pos[a] += t;
      float* v = verts + 3L * vid[3 * p + a];
      v[0] = pos[0]; v[1] = pos[1]; v[2] = pos[2];
    }
  }
  const int cnt = ccount[p];
  if (cnt > 0) {
    const int m = mc_case(u, p, ny, nz, iso);
    int* out = tris + 3L * coff[p];
    for (int q = 0; q < cnt; ++q) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int e = tri_table[(m * 5 + q) * 3 + c];
        const int* ed = edge_table + 4 * e;
        const long pe = p + ed[0] * sx + ed[1] * sy + ed[2];
        out[3 * q + c] = vid[3 * pe + ed[3]];
      }
    }
  }
}

extern "C" int avc_mc_classify(const float* u, int nx, int ny, int nz, float iso, const int* ntri_table, int* vflag,
                               int* ccount, void* stream) {
  const long n = (long)nx * ny * nz;
  if (n <= 0) return 0;
  if (nx < 2 || ny < 2 || nz < 2) { avc_set_error("avc_mc_classify: the grid needs at least 2 points per axis"); return 1; }
  const long blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffL) { avc_set_error("avc_mc_classify: grid too large"); return 1; }
  hipLaunchKernelGGL(mc_classify_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, u, nx, ny, nz, iso,
                     ntri_table, vflag, ccount);
  return avc_check_launch("avc_mc_classify");
}

extern "C" int avc_mc_emit(const float* u, int nx, int ny, int nz, float iso, const int* vflag, const int* vid,
                           const int* ccount, const int* coff, const signed char* tri_table, const int* edge_table,
                           float* verts, int* tris, void* stream) {
  const long n = (long)nx * ny * nz;
  if (n <= 0) return 0;
  const long blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffL) { avc_set_error("avc_mc_emit: grid too large"); return 1; }
  hipLaunchKernelGGL(mc_emit_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, u, nx, ny, nz, iso, vflag, vid,
                     ccount, coff, tri_table, edge_table, verts, tris);
  return avc_check_launch("avc_mc_emit");
}
