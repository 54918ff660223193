// Silhouette + flat-shading rasteriser of the SMPL prior: the forward pass of
// `neural_renderer` as AppearanceGen uses it (AvatarGen/AppearanceGen/models/utils.py:108-125, render_one_batch: white
// texture, ambient 0.5 + directional 0.5 face lighting, anti-aliased 256 x 256) -- no host round trip, no neural_renderer.
// Algorithm = the published one of neural_renderer's rasterize_cuda_kernel.cu (restated in oracle/nr_oracle.py, which this
// kernel is tested against): per pixel of the is x is super-sampled grid, nearest front-facing face whose three edge functions
// contain the pixel centre; perspective-correct 1/z from clamped barycentric weights; strict z test (ties -> lower face index).
//
// HBM-bound index/byte work, no MFMA: F <= ~30k faces x 9 floats, is^2 <= 512^2 pixels.  One 256-thread workgroup per
// 16 x 16 pixel tile: the faces are scanned in chunks of 256 (one bounding-box test per thread), survivors are compacted
// into LDS with a wave ballot + prefix count, then every pixel of the tile walks the short survivor list.
#include "avc_common.h"
#include "../../include/avc.h"

#pragma clang fp contract(off)   // same roundings as the fp32 restatement (edge tests are sign tests)

#define RS_TILE 16
#define RS_CHUNK 256

__global__ __launch_bounds__(256) void raster_kernel(const float* __restrict__ faces /* [F,9] x,y (NDC), z (depth) */,
                                                     const float* __restrict__ light /* [F] */, int F, int is, float near,
                                                     float far, float* __restrict__ image /* [is,is], row 0 = top */) {
  __shared__ float sf[RS_CHUNK][10];
  __shared__ int scount;
  __shared__ int wbase[4];
  const int tid = threadIdx.x;
  const int tx0 = blockIdx.x * RS_TILE, ty0 = blockIdx.y * RS_TILE;
  const int xi = tx0 + (tid & 15), yi = ty0 + (tid >> 4);
  const float xp = (2.f * xi + 1.f - is) / is;
  const float yp = (2.f * yi + 1.f - is) / is;
  // tile extent in NDC (pixel centres)
  const float txl = (2.f * tx0 + 1.f - is) / is, txh = (2.f * (tx0 + RS_TILE - 1) + 1.f - is) / is;
  const float tyl = (2.f * ty0 + 1.f - is) / is, tyh = (2.f * (ty0 + RS_TILE - 1) + 1.f - is) / is;
  float depth_min = far;
  float val = 0.f;
  int best = -1;
  for (int c0 = 0; c0 < F; c0 += RS_CHUNK) {
    const int fn = c0 + tid;
    bool keep = false;
    float f[9];
    if (fn < F) {
#pragma unroll
      for (int k = 0; k < 9; ++k) f[k] = faces[(long)fn * 9 + k];
      const bool back = (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
      const float xl = fminf(f[0], fminf(f[3], f[6])), xh = fmaxf(f[0], fmaxf(f[3], f[6]));
      const float yl = fminf(f[1], fminf(f[4], f[7])), yh = fmaxf(f[1], fmaxf(f[4], f[7]));
      keep = !back && xh >= txl && xl <= txh && yh >= tyl && yl <= tyh;
    }
    __syncthreads();                       // previous chunk's list fully consumed
    const unsigned long long m = __ballot(keep);
    const int wv = tid >> 6, ln = tid & 63;
    if (ln == 0) wbase[wv] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += wbase[w];
    if (tid == 0) scount = wbase[0] + wbase[1] + wbase[2] + wbase[3];
    if (keep) {
      const int pos = base + __popcll(m & ((1ull << ln) - 1ull));   // chunk order == face order: deterministic z ties
#pragma unroll
      for (int k = 0; k < 9; ++k) sf[pos][k] = f[k];
      sf[pos][9] = light[fn];
    }
    __syncthreads();
    const int n = scount;
    for (int q = 0; q < n; ++q) {
      const float x0 = sf[q][0], y0 = sf[q][1], z0 = sf[q][2], x1 = sf[q][3], y1 = sf[q][4], z1 = sf[q][5], x2 = sf[q][6],
                  y2 = sf[q][7], z2 = sf[q][8];
      if (((yp - y0) * (x1 - x0) < (xp - x0) * (y1 - y0)) || ((yp - y1) * (x2 - x1) < (xp - x1) * (y2 - y1)) ||
          ((yp - y2) * (x0 - x2) < (xp - x2) * (y0 - y2)))
        continue;
      // pixel-space vertices and the inverse of their homogeneous matrix (rasterize_cuda_kernel.cu, kernel 1)
      const float p0x = 0.5f * (x0 * is + is - 1), p0y = 0.5f * (y0 * is + is - 1);
      const float p1x = 0.5f * (x1 * is + is - 1), p1y = 0.5f * (y1 * is + is - 1);
      const float p2x = 0.5f * (x2 * is + is - 1), p2y = 0.5f * (y2 * is + is - 1);
      const float den = p2x * (p0y - p1y) + p0x * (p1y - p2y) + p1x * (p2y - p0y);
      if (den == 0.f) continue;
      float w0 = ((p1y - p2y) * xi + (p2x - p1x) * yi + (p1x * p2y - p2x * p1y)) / den;
      float w1 = ((p2y - p0y) * xi + (p0x - p2x) * yi + (p2x * p0y - p0x * p2y)) / den;
      float w2 = ((p0y - p1y) * xi + (p1x - p0x) * yi + (p0x * p1y - p1x * p0y)) / den;
      w0 = fminf(fmaxf(w0, 0.f), 1.f); w1 = fminf(fmaxf(w1, 0.f), 1.f); w2 = fminf(fmaxf(w2, 0.f), 1.f);
      const float ws = fmaxf(w0 + w1 + w2, 1e-10f);
      const float zp = 1.f / ((w0 / z0 + w1 / z1 + w2 / z2) / ws);
      if (zp <= near || far <= zp) continue;
      if (zp < depth_min) { depth_min = zp; val = sf[q][9]; best = q; }
    }
  }
  (void)best;
  if (xi < is && yi < is) image[(long)(is - 1 - yi) * is + xi] = val;   // rasterize.py flips the rows (y up -> row 0 = top)
}

extern "C" int avc_rasterize_faces(const float* faces, const float* light, int F, int image_size, float near, float far,
                                   float* image, void* stream) {
  if (image_size <= 0 || (image_size % RS_TILE)) { avc_set_error("avc_rasterize_faces: image_size must be a multiple of 16"); return 1; }
  if (F < 0) { avc_set_error("avc_rasterize_faces: F < 0"); return 1; }
  hipLaunchKernelGGL(raster_kernel, dim3(image_size / RS_TILE, image_size / RS_TILE), dim3(256), 0, (hipStream_t)stream, faces, light,
                     F, image_size, near, far, image);
  return avc_check_launch("avc_rasterize_faces");
}
