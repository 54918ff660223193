// Silhouette + flat-shading rasteriser of the SMPL prior: the forward pass of
// `neural_renderer` as AppearanceGen uses it (AvatarGen/AppearanceGen/models/utils.py:108-125, render_one_batch: white
// texture, ambient 0.5 + directional 0.5 face lighting, anti-aliased 256 x 256) -- no host round trip, no neural_renderer.
// Algorithm = the published one of neural_renderer's rasterize_cuda_kernel.cu (restated in oracle/nr_oracle.py, which this
// kernel is tested against): per pixel of the is x is super-sampled grid, nearest front-facing face whose three edge functions
// contain the pixel centre; perspective-correct 1/z from clamped barycentric weights; strict z test (ties -> lower face index).
//
// Index / compare work, no MFMA: F <= ~30k faces x 9 floats, is^2 <= 512^2 pixels, and a face of the posed SMPL body covers a handful
// of pixels.  Face-parallel with a z-buffer of 64-bit keys: one wavefront per face walks the pixels of the face's bounding box (lanes
// = pixels, so a large face is 64-wide too), evaluates the edge functions and the depth exactly as above and does ONE
// atomicMin(zbuf[pixel], depth_bits << 32 | face) per covered pixel -- depths are positive floats (near < z), so the integer order
// of the key is (depth, face index): nearest face, ties to the lower index, independent of the order the atomics arrive in.  A
// second launch turns the keys into the faces' light values and puts the all-ones "empty" key back.  (The first version of this
// file gave every 16 x 16 pixel tile a scan over ALL faces with an LDS survivor list: 0.37 ms for the 27 552 faces of the prior at
// 512^2, bound by the few tiles over the head and the hands where a thousand small faces survive the box test; staging the faces
// through LDS or scanning 1 024 per round did not move that.  profiles/r04_ab_kernels.txt)
#include "avc_common.h"
#include "../../include/avc.h"

#pragma clang fp contract(off)   // same roundings as the fp32 restatement (edge tests are sign tests)

#define RS_EMPTY 0xFFFFFFFFFFFFFFFFull

__global__ __launch_bounds__(256) void raster_faces_kernel(const float* __restrict__ faces /* [F,9] x,y (NDC), z (depth) */, int F, int is,
                                                           float near, float far, unsigned long long* __restrict__ zbuf /* [is,is], y up */) {
  const int fn = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (fn >= F) return;
  float f[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) f[k] = faces[(long)fn * 9 + k];
  const float x0 = f[0], y0 = f[1], z0 = f[2], x1 = f[3], y1 = f[4], z1 = f[5], x2 = f[6], y2 = f[7], z2 = f[8];
  if ((y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0)) return;          // back-facing
  const float xl = fminf(x0, fminf(x1, x2)), xh = fmaxf(x0, fmaxf(x1, x2));
  const float yl = fminf(y0, fminf(y1, y2)), yh = fmaxf(y0, fmaxf(y1, y2));
  if (!(xh >= -1.f && xl <= 1.f && yh >= -1.f && yl <= 1.f)) return;  // off screen (or NaN)
  // pixel-space vertices and the inverse of their homogeneous matrix (rasterize_cuda_kernel.cu, kernel 1)
  const float p0x = 0.5f * (x0 * is + is - 1), p0y = 0.5f * (y0 * is + is - 1);
  const float p1x = 0.5f * (x1 * is + is - 1), p1y = 0.5f * (y1 * is + is - 1);
  const float p2x = 0.5f * (x2 * is + is - 1), p2y = 0.5f * (y2 * is + is - 1);
  const float den = p2x * (p0y - p1y) + p0x * (p1y - p2y) + p1x * (p2y - p0y);
  if (den == 0.f) return;
  // pixels whose centres can lie inside: the box in pixel coordinates, one pixel of slack each way (the edge functions decide)
  const int xa = max(0, (int)floorf(0.5f * (xl * is + is - 1)) - 1), xb = min(is - 1, (int)ceilf(0.5f * (xh * is + is - 1)) + 1);
  const int ya = max(0, (int)floorf(0.5f * (yl * is + is - 1)) - 1), yb = min(is - 1, (int)ceilf(0.5f * (yh * is + is - 1)) + 1);
  const int w = xb - xa + 1, h = yb - ya + 1;
  if (w <= 0 || h <= 0) return;
  const long n = (long)w * h;
  for (long idx = lane; idx < n; idx += 64) {
    const int xi = xa + (int)(idx % w), yi = ya + (int)(idx / w);
    const float xp = (2.f * xi + 1.f - is) / is;
    const float yp = (2.f * yi + 1.f - is) / is;
    if (((yp - y0) * (x1 - x0) < (xp - x0) * (y1 - y0)) || ((yp - y1) * (x2 - x1) < (xp - x1) * (y2 - y1)) ||
        ((yp - y2) * (x0 - x2) < (xp - x2) * (y0 - y2)))
      continue;
    float w0 = ((p1y - p2y) * xi + (p2x - p1x) * yi + (p1x * p2y - p2x * p1y)) / den;
    float w1 = ((p2y - p0y) * xi + (p0x - p2x) * yi + (p2x * p0y - p0x * p2y)) / den;
    float w2 = ((p0y - p1y) * xi + (p1x - p0x) * yi + (p0x * p1y - p1x * p0y)) / den;
    w0 = fminf(fmaxf(w0, 0.f), 1.f); w1 = fminf(fmaxf(w1, 0.f), 1.f); w2 = fminf(fmaxf(w2, 0.f), 1.f);
    const float ws = fmaxf(w0 + w1 + w2, 1e-10f);
    const float zp = 1.f / ((w0 / z0 + w1 / z1 + w2 / z2) / ws);
    if (!(zp > near && zp < far)) continue;        // (zp > near >= 0: its bit pattern orders like its value)
    const unsigned long long key = ((unsigned long long)__float_as_uint(zp) << 32) | (unsigned)fn;
    atomicMin(&zbuf[(long)yi * is + xi], key);
  }
}
// image = light of the winning face (0: background), rows flipped (rasterize.py: y up -> row 0 = top); the z-buffer is left empty
__global__ __launch_bounds__(256) void raster_resolve_kernel(unsigned long long* __restrict__ zbuf, const float* __restrict__ light, int is,
                                                             float* __restrict__ image) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= is * is) return;
  const unsigned long long key = zbuf[p];
  const int yi = p / is, xi = p % is;
  image[(long)(is - 1 - yi) * is + xi] = key == RS_EMPTY ? 0.f : light[(unsigned)(key & 0xFFFFFFFFull)];
  zbuf[p] = RS_EMPTY;
}

extern "C" int avc_rasterize_faces(const float* faces, const float* light, int F, int image_size, float near, float far,
                                   float* image, void* zbuf, void* stream) {
  if (image_size <= 0) { avc_set_error("avc_rasterize_faces: image_size <= 0"); return 1; }
  if (F < 0 || near < 0.f) { avc_set_error("avc_rasterize_faces: F < 0 or near < 0"); return 1; }
  if (!image || !zbuf || (F && (!faces || !light))) { avc_set_error("avc_rasterize_faces: NULL buffer"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  if (F) hipLaunchKernelGGL(raster_faces_kernel, dim3((F + 3) / 4), dim3(256), 0, s, faces, F, image_size, near, far, (unsigned long long*)zbuf);
  hipLaunchKernelGGL(raster_resolve_kernel, dim3((image_size * image_size + 255) / 256), dim3(256), 0, s, (unsigned long long*)zbuf, light,
                     image_size, image);
  return avc_check_launch("avc_rasterize_faces");
}
