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
// second launch turns the keys into the faces' light values and puts the all-ones "empty" key back.  A face whose box holds more than
// RS_LARGE pixels (a close-up, a face cut by the near plane: one wavefront would walk up to the whole image) is only listed; a
// tile-parallel launch in between gives every 16 x 16 pixel tile a scan over that (short) list -- the first version's scheme, which
// is the right one for exactly those faces.  (The first version of this
// file gave every 16 x 16 pixel tile a scan over ALL faces with an LDS survivor list: 0.37 ms for the 27 552 faces of the prior at
// 512^2, bound by the few tiles over the head and the hands where a thousand small faces survive the box test; staging the faces
// through LDS or scanning 1 024 per round did not move that.  profiles/r04_ab_kernels.txt)
#include "avc_common.h"
#include "../../include/avc.h"

#pragma clang fp contract(off)   // same roundings as the fp32 restatement (edge tests are sign tests)

#define RS_EMPTY 0xFFFFFFFFFFFFFFFFull
#define RS_LARGE 1024     // pixels in a face's box from which it goes to the tile-parallel pass
#define RS_TILE 16

// depth of face (x0..z2) at pixel (xi, yi), or a negative number if the pixel centre is outside / the depth out of range
struct FaceEq {
  float x0, y0, z0, x1, y1, z1, x2, y2, z2;
  float p0x, p0y, p1x, p1y, p2x, p2y, den;
};
__device__ __forceinline__ bool face_setup(const float* __restrict__ f, int is, FaceEq& e) {
  e.x0 = f[0]; e.y0 = f[1]; e.z0 = f[2]; e.x1 = f[3]; e.y1 = f[4]; e.z1 = f[5]; e.x2 = f[6]; e.y2 = f[7]; e.z2 = f[8];
  if ((e.y2 - e.y0) * (e.x1 - e.x0) < (e.y1 - e.y0) * (e.x2 - e.x0)) return false;          // back-facing
  // pixel-space vertices and the inverse of their homogeneous matrix (rasterize_cuda_kernel.cu, kernel 1)
  e.p0x = 0.5f * (e.x0 * is + is - 1); e.p0y = 0.5f * (e.y0 * is + is - 1);
  e.p1x = 0.5f * (e.x1 * is + is - 1); e.p1y = 0.5f * (e.y1 * is + is - 1);
  e.p2x = 0.5f * (e.x2 * is + is - 1); e.p2y = 0.5f * (e.y2 * is + is - 1);
  e.den = e.p2x * (e.p0y - e.p1y) + e.p0x * (e.p1y - e.p2y) + e.p1x * (e.p2y - e.p0y);
  return e.den != 0.f;
}
__device__ __forceinline__ float face_depth(const FaceEq& e, int xi, int yi, int is, float near, float far) {
  const float xp = (2.f * xi + 1.f - is) / is;
  const float yp = (2.f * yi + 1.f - is) / is;
  if (((yp - e.y0) * (e.x1 - e.x0) < (xp - e.x0) * (e.y1 - e.y0)) || ((yp - e.y1) * (e.x2 - e.x1) < (xp - e.x1) * (e.y2 - e.y1)) ||
      ((yp - e.y2) * (e.x0 - e.x2) < (xp - e.x2) * (e.y0 - e.y2)))
    return -1.f;
  float w0 = ((e.p1y - e.p2y) * xi + (e.p2x - e.p1x) * yi + (e.p1x * e.p2y - e.p2x * e.p1y)) / e.den;
  float w1 = ((e.p2y - e.p0y) * xi + (e.p0x - e.p2x) * yi + (e.p2x * e.p0y - e.p0x * e.p2y)) / e.den;
  float w2 = ((e.p0y - e.p1y) * xi + (e.p1x - e.p0x) * yi + (e.p0x * e.p1y - e.p1x * e.p0y)) / e.den;
  w0 = fminf(fmaxf(w0, 0.f), 1.f); w1 = fminf(fmaxf(w1, 0.f), 1.f); w2 = fminf(fmaxf(w2, 0.f), 1.f);
  const float ws = fmaxf(w0 + w1 + w2, 1e-10f);
  const float zp = 1.f / ((w0 / e.z0 + w1 / e.z1 + w2 / e.z2) / ws);
  return (zp > near && zp < far) ? zp : -1.f;        // (zp > near >= 0: its bit pattern orders like its value)
}
// the face's box in pixel indices, one pixel of slack each way (the edge functions decide); false: off screen (or NaN)
__device__ __forceinline__ bool face_box(const FaceEq& e, int is, int& xa, int& xb, int& ya, int& yb) {
  const float xl = fminf(e.x0, fminf(e.x1, e.x2)), xh = fmaxf(e.x0, fmaxf(e.x1, e.x2));
  const float yl = fminf(e.y0, fminf(e.y1, e.y2)), yh = fmaxf(e.y0, fmaxf(e.y1, e.y2));
  if (!(xh >= -1.f && xl <= 1.f && yh >= -1.f && yl <= 1.f)) return false;
  // (clamped in float first: a vertex near the camera plane projects to 1e30, which no int holds)
  xa = max(0, (int)floorf(fmaxf(0.5f * (xl * is + is - 1), -1.f)) - 1); xb = min(is - 1, (int)ceilf(fminf(0.5f * (xh * is + is - 1), (float)is)) + 1);
  ya = max(0, (int)floorf(fmaxf(0.5f * (yl * is + is - 1), -1.f)) - 1); yb = min(is - 1, (int)ceilf(fminf(0.5f * (yh * is + is - 1), (float)is)) + 1);
  return xb >= xa && yb >= ya;
}

// the nine floats of face fn: from faces [F,9], or (idx != NULL) gathered from the projected vertices faces = ndc [V,3] through idx [F,3]
__device__ __forceinline__ void load_face(const float* __restrict__ faces, const int* __restrict__ idx, int fn, float (&f)[9]) {
  if (idx) {
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const long vi = idx[3 * (long)fn + v];
#pragma unroll
      for (int k = 0; k < 3; ++k) f[3 * v + k] = faces[3 * vi + k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = faces[(long)fn * 9 + k];
  }
}
// neural_renderer's look + perspective (look.py, perspective.py; models/utils.py:108-125): v_cam = (v - eye) . (x, y, z axes),
// ndc = (x / z / width, y / z / width, z), (0, 0, 0) for z <= 0; cam = device [12]: eye, x axis, y axis, z axis
__global__ __launch_bounds__(256) void prior_project_kernel(const float* __restrict__ vw, int V, const float* __restrict__ cam, float width,
                                                            float* __restrict__ ndc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= V) return;
  const float d0 = vw[3 * i] - cam[0], d1 = vw[3 * i + 1] - cam[1], d2 = vw[3 * i + 2] - cam[2];
  float c[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) c[j] = fmaf(d2, cam[3 + 3 * j + 2], fmaf(d1, cam[3 + 3 * j + 1], d0 * cam[3 + 3 * j]));
  const bool behind = c[2] <= 0.f;      // the patch the reference's README.md:126-134 prescribes for perspective.py: behind the camera -> (0, 0, 0)
  ndc[3 * i] = behind ? 0.f : c[0] / c[2] / width;
  ndc[3 * i + 1] = behind ? 0.f : c[1] / c[2] / width;
  ndc[3 * i + 2] = behind ? 0.f : c[2];
}
// `large` = [count - 1 (0xFFFFFFFF = none), face indices ...]
__global__ __launch_bounds__(256) void raster_faces_kernel(const float* __restrict__ faces /* [F,9] x,y (NDC), z (depth) */,
                                                           const int* __restrict__ idx, int F, int is,
                                                           float near, float far, unsigned long long* __restrict__ zbuf /* [is,is], y up */,
                                                           unsigned* __restrict__ large) {
  const int fn = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (fn >= F) return;
  float f[9];
  load_face(faces, idx, fn, f);
  FaceEq e;
  if (!face_setup(f, is, e)) return;
  int xa, xb, ya, yb;
  if (!face_box(e, is, xa, xb, ya, yb)) return;
  const int w = xb - xa + 1, h = yb - ya + 1;
  const int n = w * h;
  if (n > RS_LARGE) {
    if (lane == 0) large[1 + (atomicAdd(&large[0], 1u) + 1u)] = (unsigned)fn;
    return;
  }
  for (int idx = lane; idx < n; idx += 64) {
    const int xi = xa + idx % w, yi = ya + idx / w;
    const float zp = face_depth(e, xi, yi, is, near, far);
    if (zp < 0.f) continue;
    atomicMin(&zbuf[(long)yi * is + xi], ((unsigned long long)__float_as_uint(zp) << 32) | (unsigned)fn);
  }
}
// the listed large faces, tile-parallel: thread = pixel of a 16 x 16 tile, every face of the list whose box meets the tile is
// evaluated at the tile's pixels; the running minimum joins the key the small faces left (plain read-modify-write: one thread per pixel)
__global__ __launch_bounds__(256) void raster_large_kernel(const float* __restrict__ faces, const int* __restrict__ idx, int is, float near,
                                                           float far, unsigned long long* __restrict__ zbuf, const unsigned* __restrict__ large) {
  const unsigned nl = large[0] + 1u;
  if (nl == 0u) return;
  const int tx0 = blockIdx.x * RS_TILE, ty0 = blockIdx.y * RS_TILE;
  const int xi = tx0 + (threadIdx.x & 15), yi = ty0 + (threadIdx.x >> 4);
  unsigned long long best = RS_EMPTY;
  for (unsigned q = 0; q < nl; ++q) {
    const int fn = (int)large[1 + q];
    float f[9];
    load_face(faces, idx, fn, f);
    FaceEq e;
    if (!face_setup(f, is, e)) continue;
    int xa, xb, ya, yb;
    if (!face_box(e, is, xa, xb, ya, yb)) continue;
    if (xb < tx0 || xa > tx0 + RS_TILE - 1 || yb < ty0 || ya > ty0 + RS_TILE - 1) continue;     // (uniform over the workgroup)
    if (xi >= is || yi >= is) continue;
    const float zp = face_depth(e, xi, yi, is, near, far);
    if (zp < 0.f) continue;
    const unsigned long long key = ((unsigned long long)__float_as_uint(zp) << 32) | (unsigned)fn;
    best = key < best ? key : best;
  }
  if (xi < is && yi < is && best != RS_EMPTY) {
    unsigned long long* z = &zbuf[(long)yi * is + xi];
    if (best < *z) *z = best;
  }
}
// image = light of the winning face (0: background), rows flipped (rasterize.py: y up -> row 0 = top); the scratch is left empty
__global__ __launch_bounds__(256) void raster_resolve_kernel(unsigned long long* __restrict__ zbuf, const float* __restrict__ light, int is,
                                                             float* __restrict__ image, unsigned* __restrict__ large) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p == 0) large[0] = 0xFFFFFFFFu;
  if (p >= is * is) return;
  const unsigned long long key = zbuf[p];
  const int yi = p / is, xi = p % is;
  image[(long)(is - 1 - yi) * is + xi] = key == RS_EMPTY ? 0.f : light[(unsigned)(key & 0xFFFFFFFFull)];
  zbuf[p] = RS_EMPTY;
}

// the same + what models/utils.py:108-125 does next, for the 2 x super-sampled render (anti_aliasing): 2 x 2 average (avg_pool2d: the window
// summed row by row, then / 4), optionally the x flip of models/utils.py:124 (`[:, ::-1]`) and the white texture's three equal channels.
// out [S,S] (channels == 1) or [S,S,3], S = is / 2.
__global__ __launch_bounds__(256) void raster_resolve_pool_kernel(unsigned long long* __restrict__ zbuf, const float* __restrict__ light, int is,
                                                                  float* __restrict__ out, int flip_x, int channels, unsigned* __restrict__ large) {
  const int S = is >> 1;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p == 0) large[0] = 0xFFFFFFFFu;
  if (p >= S * S) return;
  const int y = p / S, x = p % S;
  float acc = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int r = 2 * y + dy, c = 2 * x + dx;                       // image row r (0 = top) is z-buffer row is - 1 - r
      unsigned long long* z = &zbuf[(long)(is - 1 - r) * is + c];
      const unsigned long long key = *z;
      acc += key == RS_EMPTY ? 0.f : light[(unsigned)(key & 0xFFFFFFFFull)];
      *z = RS_EMPTY;
    }
  const float v = acc / 4.f;
  const int xo = flip_x ? S - 1 - x : x;
  for (int ch = 0; ch < channels; ++ch) out[((long)y * S + xo) * channels + ch] = v;
}

extern "C" long avc_rasterize_scratch_bytes(int F, int image_size) {
  return (long)image_size * image_size * 8 + ((long)F + 2) * 4;
}
extern "C" int avc_rasterize_faces(const float* faces, const float* light, int F, int image_size, float near, float far,
                                   float* image, void* scratch, void* stream) {
  if (image_size <= 0) { avc_set_error("avc_rasterize_faces: image_size <= 0"); return 1; }
  if (F < 0 || near < 0.f) { avc_set_error("avc_rasterize_faces: F < 0 or near < 0"); return 1; }
  if (!image || !scratch || (F && (!faces || !light))) { avc_set_error("avc_rasterize_faces: NULL buffer"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* zbuf = (unsigned long long*)scratch;
  unsigned* large = (unsigned*)(zbuf + (long)image_size * image_size);
  if (F) {
    hipLaunchKernelGGL(raster_faces_kernel, dim3((F + 3) / 4), dim3(256), 0, s, faces, (const int*)nullptr, F, image_size, near, far, zbuf, large);
    const int nt = (image_size + RS_TILE - 1) / RS_TILE;
    hipLaunchKernelGGL(raster_large_kernel, dim3(nt, nt), dim3(256), 0, s, faces, (const int*)nullptr, image_size, near, far, zbuf, large);
  }
  hipLaunchKernelGGL(raster_resolve_kernel, dim3((image_size * image_size + 255) / 256), dim3(256), 0, s, zbuf, light, image_size, image, large);
  return avc_check_launch("avc_rasterize_faces");
}
// The whole prior render of models/utils.py:108-125 from the world-space mesh: projection of the V vertices (cam = device [12]: eye + the
// look frame's x, y, z axes; width = tan(viewing angle)), the rasteriser above on faces gathered through idx [F,3] (fill_back copies
// included) at the 2 x super-sampled size 2 S, and the 2 x 2 average (+ x flip, + 3 equal channels) -> out [S,S(,3)].  ndc: [V,3] scratch.
extern "C" int avc_rasterize_mesh(const float* v_world, int V, const int* idx, int F, const float* cam, float width, const float* light,
                                  int S, float near, float far, float* ndc, float* out, int flip_x, int channels, void* scratch, void* stream) {
  if (S <= 0 || V <= 0 || F < 0 || near < 0.f || (channels != 1 && channels != 3)) { avc_set_error("avc_rasterize_mesh: bad sizes"); return 1; }
  if (!v_world || !cam || !ndc || !out || !scratch || (F && (!idx || !light))) { avc_set_error("avc_rasterize_mesh: NULL buffer"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  const int is = 2 * S;
  unsigned long long* zbuf = (unsigned long long*)scratch;
  unsigned* large = (unsigned*)(zbuf + (long)is * is);
  hipLaunchKernelGGL(prior_project_kernel, dim3((V + 255) / 256), dim3(256), 0, s, v_world, V, cam, width, ndc);
  if (F) {
    hipLaunchKernelGGL(raster_faces_kernel, dim3((F + 3) / 4), dim3(256), 0, s, ndc, idx, F, is, near, far, zbuf, large);
    const int nt = (is + RS_TILE - 1) / RS_TILE;
    hipLaunchKernelGGL(raster_large_kernel, dim3(nt, nt), dim3(256), 0, s, ndc, idx, is, near, far, zbuf, large);
  }
  hipLaunchKernelGGL(raster_resolve_pool_kernel, dim3((S * S + 255) / 256), dim3(256), 0, s, zbuf, light, is, out, flip_x, channels, large);
  return avc_check_launch("avc_rasterize_mesh");
}
