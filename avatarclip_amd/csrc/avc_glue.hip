// The per-pixel glue of one train_clip iteration between the renderer and CLIP, fused (AvatarGen/AppearanceGen/main.py:426-534):
//   avc_shade_loss_fwd / _bwd   random-light Lambert shading of the rendered normals (main.py:426-453), the scatter of the silhouette
//                               rays back into full images over the augmentation background (:461-487), and the per-pixel terms of the
//                               colour L1 (:491-492) and mask BCE (:497) losses -- ~85 small torch launches forward and ~100 backward
//                               become one kernel each way.
//   avc_resize_norm_fwd / _bwd  CLIP's preprocessing of the two images (:510-511,516-517): bilinear resize to 224 x 224 (F.interpolate,
//                               align_corners = False, no antialias) + Normalize, [B,H,W,3] -> [B,3,224,224].
// fp32 throughout; the formulas are the reference's line by line (tests/test_gpu_glue.py compares values and gradients with the torch
// statement of the same lines, Runner.shade_and_scatter / assemble_loss, which tests/test_glue_golden.py pins against the reference).
// One thread per pixel of the (small) image; HBM / latency work, no matrix core.
#include "avc_common.h"
#include "../../include/avc.h"

#define GLUE_THREADS 256

struct GlueIn {
  const float* color;      // [R,3] color_fine
  const float* extra;      // [R,3] extra_color_fine
  const float* wsum;       // [R]   weight_sum
  const float* nsum;       // [R,3] sum_i w_i n_i (main.py:428 before its normalisation); NULL: no shading
  const float* true_rgb;   // [P,3]
  const float* mask;       // [P]   (already thresholded / all ones: what the losses use)
  const int* ray_of_pixel; // [P] ray index of a pixel or -1; NULL: pixel p = ray p (full-frame mode)
  const float* bg;         // [P] grey background of the CLIP images outside the silhouette, or NULL: bg_const
  const float* light;      // device [4]: unit light direction, ambience
  float bg_const;
  int P;
  int img0_is_extra;       // image 0 = extra_color (texture_cast_light off) instead of texture_shading
};

// per pixel: everything the forward produces, recomputed in the backward (a dozen flops)
struct Shade {
  float tex[3], rsh[3];   // texture_shading, rand_shading_rgb
  float s, sp;            // rand_shading before / after the background rule
  float dot, rho;         // n^ . l^ (before the clamp), |N|
  bool bgm, dnan;
};
__device__ __forceinline__ Shade shade_pixel(const GlueIn& g, int r) {
  Shade o;
  const float E[3] = {g.extra[3 * r], g.extra[3 * r + 1], g.extra[3 * r + 2]};
  if (!g.nsum) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { o.tex[c] = E[c]; o.rsh[c] = E[c]; }
    o.s = o.sp = 1.f; o.dot = 0.f; o.rho = 0.f; o.bgm = true; o.dnan = false;
    return o;
  }
  const float N[3] = {g.nsum[3 * r], g.nsum[3 * r + 1], g.nsum[3 * r + 2]};
  o.rho = sqrtf(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
  const float inv = 1.f / (o.rho + 1e-7f);
  o.dot = (N[0] * inv) * g.light[0] + (N[1] * inv) * g.light[1] + (N[2] * inv) * g.light[2];
  float d = fminf(fmaxf(o.dot, 0.f), 1.f);
  o.dnan = isnan(o.dot);
  if (o.dnan) d = 1.f;                                   // main.py:437 (nan -> 1)
  const float a = g.light[3];
  o.s = a + (1.f - a) * d;
  o.bgm = g.wsum[r] < 0.5f;                              // main.py:444
  o.sp = o.bgm ? 1.f : o.s;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o.rsh[c] = o.bgm ? E[c] : o.s;
    o.tex[c] = fminf(fmaxf(E[c] * o.sp, 0.f), 1.f);
  }
  return o;
}

// images [2][P][3]: 0 = texture_shading (or extra_color when there is no shading), 1 = rand_shading_rgb;
// partial [gridDim.x][4] = per-block sums of (|color - true| mask, mask, BCE term, (color - true)^2 mask); the block that finishes
// last (ticket: a zero-initialised device word that it resets) adds the partials up in a fixed order into sums[4]
__global__ __launch_bounds__(GLUE_THREADS) void shade_loss_fwd_kernel(GlueIn g, float* __restrict__ images, float* __restrict__ partial,
                                                                      float* __restrict__ sums, unsigned* __restrict__ ticket) {
  const int p = blockIdx.x * GLUE_THREADS + threadIdx.x;
  float l1 = 0.f, ms = 0.f, bce = 0.f, sq = 0.f;
  if (p < g.P) {
    const int r = g.ray_of_pixel ? g.ray_of_pixel[p] : p;
    float C[3] = {0.f, 0.f, 0.f}, ws = 0.f;
    float i0[3], i1[3];
    if (r >= 0) {
      const Shade s = shade_pixel(g, r);
#pragma unroll
      for (int c = 0; c < 3; ++c) { C[c] = g.color[3 * r + c]; i0[c] = g.img0_is_extra ? g.extra[3 * r + c] : s.tex[c]; i1[c] = s.rsh[c]; }
      ws = g.wsum[r];
    } else {
      const float b = g.bg ? g.bg[p] : g.bg_const;
#pragma unroll
      for (int c = 0; c < 3; ++c) { i0[c] = b; i1[c] = b; }
    }
    const float m = g.mask[p];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      images[(long)p * 3 + c] = i0[c];
      images[((long)g.P + p) * 3 + c] = i1[c];
      const float e = (C[c] - g.true_rgb[3 * p + c]);
      l1 += fabsf(e * m);
      sq += e * e * m;
    }
    ms = m;
    const float x = fminf(fmaxf(ws, 1e-3f), 1.f - 1e-3f);
    bce = -(m * fmaxf(logf(x), -100.f) + (1.f - m) * fmaxf(logf(1.f - x), -100.f));   // F.binary_cross_entropy (log clamped at -100)
  }
  __shared__ float red[4][GLUE_THREADS / 64];
  __shared__ bool last;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { l1 += __shfl_xor(l1, d); ms += __shfl_xor(ms, d); bce += __shfl_xor(bce, d); sq += __shfl_xor(sq, d); }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = l1; red[1][wv] = ms; red[2][wv] = bce; red[3][wv] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
    for (int w = 0; w < GLUE_THREADS / 64; ++w) { a += red[0][w]; b += red[1][w]; c += red[2][w]; d += red[3][w]; }
    partial[4 * blockIdx.x + 0] = a; partial[4 * blockIdx.x + 1] = b; partial[4 * blockIdx.x + 2] = c; partial[4 * blockIdx.x + 3] = d;
    __threadfence();                                        // the partials are visible device-wide before the ticket moves
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // fixed order: thread t adds the blocks t, t + 256, ...; then the lanes of a wave, then the four waves
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < (int)gridDim.x; b += GLUE_THREADS) {
    const f4 v = reinterpret_cast<const f4*>(partial)[b];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc[k] += __shfl_xor(acc[k], d);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][wv] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.f;
    for (int w = 0; w < GLUE_THREADS / 64; ++w) v += red[threadIdx.x][w];
    sums[threadIdx.x] = v;
  }
  if (threadIdx.x == 0) *ticket = 0u;
}

// gs (device [4], the gradient of the forward's sums) : [0] = d loss / d (l1 sum), [2] = d loss / d (bce sum); dimages [2][P][3] (either half may be NULL: that image is not used)
__global__ __launch_bounds__(GLUE_THREADS) void shade_loss_bwd_kernel(GlueIn g, const float* __restrict__ dimg0, const float* __restrict__ dimg1,
                                                                      const float* __restrict__ gs, float* __restrict__ dcolor,
                                                                      float* __restrict__ dextra, float* __restrict__ dwsum, float* __restrict__ dnsum) {
  const int p = blockIdx.x * GLUE_THREADS + threadIdx.x;
  if (p >= g.P) return;
  const int r = g.ray_of_pixel ? g.ray_of_pixel[p] : p;
  if (r < 0) return;
  const float m = g.mask[p], gl1 = gs[0], gbce = gs[2];
  // colour L1: d |e| = sign(e), e = (C - T) m
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float e = (g.color[3 * r + c] - g.true_rgb[3 * p + c]) * m;
    dcolor[3 * r + c] = gl1 * m * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
  }
  // mask BCE through the clip (gradient passes inside [1e-3, 1 - 1e-3], ends included)
  const float ws = g.wsum[r];
  float dws = 0.f;
  if (ws >= 1e-3f && ws <= 1.f - 1e-3f) {
    // (the -100 clamp of the logarithms is inactive inside the clip range)
    dws = gbce * (-m / ws + (1.f - m) / (1.f - ws));
  }
  dwsum[r] = dws;
  const Shade s = shade_pixel(g, r);
  float dE[3] = {0.f, 0.f, 0.f}, ds = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float E = g.extra[3 * r + c];
    const float g0 = dimg0 ? dimg0[(long)p * 3 + c] : 0.f, g1 = dimg1 ? dimg1[(long)p * 3 + c] : 0.f;
    if (!g.nsum) { dE[c] = g0 + g1; continue; }
    if (g.img0_is_extra) {
      dE[c] += g0;
    } else {
      const float t = E * s.sp;
      const float gt = (t >= 0.f && t <= 1.f) ? g0 : 0.f;   // clamp(min=0, max=1)
      dE[c] += gt * s.sp;
      if (!s.bgm) ds += gt * E;                              // s' = s on the body, the constant 1 on the background
    }
    if (s.bgm) dE[c] += g1; else ds += g1;                   // rand_shading_rgb = extra on the background, s (x3) on the body
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) dextra[3 * r + c] = dE[c];
  if (g.nsum) {
    float dN[3] = {0.f, 0.f, 0.f};
    const float dd = (1.f - g.light[3]) * ds;
    if (!s.dnan && s.dot >= 0.f && s.dot <= 1.f && dd != 0.f) {
      // dot = n^ . l^, n^ = N / (rho + eps):  dN = dn^ / (rho + eps) - N (N . dn^) / (rho (rho + eps)^2), dn^ = dd l^
      const float N[3] = {g.nsum[3 * r], g.nsum[3 * r + 1], g.nsum[3 * r + 2]};
      const float re = s.rho + 1e-7f;
      const float ndl = (N[0] * g.light[0] + N[1] * g.light[1] + N[2] * g.light[2]) * dd;
      const float k = s.rho > 0.f ? ndl / (s.rho * re * re) : 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) dN[c] = dd * g.light[c] / re - N[c] * k;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dnsum[3 * r + c] = dN[c];
  }
}

extern "C" int avc_shade_loss_blocks(int P) { return (P + GLUE_THREADS - 1) / GLUE_THREADS; }

static GlueIn glue_in(const float* color, const float* extra, const float* wsum, const float* nsum, const float* true_rgb, const float* mask,
                      const int* ray_of_pixel, const float* bg, float bg_const, const float* light, int P, int img0_is_extra) {
  GlueIn g;
  g.color = color; g.extra = extra; g.wsum = wsum; g.nsum = nsum; g.true_rgb = true_rgb; g.mask = mask; g.ray_of_pixel = ray_of_pixel;
  g.bg = bg; g.light = light; g.bg_const = bg_const; g.P = P; g.img0_is_extra = img0_is_extra;
  return g;
}

extern "C" int avc_shade_loss_fwd(const float* color, const float* extra, const float* wsum, const float* nsum, const float* true_rgb,
                                  const float* mask, const int* ray_of_pixel, const float* bg, float bg_const, const float* light, int P,
                                  int img0_is_extra, float* images, float* partial, float* sums, unsigned* ticket, void* stream) {
  if (P <= 0) return 0;
  if (!color || !extra || !wsum || !true_rgb || !mask || !images || !partial || !sums || !ticket || (nsum && !light)) {
    avc_set_error("avc_shade_loss_fwd: NULL buffer");
    return 1;
  }
  hipLaunchKernelGGL(shade_loss_fwd_kernel, dim3(avc_shade_loss_blocks(P)), dim3(GLUE_THREADS), 0, (hipStream_t)stream,
                     glue_in(color, extra, wsum, nsum, true_rgb, mask, ray_of_pixel, bg, bg_const, light, P, img0_is_extra), images, partial,
                     sums, ticket);
  return avc_check_launch("avc_shade_loss_fwd");
}

extern "C" int avc_shade_loss_bwd(const float* color, const float* extra, const float* wsum, const float* nsum, const float* true_rgb,
                                  const float* mask, const int* ray_of_pixel, const float* light, int P, int img0_is_extra,
                                  const float* dimg0, const float* dimg1, const float* gs, float* dcolor, float* dextra, float* dwsum,
                                  float* dnsum, void* stream) {
  if (P <= 0) return 0;
  if (!color || !extra || !wsum || !true_rgb || !mask || !gs || !dcolor || !dextra || !dwsum || (nsum && (!light || !dnsum))) {
    avc_set_error("avc_shade_loss_bwd: NULL buffer");
    return 1;
  }
  hipLaunchKernelGGL(shade_loss_bwd_kernel, dim3(avc_shade_loss_blocks(P)), dim3(GLUE_THREADS), 0, (hipStream_t)stream,
                     glue_in(color, extra, wsum, nsum, true_rgb, mask, ray_of_pixel, nullptr, 0.f, light, P, img0_is_extra), dimg0, dimg1, gs,
                     dcolor, dextra, dwsum, dnsum);
  return avc_check_launch("avc_shade_loss_bwd");
}

// ---------------------------------------------------------------------------------------------------------------
// F.interpolate(mode='bilinear', align_corners=False) source coordinates: src = scale (dst + 0.5) - 0.5, clamped at 0
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp lerp_of(int dst, int in_size, float scale) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lerp l;
  l.i0 = (int)src;
  if (l.i0 > in_size - 1) l.i0 = in_size - 1;
  l.i1 = l.i0 + (l.i0 < in_size - 1 ? 1 : 0);
  l.w1 = src - (float)l.i0;
  l.w0 = 1.f - l.w1;
  return l;
}
#define CLIP_RES 224
struct Norm3 { float mean[3], istd[3]; };

// in [B,H,W,3] -> out [B,3,224,224] = (resize(in) - mean) / std
__global__ __launch_bounds__(GLUE_THREADS) void resize_norm_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                                       Norm3 nm) {
  const int idx = blockIdx.x * GLUE_THREADS + threadIdx.x;
  if (idx >= B * CLIP_RES * CLIP_RES) return;
  const int x = idx % CLIP_RES, y = (idx / CLIP_RES) % CLIP_RES, b = idx / (CLIP_RES * CLIP_RES);
  const float* ib = in + (long)b * H * W * 3;
  float v[3];
  if (H == CLIP_RES && W == CLIP_RES) {
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = ib[((long)y * W + x) * 3 + c];
  } else {
    const Lerp ly = lerp_of(y, H, (float)H / CLIP_RES), lx = lerp_of(x, W, (float)W / CLIP_RES);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = ib[((long)ly.i0 * W + lx.i0) * 3 + c], bb = ib[((long)ly.i0 * W + lx.i1) * 3 + c];
      const float cc = ib[((long)ly.i1 * W + lx.i0) * 3 + c], d = ib[((long)ly.i1 * W + lx.i1) * 3 + c];
      v[c] = ly.w0 * (lx.w0 * a + lx.w1 * bb) + ly.w1 * (lx.w0 * cc + lx.w1 * d);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) out[(((long)b * 3 + c) * CLIP_RES + y) * CLIP_RES + x] = (v[c] - nm.mean[c]) * nm.istd[c];
}
// din [B,H,W,3] = the transpose of the map above applied to dout [B,3,224,224], as a GATHER: one thread per input pixel adds up, in a
// fixed order, the output pixels whose two-tap stencil touches it.  (Rounds 3-4 scattered with atomicAdd: correct, but the order of
// the float additions -- hence the last bits of every pixel gradient, hence after Adam's sign-sensitive first steps the 5th digit of the
// next losses -- changed from run to run; an iteration is now bit-reproducible, which is what lets "two replicas side by side equal
// their solo runs" be asserted, tests/test_gpu_parallel.py.)
// weight with which output index o feeds input index i (0 if it does not); both taps can land on the last input index
__device__ __forceinline__ float tap_weight(int o, int i, int in_size, float scale) {
  const Lerp l = lerp_of(o, in_size, scale);
  return (l.i0 == i ? l.w0 : 0.f) + (l.i1 == i ? l.w1 : 0.f);
}
// output indices whose stencil can touch input index i: src = scale (o + 0.5) - 0.5 in (i - 1, i + 1), one index of margin either way
__device__ __forceinline__ void tap_range(int i, int in_size, float scale, int& lo, int& hi) {
  const float inv = 1.f / scale;
  lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
  if (i == 0 || lo < 0) lo = 0;
  if (i == in_size - 1 || hi > CLIP_RES - 1) hi = CLIP_RES - 1;
}
__global__ __launch_bounds__(GLUE_THREADS) void resize_norm_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int H, int W,
                                                                       Norm3 nm) {
  const long idx = (long)blockIdx.x * GLUE_THREADS + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), b = (int)(idx / ((long)W * H));
  const float* ob = dout + (long)b * 3 * CLIP_RES * CLIP_RES;
  float acc[3] = {0.f, 0.f, 0.f};
  if (H == CLIP_RES && W == CLIP_RES) {
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = ob[((long)c * CLIP_RES + y) * CLIP_RES + x];
  } else {
    const float sy = (float)H / CLIP_RES, sx = (float)W / CLIP_RES;
    int y0, y1, x0, x1;
    tap_range(y, H, sy, y0, y1);
    tap_range(x, W, sx, x0, x1);
    for (int oy = y0; oy <= y1; ++oy) {
      const float wy = tap_weight(oy, y, H, sy);
      if (wy == 0.f) continue;
      float row[3] = {0.f, 0.f, 0.f};
      for (int ox = x0; ox <= x1; ++ox) {
        const float wx = tap_weight(ox, x, W, sx);
        if (wx == 0.f) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) row[c] += wx * ob[((long)c * CLIP_RES + oy) * CLIP_RES + ox];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += wy * row[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) din[(((long)b * H + y) * W + x) * 3 + c] = acc[c] * nm.istd[c];
}

static Norm3 norm3(const float* mean, const float* stdv) {
  Norm3 n;
  for (int c = 0; c < 3; ++c) { n.mean[c] = mean[c]; n.istd[c] = 1.f / stdv[c]; }
  return n;
}
extern "C" int avc_resize_norm_fwd(const float* images, int B, int H, int W, const float* mean /* host [3] */, const float* stdv /* host [3] */,
                                   float* out, void* stream) {
  if (B <= 0) return 0;
  if (!images || !out || !mean || !stdv || H < 1 || W < 1) { avc_set_error("avc_resize_norm_fwd: bad arguments"); return 1; }
  const int n = B * CLIP_RES * CLIP_RES;
  hipLaunchKernelGGL(resize_norm_fwd_kernel, dim3((n + GLUE_THREADS - 1) / GLUE_THREADS), dim3(GLUE_THREADS), 0, (hipStream_t)stream, images, out, B,
                     H, W, norm3(mean, stdv));
  return avc_check_launch("avc_resize_norm_fwd");
}
extern "C" int avc_resize_norm_bwd(const float* dout, int B, int H, int W, const float* mean /* host [3] */, const float* stdv /* host [3] */,
                                   float* dimages, void* stream) {
  if (B <= 0) return 0;
  if (!dout || !dimages || !mean || !stdv || H < 1 || W < 1) { avc_set_error("avc_resize_norm_bwd: bad arguments"); return 1; }
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(resize_norm_bwd_kernel, dim3((unsigned)((n + GLUE_THREADS - 1) / GLUE_THREADS)), dim3(GLUE_THREADS), 0, (hipStream_t)stream, dout,
                     dimages, B, H, W, norm3(mean, stdv));
  return avc_check_launch("avc_resize_norm_bwd");
}

// ---------------------------------------------------------------------------------------------------------------
// The head of an iteration: rays of a view (dataset.py:277-293 gen_rays_pose / :252-275 the selected pixels of gen_rays_silhouettes),
// their near / far from the unit sphere (:331-342), and the prior image resampled to the ray grid (main.py:376-380: nearest) -- ~45 small
// torch launches as one.  torch.linspace's own formula (start + i step below the middle, end - (n - 1 - i) step above) so that the
// pixel centres are the same floats.
__device__ __forceinline__ float linspace_at(float end, int n, int i) {   // torch.linspace(0, end, n)[i], float32
  if (n == 1) return 0.f;
  const float step = end / (float)(n - 1);
  return i < n / 2 ? step * (float)i : end - step * (float)(n - 1 - i);
}
struct RaysIn {
  const float* pose;        // device [16]: camera-to-world 4 x 4, row major
  const long* sel;          // [R] row-major pixel of ray r in the Hn x Wn grid, or NULL: ray r = pixel r
  const float* prior;       // [Hp, Wp, 3] prior render (0 = background), or NULL
  float W, H, focal;        // the dataset's pinhole camera
  int Wn, Hn, R, Hp, Wp;
};
__global__ __launch_bounds__(GLUE_THREADS) void gen_rays_kernel(RaysIn a, float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                                float* __restrict__ near, float* __restrict__ far,
                                                                float* __restrict__ true_rgb, float* __restrict__ mask) {
#pragma clang fp contract(off)   // every float32 operation rounds like the torch op it replaces (a fused multiply-add would move a ray by an ulp)
  const int t = blockIdx.x * GLUE_THREADS + threadIdx.x;
  if (t < a.R) {
    const long p = a.sel ? a.sel[t] : (long)t;
    const int iy = (int)(p / a.Wn), ix = (int)(p % a.Wn);
    const float px = linspace_at(a.W - 1.f, a.Wn, ix), py = linspace_at(a.H - 1.f, a.Hn, iy);
    const float c0 = (px - 0.5f * a.W) / a.focal, c1 = -(py - 0.5f * a.H) / a.focal, c2 = -1.f;
    const float nrm = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
    const float v0 = c0 / nrm, v1 = c1 / nrm, v2 = c2 / nrm;
    float d[3], o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      d[i] = v0 * a.pose[4 * i] + v1 * a.pose[4 * i + 1] + v2 * a.pose[4 * i + 2];
      o[i] = a.pose[4 * i + 3];
      rays_d[3 * t + i] = d[i];
      rays_o[3 * t + i] = o[i];
    }
    const float aa = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const float bb = 2.f * (o[0] * d[0] + o[1] * d[1] + o[2] * d[2]);
    const float mid = 0.5f * (-bb) / aa;
    near[t] = fmaxf(mid - 1.f, 0.f);
    far[t] = mid + 1.f;
  }
  if (a.prior && t < a.Wn * a.Hn) {     // F.interpolate(mode='nearest'): source index floor(dst * in / out)
    const int iy = t / a.Wn, ix = t % a.Wn;
    int sy = (int)floorf((float)iy * ((float)a.Hp / (float)a.Hn)), sx = (int)floorf((float)ix * ((float)a.Wp / (float)a.Wn));
    if (sy > a.Hp - 1) sy = a.Hp - 1;
    if (sx > a.Wp - 1) sx = a.Wp - 1;
    const float* s = a.prior + ((long)sy * a.Wp + sx) * 3;
    true_rgb[3 * t] = s[0]; true_rgb[3 * t + 1] = s[1]; true_rgb[3 * t + 2] = s[2];
    mask[t] = s[0] != 0.f ? 1.f : 0.f;                 // main.py:379-380: mask = (true_rgb != 0)[..., :1]
  }
}
extern "C" int avc_gen_rays(const float* pose, const long* sel, const float* prior, int Hp, int Wp, float W, float H, float focal, int Wn, int Hn,
                            int R, float* rays_o, float* rays_d, float* near, float* far, float* true_rgb, float* mask, void* stream) {
  if (R <= 0 && !prior) return 0;
  if (!pose || (R > 0 && (!rays_o || !rays_d || !near || !far)) || (prior && (!true_rgb || !mask))) { avc_set_error("avc_gen_rays: NULL buffer"); return 1; }
  RaysIn a;
  a.pose = pose; a.sel = sel; a.prior = prior; a.W = W; a.H = H; a.focal = focal; a.Wn = Wn; a.Hn = Hn; a.R = R; a.Hp = Hp; a.Wp = Wp;
  const int n = prior ? (R > Wn * Hn ? R : Wn * Hn) : R;
  hipLaunchKernelGGL(gen_rays_kernel, dim3((n + GLUE_THREADS - 1) / GLUE_THREADS), dim3(GLUE_THREADS), 0, (hipStream_t)stream, a, rays_o, rays_d, near,
                     far, true_rgb, mask);
  return avc_check_launch("avc_gen_rays");
}

// main.py:398-405: 0.2 / 0.8 chess board of L-pixel squares under torchvision's GaussianBlur(kernel (5, 9), sigma): separable taps with
// reflect padding (5 along x, 9 along y), one thread per pixel.  taps: device [14] = kx[5], ky[9] (normalised, computed on the host).
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__global__ __launch_bounds__(GLUE_THREADS) void chess_bg_kernel(float* __restrict__ out, int H, int W, int L, const float* __restrict__ taps) {
  const int t = blockIdx.x * GLUE_THREADS + threadIdx.x;
  if (t >= H * W) return;
  const int y = t / W, x = t % W;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < 9; ++ky) {
    const int yy = reflect_idx(y + ky - 4, H);
    float row = 0.f;
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
      const int xx = reflect_idx(x + kx - 2, W);
      row += taps[kx] * ((((yy / L) + (xx / L)) & 1) == 0 ? 0.8f : 0.2f);
    }
    acc += taps[5 + ky] * row;
  }
  out[t] = acc;
}
extern "C" int avc_chess_background(float* out, int H, int W, int chess_length, const float* taps, void* stream) {
  if (H <= 0 || W <= 0) return 0;
  if (!out || !taps || chess_length < 1 || H < 5 || W < 3) { avc_set_error("avc_chess_background: bad arguments"); return 1; }
  hipLaunchKernelGGL(chess_bg_kernel, dim3((H * W + GLUE_THREADS - 1) / GLUE_THREADS), dim3(GLUE_THREADS), 0, (hipStream_t)stream, out, H, W,
                     chess_length, taps);
  return avc_check_launch("avc_chess_background");
}

// renderer.py:311-322: the coarse sample depths z[r][i] = near[r] + (far[r] - near[r]) * linspace(0, 1, n)[i] (+ (jitter[r] - 0.5) * 2 / n
// with perturb), every float32 operation rounded like the torch op it replaces (torch.linspace: start + step * i below the middle,
// end - step * (n - 1 - i) above).  One launch instead of nine.
__global__ __launch_bounds__(GLUE_THREADS) void coarse_z_kernel(const float* __restrict__ near, const float* __restrict__ far,
                                                                const float* __restrict__ jitter, int R, int n, float* __restrict__ z) {
#pragma clang fp contract(off)   // (as in gen_rays_kernel: the torch ops this replaces are separate launches, each product and sum rounded)
  const int t = blockIdx.x * GLUE_THREADS + threadIdx.x;
  if (t >= R * n) return;
  const int r = t / n, i = t % n;
  const float lin = linspace_at(1.f, n, i);     // (torch.linspace's kernel itself contracts end - step * k into one fma: so does linspace_at)
  const float nr = near[r];
  float v = nr + (far[r] - nr) * lin;
  if (jitter) v = v + (jitter[r] - 0.5f) * 2.0f / (float)n;
  z[t] = v;
}
extern "C" int avc_coarse_z(const float* near, const float* far, const float* jitter, int R, int n, float* z, void* stream) {
  if (R <= 0 || n <= 0) return 0;
  if (!near || !far || !z) { avc_set_error("avc_coarse_z: NULL buffer"); return 1; }
  hipLaunchKernelGGL(coarse_z_kernel, dim3(((long)R * n + GLUE_THREADS - 1) / GLUE_THREADS), dim3(GLUE_THREADS), 0, (hipStream_t)stream, near, far,
                     jitter, R, n, z);
  return avc_check_launch("avc_coarse_z");
}

// ---------------------------------------------------------------------------------------------------------------
// The scalar tail of the iteration's loss (main.py:491-534) in one launch each way: the two CLIP cosines
//   cos_b = < e_b / max(|e_b|, 1e-8), t / max(|t|, 1e-8) >,   e_b = torch.mean(enc[b:b+1], dim=0), t = torch.mean(text, dim=0)
// (torch.cosine_similarity normalises first), the colour / mask terms from the sums of avc_shade_loss_fwd and
//   loss = l1 / (mask_sum + 1e-5) + eikonal * igr_weight + (bce / P) * mask_weight + sum_b (1 - cos_b) * clip_weight
// in the reference's order of additions.  One 512-thread block, thread = embedding channel.
#define TAIL_D 512
struct TailW { float igr_w, mask_w, clip_w, P; };
__device__ __forceinline__ float tail_block_sum(float v, float (*red)[TAIL_D / 64], int slot) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  if ((threadIdx.x & 63) == 0) red[slot][threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < TAIL_D / 64; ++w) t += red[slot][w];
  return t;
}
// out[8] = loss, colour loss, mask loss, cos_0, cos_1, 0, 0, 0;  saved[4 B] = per image (|e| clamped, cos, |e| raw, 0); saved[4 B] = |t| clamped
__global__ __launch_bounds__(TAIL_D) void loss_tail_fwd_kernel(const float* __restrict__ enc, const float* __restrict__ text, int B, int T,
                                                               const float* __restrict__ sums, const float* __restrict__ eik, TailW w,
                                                               float* __restrict__ loss, float* __restrict__ out, float* __restrict__ saved) {
  __shared__ float red[2 + 2 * 4][TAIL_D / 64];
  const int c = threadIdx.x;
  float tm = 0.f;
  for (int t = 0; t < T; ++t) tm += text[(long)t * TAIL_D + c];
  tm /= (float)T;
  const float nt = fmaxf(sqrtf(tail_block_sum(tm * tm, red, 0)), 1e-8f);
  const float th = tm / nt;
  float cosv[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < B; ++b) {
    const float e = enc[(long)b * TAIL_D + c];
    const float nr = sqrtf(tail_block_sum(e * e, red, 2 + 2 * b)), ne = fmaxf(nr, 1e-8f);
    const float cs = tail_block_sum((e / ne) * th, red, 3 + 2 * b);
    cosv[b] = cs;
    if (c == 0) { saved[4 * b] = ne; saved[4 * b + 1] = cs; saved[4 * b + 2] = nr; saved[4 * b + 3] = 0.f; }
  }
  if (c == 0) {
    saved[4 * B] = nt;
    const float colour = sums[0] / (sums[1] + 1e-5f), maskl = sums[2] / w.P;
    float l = colour + eik[0] * w.igr_w + maskl * w.mask_w;
    for (int b = 0; b < B; ++b) l = l + (1.f - cosv[b]) * w.clip_w;
    loss[0] = l;
    out[0] = l; out[1] = colour; out[2] = maskl; out[3] = cosv[0]; out[4] = cosv[1]; out[5] = 0.f; out[6] = 0.f; out[7] = 0.f;
  }
}
// g = d L / d loss (device scalar).  d_enc [B,512]; dd[8]: [0..3] = gradient of sums (l1, mask_sum: 0, bce, sq: 0), [4] = of the eikonal term
__global__ __launch_bounds__(TAIL_D) void loss_tail_bwd_kernel(const float* __restrict__ g, const float* __restrict__ enc,
                                                               const float* __restrict__ text, int B, int T, const float* __restrict__ sums,
                                                               const float* __restrict__ saved, TailW w, float* __restrict__ d_enc,
                                                               float* __restrict__ dd) {
  const int c = threadIdx.x;
  const float gl = g[0];
  float tm = 0.f;
  for (int t = 0; t < T; ++t) tm += text[(long)t * TAIL_D + c];
  tm /= (float)T;
  const float th = tm / saved[4 * B];
  for (int b = 0; b < B; ++b) {
    const float ne = saved[4 * b], cs = saved[4 * b + 1], nr = saved[4 * b + 2];
    const float e = enc[(long)b * TAIL_D + c];
    // cos = <e / ne, th>: d e = th / ne - (the norm's branch, only where the clamp is inactive) cos * e / ne^2
    float d = th / ne;
    if (nr > 1e-8f) d -= cs * e / (ne * ne);
    d_enc[(long)b * TAIL_D + c] = -gl * w.clip_w * d;
  }
  if (c == 0) {
    dd[0] = gl / (sums[1] + 1e-5f); dd[1] = 0.f; dd[2] = gl * w.mask_w / w.P; dd[3] = 0.f;
    dd[4] = gl * w.igr_w; dd[5] = 0.f; dd[6] = 0.f; dd[7] = 0.f;
  }
}
extern "C" int avc_loss_tail_fwd(const float* enc, const float* text, int B, int T, int D, const float* sums, const float* eikonal, float igr_weight,
                                 float mask_weight, float clip_weight, float P, float* loss, float* out, float* saved, void* stream) {
  if (D != TAIL_D || B < 1 || B > 4 || T < 1) { avc_set_error("avc_loss_tail_fwd: built for 512-wide embeddings, 1-4 images"); return 1; }
  if (!enc || !text || !sums || !eikonal || !loss || !out || !saved) { avc_set_error("avc_loss_tail_fwd: NULL buffer"); return 1; }
  const TailW w = {igr_weight, mask_weight, clip_weight, P};
  hipLaunchKernelGGL(loss_tail_fwd_kernel, dim3(1), dim3(TAIL_D), 0, (hipStream_t)stream, enc, text, B, T, sums, eikonal, w, loss, out, saved);
  return avc_check_launch("avc_loss_tail_fwd");
}
extern "C" int avc_loss_tail_bwd(const float* g, const float* enc, const float* text, int B, int T, int D, const float* sums, const float* saved,
                                 float igr_weight, float mask_weight, float clip_weight, float P, float* d_enc, float* dd, void* stream) {
  if (D != TAIL_D || B < 1 || B > 4 || T < 1) { avc_set_error("avc_loss_tail_bwd: built for 512-wide embeddings, 1-4 images"); return 1; }
  if (!g || !enc || !text || !sums || !saved || !d_enc || !dd) { avc_set_error("avc_loss_tail_bwd: NULL buffer"); return 1; }
  const TailW w = {igr_weight, mask_weight, clip_weight, P};
  hipLaunchKernelGGL(loss_tail_bwd_kernel, dim3(1), dim3(TAIL_D), 0, (hipStream_t)stream, g, enc, text, B, T, sums, saved, w, d_enc, dd);
  return avc_check_launch("avc_loss_tail_bwd");
}
