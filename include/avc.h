/* avc.h -- C ABI of libavc.so, the MI355X (gfx950) implementation of the AvatarCLIP AppearanceGen hot path.
 *
 * The reference (hongfz16/AvatarCLIP) has no FFI of its own: the hot path is the Python call chain
 *   AvatarGen/AppearanceGen/main.py:418-420   Runner.train_clip -> NeuSRenderer.render
 *   AvatarGen/AppearanceGen/models/renderer.py:302-397 (render), :133-193 (up_sample/cat_z_vals), :195-300 (render_core)
 *   AvatarGen/AppearanceGen/models/fields.py:72-107,154-185 (SDFNetwork / RenderingNetwork)
 *   AvatarGen/AppearanceGen/main.py:512 (perceptor.encode_image -- OpenAI CLIP ViT-B/32, un-vendored)
 * Each entry point below names the reference lines it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions: every pointer is a DEVICE pointer (hipMalloc / torch allocation) unless marked host; buffers are
 * caller-allocated, no ownership transfer; `stream` is a hipStream_t; all kernels are enqueued on it and the
 * call returns without synchronising.  Return value 0 = ok, otherwise avc_last_error() describes the failure.
 * No CPU fallback exists: with no GPU the calls fail.
 */
#ifndef AVC_H
#define AVC_H
#ifdef __cplusplus
extern "C" {
#endif

#define AVC_NET_FULL 0  /* confs/examples (all 144):   SDF 39-256-256-256-217(+39)-257, colour 262-256-256-{3,3} */
#define AVC_NET_SMALL 1 /* confs/examples_small:        SDF 39-128-128-89(+39)-129,     colour 134-128-{3,3}     */

/* Interface revision of this header.  It changes whenever an existing entry point's argument list or meaning changes (2: round 5 put
 * `colsum` into avc_render_points_bwd and moved the second-order row-0 term out of the weight-gradient products), so a caller built
 * against an older header can refuse the library instead of passing arguments in the wrong slots:
 *     if (avc_version() != AVC_ABI_VERSION) abort();       (avatarclip_amd/lib.py: load() does exactly this) */
#define AVC_ABI_VERSION 2

const char* avc_last_error(void);
int avc_version(void);
/* number of int32 slots in the `offs` arrays below (== OFF_COUNT of csrc/avc_common.h) */
int avc_num_offsets(void);

/* SDFNetwork.sdf (fields.py:90-91) without autograd: renderer.py:337-338 (coarse), :187 (new samples),
 * :403 (extract_geometry query).  Points are either pts[N,3] or rays_o/rays_d[R,3] + z[R,ldz] (S per ray).
 * If slot != NULL the value of sample (ray,j) is scattered to sdf_out[ray*ld_out + slot[ray*S+j]]
 * (the merge of cat_z_vals, renderer.py:179-193). wf16/tab/offs: packed parameters (avatarclip_amd/packing.py). */
int avc_sdf_forward(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z, int S,
                    int ldz, long npts, const void* wf16, const float* tab, const int* offs /* host */,
                    float* sdf_out, const int* slot, int ld_out, void* stream);

/* One step of NeuSRenderer.up_sample + sample_pdf(det=True) + the sort/merge of cat_z_vals
 * (renderer.py:133-177, 39-69, 179-193) for R rays with n current samples, producing m new ones.
 * z_out/sdf_out [R, n+m] receive the merged z and the old sdf values at their merged positions; z_new/slot_new
 * [R,m] the new depths and their positions in the merged row (sdf of those is filled by avc_sdf_forward). */
int avc_upsample_step(const float* rays_o, const float* rays_d, const float* z_in, const float* sdf_in, int R,
                      int n, int m, float inv_s, float* z_out, float* sdf_out, float* z_new, int* slot_new,
                      void* stream);
/* the same with the work distribution as an argument: lanes_per_ray = 0 (chosen from n and m: 16 / 32 lanes per ray for n + m <= 64 /
 * 128, avc_upsample_step's choice) or 64 (one wavefront per ray: the cross-check of the grouped kernels) */
int avc_upsample_step_lanes(const float* rays_o, const float* rays_d, const float* z_in, const float* sdf_in, int R,
                            int n, int m, float inv_s, float* z_out, float* sdf_out, float* z_new, int* slot_new,
                            int lanes_per_ray, void* stream);

/* sdf_network(pts) + sdf_network.gradient(pts) + color_network(...) of render_core (renderer.py:221-232) at the
 * section mid-points of z[R,S] (or at pts[N,3]): sdf[N], normal[N,3] (= d sdf/dx), rgb[N,6] = sigmoid([rgb ; extra]). */
int avc_render_points_fwd(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                          int S, int ldz, float sample_dist, long npts, const void* wf16, const float* tab,
                          const int* offs /* host */, float* sdf_out, float* normal_out, float* rgb_out, long max_waves,
                          void* scratch /* max_waves * avc_fwd_scratch_bytes_per_wave(net) bytes */, void* stream);
/* bytes of the per-wavefront slot in which avc_render_points_fwd parks the trunk activations between the forward and the
 * normal sweep (persistent workgroups of 8 wavefronts; max_waves bounds the resident grid) */
long avc_fwd_scratch_bytes_per_wave(int net);

/* NeuS alpha + compositing of render_core (renderer.py:234-286), one wavefront per ray.
 * bg_mode 0: none, 1: bg[3] shared, 2: bg[R] grey per ray (main.py:387-415); background is composited into
 * `extra` only (renderer.py:277-281).  Outputs: color[R,3], extra[R,3], weights[R,S], cdf[R,S], mid_z[R,S],
 * inside[R,S], eik[R,2] = per-ray (sum relax*(|n|-1)^2, sum relax); optional (NULL = skip) per-ray reductions of the weights
 * that the callers of render() take next: wstat[2][R] = (sum_i w_i, max_i w_i) (renderer.py:391-392 weight_sum / weight_max)
 * and nsum[R,3] = sum_i w_i n_i (the shading normal of main.py:428 before its normalisation). */
int avc_composite_fwd(const float* sdf, const float* normal, const float* rgb, const float* z, const float* rays_o,
                      const float* rays_d, int R, int S, const float* inv_s /* device scalar */, float sample_dist,
                      float cos_anneal, const float* bg, int bg_mode, float* color, float* extra, float* weights,
                      float* cdf, float* mid_z, float* inside, float* eik, float* wstat, float* nsum, void* stream);

/* The scalar reductions around the compositing kernels in one launch each: column sums of x [R,C] (C <= 4) in a fixed order; mode 1
 * (x = avc_composite_fwd's eik): out[1] = sum x[:,1] + 1e-5, out[0] = sum x[:,0] / out[1] = the eikonal term of renderer.py:283-285.
 * avc_inv_s: out[0] = exp(10 variance).clip(1e-6, 1e6), out[1] = 1 / out[0] (fields.py:275-276, renderer.py:234,288); with g != NULL:
 * out[0] = its backward g[0] * d/dv. */
long avc_colsum_scratch_bytes(void);   /* zero-initialised once by the caller; every call leaves its ticket word zero */
int avc_colsum(const float* x, long R, int C, int mode, float* out, void* scratch, void* stream);
int avc_inv_s(const float* variance, const float* g, float* out, void* stream);

/* Reverse of avc_composite_fwd.  Upstream: d_color[R,3], d_extra[R,3], d_weights[R,S] (may be NULL), d_normal_up[R,S,3] (may
 * be NULL), d_wsum[R] / d_nsum[R,3] = gradients of wstat[:,0] / nsum (may be NULL), eik_scale = d(loss)/d(eik) / (sum relax +
 * 1e-5) (device scalar).  Outputs: d_sdf[R,S], d_normal[R,S,3], d_rgb[R,S,6], d_inv_s[R] (per-ray partial of d loss / d inv_s). */
int avc_composite_bwd(const float* sdf, const float* normal, const float* rgb, const float* z, const float* rays_o,
                      const float* rays_d, int R, int S, const float* inv_s, float sample_dist, float cos_anneal,
                      const float* bg, int bg_mode, const float* d_color, const float* d_extra,
                      const float* d_weights, const float* d_normal_up, const float* d_wsum, const float* d_nsum,
                      const float* eik_scale, float* d_sdf, float* d_normal, float* d_rgb, float* d_inv_s, void* stream);

/* The differentiable forward of render_core (renderer.py:221-232, once per iteration as in the reference): the same outputs
 * as avc_render_points_fwd, plus everything the backward pass and the weight-gradient products need from the forward pass,
 * written to the F REGION of the OPERAND PANELS: per 32-point block avc_fwd_panel_tiles(net) tiles of 2 KiB, each tile = the
 * two 16-bit B-operand fragments [k-step][lane = point + 32 half][8 features] of 32 features x 32 points (f16: PE values,
 * h_l, g_a,l, feature vector, [x,n], r1, r2), and the ReLU masks of r1 / r2 (avc_mask_u16_per_block(net) x 16 bits per
 * block: [layer][tile][lane], bit (r >> 1) + 8 (r & 1) = feature row r of the lane's 16 rows is > 0 -- opaque to the caller, the
 * backward kernel is the only reader).  Both buffers need (nblk + 1) blocks, nblk = ceil(npts / 32): the last block is a sink for wavefronts past the end.
 * The gradient-type operands (bf16: gbar_h, abar, delta, ybar) live in a separate G REGION of avc_grad_panel_tiles(net) tiles
 * per block that only ever holds one SLAB of blocks (see avc_render_points_bwd). */
int avc_fwd_panel_tiles(int net);
int avc_grad_panel_tiles(int net);
int avc_mask_u16_per_block(int net);
int avc_render_points_fwd_train(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                                int S, int ldz, float sample_dist, long npts, const void* wf16, const float* tab,
                                const int* offs /* host */, float* sdf_out, float* normal_out, float* rgb_out,
                                long max_waves, void* fpanels, void* masks, void* stream);

/* Backward of avc_render_points_fwd_train wrt every dense weight (autograd incl. the double backward of
 * SDFNetwork.gradient, fields.py:96-107; main.py:537) for one SLAB of npts points (a block-aligned sub-range of the forward's
 * points: the ray / z / gradient pointers and fpanels / masks point at the slab's first ray resp. first block).  Recomputes
 * nothing of the forward: reads h_l, g_a,l (fpanels), the masks and the colours (rgb_fwd = the forward's rgb_out) back, runs
 * the colour backward, the second-order sweep and the reverse sweep (bf16 operands) and writes the gradient-type operand
 * tiles of the slab to gpanels ((nblk + 1) blocks of avc_grad_panel_tiles(net) tiles, block 0 = the slab's first block);
 * avc_weight_grad_all then contracts both regions over the slab's points.  max_waves bounds the resident grid (persistent
 * workgroups of 8 wavefronts).
 * One term of the weight gradient is not a product and does not go through the panels: the second-order term of row 0 of the last
 * SDF layer, dW_last[0, :] += sum_points [gbar_hs ; gbar_h0] / sqrt2 (fields.py:96-107 under main.py:537).  The kernel
 * reduces it over each wavefront's points and writes colsum[avc_bwd_colsum_rows(npts, max_waves)][avc_bwd_colsum_floats(net)]: per row
 * [ST tiles][2 halves][16 accumulator registers] of gbar_hs, then [3 fragments][2 halves][8 slots] of gbar_h0; the caller adds the rows
 * (and the slabs) up and scatters them into the dense gradient (packing.Layout.cs_src / cs_tgt / cs_scale). */
int avc_bwd_colsum_floats(int net);
long avc_bwd_colsum_rows(long npts, long max_waves);
int avc_render_points_bwd(int net, const float* pts, const float* rays_o, const float* rays_d, const float* z,
                          int S, int ldz, float sample_dist, long npts, const void* wbf16, const float* tab,
                          const int* offs /* host */, const float* d_sdf, const float* d_normal, const float* d_rgb,
                          const float* rgb_fwd, const void* fpanels, void* gpanels, const void* masks, float* colsum,
                          long max_waves, void* stream);

/* Every weight-gradient product of one slab in one launch.  pairs (host) = npairs x {pa, ta, pb, tb, out_off, bias_off,
 * type_a, type_b}: pair i contracts A tiles pa .. pa+ta-1 with B tiles pb .. pb+tb-1 (1 <= ta <= 8, 1 <= tb <= 9, or the merged
 * last-layer product ta = tb = 9) over the points of `nblk` blocks; type selects the operand's region and element type: 0 = F
 * region (fpanels, ftiles tiles per block, f16), 1 = G region (gpanels, gtiles per block, bf16), tile indices are
 * region-local.  The tiles are copied to LDS by global->LDS DMA in a chunk order that lets gfx950's LDS transpose read
 * (ds_read_b64_tr_b16) hand every lane its feature-major fragment, f16 operands are converted to bf16 after the read, and the
 * products accumulate in fp32.  partial[split][out_off + ((ta_i * tb + tb_j) * 64 + lane) * 16 + r] = element
 * dW[32 ta_i + (r&3)+8(r>>2)+4h][32 tb_j + n] of lane (n,h); bias_partial[split][bias_off + 32 ta_i + n] = sum_points
 * A[:, 32 ta_i + n] (bias_off < 0: none).  nsplit = split-K factor (grid x); split s writes its slab at
 * partial + s*out_stride (bias_partial + s*bias_stride), floats; the caller sums the slabs (no atomics). */
int avc_weight_grad_all(const void* fpanels, int ftiles, const void* gpanels, int gtiles, int npairs,
                        const int* pairs /* host */, long nblk, float* partial, float* bias_partial, int nsplit,
                        int out_stride, int bias_stride, void* stream);

/* What follows avc_weight_grad_all (the tail of main.py:537's backward for the MLP weights): acc [gout_size + gbias_size] (+)= the sum
 * over the nsplit slabs of partial / bias_partial, added in split order (accumulate != 0: on top of acc, i.e. the previous slab of
 * points); then grad[t] = sum_{k = off[t] .. off[t+1]-1} acc[src[k]] * scale[k] maps the products' tile layout back to the dense
 * parameter vector (device tables built from packing.Layout.un_* / ub_*, engine._DevLayout). */
int avc_weight_grad_reduce(const float* partial, const float* bias_partial, int nsplit, int out_stride, int bias_stride, int gout_size,
                           int gbias_size, float* acc, int accumulate, void* stream);
int avc_weight_grad_unpack(const float* acc, const int* off, const int* src, const float* scale, int nparam, float* grad, void* stream);

/* The per-pixel glue between the renderer and CLIP in one launch each way (main.py:426-453 random-light Lambert shading of the rendered
 * normals, :461-487 scatter of the silhouette rays into full images over the augmentation background, :491-492 / :497 the per-pixel terms
 * of the colour L1 and mask BCE losses).  P pixels of the H x W image; ray_of_pixel[P] = the ray of a pixel or -1 (NULL: pixel p = ray p,
 * the full-frame mode); color / extra [R,3], wsum [R], nsum [R,3] = sum_i w_i n_i (NULL: no shading, both images = extra_color);
 * true_rgb [P,3], mask [P] (what the losses use); bg [P] grey background outside the silhouette (NULL: bg_const); light = device
 * [4] (unit light direction, ambience).  Outputs: images [2][P][3] (0 = texture_shading, or extra_color if img0_is_extra; 1 =
 * rand_shading_rgb), partial [avc_shade_loss_blocks(P)][4] = per-block sums of (|color - true| mask, mask, BCE term, (color - true)^2
 * mask: the logged psnr of main.py:493) and sums [4] = their totals, added up in a fixed order by the block that finishes last
 * (ticket: one zero-initialised device word per stream of calls; the kernel leaves it zero).  The backward takes dimg0 / dimg1 [P,3] (NULL:
 * unused image) and gs = device [4], the gradient of `sums` ([0]: d loss / d l1-sum, [2]: d loss / d bce-sum), and writes dcolor / dextra
 * [R,3], dwsum [R], dnsum [R,3] for every ray that owns a pixel. */
int avc_shade_loss_blocks(int P);
int avc_shade_loss_fwd(const float* color, const float* extra, const float* wsum, const float* nsum, const float* true_rgb,
                       const float* mask, const int* ray_of_pixel, const float* bg, float bg_const, const float* light, int P,
                       int img0_is_extra, float* images, float* partial, float* sums, unsigned* ticket, void* stream);
int avc_shade_loss_bwd(const float* color, const float* extra, const float* wsum, const float* nsum, const float* true_rgb,
                       const float* mask, const int* ray_of_pixel, const float* light, int P, int img0_is_extra, const float* dimg0,
                       const float* dimg1, const float* gs, float* dcolor, float* dextra, float* dwsum, float* dnsum, void* stream);
/* engine.Packed in one launch: w_f16 / w_bf16 [n16] = flat[idx16] * scale16 (both 16-bit types), tab [n32] = flat[idx32] * scale32; an
 * index >= nparam reads the appended zero (packing.py: the blobs are pure index gathers of the dense vector, fields.py:65-66,139-143). */
int avc_pack_params(const float* flat, int nparam, const long* idx16, const float* scale16, int n16, const long* idx32,
                    const float* scale32, int n32, void* w_f16, void* w_bf16, float* tab, void* stream);
/* renderer.py:311-322: z [R,n] = near + (far - near) * linspace(0, 1, n) (+ (jitter - 0.5) * 2 / n; jitter [R] or NULL), rounded like
 * the torch ops */
int avc_coarse_z(const float* near_, const float* far_, const float* jitter, int R, int n, float* z, void* stream);
/* The scalar tail of the loss (main.py:491-534) in one launch each way: cos_b = torch.cosine_similarity(torch.mean(enc[b:b+1], 0),
 * torch.mean(text, 0), dim=0) for the B images (enc [B,512], text [T,512]) and
 *   loss = sums[0] / (sums[1] + 1e-5) + eikonal * igr_weight + (sums[2] / P) * mask_weight + sum_b (1 - cos_b) * clip_weight
 * (the reference's order of additions).  loss: device scalar; out [8] = loss, colour loss, mask loss, cos_0, cos_1 (statistics); saved
 * [4 B + 1]: what the backward needs.  The backward takes g = d L / d loss (device scalar) and writes d_enc [B,512] and dd [8]: [0..3] the
 * gradient of sums (what avc_shade_loss_bwd takes as gs), [4] that of the eikonal term. */
int avc_loss_tail_fwd(const float* enc, const float* text, int B, int T, int D, const float* sums, const float* eikonal, float igr_weight,
                      float mask_weight, float clip_weight, float P, float* loss, float* out, float* saved, void* stream);
int avc_loss_tail_bwd(const float* g, const float* enc, const float* text, int B, int T, int D, const float* sums, const float* saved,
                      float igr_weight, float mask_weight, float clip_weight, float P, float* d_enc, float* dd, void* stream);
/* CLIP's preprocessing (main.py:261-267,510-511: RandomResizedCrop(224, scale=(1,1)) of a square image = bilinear resize,
 * align_corners = False, no antialias; Normalize): images [B,H,W,3] -> out [B,3,224,224] = (resize - mean) / std, and its transpose
 * (dimages is overwritten).  mean / std: HOST arrays of 3 floats. */
int avc_resize_norm_fwd(const float* images, int B, int H, int W, const float* mean, const float* stdv, float* out, void* stream);
int avc_resize_norm_bwd(const float* dout, int B, int H, int W, const float* mean, const float* stdv, float* dimages, void* stream);

/* The rays of a view in one launch (dataset.py:277-293 gen_rays_pose: pixel centres linspace(0, W - 1, Wn) x linspace(0, H - 1, Hn) of the
 * dataset's pinhole camera (W, H, focal), direction = pose[:3,:3] p / |p| with p = ((x - W/2) / f, -(y - H/2) / f, -1), origin = pose[:3,3];
 * sel != NULL: only the R listed row-major pixels, the selected rays of gen_rays_silhouettes :252-275) with near / far from the unit
 * sphere (:331-342), and -- prior != NULL -- the prior render [Hp,Wp,3] resampled to the Hn x Wn grid (main.py:376-380: nearest) as
 * true_rgb [Hn*Wn,3] and mask [Hn*Wn] = (true_rgb[..., 0] != 0).  pose: device [16], camera-to-world, row major. */
int avc_gen_rays(const float* pose, const long* sel, const float* prior, int Hp, int Wp, float W, float H, float focal, int Wn, int Hn,
                 int R, float* rays_o, float* rays_d, float* near, float* far, float* true_rgb, float* mask, void* stream);
/* main.py:398-405: the blurred chess-board background of the augmentation, out [H*W] (0.2 / 0.8 squares of chess_length pixels,
 * GaussianBlur kernel (5, 9) with reflect padding; taps = device [14]: the 5 normalised taps along x, then the 9 along y). */
int avc_chess_background(float* out, int H, int W, int chess_length, const float* taps, void* stream);

/* Dense-parameter assembly of one optimisation step: W_l = g_l * v_l / ||v_l||_row (nn.utils.weight_norm, fields.py:65-66,139-143;
 * g[l] == NULL: the plain weight) and the biases of `n` linears written into the flat dense vector `flat` at w_off[l] (row-major
 * [rows, cols]) / b_off[l] (floats), and the backward of that map: dflat -> dv[l], dg[l], db[l] (b[l] / db[l] may be NULL).
 * The arrays of pointers / shapes are HOST arrays (n <= 16). */
int avc_dense_params_fwd(int n, const void* const* v, const void* const* g, const void* const* b, const int* rows,
                         const int* cols, const long* w_off, const long* b_off, float* flat, void* stream);
int avc_dense_params_bwd(int n, const void* const* v, const void* const* g, void* const* dv, void* const* dg, void* const* db,
                         const int* rows, const int* cols, const long* w_off, const long* b_off, const float* dflat,
                         void* stream);

/* ---- mesh extraction (Runner.validate_mesh, main.py:850-919; renderer.py:10-36; mcubes.marching_cubes) ----
 * Marching cubes over u[nx][ny][nz] (row-major, the reference's extract_fields layout) at iso level `iso`, inside = u > iso,
 * one welded vertex per sign-changing grid edge.  Pass 1 writes, per grid point p, the flags of the three edges it owns
 * (vflag[3p + axis]) and the triangle count of the cell whose low corner it is (ccount[p]); the caller turns both into
 * exclusive prefix sums (vid, coff); pass 2 writes vertices (index coordinates, float32: the caller applies
 * renderer.py:35) and triangles (vertex ids).  ntri_table[256], tri_table[256][5][3], edge_table[12][4] = (dx,dy,dz,axis):
 * device copies of avatarclip_amd/mc_tables.py. */
int avc_mc_classify(const float* u, int nx, int ny, int nz, float iso, const int* ntri_table, int* vflag, int* ccount,
                    void* stream);
int avc_mc_emit(const float* u, int nx, int ny, int nz, float iso, const int* vflag, const int* vid, const int* ccount,
                const int* coff, const signed char* tri_table, const int* edge_table, float* verts, int* tris, void* stream);

/* ---- CLIP ViT-B/32 image encoder (perceptor.encode_image, main.py:512,518,524; OpenAI clip/model.py) ----
 * y[M,N] = act(x[M,K] W^T + bias) (+ residual); W pre-packed bf16 [N/32][K/16][64][8] (lane (n,h): W[32t+n][16s+8h+j]).
 * act 1 = QuickGELU (y_pre, if given, receives the pre-activation for the backward).  The backward dX = dY W is the same
 * call with the packed W^T (weights are frozen in AvatarCLIP: main.py:260).  Any M: up to 128 rows (the per-iteration calls) one
 * 8-wavefront split-K workgroup per (32 columns, 32 rows); beyond (batched scoring) an LDS-staged GEMM, one 4-wavefront workgroup
 * per 128 x 128 output block (N % 128 == 0, K % 32 == 0; csrc/avc_vit_gemm.hip). */
int avc_vit_linear(const float* x, const void* w_packed, const float* bias, const float* residual, float* y,
                   float* y_pre, int M, int N, int K, int act, void* workspace /* avc_vit_workspace_bytes(M, K) */,
                   void* stream);
/* backward through a QuickGELU layer and the linear in front of it in one call: dx[M,N] = (dy[M,K] * gelu'(pre[M,K])) W, with
 * the packed W^T (N = the layer's input width, K = its output width); the activation derivative is applied while dy is packed. */
int avc_vit_linear_bwd_gelu(const float* dy, const float* pre, const void* wt_packed, float* dx, int M, int N, int K,
                            void* workspace, void* stream);
/* bytes of the bf16 fragment copy of x that avc_vit_linear builds in `workspace` */
long avc_vit_workspace_bytes(int M, int K);
/* The batched scoring calls (no gradient; hundreds of images: ShapeGen/main.py:104-128, AvatarAnimate pose_generation.py:79-110) keep
 * the activations between the kernels as packed bf16 operands ([row tile][k-step][lane (row, half)][8], avc_vit_workspace_bytes(M, K)
 * bytes for an [M,K] activation) instead of fp32 rows + a packing pass per linear:
 *   avc_vit_ln_pack               LayerNorm(x[M,768]; gamma, beta, eps) (clip/model.py LayerNorm, fp32 statistics) -> packed
 *   avc_vit_linear_packed         y = act(xs W^T + b) (+ residual) from a packed operand; with ys_packed != NULL the result leaves
 *                                 packed for the next linear (then y, residual must be NULL), else fp32 y[M,N]
 *   avc_vit_attention_fwd_packed  attention with its output packed for the out-projection (rows = b * 50 + token)
 * Shapes: more than 128 rows' worth is not required, but N % 128 == 0 and K % 32 == 0 are. */
int avc_vit_ln_pack(const float* x, const float* gamma, const float* beta, float eps, int M, int K, void* xs_packed, void* stream);
int avc_vit_linear_packed(const void* xs_packed, const void* w_packed, const float* bias, const float* residual, float* y,
                          void* ys_packed, int M, int N, int K, int act, void* stream);
int avc_vit_attention_fwd_packed(const float* qkv, void* out_packed, int B, int T, int width, int heads, void* stream);
/* The per-iteration call of the training loop (main.py:512,524: 1-2 images WITH a gradient to the pixels, M <= 128 rows) hands its
 * activations over the same way, forward AND backward, on the split-K latency kernel (clip_vit.BlocksFn: the residual blocks of
 * clip/model.py ResidualAttentionBlock as one autograd node; weights frozen, main.py:260):
 *   avc_vit_pack          fp32 rows (optionally times QuickGELU'(gelu_pre)) -> packed operand
 *   avc_vit_linear_small  y = f(xs W^T + b) (+ residual) from a packed operand; act 0 identity, 1 QuickGELU (y_pre receives the
 *                         pre-activation), 2 times QuickGELU'(gelu_pre[M,N]) (backward of a QuickGELU layer); the result leaves as
 *                         fp32 rows y and / or as the packed operand ys_packed of the next linear
 *   avc_vit_ln_bwd        dx = (d LayerNorm(x; gamma) / dx)^T dy (+ residual_grad): fp32 rows and, xs_packed != NULL, the packed
 *                         operand of the transposed linear behind it (statistics recomputed from x, eps as in avc_vit_ln_pack) */
int avc_vit_pack(const float* x, const float* gelu_pre, void* xs_packed, int M, int K, void* stream);
/* avc_vit_attention_bwd with dqkv leaving as the packed operand ([B * 50, 3 W]) of the transposed in-projection */
int avc_vit_attention_bwd_packed(const float* qkv, const float* dout, void* dqkv_packed, int B, int T, int width, int heads, void* stream);
int avc_vit_linear_small(const void* xs_packed, const void* w_packed, const float* bias, const float* residual, const float* gelu_pre,
                         float* y, float* y_pre, void* ys_packed, int M, int N, int K, int act, void* stream);
int avc_vit_ln_bwd(const float* dy, const float* x, const float* gamma, float eps, const float* residual_grad, float* dx,
                   void* xs_packed, int M, int K, void* stream);
/* multi-head self-attention of ResidualAttentionBlock over T=50 tokens, head dim 64: qkv[B,T,3W] -> out[B,T,W] */
int avc_vit_attention_fwd(const float* qkv, float* out, int B, int T, int width, int heads, void* stream);
/* text tower (perceptor.encode_text, main.py:276-288; clip/model.py): self-attention over T <= 128 tokens, head dim 64, with
 * the causal mask of CLIP's text transformer when `causal` != 0.  Forward only (prompts are encoded once, detached). */
int avc_text_attention_fwd(const float* qkv, float* out, int B, int T, int width, int heads, int causal, void* stream);
int avc_vit_attention_bwd(const float* qkv, const float* dout, float* dqkv, int B, int T, int width, int heads,
                          void* stream);

/* test hook: one v_mfma_f32_32x32x16_{f16,bf16} on caller-provided per-lane fragments (64 lanes x 8 / x16) */
int avc_probe_mfma(const void* a_f16, const void* b_f16, float* d, const void* a_bf16, const void* b_bf16, float* d_bf,
                   void* stream);

/* The forward pass of neural_renderer as the SMPL prior uses it (AvatarGen/AppearanceGen/models/utils.py:108-125,
 * render_one_batch -> nr.Renderer(camera_mode='look')(vertices, faces, white textures); main.py:360): nearest front-facing
 * face per pixel of an image_size x image_size grid (the caller passes the 2x super-sampled size and average-pools,
 * anti_aliasing=True), value = that face's light intensity, 0 = background.
 * faces[F,9]: per face three vertices (x, y in NDC after look + perspective, z = camera depth), the fill_back copies
 * (reversed vertex order) included by the caller; light[F]: ambient + directional intensity per face
 * (lighting.py, world space).  image[image_size, image_size], row 0 = top (rasterize.py's final flip applied).  scratch: the
 * z-buffer of 64-bit (depth bits, face index) keys the faces race into with atomicMin + the list of the faces too large for that
 * (handled tile by tile) -- avc_rasterize_scratch_bytes(F, image_size) bytes that the caller fills with 0xFF once; every call hands
 * them back that way. */
/* The same from the world-space mesh in four launches (projection look.py / perspective.py -> rasteriser at the 2 x super-sampled size on
 * faces gathered through idx [F,3] -> 2 x 2 average, optional x flip (models/utils.py:124) and three equal channels): v_world [V,3], cam =
 * device [12] (eye, x / y / z axis of the look frame), width = tan(viewing angle); ndc [V,3] scratch; out [S,S] or [S,S,3];
 * scratch: avc_rasterize_scratch_bytes(F, 2 S). */
int avc_rasterize_mesh(const float* v_world, int V, const int* idx, int F, const float* cam, float width, const float* light, int S,
                       float near_, float far_, float* ndc, float* out, int flip_x, int channels, void* scratch, void* stream);
long avc_rasterize_scratch_bytes(int F, int image_size);
int avc_rasterize_faces(const float* faces, const float* light, int F, int image_size, float near_, float far_,
                        float* image, void* scratch /* avc_rasterize_scratch_bytes, 0xFF-filled on entry; left so */, void* stream);

#ifdef __cplusplus
}
#endif
#endif
