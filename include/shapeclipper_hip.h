/* shapeclipper_hip.h -- C ABI of libshapeclipper_hip.so (hand-written gfx950 / MI355X kernels).
 *
 * Drop-in boundary for the ShapeClipper hot path.  Plain pointers and sizes only: every pointer is
 * a DEVICE pointer into memory the caller owns (nothing is allocated or freed inside, there is no
 * global state, calls are re-entrant per device), `stream` is a hipStream_t (pass torch's current
 * stream), and the caller has already made the right device current.  Return value: 0 on success,
 * otherwise the hipError_t of the failed launch (the Python side raises).
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   sc_chamfer3d_forward   <- chamfer_3D.forward  (external/chamfer3D/chamfer_cuda.cpp:17-19,31;
 *                                                  kernel chamfer3D.cu:12-154)
 *   sc_chamfer3d_backward  <- chamfer_3D.backward (chamfer_cuda.cpp:22-27,32; chamfer3D.cu:155-195)
 *   sc_sdf_forward         <- SDFNetwork.get_conditional_output (model/implicit.py:163-189) and
 *                             compute_level_grid's inner call (utils/eval_3D.py:21-38)
 *   sc_sdf_backward        <- autograd of the above incl. the create_graph=True double backward
 *   sc_render_*            <- Renderer.forward (model/renderer.py:57-185) and its autograd
 *   sc_loss_*              <- Loss.MSE_loss / mask_loss / normal_loss (model/loss.py:19-97)
 */
#ifndef SHAPECLIPPER_HIP_H
#define SHAPECLIPPER_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * Chamfer3D.  xyz1 [b,n,3], xyz2 [b,m,3] fp32 contiguous.  dist1 [b,n], dist2 [b,m]: SQUARED
 * nearest-neighbour distance d = fma(dy,dy,dx*dx) + dz*dz (the reference extension's arithmetic on this GPU: oracle/chamfer_ref.c); idx1 [b,n], idx2 [b,m] int32: index
 * of the nearest neighbour, lowest index among exact ties (chamfer3D.cu:36,126).  Outputs are
 * fully overwritten (if n==0 or m==0 they are left untouched, as the reference does).
 * Launches on `stream` (the reference used the legacy default stream).                          */
int sc_chamfer3d_forward(const float* xyz1, const float* xyz2, float* dist1, float* dist2,
                         int32_t* idx1, int32_t* idx2, int b, int n, int m, void* stream);

/* Same results (bit for bit): the targets are split over `nsplit` workgroup slices and merged with 64-bit atomicMin on
 * (distance bits, index) keys; nsplit <= 0: chosen per direction so that the launch is a whole number of full rounds of the
 * chip (5 workgroups per CU at a time).  workspace: (b*n + b*m) * 8 bytes of device scratch.   */
int sc_chamfer3d_forward_split(const float* xyz1, const float* xyz2, float* dist1, float* dist2,
                               int32_t* idx1, int32_t* idx2, int b, int n, int m, int nsplit,
                               void* workspace, void* stream);

/* Same results (bit for bit) from an exact accelerated search (csrc/chamfer_grid.hip): each cloud is binned into a uniform
 * grid and every query walks rings of cells around its own, evaluating candidates with the same expression and accepting on
 * d < best || (d == best && index < best_index); the walk stops only when no unseen target can tie or beat the best (rounding
 * of the binning and of d included).  Queries that do not terminate within a few rings (far outside the other cloud, one huge
 * cell, non-finite coordinates) are answered by the brute-force scan, so the worst case is about sc_chamfer3d_forward's cost.
 * Surface-like targets (>= 8 per occupied cell: the evaluation's clouds) are walked one wave per 64 queries of a 2 x 2 x 2 tile of cells.
 * workspace: sc_chamfer3d_grid_workspace_bytes(b, n, m) bytes of device scratch (contents irrelevant on entry).       */
long long sc_chamfer3d_grid_workspace_bytes(int b, int n, int m);
int sc_chamfer3d_forward_grid(const float* xyz1, const float* xyz2, float* dist1, float* dist2,
                              int32_t* idx1, int32_t* idx2, int b, int n, int m, void* workspace, void* stream);

/* gradxyz1 [b,n,3], gradxyz2 [b,m,3] must be ZERO-FILLED by the caller (atomicAdd accumulation,
 * chamfer3D.cu:166-171,177-178).                                                                */
int sc_chamfer3d_backward(const float* xyz1, const float* xyz2, float* gradxyz1, float* gradxyz2,
                          const float* graddist1, const float* graddist2, const int32_t* idx1,
                          const int32_t* idx2, int b, int n, int m, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Conditional SDF MLP, shipped architecture only (options/pix3d/config.yaml: 5 hidden x 64,
 * softplus(100), skips at layers 1,2, pos_enc 6, force_symmetry).
 *
 * w_pack  : kernel-ready weight image, SC_SDF_PACK_FLOATS floats (layout: mlp_tile.hpp SdfPack;
 *           produced by shapeclipper_amd.packing.pack_sdf -- PE columns in slot order, skip
 *           scale 1/sqrt(2) applied, latent columns removed).
 * cbias   : [n_images][5][64] per-image biases c_l = b_l + W_l[:, latent] z (scaled for skips).
 * points  : [n_points][3], image-major; image(i) = min(i / n_per_image, n_images-1).
 * Outputs (any may be NULL):  sdf [n_points];  grad [n_points][3] = d sdf / d point (its presence
 * selects the gradient kernel);  feat / stash_a / stash_p: tile-blocked 64-channel tensors
 * (TBL64: [ceil(n/16)][16][16][4] floats), stash_a holds 5 and stash_p 4 such tensors back to back
 * (since round 5 the ACTIVATIONS h_0..h_4 = softplus(a_l) -- a_l before -- and the adjoints p_0..p_3, kept for sc_sdf_backward[_fused]: with
 * u = exp(-100 h) the reverse passes get sp'(a) = 1 - u and sp''(a) = 100 (1 - u) u from ONE transcendental and sp(a) = h from none; the buffer
 * is opaque to callers: whatever sc_sdf_forward wrote is what the backward entry points of the SAME library build read).
 * scratch: required when grad != NULL and stash_a == NULL: 256*8*5*1024 floats of per-wave scratch.  */
#define SC_SDF_PACK_FLOATS (64*48 + 2*64*112 + 2*64*64 + 65*64 + 65)
#define SC_RGB_PACK_FLOATS (64*112 + 2*64*64 + 3*64 + 4)
#define SC_TILE_POINTS 16
int sc_sdf_forward(const float* points, const float* w_pack, const float* cbias, int n_points,
                   int n_per_image, int n_images, int symmetric, float* sdf, float* grad,
                   float* feat, float* stash_a, float* stash_p, float* scratch, void* stream);

/* Reverse pass of sc_sdf_forward incl. the second-order terms of d/dtheta[d sdf/dx] (what autograd's
 * double backward does for model/implicit.py:180-186 + model/renderer.py:101-107).
 * stash_a / stash_p: as written by sc_sdf_forward.  g_sdf [n], g_grad [n][3], g_feat (TBL64): upstream
 * gradients, any may be NULL (= zero).  Outputs: g_points [n][3] (may be NULL); ga (5 x TBL64), gp (4 x
 * TBL64, only written when g_grad != NULL), r0 (TBL64): operands of the weight-gradient GEMMs (sc_wgrad). */
int sc_sdf_backward(const float* points, const float* w_pack, int n_points, int symmetric,
                    const float* stash_a, const float* stash_p, const float* g_sdf, const float* g_grad,
                    const float* g_feat, float* g_points, float* ga, float* gp, float* r0, void* stream);

/* ---------------------------------------------------------------------------------------------
 * RGB MLP + Laplace density + alpha compositing, one wavefront per ray of 64 samples
 * (RGBNetwork.forward model/implicit.py:220-239, LaplaceDensity :65-83, Renderer.volume_rendering
 * model/renderer.py:187-209 and the per-ray reductions :117-152).
 * points [n_rays*64][3], z_vals [n_rays][64], depth_fac [n_rays], sdf [P], grad [P][3] (= d sdf/dx),
 * feat TBL64 (from sc_sdf_forward), v_pack (SC_RGB_PACK_FLOATS, packing.pack_rgb), dbias
 * [n_images][3][64], beta_param: the raw scalar parameter renderer.density.beta in device memory.
 * Outputs: rgb [n_rays][3], mask/mask_hard/depth [n_rays], normal [n_rays][3]; optional weights/alpha
 * [n_rays][64] and rgb_flat [P][3] (rgb_flat is required by the backward).                          */
int sc_rgb_composite_forward(const float* points, const float* z_vals, const float* depth_fac,
                             const float* sdf, const float* grad, const float* feat,
                             const float* v_pack, const float* dbias, const float* beta_param,
                             int n_rays, int rays_per_image, int n_images, int symmetric,
                             float beta_min, float bgcolor, float normal_pow,
                             float* rgb, float* mask, float* mask_hard, float* depth, float* normal,
                             float* weights, float* alpha, float* rgb_flat, void* stream);
/* The same, and the post-ReLU activations r0, r1, r2 of the three hidden layers of RGBNetwork (model/implicit.py:233-237) are parked in
 * rr (3 x TBL64 = 3 x n_rays * 4 * 1024 floats, layer-major; NULL = not parked) for sc_rgb_composite_backward_fused_stash.        */
int sc_rgb_composite_forward_stash(const float* points, const float* z_vals, const float* depth_fac,
                                   const float* sdf, const float* grad, const float* feat,
                                   const float* v_pack, const float* dbias, const float* beta_param,
                                   int n_rays, int rays_per_image, int n_images, int symmetric,
                                   float beta_min, float bgcolor, float normal_pow,
                                   float* rgb, float* mask, float* mask_hard, float* depth, float* normal,
                                   float* weights, float* alpha, float* rgb_flat, float* rr, void* stream);
/* round 6: sc_rgb_composite_forward_stash with the RGB network in the exact three-piece bf16 split arithmetic, weights pre-split in LDS
 * (csrc/mlp_presplit.hpp; six piece products per fp32 product on the bf16 matrix pipe, fp32 accumulation -- error against float64 that of
 * the fp32 chain).  Same operands and outputs; mask / mask_hard / depth / normal do not depend on the RGB network and are bit-identical
 * to the fp32-MFMA form's, the colours differ by fp32 rounding.                                                                         */
int sc_rgb_composite_forward_split(const float* points, const float* z_vals, const float* depth_fac,
                                   const float* sdf, const float* grad, const float* feat,
                                   const float* v_pack, const float* dbias, const float* beta_param,
                                   int n_rays, int rays_per_image, int n_images, int symmetric,
                                   float beta_min, float bgcolor, float normal_pow,
                                   float* rgb, float* mask, float* mask_hard, float* depth, float* normal,
                                   float* weights, float* alpha, float* rgb_flat, float* rr, void* stream);

/* Reverse pass.  G_* are the upstream per-ray gradients (NULL = zero).  g_beta: SC_RGB_BWD_BETA_PARTS floats, fully
 * written: one partial of d/d(raw beta parameter) per wave of the grid; the gradient is their sum in index order
 * (sc_partial_reduce(g_beta, SC_RGB_BWD_BETA_PARTS, 1, 1, out)) -- a fixed summation order, no float atomics.
 * gy (3 x TBL64), rr (3 x TBL64), gy3 [P][3]: operands for the RGB weight gradients.                          */
#define SC_RGB_BWD_BETA_PARTS 2048
int sc_rgb_composite_backward(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* gy, float* rr, float* gy3, void* stream);
/* The same with the gradient of the 3-row output layer folded in: v3_part [SC_RGB_BWD_BETA_PARTS][196] (fully written) holds, per wave of
 * the grid, the sums over its points of gy3_j * r2[ch] (dV3 [3][64]), of gy3_j (db3 [3]) and a zero; the gradient is their sum in index
 * order (sc_partial_reduce(v3_part, SC_RGB_BWD_BETA_PARTS, 196, 196, out)).  rr[2] and gy3 are then NOT written (may be NULL past rr[1]).
 * v3_part == NULL: exactly sc_rgb_composite_backward.                                                                                  */
int sc_rgb_composite_backward_v3(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* gy, float* rr, float* gy3, float* v3_part, void* stream);

/* The same reverse pass with the weight gradients of V0, V1, V2 and the per-image bias gradients formed INSIDE the kernel (round 5: four
 * weight-gradient waves beside the four chain waves, operands handed over through LDS -- the scheme of sc_sdf_backward_fused) instead of
 * the gy / rr hand-off tensors and three sc_wgrad launches.  partial: [sc_rgb_composite_backward_fused_parts(n_rays)]
 * [sc_rgb_composite_backward_fused_partial_floats(n_images)] floats, one image per workgroup = d/d(V0 | V1 | V2) (RgbPack order, 15,360
 * floats) then [n_images][3][64] bias gradients, fully written; sum them in index order (sc_partial_reduce).  v3_part as above.
 * n_images <= 256.                                                                                                             */
int sc_rgb_composite_backward_fused_parts(int n_rays);
int sc_rgb_composite_backward_fused_partial_floats(int n_images);
int sc_rgb_composite_backward_fused(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* partial, float* v3_part, void* stream);
/* sc_rgb_composite_backward_fused reading the hidden activations the forward pass parked (rr of sc_rgb_composite_forward_stash) instead
 * of recomputing the forward chain (a third of the kernel's time); same outputs, same summation orders.                            */
int sc_rgb_composite_backward_fused_stash(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* partial, float* v3_part, const float* rr, void* stream);
/* round 6: the same reverse pass with its three transposed products (V2^T, V1^T, V0f^T) and the encoding's Jacobian in the exact bf16x3
 * split arithmetic from pre-split fragments (csrc/mlp_presplit.hpp); same operands and outputs, gradients equal up to fp32 rounding.     */
int sc_rgb_composite_backward_fused_split(
    const float* points, const float* z_vals, const float* depth_fac, const float* sdf, const float* grad,
    const float* feat, const float* v_pack, const float* dbias, const float* beta_param, const float* rgb_flat,
    int n_rays, int rays_per_image, int n_images, int symmetric, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal,
    float* g_sdf, float* g_grad, float* g_feat, float* g_points, float* g_z, float* g_depth_fac, float* g_beta,
    float* partial, float* v3_part, const float* rr, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight-gradient GEMM  dW[64][nb0+nb1] = sum_points A(p) (x) [B0(p) | B1(p)]  over one or two terms.
 * Operand transform codes: 1 plain TBL64, 2 softplus(a), 3 p*softplus'(a), 4 w5row*softplus'(a),
 * 5 positional encoding of the point (48 columns), 6 g_grad-weighted PE Jacobian (48 columns).
 * Every one of the `nparts` workgroups writes its partial result at
 * partial[part*partial_stride + out_offset + row*out_ld + col]; sc_partial_reduce sums the parts.
 * rowsum (may be NULL): [nparts * 4][n_images][64], fully written: per WAVE of the grid and per image, the sum over
 * the wave's points of term 0's A operand (= the bias / per-image latent gradient of that layer) -- the operand is in
 * registers anyway; the caller adds the nparts * 4 partial images in index order (sc_partial_reduce): a fixed summation
 * order instead of float atomics.  Requires n_per_image % 16 == 0 (a 16-point tile never straddles two images).       */
int sc_wgrad(int nterms,
             const float* a0_0, const float* a1_0, int aop_0, const float* b0_0, int bop0_0, const float* b1_0, int bop1_0,
             const float* a0_1, const float* a1_1, int aop_1, const float* b0_1, int bop0_1, const float* b1_1, int bop1_1,
             const float* points, const float* g_grad, const float* w5row, int n_points, int symmetric,
             int nb0, int nb1, float* partial, int nparts, int partial_stride, int out_offset, int out_ld,
             float* rowsum, int n_per_image, int n_images, void* stream);
/* out[i] = sum_part partial[part*stride + i], i < n, in a fixed order (interleaved sequential sums combined by a fixed binary tree):
 * results do not depend on timing.  out is assigned (need not be zero-filled).                              */
int sc_partial_reduce(const float* partial, int nparts, int stride, int n, float* out, void* stream);

/* out_t[img][k][ch] += sum_{p in img} coef[p][k] * x_t[ch][p] for t < n_tensors (<= 8) TBL64 tensors in ONE
 * launch (coef NULL: K = 1, coefficient 1; else K = 3).  xs / outs are HOST arrays of device pointers; every
 * Two modes.  part != NULL (needs n_per_image % 16 == 0): fixed summation order -- every block writes a partial image into
 * `part` (sc_tbl_sum_blocks(n_points) * n_tensors * n_images * K * 64 floats of workspace) and a second launch adds them in
 * block order into outs[0], which must be ONE buffer holding all tensors back to back ([n_tensors][n_images][K][64], fully
 * written).  part == NULL: every outs[t] must be zero-filled, float atomicAdd (summation order depends on timing).
 * Used for bias / latent gradients and the 3-row output layer of the RGB net.                                   */
int sc_tbl_sum_blocks(int n_points);
int sc_tbl_sum(const float* const* xs, int n_tensors, const float* coef, int n_points, int n_per_image,
               int n_images, float* const* outs, float* part, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loss reductions of one render in one launch (Loss.MSE_loss / mask_loss / normal_loss, model/loss.py:19-97
 * as called from Graph.compute_loss, model/graph.py:224-236, 252-264).
 * rgb, rgb_t, normal, normal_t [B][R][3]; mask, mask_t [B][R]; eik [B][E] or NULL.
 * normal mask = (mask_t > 0.5) & (mask > 0.5); keep_frac = 1 - reg.normal_tol (double: the cut is
 * int(n * keep_frac) as in loss.py:62).  out4: EIGHT floats, zero-filled by the caller: [0..3] receive
 * (render MSE, IoU + mask_mse*MSE, robust normal loss, eikonal MSE), [4] is the arrival counter of the fixed-order
 * reduction over the images (results do not depend on timing), [5..7] pad; g_* receive d loss_k / d prediction;
 * g_normal_t [B][R][3] (or NULL) receives d normal loss / d normal_t -- the target is
 * camera.transform_normal(input normal, predicted pose) (graph.py:85,260), so autograd carries it into the estimator.
 * ang_ws: B*R + 4*B floats of workspace.                                                           */
int sc_loss_fused_forward(const float* rgb, const float* rgb_t, const float* mask, const float* mask_t,
                          const float* normal, const float* normal_t, const float* eik, int B, int R, int E,
                          float normal_l1, float mask_mse, double keep_frac, float* out4, float* g_rgb,
                          float* g_mask, float* g_normal, float* g_eik, float* g_normal_t, float* ang_ws,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * CLIP ViT image tower forward (replaces clip_encoder.encode_image, CLIP_anno.py:166; third-party
 * openai/CLIP, parity unpinned).  image [B][C][H][W] fp32 (already CLIP-normalised) -> out [B][proj_dim]
 * fp32 (NOT L2-normalised).  Requirements: head dim 64, D % 64 == 0, mlp % 64 == 0; any token count
 * T = (H/patch)*(W/patch)+1 whose transposed V tile fits LDS (ViT-B/32: 50, ViT-L/14: 257).
 * w_bf16: bf16 matrices, row-major [out][in], in this order: patch [D][Kp], Kp = C*patch*patch rounded up to a
 *   multiple of 64 with zero columns; per layer
 *   qkv [3D][D] (q rows, k rows, v rows), out_proj [D][D], fc1 [mlp][D], fc2 [D][mlp]; then proj [proj_dim][D].
 * w_f32: fp32 vectors in this order: class_embedding [D], position_embedding [T][D], ln_pre gamma, beta;
 *   per layer ln_1 gamma, beta, qkv bias [3D], out_proj bias, ln_2 gamma, beta, fc1 bias [mlp], fc2 bias;
 *   then ln_post gamma, beta.
 * workspace: >= sc_clip_vit_workspace_bytes(...) bytes of device memory.                               */
long long sc_clip_vit_workspace_bytes(int B, int C, int H, int W, int patch, int D, int mlp);
int sc_clip_vit_forward(const float* image, int B, int C, int H, int W, int patch, int D, int mlp, int layers,
                        int heads, int proj_dim, const uint16_t* w_bf16, const float* w_f32, float ln_eps,
                        float* out, void* workspace, long long workspace_bytes, void* stream);
/* out[M][N] = A[M][K] (bf16) * Wt[N][K]^T (bf16) + bias; epi 0: fp32 store, 1: fp32 +=, 2: quick_gelu -> bf16,
 * 3: bf16.  K % 64 == 0.                                                                               */
int sc_gemm_bf16(int epi, const uint16_t* A, const uint16_t* Wt, const float* bias, void* out, int M, int N, int K, void* stream);
int sc_f32_to_bf16(const float* x, uint16_t* y, long long n, void* stream);
/* The same three entry points with IEEE fp16 operands instead of bf16 (16-bit images / activations are fp16 bit patterns, fp32
 * accumulation, LayerNorm / softmax statistics / residual stream in fp32): the arithmetic the reference's dependency uses on a GPU --
 * clip.load("ViT-L/14", device="cuda") keeps weights and activations in fp16 with fp32 LayerNorm (CLIP_anno.py:16,166-167).  Default
 * of shapeclipper_amd.model.clip_vit.ClipVisionTower since round 3.                                          */
int sc_clip_vit_forward_f16(const float* image, int B, int C, int H, int W, int patch, int D, int mlp, int layers,
                            int heads, int proj_dim, const uint16_t* w_f16, const float* w_f32, float ln_eps,
                            float* out, void* workspace, long long workspace_bytes, void* stream);
int sc_gemm_f16(int epi, const uint16_t* A, const uint16_t* Wt, const float* bias, void* out, int M, int N, int K, void* stream);
int sc_f32_to_f16(const float* x, uint16_t* y, long long n, void* stream);
/* Small-batch form of the ViT-B layers (csrc/clip_cluster.hpp; same call site, CLIP_anno.py:166-167, at the batch the annotator
 * uses): all transformer layers in ONE launch, an image per cluster of 8 CUs of one XCD, four bounded cluster barriers per layer, no
 * chip-wide synchronisation.  It reads the layer matrices from a RE-PACKED image (same values, the order its waves consume them in: every
 * MFMA fragment one contiguous KB) that sc_clip_cluster_pack builds once per model from the row-major image above.
 * sc_clip_cluster_supported: 1 for width 768 / MLP 3072 / 12 heads / at most 64 tokens per image, else 0.
 * sc_clip_cluster_pack_elems: 16-bit values of the re-packed image (layers * 12 * 768 * 768).
 * sc_clip_cluster_pack: w16 = the tower's 16-bit image (bf16 or fp16 alike), Kp as above, out = the re-packed layers.
 * sc_clip_vit_forward_packed: sc_clip_vit_forward (fp16 == 0) / sc_clip_vit_forward_f16 (fp16 != 0) with w_cluster beside w16; batches of
 * min_b <= B <= max_b images (default 26..32, where it is the faster form; sc_clip_cluster_set_batch_range, or SC_CLIP_CLUSTER_MIN_B / _MAX_B
 * in the environment at load time) on a device with >= 256 CUs take the cluster form, everything else the launch-per-operation form
 * (w_cluster may be NULL then).  A device that cannot hold the grid at once ends every wait after 0.2 s and returns NaN embeddings.       */
int sc_clip_cluster_supported(int D, int mlp, int heads, int tokens);
int sc_clip_cluster_set_batch_range(int min_b, int max_b);
long long sc_clip_cluster_pack_elems(int layers);
int sc_clip_cluster_pack(const uint16_t* w16, int Kp, int layers, uint16_t* out, void* stream);
int sc_clip_vit_forward_packed(const float* image, int B, int C, int H, int W, int patch, int D, int mlp, int layers, int heads,
                               int proj_dim, const uint16_t* w16, const float* w_f32, const uint16_t* w_cluster, int fp16,
                               float ln_eps, float* out, void* workspace, long long workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Ray sampling (UniformSampler.get_z_vals + point generation, model/renderer.py:13-37,84-86).
 * cam_loc, ray_dirs [n_rays][3]; scale_dist [n_images]; u [n_rays][64] stratified jitter in [0,1) or NULL
 * (evaluation: plain linspace).  Outputs z_vals [n_rays][64], points [n_rays*64][3].  Arithmetic follows the
 * reference's fp32 op order (no fma contraction) so z_vals / points are bit-identical to the torch ops.   */
int sc_ray_sample_forward(const float* cam_loc, const float* ray_dirs, const float* scale_dist, const float* u,
                          int n_rays, int rays_per_image, int n_images, float cam_dist, float* z_vals,
                          float* points, void* stream);
/* adjoint: g_points [P][3] (+ g_z_extra [n_rays][64] from the compositing, may be NULL) -> g_cam_loc,
 * g_ray_dirs [n_rays][3], g_scale_dist [n_rays]: per-RAY contribution; d/d scale_dist[b] is its sum over the
 * rays of image b (left to the caller: one tiny reduction instead of 16K contended atomics).            */
int sc_ray_sample_backward(const float* ray_dirs, const float* z_vals, const float* g_points,
                           const float* g_z_extra, int n_rays, int rays_per_image, int n_images, float cam_dist,
                           float* g_cam_loc, float* g_ray_dirs, float* g_scale_dist, void* stream);
/* The same pair with the eikonal sample points of a training render (model/renderer.py:154-165) produced / differentiated on the way:
 * eik_points [n_images][2 rays_per_image][3] = per image the rays_per_image uniform points (eik_uniform [n_rays][3], copied) followed by the
 * near-surface point of every ray = its sample eik_idx[ray] (int64 [n_rays]; cam_loc + z_eik ray_dir, bit-identical to that entry of
 * `points`).  backward: g_eik_points (may be NULL) adds the near points' gradients to the samples they are; g_points may be NULL (zeros).
 * n_rays == n_images * rays_per_image.                                                                                              */
int sc_ray_sample_forward_eik(const float* cam_loc, const float* ray_dirs, const float* scale_dist, const float* u, const long long* eik_idx,
                              const float* eik_uniform, int n_rays, int rays_per_image, int n_images, float cam_dist, float* z_vals,
                              float* points, float* eik_points, void* stream);
int sc_ray_sample_backward_eik(const float* ray_dirs, const float* z_vals, const float* g_points, const float* g_z_extra, const long long* eik_idx,
                               const float* g_eik_points, int n_rays, int rays_per_image, int n_images, float cam_dist, float* g_cam_loc,
                               float* g_ray_dirs, float* g_scale_dist, void* stream);

/* One render in a single call (Renderer.forward, model/renderer.py:57-152):
 * sc_ray_sample_forward -> sc_sdf_forward -> sc_rgb_composite_forward.  z_vals, points, sdf, grad, feat and
 * scratch are caller-provided work buffers (sizes as in the three entry points).  Inference: stash_a = stash_p =
 * rgb_flat = NULL.  Training: pass stash_a (5 x TBL64), stash_p (4 x TBL64) and rgb_flat [P][3]; together with
 * z_vals / points / sdf / grad / feat they are what sc_render_backward needs (scratch may then be NULL).    */
/* (sc_render_forward chains sc_ray_sample_forward, sc_sdf_forward and sc_rgb_composite_forward -- the fp32-MFMA forms; the round-6 split
 *  forms sc_sdf_forward_stream / sc_rgb_composite_forward_split are what the Python host calls one by one.)                          */
int sc_render_forward(const float* cam_loc, const float* ray_dirs, const float* depth_fac, const float* scale_dist,
                      const float* u, const float* sdf_pack, const float* sdf_cbias, const float* rgb_pack,
                      const float* rgb_dbias, const float* beta_param, int n_rays, int rays_per_image, int n_images,
                      int symmetric, float cam_dist, float beta_min, float bgcolor, float normal_pow,
                      float* rgb, float* mask, float* mask_hard, float* depth, float* normal,
                      float* z_vals, float* points, float* sdf, float* grad, float* feat, float* scratch,
                      float* stash_a, float* stash_p, float* rgb_flat, void* stream);

/* The whole reverse pass of one training render (what autograd does for model/renderer.py:57-185 and the MLPs of
 * model/implicit.py:138-239, incl. the double backward through d sdf/dx): sc_rgb_composite_backward, the RGB
 * weight-gradient GEMMs, sc_sdf_backward, the SDF weight-gradient GEMMs and sc_ray_sample_backward in one call.
 * Inputs: the forward's operands and saved tensors (see sc_render_forward) and the upstream gradients of the per-ray
 * outputs G_rgb [n_rays][3], G_mask, G_depth [n_rays], G_normal [n_rays][3] (NULL = zero), G_z_extra [n_rays][64]
 * (extra gradient on z_vals, e.g. from the eikonal near-surface samples; may be NULL).
 * Outputs (all fully written): g_sdf_pack [SC_SDF_PACK_FLOATS], g_cbias [5][n_images][64], g_rgb_pack
 * [SC_RGB_PACK_FLOATS], g_dbias [3][n_images][64], g_beta [1], and per ray g_cam_loc [.][3], g_ray_dirs [.][3],
 * g_scale_dist [.] (sum over the rays of an image = d/d scale_dist), g_depth_fac [.].
 * workspace: sc_render_backward_workspace_bytes(n_rays) bytes of device memory (~17 TBL64 tensors).          */
long long sc_render_backward_workspace_bytes(int n_rays);
int sc_render_backward(
    const float* ray_dirs, const float* depth_fac, const float* sdf_pack, const float* rgb_pack, const float* rgb_dbias,
    const float* beta_param, const float* z_vals, const float* points, const float* sdf, const float* grad, const float* feat,
    const float* stash_a, const float* stash_p, const float* rgb_flat, int n_rays, int rays_per_image, int n_images,
    int symmetric, float cam_dist, float beta_min, float bgcolor, float normal_pow,
    const float* G_rgb, const float* G_mask, const float* G_depth, const float* G_normal, const float* G_z_extra,
    float* g_sdf_pack, float* g_cbias, float* g_rgb_pack, float* g_dbias, float* g_beta,
    float* g_cam_loc, float* g_ray_dirs, float* g_scale_dist, float* g_depth_fac,
    void* workspace, long long workspace_bytes, void* stream);

/* compute_level_grid (utils/eval_3D.py:9-38): SDF on linspace(lo,hi,n_axis)^3 ('ij' order) for every image.
 * points_ws: [n_images*n_axis^3][3] floats of workspace; level: [n_images][n_axis][n_axis][n_axis].       */
int sc_sdf_grid_forward(const float* sdf_pack, const float* sdf_cbias, float lo, float hi, int n_axis, int n_images,
                        int symmetric, float* points_ws, float* level, void* stream);
/* round 6: the same grid through the value-only chain in the exact three-piece bf16 split arithmetic with the weights pre-split in LDS
 * (csrc/sdf_value_split.hip: six piece products per fp32 product on v_mfma_f32_16x16x32_bf16, fp32 accumulation -- the arithmetic of the
 * trunk convolutions; error against float64 that of the fp32 chain).  sc_sdf_value_forward_split: the same chain on given points
 * (operands as sc_sdf_forward; sdf [n_points] is the only output).                                                                       */
int sc_sdf_grid_forward_split(const float* sdf_pack, const float* sdf_cbias, float lo, float hi, int n_axis, int n_images,
                              int symmetric, float* points_ws, float* level, void* stream);
int sc_sdf_value_forward_split(const float* points, const float* w_pack, const float* cbias, int n_points, int n_per_image,
                               int n_images, int symmetric, float* sdf, void* stream);
/* round 6: sc_sdf_forward (value, feature, d sdf/dx, training stashes: same operands and outputs) in the same split arithmetic with the
 * pre-split weight fragments STREAMED through LDS one layer at a time (csrc/sdf_fwd_stream.hip: 324 KiB of fragments incl. the transposed
 * set of the adjoint sweep; the 8 waves of a workgroup walk the chain in lock step, two LDS buffers).
 *   sc_sdf_stream_pack_bytes()      bytes of the streamed image (331,776)
 *   sc_sdf_stream_pack              w_pack (fp32 SdfPack image) -> the streamed image; once per weight update
 *   sc_sdf_forward_stream           grad required; stash_a, stash_p and feat all given = the training render (else per-wave `scratch`
 *                                   as in sc_sdf_forward); w_pack is read for the fp32 sdf row of the output layer and its biases      */
long long sc_sdf_stream_pack_bytes(void);
int sc_sdf_stream_pack(const float* w_pack, void* w_stream, void* stream);
int sc_sdf_forward_stream(const float* points, const void* w_stream, const float* w_pack, const float* cbias, int n_points,
                          int n_per_image, int n_images, int symmetric, float* sdf, float* grad, float* feat,
                          float* stash_a, float* stash_p, float* scratch, void* stream);

/* Workgroup-cooperative reverse pass of the SDF MLP (csrc/sdf_bwdw.hip): what sc_sdf_backward + the eight sc_wgrad launches
 * + sc_tbl_sum of the SDF network do, in ONE launch and without the Ga/Gp/r0 hand-off tensors (chain waves and
 * weight-gradient waves of a workgroup exchange operands through LDS).  Requires g_grad and stash_p (the d sdf/dx output
 * is differentiated: every training render) and n_per_image % 16 == 0.
 *   park     workspace, sc_sdf_backward_fused_parts(n_points) * 4 * 4 * 1024 floats (per-wave scratch, stays in L2)
 *   partial  [sc_sdf_backward_fused_parts(n_points)][S], S = sc_sdf_backward_fused_partial_floats(n_images): one partial image
 *            per workgroup, fully written: SdfPack floats of d/d(w_pack) followed (n_images <= 256) by [n_images][5][64] of
 *            d/d(per-image biases); sum them with sc_partial_reduce(partial, parts, S, S, out) -- a fixed summation order
 *   g_cbias  only for n_images > 256 (S == SdfPack floats): [n_images][5][64], zero-filled by the caller, float atomicAdd;
 *            ignored (may be NULL) otherwise
 *   g_points [n_points][3] or NULL.                                                                              */
int sc_sdf_backward_fused_parts(int n_points);
int sc_sdf_backward_fused_partial_floats(int n_images);
int sc_sdf_backward_fused(const float* points, const float* w_pack, int n_points, int n_per_image, int n_images,
                          int symmetric, const float* stash_a, const float* stash_p, const float* g_sdf,
                          const float* g_grad, const float* g_feat, float* g_points, float* park, float* partial,
                          float* g_cbias, void* stream);

/* backward of sc_loss_fused_forward: scales the stored gradients in place by the upstream dL/dloss_k (G4[4],
 * device memory).  g_eik and g_normal_t (n_normal elements, scaled by G4[2]) may be NULL.              */
int sc_loss_fused_backward(const float* G4, float* g_rgb, long long n_rgb, float* g_mask, long long n_mask,
                           float* g_normal, long long n_normal, float* g_eik, long long n_eik, float* g_normal_t,
                           void* stream);

/* ---- encoder glue: BatchNorm2d fused with the residual add / ReLU / stem max-pool around it --------
 * Replaces, between the MIOpen convolutions of the reference's torchvision ResNet-18/34 (model/graph.py:52-54,
 * model/view_estimator.py:40-42; torchvision BasicBlock.forward), nn.BatchNorm2d + `out += identity` + ReLU.
 * All tensors NCHW fp32, contiguous, 16-byte aligned.  N images, C channels, HW = H*W.
 * sc_bn_splits: number S of per-channel partial blocks the kernels use for one group.                       */
int sc_bn_splits(int N, int C);

/* y = [relu]( gamma*(x-mean)*rstd + beta [+ res] ).  training != 0: batch statistics (biased variance), running
 * statistics updated with `momentum` (unbiased variance) and *n_tracked += groups (both may be NULL); training == 0:
 * the running statistics are used.  groups = G >= 1 (N % G == 0): the batch is G independent sub-batches of N/G
 * images (what the reference feeds through the network in G consecutive calls): statistics per sub-batch, running
 * statistics updated G times in order.  save_mean / save_rstd [G][C] are written for the backward.
 * `partial` must hold 2*(2048 + C*G) floats.                                                                  */
int sc_bn_act_forward(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                      float* save_mean, float* save_rstd, float* run_mean, float* run_var, int64_t* n_tracked,
                      float* partial, int N, int C, int HW, int relu, int training, int groups, float eps,
                      float momentum, void* stream);

/* Backward of sc_bn_act_forward.  y (the forward output) is needed only when a residual was added AND relu != 0
 * (pass NULL otherwise: the ReLU mask is then recomputed from x).  dres (may be NULL) receives the gradient of the
 * residual input (= the ReLU-masked dy); dx may be NULL; dgamma / dbeta [C] are always written.               */
int sc_bn_act_backward(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                       const float* mean, const float* rstd, float* partial, float* dx, float* dres, float* dgamma,
                       float* dbeta, int N, int C, int HW, int relu, int training, int groups, void* stream);

/* ResNet stem: y = maxpool3x3/stride2/pad1( relu( bn(x) ) ), x [N,C,H,W] -> y [N,C,Ho,Wo], Ho = (H-1)/2+1.
 * idx [N,C,Ho,Wo] int32: argmax position h*W+w (first maximum in scan order, as torch.nn.MaxPool2d).          */
int sc_bn_relu_pool_forward(const float* x, const float* gamma, const float* beta, float* y, int* idx,
                            float* save_mean, float* save_rstd, float* run_mean, float* run_var, int64_t* n_tracked,
                            float* partial, int N, int C, int H, int W, int training, int groups, float eps,
                            float momentum, void* stream);
int sc_bn_relu_pool_backward(const float* dy, const int* idx, const float* x, const float* gamma, const float* beta,
                             const float* mean, const float* rstd, float* partial, float* dx, float* dgamma,
                             float* dbeta, int N, int C, int H, int W, int training, int groups, void* stream);

/* ---- iso-surface extraction for evaluation (replaces mcubes.marching_cubes, utils/eval_3D.py:125) -----
 * level [n_images][n_axis]^3 fp32.  sc_isosurface_count writes counts[cube] (cube = ((b*Nc + x)*Nc + y)*Nc + z,
 * Nc = n_axis-1): number of triangles of the marching-tetrahedra surface {level = iso} inside the cube (0..12).
 * The caller turns counts into exclusive offsets (int64 prefix sum) and allocates tris [total][3][3];
 * sc_isosurface_emit writes the triangles of every cube at its offset, vertices in grid-index units.           */
int sc_isosurface_count(const float* level, int n_images, int n_axis, float iso, int* counts, void* stream);
int sc_isosurface_emit(const float* level, int n_images, int n_axis, float iso, const int* counts,
                       const long long* offsets, float* tris, void* stream);

/* sc_marching_cubes_*: the same contract with marching CUBES (the algorithm of the reference's PyMCubes call, utils/eval_3D.py:
 * 123-153): up to 5 triangles per cube from the 256-case table of csrc/mc_table.hpp (generated by tools/gen_mc_table.py).  The
 * vertex set is the one every marching-cubes implementation produces -- one vertex per grid edge whose end values lie on
 * different sides of iso (inside = value < iso), at the linear interpolation point -- the triangulation of ambiguous faces
 * (diagonal corners inside) cuts off the inside corners, identically on both sides of the face.                          */
int sc_marching_cubes_count(const float* level, int n_images, int n_axis, float iso, int* counts, void* stream);
int sc_marching_cubes_emit(const float* level, int n_images, int n_axis, float iso, const int* counts,
                           const long long* offsets, float* tris, void* stream);

/* Block form of both (what the Python side uses): a workgroup owns 1,024 consecutive cubes of one image.
 *   sc_isosurface_blocks_per_image(n_axis)   blocks per image = ceil((n_axis-1)^3 / 1024)  (-1: n_axis outside 2..1024)
 *   *_block_count    block_counts [n_images * blocks_per_image] int: triangles of each block
 *   sc_isosurface_block_scan   (round 6) block_offsets [n_images * blocks_per_image + 1] = exclusive int64 prefix sum of block_counts with
 *                    the total as its last entry, per_image [n_images] = triangles of each image: one launch of one workgroup between
 *                    the two passes (the caller reads per_image to size `tris`).
 *   *_block_emit     block_offsets: that array (n_images * blocks_per_image + 1 entries: a block whose neighbours' offsets are equal holds
 *                    no triangle and is left at once); the kernel recomputes the per-cube counts and takes
 *                    their prefix inside the workgroup -- no per-cube count / offset arrays (12 bytes per cube), and the prefix sum
 *                    between the launches runs over 1/1024 of the values.  Same triangles in the same order as the per-cube form. */
int sc_isosurface_blocks_per_image(int n_axis);
int sc_isosurface_block_scan(const int* block_counts, int n_images, int n_axis, long long* block_offsets, long long* per_image, void* stream);
int sc_isosurface_block_count(const float* level, int n_images, int n_axis, float iso, int* block_counts, void* stream);
int sc_isosurface_block_emit(const float* level, int n_images, int n_axis, float iso, const long long* block_offsets, float* tris,
                             void* stream);
int sc_marching_cubes_block_count(const float* level, int n_images, int n_axis, float iso, int* block_counts, void* stream);
int sc_marching_cubes_block_emit(const float* level, int n_images, int n_axis, float iso, const long long* block_offsets, float* tris,
                                 void* stream);
/* round 6, what the Python side runs for marching cubes: the count pass also leaves the case index of every cube (masks
 * [n_images * (n_axis-1)^3] bytes, caller-allocated), the emit pass reads them instead of re-reading 8 corners per cube and deals the
 * vertices of a 256-cube group out to all lanes (coalesced 12-byte stores).  Same triangles, same order, bit for bit.                   */
int sc_marching_cubes_block_count_masks(const float* level, int n_images, int n_axis, float iso, int* block_counts, unsigned char* masks,
                                        void* stream);
int sc_marching_cubes_block_emit_masks(const float* level, int n_images, int n_axis, float iso, const long long* block_offsets,
                                       const unsigned char* masks, float* tris, void* stream);

/* ---- camera algebra of a render -------------------------------------------------------------------
 * sc_camera_rays_*: utils/camera.py:157-196 (get_center_and_ray on the rendered pixels only) + the normalisation of
 * model/renderer.py:69-76.  pose [n_images][3][4] = [R|t] world->camera, intr [n_images][3][3], ray_idx
 * [n_images][rays_per_image] int64 pixel indices (y*W + x) or NULL (= all pixels 0..rays_per_image-1).  Outputs per
 * ray: cam_loc [.][3] (camera centre, repeated), ray_dirs [.][3] (unit), depth_fac [.] (= |dir| / |unnormalised ray|).
 * Perspective model.  The backward returns d/d pose and d/d intr (g_* inputs may be NULL = zero).                */
int sc_camera_rays_forward(const float* pose, const float* intr, const long long* ray_idx, int n_images,
                           int rays_per_image, int image_width, float* cam_loc, float* ray_dirs, float* depth_fac,
                           void* stream);
int sc_camera_rays_backward(const float* pose, const float* intr, const long long* ray_idx, int n_images,
                            int rays_per_image, int image_width, const float* g_cam_loc, const float* g_ray_dirs,
                            const float* g_depth_fac, float* g_pose, float* g_intr, void* stream);

/* sc_pose_from_trig_*: model/graph.py:272-293 (pred_pose) with utils/camera.py:105-155,198-211: (cos,sin) of azimuth,
 * elevation, roll [n_images][2], scale_focal, scale_dist [n_images] -> pose [n_images][3][4] = [Rz Rx Ry P | (0,0,
 * cam_dist*scale_dist)], intr [n_images][3][3] = [[f W,0,W/2],[0,f H,H/2],[0,0,1]], f = focal*scale_focal.        */
int sc_pose_from_trig_forward(const float* azim, const float* elev, const float* theta, const float* scale_focal,
                              const float* scale_dist, int n_images, float cam_dist, float focal, int image_width,
                              int image_height, float* pose, float* intr, void* stream);
int sc_pose_from_trig_backward(const float* azim, const float* elev, const float* theta, const float* scale_focal,
                               const float* scale_dist, int n_images, float cam_dist, float focal, int image_width,
                               int image_height, const float* g_pose, const float* g_intr, float* g_azim,
                               float* g_elev, float* g_theta, float* g_scale_focal, float* g_scale_dist, void* stream);

/* ---- the [n_images]-sized arithmetic around the view estimator (csrc/camera_prior.hip): one launch each way where the
 * reference spends dozens of [B]-shaped torch operators.
 * sc_estimator_head_*: model/view_estimator.py:62-75.  trig [n_rows][6] (extr_fc output), size_lin / persp_lin [n_rows]
 *   (size_fc / perspect_fc outputs) -> azim, elev, theta [n_rows][2] = F.normalize of the three pairs (eps 1e-12),
 *   scale_focal = 1 + tanh(persp_lin) * persp_range, scale_dist = (1 + tanh(size_lin) * size_range) * scale_focal.
 *   Backward: the rows are n_groups stacked image sets of n_rows / n_groups rows (<= 8 groups); grads is a HOST array of
 *   5 * n_groups device pointers (group-major: azim, elev, theta, scale_focal, scale_dist of that group, each
 *   [rows][2] or [rows]; NULL = that output was not differentiated).  Returns -1 for a bad group count.            */
int sc_estimator_head_forward(const float* trig, const float* size_lin, const float* persp_lin, int n_rows,
                              float size_range, float persp_range, float* azim, float* elev, float* theta,
                              float* scale_focal, float* scale_dist, void* stream);
int sc_estimator_head_backward(const float* trig, const float* size_lin, const float* persp_lin, int n_rows,
                               float size_range, float persp_range, const float* const* grads, int n_groups,
                               float* g_trig, float* g_size_lin, float* g_persp_lin, void* stream);

/* sc_camera_prior_*: model/loss.py:99-167 -- cam_margin_loss (elevation and roll ranges in degrees, eps = margin_eps),
 *   cam_uniform_loss (sorted (cos, sin, cos*sin) of the azimuth against the sorted uniform grid; emd_p 1 or 2) and
 *   cam_sym_loss against the estimator's outputs on the mirrored images (flip_*), all [n_images][2].
 *   out [3] = (cam_margin, cam_uniform, cam_sym); grads [6][n_images][2] = d margin/d elev, d margin/d theta,
 *   d uniform/d azim, d sym/d azim, d sym/d elev, d sym/d theta.  One workgroup; n_images <= sc_camera_prior_max_images()
 *   (1024), otherwise -1.  The backward scales by the upstream gradients of the three values (device scalars, NULL = 0)
 *   and also returns d sym/d flip_*.                                                                               */
int sc_camera_prior_max_images(void);
int sc_camera_prior_forward(const float* azim, const float* elev, const float* theta, const float* flip_azim,
                            const float* flip_elev, const float* flip_theta, int n_images, float elev_lo, float elev_hi,
                            float theta_lo, float theta_hi, float margin_eps, int emd_p, float* out, float* grads,
                            void* stream);
int sc_camera_prior_backward(const float* grads, int n_images, const float* G_margin, const float* G_uniform,
                             const float* G_sym, float* g_azim, float* g_elev, float* g_theta, float* g_flip_azim,
                             float* g_flip_elev, float* g_flip_theta, void* stream);

/* sc_transform_normal_*: utils/camera.py:98-103 -- out[b][r][:] = normals[b][r][:] @ R_b with R_b the rotation block of
 *   pose [n_images][3][4].  Backward: g_pose [n_images][3][4] (translation column zero); the normals are data.      */
int sc_transform_normal_forward(const float* normals, const float* pose, int n_images, int n_per_image, float* out,
                                void* stream);
int sc_transform_normal_backward(const float* normals, const float* g_out, int n_images, int n_per_image, float* g_pose,
                                 void* stream);

/* sc_loss_total_*: model/runner.py:294-305 -- total = sum_k weights[k] * *values[k] added in index order (<= 16 terms;
 *   values is a HOST array of device scalars, weights a HOST array), bad = 1 if any term is NaN/Inf.  Backward:
 *   g_values[k] = weights[k] * *G.                                                                                 */
int sc_loss_total_forward(const float* const* values, const float* weights, int n_terms, float* total, unsigned char* bad,
                          void* stream);
int sc_loss_total_backward(const float* weights, int n_terms, const float* G, float* g_values, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolution of the ResNet-18/34 trunks (torchvision BasicBlock conv1/conv2 behind
 * model/graph.py:50-54 and model/view_estimator.py:40-42), NCHW fp32, on the fp32 matrix pipe (csrc/conv3x3.hip).
 * hw = side of the square feature map: 56, 28, 14 or 7 (anything else returns -1: the caller keeps MIOpen for it).
 *   sc_conv3x3_pack_floats  size of the kernel-ready weight image (floats), -1 for an unsupported shape
 *   sc_conv3x3_pack         w [cout][cin][3][3] -> w_pack.  transpose_flip = 1 writes the filter of the backward-data pass,
 *                           w'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx], for w stored [cin][cout][3][3] in THIS call's naming
 *                           (cin = channels of dL/dy, cout = channels of dL/dx)
 *   sc_conv3x3_forward      out [batch][cout][hw][hw] = conv(x [batch][cin][hw][hw], w), fully overwritten.
 *                           dL/dx = sc_conv3x3_forward(dL/dy, pack(w, transpose_flip = 1)).
 *                           workspace: sc_conv3x3_workspace_floats(hw) floats of device scratch (partial tiles of the
 *                           workgroups that share the last round's tiles; summed in a fixed order by a second launch).
 *   sc_conv3x3_pack_multi   the same images for MANY filters in one launch (a network's filters change once per optimizer
 *                           step): table = n <= 128 rows of 6 int64 on the device, {address of w, first float of the image inside dst,
 *                           cin, cout, sc_conv3x3_tile_channels(hw), transpose_flip}, rows sorted by first float; total = floats
 *                           of all images.  Image e is then  dst + row[1]  and has sc_conv3x3_pack_floats(cin, cout, hw) floats.  */
long long sc_conv3x3_pack_floats(int cin, int cout, int hw);
long long sc_conv3x3_workspace_floats(int hw);
int sc_conv3x3_tile_channels(int hw);
int sc_conv3x3_pack_multi(const long long* table, int n, float* dst, long long total, void* stream);
/* The same table, for the case that EVERY row is a split image (transpose_flip bit 1) with 64-channel tiles (sc_conv3x3_tile_channels_split
 * = 64): one workgroup per 64 x 16 x 9 unit of a filter (a pair of K-steps: nine tap-pair images, see csrc/conv3x3.hip), contiguous reads
 * and writes (byte-identical images); total % 13824 must be 0.                                                                          */
int sc_conv3x3_pack_multi_units(const long long* table, int n, float* dst, long long total, void* stream);
int sc_conv3x3_pack(const float* w, float* w_pack, int cin, int cout, int hw, int transpose_flip, void* stream);
int sc_conv3x3_forward(const float* x, const float* w_pack, float* out, float* workspace, int batch, int cin, int cout, int hw,
                       void* stream);

/* The same convolution with fp32-ACCURATE products on the bf16 matrix pipe (opt-in, `--hip.conv3x3_split`): every operand is split
 * exactly into three bf16 pieces (x = p0 + p1 + p2), the six products a_p b_q with p + q <= 2 are accumulated in fp32 (what is dropped
 * is < 2^-23 |a||b| per product); 6 bf16 MFMAs replace 16 fp32-MFMA passes.  Same tensors and semantics as sc_conv3x3_forward; the
 * filter image is written by sc_conv3x3_pack with bit 1 of transpose_flip set (or flags | 2 in a sc_conv3x3_pack_multi row) and has
 * sc_conv3x3_pack_floats_split floats; tile width / workspace: the _split queries.                                                   */
long long sc_conv3x3_pack_floats_split(int cin, int cout, int hw);
long long sc_conv3x3_workspace_floats_split(int hw);
int sc_conv3x3_tile_channels_split(int hw);
int sc_conv3x3_forward_split(const float* x, const float* w_pack, float* out, float* workspace, int batch, int cin, int cout, int hw,
                             void* stream);
/* out = conv(x, w) + addend, the sum formed in the store epilogue (addend: a tensor of the output's shape; split != 0: w_pack is a split image).
 * Used for d L / d x of a residual block: backward-data of conv1 + the gradient of the skip branch (model/graph.py's torchvision BasicBlock:
 * autograd's accumulation of the two uses of the block input).                                                                      */
int sc_conv3x3_forward_add(const float* x, const float* w_pack, const float* addend, float* out, float* workspace, int batch, int cin,
                           int cout, int hw, int split, void* stream);

/* 3x3 / stride 2 / pad 1 (BasicBlock.conv1 of layer2-4; torchvision layer{2,3,4}.0.conv1 behind model/graph.py:50-54 and
 * model/view_estimator.py:40-42), same kernel family: hw = side of the INPUT map (56, 28 or 14), out [batch][cout][hw/2][hw/2].
 * Forward filter image: sc_conv3x3_pack with bit 2 (value 4) of transpose_flip set, sc_conv3x3s2_pack_floats floats.
 * Backward-data (sc_conv3x3s2_backward_data: gx [batch][cin][hw][hw] from gy [batch][cout][hw/2][hw/2]): four stride-1 sub-convolutions
 *   over the gradient map, one per output parity, each with the 1 / 2 / 2 / 4 filter taps that reach it -- every product of the transposed
 *   convolution once; exact three-piece bf16 operand splits with fp32 accumulation (the arithmetic of sc_conv3x3_forward_split).  Filter
 *   image: sc_conv3x3s2_bd_pack from the FORWARD filter w [cout][cin][3][3], sc_conv3x3s2_bd_pack_floats floats; workspace
 *   sc_conv3x3s2_bd_workspace_floats(hw) floats.  cout % 8 == 0.
 * Backward-weight (sc_conv3x3s2_wgrad: dw [cout][cin][3][3] from gy and x [batch][cin][hw][hw]): csrc/conv3x3_wgrad.hip with stride-2 patch
 *   addressing, fp32 MFMA, fixed summation order; workspace sc_conv3x3_wgrad_workspace_floats(cin, cout) floats; cin, cout % 64 == 0. */
long long sc_conv3x3s2_bd_pack_floats(int cin, int cout, int hw);
long long sc_conv3x3s2_bd_workspace_floats(int hw);
int sc_conv3x3s2_bd_pack(const float* w, float* w_pack, int cin, int cout, int hw, void* stream);
int sc_conv3x3s2_backward_data(const float* gy, const float* w_pack, float* gx, float* workspace, int batch, int cin, int cout, int hw,
                               void* stream);
int sc_conv3x3s2_wgrad(const float* gy, const float* x, float* dw, float* workspace, int batch, int cin, int cout, int hw, void* stream);
long long sc_conv3x3s2_pack_floats(int cin, int cout, int hw);
long long sc_conv3x3s2_workspace_floats(int hw);
int sc_conv3x3s2_forward(const float* x, const float* w_pack, float* out, float* workspace, int batch, int cin, int cout, int hw,
                         void* stream);

/* Weight gradient of the same convolution (csrc/conv3x3_wgrad.hip):  dw [cout][cin][3][3] = sum over the batch of gy (x) shifted x,
 * gy [batch][cout][hw][hw], x [batch][cin][hw][hw], fully overwritten, fixed summation order.  cin and cout must be multiples of 64
 * (otherwise hipErrorInvalidValue; sc_conv3x3_wgrad_workspace_floats returns -1): the caller keeps MIOpen for other shapes.
 * workspace: sc_conv3x3_wgrad_workspace_floats(cin, cout) floats (one partial 64 x 64 x 9 block per workgroup).                       */
long long sc_conv3x3_wgrad_workspace_floats(int cin, int cout);
int sc_conv3x3_wgrad(const float* gy, const float* x, float* dw, float* workspace, int batch, int cin, int cout, int hw, void* stream);
/* The same gradient with fp32-accurate products on the bf16 matrix pipe (the arithmetic of sc_conv3x3_forward_split: exact three-way
 * bf16 split of BOTH operands, the six piece products with p + q <= 2, smallest first, fp32 accumulate); same arguments and workspace. */
int sc_conv3x3_wgrad_split(const float* gy, const float* x, float* dw, float* workspace, int batch, int cin, int cout, int hw, void* stream);

/* The stem of the trunks (torchvision ResNet.conv1: 7x7 / stride 2 / pad 3, 3 -> 64 channels, 224 x 224 inputs only; csrc/conv_stem.hip):
 *   sc_conv_stem_forward   out [batch][64][112][112] = conv(x [batch][3][224][224], w [64][3][7][7]), fully overwritten
 *   sc_conv_stem_wgrad     dw [64][3][7][7] = weight gradient from gy [batch][64][112][112] and x, fixed summation order;
 *                          workspace: sc_conv_stem_wgrad_workspace_floats() floats.  (The input is data: no backward-data.)          */
long long sc_conv_stem_wgrad_workspace_floats(void);
int sc_conv_stem_forward(const float* x, const float* w, float* out, int batch, void* stream);
int sc_conv_stem_wgrad(const float* gy, const float* x, float* dw, float* workspace, int batch, void* stream);

/* The 1x1 / stride 2 shortcut convolutions of the trunks (torchvision BasicBlock.downsample[0]; csrc/conv1x1s2.hip), NCHW fp32,
 * hin = side of the (square, even) input map, cin and cout multiples of 64 (otherwise hipErrorInvalidValue / -1):
 *   sc_conv1x1s2_forward        out [batch][cout][hin/2][hin/2] = sum_ci w[cout][cin] x[batch][cin][2y][2x]
 *   sc_conv1x1s2_backward_data  gx [batch][cin][hin][hin] from gy [batch][cout][hin/2][hin/2]: fully written (zeros at odd positions)
 *   sc_conv1x1s2_wgrad          dw [cout][cin] from gy and x, fixed summation order; workspace: the _workspace_floats query.        */
long long sc_conv1x1s2_wgrad_workspace_floats(int cin, int cout);
int sc_conv1x1s2_forward(const float* x, const float* w, float* out, int batch, int cin, int cout, int hin, void* stream);
int sc_conv1x1s2_backward_data(const float* gy, const float* w, float* gx, int batch, int cin, int cout, int hin, void* stream);
int sc_conv1x1s2_wgrad(const float* gy, const float* x, float* dw, float* workspace, int batch, int cin, int cout, int hin, void* stream);

/* ---- one call per BasicBlock (csrc/block.hip) ------------------------------------------------------------------------------------------
 * torchvision BasicBlock with stride 1 and no shortcut convolution, as the reference's ResNet-18 / 34 trunks run it (model/graph.py:50-54,
 * model/view_estimator.py:40-42):   out = relu( bn2( conv2( relu( bn1( conv1(x) ) ) ) ) + x ),   x, y1, a1, y2, out: [batch][channels][hw][hw].
 * Host glue over the entry points above (same launches, same order, same results): the caller fills one argument block per call.
 *   forward : reads x, pf1, pf2 (filter images of sc_conv3x3_pack, forward orientation), the BatchNorm parameters / running statistics
 *             (rm / rv / nt may be NULL); writes y1 = conv1(x), a1 = relu(bn1(y1)), y2 = conv2(a1), out, st1, st2 ([2][groups][channels]:
 *             save_mean | save_rstd).
 *   backward: reads d_out and what the forward wrote, pb1 / pb2 (backward-data filter images); writes dy2, da1, dy1 (scratch the caller
 *             provides), dgb1 / dgb2 ([2][channels]: dgamma | dbeta), gw1 / gw2 ([channels][channels][3][3], NULL: skipped) and -- when
 *             need_dx -- dres (scratch) and dx = dL/dx (both uses of x summed).
 * Workspaces: conv_ws sc_conv3x3_workspace_floats[_split](hw), bn_ws 2 * (2048 + channels * groups), wgrad_ws
 * sc_conv3x3_wgrad_workspace_floats(channels, channels) floats.  split != 0: the bf16x3-split arithmetic of sc_conv3x3_forward_split /
 * sc_conv3x3_wgrad_split.  Returns the first non-zero status of the calls it makes.                                                       */
typedef struct sc_block_args {
    const float *x, *pf1, *pf2, *pb1, *pb2;
    const float *g1, *b1, *g2, *b2;
    float *rm1, *rv1, *rm2, *rv2;
    int64_t *nt1, *nt2;
    float *y1, *a1, *y2, *out, *st1, *st2;
    float *conv_ws, *bn_ws, *wgrad_ws;
    const float* d_out;
    float *dy2, *dres, *da1, *dy1, *dx, *gw1, *gw2, *dgb1, *dgb2;
    int batch, channels, hw, groups, training, split, need_dx;
    float mom1, eps1, mom2, eps2;
} sc_block_args;
int sc_basic_block_forward(const sc_block_args* args, void* stream);
int sc_basic_block_backward(const sc_block_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 1x1 "Bottleneck_Linear" blocks (csrc/bottleneck.hip): a linear layer WITHOUT bias on row vectors followed by BatchNorm over the rows,
 * optional residual and ReLU, one launch per linear each way.  Replaces, per block, the reference's two nn.Conv2d(C, C, 1) + two
 * nn.BatchNorm2d on 1x1 maps (model/view_estimator.py:6-33 heads, model/graph.py:16-44 latent projectors): rocBLAS products + separate
 * BatchNorm launches before.  x [N][Cin], w [Cout][Cin] (torch layout), y / out / res [N][Cout]; `groups` stacked sub-batches of N / groups
 * rows with their own statistics (save_mean / save_rstd [groups][Cout]); running statistics (may be NULL when training) are updated
 * once per group in order, *n_tracked += groups.  Limits: N <= 128, groups <= 4, Cin % 64 == 0, Cout % 16 == 0
 * (sc_linear_bn_supported says whether a shape is taken).  Fixed summation orders: bit-reproducible.
 *   forward:   y = x w^T;  out = [relu]( gamma (y - mean) rstd + beta [+ res] )
 *   backward:  g = (g_out, or gy_next [N][Cnext] x w_next [Cnext][Cout] when g_out is NULL) [+ g_add]; ReLU mask from `out`;
 *              g_res (may be NULL) receives the masked gradient (= gradient of `res`); BatchNorm backward -> gy [N][Cout];
 *              dw [Cout][Cin] = gy^T x; dgamma, dbeta [Cout]
 *   data:      dx [N][Cin] = gy [N][Cout] w [Cout][Cin] [+ g_add]                                                                  */
int sc_linear_bn_supported(int N, int Cin, int Cout, int groups);
int sc_linear_bn_forward(const float* x, const float* w, const float* gamma, const float* beta, const float* res, float* y, float* out,
                         float* save_mean, float* save_rstd, float* run_mean, float* run_var, int64_t* n_tracked, int N, int Cin,
                         int Cout, int groups, int training, int relu, float eps, float momentum, void* stream);
int sc_linear_bn_backward(const float* g_out, const float* gy_next, const float* w_next, const float* g_add, int Cnext, const float* out,
                          const float* y, const float* save_mean, const float* save_rstd, const float* gamma, const float* x, float* gy,
                          float* g_res, float* dw, float* dgamma, float* dbeta, int N, int Cin, int Cout, int groups, int training, int relu,
                          void* stream);
int sc_linear_backward_data(const float* gy, const float* w, const float* g_add, float* dx, int N, int Cin, int Cout, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-image biases of the conditioned MLP layers (csrc/latent_bias.hip): the latent columns of a layer act on the [B][Z] latent once per
 * image -- the reference repeats the latent per sample point (model/implicit.py:166, model/renderer.py:89).
 *   out [B][NL][64] = bias [NL][64] + (l < L ? post[l] : 0) * z [B][Z] lat[l * 64 + ch][Z]^T      (post may be NULL = 1)
 *   backward: g [B][NL][64] -> g_z [B][Z] (may be NULL), g_lat [L * 64][Z], g_bias [NL][64].  Every element is one fixed-order sum:
 *   results do not depend on the batch size.                                                                                          */
int sc_latent_bias_forward(const float* z, const float* lat, const float* bias, const float* post, float* out, int B, int Z, int L, int NL,
                           void* stream);
int sc_latent_bias_backward(const float* g, const float* z, const float* lat, const float* post, float* g_z, float* g_lat, float* g_bias,
                            int B, int Z, int L, int NL, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Launch policy (csrc/device.hip) -- the one process-level setting of the library.  The persistent one-workgroup-per-CU grids
 * (stream-K 3x3 convolutions and their weight gradients, stem / 1x1 / stride-2 gradients) are sized for
 * sc_grid_cus() = device CUs - reserved.  Reserve CUs when another stream must make progress beside them: RCCL's
 * all-reduce kernels in a multi-GPU step (the reference leaves this to DDP / NCCL: model/runner.py:121).  Call before sizing
 * workspaces (the *_workspace_floats queries follow the grid); default 0, or SHAPECLIPPER_RESERVE_CUS.  Host-only calls.       */
int sc_set_reserved_cus(int n);
int sc_grid_cus(void);
/* The convolution launches keep one small device table per (device, shape, grid) -- the stream-K span cut -- created by the first launch
 * of a shape outside a stream capture (stream-ordered copy on the launch stream).  This frees them all (rebuilt on demand); call
 * with no convolution in flight.                                                                                               */
int sc_conv3x3_release_tables(void);

#ifdef __cplusplus
}
#endif
#endif
