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
 * nearest-neighbour distance d = fma(dz,dz,fma(dy,dy,dx*dx)); idx1 [b,n], idx2 [b,m] int32: index
 * of the nearest neighbour, lowest index among exact ties (chamfer3D.cu:36,126).  Outputs are
 * fully overwritten (if n==0 or m==0 they are left untouched, as the reference does).
 * Launches on `stream` (the reference used the legacy default stream).                          */
int sc_chamfer3d_forward(const float* xyz1, const float* xyz2, float* dist1, float* dist2,
                         int32_t* idx1, int32_t* idx2, int b, int n, int m, void* stream);

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
 * (pre-activations a_0..a_4 and adjoints p_0..p_3 kept for sc_sdf_backward).                    */
#define SC_SDF_PACK_FLOATS (64*48 + 2*64*112 + 2*64*64 + 65*64 + 65)
#define SC_RGB_PACK_FLOATS (64*112 + 2*64*64 + 3*64 + 4)
#define SC_TILE_POINTS 16
int sc_sdf_forward(const float* points, const float* w_pack, const float* cbias, int n_points,
                   int n_per_image, int n_images, int symmetric, float* sdf, float* grad,
                   float* feat, float* stash_a, float* stash_p, void* stream);

#ifdef __cplusplus
}
#endif
#endif
