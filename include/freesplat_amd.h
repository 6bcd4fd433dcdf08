/*
 * freesplat_amd.h -- C ABI of libfreesplat_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary of the FreeSplat hot path.  Every entry point takes plain
 * device pointers + sizes + a hipStream_t (passed as void*), returns an int status
 * (0 = ok, <0 = FS_ERR_*), performs NO host synchronisation and NO allocation, and is
 * therefore stream-ordered and hipGraph-capturable.  All tensors are dense, row-major,
 * fp32 unless noted.  Nothing here depends on torch.
 *
 * Reference interfaces replaced (paths relative to the FreeSplat tree):
 *   fs_raster_forward / fs_raster_backward
 *       the un-vendored CUDA extension `diff_gaussian_rasterization_depth`
 *       (requirements.txt:17) as called from src/model/decoder/cuda_splatting.py:100-127
 *       (GaussianRasterizationSettings 12 fields + GaussianRasterizer.forward 6 tensors ->
 *       (color[3,H,W], radii[N], depth[H,W], aux)), and its autograd backward.
 *   fs_cost_volume_forward
 *       AVGFeatureVolumeManager.build_cost_volume,
 *       src/model/encoder/modules/cost_volume.py:429-619 (+ sr_utils/geometry_utils.py:22-89,
 *       src/model/encoder/modules/networks.py:218-236).
 *   fs_ptf_*
 *       the device steps of EncoderFreeSplat.fuse_gaussians,
 *       src/model/encoder/encoder_freesplat.py:431-522.
 */
#ifndef FREESPLAT_AMD_H
#define FREESPLAT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_OK 0
#define FS_ERR_INVALID_ARG (-1)
#define FS_ERR_LAUNCH (-2)
#define FS_ERR_UNSUPPORTED (-3)

#define FS_TILE 16 /* rasterizer tile edge in pixels (16x16 = 4 wavefronts of 64) */

/* Library / build identification: "freesplat_amd <ver> gfx950". */
const char* fs_version(void);
/* Integer ABI revision: bumped whenever a signature, a flag or the layout of an opaque buffer changes.  A binding checks
 * it right after loading the library (freesplat_amd/_lib.py does): a stale build would otherwise accept calls with
 * shifted pointers.  3 = round 3 (single-pass binning: scratch = per-tile key areas, counters[1] = largest tile list on
 * overflow, geom without the mask / depth arrays; fused sort + blend). */
#define FS_ABI_VERSION 6
int fs_abi_version(void);
/* Last HIP error string observed by a failing call on this thread (never NULL). */
const char* fs_last_error(void);

/* Optional per-kernel timing: while bit i of stage_mask is set, every launch of stage i of this library is
 * bracketed by a hipEvent pair recorded on the launch stream (stream-ordered, no sync; -1 = every stage, 0 = off).
 * fs_profile_collect() synchronises the recorded events, sums milliseconds / launch counts per stage into the
 * first n slots (stage i is named fs_profile_stage_name(i), NULL past the last stage) and resets. */
int fs_profile_enable(int stage_mask);
int fs_profile_collect(int n, float* ms_total, int32_t* launches);
const char* fs_profile_stage_name(int i);

/* ------------------------------------------------------------------------------------ *
 * Rasterizer                                                                            *
 * ------------------------------------------------------------------------------------ */

/* Scalars of GaussianRasterizationSettings (cuda_splatting.py:100-113).  The tensor-valued
 * settings (bg[3], viewmatrix[16], projmatrix[16], campos[3]) stay on the device and are
 * passed as pointers so that no device->host copy is ever needed. */
typedef struct fs_raster_dims {
    int32_t N;         /* gaussians */
    int32_t M;         /* SH coefficients per colour channel stored in `shs` ([N, M, 3]); 0 with colors_precomp */
    int32_t H, W;      /* image_height, image_width */
    int32_t sh_degree; /* active SH degree, 0..3, (sh_degree+1)^2 <= M */
    float tanfovx, tanfovy;
    int32_t flags;     /* FS_RASTER_* bits */
} fs_raster_dims;

/* Drop (gaussian, tile) instances that provably cannot reach alpha >= 1/255 anywhere in the tile
 * (conservative bound on the exponent over the tile's pixel rectangle).  Images are bit-identical
 * with and without; without it the tile lists equal the reference's 3-sigma-square lists, with it
 * they are an order-preserving subsequence. */
#define FS_RASTER_TILE_CULL 1
/* `shs` holds IEEE half-precision values ([N,M,3] fp16, 94 instead of 148 input bytes per Gaussian at degree 2):
 * storage only -- they are widened on load and all math stays fp32 (BASELINE config 5).  Gradients stay fp32. */
#define FS_RASTER_SH_FP16 2
/* Reference-native tensor layouts, so that the caller needs no transposed / gathered copies
 * (src/model/types.py:7-12; cuda_splatting.py:78 `rearrange(... "b g xyz n -> b g n xyz")` and :126
 * `gaussian_covariances[:, :, row, col]`):
 * SH_CHANNEL_MAJOR: `shs` / `dL_dshs` rows are [3][M] (harmonics [G,3,d_sh]) instead of [M][3];
 * COV_FULL: `cov3D` / `dL_dcov3D` rows are the row-major 3x3 matrix (9 floats), of which the upper triangle
 * is read; the gradient goes to the upper triangle and the three entries below it are written as 0. */
#define FS_RASTER_SH_CHANNEL_MAJOR 4
#define FS_RASTER_COV_FULL 8
/* Hardware exponential (v_exp_f32, <= 1 ulp) in the blend loops of the forward AND the backward instead of the
 * CPU-reproducible polynomial exp of the bit-exact contract: 11 instead of 26 VALU cycles per (Gaussian, pixel) pair
 * on gfx950.  Tile ranges, list order and radii are unaffected, and so are the alpha >= 1/255 accept / reject decisions:
 * a blend step with an alpha within 16 ulp of the threshold is re-evaluated with the contract exp.  Colours / depth /
 * alpha move by ~1e-6; the remaining discontinuity is the T >= 1e-4 termination, whose flip changes a pixel by less
 * than 1e-4 x colour (counted by tests/test_raster_hip.py and bench.py).  Off by default in the Python layer (not
 * bit-exact).  A forward and its backward must use the same setting. */
#define FS_RASTER_FAST_EXP 16
/* Inference: the caller will not run fs_raster_backward on this forward.  The blend does not track the per-pixel
 * contributor count (n_contrib of the image buffer is left unwritten; colour / depth / alpha / final_T are the same
 * bits) and a tile's sorted list stays in LDS instead of being written to `binning` (only the tile ranges are, and the
 * lists of tiles with more than 2048 entries, which are sorted through global memory -- `binning` must still be a buffer
 * of the size fs_raster_buffer_sizes reports); fs_raster_backward(_views) refuses dims that carry the flag. */
#define FS_RASTER_NO_BACKWARD_STATE 32

/* Byte sizes of the four caller-owned device buffers for (N, H, W, instance capacity):
 *   out[0] geom    : per-Gaussian screen-space state            (saved for backward)
 *   out[1] binning : per-tile ranges + depth-sorted id list      (saved for backward)
 *   out[2] image   : per-pixel final transmittance + n_contrib   (saved for backward)
 *   out[3] scratch : tile counters + one FIXED key area per tile  (reusable after the call)
 * `inst_capacity` bounds the number of (gaussian, tile) instances (the saved lists hold that many entries) and sizes
 * the key areas: every tile owns tile_capacity = the power of two >= max(2048, 4 * inst_capacity / T) key slots (T =
 * tiles), so that the projection kernel can bin in the same launch; see fs_raster_forward for the overflow report. */
int fs_raster_buffer_sizes(int32_t N, int32_t H, int32_t W, int64_t inst_capacity, size_t out[4]);

/*
 * Forward.  Inputs (device): means3D[N,3], cov3D[N,6] (upper-triangular xx,xy,xz,yy,yz,zz),
 * exactly one of shs[N,M,3] / colors_precomp[N,3], opacities[N], bg[3], viewmatrix[16],
 * projmatrix[16] (both as torch passes them: transposed, i.e. column-major), campos[3].
 * Outputs (device): out_color[3,H,W], out_depth[H,W] (sum z*alpha*T, un-normalised),
 * out_alpha[H,W] (1 - T_final), radii[N] (int32, 0 = culled),
 * counters[2] (uint32): {number of instances I, overflow}.  overflow == 0: fine.  overflow != 0 (I > inst_capacity, or
 * a tile list longer than its key area): the value is the LARGEST TILE LIST n_max (>= 1); the image outputs are
 * undefined and the caller must retry with a capacity such that inst_capacity >= I and 4 * inst_capacity / T >= n_max
 * (the library never allocates).
 * Optional device-resident settings (NULL = unused), so that a multi-view caller never has to
 * read a tensor back to the host:
 *   tanfov_dev[2]  overrides dims->tanfovx/tanfovy;
 *   scale_dev[1]   the scale-invariant rescale of cuda_splatting.py:64-71 folded into the kernel:
 *                  means3D * s and cov3D * (s*s) are formed on the fly (bit-identical to doing it
 *                  in torch first); viewmatrix/projmatrix/campos must already be the scaled ones.
 */
int fs_raster_forward(const fs_raster_dims* dims, const float* means3D, const float* cov3D,
                      const float* shs, const float* colors_precomp, const float* opacities,
                      const float* bg, const float* viewmatrix, const float* projmatrix,
                      const float* campos, const float* tanfov_dev, const float* scale_dev,
                      void* geom, void* binning, void* image, void* scratch,
                      int64_t inst_capacity, float* out_color, float* out_depth, float* out_alpha,
                      int32_t* radii, uint32_t* counters, void* stream);

/* Number of scratch (key-area) buffers fs_raster_forward_views needs for v views on n_streams streams: one per view in
 * flight, at most 16 (beyond that the call joins its streams and reuses them). */
int fs_raster_scratch_slots(int32_t v, int32_t n_streams);

/* v views of ONE Gaussian set in one host call (the decoder's path: decoder_splatting_cuda.py:55-75 renders
 * v views per scene; cuda_splatting.py:89-132 is the per-view loop this replaces).  Per-view arrays are packed:
 * bg [v,3], viewmatrix / projmatrix [v,16], campos [v,3], tanfov [v,2] | NULL, scale [v] | NULL (fs_frame_views fills all
 * of them), out_color [v,3,H,W], out_depth / out_alpha [v,H,W], radii [v,N], counters [v,2].  geom / binning / image hold v
 * buffers strides[0..2] bytes apart (>= the sizes of fs_raster_buffer_sizes); scratch holds
 * fs_raster_scratch_slots(v, n_streams) key-area buffers strides[3] bytes apart.
 * ONE launch set for a batch of views (ABI revision 6): the tile counters of the batch are cleared by one launch, every
 * Gaussian is read once and projected + binned into all cameras of the batch by one launch (preprocess_views_kernel), one
 * launch scans the tile counts of all its views; then one fused sort + blend launch per view.  Batches hold
 * FREESPLAT_RASTER_BATCH views (environment, default 16).
 * n_streams > 1: the projection launch sets run on main_stream, the blend of view i on streams[i % n_streams], ordered after
 * its batch's projection by an event -- so the next batch's projection overlaps the blends of this one; main_stream is ordered
 * after all of them before the call returns (fork / join with events, no host sync), so to the caller the call is
 * stream-ordered on main_stream like every other entry point.  n_streams <= 1: everything on main_stream.
 * (FREESPLAT_PREPROCESS=legacy in the environment selects the per-view projection kernel of revisions <= 5 for A/B runs.) */
int fs_raster_forward_views(const fs_raster_dims* dims, int32_t v, const float* means3D, const float* cov3D,
                            const float* shs, const float* colors_precomp, const float* opacities,
                            const float* bg, const float* viewmatrix, const float* projmatrix,
                            const float* campos, const float* tanfov, const float* scale, void* geom,
                            void* binning, void* image, void* scratch, const size_t strides[4], int64_t cap,
                            float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                            uint32_t* counters, int32_t n_streams, void* const* streams, void* main_stream);

/*
 * Backward.  dL_dcolor[3,H,W] (required), dL_ddepth[H,W] (may be NULL).  geom/binning/image
 * are the buffers filled by the matching forward, `counters` the forward's counter pair (may be NULL): when its
 * overflow flag is set the forward produced no image and no valid lists, and the backward of that view yields
 * ZERO gradients instead of walking unbacked tile ranges (a caller that defers the capacity check -- decoder
 * check="deferred" -- may reach backward before it has seen the flag).  `grad_scratch` >= N*12*4 bytes.
 * Outputs (device; overwritten when accumulate == 0, added to when accumulate != 0 -- the
 * multi-view decoder sums the per-view gradients of one shared Gaussian set this way):
 * dL_dmeans3D[N,3], dL_dmeans2D[N,3] (screen-space grad
 * sink, z = 0; cuda_splatting.py:94-98), dL_dcov3D[N,6] (off-diagonals carry both symmetric
 * positions), dL_dshs[N,M,3] or dL_dcolors[N,3] (the other NULL), dL_dopacities[N].
 * `opacities` [N] = the forward's (ABI revision 6: the blend backward sums the geometric moments of dL/dalpha * G and the
 * per-Gaussian pass multiplies them by the opacity).
 */
int fs_raster_backward(const fs_raster_dims* dims, const float* means3D, const float* cov3D,
                       const float* shs, const float* colors_precomp, const float* opacities, const float* bg,
                       const float* viewmatrix, const float* projmatrix, const float* campos,
                       const float* tanfov_dev, const float* scale_dev,
                       const void* geom, const void* binning, const void* image, const uint32_t* counters,
                       const float* dL_dcolor, const float* dL_ddepth, void* grad_scratch,
                       float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcov3D, float* dL_dshs,
                       float* dL_dcolors, float* dL_dopacities, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------ *
 * Plane-sweep cost volume                                                               *
 * ------------------------------------------------------------------------------------ */

/* generate_depth_planes (cost_volume.py:98-134) in one launch: planes[d] = 1 / (1/min + (1/max - 1/min) * ramp[d]), every
 * operation rounded as torch rounds it; min_depth, max_depth: device scalars, ramp[D] = the module's `linear_ramp_1d11`. */
int fs_cost_volume_depth_planes(int32_t D, const float* min_depth, const float* max_depth, const float* ramp,
                                float* planes, void* stream);

/* Bytes of the re-layout workspace fs_cost_volume_forward needs (pixel-major copies of the
 * current and source feature maps). */
size_t fs_cost_volume_workspace_bytes(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w);

/*
 * AVGFeatureVolumeManager.build_cost_volume (cost_volume.py:429-619) fused into one pass.
 * cur_feats[B,C,h,w], src_feats[B,K,C,h,w], src_extrinsics[B,K,4,4] (source<-current),
 * src_Ks[B,K,4,4] (pixel units at the matching resolution), cur_invK[B,4,4];
 * planes: depth of plane d for batch b at pixel p = planes[b*stride_b + d*stride_d + p*stride_pix]
 * (the generated planes of cost_volume.py:98-134 use strides (0,1,0) over a [D] array);
 * MLP (networks.py:218-236): w1[32,C+1], b1[32], w2[32,32], b2[32], w3[1,32], b3[1], LeakyReLU(0.01).
 * out[B,D,h,w].  C must be 48 (FreeSplat) or 16, and one feature map below 4 GB (h*w*(C+32)*4 < 2^32), else
 * FS_ERR_UNSUPPORTED.  fp32 throughout; the 49->32->32 layers run on
 * v_mfma_f32_32x32x2_f32 (exact fp32).
 */
int fs_cost_volume_forward(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                           const float* cur_feats, const float* src_feats,
                           const float* src_extrinsics, const float* src_Ks, const float* cur_invK,
                           const float* planes, int64_t plane_stride_b, int64_t plane_stride_d,
                           int64_t plane_stride_pix, const float* w1, const float* b1,
                           const float* w2, const float* b2, const float* w3, const float* b3,
                           void* workspace, float* out, void* stream);

/* The same forward reading feature maps that are ALREADY pixel-major (channels_last tensors: [h*w][C] records, what the K >= 2
 * sweep gathers from): layout bit 0 = cur_feats is [B, h, w, C], bit 1 = src_feats is [B, K, h, w, C].  A map given this way is read
 * in place -- no re-layout pass (425 of the 755 MB a 10-view K = 8 call moves with [C, h, w] maps).  layout != 0 always takes
 * the 16-pixel sweep (the K = 1 projected sweep reads the caller's [C, h, w] current map).  Inference only: the backward
 * entry points take [C, h, w] maps. */
int fs_cost_volume_forward_layout(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                                  const float* cur_feats, const float* src_feats,
                                  const float* src_extrinsics, const float* src_Ks,
                                  const float* cur_invK, const float* planes, int64_t plane_stride_b,
                                  int64_t plane_stride_d, int64_t plane_stride_pix, const float* w1,
                                  const float* b1, const float* w2, const float* b2, const float* w3,
                                  const float* b3, void* workspace, float* out, int32_t layout, void* stream);

/*
 * Training forward: fs_cost_volume_forward plus `saved` (fs_cost_volume_saved_bytes) -- the MLP's input of every (view,
 * plane, pixel) point (the C averaged warped features, the averaged score, the sources' validity bits: C + 2 floats per
 * point), which fs_cost_volume_backward_train starts from instead of gathering the K sources' taps again.  This is what
 * autograd keeps of cost_volume.py:506-615 (which retains every plane's warped [B*K,C,h,w] features: 8 times more at
 * K = 8).  Same output bits as fs_cost_volume_forward's general sweep (FS_CV_PROJECTED=0).
 */
size_t fs_cost_volume_saved_bytes(int32_t B, int32_t C, int32_t h, int32_t w, int32_t D);
int fs_cost_volume_forward_train(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                                 const float* cur_feats, const float* src_feats,
                                 const float* src_extrinsics, const float* src_Ks, const float* cur_invK,
                                 const float* planes, int64_t plane_stride_b, int64_t plane_stride_d,
                                 int64_t plane_stride_pix, const float* w1, const float* b1,
                                 const float* w2, const float* b2, const float* w3, const float* b3,
                                 void* workspace, float* out, void* saved, void* stream);

size_t fs_cost_volume_backward_workspace_bytes(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w,
                                               int32_t D);
/* The same for ONE call: plane_stride_pix != 0 (per-pixel planes), K > 16 or FS_CV_BWD_ATOMIC=1 select the one-kernel scatter
 * form, whose workspace holds no records (ABI 5).  The first function stays the upper bound for these dimensions. */
size_t fs_cost_volume_backward_workspace_bytes_for(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w,
                                                   int32_t D, int64_t plane_stride_pix);

/*
 * Backward of fs_cost_volume_forward w.r.t. the features and the MLP.  grad_out[B,D,h,w] ->
 * d_cur_feats[B,C,h,w], d_src_feats[B,K,C,h,w] and ALL six parameter gradients d_w1[32,C+1], d_b1[32],
 * d_w2[32,32], d_b2[32], d_w3[32], d_b3[1] (everything overwritten).  The weight gradients -- sums of outer products
 * over all B*D*h*w points -- are accumulated on the matrix cores inside the kernel.  With plane depths that do not vary
 * per pixel (plane_stride_pix == 0: the module's generated planes) and K <= 16 the backward runs in TWO passes with no
 * global float atomics on the source maps: the first writes one record of C + 2 floats per (view, plane, pixel) into the
 * workspace (fs_cost_volume_backward_workspace_bytes therefore grows with D), the second owns tiles of source texels and
 * collects from the pixels that sample them through the inverse plane homography.  Otherwise (per-pixel planes, K > 16,
 * or FS_CV_BWD_ATOMIC=1) one kernel scatters with float atomics.
 */
int fs_cost_volume_backward(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                            const float* cur_feats, const float* src_feats,
                            const float* src_extrinsics, const float* src_Ks, const float* cur_invK,
                            const float* planes, int64_t plane_stride_b, int64_t plane_stride_d,
                            int64_t plane_stride_pix, const float* w1, const float* b1,
                            const float* w2, const float* b2, const float* w3, const float* grad_out,
                            void* workspace, float* d_cur_feats, float* d_src_feats, float* d_w1,
                            float* d_b1, float* d_w2, float* d_b2, float* d_w3, float* d_b3, void* stream);

/* The same from the `saved` buffer of fs_cost_volume_forward_train (same call arguments): no forward recompute. */
int fs_cost_volume_backward_train(int32_t B, int32_t K, int32_t C, int32_t h, int32_t w, int32_t D,
                                  const float* cur_feats, const float* src_feats,
                                  const float* src_extrinsics, const float* src_Ks, const float* cur_invK,
                                  const float* planes, int64_t plane_stride_b, int64_t plane_stride_d,
                                  int64_t plane_stride_pix, const float* w1, const float* b1,
                                  const float* w2, const float* b2, const float* w3, const float* grad_out,
                                  void* workspace, const void* saved, float* d_cur_feats, float* d_src_feats,
                                  float* d_w1, float* d_b1, float* d_w2, float* d_b2, float* d_w3, float* d_b3,
                                  void* stream);

/* ------------------------------------------------------------------------------------ *
 * Pixel-wise Triplet Fusion: matching step                                              *
 * ------------------------------------------------------------------------------------ */

size_t fs_ptf_scratch_bytes(int32_t M, int32_t h, int32_t w);

/*
 * One fold step of EncoderFreeSplat.fuse_gaussians (encoder_freesplat.py:443-482, 508): project the
 * M global Gaussians xyz[M,3] into view i (w2c[16] = inverse(extrinsics_i) row-major, kpix[4] =
 * {fx, fy, cx, cy} in pixels, both on the device), z-buffer per pixel, depth-consistency test against
 * depth_i[h*w] with threshold max(0.05*d, depth_thres), and write ascending index lists:
 *   keep_idx[n_keep]  global entries that stay as they are          (global[~mask])
 *   fuse_idx[n_fuse]  global entries fused with view i, fuse_pix[n_fuse] their pixel in view i
 *   append_pix[n_app] pixels of view i that start new Gaussians     (~fusion_mask)
 * counts[4] = {n_keep, n_fuse, n_app, n_keep + n_fuse + n_app} (device).  Index buffers must hold M (resp. h*w) int64.
 * Bit-exact index semantics (round-half-even pixel, exact z equality, ties fuse together).
 */
int fs_ptf_match(int32_t M, int32_t h, int32_t w, const float* xyz, const float* w2c,
                 const float* kpix, const float* depth_i, float depth_thres, void* scratch,
                 int64_t* keep_idx, int64_t* fuse_idx, int64_t* fuse_pix, int64_t* append_pix,
                 int32_t* counts, void* stream);

/* Inference-path data movement of the same fold step (encoder_freesplat.py:485-519), given the lists of
 * fs_ptf_match.  State arrays: G[M,64] latents, X[M,3], R[M] densities, O[M] weights, E[M,16], D[M]
 * depths; view-i arrays g_i[P,64], x_i[P,3], rho_i[P], om_i[P], d_i[P], E_i[16].
 *   fs_ptf_gru_inputs: cat[n_fuse,176] = [G[m] | PE6(rho_i[p], O[m]) | g_i[p] | PE6(R[m], om_i[p])]
 *                      (the GRU's concatenated input, networks.py:201-206; its linears are the
 *                      caller's GEMMs);
 *   fs_ptf_write_state: next state o* of n_keep+n_fuse+n_app rows in the reference's order
 *                      [kept | fused (fused[n_fuse,64] = GRU output, density-weighted blends) | appended]. */
int fs_ptf_gru_inputs(int32_t n_fuse, const int64_t* fuse_idx, const int64_t* fuse_pix, const float* G,
                      const float* R, const float* O, const float* g_i, const float* rho_i, const float* om_i,
                      float* cat, void* stream);
int fs_ptf_write_state(int32_t n_keep, int32_t n_fuse, int32_t n_app, const int64_t* keep_idx,
                       const int64_t* fuse_idx, const int64_t* fuse_pix, const int64_t* append_pix,
                       const float* G, const float* X, const float* R, const float* O, const float* E,
                       const float* D, const float* g_i, const float* x_i, const float* rho_i,
                       const float* om_i, const float* d_i, const float* E_i, const float* fused, float* oG,
                       float* oX, float* oR, float* oO, float* oE, float* oD, void* stream);

/* One whole fold step of PTF (fs_ptf_match -> fs_ptf_gru_inputs -> fs_ptf_gru_forward -> fs_ptf_write_state) with
 * every data-dependent size kept ON THE DEVICE, so that the steps of all views can be queued back to back without a
 * host sync (the reference syncs four times per view: encoder_freesplat.py:450-453,474,484).
 * The current state G [.,64], X [.,3], R, O, E [.,16], D has *M_dev valid rows (M_dev NULL: M_max rows); M_max is a
 * host-side upper bound of it (i * h*w after i views) that sizes grids and the scratch.  Outputs must hold
 * M_max + h*w rows; counts[4] = {n_keep, n_fuse, n_app, rows of the new state} -- pass &counts[3] as the next step's
 * M_dev.  scratch: fs_ptf_fold_scratch_bytes(M_max, h, w). */
size_t fs_ptf_fold_scratch_bytes(int32_t M_max, int32_t h, int32_t w);
int fs_ptf_fold_step(int32_t M_max, const int32_t* M_dev, int32_t h, int32_t w, const float* G, const float* X,
                     const float* R, const float* O, const float* E, const float* D, const float* g_i,
                     const float* x_i, const float* rho_i, const float* om_i, const float* d_i, const float* E_i,
                     const float* w2c, const float* kpix, float depth_thres, const float* gru_tables,
                     void* scratch, float* oG, float* oX, float* oR, float* oO, float* oE, float* oD,
                     int32_t* counts, void* stream);
/* The same step for a TRAINING fold (ABI 6, round 6): the GRU additionally leaves, for fused pair t (the order of the step's fuse
 * list), row t of `side` [min(M_max, h w), fs_ptf_gru_side_cols()] -- columns 6 .. 9 only: relu(r1), relu(z1), relu(n1), r * hid, the
 * activations the weight gradients pair with -- and `act` [min(M_max, h w) rounded up to 16, fs_ptf_gru_act_cols()] = the gates r, z, q in the kernels' own lane order
 * (opaque: [group of 16 pairs][r, z, q][4 blocks][64 lanes] float4), and row t of `cat` [min(M_max, h w), 176] = the pair's gathered and
 * encoded input row (exactly what fs_ptf_gru_inputs would re-gather for the backward), so that
 * fs_ptf_gru_backward_saved runs the transposed layers only.  Requires fs_ptf_gru_stream_t_rows() > 0 (the 16-pair kernels). */
int fs_ptf_fold_step_save(int32_t M_max, const int32_t* M_dev, int32_t h, int32_t w, const float* G, const float* X,
                          const float* R, const float* O, const float* E, const float* D, const float* g_i,
                          const float* x_i, const float* rho_i, const float* om_i, const float* d_i, const float* E_i,
                          const float* w2c, const float* kpix, float depth_thres, const float* gru_tables,
                          void* scratch, float* oG, float* oX, float* oR, float* oO, float* oE, float* oD,
                          int32_t* counts, float* side, float* act, float* cat, void* stream);

/* All fold steps of one scene in one host call (no host sync, no allocation): views 1 .. V-1 are folded into the state
 * that starts as view 0.  lat [V,P,64], xs [V,P,3], rho / om / dep [V,P], Es [V,16] (camera-to-world), w2c [V,16]
 * (its inverse; NULL = formed here by fs_invert_4x4 -- a pixel's rounding can hinge on the inverse's last bit, so a caller
 * that must reproduce another inverse bit for bit, e.g. torch's, hands its own matrices over), Kn [V,9] (normalised intrinsics; scaled to pixels on the device, encoder_freesplat.py:445-448).  bufA / bufB: two sets of 6 state arrays {G [.,64], X [.,3], R, O, E [.,16], D} with V*P rows
 * each (2*P for V == 2, where bufB is unused), written alternately; counts [V,4].  The final state is set A if
 * (V - 1) is odd, else B, with counts[V-1][3] rows.  scratch: fs_ptf_fold_bytes(V, h, w). */
size_t fs_ptf_fold_bytes(int32_t V, int32_t h, int32_t w);
int fs_ptf_fold(int32_t V, int32_t h, int32_t w, const float* lat, const float* xs, const float* rho,
                const float* om, const float* dep, const float* Es, const float* w2c, const float* Kn,
                float depth_thres, const float* gru_tables, void* scratch, float* const* bufA, float* const* bufB,
                int32_t* counts, void* stream);

/* The GRU of the fold step on the fp32 matrix cores (networks.py:188-214): cat[n,176] rows from
 * fs_ptf_gru_inputs -> fused[n,64].  `tables` = the six weight matrices and biases pre-arranged in MFMA
 * operand order, fs_ptf_gru_table_rows() rows of 64 floats (layout: csrc/ptf_gru.hip; builder:
 * freesplat_amd/ptf.py:gru_tables). */
int32_t fs_ptf_gru_table_rows(void);
/* Layout of `tables` (ABI 6): 0 = the 32-pair kernels' (operand rows of the forward in consumption order, then 192 bias rows); 1 = the
 * 16-pair forward kernel's (the default; FS_GRU_FWD16=0 or FS_GRU_BWD16=0 selects 0): its 696 operand rows of v_mfma_f32_16x16x4_f32
 * padded to whole chunks and interleaved by quads as operand-stream layout 2, then six rows = the bias vectors. */
int32_t fs_ptf_gru_table_layout(void);
int fs_ptf_gru_forward(int32_t n, const float* cat, const float* tables, float* fused, void* stream);

/* Backward of the GRU on the fp32 matrix cores (autograd of networks.py:201-214 w.r.t. its input rows): the forward
 * is re-run from cat[n,176], then every linear layer runs transposed (dX^T = W^T dY^T).  `operand_stream` =
 * fs_ptf_gru_stream_rows() rows of 64 floats: the forward's operand rows (`tables`) followed by the rows of the six
 * transposed matrices (fs_ptf_gru_table_t_rows() of them), re-ordered into the order in which the kernel consumes
 * them -- the workgroup streams them through LDS once for its four wavefronts (layout: csrc/ptf_gru.hip; builders:
 * freesplat_amd/ptf.py:gru_tables_t, gru_operand_stream).  `tables` supplies the biases.
 * g_fused[n,64] = gradient of the GRU output -> dcat[n,176] = gradient of the input rows (fed to
 * fs_ptf_gru_inputs_backward), and side[n, fs_ptf_gru_side_cols()] =
 *   [dr1 | dz1 | dR | dZ | dn1 | dN | relu(r1) | relu(z1) | relu(n1) | r*hid]   (64 floats each)
 * the pre-activation gradients of the six layers and the hidden activations they pair with.
 * fs_ptf_gru_weight_grads contracts them over the n pairs, one launch: grads [fs_ptf_gru_grad_floats() = 44928] +=
 * dW = dY^T X and the bias sums of the 12 parameters, concatenated in the order mlp_r[0].weight [64,176], .bias [64],
 * mlp_r[2].weight [64,64], .bias, mlp_z[0] .., mlp_z[2] .., mlp_n[0].weight [64,152], .bias, mlp_n[2].weight, .bias
 * (networks.py:188-199).  ADDED to `grads` (zero it before the first fold step of a backward pass); the summation order
 * over the pairs is fixed.  workspace: fs_ptf_gru_weight_grads_bytes(n) bytes of device memory (per-workgroup partial
 * sums, reduced by a second launch). */
int32_t fs_ptf_gru_table_t_rows(void);
int32_t fs_ptf_gru_stream_rows(void);
/* Memory layout of `operand_stream` (ABI 6): 0 = row r is 64 consecutive floats; 1 = interleaved by quads of rows -- with
 * c = fs_ptf_gru_stream_chunk_rows() rows per LDS chunk, element [chunk][owner wavefront (4)][quad (c/16)][lane (64)][row of the
 * quad (4)] holds row chunk*c + wavefront*(c/4) + 4*quad + row, lane `lane`, so that a lane's four consecutive operand rows are one
 * float4 (one ds_read_b128 per four MFMAs); 2 = the stream of the 16-pair backward kernel (the default; FS_GRU_BWD16=0 in the
 * environment selects the 32-pair kernel and layout 1): 1 400 operand rows of v_mfma_f32_16x16x4_f32 in consumption order, padded to whole
 * chunks and interleaved as in layout 1, followed by six rows = the bias vectors br1, bz1, br2, bz2, bn1, bn2.
 * freesplat_amd/ptf.py:gru_operand_stream builds what the library reports. */
int32_t fs_ptf_gru_stream_layout(void);
int32_t fs_ptf_gru_stream_chunk_rows(void);
int32_t fs_ptf_gru_side_cols(void);
int fs_ptf_gru_backward(int32_t n, const float* cat, const float* tables, const float* operand_stream,
                        const float* g_fused, float* dcat, float* side, void* stream);
/* Backward of the GRU over the n fused pairs of a step that fs_ptf_fold_step_save ran (replaces the re-run of the forward inside
 * fs_ptf_gru_backward: 704 instead of 1 400 matrix instructions per 16 pairs; reference: autograd through networks.py:188-214).
 * stream_t: the fs_ptf_gru_stream_t_rows() transposed operand rows of stream layout 2 alone (rows 696 .. 1399, interleaved by quads
 * inside chunks exactly like layout 2, no bias rows; freesplat_amd/ptf.py:gru_operand_stream_t); 0 rows = unavailable in this mode.
 * side: the step's buffer (columns 6 .. 9 filled by the forward; columns 0 .. 5 are written here), act: the step's gates. */
int32_t fs_ptf_gru_act_cols(void);
int32_t fs_ptf_gru_stream_t_rows(void);
int fs_ptf_gru_backward_saved(int32_t n, const float* cat, const float* stream_t, const float* act, const float* g_fused,
                              float* dcat, float* side, void* stream);
int32_t fs_ptf_gru_grad_floats(void);
size_t fs_ptf_gru_weight_grads_bytes(int32_t n);
int fs_ptf_gru_weight_grads(int32_t n, const float* cat, const float* side, float* grads, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------ *
 * Gaussian adapter steps either side of PTF                                             *
 * ------------------------------------------------------------------------------------ */

/* GaussianAdapter.forward(fusion=True) / Create_from_depth_map.project (gaussian_adapter.py:36-79,
 * 174-188): depths[V,h*w] -> world xyz[V,h*w,3] = c2w_v @ ((u-cx)/fx z, (v-cy)/fy z, z, 1) with integer
 * pixel coordinates; extrinsics[V,16] c2w row-major, k0_pix[4] = {fx,fy,cx,cy} of view 0 in pixels
 * (the reference uses view 0's intrinsics for every view, :177-181).  Backward: g_xyz -> g_depths. */
int fs_unproject_forward(int32_t V, int32_t h, int32_t w, const float* depths, const float* extrinsics,
                         const float* k0_pix, float* xyz, void* stream);
int fs_unproject_backward(int32_t V, int32_t h, int32_t w, const float* extrinsics, const float* k0_pix,
                          const float* g_xyz, float* g_depths, void* stream);

/* GaussianAdapter.forward(fusion=False, coords given) (gaussian_adapter.py:151-172, 191-201;
 * common/gaussians.py:8-44): raw[M,34] = (scale 3, rotation xyzw 4, SH 27 as (xyz, d_sh)), depths[M],
 * extrinsics[M,16] (blended c2w; only the 3x3 block is used), multiplier[m * mult_stride]
 * (get_scale_multiplier, :203-214; stride 0 = one scalar), sh_mask[9] ->
 * cov[M,9] = Rc (R S S^T R^T) Rc^T, harmonics[M,27] = SH * mask, scales[M,3], rotations[M,4] (unit).
 * Backward: g_cov / g_harmonics / g_scales / g_rotations (each may be NULL = zero) ->
 * g_raw[M,34], g_depths[M], g_extrinsics[M,16]. */
int fs_gaussian_head_forward(int64_t M, const float* raw, const float* depths, const float* extrinsics,
                             const float* multiplier, int64_t mult_stride, const float* sh_mask,
                             float scale_min, float scale_max, float* cov, float* harmonics, float* scales,
                             float* rotations, void* stream);
int fs_gaussian_head_backward(int64_t M, const float* raw, const float* depths, const float* extrinsics,
                              const float* multiplier, int64_t mult_stride, const float* sh_mask,
                              float scale_min, float scale_max, const float* g_cov, const float* g_harmonics,
                              const float* g_scales, const float* g_rotations, float* g_raw, float* g_depths,
                              float* g_extrinsics, void* stream);

/*
 * Per-pixel latents of the fold (encoder_freesplat.py:311-316: `gaussians = head[:, 1:] + skip`, rearranged
 * "(b v) c h w -> b v (h w) c"; densities from head[:, :1]), ABI 5.  head [V, C + 1, P] and skip [V, C, P] are the channel-major
 * maps of the depth decoder's last convolution and of the full-resolution skip convolution (C = 64, P = h * w);
 * latents [V, P, C] = head[:, 1 + c] + skip[:, c] in the pixel-major layout fs_ptf_fold reads, dens [V, P] (optional) = head[:, 0]
 * (the raw density logit).  One pass through an LDS tile each way instead of torch's add + three transposing copies per
 * training step.  Backward: g_latents [V, P, C] (NULL = zero), g_dens [V, P] (NULL = zero) -> g_head [V, C + 1, P] (every
 * channel written), g_skip [V, C, P]; either output may be NULL.
 */
int fs_latents_pack_forward(int32_t V, int64_t P, int32_t C, const float* head, const float* skip, float* latents,
                            float* dens, void* stream);
int fs_latents_pack_backward(int32_t V, int64_t P, int32_t C, const float* g_latents, const float* g_dens, float* g_head,
                             float* g_skip, void* stream);

/* ------------------------------------------------------------------------------------ *
 * Camera framing of the decoder (cuda_splatting.py:17-44, :64-87; projection.py:233-247) *
 * ------------------------------------------------------------------------------------ */

/* v views: extrinsics [v,4,4] (camera-to-world), intrinsics [v,3,3] (normalised), near / far [v] ->
 * view [v,16] and full [v,16] (the transposed world-to-camera / full-projection matrices exactly as
 * GaussianRasterizationSettings.viewmatrix / projmatrix expect them), campos [v,3], tanfov [v,2]
 * (tan(fov_x/2), tan(fov_y/2)) and scale [v] (1/near when scale_invariant, else 1): the device-resident
 * per-view settings fs_raster_forward takes (tanfov_dev, scale_dev).  A singular matrix gives NaNs, as
 * torch.linalg.inv_ex does without check. */
int fs_frame_views(int32_t v, const float* extrinsics, const float* intrinsics, const float* near,
                   const float* far, int32_t scale_invariant, float* view, float* full, float* campos,
                   float* tanfov, float* scale, void* stream);

/* n row-major 4x4 matrices -> their inverses in ONE launch (double precision inside, each entry rounded once to fp32; a
 * singular matrix gives NaNs).  The world-to-camera matrices of the PTF fold (`extrinsic.inverse()`,
 * encoder_freesplat.py:455): torch's batched LU inverse costs ~0.11 ms of host time per fold, a third of a 2-view call. */
int fs_invert_4x4(int32_t n, const float* src, float* dst, void* stream);

/* ------------------------------------------------------------------------------------ *
 * Depth-regression tail of the DepthDecoder (networks.py:130-152)                       *
 * ------------------------------------------------------------------------------------ */

/* logits[B,D,h2,w2], candidates[D] -> stats[B,2,h2*w2] (softmax max / sum, saved for backward),
 * coarse[B,h2*w2] = sum_d cand_d softmax(logits)_d, depth = exp(coarse) (log_planes) | 1/coarse, and, when
 * depth_map / depth_weights / argmax are non-NULL (all three together), at [B,2h2,2w2]:
 * depth_map = exp | 1/ of the x2 align_corners bilinear of coarse, depth_weights = max_d of the
 * x2-upsampled probabilities, argmax = that plane (saved for backward). */
int fs_depth_tail_forward(int32_t B, int32_t D, int32_t h2, int32_t w2, const float* logits,
                          const float* candidates, int32_t log_planes, float* stats, float* coarse,
                          float* depth, float* depth_map, float* depth_weights, int32_t* argmax, void* stream);
/* g_coarse / g_depth [B,h2*w2], g_map / g_weights [B,2h2,2w2] (each may be NULL) -> g_logits[B,D,h2,w2].
 * One launch: every coarse pixel gathers the fine pixels that tap it (the gradient of the upsampled probabilities lives in
 * LDS only).  scratch_gE / scratch_gprob: unused since ABI revision 4 (the scatter form of revisions 1 - 3 needed
 * [B,h2*w2] and [B,D,h2*w2] floats); pass NULL. */
int fs_depth_tail_backward(int32_t B, int32_t D, int32_t h2, int32_t w2, const float* logits,
                           const float* candidates, int32_t log_planes, const float* stats,
                           const float* coarse, const float* depth, const float* depth_map,
                           const int32_t* argmax, const float* g_coarse, const float* g_depth,
                           const float* g_map, const float* g_weights, float* scratch_gE,
                           float* scratch_gprob, float* g_logits, void* stream);

/* Backward of v views of ONE Gaussian set in one host call (counterpart of fs_raster_forward_views; same packed
 * per-view arrays and buffer strides[0..2] = geom / binning / image; counters [v,2] | NULL as in fs_raster_backward).
 * dL_dcolor [v,3,H,W], dL_ddepth [v,H,W] | NULL.
 * grad_scratch: v buffers of align_up(N*48, 256) bytes.  The blend backward of the views alternates over the
 * streams; after the join ONE pass over the Gaussians turns the screen-space gradients of all views into the
 * parameter gradients (inputs read once, sums in registers, outputs written once; `accumulate` adds to their
 * current contents).  Stream-ordered on main_stream like fs_raster_forward_views. */
int fs_raster_backward_views(const fs_raster_dims* dims, int32_t v, const float* means3D, const float* cov3D,
                             const float* shs, const float* colors_precomp, const float* opacities, const float* bg,
                             const float* viewmatrix, const float* projmatrix, const float* campos,
                             const float* tanfov, const float* scale, const void* geom, const void* binning,
                             const void* image, const uint32_t* counters, const size_t strides[3], const float* dL_dcolor,
                             const float* dL_ddepth, void* grad_scratch, float* dL_dmeans3D, float* dL_dmeans2D,
                             float* dL_dcov3D, float* dL_dshs, float* dL_dcolors, float* dL_dopacities,
                             int32_t accumulate, int32_t n_streams, void* const* streams, void* main_stream);

/* The same backward with its per-Gaussian pass restricted to rows [row0, row0 + nrows) of the Gaussian set (ABI revision 6): a
 * caller that sums the gradients over GPUs chunk by chunk (view_sharding.GradExchange("chunked")) passes with_blend = 1 with its
 * first chunk -- the blend backward of all v views runs, then the pass over that chunk's rows -- and with_blend = 0 with the others
 * (only the pass over their rows; the blend's screen-space gradients are still in grad_scratch), so that the reduce-scatter of
 * chunk c overlaps the pass over chunk c + 1.  Rows outside the chunk are not written. */
int fs_raster_backward_views_rows(const fs_raster_dims* dims, int32_t v, const float* means3D, const float* cov3D,
                                  const float* shs, const float* colors_precomp, const float* opacities, const float* bg,
                                  const float* viewmatrix, const float* projmatrix, const float* campos,
                                  const float* tanfov, const float* scale, const void* geom, const void* binning,
                                  const void* image, const uint32_t* counters, const size_t strides[3],
                                  const float* dL_dcolor, const float* dL_ddepth, void* grad_scratch,
                                  float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcov3D, float* dL_dshs,
                                  float* dL_dcolors, float* dL_dopacities, int32_t accumulate, int32_t n_streams,
                                  void* const* streams, void* main_stream, int32_t row0, int32_t nrows, int32_t with_blend);

/* ---- PTF training path: backward of one fold step's data movement (encoder_freesplat.py:485-519) ----
 * fs_ptf_fold_step_lists: device pointers (into the step's scratch) of the four ordered index lists the step left
 * behind, lists[0..3] = keep_idx, fuse_idx, fuse_pix, append_pix (lengths = counts[0..2] of that step).
 * fs_ptf_write_state_backward: gradient of the step's out state -> gradient of its in state and of the view's
 * arrays.  g_out[6] / g_in[6] in the order G, X, R, O, E, D (g_out entries may be NULL = zero; every row of g_in is
 * written, except g_in G of the fused rows, which fs_ptf_gru_inputs_backward writes).  View gradients g_lat_i [P,64],
 * g_x_i [P,3], g_rho_i / g_om_i / g_d_i [P] are ACCUMULATED (zero them first; tied Gaussians may share a pixel).
 * The gradient of the GRU output rows is g_out[0] + n_keep*64 (n_fuse contiguous rows).
 * fs_ptf_gru_inputs_backward: dcat [n_fuse,176] -> g_G rows fuse_idx (stored), g_R / g_O (added), view gradients
 * (accumulated) through the gather and the positional encodings (:62-77, 485-486). */
/* fs_ptf_cameras: kpix [V,4] = (fx w, fy h, cx w, cy h) from the normalised intrinsics Kn [V,9] and E0 [h*w,16] = view 0's
 * camera-to-world matrix on every row (the initial per-Gaussian extrinsics, encoder_freesplat.py:441), one launch: what
 * fs_ptf_fold prepares internally, for a caller that drives fs_ptf_fold_step itself (the training path). */
int fs_ptf_cameras(int32_t V, int32_t h, int32_t w, const float* Es, const float* Kn, float* kpix, float* E0, void* stream);
int fs_ptf_fold_step_lists(int32_t M_max, int32_t h, int32_t w, void* scratch, int64_t** lists);
int fs_ptf_write_state_backward(int32_t n_keep, int32_t n_fuse, int32_t n_app, const int64_t* keep_idx,
                                const int64_t* fuse_idx, const int64_t* fuse_pix, const int64_t* append_pix,
                                const float* X, const float* R, const float* E, const float* D, const float* x_i,
                                const float* rho_i, const float* d_i, const float* E_i, float* const* g_out,
                                float* const* g_in, float* g_lat_i, float* g_x_i, float* g_rho_i, float* g_om_i,
                                float* g_d_i, void* stream);
int fs_ptf_gru_inputs_backward(int32_t n_fuse, const int64_t* fuse_idx, const int64_t* fuse_pix, const float* R,
                               const float* O, const float* rho_i, const float* om_i, const float* dcat, float* g_G,
                               float* g_R, float* g_O, float* g_lat_i, float* g_rho_i, float* g_om_i, void* stream);

/* Debug/test accessors into the opaque buffers (device pointers, no copies). */
const uint32_t* fs_raster_tile_ranges(const void* binning, int32_t H, int32_t W);  /* [T+1] offsets */
const uint32_t* fs_raster_point_list(const void* binning, int32_t H, int32_t W);   /* [I] (id << 4) | 8x8-quadrant mask */
const float* fs_raster_geom_records(const void* geom);                             /* [N,12] */
const float* fs_raster_final_T(const void* image);                                 /* [H*W] */
const int32_t* fs_raster_n_contrib(const void* image, int32_t H, int32_t W);       /* [H*W] */

#ifdef __cplusplus
}
#endif
#endif /* FREESPLAT_AMD_H */
