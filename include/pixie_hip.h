/*
 * include/pixie_hip.h -- C ABI of libpixie_hip.so, the MI355X (gfx950) implementation of
 * Pixie's material_mode=neural inference hot path.
 *
 * The reference (vlongle/pixie) has no FFI/plugin layer for this path: its stages are Python
 * classes calling torch ops and NVIDIA-Warp kernels in-process.  This
 * header is therefore the boundary a maintainer would bind from the reference's Python with
 * ctypes (INTEGRATION.md shows the stubs); each entry point cites the reference call it
 * replaces.  Paths are relative to the reference root:
 *   WG = third_party/Wavelet-Generation,  PG = third_party/PhysGaussian.
 *
 * Conventions
 *   - plain C types only: pointers, sizes, scalars.  No torch / HIP types in signatures;
 *     `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - every pointer named d_* is DEVICE memory owned by the caller (e.g. a torch tensor's
 *     data_ptr()); the library never frees it.  Handles own their internal device state.
 *   - all launches are asynchronous on `stream`; nothing here synchronises the device
 *     unless documented.
 *   - return value 0 = success; non-zero = failure, message via pixie_last_error().
 *   - thread-compatible per handle; pixie_last_error() is thread-local.
 */
#ifndef PIXIE_HIP_H
#define PIXIE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* pixie_last_error(void);
/* Library/version probe: returns the gfx arch string the kernels were compiled for. */
const char* pixie_build_arch(void);
/* (No process-wide switches: every option lives on a handle -- pixie_unet_set_option, pixie_mpm_set_scalar.) */

/* ======================================================================================
 * (B) MLS-MPM solver -- replaces PG/mpm_solver_warp/mpm_solver_warp.py: MPM_Simulator_WARP
 * ====================================================================================== */
typedef struct pixie_mpm pixie_mpm;

/* MPM_Simulator_WARP.__init__/initialize (mpm_solver_warp.py:48-180): allocates particle and
 * grid state for n_particles and an n_grid^3 grid over [0,grid_lim]^3; v=0, C=0, F_trial=I. */
int pixie_mpm_create(pixie_mpm** out, int n_particles, int n_grid, double grid_lim);
int pixie_mpm_destroy(pixie_mpm* h);
/* set_parameters_dict with a new n_grid / grid_lim after the particles were loaded (mpm_solver_warp.py:315-342): the
 * reference re-allocates grid_m / grid_v_in / grid_v_out and recomputes dx, inv_dx; particle state, model scalars,
 * boundary conditions, particle modifiers and the time are untouched.  Same here, in place.  Synchronises `stream`. */
int pixie_mpm_regrid(pixie_mpm* h, int n_grid, double grid_lim, void* stream);

/* Field import/export in the reference's AoS layouts (warp_utils.py:42-74):
 *   "x","v": float[n][3]; "F","F_trial","C","stress": float[n][9] row-major;
 *   "vol","mass","density","E","nu","mu","lam","bulk","yield_stress": float[n];
 *   "material","selection": int32[n]; "init_cov","cov": float[n][6];
 *   grids (export only): "grid_m": float[ng^3], "grid_v_in","grid_v_out": float[ng^3][3].
 * Replaces import_particle_*_from_torch / export_particle_*_to_torch
 * (mpm_solver_warp.py:659-741) and wp.from_torch assignments such as gs_simulation.py:528.
 * `count` is the number of scalars in the caller's buffer (checked). */
int pixie_mpm_set_field(pixie_mpm* h, const char* name, const void* d_src, int64_t count, void* stream);
int pixie_mpm_get_field(pixie_mpm* h, const char* name, void* d_dst, int64_t count, void* stream);
/* set_value_to_{float,int}_array (warp_utils.py:222-230) as used by set_parameters_dict. */
int pixie_mpm_fill_field(pixie_mpm* h, const char* name, double value, void* stream);

/* Scalars of MPMModelStruct set by set_parameters_dict (mpm_solver_warp.py:386-433):
 * "rpic_damping","grid_v_damping_scale","hardening","xi","softening","plastic_viscosity",
 * "friction_angle","gx","gy","gz","time".
 * No reference counterpart: "compensated_x" (0, the default = the reference's float32 `x += dt v`, mpm_utils.py:447; 1 = the
 * rounding error of that sum is carried in three more words per particle and fed into the next increment, so the stored x is the
 * float32 rounding of the accumulated position: displacement error vs float64 4-8x smaller in quiet scenes, +5 % per substep),
 * "scatter_bits" (0 = by the particle-mass contrast, 32, 64), "occupancy", "item_cap" (0 = automatic: decided at every re-binning from
 * that binning's block histogram, with hysteresis; get_scalar "item_cap" returns the capacity in force), "wide", "sparse_tiles",
 * "grid_rb", "resort_interval", "xcd_order" (1, the default: the block kernel's workgroups take the block-ordered work list in
 * contiguous runs per XCD -- same bits, 2.6 % faster; 0: round-robin) (kernel variants, see csrc/mpm.hip).
 * "profile" / "trace": diagnostic switches of the PIXIE_DIAG build (libpixie_hip_diag.so); the product library accepts 0 (a no-op)
 * and refuses any other value. */
int pixie_mpm_set_scalar(pixie_mpm* h, const char* key, double value);
int pixie_mpm_get_scalar(pixie_mpm* h, const char* key, double* value);

/* get_float_array_product(density, vol, mass) (mpm_solver_warp.py:416-425). */
int pixie_mpm_update_mass(pixie_mpm* h, void* stream);
/* finalize_mu_lam / compute_bulk (mpm_solver_warp.py:465-471, 505-511; mpm_utils.py:282-293). */
int pixie_mpm_finalize_mu_lam(pixie_mpm* h, int with_bulk, void* stream);
/* apply_additional_params (mpm_utils.py:591-610): box-select E, nu, density, material. */
int pixie_mpm_apply_additional_params(pixie_mpm* h, const double point[3], const double size[3], double E,
                                      double nu, double density, int material, void* stream);
/* The same for a LIST of boxes in one launch, with the result of applying them one after the other in list order (a particle
 * inside several boxes keeps the LAST one's values).  material_field.py:343-363 uploads the material field as one 1 mm box
 * per particle -- N launches of N threads in the reference; here one launch.  Device arrays: d_boxes[n_boxes][6] =
 * (point.xyz, size.xyz) float32, d_params[n_boxes][3] = (E, nu, density) float32, d_material[n_boxes] int32. */
int pixie_mpm_apply_additional_params_batch(pixie_mpm* h, int64_t n_boxes, const float* d_boxes, const float* d_params,
                                            const int32_t* d_material, void* stream);

/* Grid boundary conditions, applied in registration order after the grid update. */
enum { PIXIE_BC_SURFACE = 0, PIXIE_BC_CUBOID = 1, PIXIE_BC_BBOX = 2 };
typedef struct pixie_bc_desc {
    int32_t type;          /* PIXIE_BC_* */
    int32_t surface_type;  /* surface collider: 0 sticky, 1 slip, 11 cut, 2 other (mpm_solver_warp.py:770-779) */
    int32_t reset;         /* cuboid (mpm_solver_warp.py:895-897) */
    int32_t pad_;
    double point[3], size[3], velocity[3], normal[3];
    double start_time, end_time, friction;
} pixie_bc_desc;
/* add_surface_collider (:749-843), set_velocity_on_cuboid (:853-908), add_bounding_box (:910-977). */
int pixie_mpm_add_bc(pixie_mpm* h, const pixie_bc_desc* bc);

/* Pre-P2G particle modifiers; masks are fixed at registration from the current positions. */
enum { PIXIE_PM_IMPULSE = 0, PIXIE_PM_TRANSLATION = 1, PIXIE_PM_ROTATION = 2 };
typedef struct pixie_pmod_desc {
    int32_t type; /* PIXIE_PM_* */
    int32_t pad_;
    double point[3], size[3];          /* box selection (mpm_utils.py:613-642) */
    double force[3];                   /* impulse: v += force/mass*dt (mpm_solver_warp.py:1015-1027) */
    double velocity[3];                /* translation pin (:1061-1073) */
    double normal[3], h1[3], h2[3];    /* rotation cylinder axes, prepared as :1092-1117 */
    double half_height, radius, rotation_scale, translation_scale;
    double start_time, end_time;
} pixie_pmod_desc;
/* add_impulse_on_particles (:982-1029), enforce_particle_velocity_translation (:1031-1075),
 * enforce_particle_velocity_rotation (:1080-1181). */
int pixie_mpm_add_particle_modifier(pixie_mpm* h, const pixie_pmod_desc* pm, void* stream);

/* p2g2p (mpm_solver_warp.py:514-637), n_substeps times with the same dt; advances h->time. */
int pixie_mpm_step(pixie_mpm* h, double dt, int n_substeps, void* stream);

/* compute_cov_from_F (mpm_utils.py:529-553) and compute_R_from_F (:556-580) as used by
 * export_particle_cov_to_torch / export_particle_R_to_torch (mpm_solver_warp.py:702-741). */
int pixie_mpm_export_cov(pixie_mpm* h, float* d_cov /* [n][6] */, void* stream);
int pixie_mpm_export_R(pixie_mpm* h, float* d_R /* [n][9] */, void* stream);
/* Per-frame export for the rasteriser (PG/gs_simulation.py:591-600 with PG/utils/transformation_utils.py:19-20,108-130):
 * positions and covariances of the first n_out particles (caller order) back in the original scene frame,
 *   pos = ((x - shift) / scale + mean) @ M,   cov = M^T (F_trial init_cov F_trial^T / scale^2) M
 * where shift = (1, 1, 1 + z_shift) (undoshift2center111), scale / mean come from transform2origin and
 * M = R_k ... R_1 is the product apply_inverse_rotations walks (row-major 3x3).  d_cov may be NULL.  One launch. */
int pixie_mpm_export_frame(pixie_mpm* h, int n_out, const double shift[3], double scale, const double mean[3],
                           const double inv_rotation[9], float* d_pos /* [n_out][3] */, float* d_cov /* [n_out][6] or NULL */,
                           void* stream);
/* Particles whose 3x3x3 stencil left the grid (undefined behaviour in the reference).  Here such a particle is frozen
 * (selection <- 2: it neither moves nor scatters mass again) and counted ONCE; slow-path particles that drifted out of
 * every active block between two re-binnings are dropped from that substep's P2G and counted too.  Synchronises
 * `stream`.  pixie_mpm_get_scalar("lost_particles_seen") returns the same count as of the last re-binning without
 * synchronising (the Python shim warns when it becomes non-zero). */
int pixie_mpm_out_of_bounds(pixie_mpm* h, int64_t* count, void* stream);

/* ======================================================================================
 * (A) 3D U-Net operators -- replace the torch ops under WG/models/module/diffusion_network.py
 *     (MyUNetModel :712-935, MyResBlock :639-710, FeatureProjector :534-589, AttentionBlock
 *     :192-242, Upsample/Downsample :51-97).  Activations are NCDHW float32, batch 1.
 * ====================================================================================== */

/* Prologue applied to every input element before the convolution (i.e. the normalisation +
 * activation that precede each conv in the reference graph), fused into the tile load:
 *   t = x * a[c] + b[c]                       (a,b: per-channel, from pixie_norm_finalize)
 *   t = t * gamma[d,h,w] + beta[d,h,w]        (if d_gamma != NULL: spatial LayerNorm affine)
 *   t = act(t):  0 none, 1 LeakyReLU(0.02), 2 SiLU
 * Zero padding applies to the activated tensor, as in the reference. */
typedef struct pixie_conv_desc {
    /* input: channel-concatenation of up to two tensors (th.cat([h, skip]), :932) */
    const float* d_in0; int32_t c0;
    const float* d_in1; int32_t c1;           /* d_in1 may be NULL (c1 = 0) */
    int32_t in_d, in_h, in_w;                 /* spatial size of the stored input tensors */
    int32_t upsample;                         /* 1: nearest x2 before the conv (Upsample :67-72) */
    int32_t stride;                           /* 1, or 2 (Downsample :89-91) */
    int32_t ksize;                            /* 3 (padding 1) or 1 (padding 0) */
    /* prologue */
    const float* d_pro_a; const float* d_pro_b;   /* [c0+c1] or NULL = identity */
    const float* d_gamma; const float* d_beta;    /* [D][H][W] of the conv-input grid, or NULL */
    int32_t act;
    /* weights, repacked by pixie_conv_pack_weights: [k^3][c_in][c_out_padded] */
    const float* d_w; const float* d_bias;        /* bias [c_out] or NULL */
    int32_t c_out;
    /* epilogue: out = conv + bias (+ residual) */
    const float* d_residual;                      /* [c_out][OD][OH][OW] or NULL; may alias d_out */
    float* d_out;                                 /* [c_out][OD][OH][OW] */
    /* f16x3 path (conv3d_f16x3.hip): used instead of d_w when d_w16 != NULL.  Requires stride 1, c0 % 8 == 0 and
     * (c0+c1) % 16 == 0.  The kernel scales the (prologue-transformed) input by a power of two chosen from
     * a bound on its magnitude so that it sits just below the fp16 range before the hi/lo split:
     *   - raw inputs (no prologue): d_in_amax0/1 point at the tensors' |x|max as float bits
     *     (pixie_channel_stats / pixie_tensor_amax keep them on the device -- no host sync);
     *   - normalised inputs: in_bound is a host-side bound on |prologue(x)| (sqrt(N)*max|gamma|+max|beta|). */
    const void* d_w16;                            /* from pixie_conv_pack_weights_f16x2, or NULL */
    const uint32_t* d_in_amax0; const uint32_t* d_in_amax1;
    float in_bound;
    /* f16x3 path only, optional: statistics of the OUTPUT taken in the epilogue (what the next layer's LayerNorm /
     * GroupNorm and the next f16x3 conv's scaling need), instead of a separate pass over the tensor:
     *   d_out_stats: pixie_conv_stats_floats(desc) floats of per-tile partial sums, finalised into the
     *   double[2*c_out] layout of pixie_channel_sums by pixie_stats_finalize; d_out_amax: |out|max (float bits),
     *   atomicMax'ed (caller zeroes). */
    float* d_out_stats;
    uint32_t* d_out_amax;
    /* f16x3 path only, optional: pixie_conv_workspace_bytes(desc) bytes of device scratch.  With it, layers whose output
     * is too small to fill the chip (the 16^3 / 32^3 levels) split their channel chunks over up to 8 workgroup slices
     * (deterministic: partial outputs are added in a fixed order); such layers do not produce d_out_stats. */
    void* d_workspace;
    /* Output extent, 0 = the natural size ((in * (upsample ? 2 : 1) + 2 pad - ksize) / stride + 1).  A smaller value crops
     * the trailing planes / rows / columns: the odd-grid crop h[..., :-1] that MyUNetModel.forward applies to an
     * up-sampled tensor before concatenating it with an odd-sized skip tensor (diffusion_network.py:925-930), done by not
     * computing the cropped voxels (d_out, d_residual and the output statistics all have the cropped extent). */
    int32_t out_d, out_h, out_w;
    /* f16x3 path only, optional: the residual block's 1x1x1 skip convolution (MyResBlock.skip_connection,
     * diffusion_network.py:691,705) folded into this launch:  out = conv(prologue(in)) + bias + skip_conv(skip_in) + skip_bias.
     * skip_in is the channel concatenation of up to two RAW tensors with the spatial size of the output; d_skip_w16 comes from
     * pixie_conv_pack_weights_f16x2(w_skip, ., c_out, skip_c0 + skip_c1, 1), d_skip_amax0/1 are the tensors' |x|max slots.
     * The skip tensor is then never written or read.  Only where pixie_conv_skip_foldable() says so (stride 1, no upsample,
     * no split-K); d_residual may still be given as well. */
    const float* d_skip_in0; int32_t skip_c0;
    const float* d_skip_in1; int32_t skip_c1;
    const void* d_skip_w16; const float* d_skip_bias;
    const uint32_t* d_skip_amax0; const uint32_t* d_skip_amax1;
} pixie_conv_desc;

/* Repack an nn.Conv3d / nn.Conv1d weight (c_out, c_in, k,k,k) into the kernel's
 * [tap][c_in][c_out_padded] layout; c_out_padded = pixie_conv_cout_padded(c_out). */
int pixie_conv_cout_padded(int c_out);
int pixie_conv_pack_weights(const float* d_w_oidhw, float* d_w_packed, int c_out, int c_in, int ksize, void* stream);
/* The same weight, split into fp16 hi/lo halves (w*s = hi + lo, s a power of two chosen on the device from
 * |w|max) and swizzled for the f16 MFMA A operand; d_packed needs pixie_conv_packed16_bytes() bytes. */
int64_t pixie_conv_packed16_bytes(int c_out, int c_in, int ksize);
int pixie_conv_pack_weights_f16x2(const float* d_w_oidhw, void* d_packed, int c_out, int c_in, int ksize, void* stream);
/* F.conv3d / nn.Conv3d forward: exact-fp32 MFMA path (d_w), or the f16x3 split path (d_w16). */
int pixie_conv3d_forward(const pixie_conv_desc* desc, void* stream);
/* Epilogue statistics of the f16x3 path: buffer size in floats for desc->d_out_stats (0: layer not on that path), and
 * the reduction of that buffer to d_sums[2*c] = (sum, sum of squares) in float64. */
int64_t pixie_conv_stats_floats(const pixie_conv_desc* desc);
int64_t pixie_conv_workspace_bytes(const pixie_conv_desc* desc);
int pixie_stats_finalize(const float* d_stats, const pixie_conv_desc* desc, double* d_sums, void* stream);
/* 1 if this descriptor's launch (its shape fields, d_w16 and d_workspace as they will be passed) can take a folded skip
 * convolution with skip_c0 + skip_c1 input channels; the skip pointers themselves need not be set yet. */
int pixie_conv_skip_foldable(const pixie_conv_desc* desc);

/* Per-channel sum and sum of squares over the spatial extent: d_sums[2*c] (float64). */
int pixie_channel_sums(const float* d_x, int channels, int64_t spatial, double* d_sums, void* stream);
/* Same, and additionally atomicMax's the tensor's |x|max (as float bits) into *d_amax (caller zeroes it). */
int pixie_channel_stats(const float* d_x, int channels, int64_t spatial, double* d_sums, uint32_t* d_amax, void* stream);
/* |x|max of `count` floats, atomicMax'ed (as float bits) into *d_slot (caller zeroes it). */
int pixie_tensor_amax(const float* d_x, int64_t count, uint32_t* d_slot, void* stream);
/* Turn channel sums into the prologue's (a,b):
 *  mode 0: LayerNorm([D,H,W]) statistics per channel (biased var, eps): a = rstd, b = -mean*rstd
 *          (affine gamma/beta are spatial and applied in the conv prologue);
 *  mode 1: GroupNorm(groups, channels): a = rstd_g*weight[c], b = bias[c] - mean_g*rstd_g*weight[c]. */
int pixie_norm_finalize(const double* d_sums, int channels, int64_t spatial, int mode, int groups, double eps,
                        const float* d_weight, const float* d_bias, float* d_a, float* d_b, void* stream);

/* QKVAttention (diffusion_network.py:218-242), single head: qkv [3C][T] -> out [C][T],
 * softmax over keys of (q*s)^T(k*s), s = C^-1/4, streamed (no T x T buffer). */
int pixie_attention_forward(const float* d_qkv, float* d_out, int channels, int tokens, void* stream);

/* y = x*a[c] + b[c] materialised (GroupNorm output feeding a non-conv consumer). */
int pixie_channel_affine(const float* d_x, const float* d_a, const float* d_b, float* d_y, int channels,
                         int64_t spatial, void* stream);

/* The voxel-grid loader of the reference's dataset item (WG/data_utils/my_data.py:160-224; features are written as
 * (D,H,W,C) float16 by pixie/voxel/voxelize.py:86,111): .astype(float32) + permute to (C,D,H,W), on the device. */
int pixie_voxel_grid_to_ncdhw(const void* d_feat_dhwc_f16, int d, int h, int w, int channels, float* d_out_cdhw, void* stream);

/* FeatureProjector.net[0] -- Conv3d(C, c_out, 1) + bias (diffusion_network.py:556-560) -- of one or two networks applied
 * to the voxel grid as the reference stores it: (voxels, C) float16, channels last (pixie/voxel/voxelize.py:86,111; what
 * WG/data_utils/my_data.py:160-224 converts to float32 and permutes first).  One launch reads the grid once and writes
 * d_out[n] = (c_out, voxels) float32 for each network n.  d_w16[n] comes from pixie_conv_pack_weights_f16x2(w, ., c_out,
 * C, 1): the inputs are exact fp16 numbers, so only the weights are split (two f16 MFMAs per product, fp32 accumulate).
 * Requires C % 16 == 0 and n_networks * padded(c_out) in {64, 128, 256} (the reference shapes: c_out = 128, one or two
 * networks); other shapes go through pixie_voxel_grid_to_ncdhw + pixie_conv3d_forward. */
int pixie_projector_conv0(const void* d_feat_dhwc_f16, int64_t voxels, int channels, int n_networks, const void* const* d_w16,
                          const float* const* d_bias, float* const* d_out, int c_out, void* stream);

/* The field exchange between ranks (the reference writes a 44 B/voxel (11, D, D, D) float file per scene and has no
 * exchange): n_scenes x (3, V) float32 continuous channels + n_scenes x V int32 class ids -> the 13 B/voxel wire buffer
 * [3 n V float32 | n V uint8 | zero pad to a multiple of 16 B] that ONE all-gather moves (pixie_amd/distributed.py).  One launch. */
int pixie_pack_fields(const float* d_cont, const int32_t* d_seg_pred, int64_t n_scenes, int64_t spatial, void* d_wire, int64_t wire_bytes,
                      void* stream);

/* process_batch/save_predictions (WG/trainer/inference_combined.py:124-126,186-195):
 * combined[0:3] = cont_pred; combined[3+k] = (argmax_c logits == k), ties -> lowest index. */
int pixie_combine_predictions(const float* d_logits, int num_classes, const float* d_cont, int64_t spatial,
                              float* d_combined, int32_t* d_argmax /* may be NULL */, void* stream);

/* save_predictions given class ids (WG/trainer/inference_combined.py:173-199): combined[0:3] = cont_pred,
 * combined[3 + k] = (seg_pred == k) for k < num_classes (an id outside [0, num_classes) leaves every class channel 0). */
int pixie_combine_class_ids(const int32_t* d_seg_pred, int num_classes, const float* d_cont, int64_t spatial, float* d_combined,
                            void* stream);

/* --------------------------------------------------------------------------------------
 * (A') One network as a handle -- what the reference does with an nn.Module: construction (MyUNetModel.__init__,
 *      diffusion_network.py:712-873, FeatureProjector :534-589, behind SegmentationUNet / RegressionUNet,
 *      trainer/training_discrete.py:50-88, trainer/training_continuous_mse.py:48-89), load_state_dict and
 *      forward (:875-935) -- so that a scene costs ONE foreign call per network instead of ~400 operator calls.
 *      The launch sequence equals pixie_amd/unet.py: UNetRunner.forward (bit-identical results); it allocates
 *      nothing and never synchronises once the weights are packed, i.e. it can be captured into a HIP graph.
 * -------------------------------------------------------------------------------------- */
typedef struct pixie_unet pixie_unet;
typedef struct pixie_unet_config {          /* the constructor arguments of the two wrappers */
    int32_t feature_channels, cond_dim, model_channels, num_res_blocks;
    int32_t n_channel_mult; int32_t channel_mult[8];
    int32_t n_attention_resolutions; int32_t attention_resolutions[8];
    int32_t grid_size;                      /* D = H = W of the feature grid (the LayerNorm([D,H,W]) parameters have this shape) */
    int32_t out_channels;                   /* num_classes (segmentation) or 3 (regression) */
    int32_t precision;                      /* 0: f16x3 convolutions where the shapes allow (default path), 1: exact fp32 MFMA everywhere */
} pixie_unet_config;

int pixie_unet_create(pixie_unet** out, const pixie_unet_config* cfg);
int pixie_unet_destroy(pixie_unet* h);
/* The state_dict: keys in the reference's registration order (a reference checkpoint loads key by key). */
int pixie_unet_param_count(const pixie_unet* h);
int pixie_unet_param_info(const pixie_unet* h, int index, const char** key, int64_t* numel, int32_t* ndim, int64_t shape[5]);
/* Point parameter `key` at `numel` float32 values in device memory.  The memory stays the caller's and must outlive the
 * handle's use of it; call again after changing the values (conv weights are re-packed, normalisation bounds re-taken, on
 * the next forward).  An unknown key or a wrong size is an error, as with load_state_dict(strict=True). */
int pixie_unet_set_param(pixie_unet* h, const char* key, const float* d_values, int64_t numel);
/* Bytes of device scratch one forward pass over a (d, h, w) grid needs (all activations, statistics and split-K buffers). */
int64_t pixie_unet_workspace_bytes(pixie_unet* h, int d, int hh, int w);
/* forward: d_feat (feature_channels, d, h, w) float32 -> d_out (out_channels, d, h, w) float32, all launches on `stream`.
 * d_proj0 (optional, hidden-128 projector only): the output of projector.net[0] computed elsewhere (pixie_projector_conv0 on
 * the channels-last grid); d_feat may then be NULL.  The first call after pixie_unet_set_param packs weights (hipMalloc) and
 * synchronises the stream once to read the normalisation bounds; later calls are launch-only. */
int pixie_unet_forward(pixie_unet* h, const float* d_feat, const float* d_proj0, int d, int hh, int w, float* d_out,
                       void* d_workspace, int64_t workspace_bytes, void* stream);
/* Options.  "graph" = 1: pixie_unet_forward records its launch sequence the first time it sees a combination of
 * (d_feat, d_proj0, d_out, d_workspace) pointers and replays it as ONE hipGraphLaunch on later calls with the same pointers
 * (a caller that keeps its buffers; up to 4 combinations are kept; pixie_unet_set_param invalidates them).  Needs a
 * non-default stream.  Same kernels, same arguments: bit-identical output. */
int pixie_unet_set_option(pixie_unet* h, const char* key, int value);

/* ======================================================================================
 * (C) Field -> particle transfer between the two halves: replaces the PLY round trip
 *     pixie/voxel/map_pred_to_coords.py:41-75,192-252 (unscale_prediction + masked voxel point list) and
 *     PG/material_field.py:228-300 perform_knn_smoothing (sklearn K-NN + a Python loop over every particle;
 *     MaterialProperties.assign_from_neighbors :52-78, .get_defaults :38-50).
 * ====================================================================================== */
typedef struct pixie_field_desc {
    const float* d_pred;        /* network output (3 + n_classes, D, H, W): cont channels in [-1,1] units, class scores / one-hot */
    const uint8_t* d_mask;      /* (D, H, W) occupancy, > 0 = a material point (clip_features_mask) */
    const float* d_axis_x; const float* d_axis_y; const float* d_axis_z;  /* voxel coordinates: float32(np.linspace(min, max, D)) per axis */
    int32_t n_classes, d, h, w;
    double min_spacing;         /* smallest lattice spacing of the three axes */
    double density_min, density_max, E_min, E_max, nu_min, nu_max;  /* normalization_stats/normalization_ranges.yaml */
} pixie_field_desc;
/* For each of the n particles (d_pos: float[n][3] in the field's coordinate frame) the k (<= 16) nearest material points;
 * continuous properties = mean (or inverse-distance weighted mean) of the un-scaled neighbours, material id / part label =
 * mode (ties: first in distance order; weighted: largest vote, ties to the smallest id), d_nearest = distance to the nearest
 * point.  Particles whose nearest point is farther than nn_distance_threshold get the defaults (mean over all material
 * points, default_material, default_part_label).  d_scratch: 64 bytes of device memory; after the call it holds
 * double[5] = sums of (density, E, nu, conf) over the material points and their count, then uint64 = number of too-far
 * particles.  Asynchronous on `stream`. */
int pixie_field_to_particles(const pixie_field_desc* field, const float* d_pos, int n, int k, double nn_distance_threshold,
                             int weighted, int default_material, int default_part_label, float* d_density, float* d_E,
                             float* d_nu, int32_t* d_material, int32_t* d_part_label, float* d_conf, float* d_nearest,
                             void* d_scratch, void* stream);

/* unscale_prediction (pixie/voxel/map_pred_to_coords.py:41-75): d_out[(channels, spatial)] = d_pred with channels 0..2
 * clipped to [-1, 1] and mapped back to physical units -- density and E through 10^(log-range), nu linearly, with the
 * ranges of normalization_stats/normalization_ranges.yaml; the class channels (3..) are copied.  float32 arithmetic,
 * as numpy does on the float32 array.  d_out may alias d_pred. */
int pixie_unscale_prediction(const float* d_pred, int channels, int64_t spatial, double density_min, double density_max,
                             double E_min, double E_max, double nu_min, double nu_max, float* d_out, void* stream);
/* The masked voxel point list map_pred_to_ply writes to its PLY (map_pred_to_coords.py:192-252): for every voxel with
 * mask > 0, in C order of the grid, its coordinates (the float32 lattice axes of `field`), un-scaled density / E / nu,
 * material id (argmax of the class channels, first maximum; a single class channel is the id itself, :122-126) and
 * confidence (the winning class score; 1 for a single channel).  *d_count (device int64) receives the number of points;
 * at most `capacity` records are written (call with capacity 0 to only count).  d_scratch needs
 * pixie_field_points_scratch_bytes(field) bytes.  Stable compaction, three launches, asynchronous on `stream`. */
int64_t pixie_field_points_scratch_bytes(const pixie_field_desc* field);
int pixie_field_points(const pixie_field_desc* field, int64_t capacity, float* d_xyz /* [n][3] */, float* d_density, float* d_E,
                       float* d_nu, int32_t* d_material, float* d_conf, int64_t* d_count, void* d_scratch, void* stream);

/* Density clustering of the "stationary" particles -- sklearn.cluster.DBSCAN(eps, min_samples).fit_predict at
 * PhysGaussian/material_field.py:405-406 -- up to the numbering: d_root[i] (ORIGINAL order) = the lowest original index among
 * the core points of point i's cluster (core points: their component; border points: the smallest such root among the core
 * points within eps), or -1 for noise.  Clusters numbered by ascending root are DBSCAN's labels.  The caller bins the points:
 * d_pos_sorted [n][3] sorted by cell id ((cx * ny + cy) * nz + cz of floor((p - lo) / cell_size), clamped), d_orig_index[s] the
 * original index of sorted point s, d_cell_start[nx*ny*nz + 1] the first sorted index of every cell; cell_size >= eps.
 * d_core_scratch / d_parent_scratch: n int32 each.  Distances in float64 on the float32 coordinates.  Three launches. */
int pixie_dbscan_roots(const float* d_pos_sorted, const int32_t* d_orig_index, const int32_t* d_cell_start, int n, int nx, int ny, int nz,
                       const double lo[3], double cell_size, double eps, int min_samples, int32_t* d_core_scratch,
                       int32_t* d_parent_scratch, int32_t* d_root, void* stream);

/* ======================================================================================
 * (D) Particle pre-pass of the MPM program: replaces the Taichi kernels of
 *     PG/particle_filling/filling.py that gs_simulation.py:442-482 runs before the solver is created.
 * ====================================================================================== */
/* densify_grids + compute_density (filling.py:13-92): for each of n Gaussians (d_pos [n][3], d_opacity [n], d_cov6 [n][6] =
 * xx,xy,xz,yy,yz,zz) count it in its cell (d_grid_count[grid_n^3] += 1) and splat opacity * mean_corners exp(-d^T C^-1 d / 2)
 * onto every cell within ceil(sqrt(max eigenvalue) / grid_dx) cells (d_grid_density[grid_n^3], accumulated; caller zeroes
 * both grids).  Particles outside the grid are not counted (the reference indexes unchecked). */
int pixie_fill_densify(const float* d_pos, const float* d_opacity, const float* d_cov6, int n, int grid_n, double grid_dx,
                       int32_t* d_grid_count, float* d_grid_density, void* stream);
/* fill_dense_grids (filling.py:95-121): every cell with density > density_thres and fewer than max_particles_per_cell
 * particles is topped up: new points at (cell + u) * grid_dx, u uniform in [0,1)^3 from a counter-based hash of (seed, cell,
 * k) -- ti.random() in the reference -- appended at d_new_particles[*d_counter ...]; *d_counter (device uint64) advances even
 * past max_samples (nothing is written there), so the caller can detect the overflow the reference would write through. */
int pixie_fill_dense_cells(int32_t* d_grid_count, const float* d_grid_density, int grid_n, double grid_dx, double density_thres,
                           int max_particles_per_cell, float* d_new_particles /* [max_samples][3] */, int64_t max_samples,
                           uint64_t* d_counter, uint32_t seed, void* stream);
/* internal_filling with collision_search / collision_times (filling.py:124-244): an EMPTY cell whose rays in all six axis
 * directions except exclude_dir (0:+x 1:-x 2:+y 3:-y 4:+z 5:-z) meet a cell with density > threshold, and whose ray along
 * ray_cast_dir crosses the surface an odd number of times, is filled with max_particles_per_cell points. */
int pixie_fill_internal_cells(int32_t* d_grid_count, const float* d_grid_density, int grid_n, double grid_dx, int max_particles_per_cell,
                              int exclude_dir, int ray_cast_dir, double threshold, float* d_new_particles, int64_t max_samples,
                              uint64_t* d_counter, uint32_t seed, void* stream);
/* get_particle_volume (filling.py:247-288): d_vol[p] = grid_dx^3 / (particles in p's cell).  d_grid_count_scratch:
 * grid_n^3 int32 of scratch (zeroed here). */
int pixie_particle_volume(const float* d_pos, int n, int grid_n, double grid_dx, int32_t* d_grid_count_scratch, float* d_vol, void* stream);
/* get_attr_from_closest (filling.py:383-403): d_nearest[i] = index of the original particle closest to new particle i
 * (first minimum in index order; brute force through LDS tiles). */
int pixie_nearest_particle(const float* d_pos, int n, const float* d_new_pos, int n_new, int32_t* d_nearest, void* stream);

/* ======================================================================================
 * Diagnostic entry points -- NOT part of the drop-in ABI.  They exist only in the -DPIXIE_DIAG build of the same sources,
 * libpixie_hip_diag.so, which the parity tests (per-phase comparison with the oracle) and the profilers (per-launch timings)
 * load; the production library libpixie_hip.so exports none of them and carries no trace buffer.
 * ====================================================================================== */
#ifdef PIXIE_DIAG
/* One phase of a substep, for per-kernel parity tests: 0 = pre-P2G modifiers + stress + P2G,
 * 1 = grid update + damping + BCs (+ host `modify`), 2 = G2P.  Does not advance time. */
int pixie_mpm_phase(pixie_mpm* h, int phase, double dt, void* stream);

/* Average duration in ms of the fused particle kernel / grid kernel over the launches since the
 * last call, measured with HIP events on `stream` (enable with set_scalar "profile"=1). */
int pixie_mpm_kernel_times(pixie_mpm* h, double* particle_ms, double* grid_ms, int64_t* n_launches);

/* Which kernel instantiation pixie_conv3d_forward picks for this descriptor: ksize*100 + MB*10 + NB of
 * conv3d_f16x3_kernel<ksize,MB,NB> (and its split-K factor in *slices), 9324 = conv3d_f16x3_c64_fullres_kernel (the <3,2,4>
 * code under its own symbol for the 64 -> 64 full-resolution 3^3 layers), 0 = the exact-fp32 kernel.  For profilers that
 * want to group per-launch timings by kernel name, as rocprofv3 does; no reference counterpart. */
int pixie_conv_kernel_variant(const pixie_conv_desc* desc, int* slices);
#endif /* PIXIE_DIAG */

#ifdef __cplusplus
}
#endif
#endif /* PIXIE_HIP_H */
