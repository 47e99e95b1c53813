/*
 * njf_hip.h -- C ABI of the MI355X (gfx950) volumetric-rendering hot path of Neural Jacobian Fields.
 *
 * The reference (sizhe-li/neural-jacobian-field) has no FFI: its seam is the Python object API of
 * project/neural_jacobian_field/models/model.py (Model.forward :316, encode_image :458,
 * compute_density :416) and the operator registry in models/decoder/__init__.py:11-44.  This header
 * is the boundary placed *underneath* that API: every entry point names the
 * reference code it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer to fp32 unless noted; tensors are dense row-major.
 *   - Caller owns all memory; the library never allocates, frees or keeps state between calls.
 *   - All work is enqueued on `stream` (a hipStream_t passed as void*); no host synchronisation.
 *   - Return value: 0 = ok; negative = invalid argument (nothing was launched, see
 *     njf_error_string); positive = a hipError_t from the launch.
 *   - Re-entrant and thread-safe.
 *
 * Packed-weight layout ("fragment-major")
 *   The fused kernels keep activations in MFMA C/D registers between layers
 *   (v_mfma_f32_32x32x2_f32; a wave owns 32 points, lane l owns point l&31 and half l>>5 of the
 *   features).  A layer y = W x + b with W [d_out, d_in] (torch.nn.Linear layout) is stored as
 *   P[kb][q][mb][lane][e] = W[fo(mb, lane&31)][16*KB*(lane>>5) + 16*kb + 4*q + e]
 *   with MB = ceil(d_out/32), KB = ceil(d_in/32), zero padding, and
 *   fo(mb, i) = 16*MB*((i>>2)&1) + 16*mb + (i&3) + 4*(i>>3).
 *   njf_pack_* produce these blobs from reference-layout tensors; callers never build them by hand.
 */
#ifndef NJF_HIP_H
#define NJF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NJF_ABI_VERSION 20
#define NJF_MAX_ACTION_DIM 10   /* 3*A <= 32 outputs of the Jacobian head */
#define NJF_HIDDEN 128          /* MlpCfg.d_hidden (model_components/resnet_fc.py:12-18) */
#define NJF_LATENT 512          /* encoder feature channels (models/encoder/encoder_resnet.py:88) */
#define NJF_PE_DIM 63           /* NeRFEncoding(3, 10 freqs, include_input) output width */
#define NJF_ZDIM 384            /* 3 lin_z layers x 128: channels of one net's hoisted feature map */

/* floats in one packed ResnetFC blob (weights) and its bias blob */
#define NJF_RESNET_CHUNKS 22
#define NJF_RESNET_CHUNKS_F16 12  /* NJF_PRECISION_F16 fills the first 12 chunk slots of the same NJF_RESNET_W_FLOATS blob */
#define NJF_CHUNK_FLOATS 8192
#define NJF_RESNET_W_FLOATS (NJF_RESNET_CHUNKS * NJF_CHUNK_FLOATS)
#define NJF_RESNET_B_FLOATS (10 * 128 + 32)
#define NJF_COLOR_W_FLOATS NJF_CHUNK_FLOATS
#define NJF_COLOR_B_FLOATS (64 + 32)
/* folded transformer Jacobian head: 14 half-chunks [query | (Mqk, Nov, W1, W2) x 3 | head]; 64 hoisted query channels */
#define NJF_TRANSFORMER_CHUNKS 7
#define NJF_TRANSFORMER_W_FLOATS (NJF_TRANSFORMER_CHUNKS * NJF_CHUNK_FLOATS)
#define NJF_TRANSFORMER_B_FLOATS (3 * 256 + 32)
#define NJF_QDIM 64

/* MFMA precision of the fused MLPs.  A packed blob and the forward call that consumes it must agree. */
#define NJF_PRECISION_F32 0    /* v_mfma_f32_32x32x2_f32: exact fp32 products */
#define NJF_PRECISION_F16X2 1  /* fp32 operands split hi+lo into two fp16; hi*hi + hi*lo + lo*hi accumulated in fp32
                                  by v_mfma_f32_32x32x16_f16: fp32-class accuracy (dropped term 2^-22) at 3/16 the cost */
#define NJF_PRECISION_F16F6 2  /* hi*hi as above; the two correction products hi*lo + lo*hi (2^-11 of the result) in
                                  block-scaled fp6 (e2m3, one power-of-two scale per lane and 32 K-values) by
                                  v_mfma_scale_f32_32x32x64_f8f6f4 at 4x the f16 rate: half the matrix time of F16X2,
                                  ~1.5e-5 relative error per ResnetFC (tools/sim_split_precision.py).  Applies to the
                                  128-wide layers; the narrow layers (lin_out, colour head, transformer head) and
                                  the feature projection keep the F16X2 form */

#define NJF_PRECISION_F16 3    /* PLAIN fp16 products (BASELINE config 5: "fp16 MFMA fused-MLP"): weights packed once as single
                                  fp16, layer inputs rounded to fp16 behind the ReLU, v_mfma_f32_32x32x16_f16 with fp32
                                  accumulation and fp32 biases / residuals -- no hi/lo split, no correction products (issue
                                  factor 1).  A REDUCED-precision mode with its own stated tolerance (every matrix operand is
                                  rounded to 11 significant bits: ~1e-3 norm-wise per network output),
                                  never the default.  Differences at the boundary: (a) the hoisted map such a network reads is
                                  a map of HALVES -- njf_project_features* / njf_project_pyramid with this precision write
                                  `out` as _Float16 [B, Hf*Wf, N] (the projection itself stays error-compensated, its fp32
                                  result is rounded once), NjfFeatureMap.data points to halves and NjfFeatureMap.stride / the
                                  gmap offsets count halves (multiples of 8); (b) a whole 128 x 128 layer is one weight chunk:
                                  the packed ResnetFC keeps its NJF_RESNET_W_FLOATS extent and fills the first
                                  NJF_RESNET_CHUNKS_F16 chunk slots; (c) inference only: the training forwards (activation
                                  dumps) return the "unknown mode" error; (d) never part of a MIXED code */

/* njf_render_forward / njf_points_forward: density + colour networks in precision `d`, Jacobian head in `j` (both one of
 * F32 / F16X2 / F16F6; mixed forms exist for the two split precisions).  A plain NJF_PRECISION_* value means d = j. */
#define NJF_PRECISION_MIXED(d, j) ((d) | (((j) + 1) << 4))

#define NJF_JACOBIAN_NONE 0
#define NJF_JACOBIAN_MLP 1          /* ActionDecoderJacobianMLP (action_decoder_jacobian.py:261-337) */
#define NJF_JACOBIAN_TRANSFORMER 2  /* ActionDecoderJacobianTransformer (:340-446), host-folded, see decoder.py */

/* Reference-layout tensors of one ResnetFC (model_components/resnet_fc.py:82-128), all device fp32. */
typedef struct NjfResnetFcWeights {
  const float* lin_in_w;    /* [128, 63]  */
  const float* lin_in_b;    /* [128]      */
  const float* fc0_w[5];    /* blocks.i.fc_0.weight [128,128] */
  const float* fc0_b[5];    /* [128] */
  const float* fc1_w[5];    /* blocks.i.fc_1.weight [128,128] */
  const float* fc1_b[5];
  const float* lin_z_w[3];  /* lin_z.i.weight [128, 512] */
  const float* lin_z_b[3];  /* [128] */
  const float* lin_out_w;   /* [d_out, 128] */
  const float* lin_out_b;   /* [d_out] */
  int d_out;                /* 1 (proposal), 16 (density+15 features), 3*A (Jacobian) ; <= 32 */
} NjfResnetFcWeights;

/* color_head of ActionDecoderJacobian* (models/decoder/action_decoder_jacobian.py:315-322). */
typedef struct NjfColorHeadWeights {
  const float* w0; const float* b0;   /* [64, 31], [64]  (15 geometry features ++ 16 SH) */
  const float* w1; const float* b1;   /* [64, 64], [64] */
  const float* w2; const float* b2;   /* [3, 64],  [3]  */
} NjfColorHeadWeights;

/* Cameras + scene bounds shared by the fused kernels.  Inverses are taken by the caller
 * (torch.linalg.inv on device), mirroring transform_world2cam (rendering/geometry.py:59-65). */
typedef struct NjfCameras {
  const float* ctxt_w2c;     /* [B,4,4] inverse of the context cam2world */
  const float* ctxt_k;       /* [B,3,3] normalised context intrinsics */
  const float* trgt_w2c;     /* [B,4,4] inverse of the target cam2world (may be NULL when no flow is rendered) */
  const float* trgt_k;       /* [B,3,3] target intrinsics in pixels */
  const float* z_near;       /* [B] */
  const float* z_far;        /* [B] */
  const float* action;       /* [B,A] robot command (may be NULL -> flow outputs are skipped) */
  int batch;                 /* B */
  int action_dim;            /* A <= NJF_MAX_ACTION_DIM */
} NjfCameras;

/* Hoisted, channels-last feature map: G[b][y][x][c] = lin_z(F)[c] (see njf_project_features). */
typedef struct NjfFeatureMap {
  const float* data;   /* [B, Hf, Wf, stride] */
  int height, width;   /* Hf, Wf */
  int stride;          /* floats per texel (>= offset + NJF_ZDIM) */
} NjfFeatureMap;

/* ---- library info ------------------------------------------------------------------------ */
int njf_abi_version(void);
/* Rays one workgroup of the fused ray kernels renders (one per wave): NjfRenderOutputs.frame_partials has
 * ceil(B*R / njf_rays_per_workgroup()) rows. */
int njf_rays_per_workgroup(void);
const char* njf_error_string(int code);

/* ---- camera matrices -------------------------------------------------------------------------- */
/* out[i] = inverse(matrices[i]) for `count` row-major 4x4 fp32 matrices (Gauss-Jordan with partial pivoting evaluated
 * in float64 and rounded once, one thread per matrix).  Replaces torch.inverse on camera extrinsics (transform_world2cam, rendering/geometry.py:59-65;
 * project_world_coords_to_camera, :206-215), which PyTorch-ROCm runs as six rocSOLVER launches per call. */
int njf_invert_4x4(const float* matrices, int count, float* out, void* stream);

/* ---- weight packing (one-off per weight update) -------------------------------------------- */
/* Packs one ResnetFC into `w_out` [NJF_RESNET_W_FLOATS] / `b_out` [NJF_RESNET_B_FLOATS] and its
 * three lin_z layers into `wz_out` [512,384] (k-major) / `bz_out` [384] (inputs of
 * njf_project_features).  Replaces nothing in the reference: it is the layout change that lets
 * ResnetFC.forward (resnet_fc.py:130-154) run as one fused kernel.  NJF_PRECISION_F16: `wz_out` / `bz_out` are REQUIRED -- the
 * biases of blocks 0 and 1's fc_1 are folded into `bz_out` (the latent of the next block is added right behind them and a
 * bilinear footprint's weights sum to 1), so the weight blob and the lin_z pack of one network always come from the same call. */
int njf_pack_resnetfc(const NjfResnetFcWeights* src, float* w_out, float* b_out, float* wz_out, float* bz_out,
                      int precision, void* stream);
/* Same, writing lin_z into a wider [512, wz_ld] matrix (several nets side by side: pass wz_out + column offset). */
int njf_pack_resnetfc_ld(const NjfResnetFcWeights* src, float* w_out, float* b_out, float* wz_out, int wz_ld,
                         float* bz_out, int precision, void* stream);
int njf_pack_color_head(const NjfColorHeadWeights* src, float* w_out, float* b_out, int precision, void* stream);
/* One torch.nn.Linear [d_out, d_in] -> fragment-major block of ceil(d_in/32)*4*ceil(d_out/32)*256 floats
 * (+ zero-padded bias of 32*ceil(d_out/32) floats when b_out != NULL).  kind 0: plain.  kind 1: the input is
 * the 63-d positional encoding (slot order [sin 30 | x | y || cos 30 | z | 1], bias folded into slot 63). */
int njf_pack_linear(const float* w, const float* b, int d_out, int d_in, int kind, float* w_out, float* b_out,
                    int precision, void* stream);

/* ---- per-image feature projection ("lin_z hoist") ------------------------------------------ */
/* G[b,p,n] = sum_k F[b,k,p] * wz[k,n] + bz[n];  F is the encoder output [B,512,Hf,Wf] (NCHW),
 * wz [512,N] (k-major, as written by njf_pack_resnetfc*), bz [N], out [B, Hf*Wf, N].  Because bilinear interpolation is linear with weights
 * summing to 1, lin_z(grid_sample(F)) == grid_sample(G): this moves resnet_fc.py:138-141's
 * 3 x (512->128) GEMMs from per-point to per-texel.  `precision` selects the MFMA path as in the fused kernels
 * (operands stay fp32 in memory; NJF_PRECISION_F16X2 splits them on the fly). */
int njf_project_features(const float* feats, const float* wz, const float* bz, int batch, int hw, int n,
                         float* out, int precision, void* stream);
int njf_project_features_ld(const float* feats, const float* wz, int wz_ld, const float* bz, int batch, int hw, int n,
                            float* out, int precision, void* stream);

/* ---- feature-pyramid producer: encoder_resnet.py:78-86 fused with the lin_z hoist ----------- */
/* The encoder output is cat_l(bilinear_upsample(latent_l)) (align_corners=False) of its conv1 / layer1..3 latents.
 * Given the latents themselves (NCHW, level 0 at the output resolution, channel counts summing to 512 in
 * concatenation order) this computes the same hoisted map G = F . wz + bz without ever forming F: every level is
 * projected at its own resolution, the coarser ones are bilinearly up-sampled and added.  `workspace` (caller-owned)
 * holds the projected coarser levels: sum over l >= 1 of batch * height_l * width_l * n floats.  NJF_PRECISION_F16 with more
 * than one level: the sum is formed in fp32 and rounded once into `out` (halves); the workspace then holds the fp32 level-0
 * map IN FRONT of the coarser levels (batch * height_0 * width_0 * n more floats). */
typedef struct NjfPyramidLevel {
  const float* feats; /* [B, channels, height, width] */
  int channels;       /* multiple of 16 */
  int height, width;
} NjfPyramidLevel;
int njf_project_pyramid(const NjfPyramidLevel* levels, int num_levels, const float* wz, int wz_ld, const float* bz,
                        int batch, int n, float* out, float* workspace, int precision, void* stream);

/* The encoder output itself, channels-last: out [B*H_0*W_0, sum C_l] = cat_l(upsample_l(latent_l)) per texel
 * (models/encoder/encoder_resnet.py:78-86: F.interpolate(bilinear, align_corners=False) to the level-0 resolution +
 * torch.cat) in one pass from the NCHW latents -- the matrix the lin_z weight gradients contract against on the training
 * path (the forward pass never forms it, see njf_project_pyramid).  C_l % 4 == 0. */
int njf_upsample_concat(const NjfPyramidLevel* levels, int num_levels, int batch, float* out, void* stream);
/* Adjoint of njf_upsample_concat (the encoder tail's backward pass: what autograd runs as slice +
 * upsample_bilinear2d_backward for encoder_resnet.py:78-86).  grad [B*H_0*W_0, sum C_l] channels-last (the gradient
 * w.r.t. the matrix njf_upsample_concat writes) -> levels[l].feats receives the gradient of latent l, [B,C_l,H_l,W_l]
 * NCHW (the `feats` pointers of `levels` are the OUTPUTS here; geometry as for njf_upsample_concat).  Gather form, no
 * atomics: bit-reproducible.  One launch per level. */
int njf_upsample_concat_backward(const float* grad, const NjfPyramidLevel* levels, int num_levels, int batch, void* stream);

/* Channel order of the hoisted map.  Inside every block of `block_channels` channels (128 for a ResnetFC's lin_z layer,
 * 64 for the transformer head's query projection) logical feature f of the layer is stored at position
 * njf_hoisted_channel(f, block_channels, precision) -- the order in which the fused kernels' gather reads it, which
 * follows the MFMA precision (NJF_PRECISION_F32 / _F16X2 / _F16F6 / _F16, not a MIXED code) the network that owns the block
 * is packed for: F32 and F16X2 networks fetch per lane (the two lanes that own a point read adjacent 16-byte pieces), F16
 * networks likewise from a map of halves (8-channel pieces), F16F6 networks per quad of lanes (csrc/njf_device.h:
 * add_hoisted_latent).  The njf_pack_* entry points apply it
 * themselves.  Host function, no GPU work; for callers that write hoisted channels directly (flow_mlp's per-image action
 * bias, the transformer head's folded query weights).  Returns a negative error code for an invalid argument. */
int njf_hoisted_channel(int feature, int block_channels, int precision);

/* ---- ray generation: rendering/geometry.py:117-134 + :170-203 ------------------------------ */
/* coords [B,R,2] normalised pixel centres (NULL -> full H x W grid of get_pixel_coordinates),
 * k_inv [B,3,3] inverse normalised intrinsics, c2w [B,4,4];  outputs origins/directions [B,R,3], z [B,R]. */
int njf_generate_rays(const float* coords, int height, int width, const float* k_inv, const float* c2w,
                      int batch, int rays, float* origins, float* directions, float* z, void* stream);

/* ---- training forward: inputs of one ResnetFC's backward pass (resnet_fc.py:130-154) --------- */
/* Written per point p (P = points of the launch) when passed to a forward entry point; all four pointers must be
 * set.  The backward chain itself runs as library GEMMs on these matrices (host side: training.py). */
typedef struct NjfActivationDump {
  float* act;      /* [11, P, 128] ReLU'd input of fc_0 / fc_1 of block b at 2b / 2b+1, of lin_out at 10 */
  float* pe;       /* [P, 64] positional encoding in slot order [sin 30 | x | y | cos 30 | z | 1] */
  int* foot_idx;   /* [P, 4] texel indices (b*Hf*Wf + y*Wf + x) of the bilinear footprint */
  float* foot_w;   /* [P, 4] bilinear weights (nw, ne, sw, se) */
  unsigned* mask;  /* ABI v17, may be NULL: [11, P, 4] ReLU masks of `act` -- bit i of the 128 bits of (layer, point) says whether
                    * one of that layer input's 128 values is > 0, in an order only njf_resnetfc_backward needs to know (its
                    * `masks` argument): the backward chain reads these 16 bytes per point and layer instead of the 512 of `act` */
  int act_f16;     /* ABI v17: non-zero = `act` addresses [11, P, 128] HALVES (16-bit training storage: the weight-gradient GEMMs
                    * then read fp16 operands with fp32 accumulation; the reference trains on TF32 products) */
} NjfActivationDump;

/* ---- fused proposal pass: ray_samplers.py:497-552 (level loop body) ------------------------ */
/* For every ray: sample `s_in` bins (bins_in: [s_in+1] shared, or [B*R, s_in+1] when
 * bins_per_ray), evaluate DensityDecoderMlp.get_density (density_decoder.py:45-71), get_weights
 * (ray_samplers.py:77-101), weights**anneal (:529), PDFSampler (:351-451) with `u` ([s_out+1]
 * shared or per ray) -> bins_out [B*R, s_out+1].  Optional per-sample outputs (may be NULL):
 * weights_out, density_out [B*R, s_in]; `dump` (NULL for inference) receives the proposal net's activations
 * with P = B*R*s_in. */
int njf_proposal_forward(const float* origins, const float* directions, int rays_per_batch,
                         const NjfCameras* cams, const NjfFeatureMap* gmap, int gmap_offset,
                         const float* w_pack, const float* b_pack,
                         const float* bins_in, int bins_per_ray, int s_in,
                         const float* u, int u_per_ray, int s_out, float anneal,
                         float* bins_out, float* weights_out, float* density_out, const NjfActivationDump* dump,
                         int precision, void* stream);

/* ---- fused final pass: action_decoder_jacobian.py:147-215 + model.py:257-314 --------------- */
typedef struct NjfRenderOutputs {
  float* rgb;             /* [B*R,3]  render_rgb (model.py:257-270) */
  float* depth;           /* [B*R]    render_depth before the tensor-global clip (model.py:276) */
  float* step_minmax;     /* [B*R,2]  per-ray min/max of the sample mid-points (inputs of the clip, model.py:277) */
  float* flow;            /* [B*R,2]  render_optical_flow (model.py:288-314); NULL to skip */
  float* pos;             /* [B*R,3]  sum_s w x            (vis_output.ray_positions); NULL to skip */
  float* pos_warped;      /* [B*R,3]  sum_s w (x + flow)   (vis_output.ray_positions_warped); NULL to skip */
  float* action_features; /* [B*R,3A] sum_s w J            (render_action_features, model.py:281-286); NULL to skip */
  float* weights;         /* [B*R,S]  per-sample weights   (training_output / vis_output); NULL to skip */
  float* density;         /* [B*R,S]  per-sample density; NULL to skip */
  float* color;           /* [B*R,S,3]; NULL to skip */
  float* sample_flow;     /* [B*R,S,3] per-sample 3-D flow; NULL to skip */
  float* jacobian;        /* [B*R,S,3A] per-sample action features (encode_image, model.py:458-495); NULL to skip */
  /* training forward, all NULL for inference (P = B*R*S; layouts as in NjfActivationDump).  jac_pe / foot_idx /
   * foot_w select a training forward and are shared by both modes: with den_act + col_* it is the perception mode
   * (backward of the density net and the colour head), otherwise the action mode (backward of the Jacobian head:
   * jac_act is required for NJF_JACOBIAN_MLP; for NJF_JACOBIAN_TRANSFORMER it is optional and has another shape -- the head's
   * residual stream [4, P, 64] that njf_transformer_backward reads). */
  float* jac_act;         /* [11, P, 128] activations of the Jacobian ResnetFC; transformer head: [4, P, 64] residual stream */
  float* jac_pe;          /* [P, 64] positional encoding (the density and Jacobian nets see the same one) */
  int* foot_idx;          /* [P, 4] */
  float* foot_w;          /* [P, 4] */
  float* den_act;         /* [11, P, 128] activations of the density ResnetFC */
  float* col_in;          /* [P, 32] colour-head input [geo 15 | 1 | sh 16] (action_decoder_jacobian.py:315-322) */
  float* col_act;         /* [2, P, 64] ReLU'd outputs of the colour head's first and second layer */
  /* frame-level reductions folded into the kernel's epilogue (ABI v15; all may be NULL).  frame_partials
   * [ceil(B*R / njf_rays_per_workgroup()), 4]: per workgroup (min_t, max_t, sum (rgb - trgt_rgb)^2, sum (flow - trgt_flow)^2) --
   * the bounds of render_depth's tensor-global clip (model.py:277) and the numerators of the photometric / flow mse
   * (model_wrapper.py:117-163) of this launch's rays, reduced in a fixed order (bit-reproducible); the sums are 0 where
   * the target pointer is NULL.  njf_reduce_frame_partials folds the rows into one 4-vector. */
  float* frame_partials;
  const float* trgt_rgb;  /* [B*R,3] target colours of the rays, or NULL */
  const float* trgt_flow; /* [B*R,2] target optical flow of the rays, or NULL */
  /* ABI v17, training forwards, may be NULL: ReLU masks [11, P, 4] of jac_act / den_act (NjfActivationDump.mask) */
  unsigned* jac_mask;
  unsigned* den_mask;
  int dump_f16;           /* non-zero: jac_act / den_act address halves (NjfActivationDump.act_f16) */
} NjfRenderOutputs;

/* bins [B*R, S+1] are spacing-domain bin edges in [0,1] (output of njf_proposal_forward or a
 * sampler); the kernel maps them to Euclidean t = b*far + (1-b)*near (ray_samplers.py:240-243). */
int njf_render_forward(const float* origins, const float* directions, int rays_per_batch,
                       const NjfCameras* cams, const NjfFeatureMap* gmap, int gmap_offset_density,
                       int gmap_offset_jacobian, int jacobian_kind /* NJF_JACOBIAN_* */,
                       const float* w_density, const float* b_density,
                       const float* w_color, const float* b_color,
                       const float* w_jacobian, const float* b_jacobian,
                       const float* bins, int samples, const NjfRenderOutputs* out, int precision, void* stream);

/* ---- frame-level reductions of a (ray-sharded) render: model.py:277, model_wrapper.py:117-163 ------------------------------ */
/* partials [groups,4] (NjfRenderOutputs.frame_partials) -> out4 = (min, max, sum, sum) over the rows, one workgroup, fixed
 * summation order.  This is the 16-byte record a rank contributes to the one collective of a ray-sharded step. */
int njf_reduce_frame_partials(const float* partials, int groups, float* out4, void* stream);
/* Assemble the frame of a ray-sharded step from the all-gathered per-rank packets.  packets [world, packet_floats]; the
 * packet of rank k holds rgb [B,n_k,3] | depth [B,n_k] | flow [B,n_k,2] of its contiguous ray shard (n_k = R/world rays
 * per batch element, the first R % world ranks one more: parallel.shard_bounds) at float offsets 0, 3*B*cap, 4*B*cap with
 * cap = ceil(R / world), and its njf_reduce_frame_partials record in the last four floats.  Writes frame [B,R,6]
 * (rgb | depth | flow per ray) with depth clipped to the GLOBAL [min, max] (render_depth's clip, model.py:277, which the
 * sharded render defers to this point) and scalars6 = (global min, global max, S_rgb = global sum (rgb - trgt)^2, S_flow =
 * global sum (flow - trgt)^2, S_rgb * rgb_scale, S_flow * flow_scale), ranks folded in rank order: with rgb_scale =
 * 1 / (3 B R) and flow_scale = 0.01 / (2 B R) the last two are the frame's rgb loss and flow loss
 * (model_wrapper.py:119-121,148-160). */
int njf_assemble_frame(const float* packets, int world, int packet_floats, int batch, int rays_per_batch, float rgb_scale,
                       float flow_scale, float* frame, float* scalars6, void* stream);

/* ---- point-list evaluation (arbitrary xyz): density_decoder.py:45-71, model.py:416-456 ----- */
/* xyz [B,N,3] world-space points, dirs [B,N,3] or NULL.  mode 0: proposal net -> density [B*N].
 * mode 1: decoder -> density [B*N], color [B*N,3], flow [B*N,3], jacobian [B*N,3A], geo [B*N,15]
 * (any may be NULL); the Jacobian head is selected by jacobian_kind.  The decoder blobs must be
 * one allocation laid out [density | colour | jacobian] (also for njf_render_forward).
 * `features` (ABI v18, may be NULL; mode 1 with NJF_JACOBIAN_MLP, not NJF_PRECISION_F16): [5, B*N, 128] -- the head's residual
 * stream after each of its five blocks, i.e. ResnetFC.forward(compute_features=True).features (model_components/resnet_fc.py:
 * 141-151) block-major; the 640 hidden "action features" ActionDecoderFlowMlp.compute_flow returns (action_decoder_flow.py:
 * 168-176) are its transpose [B*N, 5, 128]. */
int njf_points_forward(const float* xyz, const float* dirs, int points_per_batch, const NjfCameras* cams,
                       const NjfFeatureMap* gmap, int gmap_offset_density, int gmap_offset_jacobian, int mode,
                       int jacobian_kind /* NJF_JACOBIAN_* */, const float* w_density, const float* b_density, const float* w_color, const float* b_color,
                       const float* w_jacobian, const float* b_jacobian,
                       float* density, float* color, float* flow, float* jacobian, float* geo, float* features, int precision,
                       void* stream);

/* ---- stand-alone sampler / compositing ops (API parity with the un-fused reference calls) -- */
/* RaySamples.get_weights (ray_samplers.py:77-101): deltas, densities [N,S] -> weights [N,S]. */
int njf_alpha_weights(const float* deltas, const float* densities, int rays, int samples, float* weights, void* stream);
/* PDFSampler.generate_ray_samples (ray_samplers.py:351-451) on spacing bins. */
int njf_pdf_resample(const float* weights, const float* bins_in, int bins_per_ray, int s_in, const float* u,
                     int u_per_ray, int s_out, float anneal, int rays, float* bins_out, void* stream);

/* ---- training: backward of alpha compositing ------------------------------------------------------------------------------- */
/* What autograd runs for RaySamples.get_weights (ray_samplers.py:77-101), render_rgb and the UN-clipped render_depth
 * (model.py:257-279), as one launch: deltas, steps (sample mid-points), sigma [rays,S], color [rays,S,3] (or NULL) are the
 * forward pass's per-sample fields; g_weights [rays,S], g_rgb [rays,3], g_depth [rays] (each may be NULL) the gradients
 * w.r.t. the weights, the composited colour and the depth before render_depth's clip.  Writes g_sigma [rays,S] and (when not
 * NULL) g_color [rays,S,3].  One wave per ray, fixed summation order (bit-reproducible). */
int njf_composite_backward(const float* deltas, const float* steps, const float* sigma, const float* color,
                           const float* g_weights, const float* g_rgb, const float* g_depth, int rays, int samples,
                           float* g_sigma, float* g_color, void* stream);

/* ---- training: backward of the pixel-aligned bilinear sampling -------------------------------- */
/* Input gradient of F.grid_sample(bilinear, border, align_corners=True) as get_pixel_aligned_features uses it
 * (model_components/pixel_aligned_features.py:29-33), in hoisted order: grad [P,channels] is the gradient w.r.t. the
 * sampled (already lin_z-projected) latent of every point, foot_idx [P,4] (int32 texel index on the flattened
 * [B*Hf*Wf] grid) / foot_w [P,4] the bilinear footprint the training forward dumped (NjfActivationDump).
 * out [texels,channels] += sum over points and footprint corners (accumulates: zero it first for a fresh gradient).
 * run_length >= 1: points p*run_length .. (p+1)*run_length-1 are handled by one thread per channel, which merges
 * consecutive contributions to the same texel before touching memory -- pass the samples per ray (points are ordered
 * ray-major, neighbouring samples mostly share texels); 1 = no merging.  The result does not depend on it beyond
 * rounding.  fp32 hardware atomics: the summation order, hence the last bits, vary from run to run (as they do for
 * ATen's grid_sampler_2d_backward on a GPU).
 * `slices` >= 1 gradients of the same points are scattered in one launch: slice s is read at grad + s * slice_stride
 * ([P,channels] each; the three lin_z latents of a ResnetFC are deltas[0], [2], [4] of njf_resnetfc_backward, i.e.
 * slice_stride = 2*P*128) and lands in columns [s*channels, (s+1)*channels) of out [texels, slices*channels]
 * (slices > 1 needs channels % 64 == 0). */
int njf_scatter_footprint(const float* grad, int slices, long long slice_stride, const int* foot_idx, const float* foot_w,
                          int points, int channels, int texels, int run_length, float* out, void* stream);

/* The whole data-gradient chain of one ResnetFC's backward pass in one launch: what autograd runs as 11 x (GEMM with the
 * transposed weight + ReLU mask + residual add) for model_components/resnet_fc.py:69-79,130-154.  `w_backward`
 * [NJF_RESNET_BACKWARD_CHUNKS * NJF_CHUNK_FLOATS] comes from njf_pack_resnetfc_backward (transposed weights, re-packed
 * after every optimiser step); `activations` [11,P,128] are the ReLU'd layer inputs the training forward dumped
 * (NjfActivationDump.act / NjfRenderOutputs.jac_act / den_act); `d_out` [P,d_out_dim] is the gradient w.r.t. lin_out's
 * output.  Writes `deltas` [11,P,128]: deltas[l+1] is the gradient w.r.t. the OUTPUT of the layer whose input is
 * activations[l] (l = 0..9), so that layer's weight gradient is deltas[l+1]^T activations[l] and its bias gradient the
 * column sum of deltas[l+1]; deltas[0], deltas[2], deltas[4] are the gradients w.r.t. the three hoisted latents
 * (lin_z outputs) and deltas[0] also w.r.t. lin_in's output.  Exact-fp32 MFMA.
 * `colsum_partial` (may be NULL) [ceil(P / 32), 11, 128]: per 32-point tile, the column sums of every deltas slice (the
 * bias gradients are their sum over the tiles: 32x less data than re-reading deltas; fixed summation order inside a
 * tile).
 * `masks` (ABI v17, may be NULL) [11,P,4]: the ReLU masks the same training forward dumped next to the activations
 * (NjfActivationDump.mask / NjfRenderOutputs.jac_mask / den_mask).  The chain needs only the SIGN of activations[l]; with `masks`
 * it does not read `activations` at all (which may then be NULL): 16 instead of 512 bytes per point and layer.
 * `precision` (ABI v17, of BOTH entry points: the blob is packed for it): NJF_PRECISION_F32 -- exact fp32 products, the default of
 * the host side -- or NJF_PRECISION_F16X2: hi*hi + hi*lo + lo*hi of fp16 halves (fp32-class: 2^-22 per product) on gradients the
 * kernel scales by a power of two taken from `d_out_absmax` (device scalar max|d_out|, required then) and scales back on the way
 * out.  (The reference trains on TF32 products, train.py:64-65.)
 * `deltas16` (ABI v17, may be NULL; needs `masks` and `d_out_absmax`): 16-bit training storage -- [11,P,128] HALVES receive
 * deltas x 2^k, k = 6 - exponent(max|d_out|) (the caller divides the weight-gradient products by 2^k), and `deltas` is then a
 * compact [3,P,128] fp32 array that receives slices 0, 2, 4 only (the latent gradients njf_scatter_footprint reads). */
#define NJF_RESNET_BACKWARD_CHUNKS 21
int njf_pack_resnetfc_backward(const NjfResnetFcWeights* src, float* w_out, int precision, void* stream);
int njf_resnetfc_backward(const float* d_out, int d_out_dim, const float* activations, const float* w_backward, int points,
                          float* deltas, float* colsum_partial, const unsigned* masks, int precision,
                          const float* d_out_absmax, void* deltas16, void* stream);

/* The data-gradient chain of the folded Jacobian transformer head (ActionDecoderJacobianTransformer.compute_jacobian,
 * action_decoder_jacobian.py:418-446 with model_components/transformer.py:38-135, as njf_render_forward evaluates it: decoder.py
 * folds keys / values / LayerNorm affines into 64 x 64 matrices) in one launch, exact fp32 MFMA -- what autograd runs for the
 * reference's parameterisation of the head.  Per layer l: n = norm(x), a = softmax_8(Mqk n + bqk), xm = x + Nov a + bo,
 * n2 = norm(xm), u = W1' n2 + b1', h = gelu(u), x_next = xm + W2 h + b2.
 * x [4,P,64]: the residual stream in front of layers 0, 1, 2 and behind layer 2 -- NjfRenderOutputs.jac_act of an action-mode
 * training forward with NJF_JACOBIAN_TRANSFORMER (slice 3 is not read here: the caller contracts it with d_out for the output
 * Linear's gradient).  d_out [P,d_out_dim]: gradient w.r.t. the head's 3A outputs.  keys = A (valid key slots per head).
 * Outputs: wg_x, wg_dy [12,P,64] -- per layer l the pairs the weight gradients contract over the points, at 4l + (0, 1, 2, 3) for
 * (Mqk, Nov, W1', W2): X = (n, a, n2, h), dY = (d dots, d xm, d u, d x_next); dW = dY^T X (K = points: library GEMM on the host),
 * bias gradients = column sums of dY (colsum_partial, may be NULL: [ceil(P / 32), 12, 64] per-tile sums, to be added over the
 * tiles).  half_storage != 0 (16-bit training storage, as `deltas16` of njf_resnetfc_backward): wg_x / wg_dy address [12,P,64]
 * HALVES, the dY stored x 2^k with k = 6 - exponent(*d_out_absmax) (device scalar max|d_out|, required then).  dx0 [P,64]: gradient w.r.t. the head's input (the query MLP's output): its weight gradient
 * contracts with the positional encoding, its hoisted feature part goes through njf_scatter_footprint.
 * njf_pack_transformer_backward: mats [3,4,64,64] = (Mqk, Nov, W1', W2) per layer, row-major [out][in]; biases [3,3,64] = (bqk, bo,
 * b1'); head_w [d_out,64]; -> w_out (NJF_TRANSFORMER_BACKWARD_CHUNKS chunks), b_out [3,192].
 * precision (ABI v19, both entry points, the same value): NJF_PRECISION_F32 -- exact fp32 MFMA -- or NJF_PRECISION_F16X2: split fp16
 * products (hi*hi + hi*lo + lo*hi, fp32-class inside fp16's range) for the layer's re-evaluation and for the chain, which then runs on
 * d_out x 2^k (d_out_absmax required) and scales its fp32 results back; the opt-in TF32-class form of njf_resnetfc_backward, for
 * callers whose reference trains on TF32 products (train.py:64-65). */
#define NJF_TRANSFORMER_BACKWARD_CHUNKS 13
int njf_pack_transformer_backward(const float* mats, const float* biases, const float* head_w, int d_out, float* w_out,
                                  float* b_out, int precision, void* stream);
int njf_transformer_backward(const float* x, const float* d_out, int d_out_dim, int keys, int points, const float* w_backward,
                             const float* b_backward, float* wg_x, float* wg_dy, float* dx0, float* colsum_partial,
                             int half_storage, const float* d_out_absmax, int precision, void* stream);

/* One layer step of the ResnetFC backward chain (model_components/resnet_fc.py:69-79,130-154 differentiated; what
 * autograd runs as compare + multiply + add + sum kernels):  out [P,C] = residual + upstream * [act > 0], with act the
 * ReLU'd forward activation the training forward dumped (residual may be NULL), and the bias gradient of the layer below
 * as column sums of out: partial_colsum [ceil(P / rows_per_block), C] receives one deterministic partial row per
 * workgroup (the caller adds the rows; NULL = no sums).  C % 4 == 0 and C/4 must divide 256 (C = 64, 128, ...). */
int njf_relu_backward(const float* upstream, const float* act, const float* residual, int points, int channels,
                      int rows_per_block, float* out, float* partial_colsum, void* stream);

/* Epilogue of a convolution of the frozen encoder trunk in eval mode (models/encoder/encoder_resnet.py:24-89: the torchvision
 * BasicBlock's  relu(bn(conv(x)))  and  relu(bn(conv(y)) + skip)  with batch norm on its running statistics; ABI v20):
 * out = [relu]( (x - running_mean[c]) / sqrt(running_var[c] + eps) * gamma[c] + beta[c] [+ skip] ), x / skip / out [batch, channels,
 * hw] fp32 NCHW (out may alias x; skip may be NULL), one launch instead of the library's batch-norm + add + ReLU kernels. */
int njf_bn_act(const float* x, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
               float eps, const float* skip, int relu, int batch, int channels, int hw, float* out, void* stream);

/* ---- inverse dynamics on the composited Jacobian field ---------------------------------------- */
/* The control loop of notebooks/real_world/2_inverse_dynamics.ipynb (cells 26-29: 100 Adam steps through
 * Model.infer_optical_flow, model.py:497-525) as a least-squares problem: optical_flow(a) = proj(x + M a) - proj(x),
 * x = mean_position [B,R,3] and M = jacobian [B,R,3,A] being the composited outputs of njf_render_forward (pos,
 * action_features re-ordered spatial-major), projection [B,3,4] = K . inv(E)[:3] of the target camera.  Runs
 * `iterations` Levenberg-Marquardt steps on sum_r mask_r |flow_r(a) - target_flow_r|^2 inside one launch (one
 * workgroup per batch element, deterministic reductions) and writes the command to action [B,A].
 * visible_mask [B,R] and init_action [B,A] may be NULL (all rays, start from zero).  A <= 16. */
int njf_solve_action(const float* mean_position, const float* jacobian, const float* projection, const float* target_flow,
                     const float* visible_mask, const float* init_action, int batch, int rays, int action_dim,
                     int iterations, float damping, float* action, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NJF_HIP_H */
