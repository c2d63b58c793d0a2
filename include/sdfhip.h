/* sdfhip — C ABI of the MI355X-native SDF volume-rendering hot path (libsdfhip.so, gfx950).
 *
 * Drop-in boundary for sdfstudio's Field / Sampler / Renderer plugin surface.  Each entry point names the
 * reference interface it replaces (paths relative to the upstream `nerfstudio/` package).  Conventions:
 *   - plain C, device pointers + sizes only, no torch / C++ types; row-major contiguous fp32 unless stated;
 *   - every call takes the HIP stream to launch on (pass torch's current stream) and never synchronises;
 *   - no hidden allocation on the hot path: the caller owns workspaces (sizes from the *_size queries);
 *     only *_create allocates (small immutable descriptor tables);
 *   - returns 0 on success, negative on error; sdfhip_last_error() returns a thread-local message;
 *   - per-point outputs are sized to sdfhip_padded_points(P) (P rounded up to 128) rows.
 */
#ifndef SDFHIP_H_
#define SDFHIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sdfhip_stream_t; /* hipStream_t */

int sdfhip_version(void);
const char* sdfhip_last_error(void);
int64_t sdfhip_padded_points(int64_t n_points);

/* ---------------------------------------------------------------------------------------------- hash grid
 * Replaces tcnn.Encoding(otype="HashGrid") as configured at fields/sdf_field.py:230-241 and
 * fields/density_fields.py:75-94 (n_levels, n_features_per_level, log2_hashmap_size, base_resolution,
 * per_level_scale, interpolation). */
typedef struct {
  int32_t n_levels;            /* <= 16 */
  int32_t n_features;          /* 2 */
  int32_t log2_hashmap_size;
  int32_t base_resolution;
  float per_level_scale;
  int32_t smoothstep;          /* 1: Smoothstep, 0: Linear */
} SdfHipGridCfg;

typedef struct {
  float scale;
  uint32_t resolution;
  uint32_t size;    /* entries */
  uint32_t offset;  /* first entry */
  uint32_t hashed;
} SdfHipGridLevel;

/* Host-only: per-level table (no GPU needed). levels: [n_levels]. Returns total entries via n_entries. */
int sdfhip_grid_levels(const SdfHipGridCfg* cfg, SdfHipGridLevel* levels, int64_t* n_entries);

/* The encoding as a standalone operator - what the reference gets from tcnn.Encoding("HashGrid") where it is not fused into a
 * field kernel (fields/nerfacto_field.py:137-156, the background field of BASELINE config 5).  x: [n_points, 3] in [0,1]
 * (positions outside wrap like tiny-cuda-nn's); feat: [n_points, n_levels * n_features] row-major, level-major columns.
 * Backward: table_bar (same shape as the table) is ACCUMULATED with atomics (caller zeroes); no gradient w.r.t. x. */
int sdfhip_grid_encode_forward(const SdfHipGridCfg* grid, const float* table, const float* x, int64_t n_points, float* feat,
                               sdfhip_stream_t stream);
int sdfhip_grid_encode_backward(const SdfHipGridCfg* grid, const float* x, int64_t n_points, const float* feat_bar, float* table_bar,
                                sdfhip_stream_t stream);
/* Test / diagnostic entry: the cell of every (point, level) exactly as every kernel of this library indexes the table (tcnn
 * grid_index / pos_fract; the hashed branch is the hash of the reference's own HashEncoding.hash_fn, field_components/
 * encodings.py:338-355, for power-of-two tables).  idx: [n_points, n_levels, 8] entry indices INCLUDING the level offset, corner k =
 * bit 0 -> +x, bit 1 -> +y, bit 2 -> +z; w: [n_points, n_levels, 3] interpolation weights (after Smoothstep where configured). */
int sdfhip_grid_cell_dump(const SdfHipGridCfg* grid, const float* x, int64_t n_points, uint32_t* idx, float* w, sdfhip_stream_t stream);

/* ---------------------------------------------------------------------------------------------- SDF field
 * Replaces nerfstudio.fields.sdf_field.SDFField's compute: forward_geonetwork (:380-410), get_sdf (:412-418),
 * the analytic gradient (:646-654), get_colors (:532-612), and their autograd backward (including the
 * double-backward of the gradient path). */
typedef struct {
  int32_t num_layers;            /* SDFFieldConfig.num_layers (hidden layers of the geometry MLP) */
  int32_t hidden_dim;
  int32_t geo_feat_dim;
  int32_t num_layers_color;
  int32_t hidden_dim_color;
  int32_t skip_layer;            /* 4 (SDFField.skip_in) or -1 when num_layers < 4 */
  int32_t pe_degree;             /* position_encoding_max_degree */
  int32_t use_position_encoding;
  int32_t appearance_dim;        /* appearance_embedding_dim */
  int32_t contract;              /* SceneContraction on the sample positions: 0 none, 1 order = inf, 2 order = None (L2) */
  float rgb_padding;
  SdfHipGridCfg grid;            /* n_levels = 0: no grid features (NeRFField) */
  /* The same fused kernels serve the background fields of the surface models (NeRFField, fields/
   * vanilla_nerf_field.py:37-114; TCNNNerfactoField's two MLPs, fields/nerfacto_field.py:128-156,211-225) through the first-order
   * entries sdfhip_geo_forward / _backward and sdfhip_color_forward / _backward: */
  int32_t activation;            /* hidden activation of the geometry-type network: 0 Softplus(beta = 100) (sdf_field.py:290), 1 ReLU */
  int32_t skip_style;            /* 0: cat([h, in0]) / sqrt(2), the layer below the skip H - in0 wide (sdf_field.py:280,403-404);
                                    1: cat([in0, h]), every layer H wide (field_components/mlp.py:86-88) */
  /* the ref-nerf options of get_colors (sdf_field.py:536-549, 566-583, 596-607; the bakedsdf / bakedangelo field settings,
   * configs/method_configs.py:270-286): bit 0 use_diffuse_color (the colour network's inputs lose the position and the gradient, its
   * kernels return the bare sigmoid: combine with sdfhip_refnerf_forward), bit 1 use_specular_tint, bit 2 use_reflections (the
   * direction encoding takes 2 (n . -d) n + d), bit 3 use_n_dot_v (one more input column).  Analytic-normal path only. */
  int32_t ref_flags;
  int32_t pe_off_axis;           /* SDFFieldConfig.off_axis: the position encoding projects on NeRFEncoding's 21 icosahedron directions
                                    (field_components/encodings.py:139-163, 191): in0 = 3 + 42 pe_degree + grid features */
} SdfHipFieldCfg;

#define SDFHIP_REF_DIFFUSE 1
#define SDFHIP_REF_TINT 2
#define SDFHIP_REF_REFLECT 4
#define SDFHIP_REF_NDOTV 8

typedef struct SdfHipField SdfHipField;

int sdfhip_field_create(const SdfHipFieldCfg* cfg, SdfHipField** out);
void sdfhip_field_destroy(SdfHipField* f);

/* Flat EFFECTIVE (weight-norm folded) parameter vector theta: for each geometry layer l = 0..num_layers
 * (glin{l}): W [out,in] row-major then b [out]; then each colour layer (clin{l}) likewise.
 * n_linear = (num_layers + 1) + (num_layers_color + 1). w_off/b_off/out_dim/in_dim: [n_linear]. */
int64_t sdfhip_field_theta_size(const SdfHipField* f);
int32_t sdfhip_field_num_linear(const SdfHipField* f);
int sdfhip_field_theta_layout(const SdfHipField* f, int64_t* w_off, int64_t* b_off, int32_t* out_dim, int32_t* in_dim);
int64_t sdfhip_field_table_size(const SdfHipField* f);    /* floats in the hash table (tcnn `params`) */
int64_t sdfhip_field_packed_size(const SdfHipField* f);   /* floats in the MFMA-packed weight blob */
/* bytes.  level: 0 = SDFHIP_MODE_SDF / SDFHIP_MODE_GEO calls; 1 = SDFHIP_MODE_FULL with training = 1 (every tensor the backward
 * needs, its staging buffers and split-K partials: ~48 KB per point at 8 x 256 + 4 x 256); 2 = SDFHIP_MODE_FULL with training = 0
 * (forward only, e.g. rendering under torch.no_grad(): ~11 KB per point). */
int64_t sdfhip_field_workspace_size(const SdfHipField* f, int64_t n_points, int32_t level);

/* theta -> MFMA operand order: every weight as five 16-bit parts (bf16 w0 + w1 + w2, fp16 hi + lo; a kernel streams the two of its
 * precision mode), W and W^T chunked per 32-wide k block (once per optimiser step). */
int sdfhip_field_pack(const SdfHipField* f, const float* theta, float* packed, sdfhip_stream_t stream);
/* Weight-normalised parameters -> theta and back, one launch each.  Every Linear of the SDF field is nn.utils.weight_norm'ed
 * (fields/sdf_field.py:314-317, 362): W = g v / ||v|| per output row; theta is the flat [W_0 | b_0 | W_1 | b_1 ...] vector of
 * sdfhip_field_theta_layout.  v / g / b: HOST arrays of n_lin = sdfhip_field_num_linear(f) DEVICE pointers (weight_v [out, in],
 * weight_g [out, 1], bias [out]).  inv_norm: [sdfhip_field_weightnorm_rows(f)] scratch the backward needs.  Backward: v_bar / g_bar /
 * b_bar are host arrays of device pointers the gradients are written to (accumulate = 1: added to) - e.g. straight into the
 * flat gradient buffer of a data-parallel exchange; a null entry skips that output. */
int64_t sdfhip_field_weightnorm_rows(const SdfHipField* f);
int sdfhip_field_theta_from_weightnorm(const SdfHipField* f, const float* const* v, const float* const* g, const float* const* b,
                                       int32_t n_lin, float* theta, float* inv_norm, sdfhip_stream_t stream);
int sdfhip_field_theta_backward_weightnorm(const SdfHipField* f, const float* const* v, const float* const* g, const float* const* b,
                                           int32_t n_lin, const float* inv_norm, const float* theta_bar, float* const* v_bar,
                                           float* const* g_bar, float* const* b_bar, int32_t accumulate, sdfhip_stream_t stream);

enum {
  SDFHIP_MODE_SDF = 0,     /* get_sdf: sdf only                                  */
  SDFHIP_MODE_GEO = 1,     /* forward_geonetwork: sdf + geometry feature         */
  SDFHIP_MODE_FULL = 2     /* get_outputs: sdf, d sdf/dx, rgb (+ saves for backward when training) */
};

/* Sample positions are origins[ray] + dirs[ray] * starts[ray, s] (cameras/rays.py:61-73); pass dirs = starts = NULL
 * and n_samples = 1 to evaluate at explicit positions origins[P,3].
 * Outputs (rows = sdfhip_padded_points(P)): sdf [rows], grad [rows,3], rgb [rows,3], feat [rows, geo_feat_dim]
 * (feat may be NULL; grad/rgb only written in MODE_FULL). emb: per-ray appearance embedding [n_rays, appearance_dim]
 * or NULL (zeros, sdf_field.py:554-564). level_mask: [n_levels*n_features] (hash_encoding_mask).
 * training (MODE_FULL only): 1 = sdfhip_field_backward will follow with the same workspace (level 1); 0 = forward only: the
 * kernels save nothing (no r_l / h_l stores, workspace level 2). */
int sdfhip_field_forward(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                         const float* origins, const float* dirs, const float* starts, int64_t n_rays, int32_t n_samples,
                         const float* emb, int32_t mode, int32_t training, void* workspace,
                         float* sdf, float* grad, float* rgb, float* feat, sdfhip_stream_t stream);

/* Backward of a MODE_FULL training forward with the same workspace.  Upstream gradients (any may be NULL):
 * sdf_bar [P], grad_bar [P,3], rgb_bar [P,3].  theta_bar [theta_size] is overwritten; table_bar [table_size] and
 * emb_bar [n_rays, appearance_dim] (may be NULL) are accumulated into (caller zeroes). */
int sdfhip_field_backward(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                          int64_t n_rays, int32_t n_samples, void* workspace,
                          const float* sdf_bar, const float* grad_bar, const float* rgb_bar,
                          float* theta_bar, float* table_bar, float* emb_bar, sdfhip_stream_t stream);

/* The same with one more upstream cotangent: feat_bar [P, geo_feat_dim] (or NULL) for a consumer of the geometry feature OUTSIDE the colour
 * network - the ref-nerf diffuse / tint heads (sdfhip_refnerf_backward), fed by sdfhip_field_forward's `feat` output in MODE_FULL. */
int sdfhip_field_backward_feat(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                               int64_t n_rays, int32_t n_samples, void* workspace,
                               const float* sdf_bar, const float* grad_bar, const float* rgb_bar, const float* feat_bar,
                               float* theta_bar, float* table_bar, float* emb_bar, sdfhip_stream_t stream);

/* The ref-nerf colour combination of SDFField.get_colors (sdf_field.py:536-540, 596-607), use_diffuse_color:
 *   rgb = clamp(tint * s + sigmoid(W_d feat + b_d - log 3), 0, 1) * (1 + 2 pad) - pad,  tint = sigmoid(W_t feat + b_t) or 0.5 (w_t NULL)
 * s_rgb [P,3]: the colour network's sigmoid output (sdfhip_field_forward's rgb under SDFHIP_REF_DIFFUSE), feat [P, geo_feat_dim],
 * w_d / w_t [3, geo_feat_dim] and b_d / b_t [3]: diffuse_color_pred / specular_tint_pred (sdf_field.py:333-336; plain Linear layers).
 * backward: workspace of sdfhip_refnerf_workspace_size bytes; s_bar [P,3] (-> sdfhip_field_backward_feat's rgb_bar), feat_bar
 * [P, geo_feat_dim] (-> its feat_bar) and the heads' gradients are OVERWRITTEN.  Deterministic (no atomics). */
int64_t sdfhip_refnerf_workspace_size(int64_t n_points, int32_t geo_feat_dim);
int sdfhip_refnerf_forward(const float* s_rgb, const float* feat, const float* w_d, const float* b_d, const float* w_t, const float* b_t,
                           int64_t n_points, int32_t geo_feat_dim, float rgb_padding, float* rgb, sdfhip_stream_t stream);
int sdfhip_refnerf_backward(const float* s_rgb, const float* feat, const float* w_d, const float* b_d, const float* w_t, const float* b_t,
                            int64_t n_points, int32_t geo_feat_dim, float rgb_padding, const float* rgb_bar, void* workspace,
                            float* s_bar, float* feat_bar, float* w_d_bar, float* b_d_bar, float* w_t_bar, float* b_t_bar,
                            sdfhip_stream_t stream);

/* Differentiable geometry network on explicit positions: SDFField.forward_geonetwork (sdf_field.py:380-410) under autograd -
 * what the reference differentiates through in the sparse-SfM loss (base_surface_model.py:463) and, six more times per sample,
 * in the numerical-gradient path (sdf_field.py:433-453).  First order only (no d sdf / dx, no second-order terms).
 * positions [P,3] are used as given (no contraction).  sdf [rows], feat [P, geo_feat_dim] or NULL; rows = sdfhip_padded_points(P).
 * workspace: sdfhip_geo_workspace_size(f, P) bytes, kept by the caller until the backward. */
int64_t sdfhip_geo_workspace_size(const SdfHipField* f, int64_t n_points);
int sdfhip_geo_forward(const SdfHipField* f, const float* packed, const float* table, const float* level_mask,
                       const float* positions, int64_t n_points, void* workspace, float* sdf, float* feat, sdfhip_stream_t stream);
/* sdf_bar [P] and feat_bar [P, geo_feat_dim] (either may be NULL = zero).  The geometry-network entries of theta_bar [theta_size]
 * are overwritten (the colour-network entries are left untouched: the caller zeroes the vector); table_bar is accumulated into. */
int sdfhip_geo_backward(const SdfHipField* f, const float* packed, const float* level_mask, int64_t n_points, void* workspace,
                        const float* sdf_bar, const float* feat_bar, float* theta_bar, float* table_bar, sdfhip_stream_t stream);
/* The same pair when only the first n_feat_points points' geometry feature is taken (feat [n_feat_points, geo_feat_dim], feat_bar
 * likewise; the rest contribute through sdf only): the numerical-gradient branch (sdf_field.py:433-453, 639-644) evaluates 7 P points
 * and uses the feature of the P centre points - no layout conversion of the other 6 P x geo_feat_dim values in either direction. */
int sdfhip_geo_forward_n(const SdfHipField* f, const float* packed, const float* table, const float* level_mask, const float* positions,
                         int64_t n_points, int64_t n_feat_points, void* workspace, float* sdf, float* feat, sdfhip_stream_t stream);
int sdfhip_geo_backward_n(const SdfHipField* f, const float* packed, const float* level_mask, int64_t n_points, void* workspace,
                          int64_t n_feat_points, const float* sdf_bar, const float* feat_bar, float* theta_bar, float* table_bar,
                          sdfhip_stream_t stream);
/* The forward on positions taken from ray frustums - what the density / background fields evaluate (fields/nerfacto_field.py:225-246
 * get_density: ray_samples.frustums.get_positions() = origins + directions * (starts + ends) / 2, rays.py:46-55, then the field's
 * spatial distortion, field_components/spatial_distortions.py:66-92, SdfHipFieldCfg.contract).  origins / dirs [n_rays,3], starts / ends
 * [n_rays, n_samples]; ends == NULL: the frustum START points (rays.py:61-73).  x_out [P,3] or NULL: the contracted positions.  The
 * workspace and the backward are sdfhip_geo_backward's (P = n_rays * n_samples points). */
int sdfhip_geo_forward_rays(const SdfHipField* f, const float* packed, const float* table, const float* level_mask, const float* origins,
                            const float* dirs, const float* starts, const float* ends, int64_t n_rays, int32_t n_samples, void* workspace,
                            float* sdf, float* feat, float* x_out, sdfhip_stream_t stream);

/* Colour network as its own operator: SDFField.get_colors (sdf_field.py:532-612, ref-nerf options off) with every input supplied
 * by the caller - the numerical-gradient path (sdf_field.py:639-644) feeds it the finite-difference d sdf / dx.
 * feat [P, geo_feat_dim], x [P,3] (contracted positions), dirs [n_rays,3], grad [P,3] (RAW gradient, :572-578),
 * emb [n_rays, appearance_dim] or NULL (zeros, :554-564); rgb [rows,3], rows = sdfhip_padded_points(P), P = n_rays * n_samples.
 * workspace: sdfhip_color_workspace_size(f, P) bytes, kept until the backward. */
int64_t sdfhip_color_workspace_size(const SdfHipField* f, int64_t n_points);
int sdfhip_color_forward(const SdfHipField* f, const float* packed, const float* feat, const float* x, const float* dirs,
                         const float* grad, const float* emb, int64_t n_rays, int32_t n_samples, void* workspace, float* rgb,
                         sdfhip_stream_t stream);
/* rgb_bar [P,3].  The colour-network entries of theta_bar are overwritten (geometry entries untouched: the caller zeroes the
 * vector); feat_bar [P, geo_feat_dim] and grad_bar [P,3] (either may be NULL) are overwritten; emb_bar [n_rays, appearance_dim]
 * (may be NULL) is accumulated into. */
int sdfhip_color_backward(const SdfHipField* f, const float* packed, int64_t n_rays, int32_t n_samples, void* workspace,
                          const float* rgb_bar, float* theta_bar, float* feat_bar, float* grad_bar, float* emb_bar,
                          sdfhip_stream_t stream);

/* SDFField.get_outputs with use_numerical_gradients (fields/sdf_field.py:629-655; the field mode of the neus-facto-angelo preset,
 * configs/method_configs.py:381-450) as ONE forward and ONE backward operator: the geometry network on the P = n_rays * n_samples
 * contracted START positions and their six taps x +- delta e_k (sdf_field.py:431-453; 7 P points, tap-major), the central-difference
 * normal, get_colors on it.  origins / dirs [n_rays,3], starts [n_rays,n_samples], emb [n_rays, appearance_dim] or NULL.
 * Outputs: sdf7 [sdfhip_numfield_sdf_rows(P)] (rows 0..P-1: the samples' sdf; rows P k + i: tap k - 1 of sample i), grad [P,3],
 * rgb [P,3], taps [P,6] (`sampled_sdf`, sdf_field.py:644; may be NULL), x_out [P,3] contracted positions (may be NULL).
 * training != 0 keeps what the backward needs in `workspace` (sdfhip_numfield_workspace_size(f, P) bytes, caller-owned until then);
 * training == 0 needs sdfhip_numfield_inference_workspace_size(f, P) bytes only (nothing saved: eval renders of config 5 in chunks).
 * Below delta = 2e-3 the seven sdf evaluations use 24-bit products where the field's shape has such kernels. */
int64_t sdfhip_numfield_workspace_size(const SdfHipField* f, int64_t n_points);
int64_t sdfhip_numfield_inference_workspace_size(const SdfHipField* f, int64_t n_points);
int64_t sdfhip_numfield_sdf_rows(int64_t n_points);
int sdfhip_numfield_forward(const SdfHipField* f, const float* packed, const float* table, const float* level_mask, const float* origins,
                            const float* dirs, const float* starts, int64_t n_rays, int32_t n_samples, const float* emb, float delta,
                            int32_t training, void* workspace, float* sdf7, float* grad, float* rgb, float* taps, float* x_out,
                            sdfhip_stream_t stream);
/* Cotangents sdf_bar [P], grad_bar [P,3], rgb_bar [P,3], taps_bar [P,6] (each may be NULL = zero).  theta_bar [theta_size] is
 * overwritten, table_bar and emb_bar [n_rays, appearance_dim] (may be NULL) are accumulated into. */
int sdfhip_numfield_backward(const SdfHipField* f, const float* packed, const float* level_mask, int64_t n_rays, int32_t n_samples,
                             float delta, void* workspace, const float* sdf_bar, const float* grad_bar, const float* rgb_bar,
                             const float* taps_bar, float* theta_bar, float* table_bar, float* emb_bar, sdfhip_stream_t stream);

/* Small per-ray operators of the "grid" background field (fields/nerfacto_field.py:65-332, base_surface_model.py:181-187).
 * sdfhip_sh4_embed: out [n_rays, 16 + emb_dim] = [tiny-cuda-nn SphericalHarmonics degree 4 of get_normalized_directions(dirs) = (dirs + 1) / 2
 * (nerfacto_field.py:128-134,283-285) | emb [n_rays, emb_dim] or zeros (NULL)]: the per-ray inputs of the field's colour network.
 * sdfhip_embedding_backward: backward of rows = weight[idx] (field_components/embedding.py): out [n_rows, dim] (overwritten) = sum of
 * grad [n, dim] over the n with idx[n] == row (idx: int64 device tensor), fixed summation order, dim <= 64. */
/* Pinhole rays for a batch of (camera, pixel) draws - the per-batch arithmetic of the reference's PixelSampler + RayGenerator
 * (data/utils/pixel_samplers.py:47-50; model_components/ray_generators.py:49-63 -> cameras/cameras.py:462-640 for a perspective camera
 * without distortion).  u [n_rays,3] uniforms in [0,1): camera = floor(u0 n_cams), y = floor(u1 height), x = floor(u2 width); direction
 * = R_cam [(x + 0.5 - cx) / fx, (y + 0.5 - cy) / fy, 1] with rot [n_cams,3,3] row-major camera-to-world (x right, y down, z forward).
 * origins [n,3] = centers[cam], dirs [n,3] unit, norm [n] = the direction's length before normalisation (RayBundle.directions_norm),
 * cam [n] int64. */
int sdfhip_generate_rays(const float* u, const float* centers, const float* rot, int32_t n_cams, int32_t height, int32_t width, float fx,
                         float fy, float cx, float cy, int64_t n_rays, float* origins, float* dirs, float* norm, int64_t* cam,
                         sdfhip_stream_t stream);
int sdfhip_sh4_embed(const float* dirs, const float* emb, int64_t n_rays, int32_t emb_dim, float* out, sdfhip_stream_t stream);
int sdfhip_embedding_backward(const int64_t* idx, const float* grad, int64_t n, int32_t dim, int64_t n_rows, float* out,
                              sdfhip_stream_t stream);

/* ---------------------------------------------------------------------------------------------- proposal density field
 * Replaces nerfstudio.fields.density_fields.HashMLPDensityField.get_density / density_fn (:99-118; base_field.py:48-65):
 * L-inf contraction of the frustum MIDPOINT, (x+2)/4, tcnn HashGrid(5 levels, F=2, linear) + FullyFusedMLP(16, ReLU,
 * no bias) -> trunc_exp.  Parameters: table, w1 [16,10], w2 [1,16]. */
int sdfhip_proposal_forward(const SdfHipGridCfg* grid, const float* table, const float* w1, const float* w2,
                            const float* origins, const float* dirs, const float* starts, const float* ends,
                            int64_t n_rays, int32_t n_samples, int32_t contract, float* density, sdfhip_stream_t stream);
/* workspace: sdfhip_proposal_workspace_size() bytes. table_bar accumulated (caller zeroes); w1_bar / w2_bar overwritten. */
int64_t sdfhip_proposal_workspace_size(void);
int sdfhip_proposal_backward(const SdfHipGridCfg* grid, const float* table, const float* w1, const float* w2,
                             const float* origins, const float* dirs, const float* starts, const float* ends,
                             int64_t n_rays, int32_t n_samples, int32_t contract, const float* density_bar,
                             void* workspace, float* table_bar, float* w1_bar, float* w2_bar, sdfhip_stream_t stream);

/* ---------------------------------------------------------------------------------------------- samplers
 * spaced bins: UniformLinDispPiecewiseSampler (model_components/ray_samplers.py:80-127, 221-247).
 * jitter: [n_rays] single-jitter draw (training) or NULL (eval).  bins: [n_rays, S+1]; starts/ends [n_rays, S]. */
int sdfhip_sample_spaced(const float* nears, const float* fars, const float* jitter, int64_t n_rays, int32_t n_samples,
                         float* bins, float* starts, float* ends, sdfhip_stream_t stream);
/* UniformSampler (ray_samplers.py:130-151): identity spacing, euclidean = x far + (1 - x) near.
 * jitter: [n_rays] (single_jitter) or, with jitter_per_sample != 0, [n_rays, n_samples+1] (ray_samplers.py:107-110), or NULL. */
int sdfhip_sample_uniform(const float* nears, const float* fars, const float* jitter, int32_t jitter_per_sample, int64_t n_rays,
                          int32_t n_samples, float* bins, float* starts, float* ends, sdfhip_stream_t stream);
/* Any SpacedSampler subclass (ray_samplers.py:55-247): spacing = SDFHIP_SPACING_* selects spacing_fn / spacing_fn_inv
 * (piecewise :221-247, uniform :130-151, linear disparity :154-175, sqrt :178-198, log :201-218); jitter as above
 * (single draw per ray, or per bin edge with jitter_per_sample != 0, ray_samplers.py:105-113). */
#define SDFHIP_SPACING_PIECEWISE 0
#define SDFHIP_SPACING_UNIFORM 1
#define SDFHIP_SPACING_LINDISP 2
#define SDFHIP_SPACING_SQRT 3
#define SDFHIP_SPACING_LOG 4
int sdfhip_sample_spacing(int32_t spacing, const float* nears, const float* fars, const float* jitter, int32_t jitter_per_sample,
                          int64_t n_rays, int32_t n_samples, float* bins, float* starts, float* ends, sdfhip_stream_t stream);
/* torch.optim.Adam step (the reference builds one Adam per parameter group: engine/optimizers.py:93-160, eps 1e-15 and the
 * learning rates of method_configs.py:483-500) over one contiguous slice of the flat parameter / gradient / moment buffers.
 * step = 1 for the first update (bias corrections 1 - beta^step); lr already carries the scheduler's factor
 * (engine/schedulers.py:170-215); grad_scale multiplies the gradient as it is read (1 / world_size after a SUM all-reduce:
 * the data-parallel mean costs no extra pass).  The four pointers must be the same slice of four equally aligned flat buffers
 * (same offset from a 16-byte boundary); n in floats. */
int sdfhip_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int64_t step, float grad_scale, sdfhip_stream_t stream);
/* torch.optim.AdamW step (engine/optimizers.py:60-64 AdamWOptimizerConfig; the `fields` group of the neuralangelo / bakedangelo presets,
 * configs/method_configs.py:229-232, 156-159): as sdfhip_adam_step, with weight_decay DECOUPLED - the parameter is multiplied by
 * 1 - lr * weight_decay ahead of the Adam update and the gradient carries no L2 term (optim/adamw.py). */
int sdfhip_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int64_t step, float grad_scale, sdfhip_stream_t stream);
/* PDFSampler (ray_samplers.py:250-370) in any spacing domain and with either jitter mode (one draw per ray, or per bin edge with
 * jitter_per_sample != 0: jitter [n_rays, s_out + 1], :321-326); include_original = False (the merge with the existing bins of
 * include_original = True is a sort of the two bin sets on the host side).  Otherwise as sdfhip_sample_pdf. */
int sdfhip_sample_pdf_spacing(int32_t spacing, const float* weights, const float* bins_in, const float* nears, const float* fars,
                              const float* jitter, int32_t jitter_per_sample, int64_t n_rays, int32_t s_in, int32_t s_out, float anneal,
                              float histogram_padding, float* bins_out, float* starts, float* ends, sdfhip_stream_t stream);
/* UniSurfSampler's surface search (ray_samplers.py:1030-1075): first outside-to-inside sign change of sdf [n_rays, n_samples] along
 * the samples at depths starts [n_rays, n_samples], depth z by linear interpolation, and the shrunk sampling interval
 * [max(z - (far - near) delta, near), min(z + (far - near) delta, far)] (unchanged where no surface was found: mask = 0). */
int sdfhip_surface_root(const float* sdf, const float* starts, const float* nears, const float* fars, int64_t n_rays, int32_t n_samples,
                        float delta, int32_t* mask, float* z, float* new_nears, float* new_fars, sdfhip_stream_t stream);
/* VolSDF's compositing in one launch per direction (models/volsdf.py:62-79): Laplace density of the SDF with
 * beta = LaplaceDensity.get_beta() (fields/sdf_field.py:48-76) -> weights (cameras/rays.py:146-192) -> rgb + background (1 - acc),
 * expected depth clipped to the batch's sample range, normal = sum w normalize(grad), accumulation
 * (model_components/renderers.py:81-92,196,245-259,294), and bg_trans [n_rays] = the transmittance in front of the LAST sample,
 * which the background model multiplies its colour with (volsdf.py:67-68).  density, weights: [n_rays, n_samples].
 * Backward: gradients w.r.t. sdf, grad (through the normal), rgb and beta (beta_bar [1] is ACCUMULATED); any upstream gradient may
 * be NULL. */
int sdfhip_volsdf_render_forward(const float* sdf, const float* grad, const float* rgb, const float* starts, const float* ends,
                                 const float* beta, const float* background, int64_t n_rays, int32_t n_samples, float* density,
                                 float* weights, float* out_rgb, float* out_depth_raw, float* out_depth, float* out_normal, float* out_acc,
                                 float* bg_trans, float* steps_minmax, sdfhip_stream_t stream);
int sdfhip_volsdf_render_backward(const float* sdf, const float* grad, const float* rgb, const float* starts, const float* ends,
                                  const float* beta, const float* background, int64_t n_rays, int32_t n_samples, const float* density,
                                  const float* weights, const float* out_depth_raw, const float* out_acc, const float* bg_trans,
                                  const float* steps_minmax, const float* rgb_bar, const float* depth_bar, const float* normal_bar,
                                  const float* acc_bar, const float* weights_bar, const float* bg_trans_bar, float* sdf_bar, float* grad_bar,
                                  float* rgbs_bar, float* beta_bar, sdfhip_stream_t stream);

/* ---- packed-sample path of NeuS-acc.  The reference calls three nerfacc (== 0.3.5, CUDA-only) operators; these
 * replace them.  "Packed": the samples of all rays in one array, ray r owning [offsets[r], offsets[r] + counts[r]).
 *
 * nerfacc.cuda.ray_marching (model_components/ray_samplers.py:1474-1484; ContractionType AABB, cone_angle 0): march each ray
 * from t_min to t_max in steps of `step`, keep the intervals whose mid point lies in an occupied voxel of binary
 * [resolution]^3 (torch.bool storage, x-major) over the region roi_aabb6 = (min xyz, max xyz) - a HOST array -, skip empty voxels
 * to their far boundary in whole steps.  Two passes: counts [n_rays] (int32); then, with offsets = exclusive scan of the
 * counts (int64), ray_indices (int64) / t_starts / t_ends [sum of counts]. */
int sdfhip_march_count(const float* origins, const float* dirs, const float* t_min, const float* t_max, const float* roi_aabb6_host,
                       const uint8_t* binary, int64_t n_rays, int32_t resolution, float step, int32_t* counts, sdfhip_stream_t stream);
int sdfhip_march_write(const float* origins, const float* dirs, const float* t_min, const float* t_max, const float* roi_aabb6_host,
                       const uint8_t* binary, int64_t n_rays, int32_t resolution, float step, const int64_t* offsets,
                       int64_t* ray_indices, float* t_starts, float* t_ends, sdfhip_stream_t stream);
/* The same two passes for a host that does not read the sample count back: step_dev (or NULL) is the step as a DEVICE scalar - NeuS-acc
 * recomputes it from a trained parameter every iteration (ray_samplers.py:1379-1382) - and the write pass drops samples at packed positions
 * >= capacity (< 0: unbounded): the caller allocates `capacity` entries, clamps (offsets, counts) on the device and checks for overflow at
 * its leisure.  What the reference does instead: nerfacc returns exact-size tensors, i.e. one device -> host read per step (:1474-1484). */
int sdfhip_march_count_dev(const float* origins, const float* dirs, const float* t_min, const float* t_max, const float* roi_aabb6_host,
                           const uint8_t* binary, int64_t n_rays, int32_t resolution, float step, const float* step_dev, int32_t* counts,
                           sdfhip_stream_t stream);
int sdfhip_march_write_capped(const float* origins, const float* dirs, const float* t_min, const float* t_max, const float* roi_aabb6_host,
                              const uint8_t* binary, int64_t n_rays, int32_t resolution, float step, const float* step_dev,
                              const int64_t* offsets, int64_t capacity, int64_t* ray_indices, float* t_starts, float* t_ends,
                              sdfhip_stream_t stream);
/* nerfacc.ray_resampling (model_components/ray_samplers.py:1496-1498; NeuSAccSampler(importance_sampling=True)): every ray that has
 * samples gets n_out new intervals whose n_out + 1 edges are the inverse CDF of its packed weights (padded to a sum >= 1e-5) at
 * u_j = 1 / (2 (n_out + 1)) + j (1 - 1 / (n_out + 1)) / n_out, linear inside the source intervals.  out_offsets [n_rays]: where
 * ray r's n_out outputs start (non-empty rays packed in order); rays without samples write nothing. */
int sdfhip_packed_resample(const float* t_starts, const float* t_ends, const float* weights, const int64_t* offsets,
                           const int32_t* counts, int64_t n_rays, int32_t n_out, const int64_t* out_offsets, float* out_starts,
                           float* out_ends, sdfhip_stream_t stream);
/* nerfacc.render_weight_from_alpha (models/neus_acc.py:103-107): weights_i = alpha_i T_i with T_i = prod_{j<i} (1 - alpha_j)
 * inside each ray's segment; trans receives T (the backward reads it).  Backward: alpha_bar from weights_bar. */
int sdfhip_packed_weights_forward(const float* alpha, const int64_t* offsets, const int32_t* counts, int64_t n_rays, float* weights,
                                  float* trans, sdfhip_stream_t stream);
int sdfhip_packed_weights_backward(const float* alpha, const float* weights, const float* trans, const float* weights_bar,
                                   const int64_t* offsets, const int32_t* counts, int64_t n_rays, float* alpha_bar, sdfhip_stream_t stream);
/* nerfacc.accumulate_along_rays (models/neus_acc.py:108-121): out[r, :] = sum over ray r's segment of weights_i * values[i, :]
 * (values NULL: of weights_i, dim = 1); rays without samples get zeros.  Deterministic (one wavefront per ray, no atomics). */
int sdfhip_packed_accumulate(const float* weights, const float* values, const int64_t* offsets, const int32_t* counts, int64_t n_rays,
                             int32_t dim, float* out, sdfhip_stream_t stream);
/* Scalar losses behind the renderer, fused: loss4 = { L1 colour loss F.l1_loss(image, rgb) (models/
 * base_surface_model.py:402), eikonal ((|grad| - 1)^2).mean() (:406), curvature |(tap sums - 2 sdf) / delta^2|.mean() (models/
 * neus_facto.py:313-325), MonoSDF normal loss (model_components/losses.py:264-275: L1 + cosine of the normalised normals) }, each
 * multiplied by scale4_host[k] (HOST array: multiplier / element count; 0 for a loss that is off).  grad [P,3], sdf [P] + taps [P,6],
 * n_pred + n_gt [N,3] may be NULL (those losses are then 0).  workspace: sdfhip_surface_loss_workspace_floats() floats.  Deterministic
 * (per-block partial sums, fixed-order finish).  Backward: loss_bar4 = HOST array of 4 DEVICE scalar pointers (NULL: not
 * differentiated); outputs may be NULL. */
int64_t sdfhip_surface_loss_workspace_floats(void);
int sdfhip_surface_loss_forward(const float* rgb, const float* image, int64_t n_rays, const float* grad, const float* sdf, const float* taps,
                                float delta, int64_t n_points, const float* n_pred, const float* n_gt, const float* scale4_host,
                                float* workspace, float* loss4, sdfhip_stream_t stream);
int sdfhip_surface_loss_backward(const float* rgb, const float* image, int64_t n_rays, const float* grad, const float* sdf, const float* taps,
                                 float delta, int64_t n_points, const float* n_pred, const float* n_gt, const float* scale4_host,
                                 const float* const* loss_bar4, float* rgb_bar, float* grad_bar, float* sdf_bar, float* taps_bar,
                                 float* n_pred_bar, sdfhip_stream_t stream);

/* MonoSDF depth prior: ScaleAndShiftInvariantLoss(alpha, scales = 1) (model_components/losses.py:278-409) exactly as the surface models
 * call it (models/base_surface_model.py:227,427-437): the N rays viewed as ONE rows x (N / rows) image (rows = 32), all-ones mask,
 * target = depth_gt * gt_scale + gt_shift (50, 0.5): closed-form scale / shift fit (:278-301), MSE / 2 + alpha * gradient matching,
 * batch-based reduction.  loss[1]; state10 carries the fit and the sums the backward needs to differentiate THROUGH it (as autograd
 * does in the reference).  One deterministic launch each way.  n_rays must be a multiple of rows. */
int sdfhip_mono_depth_loss_forward(const float* depth_pred, const float* depth_gt, int64_t n_rays, int32_t rows, float gt_scale,
                                   float gt_shift, float alpha, float* loss, float* state10, sdfhip_stream_t stream);
int sdfhip_mono_depth_loss_backward(const float* depth_pred, const float* depth_gt, int64_t n_rays, int32_t rows, float gt_scale,
                                    float gt_shift, float alpha, const float* state10, const float* loss_bar, float* pred_bar,
                                    sdfhip_stream_t stream);

/* Foreground-mask loss (models/base_surface_model.py:415-420): mult * binary_cross_entropy(clip(acc, 1e-3, 1 - 1e-3), label) with
 * acc[N] = the per-ray sum of the rendering weights; loss[1]; acc_bar[N].  One deterministic launch each way. */
int sdfhip_fg_mask_loss_forward(const float* acc, const float* label, int64_t n_rays, float mult, float* loss, sdfhip_stream_t stream);
int sdfhip_fg_mask_loss_backward(const float* acc, const float* label, int64_t n_rays, float mult, const float* loss_bar, float* acc_bar,
                                 sdfhip_stream_t stream);

/* Sensor-depth losses (model_components/losses.py:628-676 SensorDepthLoss, called at models/base_surface_model.py:440-449): the L1 between
 * the rendered depth depth_pred[N] (already divided by directions_norm, :303) and the sensor depth depth_gt[N] over the valid rays
 * (depth_gt > 0), the free-space loss on the samples in front of the truncation band (z < d - t: relu(t - sdf)^2) and the sdf loss inside
 * it (((z + sdf) - d)^2), z = starts / directions_norm (directions_norm[N] or NULL = 1), both means over ALL N * S samples weighted by
 * 1 - n_front / n resp. 1 - n_near / n.  losses3 = {l1, free space, sdf} WITHOUT the model's multipliers; state4 carries the weights the
 * backward needs.  workspace: sdfhip_sensor_depth_loss_workspace_size() bytes.  Forward: two deterministic launches (per-block sums in
 * double, one block adds them in a fixed order); backward: one launch writing sdf_bar[N * S] and depth_bar[N]. */
size_t sdfhip_sensor_depth_loss_workspace_size(void);
int sdfhip_sensor_depth_loss_forward(const float* depth_pred, const float* depth_gt, const float* sdf, const float* starts,
                                     const float* directions_norm, int64_t n_rays, int64_t n_samples, float truncation, void* workspace,
                                     float* losses3, float* state4, sdfhip_stream_t stream);
int sdfhip_sensor_depth_loss_backward(const float* depth_pred, const float* depth_gt, const float* sdf, const float* starts,
                                      const float* directions_norm, int64_t n_rays, int64_t n_samples, float truncation,
                                      const float* state4, const float* losses_bar3, float* sdf_bar, float* depth_bar,
                                      sdfhip_stream_t stream);

/* interlevel_loss_zip (model_components/losses.py:116-172), the part per proposal level: the field histogram (c [n_rays, s+1]
 * spacing bins, w [n_rays, s] weights; both constants) blurred with half-width `radius` (0.03 / 0.003 for the two levels, :138)
 * and resampled at the proposal bins cp [n_rays, s_p+1]; against the proposal weights wp [n_rays, s_p]:
 *   term = clip(w_gt - wp, 0)^2 / (wp + 1e-5),  dterm = d term / d wp   ([n_rays, s_p] each; w_gt optional, may be NULL).
 * The loss of the level is mean(term) (:171); only wp carries gradient. */
int sdfhip_interlevel_terms(const float* c, const float* w, const float* cp, const float* wp, int64_t n_rays, int32_t s, int32_t s_p,
                            float radius, float* term, float* dterm, float* w_gt, sdfhip_stream_t stream);
/* PDFSampler(include_original=False, single_jitter) (ray_samplers.py:275-370) applied to weights^anneal
 * (ProposalNetworkSampler :562).  Outputs are constants w.r.t. autograd (bins.detach(), :358). */
int sdfhip_sample_pdf(const float* weights, const float* bins_in, const float* nears, const float* fars,
                      const float* jitter, int64_t n_rays, int32_t s_in, int32_t s_out, float anneal,
                      float histogram_padding, float* bins_out, float* starts, float* ends, sdfhip_stream_t stream);

/* PDFSampler in UniformSampler spacing (the samplers NeuSSampler / ErrorBoundedSampler own, ray_samplers.py:607-611, 843-847);
 * jitter [n_rays] or [n_rays, s_out+1] (jitter_per_sample) or NULL. */
int sdfhip_sample_pdf_uniform(const float* weights, const float* bins_in, const float* nears, const float* fars, const float* jitter,
                              int32_t jitter_per_sample, int64_t n_rays, int32_t s_in, int32_t s_out, float histogram_padding,
                              float* bins_out, float* starts, float* ends, sdfhip_stream_t stream);
/* ErrorBoundedSampler.merge_ray_samples (ray_samplers.py:757-786), UniformSampler spacing: bins_1 [N,s1+1], bins_2 [N,s2+1] ->
 * merged_bins [N,s1+s2+1], merged_index [N,s1+s2] (int32, into cat(starts_1, starts_2)), euclidean merged_starts / merged_ends. */
int sdfhip_merge_uniform(const float* bins_1, const float* bins_2, const float* nears, const float* fars, int64_t n_rays, int32_t s1,
                         int32_t s2, float* merged_bins, int32_t* merged_index, float* merged_starts, float* merged_ends,
                         sdfhip_stream_t stream);
/* One outer iteration of VolSDF Algorithm 1 (ErrorBoundedSampler.generate_ray_samples, ray_samplers.py:650-683) up to the
 * resampling weights: sdf = gather(cat(sdf_a, sdf_b), index) ; d* (get_dstar :704-726) ; beta by bisection on the error bound
 * (get_updated_beta / get_error_bound :728-755) ; weights = get_weights_and_transmittance(laplace_density(sdf, beta)) ;
 * err_weights = (clamp(exp(cumsum(error per section)), 1e6) - 1) * transmittance (:676-683).
 * beta_in / beta_out [N]; beta0 [1] = density_fn.get_beta(); not_converged [1]: OR over rays of (beta_out > beta0), accumulated
 * (the caller zeroes it; it is the reference's `beta.max() > beta0` host decision, :671). */
int sdfhip_volsdf_bound_step(const float* bins_in, const float* sdf_a, const float* sdf_b, const int32_t* index, const float* nears,
                             const float* fars, const float* beta_in, const float* beta0, int64_t n_rays, int32_t s_a, int32_t s_b,
                             float eps, int32_t beta_iters, float* sdf_merged, float* beta_out, float* weights, float* err_weights,
                             int32_t* not_converged, sdfhip_stream_t stream);

/* One up-sampling step of NeuSSampler.generate_ray_samples (ray_samplers.py:851-886), UniformSampler spacing:
 *   sdf = gather(cat(sdf_a, sdf_b), index) (:864-868; index == NULL, s_b == 0 on the first step)
 *   alpha = rendering_sdf_with_fixed_inv_s(bins_in, sdf, inv_s) (:899-944) -> weights (rays.py:194-208) with a trailing 0 (:873)
 *   new samples = PDFSampler(histogram_padding = 1e-5, include_original = False)(weights, n_new) (:875-880, :303-358)
 *   merged = merge_ray_samples(current, new) (:757-786): sorted bins + the index into cat(starts_current, starts_new)
 * bins_in [N,S+1] (S = s_a + s_b); jitter NULL (eval: bin centres), [N] (single_jitter, :825, 836-840) or, with
 * jitter_per_sample != 0, [N,n_new+1] (one draw per new bin edge, :321-330).  Outputs: sdf_merged [N,S]; new_bins [N,n_new+1]; new_starts /
 * new_ends [N,n_new] euclidean (the caller evaluates get_sdf there and passes the result as the next step's sdf_b);
 * merged_bins [N,S+n_new+1]; merged_index [N,S+n_new] (int32); merged_starts / merged_ends [N,S+n_new] euclidean.
 * All outputs are constants w.r.t. autograd (bins.detach(), :771). */
int sdfhip_neus_upsample(const float* bins_in, const float* sdf_a, const float* sdf_b, const int32_t* index, const float* nears,
                         const float* fars, const float* jitter, int32_t jitter_per_sample, int64_t n_rays, int32_t s_a, int32_t s_b,
                         int32_t n_new, float inv_s, float* sdf_merged, float* new_bins, float* new_starts, float* new_ends, float* merged_bins,
                         int32_t* merged_index, float* merged_starts, float* merged_ends, sdfhip_stream_t stream);

/* ---------------------------------------------------------------------------------------------- weights + renderers
 * RaySamples.get_weights (cameras/rays.py:146-167) and its backward. */
int sdfhip_density_weights_forward(const float* density, const float* starts, const float* ends, int64_t n_rays,
                                   int32_t n_samples, float* weights, sdfhip_stream_t stream);
int sdfhip_density_weights_backward(const float* density, const float* starts, const float* ends, int64_t n_rays,
                                    int32_t n_samples, const float* weights_bar, float* density_bar, sdfhip_stream_t stream);

/* SDFField.get_alpha (fields/sdf_field.py:476-525) -> RaySamples.get_weights_from_alphas (cameras/rays.py:194-208)
 * -> RGBRenderer / DepthRenderer("expected") / SemanticRenderer(normals) / AccumulationRenderer
 * (model_components/renderers.py:81-92,245-259,294,196), fused, one wavefront per ray.
 * background: [3] or NULL (black).  steps_minmax: [2] scratch. depth is clipped to the batch-global [min,max] mid point. */
int sdfhip_neus_render_forward(const float* sdf, const float* grad, const float* rgb, const float* dirs,
                               const float* starts, const float* ends, const float* variance, const float* background,
                               float cos_anneal, int64_t n_rays, int32_t n_samples,
                               float* alpha, float* weights, float* out_rgb, float* out_depth_raw, float* out_depth,
                               float* out_normal, float* out_acc, float* steps_minmax, sdfhip_stream_t stream);
int sdfhip_neus_render_backward(const float* sdf, const float* grad, const float* rgb, const float* dirs,
                                const float* starts, const float* ends, const float* variance, const float* background,
                                float cos_anneal, int64_t n_rays, int32_t n_samples,
                                const float* alpha, const float* weights, const float* out_depth_raw, const float* out_acc,
                                const float* steps_minmax,
                                const float* rgb_bar, const float* depth_bar, const float* normal_bar, const float* acc_bar,
                                const float* weights_bar,
                                float* sdf_bar, float* grad_bar, float* rgbs_bar, float* variance_bar, sdfhip_stream_t stream);

/* The same with NeuS-facto's background merge (models/neus_facto.py:289-290 -> base_surface_model.py:256-290,
 * forward_background_field_and_merge) fused in: samples whose START position origins + dirs * starts lies outside the unit sphere take
 * alpha = 1 - exp(-(ends - starts) * bg_density) (RaySamples.get_alphas, cameras/rays.py:131-144) and colour bg_rgb of the background
 * field, the others the SDF field's alpha and colour; the rendered normal is the SDF field's everywhere (field_outputs[NORMAL] is not
 * merged).  origins [n_rays,3], bg_density [n_rays,n_samples], bg_rgb [n_rays,n_samples,3]; `alpha` returns the MERGED alpha.
 * Backward: bg_density_bar / bg_rgb_bar are overwritten (zero for inside samples), sdf_bar / rgbs_bar are zero for outside samples. */
int sdfhip_neus_render_bg_forward(const float* sdf, const float* grad, const float* rgb, const float* dirs, const float* starts,
                                  const float* ends, const float* variance, const float* background, float cos_anneal, int64_t n_rays,
                                  int32_t n_samples, const float* origins, const float* bg_density, const float* bg_rgb, float* alpha,
                                  float* weights, float* out_rgb, float* out_depth_raw, float* out_depth, float* out_normal,
                                  float* out_acc, float* steps_minmax, float* rgb_merged /* [n_rays,n_samples,3] or NULL */,
                                  sdfhip_stream_t stream);
int sdfhip_neus_render_bg_backward(const float* sdf, const float* grad, const float* rgb, const float* dirs, const float* starts,
                                   const float* ends, const float* variance, const float* background, float cos_anneal, int64_t n_rays,
                                   int32_t n_samples, const float* origins, const float* bg_density, const float* bg_rgb,
                                   const float* alpha, const float* weights, const float* out_depth_raw, const float* out_acc,
                                   const float* steps_minmax, const float* rgb_bar, const float* depth_bar, const float* normal_bar,
                                   const float* acc_bar, const float* weights_bar, float* sdf_bar, float* grad_bar, float* rgbs_bar,
                                   float* variance_bar, float* bg_density_bar, float* bg_rgb_bar, sdfhip_stream_t stream);

/* Data-parallel hosts (replaces nothing in the reference: DDP's bucket hooks, torch/nn/parallel/distributed.py, see the field's backward
 * as ONE autograd node): sdfhip_field_backward / sdfhip_numfield_backward on THIS field call `cb(user, table_bar, stream)` right after the
 * hash-table scatter has been enqueued on `stream` and before the weight-gradient GEMMs are - work enqueued behind `stream` inside the
 * callback (a reduce-scatter of table_bar) overlaps those GEMMs.  State of the handle, not of the process (no global state
 * besides immutable descriptors - a second model or the viewer's render thread, viewer/server/viewer_utils.py:109-135, has its own);
 * thread-safe; NULL clears it.  The callback must not synchronise. */
typedef void (*sdfhip_table_grad_cb)(void* user, const float* table_bar, sdfhip_stream_t stream);
int sdfhip_field_set_table_grad_callback(SdfHipField* field, sdfhip_table_grad_cb cb, void* user);

/* ---------------------------------------------------------------------------------------------- measurement (bench.py)
 * Optional HIP-event timing of the library's own launches, recorded on the stream each kernel is launched on.
 * enable(1) resets and starts recording; read() waits for the recorded events of a slot and returns the total
 * milliseconds and the number of timed launches.  Off by default; not thread safe. */
int sdfhip_profile_enable(int enable);          /* every slot on / off; returns the number of slots */
int sdfhip_profile_enable_slots(uint64_t slot_mask); /* only the slots whose bit is set record (0: off); returns the number of slots */
const char* sdfhip_profile_name(int slot);
int sdfhip_profile_read(int slot, double* total_ms, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* SDFHIP_H_ */
