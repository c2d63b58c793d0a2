// sdfhip — per-configuration instantiation of the fused network kernels.  Each supported network shape is
// compiled in its own translation unit (inst_*.hip) so the build parallelises; the API layer picks the table.
#pragma once
#include "col_kernels.h"
#include "sdfrow_kernel.h"

struct FieldKernels {
  int nbh, nb0, nbf, nbs, nbc;  // block widths: hidden, in0, geometry feature, small colour inputs, colour hidden
  void (*geo_fwd)(int mode_train_geo_sdf, const GeoFwdArgs&, unsigned grid, hipStream_t);
  void (*geo_bwd)(const GeoBwdArgs&, unsigned grid, hipStream_t);   // tangent pass + data backward (after MODE_FULL)
  void (*geo_bwd1)(const GeoBwdArgs&, unsigned grid, hipStream_t);  // first-order data backward only (after sdfhip_geo_forward)
  void (*col_fwd)(const ColFwdArgs&, int save_activations, unsigned grid, hipStream_t);
  void (*col_bwd)(const ColBwdArgs&, unsigned grid, hipStream_t);
  void (*sdfrow)(const float* u_last, const float* qb_last, const float* sdfbar, int64_t n_tiles, int tiles_per_split,
                 float* partial, unsigned grid, hipStream_t);
  int act;  // hidden activation of the geometry-type network (common.h act_h): 0 Softplus(100); 1 ReLU - first-order entries only
            // (geo_fwd modes 1 - 3, geo_bwd1; geo_bwd is null)
  int layerwise = 0;  // 1: the geometry network runs one layer per launch (wide_kernels.h): the per-layer tensors are its inter-layer
                      // storage, so the workspace holds them in EVERY mode
  // first-order backward of points that have no feature cotangent (geo_bwd_kernel<FEATBAR = false>); null: use geo_bwd1 with zeros
  void (*geo_bwd1s)(const GeoBwdArgs&, unsigned grid, hipStream_t) = nullptr;
  int has_sdf_save = 0;  // geo_fwd mode 5 exists (sdf row only, activations saved: the taps of the numerical-gradient branch)
  int has_hp = 0;        // geo_fwd honours mode | kGeoHp (24-bit products in the first-order modes)
};

// Kernels that want more than 64 KiB of dynamic LDS must raise the per-function limit first.
template <class K, class A>
static inline void launch_lds(K kernel, const A& a, unsigned grid, unsigned block, size_t lds, hipStream_t s) {
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, s, a);
}

// One instantiation serves every DEPTH of a network with these block widths (layers and skip position are run-time values,
// the kernels loop over the layers).  mode: 0 = train/full (GRAD, SAVE, FEAT), 1 = geonetwork (FEAT only), 2 = sdf only,
// 3 = geonetwork saving z_l (differentiable), 4 = full forward without a backward to follow (GRAD, FEAT: d sdf / dx, nothing saved),
// 5 = sdf only, activations saved (differentiable; no feature rows: the six taps of the numerical-gradient branch).  The kernel families of one shape can live in separate translation units so the
// build parallelises: GEO_FWD_TRAIN / GEO_FWD_INFER / GEO_BWD define plain functions, COL defines the rest and the table.
// The second-order kernels run as two launches each (forward | chain, tangent | backward: geo_kernels.h, PHASE);
// -DSDFHIP_SINGLE_LAUNCH keeps them in one for A/B runs.
#ifdef SDFHIP_SINGLE_LAUNCH
#define SDFHIP_GEO_FWD_LAUNCH(GD, SAVE, a, grid, lds, s) launch_lds(geo_fwd_kernel<GD, true, SAVE, true>, a, grid, 256, lds, s)
#define SDFHIP_GEO_BWD_LAUNCH(GD, a, grid, lds, s) launch_lds(geo_bwd_kernel<GD>, a, grid, 256, lds, s)
#else
#define SDFHIP_GEO_FWD_LAUNCH(GD, SAVE, a, grid, lds, s)                                  \
  do {                                                                                     \
    launch_lds(geo_fwd_kernel<GD, true, SAVE, true, 1>, a, grid, 256, lds, s);             \
    launch_lds(geo_fwd_kernel<GD, true, SAVE, true, 2>, a, grid, 256, lds, s);             \
  } while (0)
#define SDFHIP_GEO_BWD_LAUNCH(GD, a, grid, lds, s)                                         \
  do {                                                                                     \
    launch_lds(geo_bwd_kernel<GD, true, 1>, a, grid, 256, lds, s);                         \
    launch_lds(geo_bwd_kernel<GD, true, 2>, a, grid, 256, lds, s);                         \
  } while (0)
#endif

#define SDFHIP_DEFINE_GEO_FWD_TRAIN(NAME, NBH, NB0, NBF)                                                                  \
  void sdfhip_geo_fwd_train_##NAME(const GeoFwdArgs& a, unsigned grid, hipStream_t s) {                                   \
    using GD = GeoDims<NBH, NB0, NBF>;                                                                                     \
    SDFHIP_GEO_FWD_LAUNCH(GD, true, a, grid, GD::lds_floats(kNsFwd, a.p.nl) * sizeof(float), s);                          \
  }

// mode | kGeoHp selects the 24-bit (six-term, precision mode 3) instantiation of the SDF-ROW modes (2 and 5) where the shape has one
// (HP = 1): the feature-row modes with six terms are 66 - 68 KB of code, over the 64 KB instruction cache, and nothing needs 24-bit
// feature rows (the caller runs the feature pass at the default precision and the sdf rows again: sdfhip_numfield_forward)
constexpr int kGeoHp = 0x100;
#define SDFHIP_DEFINE_GEO_FWD_INFER_(NAME, NBH, NB0, NBF, HP)                                                             \
  void sdfhip_geo_fwd_infer_##NAME(int mode_, const GeoFwdArgs& a, unsigned grid, hipStream_t s) {                        \
    using GD = GeoDims<NBH, NB0, NBF>;                                                                                     \
    const int mode = mode_ & 0xff;                                                                                         \
    if constexpr (HP != 0) {                                                                                               \
      if ((mode_ & kGeoHp) != 0 && (mode == 2 || mode == 5)) {                                                             \
        const size_t lds3 = GD::lds_floats(3, a.p.nl) * sizeof(float);                                                     \
        if (mode == 5) launch_lds(geo_fwd_kernel<GD, false, true, false, 0, 3>, a, grid, 256, lds3, s);                   \
        else launch_lds(geo_fwd_kernel<GD, false, false, false, 0, 3>, a, grid, 256, lds3, s);                            \
        return;                                                                                                            \
      }                                                                                                                    \
    }                                                                                                                      \
    const size_t lds = GD::lds_floats(kNsFwd, a.p.nl) * sizeof(float);                                                     \
    if (mode == 1) launch_lds(geo_fwd_kernel<GD, false, false, true>, a, grid, 256, lds, s);                              \
    else if (mode == 3) launch_lds(geo_fwd_kernel<GD, false, true, true>, a, grid, 256, lds, s);                          \
    else if (mode == 4) SDFHIP_GEO_FWD_LAUNCH(GD, false, a, grid, lds, s);                                                 \
    else if (mode == 5) launch_lds(geo_fwd_kernel<GD, false, true, false>, a, grid, 256, lds, s);                         \
    else launch_lds(geo_fwd_kernel<GD, false, false, false>, a, grid, 256, lds, s);                                       \
  }
#define SDFHIP_DEFINE_GEO_FWD_INFER(NAME, NBH, NB0, NBF) SDFHIP_DEFINE_GEO_FWD_INFER_(NAME, NBH, NB0, NBF, 0)
#define SDFHIP_DEFINE_GEO_FWD_INFER_HP(NAME, NBH, NB0, NBF) SDFHIP_DEFINE_GEO_FWD_INFER_(NAME, NBH, NB0, NBF, 1)

#define SDFHIP_DEFINE_GEO_BWD(NAME, NBH, NB0, NBF)                                                                        \
  void sdfhip_geo_bwd_##NAME(const GeoBwdArgs& a, unsigned grid, hipStream_t s) {                                         \
    using GD = GeoDims<NBH, NB0, NBF>;                                                                                     \
    SDFHIP_GEO_BWD_LAUNCH(GD, a, grid, GD::lds_floats(kNsGrad, a.p.nl) * sizeof(float), s);                               \
  }                                                                                                                       \
  void sdfhip_geo_bwd1_##NAME(const GeoBwdArgs& a, unsigned grid, hipStream_t s) {                                        \
    using GD = GeoDims<NBH, NB0, NBF>;                                                                                     \
    launch_lds(geo_bwd_kernel<GD, false>, a, grid, 256, GD::lds_floats(kNsGrad, a.p.nl) * sizeof(float), s);              \
  }                                                                                                                       \
  void sdfhip_geo_bwd1s_##NAME(const GeoBwdArgs& a, unsigned grid, hipStream_t s) {                                       \
    using GD = GeoDims<NBH, NB0, NBF>;                                                                                     \
    launch_lds(geo_bwd_kernel<GD, false, 0, false>, a, grid, 256, GD::lds_floats(kNsGrad, a.p.nl) * sizeof(float), s);    \
  }

#define SDFHIP_DEFINE_COL_AND_TABLE(NAME, NBH, NB0, NBF, NBS, NBC) SDFHIP_DEFINE_COL_AND_TABLE_(NAME, NBH, NB0, NBF, NBS, NBC, 0)
#define SDFHIP_DEFINE_COL_AND_TABLE_(NAME, NBH, NB0, NBF, NBS, NBC, HP)                                                   \
  void sdfhip_geo_fwd_train_##NAME(const GeoFwdArgs& a, unsigned grid, hipStream_t s);                                    \
  void sdfhip_geo_fwd_infer_##NAME(int mode, const GeoFwdArgs& a, unsigned grid, hipStream_t s);                          \
  void sdfhip_geo_bwd_##NAME(const GeoBwdArgs& a, unsigned grid, hipStream_t s);                                          \
  void sdfhip_geo_bwd1_##NAME(const GeoBwdArgs& a, unsigned grid, hipStream_t s);                                         \
  void sdfhip_geo_bwd1s_##NAME(const GeoBwdArgs& a, unsigned grid, hipStream_t s);                                        \
  namespace NAME##_ns {                                                                                                   \
  static void geo_fwd(int mode, const GeoFwdArgs& a, unsigned grid, hipStream_t s) {                                      \
    if ((mode & 0xff) == 0) sdfhip_geo_fwd_train_##NAME(a, grid, s);                                                      \
    else sdfhip_geo_fwd_infer_##NAME(mode, a, grid, s);                                                                   \
  }                                                                                                                       \
  using CD = ColDims<NBF, NBS, NBC>;                                                                                      \
  static void col_fwd(const ColFwdArgs& a, int save, unsigned grid, hipStream_t s) {                                      \
    const size_t lds = CD::lds_floats(kNsCol, a.p.nlc) * sizeof(float);                                                   \
    if (save) launch_lds(col_fwd_kernel<CD, true>, a, grid, 256, lds, s);                                                 \
    else launch_lds(col_fwd_kernel<CD, false>, a, grid, 256, lds, s);                                                     \
  }                                                                                                                       \
  static void col_bwd(const ColBwdArgs& a, unsigned grid, hipStream_t s) {                                                \
    launch_lds(col_bwd_kernel<CD>, a, grid, 256, CD::lds_floats(kNsGrad, a.p.nlc) * sizeof(float), s);                    \
  }                                                                                                                       \
  static void sdfrow(const float* z, const float* q, const float* sb, int64_t nt, int tps, float* part,                  \
                     unsigned grid, hipStream_t s) {                                                                      \
    sdfrow_grad_kernel<NBH><<<grid, 256, 0, s>>>(z, q, sb, nt, tps, part);                                                \
  }                                                                                                                       \
  }                                                                                                                       \
  const FieldKernels* sdfhip_kernels_##NAME() {                                                                           \
    static const FieldKernels k = {NBH, NB0, NBF, NBS, NBC, NAME##_ns::geo_fwd, sdfhip_geo_bwd_##NAME,                    \
                                   sdfhip_geo_bwd1_##NAME, NAME##_ns::col_fwd, NAME##_ns::col_bwd, NAME##_ns::sdfrow,     \
                                   0, 0, sdfhip_geo_bwd1s_##NAME, 1, HP};                                                 \
    return &k;                                                                                                            \
  }

// everything of one shape in one translation unit
#define SDFHIP_DEFINE_FIELD_KERNELS(NAME, NBH, NB0, NBF, NBS, NBC) \
  SDFHIP_DEFINE_GEO_FWD_TRAIN(NAME, NBH, NB0, NBF)                 \
  SDFHIP_DEFINE_GEO_FWD_INFER(NAME, NBH, NB0, NBF)                 \
  SDFHIP_DEFINE_GEO_BWD(NAME, NBH, NB0, NBF)                       \
  SDFHIP_DEFINE_COL_AND_TABLE(NAME, NBH, NB0, NBF, NBS, NBC)

// ---- first-order kernel family with a ReLU geometry-type network: the background fields.  NeRFField
// (fields/vanilla_nerf_field.py:37-114: 8 x 256 ReLU MLP with a skip, two 128-wide head layers) and TCNNNerfactoField's two 64-wide
// bias-free ReLU MLPs (fields/nerfacto_field.py:128-156, 211-225) run on the same fused kernels as the SDF field - geo_fwd_kernel
// without the analytic-normal chain (modes 1 - 3), geo_bwd_kernel<TANGENT = false>, the colour kernels, the split-K weight
// gradients - instantiated with ACT = 1.  No second-order entries: nothing differentiates these fields' outputs w.r.t. position.
#define SDFHIP_DEFINE_FIRST_ORDER_FIELD_KERNELS(NAME, NBH, NB0, NBF, NBS, NBC)                                            \
  namespace NAME##_ns {                                                                                                   \
  using GD = GeoDims<NBH, NB0, NBF, 1>;                                                                                    \
  using CD = ColDims<NBF, NBS, NBC>;                                                                                      \
  static void geo_fwd(int mode, const GeoFwdArgs& a, unsigned grid, hipStream_t s) {                                      \
    const size_t lds = GD::lds_floats(kNsFwd, a.p.nl) * sizeof(float);                                                     \
    if (mode == 1) launch_lds(geo_fwd_kernel<GD, false, false, true>, a, grid, 256, lds, s);                              \
    else if (mode == 3) launch_lds(geo_fwd_kernel<GD, false, true, true>, a, grid, 256, lds, s);                          \
    else if (mode == 2) launch_lds(geo_fwd_kernel<GD, false, false, false>, a, grid, 256, lds, s);                        \
    else abort(); /* second-order modes are refused in sdfhip_field_forward before they get here */                      \
  }                                                                                                                       \
  static void geo_bwd1(const GeoBwdArgs& a, unsigned grid, hipStream_t s) {                                               \
    launch_lds(geo_bwd_kernel<GD, false>, a, grid, 256, GD::lds_floats(kNsGrad, a.p.nl) * sizeof(float), s);              \
  }                                                                                                                       \
  static void col_fwd(const ColFwdArgs& a, int save, unsigned grid, hipStream_t s) {                                      \
    const size_t lds = CD::lds_floats(kNsCol, a.p.nlc) * sizeof(float);                                                   \
    if (save) launch_lds(col_fwd_kernel<CD, true>, a, grid, 256, lds, s);                                                 \
    else launch_lds(col_fwd_kernel<CD, false>, a, grid, 256, lds, s);                                                     \
  }                                                                                                                       \
  static void col_bwd(const ColBwdArgs& a, unsigned grid, hipStream_t s) {                                                \
    launch_lds(col_bwd_kernel<CD>, a, grid, 256, CD::lds_floats(kNsGrad, a.p.nlc) * sizeof(float), s);                    \
  }                                                                                                                       \
  static void sdfrow(const float* z, const float* q, const float* sb, int64_t nt, int tps, float* part,                  \
                     unsigned grid, hipStream_t s) {                                                                      \
    sdfrow_grad_kernel<NBH><<<grid, 256, 0, s>>>(z, q, sb, nt, tps, part);                                             \
  }                                                                                                                       \
  }                                                                                                                       \
  const FieldKernels* sdfhip_kernels_##NAME() {                                                                           \
    static const FieldKernels k = {NBH, NB0, NBF, NBS, NBC, NAME##_ns::geo_fwd, nullptr, NAME##_ns::geo_bwd1,             \
                                   NAME##_ns::col_fwd, NAME##_ns::col_bwd, NAME##_ns::sdfrow, 1};                         \
    return &k;                                                                                                            \
  }
