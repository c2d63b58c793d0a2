// Hidden width 512 (16 blocks): the reference's neus-facto-bigmlp preset (configs/method_configs.py:503-523: SDFFieldConfig(num_layers=8,
// hidden_dim=512, num_layers_color=4), everything else at its defaults - in0 = 71 (3 blocks), 256-wide geometry feature and colour MLP).
// Two 16-block accumulator sets are the whole register file of a wave, so the geometry network runs LAYER BY LAYER on the kernels of
// wide_kernels.h; the colour network, the sdf-row gradient and the weight-gradient GEMMs are the ones of the 256-wide family.
#include <cstring>
#include "field_inst.h"
#include "wide_kernels.h"

const FieldKernels* sdfhip_kernels_A();

namespace W_ns {
using GD = GeoDims<16, 3, 8>;

template <class K, class... A>
static void launch(K kernel, unsigned grid, size_t lds_bytes, hipStream_t s, A... args) {
  if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds_bytes, s, args...);
}
static size_t lds(int ns) { return WideLds<GD>::floats(ns) * sizeof(float); }
// launches of the half-width kernels: 2 workgroups per tile group, in whole sets of 8 groups x 2 halves (wide_block)
static unsigned halves(unsigned groups) { return (groups + 7) / 8 * 16; }

template <int NS, bool ADD>
static void down_pass(const WideDownArgs& d, unsigned grid, hipStream_t s) {
  for (int l = d.p.nl - 1; l >= 0; --l) {
    if (l == 0 || l == d.p.skip) launch(wide_down_kernel<GD, NS, ADD, false, true>, grid, lds(NS), s, d, l, (int64_t)grid);  // the part that goes to in0
    if (l > 0) launch(wide_down_kernel<GD, NS, ADD, true, false>, halves(grid), lds(NS), s, d, l, (int64_t)grid);
  }
}

// mode: 0 = train/full (chain, everything saved), 1 = geonetwork, 2 = sdf only, 3 = geonetwork with a backward to follow, 4 = full forward
// without a backward.  The per-layer tensors are this path's only inter-layer storage: every mode writes u_l (and the chain r_l).
static void geo_fwd(int mode, const GeoFwdArgs& a, unsigned grid, hipStream_t s) {
  const int NL = a.p.nl, SKIP = a.p.skip;
  const int64_t G = grid;
  for (int l = 0; l < NL; ++l) {
    if (l == 0) launch(wide_fwd_kernel<GD, false, true>, halves(grid), lds(kNsFwd), s, a, l, G);
    else if (l == SKIP) launch(wide_fwd_kernel<GD, true, true>, halves(grid), lds(kNsFwd), s, a, l, G);
    else launch(wide_fwd_kernel<GD, true, false>, halves(grid), lds(kNsFwd), s, a, l, G);
  }
  const bool grad = mode == 0 || mode == 4;
  if (grad) launch(wide_out_kernel<GD, true, true>, grid, lds(kNsFwd), s, a);
  else if (mode == 2) launch(wide_out_kernel<GD, false, false>, grid, lds(kNsFwd), s, a);
  else launch(wide_out_kernel<GD, false, true>, grid, lds(kNsFwd), s, a);
  if (grad) {
    WideDownArgs d;
    std::memset(&d, 0, sizeof(d));
    d.p = a.p;
    for (int l = 0; l < NL; ++l) {
      d.x_tp[l] = a.r_tp[l];
      d.y_tp[l] = a.r_tp[l];
      d.u_tp[l] = a.u_tp[l];
    }
    d.e_tp = a.e_tp;
    down_pass<kNsFwd, false>(d, grid, s);
  }
}
template <bool TANGENT>
static void bwd_impl(const GeoBwdArgs& a, unsigned grid, hipStream_t s) {
  const int NL = a.p.nl, SKIP = a.p.skip;
  const int64_t G = grid;
  if constexpr (TANGENT) {
    for (int l = 0; l < NL; ++l) {
      if (l == 0) launch(wide_tan_kernel<GD, false, true>, halves(grid), lds(kNsGrad), s, a, l, G);
      else if (l == SKIP) launch(wide_tan_kernel<GD, true, true>, halves(grid), lds(kNsGrad), s, a, l, G);
      else launch(wide_tan_kernel<GD, true, false>, halves(grid), lds(kNsGrad), s, a, l, G);
    }
  }
  launch(wide_bwd_seed_kernel<GD, TANGENT>, halves(grid), lds(kNsGrad), s, a, G);
  WideDownArgs d;
  std::memset(&d, 0, sizeof(d));
  d.p = a.p;
  for (int l = 0; l < NL; ++l) {
    d.x_tp[l] = a.zb_tp[l];
    d.y_tp[l] = a.zb_tp[l];
    d.u_tp[l] = a.u_tp[l];
  }
  d.e_tp = a.in0bar_tp;
  down_pass<kNsGrad, TANGENT>(d, grid, s);
}
static void geo_bwd(const GeoBwdArgs& a, unsigned grid, hipStream_t s) { bwd_impl<true>(a, grid, s); }
static void geo_bwd1(const GeoBwdArgs& a, unsigned grid, hipStream_t s) { bwd_impl<false>(a, grid, s); }
static void col_fwd(const ColFwdArgs& a, int save, unsigned grid, hipStream_t s) { sdfhip_kernels_A()->col_fwd(a, save, grid, s); }
static void col_bwd(const ColBwdArgs& a, unsigned grid, hipStream_t s) { sdfhip_kernels_A()->col_bwd(a, grid, s); }
static void sdfrow(const float* u, const float* q, const float* sb, int64_t nt, int tps, float* part, unsigned grid, hipStream_t s) {
  sdfrow_grad_kernel<16><<<dim3(grid, 2), 256, 0, s>>>(u, q, sb, nt, tps, part);  // two groups of 8 blocks
}
}  // namespace W_ns

const FieldKernels* sdfhip_kernels_W() {
  static const FieldKernels k = {16, 3, 8, 3, 8, W_ns::geo_fwd, W_ns::geo_bwd, W_ns::geo_bwd1, W_ns::col_fwd, W_ns::col_bwd, W_ns::sdfrow, 0, 1};
  return &k;
}
