// sdfhip - the glue between the field's torch parameters and the fused kernels, and the scalar losses behind the renderer, as
// kernels (the step carried ~220 tiny ATen launches, a third of them from exactly these two places).
//
//  weightnorm_theta_{fwd,bwd}_kernel   every Linear of the SDF field is weight-normalised (nn.utils.weight_norm, sdf_field.py:314-317,
//                                      362): W = g v / ||v|| per output row.  The reference (and rounds 1 - 2 here) ran torch._weight_norm
//                                      per layer and concatenated: 14 + 14 + 1 launches per step, plus three AccumulateGrad adds per
//                                      layer.  Here ONE launch builds the flat theta the fused kernels pack from, and ONE launch turns
//                                      theta_bar into (v_bar, g_bar, bias_bar) of every layer, written where the caller points -
//                                      straight into the flat gradient buffer of distributed.FlatGradients when there is one.
//  surface_loss_*                      L1 colour loss (base_surface_model.py:402), eikonal (:406), curvature (neus_facto.py:313-325)
//                                      and the MonoSDF normal loss (losses.py:264-275): one pass over the rendered rays / the
//                                      per-sample gradients producing per-block partial sums, a one-block finish that scales them
//                                      into the loss scalars, and one elementwise backward.
#pragma once
#include "common.h"

constexpr int kThetaMaxLin = 24;
struct ThetaLayers {
  int32_t n_lin, total_rows;
  int32_t row_start[kThetaMaxLin + 1];  // first global row of layer l
  int32_t out_dim[kThetaMaxLin], in_dim[kThetaMaxLin];
  int64_t w_off[kThetaMaxLin], b_off[kThetaMaxLin];  // into theta
  const float* v[kThetaMaxLin];  // weight_v [out][in]
  const float* g[kThetaMaxLin];  // weight_g [out][1]
  const float* b[kThetaMaxLin];  // bias [out]
  float* v_bar[kThetaMaxLin];    // backward outputs (null: skipped)
  float* g_bar[kThetaMaxLin];
  float* b_bar[kThetaMaxLin];
};

SDFHIP_D int theta_layer_of_row(const ThetaLayers& L, const int row) {
  int l = 0;
  while (l + 1 < L.n_lin && row >= L.row_start[l + 1]) ++l;
  return l;
}
SDFHIP_D float theta_wave_sum(float x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
  return x;
}

// grid = ceil(total_rows / 4), block = 256: one wavefront per output row of one layer
__global__ __launch_bounds__(256) void weightnorm_theta_fwd_kernel(const ThetaLayers L, float* __restrict__ theta, float* __restrict__ inv_norm) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= L.total_rows) return;
  const int l = theta_layer_of_row(L, row), r = row - L.row_start[l], n = L.in_dim[l];
  const float* v = L.v[l] + (size_t)r * n;
  float ss = 0.0f;
  for (int c = lane; c < n; c += 64) ss = fmaf(v[c], v[c], ss);
  ss = theta_wave_sum(ss);
  const float nrm = sqrtf(ss);              // torch.norm_except_dim(v, 2, 0)
  const float scale = L.g[l][r] / nrm;      // torch._weight_norm: v * (g / norm)
  float* w = theta + L.w_off[l] + (size_t)r * n;
  for (int c = lane; c < n; c += 64) w[c] = v[c] * scale;
  if (lane == 0) {
    theta[L.b_off[l] + r] = L.b[l][r];
    inv_norm[row] = 1.0f / nrm;
  }
}
// accumulate: += into the outputs (a second consumer of the same parameters within one backward), else overwrite
__global__ __launch_bounds__(256) void weightnorm_theta_bwd_kernel(const ThetaLayers L, const float* __restrict__ theta_bar,
                                                                     const float* __restrict__ inv_norm, const int accumulate) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= L.total_rows) return;
  const int l = theta_layer_of_row(L, row), r = row - L.row_start[l], n = L.in_dim[l];
  const float* v = L.v[l] + (size_t)r * n;
  const float* wb = theta_bar + L.w_off[l] + (size_t)r * n;
  float dot = 0.0f;
  for (int c = lane; c < n; c += 64) dot = fmaf(wb[c], v[c], dot);
  dot = theta_wave_sum(dot);
  const float inv = inv_norm[row], g = L.g[l][r];
  // W = g v / |v|:  g_bar = <W_bar, v> / |v| ;  v_bar = (g / |v|) (W_bar - v <W_bar, v> / |v|^2)
  const float gb = dot * inv, k0 = g * inv, k1 = dot * inv * inv;
  if (L.v_bar[l] != nullptr) {
    float* vb = L.v_bar[l] + (size_t)r * n;
    for (int c = lane; c < n; c += 64) {
      const float val = k0 * (wb[c] - v[c] * k1);
      vb[c] = accumulate ? vb[c] + val : val;
    }
  }
  if (lane == 0) {
    if (L.g_bar[l] != nullptr) L.g_bar[l][r] = accumulate ? L.g_bar[l][r] + gb : gb;
    if (L.b_bar[l] != nullptr) {
      const float bb = theta_bar[L.b_off[l] + r];
      L.b_bar[l][r] = accumulate ? L.b_bar[l][r] + bb : bb;
    }
  }
}

// ------------------------------------------------------------------------------------------------ scalar losses of the surface models
enum { SL_RGB = 0, SL_EIK = 1, SL_CURV = 2, SL_NORMAL = 3, SL_COUNT = 4 };  // SL_NORMAL = L1 + cosine term (losses.py:264-275)
struct SurfaceLossArgs {
  const float* rgb;      // [N,3] rendered
  const float* image;    // [N,3] target
  const float* grad;     // [P,3] d sdf / dx per sample, or null
  const float* sdf;      // [P] and
  const float* taps;     // [P,6] numerical-gradient tap values (sampled_sdf), or null
  const float* n_pred;   // [N,3] rendered normal and
  const float* n_gt;     // [N,3] monocular normal prior, or null
  int64_t n_rays, n_points;
  float inv_delta2;      // 1 / delta^2 of the curvature stencil
  float scale[SL_COUNT]; // loss_k = scale_k * sum_k
  float* partial;        // [n_blocks][SL_COUNT]
  float* loss;           // [SL_COUNT]
  int32_t n_blocks;
  // backward
  const float* loss_bar[SL_COUNT];  // device scalars (null: that loss was not differentiated)
  float* rgb_bar;        // [N,3]
  float* grad_bar;       // [P,3]
  float* sdf_bar;        // [P]
  float* taps_bar;       // [P,6]
  float* n_pred_bar;     // [N,3]
};

SDFHIP_D void unit3(const float* p, float out[3], float& nrm) {
  nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  const float d = fmaxf(nrm, 1e-12f);  // F.normalize(p = 2, eps = 1e-12)
  out[0] = p[0] / d;
  out[1] = p[1] / d;
  out[2] = p[2] / d;
}

// grid = n_blocks (grid-stride over max(N, P) items), block = 256; deterministic: per-block partials, fixed-order finish
__global__ __launch_bounds__(256) void surface_loss_partial_kernel(const SurfaceLossArgs a) {
  __shared__ float red[4][SL_COUNT];
  float acc[SL_COUNT] = {0.f, 0.f, 0.f, 0.f};
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_rays; i += stride) {
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[SL_RGB] += fabsf(a.image[i * 3 + c] - a.rgb[i * 3 + c]);
    if (a.n_pred != nullptr) {
      float p[3], g[3], np, ng;
      unit3(a.n_pred + i * 3, p, np);
      unit3(a.n_gt + i * 3, g, ng);
      acc[SL_NORMAL] += (fabsf(p[0] - g[0]) + fabsf(p[1] - g[1]) + fabsf(p[2] - g[2])) + (1.0f - (p[0] * g[0] + p[1] * g[1] + p[2] * g[2]));
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_points; i += stride) {
    if (a.grad != nullptr) {
      const float gx = a.grad[i * 3], gy = a.grad[i * 3 + 1], gz = a.grad[i * 3 + 2];
      const float d = sqrtf(gx * gx + gy * gy + gz * gz) - 1.0f;
      acc[SL_EIK] += d * d;
    }
    if (a.taps != nullptr) {
      const float c2 = 2.0f * a.sdf[i];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) acc[SL_CURV] += fabsf((a.taps[i * 6 + 2 * ax] + a.taps[i * 6 + 2 * ax + 1] - c2) * a.inv_delta2);
    }
  }
#pragma unroll
  for (int k = 0; k < SL_COUNT; ++k) acc[k] = theta_wave_sum(acc[k]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < SL_COUNT; ++k) red[wave][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < SL_COUNT) a.partial[(size_t)blockIdx.x * SL_COUNT + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// one block: loss_k = scale_k * sum over blocks (double accumulation, fixed order)
// one block of 256: thread t sums the partials of loss t % SL_COUNT over blocks t / SL_COUNT, + 64, ... in double, the 64 lanes of a loss
// combine through LDS (one thread per loss over all blocks was ~1000 dependent round trips: 45 us for four numbers)
__global__ __launch_bounds__(256) void surface_loss_finish_kernel(const SurfaceLossArgs a) {
  __shared__ double red[256];
  static_assert(256 % SL_COUNT == 0, "lanes per loss");
  const int k = threadIdx.x % SL_COUNT, g = threadIdx.x / SL_COUNT;
  double s = 0.0;
  for (int b = g; b < a.n_blocks; b += 256 / SL_COUNT) s += (double)a.partial[(size_t)b * SL_COUNT + k];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < SL_COUNT) {
    double t = 0.0;
    for (int j = 0; j < 256 / SL_COUNT; ++j) t += red[j * SL_COUNT + threadIdx.x];
    a.loss[threadIdx.x] = (float)(t * (double)a.scale[threadIdx.x]);
  }
}
// elementwise backward; grid-stride over max(N, P)
__global__ __launch_bounds__(256) void surface_loss_bwd_kernel(const SurfaceLossArgs a) {
  const float lb_rgb = a.loss_bar[SL_RGB] ? a.loss_bar[SL_RGB][0] * a.scale[SL_RGB] : 0.0f;
  const float lb_eik = a.loss_bar[SL_EIK] ? a.loss_bar[SL_EIK][0] * a.scale[SL_EIK] : 0.0f;
  const float lb_cur = a.loss_bar[SL_CURV] ? a.loss_bar[SL_CURV][0] * a.scale[SL_CURV] : 0.0f;
  const float lb_nrm = a.loss_bar[SL_NORMAL] ? a.loss_bar[SL_NORMAL][0] * a.scale[SL_NORMAL] : 0.0f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  auto sgn = [](float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); };
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_rays; i += stride) {
    if (a.rgb_bar != nullptr)
#pragma unroll
      for (int c = 0; c < 3; ++c) a.rgb_bar[i * 3 + c] = -lb_rgb * sgn(a.image[i * 3 + c] - a.rgb[i * 3 + c]);  // l1_loss(image, rgb)
    if (a.n_pred_bar != nullptr) {
      float p[3], g[3], np, ng;
      unit3(a.n_pred + i * 3, p, np);
      unit3(a.n_gt + i * 3, g, ng);
      // d / d p_hat of |p_hat - g_hat|_1 + (1 - <p_hat, g_hat>), then through the normalisation p_hat = p / max(|p|, eps)
      float u[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) u[c] = lb_nrm * (sgn(p[c] - g[c]) - g[c]);
      const float d = fmaxf(np, 1e-12f);
      const float proj = np > 1e-12f ? (u[0] * p[0] + u[1] * p[1] + u[2] * p[2]) : 0.0f;  // below eps the divisor is a constant
#pragma unroll
      for (int c = 0; c < 3; ++c) a.n_pred_bar[i * 3 + c] = (u[c] - p[c] * proj) / d;
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_points; i += stride) {
    if (a.grad_bar != nullptr) {
      const float gx = a.grad[i * 3], gy = a.grad[i * 3 + 1], gz = a.grad[i * 3 + 2];
      const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
      const float k = nrm > 0.0f ? lb_eik * 2.0f * (nrm - 1.0f) / nrm : 0.0f;  // torch: the norm's subgradient at 0 is 0
      a.grad_bar[i * 3] = k * gx;
      a.grad_bar[i * 3 + 1] = k * gy;
      a.grad_bar[i * 3 + 2] = k * gz;
    }
    if (a.taps_bar != nullptr) {
      const float c2 = 2.0f * a.sdf[i];
      float sb = 0.0f;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const float s = sgn((a.taps[i * 6 + 2 * ax] + a.taps[i * 6 + 2 * ax + 1] - c2) * a.inv_delta2) * lb_cur * a.inv_delta2;
        a.taps_bar[i * 6 + 2 * ax] = s;
        a.taps_bar[i * 6 + 2 * ax + 1] = s;
        sb -= 2.0f * s;
      }
      a.sdf_bar[i] = sb;
    }
  }
}
