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

// ------------------------------------------------------------------------------------------------ MonoSDF depth prior + foreground mask
// ScaleAndShiftInvariantLoss(alpha, scales = 1) (model_components/losses.py:278-409) as the surface models call it
// (base_surface_model.py:427-437): the ray batch viewed as ONE rows x (N / rows) image, all-ones mask, target = gt * gt_scale + gt_shift.
//   (s, t) = argmin sum (s p + t - y)^2                          closed form 2 x 2 system, (0, 0) when singular      (:278-301)
//   loss   = sum d^2 / (2 N) + alpha * sum_edges |d_j - d_k| / N          d = s p + t - y, edges = horizontal + vertical neighbours
// One block: the batch is a few thousand rays, and one block makes the sums deterministic.  state (out): s, t, det, a00, a01, b0, b1, N as
// float, then G_s = sum g p and G_t = sum g with g = d loss / d d - what the backward needs to go THROUGH the fit (the reference
// differentiates compute_scale_and_shift too).
struct DepthLossArgs {
  const float* pred;  // [N]
  const float* gt;    // [N]
  int32_t n, rows, width;
  float gt_scale, gt_shift, alpha;
  float* loss;         // [1]
  float* state;        // [10]
  const float* loss_bar;  // device scalar (backward)
  float* pred_bar;        // [N]      (backward)
};
SDFHIP_D float depth_sign(const float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
template <int NV>
SDFHIP_D void block_sum_double(double (&v)[NV], double* red /* [NV][blockDim / 64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double t = v[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
    if (lane == 0) red[i * nw + wave] = t;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[i * nw + w];
    v[i] = t;
  }
  __syncthreads();
}
// g_i = d loss / d d_i (without the 1 / N): d_i + alpha * sum over the up-to-four neighbours of sign(d_i - d_nb)
SDFHIP_D float depth_loss_g(const DepthLossArgs& a, const int i, const float s, const float t) {
  auto d_at = [&](const int j) { return fmaf(s, a.pred[j], t) - fmaf(a.gt[j], a.gt_scale, a.gt_shift); };
  const int r = i / a.width, c = i - r * a.width;
  const float d = d_at(i);
  float e = 0.0f;
  if (c + 1 < a.width) e += depth_sign(d - d_at(i + 1));
  if (c > 0) e += depth_sign(d - d_at(i - 1));
  if (r + 1 < a.rows) e += depth_sign(d - d_at(i + a.width));
  if (r > 0) e += depth_sign(d - d_at(i - a.width));
  return fmaf(a.alpha, e, d);
}
__global__ __launch_bounds__(1024) void depth_loss_fwd_kernel(const DepthLossArgs a) {
  __shared__ double red[4 * 16];
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) {
    const double p = a.pred[i], y = fmaf(a.gt[i], a.gt_scale, a.gt_shift);
    v[0] += p * p;
    v[1] += p;
    v[2] += p * y;
    v[3] += y;
  }
  block_sum_double<4>(v, red);
  const double a00 = v[0], a01 = v[1], b0 = v[2], b1 = v[3], a11 = (double)a.n;
  // the reference forms the determinant and the solution in fp32 from fp32 sums (:288-299); the sums here are exact to double, the
  // 2 x 2 solve as well, rounded once
  const double det = a00 * a11 - a01 * a01;
  const float s = det != 0.0 ? (float)((a11 * b0 - a01 * b1) / det) : 0.0f;
  const float t = det != 0.0 ? (float)((-a01 * b0 + a00 * b1) / det) : 0.0f;
  double w[4] = {0.0, 0.0, 0.0, 0.0};  // sum d^2, sum |edge|, G_s, G_t
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) {
    const int r = i / a.width, c = i - r * a.width;
    const float p = a.pred[i];
    const float d = fmaf(s, p, t) - fmaf(a.gt[i], a.gt_scale, a.gt_shift);
    w[0] += (double)d * d;
    if (c + 1 < a.width) w[1] += fabsf(d - (fmaf(s, a.pred[i + 1], t) - fmaf(a.gt[i + 1], a.gt_scale, a.gt_shift)));
    if (r + 1 < a.rows) w[1] += fabsf(d - (fmaf(s, a.pred[i + a.width], t) - fmaf(a.gt[i + a.width], a.gt_scale, a.gt_shift)));
    const float g = depth_loss_g(a, i, s, t);
    w[2] += (double)g * p;
    w[3] += (double)g;
  }
  block_sum_double<4>(w, red);
  if (threadIdx.x == 0) {
    a.loss[0] = (float)(w[0] / (2.0 * a11) + (double)a.alpha * w[1] / a11);
    a.state[0] = s;
    a.state[1] = t;
    a.state[2] = (float)det;
    a.state[3] = (float)a00;
    a.state[4] = (float)a01;
    a.state[5] = (float)b0;
    a.state[6] = (float)b1;
    a.state[7] = (float)a11;
    a.state[8] = (float)(w[2] / a11);
    a.state[9] = (float)(w[3] / a11);
  }
}
// d loss / d p_i = lb * ( g_i s / N  +  G_s ds/dp_i  +  G_t dt/dp_i ),  the fit differentiated through a00' = 2 p_i, a01' = 1, b0' = y_i
__global__ __launch_bounds__(256) void depth_loss_bwd_kernel(const DepthLossArgs a) {
  const float s = a.state[0], t = a.state[1], det = a.state[2], a00 = a.state[3], a01 = a.state[4], b0 = a.state[5], b1 = a.state[6];
  const float n = a.state[7], gs = a.state[8], gt_ = a.state[9], lb = a.loss_bar[0];
  const float inv_det = det != 0.0f ? 1.0f / det : 0.0f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    const float p = a.pred[i], y = fmaf(a.gt[i], a.gt_scale, a.gt_shift);
    const float ddet = 2.0f * (p * n - a01);
    const float ds = ((n * y - b1) - s * ddet) * inv_det;
    const float dt = ((-b0 - a01 * y + 2.0f * p * b1) - t * ddet) * inv_det;
    (void)a00;
    a.pred_bar[i] = lb * (depth_loss_g(a, i, s, t) * s / n + gs * ds + gt_ * dt);
  }
}

// Foreground-mask loss (base_surface_model.py:415-420): binary_cross_entropy(clip(sum_s weights, 1e-3, 1 - 1e-3), fg_label), mean over rays.
// acc = the per-ray weight sum.  One block, deterministic; the log terms are clamped at -100 like aten's.
struct FgLossArgs {
  const float* acc;    // [N]
  const float* label;  // [N]
  int32_t n;
  float scale;         // loss multiplier / N
  float* loss;         // [1]
  const float* loss_bar;
  float* acc_bar;      // [N]
};
__global__ __launch_bounds__(1024) void fg_loss_fwd_kernel(const FgLossArgs a) {
  __shared__ double red[16];
  double v[1] = {0.0};
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) {
    const float c = fminf(fmaxf(a.acc[i], 1e-3f), 1.0f - 1e-3f), y = a.label[i];
    v[0] -= (double)(y * fmaxf(logf(c), -100.0f) + (1.0f - y) * fmaxf(logf(1.0f - c), -100.0f));
  }
  block_sum_double<1>(v, red);
  if (threadIdx.x == 0) a.loss[0] = (float)(v[0] * (double)a.scale);
}
__global__ __launch_bounds__(256) void fg_loss_bwd_kernel(const FgLossArgs a) {
  const float lb = a.loss_bar[0] * a.scale;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    const float x = a.acc[i], y = a.label[i];
    const bool inside = x >= 1e-3f && x <= 1.0f - 1e-3f;  // clip passes the gradient on its closed interval like aten's clamp backward
    const float c = fminf(fmaxf(x, 1e-3f), 1.0f - 1e-3f);
    a.acc_bar[i] = inside ? lb * (c - y) / (c * (1.0f - c)) : 0.0f;
  }
}

// Sensor-depth losses (model_components/losses.py:628-676, SensorDepthLoss, as models/base_surface_model.py:440-449 calls it): with the sensor
// depth d of a ray (d > 0 = valid), its samples' z = start / directions_norm and sdf values,
//   l1         = sum_valid |d - depth_pred| / (n_valid + 1e-6)
//   free space = mean over ALL samples of (relu(t - sdf) * front)^2 * (1 - n_front / n),   front = valid and z < d - t
//   sdf        = mean over ALL samples of ((z + sdf) - d)^2 * near * (1 - n_near / n),     near  = valid, not front, not (z > d + t)
// with n = n_front + n_near + 1e-6 (counts: constants of the backward, as in the reference where they come out of integer sums).
// Two launches forward (per-block partial sums in double, then ONE block adds them in a fixed order: deterministic), one backward.
struct SensorDepthArgs {
  const float* depth_pred;  // [N]
  const float* depth_gt;    // [N]
  const float* sdf;         // [N * S]
  const float* starts;      // [N * S]
  const float* dnorm;       // [N] directions_norm, or null (1)
  int32_t n_rays, n_samples, n_blocks;
  float truncation;
  double* partial;          // [n_blocks][6]
  float* losses;            // [3]: l1, free space, sdf
  float* state;             // [4]: free-space weight / (N S), sdf weight / (N S), 1 / (n_valid + 1e-6), unused
  const float* losses_bar;  // [3]
  float* sdf_bar;           // [N * S]
  float* depth_bar;         // [N]
};
struct SensorSample {
  bool front, near;
  float fs, res;  // relu(t - sdf), (z + sdf) - d
};
SDFHIP_D SensorSample sensor_sample(const SensorDepthArgs& a, const int64_t i) {
  const int ray = (int)(i / a.n_samples);
  const float d = a.depth_gt[ray], t = a.truncation;
  const bool valid = d > 0.0f;
  const float z = a.dnorm != nullptr ? a.starts[i] / a.dnorm[ray] : a.starts[i];
  SensorSample s;
  s.front = valid && z < d - t;
  const bool back = valid && z > d + t;
  s.near = valid && !s.front && !back;
  const float x = a.sdf[i];
  s.fs = fmaxf(t - x, 0.0f);
  s.res = (z + x) - d;
  return s;
}
__global__ __launch_bounds__(256) void sensor_depth_partial_kernel(const SensorDepthArgs a) {
  __shared__ double red[6 * 4];
  double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // sum |d - pred| over valid rays, n_valid, n_front, n_near, sum fs^2 front, sum res^2 near
  const int64_t total = (int64_t)a.n_rays * a.n_samples;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const SensorSample s = sensor_sample(a, i);
    if (s.front) {
      v[2] += 1.0;
      v[4] += (double)(s.fs * s.fs);
    }
    if (s.near) {
      v[3] += 1.0;
      v[5] += (double)(s.res * s.res);
    }
  }
  for (int r = blockIdx.x * 256 + threadIdx.x; r < a.n_rays; r += gridDim.x * 256) {
    const float d = a.depth_gt[r];
    if (d > 0.0f) {
      v[0] += (double)fabsf(d - a.depth_pred[r]);
      v[1] += 1.0;
    }
  }
  block_sum_double<6>(v, red);
  if (threadIdx.x < 6) a.partial[(size_t)blockIdx.x * 6 + threadIdx.x] = v[threadIdx.x];
}
__global__ __launch_bounds__(256) void sensor_depth_final_kernel(const SensorDepthArgs a) {
  __shared__ double red[6 * 4];
  double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < a.n_blocks; b += 256)
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] += a.partial[(size_t)b * 6 + k];
  block_sum_double<6>(v, red);
  if (threadIdx.x == 0) {
    const double ns = (double)a.n_rays * (double)a.n_samples;
    // the reference's fp32 scalars: int64 count + 1e-6 -> float32
    const float n = (float)(v[2] + v[3]) + 1e-6f;
    const float fs_w = 1.0f - (float)v[2] / n, sdf_w = 1.0f - (float)v[3] / n;
    const float inv_valid = 1.0f / ((float)v[1] + 1e-6f);
    a.losses[0] = (float)(v[0] * (double)inv_valid);
    a.losses[1] = (float)(v[4] / ns * (double)fs_w);
    a.losses[2] = (float)(v[5] / ns * (double)sdf_w);
    a.state[0] = (float)((double)fs_w / ns);
    a.state[1] = (float)((double)sdf_w / ns);
    a.state[2] = inv_valid;
    a.state[3] = 0.0f;
  }
}
__global__ __launch_bounds__(256) void sensor_depth_bwd_kernel(const SensorDepthArgs a) {
  const float g_l1 = a.losses_bar[0] * a.state[2], g_fs = a.losses_bar[1] * a.state[0], g_sdf = a.losses_bar[2] * a.state[1];
  const int64_t total = (int64_t)a.n_rays * a.n_samples;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const SensorSample s = sensor_sample(a, i);
    float g = 0.0f;
    if (s.front) g -= g_fs * 2.0f * s.fs;  // d relu(t - x)^2 / d x = -2 relu(t - x)
    if (s.near) g += g_sdf * 2.0f * s.res;
    a.sdf_bar[i] = g;
  }
  for (int r = blockIdx.x * 256 + threadIdx.x; r < a.n_rays; r += gridDim.x * 256) {
    const float d = a.depth_gt[r];
    a.depth_bar[r] = d > 0.0f ? g_l1 * depth_sign(a.depth_pred[r] - d) : 0.0f;
  }
}
