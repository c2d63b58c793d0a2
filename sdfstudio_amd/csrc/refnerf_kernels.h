// sdfhip - the ref-nerf colour combination of SDFField.get_colors (fields/sdf_field.py:536-540, 596-607; the bakedsdf / bakedangelo field
// settings, configs/method_configs.py:270-286): with use_diffuse_color the colour network's sigmoid output s is a SPECULAR term,
//     diffuse = sigmoid(W_d feat + b_d - log 3)        (diffuse_color_pred, :333-334; "initialised around 0.25")
//     tint    = sigmoid(W_t feat + b_t)  or 0.5         (specular_tint_pred, use_specular_tint)
//     rgb     = clamp(tint * s + diffuse, 0, 1) * (1 + 2 pad) - pad
// Two 3-row heads on the geometry feature: lane-parallel dot products, one wavefront per point (the lanes stride the feature row:
// coalesced), six wavefront reductions per point.  HBM-bound on the feature rows (4 GF bytes per point each way); nothing for the MFMA pipe.
#pragma once
#include "ray_kernels.h"  // wave_sum

struct RefHeads {
  const float* w_d;  // [3][GF]
  const float* b_d;  // [3]
  const float* w_t;  // [3][GF] or null (no tint: 0.5)
  const float* b_t;  // [3]
};

// the six head pre-activations of point p (every lane returns all of them)
SDFHIP_D void ref_head_dots(const RefHeads& h, const float* __restrict__ feat_row, const int gf, const int lane, float raw_d[3], float raw_t[3]) {
  float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = lane; k < gf; k += 64) {
    const float f = feat_row[k];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a[c] = fmaf(h.w_d[c * gf + k], f, a[c]);
      if (h.w_t != nullptr) a[3 + c] = fmaf(h.w_t[c * gf + k], f, a[3 + c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    raw_d[c] = wave_sum(a[c]) + h.b_d[c];
    raw_t[c] = h.w_t != nullptr ? wave_sum(a[3 + c]) + h.b_t[c] : 0.0f;
  }
}

SDFHIP_D float ref_sigmoid(const float v) { return 1.0f / (1.0f + expf(-v)); }

struct RefCombineArgs {
  RefHeads h;
  const float* s_rgb;  // [P][3] the colour network's sigmoid output (its kernels run with rgb_padding 0 under use_diffuse_color)
  const float* feat;   // [P][GF]
  int64_t n_points;
  int32_t gf, pad_;
  float rgb_padding;
  float* rgb;          // [P][3]
};

// block = 256 threads = 4 wavefronts; a wavefront walks points wave_id, wave_id + n_waves, ...
__global__ __launch_bounds__(256) void refnerf_fwd_kernel(const RefCombineArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < a.n_points; p += n_waves) {
    float raw_d[3], raw_t[3];
    ref_head_dots(a.h, a.feat + p * a.gf, a.gf, lane, raw_d, raw_t);
    if (lane < 3) {
      const float diffuse = ref_sigmoid(raw_d[lane] - 1.09861228866810969140f);  // - log 3
      const float tint = a.h.w_t != nullptr ? ref_sigmoid(raw_t[lane]) : 0.5f;
      const float lin = tint * a.s_rgb[p * 3 + lane] + diffuse;
      a.rgb[p * 3 + lane] = fminf(fmaxf(lin, 0.0f), 1.0f) * (1.0f + 2.0f * a.rgb_padding) - a.rgb_padding;
    }
  }
}

struct RefCombineBwdArgs {
  RefHeads h;
  const float* s_rgb;    // [P][3]
  const float* feat;     // [P][GF]
  const float* rgb_bar;  // [P][3]
  int64_t n_points;
  int32_t gf, pad_;
  float rgb_padding;
  float* s_bar;          // [P][3]  d L / d (the colour network's sigmoid output)
  float* feat_bar;       // [P][GF] d L / d feat through the two heads
  float* delta;          // [P][8]  head pre-activation cotangents: diffuse 0..2, tint 3..5 (the weight-gradient pass reads them)
};

__global__ __launch_bounds__(256) void refnerf_bwd_kernel(const RefCombineBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < a.n_points; p += n_waves) {
    float raw_d[3], raw_t[3];
    ref_head_dots(a.h, a.feat + p * a.gf, a.gf, lane, raw_d, raw_t);
    float dd[3], dt[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float diffuse = ref_sigmoid(raw_d[c] - 1.09861228866810969140f);
      const float tint = a.h.w_t != nullptr ? ref_sigmoid(raw_t[c]) : 0.5f;
      const float s = a.s_rgb[p * 3 + c];
      const float lin = tint * s + diffuse;
      // torch.clamp passes the gradient where min <= x <= max
      const float g = (lin >= 0.0f && lin <= 1.0f) ? a.rgb_bar[p * 3 + c] * (1.0f + 2.0f * a.rgb_padding) : 0.0f;
      dd[c] = g * diffuse * (1.0f - diffuse);
      dt[c] = a.h.w_t != nullptr ? g * s * tint * (1.0f - tint) : 0.0f;
      if (lane == c) a.s_bar[p * 3 + c] = g * tint;
    }
    if (lane < 8) a.delta[p * 8 + lane] = lane < 3 ? dd[lane] : (lane < 6 ? dt[lane - 3] : 0.0f);
    for (int k = lane; k < a.gf; k += 64) {
      float v = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v = fmaf(a.h.w_d[c * a.gf + k], dd[c], v);
        if (a.h.w_t != nullptr) v = fmaf(a.h.w_t[c * a.gf + k], dt[c], v);
      }
      a.feat_bar[p * a.gf + k] = v;
    }
  }
}

// Head weight gradients, pass 1: block b sums delta^T feat over its chunk of points; thread = feature column (gf <= 1024, a multiple of
// 32); partial [n_blocks][6][gf + 1] (column gf: the bias sums).  Pass 2 adds the blocks up in order: no atomics, fixed summation order.
struct RefWgradArgs {
  const float* delta;  // [P][8]
  const float* feat;   // [P][GF]
  int64_t n_points;
  int32_t gf, chunk;
  float* partial;      // [n_blocks][6][gf + 1]
};
__global__ void refnerf_wgrad_kernel(const RefWgradArgs a) {
  const int k = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * a.chunk, p1 = p0 + a.chunk < a.n_points ? p0 + a.chunk : a.n_points;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t p = p0; p < p1; ++p) {
    const float f = k < a.gf ? a.feat[p * a.gf + k] : 0.0f;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float d = a.delta[p * 8 + c];
      acc[c] = fmaf(d, f, acc[c]);
      bacc[c] += d;
    }
  }
  float* out = a.partial + (size_t)blockIdx.x * 6 * (a.gf + 1);
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    if (k < a.gf) out[c * (a.gf + 1) + k] = acc[c];
    if (k == 0) out[c * (a.gf + 1) + a.gf] = bacc[c];
  }
}
__global__ void refnerf_wreduce_kernel(const float* __restrict__ partial, const int n_blocks, const int gf, float* __restrict__ w_d_bar,
                                       float* __restrict__ b_d_bar, float* __restrict__ w_t_bar, float* __restrict__ b_t_bar) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 6 * (gf + 1)) return;
  float s = 0.0f;
  for (int b = 0; b < n_blocks; ++b) s += partial[(size_t)b * 6 * (gf + 1) + idx];
  const int c = idx / (gf + 1), k = idx % (gf + 1);
  if (c < 3) {
    if (k < gf) w_d_bar[c * gf + k] = s;
    else b_d_bar[c] = s;
  } else if (w_t_bar != nullptr) {
    if (k < gf) w_t_bar[(c - 3) * gf + k] = s;
    else b_t_bar[c - 3] = s;
  }
}
