// sdfhip - optimiser step over flat buffers: torch.optim.Adam's update (engine/optimizers.py:93-160 builds
// one Adam per parameter group, eps 1e-15, method_configs.py:483-500) as ONE elementwise kernel over a group's contiguous slice
// of the flat parameter / gradient / moment buffers, with the data-parallel mean folded into the gradient read (grad_scale =
// 1 / world_size after a SUM all-reduce) instead of a separate pass over the gradients.
//   m <- m + (1 - b1) (g - m)                      (exp_avg.lerp_(grad, 1 - beta1))
//   v <- b2 v + (1 - b2) g g                       (exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2))
//   p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// weight_decay: torch.optim.Adam's L2 form (g <- g + wd p).  decay_mul: torch.optim.AdamW's DECOUPLED form, p <- p (1 - lr wd) ahead of the
// update (optim/adamw.py _single_tensor_adamw: param.mul_(1 - lr * weight_decay)); 1 = none.  The reference's neuralangelo / bakedangelo
// presets train their fields with AdamW, weight decay 0.01 (configs/method_configs.py:229-232, 156-159).
// Pure streaming: 16 B read + 12 B written per parameter; bound by HBM.
#pragma once
#include "common.h"

struct AdamArgs {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;
  int32_t head;  // leading elements before the first 16-byte boundary (all four slices share the misalignment)
  float lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, weight_decay, grad_scale, decay_mul;
};

SDFHIP_D void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
  g *= a.grad_scale;
  if (a.decay_mul != 1.0f) p *= a.decay_mul;
  if (a.weight_decay != 0.0f) g = fmaf(a.weight_decay, p, g);
  m = m + (1.0f - a.beta1) * (g - m);
  v = a.beta2 * v + (1.0f - a.beta2) * g * g;
  const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
  p = p - a.lr_over_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
  const int64_t body = a.n - a.head;
  const int64_t n4 = body >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  f32x4* p4 = reinterpret_cast<f32x4*>(a.param + a.head);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(a.grad + a.head);
  f32x4* m4 = reinterpret_cast<f32x4*>(a.exp_avg + a.head);
  f32x4* v4 = reinterpret_cast<f32x4*>(a.exp_avg_sq + a.head);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 p = p4[i], m = m4[i], v = v4[i];
    const f32x4 g = g4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = p[e], me = m[e], ve = v[e];
      adam_one(pe, g[e], me, ve, a);
      p[e] = pe;
      m[e] = me;
      v[e] = ve;
    }
    p4[i] = p;
    m4[i] = m;
    v4[i] = v;
  }
  // the < 4 elements in front of the aligned body and the < 4 behind it
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < a.head) adam_one(a.param[gid], a.grad[gid], a.exp_avg[gid], a.exp_avg_sq[gid], a);
  const int64_t t = a.head + (n4 << 2) + gid;
  if (t < a.n) adam_one(a.param[t], a.grad[t], a.exp_avg[t], a.exp_avg_sq[t], a);
}
