// sdfhip — VolSDF's compositing as one kernel per direction (models/volsdf.py:62-79): Laplace density of the SDF
// (sdf_field.py:48-76) -> weights exp(-cumsum) (cameras/rays.py:146-192) -> rgb with background / expected depth with the
// batch-global clip / normal / accumulation (model_components/renderers.py:81-92,196,245-259,294), plus the transmittance in
// front of the last sample that the background model multiplies its colour with (volsdf.py:67-68).  The density-input sibling of
// neus_render_{fwd,bwd}_kernel: one 64-lane wavefront per ray, C consecutive samples per lane.
#pragma once
#include "ray_kernels.h"

struct VolsdfRenderArgs {
  const float* sdf;     // [N,S]
  const float* grad;    // [N,S,3]
  const float* rgb;     // [N,S,3]
  const float* starts;  // [N,S]
  const float* ends;    // [N,S]
  const float* beta;    // [1]  effective beta = |beta| + beta_min (LaplaceDensity.get_beta)
  const float* bg;      // [3] background colour or null (black)
  int32_t N, S;
  float* density;       // [N,S]
  float* weights;       // [N,S]
  float* out_rgb;       // [N,3]
  float* out_depth;     // [N] unclipped expected depth
  float* out_normal;    // [N,3]
  float* out_acc;       // [N]
  float* bg_trans;      // [N]  transmittance in front of the last sample
  float* steps_minmax;  // [2]
  // backward
  const float* rgbbar;       // [N,3] or null
  const float* depthbar;     // [N] or null (w.r.t. the CLIPPED depth)
  const float* normalbar;    // [N,3] or null
  const float* accbar;       // [N] or null
  const float* weightsbar;   // [N,S] or null
  const float* bgtransbar;   // [N] or null
  float* sdfbar;    // [N,S]
  float* gradbar;   // [N,S,3]
  float* rgbsbar;   // [N,S,3]
  float* betabar;   // [1] accumulated
};

// density: laplace_density_f (ray_kernels.h; sdf_field.py:62-76: alpha (0.5 + 0.5 sign(sdf) expm1(-|sdf| / beta)), alpha = 1 / beta)

template <int C>
__global__ __launch_bounds__(256) void volsdf_render_fwd_kernel(const VolsdfRenderArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int S = a.S;
  const float beta = a.beta[0];
  float dd[C];
  float local = 0.0f, last_dd = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    dd[c] = 0.0f;
    if (s < S) {
      const int64_t i = (int64_t)ray * S + s;
      const float sigma = laplace_density_f(a.sdf[i], beta);
      a.density[i] = sigma;
      dd[c] = (a.ends[i] - a.starts[i]) * sigma;
      if (s == S - 1) last_dd = dd[c];
    }
    local += dd[c];
  }
  const float incl = wave_incl_scan_add(local, lane);
  float cum = incl - local;  // exclusive
  float acc = 0.f, r = 0.f, g = 0.f, b = 0.f, dep = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
  float mn = 3.0e38f, mx = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    if (s < S) {
      const int64_t i = (int64_t)ray * S + s;
      const float w = (1.0f - expf(-dd[c])) * expf(-cum);
      cum += dd[c];
      a.weights[i] = w;
      acc += w;
      r = fmaf(w, a.rgb[i * 3], r);
      g = fmaf(w, a.rgb[i * 3 + 1], g);
      b = fmaf(w, a.rgb[i * 3 + 2], b);
      const float mid = 0.5f * (a.starts[i] + a.ends[i]);
      dep = fmaf(w, mid, dep);
      mn = fminf(mn, mid);
      mx = fmaxf(mx, mid);
      const float gx = a.grad[i * 3], gy = a.grad[i * 3 + 1], gz = a.grad[i * 3 + 2];
      const float inv = 1.0f / fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);  // F.normalize
      nx = fmaf(w, gx * inv, nx);
      ny = fmaf(w, gy * inv, ny);
      nz = fmaf(w, gz * inv, nz);
    }
  }
  acc = wave_sum(acc);
  r = wave_sum(r);
  g = wave_sum(g);
  b = wave_sum(b);
  dep = wave_sum(dep);
  nx = wave_sum(nx);
  ny = wave_sum(ny);
  nz = wave_sum(nz);
  last_dd = wave_sum(last_dd);
  const float total = __shfl(incl, 63);
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    mn = fminf(mn, __shfl_xor(mn, m));
    mx = fmaxf(mx, __shfl_xor(mx, m));
  }
  if (lane == 0) {
    float bgr = 0.f, bgg = 0.f, bgb = 0.f;
    if (a.bg != nullptr) {
      bgr = a.bg[0];
      bgg = a.bg[1];
      bgb = a.bg[2];
    }
    a.out_rgb[ray * 3 + 0] = r + bgr * (1.0f - acc);
    a.out_rgb[ray * 3 + 1] = g + bgg * (1.0f - acc);
    a.out_rgb[ray * 3 + 2] = b + bgb * (1.0f - acc);
    a.out_depth[ray] = dep / (acc + 1e-10f);
    a.out_normal[ray * 3 + 0] = nx;
    a.out_normal[ray * 3 + 1] = ny;
    a.out_normal[ray * 3 + 2] = nz;
    a.out_acc[ray] = acc;
    a.bg_trans[ray] = expf(-(total - last_dd));
    atomic_min_f(a.steps_minmax, mn);
    atomic_max_f(a.steps_minmax + 1, mx);
  }
}

template <int C>
__global__ __launch_bounds__(256) void volsdf_render_bwd_kernel(const VolsdfRenderArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int S = a.S;
  const float beta = a.beta[0];
  float rb[3] = {0.f, 0.f, 0.f}, nb[3] = {0.f, 0.f, 0.f}, db = 0.f, ab = 0.f, bg[3] = {0.f, 0.f, 0.f}, tb = 0.f;
  if (a.rgbbar != nullptr) {
    rb[0] = a.rgbbar[ray * 3];
    rb[1] = a.rgbbar[ray * 3 + 1];
    rb[2] = a.rgbbar[ray * 3 + 2];
  }
  if (a.normalbar != nullptr) {
    nb[0] = a.normalbar[ray * 3];
    nb[1] = a.normalbar[ray * 3 + 1];
    nb[2] = a.normalbar[ray * 3 + 2];
  }
  if (a.accbar != nullptr) ab = a.accbar[ray];
  if (a.bg != nullptr) {
    bg[0] = a.bg[0];
    bg[1] = a.bg[1];
    bg[2] = a.bg[2];
  }
  const float acc = a.out_acc[ray];
  const float depth_raw = a.out_depth[ray];
  if (a.depthbar != nullptr) {
    const bool pass = depth_raw >= a.steps_minmax[0] && depth_raw <= a.steps_minmax[1];
    db = pass ? a.depthbar[ray] : 0.0f;
  }
  if (a.bgtransbar != nullptr) tb = a.bgtransbar[ray] * a.bg_trans[ray];  // d L / d (sum of dd in front of the last sample), negated below
  float dd[C], wbar[C], wv[C], delta[C];
  float local = 0.0f, lsum = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    dd[c] = 0.f;
    wbar[c] = 0.f;
    wv[c] = 0.f;
    delta[c] = 0.f;
    if (s < S) {
      const int64_t i = (int64_t)ray * S + s;
      delta[c] = a.ends[i] - a.starts[i];
      dd[c] = delta[c] * a.density[i];
      const float w = a.weights[i];
      wv[c] = w;
      const float mid = 0.5f * (a.starts[i] + a.ends[i]);
      const float gx = a.grad[i * 3], gy = a.grad[i * 3 + 1], gz = a.grad[i * 3 + 2];
      const float nrm = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);
      float t = ab + (a.weightsbar != nullptr ? a.weightsbar[i] : 0.0f);
      t += rb[0] * (a.rgb[i * 3] - bg[0]) + rb[1] * (a.rgb[i * 3 + 1] - bg[1]) + rb[2] * (a.rgb[i * 3 + 2] - bg[2]);
      t += db * (mid - depth_raw) / (acc + 1e-10f);
      t += (nb[0] * gx + nb[1] * gy + nb[2] * gz) / nrm;
      wbar[c] = t;
      a.rgbsbar[i * 3 + 0] = w * rb[0];
      a.rgbsbar[i * 3 + 1] = w * rb[1];
      a.rgbsbar[i * 3 + 2] = w * rb[2];
      lsum += t * w;
      // normal render backward: n = g / |g|
      const float n0 = gx / nrm, n1 = gy / nrm, n2 = gz / nrm;
      const float q0 = w * nb[0], q1 = w * nb[1], q2 = w * nb[2];
      const float dotn = n0 * q0 + n1 * q1 + n2 * q2;
      a.gradbar[i * 3 + 0] = (q0 - n0 * dotn) / nrm;
      a.gradbar[i * 3 + 1] = (q1 - n1 * dotn) / nrm;
      a.gradbar[i * 3 + 2] = (q2 - n2 * dotn) / nrm;
    }
    local += dd[c];
  }
  const float incl = wave_incl_scan_add(local, lane);
  const float incl2 = wave_incl_scan_add(lsum, lane);
  float suffix = __shfl(incl2, 63) - incl2;  // sum over the samples of later lanes of wbar_j w_j
  float cum = incl;                           // cumulative INCLUDING this lane's samples; walk backwards
  float bb = 0.0f;
#pragma unroll
  for (int cc = 0; cc < C; ++cc) {
    const int c = C - 1 - cc;
    const int s = lane * C + c;
    cum -= dd[c];  // exclusive cumulative for sample c
    if (s < S) {
      const int64_t i = (int64_t)ray * S + s;
      // w_i = (1 - e^{-dd_i}) e^{-cum_i}: d w_i / d dd_i = e^{-dd_i} e^{-cum_i}; d w_j / d dd_i = -w_j (j > i);
      // bg_trans = exp(-sum_{j < S-1} dd_j): d / d dd_i = -bg_trans for i < S - 1
      float g = wbar[c] * expf(-dd[c]) * expf(-cum) - suffix;
      if (s < S - 1) g -= tb;
      const float sb = g * delta[c];  // d L / d sigma_i
      const float x = a.sdf[i];
      const float sg = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
      const float e = expf(-fabsf(x) / beta);
      const float sigma = a.density[i];
      a.sdfbar[i] = sb * (-(sg * sg) * e / (2.0f * beta * beta));
      bb += sb * (-sigma / beta + 0.5f * sg * e * fabsf(x) / (beta * beta * beta));
    }
    suffix += wbar[c] * wv[c];
  }
  bb = wave_sum(bb);
  if (lane == 0 && a.betabar != nullptr) atomicAdd(a.betabar, bb);
}
