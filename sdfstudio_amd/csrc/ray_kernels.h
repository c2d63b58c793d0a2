// sdfhip — per-ray kernels: one 64-lane wavefront per ray, C = ceil(S/64) consecutive samples per lane,
// wave-level scans with DPP/shuffle (no LDS round trip), HBM-bound.
//   neus_render_fwd/bwd     NeuS alpha (sdf_field.py:476-525) -> weights (rays.py:194-230) -> rgb / depth /
//                           normal / accumulation (renderers.py:81-92,196,245-259,294), fused
//   density_weights_fwd/bwd proposal weights from density (rays.py:146-167)
//   spaced_bins_kernel      UniformLinDispPiecewiseSampler bins (ray_samplers.py:101-117, 240-241)
//   pdf_sample_kernel       PDFSampler, include_original = False (ray_samplers.py:303-358)
#pragma once
#include "common.h"

constexpr int kMaxPerLane = 16;  // S <= 1024

SDFHIP_D float wave_incl_scan_add(float v, const int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
SDFHIP_D float wave_incl_scan_mul(float v, const int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = __shfl_up(v, d);
    if (lane >= d) v *= t;
  }
  return v;
}
SDFHIP_D float wave_sum(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
SDFHIP_D float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

SDFHIP_D void atomic_min_f(float* addr, float v) {  // valid for non-negative floats (ray distances)
  atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
SDFHIP_D void atomic_max_f(float* addr, float v) {
  atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

struct NeusRenderArgs {
  const float* sdf;      // [N,S]
  const float* grad;     // [N,S,3]
  const float* rgb;      // [N,S,3]
  const float* dirs;     // [N,3]
  const float* starts;   // [N,S]
  const float* ends;     // [N,S]
  const float* variance; // [1]  deviation_network.variance
  const float* bg;       // [3] background colour or null (black)
  float cos_anneal;
  int32_t N, S;
  // forward outputs
  float* alpha;    // [N,S]
  float* weights;  // [N,S]
  float* out_rgb;  // [N,3]
  float* out_depth;   // [N] unclipped expected depth
  float* out_normal;  // [N,3]
  float* out_acc;     // [N]
  float* steps_minmax;  // [2] global min / max of sample mid points (pre-initialised to +inf / 0)
  // backward inputs
  const float* rgbbar;     // [N,3] or null
  const float* depthbar;   // [N] or null  (w.r.t. the CLIPPED depth)
  const float* normalbar;  // [N,3] or null
  const float* accbar;     // [N] or null
  const float* weightsbar; // [N,S] or null
  // backward outputs
  float* sdfbar;   // [N,S]
  float* gradbar;  // [N,S,3]
  float* rgbsbar;  // [N,S,3]
  float* variancebar;  // [1] accumulated
  // background merge of NeuS-facto (neus_facto.py:289-290 -> base_surface_model.py:266-290; null: no background model): samples whose
  // START position o + d t lies outside the unit sphere take alpha = 1 - exp(-delta sigma_bg) (rays.py:131-144) and colour rgb_bg of
  // the background field; the normal stays the SDF field's (field_outputs[NORMAL] is not merged)
  const float* origins;      // [N,3]
  const float* bg_density;   // [N,S]
  const float* bg_rgb;       // [N,S,3]
  float* bg_density_bar;     // [N,S]
  float* bg_rgb_bar;         // [N,S,3]
  float* rgb_merged;         // [N,S,3] forward: the merged per-sample colour (field_outputs[RGB] of the reference), or null
};
SDFHIP_D bool neus_inside(const NeusRenderArgs& a, const int ray, const float dx, const float dy, const float dz, const float t) {
  const float px = a.origins[ray * 3 + 0] + dx * t, py = a.origins[ray * 3 + 1] + dy * t, pz = a.origins[ray * 3 + 2] + dz * t;
  return sqrtf(px * px + py * py + pz * pz) < 1.0f;  // get_foreground_mask, base_surface_model.py:256-264
}

SDFHIP_D float neus_inv_s(const float variance) {
  return fminf(fmaxf(expf(variance * 10.0f), 1e-6f), 1e6f);  // sdf_field.py:116-118
}

// grid = ceil(N/4), block = 256 (4 rays)
template <int C>
__global__ __launch_bounds__(256) void neus_render_fwd_kernel(const NeusRenderArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int S = a.S;
  const float inv_s = neus_inv_s(a.variance[0]);
  const float dx = a.dirs[ray * 3 + 0], dy = a.dirs[ray * 3 + 1], dz = a.dirs[ray * 3 + 2];
  float al[C], one_m[C];
  float local = 1.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    al[c] = 0.0f;
    one_m[c] = 1.0f;
    if (s < S) {
      const int64_t i = (int64_t)ray * S + s;
      const float sd = a.sdf[i];
      const float tc = dx * a.grad[i * 3] + dy * a.grad[i * 3 + 1] + dz * a.grad[i * 3 + 2];
      const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.0f) * (1.0f - a.cos_anneal) + fmaxf(-tc, 0.0f) * a.cos_anneal);
      const float delta = a.ends[i] - a.starts[i];
      const float pc = sigmoidf_((sd - ic * delta * 0.5f) * inv_s);
      const float nc = sigmoidf_((sd + ic * delta * 0.5f) * inv_s);
      const float v = (pc - nc + 1e-5f) / (pc + 1e-5f);
      al[c] = fminf(fmaxf(v, 0.0f), 1.0f);
      if (a.bg_density != nullptr && !neus_inside(a, ray, dx, dy, dz, a.starts[i])) al[c] = 1.0f - expf(-(delta * a.bg_density[i]));
      one_m[c] = 1.0f - al[c] + 1e-7f;  // rays.py:205
      a.alpha[i] = al[c];
    }
    local *= one_m[c];
  }
  const float incl = wave_incl_scan_mul(local, lane);
  float T = __shfl_up(incl, 1);
  if (lane == 0) T = 1.0f;
  float acc = 0.f, r = 0.f, g = 0.f, b = 0.f, dep = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
  float mn = 3.0e38f, mx = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    if (s < S) {
      const int64_t i = (int64_t)ray * S + s;
      const float w = al[c] * T;
      a.weights[i] = w;
      acc += w;
      const float* col = (a.bg_density != nullptr && !neus_inside(a, ray, dx, dy, dz, a.starts[i])) ? a.bg_rgb : a.rgb;
      r = fmaf(w, col[i * 3], r);
      g = fmaf(w, col[i * 3 + 1], g);
      b = fmaf(w, col[i * 3 + 2], b);
      if (a.rgb_merged != nullptr) {
        a.rgb_merged[i * 3 + 0] = col[i * 3];
        a.rgb_merged[i * 3 + 1] = col[i * 3 + 1];
        a.rgb_merged[i * 3 + 2] = col[i * 3 + 2];
      }
      const float mid = 0.5f * (a.starts[i] + a.ends[i]);
      dep = fmaf(w, mid, dep);
      mn = fminf(mn, mid);
      mx = fmaxf(mx, mid);
      const float gx = a.grad[i * 3], gy = a.grad[i * 3 + 1], gz = a.grad[i * 3 + 2];
      const float inv = 1.0f / fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);  // F.normalize
      nx = fmaf(w, gx * inv, nx);
      ny = fmaf(w, gy * inv, ny);
      nz = fmaf(w, gz * inv, nz);
      T *= one_m[c];
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
    atomic_min_f(a.steps_minmax, mn);
    atomic_max_f(a.steps_minmax + 1, mx);
  }
}

// depth = clip(depth, steps.min(), steps.max())  (renderers.py:257 — a cross-ray reduction)
__global__ void depth_clip_kernel(const float* __restrict__ raw, const float* __restrict__ minmax, const int n,
                                  float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fminf(fmaxf(raw[i], minmax[0]), minmax[1]);
}

template <int C>
__global__ __launch_bounds__(256) void neus_render_bwd_kernel(const NeusRenderArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int S = a.S;
  const float var = a.variance[0];
  const float inv_s_raw = expf(var * 10.0f);
  const float inv_s = fminf(fmaxf(inv_s_raw, 1e-6f), 1e6f);
  const float dx = a.dirs[ray * 3 + 0], dy = a.dirs[ray * 3 + 1], dz = a.dirs[ray * 3 + 2];
  float rb[3] = {0.f, 0.f, 0.f}, nb[3] = {0.f, 0.f, 0.f}, db = 0.f, ab = 0.f, bg[3] = {0.f, 0.f, 0.f};
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
  // pass 1: wbar_i, the transmittance T_i, and the suffix sum of wbar_j w_j
  float wbar[C], wv[C], al[C], Tv[C];
  float local = 0.0f, lprod = 1.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    wbar[c] = 0.f;
    wv[c] = 0.f;
    al[c] = 0.f;
    if (s < S) {
      const int64_t i = (int64_t)ray * S + s;
      const float w = a.weights[i];
      wv[c] = w;
      al[c] = a.alpha[i];
      lprod *= 1.0f - al[c] + 1e-7f;
      const float mid = 0.5f * (a.starts[i] + a.ends[i]);
      const float gx = a.grad[i * 3], gy = a.grad[i * 3 + 1], gz = a.grad[i * 3 + 2];
      const float nrm = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);
      const float n0 = gx / nrm, n1 = gy / nrm, n2 = gz / nrm;
      float t = ab + (a.weightsbar != nullptr ? a.weightsbar[i] : 0.0f);
      const bool outside = a.bg_density != nullptr && !neus_inside(a, ray, dx, dy, dz, a.starts[i]);
      const float* col = outside ? a.bg_rgb : a.rgb;
      t += rb[0] * (col[i * 3] - bg[0]) + rb[1] * (col[i * 3 + 1] - bg[1]) + rb[2] * (col[i * 3 + 2] - bg[2]);
      t += db * (mid - depth_raw) / (acc + 1e-10f);
      t += nb[0] * n0 + nb[1] * n1 + nb[2] * n2;
      wbar[c] = t;
      // per-sample outputs that do not need the scan: the colour cotangent goes to whichever field supplied the colour
      a.rgbsbar[i * 3 + 0] = outside ? 0.0f : w * rb[0];
      a.rgbsbar[i * 3 + 1] = outside ? 0.0f : w * rb[1];
      a.rgbsbar[i * 3 + 2] = outside ? 0.0f : w * rb[2];
      if (a.bg_rgb_bar != nullptr) {
        a.bg_rgb_bar[i * 3 + 0] = outside ? w * rb[0] : 0.0f;
        a.bg_rgb_bar[i * 3 + 1] = outside ? w * rb[1] : 0.0f;
        a.bg_rgb_bar[i * 3 + 2] = outside ? w * rb[2] : 0.0f;
      }
      local += t * w;
    }
  }
  {
    const float pincl = wave_incl_scan_mul(lprod, lane);
    float T = __shfl_up(pincl, 1);
    if (lane == 0) T = 1.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      Tv[c] = T;
      T *= 1.0f - al[c] + 1e-7f;
    }
  }
  const float incl = wave_incl_scan_add(local, lane);
  const float total = __shfl(incl, 63);
  float suffix = total - incl;  // sum over lanes > this lane
  float vbar_acc = 0.0f;
  // walk this lane's samples from last to first so `suffix` = sum_{j>i} wbar_j w_j
#pragma unroll
  for (int cc = 0; cc < C; ++cc) {
    const int c = C - 1 - cc;
    const int s = lane * C + c;
    if (s < S) {
      const int64_t i = (int64_t)ray * S + s;
      // d w_i / d alpha_i = T_i ; d w_j / d alpha_i = -w_j / (1 - alpha_i + 1e-7)  for j > i   (rays.py:204-208)
      const float one_m = 1.0f - al[c] + 1e-7f;
      const float abar_i = wbar[c] * Tv[c] - suffix / one_m;
      suffix += wbar[c] * wv[c];
      // ---- alpha backward (sdf_field.py:494-516)
      const float sd = a.sdf[i];
      const float gx = a.grad[i * 3], gy = a.grad[i * 3 + 1], gz = a.grad[i * 3 + 2];
      const float tc = dx * gx + dy * gy + dz * gz;
      const float ca = a.cos_anneal;
      const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.0f) * (1.0f - ca) + fmaxf(-tc, 0.0f) * ca);
      const float delta = a.ends[i] - a.starts[i];
      const float ep = sd - ic * delta * 0.5f, en = sd + ic * delta * 0.5f;
      const float pc = sigmoidf_(ep * inv_s), nc = sigmoidf_(en * inv_s);
      const float den = pc + 1e-5f;
      const float v = (pc - nc + 1e-5f) / den;
      const bool outside = a.bg_density != nullptr && !neus_inside(a, ray, dx, dy, dz, a.starts[i]);
      if (a.bg_density_bar != nullptr) a.bg_density_bar[i] = outside ? abar_i * delta * expf(-(delta * a.bg_density[i])) : 0.0f;
      const bool pass = v >= 0.0f && v <= 1.0f && !outside;  // torch.clip backward mask; outside samples take the background's alpha
      float sdb = 0.f, icb = 0.f;
      if (pass && abar_i != 0.0f) {
        const float pbar = abar_i / den;              // d/d(p) with p = pc - nc
        const float cbar = -abar_i * v / den;         // d/d(c) with c = pc
        const float pcb = pbar + cbar, ncb = -pbar;
        const float epb = pcb * pc * (1.0f - pc) * inv_s;
        const float enb = ncb * nc * (1.0f - nc) * inv_s;
        vbar_acc += pcb * pc * (1.0f - pc) * ep + ncb * nc * (1.0f - nc) * en;  // d/d inv_s
        sdb = epb + enb;
        icb = (enb - epb) * delta * 0.5f;
      }
      // d iter_cos / d true_cos
      const float dic = 0.5f * (1.0f - ca) * ((-tc * 0.5f + 0.5f) > 0.0f ? 1.0f : 0.0f) + ca * ((-tc) > 0.0f ? 1.0f : 0.0f);
      const float tcb = icb * dic;
      // normal render backward: n = g / |g|
      const float nrm = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);
      const float n0 = gx / nrm, n1 = gy / nrm, n2 = gz / nrm;
      const float w = wv[c];
      const float q0 = w * nb[0], q1 = w * nb[1], q2 = w * nb[2];
      const float dotn = n0 * q0 + n1 * q1 + n2 * q2;
      a.sdfbar[i] = sdb;
      a.gradbar[i * 3 + 0] = tcb * dx + (q0 - n0 * dotn) / nrm;
      a.gradbar[i * 3 + 1] = tcb * dy + (q1 - n1 * dotn) / nrm;
      a.gradbar[i * 3 + 2] = tcb * dz + (q2 - n2 * dotn) / nrm;
    }
  }
  vbar_acc = wave_sum(vbar_acc);
  if (lane == 0 && a.variancebar != nullptr) {
    const bool pass = inv_s_raw >= 1e-6f && inv_s_raw <= 1e6f;
    if (pass) atomicAdd(a.variancebar, vbar_acc * 10.0f * inv_s_raw);
  }
}

// ------------------------------------------------------------------------------------------------ density weights
struct DensityWeightsArgs {
  const float* density;  // [N,S]
  const float* starts;
  const float* ends;
  int32_t N, S;
  float* weights;           // forward out
  const float* weightsbar;  // backward in
  float* densitybar;        // backward out
};

template <int C>
__global__ __launch_bounds__(256) void density_weights_fwd_kernel(const DensityWeightsArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  float dd[C];
  float local = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    dd[c] = 0.0f;
    if (s < a.S) {
      const int64_t i = (int64_t)ray * a.S + s;
      dd[c] = (a.ends[i] - a.starts[i]) * a.density[i];
    }
    local += dd[c];
  }
  const float incl = wave_incl_scan_add(local, lane);
  float cum = incl - local;  // exclusive
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    if (s < a.S) {
      const int64_t i = (int64_t)ray * a.S + s;
      a.weights[i] = (1.0f - expf(-dd[c])) * expf(-cum);
      cum += dd[c];
    }
  }
}

template <int C>
__global__ __launch_bounds__(256) void density_weights_bwd_kernel(const DensityWeightsArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  float dd[C], wb[C], delta[C];
  float local = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    dd[c] = 0.0f;
    wb[c] = 0.0f;
    delta[c] = 0.0f;
    if (s < a.S) {
      const int64_t i = (int64_t)ray * a.S + s;
      delta[c] = a.ends[i] - a.starts[i];
      dd[c] = delta[c] * a.density[i];
      wb[c] = a.weightsbar[i];
    }
    local += dd[c];
  }
  const float incl = wave_incl_scan_add(local, lane);
  float cum = incl - local;
  float ww[C];
  float lsum = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float T = expf(-cum);
    ww[c] = (1.0f - expf(-dd[c])) * T;
    lsum += wb[c] * ww[c];
    cum += dd[c];
  }
  const float incl2 = wave_incl_scan_add(lsum, lane);
  const float total = __shfl(incl2, 63);
  float suffix = total - incl2;
  cum = incl;  // cumulative INCLUDING this lane's samples; walk backwards
#pragma unroll
  for (int cc = 0; cc < C; ++cc) {
    const int c = C - 1 - cc;
    const int s = lane * C + c;
    cum -= dd[c];  // exclusive cumulative for sample c
    if (s < a.S) {
      const int64_t i = (int64_t)ray * a.S + s;
      // w_i = (1 - e^{-dd_i}) e^{-cum_i}; d w_i / d dd_i = e^{-dd_i} e^{-cum_i}; d w_j / d dd_i = -w_j (j > i)
      const float g = wb[c] * expf(-dd[c]) * expf(-cum) - suffix;
      a.densitybar[i] = g * delta[c];
    }
    suffix += wb[c] * ww[c];
  }
}

// ------------------------------------------------------------------------------------------------ samplers
SDFHIP_D float piecewise_fn(const float x) { return x < 1.0f ? x * 0.5f : 1.0f - 1.0f / (2.0f * x); }
SDFHIP_D float piecewise_inv(const float x) { return x < 0.5f ? 2.0f * x : 1.0f / (2.0f - 2.0f * x); }
// spacing_fn / spacing_fn_inv of the SpacedSampler subclasses (ray_samplers.py:130-247)
enum { SP_PIECEWISE = 0, SP_UNIFORM = 1, SP_LINDISP = 2, SP_SQRT = 3, SP_LOG = 4 };
SDFHIP_D float spacing_fn(const int kind, const float x) {
  switch (kind) {
    case SP_UNIFORM: return x;
    case SP_LINDISP: return 1.0f / x;
    case SP_SQRT: return sqrtf(x);
    case SP_LOG: return logf(x);
    default: return piecewise_fn(x);
  }
}
SDFHIP_D float spacing_inv(const int kind, const float x) {
  switch (kind) {
    case SP_UNIFORM: return x;
    case SP_LINDISP: return 1.0f / x;
    case SP_SQRT: return x * x;
    case SP_LOG: return expf(x);
    default: return piecewise_inv(x);
  }
}

struct BinsArgs {
  const float* nears;  // [N]
  const float* fars;   // [N]
  const float* jitter; // [N] single-jitter draw in [0,1), [N,S+1] per-sample draws (jitter_stride = S+1), or null (deterministic)
  int32_t jitter_stride;
  int32_t N, S;        // S samples -> S+1 bins
  int32_t uniform;     // SP_*: 0 UniformLinDispPiecewiseSampler (ray_samplers.py:240-241), 1 UniformSampler (:130-151), 2 LinearDisparity
                       // (:154-175), 3 Sqrt (:178-198), 4 Log (:201-218)
  float* bins;         // [N,S+1] spacing-domain bins
  float* starts;       // [N,S] euclidean
  float* ends;         // [N,S]
};

// thread per (ray, bin edge)
__global__ void spaced_bins_kernel(const BinsArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = a.S + 1;
  if (idx >= (int64_t)a.N * nb) return;
  const int ray = (int)(idx / nb), j = (int)(idx % nb);
  // torch.linspace(0, 1, S+1) (ray_samplers.py:101): symmetric formula used by ATen
  const float step = 1.0f / (float)a.S;
  auto lin = [&](int k) { return k < nb / 2 ? step * (float)k : 1.0f - step * (float)(a.S - k); };
  float b = lin(j);
  if (a.jitter != nullptr) {
    const float lo = j == 0 ? lin(0) : (lin(j) + lin(j - 1)) * 0.5f;
    const float hi = j == a.S ? lin(a.S) : (lin(j + 1) + lin(j)) * 0.5f;
    b = lo + (hi - lo) * a.jitter[a.jitter_stride ? (int64_t)ray * a.jitter_stride + j : ray];
  }
  a.bins[idx] = b;
  float e;
  if (a.uniform == SP_UNIFORM) {
    e = b * a.fars[ray] + (1.0f - b) * a.nears[ray];
  } else {
    const float sn = spacing_fn(a.uniform, a.nears[ray]), sf = spacing_fn(a.uniform, a.fars[ray]);
    e = spacing_inv(a.uniform, b * sf + (1.0f - b) * sn);
  }
  if (j < a.S) a.starts[(int64_t)ray * a.S + j] = e;
  if (j > 0) a.ends[(int64_t)ray * a.S + j - 1] = e;
}

struct PdfArgs {
  const float* weights;   // [N,S_in]
  const float* bins_in;   // [N,S_in+1]
  const float* nears;
  const float* fars;
  const float* jitter;    // [N], [N,S_out+1] (jitter_stride = S_out+1) or null
  int32_t jitter_stride, uniform;  // uniform: the SP_* spacing of the bins (0 piecewise, 1 uniform: x far + (1 - x) near, 2 .. 4)
  int32_t N, S_in, S_out;
  float anneal, histogram_padding, eps;
  float u_end;      // float(1 - 1/(S_out+1))            (ray_samplers.py:323)
  float u_center;   // float(1 / (2 (S_out+1)))          (ray_samplers.py:333)
  float* bins_out;        // [N,S_out+1]
  float* starts;          // [N,S_out]
  float* ends;            // [N,S_out]
};

// one wave per ray; cdf staged in LDS.  S_in <= 64*C
template <int C>
__global__ __launch_bounds__(256) void pdf_sample_kernel(const PdfArgs a) {
  __shared__ float cdf_s[4][64 * C + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  if (ray >= a.N) return;
  float* cdf = cdf_s[wv];
  const int Si = a.S_in;
  float w[C];
  float local = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    w[c] = 0.0f;
    if (s < Si) {
      const float wv = a.weights[(int64_t)ray * Si + s];
      w[c] = (a.anneal == 1.0f ? wv : powf(wv, a.anneal)) + a.histogram_padding;
    }
    local += w[c];
  }
  float wsum = wave_sum(local);
  const float padding = fmaxf(a.eps - wsum, 0.0f);
  const float add = padding / (float)Si;
  wsum += padding;
  local = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    if (s < Si) {
      w[c] = (w[c] + add) / wsum;
      local += w[c];
    } else {
      w[c] = 0.0f;
    }
  }
  const float incl = wave_incl_scan_add(local, lane);
  float cum = incl - local;
  if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int s = lane * C + c;
    if (s < Si) {
      cum += w[c];
      cdf[s + 1] = fminf(1.0f, cum);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave are done (single-wave producer/consumer)
  __builtin_amdgcn_wave_barrier();
  const int nbins = a.S_out + 1;
  const float near_ = a.nears[ray], far_ = a.fars[ray];
  const float sn = spacing_fn(a.uniform, near_), sf = spacing_fn(a.uniform, far_);
  const float* bin = a.bins_in + (int64_t)ray * (Si + 1);
  for (int j = lane; j < nbins; j += 64) {
    // u = linspace(0, 1 - 1/nbins, nbins)[j] + (jitter / nbins | 1/(2 nbins))      (ray_samplers.py:321-334)
    const float end = a.u_end;
    const float step = end / (float)(nbins - 1);
    float u = j < nbins / 2 ? step * (float)j : end - step * (float)(nbins - 1 - j);
    u += a.jitter != nullptr ? a.jitter[a.jitter_stride ? (int64_t)ray * a.jitter_stride + j : ray] / (float)nbins : a.u_center;
    // searchsorted(cdf, u, side="right"): first index with cdf[idx] > u
    int lo = 0, hi = Si + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    const int below = min(max(lo - 1, 0), Si), above = min(max(lo, 0), Si);
    const float c0 = cdf[below], c1 = cdf[above];
    const float b0 = bin[below], b1 = bin[above];
    float t = (u - c0) / (c1 - c0);
    if (t != t) t = 0.0f;            // nan_to_num(nan=0)
    t = fminf(fmaxf(t, 0.0f), 1.0f); // +-inf clip like torch.clip after nan_to_num
    const float b = b0 + t * (b1 - b0);
    a.bins_out[(int64_t)ray * nbins + j] = b;
    const float e = a.uniform == SP_UNIFORM ? b * far_ + (1.0f - b) * near_ : spacing_inv(a.uniform, b * sf + (1.0f - b) * sn);
    if (j < a.S_out) a.starts[(int64_t)ray * a.S_out + j] = e;
    if (j > 0) a.ends[(int64_t)ray * a.S_out + j - 1] = e;
  }
}

// ------------------------------------------------------------------------------------------------ NeuS hierarchical sampler
// One up-sampling step of NeuSSampler (ray_samplers.py:851-886): merge the sdf of the samples added by the previous step
// (:864-868), alpha with a fixed inverse variance (:899-944), weights (rays.py:194-208), PDFSampler with
// histogram_padding 1e-5 / include_original = False (:303-358), merge_ray_samples (:757-786).  One wavefront per ray, the
// ray's working set (<= 512 samples) staged in LDS; UniformSampler spacing (euclid = x far + (1 - x) near).
struct NeusUpArgs {
  const float* bins_in;   // [N,S+1]
  const float* sdf_a;     // [N,Sa]
  const float* sdf_b;     // [N,Sb] or null
  const int32_t* index;   // [N,S] into cat(sdf_a, sdf_b) or null (identity)
  const float* nears;
  const float* fars;
  const float* jitter;    // [N], [N,n_new+1] (jitter_stride = n_new+1) or null
  int32_t jitter_stride;
  int32_t N, Sa, Sb, n_new;
  float inv_s, histogram_padding, eps, u_end, u_center;
  float* sdf_merged;      // [N,S]
  float* new_bins;        // [N,n_new+1]
  float* new_starts;      // [N,n_new]  euclidean starts of the new samples (where the field is evaluated next)
  float* new_ends;        // [N,n_new]
  float* merged_bins;     // [N,S+n_new+1]
  int32_t* merged_index;  // [N,S+n_new]
  float* merged_starts;   // [N,S+n_new]
  float* merged_ends;     // [N,S+n_new]
};

constexpr int kNeusUpMaxNew = 64;
template <int C>
__global__ __launch_bounds__(256) void neus_upsample_kernel(const NeusUpArgs a) {
  __shared__ float sdf_s[4][64 * C], bin_s[4][64 * C + 1], cdf_s[4][64 * C + 1], new_s[4][kNeusUpMaxNew + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  if (ray >= a.N) return;
  const int S = a.Sa + a.Sb, Sn = a.n_new;
  float* sdf = sdf_s[wv];
  float* bin = bin_s[wv];
  float* cdf = cdf_s[wv];
  float* nb = new_s[wv];
  const float near = a.nears[ray], far = a.fars[ray];
  auto euclid = [&](const float x) { return x * far + (1.0f - x) * near; };
  auto lds_sync = [&]() {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS traffic is complete (wave-private arrays)
    __builtin_amdgcn_wave_barrier();
  };
  for (int s = lane; s <= S; s += 64) {
    bin[s] = a.bins_in[(int64_t)ray * (S + 1) + s];
    if (s < S) {
      const int src = a.index != nullptr ? a.index[(int64_t)ray * S + s] : s;
      const float v = src < a.Sa ? a.sdf_a[(int64_t)ray * a.Sa + src] : a.sdf_b[(int64_t)ray * a.Sb + (src - a.Sa)];
      sdf[s] = v;
      a.sdf_merged[(int64_t)ray * S + s] = v;
    }
  }
  lds_sync();
  // alpha_i, i < S - 1 (the last sample gets weight 0, :873)
  float al[C], one_m[C];
  float local = 1.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = lane * C + c;
    al[c] = 0.0f;
    one_m[c] = 1.0f;
    if (i < S - 1) {
      const float e0 = euclid(bin[i]), e1 = euclid(bin[i + 1]);
      const float d = e1 - e0;
      const float s0 = sdf[i], s1 = sdf[i + 1];
      float cosv = (s1 - s0) / (d + 1e-5f);
      float prev = 0.0f;
      if (i > 0) prev = (s0 - sdf[i - 1]) / ((e0 - euclid(bin[i - 1])) + 1e-5f);
      cosv = fminf(fmaxf(fminf(prev, cosv), -1e3f), 0.0f);
      const float mid = (s0 + s1) * 0.5f;
      const float pc = sigmoidf_((mid - cosv * d * 0.5f) * a.inv_s), nc = sigmoidf_((mid + cosv * d * 0.5f) * a.inv_s);
      al[c] = (pc - nc + 1e-5f) / (pc + 1e-5f);
      one_m[c] = 1.0f - al[c] + 1e-7f;
    }
    local *= one_m[c];
  }
  float T = __shfl_up(wave_incl_scan_mul(local, lane), 1);
  if (lane == 0) T = 1.0f;
  // PDF over the S weights
  float w[C];
  float wsum_l = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = lane * C + c;
    w[c] = 0.0f;
    if (i < S) {
      w[c] = (i < S - 1 ? al[c] * T : 0.0f) + a.histogram_padding;
      wsum_l += w[c];
    }
    T *= one_m[c];
  }
  float wsum = wave_sum(wsum_l);
  const float padding = fmaxf(a.eps - wsum, 0.0f);
  const float add = padding / (float)S;
  wsum += padding;
  float pl = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = lane * C + c;
    w[c] = i < S ? (w[c] + add) / wsum : 0.0f;
    pl += w[c];
  }
  float cum = wave_incl_scan_add(pl, lane) - pl;
  if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = lane * C + c;
    if (i < S) {
      cum += w[c];
      cdf[i + 1] = fminf(1.0f, cum);
    }
  }
  lds_sync();
  const int nbins = Sn + 1;
  for (int j = lane; j < nbins; j += 64) {
    const float end = a.u_end;
    const float step = end / (float)(nbins - 1);
    float u = j < nbins / 2 ? step * (float)j : end - step * (float)(nbins - 1 - j);  // torch.linspace
    u += a.jitter != nullptr ? a.jitter[a.jitter_stride ? (int64_t)ray * a.jitter_stride + j : ray] / (float)nbins : a.u_center;
    int lo = 0, hi = S + 1;
    while (lo < hi) {  // searchsorted(side = "right")
      const int m = (lo + hi) >> 1;
      if (cdf[m] > u) hi = m; else lo = m + 1;
    }
    const int below = min(max(lo - 1, 0), S), above = min(max(lo, 0), S);
    const float c0 = cdf[below], c1 = cdf[above];
    float t = (u - c0) / (c1 - c0);
    if (t != t) t = 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float b = bin[below] + t * (bin[above] - bin[below]);
    nb[j] = b;
    a.new_bins[(int64_t)ray * nbins + j] = b;
    const float e = euclid(b);
    if (j < Sn) a.new_starts[(int64_t)ray * Sn + j] = e;
    if (j > 0) a.new_ends[(int64_t)ray * Sn + j - 1] = e;
  }
  lds_sync();
  // merge: both start lists are sorted; rank by counting (list 1 first on ties, as a stable sort of cat(starts_1, starts_2))
  const int M = S + Sn;
  float* mb = a.merged_bins + (int64_t)ray * (M + 1);
  int32_t* mi = a.merged_index + (int64_t)ray * M;
  // cdf is dead: reuse it as the merged-bin scratch (the kernel is instantiated for C >= ceil((S + n_new + 1) / 64))
  float* merged = cdf;
  for (int i = lane; i < S; i += 64) {
    const float v = bin[i];
    int lo = 0, hi = Sn;
    while (lo < hi) {  // number of new starts strictly below v
      const int m = (lo + hi) >> 1;
      if (nb[m] < v) lo = m + 1; else hi = m;
    }
    mi[i + lo] = i;
    merged[i + lo] = v;
  }
  for (int j = lane; j < Sn; j += 64) {
    const float v = nb[j];
    int lo = 0, hi = S;
    while (lo < hi) {  // number of old starts <= v
      const int m = (lo + hi) >> 1;
      if (bin[m] <= v) lo = m + 1; else hi = m;
    }
    mi[j + lo] = S + j;
    merged[j + lo] = v;
  }
  if (lane == 0) merged[M] = fmaxf(bin[S], nb[Sn]);
  lds_sync();
  for (int i = lane; i <= M; i += 64) {
    const float b0 = merged[i];
    mb[i] = b0;
    if (i < M) {
      a.merged_starts[(int64_t)ray * M + i] = euclid(b0);
      a.merged_ends[(int64_t)ray * M + i] = euclid(merged[i + 1]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ VolSDF error-bounded sampler
// merge_ray_samples (ray_samplers.py:757-786) for two sorted bin sets in UniformSampler spacing: thread per element, rank by
// binary search in the other list (list 1 first on ties, as a stable sort of cat(starts_1, starts_2)).
struct MergeArgs {
  const float* bins_1;  // [N,S1+1]
  const float* bins_2;  // [N,S2+1]
  const float* nears;
  const float* fars;
  int32_t N, S1, S2;
  float* merged_bins;     // [N,S1+S2+1]
  int32_t* merged_index;  // [N,S1+S2]
};
__global__ void merge_bins_kernel(const MergeArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int M = a.S1 + a.S2;
  if (idx >= (int64_t)a.N * (M + 1)) return;
  const int ray = (int)(idx / (M + 1)), e = (int)(idx % (M + 1));
  const float* b1 = a.bins_1 + (int64_t)ray * (a.S1 + 1);
  const float* b2 = a.bins_2 + (int64_t)ray * (a.S2 + 1);
  float* mb = a.merged_bins + (int64_t)ray * (M + 1);
  int32_t* mi = a.merged_index + (int64_t)ray * M;
  if (e == M) {
    mb[M] = fmaxf(b1[a.S1], b2[a.S2]);
  } else if (e < a.S1) {
    const float v = b1[e];
    int lo = 0, hi = a.S2;
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (b2[m] < v) lo = m + 1; else hi = m;
    }
    mb[e + lo] = v;
    mi[e + lo] = e;
  } else {
    const int j = e - a.S1;
    const float v = b2[j];
    int lo = 0, hi = a.S1;
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (b1[m] <= v) lo = m + 1; else hi = m;
    }
    mb[j + lo] = v;
    mi[j + lo] = a.S1 + j;
  }
}
// euclidean starts / ends of uniform-spacing bins [N,S+1]
__global__ void uniform_euclid_kernel(const float* __restrict__ bins, const float* __restrict__ nears, const float* __restrict__ fars,
                                      const int N, const int S, float* __restrict__ starts, float* __restrict__ ends) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * (S + 1)) return;
  const int ray = (int)(idx / (S + 1)), j = (int)(idx % (S + 1));
  const float b = bins[idx];
  const float e = b * fars[ray] + (1.0f - b) * nears[ray];
  if (j < S) starts[(int64_t)ray * S + j] = e;
  if (j > 0) ends[(int64_t)ray * S + j - 1] = e;
}

// One outer iteration of VolSDF Algorithm 1 up to the resampling weights (ray_samplers.py:650-674): merge the new sdf values
// (:654-658), d* (Theorem 1, :704-726), beta by bisection on the error bound (:728-755: 1 + beta_iters bound evaluations, each two
// wave scans and a max), density -> weights / transmittance (rays.py:169-192), and the error-proportional weights (:676-683).
// One wavefront per ray, C consecutive samples per lane, everything in registers between the scans.
struct VolsdfStepArgs {
  const float* bins_in;   // [N,S+1]
  const float* sdf_a;     // [N,Sa]
  const float* sdf_b;     // [N,Sb] or null
  const int32_t* index;   // [N,S] or null
  const float* nears;
  const float* fars;
  const float* beta_in;   // [N]
  const float* beta0;     // [1]  density_fn.get_beta()
  int32_t N, Sa, Sb, beta_iters;
  float eps;
  float* sdf_merged;      // [N,S]
  float* beta_out;        // [N]
  float* weights;         // [N,S]
  float* err_weights;     // [N,S]
  int32_t* not_converged; // [1]  max over rays of (beta_out > beta0); caller zeroes
};

SDFHIP_D float laplace_density_f(const float sdf, const float beta) {  // sdf_field.py:49-71
  const float sg = sdf > 0.0f ? 1.0f : (sdf < 0.0f ? -1.0f : 0.0f);
  return (1.0f / beta) * (0.5f + 0.5f * sg * expm1f(-fabsf(sdf) / beta));
}
SDFHIP_D float wave_max(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

template <int C>
__global__ __launch_bounds__(256) void volsdf_step_kernel(const VolsdfStepArgs a) {
  __shared__ float sdf_s[4][64 * C + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  if (ray >= a.N) return;
  const int S = a.Sa + a.Sb;
  float* sl = sdf_s[wv];
  const float near = a.nears[ray], far = a.fars[ray];
  const float* bin = a.bins_in + (int64_t)ray * (S + 1);
  for (int s = lane; s < S; s += 64) {
    const int src = a.index != nullptr ? a.index[(int64_t)ray * S + s] : s;
    const float v = src < a.Sa ? a.sdf_a[(int64_t)ray * a.Sa + src] : a.sdf_b[(int64_t)ray * a.Sb + (src - a.Sa)];
    sl[s] = v;
    a.sdf_merged[(int64_t)ray * S + s] = v;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  float sd[C], dl[C], ds[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = lane * C + c;
    sd[c] = 0.0f;
    dl[c] = 0.0f;
    ds[c] = 0.0f;
    if (i < S) {
      sd[c] = sl[i];
      const float b0 = bin[i], b1 = bin[i + 1];
      dl[c] = (b1 * far + (1.0f - b1) * near) - (b0 * far + (1.0f - b0) * near);
    }
  }
  // d*: interval i uses samples i and i + 1; the last sample repeats the previous interval's value
  auto dstar_of = [&](const int i, const float dist) {
    const float s0 = sl[i], s1 = sl[i + 1];
    const float A = dist, B = fabsf(s0), Cc = fabsf(s1);
    const bool first = A * A + B * B <= Cc * Cc, second = A * A + Cc * Cc <= B * B;
    float d = 0.0f;
    if (first) d = B;
    if (second) d = Cc;
    const float sh = (A + B + Cc) * 0.5f;
    const float area = sh * (sh - A) * (sh - B) * (sh - Cc);
    if (!first && !second && (B + Cc - A > 0.0f)) d = (2.0f * sqrtf(area)) / A;
    const float sg0 = s0 > 0.0f ? 1.0f : (s0 < 0.0f ? -1.0f : 0.0f), sg1 = s1 > 0.0f ? 1.0f : (s1 < 0.0f ? -1.0f : 0.0f);
    return (sg0 * sg1 == 1.0f) ? d : 0.0f;
  };
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = lane * C + c;
    if (i < S - 1) {
      ds[c] = dstar_of(i, dl[c]);
    } else if (i == S - 1 && S >= 2) {
      const float b0 = bin[S - 2], b1 = bin[S - 1];
      ds[c] = dstar_of(S - 2, (b1 * far + (1.0f - b1) * near) - (b0 * far + (1.0f - b0) * near));
    }
  }
  // error bound for a given beta (get_error_bound); optionally returns per-sample transmittance and error integral
  auto error_bound = [&](const float beta) {
    float dd[C], es[C];
    float ldd = 0.0f, les = 0.0f;
    const float inv4b2 = 1.0f / (4.0f * beta * beta);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int i = lane * C + c;
      dd[c] = i < S ? dl[c] * laplace_density_f(sd[c], beta) : 0.0f;
      es[c] = i < S ? expf(-ds[c] / beta) * (dl[c] * dl[c]) * inv4b2 : 0.0f;
      ldd += dd[c];
      les += es[c];
    }
    float integral = wave_incl_scan_add(ldd, lane) - ldd;  // exclusive over lanes
    float err = wave_incl_scan_add(les, lane) - les;
    float mx = 0.0f;
    bool any = false;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int i = lane * C + c;
      if (i < S) {
        err += es[c];
        const float bnd = (fminf(expf(err), 1.0e6f) - 1.0f) * expf(-integral);
        mx = any ? fmaxf(mx, bnd) : bnd;
        any = true;
        integral += dd[c];
      }
    }
    // max over the S samples (all bounds are >= 0; lanes without samples contribute 0, which cannot exceed a real maximum >= 0)
    return wave_max(any ? mx : 0.0f);
  };
  const float beta0 = a.beta0[0];
  float beta = a.beta_in[ray];
  {
    const float curr = error_bound(beta0);
    if (curr <= a.eps) beta = beta0;
    float bmin = beta0, bmax = beta;
    for (int j = 0; j < a.beta_iters; ++j) {
      const float mid = (bmin + bmax) * 0.5f;
      const float e = error_bound(mid);
      if (e <= a.eps) bmax = mid; else bmin = mid;
    }
    beta = bmax;
  }
  // density -> weights, transmittance; error-proportional weights
  float dd[C], es[C];
  float ldd = 0.0f, les = 0.0f;
  const float inv4b2 = 1.0f / (4.0f * beta * beta);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = lane * C + c;
    dd[c] = i < S ? dl[c] * laplace_density_f(sd[c], beta) : 0.0f;
    es[c] = i < S ? expf(-ds[c] / beta) * (dl[c] * dl[c]) * inv4b2 : 0.0f;
    ldd += dd[c];
    les += es[c];
  }
  float integral = wave_incl_scan_add(ldd, lane) - ldd;
  float err = wave_incl_scan_add(les, lane) - les;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int i = lane * C + c;
    if (i < S) {
      const float T = expf(-integral);
      err += es[c];
      a.weights[(int64_t)ray * S + i] = (1.0f - expf(-dd[c])) * T;
      a.err_weights[(int64_t)ray * S + i] = (fminf(expf(err), 1.0e6f) - 1.0f) * T;
      integral += dd[c];
    }
  }
  if (lane == 0) {
    a.beta_out[ray] = beta;
    if (beta > beta0) atomicMax(a.not_converged, 1);
  }
}

// ------------------------------------------------------------------------------------------------ interlevel loss (Zip-NeRF)
// interlevel_loss_zip (model_components/losses.py:116-172), one proposal level per launch, one wavefront per ray, no sort:
// the blurred step function's 2 (S + 1) knots {c_i - r} U {c_i + r} are two sorted sequences, so the rank of a knot in the
// merged order is its own index plus a binary search in the other sequence (on the COMPUTED fp32 knot values, so ties fall
// exactly as a sort of those values would put them); the reference's sort + 2 gathers + 3 cumsums + searchsorted +
// 4 gathers (~40 launches, a radix sort among them) become LDS traffic of one wave.
//   y1_i   = (wn_i - wn_{i-1}) / (2 r),  wn = w / diff(c), wn_{-1} = wn_S = 0           (:120-121)
//   slope  = cumsum of (+y1 at a left knot, -y1 at a right knot) over the merged knots   (:122-126)
//   y_r    = max(0, [0, cumsum(diff(x_r) * slope)])                                       (:127-128, :143)
//   y_cum  = [0, cumsum(trapezoids of y_r)]                                               (:147-148)
//   bins_k = y_cum interpolated at the proposal bin edge cp_k (searchsorted right)        (:156-166)
//   w_gt   = diff(bins) ; loss terms clip(w_gt - wp, 0)^2 / (wp + 1e-5)                   (:168-171)
// Outputs per sample the loss term and its derivative w.r.t. wp (only the proposal weights carry gradient: c, w and cp are
// detached, :133-134 and the sampler's bins.detach()); the mean over N S_p and the chain rule are one torch op each.
struct InterlevelArgs {
  const float* c;    // [N,S+1]  field spacing bins
  const float* w;    // [N,S]    field weights
  const float* cp;   // [N,Sp+1] proposal spacing bins
  const float* wp;   // [N,Sp]   proposal weights
  int32_t N, S, Sp;
  float r;
  float* term;       // [N,Sp]  clip(w_gt - wp, 0)^2 / (wp + 1e-5)
  float* dterm;      // [N,Sp]  d term / d wp
  float* w_gt;       // [N,Sp]  (diagnostics / tests) or null
};

// inclusive scan of n fp32 values in LDS (blocked over the 64 lanes: lane l owns [l m, (l + 1) m)), in place, ACCUMULATED IN
// DOUBLE and rounded once per output - what torch.cumsum does on the CPU (at::acc_type<float, false> is double), i.e. what the
// reference's CPU path computes; a plain fp32 scan is ~16x further from the exact sums (the slopes are differences of
// normalised weights divided by 2 r = 0.006 that cancel to zero over the ray)
SDFHIP_D void wave_lds_scan_add(float* v, const int n, const int lane) {
  const int m = (n + 63) / 64;
  const int b = lane * m, e = b + m < n ? b + m : n;
  double s = 0.0;
  for (int k = b; k < e; ++k) s += (double)v[k];
  double incl = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  double run = incl - s;  // exclusive prefix of this lane's chunk
  for (int k = b; k < e; ++k) {
    run += (double)v[k];
    v[k] = (float)run;
  }
}

__global__ __launch_bounds__(256) void interlevel_kernel(const InterlevelArgs a) {
  extern __shared__ float ilds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  if (ray >= a.N) return;  // whole waves leave together; only wave-level primitives and wave-private LDS below
  const int S = a.S, Sp = a.Sp, nk = 2 * (S + 1);
  float* base = ilds + (size_t)wv * (4 * (S + 1) + 2 * nk + (Sp + 1));
  float* cl = base;               // [S+1] left knots  c_i - r
  float* cr = cl + (S + 1);       // [S+1] right knots c_i + r
  float* y1 = cr + (S + 1);       // [S+1]
  float* cc = y1 + (S + 1);       // [S+1] c
  float* xr = cc + (S + 1);       // [nk]  merged knots
  float* yy = xr + nk;            // [nk]  slope increments -> slopes -> y_r -> y_cum
  float* bb = yy + nk;            // [Sp+1] interpolated cumulative mass at the proposal bin edges
  const float* c = a.c + (size_t)ray * (S + 1);
  const float* w = a.w + (size_t)ray * S;
  const float r = a.r;
  for (int i = lane; i <= S; i += 64) {
    const float ci = c[i];
    cc[i] = ci;
    cl[i] = ci - r;
    cr[i] = ci + r;
  }
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i <= S; i += 64) {
    const float wn = i < S ? w[i] / (cc[i + 1] - cc[i]) : 0.0f;
    const float wm = i > 0 ? w[i - 1] / (cc[i] - cc[i - 1]) : 0.0f;
    y1[i] = (wn - wm) / (2.0f * r);
  }
  __builtin_amdgcn_wave_barrier();
  // merge: a left knot precedes an equal right knot (concatenation order [left | right] under a stable sort)
  for (int i = lane; i <= S; i += 64) {
    const float key = cl[i];
    int lo = 0, hi = S + 1;  // number of right knots strictly below the key
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cr[mid] < key) lo = mid + 1;
      else hi = mid;
    }
    xr[i + lo] = key;
    yy[i + lo] = y1[i];
    const float key2 = cr[i];
    lo = 0, hi = S + 1;      // number of left knots at or below the key
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cl[mid] <= key2) lo = mid + 1;
      else hi = mid;
    }
    xr[i + lo] = key2;
    yy[i + lo] = -y1[i];
  }
  __builtin_amdgcn_wave_barrier();
  wave_lds_scan_add(yy, nk - 1, lane);  // slopes on the nk - 1 intervals (the last knot's increment is not used, :124)
  __builtin_amdgcn_wave_barrier();
  // seg_k = (x_k - x_{k-1}) slope_{k-1} (seg_0 = 0) into the staging area of the knots, which is free now (cl .. cc = 2 nk floats)
  float* tmp = cl;
  for (int k = lane; k < nk; k += 64) tmp[k] = k >= 1 ? (xr[k] - xr[k - 1]) * yy[k - 1] : 0.0f;
  __builtin_amdgcn_wave_barrier();
  wave_lds_scan_add(tmp, nk, lane);     // y_r (before the clip)
  __builtin_amdgcn_wave_barrier();
  for (int k = lane; k < nk; k += 64)
    yy[k] = k >= 1 ? (fmaxf(tmp[k], 0.0f) + fmaxf(tmp[k - 1], 0.0f)) * 0.5f * (xr[k] - xr[k - 1]) : 0.0f;
  __builtin_amdgcn_wave_barrier();
  wave_lds_scan_add(yy, nk, lane);      // y_cum
  __builtin_amdgcn_wave_barrier();
  const float* cp = a.cp + (size_t)ray * (Sp + 1);
  for (int k = lane; k <= Sp; k += 64) {
    const float x = cp[k];
    int lo = 0, hi = nk;  // searchsorted(side="right"): number of knots <= x
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (xr[mid] <= x) lo = mid + 1;
      else hi = mid;
    }
    const int below = min(max(lo - 1, 0), nk - 1), above = min(max(lo, 0), nk - 1);
    const float x0 = xr[below], x1 = xr[above], b0 = yy[below], b1 = yy[above];
    float t = (x - x0) / (x1 - x0);
    t = (t != t) ? 0.0f : t;                       // nan_to_num(.., 0): 0 / 0 when below == above
    t = fminf(fmaxf(t, 0.0f), 1.0f);               // +-inf clip to the ends
    bb[k] = b0 + t * (b1 - b0);
  }
  __builtin_amdgcn_wave_barrier();
  const float* wp = a.wp + (size_t)ray * Sp;
  for (int k = lane; k < Sp; k += 64) {
    const float wg = bb[k + 1] - bb[k];
    const float p = wp[k], q = p + 1e-5f;
    const float d = fmaxf(wg - p, 0.0f);
    const size_t o = (size_t)ray * Sp + k;
    a.term[o] = d * d / q;
    a.dterm[o] = -2.0f * d / q - d * d / (q * q);
    if (a.w_gt != nullptr) a.w_gt[o] = wg;
  }
}

// ------------------------------------------------------------------------------------------------ surface root finding (UniSurf)
// UniSurfSampler's ray / surface intersection (model_components/ray_samplers.py:1030-1075): along the num_marching_steps uniform
// samples of a ray, the FIRST interval whose sdf changes sign (sdf_i sdf_{i+1} < 0), accepted only if it goes from outside to
// inside (sdf_i > 0); the depth is the linear interpolation (regula falsi step) of the two bracketing samples, and the ray's
// [near, far] shrinks to z +- (far - near) delta, clipped to the original interval (:1068-1075).  One thread per ray.
struct RootArgs {
  const float* sdf;      // [N,S]
  const float* starts;   // [N,S]
  const float* nears;    // [N]
  const float* fars;     // [N]
  int32_t N, S;
  float delta;
  int32_t* mask;         // [N]  1: a surface was found
  float* z;              // [N]  its depth (undefined where mask == 0)
  float* new_nears;      // [N]
  float* new_fars;       // [N]
};

__global__ void surface_root_kernel(const RootArgs a) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= a.N) return;
  const float* sdf = a.sdf + (size_t)ray * a.S;
  const float* st = a.starts + (size_t)ray * a.S;
  int idx = -1;
  float prev = sdf[0];
  for (int i = 0; i + 1 < a.S; ++i) {
    const float next = sdf[i + 1];
    if (prev * next < 0.0f) {  // torch.sign(sdf_i sdf_{i+1}) == -1; the argmin over sign x (S - i) picks the first such i (:1039-1045)
      idx = i;
      break;
    }
    prev = next;
  }
  const float near_ = a.nears[ray], far_ = a.fars[ray];
  bool ok = idx >= 0 && sdf[idx] > 0.0f;
  float z = 0.0f, nn = near_, nf = far_;
  if (ok) {
    const int hi = idx + 1 < a.S ? idx + 1 : a.S - 1;
    const float d_low = st[idx], v_low = sdf[idx], d_high = st[hi], v_high = sdf[hi];
    z = (v_low * d_high - v_high * d_low) / (v_low - v_high);
    const float w = (far_ - near_) * a.delta;
    nn = fmaxf(z - w, near_);
    nf = fminf(z + w, far_);
  }
  a.mask[ray] = ok ? 1 : 0;
  a.z[ray] = z;
  a.new_nears[ray] = nn;
  a.new_fars[ray] = nf;
}
