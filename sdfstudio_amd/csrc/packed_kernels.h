// sdfhip — packed-sample path of NeuS-acc: occupancy-grid ray marching and segmented compositing.  Restates
// the three nerfacc (== 0.3.5, pyproject.toml:31; CUDA-only, not vendored) operators the reference calls:
//   nerfacc.cuda.ray_marching        (model_components/ray_samplers.py:1474-1484)  -> march_kernel (count pass, write pass)
//   nerfacc.render_weight_from_alpha (models/neus_acc.py:103-107)                  -> packed_weights_{fwd,bwd}_kernel
//   nerfacc.accumulate_along_rays    (models/neus_acc.py:108-121)                  -> packed_accumulate_kernel
// "Packed": the samples of all rays in one array, ray r owning [offset_r, offset_r + count_r) (nerfacc's packed_info).
#pragma once
#include "ray_kernels.h"

struct MarchArgs {
  const float* origins;   // [N,3]
  const float* dirs;      // [N,3]
  const float* t_min;     // [N]
  const float* t_max;     // [N]
  const uint8_t* binary;  // [R,R,R] occupancy (x-major: idx = (ix * R + iy) * R + iz), torch.bool storage
  float roi_min[3], roi_max[3];
  int32_t N, R;
  float step;             // dt of every sample (cone_angle = 0: nerfacc's calc_dt clamps to dt_min = step)
  const float* step_dev;  // or null: the step as a DEVICE scalar (overrides `step`): NeuS-acc derives it from a trained parameter every
                          // iteration (ray_samplers.py:1379-1382: 14 / inv_s / 16) - read here, the host never has to
  int64_t capacity;       // write pass: samples at packed positions >= capacity are dropped (< 0: no bound)
  // count pass: counts [N].  write pass: offsets [N] (exclusive scan of the counts) -> ray_indices / t_starts / t_ends [P]
  int32_t* counts;
  const int64_t* offsets;
  int64_t* ray_indices;
  float* t_starts;
  float* t_ends;
};

// occupancy of the voxel holding xyz; outside the region of interest: empty (nerfacc grid_occupied_at, ContractionType::AABB)
SDFHIP_D bool march_occupied(const MarchArgs& a, const float x, const float y, const float z) {
  if (x < a.roi_min[0] || x > a.roi_max[0] || y < a.roi_min[1] || y > a.roi_max[1] || z < a.roi_min[2] || z > a.roi_max[2]) return false;
  const float R = (float)a.R;
  const float ux = (x - a.roi_min[0]) / (a.roi_max[0] - a.roi_min[0]) * R;
  const float uy = (y - a.roi_min[1]) / (a.roi_max[1] - a.roi_min[1]) * R;
  const float uz = (z - a.roi_min[2]) / (a.roi_max[2] - a.roi_min[2]) * R;
  const int ix = min(max((int)ux, 0), a.R - 1), iy = min(max((int)uy, 0), a.R - 1), iz = min(max((int)uz, 0), a.R - 1);
  return a.binary[((int64_t)ix * a.R + iy) * a.R + iz] != 0;
}
SDFHIP_D float march_sign(const float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
// distance along the ray to the boundary of the current voxel (nerfacc distance_to_next_voxel), then fixed steps of dt up to it
SDFHIP_D float march_advance(const MarchArgs& a, float t, const float x, const float y, const float z, const float d[3], const float inv_d[3]) {
  const float R = (float)a.R;
  const float p[3] = {x, y, z};
  float tt = 3.0e38f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float ext = a.roi_max[k] - a.roi_min[k];
    const float u = (p[k] - a.roi_min[k]) / ext * R;
    const float tk = ((floorf(u + 0.5f + 0.5f * march_sign(d[k])) - u) * inv_d[k]) / R * ext;
    tt = fminf(tt, tk);
  }
  const float target = t + fmaxf(tt, 0.0f);
  do {
    t += a.step;
  } while (t < target);
  return t;
}

// thread per ray.  WRITE = false: count the samples; WRITE = true: emit them at the ray's offset
template <bool WRITE>
__global__ void march_kernel(const MarchArgs a_in) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= a_in.N) return;
  MarchArgs a = a_in;
  if (a.step_dev != nullptr) a.step = *a.step_dev;
  const float o[3] = {a.origins[ray * 3], a.origins[ray * 3 + 1], a.origins[ray * 3 + 2]};
  const float d[3] = {a.dirs[ray * 3], a.dirs[ray * 3 + 1], a.dirs[ray * 3 + 2]};
  const float inv_d[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
  const float near = a.t_min[ray], far = a.t_max[ray];
  int64_t base = 0;
  if constexpr (WRITE) base = a.offsets[ray];
  int j = 0;
  float t0 = near, t1 = t0 + a.step, tm = (t0 + t1) * 0.5f;
  while (tm < far) {
    // one rounding per coordinate (the CUDA original contracts to fma as well); the oracle forms the same value
    const float x = __builtin_fmaf(tm, d[0], o[0]), y = __builtin_fmaf(tm, d[1], o[1]), z = __builtin_fmaf(tm, d[2], o[2]);
    if (march_occupied(a, x, y, z)) {
      if constexpr (WRITE) {
        if (a.capacity < 0 || base + j < a.capacity) {
          a.ray_indices[base + j] = ray;
          a.t_starts[base + j] = t0;
          a.t_ends[base + j] = t1;
        }
      }
      ++j;
      t0 = t1;
      t1 = t0 + a.step;
      tm = (t0 + t1) * 0.5f;
    } else {
      tm = march_advance(a, tm, x, y, z, d, inv_d);
      t0 = tm - a.step * 0.5f;
      t1 = tm + a.step * 0.5f;
    }
  }
  if constexpr (!WRITE) a.counts[ray] = j;
}

// nerfacc.ray_resampling (model_components/ray_samplers.py:1496-1498, importance_sampling = True): per ray, n_out new intervals whose
// edges are the inverse CDF of the packed weights at the n_out + 1 centres u_j = (j + 1/2) / (n_out + 1) ... spaced by
// (1 - 1 / (n_out + 1)) / n_out, linear inside the source intervals; weights are padded to a sum of at least 1e-5.  Rays without
// samples stay empty.  Thread per ray (a serial walk over the ray's two short lists).
struct ResampleArgs {
  const int64_t* offsets;      // [N] source
  const int32_t* counts;       // [N]
  const float* t_starts;       // [P]
  const float* t_ends;         // [P]
  const float* weights;        // [P]
  int32_t N, n_out;
  float* out_starts;           // [N_nonempty * n_out], ray r at (number of non-empty rays before r) * n_out
  float* out_ends;
  const int64_t* out_offsets;  // [N]
};
__global__ void packed_resample_kernel(const ResampleArgs a) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= a.N) return;
  const int steps = a.counts[ray];
  if (steps == 0) return;
  const float* st = a.t_starts + a.offsets[ray];
  const float* en = a.t_ends + a.offsets[ray];
  const float* w = a.weights + a.offsets[ray];
  float* os = a.out_starts + a.out_offsets[ray];
  float* oe = a.out_ends + a.out_offsets[ray];
  float wsum = 0.0f;
  for (int j = 0; j < steps; ++j) wsum += w[j];
  const float padding = fmaxf(1e-5f - wsum, 0.0f);
  const float pad_step = padding / (float)steps;
  wsum += padding;
  const int num_bins = a.n_out + 1;
  const float cdf_step = (1.0f - 1.0f / (float)num_bins) / (float)a.n_out;
  int idx = 0, j = 0;
  float cdf_prev = 0.0f, cdf_next = (w[0] + pad_step) / wsum;
  float cdf_u = 1.0f / (float)(2 * num_bins);
  while (j < num_bins) {
    if (cdf_u < cdf_next || idx == steps - 1) {  // (the last interval also takes what round-off leaves above its cdf)
      const float scaling = (en[idx] - st[idx]) / (cdf_next - cdf_prev);
      const float t = __builtin_fmaf(cdf_u - cdf_prev, scaling, st[idx]);  // one rounding (the CUDA original contracts as well)
      if (j < num_bins - 1) os[j] = t;
      if (j > 0) oe[j - 1] = t;
      cdf_u += cdf_step;
      ++j;
    } else {
      ++idx;
      cdf_prev = cdf_next;
      cdf_next += (w[idx] + pad_step) / wsum;
    }
  }
}

struct PackedArgs {
  const int64_t* offsets;  // [N]
  const int32_t* counts;   // [N]
  int32_t N, D;
  const float* alpha;      // [P]
  float* weights;          // [P]   w_i = alpha_i T_i,  T_i = prod_{j < i} (1 - alpha_j) within the ray's segment
  float* trans;            // [P]   T_i (forward output, backward input)
  const float* wbar;       // [P]
  float* alphabar;         // [P]
  const float* values;     // [P,D] or null (accumulate the weights themselves, D = 1)
  float* out;              // [N,D]
};

// one wave per ray, 64 samples per round, the transmittance carried from round to round
__global__ __launch_bounds__(256) void packed_weights_fwd_kernel(const PackedArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int64_t off = a.offsets[ray];
  const int cnt = a.counts[ray];
  float carry = 1.0f;
  for (int b = 0; b < cnt; b += 64) {
    const int i = b + lane;
    const float al = i < cnt ? a.alpha[off + i] : 0.0f;
    const float incl = wave_incl_scan_mul(1.0f - al, lane);
    float T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.0f;
    T *= carry;
    if (i < cnt) {
      a.weights[off + i] = al * T;
      a.trans[off + i] = T;
    }
    carry *= __shfl(incl, 63);
  }
}

// alphabar_i = wbar_i T_i - (sum_{j > i} wbar_j w_j) / (1 - alpha_i): rounds in reverse order, the suffix sum carried
__global__ __launch_bounds__(256) void packed_weights_bwd_kernel(const PackedArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int64_t off = a.offsets[ray];
  const int cnt = a.counts[ray];
  float carry = 0.0f;
  const int rounds = (cnt + 63) / 64;
  for (int r = rounds - 1; r >= 0; --r) {
    const int i = r * 64 + lane;
    const bool ok = i < cnt;
    const float al = ok ? a.alpha[off + i] : 0.0f;
    const float w = ok ? a.weights[off + i] : 0.0f;
    const float wb = ok ? a.wbar[off + i] : 0.0f;
    const float term = wb * w;
    // inclusive suffix sum over the lanes: reverse the lane order, prefix-scan, reverse back
    const float rev = __shfl(term, 63 - lane);
    const float incl_rev = wave_incl_scan_add(rev, lane);
    const float suffix_incl = __shfl(incl_rev, 63 - lane);
    const float after = suffix_incl - term + carry;  // sum over j > i
    if (ok) {
      const float one_m = 1.0f - al;
      // alpha == 1 ends the ray: every later weight is zero and so is the sum (0 / 0 -> 0)
      a.alphabar[off + i] = wb * a.trans[off + i] - (one_m > 0.0f ? after / one_m : 0.0f);
    }
    carry += __shfl(incl_rev, 63);
  }
}

// out[ray] = sum over the ray's segment of w_i * values_i (values null: of w_i); empty rays write zeros
__global__ __launch_bounds__(256) void packed_accumulate_kernel(const PackedArgs a) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int64_t off = a.offsets[ray];
  const int cnt = a.counts[ray];
  for (int d = 0; d < a.D; ++d) {
    float s = 0.0f;
    for (int i = lane; i < cnt; i += 64) {
      const float w = a.weights[off + i];
      s = fmaf(w, a.values ? a.values[(off + i) * a.D + d] : 1.0f, s);
    }
    s = wave_sum(s);
    if (lane == 0) a.out[(int64_t)ray * a.D + d] = s;
  }
}
