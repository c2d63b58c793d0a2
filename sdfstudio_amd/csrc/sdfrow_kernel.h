// sdfhip - gradient of the sdf output row (instantiated per network shape, field_inst.h); the weight-gradient GEMMs live in
// wgrad_kernels.h, which only api.hip includes.
#pragma once
#include "common.h"

// Gradient of the sdf output row (lane-local dot product in geo_fwd_kernel):
//   w_sdf_bar[k] = sum_p ( sdfbar_p * u_last[p][k] + qb_last[p][k] ),   b_sdf_bar = sum_p sdfbar_p      u_last = act(z_last), as saved
// (qb_last == nullptr: first-order backward, no tangent term)
// grid = n_split, block = 256 (the 4 waves interleave over the split's tiles, then sum through LDS).
// partial: [n_split][NBH*32 + 32]  (last 32-slot holds b_sdf_bar in [0])
template <int NBH>
__global__ __launch_bounds__(256) void sdfrow_grad_kernel(const float* __restrict__ u_last, const float* __restrict__ qb_last,
                                                            const float* __restrict__ sdfbar, const int64_t n_tiles,
                                                            const int tiles_per_split, float* __restrict__ partial) {
  __shared__ float red[4][NBH * 32 + 32];
  const int lane = threadIdx.x & 63, hf = lane >> 5, wave = threadIdx.x >> 6;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_split;
  int64_t t1 = t0 + tiles_per_split;
  if (t1 > n_tiles) t1 = n_tiles;
  f32x16 acc[NBH];
#pragma unroll
  for (int b = 0; b < NBH; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
  float bsum = 0.0f;
  for (int64_t tile = t0 + wave; tile < t1; tile += 4) {
    const float sb = sdfbar[tile * 32 + (lane & 31)];
    bsum += sb;
#pragma unroll
    for (int b = 0; b < NBH; ++b) {
      const float* up = u_last + ((size_t)tile * NBH + b) * 1024 + lane;
      const float* qp = qb_last + ((size_t)tile * NBH + b) * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[b][r] += fmaf(sb, up[r * 64], qb_last != nullptr ? qp[r * 64] : 0.0f);
      }
    }
  }
  // reduce over the 32 points of a half-wave
#pragma unroll
  for (int m = 1; m < 32; m <<= 1) {
    bsum += __shfl_xor(bsum, m);
#pragma unroll
    for (int b = 0; b < NBH; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] += __shfl_xor(acc[b][r], m);
  }
  if ((lane & 31) == 0) {
#pragma unroll
    for (int b = 0; b < NBH; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave][b * 32 + tp_row(r, hf)] = acc[b][r];
    if (hf == 0) red[wave][NBH * 32] = bsum;
  }
  __syncthreads();
  float* dst = partial + (size_t)blockIdx.x * (NBH * 32 + 32);
  for (int i = threadIdx.x; i <= NBH * 32; i += 256) dst[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}
