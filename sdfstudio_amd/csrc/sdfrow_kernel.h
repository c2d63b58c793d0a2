// sdfhip - gradient of the sdf output row (instantiated per network shape, field_inst.h); the weight-gradient GEMMs live in
// wgrad_kernels.h, which only api.hip includes.
#pragma once
#include "common.h"

// Gradient of the sdf output row (lane-local dot product in geo_fwd_kernel):
//   w_sdf_bar[k] = sum_p ( sdfbar_p * u_last[p][k] + qb_last[p][k] ),   b_sdf_bar = sum_p sdfbar_p      u_last = act(z_last), as saved
// (qb_last == nullptr: first-order backward, no tangent term)
// grid = (n_split, NBH / NBG), block = 256 (the 4 waves interleave over the split's tiles, then sum through LDS).  A workgroup sums NBG <= 8
// of the row's NBH 32-feature blocks (blockIdx.y picks the group): 16 blocks at once are 256 accumulator registers plus operands and
// spilled (1.1 KB of scratch per lane, 1.8 ms per config-2-sized launch at hidden 512 against 0.4 ms of bytes).
// partial: [n_split][NBH*32 + 32]  (last 32-slot holds b_sdf_bar in [0])
template <int NBH, int NBG = (NBH > 8 ? 8 : NBH)>
__global__ __launch_bounds__(256) void sdfrow_grad_kernel(const float* __restrict__ u_last, const float* __restrict__ qb_last,
                                                            const float* __restrict__ sdfbar, const int64_t n_tiles,
                                                            const int tiles_per_split, float* __restrict__ partial) {
  static_assert(NBH % NBG == 0, "block groups tile the row");
  __shared__ float red[4][NBG * 32 + 32];
  const int b0 = blockIdx.y * NBG;  // first block of this workgroup's group
  const int lane = threadIdx.x & 63, hf = lane >> 5, wave = threadIdx.x >> 6;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_split;
  int64_t t1 = t0 + tiles_per_split;
  if (t1 > n_tiles) t1 = n_tiles;
  f32x16 acc[NBG];
#pragma unroll
  for (int b = 0; b < NBG; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
  float bsum = 0.0f;
  for (int64_t tile = t0 + wave; tile < t1; tile += 4) {
    const float sb = sdfbar[tile * 32 + (lane & 31)];
    bsum += sb;
#pragma unroll
    for (int b = 0; b < NBG; ++b) {
      const float* up = u_last + ((size_t)tile * NBH + b0 + b) * 1024 + lane;
      const float* qp = qb_last + ((size_t)tile * NBH + b0 + b) * 1024 + lane;
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
    for (int b = 0; b < NBG; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] += __shfl_xor(acc[b][r], m);
  }
  if ((lane & 31) == 0) {
#pragma unroll
    for (int b = 0; b < NBG; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave][b * 32 + tp_row(r, hf)] = acc[b][r];
    if (hf == 0) red[wave][NBG * 32] = bsum;
  }
  __syncthreads();
  float* dst = partial + (size_t)blockIdx.x * (NBH * 32 + 32);
  for (int i = threadIdx.x; i < NBG * 32; i += 256) dst[b0 * 32 + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  if (blockIdx.y == 0 && threadIdx.x == 0) dst[NBH * 32] = red[0][NBG * 32] + red[1][NBG * 32] + red[2][NBG * 32] + red[3][NBG * 32];
}
