// sdfhip — the multi-resolution hash-grid encoding as a standalone operator: positions in [0,1]^3 -> [P, L * F] features
// (row-major, level-major columns), and the scatter of d L / d features into the table gradient.  This is what the reference
// gets from tcnn.Encoding("HashGrid") when the encoding is not fused into a field kernel (fields/nerfacto_field.py:137-156: the
// background field of BASELINE config 5).  Same cell / index / weight arithmetic as the fused kernels (point_kernels.h grid_cell).
#pragma once
#include "point_kernels.h"

struct GridEncodeArgs {
  GridDev grid;
  const float* x;        // [P,3] in [0,1] (positions outside wrap like tiny-cuda-nn's)
  int64_t n_points;
  const float* table;    // [entries][F]
  float* feat;           // [P, L*F]
  const float* featbar;  // [P, L*F]
  float* tablebar;       // [entries][F], accumulated
};

// grid = (ceil(P / 256), L * F / 2): one thread per (point, level, feature pair)
template <bool BWD>
__global__ __launch_bounds__(256) void grid_encode_kernel(const GridEncodeArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n_points) return;
  const int F = a.grid.n_features, pairs = F >> 1;
  const int level = blockIdx.y / pairs, pair = blockIdx.y % pairs;
  const float pp[3] = {a.x[p * 3], a.x[p * 3 + 1], a.x[p * 3 + 2]};
  GridCell c;
  grid_cell(a.grid.lv[level], a.grid.smoothstep != 0, pp, c);
  const int64_t col = (int64_t)level * F + pair * 2, width = (int64_t)a.grid.n_levels * F;
  if constexpr (!BWD) {
    const float2* tab = reinterpret_cast<const float2*>(a.table);
    float y0 = 0.0f, y1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float2 v = tab[(size_t)c.idx[k] * pairs + pair];
      const float w = corner_w(c, k);
      y0 = fmaf(w, v.x, y0);
      y1 = fmaf(w, v.y, y1);
    }
    a.feat[p * width + col] = y0;
    a.feat[p * width + col + 1] = y1;
  } else {
    const float g0 = a.featbar[p * width + col], g1 = a.featbar[p * width + col + 1];
    if (g0 == 0.0f && g1 == 0.0f) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = corner_w(c, k);
      float* dst = a.tablebar + ((size_t)c.idx[k] * pairs + pair) * 2;
      atomicAdd(dst, w * g0);
      atomicAdd(dst + 1, w * g1);
    }
  }
}

// ---- test / diagnostic entry: the cell of every (point, level) exactly as the fused kernels index the table - the 8 corner entry
// indices (level offset included; corner k: bit 0 -> +x, bit 1 -> +y, bit 2 -> +z) and the three interpolation weights.  The GPU
// parity test compares the indices BIT FOR BIT with oracle/hashgrid.py level_cell, whose hashed branch is pinned on the
// reference's own HashEncoding.hash_fn (field_components/encodings.py:338-355) by the CPU suite.
struct GridDumpArgs {
  GridDev grid;
  const float* x;   // [P,3]
  int64_t n_points;
  uint32_t* idx;    // [P][L][8]
  float* w;         // [P][L][3]
};
__global__ __launch_bounds__(256) void grid_cell_dump_kernel(const GridDumpArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n_points) return;
  const int level = blockIdx.y, L = a.grid.n_levels;
  const float pp[3] = {a.x[p * 3], a.x[p * 3 + 1], a.x[p * 3 + 2]};
  GridCell c;
  grid_cell(a.grid.lv[level], a.grid.smoothstep != 0, pp, c);
#pragma unroll
  for (int k = 0; k < 8; ++k) a.idx[(p * L + level) * 8 + k] = c.idx[k];
#pragma unroll
  for (int d = 0; d < 3; ++d) a.w[(p * L + level) * 3 + d] = c.w[d];
}
