// sdfhip — fused colour network (fields/sdf_field.py:532-612 get_colors, ref-nerf options off):
//   c_in = [ geo feature (NBF blocks, straight from the geometry kernel, TP) | small inputs (NBS blocks) ]
//   small inputs = x(3), NeRF-PE(view dir, 4 freqs, include_input)(27), RAW d sdf/dx (3), appearance embedding
//   NLC x (Linear + ReLU)  ->  Linear(3)  ->  sigmoid  ->  rgb * (1 + 2 pad) - pad
// Hidden layers run on the MFMA core; the 3-row output layer is a lane-local dot product.
#pragma once
#include "geo_kernels.h"

template <int NBF_, int NBS_, int NBC_, int NLC_>
struct ColDims {
  static constexpr int NBF = NBF_, NBS = NBS_, NBC = NBC_, NLC = NLC_;
  static constexpr int kb(int l) { return l == 0 ? NBF + NBS : NBC; }
  static constexpr int MAXB = (NBF + NBS) > NBC ? (NBF + NBS) : NBC;
  static constexpr int LDS_FLOATS = 2 * MAXB * 1024;
};

struct ColPtrs {
  const float* wp[kMaxLayers];    // packed W_l   [kb][NBC][16][64]
  const float* wpT[kMaxLayers];   // packed W_l^T [NBC][kb][16][64]
  const float* bias[kMaxLayers];  // padded natural order
  const float* w_out;             // [3][NBC*32]
  const float* b_out;             // [3]
  float rgb_padding;
};

struct ColFwdArgs {
  ColPtrs p;
  const float* feat_tp;    // [T][NBF]
  const float* csmall_tp;  // [T][NBS]
  float* h_tp[kMaxLayers]; // [T][NBC] post-ReLU activations (training only)
  float* rgb;              // [T*32][3]
};

template <class D, bool SAVE>
__global__ __launch_bounds__(256, 1) void col_fwd_kernel(const ColFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int MAXB = D::MAXB;

  f32x16 H[MAXB];
#pragma unroll
  for (int b = 0; b < D::NBF; ++b) H[b] = tp_load_blk(a.feat_tp, tile, D::NBF, b, lane);
#pragma unroll
  for (int b = 0; b < D::NBS; ++b) H[D::NBF + b] = tp_load_blk(a.csmall_tp, tile, D::NBS, b, lane);

  static_for<0, D::NLC>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value;
    constexpr int KB = D::kb(l);
    f32x16 acc[MAXB];
    tp_load_rowvec<D::NBC>(acc, a.p.bias[l], hf);
    tp_gemm<KB, D::NBC>(acc, H, a.p.wp[l], lds, tid, lane);
#pragma unroll
    for (int b = 0; b < D::NBC; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) H[b][r] = fmaxf(acc[b][r], 0.0f);
      if constexpr (SAVE) tp_store_blk(H[b], a.h_tp[l], tile, D::NBC, b, lane);
    }
  });

  float o[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float part = 0.0f;
    const float* w = a.p.w_out + c * (D::NBC * 32);
#pragma unroll
    for (int b = 0; b < D::NBC; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) part = fmaf(w[b * 32 + tp_row(r, hf)], H[b][r], part);
    part += __shfl_xor(part, 32);
    const float s = 1.0f / (1.0f + expf(-(part + a.p.b_out[c])));
    o[c] = s * (1.0f + 2.0f * a.p.rgb_padding) - a.p.rgb_padding;
  }
  if (hf == 0) {
    float* dst = a.rgb + (tile * 32 + lane) * 3;
    dst[0] = o[0];
    dst[1] = o[1];
    dst[2] = o[2];
  }
}

struct ColBwdArgs {
  ColPtrs p;
  const float* rgb;          // [T*32][3] forward output
  const float* rgbbar;       // [P][3]    upstream gradient (rows >= n_points read as zero)
  int64_t n_points;
  const float* h_tp[kMaxLayers];
  float* d_tp[kMaxLayers];   // [T][NBC] pre-activation gradients delta_l (for the weight-gradient GEMMs)
  float* dout_tp;            // [T][1]   delta of the 3-row output layer in rows 0..2
  float* featbar_tp;         // [T][NBF]
  float* csmallbar_tp;       // [T][NBS]
};

template <class D>
__global__ __launch_bounds__(256, 1) void col_bwd_kernel(const ColBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int MAXB = D::MAXB;

  const int64_t p = tile * 32 + (lane & 31);
  float dl[3];
  {
    const float pad = a.p.rgb_padding, k = 1.0f + 2.0f * pad;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float g = p < a.n_points ? a.rgbbar[p * 3 + c] : 0.0f;
      const float s = (a.rgb[p * 3 + c] + pad) / k;
      dl[c] = g * k * s * (1.0f - s);
    }
    // TP block with delta_out in feature rows 0..2 (reg 0..2 of the hf == 0 half)
    float* dst = a.dout_tp + (size_t)tile * 1024 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r * 64] = (hf == 0 && r < 3) ? dl[r] : 0.0f;
  }
  f32x16 hb[MAXB];
#pragma unroll
  for (int b = 0; b < D::NBC; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = b * 32 + tp_row(r, hf);
      hb[b][r] = a.p.w_out[k] * dl[0] + a.p.w_out[D::NBC * 32 + k] * dl[1] + a.p.w_out[2 * D::NBC * 32 + k] * dl[2];
    }

  static_for<0, D::NLC>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = D::NLC - 1 - decltype(lc)::value;
    constexpr int KB = D::kb(l);
#pragma unroll
    for (int b = 0; b < D::NBC; ++b) {
      const f32x16 h = tp_load_blk(a.h_tp[l], tile, D::NBC, b, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) hb[b][r] = h[r] > 0.0f ? hb[b][r] : 0.0f;
      tp_store_blk(hb[b], a.d_tp[l], tile, D::NBC, b, lane);
    }
    f32x16 un[MAXB];
#pragma unroll
    for (int b = 0; b < KB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) un[b][r] = 0.0f;
    tp_gemm<D::NBC, KB>(un, hb, a.p.wpT[l], lds, tid, lane);
    if constexpr (l == 0) {
#pragma unroll
      for (int b = 0; b < D::NBF; ++b) tp_store_blk(un[b], a.featbar_tp, tile, D::NBF, b, lane);
#pragma unroll
      for (int b = 0; b < D::NBS; ++b) tp_store_blk(un[D::NBF + b], a.csmallbar_tp, tile, D::NBS, b, lane);
    } else {
#pragma unroll
      for (int b = 0; b < D::NBC; ++b) hb[b] = un[b];
    }
  });
}
