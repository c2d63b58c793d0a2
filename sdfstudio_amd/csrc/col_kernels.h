// sdfhip — fused colour network (fields/sdf_field.py:532-612 get_colors, ref-nerf options off):
//   c_in = [ geo feature (NBF blocks, straight from the geometry kernel, TP) | small inputs (NBS blocks) ]
//   small inputs = x(3), NeRF-PE(view dir, 4 freqs, include_input)(27), RAW d sdf/dx (3), appearance embedding
//   NLC x (Linear + ReLU)  ->  Linear(3)  ->  sigmoid  ->  rgb * (1 + 2 pad) - pad
// Hidden layers run on the MFMA core (mlp_core.h: activations produced block by block from the previous layer's
// accumulators); the 3-row output layer is a lane-local dot product.
#pragma once
#include "geo_kernels.h"

template <int NBF_, int NBS_, int NBC_, int NLC_>
struct ColDims {
  static constexpr int NBF = NBF_, NBS = NBS_, NBC = NBC_, NLC = NLC_;
  static constexpr int kb(int l) { return l == 0 ? NBF + NBS : NBC; }
  static constexpr int MAXB = (NBF + NBS) > NBC ? (NBF + NBS) : NBC;
  static constexpr int buf_floats(int ns) { return chunk_pieces(MAXB, ns) * 256; }
  static constexpr int CW = NBC * 32;
  static constexpr int CVEC_FLOATS = (NLC + 3) * CW;  // biases of the NLC hidden layers, then the 3 output rows
  static constexpr int lds_floats(int ns) { return 2 * buf_floats(ns) + CVEC_FLOATS; }
};
constexpr int kNsCol = kNsFwd;  // forward: 6-term products (a 3-term forward flips ReLU branches at |z| ~ 1e-5)

struct ColPtrs {
  const float* wp[kMaxLayers];    // packed W_l   [kb][3][NBC][2][64] x 8 bf16
  const float* wpT[kMaxLayers];   // packed W_l^T [NBC][3][kb][2][64] x 8 bf16
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
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int MAXB = D::MAXB, W = D::CW;
  constexpr int NS = kNsCol;
  float* cvec = lds + 2 * D::buf_floats(NS);

  WStream ws{lds, D::buf_floats(NS), 0, wave, lane};
  ws.issue(a.p.wp[0], chunk_pieces(D::NBC, NS), true);
  static_for<0, D::NLC>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value;
    if (tid < W) cvec[l * W + tid] = a.p.bias[l][tid];
  });
  for (int i = tid; i < 3 * W; i += 256) cvec[D::NLC * W + i] = a.p.w_out[i];
  __syncthreads();

  f32x16 accA[MAXB], accB[MAXB];
  Raw carry;
  auto in_fetch = [&](auto kbc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kbc)::value;
    if constexpr (kb < D::NBF) return BlkSrc<1>{{tp_block_ptr(a.feat_tp, tile, D::NBF, kb)}};
    else return BlkSrc<1>{{tp_block_ptr(a.csmall_tp, tile, D::NBS, kb - D::NBF)}};
  };
  carry = load_src(in_fetch(IC<0>{}), lane);
  static_for<0, D::NLC>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value;
    constexpr int KB = D::kb(l);
    auto& in = pick<(l % 2) == 0>(accA, accB);
    auto& out = pick<(l % 2) == 0>(accB, accA);
#pragma unroll
    for (int b = 0; b < D::NBC; ++b) out[b] = tp_rowvec_blk(cvec + l * W, b, hf);
    auto fetch = [&](auto kbc) __attribute__((always_inline)) {
      if constexpr (l == 0) return in_fetch(kbc);
      else return BlkSrc<0>{};
    };
    auto make = [&](auto kbc, const Raw& raw, auto ec) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
      if constexpr (l == 0) {
        return raw.a[e];
      } else {
        const float h = fmaxf(in[kb][e], 0.0f);
        if constexpr (SAVE) *tp_elem(a.h_tp[l > 0 ? l - 1 : 0], tile, D::NBC, kb, e, lane) = h;
        return h;
      }
    };
    tp_gemm<KB, D::NBC, Stores<((l > 0 && SAVE) ? 16 : 0)>, NS, (l + 1 < D::NLC ? chunk_pieces(D::NBC, NS) : 0)>(
        out, carry, fetch, make, NoFetch{}, ws, a.p.wp[l], l + 1 < D::NLC ? a.p.wp[l + 1 < D::NLC ? l + 1 : l] : nullptr);
  });

  // last hidden activation + the 3-row output layer
  auto& z = pick<(D::NLC % 2) == 0>(accA, accB);
  float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < D::NBC; ++b) {
    f32x16 h;
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = fmaxf(z[b][r], 0.0f);
    if constexpr (SAVE) tp_store_blk(h, a.h_tp[D::NLC - 1], tile, D::NBC, b, lane);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[c] = fmaf(cvec[(D::NLC + c) * W + b * 32 + tp_row(r, hf)], h[r], part[c]);
  }
  float o[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t = part[c] + __shfl_xor(part[c], 32);
    const float s = 1.0f / (1.0f + expf(-(t + a.p.b_out[c])));
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
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int MAXB = D::MAXB, W = D::CW;
  constexpr int NS = kNsGrad;
  float* cvec = lds + 2 * D::buf_floats(NS);

  WStream ws{lds, D::buf_floats(NS), 0, wave, lane};
  ws.issue(a.p.wpT[D::NLC - 1], chunk_pieces(D::kb(D::NLC - 1), NS), true);
  for (int i = tid; i < 3 * W; i += 256) cvec[i] = a.p.w_out[i];
  __syncthreads();

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
  f32x16 accA[MAXB], accB[MAXB];
  // hbar of the last hidden layer = w_out^T delta_out
#pragma unroll
  for (int b = 0; b < D::NBC; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = b * 32 + tp_row(r, hf);
      accA[b][r] = cvec[k] * dl[0] + cvec[W + k] * dl[1] + cvec[2 * W + k] * dl[2];
    }

  Raw carry;
  auto h_fetch = [&](auto lc, auto bc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value, b = decltype(bc)::value;
    return BlkSrc<1>{{tp_block_ptr(a.h_tp[l], tile, D::NBC, b)}};
  };
  carry = load_src(h_fetch(IC<D::NLC - 1>{}, IC<0>{}), lane);
  static_for<0, D::NLC>([&](auto sc) __attribute__((always_inline)) {
    constexpr int step = decltype(sc)::value;
    constexpr int l = D::NLC - 1 - step;
    constexpr int KB = D::kb(l);
    auto& hb = pick<(step % 2) == 0>(accA, accB);
    auto& un = pick<(step % 2) == 0>(accB, accA);
#pragma unroll
    for (int b = 0; b < KB; ++b) un[b] = f32x16_zero();
    // delta_l = hbar_l masked by the ReLU
    auto fetch = [&](auto bc) __attribute__((always_inline)) { return h_fetch(IC<l>{}, bc); };
    auto make = [&](auto bc, const Raw& raw, auto ec) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value, e = decltype(ec)::value;
      const float d = raw.a[e] > 0.0f ? hb[b][e] : 0.0f;
      *tp_elem(a.d_tp[l], tile, D::NBC, b, e, lane) = d;
      return d;
    };
    auto next_fetch = [&]() __attribute__((always_inline)) {
      if constexpr (l > 0) return h_fetch(IC<(l > 0 ? l - 1 : 0)>{}, IC<0>{});
      else return BlkSrc<0>{};
    };
    tp_gemm<D::NBC, KB, Stores<16>, NS, (l > 0 ? chunk_pieces(D::kb(l > 0 ? l - 1 : 0), NS) : 0)>(
        un, carry, fetch, make, next_fetch, ws, a.p.wpT[l], l > 0 ? a.p.wpT[l > 0 ? l - 1 : 0] : nullptr);
    if constexpr (l == 0) {
#pragma unroll
      for (int b = 0; b < D::NBF; ++b) tp_store_blk(un[b], a.featbar_tp, tile, D::NBF, b, lane);
#pragma unroll
      for (int b = 0; b < D::NBS; ++b) tp_store_blk(un[D::NBF + b], a.csmallbar_tp, tile, D::NBS, b, lane);
    }
  });
}
