// sdfhip — fused colour network (fields/sdf_field.py:532-612 get_colors, ref-nerf options off):
//   c_in = [ geo feature (NBF blocks, straight from the geometry kernel, TP) | small inputs (NBS blocks) ]
//   small inputs = x(3), NeRF-PE(view dir, 4 freqs, include_input)(27), RAW d sdf/dx (3), appearance embedding
//   NLC x (Linear + ReLU)  ->  Linear(3)  ->  sigmoid  ->  rgb * (1 + 2 pad) - pad
// Hidden layers run on the MFMA core (mlp_core.h: activations produced block by block from the previous layer's
// accumulators); the 3-row output layer is a lane-local dot product.  Like the geometry kernels (geo_kernels.h) the layers
// after the first are ONE instance of the hidden -> hidden gemm in a run-time loop: the depth NLC is a run-time property.
#pragma once
#include "geo_kernels.h"

template <int NBF_, int NBS_, int NBC_>
struct ColDims {
  static constexpr int NBF = NBF_, NBS = NBS_, NBC = NBC_;
  static constexpr int KB0 = NBF + NBS;  // input blocks of layer 0
  static constexpr int MAXO = KB0 > NBC ? KB0 : NBC;  // widest chunk: the first layer's transposed weights (KB0 out-blocks)
  static constexpr int buf_floats(int ns) { return chunk_pieces(MAXO, ns) * 256; }
  static constexpr int CW = NBC * 32;
  // LDS: two chunk buffers, then the biases of the NLC hidden layers and the 3 output rows
  static constexpr int lds_floats(int ns, int nlc) { return 2 * buf_floats(ns) + (nlc + 3) * CW; }
};
constexpr int kNsCol = kNsFwd;  // forward: fp32-class products (a 3-term bf16 forward flips ReLU branches at |z| ~ 1e-5)

struct ColPtrs {
  int32_t nlc, pad_;              // hidden layers (>= 1)
  const float* wp[kMaxLayers];    // packed W_l   [kb][parts][NBC][2][64] x 8
  const float* wpT[kMaxLayers];   // packed W_l^T [NBC][parts][kb][2][64] x 8
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
  constexpr int W = D::CW, NS = kNsCol, PCS = chunk_pieces(D::NBC, NS);
  const int NLC = a.p.nlc;
  float* cvec = lds + 2 * D::buf_floats(NS);

  WStream ws{lds, D::buf_floats(NS), 0, wave, lane};
  ws.issue(a.p.wp[0], PCS, true);
  for (int l = 0; l < NLC; ++l)
    if (tid < W) cvec[l * W + tid] = a.p.bias[l][tid];
  for (int i = tid; i < 3 * W; i += 256) cvec[NLC * W + i] = a.p.w_out[i];
  __syncthreads();

  f32x16 accIn[D::NBC], accOut[D::NBC];
  Raw carry;
  auto in_fetch = [&](auto kbc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kbc)::value;
    if constexpr (kb < D::NBF) return BlkSrc<1>{{tp_block_ptr(a.feat_tp, tile, D::NBF, kb)}};
    else return BlkSrc<1>{{tp_block_ptr(a.csmall_tp, tile, D::NBS, kb - D::NBF)}};
  };
  carry = load_src(in_fetch(IC<0>{}), lane);
  // layer 0: [feature | small inputs] -> hidden
  {
#pragma unroll
    for (int b = 0; b < D::NBC; ++b) accOut[b] = tp_rowvec_blk(cvec, b, hf);
    auto make = [&](auto, const Raw& raw, auto ec) __attribute__((always_inline)) { return raw.a[decltype(ec)::value]; };
    tp_gemm<D::KB0, D::NBC, Stores<0>, NS, PCS>(accOut, carry, in_fetch, make, NoFetch{}, ws, a.p.wp[0], a.p.wp[NLC > 1 ? 1 : 0]);
    acc_copy(accIn, accOut);
  }
  // layers 1 .. NLC-1: hidden -> hidden, input = relu of the layer below (saved on the way for the backward)
#pragma unroll 1
  for (int l = 1; l < NLC; ++l) {
    const float* bias = cvec + l * W;
#pragma unroll
    for (int b = 0; b < D::NBC; ++b) accOut[b] = tp_rowvec_blk(bias, b, hf);
    float* hprev = a.h_tp[l - 1];
    auto make = [&](auto kbc, const Raw&, auto ec) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
      const float h = act_h<1, NS == 4>(accIn[kb][e]);  // ReLU, bounded for fp16 operand parts by the same v_med3_f32
      if constexpr (SAVE) *tp_elem(hprev, tile, D::NBC, kb, e, lane) = h;
      return InRange{h};
    };
    tp_gemm<D::NBC, D::NBC, Stores<(SAVE ? 16 : 0)>, NS, PCS>(accOut, carry, NoFetch{}, make, NoFetch{}, ws, a.p.wp[l],
                                                              a.p.wp[l + 1 < NLC ? l + 1 : l]);
    acc_copy(accIn, accOut);
  }

  // last hidden activation + the 3-row output layer
  float part[3] = {0.f, 0.f, 0.f};
  float* hlast = a.h_tp[NLC - 1];
  const float* wout = cvec + NLC * W;
#pragma unroll
  for (int b = 0; b < D::NBC; ++b) {
    f32x16 h;
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = fmaxf(accIn[b][r], 0.0f);
    if constexpr (SAVE) tp_store_blk(h, hlast, tile, D::NBC, b, lane);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[c] = fmaf(wout[c * W + b * 32 + tp_row(r, hf)], h[r], part[c]);
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
  constexpr int W = D::CW, NS = kNsGrad;
  constexpr int PCS = chunk_pieces(D::MAXO, NS);  // prefetch size: the first layer's transposed chunk is the widest
  const int NLC = a.p.nlc;
  float* cvec = lds + 2 * D::buf_floats(NS);

  WStream ws{lds, D::buf_floats(NS), 0, wave, lane};
  ws.issue(a.p.wpT[NLC - 1], PCS, true);
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
  f32x16 accIn[D::NBC], accOut[D::NBC];
  // hbar of the last hidden layer = w_out^T delta_out
#pragma unroll
  for (int b = 0; b < D::NBC; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = b * 32 + tp_row(r, hf);
      accIn[b][r] = cvec[k] * dl[0] + cvec[W + k] * dl[1] + cvec[2 * W + k] * dl[2];
    }

  Raw carry;
  carry = load_src(BlkSrc<1>{{tp_block_ptr(a.h_tp[NLC - 1], tile, D::NBC, 0)}}, lane);
  // layers NLC-1 .. 1: delta_l = hbar_l masked by the ReLU (saved for the weight gradients); hbar_{l-1} = W_l^T delta_l
#pragma unroll 1
  for (int l = NLC - 1; l >= 1; --l) {
    const float* hl = a.h_tp[l];
    const float* hbelow = a.h_tp[l - 1];
    float* dlp = a.d_tp[l];
#pragma unroll
    for (int b = 0; b < D::NBC; ++b) accOut[b] = f32x16_zero();
    auto fetch = [&](auto bc) __attribute__((always_inline)) { return BlkSrc<1>{{tp_block_ptr(hl, tile, D::NBC, decltype(bc)::value)}}; };
    auto make = [&](auto bc, const Raw& raw, auto ec) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value, e = decltype(ec)::value;
      const float d = raw.a[e] > 0.0f ? accIn[b][e] : 0.0f;
      *tp_elem(dlp, tile, D::NBC, b, e, lane) = d;
      return d;
    };
    auto next_fetch = [&]() __attribute__((always_inline)) { return BlkSrc<1>{{tp_block_ptr(hbelow, tile, D::NBC, 0)}}; };
    tp_gemm<D::NBC, D::NBC, Stores<16>, NS, PCS>(accOut, carry, fetch, make, next_fetch, ws, a.p.wpT[l], a.p.wpT[l - 1]);
    acc_copy(accIn, accOut);
  }
  // layer 0: the gradient of [feature | small inputs]
  {
    f32x16 acc0[D::KB0];
#pragma unroll
    for (int b = 0; b < D::KB0; ++b) acc0[b] = f32x16_zero();
    const float* hl = a.h_tp[0];
    float* dlp = a.d_tp[0];
    auto fetch = [&](auto bc) __attribute__((always_inline)) { return BlkSrc<1>{{tp_block_ptr(hl, tile, D::NBC, decltype(bc)::value)}}; };
    auto make = [&](auto bc, const Raw& raw, auto ec) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value, e = decltype(ec)::value;
      const float d = raw.a[e] > 0.0f ? accIn[b][e] : 0.0f;
      *tp_elem(dlp, tile, D::NBC, b, e, lane) = d;
      return d;
    };
    tp_gemm<D::NBC, D::KB0, Stores<16>, NS, 0>(acc0, carry, fetch, make, NoFetch{}, ws, a.p.wpT[0], nullptr);
#pragma unroll
    for (int b = 0; b < D::NBF; ++b) tp_store_blk(acc0[b], a.featbar_tp, tile, D::NBF, b, lane);
#pragma unroll
    for (int b = 0; b < D::NBS; ++b) tp_store_blk(acc0[D::NBF + b], a.csmallbar_tp, tile, D::NBS, b, lane);
  }
}
