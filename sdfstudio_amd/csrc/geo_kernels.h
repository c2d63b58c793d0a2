// sdfhip — fused geometry network (SDF head): forward + analytic d sdf / d in0 chain, and the backward
// (tangent pass for the second-order terms + data backward).  Restates fields/sdf_field.py:380-410
// (forward_geonetwork) and the autograd.grad(..., create_graph=True) at :646-654 as explicit kernels.
//
// Per point (lane-owned, TP layout), with u_l the input of layer l, s = softplus(beta=100):
//   forward   z_l = W_l u_l + b_l ;  u_{l+1} = s(z_l) ;  u_SKIP = [s(z_{SKIP-1}) | in0] / sqrt(2) (1/sqrt2 folded into W_SKIP)
//             feat = W_f u_NL + b_f ; sdf = w_s . u_NL + b_s
//   chain     q_NL = w_s ; r_l = q_{l+1} * s'(z_l) ; q_l = W_l^T r_l ;   e = d sdf / d in0 = q_0 (+ skip part)
//   tangent   (backward of the chain == forward-mode with tangent ebar = J_in0 gbar)
//             v_l = W_l qb_l ; qb_{l+1} = s'(z_l) v_l ; zc_l = v_l * r_l * 100 (1 - s'(z_l))
//   backward  ub_NL = W_f^T featbar + w_s sdfbar ; zb_l = ub_{l+1} s'(z_l) + zc_l ; ub_l = W_l^T zb_l
//   weight gradients are separate split-K GEMMs (wgrad_kernel): Wb_l = Zb_l^T U_l + R_l^T Qb_l.
//
// Every line above is one tp_gemm (mlp_core.h) whose input blocks are produced just in time from the previous gemm's
// accumulators; the elementwise work (softplus, its derivatives, loads / stores of the saved tensors) rides in the
// producer, interleaved with the MFMAs of the previous block.
#pragma once
#include "mlp_core.h"

constexpr int kMaxLayers = 10;
// Precision modes (mlp_core.h).  Everything the forward call returns (sdf, geo feature, d sdf / dx, rgb) carries the parity
// targets and the raw d sdf / dx feeds the colour network's ReLUs: fp32-class products, by default as fp16 hi + lo parts with
// 3 terms (mode 4; -DSDFHIP_NS_FWD=3 selects the 6-term bf16 form, twice the matrix instructions for the last two mantissa
// bits); the backward kernels (tangent pass, data backward, colour backward) use 3-term bf16 products (mode 2).
#ifndef SDFHIP_NS_FWD
#define SDFHIP_NS_FWD 4
#endif
#ifndef SDFHIP_NS_GRAD
#define SDFHIP_NS_GRAD 2
#endif
constexpr int kNsFwd = SDFHIP_NS_FWD, kNsGrad = SDFHIP_NS_GRAD;
constexpr int kNsMax = ns_parts(kNsFwd) > ns_parts(kNsGrad) ? kNsFwd : kNsGrad;  // the mode with the larger weight chunks

template <int NBH_, int NB0_, int NB3_, int NL_, int SKIP_, int NBF_>
struct GeoDims {
  static constexpr int NBH = NBH_, NB0 = NB0_, NB3 = NB3_, NL = NL_, SKIP = SKIP_, NBF = NBF_;
  // input blocks of layer l (l in [0, NL]; l == NL is the output layer)
  static constexpr int kb(int l) { return l == 0 ? NB0 : (l == SKIP ? NB3 + NB0 : NBH); }
  // output blocks of layer l
  static constexpr int nbo(int l) { return l == NL ? NBF : ((l + 1 == SKIP) ? NB3 : NBH); }
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  static constexpr int MAXB = cmax(cmax(NBH, NBF), cmax(NB0, SKIP >= 0 ? NB3 + NB0 : 0));
  static constexpr int buf_floats(int ns) { return chunk_pieces(MAXB, ns) * 256; }  // one weight chunk buffer
  static constexpr int CW = cmax(NBH, NBF) * 32;                         // stride of the constant-vector area
  static constexpr int CVEC_FLOATS = (NL + 2) * CW;                      // biases of layers 0..NL, then w_sdf
  static constexpr int lds_floats(int ns) { return 2 * buf_floats(ns) + CVEC_FLOATS; }
};

struct GeoPtrs {
  const float* wp[kMaxLayers];    // packed W_l      [kb][3][nbo][2][64] x 8 bf16
  const float* wpT[kMaxLayers];   // packed W_l^T    [nbo][3][kb][2][64] x 8 bf16
  const float* bias[kMaxLayers];  // natural order, padded to nbo*32
  const float* w_sdf;             // [NBH*32]  row of the output layer that produces sdf
  const float* b_sdf;             // [1]
};

struct GeoFwdArgs {
  GeoPtrs p;
  const float* in0_tp;  // [T][NB0]
  float* z_tp[kMaxLayers];  // [T][nbo(l)]   (training only)
  float* r_tp[kMaxLayers];  // [T][nbo(l)]   (training only)
  float* feat_tp;           // [T][NBF]
  float* sdf;               // [T*32]
  float* e_tp;              // [T][NB0]   d sdf / d in0
};

// biases (layers 0..NL) and w_sdf into the constant area of LDS: cvec[l * CW + i], w_sdf at l = NL + 1
template <class D>
SDFHIP_D void geo_stage_cvec(float* cvec, const GeoPtrs& p, const int tid) {
  static_for<0, D::NL + 1>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value;
    if (tid < D::nbo(l) * 32) cvec[l * D::CW + tid] = p.bias[l][tid];
  });
  if (tid < D::NBH * 32) cvec[(D::NL + 1) * D::CW + tid] = p.w_sdf[tid];
}

template <class D, bool GRAD, bool SAVE, bool FEAT>
__global__ __launch_bounds__(256, 1) void geo_fwd_kernel(const GeoFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int MAXB = D::MAXB, W = D::CW;
  constexpr int NSF = kNsFwd, NSC = kNsFwd, NSB = kNsFwd;
  float* cvec = lds + 2 * D::buf_floats(NSB);

  WStream ws{lds, D::buf_floats(NSB), 0, wave, lane};
  ws.issue(a.p.wp[0], chunk_pieces(D::nbo(0), NSF), true);
  geo_stage_cvec<D>(cvec, a.p, tid);
  __syncthreads();

  f32x16 accA[MAXB], accB[MAXB];
  Raw carry;

  // HBM operands of input block kb of forward layer l (only in0 blocks come from memory)
  auto fwd_fetch = [&](auto lc, auto kbc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value, kb = decltype(kbc)::value;
    if constexpr (l == 0) return BlkSrc<1>{{tp_block_ptr(a.in0_tp, tile, D::NB0, kb)}};
    else if constexpr (l == D::SKIP && kb >= D::NB3) return BlkSrc<1>{{tp_block_ptr(a.in0_tp, tile, D::NB0, kb - D::NB3)}};
    else return BlkSrc<0>{};
  };
  // HBM operands of block b of the chain step through layer l
  auto chain_fetch = [&](auto lc, auto bc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value, b = decltype(bc)::value;
    return BlkSrc<1>{{tp_block_ptr(a.z_tp[l], tile, D::nbo(l), b)}};
  };

  // ---- forward layers 0 .. NL-1: out_l = b_l + W_l u_l
  carry = load_src(fwd_fetch(IC<0>{}, IC<0>{}), lane);
  static_for<0, D::NL>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value;
    constexpr int KB = D::kb(l), NBO = D::nbo(l);
    auto& in = pick<(l % 2) == 0>(accA, accB);   // accumulators of layer l - 1 (l >= 1)
    auto& out = pick<(l % 2) == 0>(accB, accA);
#pragma unroll
    for (int b = 0; b < NBO; ++b) out[b] = tp_rowvec_blk(cvec + l * W, b, hf);
    auto fetch = [&](auto kbc) __attribute__((always_inline)) { return fwd_fetch(lc, kbc); };
    auto make = [&](auto kbc, const Raw& raw, auto ec) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
      if constexpr (l == 0 || (l == D::SKIP && kb >= D::NB3)) {
        return raw.a[e];
      } else {
        const float z = in[kb][e];
        if constexpr (SAVE || GRAD) *tp_elem(a.z_tp[l > 0 ? l - 1 : 0], tile, D::nbo(l > 0 ? l - 1 : 0), kb, e, lane) = z;
        return softplus100_h(z);
      }
    };
    constexpr bool last = l + 1 == D::NL;
    const float* next = last ? (FEAT ? a.p.wp[D::NL] : (GRAD ? a.p.wpT[D::NL - 1] : nullptr)) : a.p.wp[l + 1];
    constexpr int next_pieces = last ? (FEAT ? chunk_pieces(D::NBF, NSF) : (GRAD ? chunk_pieces(D::kb(D::NL - 1), NSC) : 0)) : chunk_pieces(D::nbo(l + 1), NSF);
    auto next_fetch = [&]() __attribute__((always_inline)) {
      if constexpr (!last) return fwd_fetch(IC<(last ? l : l + 1)>{}, IC<0>{});
      else return BlkSrc<0>{};
    };
    constexpr int ZS = (SAVE || GRAD) ? 16 : 0;  // z stores per produced block
    using ST = Stores<(l == 0 ? 0 : ZS), (l == 0 ? 0 : (l == D::SKIP ? 0 : ZS)), (l == D::SKIP ? D::NB3 : (1 << 30))>;
    tp_gemm<KB, NBO, ST, NSF, next_pieces>(out, carry, fetch, make, next_fetch, ws, a.p.wp[l], next);
  });

  // ---- output layer: the sdf row as a lane-local dot product riding in the producer, feature rows on the MFMA path
  {
    auto& in = pick<(D::NL % 2) == 0>(accA, accB);
    auto& out = pick<(D::NL % 2) == 0>(accB, accA);
    float part = 0.0f;
    auto make = [&](auto kbc, const Raw&, auto ec) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
      const float z = in[kb][e];
      if constexpr (SAVE || GRAD) *tp_elem(a.z_tp[D::NL - 1], tile, D::NBH, kb, e, lane) = z;
      const float h = softplus100_h(z);
      part = fmaf(cvec[(D::NL + 1) * W + kb * 32 + tp_row(e, hf)], h, part);
      return h;
    };
    if constexpr (FEAT) {
#pragma unroll
      for (int b = 0; b < D::NBF; ++b) out[b] = tp_rowvec_blk(cvec + D::NL * W, b, hf);
      auto next_fetch = [&]() __attribute__((always_inline)) {
        if constexpr (GRAD) return chain_fetch(IC<D::NL - 1>{}, IC<0>{});
        else return BlkSrc<0>{};
      };
      tp_gemm<D::NBH, D::NBF, Stores<((SAVE || GRAD) ? 16 : 0)>, NSF, (GRAD ? chunk_pieces(D::kb(D::NL - 1), NSC) : 0)>(
          out, carry, NoFetch{}, make, next_fetch, ws, a.p.wp[D::NL], GRAD ? a.p.wpT[D::NL - 1] : nullptr);
#pragma unroll
      for (int b = 0; b < D::NBF; ++b) tp_store_blk(out[b], a.feat_tp, tile, D::NBF, b, lane);
    } else {
      static_for<0, D::NBH>([&](auto kbc) __attribute__((always_inline)) {
        static_for<0, 16>([&](auto ec) __attribute__((always_inline)) { (void)make(kbc, carry, ec); });
      });
      if constexpr (GRAD) carry = load_src(chain_fetch(IC<D::NL - 1>{}, IC<0>{}), lane);
    }
    part += __shfl_xor(part, 32);
    if (hf == 0) a.sdf[tile * 32 + lane] = part + a.p.b_sdf[0];
  }

  // ---- chain: q_l = W_l^T (q_{l+1} * s'(z_l)),  q_NL = w_sdf
  if constexpr (GRAD) {
#pragma unroll
    for (int b = 0; b < D::NBH; ++b) accA[b] = tp_rowvec_blk(cvec + (D::NL + 1) * W, b, hf);
    static_for<0, D::NL>([&](auto sc) __attribute__((always_inline)) {
      constexpr int step = decltype(sc)::value;
      constexpr int l = D::NL - 1 - step;
      constexpr int KB = D::kb(l), NBO = D::nbo(l);
      auto& q = pick<(step % 2) == 0>(accA, accB);
      auto& qn = pick<(step % 2) == 0>(accB, accA);
#pragma unroll
      for (int b = 0; b < KB; ++b) qn[b] = f32x16_zero();
      auto fetch = [&](auto bc) __attribute__((always_inline)) { return chain_fetch(IC<l>{}, bc); };
      auto make = [&](auto bc, const Raw& raw, auto ec) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value, e = decltype(ec)::value;
        const float r = q[b][e] * softplus100_d1(raw.a[e]);
        if constexpr (SAVE) *tp_elem(a.r_tp[l], tile, NBO, b, e, lane) = r;
        return r;
      };
      auto next_fetch = [&]() __attribute__((always_inline)) {
        if constexpr (l > 0) return chain_fetch(IC<(l > 0 ? l - 1 : 0)>{}, IC<0>{});
        else return BlkSrc<0>{};
      };
      tp_gemm<NBO, KB, Stores<(SAVE ? 16 : 0)>, NSC, (l > 0 ? chunk_pieces(D::kb(l > 0 ? l - 1 : 0), NSC) : 0)>(
          qn, carry, fetch, make, next_fetch, ws, a.p.wpT[l], l > 0 ? a.p.wpT[l > 0 ? l - 1 : 0] : nullptr);
      if constexpr (l == D::SKIP) {
        // the part of d sdf / d (layer input) that goes straight to in0: park it in e_tp, layer 0 adds to it
#pragma unroll
        for (int b = 0; b < D::NB0; ++b) tp_store_blk(qn[D::NB3 + b], a.e_tp, tile, D::NB0, b, lane);
      }
      if constexpr (l == 0) {
#pragma unroll
        for (int b = 0; b < D::NB0; ++b) {
          if constexpr (D::SKIP > 0) qn[b] += tp_load_blk(a.e_tp, tile, D::NB0, b, lane);
          tp_store_blk(qn[b], a.e_tp, tile, D::NB0, b, lane);
        }
      }
    });
  }
}

struct GeoBwdArgs {
  GeoPtrs p;
  const float* ebar_tp;     // [T][NB0]   tangent seed  J_in0 gbar
  const float* featbar_tp;  // [T][NBF]
  const float* sdfbar;      // [T*32]
  const float* z_tp[kMaxLayers];
  const float* r_tp[kMaxLayers];
  float* qb_tp[kMaxLayers + 1];  // [T][kb(l)]  tangent entering layer l  (l == NL: the tangent reaching the sdf row)
  float* zb_tp[kMaxLayers];      // [T][nbo(l)] holds zc_l after the tangent pass, zbar_l after the backward pass
  float* in0bar_tp;              // [T][NB0]
};

// TANGENT = false: first-order backward only (no second-order terms: the caller differentiated sdf / feature, not d sdf / dx).
template <class D, bool TANGENT = true>
__global__ __launch_bounds__(256, 1) void geo_bwd_kernel(const GeoBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int MAXB = D::MAXB;
  constexpr int NS = kNsGrad;
  float* cvec = lds + 2 * D::buf_floats(NS);

  WStream ws{lds, D::buf_floats(NS), 0, wave, lane};
  if constexpr (TANGENT) ws.issue(a.p.wp[0], chunk_pieces(D::nbo(0), NS), true);
  else ws.issue(a.p.wpT[D::NL], chunk_pieces(D::NBH, NS), true);
  if (tid < D::NBH * 32) cvec[tid] = a.p.w_sdf[tid];
  __syncthreads();

  f32x16 accA[MAXB], accB[MAXB];
  Raw carry;

  // HBM operands of input block kb of tangent layer l: the seed blocks, or (z, r) of the layer below
  auto tan_fetch = [&](auto lc, auto kbc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value, kb = decltype(kbc)::value;
    if constexpr (l == 0) {
      return BlkSrc<1>{{tp_block_ptr(a.ebar_tp, tile, D::NB0, kb)}};  // qb_0 == ebar (already in HBM)
    } else if constexpr (l == D::SKIP && kb >= D::NB3) {
      return BlkSrc<1>{{tp_block_ptr(a.ebar_tp, tile, D::NB0, kb - D::NB3)}};
    } else {
      return BlkSrc<2>{{tp_block_ptr(a.z_tp[l > 0 ? l - 1 : 0], tile, D::nbo(l > 0 ? l - 1 : 0), kb),
                        tp_block_ptr(a.r_tp[l > 0 ? l - 1 : 0], tile, D::nbo(l > 0 ? l - 1 : 0), kb)}};
    }
  };
  // tangent epilogue of layer l on element e of block b:  qb_{l+1} = s'(z_l) v ;  zc_l = v r_l 100 (1 - s'(z_l))  (-> zb_tp[l])
  auto tangent_elem = [&](auto lc, auto bc, auto ec, const float v, const Raw& raw) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value, b = decltype(bc)::value, e = decltype(ec)::value;
    const float d1 = softplus100_d1(raw.a[e]);
    *tp_elem(a.zb_tp[l], tile, D::nbo(l), b, e, lane) = v * raw.b[e] * (100.0f * (1.0f - d1));
    return d1 * v;
  };
  auto bwd_fetch = [&](auto lc, auto bc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value, b = decltype(bc)::value;
    if constexpr (TANGENT) return BlkSrc<2>{{tp_block_ptr(a.z_tp[l], tile, D::nbo(l), b), tp_block_ptr(a.zb_tp[l], tile, D::nbo(l), b)}};
    else return BlkSrc<1>{{tp_block_ptr(a.z_tp[l], tile, D::nbo(l), b)}};
  };

  if constexpr (TANGENT) {
    // ---- tangent pass (second-order terms): v_l = W_l qb_l
    carry = load_src(tan_fetch(IC<0>{}, IC<0>{}), lane);
    static_for<0, D::NL>([&](auto lc) __attribute__((always_inline)) {
      constexpr int l = decltype(lc)::value;
      constexpr int KB = D::kb(l), NBO = D::nbo(l);
      auto& in = pick<(l % 2) == 0>(accA, accB);  // v_{l-1}
      auto& out = pick<(l % 2) == 0>(accB, accA);
#pragma unroll
      for (int b = 0; b < NBO; ++b) out[b] = f32x16_zero();
      auto fetch = [&](auto kbc) __attribute__((always_inline)) { return tan_fetch(lc, kbc); };
      auto make = [&](auto kbc, const Raw& raw, auto ec) __attribute__((always_inline)) {
        constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
        if constexpr (l == 0) {
          return raw.a[e];
        } else if constexpr (l == D::SKIP && kb >= D::NB3) {
          *tp_elem(a.qb_tp[l], tile, KB, kb, e, lane) = raw.a[e];
          return raw.a[e];
        } else {
          const float qn = tangent_elem(IC<(l > 0 ? l - 1 : 0)>{}, kbc, ec, in[kb][e], raw);
          *tp_elem(a.qb_tp[l], tile, KB, kb, e, lane) = qn;
          return qn;
        }
      };
      constexpr bool last = l + 1 == D::NL;
      auto next_fetch = [&]() __attribute__((always_inline)) {
        if constexpr (!last) return tan_fetch(IC<(last ? l : l + 1)>{}, IC<0>{});
        else return BlkSrc<0>{};
      };
      // stores per produced block: zc + qb (32) for blocks computed from the layer below, qb only (16) for the seed blocks of
      // the skip layer, none for layer 0
      using ST = Stores<(l == 0 ? 0 : 32), (l == 0 ? 0 : (l == D::SKIP ? 16 : 32)), (l == D::SKIP ? D::NB3 : (1 << 30))>;
      tp_gemm<KB, NBO, ST, NS, chunk_pieces(last ? D::NBH : D::nbo(last ? l : l + 1), NS)>(
          out, carry, fetch, make, next_fetch, ws, a.p.wp[l], last ? a.p.wpT[D::NL] : a.p.wp[last ? l : l + 1]);
    });
    {
      // epilogue of the last hidden layer: qb_NL (tangent reaching the sdf row; only the weight gradient needs it) and zc_{NL-1}
      auto& v = pick<(D::NL % 2) == 0>(accA, accB);
      static_for<0, D::NBH>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        const Raw raw = load_src(tan_fetch(IC<D::NL>{}, bc), lane);
        static_for<0, 16>([&](auto ec) __attribute__((always_inline)) {
          constexpr int e = decltype(ec)::value;
          *tp_elem(a.qb_tp[D::NL], tile, D::NBH, b, e, lane) = tangent_elem(IC<D::NL - 1>{}, bc, ec, v[b][e], raw);
        });
      });
    }

  }

  // ---- backward pass: ub_NL = w_s sdfbar + W_f^T featbar
  {
    const float sb = a.sdfbar[tile * 32 + (lane & 31)];
#pragma unroll
    for (int b = 0; b < D::NBH; ++b) {
      const f32x16 w = tp_rowvec_blk(cvec, b, hf);
#pragma unroll
      for (int i = 0; i < 16; ++i) accA[b][i] = w[i] * sb;
    }
    auto fetch = [&](auto bc) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value;
      return BlkSrc<1>{{tp_block_ptr(a.featbar_tp, tile, D::NBF, b)}};
    };
    auto make = [&](auto, const Raw& raw, auto ec) __attribute__((always_inline)) { return raw.a[decltype(ec)::value]; };
    auto next_fetch = [&]() __attribute__((always_inline)) { return bwd_fetch(IC<D::NL - 1>{}, IC<0>{}); };
    carry = load_src(fetch(IC<0>{}), lane);
    tp_gemm<D::NBF, D::NBH, Stores<0>, NS, chunk_pieces(D::kb(D::NL - 1), NS)>(accA, carry, fetch, make, next_fetch, ws, a.p.wpT[D::NL],
                                                                               a.p.wpT[D::NL - 1]);
  }
  static_for<0, D::NL>([&](auto sc) __attribute__((always_inline)) {
    constexpr int step = decltype(sc)::value;
    constexpr int l = D::NL - 1 - step;
    constexpr int KB = D::kb(l), NBO = D::nbo(l);
    auto& ub = pick<(step % 2) == 0>(accA, accB);
    auto& un = pick<(step % 2) == 0>(accB, accA);
#pragma unroll
    for (int b = 0; b < KB; ++b) un[b] = f32x16_zero();
    // zb_l = ub * s'(z_l) + zc_l
    auto fetch = [&](auto bc) __attribute__((always_inline)) { return bwd_fetch(IC<l>{}, bc); };
    auto make = [&](auto bc, const Raw& raw, auto ec) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value, e = decltype(ec)::value;
      const float zb = TANGENT ? fmaf(ub[b][e], softplus100_d1(raw.a[e]), raw.b[e]) : ub[b][e] * softplus100_d1(raw.a[e]);
      *tp_elem(a.zb_tp[l], tile, NBO, b, e, lane) = zb;
      return zb;
    };
    auto next_fetch = [&]() __attribute__((always_inline)) {
      if constexpr (l > 0) return bwd_fetch(IC<(l > 0 ? l - 1 : 0)>{}, IC<0>{});
      else return BlkSrc<0>{};
    };
    tp_gemm<NBO, KB, Stores<16>, NS, (l > 0 ? chunk_pieces(D::kb(l > 0 ? l - 1 : 0), NS) : 0)>(
        un, carry, fetch, make, next_fetch, ws, a.p.wpT[l], l > 0 ? a.p.wpT[l > 0 ? l - 1 : 0] : nullptr);
    if constexpr (l == D::SKIP) {
#pragma unroll
      for (int b = 0; b < D::NB0; ++b) tp_store_blk(un[D::NB3 + b], a.in0bar_tp, tile, D::NB0, b, lane);
    }
    if constexpr (l == 0) {
#pragma unroll
      for (int b = 0; b < D::NB0; ++b) {
        if constexpr (D::SKIP > 0) un[b] += tp_load_blk(a.in0bar_tp, tile, D::NB0, b, lane);
        tp_store_blk(un[b], a.in0bar_tp, tile, D::NB0, b, lane);
      }
    }
  });
}
