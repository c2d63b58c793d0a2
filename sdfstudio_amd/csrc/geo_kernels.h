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
#pragma once
#include "mlp_core.h"

constexpr int kMaxLayers = 10;

template <int NBH_, int NB0_, int NB3_, int NL_, int SKIP_, int NBF_>
struct GeoDims {
  static constexpr int NBH = NBH_, NB0 = NB0_, NB3 = NB3_, NL = NL_, SKIP = SKIP_, NBF = NBF_;
  // input blocks of layer l (l in [0, NL]; l == NL is the output layer)
  static constexpr int kb(int l) { return l == 0 ? NB0 : (l == SKIP ? NB3 + NB0 : NBH); }
  // output blocks of layer l
  static constexpr int nbo(int l) { return l == NL ? NBF : ((l + 1 == SKIP) ? NB3 : NBH); }
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  static constexpr int MAXB = cmax(cmax(NBH, NBF), cmax(NB0, SKIP >= 0 ? NB3 + NB0 : 0));
  static constexpr int LDS_FLOATS = 2 * MAXB * 1024;
};

struct GeoPtrs {
  const float* wp[kMaxLayers];    // packed W_l      [kb][nbo][16][64]
  const float* wpT[kMaxLayers];   // packed W_l^T    [nbo][kb][16][64]
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

// one TP block (16 registers) load / store
SDFHIP_D f32x16 tp_load_blk(const float* __restrict__ base, const int64_t tile, const int nb, const int b, const int lane) {
  const float* p = base + ((size_t)tile * nb + b) * 1024 + lane;
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = p[r * 64];
  return v;
}
SDFHIP_D void tp_store_blk(const f32x16 v, float* __restrict__ base, const int64_t tile, const int nb, const int b, const int lane) {
  float* p = base + ((size_t)tile * nb + b) * 1024 + lane;
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r * 64] = v[r];
}

template <class D, bool GRAD, bool SAVE, bool FEAT>
__global__ __launch_bounds__(256, 1) void geo_fwd_kernel(const GeoFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int MAXB = D::MAXB;

  f32x16 H[MAXB];
  tp_load<D::NB0>(H, a.in0_tp, tile, lane);

  static_for<0, D::NL>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value;
    constexpr int KB = D::kb(l), NBO = D::nbo(l);
    if constexpr (l == D::SKIP) {
#pragma unroll
      for (int b = 0; b < D::NB0; ++b) H[D::NB3 + b] = tp_load_blk(a.in0_tp, tile, D::NB0, b, lane);
    }
    f32x16 acc[MAXB];
    tp_load_rowvec<NBO>(acc, a.p.bias[l], hf);
    tp_gemm<KB, NBO>(acc, H, a.p.wp[l], lds, tid, lane);
#pragma unroll
    for (int b = 0; b < NBO; ++b) {
      if constexpr (SAVE || GRAD) tp_store_blk(acc[b], a.z_tp[l], tile, NBO, b, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        H[b][r] = softplus100_h(acc[b][r]);
      }
    }
  });

  // output layer: feature rows through the MFMA path, the sdf row as a lane-local dot product
  {
    float part = 0.0f;
#pragma unroll
    for (int b = 0; b < D::NBH; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) part = fmaf(a.p.w_sdf[b * 32 + tp_row(r, hf)], H[b][r], part);
    part += __shfl_xor(part, 32);
    if (hf == 0) a.sdf[tile * 32 + lane] = part + a.p.b_sdf[0];
  }
  if constexpr (FEAT) {
    f32x16 acc[MAXB];
    tp_load_rowvec<D::NBF>(acc, a.p.bias[D::NL], hf);
    tp_gemm<D::NBH, D::NBF>(acc, H, a.p.wp[D::NL], lds, tid, lane);
    tp_store<D::NBF>(acc, a.feat_tp, tile, lane);
  }

  if constexpr (GRAD) {
    f32x16 q[MAXB];
    tp_load_rowvec<D::NBH>(q, a.p.w_sdf, hf);
    static_for<0, D::NL>([&](auto lc) __attribute__((always_inline)) {
      constexpr int l = D::NL - 1 - decltype(lc)::value;
      constexpr int KB = D::kb(l), NBO = D::nbo(l);
      // r_l = q * s'(z_l)   (in place in q)
#pragma unroll
      for (int b = 0; b < NBO; ++b) {
        const f32x16 z = tp_load_blk(a.z_tp[l], tile, NBO, b, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) q[b][r] *= softplus100_d1(z[r]);
        if constexpr (SAVE) tp_store_blk(q[b], a.r_tp[l], tile, NBO, b, lane);
      }
      f32x16 qn[MAXB];
#pragma unroll
      for (int b = 0; b < KB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) qn[b][r] = 0.0f;
      tp_gemm<NBO, KB>(qn, q, a.p.wpT[l], lds, tid, lane);
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
      } else {
        constexpr int NC = D::nbo(l - 1);
#pragma unroll
        for (int b = 0; b < NC; ++b) q[b] = qn[b];
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

template <class D>
__global__ __launch_bounds__(256, 1) void geo_bwd_kernel(const GeoBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int MAXB = D::MAXB;

  // ---- tangent pass (second-order terms)
  {
    f32x16 qb[MAXB];
    tp_load<D::NB0>(qb, a.ebar_tp, tile, lane);
    static_for<0, D::NL>([&](auto lc) __attribute__((always_inline)) {
      constexpr int l = decltype(lc)::value;
      constexpr int KB = D::kb(l), NBO = D::nbo(l);
      if constexpr (l == D::SKIP) {
#pragma unroll
        for (int b = 0; b < D::NB0; ++b) qb[D::NB3 + b] = tp_load_blk(a.ebar_tp, tile, D::NB0, b, lane);
      }
      if constexpr (l > 0) tp_store<KB>(qb, a.qb_tp[l], tile, lane);  // (l == 0: qb_0 == ebar, already in HBM)
      f32x16 v[MAXB];
#pragma unroll
      for (int b = 0; b < NBO; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[b][r] = 0.0f;
      tp_gemm<KB, NBO>(v, qb, a.p.wp[l], lds, tid, lane);
#pragma unroll
      for (int b = 0; b < NBO; ++b) {
        const f32x16 z = tp_load_blk(a.z_tp[l], tile, NBO, b, lane);
        const f32x16 rr = tp_load_blk(a.r_tp[l], tile, NBO, b, lane);
        f32x16 zc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d1 = softplus100_d1(z[r]);
          const float vv = v[b][r];
          qb[b][r] = d1 * vv;
          zc[r] = vv * rr[r] * (100.0f * (1.0f - d1));
        }
        tp_store_blk(zc, a.zb_tp[l], tile, NBO, b, lane);
      }
    });
    tp_store<D::NBH>(qb, a.qb_tp[D::NL], tile, lane);
  }

  // ---- backward pass
  f32x16 ub[MAXB];
  {
    f32x16 fb[MAXB];
    tp_load<D::NBF>(fb, a.featbar_tp, tile, lane);
    const float sb = a.sdfbar[tile * 32 + (lane & 31)];
#pragma unroll
    for (int b = 0; b < D::NBH; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) ub[b][r] = a.p.w_sdf[b * 32 + tp_row(r, hf)] * sb;
    tp_gemm<D::NBF, D::NBH>(ub, fb, a.p.wpT[D::NL], lds, tid, lane);
  }
  static_for<0, D::NL>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = D::NL - 1 - decltype(lc)::value;
    constexpr int KB = D::kb(l), NBO = D::nbo(l);
    // zb_l = ub * s'(z_l) + zc_l   (in place in ub)
#pragma unroll
    for (int b = 0; b < NBO; ++b) {
      const f32x16 z = tp_load_blk(a.z_tp[l], tile, NBO, b, lane);
      const f32x16 zc = tp_load_blk(a.zb_tp[l], tile, NBO, b, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) ub[b][r] = fmaf(ub[b][r], softplus100_d1(z[r]), zc[r]);
      tp_store_blk(ub[b], a.zb_tp[l], tile, NBO, b, lane);
    }
    f32x16 un[MAXB];
#pragma unroll
    for (int b = 0; b < KB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) un[b][r] = 0.0f;
    tp_gemm<NBO, KB>(un, ub, a.p.wpT[l], lds, tid, lane);
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
    } else {
      constexpr int NC = D::nbo(l - 1);
#pragma unroll
      for (int b = 0; b < NC; ++b) ub[b] = un[b];
    }
  });
}
