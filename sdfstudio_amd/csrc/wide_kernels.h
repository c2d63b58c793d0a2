// sdfhip — LAYER-AT-A-TIME geometry kernels for hidden widths whose two accumulator sets do not fit a wave: NBH = 16 (hidden 512, the
// reference's neus-facto-bigmlp preset, configs/method_configs.py:503-523).
//
// The fused kernels of geo_kernels.h keep the previous layer's result (accIn) and the current one (accOut) in registers: 2 x NBH x 16
// accumulator registers per lane, the whole 512-register file at NBH = 16.  Here ONE layer is one launch: its input blocks come from the
// tensors the training data flow keeps per layer anyway (u_l, r_l, qb_l, zb_l: geo_kernels.h header) through the memory-operand path of
// tp_gemm (the one the in0 gemms use), its NBH output blocks are the only accumulator set, and the element-wise step that the fused
// kernels run in the NEXT gemm's producer runs as this launch's epilogue.  Same gemm core, same packed weights, same precision modes,
// same saved tensors (so the weight-gradient GEMMs, the sdf-row gradient and everything around the network are shared); the price is
// that every per-layer tensor is read back once more than in the fused kernels.  Passes (names as in geo_kernels.h):
//   up / forward    u_l     = act(W_l u_{l-1} [+ W_l,in0 in0] + b_l)                                   wide_fwd_kernel
//   output layer    feat, sdf  (and the chain's seed r_{NL-1} = w_s s'(u_{NL-1}))                      wide_out_kernel
//   down / chain    r_{l-1} = (W_l^T r_l) s'(u_{l-1}) ;  e (+)= W_l,in0^T r_l                          wide_down_kernel<.., ADD = false>
//   up / tangent    v_l = W_l qb_l [+ W_l,in0 ebar] ; qb_{l+1} = s'(u_l) v_l ; zc_l = v_l r_l 100 (1 - s'(u_l))   wide_tan_kernel
//   backward seed   zb_{NL-1} = (w_s sdfbar + W_f^T featbar) s'(u_{NL-1}) + zc_{NL-1}                  wide_bwd_seed_kernel
//   down / backward zb_{l-1} = (W_l^T zb_l) s'(u_{l-1}) + zc_{l-1} ; in0bar (+)= W_l,in0^T zb_l        wide_down_kernel<.., ADD = TANGENT>
// Which parts a layer has (a hidden input, an in0 input) is a template parameter - the LAST gemm of a launch must not prefetch a
// successor's weights (an LDS-DMA still in flight when the workgroup ends would land in somebody else's LDS) - and the host loop picks
// the instantiation per layer.
//
// Round 5: HALF-WIDTH workgroups, two per CU.  A launch used to be the SUM of its matrix time and its HBM time: one wave per SIMD (16
// accumulator blocks = 256 of its 512 registers), so a wave's epilogue - 16 blocks of loads, element-wise work and stores - had nobody to
// hide behind.  Now a workgroup owns kWideNbo = 8 of the layer's 16 out-blocks (blockIdx -> (tile group, half), wide_block): 128
// accumulator registers, the whole kernel under 256, LDS 2 x 32 KB of weight chunks - so TWO workgroups are resident per CU (launch
// bounds (256, 2)) and one's epilogue runs beside the other's gemm.  The two halves of a tile group read the same input blocks; they
// are adjacent on the same XCD (workgroups go to the XCDs round robin: ids 16 g + 0..7 and 16 g + 8..15 are the two halves of tile
// groups 8 g .. 8 g + 7), so the second read is an L2 hit.  The packed weights of a 16-out-block matrix are stored as two
// 8-out-block matrices, half after half (api.hip: field_create).
#pragma once
#include "geo_kernels.h"

constexpr int kWideG = 2;    // out-blocks per weight-fragment group of the gemm (mlp_core.h GCAP): 256 registers per wave
template <class D>
struct WideLds {
  static_assert(D::NBH == 2 * kWideNbo && D::NBF <= kWideNbo, "wide kernels: 16 hidden blocks as two halves of 8");
  static constexpr int buf_floats(int ns) { return chunk_pieces(kWideNbo, ns) * 256; }          // one weight chunk buffer (8 out-blocks)
  static constexpr int floats(int ns) { return 2 * buf_floats(ns) + 2 * D::CW; }               // two chunk buffers, bias_l, w_sdf
};
SDFHIP_D BlkSrc<1> wide_src(const float* base, const int64_t tile, const int nb, const int b) { return BlkSrc<1>{{tp_block_ptr(base, tile, nb, b)}}; }
// blockIdx.x of a 2 G-block launch -> (tile group, half): both halves of a group on one XCD, 8 ids apart
SDFHIP_D void wide_block(int64_t& group, int& half) {
  const unsigned L = blockIdx.x;
  half = (int)((L >> 3) & 1u);
  group = (int64_t)(((L >> 4) << 3) | (L & 7u));
}
// launches that do not split (the output layer: 8 feature blocks; the in0 part of a down pass: 3 blocks) use blockIdx.x as the group
// floats between the two halves of a packed matrix of `kb` k-chunks
SDFHIP_HD constexpr size_t wide_half_stride(const int kb) { return (size_t)kb * kWideNbo * kChunkBlockFloats; }

// ---- forward, hidden layer l
template <class D, bool HID, bool IN0>
__global__ __launch_bounds__(256, 2) void wide_fwd_kernel(const GeoFwdArgs a, const int l, const int64_t n_groups) {
  static_assert(HID || IN0, "a layer has an input");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t group;
  int half;
  wide_block(group, half);
  if (group >= n_groups) return;  // the launch is padded to whole sets of 8 groups x 2 halves (workgroup-uniform: before any barrier)
  const int64_t tile = group * 4 + wave;
  constexpr int NS = kNsFwd, NBO = kWideNbo, PCS = chunk_pieces(NBO, NS);
  constexpr int KBL = (HID ? D::NBH : 0) + (IN0 ? D::NB0 : 0);             // k-chunks of this layer's packed matrix
  const int ob0 = half * NBO;
  float* cvec = lds + 2 * WideLds<D>::buf_floats(NS);
  const float* w_h = a.p.wp[l] + half * wide_half_stride(KBL);             // NBH chunks over the hidden input ...
  const float* w_i = HID ? w_h + wide_half_stride(D::NBH) : w_h;           // ... then NB0 chunks over in0 (layer 0: only those)
  WStream ws{lds, WideLds<D>::buf_floats(NS), 0, wave, lane};
  ws.issue(HID ? w_h : w_i, PCS, true);
  for (int i = tid; i < D::NBH * 32; i += 256) cvec[i] = a.p.bias[l][i];  // NBH * 32 = 512 values, 256 threads
  __syncthreads();
  f32x16 acc[NBO];
#pragma unroll
  for (int b = 0; b < NBO; ++b) acc[b] = tp_rowvec_blk(cvec, ob0 + b, hf);
  Raw carry;
  auto ident = [&](auto, const Raw& raw, auto ec) __attribute__((always_inline)) { return raw.a[decltype(ec)::value]; };
  auto in0_fetch = [&](auto kbc) __attribute__((always_inline)) { return wide_src(a.in0_tp, tile, D::NB0, decltype(kbc)::value); };
  if constexpr (HID) {
    const float* uprev = a.u_tp[l - 1];
    auto fetch = [&](auto kbc) __attribute__((always_inline)) { return wide_src(uprev, tile, D::NBH, decltype(kbc)::value); };
    carry = load_src(fetch(IC<0>{}), lane);
    if constexpr (IN0) {
      auto next_fetch = [&]() __attribute__((always_inline)) { return wide_src(a.in0_tp, tile, D::NB0, 0); };
      tp_gemm<D::NBH, NBO, Stores<0>, NS, PCS, kWideG>(acc, carry, fetch, ident, next_fetch, ws, w_h, w_i);
    } else {
      tp_gemm<D::NBH, NBO, Stores<0>, NS, 0, kWideG>(acc, carry, fetch, ident, NoFetch{}, ws, w_h, nullptr);
    }
  } else {
    carry = load_src(in0_fetch(IC<0>{}), lane);
  }
  if constexpr (IN0) tp_gemm<D::NB0, NBO, Stores<0>, NS, 0, kWideG>(acc, carry, in0_fetch, ident, NoFetch{}, ws, w_i, nullptr);
  float* ul = a.u_tp[l];
#pragma unroll
  for (int b = 0; b < NBO; ++b) {
    f32x16 u;
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = act_h<D::ACT>(acc[b][r]);
    tp_store_blk(u, ul, tile, D::NBH, ob0 + b, lane);
  }
}

// ---- output layer: feature rows on the MFMA path, the sdf row as a lane-local dot product in the producer; with GRAD the chain's
// seed r_{NL-1} = w_s s'(u_{NL-1}) is stored from the same pass over u_{NL-1}.  NBF = 8 out-blocks: one workgroup per tile group.
template <class D, bool GRAD, bool FEAT>
__global__ __launch_bounds__(256, 2) void wide_out_kernel(const GeoFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int NS = kNsFwd;
  const int NL = a.p.nl;
  float* cvec = lds + 2 * WideLds<D>::buf_floats(NS);
  WStream ws{lds, WideLds<D>::buf_floats(NS), 0, wave, lane};
  if constexpr (FEAT) ws.issue(a.p.wp[NL], chunk_pieces(D::NBF, NS), true);
  if (tid < D::NBF * 32) cvec[tid] = a.p.bias[NL][tid];
  for (int i = tid; i < D::NBH * 32; i += 256) cvec[D::CW + i] = a.p.w_sdf[i];
  __syncthreads();
  const float* wsdf = cvec + D::CW;
  const float* ulast = a.u_tp[NL - 1];
  float* rlast = a.r_tp[NL - 1];
  float part = 0.0f;
  auto make = [&](auto kbc, const Raw& raw, auto ec) __attribute__((always_inline)) {
    constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
    const float u = raw.a[e], w = wsdf[kb * 32 + tp_row(e, hf)];
    part = fmaf(w, u, part);
    pin_here(part);
    if constexpr (GRAD) *tp_elem(rlast, tile, D::NBH, kb, e, lane) = w * act_d1h<D::ACT>(u);
    return u;
  };
  auto fetch = [&](auto kbc) __attribute__((always_inline)) { return wide_src(ulast, tile, D::NBH, decltype(kbc)::value); };
  if constexpr (FEAT) {
    f32x16 acc[D::NBF];
#pragma unroll
    for (int b = 0; b < D::NBF; ++b) acc[b] = tp_rowvec_blk(cvec, b, hf);
    Raw carry = load_src(fetch(IC<0>{}), lane);
    tp_gemm<D::NBH, D::NBF, Stores<(GRAD ? 16 : 0)>, NS, 0, kWideG>(acc, carry, fetch, make, NoFetch{}, ws, a.p.wp[NL], nullptr);
#pragma unroll
    for (int b = 0; b < D::NBF; ++b) tp_store_blk(acc[b], a.feat_tp, tile, D::NBF, b, lane);
  } else {
    static_for<0, D::NBH>([&](auto kbc) __attribute__((always_inline)) {
      const Raw raw = load_src(fetch(kbc), lane);
      static_for<0, 16>([&](auto ec) __attribute__((always_inline)) { (void)make(kbc, raw, ec); });
    });
  }
  part += __shfl_xor(part, 32);
  if (hf == 0) a.sdf[tile * 32 + lane] = part + a.p.b_sdf[0];
}

// ---- down pass, layer l: x_l (r_l in the chain, zb_l in the backward) times W_l^T.
//   IN0 (l == 0 or l == skip): e (+)= W_{l,in0}^T x_l     (the skip layer parks its part in `e`, layer 0 adds to it)    one workgroup per tile group
//   HID (l > 0):               y_{l-1} = (W_l^T x_l) s'(u_{l-1}) [+ y_{l-1} as stored: the tangent pass's zc_{l-1}]      two halves
// The host runs the skip layer as TWO launches (IN0 only, then HID only).
struct WideDownArgs {
  GeoPtrs p;
  const float* x_tp[kMaxLayers];  // [T][NBH]
  float* y_tp[kMaxLayers];        // [T][NBH]
  const float* u_tp[kMaxLayers];
  float* e_tp;                    // [T][NB0]
};
template <class D, int NS, bool ADD, bool HID, bool IN0>
__global__ __launch_bounds__(256, 2) void wide_down_kernel(const WideDownArgs a, const int l, const int64_t n_groups) {
  static_assert(HID != IN0, "one part per launch");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t group = blockIdx.x;
  int half = 0;
  if constexpr (HID) wide_block(group, half);
  if (group >= n_groups) return;
  const int64_t tile = group * 4 + wave;
  constexpr int NBO = kWideNbo;
  const int ob0 = half * NBO;
  const float* w_in0 = l == 0 ? a.p.wpT[0] : a.p.wpT_in0;
  const float* w_hid = a.p.wpT[l] + half * wide_half_stride(D::NBH);
  WStream ws{lds, WideLds<D>::buf_floats(NS), 0, wave, lane};
  ws.issue(IN0 ? w_in0 : w_hid, IN0 ? chunk_pieces(D::NB0, NS) : chunk_pieces(NBO, NS), true);
  const float* xl = a.x_tp[l];
  auto fetch = [&](auto kbc) __attribute__((always_inline)) { return wide_src(xl, tile, D::NBH, decltype(kbc)::value); };
  auto ident = [&](auto, const Raw& raw, auto ec) __attribute__((always_inline)) { return raw.a[decltype(ec)::value]; };
  Raw carry = load_src(fetch(IC<0>{}), lane);
  if constexpr (IN0) {
    f32x16 accE[D::NB0];
#pragma unroll
    for (int b = 0; b < D::NB0; ++b) accE[b] = f32x16_zero();
    tp_gemm<D::NBH, D::NB0, Stores<0>, NS, 0, kWideG>(accE, carry, fetch, ident, NoFetch{}, ws, w_in0, nullptr);
    if (l == 0 && a.p.skip > 0) {  // layer 0 of a network with a skip layer: add what the skip layer parked
#pragma unroll
      for (int b = 0; b < D::NB0; ++b) accE[b] += tp_load_blk(a.e_tp, tile, D::NB0, b, lane);
    }
#pragma unroll
    for (int b = 0; b < D::NB0; ++b) tp_store_blk(accE[b], a.e_tp, tile, D::NB0, b, lane);
  }
  if constexpr (HID) {
    f32x16 acc[NBO];
#pragma unroll
    for (int b = 0; b < NBO; ++b) acc[b] = f32x16_zero();
    tp_gemm<D::NBH, NBO, Stores<0>, NS, 0, kWideG>(acc, carry, fetch, ident, NoFetch{}, ws, w_hid, nullptr);
    const float* ub = a.u_tp[l - 1];
    float* yb = a.y_tp[l - 1];
#pragma unroll
    for (int b = 0; b < NBO; ++b) {
      const f32x16 u = tp_load_blk(ub, tile, D::NBH, ob0 + b, lane);
      f32x16 y;
      if constexpr (ADD) y = tp_load_blk(yb, tile, D::NBH, ob0 + b, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = acc[b][r] * act_d1h<D::ACT>(u[r]);
        y[r] = ADD ? y[r] + t : t;
      }
      tp_store_blk(y, yb, tile, D::NBH, ob0 + b, lane);
    }
  }
}

// ---- tangent pass, layer l: v_l = W_l qb_l [+ W_{l,in0} ebar]; epilogue qb_{l+1} = s'(u_l) v_l, zc_l = v_l r_l 100 (1 - s'(u_l)) -> zb_tp[l]
template <class D, bool HID, bool IN0>
__global__ __launch_bounds__(256, 2) void wide_tan_kernel(const GeoBwdArgs a, const int l, const int64_t n_groups) {
  static_assert(HID || IN0, "a layer has an input");
  static_assert(D::ACT == 0, "the tangent pass uses Softplus(100)'s second derivative");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t group;
  int half;
  wide_block(group, half);
  if (group >= n_groups) return;
  const int64_t tile = group * 4 + wave;
  constexpr int NS = kNsGrad, NBO = kWideNbo, PCS = chunk_pieces(NBO, NS);
  constexpr int KBL = (HID ? D::NBH : 0) + (IN0 ? D::NB0 : 0);
  const int ob0 = half * NBO;
  const int SKIP = a.p.skip;
  const float* w_h = a.p.wp[l] + half * wide_half_stride(KBL);
  const float* w_i = HID ? w_h + wide_half_stride(D::NBH) : w_h;
  WStream ws{lds, WideLds<D>::buf_floats(NS), 0, wave, lane};
  ws.issue(HID ? w_h : w_i, PCS, true);
  f32x16 acc[NBO];
#pragma unroll
  for (int b = 0; b < NBO; ++b) acc[b] = f32x16_zero();
  constexpr int qb_nb = HID ? (IN0 ? D::NBH + D::NB0 : D::NBH) : D::NB0;  // blocks per tile of qb_tp[l]
  float* qbl = a.qb_tp[l];
  auto ident = [&](auto, const Raw& raw, auto ec) __attribute__((always_inline)) { return raw.a[decltype(ec)::value]; };
  auto seed_fetch = [&](auto kbc) __attribute__((always_inline)) { return wide_src(a.ebar_tp, tile, D::NB0, decltype(kbc)::value); };
  // the in0 part of the skip layer: the tangent seed again; its copy in qb_tp[SKIP] is the weight gradient's operand (written by half 0)
  auto seed_make = [&](auto kbc, const Raw& raw, auto ec) __attribute__((always_inline)) {
    constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
    if constexpr (HID) {
      if (half == 0) *tp_elem(qbl, tile, qb_nb, D::NBH + kb, e, lane) = raw.a[e];
    }
    return raw.a[e];
  };
  Raw carry;
  if constexpr (HID) {
    auto fetch = [&](auto kbc) __attribute__((always_inline)) { return wide_src(qbl, tile, qb_nb, decltype(kbc)::value); };
    carry = load_src(fetch(IC<0>{}), lane);
    if constexpr (IN0) {
      auto next_fetch = [&]() __attribute__((always_inline)) { return wide_src(a.ebar_tp, tile, D::NB0, 0); };
      tp_gemm<D::NBH, NBO, Stores<0>, NS, PCS, kWideG>(acc, carry, fetch, ident, next_fetch, ws, w_h, w_i);
    } else {
      tp_gemm<D::NBH, NBO, Stores<0>, NS, 0, kWideG>(acc, carry, fetch, ident, NoFetch{}, ws, w_h, nullptr);
    }
  } else {
    carry = load_src(seed_fetch(IC<0>{}), lane);
  }
  if constexpr (IN0) tp_gemm<D::NB0, NBO, Stores<(HID ? 16 : 0)>, NS, 0, kWideG>(acc, carry, seed_fetch, seed_make, NoFetch{}, ws, w_i, nullptr);
  const float* ul = a.u_tp[l];
  const float* rl = a.r_tp[l];
  float* zbl = a.zb_tp[l];
  float* qbn = a.qb_tp[l + 1];
  const int qbn_nb = (l + 1 == SKIP) ? D::NBH + D::NB0 : D::NBH;  // l + 1 == NL: the tangent reaching the sdf row, NBH blocks
#pragma unroll
  for (int b = 0; b < NBO; ++b) {
    const f32x16 u = tp_load_blk(ul, tile, D::NBH, ob0 + b, lane), r = tp_load_blk(rl, tile, D::NBH, ob0 + b, lane);
    f32x16 zc, qn;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v = acc[b][i], d1 = act_d1h<D::ACT>(u[i]);
      zc[i] = v * r[i] * (100.0f * (1.0f - d1));
      qn[i] = d1 * v;
    }
    tp_store_blk(zc, zbl, tile, D::NBH, ob0 + b, lane);
    tp_store_blk(qn, qbn, tile, qbn_nb, ob0 + b, lane);
  }
}

// ---- backward seed: zb_{NL-1} = (w_s sdfbar + W_f^T featbar) s'(u_{NL-1}) [+ zc_{NL-1}]
template <class D, bool TANGENT>
__global__ __launch_bounds__(256, 2) void wide_bwd_seed_kernel(const GeoBwdArgs a, const int64_t n_groups) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t group;
  int half;
  wide_block(group, half);
  if (group >= n_groups) return;
  const int64_t tile = group * 4 + wave;
  constexpr int NS = kNsGrad, NBO = kWideNbo;
  const int ob0 = half * NBO;
  const int NL = a.p.nl;
  float* cvec = lds + 2 * WideLds<D>::buf_floats(NS);
  WStream ws{lds, WideLds<D>::buf_floats(NS), 0, wave, lane};
  const float* wf = a.p.wpT[NL] + half * wide_half_stride(D::NBF);
  ws.issue(wf, chunk_pieces(NBO, NS), true);
  for (int i = tid; i < D::NBH * 32; i += 256) cvec[i] = a.p.w_sdf[i];
  __syncthreads();
  const float sb = a.sdfbar[tile * 32 + (lane & 31)];
  f32x16 acc[NBO];
#pragma unroll
  for (int b = 0; b < NBO; ++b) {
    const f32x16 w = tp_rowvec_blk(cvec, ob0 + b, hf);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = w[i] * sb;
  }
  auto fetch = [&](auto kbc) __attribute__((always_inline)) { return wide_src(a.featbar_tp, tile, D::NBF, decltype(kbc)::value); };
  auto ident = [&](auto, const Raw& raw, auto ec) __attribute__((always_inline)) { return raw.a[decltype(ec)::value]; };
  Raw carry = load_src(fetch(IC<0>{}), lane);
  tp_gemm<D::NBF, NBO, Stores<0>, NS, 0, kWideG>(acc, carry, fetch, ident, NoFetch{}, ws, wf, nullptr);
  const float* ul = a.u_tp[NL - 1];
  float* zbl = a.zb_tp[NL - 1];
#pragma unroll
  for (int b = 0; b < NBO; ++b) {
    const f32x16 u = tp_load_blk(ul, tile, D::NBH, ob0 + b, lane);
    f32x16 z;
    if constexpr (TANGENT) z = tp_load_blk(zbl, tile, D::NBH, ob0 + b, lane);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float t = acc[b][i] * act_d1h<D::ACT>(u[i]);
      z[i] = TANGENT ? z[i] + t : t;
    }
    tp_store_blk(z, zbl, tile, D::NBH, ob0 + b, lane);
  }
}
