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
//
// Code size and the layer loops.  Unrolled over the layers these kernels were 330 - 380 KB of straight-line code against
// a 64 KB instruction cache shared by two CUs: every workgroup streamed the whole kernel through it.  Most MI355X boxes hide
// that behind the sequential instruction prefetch; on some the fetch rate of code that misses the cache drops from 1.2 to
// 2.9 cycles per instruction (tools/probe_icache.hip) and exactly the kernels larger than the cache ran 2.4x slower
// (DESIGN.md section 5).  Each pass below is therefore a RUN-TIME loop over the layers around ONE instance of the
// hidden -> hidden gemm (~25 KB) and one of the small in0 gemm, and each pass is its own LAUNCH (template parameter PHASE):
//   * every hidden layer is NBH blocks wide - the layer below the skip concatenation is padded from H - D0 to H rows
//     (zero weights, NB3 == NBH), so that layer and the skip layer have the regular shape;
//   * the skip layer's input cat([h, in0]) is two gemms into the same accumulators: the regular one over h and the in0 gemm
//     that also serves layer 0 (forward / tangent: in0 -> hidden; chain / backward: hidden -> in0);
//   * accumulator roles are fixed (accIn: the previous layer's result, accOut: this layer's), with a register copy per layer
//     in place of the compile-time ping-pong; pointers, biases and weight chunks are indexed by the run-time layer.
#pragma once
#include "mlp_core.h"

constexpr int kMaxLayers = 10;
// layer-at-a-time kernels (wide_kernels.h, hidden width 512): out-blocks one workgroup owns; the packed weights of a 16-out-block matrix
// are stored as two 8-out-block matrices, half after half (api.hip: field_create / fill_geo_ptrs)
constexpr int kWideNbo = 8;
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

// Block widths of a geometry network (32 features per block): hidden, in0 (position + encodings + grid features), geometry
// feature.  The DEPTH (hidden layers NL, skip layer SKIP or -1) is a run-time property carried in GeoPtrs: the kernels loop.
template <int NBH_, int NB0_, int NBF_, int ACT_ = 0>
struct GeoDims {
  static constexpr int NBH = NBH_, NB0 = NB0_, NBF = NBF_;
  static constexpr int ACT = ACT_;  // hidden activation (common.h act_h / act_d1): 0 Softplus(100), 1 ReLU (first-order kernels only)
  static constexpr int cmax(int a, int b) { return a > b ? a : b; }
  static constexpr int MAXO = cmax(NBH, NBF);                            // widest chunk (out-blocks) any gemm streams
  static constexpr int buf_floats(int ns) { return chunk_pieces(MAXO, ns) * 256; }  // one weight chunk buffer
  static constexpr int CW = MAXO * 32;                                   // stride of the constant-vector area
  // LDS: two chunk buffers, then biases of layers 0..NL and w_sdf
  static constexpr int lds_floats(int ns, int nl) { return 2 * buf_floats(ns) + (nl + 2) * CW; }
  // every gemm prefetches the first chunk of whatever follows it (a run-time choice) at this size; a smaller chunk is
  // over-read into the one behind it (the packed buffer ends in slack)
  static constexpr int pieces(int ns) { return chunk_pieces(MAXO, ns); }
};
// tensor layouts that depend on the depth: qb_tp[l] has kb(l) blocks per tile (NB0 / NBH + NB0 at the skip layer / NBH),
// u_tp[l], r_tp[l], zb_tp[l] have NBH (every hidden layer is NBH wide: the one below the skip concatenation is padded)

struct GeoPtrs {
  int32_t nl, skip;               // hidden layers (the output layer is layer nl), skip layer (-1: none; 1 <= skip < nl)
  const float* wp[kMaxLayers];    // packed W_l      [kb][parts][nbo][2][64] x 8 ; the skip layer: NBH chunks over h, then NB0 over in0
  const float* wpT[kMaxLayers];   // packed W_l^T    [nbo][parts][kb][2][64] x 8 ; the skip layer: the NBH columns over h only
  const float* wpT_in0;           // packed W_SKIP^T restricted to the in0 columns: [NBH][parts][NB0][2][64] x 8
  const float* bias[kMaxLayers];  // natural order, padded to nbo*32
  const float* w_sdf;             // [NBH*32]  row of the output layer that produces sdf
  const float* b_sdf;             // [1]
};

struct GeoFwdArgs {
  GeoPtrs p;
  const float* in0_tp;  // [T][NB0]
  float* u_tp[kMaxLayers];  // [T][nbo(l)]   saved ACTIVATIONS act(z_l) (training only): what the next layer and the weight gradient
                            // consume as they are; the backward-type passes recover s'(z_l) from them (common.h act_d1h)
  float* r_tp[kMaxLayers];  // [T][nbo(l)]   (training only)
  float* feat_tp;           // [T][NBF]
  float* sdf;               // [T*32]
  float* e_tp;              // [T][NB0]   d sdf / d in0
};

// biases (layers 0..NL) and w_sdf into the constant area of LDS: cvec[l * CW + i], w_sdf at l = NL + 1
template <class D>
SDFHIP_D void geo_stage_cvec(float* cvec, const GeoPtrs& p, const int tid) {
  for (int l = 0; l < p.nl; ++l)
    if (tid < D::NBH * 32) cvec[l * D::CW + tid] = p.bias[l][tid];
  if (tid < D::NBF * 32) cvec[p.nl * D::CW + tid] = p.bias[p.nl][tid];
  if (tid < D::NBH * 32) cvec[(p.nl + 1) * D::CW + tid] = p.w_sdf[tid];
}

template <int N>
SDFHIP_D void acc_copy(f32x16 (&dst)[N], const f32x16 (&src)[N]) {
#pragma unroll
  for (int b = 0; b < N; ++b) dst[b] = src[b];
}
template <int N, int M>
SDFHIP_D void acc_copy_n(f32x16 (&dst)[N], const f32x16 (&src)[M]) {
  static_assert(N <= M, "");
#pragma unroll
  for (int b = 0; b < N; ++b) dst[b] = src[b];
}
// first chunk of the in0 part of the skip layer's packed weights (NBH chunks of NBH out-blocks come first)
template <class D>
SDFHIP_D const float* geo_skip_in0(const float* wp_skip) {
  return wp_skip + (size_t)D::NBH * D::NBH * kChunkBlockFloats;
}

// PHASE splits the kernel in two launches (0: everything in one).  The chain starts from a constant (q_NL = w_sdf) and reads z_l
// from HBM either way, so running it as its own launch (PHASE 2) after the forward (PHASE 1) moves no extra data - and every CU
// of a launch then sits in the SAME layer loop.  That matters on the boxes of DESIGN.md section 5: two CUs share an instruction
// cache, and a forward loop (25 KB) next to a chain loop (37 KB) does not fit its fast half.
// NS_: precision mode of the products (mlp_core.h).  Default kNsFwd (fp16 hi + lo: 22 mantissa bits); 3 (three bf16 parts, six terms: all 24
// bits, the error class of an fp32 GEMM) is instantiated for the first-order modes of BASELINE config 5's shape, whose numerical normals
// divide sdf DIFFERENCES by a delta that shrinks to 2.4e-4 (sdfhip_numfield_forward).
template <class D, bool GRAD, bool SAVE, bool FEAT, int PHASE = 0, int NS_ = kNsFwd>
__global__ __launch_bounds__(256, 1) void geo_fwd_kernel(const GeoFwdArgs a) {
  static_assert(PHASE == 0 || GRAD, "the chain only exists with GRAD");
  static_assert(!GRAD || D::ACT == 0, "the analytic-normal chain is written for Softplus(100) networks");
  constexpr bool CHAIN = GRAD && PHASE != 1;  // this launch runs (and prefetches for) the chain
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int W = D::CW, NS = NS_, PCS = D::pieces(NS);
  const int NL = a.p.nl, SKIP = a.p.skip;
  float* cvec = lds + 2 * D::buf_floats(NS);

  // first gemm of the chain: the in0 part if the last hidden layer is the skip layer, the hidden part otherwise; its operands
  // are block 0 of z_{NL-1} either way
  auto chain_first_src = [&]() __attribute__((always_inline)) { return BlkSrc<1>{{tp_block_ptr(a.u_tp[NL - 1], tile, D::NBH, 0)}}; };
  const float* chain_first_w = SKIP == NL - 1 ? a.p.wpT_in0 : a.p.wpT[NL - 1];

  WStream ws{lds, D::buf_floats(NS), 0, wave, lane};
  ws.issue(PHASE == 2 ? chain_first_w : a.p.wp[0], PHASE == 2 ? PCS : chunk_pieces(D::NBH, NS), true);
  geo_stage_cvec<D>(cvec, a.p, tid);
  __syncthreads();

  f32x16 accIn[D::NBH], accOut[D::MAXO];
  Raw carry;
  auto in0_blk0 = [&]() __attribute__((always_inline)) { return BlkSrc<1>{{tp_block_ptr(a.in0_tp, tile, D::NB0, 0)}}; };
#ifdef SDFHIP_ABL_FWD_NOSTORE
  constexpr int ZS = 0;
#else
  constexpr int ZS = (SAVE || GRAD) ? 16 : 0;  // activation stores per produced block
#endif

  if constexpr (PHASE == 2) carry = load_src(chain_first_src(), lane);
  if constexpr (PHASE != 2) {
  // ---- forward layers 0 .. NL-1: out_l = b_l + W_l u_l
  carry = load_src(in0_blk0(), lane);
#pragma unroll 1
  for (int l = 0; l < NL; ++l) {
    // weights that follow the last hidden layer: the output layer's, else the first chain gemm's, else nothing (re-read own)
    const float* after_last = FEAT ? a.p.wp[NL] : (CHAIN ? chain_first_w : a.p.wp[l]);
    {
      const float* bias = cvec + l * W;
#pragma unroll
      for (int b = 0; b < D::NBH; ++b) accOut[b] = tp_rowvec_blk(bias, b, hf);
    }
    if (l > 0) {
      // hidden -> hidden: u_l = softplus(z_{l-1}) made from the accumulators of the layer below, z_{l-1} saved on the way
      float* uprev = a.u_tp[l - 1];
      auto make = [&](auto kbc, const Raw&, auto ec) __attribute__((always_inline)) {
        constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
        const float h = act_h<D::ACT, NS == 4>(accIn[kb][e]);  // bounded for fp16 operand parts in the activation's own clamp
#ifndef SDFHIP_ABL_FWD_NOSTORE  // timing ablation: the forward launch without its activation stores
        if constexpr (SAVE || GRAD) *tp_elem(uprev, tile, D::NBH, kb, e, lane) = h;
#endif
        return InRange{h};
      };
      const float* nxt = l == SKIP ? geo_skip_in0<D>(a.p.wp[l]) : (l + 1 < NL ? a.p.wp[l + 1] : after_last);
      tp_gemm<D::NBH, D::NBH, Stores<ZS>, NS, PCS>(accOut, carry, NoFetch{}, make, in0_blk0, ws, a.p.wp[l], nxt);
    }
    if (l == 0 || l == SKIP) {
      // in0 -> hidden: layer 0, and the in0 columns of the skip layer (cat([h, in0]) / sqrt(2), the factor folded into W)
      auto fetch = [&](auto kbc) __attribute__((always_inline)) {
        return BlkSrc<1>{{tp_block_ptr(a.in0_tp, tile, D::NB0, decltype(kbc)::value)}};
      };
      auto make = [&](auto, const Raw& raw, auto ec) __attribute__((always_inline)) { return raw.a[decltype(ec)::value]; };
      const float* w = l == 0 ? a.p.wp[0] : geo_skip_in0<D>(a.p.wp[l]);
      tp_gemm<D::NB0, D::NBH, Stores<0>, NS, PCS>(accOut, carry, fetch, make, NoFetch{}, ws, w, l + 1 < NL ? a.p.wp[l + 1] : after_last);
    }
    acc_copy_n(accIn, accOut);
  }

  // ---- output layer: the sdf row as a lane-local dot product riding in the producer, feature rows on the MFMA path
  {
    // The sdf row is a lane-local dot product over 128 of the last layer's 256 activations (the other half-wave holds the rest).  A plain
    // fma chain rounds its running sum (magnitude ~1.5 at geometric initialisation: 1.2e-7 per ulp) 128 times: 1.8e-7 rms, 8e-7 worst -
    // twice the error of the blocked sum an fp32 GEMM makes, and the LARGEST term of the 24-bit evaluation's error.  The 24-bit mode
    // (NS == 3: the numerical normal divides sdf differences by 2 delta = 4.8e-4) therefore accumulates compensated (Kahan): the running
    // sum's rounding errors are carried in `comp` and taken out once at the end, which leaves the final rounding (6e-8) alone.
    constexpr bool KAHAN = NS == 3;
    float part = 0.0f, comp = 0.0f;
    float* ulast = a.u_tp[NL - 1];
    const float* wsdf = cvec + (NL + 1) * W;
    auto make = [&](auto kbc, const Raw&, auto ec) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
      const float h = act_h<D::ACT, NS == 4>(accIn[kb][e]);
      if constexpr (SAVE || GRAD) *tp_elem(ulast, tile, D::NBH, kb, e, lane) = h;
      if constexpr (KAHAN) {
        const float y = fmaf(wsdf[kb * 32 + tp_row(e, hf)], h, -comp);
        const float t = part + y;
        comp = (t - part) - y;  // what the addition lost, negated: the true sum is part - comp
        part = t;
        pin_here(comp);
      } else {
        part = fmaf(wsdf[kb * 32 + tp_row(e, hf)], h, part);
      }
      pin_here(part);
      return InRange{h};
    };
    if constexpr (FEAT) {
#pragma unroll
      for (int b = 0; b < D::NBF; ++b) accOut[b] = tp_rowvec_blk(cvec + NL * W, b, hf);
      auto next_fetch = [&]() __attribute__((always_inline)) {
        if constexpr (CHAIN) return chain_first_src();
        else return BlkSrc<0>{};
      };
      tp_gemm<D::NBH, D::NBF, Stores<ZS>, NS, (CHAIN ? PCS : 0)>(accOut, carry, NoFetch{}, make, next_fetch, ws, a.p.wp[NL],
                                                                 CHAIN ? chain_first_w : nullptr);
#pragma unroll
      for (int b = 0; b < D::NBF; ++b) tp_store_blk(accOut[b], a.feat_tp, tile, D::NBF, b, lane);
    } else {
      static_for<0, D::NBH>([&](auto kbc) __attribute__((always_inline)) {
        static_for<0, 16>([&](auto ec) __attribute__((always_inline)) { (void)make(kbc, carry, ec); });
      });
      if constexpr (CHAIN) carry = load_src(chain_first_src(), lane);
    }
    if constexpr (KAHAN) {
      // the two half-waves' sums and the bias by error-free additions (TwoSum); every rounding error joins `comp`
      const float p1 = __shfl_xor(part, 32), c1 = __shfl_xor(comp, 32);
      const float s2 = part + p1, v2 = s2 - part;
      const float e2 = (part - (s2 - v2)) + (p1 - v2);  // part + p1 == s2 + e2 exactly
      const float b = a.p.b_sdf[0];
      const float s3 = s2 + b, v3 = s3 - s2;
      const float e3 = (s2 - (s3 - v3)) + (b - v3);     // s2 + b == s3 + e3 exactly
      if (hf == 0) a.sdf[tile * 32 + lane] = s3 + ((e2 + e3) - (comp + c1));
    } else {
      part += __shfl_xor(part, 32);
      if (hf == 0) a.sdf[tile * 32 + lane] = part + a.p.b_sdf[0];
    }
  }
  }  // PHASE != 2

  // ---- chain: q_l = W_l^T (q_{l+1} * s'(z_l)),  q_NL = w_sdf.  accIn holds q_{l+1}
  if constexpr (CHAIN) {
#pragma unroll
    for (int b = 0; b < D::NBH; ++b) accIn[b] = tp_rowvec_blk(cvec + (NL + 1) * W, b, hf);
#pragma unroll 1
    for (int l = NL - 1; l >= 0; --l) {
      const float* ul = a.u_tp[l];
      float* rl = a.r_tp[l];
      auto fetch = [&](auto bc) __attribute__((always_inline)) { return BlkSrc<1>{{tp_block_ptr(ul, tile, D::NBH, decltype(bc)::value)}}; };
      auto make = [&](auto bc, const Raw& raw, auto ec) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value, e = decltype(ec)::value;
        const float r = accIn[b][e] * act_d1h<D::ACT>(raw.a[e]);
        if constexpr (SAVE) *tp_elem(rl, tile, D::NBH, b, e, lane) = r;
        return r;
      };
      if (l == 0 || l == SKIP) {
        // hidden -> in0: layer 0, and the part of the skip layer's input gradient that goes straight to in0 (parked in e_tp,
        // layer 0 adds to it).  The hidden part of the skip layer follows with the same operands (z_l block 0)
        // accumulates in the first NB0 blocks of accOut: dead here (the hidden gemm below clears it before it writes), and a separate
        // NB0-block set next to accIn / accOut does not fit the register file at NB0 = 6 (config 5's shape spilled 196 B in the backward)
        static_assert(D::NB0 <= D::MAXO, "the in0 accumulators live in accOut");
        auto& accE = accOut;
#pragma unroll
        for (int b = 0; b < D::NB0; ++b) accE[b] = f32x16_zero();
        auto next_fetch = [&]() __attribute__((always_inline)) { return BlkSrc<1>{{tp_block_ptr(ul, tile, D::NBH, 0)}}; };
        tp_gemm<D::NBH, D::NB0, Stores<(SAVE ? 16 : 0)>, NS, PCS>(accE, carry, fetch, make, next_fetch, ws, l == 0 ? a.p.wpT[0] : a.p.wpT_in0,
                                                                  a.p.wpT[l]);
        if (l == 0 && SKIP > 0) {
#pragma unroll
          for (int b = 0; b < D::NB0; ++b) accE[b] += tp_load_blk(a.e_tp, tile, D::NB0, b, lane);
        }
#pragma unroll
        for (int b = 0; b < D::NB0; ++b) tp_store_blk(accE[b], a.e_tp, tile, D::NB0, b, lane);
      }
      if (l > 0) {
#pragma unroll
        for (int b = 0; b < D::NBH; ++b) accOut[b] = f32x16_zero();
        const float* ubelow = a.u_tp[l - 1];
        auto next_fetch = [&]() __attribute__((always_inline)) { return BlkSrc<1>{{tp_block_ptr(ubelow, tile, D::NBH, 0)}}; };
        tp_gemm<D::NBH, D::NBH, Stores<(SAVE ? 16 : 0)>, NS, PCS>(accOut, carry, fetch, make, next_fetch, ws, a.p.wpT[l],
                                                                  l - 1 == SKIP ? a.p.wpT_in0 : a.p.wpT[l - 1]);
        acc_copy_n(accIn, accOut);
      }
    }
  }
}

struct GeoBwdArgs {
  GeoPtrs p;
  const float* ebar_tp;     // [T][NB0]   tangent seed  J_in0 gbar
  const float* featbar_tp;  // [T][NBF]
  const float* sdfbar;      // [T*32]
  const float* u_tp[kMaxLayers];
  const float* r_tp[kMaxLayers];
  float* qb_tp[kMaxLayers + 1];  // [T][kb(l)]  tangent entering layer l  (l == NL: the tangent reaching the sdf row); qb_tp[0] == ebar_tp
  float* zb_tp[kMaxLayers];      // [T][nbo(l)] holds zc_l after the tangent pass, zbar_l after the backward pass
  float* in0bar_tp;              // [T][NB0]
};

// TANGENT = false: first-order backward only (no second-order terms: the caller differentiated sdf / feature, not d sdf / dx).
// PHASE (TANGENT only): 0 = tangent pass and data backward in one launch, 1 = tangent pass only, 2 = data backward only (zc from
// the tangent launch is in zb_tp).  The backward starts from ub_NL = w_s sdfbar + W_f^T featbar, not from registers of the
// tangent pass, so the split moves no extra data; the reason for it is the one given at geo_fwd_kernel.
// FEATBAR = false (first-order only): the caller has no cotangent for the geometry feature of these points (the six taps of the
// numerical-gradient branch are evaluated for their sdf alone): ub_NL = w_s sdfbar, the W_f^T featbar gemm and its reads are skipped.
template <class D, bool TANGENT = true, int PHASE = 0, bool FEATBAR = true>
__global__ __launch_bounds__(256, 1) void geo_bwd_kernel(const GeoBwdArgs a) {
  static_assert(PHASE == 0 || TANGENT, "phases split the second-order kernel");
  static_assert(FEATBAR || !TANGENT, "the feature-less variant is a first-order kernel");
  static_assert(!TANGENT || D::ACT == 0, "the tangent pass uses Softplus(100)'s second derivative");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  constexpr int NS = kNsGrad, PCS = D::pieces(NS);
  const int NL = a.p.nl, SKIP = a.p.skip;
  float* cvec = lds + 2 * D::buf_floats(NS);
  // first gemm of the backward pass proper (after the feature gemm)
  const float* bwd_first_w = SKIP == NL - 1 ? a.p.wpT_in0 : a.p.wpT[NL - 1];

  WStream ws{lds, D::buf_floats(NS), 0, wave, lane};
  if constexpr (TANGENT && PHASE != 2) ws.issue(a.p.wp[0], chunk_pieces(D::NBH, NS), true);
  else if constexpr (FEATBAR) ws.issue(a.p.wpT[NL], chunk_pieces(D::NBH, NS), true);
  else ws.issue(bwd_first_w, PCS, true);  // the widest chunk any gemm streams, over-read into the packed buffer's slack (geo_fwd PHASE 2)
  if (tid < D::NBH * 32) cvec[tid] = a.p.w_sdf[tid];
  __syncthreads();

  f32x16 accIn[D::NBH], accOut[D::NBH];
  Raw carry;

  if constexpr (TANGENT && PHASE != 2) {
    // ---- tangent pass (second-order terms): v_l = W_l qb_l ; accIn holds v_{l-1}
    carry = load_src(BlkSrc<1>{{tp_block_ptr(a.ebar_tp, tile, D::NB0, 0)}}, lane);
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
#pragma unroll
      for (int b = 0; b < D::NBH; ++b) accOut[b] = f32x16_zero();
      float* qbl = a.qb_tp[l];
      const int qb_nb = l == 0 ? D::NB0 : (l == SKIP ? D::NBH + D::NB0 : D::NBH);  // blocks per tile of qb_tp[l]
      if (l > 0) {
        // hidden -> hidden.  The producer is the tangent epilogue of the layer below on element e of block kb:
        //   qb_l = s'(z_{l-1}) v_{l-1} ;  zc_{l-1} = v_{l-1} r_{l-1} 100 (1 - s'(z_{l-1}))  (-> zb_tp[l-1])
        const float* up = a.u_tp[l - 1];
        const float* rp = a.r_tp[l - 1];
        float* zbp = a.zb_tp[l - 1];
        auto fetch = [&](auto kbc) __attribute__((always_inline)) {
          constexpr int kb = decltype(kbc)::value;
          return BlkSrc<2>{{tp_block_ptr(up, tile, D::NBH, kb), tp_block_ptr(rp, tile, D::NBH, kb)}};
        };
        auto make = [&](auto kbc, const Raw& raw, auto ec) __attribute__((always_inline)) {
          constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
          const float v = accIn[kb][e];
          const float d1 = act_d1h<D::ACT>(raw.a[e]);
          *tp_elem(zbp, tile, D::NBH, kb, e, lane) = v * raw.b[e] * (100.0f * (1.0f - d1));
          const float qn = d1 * v;
          *tp_elem(qbl, tile, qb_nb, kb, e, lane) = qn;
          return qn;
        };
        // what follows: the in0 part of this layer (operands: seed block 0), or the next layer (operands: (z_l, r_l) block 0)
        const float* na = l == SKIP ? a.ebar_tp : a.u_tp[l];
        const float* nb = l == SKIP ? a.ebar_tp : a.r_tp[l];
        const int nnb = l == SKIP ? D::NB0 : D::NBH;
        auto next_fetch = [&]() __attribute__((always_inline)) { return BlkSrc<2>{{tp_block_ptr(na, tile, nnb, 0), tp_block_ptr(nb, tile, nnb, 0)}}; };
        const float* nxt = l == SKIP ? geo_skip_in0<D>(a.p.wp[l]) : (l + 1 < NL ? a.p.wp[l + 1] : a.p.wpT[NL]);
        tp_gemm<D::NBH, D::NBH, Stores<32>, NS, PCS>(accOut, carry, fetch, make, next_fetch, ws, a.p.wp[l], nxt);
      }
      if (l == 0 || l == SKIP) {
        // in0 -> hidden on the tangent seed (qb_0 == ebar; the seed blocks of the skip layer's qb are copies of it)
        const int b0 = l == 0 ? 0 : D::NBH;
        auto fetch = [&](auto kbc) __attribute__((always_inline)) {
          return BlkSrc<1>{{tp_block_ptr(a.ebar_tp, tile, D::NB0, decltype(kbc)::value)}};
        };
        auto make = [&](auto kbc, const Raw& raw, auto ec) __attribute__((always_inline)) {
          constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
          *tp_elem(qbl, tile, qb_nb, b0 + kb, e, lane) = raw.a[e];  // layer 0: rewrites ebar with itself (qb_tp[0] == ebar_tp)
          return raw.a[e];
        };
        auto next_fetch = [&]() __attribute__((always_inline)) {
          return BlkSrc<2>{{tp_block_ptr(a.u_tp[l], tile, D::NBH, 0), tp_block_ptr(a.r_tp[l], tile, D::NBH, 0)}};
        };
        const float* w = l == 0 ? a.p.wp[0] : geo_skip_in0<D>(a.p.wp[l]);
        tp_gemm<D::NB0, D::NBH, Stores<16>, NS, PCS>(accOut, carry, fetch, make, next_fetch, ws, w,
                                                     l + 1 < NL ? a.p.wp[l + 1] : a.p.wpT[NL]);
      }
      acc_copy(accIn, accOut);
    }
    {
      // epilogue of the last hidden layer: qb_NL (tangent reaching the sdf row; only the weight gradient needs it) and zc_{NL-1}
      const float* ulast = a.u_tp[NL - 1];
      const float* rlast = a.r_tp[NL - 1];
      float* zblast = a.zb_tp[NL - 1];
      float* qblast = a.qb_tp[NL];
      static_for<0, D::NBH>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        const Raw raw = load_src(BlkSrc<2>{{tp_block_ptr(ulast, tile, D::NBH, b), tp_block_ptr(rlast, tile, D::NBH, b)}}, lane);
        static_for<0, 16>([&](auto ec) __attribute__((always_inline)) {
          constexpr int e = decltype(ec)::value;
          const float v = accIn[b][e];
          const float d1 = act_d1h<D::ACT>(raw.a[e]);
          *tp_elem(zblast, tile, D::NBH, b, e, lane) = v * raw.b[e] * (100.0f * (1.0f - d1));
          *tp_elem(qblast, tile, D::NBH, b, e, lane) = d1 * v;
        });
      });
    }
  }

  auto bwd_src = [&](const int l, const int b) __attribute__((always_inline)) {
    if constexpr (TANGENT) return BlkSrc<2>{{tp_block_ptr(a.u_tp[l], tile, D::NBH, b), tp_block_ptr(a.zb_tp[l], tile, D::NBH, b)}};
    else return BlkSrc<1>{{tp_block_ptr(a.u_tp[l], tile, D::NBH, b)}};
  };

  if constexpr (PHASE == 1) return;
  // ---- backward pass: ub_NL = w_s sdfbar + W_f^T featbar
  {
    const float sb = a.sdfbar[tile * 32 + (lane & 31)];
#pragma unroll
    for (int b = 0; b < D::NBH; ++b) {
      const f32x16 w = tp_rowvec_blk(cvec, b, hf);
#pragma unroll
      for (int i = 0; i < 16; ++i) accIn[b][i] = w[i] * sb;
    }
    auto fetch = [&](auto bc) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value;
      return BlkSrc<1>{{tp_block_ptr(a.featbar_tp, tile, D::NBF, b)}};
    };
    auto make = [&](auto, const Raw& raw, auto ec) __attribute__((always_inline)) { return raw.a[decltype(ec)::value]; };
    auto next_fetch = [&]() __attribute__((always_inline)) { return bwd_src(NL - 1, 0); };
    if constexpr (FEATBAR) {
      carry = load_src(fetch(IC<0>{}), lane);
      tp_gemm<D::NBF, D::NBH, Stores<0>, NS, PCS>(accIn, carry, fetch, make, next_fetch, ws, a.p.wpT[NL], bwd_first_w);
    } else {
      carry = load_src(next_fetch(), lane);
    }
  }
  // accIn holds ub_{l+1}.  zb_l = ub * s'(z_l) + zc_l (zc from the tangent pass, in zb_tp[l]) is produced, stored for the weight
  // gradient and multiplied by W_l^T: its in0 columns first where the layer has them (skip layer -> parked in in0bar, layer 0),
  // then the hidden columns.  The second gemm of the skip layer finds the FINISHED zb_l in zb_tp[l] (this lane's own stores)
#pragma unroll 1
  for (int l = NL - 1; l >= 0; --l) {
    float* zbl = a.zb_tp[l];
    auto fetch = [&](auto bc) __attribute__((always_inline)) { return bwd_src(l, decltype(bc)::value); };
    if (l == 0 || l == SKIP) {
      auto make = [&](auto bc, const Raw& raw, auto ec) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value, e = decltype(ec)::value;
        const float zb = TANGENT ? fmaf(accIn[b][e], act_d1h<D::ACT>(raw.a[e]), raw.b[e]) : accIn[b][e] * act_d1h<D::ACT>(raw.a[e]);
        *tp_elem(zbl, tile, D::NBH, b, e, lane) = zb;
        return zb;
      };
      static_assert(D::NB0 <= D::NBH, "the in0 accumulators live in accOut");
      auto& accE = accOut;  // dead here: see the chain of geo_fwd_kernel
#pragma unroll
      for (int b = 0; b < D::NB0; ++b) accE[b] = f32x16_zero();
      auto next_fetch = [&]() __attribute__((always_inline)) { return bwd_src(l, 0); };
      tp_gemm<D::NBH, D::NB0, Stores<16>, NS, PCS>(accE, carry, fetch, make, next_fetch, ws, l == 0 ? a.p.wpT[0] : a.p.wpT_in0, a.p.wpT[l]);
      // TANGENT: ub_{l+1} has been consumed for good - layer 0 ends the sweep, and the skip layer's hidden gemm below takes the FINISHED
      // zb_l from memory (`done`).  The compiler cannot see that across the two run-time conditions and kept all 8 blocks alive through
      // this gemm (NB0 = 6: 196 B of scratch); overwriting them ends their live ranges block by block as the producer passes.
      // The first-order kernel has no zc operand to fetch zb_l back through: its hidden gemm RECOMPUTES zb_l = ub_{l+1} s'(z_l) from accIn,
      // which must therefore survive (round 5: clearing it here zeroed every gradient at and below the skip layer in the first-order
      // backward - forward_geonetwork under autograd, the ReLU background fields, the numerical-gradient field of networks with a skip).
      if constexpr (TANGENT) {
#pragma unroll
        for (int b = 0; b < D::NBH; ++b) accIn[b] = f32x16_zero();
      }
      if (l == 0 && SKIP > 0) {
#pragma unroll
        for (int b = 0; b < D::NB0; ++b) accE[b] += tp_load_blk(a.in0bar_tp, tile, D::NB0, b, lane);
      }
#pragma unroll
      for (int b = 0; b < D::NB0; ++b) tp_store_blk(accE[b], a.in0bar_tp, tile, D::NB0, b, lane);
    }
    if (l > 0) {
      const bool done = l == SKIP;  // zb_l already finished by the in0 gemm above: take it as stored
      auto make = [&](auto bc, const Raw& raw, auto ec) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value, e = decltype(ec)::value;
        float zb;
        if constexpr (TANGENT) zb = done ? raw.b[e] : fmaf(accIn[b][e], act_d1h<D::ACT>(raw.a[e]), raw.b[e]);
        else zb = accIn[b][e] * act_d1h<D::ACT>(raw.a[e]);
        *tp_elem(zbl, tile, D::NBH, b, e, lane) = zb;
        return zb;
      };
#pragma unroll
      for (int b = 0; b < D::NBH; ++b) accOut[b] = f32x16_zero();
      auto next_fetch = [&]() __attribute__((always_inline)) { return bwd_src(l - 1, 0); };
      tp_gemm<D::NBH, D::NBH, Stores<16>, NS, PCS>(accOut, carry, fetch, make, next_fetch, ws, a.p.wpT[l],
                                                   l - 1 == SKIP ? a.p.wpT_in0 : a.p.wpT[l - 1]);
      acc_copy(accIn, accOut);
    }
  }
}
