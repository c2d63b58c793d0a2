// sdfhip — the sdf-only geometry forward (SDFHIP_MODE_SDF: get_sdf, the samplers' callbacks, dense SDF grids for mesh extraction) on TWO
// waves per SIMD ("pair-wave" form, round 5; the structure of tools/probe_pair.hip carried into the product network).
//
// The one-wave-per-SIMD kernels (geo_kernels.h) keep a 32-point tile's whole activation state in one wave's registers; the wave's
// producer work (Softplus = two transcendentals, the hi / lo split) and its MFMAs then share one instruction stream and do not
// overlap: matrix pipe 54 % + VALU 38 % busy in the MODE_SDF kernel, which saves nothing and is bound by exactly that
// (profiles/r5_eval_pmc_summary.csv).  Here waves w and w + 4 of an 8-wave workgroup - the SAME SIMD: a workgroup's waves go to the SIMDs
// round robin - co-own tile w: ROLE r = wave >> 2 owns the out-blocks of parity r (4 accumulator blocks of 16 registers instead of 8:
// two sets fit 256 registers), produces the input blocks of parity r of the next layer just in time from them and publishes them through
// a 2-slot LDS ring per tile; both waves read every B operand of a hidden layer from the ring.  One wave's producer work runs under the
// other's MFMAs.  Weights stream L2 -> LDS by DMA as in the one-wave kernels (same packed chunks, all 8 waves share a chunk and move a
// quarter... an eighth of it each), one s_barrier per k-block step.
//
// Every wait DRAINS the wave's memory queue (vmcnt(0) lgkmcnt(0)) before the barrier: mlp_core.h explains why no counted form is used.
// The in0 blocks (layer 0 and the skip layer's in0 columns) come from HBM, not from a partner's accumulators: both waves load and split
// them themselves.  The sdf row is a lane-local dot product over the wave's own blocks; the two roles' partial sums meet in LDS.
#pragma once
#include <cstdlib>

#include "geo_kernels.h"

template <class D>
struct PairLds {
  static constexpr int NS = kNsFwd;
  static constexpr int buf_floats = chunk_pieces(D::NBH, NS) * 256;  // one weight chunk: NBH out-blocks x parts x 2 k halves x 1 KiB
  static constexpr int slot_floats = ns_parts(NS) * 2 * 64 * 4;      // one B block in operand form: parts x k halves x 64 lanes x 16 B
  static constexpr int ring_floats = 4 * 2 * slot_floats;            // 4 tiles x 2 slots
  static constexpr int floats(int nl) { return 2 * buf_floats + ring_floats + (nl + 2) * D::CW + 4 * 32; }
};

template <class D, int ROLE>
SDFHIP_D void geo_sdf_pair_body(const GeoFwdArgs& a, float* lds, const int wave, const int lane, const int64_t tile) {
  constexpr int NS = kNsFwd, NP = ns_parts(NS);
  static_assert(NS == 4, "the pair-wave forward is written for fp16 hi + lo parts (three product terms)");
  static_assert(D::ACT == 0, "Softplus networks");
  constexpr int NB = D::NBH, NB0 = D::NB0, HB = NB / 2;
  static_assert(NB % 2 == 0 && NB >= 4, "out-blocks split by parity over the two roles");
  constexpr int NT = 3;
  constexpr int ta[NT] = {1, 0, 0}, tb[NT] = {0, 1, 0};  // (weight part, activation part) of every product term, smallest first
  constexpr int BUF = PairLds<D>::buf_floats, SL = PairLds<D>::slot_floats, W = D::CW;
  constexpr int PPW = chunk_pieces(NB, NS) / 8;  // DMA pieces (1 KiB) per wave and chunk
  static_assert(chunk_pieces(NB, NS) % 8 == 0 && PPW <= 2 * NT, "the chunk's pieces ride in the first two MFMA groups of a step");
  const int hf = lane >> 5;
  const int NL = a.p.nl, SKIP = a.p.skip;
  float* wbuf = lds;
  float* ring = lds + 2 * BUF + (wave & 3) * 2 * SL;
  float* cvec = lds + 2 * BUF + PairLds<D>::ring_floats;
  float* sdfx = cvec + (NL + 2) * W;
  int cur = 0;  // buffer that holds the chunk the next step multiplies

  auto dma_piece = [&](const float* __restrict__ gsrc, const int buf, const int i) __attribute__((always_inline)) {
    const int piece = i * 8 + wave;  // 1 KiB per wave instruction: lane l supplies bytes [16 l, 16 l + 16)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + piece * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(wbuf + buf * BUF + piece * 256), 16, 0, 0);
  };
  // the chunk about to be multiplied has landed for every wave, the ring slot about to be read has been published, and everybody is done
  // with the other chunk buffer and the other ring slot
  auto step_sync = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto publish = [&](const int slot, const SplitBlk<NS>& b) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) *reinterpret_cast<bf16x8*>(ring + slot * SL + ((q * 2 + kk) * 64 + lane) * 4) = b.p[q][kk];
  };

  f32x16 in[HB], out[HB];  // own blocks: index j <-> block 2 j + ROLE

  // One k-block step: 2 k halves x HB own out-blocks x 3 terms, weight fragments three groups deep in flight from LDS, the next chunk's
  // DMA pieces in the first gaps, and - hidden layers - `ne` elements of the block this wave is producing, in two batches.
  // B: the operand parts of input block kb.  prod(e): produce element e (called for e0 .. e0 + ne - 1, in order).
  auto mma_step = [&](const bf16x8 (&bfr)[NP][2], const float* __restrict__ gnext, auto e0c, auto nec, auto&& prod) __attribute__((always_inline)) {
    constexpr int e0 = decltype(e0c)::value, ne = decltype(nec)::value;
    const float* wcur = wbuf + cur * BUF + lane * 4;
    bf16x8 af[3][NP];
    auto load_a = [&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value, kk = g / HB, i = g % HB;
#pragma unroll
      for (int q = 0; q < NP; ++q) af[g % 3][q] = *reinterpret_cast<const bf16x8*>(wcur + ((q * NB + 2 * i + ROLE) * 2 + kk) * 256);
    };
    load_a(IC<0>{});
    load_a(IC<1>{});
    constexpr int NG = 2 * HB, NBATCH = ne / 4;
    static_for<0, NG>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value, kk = g / HB, i = g % HB;
      if constexpr (g + 2 < NG) load_a(IC<(g + 2 < NG ? g + 2 : 0)>{});
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, m = g * NT + t;
        out[i] = mfma_term<NS>(af[g % 3][ta[t]], bfr[tb[t]][kk], out[i]);
        if constexpr (m < PPW) dma_piece(gnext, cur ^ 1, m);
      });
      if constexpr (NBATCH > 0) {
        constexpr int jlo = (g * NBATCH) / NG, jhi = ((g + 1) * NBATCH) / NG;  // batch j follows group floor((j + 1) NG / NBATCH) - 1
        if constexpr (jlo < jhi) {
          __builtin_amdgcn_sched_barrier(0);
          static_for<jlo * 4, jhi * 4>([&](auto jc) __attribute__((always_inline)) { prod(IC<e0 + decltype(jc)::value>{}); });
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    cur ^= 1;
  };

  // hidden -> hidden: input block kb = Softplus of the layer below's block kb, produced by the role that owns it, half a block per step
  // (second half of block kb + 1, else first half of block kb + 2), published to the ring slot (kb + 1) & 1 at the end of step kb
  auto hidden_gemm = [&](const float* __restrict__ wp, const float* __restrict__ next_wp) __attribute__((always_inline)) {
    SplitBlk<NS> np;
    if constexpr (ROLE == 0) {
      static_for<0, 16>([&](auto ec) __attribute__((always_inline)) {
        split_put<NS, decltype(ec)::value>(np, InRange{act_h<D::ACT, true>(in[0][decltype(ec)::value])});
      });
      publish(0, np);
    } else {
      static_for<0, 8>([&](auto ec) __attribute__((always_inline)) {
        split_put<NS, decltype(ec)::value>(np, InRange{act_h<D::ACT, true>(in[0][decltype(ec)::value])});
      });
    }
    static_for<0, NB>([&](auto kbc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value;
      constexpr bool own1 = kb + 1 < NB && ((kb + 1) & 1) == ROLE;
      constexpr bool own2 = kb + 2 < NB && ((kb + 2) & 1) == ROLE;
      constexpr int pb = own1 ? kb + 1 : kb + 2;  // block being produced
      constexpr int e0 = own1 ? 8 : 0, ne = (own1 || own2) ? 8 : 0;
      step_sync();
      bf16x8 bfr[NP][2];
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) bfr[q][kk] = *reinterpret_cast<const bf16x8*>(ring + (kb & 1) * SL + ((q * 2 + kk) * 64 + lane) * 4);
      const float* gnext = kb + 1 < NB ? wp + (size_t)(kb + 1) * NB * kChunkBlockFloats : next_wp;
      mma_step(bfr, gnext, IC<e0>{}, IC<ne>{}, [&](auto ec) __attribute__((always_inline)) {
        constexpr int e = decltype(ec)::value;
        split_put<NS, e>(np, InRange{act_h<D::ACT, true>(in[(pb >> 1) < HB ? (pb >> 1) : 0][e])});
      });
      if constexpr (own1) publish((kb + 1) & 1, np);
    });
  };

  // in0 -> hidden (layer 0; the in0 columns of the skip layer): the B blocks are tile-packed in0 from HBM, loaded and split by both roles
  auto in0_gemm = [&](const float* __restrict__ wp, const float* __restrict__ next_wp) __attribute__((always_inline)) {
    Raw raw = load_src(BlkSrc<1>{{tp_block_ptr(a.in0_tp, tile, NB0, 0)}}, lane);
    static_for<0, NB0>([&](auto kbc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value;
      step_sync();  // (drains the block's loads as well)
      SplitBlk<NS> blk;
      static_for<0, 16>([&](auto ec) __attribute__((always_inline)) { split_put<NS, decltype(ec)::value>(blk, raw.a[decltype(ec)::value]); });
      if constexpr (kb + 1 < NB0) raw = load_src(BlkSrc<1>{{tp_block_ptr(a.in0_tp, tile, NB0, kb + 1 < NB0 ? kb + 1 : 0)}}, lane);
      const float* gnext = kb + 1 < NB0 ? wp + (size_t)(kb + 1) * NB * kChunkBlockFloats : next_wp;
      mma_step(blk.p, gnext, IC<0>{}, IC<0>{}, [&](auto) __attribute__((always_inline)) {});
    });
  };

#pragma unroll 1
  for (int l = 0; l < NL; ++l) {
    {
      const float* bias = cvec + l * W;
#pragma unroll
      for (int j = 0; j < HB; ++j) out[j] = tp_rowvec_blk(bias, 2 * j + ROLE, hf);
    }
    const float* after = l + 1 < NL ? a.p.wp[l + 1] : a.p.wp[0];  // after the last layer: a chunk nobody multiplies (the stream never branches)
    if (l > 0) hidden_gemm(a.p.wp[l], l == SKIP ? geo_skip_in0<D>(a.p.wp[l]) : after);
    if (l == 0 || l == SKIP) in0_gemm(l == 0 ? a.p.wp[0] : geo_skip_in0<D>(a.p.wp[l]), after);
#pragma unroll
    for (int j = 0; j < HB; ++j) in[j] = out[j];
  }

  // the sdf row: lane-local dot product over the own blocks of Softplus(z_{NL-1}); the two roles' sums meet in LDS
  float part = 0.0f;
  const float* wsdf = cvec + (NL + 1) * W;
  static_for<0, HB>([&](auto jc) __attribute__((always_inline)) {
    static_for<0, 16>([&](auto ec) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value, e = decltype(ec)::value;
      part = fmaf(wsdf[(2 * j + ROLE) * 32 + tp_row(e, hf)], act_h<D::ACT, true>(in[j][e]), part);
    });
  });
  part += __shfl_xor(part, 32);
  if constexpr (ROLE == 1) {
    if (hf == 0) sdfx[(wave & 3) * 32 + lane] = part;
  }
  step_sync();
  if constexpr (ROLE == 0) {
    if (hf == 0) a.sdf[tile * 32 + lane] = part + sdfx[(wave & 3) * 32 + lane] + a.p.b_sdf[0];
  }
}

template <class D>
__global__ __launch_bounds__(512, 2) void geo_sdf_pair_kernel(const GeoFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + (wave & 3);
  constexpr int NS = kNsFwd, BUF = PairLds<D>::buf_floats;
  {
    // first chunk (layer 0, in0 block 0) into buffer 0: 8 waves x PPW pieces
    constexpr int PPW = chunk_pieces(D::NBH, NS) / 8;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = i * 8 + wave;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.p.wp[0] + piece * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(lds + piece * 256), 16, 0, 0);
    }
  }
  geo_stage_cvec<D>(lds + 2 * BUF + PairLds<D>::ring_floats, a.p, tid);
  __syncthreads();
  if (wave < 4) geo_sdf_pair_body<D, 0>(a, lds, wave, lane, tile);
  else geo_sdf_pair_body<D, 1>(a, lds, wave, lane, tile);
}

// SDFHIP_PAIR_SDF=0 / 1 selects the one-wave / pair-wave form of the MODE_SDF forward at run time (same-box A/Bs, the parity test that
// holds the two against each other); read per launch: a getenv is nanoseconds beside a launch
#ifndef SDFHIP_PAIR_SDF_DEFAULT
#define SDFHIP_PAIR_SDF_DEFAULT 0
#endif
static inline bool sdfhip_pair_sdf_enabled() {
  const char* e = getenv("SDFHIP_PAIR_SDF");
  return e != nullptr ? atoi(e) != 0 : SDFHIP_PAIR_SDF_DEFAULT != 0;
}
