// sdfhip — MFMA core shared by the fused geometry / colour network kernels (gfx950).
//
// One workgroup = 4 waves = 128 points; one wave per SIMD, each wave owns a 32-point tile.  A fused network is a
// sequence of "gemms"   acc_out[0..NBO) (+)= W (NBO x KB blocks) * B[0..KB)   where the k-th 32-wide input block B[kb]
// is PRODUCED just in time from the previous layer's accumulators by a small functor (activation, derivative scaling,
// loads of saved tensors, stores of tensors the backward needs):
//
//   for kb in 0..KB (unrolled WITHIN a gemm, every register index a compile-time constant; the layers around the gemms are run-time loops):
//       wait for weight chunk kb in LDS: s_waitcnt vmcnt(0) - the wave's WHOLE vector-memory queue, i.e. the chunk's LDS-DMA and every
//       load / store issued around it - then a bare s_barrier (all four waves' pieces have landed, everybody is done with the other buffer)
//       start the LDS-DMA of the next chunk (possibly the first chunk of the NEXT gemm: the weight stream itself never stalls on a layer boundary)
//       issue the global loads the producer of block kb + 2 needs
//       produce block kb + 1 (VALU) and split it into bf16 parts, interleaved with
//       2 x NBO x {3 | 6}  v_mfma_f32_32x32x16_{bf16,f16}  with A = weight fragments from LDS (ds_read_b128: 8 k per read),
//                                                        B = the parts of block kb (own registers)
//
// Numerics: fp32 in, fp32 accumulate, products by SPLIT 16-bit parts.  x = x0 + x1 (+ x2) with x0 = round16(x), x1 = round16(x - x0), ...
// (a product of two bf16 / fp16 values is exact in fp32), and
//   NS = 2:  bf16 hi + lo, w x ~ w1 x0 + w0 x1 + w0 x0                      (3 MFMAs, relative product error ~2^-17)
//   NS = 3:  three bf16 parts, w1 x1 + w2 x0 + w0 x2 + w1 x0 + w0 x1 + w0 x0 (6 MFMAs, all 24 mantissa bits: fp32-class)
//   NS = 4:  fp16 hi + lo, the same three terms as NS = 2                    (3 MFMAs, 22 mantissa bits: fp32-class for O(1) data)
// The matrix pipe runs 16-bit inputs 16x faster than fp32 (MI355X: 2.5 PFLOP/s vs 157 TFLOP/s), so the 3-term forms are ~5x cheaper
// than v_mfma_f32_32x32x2_f32.  The forward passes (whose outputs have parity targets: 1e-5 on SDF) use NS = 4; the derivative /
// gradient passes, whose operands have arbitrary scale, use NS = 2.  Activations never leave the register file between layers and
// the weights move L2 -> LDS by DMA (buffer_load_dwordx4 ... lds: WStream::dma_piece) without passing through registers.
//
// Weight chunk layout (pack_kernel), 16-bit:  Wp[kb][part 0..4 = bf16 x 3, fp16 x 2][ob][kk 0..1][lane][j 0..7]
//   = part of  W[out = 32 ob + (lane & 31)][k = 32 kb + tp_row(8 kk + j, lane >> 5)]
// One ds_read_b128 is the A operand of one MFMA; a gemm streams only the parts of its mode (chunk_part_offset).
#pragma once
#include "common.h"

template <int N>
using IC = std::integral_constant<int, N>;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));  // also the raw 16-byte carrier of 8 fp16 (NS == 4)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Precision modes ("NS").  2: bf16 hi + lo, 3-term products.  3: three bf16 parts, 6-term products (all 24 mantissa bits).
// 4: fp16 hi + lo, 3-term products: 22 mantissa bits (absolute 2^-25 below 0.125), measured on the 8 x 256^2 softplus stack
// at 5.0e-7 of the result against 4.2e-7 for the 6-term bf16 form and 5.6e-6 for the 3-term bf16 form
// (tools/probe_split.hip -DPROBE_F16), at HALF the matrix instructions of the 6-term form.  fp16 has no exponent headroom
// (max 65504, the split saturates there), so it serves the forward passes (activations, d sdf / dx chain: O(1) data); the
// backward passes carry loss gradients of arbitrary scale and stay on bf16 parts.
SDFHIP_HD constexpr int ns_parts(const int ns) { return ns == 4 ? 2 : ns; }   // operand parts held / streamed
SDFHIP_HD constexpr int ns_first(const int ns) { return ns == 4 ? 3 : 0; }    // first part of a packed chunk the mode reads
constexpr int kChunkParts = 5;  // packed chunk: bf16 parts 0..2, then fp16 parts 0..1
constexpr int kChunkBlockFloats = kChunkParts * 512;  // HBM: one (k block, out block) pair = 5 parts x 32 x 32 16-bit weights
// KiB pieces the DMA moves for one chunk of nbo out-blocks in mode ns (4 waves x 1 KiB per instruction round)
SDFHIP_HD constexpr int chunk_pieces(const int nbo, const int ns) { return (nbo * ns_parts(ns) * 2 + 3) / 4 * 4; }
// offset (floats) of the parts mode ns streams inside a packed chunk of nbo out-blocks: the host adds it to the weight pointers
SDFHIP_HD constexpr int chunk_part_offset(const int nbo, const int ns) { return ns_first(ns) * nbo * 512; }

struct WStream {
  float* lds;        // two chunk buffers of buf_floats each
  int buf_floats;
  int cur;           // buffer holding the chunk that the next mfma step consumes
  int wave, lane;

  // ONE 1 KiB DMA instruction: piece `piece` of the chunk at gsrc -> the same piece of the LDS buffer at dst; lane l moves bytes [16 l, 16 l + 16).
  // The BUFFER form of the LDS-DMA (buffer_load_dwordx4 ... lds), not the global form (global_load_lds_dwordx4) rounds 1 - 5 used: hipcc
  // (ROCm 7.2) books a global_load_lds as a FLAT access that touches both memory and LDS, and while one is pending it turns EVERY s_waitcnt it
  // inserts - vmcnt and lgkmcnt - into a full drain ("pending flat").  A weight chunk's DMA is pending through the whole step before it, so all
  // 161 LDS waits of the MODE_SDF kernel were lgkmcnt(0): each group's first MFMA waited for the NEXT group's eight ds_read_b128 as well - the
  // two-deep weight-fragment registers bought nothing, and four LDS round trips per step were exposed.  The MUBUF form is an ordinary VMEM
  // operation to the compiler: exact lgkmcnt(N) / vmcnt(N) counts come back (the hand-written vmcnt(0) + s_barrier of wait_sync stays: it is
  // what makes the chunk visible).  -DSDFHIP_GLDS_FLAT selects the old form for A/B runs.
  SDFHIP_D void dma_piece(const float* __restrict__ gsrc, float* dst, const int piece) const {
#ifdef SDFHIP_GLDS_FLAT
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + piece * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(dst + piece * 256), 16, 0, 0);
#else
    // raw buffer over the packed weights: stride 0, no range check that could bite (the chunks are over-read into slack by design), gfx9 word 3
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gsrc), 0, -1, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + piece * 256), 16, lane * 16, piece * 1024, 0, 0);
#endif
  }
  // Start the DMA of `pieces` KiB (a multiple of 4: every wave issues the same count, so vmcnt bookkeeping is uniform)
  // from gsrc into the buffer that is NOT current.
  SDFHIP_D void issue(const float* __restrict__ gsrc, const int pieces, const bool into_current = false) {
    float* dst = lds + ((into_current ? cur : cur ^ 1) * buf_floats);
    for (int i = 0; i < pieces; i += 4) dma_piece(gsrc, dst, i + wave);  // 1 KiB per wave instruction
  }
  // one DMA instruction of the chunk going into the buffer that is NOT current: piece 4 j + wave
  SDFHIP_D void issue_piece(const float* __restrict__ gsrc, const int j) { dma_piece(gsrc, lds + (cur ^ 1) * buf_floats, 4 * j + wave); }
  // The weight chunk about to be consumed has landed in LDS for every wave, and every wave is done reading the other
  // buffer.  NEWER = vector-memory operations allowed to stay outstanding: 0 in the product (tp_gemm explains why no counted form -
  // "the operations issued after the chunk's DMA may stay in flight" - is safe here).
  // A bare s_barrier (no workgroup release fence) keeps the compiler from adding its own vmcnt(0) in front of it.
  template <int NEWER>
  SDFHIP_D void wait_sync() {
    static_assert(NEWER >= 0 && NEWER < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEWER) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  SDFHIP_D const float* current() const { return lds + cur * buf_floats + lane * 4; }
  SDFHIP_D void flip() { cur ^= 1; }
};

// One 32-feature activation block split into the parts of mode NS; p[q][kk] is the B operand of the MFMAs of k half kk.
template <int NS>
struct SplitBlk {
  bf16x8 p[ns_parts(NS)][2];
  float pend;  // the even element of a pair, waiting for its odd neighbour (split_put converts element pairs with packed conversions)
};
// one product term of mode NS: acc += A * B
template <int NS>
SDFHIP_D f32x16 mfma_term(const bf16x8 a, const bf16x8 b, const f32x16 acc) {
  if constexpr (NS == 4) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

// What a producer reads from HBM for one input block: up to two TP blocks (e.g. z_l and zc_l); unused members cost nothing.
struct Raw {
  f32x16 a, b;
};
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) const float gcfloat;
// Where those blocks live: N wave-uniform pointers to the first element of the wave's TP block (tp_block_ptr).  The fused
// kernels hand these to tp_gemm, which spreads the 16 N element loads over the MFMAs of a step.
template <int N>
struct BlkSrc {
  static constexpr int n = N;
  const gfloat* p[N > 0 ? N : 1];
};
struct NoFetch {  // producer without HBM operands
  template <class... K>
  SDFHIP_D BlkSrc<0> operator()(K...) const {
    return BlkSrc<0>{};
  }
};
template <int E, int N>
SDFHIP_D void load_src_elem(const BlkSrc<N>& src, Raw& r, const int lane) {
  if constexpr (N >= 1) r.a[E] = src.p[0][(unsigned)lane + (unsigned)E * 64u];
  if constexpr (N >= 2) r.b[E] = src.p[1][(unsigned)lane + (unsigned)E * 64u];
}
template <int N>
SDFHIP_D Raw load_src(const BlkSrc<N>& src, const int lane) {
  Raw r;
  static_for<0, 16>([&](auto ec) __attribute__((always_inline)) { load_src_elem<decltype(ec)::value>(src, r, lane); });
  return r;
}

// acc[0..NBO) += W * B with B[kb] produced just in time.  The producer of a block is split so that neither HBM latency nor
// its VALU work ever sits in front of the matrix pipe (one wave per SIMD: nothing else would cover it):
//   fetch(IC<kb>) -> BlkSrc<N>         where the HBM operands of block kb live; their 16 N element loads are issued during
//                                      step kb - 2, one or two per MFMA gap
//   make(IC<kb>, Raw, IC<e>) -> float  element e of the block (VALU + the element's stores); block kb + 1 is produced
//                                      during step kb, the compiler interleaves it with the step's MFMAs
//   next_fetch() -> BlkSrc<N>          operands of block 0 of the FOLLOWING gemm, loaded during this gemm's last step;
//                                      they travel in `carry`, which on entry holds this gemm's own block-0 operands
//   ST::at(kb)                         number of global stores make(IC<kb>, ..) issues (bookkeeping for the experimental counted waits
//                                      below; the product wait drains the queue and does not depend on it)
//   NS                                 bf16 parts per operand (2: 3-term products, 3: 6-term, fp32-class)
// wp: this gemm's packed weights (first chunk already in flight / landed in the current buffer).
// next_wp / NEXTP: first chunk of the gemm that follows, chunk_pieces(its NBO, its NS) (0: none).
// global stores make(IC<kb>, ..) issues for one block: A for blocks kb < FROM, B for the rest
template <int A, int B = A, int FROM = 1 << 30>
struct Stores {
  static constexpr int at(int kb) { return kb < FROM ? A : B; }  // global stores make(IC<kb>, ..) issues
};

// out-blocks per operand-read group: a divisor of NBO, small enough that two groups of weight fragments (the one being
// multiplied and the one in flight from LDS) fit the register budget next to 256 accumulator registers
constexpr int gemm_group(const int nbo, const int ns, const int gcap = 4) {
  const int gmax = (ns_parts(ns) == 3 ? 2 : 4) < gcap ? (ns_parts(ns) == 3 ? 2 : 4) : gcap;
  for (int g = gmax; g > 1; --g)
    if (nbo % g == 0) return g;
  return nbo <= 5 ? nbo : 1;
}

// element e (TP register index) of a block under construction -> its slot in the split operand.  Elements arrive in order
// e = 0 .. 15; an even element waits in `pend` and is converted TOGETHER with its odd neighbour: the two share one dword of the
// operand vector, so hi and lo parts are one packed conversion each (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32, round to nearest even
// like the scalar conversions: same numbers) instead of two scalar conversions and a pack apiece.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// A producer whose value is known to lie inside fp16's finite range (an activation bounded on the way: common.h act_h<ACT, true>)
// returns it wrapped in InRange: the split then skips the saturation (one v_med3_f32 per element).
struct InRange {
  float v;
};
// lo parts of an fp16 hi / lo pair:  lo_i = fp16(x_i - hi_i).  The difference is formed and rounded by ONE instruction per element,
// v_fma_mix{lo,hi}_f16 (hi_i (f16) * -1 + x_i in fp32 - exact here - rounded to nearest even into the low / high half of the result):
// the same numbers as "convert hi back to fp32, subtract, packed conversion" (rounds 1 - 4: 2 + 2 + 1 instructions per pair).
SDFHIP_D uint32_t f16_lo_pair(const uint32_t hpk, const float x0, const float x1) {
  uint32_t l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hpk), "v"(x0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hpk), "v"(x1));
  return l;
}
template <int NS, int E, bool CLAMP = true>
SDFHIP_D void split_put(SplitBlk<NS>& s, float r) {
  if constexpr (NS == 4 && CLAMP) r = __builtin_amdgcn_fmed3f(r, -65504.0f, 65504.0f);  // fp16 has no exponent headroom: saturate instead of inf
  if constexpr ((E & 1) == 0) {
    s.pend = r;
  } else {
    constexpr int kk = E >> 3, j = (E & 7) - 1;
    f32x2_t v = {s.pend, r};
    // the pair (j, j + 1), j even, is dword j / 2 of the 8-element operand vector: the packed conversion's result goes in whole
    // (element-wise _Float16 -> __bf16 bit casts of the pair miscompile on this toolchain: tools/probe_splitput.hip)
    auto put = [&](bf16x8& dst, const uint32_t w) __attribute__((always_inline)) {
      u32x4_t t = __builtin_bit_cast(u32x4_t, dst);
      t[j >> 1] = w;
      dst = __builtin_bit_cast(bf16x8, t);
    };
    if constexpr (NS == 4) {
      const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
      put(s.p[0][kk], h);
#ifndef SDFHIP_OLD_SPLIT
      put(s.p[1][kk], f16_lo_pair(h, v[0], v[1]));
#else
      const f16x2_t l = __builtin_convertvector(v - __builtin_convertvector(__builtin_bit_cast(f16x2_t, h), f32x2_t), f16x2_t);
      put(s.p[1][kk], __builtin_bit_cast(uint32_t, l));
#endif
    } else {
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
        put(s.p[q][kk], __builtin_bit_cast(uint32_t, h));
        if (q + 1 < NS) v -= __builtin_convertvector(h, f32x2_t);
      }
    }
  }
}
template <int NS, int E>
SDFHIP_D void split_put(SplitBlk<NS>& s, const InRange r) {
  split_put<NS, E, false>(s, r.v);
}

// GCAP: upper bound on the out-blocks per operand-read group (kernels that live in 256 registers - two workgroups per CU,
// wide_kernels.h - hold two groups of 2 x 2 weight fragments instead of 2 x 4)
template <int KB, int NBO, class ST, int NS, int NEXTP, int GCAP = 4, int MAXA, class Fetch, class Make, class NextFetch>
SDFHIP_D void tp_gemm(f32x16 (&acc)[MAXA], Raw& carry, Fetch&& fetch, Make&& make, NextFetch&& next_fetch, WStream& ws,
                      const float* __restrict__ wp, const float* __restrict__ next_wp) {
  static_assert(NBO <= MAXA, "accumulator tile too small");
  static_assert(NS == 2 || NS == 3 || NS == 4, "bf16 x 2, bf16 x 3 or fp16 x 2 parts");
  static_assert(NEXTP % 4 == 0, "whole DMA rounds");
  // (weight part, activation part) of every product term, smallest magnitude first
  constexpr int NP = ns_parts(NS);
  constexpr int NT = NP == 2 ? 3 : 6;
  constexpr int ta[6] = {1, NP == 2 ? 0 : 2, 0, 1, 0, 0};
  constexpr int tb[6] = {NP == 2 ? 0 : 1, NP == 2 ? 1 : 0, NP == 2 ? 0 : 2, 0, 1, 0};
  constexpr int G = gemm_group(NBO, NS, GCAP), NG = NBO / G;  // groups per k half
  static_assert(NBO % G == 0, "operand groups must tile the out-blocks");
  constexpr int NGRP = 2 * NG, MPG = NT * G, NM = NGRP * MPG;  // groups, MFMAs per group, MFMAs per step
  const int lane = ws.lane;
  // r1: operands of the block made during the current step; r2: operands of the block after that (in flight)
  Raw r1 = carry, r2;
  if constexpr (KB > 1) r2 = load_src(fetch(IC<(KB > 1 ? 1 : 0)>{}), lane);
  SplitBlk<NS> blk;
  static_for<0, 16>([&](auto ec) __attribute__((always_inline)) { split_put<NS, decltype(ec)::value>(blk, make(IC<0>{}, r1, ec)); });
  static_for<0, KB>([&](auto kbc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kbc)::value;
    constexpr bool more = kb + 1 < KB;
    {
      // vector-memory operations issued after the DMA of chunk kb: the loads of fetch(kb + 1), then the stores of make(kb)
      constexpr int newer_ = ST::at(kb) + (more ? 16 * decltype(fetch(IC<(more ? kb + 1 : 0)>{}))::n : 0);
      // The wait DRAINS the wave's vector-memory queue (vmcnt(0)).  Rounds 1 - 2 waited with the count above ("at most `newer_`
      // operations outstanding": what was issued after the chunk's DMA may stay in flight), on the premise that vmcnt retires in issue
      // order.  It does not, in two ways, both established with the bit-reproducibility tests (tests/test_gpu_bitrepro.py,
      // test_field_forward_is_bit_reproducible): stores retire out of order with respect to loads (round 3: one no-grad forward in five of
      // the 64-wide golden network came back with a workgroup's colour outputs off by ~5e-5 - a gemm had started on a weight chunk that
      // was still landing), and an LDS-DMA does not retire in order with the ordinary loads issued after it either (round 4: counting
      // only the younger LOADS and letting every store drain - the one variant the store argument leaves open - fails the same test, 2 of
      // 2 modes, while being 0.07 - 0.2 ms per step faster).  A counted wait is therefore only sound for a queue that holds nothing but
      // LDS-DMAs (the guide's GEMM templates); these kernels interleave the producer's loads and stores with the weight stream, and all
      // four waves of a workgroup own a full register file, so there is no room for a fifth, DMA-only wave.  The drain costs 0.2 ms per
      // config-2 step against the (racy) counted form.
#ifdef SDFHIP_ABL_WAIT_SLACK  // TIMING ABLATION ONLY (racy on purpose, wrong numerics allowed): what the wait costs through the queue it drains
      constexpr int newer = newer_ + SDFHIP_ABL_WAIT_SLACK;
#else
      constexpr int newer = 0;
      (void)newer_;
#endif
      ws.template wait_sync<(newer < 63 ? newer : 63)>();
    }
    r1 = r2;
    // This step's vector-memory work, in issue order: the DMA pieces of the next weight chunk, then the element loads of
    // the block after next (the last step: of block 0 of the next gemm), then - from MFMA S0 on - the stores of the block
    // being produced.  Everything is spread over the MFMA gaps: an LDS-DMA piece or a global load costs the wave tens of
    // issue cycles, and issued in one burst at the head of the step they would leave the matrix pipe idle.
    const auto src = [&]() __attribute__((always_inline)) {
      if constexpr (kb + 2 < KB) return fetch(IC<(kb + 2 < KB ? kb + 2 : 0)>{});
      else if constexpr (!more) return next_fetch();
      else return BlkSrc<0>{};
    }();
    constexpr int NSRC = std::remove_cv_t<std::remove_reference_t<decltype(src)>>::n;
    constexpr int ND = (more ? chunk_pieces(NBO, NS) : NEXTP) / 4;
    const float* dsrc = more ? wp + (size_t)(kb + 1) * NBO * kChunkBlockFloats : next_wp;
    constexpr int NOPS = ND + (NSRC > 0 ? 16 : 0);
    constexpr int P1 = NM / 2 > 0 ? NM / 2 : 1;
    constexpr int PER = NOPS == 0 ? 1 : (NOPS + P1 - 1) / P1;          // memory operations per MFMA gap
    constexpr int S0 = NOPS == 0 ? NM / 4 : (NOPS + PER - 1) / PER;     // first gap that produces an element
    constexpr int W = NM - S0 > 0 ? NM - S0 : 1;
    const float* cur = ws.current();
    bf16x8 a[2][NP][G];
    auto load_group = [&](auto gc) __attribute__((always_inline)) {
      constexpr int gi = decltype(gc)::value, kk = gi / NG, g0 = (gi % NG) * G;
#ifdef SDFHIP_ABLATE_LDS_READS  // timing experiment only (wrong numerics): weight fragments read for the first two groups of a step
      if constexpr (gi >= 2) return;
#endif
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int i = 0; i < G; ++i) a[gi & 1][q][i] = *reinterpret_cast<const bf16x8*>(cur + ((q * NBO + g0 + i) * 2 + kk) * 256);
    };
    load_group(IC<0>{});
    SplitBlk<NS> nxt;
    static_for<0, NGRP>([&](auto gc) __attribute__((always_inline)) {
      constexpr int gi = decltype(gc)::value, kk = gi / NG, g0 = (gi % NG) * G;
      if constexpr (gi + 1 < NGRP) load_group(IC<(gi + 1 < NGRP ? gi + 1 : 0)>{});
      __builtin_amdgcn_sched_barrier(0);  // the LDS reads of the next group stay ahead of this group's MFMAs
      static_for<0, MPG>([&](auto mc) __attribute__((always_inline)) {
        constexpr int mi = decltype(mc)::value, t = mi / G, i = mi % G, m = gi * MPG + mi;
        acc[g0 + i] = mfma_term<NS>(a[gi & 1][ta[t]][i], blk.p[tb[t]][kk], acc[g0 + i]);
        constexpr int olo = m * PER < NOPS ? m * PER : NOPS, ohi = (m + 1) * PER < NOPS ? (m + 1) * PER : NOPS;
        if constexpr (olo < ohi) {
          static_for<olo, ohi>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < ND) {
              ws.issue_piece(dsrc, j);
            } else {
              if constexpr (kb + 2 < KB) load_src_elem<j - ND>(src, r2, lane);
              else load_src_elem<j - ND>(src, carry, lane);
            }
          });
        }
        // element e of the next input block is due after MFMA number S0 + floor(e W / 16)
        constexpr int mm = m - S0;
        constexpr int elo = (!more || mm < 0) ? 0 : (16 * mm + W - 1) / W, ehi_ = (!more || mm < 0) ? 0 : (16 * (mm + 1) + W - 1) / W;
        constexpr int ehi = m + 1 == NM && more ? 16 : (ehi_ < 16 ? ehi_ : 16);
        if constexpr (elo < ehi) {
          static_for<elo, ehi>([&](auto ec) __attribute__((always_inline)) {
            split_put<NS, decltype(ec)::value>(nxt, make(IC<(more ? kb + 1 : 0)>{}, r1, ec));
          });
        }
        if constexpr (olo < ohi || elo < ehi) __builtin_amdgcn_sched_barrier(0);
      });
    });
    ws.flip();
    if constexpr (more) blk = nxt;
  });
}

// Wave-uniform GLOBAL pointer: pins a pointer the compiler cannot prove uniform into scalar registers, so that the global
// loads / stores below use the SGPR-base + 32-bit VGPR offset + immediate form.  One shared offset register (lane * 4) then
// serves every tensor and every element; 64-bit per-element addresses in VGPRs cost the fused kernels their register budget.
// The result is typed address_space(1): after the integer round trip the compiler could no longer infer "global" and
// would fall back to flat_load / flat_store, which also tick lgkmcnt and so stall the LDS -> MFMA stream.
SDFHIP_D gfloat* uniform_gptr(const float* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<gfloat*>(((uint64_t)hi << 32) | lo);
}
// first element of TP block b of the wave's tile:  A[tile][nb blocks][16][64]
SDFHIP_D gfloat* tp_block_ptr(const float* __restrict__ base, const int64_t tile, const int nb, const int b) {
  return uniform_gptr(base + ((size_t)tile * nb + b) * 1024);
}
// element e of TP block b: address of (tile, nb, b, e, lane)
SDFHIP_D gfloat* tp_elem(float* __restrict__ base, const int64_t tile, const int nb, const int b, const int e, const int lane) {
  return tp_block_ptr(base, tile, nb, b) + ((unsigned)lane + (unsigned)e * 64u);
}

// one TP block (16 registers) load / store
SDFHIP_D f32x16 tp_load_blk(const float* __restrict__ base, const int64_t tile, const int nb, const int b, const int lane) {
  const gfloat* p = tp_block_ptr(base, tile, nb, b);
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = p[(unsigned)lane + (unsigned)r * 64u];
  return v;
}
SDFHIP_D void tp_store_blk(const f32x16 v, float* __restrict__ base, const int64_t tile, const int nb, const int b, const int lane) {
  gfloat* p = tp_block_ptr(base, tile, nb, b);
#pragma unroll
  for (int r = 0; r < 16; ++r) p[(unsigned)lane + (unsigned)r * 64u] = v[r];
}
// per-feature vector (natural order, in LDS) broadcast into one TP block: element r <-> feature 32 b + tp_row(r, hf)
SDFHIP_D f32x16 tp_rowvec_blk(const float* v, const int b, const int hf) {
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = v[b * 32 + tp_row(r, hf)];
  return o;
}
SDFHIP_D f32x16 f32x16_zero() {
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.0f;
  return o;
}

// compile-time ping-pong selection
template <bool FIRST, class T>
SDFHIP_D T& pick(T& a, T& b) {
  if constexpr (FIRST) return a;
  else return b;
}
