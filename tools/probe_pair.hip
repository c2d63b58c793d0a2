// Dev probe: the fused 8 x (256 x 256) softplus stack of probe_split.hip restructured for TWO waves per SIMD.
//
// Workgroup = 8 waves = 4 point tiles of 32.  Waves w and w + 4 (same SIMD: a workgroup's waves go to the SIMDs round
// robin) co-own tile w: role r = wave >> 2 owns the out-blocks of parity r (4 accumulator blocks = 64 registers instead
// of 8), produces the input blocks of parity r of the next layer just in time (activation + bf16 split) and publishes
// them through a 2-slot LDS ring; both waves read every B operand from the ring.  Weights stream L2 -> LDS by DMA exactly
// as in the one-wave-per-SIMD kernel (same packed chunk format), all 8 waves share a chunk, one s_barrier per k-block.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "common.h"
void sdfhip_set_error(const char*, ...) {}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef PROBE_L
#define PROBE_L 8
#endif
constexpr int NB = 8, L = PROBE_L;
#ifndef PAIR_ABL
#define PAIR_ABL 0  // timing ablations (wrong numerics): 1 = no weight DMA after the prologue, 2 = no activation / split VALU work
#endif
#ifndef PAIR_NBUF
#define PAIR_NBUF 2  // weight chunk buffers: 2 = the next chunk streams in during a step, 3 = two chunks ahead
#endif
#ifndef PAIR_SB
#define PAIR_SB 1  // 1: sched_barriers keep the MFMA groups / production batches where they are written; 0: compiler schedule
#endif
#ifndef PAIR_BATCH
#define PAIR_BATCH 4  // elements produced per batch
#endif
#ifndef PAIR_BAL
#define PAIR_BAL 1  // 1: every wave produces HALF a block per step (first half kept in registers); 0: a whole block every other step
#endif
template <int N>
using IC = std::integral_constant<int, N>;
template <bool FIRST, class T>
__device__ __forceinline__ T& pick_ref(T& a, T& b) {
  if constexpr (FIRST) return a;
  else return b;
}

template <int NS, bool SAVE, int ROLE>
__device__ __forceinline__ void pair_body(const float* __restrict__ in_tp, const __bf16* __restrict__ wp, float* __restrict__ out_tp,
                                          float* __restrict__ save_tp, __bf16* ldsb, const int wave, const int lane, const int64_t tile) {
  constexpr int CH = NS * NB * 2 * 64 * 8;  // bf16 per weight chunk (one k block, all out blocks): [part][ob][kk][lane][8]
  constexpr int SL = NS * 2 * 64 * 8;       // bf16 per ring slot (one input block): [part][kk][lane][8]
  constexpr int NT = NS == 2 ? 3 : 6;
  constexpr int ta[6] = {1, NS == 2 ? 0 : 2, 0, 1, 0, 0};
  constexpr int tb[6] = {NS == 2 ? 0 : 1, NS == 2 ? 1 : 0, NS == 2 ? 0 : 2, 0, 1, 0};
  constexpr int PPW = NS * 2;  // DMA pieces (1 KiB) per wave per chunk
  __bf16* wbuf = ldsb;
  __bf16* ring = ldsb + PAIR_NBUF * CH + (wave & 3) * 2 * SL;
  auto dma_piece = [&](const __bf16* g, const int buf, const int i) __attribute__((always_inline)) {
    const int piece = i * 8 + wave;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 512 + lane * 8),
                                     (__attribute__((address_space(3))) void*)(wbuf + buf * CH + piece * 512), 16, 0, 0);
  };
  f32x16 accA[NB / 2], accB[NB / 2];  // own blocks: index j <-> block 2 j + ROLE
  {
    const float* p = in_tp + (size_t)tile * NB * 1024 + lane;
#pragma unroll
    for (int j = 0; j < NB / 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[j][r] = p[((2 * j + ROLE) * 16 + r) * 64];
  }
#pragma unroll
  for (int cc = 0; cc < PAIR_NBUF - 1; ++cc)
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma_piece(wp + (size_t)cc * CH, cc, i);

  // element e of input block b of a layer from the previous layer's accumulators (+ the training kernels' z store)
  auto make_elem = [&](const int l, const int b, const int e, const float z) __attribute__((always_inline)) {
    if constexpr (SAVE) save_tp[(((size_t)tile * L + l) * NB + b) * 1024 + e * 64 + lane] = z;
    if constexpr ((PAIR_ABL & 2) != 0) return z;
    return softplus100_h(z * 0.01f) * 100.0f;
  };
  auto publish = [&](const int slot, const bf16x8 (&p)[NS][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NS; ++q)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) *reinterpret_cast<bf16x8*>(ring + slot * SL + ((q * 2 + kk) * 64 + lane) * 8) = p[q][kk];
  };
  auto put = [&](bf16x8 (&p)[NS][2], const int e, float r) __attribute__((always_inline)) {
    if constexpr ((PAIR_ABL & 2) != 0) {
      if (e == 0) {
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) p[q][kk][j] = (__bf16)r;
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const __bf16 h = (__bf16)r;
      p[q][e >> 3][e & 7] = h;
      if (q + 1 < NS) r -= (float)h;
    }
  };

  static_for<0, L>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value;
    auto& in = pick_ref<(l % 2) == 0>(accA, accB);
    auto& out = pick_ref<(l % 2) == 0>(accB, accA);
#pragma unroll
    for (int j = 0; j < NB / 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[j][r] = 0.0f;
    bf16x8 np[NS][2];  // the block this wave is producing
    if constexpr (ROLE == 0) {  // block 0 of this layer (slot 0 was last read two barriers ago)
      static_for<0, 16>([&](auto ec) __attribute__((always_inline)) { put(np, decltype(ec)::value, make_elem(l, 0, decltype(ec)::value, in[0][decltype(ec)::value])); });
      publish(0, np);
    } else if constexpr (PAIR_BAL) {  // first half of block 1
      static_for<0, 8>([&](auto ec) __attribute__((always_inline)) { put(np, decltype(ec)::value, make_elem(l, 1, decltype(ec)::value, in[0][decltype(ec)::value])); });
    }
    static_for<0, NB>([&](auto kbc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value;
      constexpr int c = l * NB + kb;  // chunk index
      // what this wave produces during this step: PAIR_BAL: the second half (elements 8..15) of block kb + 1 if it owns it,
      // else the first half of block kb + 2 (kept in registers until the next step); !PAIR_BAL: all of block kb + 1 if owned
      constexpr bool own1 = kb + 1 < NB && ((kb + 1) & 1) == ROLE;
      constexpr bool own2 = PAIR_BAL && kb + 2 < NB && ((kb + 2) & 1) == ROLE;
      constexpr int pb = own1 ? kb + 1 : kb + 2;                              // block being produced
      constexpr int e0 = PAIR_BAL ? (own1 ? 8 : 0) : 0, ne = (own1 || own2) ? (PAIR_BAL ? 8 : 16) : 0;
      // z stores this wave issued after the LAST DMA piece of chunk c.  The chunk's pieces ride behind MFMAs 0 .. PPW-1 of the
      // previous step, each ahead of that slot's elements; element j of ne follows MFMA floor(j NM / ne).  Before step 0 the
      // stores are those of the top-of-layer production (the previous layer's last step produces nothing).
      constexpr bool pown1 = ((kb & 1) == ROLE), pown2 = PAIR_BAL && kb + 1 < NB && (((kb + 1) & 1) == ROLE);
      constexpr int pne = (pown1 || pown2) ? (PAIR_BAL ? 8 : 16) : 0;
      constexpr int NMc = 8 * NT;
      // (batches follow their MFMA group: batch j after group floor((j + 1) 8 / NBATCH) - 1 >= the group holding the last piece
      // whenever NBATCH <= 4, so every store of the previous step is newer than the chunk's DMA)
      constexpr int zs_prev = pne;
      static_assert(PPW <= 2 * NT, "the DMA pieces must sit in the first two MFMA groups");
      constexpr int zs_top = ROLE == 0 ? 16 : (PAIR_BAL ? 8 : 0);
      constexpr int newer = !SAVE ? 0 : (kb == 0 ? zs_top : zs_prev);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((PAIR_ABL & 1) ? 0 : newer + (PAIR_NBUF - 2) * PPW) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const __bf16* wcur = wbuf + (c % PAIR_NBUF) * CH + lane * 8;
      const __bf16* rcur = ring + (kb & 1) * SL + lane * 8;
      bf16x8 bfr[NS][2];
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) bfr[q][kk] = *reinterpret_cast<const bf16x8*>(rcur + (q * 2 + kk) * 512);
      constexpr int cn = c + PAIR_NBUF - 1;
      const __bf16* gnext = wp + (size_t)(cn < L * NB ? cn : L * NB - 1) * CH;
      bf16x8 a[3][NS];  // weight fragments: the group being multiplied and the next two, in flight from LDS
      auto load_a = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, kk = g / 4, i = g % 4;
#pragma unroll
        for (int q = 0; q < NS; ++q) a[g % 3][q] = *reinterpret_cast<const bf16x8*>(wcur + ((q * NB + 2 * i + ROLE) * 2 + kk) * 512);
      };
      load_a(IC<0>{});
      load_a(IC<1>{});
      // production: batches of PAIR_BATCH elements (independent dependency chains the compiler interleaves) after every
      // 8 / nbatch-th MFMA group; the partner wave's MFMAs cover them
      constexpr int NBATCH = ne / PAIR_BATCH;
      static_for<0, 8>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, kk = g / 4, i = g % 4;
        if constexpr (g + 2 < 8) load_a(IC<(g + 2 < 8 ? g + 2 : 0)>{});
        if constexpr (PAIR_SB) __builtin_amdgcn_sched_barrier(0);
        static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
          constexpr int t = decltype(tc)::value;
          out[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g % 3][ta[t]], bfr[tb[t]][kk], out[i], 0, 0, 0);
          constexpr int m = g * NT + t;
          if constexpr (m < PPW && (PAIR_ABL & 1) == 0) dma_piece(gnext, cn % PAIR_NBUF, m);  // the next chunk's DMA pieces ride in the first gaps
        });
        // batch j of NBATCH after group floor((j + 1) 8 / NBATCH) - 1
        if constexpr (NBATCH > 0) {
          constexpr int jlo = (g * NBATCH) / 8, jhi = ((g + 1) * NBATCH) / 8;
          if constexpr (jlo < jhi) {
            if constexpr (PAIR_SB) __builtin_amdgcn_sched_barrier(0);
            static_for<jlo * PAIR_BATCH, jhi * PAIR_BATCH>([&](auto jc) __attribute__((always_inline)) {
              constexpr int e = e0 + decltype(jc)::value;
              put(np, e, make_elem(l, pb, e, in[pb >> 1][e]));
            });
          }
        }
        if constexpr (PAIR_SB) __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (own1) publish((kb + 1) & 1, np);
    });
  });
  auto& fin = pick_ref<(L % 2) == 0>(accA, accB);
  {
    float* p = out_tp + (size_t)tile * NB * 1024 + lane;
#pragma unroll
    for (int j = 0; j < NB / 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) p[((2 * j + ROLE) * 16 + r) * 64] = fin[j][r];
  }
}

template <int NS, bool SAVE>
__global__ __launch_bounds__(512, 2) void pair_kernel(const float* __restrict__ in_tp, const __bf16* __restrict__ wp, float* __restrict__ out_tp,
                                                      float* __restrict__ save_tp, unsigned long long* __restrict__ clk) {
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsb[];
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + (wave & 3);
  if (wave < 4) pair_body<NS, SAVE, 0>(in_tp, wp, out_tp, save_tp, ldsb, wave, lane, tile);
  else pair_body<NS, SAVE, 1>(in_tp, wp, out_tp, save_tp, ldsb, wave, lane, tile);
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
    clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
  }
}

static float act_ref(double z) { return z > 20 ? z : log1p(exp(z)); }

template <int NS, bool SAVE>
static void run(const char* name, int64_t P, const float* d_in, float* d_out, float* d_save, unsigned long long* d_clk, const std::vector<float>& W,
                const std::vector<float>& X) {
  constexpr int CH = NS * NB * 2 * 64 * 8, SL = NS * 2 * 64 * 8;
  std::vector<__bf16> Wp((size_t)L * NB * CH);
  for (int l = 0; l < L; ++l)
    for (int kb = 0; kb < NB; ++kb)
      for (int ob = 0; ob < NB; ++ob)
        for (int kk = 0; kk < 2; ++kk)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int o = ob * 32 + (lane & 31), k = kb * 32 + tp_row(kk * 8 + j, lane >> 5);
              float r = W[((size_t)l * 256 + o) * 256 + k];
              for (int q = 0; q < NS; ++q) {
                const __bf16 h = (__bf16)r;
                Wp[(size_t)(l * NB + kb) * CH + (((q * NB + ob) * 2 + kk) * 64 + lane) * 8 + j] = h;
                r -= (float)h;
              }
            }
  __bf16* d_wp;
  hipMalloc(&d_wp, Wp.size() * 2);
  hipMemcpy(d_wp, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice);
  const unsigned grid = (unsigned)(P / 128);
  const size_t lds = (size_t)(PAIR_NBUF * CH + 8 * SL) * 2;
  hipFuncSetAttribute((const void*)pair_kernel<NS, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) pair_kernel<NS, SAVE><<<grid, 512, lds>>>(d_in, d_wp, d_out, d_save, d_clk);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) pair_kernel<NS, SAVE><<<grid, 512, lds>>>(d_in, d_wp, d_out, d_save, d_clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  const hipError_t err_launch = hipGetLastError();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flops = 2.0 * 256 * 256 * L * (double)P;
  std::vector<float> out(NB * 1024);
  double err = 0, sc = 0;
  for (int64_t tile : {(int64_t)3, (int64_t)(P / 32 - 2)}) {
    hipMemcpy(out.data(), d_out + tile * NB * 1024, out.size() * sizeof(float), hipMemcpyDeviceToHost);
    for (int pl = 0; pl < 32; pl += 5) {
      std::vector<double> h(256), nh(256);
      for (int f = 0; f < 256; ++f) h[f] = X[tp_index(tile * 32 + pl, f, NB)];
      for (int l = 0; l < L; ++l) {
        for (int o = 0; o < 256; ++o) {
          double s = 0;
          for (int k = 0; k < 256; ++k) s += (double)W[((size_t)l * 256 + o) * 256 + k] * act_ref(h[k]);
          nh[o] = s;
        }
        h = nh;
      }
      for (int f = 0; f < 256; ++f) {
        err = fmax(err, fabs(out[tp_index(pl, f, NB)] - h[f]));
        sc = fmax(sc, fabs(h[f]));
      }
    }
  }
  std::vector<unsigned long long> c(2048);
  hipMemcpy(c.data(), d_clk, c.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < 1024; ++i) { cyc += (double)c[2 * i]; wall += (double)c[2 * i + 1]; }
  const double mfma_ms = (NS == 2 ? 3.0 : 6.0) * flops / 2.5e15 * 1e3;
  printf("[%4.0f MHz, %7.0f cyc/WG] %-28s %7.3f ms  %7.1f fp32-equivalent TFLOP/s  (bf16 pipe %4.1f %%)  max|err| %.2e (scale %.2e, rel %.1e) %s\n",
         cyc / wall * 100.0, cyc / 1024, name, ms, flops / ms / 1e9, 100.0 * mfma_ms / ms, err, sc, err / sc, hipGetErrorString(err_launch));
  hipFree(d_wp);
}

int main() {
  const int64_t P = 524288;
  std::vector<float> W((size_t)L * 256 * 256), X((size_t)P * 256);
  srand(1);
  for (auto& w : W) w = ((rand() % 2001) / 1000.0f - 1.0f) * (L > 8 ? 0.07f : 0.09f);
  for (auto& x : X) x = (rand() % 2001) / 1000.0f - 1.0f;
  float *d_in, *d_out, *d_save;
  unsigned long long* d_clk;
  hipMalloc(&d_in, X.size() * 4);
  hipMalloc(&d_out, X.size() * 4);
  hipMalloc(&d_save, X.size() * 4 * L);
  hipMalloc(&d_clk, 2048 * 8);
  hipMemcpy(d_in, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  run<2, false>("pair 3-term softplus", P, d_in, d_out, d_save, d_clk, W, X);
  if (PAIR_NBUF == 2) run<3, false>("pair 6-term softplus", P, d_in, d_out, d_save, d_clk, W, X);
#ifdef PAIR_ZSAVE
  run<2, true>("pair 3-term softplus +zsave", P, d_in, d_out, d_save, d_clk, W, X);
  run<3, true>("pair 6-term softplus +zsave", P, d_in, d_out, d_save, d_clk, W, X);
#endif
  return 0;
}
