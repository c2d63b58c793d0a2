// Dev probe: fused 8 x (256 x 256) stack with split 16-bit products on v_mfma_f32_32x32x16_{bf16,f16} (fp32 accumulate).
//   NS = 2: x = hi + lo,        3 MFMAs per product  (bf16: error ~2^-16 per product)
//   NS = 3: x = hi + mid + lo,  6 MFMAs per product  (fp32-class error)
//   -DPROBE_F16:   the parts are fp16 instead of bf16 (NS = 2 then carries 22 mantissa bits: fp32-class with 3 MFMAs - the forward
//                  mode of the product kernels, mlp_core.h NS = 4)
//   -DPROBE_PKRTZ: (with PROBE_F16) hi parts by v_cvt_pkrtz_f16_f32 on element pairs (round toward zero; the lo part absorbs the
//                  truncation) - a cheaper producer, to be measured for accuracy and speed
//   -DPROBE_RELU / -DPROBE_LOOPED / -DABL=n: see main() and the kernel
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "common.h"
void sdfhip_set_error(const char*, ...) {}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
#ifndef PROBE_L
#define PROBE_L 8
#endif
constexpr int NB = 8, L = PROBE_L;
#ifndef ABL
#define ABL 0
#endif

template <int NS>
struct Split {
  bf16x8 p[NS][2];  // [part][k half]
};
template <int NS>
__device__ __forceinline__ Split<NS> split_block(const f32x16& v) {
  Split<NS> s;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float r = v[kk * 8 + j];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
#ifdef PROBE_F16
        r = __builtin_amdgcn_fmed3f(r, -65504.0f, 65504.0f);
        const _Float16 h = (_Float16)r;
        s.p[q][kk][j] = __builtin_bit_cast(__bf16, h);
#else
        const __bf16 h = (__bf16)r;
        s.p[q][kk][j] = h;
#endif
        r -= (float)h;
      }
    }
#if defined(PROBE_F16) && defined(PROBE_PKRTZ)
  // hi parts of element pairs in one instruction (round toward zero), lo = fp16(x - hi)
  typedef __fp16 half2_t __attribute__((ext_vector_type(2)));
  if constexpr (NS == 2)  // hi + lo only (the three-part form keeps the rounding conversions above)
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const float r0 = __builtin_amdgcn_fmed3f(v[kk * 8 + j], -65504.0f, 65504.0f), r1 = __builtin_amdgcn_fmed3f(v[kk * 8 + j + 1], -65504.0f, 65504.0f);
      const half2_t h = __builtin_amdgcn_cvt_pkrtz(r0, r1);
      s.p[0][kk][j] = __builtin_bit_cast(__bf16, (_Float16)h[0]);
      s.p[0][kk][j + 1] = __builtin_bit_cast(__bf16, (_Float16)h[1]);
      s.p[1][kk][j] = __builtin_bit_cast(__bf16, (_Float16)(r0 - (float)h[0]));
      s.p[1][kk][j + 1] = __builtin_bit_cast(__bf16, (_Float16)(r1 - (float)h[1]));
    }
#endif
  return s;
}

template <int NS, int ACT, int NBUF, bool LOOPED>
__global__ __launch_bounds__(256, 1) void split_kernel(const float* __restrict__ in_tp, const __bf16* __restrict__ wp,
                                                       float* __restrict__ out_tp, unsigned long long* __restrict__ clk) {
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsb[];
  constexpr int CH = NS * NB * 2 * 64 * 8;  // bf16 per chunk (one k block): [part][ob][kk][lane][8]
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  int cur = 0;
  constexpr int PER = CH * 2 / 1024 / 4;  // DMA instructions per wave per chunk
  auto issue = [&](const __bf16* g, const int buf) {
    // chunk bytes = CH * 2; 1 KiB per wave instruction
    constexpr int pieces = CH * 2 / 1024;
    for (int i = wave; i < pieces; i += 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * 512 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(ldsb + buf * CH + i * 512), 16, 0, 0);
  };
  for (int c = 0; c < NBUF - 1; ++c) issue(wp + (size_t)c * CH, c);
  f32x16 accA[NB], accB[NB];
  {
    const float* p = in_tp + (size_t)tile * NB * 1024 + lane;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[b][r] = p[(b * 16 + r) * 64];
  }
  int chunk = 0;
  auto layer = [&](f32x16 (&in)[NB], f32x16 (&out)[NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[b][r] = 0.0f;
    auto make = [&](const int kb_dummy, const f32x16& z) __attribute__((always_inline)) {
      f32x16 h;
      if constexpr ((ABL & 1) != 0) {
        Split<NS> s0;
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) s0.p[q][kk][j] = (__bf16)z[0];
        return s0;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) h[r] = ACT == 0 ? fmaxf(z[r], 0.0f) : softplus100_h(z[r] * 0.01f) * 100.0f;
      return split_block<NS>(h);
    };
    Split<NS> blk = make(0, in[0]);
    static_for<0, NB>([&](auto kbc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value;
      if constexpr ((ABL & 8) == 0) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PER) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if constexpr ((ABL & 2) == 0) {
        const int nc = chunk + NBUF - 1;  // past the end: re-read the last chunk (keeps vmcnt accounting uniform)
        issue(wp + (size_t)(nc < L * NB ? nc : L * NB - 1) * CH, (cur + NBUF - 1) % NBUF);
        ++chunk;
      }
      Split<NS> nxt;
      if constexpr (kb + 1 < NB) nxt = make(kb + 1, in[kb + 1 < NB ? kb + 1 : 0]);
      const __bf16* base = ldsb + cur * CH + lane * 8;
      constexpr int NT = NS == 2 ? 3 : 6;
      constexpr int ta[6] = {NS == 2 ? 1 : 1, NS == 2 ? 0 : 2, NS == 2 ? 0 : 0, 1, 0, 0};
      constexpr int tb[6] = {NS == 2 ? 0 : 1, NS == 2 ? 1 : 0, NS == 2 ? 0 : 2, 0, 1, 0};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a[NS][NB];
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
          for (int ob = 0; ob < NB; ++ob) {
            if constexpr ((ABL & 4) != 0) a[q][ob] = blk.p[q][kk];
            else a[q][ob] = *reinterpret_cast<const bf16x8*>(base + ((q * NB + ob) * 2 + kk) * 512);
          }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int ob = 0; ob < NB; ++ob)
#ifdef PROBE_F16
            out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[ta[t]][ob]), __builtin_bit_cast(f16x8_t, blk.p[tb[t]][kk]),
                                                             out[ob], 0, 0, 0);
#else
            out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta[t]][ob], blk.p[tb[t]][kk], out[ob], 0, 0, 0);
#endif
      }
      cur = (cur + 1) % NBUF;
      if constexpr (kb + 1 < NB) blk = nxt;
    });
  };
  if constexpr (LOOPED) {
    static_assert(NBUF == 2 && L % 2 == 0, "looped variant: two chunk buffers, layer pairs");
#pragma unroll 1
    for (int lp = 0; lp < L / 2; ++lp) {
      layer(accA, accB);
      layer(accB, accA);
    }
  } else {
    static_for<0, L>([&](auto lc) __attribute__((always_inline)) {
      constexpr int l = decltype(lc)::value;
      if constexpr ((l % 2) == 0) layer(accA, accB);
      else layer(accB, accA);
    });
  }
  auto& fin = (L % 2) == 0 ? accA : accB;
  {
    float* p = out_tp + (size_t)tile * NB * 1024 + lane;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) p[(b * 16 + r) * 64] = fin[b][r];
  }
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
    clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
  }
}

static float act_ref(double z, int act) { return act == 0 ? (z > 0 ? z : 0) : (z > 20 ? z : log1p(exp(z))); }
static __bf16 to_bf16(float x) { return (__bf16)x; }

template <int NS, int ACT, int NBUF, bool LOOPED = false>
static void run(const char* name, int64_t P, const float* d_in, float* d_out, unsigned long long* d_clk, const std::vector<float>& W,
                const std::vector<float>& X) {
  constexpr int CH = NS * NB * 2 * 64 * 8;
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
#ifdef PROBE_F16
                const _Float16 h16 = (_Float16)r;
                const __bf16 h = __builtin_bit_cast(__bf16, h16);
                Wp[(size_t)(l * NB + kb) * CH + (((q * NB + ob) * 2 + kk) * 64 + lane) * 8 + j] = h;
                r -= (float)h16;
                continue;
#else
                const __bf16 h = to_bf16(r);
                Wp[(size_t)(l * NB + kb) * CH + (((q * NB + ob) * 2 + kk) * 64 + lane) * 8 + j] = h;
#endif
                r -= (float)h;
              }
            }
  __bf16* d_wp;
  hipMalloc(&d_wp, Wp.size() * 2);
  hipMemcpy(d_wp, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice);
  const unsigned grid = (unsigned)(P / 128);
  const size_t lds = (size_t)NBUF * CH * 2;
  hipFuncSetAttribute((const void*)split_kernel<NS, ACT, NBUF, LOOPED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) split_kernel<NS, ACT, NBUF, LOOPED><<<grid, 256, lds>>>(d_in, d_wp, d_out, d_clk);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) split_kernel<NS, ACT, NBUF, LOOPED><<<grid, 256, lds>>>(d_in, d_wp, d_out, d_clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flops = 2.0 * 256 * 256 * L * (double)P;
  std::vector<float> out(NB * 1024);
  const int64_t tile = 3;
  hipMemcpy(out.data(), d_out + tile * NB * 1024, out.size() * sizeof(float), hipMemcpyDeviceToHost);
  double err = 0, sc = 0;
  for (int pl = 0; pl < 32; pl += 5) {
    std::vector<double> h(256), nh(256);
    for (int f = 0; f < 256; ++f) h[f] = X[tp_index(tile * 32 + pl, f, NB)];
    for (int l = 0; l < L; ++l) {
      for (int o = 0; o < 256; ++o) {
        double s = 0;
        for (int k = 0; k < 256; ++k) s += (double)W[((size_t)l * 256 + o) * 256 + k] * act_ref(h[k], ACT);
        nh[o] = s;
      }
      h = nh;
    }
    for (int f = 0; f < 256; ++f) {
      err = fmax(err, fabs(out[tp_index(pl, f, NB)] - h[f]));
      sc = fmax(sc, fabs(h[f]));
    }
  }
  std::vector<unsigned long long> c(2048);
  hipMemcpy(c.data(), d_clk, c.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < 1024; ++i) { cyc += (double)c[2 * i]; wall += (double)c[2 * i + 1]; }
  printf("[%4.0f MHz, %7.0f cyc/WG] %-22s %7.3f ms  %7.1f fp32-equivalent TFLOP/s   max|err| %.2e (scale %.2e, rel %.1e)\n", cyc / wall * 100.0,
         cyc / 1024, name, ms, flops / ms / 1e9, err, sc, err / sc);
  hipFree(d_wp);
}

int main() {
  const int64_t P = 524288;
  std::vector<float> W((size_t)L * 256 * 256), X((size_t)P * 256);
  srand(1);
  for (auto& w : W) w = ((rand() % 2001) / 1000.0f - 1.0f) * (L > 8 ? 0.07f : 0.09f);
  for (auto& x : X) x = (rand() % 2001) / 1000.0f - 1.0f;
  float *d_in, *d_out;
  unsigned long long* d_clk;
  hipMalloc(&d_in, X.size() * 4);
  hipMalloc(&d_out, X.size() * 4);
  hipMalloc(&d_clk, 2048 * 8);
  hipMemcpy(d_in, X.data(), X.size() * 4, hipMemcpyHostToDevice);
#ifdef PROBE_RELU
  run<2, 0, 2>("3-term relu (no transcendentals)", P, d_in, d_out, d_clk, W, X);
  run<3, 0, 2>("6-term relu (no transcendentals)", P, d_in, d_out, d_clk, W, X);
#elif defined(PROBE_LOOPED)
  run<2, 1, 2, true>("3-term softplus, layers looped", P, d_in, d_out, d_clk, W, X);
  run<3, 1, 2, true>("6-term softplus, layers looped", P, d_in, d_out, d_clk, W, X);
#else
  run<2, 1, 2>("3-term softplus", P, d_in, d_out, d_clk, W, X);
  run<3, 1, 2>("6-term softplus", P, d_in, d_out, d_clk, W, X);
#endif
  return 0;
}
