// Dev probe: throughput + correctness of the MFMA core (mlp_core.h) on a stack of L 256x256 layers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sdfstudio_amd/csrc tools/probe_mlp.hip -o /tmp/probe_mlp && /tmp/probe_mlp
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "mlp_core.h"
void sdfhip_set_error(const char*, ...) {}

constexpr int NB = 8, L = 8;

template <int SAVE, int ACT>
__global__ __launch_bounds__(256, 1) void probe_kernel(const float* __restrict__ in_tp, const float* __restrict__ wp,
                                                       float* __restrict__ z_tp, float* __restrict__ out_tp,
                                                       unsigned long long* __restrict__ clk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  WStream ws{lds, NB * 1024, 0, wave, lane};
  ws.issue(wp, NB, true);
  f32x16 accA[NB], accB[NB];
  Raw carry;
#pragma unroll
  for (int b = 0; b < NB; ++b) accA[b] = tp_load_blk(in_tp, tile, NB, b, lane);
  static_for<0, L>([&](auto lc) __attribute__((always_inline)) {
    constexpr int l = decltype(lc)::value;
    auto& in = pick<(l % 2) == 0>(accA, accB);
    auto& out = pick<(l % 2) == 0>(accB, accA);
#pragma unroll
    for (int b = 0; b < NB; ++b) out[b] = f32x16_zero();
    auto make = [&](auto kbc, const Raw&, auto ec) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value, e = decltype(ec)::value;
      const float z = in[kb][e];
      if constexpr (SAVE) *tp_elem(z_tp + (size_t)l * gridDim.x * 4 * NB * 1024, tile, NB, kb, e, lane) = z;
      return ACT == 0 ? fmaxf(z, 0.0f) : softplus100_h(z * 0.01f) * 100.0f;
    };
    const float* w = wp + (size_t)l * NB * NB * 1024;
    tp_gemm<NB, NB, Stores<(SAVE ? 16 : 0)>>(out, carry, NoFetch{}, make, NoFetch{}, ws, w, l + 1 < L ? w + NB * NB * 1024 : nullptr, NB);
  });
  auto& fin = pick<(L % 2) == 0>(accA, accB);
#pragma unroll
  for (int b = 0; b < NB; ++b) tp_store_blk(fin[b], out_tp, tile, NB, b, lane);
  if (threadIdx.x == 0 && blockIdx.x < 1024) {  // shader cycles and constant-rate (100 MHz) ticks spent by this workgroup
    clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0;
    clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
  }
}

static float act_ref(float z, int act) {
  if (act == 0) return z > 0 ? z : 0;
  float t = z;  // softplus(beta=1) scaled form used in the probe
  return t > 20.f ? z : log1pf(expf(t));
}

static unsigned long long* d_clk;
template <int SAVE, int ACT>
static void run(int sustain, const char* name, int64_t P, const float* d_in, const float* d_wp, float* d_z, float* d_out, const std::vector<float>& W,
                const std::vector<float>& X) {
  const unsigned grid = (unsigned)(P / 128);
  const size_t lds = 2 * NB * 1024 * sizeof(float);
  hipFuncSetAttribute((const void*)probe_kernel<SAVE, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) probe_kernel<SAVE, ACT><<<grid, 256, lds>>>(d_in, d_wp, d_z, d_out, d_clk);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) probe_kernel<SAVE, ACT><<<grid, 256, lds>>>(d_in, d_wp, d_z, d_out, d_clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  if (sustain > 0) {
    // sustained load: TFLOP/s per batch of 20 launches (does the chip hold its clock?)
    printf("  sustained:");
    for (int b = 0; b < sustain; ++b) {
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) probe_kernel<SAVE, ACT><<<grid, 256, lds>>>(d_in, d_wp, d_z, d_out, d_clk);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float m2;
      hipEventElapsedTime(&m2, e0, e1);
      printf(" %.0f", 2.0 * 256 * 256 * L * (double)P / (m2 / 20) / 1e9);
    }
    printf(" TFLOP/s\n");
  }
  const double flops = 2.0 * 256 * 256 * L * (double)P;
  // check point 5 of tile 3 against a CPU fp64 evaluation
  std::vector<float> out(NB * 1024);
  const int64_t tile = 3, pl = 5;
  hipMemcpy(out.data(), d_out + tile * NB * 1024, out.size() * sizeof(float), hipMemcpyDeviceToHost);
  std::vector<double> h(256), nh(256);
  for (int f = 0; f < 256; ++f) h[f] = X[tp_index(tile * 32 + pl, f, NB)];
  for (int l = 0; l < L; ++l) {
    for (int o = 0; o < 256; ++o) {
      double s = 0;
      for (int k = 0; k < 256; ++k) s += (double)W[((size_t)l * 256 + o) * 256 + k] * act_ref((float)h[k], ACT);
      nh[o] = s;
    }
    h = nh;
  }
  double err = 0, sc = 0;
  for (int f = 0; f < 256; ++f) {
    const double g = out[tp_index(pl, f, NB)];
    err = fmax(err, fabs(g - h[f]));
    sc = fmax(sc, fabs(h[f]));
  }
  {
    std::vector<unsigned long long> c(2048);
    hipMemcpy(c.data(), d_clk, c.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < 1024; ++i) { cyc += (double)c[2 * i]; wall += (double)c[2 * i + 1]; }
    printf("[shader clock %.0f MHz, %.0f cycles/WG] ", cyc / wall * 100.0, cyc / 1024);
  }
  printf("%-28s %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)   max|err| %.2e (scale %.2e)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100, err,
         sc);
}

int main() {
  const int64_t P = 524288;
  std::vector<float> W((size_t)L * 256 * 256), X((size_t)P * 256), Wp(W.size());
  srand(1);
  for (auto& w : W) w = ((rand() % 2001) / 1000.0f - 1.0f) * 0.09f;
  for (auto& x : X) x = (rand() % 2001) / 1000.0f - 1.0f;
  // pack: Wp[l][kb][ob][r4][lane][j]
  for (int l = 0; l < L; ++l)
    for (int kb = 0; kb < NB; ++kb)
      for (int ob = 0; ob < NB; ++ob)
        for (int r4 = 0; r4 < 4; ++r4)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
              const int o = ob * 32 + (lane & 31), k = kb * 32 + tp_row(r4 * 4 + j, lane >> 5);
              Wp[((((size_t)(l * NB + kb) * NB + ob) * 4 + r4) * 64 + lane) * 4 + j] = W[((size_t)l * 256 + o) * 256 + k];
            }
  float *d_in, *d_wp, *d_z, *d_out;
  hipMalloc(&d_in, X.size() * 4);
  hipMalloc(&d_out, X.size() * 4);
  hipMalloc(&d_z, X.size() * 4 * L);
  hipMalloc(&d_wp, Wp.size() * 4);
  hipMalloc(&d_clk, 2048 * 8);
  hipMemcpy(d_in, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_wp, Wp.data(), Wp.size() * 4, hipMemcpyHostToDevice);
  run<0, 0>(0, "relu", P, d_in, d_wp, d_z, d_out, W, X);
  run<0, 1>(0, "softplus", P, d_in, d_wp, d_z, d_out, W, X);
  run<1, 1>(30, "softplus + store z", P, d_in, d_wp, d_z, d_out, W, X);
  return 0;
}
