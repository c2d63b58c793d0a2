// Dev probe: device-side check of the producer arithmetic of mlp_core.h / common.h in isolation: split_put (hi / lo 16-bit parts of
// an activation block, packed pair conversions) must reconstruct every element to 2^-21 (fp16 parts, NS = 4) / 2^-16 (bf16, NS = 2) and
// land in the operand slots the MFMA expects; softplus100_h / _d1 against libm on a sweep of pre-activations.
#include <cmath>
#include <cstdio>
#include <vector>
#include "mlp_core.h"
void sdfhip_set_error(const char*, ...) {}

template <int NS>
__global__ void split_kernel(const float* in, float* rec) {
  SplitBlk<NS> s;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i];
  static_for<0, 16>([&](auto ec) __attribute__((always_inline)) { split_put<NS, decltype(ec)::value>(s, v[decltype(ec)::value]); });
  for (int e = 0; e < 16; ++e) {
    float r = 0.f;
    for (int q = ns_parts(NS) - 1; q >= 0; --q) {
      const __bf16 h = s.p[q][e >> 3][e & 7];
      if constexpr (NS == 4) r += (float)__builtin_bit_cast(_Float16, h);
      else r += (float)h;
    }
    rec[threadIdx.x * 16 + e] = r;
  }
}
__global__ void sp_kernel(const float* z, float* h, float* d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  h[i] = softplus100_h(z[i]);
  d[i] = softplus100_d1(z[i]);
}
int main() {
  const int n = 64 * 16;
  std::vector<float> in(n), rec(n);
  for (int i = 0; i < n; ++i) in[i] = (float)std::sin(i * 0.7) * std::pow(10.0f, (i % 9) - 4);
  in[5] = 1e6f; in[6] = -7e4f; in[7] = 0.f;
  float *din, *drec;
  hipMalloc(&din, n * 4); hipMalloc(&drec, n * 4);
  hipMemcpy(din, in.data(), n * 4, hipMemcpyHostToDevice);
  for (int ns : {4, 2}) {
    if (ns == 4) split_kernel<4><<<1, 64>>>(din, drec); else split_kernel<2><<<1, 64>>>(din, drec);
    hipMemcpy(rec.data(), drec, n * 4, hipMemcpyDeviceToHost);
    double worst = 0; int at = -1;
    for (int i = 0; i < n; ++i) {
      const float want = ns == 4 ? std::fmin(std::fmax(in[i], -65504.f), 65504.f) : in[i];
      const double rel = std::fabs((double)rec[i] - want) / (std::fabs(want) + 1e-30);
      if (std::fabs(want) > 1e-4 && rel > worst) { worst = rel; at = i; }
    }
    printf("split_put NS=%d: worst relative reconstruction error %.3e at %d (in %g rec %g)\n", ns, worst, at, in[at], rec[at]);
  }
  const int m = 4096;
  std::vector<float> z(m), h(m), d(m);
  for (int i = 0; i < m; ++i) z[i] = -0.5f + 2.0f * i / m;
  z[0] = -50.f; z[1] = 50.f; z[2] = 0.2f; z[3] = 0.19999f; z[4] = 0.20001f; z[5] = 1e30f; z[6] = -1e30f;
  float *dz, *dh, *dd;
  hipMalloc(&dz, m * 4); hipMalloc(&dh, m * 4); hipMalloc(&dd, m * 4);
  hipMemcpy(dz, z.data(), m * 4, hipMemcpyHostToDevice);
  sp_kernel<<<m / 256, 256>>>(dz, dh, dd);
  hipMemcpy(h.data(), dh, m * 4, hipMemcpyDeviceToHost);
  hipMemcpy(d.data(), dd, m * 4, hipMemcpyDeviceToHost);
  double eh = 0, ed = 0; int ah = -1, ad = -1;
  for (int i = 0; i < m; ++i) {
    const double t = 100.0 * z[i];
    const double hr = t > 20 ? z[i] : std::log1p(std::exp(t)) / 100.0, dr = t > 20 ? 1.0 : 1.0 / (1.0 + std::exp(-t));
    if (!(std::fabs(h[i] - hr) <= eh)) { if (std::fabs(h[i] - hr) > eh || h[i] != h[i]) { eh = h[i] != h[i] ? 1e30 : std::fabs(h[i] - hr); ah = i; } }
    if (!(std::fabs(d[i] - dr) <= ed)) { if (std::fabs(d[i] - dr) > ed || d[i] != d[i]) { ed = d[i] != d[i] ? 1e30 : std::fabs(d[i] - dr); ad = i; } }
  }
  printf("softplus100_h: worst abs error %.3e at z = %g (got %g) ; softplus100_d1: %.3e at z = %g (got %g)\n", eh, z[ah], h[ah], ed, z[ad], d[ad]);
  return 0;
}
