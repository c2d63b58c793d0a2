// sdfhip — shared device/host helpers (gfx950 only).
//
// "Tile-packed" (TP) activation layout
// ------------------------------------
// Every per-point feature matrix that an MFMA kernel produces or consumes lives in HBM in the
// accumulator layout of v_mfma_f32_32x32x2_f32 so that a wave can load / store it with perfectly
// coalesced 256-byte rows and feed it to the next MFMA with no LDS round trip and no transposition:
//
//     A[tile][blk][reg][lane]      tile = point / 32, blk = feature / 32, reg in [0,16), lane in [0,64)
//     point   = 32*tile + (lane & 31)
//     feature = 32*blk  + (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
//
// i.e. lane l of a wave owns point (l & 31); the two half-waves own complementary halves of that point's
// features.  A "layer" D[out][point] = sum_k W[out][k] * H[point][k] then runs with W as the MFMA A operand
// (pre-packed in the matching k order, see pack_kernel) and the wave's own registers as the B operand.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SDFHIP_HD __host__ __device__ __forceinline__
#define SDFHIP_D __device__ __forceinline__

SDFHIP_HD int tp_row(int reg, int hf) { return (reg & 3) + 8 * (reg >> 2) + 4 * hf; }
SDFHIP_HD int tp_reg_of_row(int row) { return (row & 3) | ((row >> 3) << 2); }
SDFHIP_HD int tp_hf_of_row(int row) { return (row >> 2) & 1; }
// flat index of (point p, feature f) in a TP array with nb feature blocks
SDFHIP_HD size_t tp_index(int64_t p, int f, int nb) {
  const int64_t tile = p >> 5;
  const int row = f & 31;
  return (size_t)(((tile * nb + (f >> 5)) * 16 + tp_reg_of_row(row)) * 64 + (p & 31) + 32 * tp_hf_of_row(row));
}

template <int I, int N, class F>
SDFHIP_D void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// A value that is only CONSUMED at the end of a long unrolled region (the lane-local sdf-row dot product of the output layer: 128 - 256
// fused multiply-adds riding in the producers of as many MFMA steps) is fair game for LLVM's code sinking: the whole chain is moved
// behind the region and its operands stay live across it - wide_out_kernel kept 2 x 256 of them in scratch (884 B per lane).  An empty
// volatile asm that reads and writes the accumulator pins each step of the chain where it is written.
SDFHIP_D void pin_here(float& x) { asm volatile("" : "+v"(x)); }

// ---- softplus(beta=100, threshold=20) and its derivatives (aten softplus / softplus_backward / softplus_double_backward,
// used at sdf_field.py:365,409) on the hardware transcendental unit: v_exp_f32 / v_log_f32 (base 2, 1 ulp) and v_rcp_f32.
//   t = 100 z ; e = exp(t) ; h = log(1 + e) / 100 ; s'(z) = e / (1 + e) ; s''(z) = 100 s' (1 - s')
// log(1 + e) is formed as log2(1 + e) ln2: for tiny e the rounding of 1 + e bounds the ABSOLUTE error of h by 6e-10, far
// below the fp32 resolution of the next layer's accumulation.  A libm expf + log1pf + IEEE division here costs ~150 VALU
// instructions per element and made the fused kernels instruction-fetch bound (profiles/r1_notes.md).
// The exponent's argument is NOT clamped: for t > 20 the linear branch is selected and whatever the other branch holds (inf, or
// inf * 0 = NaN in the derivative) is discarded by the select; z * (100 log2 e) is one multiply and the threshold test is on z
// itself (t > 20 <=> z > 0.2): two VALU instructions fewer per element than "t = 100 z; min(t, 20) * log2 e" in the producers of
// the fused kernels, which are VALU-bound in the forward launch.
constexpr float kSp100Log2e = 144.269504088896340736f;  // 100 / ln 2
constexpr float kSp100Thr = 0.2f;                       // threshold 20 / beta 100
SDFHIP_D void softplus100(float z, float& h, float& d1) {
  const float e = __builtin_amdgcn_exp2f(z * kSp100Log2e);
  const float u = 1.0f + e;
  const float hs = __builtin_amdgcn_logf(u) * (0.69314718055994530942f * 0.01f);
  const float ds = e * __builtin_amdgcn_rcpf(u);
  const bool lin = z > kSp100Thr;
  h = lin ? z : hs;
  d1 = lin ? 1.0f : ds;
}
SDFHIP_D float softplus100_d1(float z) {
  const float e = __builtin_amdgcn_exp2f(z * kSp100Log2e);
  const float ds = e * __builtin_amdgcn_rcpf(1.0f + e);
  return z > kSp100Thr ? 1.0f : ds;
}
// h alone, in the overflow-free form  softplus(z) = max(z, 0) + log(1 + e^{-|100 z|}) / 100  (round 5): the exponential's argument is never
// positive, so no branch has to be discarded - multiply (the |.| and the sign ride in the instruction's source modifiers), exp, add, log, one
// clamp, one fma: six vector instructions where the select form took seven, and for 0 < 100 z < 20 the sum z + (small) is closer to aten's
// log1p(exp(.)) than log2(1 + e^{100 z}) ln2 / 100 was.  Above the threshold (100 z > 20) the correction is < 2.1e-11, below half an ulp of
// z >= 0.2: the result is z exactly, like aten's linear branch.  HI bounds the result from above in the same instruction as the max with 0
// (v_med3_f32): producers that hand the value to fp16 operand parts pass 65504 (mlp_core.h InRange) and need no clamp of their own.
#ifndef SDFHIP_OLD_SOFTPLUS
template <bool CLAMP16 = false>
SDFHIP_D float softplus100_h(float z) {
  const float e = __builtin_amdgcn_exp2f(__builtin_fabsf(z) * -kSp100Log2e);
  const float l = __builtin_amdgcn_logf(1.0f + e);
  return __builtin_fmaf(l, 0.69314718055994530942f * 0.01f, __builtin_amdgcn_fmed3f(z, 0.0f, CLAMP16 ? 65504.0f : __builtin_inff()));
}
#else  // the select form of rounds 1 - 4 (A/B builds)
template <bool CLAMP16 = false>
SDFHIP_D float softplus100_h(float z) {
  const float e = __builtin_amdgcn_exp2f(z * kSp100Log2e);
  const float hs = __builtin_amdgcn_logf(1.0f + e) * (0.69314718055994530942f * 0.01f);
  const float h = z > kSp100Thr ? z : hs;
  return CLAMP16 ? __builtin_amdgcn_fmed3f(h, -65504.0f, 65504.0f) : h;
}
#endif

// s'(z) recovered from the SAVED ACTIVATION h = softplus(z) (round 3: the fused kernels save h_l, not z_l): e^{100 h} = 1 + e^{100 z}, so
//   s'(z) = e^{100 z} / (1 + e^{100 z}) = 1 - e^{-100 h}
// one transcendental (quarter rate on the VALU) and two full-rate instructions in place of exp + rcp + five; above the threshold h = z
// and e^{-100 h} < 2e-9 rounds the result to 1 like aten's linear branch.  For h -> 0 the subtraction is exact to half an ulp of 1
// (6e-8 ABSOLUTE, where exp / (1 + exp) is 6e-8 relative): the same error class as the rounding of every s' near 1, which is what the
// sums these derivatives enter are made of; parity is held by the fp64-anchored gradient tests of tests/test_gpu_parity.py.
SDFHIP_D float softplus100_d1_from_h(float h) { return 1.0f - __builtin_amdgcn_exp2f(-h * kSp100Log2e); }

// Hidden activation of a fused geometry-type network, selected at compile time by the network's dims class (GeoDims::ACT):
//   0  Softplus(beta = 100)   the SDF field (sdf_field.py:290, 409)
//   1  ReLU                   the background fields (field_components/mlp.py:93 of NeRFField; tcnn's FullyFusedMLP in TCNNNerfactoField)
// act_d1 is the derivative the first-order backward multiplies with; second-order passes exist for ACT = 0 only.
// CLAMP16: the result is handed to fp16 operand parts (precision mode 4) - bound it by fp16's largest finite value in the activation's own
// max / min instruction (v_med3_f32), so that the split needs no clamp of its own (mlp_core.h: InRange)
template <int ACT, bool CLAMP16 = false>
SDFHIP_D float act_h(const float z) {
  if constexpr (ACT == 1) return __builtin_amdgcn_fmed3f(z, 0.0f, CLAMP16 ? 65504.0f : __builtin_inff());
  else return softplus100_h<CLAMP16>(z);
}
template <int ACT>
SDFHIP_D float act_d1(const float z) {
  if constexpr (ACT == 1) return z > 0.0f ? 1.0f : 0.0f;
  else return softplus100_d1(z);
}
// the same derivative from the saved activation h = act(z) (what the training kernels keep per layer; the weight gradient reads it as is)
template <int ACT>
SDFHIP_D float act_d1h(const float h) {
  if constexpr (ACT == 1) return h > 0.0f ? 1.0f : 0.0f;
  else return softplus100_d1_from_h(h);
}

// thread-local last-error string (extern "C" API returns 0 or a negative code)
void sdfhip_set_error(const char* fmt, ...);
#define SDFHIP_CHECK_HIP(expr)                                                            \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      sdfhip_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)
#define SDFHIP_REQUIRE(cond, ...)       \
  do {                                  \
    if (!(cond)) {                      \
      sdfhip_set_error(__VA_ARGS__);    \
      return -1;                        \
    }                                   \
  } while (0)
