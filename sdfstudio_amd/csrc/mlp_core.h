// sdfhip — MFMA core shared by the fused geometry / colour network kernels (gfx950).
//
// One workgroup = 4 waves = 128 points; one wave per SIMD, each wave owns a 32-point tile whose
// activations stay in VGPRs in tile-packed (TP) order across all layers.  Per layer the effective
// (weight-norm folded) weight matrix streams L2 -> LDS in 16-k-step chunks shared by the 4 waves
// (double buffered, one barrier per chunk) and is consumed as the MFMA A operand:
//
//   acc[ob] (+)= v_mfma_f32_32x32x2_f32( A = Wp[kb][ob][r][lane],  B = H[kb][r] (own register) )
//
// fp32 in / fp32 accumulate: exact-f32 numerics (the reference trains in fp32, parity target 1e-5 on SDF).
#pragma once
#include "common.h"

// floats in one weight chunk (all NBO output blocks of one 32-wide k block)
template <int NBO>
struct Chunk {
  static constexpr int kFloats = NBO * 1024;
};

// acc[0..NBO) += W (packed at wp) * H[0..KB)    — all 4 waves of the workgroup must call this together.
// The k-block loop is a real loop: after block H[0] is consumed the register tile is rotated by one block
// (KB v_mov groups, hidden under the 16*NBO MFMAs) so that every register index stays a compile-time constant;
// after KB iterations H is back in its original order.
template <int KB, int NBO, int MAXA, int MAXH>
SDFHIP_D void tp_gemm(f32x16 (&acc)[MAXA], f32x16 (&H)[MAXH], const float* __restrict__ wp, float* lds,
                      const int tid, const int lane) {
  static_assert(KB <= MAXH && NBO <= MAXA, "register tile too small");
  constexpr int CH = NBO * 1024;
  f32x4 st[NBO];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(wp);
#pragma unroll
    for (int i = 0; i < NBO; ++i) st[i] = src[i * 256 + tid];
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
#pragma unroll
    for (int i = 0; i < NBO; ++i) dst[i * 256 + tid] = st[i];
  }
  __syncthreads();
#pragma unroll 1
  for (int kb = 0; kb < KB; ++kb) {
    const float* cur = lds + (kb & 1) * CH + lane;
    const bool more = kb + 1 < KB;
    if (more) {
      const f32x4* src = reinterpret_cast<const f32x4*>(wp + (size_t)(kb + 1) * CH);
#pragma unroll
      for (int i = 0; i < NBO; ++i) st[i] = src[i * 256 + tid];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float b = H[0][r];
#pragma unroll
      for (int ob = 0; ob < NBO; ++ob) {
        const float a = cur[(ob * 16 + r) * 64];
        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ob], 0, 0, 0);
      }
    }
    if (more) {
      f32x4* dst = reinterpret_cast<f32x4*>(lds + ((kb + 1) & 1) * CH);
#pragma unroll
      for (int i = 0; i < NBO; ++i) dst[i * 256 + tid] = st[i];
    }
    __syncthreads();
    const f32x16 h0 = H[0];
#pragma unroll
    for (int i = 0; i + 1 < KB; ++i) H[i] = H[i + 1];
    H[KB - 1] = h0;
  }
}

// TP load / store of NB blocks for one tile
template <int NB, int MAXH>
SDFHIP_D void tp_load(f32x16 (&H)[MAXH], const float* __restrict__ base, const int64_t tile, const int lane) {
  const float* p = base + (size_t)tile * NB * 1024 + lane;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) H[b][r] = p[(b * 16 + r) * 64];
}
template <int NB, int MAXH>
SDFHIP_D void tp_store(const f32x16 (&H)[MAXH], float* __restrict__ base, const int64_t tile, const int lane) {
  float* p = base + (size_t)tile * NB * 1024 + lane;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) p[(b * 16 + r) * 64] = H[b][r];
}
// per-feature vector (natural order, padded to NB*32) broadcast into TP registers
template <int NB, int MAXH>
SDFHIP_D void tp_load_rowvec(f32x16 (&H)[MAXH], const float* __restrict__ v, const int hf) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) H[b][r] = v[b * 32 + tp_row(r, hf)];
}
