// Superseded weight-gradient kernels, kept for reference only (NOT compiled into libsdfhip.so since round 4; they were reachable through
// SDFHIP_WGRAD_FP32=1 / SDFHIP_WGRAD_V1=1 in rounds 1 - 3): the exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32, round 1) and the 4-wave
// split-bf16 kernel the 8-wave double-buffered wgrad_bf16x8_kernel of csrc/wgrad_kernels.h replaced in round 2.  To build one again,
// include this header after wgrad_kernels.h in a scratch translation unit.
#pragma once

constexpr int kWgRow = 36;              // floats per LDS row (32 points + 4 pad: 16-byte aligned, bank-spreading)
constexpr int kWgBlk = 32 * kWgRow;     // floats per staged block
constexpr int kWgLdsBytes = 16 * kWgBlk * 4;

// grid = n_split, block = 256.  One macro tile (<= 8 x 8 blocks at ob_base / ib_base) per launch.  Wave (qi, qj) owns row
// blocks ob_base + qi + 2 i (i < NA) and column blocks ib_base + qj + 2 j (j < NB): the interleaved assignment keeps the four
// waves balanced for 6- or 3-block operands, and NA / NB are compile-time so the 16 NA NB MFMAs of each 8-point group form
// one straight-line block.  Slots beyond the operand's extent stay zero in LDS (multiplied but never written back).
template <int NA, int NB>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [16 slots][32][36]: slots 0..7 = A blocks, 8..15 = B blocks
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the staging predicates scalar
  const int qi = wave >> 1, qj = wave & 1;
  const int split = blockIdx.x;
  const int ob_base = a.ob_base, ib_base = a.ib_base;

  f32x16 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float colsum[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) colsum[i] = 0.0f;

  const int64_t t0 = (int64_t)split * a.tiles_per_split;
  int64_t t1 = t0 + a.tiles_per_split;
  if (t1 > a.n_tiles) t1 = a.n_tiles;
  const int n_t = t1 > t0 ? (int)(t1 - t0) : 0;
  const int n_stage = n_t * a.n_pairs;

  for (int i = tid; i < 16 * kWgBlk / 4; i += 256) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging role of this wave: slots {wave, wave + 4} (A) and {8 + wave, 12 + wave} (B)
  const int slot[4] = {wave, wave + 4, 8 + wave, 12 + wave};
  bool valid[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) valid[q] = slot[q] < 8 ? (ob_base + slot[q] < a.nba) : (ib_base + slot[q] - 8 < a.nbb);

  // per-slot source description, resolved once (wave-uniform -> scalar registers): base pointer, floats per tile
  const float* src0[4];
  const float* src1[4];
  int stride0[4], stride1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int blk = slot[q] < 8 ? ob_base + slot[q] : ib_base + slot[q] - 8;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const TpOperand& op = slot[q] < 8 ? a.A[pr] : a.B[pr];
      const int seg = blk >= op.nb[0];
      const int lb = blk - (seg ? op.nb[0] : 0);
      const float* base = op.ptr[seg] + (size_t)lb * 1024;
      const int stride = op.nb[seg] * 1024;
      if (pr == 0) {
        src0[q] = base;
        stride0[q] = stride;
      } else {
        src1[q] = base;
        stride1[q] = stride;
      }
    }
  }

  f32x4 pre[4][4];
  auto load_stage = [&](const int st) {
    const bool p1 = st >= n_t;
    const int64_t tile = t0 + (p1 ? st - n_t : st);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!valid[q]) continue;
      const float* base = p1 ? src1[q] : src0[q];
      const int stride = p1 ? stride1[q] : stride0[q];
      const f32x4* src = reinterpret_cast<const f32x4*>(base + (size_t)tile * stride) + lane;
#pragma unroll
      for (int i = 0; i < 4; ++i) pre[q][i] = src[i * 64];
    }
  };
  auto store_stage = [&](const int st) {
    const bool p1 = st >= n_t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!valid[q]) continue;
      // lane holds, for i = 0..3, TP row r = 4 i + (lane >> 4), half (lane >> 3) & 1, points 4 (lane & 7) .. + 3
      float* dst = lds + slot[q] * kWgBlk + (lane & 7) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = pre[q][i];
        const int f = tp_row(i * 4 + (lane >> 4), (lane >> 3) & 1);
        *reinterpret_cast<f32x4*>(dst + f * kWgRow) = v;
      }
    }
  };

  if (n_stage > 0) load_stage(0);
  const float* la = lds + qi * kWgBlk + (lane & 31) * kWgRow + 4 * (lane >> 5);
  const float* lb = lds + (8 + qj) * kWgBlk + (lane & 31) * kWgRow + 4 * (lane >> 5);
  for (int st = 0; st < n_stage; ++st) {
    __syncthreads();  // every wave is done reading the previous stage (and, first time round, the zero fill has landed)
    store_stage(st);
    __syncthreads();
    if (st + 1 < n_stage) load_stage(st + 1);
    const bool first_pair = st < n_t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a4[NA], b4[NB];
#pragma unroll
      for (int i = 0; i < NA; ++i) a4[i] = *reinterpret_cast<const f32x4*>(la + 2 * i * kWgBlk + g * 8);
#pragma unroll
      for (int j = 0; j < NB; ++j) b4[j] = *reinterpret_cast<const f32x4*>(lb + 2 * j * kWgBlk + g * 8);
      if (first_pair) {
#pragma unroll
        for (int i = 0; i < NA; ++i) colsum[i] += (a4[i][0] + a4[i][1]) + (a4[i][2] + a4[i][3]);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][s], b4[j][s], acc[i][j], 0, 0, 0);
    }
  }

  const int ldc = a.nbb * 32;
  const int hf = lane >> 5;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int ob = ob_base + qi + 2 * i, ib = ib_base + qj + 2 * j;
      if (qi + 2 * i < 8 && qj + 2 * j < 8 && ob < a.nba && ib < a.nbb) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          a.partial[wg_partial_index(split, (int)gridDim.x, ob * 32 + tp_row(r, hf), ib * 32 + (lane & 31), ldc)] = acc[i][j][r];
      }
    }
  if (a.bpartial != nullptr && ib_base == 0 && qj == 0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const float t = colsum[i] + __shfl_xor(colsum[i], 32);
      const int ob = ob_base + qi + 2 * i;
      if (hf == 0 && qi + 2 * i < 8 && ob < a.nba) a.bpartial[(size_t)split * a.nba * 32 + ob * 32 + lane] = t;
    }
  }
}


// ---- 4-wave split-bf16 kernel (round 1)
template <int NA, int NB>
__global__ __launch_bounds__(256, 1) void wgrad_bf16_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsb[];  // [16 slots][hi|lo][32][40]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = wave >> 1, qj = wave & 1;
  const int split = blockIdx.x;
  const int ob_base = a.ob_base, ib_base = a.ib_base;

  f32x16 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // column sums of pair 0's A blocks this wave stages (slots wave, wave + 4): per lane 4 TP rows per slot
  float colsum[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) colsum[q][i] = 0.0f;

  const int64_t t0 = (int64_t)split * a.tiles_per_split;
  int64_t t1 = t0 + a.tiles_per_split;
  if (t1 > a.n_tiles) t1 = a.n_tiles;
  const int n_t = t1 > t0 ? (int)(t1 - t0) : 0;
  const int n_stage = n_t * a.n_pairs;

  for (int i = tid; i < kWbLdsBytes / 16; i += 256) reinterpret_cast<f32x4*>(ldsb)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int slot[4] = {wave, wave + 4, 8 + wave, 12 + wave};
  bool valid[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) valid[q] = slot[q] < 8 ? (ob_base + slot[q] < a.nba) : (ib_base + slot[q] - 8 < a.nbb);
  const float* src0[4];
  const float* src1[4];
  int stride0[4], stride1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int blk = slot[q] < 8 ? ob_base + slot[q] : ib_base + slot[q] - 8;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const TpOperand& op = slot[q] < 8 ? a.A[pr] : a.B[pr];
      const int seg = blk >= op.nb[0];
      const int lb = blk - (seg ? op.nb[0] : 0);
      const float* base = op.ptr[seg] + (size_t)lb * 1024;
      const int stride = op.nb[seg] * 1024;
      if (pr == 0) {
        src0[q] = base;
        stride0[q] = stride;
      } else {
        src1[q] = base;
        stride1[q] = stride;
      }
    }
  }

  f32x4 pre[4][4];
  auto load_stage = [&](const int st) {
    const bool p1 = st >= n_t;
    const int64_t tile = t0 + (p1 ? st - n_t : st);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!valid[q]) continue;
      const float* base = p1 ? src1[q] : src0[q];
      const int stride = p1 ? stride1[q] : stride0[q];
      const f32x4* src = reinterpret_cast<const f32x4*>(base + (size_t)tile * stride) + lane;
#pragma unroll
      for (int i = 0; i < 4; ++i) pre[q][i] = src[i * 64];
    }
  };
  auto store_stage = [&](const int st) {
    const bool p1 = st >= n_t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!valid[q]) continue;
      // lane holds, for i = 0..3, TP row r = 4 i + (lane >> 4), half (lane >> 3) & 1, points 4 (lane & 7) .. + 3
      __bf16* dst = ldsb + slot[q] * kWbSlot + (lane & 7) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = pre[q][i];
        if (q < 2 && !p1) colsum[q][i] += (v[0] + v[1]) + (v[2] + v[3]);
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hi[e] = (__bf16)v[e];
          lo[e] = (__bf16)(v[e] - (float)hi[e]);
        }
        const int f = tp_row(i * 4 + (lane >> 4), (lane >> 3) & 1);
        *reinterpret_cast<bf16x4*>(dst + f * kWbRow) = hi;
        *reinterpret_cast<bf16x4*>(dst + kWbTile + f * kWbRow) = lo;
      }
    }
  };

  if (n_stage > 0) load_stage(0);
  // lane (row = lane & 31, k half = lane >> 5) reads 8 consecutive points of "its" feature row
  const __bf16* la = ldsb + qi * kWbSlot + (lane & 31) * kWbRow + 8 * (lane >> 5);
  const __bf16* lb = ldsb + (8 + qj) * kWbSlot + (lane & 31) * kWbRow + 8 * (lane >> 5);
  for (int st = 0; st < n_stage; ++st) {
    __syncthreads();
    store_stage(st);
    __syncthreads();
    if (st + 1 < n_stage) load_stage(st + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 ah[NA], al[NA], bh[NB], bl[NB];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(la + 2 * i * kWbSlot + kk * 16);
        al[i] = *reinterpret_cast<const bf16x8*>(la + 2 * i * kWbSlot + kWbTile + kk * 16);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(lb + 2 * j * kWbSlot + kk * 16);
        bl[j] = *reinterpret_cast<const bf16x8*>(lb + 2 * j * kWbSlot + kWbTile + kk * 16);
      }
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
    }
  }

  const int ldc = a.nbb * 32;
  const int hf = lane >> 5;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int ob = ob_base + qi + 2 * i, ib = ib_base + qj + 2 * j;
      if (qi + 2 * i < 8 && qj + 2 * j < 8 && ob < a.nba && ib < a.nbb) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          a.partial[wg_partial_index(split, (int)gridDim.x, ob * 32 + tp_row(r, hf), ib * 32 + (lane & 31), ldc)] = acc[i][j][r];
      }
    }
  if (a.bpartial != nullptr && ib_base == 0) {
    // each staged A row was summed over this lane's 4 points: finish over the 8 lanes that share the row
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t = colsum[q][i];
        t += __shfl_xor(t, 1);
        t += __shfl_xor(t, 2);
        t += __shfl_xor(t, 4);
        const int ob = ob_base + slot[q];
        if (valid[q] && (lane & 7) == 0 && ob < a.nba)
          a.bpartial[(size_t)split * a.nba * 32 + ob * 32 + tp_row(i * 4 + (lane >> 4), (lane >> 3) & 1)] = t;
      }
  }
}

