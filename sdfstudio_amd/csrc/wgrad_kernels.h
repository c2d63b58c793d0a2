// sdfhip — weight packing and weight-gradient kernels for the fused networks.
//
//  pack_kernel    natural [out][in] fp32 weights -> split-bf16 MFMA A-operand order  Wp[kb][part][ob][kk][lane][j]
//                 (and the transposed pack used by the chain / data-backward passes)
//  wgrad_kernel   split-K GEMM over points:  C[o][i] = sum_p A[p][o] * B[p][i]  with A, B tile-packed in HBM,
//                 one workgroup per (point-split, 8x8 block macro tile), tiles transposed through LDS; fp32 MFMA 32x32x2.
//                 Column sums of A (bias gradients) fall out of the same reads.
//  wreduce_kernel sums the split partials and scatters them into the natural-layout gradient vector.
#pragma once
#include "mlp_core.h"  // chunk layout constants of the packed weights

// One descriptor per packed matrix. All offsets are in floats / ints relative to the base pointers.
struct PackDesc {
  int64_t src_off;   // into theta (natural [n_rows][ld])
  int64_t dst_off;   // into the packed blob
  int32_t ld;        // natural leading dimension
  int32_t kb, nbo;   // packed k blocks / output blocks
  int32_t rowmap_off;  // int32[nbo*32]  packed output row -> natural row  (-1 = zero)
  int32_t colmap_off;  // int32[kb*32]   packed k index    -> natural col  (-1 = zero)
  int32_t transpose;   // 0: out = rows, k = cols ; 1: out = cols, k = rows  (packs W^T)
  float scale;
  int32_t pad_;
};

// grid = (ceil(kb*nbo*1024 / 256), n_desc).  One thread per weight: writes its three bf16 parts (w = w0 + w1 + w2, each part the
// bf16 rounding of what the previous parts left) and its two fp16 parts into the chunk layout of mlp_core.h:
//   Wp[kb][part 0..4][ob][kk][lane][j]  <-  W[out = 32 ob + (lane & 31)][k = 32 kb + tp_row(8 kk + j, lane >> 5)]
static __global__ void pack_kernel(const float* __restrict__ theta, const PackDesc* __restrict__ descs,
                            const int32_t* __restrict__ maps, float* __restrict__ packed) {
  const PackDesc d = descs[blockIdx.y];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = d.kb * d.nbo * 1024;
  if (idx >= total) return;
  const int j = idx & 7, lane = (idx >> 3) & 63, kk = (idx >> 9) & 1, ob = (idx >> 10) % d.nbo, kb = (idx >> 10) / d.nbo;
  const int o = ob * 32 + (lane & 31);                     // MFMA A-operand row  (output feature)
  const int k = kb * 32 + tp_row(kk * 8 + j, lane >> 5);   // contraction index in TP order
  const int32_t* rowmap = maps + d.rowmap_off;
  const int32_t* colmap = maps + d.colmap_off;
  float v = 0.0f;
  if (!d.transpose) {
    const int nr = rowmap[o], nc = colmap[k];
    if (nr >= 0 && nc >= 0) v = theta[d.src_off + (int64_t)nr * d.ld + nc] * d.scale;
  } else {
    const int nr = rowmap[k], nc = colmap[o];
    if (nr >= 0 && nc >= 0) v = theta[d.src_off + (int64_t)nr * d.ld + nc] * d.scale;
  }
  __bf16* chunk = reinterpret_cast<__bf16*>(packed + d.dst_off) + (size_t)kb * d.nbo * (2 * kChunkBlockFloats);
  const size_t at = ((size_t)(ob * 2 + kk) * 64 + lane) * 8 + j, part = (size_t)d.nbo * 1024;  // 16-bit elements per part
  float r = v;
#pragma unroll
  for (int q = 0; q < 3; ++q) {  // parts 0..2: bf16 (modes 2 and 3 of mlp_core.h)
    const __bf16 h = (__bf16)r;
    chunk[q * part + at] = h;
    r -= (float)h;
  }
  {  // parts 3, 4: fp16 hi + lo (mode 4), saturating like the activation split
    const float c = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
    const _Float16 h = (_Float16)c;
    _Float16* c16 = reinterpret_cast<_Float16*>(chunk);
    c16[3 * part + at] = h;
    c16[4 * part + at] = (_Float16)(c - (float)h);
  }
}

// padded natural-order vectors (biases, output rows): dst[i] = map[i] >= 0 ? theta[src_off + map[i]*stride] : 0
struct VecDesc {
  int64_t src_off, dst_off;
  int32_t n, map_off, stride, pad_;
};
static __global__ void packvec_kernel(const float* __restrict__ theta, const VecDesc* __restrict__ descs,
                               const int32_t* __restrict__ maps, float* __restrict__ packed) {
  const VecDesc d = descs[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.n) return;
  const int m = maps[d.map_off + i];
  packed[d.dst_off + i] = m >= 0 ? theta[d.src_off + (int64_t)m * d.stride] : 0.0f;
}

// ------------------------------------------------------------------------------------------------ wgrad
// C[o][i] = sum over points of A[p][o] * B[p][i]: a GEMM whose contraction runs over the 524 288 points, with both
// operands stored tile-packed (lane <-> point).  The MFMA wants lane <-> feature, so every 32-point tile is transposed
// through LDS: the workgroup (4 waves) loads the <= 8 A blocks and <= 8 B blocks of one tile with coalesced 1 KiB
// dwordx4 wave loads (each HBM byte is read once per macro tile; every operand is used as stored - the layer kernels save
// activations, not pre-activations), writes them as [block][feature 0..31][point 0..31] rows of 36 floats (ds_write_b128, conflict free),
// and each wave reads its 4 + 4 blocks back with ds_read_b128 (4 consecutive points of "its" feature; the 36-float row
// stride spreads a 16-lane read group over all 64 banks).  Each wave owns a 4x4-block (128 x 128) quadrant of the
// 256 x 256 macro tile: 256 accumulator registers, 256 MFMAs per tile against 32 LDS reads.  The next tile's global loads
// are in flight in registers while the current one is multiplied.  Split over points; partials are reduced by
// wreduce_kernel.  Column sums of pair 0's A (bias gradients) fall out of the A reads.
struct TpOperand {
  const float* ptr[2];  // up to two concatenated TP arrays
  int32_t nb[2];        // blocks in each
};
struct WgradArgs {
  TpOperand A[2], B[2];  // up to two (A, B) pairs accumulated into the same C
  int32_t n_pairs;
  int32_t nba, nbb;      // total blocks of A (rows of C / 32) and B (cols of C / 32)
  int32_t ob_base, ib_base;  // first row / column block of the macro tile this launch computes
  // quad != 0: ONE launch computes the 2 x 2 macro tiles of a 16 x 16-block C (hidden 512).  grid = 4 n_split; workgroup L works on macro
  // tile (L >> 3) & 3 of point split ((L >> 5) << 3) | (L & 7): workgroups go to the 8 XCDs round-robin, so the four macro tiles of a
  // split are resident TOGETHER on ONE XCD and stream the same point tiles at the same time - each operand block set (8 A or 8 B
  // blocks) is read by two of them and the second read is an L2 hit.  As four launches, each streamed its 2 KiB per point from HBM.
  int32_t quad, n_split;
  int64_t n_tiles;       // point tiles
  int32_t tiles_per_split;
  float* partial;        // [n_split][nba*32][nbb*32]
  float* bpartial;       // [n_split][nba*32]   column sums of pair 0's A   (may be null)
};

// Split-K partials are stored chunk-major: element idx = row * cols + col of split s lives at  P[idx / 256][s][idx % 256], so the
// 256 partials of 256 consecutive elements are ONE contiguous 256 KiB run that a wreduce block streams (split-major [s][rows][cols]
// put them 256 KiB apart: every load of the reduction opened a new page, 62 us for 64 MB out of the Infinity Cache).  A wave of the
// GEMM still writes whole 128-byte segments (32 consecutive columns of a row never straddle a chunk).
SDFHIP_D size_t wg_partial_index(const int split, const int n_split, const int row, const int col, const int ldc) {
  const int idx = row * ldc + col;
  return ((size_t)(idx >> 8) * n_split + split) * 256 + (idx & 255);
}

// ------------------------------------------------------------------------------------------------ wgrad, split-bf16 products
// Same GEMM, same data flow, but every fp32 operand x is split as x = hi + lo + O(2^-17 |x|) into two bf16 values when the tile is
// written to LDS, and a product a b is formed as a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation: 6 MFMAs of 32 cycles per 32 x 32 x 32 block product instead of 16 fp32 MFMAs of 64 cycles (5.3x less matrix-pipe
// time).  The dropped a_lo b_lo term and the split residual are ~2^-16 relative per product, below the fp32 round-off of the
// 524 288-term sums these gradients are (sqrt(N) 2^-24 ~ 4e-5); parameter-gradient parity (1e-3) is unaffected.  With the
// matrix pipe out of the way the kernel is bound by streaming the operand tiles from HBM.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kWbRow = 40;                  // bf16 per LDS row: 32 points + 8 pad (80 B: spreads a 16-lane b128 group over all banks)
constexpr int kWbTile = 32 * kWbRow;        // one hi or lo tile
constexpr int kWbSlot = 2 * kWbTile;        // [hi | lo]
constexpr int kWbLdsBytes = 16 * kWbSlot * 2;

// ---- 8-wave kernel.  (Rounds 1 - 2 also carried an exact-fp32 MFMA kernel and a 4-wave split-bf16 one behind environment switches:
// both superseded, moved out of the product to tools/legacy_wgrad_kernels.h.)  A 4-wave kernel is bound by the VALU work of the staging pass (bf16 split of 256 values
// per wave and tile (until round 3 also the softplus of saved pre-activations): ~1500 VALU instructions against 96 MFMAs, with the
// matrix pipe idle meanwhile and two barriers per tile).  Here the macro tile is shared by 8 waves (two per SIMD: their VALU
// streams issue side by side), each wave stages two blocks instead of four and owns a 2 x 4 block patch (128 accumulator
// registers), and LDS is double buffered so that tile t + 1 is split and written while tile t is multiplied: one barrier
// per tile.  LDS: 2 x 80 KiB = all of it.
constexpr int kW8LdsBytes = 2 * kWbLdsBytes;
constexpr int kWbBuf = 16 * kWbSlot;  // bf16 elements per stage buffer

template <int NA, int NB>
__global__ __launch_bounds__(512, 2) void wgrad_bf16x8_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsb[];  // [2 buffers][16 slots][hi|lo][32][40]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qi = wave >> 1, qj = wave & 1;  // row blocks qi + 4 i (i < NA), column blocks qj + 2 j (j < NB)
  const int L = blockIdx.x, macro = (L >> 3) & 3;
  const int split = a.quad ? (((L >> 5) << 3) | (L & 7)) : L;
  const int ob_base = a.quad ? (macro >> 1) * 8 : a.ob_base, ib_base = a.quad ? (macro & 1) * 8 : a.ib_base;

  f32x16 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float colsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // column sums of pair 0's A block this wave stages (slot `wave`): 4 TP rows per lane

  const int64_t t0 = (int64_t)split * a.tiles_per_split;
  int64_t t1 = t0 + a.tiles_per_split;
  if (t1 > a.n_tiles) t1 = a.n_tiles;
  const int n_t = t1 > t0 ? (int)(t1 - t0) : 0;
  const int n_stage = n_t * a.n_pairs;

  for (int i = tid; i < kW8LdsBytes / 16; i += 512) reinterpret_cast<f32x4*>(ldsb)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging role: slot `wave` (A block ob_base + wave) and slot 8 + wave (B block ib_base + wave).  A slot beyond the operand's extent
  // stages a copy of the operand's LAST block (clamped index: the loads hit the cache behind the wave that owns that block) - it feeds
  // accumulator blocks that are never written back.  That keeps the whole pipeline below free of branches, which is what lets the
  // compiler's s_waitcnt insertion count the loads in flight exactly (see the loop).
  const int slot[2] = {wave, 8 + wave};
  const bool valid0 = ob_base + wave < a.nba;
  const float* src0[2];
  const float* src1[2];
  int stride0[2], stride1[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int want = q == 0 ? ob_base + wave : ib_base + wave;
    const int blk = min(want, (q == 0 ? a.nba : a.nbb) - 1);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const TpOperand& op = q == 0 ? a.A[pr] : a.B[pr];
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
  if (a.n_pairs < 2) {  // pair 1 absent: never selected below (p1 is false for every stage), but keep the pointers valid
    src1[0] = src0[0], src1[1] = src0[1];
    stride1[0] = stride0[0], stride1[1] = stride0[1];
  }

  // Operand tile in flight from HBM: 8 staging units per wave and stage (unit u = 16-byte piece u & 3 of slot u >> 2), each in its own
  // four registers.  The load of unit u of stage st + 2 is issued the moment unit u of stage st + 1 has been consumed, so every load has
  // a whole stage to return and 8 are in flight per wave at all times; its consumer waits with vmcnt(7) - the seven younger loads stay
  // in flight.  (Until round 3 the eight loads of a stage were issued together at the end of the previous one, behind validity branches:
  // the compiler had to wait for five of them before the FIRST unit and the HBM latency was exposed once per stage.)
  f32x4 pre[2][4];
  auto load_unit = [&](const int st_want, auto qc, auto ic) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, i = decltype(ic)::value;
    const int st = min(st_want, n_stage - 1);  // past the end: re-load the last stage (never consumed into a buffer that is read)
    const bool p1 = st >= n_t;
    const int64_t tile = t0 + (p1 ? st - n_t : st);
    const float* base = p1 ? src1[q] : src0[q];
    const int stride = p1 ? stride1[q] : stride0[q];
    pre[q][i] = reinterpret_cast<const f32x4*>(base + (size_t)tile * stride)[lane + i * 64];
  };
  // one staging unit = one f32x4 (4 consecutive points of one TP row) of slot q: bias column sums, bf16 hi / lo split, two 8-byte LDS writes
  auto store_unit = [&](const int st, auto qc, auto ic) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, i = decltype(ic)::value;
    const bool p1 = st >= n_t;
    // lane holds, for i = 0..3, TP row r = 4 i + (lane >> 4), half (lane >> 3) & 1, points 4 (lane & 7) .. + 3
    __bf16* dst = ldsb + (st & 1) * kWbBuf + slot[q] * kWbSlot + (lane & 7) * 4;
    const f32x4 v = pre[q][i];
    if (q == 0) {
      const float t = (v[0] + v[1]) + (v[2] + v[3]);
      colsum[i] += p1 ? 0.0f : t;  // pair 0 only; "stage n_stage" (staged behind the last one, never multiplied) has st >= n_t as well
    }
    bf16x4 hi, lo;
#ifdef SDFHIP_ABL_WGRAD_NOSPLIT  // timing ablation (wrong numerics): what the staging pass would cost if the operands arrived split
    {
      typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
      u32x2_t t;
      t[0] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, v[1]), __builtin_bit_cast(uint32_t, v[0]), 0x07060302u);
      t[1] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, v[3]), __builtin_bit_cast(uint32_t, v[2]), 0x07060302u);
      hi = __builtin_bit_cast(bf16x4, t);
      lo = hi;
    }
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (__bf16)v[e];
      lo[e] = (__bf16)(v[e] - (float)hi[e]);
    }
#endif
    const int f = tp_row(i * 4 + (lane >> 4), (lane >> 3) & 1);
    *reinterpret_cast<bf16x4*>(dst + f * kWbRow) = hi;
    *reinterpret_cast<bf16x4*>(dst + kWbTile + f * kWbRow) = lo;
  };

  if (n_stage > 0) {
    static_for<0, 8>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      load_unit(0, std::integral_constant<int, (u >> 2)>{}, std::integral_constant<int, (u & 3)>{});
    });
    __syncthreads();  // the zero fill has landed
    static_for<0, 8>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      store_unit(0, std::integral_constant<int, (u >> 2)>{}, std::integral_constant<int, (u & 3)>{});
      load_unit(1, std::integral_constant<int, (u >> 2)>{}, std::integral_constant<int, (u & 3)>{});
    });
  }
  // lane (row = lane & 31, k half = lane >> 5) reads 8 consecutive points of "its" feature row
  const int frag = (lane & 31) * kWbRow + 8 * (lane >> 5);
  // One stage: 6 NA NB MFMAs on buffer st & 1 with the eight staging units of stage st + 1 (VALU + ds_write into the other
  // buffer, then the global load of the same unit of stage st + 2) placed between them at equal distances and pinned there.  The two
  // waves of a SIMD share its matrix pipe: whichever loses the arbitration for a run of MFMAs falls behind by that run and from then
  // on does its VALU unit while the partner multiplies - the pattern settles into complementary phases instead of both waves sitting
  // in their VALU part together.  The last iteration stages a "stage n_stage" nobody reads (no branch: see above).
  constexpr int M = 6 * NA * NB;
  for (int st = 0; st < n_stage; ++st) {
    __syncthreads();  // stage st is complete in buffer st & 1; everybody is done reading the other buffer (stage st - 1)
    const __bf16* la = ldsb + (st & 1) * kWbBuf + qi * kWbSlot + frag;
    const __bf16* lb = ldsb + (st & 1) * kWbBuf + (8 + qj) * kWbSlot + frag;
    static_for<0, 2>([&](auto kc) __attribute__((always_inline)) {
      constexpr int kk = decltype(kc)::value;
      bf16x8 ah[NA], al[NA], bh[NB], bl[NB];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(la + 4 * i * kWbSlot + kk * 16);
        al[i] = *reinterpret_cast<const bf16x8*>(la + 4 * i * kWbSlot + kWbTile + kk * 16);
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(lb + 2 * j * kWbSlot + kk * 16);
        bl[j] = *reinterpret_cast<const bf16x8*>(lb + 2 * j * kWbSlot + kWbTile + kk * 16);
      }
      static_for<0, 3 * NA * NB>([&](auto mc) __attribute__((always_inline)) {
        constexpr int mi = decltype(mc)::value, t = mi / (NA * NB), i = (mi / NB) % NA, j = mi % NB;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 2 ? al[i] : ah[i], t == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
        constexpr int m = kk * 3 * NA * NB + mi;                      // MFMA index within the stage
        constexpr int ulo = (m * 8) / M, uhi = ((m + 1) * 8) / M;      // staging units due after this MFMA
        if constexpr (ulo < uhi) {
          __builtin_amdgcn_sched_barrier(0);
          static_for<ulo, uhi>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            store_unit(st + 1, std::integral_constant<int, (u >> 2)>{}, std::integral_constant<int, (u & 3)>{});
            load_unit(st + 2, std::integral_constant<int, (u >> 2)>{}, std::integral_constant<int, (u & 3)>{});
          });
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    });
  }

  const int ldc = a.nbb * 32;
  const int hf = lane >> 5;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int ob = ob_base + qi + 4 * i, ib = ib_base + qj + 2 * j;
      if (qi + 4 * i < 8 && qj + 2 * j < 8 && ob < a.nba && ib < a.nbb) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          a.partial[wg_partial_index(split, a.n_split, ob * 32 + tp_row(r, hf), ib * 32 + (lane & 31), ldc)] = acc[i][j][r];
      }
    }
  if (a.bpartial != nullptr && ib_base == 0) {
    // each staged A row was summed over this lane's 4 points: finish over the 8 lanes that share the row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = colsum[i];
      t += __shfl_xor(t, 1);
      t += __shfl_xor(t, 2);
      t += __shfl_xor(t, 4);
      const int ob = ob_base + wave;
      if (valid0 && (lane & 7) == 0 && ob < a.nba)
        a.bpartial[(size_t)split * a.nba * 32 + ob * 32 + tp_row(i * 4 + (lane >> 4), (lane >> 3) & 1)] = t;
    }
  }
}

typedef void (*WgradKernelFn)(const WgradArgs);
// 8-wave kernel: na = ceil(row blocks / 4) in {1, 2}, nb = ceil(column blocks / 2) in {1..4}
static WgradKernelFn wgrad8_pick(const int na, const int nb) {
  if (na <= 1) {
    switch (nb) {
      case 1: return wgrad_bf16x8_kernel<1, 1>;
      case 2: return wgrad_bf16x8_kernel<1, 2>;
      case 3: return wgrad_bf16x8_kernel<1, 3>;
      default: return wgrad_bf16x8_kernel<1, 4>;
    }
  }
  switch (nb) {
    case 1: return wgrad_bf16x8_kernel<2, 1>;
    case 2: return wgrad_bf16x8_kernel<2, 2>;
    case 3: return wgrad_bf16x8_kernel<2, 3>;
    default: return wgrad_bf16x8_kernel<2, 4>;
  }
}

// out[dst_off + rowmap[o]*ld + colmap[i]] = scale * sum_s partial[s][o][i]      (grid-stride over o,i)
struct WreduceArgs {
  const float* partial;
  const float* bpartial;
  int32_t n_split, rows, cols;  // packed rows / cols (multiples of 32)
  const int32_t* rowmap;        // [rows] -> natural row or -1
  const int32_t* colmap;        // [cols] -> natural col or -1
  float* theta_bar;
  int64_t w_off;
  int32_t ld;
  float scale;
  int64_t b_off;   // bias gradient destination (index by natural row), or -1
  int32_t accumulate;
};
// block = 64 x kWrG threads: 64 x 4 consecutive elements (one 16-byte load per thread and split: a wave reads 1 KiB per instruction; with
// 4-byte loads the kernel was bound by the latency of 256-byte requests, PMC: 83 % of its wave cycles parked) x kWrG interleaved split
// groups, combined through LDS; grid = ceil(rows * cols / 256): one block streams the contiguous [n_split][256] run of its chunk
// (wg_partial_index), a thread sums n_split / kWrG partials in batches of 8.  The group count moves time between this kernel and the
// GEMM that follows it, not their sum (same-box kernel durations, tools/ab_rocprof.sh: 4 / 8 / 16 groups -> wreduce 67 / 43 / 107 us per
// launch, wgrad + wreduce 5.38 / 5.36 / 5.38 ms per step): what the pair costs is the write-back of the 64 MB of partials and their
// read, wherever the counters book it.
#ifndef SDFHIP_WREDUCE_GROUPS
#define SDFHIP_WREDUCE_GROUPS 8
#endif
constexpr int kWrG = SDFHIP_WREDUCE_GROUPS;
SDFHIP_D void wreduce_block(const WreduceArgs& a, const int bx, f32x4 (*red)[64]) {
  const int ix = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int total = a.rows * a.cols;  // a multiple of 1024 (both are multiples of 32)
  const int idx = (bx * 64 + ix) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (idx < total) {
    const float* p = a.partial + (size_t)(idx >> 8) * a.n_split * 256 + (idx & 255);  // chunk-major (wg_partial_index): split stride 256
    int k = sg;
    for (; k + 7 * kWrG < a.n_split; k += 8 * kWrG) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(k + kWrG * u) * 256);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < a.n_split; k += kWrG) s += *reinterpret_cast<const f32x4*>(p + (size_t)k * 256);
  }
  red[sg][ix] = s;
  __syncthreads();
  if (sg == 0 && idx < total) {
    s = red[0][ix];
#pragma unroll
    for (int g = 1; g < kWrG; ++g) s += red[g][ix];
    const int o = idx / a.cols, i0 = idx % a.cols;  // cols is a multiple of 4: the four elements share the row
    const int nr = a.rowmap[o];
    if (nr >= 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int nc = a.colmap[i0 + e];
        if (nc >= 0) {
          float* dst = a.theta_bar + a.w_off + (int64_t)nr * a.ld + nc;
          *dst = (a.accumulate ? *dst : 0.0f) + s[e] * a.scale;
        }
      }
    }
  }
  // bias gradients: rows elements, 64 per block, handled by the first ceil(rows / 64) blocks
  if (a.bpartial != nullptr && a.b_off >= 0 && bx * 64 < a.rows) {  // block-uniform condition (barriers inside)
    const int row = bx * 64 + ix;
    float t = 0.0f;
    if (row < a.rows)
      for (int k = sg; k < a.n_split; k += kWrG) t += a.bpartial[(size_t)k * a.rows + row];
    __syncthreads();
    red[sg][ix][0] = t;
    __syncthreads();
    if (sg == 0 && row < a.rows) {
      const int nr = a.rowmap[row];
      if (nr >= 0) {
        float tot = red[0][ix][0];
#pragma unroll
        for (int g = 1; g < kWrG; ++g) tot += red[g][ix][0];
        float* dst = a.theta_bar + a.b_off + nr;
        *dst = (a.accumulate ? *dst : 0.0f) + tot;
      }
    }
  }
}
static __global__ __launch_bounds__(64 * kWrG) void wreduce_kernel(const WreduceArgs a) {
  __shared__ f32x4 red[kWrG][64];
  wreduce_block(a, (int)blockIdx.x, red);
}
SDFHIP_HD int wreduce_blocks(const WreduceArgs& r) {
  // 256 elements per block; the bias rows (64 per block) need ceil(rows / 64) blocks, which rows * cols / 256 covers for cols >= 4
  const int total = r.rows * r.cols;
  const int a = (total + 255) / 256, b = (r.rows + 63) / 64;
  return a > b ? a : b;
}
// Every reduction of one backward call in ONE launch: each split-K GEMM keeps its own partial region and the reductions run together
// at the end of the call.  One reduction per GEMM (rounds 1 - 6) was 14 launches of ~44 us on config 2 (0.60 ms per step: 67 MB each, too
// short to stream) in BETWEEN the GEMMs; here the GEMMs follow each other and ~0.9 GB of partials are read by one launch that fills the
// chip: 0.17 ms.  The GEMMs lose part of it (their partials no longer sit in the Infinity Cache under the next GEMM's writes: 5.85 -> 6.15 ms
// by the events); net -0.07 ms per config-2 step, -0.16 ms per config-5 step, same box, alternating runs (profiles/r6_wgrad_launch_forms.txt).
// Same sums in the same order: the gradients are bit-identical to the per-GEMM form (tests/test_gpu_bitrepro.py).
// (Also measured there: the GEMMs of one kernel instantiation merged into ONE launch - no gain, +0.09 ms; not kept.)
constexpr int kWreduceBatchMax = 24;
struct WreduceBatchArgs {
  WreduceArgs r[kWreduceBatchMax];
  int32_t first_block[kWreduceBatchMax + 1];
  int32_t n;
};
static __global__ __launch_bounds__(64 * kWrG) void wreduce_batch_kernel(const WreduceBatchArgs a) {
  __shared__ f32x4 red[kWrG][64];
  const int b = (int)blockIdx.x;
  int i = 0;
  while (i + 1 < a.n && b >= a.first_block[i + 1]) ++i;  // wave-uniform, <= 24 steps
  wreduce_block(a.r[i], b - a.first_block[i], red);
}
