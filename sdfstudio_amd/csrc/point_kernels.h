// sdfhip — per-point kernels around the fused networks (gfx950; HBM / cache-bandwidth bound gather work):
//   geo_encode_kernel      start position -> L-inf contraction -> [x | NeRF-PE(x) | hash-grid features * mask]  (TP)
//                          + d feature / d p for the analytic normal            (sdf_field.py:380-392, 629)
//   grad_assemble_kernel   d sdf / d x = J_in0^T e ; builds the colour network's small-input block
//   bwd_prep_kernel        total d L / d grad -> tangent seed ebar = J_in0 gbar (TP), sdfbar padded
//   grid_bwd_kernel        scatter-add (fp32 hardware atomics) of first- and second-order terms into the table gradient
//   prop_fwd / prop_bwd    fused proposal density field: contraction, 5-level linear grid, 10->16->1 ReLU MLP, exp
//                          (density_fields.py:99-118, activations.py:23-39)
// The grid arithmetic restates tinycudann's GridEncoding (see oracle/hashgrid.py for the algorithm statement).
#pragma once
#include "common.h"

constexpr int kMaxLevels = 16;
struct GridLevelDev {
  float scale;
  uint32_t res;
  uint32_t size;    // entries in level
  uint32_t offset;  // first entry
  uint32_t hashed;
};
struct GridDev {
  int32_t n_levels, n_features, smoothstep, pad_;
  GridLevelDev lv[kMaxLevels];
};

// order: 1 = L-inf (surface models' default, base_surface_model.py:148-155), 2 = L2 (SceneContraction(order=None))
SDFHIP_D void contract_inf(float x[3], const int order = 1) {
  // spatial_distortions.py:66-73
  const float mag = order == 2 ? sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x[0], x[0]), __fmul_rn(x[1], x[1])), __fmul_rn(x[2], x[2])))
                               : fmaxf(fabsf(x[0]), fmaxf(fabsf(x[1]), fabsf(x[2])));
  if (mag >= 1.0f) {
    const float k = (2.0f - 1.0f / mag);
    x[0] = k * (x[0] / mag);
    x[1] = k * (x[1] / mag);
    x[2] = k * (x[2] / mag);
  }
}

struct GridCell {
  uint32_t idx[8];
  float w[3], dw[3];  // interpolation weight along each axis and its derivative w.r.t. the [0,1] position
};
SDFHIP_D void grid_cell(const GridLevelDev& L, const bool smooth, const float p[3], GridCell& c) {
  uint32_t g[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    // ONE rounding, as tiny-cuda-nn's pos_fract does: fmaf(scale, x, 0.5f) (grid.h); the oracle (oracle/hashgrid.py) forms the
    // same value through an exact fp64 product.  At scale ~2e3 the un-fused form moves the interpolation weight by up to 1e-4.
    const float pos = __builtin_fmaf(L.scale, p[d], 0.5f);
    const float fl = floorf(pos);
    const float f = pos - fl;
    g[d] = (uint32_t)(int)fl;
    if (smooth) {
      c.w[d] = f * f * (3.0f - 2.0f * f);
      c.dw[d] = 6.0f * f * (1.0f - f) * L.scale;
    } else {
      c.w[d] = f;
      c.dw[d] = L.scale;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t cx = g[0] + (k & 1), cy = g[1] + ((k >> 1) & 1), cz = g[2] + ((k >> 2) & 1);
    uint32_t idx;
    if (L.hashed) {
      idx = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & (L.size - 1u);  // hashed levels: size is a power of two
    } else {
      idx = cx + cy * L.res + cz * L.res * L.res;
      // tiny-cuda-nn reduces the uint32 sum modulo the level size (grid.h grid_index): inside the unit cube only the x == 1.0
      // face wraps, but get_sdf / gradient() take UNcontracted positions (sdf_field.py:412-418), which may lie outside it
      if (idx >= L.size) idx %= L.size;
    }
    c.idx[k] = L.offset + idx;
  }
}
SDFHIP_D float corner_w(const GridCell& c, const int k) {
  const float wx = (k & 1) ? c.w[0] : 1.0f - c.w[0];
  const float wy = (k & 2) ? c.w[1] : 1.0f - c.w[1];
  const float wz = (k & 4) ? c.w[2] : 1.0f - c.w[2];
  return wx * wy * wz;
}
// d corner weight / d p_d
SDFHIP_D void corner_dw(const GridCell& c, const int k, float out[3]) {
  const float wx = (k & 1) ? c.w[0] : 1.0f - c.w[0];
  const float wy = (k & 2) ? c.w[1] : 1.0f - c.w[1];
  const float wz = (k & 4) ? c.w[2] : 1.0f - c.w[2];
  out[0] = ((k & 1) ? c.dw[0] : -c.dw[0]) * wy * wz;
  out[1] = ((k & 2) ? c.dw[1] : -c.dw[1]) * wx * wz;
  out[2] = ((k & 4) ? c.dw[2] : -c.dw[2]) * wx * wy;
}

SDFHIP_D void start_position(const float* __restrict__ origins, const float* __restrict__ dirs,
                             const float* __restrict__ starts, const int64_t p, const int S, float x[3]) {
  if (dirs == nullptr) {  // origins already holds [P,3] positions
    x[0] = origins[p * 3 + 0];
    x[1] = origins[p * 3 + 1];
    x[2] = origins[p * 3 + 2];
  } else {
    const int64_t ray = p / S;
    const float t = starts[p];
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = origins[ray * 3 + d] + dirs[ray * 3 + d] * t;  // rays.py:61-73
  }
}


// ---- wave-level run reduction in front of the table-gradient atomics
// Lanes of a wavefront hold consecutive samples of a ray, so neighbouring lanes often fall into the same grid cell (always
// on the coarse levels, and more so once the sampler has concentrated the samples at the surface).  Device-scope fp32
// atomics execute memory-side on MI355X (~20 G/s chip-wide, measured; same-address updates serialise), so every run of
// equal indices is summed in registers first (segmented Hillis-Steele scan, 6 shuffle steps) and only the run's last lane
// issues the atomic.  Returns true for the lane that must issue; v0 / v1 then hold the run totals.
SDFHIP_D bool wave_run_reduce(const uint32_t idx, const bool active, float& v0, float& v1) {
  const int lane = threadIdx.x & 63;
  const uint32_t key = active ? idx : 0xffffffffu;  // inactive lanes never join a run of active ones
  const uint32_t prev = __shfl_up(key, 1);
  const uint32_t next = __shfl_down(key, 1);
  int head = (lane == 0) || (prev != key);
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float u0 = __shfl_up(v0, d);
    const float u1 = __shfl_up(v1, d);
    const int uh = __shfl_up(head, d);
    if (lane >= d && !head) {
      v0 += u0;
      v1 += u1;
      head |= uh;
    }
  }
  return active && ((lane == 63) || (next != key));
}

// ---- positional encoding of the geometry network's input (NeRFEncoding, field_components/encodings.py:118-208; include_input = False)
// `pe_code` = number of frequencies | (off_axis << 8).  Axis-aligned: the 3 coordinates; off_axis (:160,:191, the bakedsdf / bakedangelo
// field settings, configs/method_configs.py:270-286): the projections on the 21 icosahedron directions of self.P (:139-163).  Columns of the
// block (after the 3 raw coordinates): sin(p_j 2^f) at j * nf + f, then sin(p_j 2^f + pi / 2) at na * nf + j * nf + f.
constexpr int kPeOffAxis = 0x100;
SDFHIP_HD int pe_freqs(const int pe_code) { return pe_code & 0xff; }
SDFHIP_HD int pe_axes(const int pe_code) { return (pe_code & kPeOffAxis) ? 21 : 3; }
SDFHIP_HD int pe_cols(const int pe_code) { return 2 * pe_axes(pe_code) * pe_freqs(pe_code); }
__device__ __constant__ static const float kPeDirs[21][3] = {
    {0.8506508f, 0.f, 0.5257311f},    {0.809017f, 0.5f, 0.309017f},    {0.5257311f, 0.8506508f, 0.f},  {1.f, 0.f, 0.f},
    {0.809017f, 0.5f, -0.309017f},    {0.8506508f, 0.f, -0.5257311f},  {0.309017f, 0.809017f, -0.5f},  {0.f, 0.5257311f, -0.8506508f},
    {0.5f, 0.309017f, -0.809017f},    {0.f, 1.f, 0.f},                 {-0.5257311f, 0.8506508f, 0.f}, {-0.309017f, 0.809017f, -0.5f},
    {0.f, 0.5257311f, 0.8506508f},    {-0.309017f, 0.809017f, 0.5f},   {0.309017f, 0.809017f, 0.5f},   {0.5f, 0.309017f, 0.809017f},
    {0.5f, -0.309017f, 0.809017f},    {0.f, 0.f, 1.f},                 {-0.5f, 0.309017f, 0.809017f},  {-0.809017f, 0.5f, 0.309017f},
    {-0.809017f, 0.5f, -0.309017f}};
// the j-th encoded coordinate of v
SDFHIP_D float pe_project(const int pe_code, const float v[3], const int j) {
  if (!(pe_code & kPeOffAxis)) return v[j];
  return (v[0] * kPeDirs[j][0] + v[1] * kPeDirs[j][1]) + v[2] * kPeDirs[j][2];
}
// g += (d p_j / d v)^T s
SDFHIP_D void pe_project_T(const int pe_code, const int j, const float s, float g[3]) {
  if (!(pe_code & kPeOffAxis)) {
    g[j] += s;
    return;
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) g[d] = fmaf(kPeDirs[j][d], s, g[d]);
}

// ---- the colour network's small-input block (sdf_field.py:532-612 get_colors).  Column order = the reference's torch.cat order with the
// geometry feature taken out (it travels as its own blocks): [x (3)] [D (27): direction encoding of the view direction - or of the
// REFLECTED direction, use_reflections] [d sdf / dx (3)] [appearance embedding] [n . v (1), use_n_dot_v]; with use_diffuse_color x and the
// gradient are not inputs (:566-571).  Offsets; -1 = absent.
constexpr int kRefDiffuse = 1, kRefTint = 2, kRefReflect = 4, kRefNdotV = 8;
struct CsmallLayout {
  int x, D, g, emb, ndv, width;
};
SDFHIP_HD CsmallLayout csmall_layout(const int flags, const int emb_dim) {
  CsmallLayout L;
  int c = 0;
  const bool diffuse = (flags & kRefDiffuse) != 0;
  L.x = diffuse ? -1 : c;
  c += diffuse ? 0 : 3;
  L.D = c;
  c += 27;
  L.g = diffuse ? -1 : c;
  c += diffuse ? 0 : 3;
  L.emb = c;
  c += emb_dim;
  L.ndv = (flags & kRefNdotV) ? c : -1;
  c += (flags & kRefNdotV) ? 1 : 0;
  L.width = c;
  return L;
}
// normal n = g / max(|g|, 1e-12) (F.normalize, :543), c = n . d, and the direction the encoding D takes: d, or 2 (n . -d) n + d (:547)
struct RefGeom {
  float n[3], inv_len, c, r[3];
};
SDFHIP_D RefGeom ref_geom(const int flags, const float g[3], const float d[3]) {
  RefGeom R;
  const float len = sqrtf((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]);
  R.inv_len = 1.0f / fmaxf(len, 1e-12f);
#pragma unroll
  for (int k = 0; k < 3; ++k) R.n[k] = g[k] * R.inv_len;
  R.c = (R.n[0] * d[0] + R.n[1] * d[1]) + R.n[2] * d[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) R.r[k] = (flags & kRefReflect) ? 2.0f * (-R.c) * R.n[k] + d[k] : d[k];
  return R;
}

struct EncodeArgs {
  GridDev grid;
  const float* origins;  // [N,3] (or [P,3] positions when dirs == null)
  const float* dirs;     // [N,3] or null
  const float* starts;   // [N,S] or null
  int64_t n_points, n_padded;
  int32_t S, contract, pe_degree, use_pe;
  int32_t nb0;           // in0 blocks
  const float* table;    // [entries][F]
  const float* mask;     // [L*F]
  float* x_out;          // [n_padded][3] contracted positions
  float* in0_tp;         // [n_padded/32][nb0][16][64]
  float* dydp;           // [(L*F)*3][n_padded]  or null
  // numerical-gradient branch (sdf_field.py:431-453): n_points = 7 tap_points, point tap_points * k + i is point i displaced by the
  // k-th tap offset AFTER the contraction (k = 0: the point itself; 1..6: +d, -d along x, then y, then z).  0: plain points.
  int64_t tap_points;
  float tap_delta;
  const float* ends;     // [N,S] or null.  Non-null (with dirs): the frustum MID point o + d (start + end) / 2 (rays.py:46-55, what the
                         // density / nerfacto fields evaluate) instead of the start point
};

// position of encode point p: the start (or mid) position of point q = p mod tap_points (or p), contracted, then displaced by its tap
SDFHIP_D void encode_position(const EncodeArgs& a, const int64_t p, float x[3]) {
  const int64_t q = a.tap_points > 0 ? p % a.tap_points : p;
  if (a.ends != nullptr && a.dirs != nullptr) {
    const int64_t ray = q / a.S;
    const float t2 = a.starts[q] + a.ends[q];
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = a.origins[ray * 3 + d] + a.dirs[ray * 3 + d] * t2 * 0.5f;  // origins + directions * (starts + ends) / 2
  } else {
    start_position(a.origins, a.dirs, a.starts, q, a.S, x);
  }
  if (a.contract) contract_inf(x, a.contract);
  if (a.tap_points > 0) {
    const int tap = (int)(p / a.tap_points);
    if (tap > 0) x[(tap - 1) >> 1] += ((tap - 1) & 1) ? -a.tap_delta : a.tap_delta;
  }
}

// grid = (n_padded / 256, n_levels + 1), block = 256
// (Round 6 tried the launch in XCD order - level l gathered for ALL points by XCD l % 8, so that a level's table crosses the fabric once
// instead of eight times: FETCH_SIZE 264 -> 162 MB per launch, and the kernel 0.204 -> 0.243 ms, two alternating runs each on one box.
// The gather is not bound by the fabric: 67 M eight-byte gathers in 0.2 ms are 21 TB/s of 64-byte L2 -> L1 transfers against the L2s'
// 34.5 TB/s.  Not shipped; tools/experiments/encode_xcd_order.patch, profiles/r6_encode_xcd_order_ab.txt.)
__global__ __launch_bounds__(256) void geo_encode_kernel(const EncodeArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n_padded) return;
  const bool live = p < a.n_points;
  float x[3] = {0.f, 0.f, 0.f};
  if (live) encode_position(a, p, x);
  // An F-feature level is F / 2 two-feature gathers that share the cell (hash_features_per_level = 8 in the neus-facto-angelo preset:
  // geo_encode8_kernel below).
  // blockIdx.y = 0: the position / positional-encoding block (36 sinf per point: the longest blocks of the launch - scheduled FIRST, they
  // overlap with the gathers instead of forming its tail); y - 1 = level * (F / 2) + feature pair
  const int L = a.grid.n_levels, F = a.grid.n_features, pairs = F >> 1;
  const int yy = (int)blockIdx.y - 1;
  const int level = yy / pairs, pair = yy % pairs;
  const int pe_dims = pe_cols(a.pe_degree);
  const int feat0 = 3 + pe_dims;
  if (yy < 0) {
    // position + positional encoding + zero padding  (encodings.py:167-208: sin(cat[x f, x f + pi/2]))
    a.x_out[p * 3 + 0] = x[0];
    a.x_out[p * 3 + 1] = x[1];
    a.x_out[p * 3 + 2] = x[2];
#pragma unroll
    for (int d = 0; d < 3; ++d) a.in0_tp[tp_index(p, d, a.nb0)] = x[d];
    const int nf = pe_freqs(a.pe_degree), na = pe_axes(a.pe_degree);
    for (int j = 0; j < na; ++j) {
      const float pj = pe_project(a.pe_degree, x, j);
      for (int f = 0; f < nf; ++f) {
        const float u = pj * (float)(1 << f);
        const float s1 = (live && a.use_pe) ? sinf(u) : 0.0f;
        const float s2 = (live && a.use_pe) ? sinf(u + 1.57079632679489661923f) : 0.0f;
        a.in0_tp[tp_index(p, 3 + j * nf + f, a.nb0)] = s1;
        a.in0_tp[tp_index(p, 3 + na * nf + j * nf + f, a.nb0)] = s2;
      }
    }
    for (int c = feat0 + L * a.grid.n_features; c < a.nb0 * 32; ++c) a.in0_tp[tp_index(p, c, a.nb0)] = 0.0f;
    return;
  }
  // ---- one feature pair of one grid level
  const int c0 = level * F + pair * 2;
  const float m0 = a.mask[c0 + 0], m1 = a.mask[c0 + 1];
  float y0 = 0.f, y1 = 0.f, d0[3] = {0.f, 0.f, 0.f}, d1[3] = {0.f, 0.f, 0.f};
  if (live && (m0 != 0.0f || m1 != 0.0f)) {
    const float pp[3] = {(x[0] + 2.0f) * 0.25f, (x[1] + 2.0f) * 0.25f, (x[2] + 2.0f) * 0.25f};  // sdf_field.py:384
    GridCell c;
    grid_cell(a.grid.lv[level], a.grid.smoothstep != 0, pp, c);
    const float2* tab = reinterpret_cast<const float2*>(a.table);
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tab[(size_t)c.idx[k] * pairs + pair];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = corner_w(c, k);
      y0 = fmaf(w, v[k].x, y0);
      y1 = fmaf(w, v[k].y, y1);
      if (a.dydp != nullptr) {
        float dw[3];
        corner_dw(c, k, dw);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          d0[d] = fmaf(dw[d], v[k].x, d0[d]);
          d1[d] = fmaf(dw[d], v[k].y, d1[d]);
        }
      }
    }
  }
  a.in0_tp[tp_index(p, feat0 + c0 + 0, a.nb0)] = y0 * m0;
  a.in0_tp[tp_index(p, feat0 + c0 + 1, a.nb0)] = y1 * m1;
  if (a.dydp != nullptr) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      a.dydp[(size_t)((c0 + 0) * 3 + d) * a.n_padded + p] = d0[d];
      a.dydp[(size_t)((c0 + 1) * 3 + d) * a.n_padded + p] = d1[d];
    }
  }
}

// ---- 8 features per entry (BASELINE config 5): one block gathers WHOLE 32-byte entries (two 16-byte loads per corner) and produces all 8
// features of a level.  With one feature pair per block (above) the four pair-blocks of a level visit every entry again, each for 8 of
// its 32 bytes: on a 2.1 GB table those visits are HBM sectors fetched four times (PMC: 7.3 GB per step for 3.4 GB of entries).
// grid = (n_padded / 256, n_levels + 1), block = 256; y = 0 writes position / positional encoding / padding like the kernel above.
__global__ __launch_bounds__(256) void geo_encode8_kernel(const EncodeArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n_padded) return;
  const bool live = p < a.n_points;
  float x[3] = {0.f, 0.f, 0.f};
  if (live) encode_position(a, p, x);
  const int L = a.grid.n_levels;
  const int level = (int)blockIdx.y - 1;  // y = 0: the position / positional-encoding block, first (see geo_encode_kernel)
  const int pe_dims = pe_cols(a.pe_degree);
  const int feat0 = 3 + pe_dims;
  if (level < 0) {
    a.x_out[p * 3 + 0] = x[0];
    a.x_out[p * 3 + 1] = x[1];
    a.x_out[p * 3 + 2] = x[2];
#pragma unroll
    for (int d = 0; d < 3; ++d) a.in0_tp[tp_index(p, d, a.nb0)] = x[d];
    const int nf = pe_freqs(a.pe_degree), na = pe_axes(a.pe_degree);
    for (int j = 0; j < na; ++j) {
      const float pj = pe_project(a.pe_degree, x, j);
      for (int f = 0; f < nf; ++f) {
        const float u = pj * (float)(1 << f);
        const float s1 = (live && a.use_pe) ? sinf(u) : 0.0f;
        const float s2 = (live && a.use_pe) ? sinf(u + 1.57079632679489661923f) : 0.0f;
        a.in0_tp[tp_index(p, 3 + j * nf + f, a.nb0)] = s1;
        a.in0_tp[tp_index(p, 3 + na * nf + j * nf + f, a.nb0)] = s2;
      }
    }
    for (int c = feat0 + L * 8; c < a.nb0 * 32; ++c) a.in0_tp[tp_index(p, c, a.nb0)] = 0.0f;
    return;
  }
  const int c0 = level * 8;
  float m[8], y[8], dy[8][3];
  bool any = false;
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    m[f] = a.mask[c0 + f];
    any |= m[f] != 0.0f;
    y[f] = 0.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) dy[f][d] = 0.0f;
  }
  if (live && any) {
    const float pp[3] = {(x[0] + 2.0f) * 0.25f, (x[1] + 2.0f) * 0.25f, (x[2] + 2.0f) * 0.25f};
    GridCell c;
    grid_cell(a.grid.lv[level], a.grid.smoothstep != 0, pp, c);
    const f32x4* tab = reinterpret_cast<const f32x4*>(a.table);
    f32x4 v[8][2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k][0] = tab[(size_t)c.idx[k] * 2];
      v[k][1] = tab[(size_t)c.idx[k] * 2 + 1];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = corner_w(c, k);
      float dw[3] = {0.f, 0.f, 0.f};
      if (a.dydp != nullptr) corner_dw(c, k, dw);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const float t = v[k][f >> 2][f & 3];
        y[f] = fmaf(w, t, y[f]);
        if (a.dydp != nullptr) {
#pragma unroll
          for (int d = 0; d < 3; ++d) dy[f][d] = fmaf(dw[d], t, dy[f][d]);
        }
      }
    }
  }
#pragma unroll
  for (int f = 0; f < 8; ++f) a.in0_tp[tp_index(p, feat0 + c0 + f, a.nb0)] = y[f] * m[f];
  if (a.dydp != nullptr) {
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int d = 0; d < 3; ++d) a.dydp[(size_t)((c0 + f) * 3 + d) * a.n_padded + p] = dy[f][d];
  }
}

struct AssembleArgs {
  const float* e_tp;    // [T][nb0]  d sdf / d in0
  const float* x;       // [n_padded][3]
  const float* dydp;    // [(L*F)*3][n_padded]
  const float* mask;    // [L*F]
  const float* dirs;    // [N,3]
  const float* emb;     // [N][emb_dim] per-ray appearance embedding or null (zeros)
  int64_t n_points, n_padded;
  int32_t S, pe_degree, use_pe, n_feat, nb0, nbs, emb_dim;
  int32_t ref_flags;    // kRef*: the ref-nerf options of get_colors (sdf_field.py:536-549, 566-583)
  float* grad;          // [n_padded][3]  (null with grad_in)
  float* csmall_tp;     // [T][nbs]
  const float* grad_in; // [P][3] or null: take d sdf / dx from the caller instead of assembling it from e_tp (numerical gradients)
  float* g_save;        // [n_padded][3] or null: d sdf / dx and the ray direction per point, kept for the backward of the ref-nerf
  float* d_save;        // [n_padded][3] or null  inputs that depend on the normal (bwd_prep_kernel)
};

// block = 256 threads over points
__global__ __launch_bounds__(256) void grad_assemble_kernel(const AssembleArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n_padded) return;
  const bool live = p < a.n_points;
  float g[3] = {0.f, 0.f, 0.f}, x[3] = {0.f, 0.f, 0.f}, dir[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      x[d] = a.x[p * 3 + d];
      g[d] = a.grad_in != nullptr ? a.grad_in[p * 3 + d] : a.e_tp[tp_index(p, d, a.nb0)];
    }
    if (a.use_pe && a.grad_in == nullptr) {
      const int nf = pe_freqs(a.pe_degree), na = pe_axes(a.pe_degree);
      for (int j = 0; j < na; ++j) {
        const float pj = pe_project(a.pe_degree, x, j);
        for (int f = 0; f < nf; ++f) {
          const float fr = (float)(1 << f);
          const float u = pj * fr;
          const float e1 = a.e_tp[tp_index(p, 3 + j * nf + f, a.nb0)];
          const float e2 = a.e_tp[tp_index(p, 3 + na * nf + j * nf + f, a.nb0)];
          pe_project_T(a.pe_degree, j, fr * (cosf(u) * e1 + cosf(u + 1.57079632679489661923f) * e2), g);
        }
      }
    }
    const int feat0 = 3 + pe_cols(a.pe_degree);
    for (int c = 0; c < (a.grad_in == nullptr ? a.n_feat : 0); ++c) {
      const float ec = a.e_tp[tp_index(p, feat0 + c, a.nb0)] * a.mask[c] * 0.25f;
#pragma unroll
      for (int d = 0; d < 3; ++d) g[d] = fmaf(ec, a.dydp[(size_t)(c * 3 + d) * a.n_padded + p], g[d]);
    }
    const int64_t ray = p / a.S;
#pragma unroll
    for (int d = 0; d < 3; ++d) dir[d] = a.dirs[ray * 3 + d];
  }
  if (a.grad != nullptr) {
#pragma unroll
    for (int d = 0; d < 3; ++d) a.grad[p * 3 + d] = g[d];
  }
  if (a.g_save != nullptr) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      a.g_save[p * 3 + d] = g[d];
      a.d_save[p * 3 + d] = dir[d];
    }
  }
  // colour-network small inputs (csmall_layout): [x] | D = PE(r): sin(r 2^f) (12), sin(r 2^f + pi/2) (12), r (3) | [grad] | emb | [n . v]
  const CsmallLayout L = csmall_layout(a.ref_flags, a.emb_dim);
  RefGeom R;
  if (a.ref_flags & (kRefReflect | kRefNdotV)) R = ref_geom(a.ref_flags, g, dir);
  else {
#pragma unroll
    for (int d = 0; d < 3; ++d) R.r[d] = dir[d];
  }
  if (L.x >= 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) a.csmall_tp[tp_index(p, L.x + d, a.nbs)] = x[d];
  }
  for (int d = 0; d < 3; ++d)
    for (int f = 0; f < 4; ++f) {
      const float u = R.r[d] * (float)(1 << f);
      a.csmall_tp[tp_index(p, L.D + d * 4 + f, a.nbs)] = live ? sinf(u) : 0.0f;
      a.csmall_tp[tp_index(p, L.D + 12 + d * 4 + f, a.nbs)] = live ? sinf(u + 1.57079632679489661923f) : 0.0f;
    }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    a.csmall_tp[tp_index(p, L.D + 24 + d, a.nbs)] = live ? R.r[d] : 0.0f;
    if (L.g >= 0) a.csmall_tp[tp_index(p, L.g + d, a.nbs)] = g[d];
  }
  for (int j = 0; j < a.emb_dim; ++j) {
    float v = 0.0f;
    if (live && a.emb != nullptr) v = a.emb[(p / a.S) * a.emb_dim + j];
    a.csmall_tp[tp_index(p, L.emb + j, a.nbs)] = v;
  }
  if (L.ndv >= 0) a.csmall_tp[tp_index(p, L.ndv, a.nbs)] = live ? R.c : 0.0f;
  for (int c = L.width; c < a.nbs * 32; ++c) a.csmall_tp[tp_index(p, c, a.nbs)] = 0.0f;
}

struct BwdPrepArgs {
  const float* gradbar;       // [P][3] upstream d L / d (d sdf/dx)  or null
  const float* sdfbar_in;     // [P] upstream or null
  const float* csmallbar_tp;  // [T][nbs] from the colour backward or null
  const float* x;
  const float* dydp;
  const float* mask;
  int64_t n_points, n_padded;
  int32_t pe_degree, use_pe, n_feat, nb0, nbs, emb_dim;
  int32_t S, ref_flags;
  const float* d_pt;          // [n_padded][3] ray direction and d sdf / dx per point (AssembleArgs::d_save / g_save; ref-nerf options only: the
  const float* g_pt;          // [n_padded][3] normal enters the colour inputs through the reflected direction and n . v - sdf_field.py:543-549,
                              //               580-583 - and their cotangents come back through it)
  float* gtot;       // [n_padded][3]
  float* ebar_tp;    // [T][nb0]
  float* sdfbar;     // [n_padded]
  float* embbar;     // [N][emb_dim] accumulated with atomics, or null
};

// d L / d (d sdf / dx) out of the colour backward's small-input cotangents: the gradient slot itself (when it is an input), plus - with
// use_reflections / use_n_dot_v - the chain through the reflected direction's encoding and n . v back to the normal and through F.normalize
SDFHIP_D void csmall_grad_cotangent(const float* __restrict__ csmallbar_tp, const int64_t p, const int nbs, const int flags, const CsmallLayout& L,
                                    const float g[3], const float dir[3], float gb[3]) {
  if (L.g >= 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) gb[d] += csmallbar_tp[tp_index(p, L.g + d, nbs)];
  }
  if (!(flags & (kRefReflect | kRefNdotV))) return;
  const RefGeom R = ref_geom(flags, g, dir);
  float nbar[3] = {0.f, 0.f, 0.f};
  if (flags & kRefReflect) {
    float rbar[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float acc = csmallbar_tp[tp_index(p, L.D + 24 + d, nbs)];
      for (int f = 0; f < 4; ++f) {
        const float fr = (float)(1 << f);
        const float u = R.r[d] * fr;
        acc += fr * (cosf(u) * csmallbar_tp[tp_index(p, L.D + d * 4 + f, nbs)] +
                     cosf(u + 1.57079632679489661923f) * csmallbar_tp[tp_index(p, L.D + 12 + d * 4 + f, nbs)]);
      }
      rbar[d] = acc;
    }
    const float nr = (R.n[0] * rbar[0] + R.n[1] * rbar[1]) + R.n[2] * rbar[2];
#pragma unroll
    for (int d = 0; d < 3; ++d) nbar[d] = -2.0f * (nr * dir[d] + R.c * rbar[d]);  // r = d - 2 (n . d) n
  }
  if (flags & kRefNdotV) {
    const float cb = csmallbar_tp[tp_index(p, L.ndv, nbs)];
#pragma unroll
    for (int d = 0; d < 3; ++d) nbar[d] = fmaf(cb, dir[d], nbar[d]);
  }
  const float len2 = (g[0] * g[0] + g[1] * g[1]) + g[2] * g[2];
  const float nn = len2 > 1e-24f ? (R.n[0] * nbar[0] + R.n[1] * nbar[1]) + R.n[2] * nbar[2] : 0.0f;  // clamped denominator: n = g / eps is linear
#pragma unroll
  for (int d = 0; d < 3; ++d) gb[d] += (nbar[d] - R.n[d] * nn) * R.inv_len;
}

__global__ __launch_bounds__(256) void bwd_prep_kernel(const BwdPrepArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n_padded) return;
  const bool live = p < a.n_points;
  const CsmallLayout L = csmall_layout(a.ref_flags, a.emb_dim);
  float gb[3] = {0.f, 0.f, 0.f}, x[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      x[d] = a.x[p * 3 + d];
      if (a.gradbar != nullptr) gb[d] = a.gradbar[p * 3 + d];
    }
    if (a.csmallbar_tp != nullptr) {
      float g[3] = {0.f, 0.f, 0.f}, dir[3] = {0.f, 0.f, 0.f};
      if (a.ref_flags & (kRefReflect | kRefNdotV)) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          g[d] = a.g_pt[p * 3 + d];
          dir[d] = a.d_pt[p * 3 + d];
        }
      }
      csmall_grad_cotangent(a.csmallbar_tp, p, a.nbs, a.ref_flags, L, g, dir, gb);
    }
    if (a.embbar != nullptr && a.csmallbar_tp != nullptr) {
      for (int j = 0; j < a.emb_dim; ++j)
        atomicAdd(a.embbar + (p / a.S) * a.emb_dim + j, a.csmallbar_tp[tp_index(p, L.emb + j, a.nbs)]);
    }
  }
  a.sdfbar[p] = (live && a.sdfbar_in != nullptr) ? a.sdfbar_in[p] : 0.0f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    a.gtot[p * 3 + d] = gb[d];
    a.ebar_tp[tp_index(p, d, a.nb0)] = gb[d];
  }
  {
    const int nf = pe_freqs(a.pe_degree), na = pe_axes(a.pe_degree);
    for (int j = 0; j < na; ++j) {
      const float pj = pe_project(a.pe_degree, x, j), gbj = pe_project(a.pe_degree, gb, j);
      for (int f = 0; f < nf; ++f) {
        const float fr = (float)(1 << f);
        const float u = pj * fr;
        const float c1 = (live && a.use_pe) ? fr * cosf(u) * gbj : 0.0f;
        const float c2 = (live && a.use_pe) ? fr * cosf(u + 1.57079632679489661923f) * gbj : 0.0f;
        a.ebar_tp[tp_index(p, 3 + j * nf + f, a.nb0)] = c1;
        a.ebar_tp[tp_index(p, 3 + na * nf + j * nf + f, a.nb0)] = c2;
      }
    }
  }
  const int feat0 = 3 + pe_cols(a.pe_degree);
  for (int c = 0; c < a.n_feat; ++c) {
    float v = 0.0f;
    if (live) {
#pragma unroll
      for (int d = 0; d < 3; ++d) v = fmaf(a.dydp[(size_t)(c * 3 + d) * a.n_padded + p], gb[d], v);
      v *= 0.25f * a.mask[c];
    }
    a.ebar_tp[tp_index(p, feat0 + c, a.nb0)] = v;
  }
  for (int c = feat0 + a.n_feat; c < a.nb0 * 32; ++c) a.ebar_tp[tp_index(p, c, a.nb0)] = 0.0f;
}

struct GridBwdArgs {
  GridDev grid;
  const float* x;          // [n_padded][3] contracted positions
  const float* in0bar_tp;  // [T][nb0]  d L / d in0 (first-order)
  const float* e_tp;       // [T][nb0]  d sdf / d in0        (second-order term) or null
  const float* gtot;       // [n_padded][3] d L / d grad     (second-order term) or null
  const float* mask;
  int64_t n_points;
  int32_t pe_degree, nb0;
  float* tablebar;         // [entries][F]  (accumulated)
  // numerical-gradient branch: the n_points = 7 tap_points points are tap-major (EncodeArgs::tap_points), the taps tap_delta apart (in
  // contracted space; x 0.25 in the grid's domain).  0: plain points.  grid_bwd8_kernel then puts the 7 taps of a sample into ADJACENT
  // lanes on every level whose cells are wider than 2 delta, where they mostly share the cell: the run reduction in front of the atomics
  // merges them (late in neus-facto-angelo's schedule delta is a quarter of the FINEST cell: up to 7 x fewer line updates).
  int64_t tap_points;
  float tap_delta;
};

// ---- line-coalesced issue of the table-gradient atomics
// The memory-side atomic unit is priced per (instruction, 64-byte line) pair: ~21 G line-updates/s chip-wide whether the
// lanes of one instruction touch 64 lines or 16 (tools/probe_atomic.hip: 21 / 42 / 84 / 334 G adds/s with 1 / 2 / 4 / 16
// lanes per line).  One point's contributions to an entry pair (x, x + 1) x (feature 0, 1) are 16 contiguous bytes on the
// dense levels (and on the hashed levels whenever x is even: the hash multiplies x by 1), so the 16 adds of a point are
// transposed through LDS: instruction (c, q) carries, in lanes 4 i .. 4 i + 3, the x-pair of corner pair c of point
// 16 q + i  ->  16 lines per instruction instead of 64.
struct ScatterStage {
  float v[64][17];     // [point of the wave][corner * 2 + feature], row padded against bank conflicts
  uint32_t e[64][9];   // [point][corner] table entry, or kNoEntry when the run reduction gave the add to another lane
};
constexpr uint32_t kNoEntry = 0xffffffffu;
SDFHIP_D void scatter_stage_put(ScatterStage& st, const int lane, const int k, const bool issue, const uint32_t entry, const float t0,
                                const float t1) {
  st.v[lane][2 * k] = t0;
  st.v[lane][2 * k + 1] = t1;
  st.e[lane][k] = issue ? entry : kNoEntry;
}
// entry_stride = features per table entry, feat_off = first feature of the pair being scattered
SDFHIP_D void scatter_stage_flush(const ScatterStage& st, const int lane, float* __restrict__ tablebar, const int entry_stride = 2,
                                  const int feat_off = 0) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // this wave's LDS writes are ordered before its reads below
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pt = q * 16 + (lane >> 2), k = c * 2 + ((lane >> 1) & 1), f = lane & 1;
      const uint32_t entry = st.e[pt][k];
      const float v = st.v[pt][2 * k + f];
      if (entry != kNoEntry) atomicAdd(tablebar + (size_t)entry * entry_stride + feat_off + f, v);
    }
  __builtin_amdgcn_wave_barrier();
}

// grid = (ceil(P/256), n_levels * F / 2): one feature pair of one level per y
__global__ __launch_bounds__(256) void grid_bwd_kernel(const GridBwdArgs a) {
  __shared__ ScatterStage stage[4];
  const int lane = threadIdx.x & 63;
  ScatterStage& st = stage[threadIdx.x >> 6];
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < a.n_points;
  const int F = a.grid.n_features, pairs = F >> 1;
  const int level = blockIdx.y / pairs, pair = blockIdx.y % pairs, c0 = level * F + pair * 2;
  const float m0 = a.mask[c0 + 0], m1 = a.mask[c0 + 1];
  if (m0 == 0.0f && m1 == 0.0f) return;  // block-uniform
  const int feat0 = 3 + pe_cols(a.pe_degree);
  float yb0 = 0.f, yb1 = 0.f, e0 = 0.f, e1 = 0.f, gb[3] = {0.f, 0.f, 0.f}, pp[3] = {0.5f, 0.5f, 0.5f};
  const bool second = a.e_tp != nullptr && a.gtot != nullptr;
  if (live) {
    yb0 = a.in0bar_tp[tp_index(p, feat0 + c0 + 0, a.nb0)] * m0;
    yb1 = a.in0bar_tp[tp_index(p, feat0 + c0 + 1, a.nb0)] * m1;
    if (second) {
      e0 = a.e_tp[tp_index(p, feat0 + c0 + 0, a.nb0)] * m0 * 0.25f;
      e1 = a.e_tp[tp_index(p, feat0 + c0 + 1, a.nb0)] * m1 * 0.25f;
#pragma unroll
      for (int d = 0; d < 3; ++d) gb[d] = a.gtot[p * 3 + d];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) pp[d] = (a.x[p * 3 + d] + 2.0f) * 0.25f;
  }
  GridCell c;
  grid_cell(a.grid.lv[level], a.grid.smoothstep != 0, pp, c);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = corner_w(c, k);
    float t0 = w * yb0, t1 = w * yb1;
    if (second) {
      float dw[3];
      corner_dw(c, k, dw);
      const float s = dw[0] * gb[0] + dw[1] * gb[1] + dw[2] * gb[2];
      t0 = fmaf(s, e0, t0);
      t1 = fmaf(s, e1, t1);
    }
    const bool issue = wave_run_reduce(c.idx[k], live, t0, t1);
    scatter_stage_put(st, lane, k, issue, c.idx[k], t0, t1);
  }
  scatter_stage_flush(st, lane, a.tablebar, F, pair * 2);
}

// ---- 8 features per entry (BASELINE config 5: 16 x 8 x 2^22): one entry is 32 bytes, an x-pair of entries one 64-byte line.  With one
// feature PAIR per block (the kernel above) the four pair-blocks of a level each touch every line of the level again - 4 lanes per line
// and instruction.  Here one block scatters all 8 features of a level: an instruction carries, in lanes 16 i .. 16 i + 15, the x-pair x
// 8 features of one corner pair of point 4 q + i - whole lines, 4 per instruction instead of 16, a quarter of the (instruction, line)
// pairs the memory-side atomic unit is priced by (tools/probe_atomic.hip: 84 -> 334 G adds/s).
template <int NV>
SDFHIP_D bool wave_run_reduce_n(const uint32_t idx, const bool active, float (&v)[NV]) {
  const int lane = threadIdx.x & 63;
  const uint32_t key = active ? idx : 0xffffffffu;
  const uint32_t prev = __shfl_up(key, 1);
  const uint32_t next = __shfl_down(key, 1);
  int head = (lane == 0) || (prev != key);
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float u[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) u[i] = __shfl_up(v[i], d);
    const int uh = __shfl_up(head, d);
    if (lane >= d && !head) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] += u[i];
      head |= uh;
    }
  }
  return active && ((lane == 63) || (next != key));
}
struct ScatterStage8 {
  float v[64][33];     // [point of the wave][(corner & 3) * 8 + feature]: four corners at a time, row padded against bank conflicts
  uint32_t e[64][5];   // [point][corner & 3] table entry, or kNoEntry
};
// grid = (ceil(P/256), n_levels): all 8 features of one level per y
__global__ __launch_bounds__(256) void grid_bwd8_kernel(const GridBwdArgs a) {
  __shared__ ScatterStage8 stage[4];
  const int lane = threadIdx.x & 63;
  ScatterStage8& st = stage[threadIdx.x >> 6];
  const int level = blockIdx.y, c0 = level * 8;
  int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool live = p < a.n_points, filler = false;
  if (a.tap_points > 0 && a.tap_delta * 0.25f * a.grid.lv[level].scale < 0.5f) {
    // taps of a sample in lanes 8 j .. 8 j + 6; lane 8 j + 7 is a zero-valued filler that repeats its left neighbour's entries, so that
    // runs still extend from one sample to the next along the ray
    const int64_t t = p;
    const int slot = (int)(t & 7);
    const int64_t i = t >> 3;
    live = i < a.tap_points && slot < 7;
    filler = i < a.tap_points && slot == 7;
    p = slot * a.tap_points + i;
  }
  float m[8];
  bool any = false;
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    m[f] = a.mask[c0 + f];
    any |= m[f] != 0.0f;
  }
  if (!any) return;  // block-uniform: a level that is still switched off (progressive levels)
  const int feat0 = 3 + pe_cols(a.pe_degree);
  float yb[8], e8[8], gb[3] = {0.f, 0.f, 0.f}, pp[3] = {0.5f, 0.5f, 0.5f};
  const bool second = a.e_tp != nullptr && a.gtot != nullptr;
#pragma unroll
  for (int f = 0; f < 8; ++f) yb[f] = e8[f] = 0.0f;
  if (live) {
#pragma unroll
    for (int f = 0; f < 8; ++f) yb[f] = a.in0bar_tp[tp_index(p, feat0 + c0 + f, a.nb0)] * m[f];
    if (second) {
#pragma unroll
      for (int f = 0; f < 8; ++f) e8[f] = a.e_tp[tp_index(p, feat0 + c0 + f, a.nb0)] * m[f] * 0.25f;
#pragma unroll
      for (int d = 0; d < 3; ++d) gb[d] = a.gtot[p * 3 + d];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) pp[d] = (a.x[p * 3 + d] + 2.0f) * 0.25f;
  }
  GridCell c;
  grid_cell(a.grid.lv[level], a.grid.smoothstep != 0, pp, c);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = half * 4 + kk;
      const float w = corner_w(c, k);
      float t[8];
      float s = 0.0f;
      if (second) {
        float dw[3];
        corner_dw(c, k, dw);
        s = dw[0] * gb[0] + dw[1] * gb[1] + dw[2] * gb[2];
      }
#pragma unroll
      for (int f = 0; f < 8; ++f) t[f] = fmaf(s, e8[f], w * yb[f]);
      uint32_t idx = c.idx[k];
      const uint32_t idx_left = __shfl_up(idx, 1);
      if (filler) idx = idx_left;  // values are zero (yb = e8 = 0 for a lane that is not live)
      const bool issue = wave_run_reduce_n<8>(idx, live || filler, t);
#pragma unroll
      for (int f = 0; f < 8; ++f) st.v[lane][kk * 8 + f] = t[f];
      st.e[lane][kk] = issue ? idx : kNoEntry;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int cp = 0; cp < 2; ++cp)
#pragma unroll 4
      for (int q = 0; q < 16; ++q) {
        const int pt = q * 4 + (lane >> 4), kk = cp * 2 + ((lane >> 3) & 1), f = lane & 7;
        const uint32_t entry = st.e[pt][kk];
        const float v = st.v[pt][kk * 8 + f];
        if (entry != kNoEntry) atomicAdd(a.tablebar + (size_t)entry * 8 + f, v);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------ proposal field
constexpr int kPropLevels = 5, kPropIn = 10, kPropHidden = 16;
struct PropArgs {
  GridDev grid;
  const float* origins;  // [N,3]
  const float* dirs;     // [N,3]
  const float* starts;   // [N,S]
  const float* ends;     // [N,S]
  int64_t n_points;
  int32_t S, contract;
  const float* table;
  const float* w1;       // [16][10]
  const float* w2;       // [16]
  float* density;        // [N,S]            (forward)
  const float* densbar;  // [N,S]            (backward)
  float* tablebar;       // accumulated      (backward)
  float* wpartial;       // [n_blocks][176]  (backward)
};

// position of point p in the grid's [0,1] domain: frustum MIDPOINT (rays.py:46-55) or explicit position, L-inf contraction, (x+2)/4
SDFHIP_D void prop_position(const PropArgs& a, const int64_t p, float pp[3]) {
  float x[3];
  if (a.dirs == nullptr) {  // explicit positions [P,3] (Field.density_fn, base_field.py:48-65)
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = a.origins[p * 3 + d];
  } else {
    const int64_t ray = p / a.S;
    const float t = 0.5f * (a.starts[p] + a.ends[p]);
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = a.origins[ray * 3 + d] + a.dirs[ray * 3 + d] * t;
  }
  if (a.contract) contract_inf(x, a.contract);
#pragma unroll
  for (int d = 0; d < 3; ++d) pp[d] = (x[d] + 2.0f) * 0.25f;
}

// grid features (5 levels x 2) and the 16 hidden pre-activations; returns the output pre-activation
// lofs: 0, possibly through an opaque register (prop_bwd_kernel): the level descriptors are then re-read from the kernel arguments where they
// are used instead of being hoisted out of the caller's grid-stride loop, where 25 of them sit in scalar registers for the whole kernel
SDFHIP_D float prop_mlp(const PropArgs& a, const float pp[3], float feat[kPropIn], float hid[kPropHidden], const int lofs = 0) {
  const float2* tab = reinterpret_cast<const float2*>(a.table);
#pragma unroll
  for (int l = 0; l < kPropLevels; ++l) {
    GridCell c;
    grid_cell(a.grid.lv[l + lofs], a.grid.smoothstep != 0, pp, c);
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tab[c.idx[k]];
    float y0 = 0.f, y1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = corner_w(c, k);
      y0 = fmaf(w, v[k].x, y0);
      y1 = fmaf(w, v[k].y, y1);
    }
    feat[2 * l] = y0;
    feat[2 * l + 1] = y1;
  }
  float pre = 0.0f;
#pragma unroll
  for (int j = 0; j < kPropHidden; ++j) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < kPropIn; ++i) s = fmaf(a.w1[j * kPropIn + i], feat[i], s);
    hid[j] = s;
    pre = fmaf(a.w2[j], fmaxf(s, 0.0f), pre);
  }
  return pre;
}

__global__ __launch_bounds__(256) void prop_fwd_kernel(const PropArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n_points) return;
  float pp[3], feat[kPropIn], hid[kPropHidden];
  prop_position(a, p, pp);
  const float pre = prop_mlp(a, pp, feat, hid);
  a.density[p] = expf(pre);
}

// Backward.  Thread <-> point (grid-stride, block-uniform trip count).  Table gradient: run-reduced, line-coalesced fp32 atomics.
// Weight gradients: w1_bar[j][i] = sum_p hb_j[p] feat_i[p] and w2_bar[j] = sum_p relu(hid_j[p]) pb[p] are 16 x 16 products
// contracted over the points, accumulated on the matrix core (v_mfma_f32_16x16x4_f32, 8 accumulator registers per
// lane instead of 176): each wave transposes its 64 points through a private LDS slab ([row][64 points + 4 pad]).
// Round 5: the per-level table scatter is a ROLLED loop (one copy of grid_cell + run reduction + staged scatter instead of five: 44.8 ->
// ~15 KB of code, no scratch where the unrolled form spilled 288 B per lane under its 256-register cap); d L / d feature of the level being
// scattered comes back from the slab (rows 27..36: written and read by the same lane).  Rows are 64 points + 4 pad floats: the b128
// operand reads of the MFMA stage (lane (m, k) reads row m) then spread over the banks (row stride 64 put all 16 rows of a k on the same
// four banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.10 for this kernel).
constexpr int kPropT = 68;     // LDS row stride (floats): 64 points + 4 pad, rows stay 16-byte aligned
constexpr int kPropRows = 37;  // rows 0..15: relu(hid_j)^T ; 16..25: feat_i^T ; 26: pb ; 27..36: d L / d feat_i ^T
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void prop_bwd_kernel(const PropArgs a) {
  __shared__ __attribute__((aligned(16))) float slab[4][kPropRows][kPropT];
  __shared__ float red[4][176];
  __shared__ ScatterStage stage[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float (*sh)[kPropT] = slab[wave];
  ScatterStage& st = stage[wave];
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  for (int64_t base = (int64_t)blockIdx.x * 256; base < a.n_points; base += (int64_t)gridDim.x * 256) {
    // compiler barrier: keeps the 176 (wave-uniform, scalar-cache resident) MLP weights from being hoisted out of the loop
    // into vector registers, which costs the kernel its occupancy
    asm volatile("" ::: "memory");
    const int64_t p = base + threadIdx.x;
    const bool live = p < a.n_points;
    const int64_t pc = live ? p : a.n_points - 1;
    const float db = live ? a.densbar[pc] : 0.0f;
    float pp[3];
    {
      float feat[kPropIn], hid[kPropHidden];
      prop_position(a, pc, pp);
      int zero;
      asm volatile("s_mov_b32 %0, 0" : "=s"(zero));  // opaque: see prop_mlp
      const float pre = prop_mlp(a, pp, feat, hid, zero);
      const float pb = db * expf(fminf(fmaxf(pre, -15.0f), 15.0f));  // activations.py:36-39
      float fb[kPropIn];
#pragma unroll
      for (int i = 0; i < kPropIn; ++i) fb[i] = 0.0f;
#pragma unroll
      for (int j = 0; j < kPropHidden; ++j) {
        const float hb = hid[j] > 0.0f ? pb * a.w2[j] : 0.0f;
        sh[j][lane] = fmaxf(hid[j], 0.0f);
#pragma unroll
        for (int i = 0; i < kPropIn; ++i) fb[i] = fmaf(a.w1[j * kPropIn + i], hb, fb[i]);
      }
#pragma unroll
      for (int i = 0; i < kPropIn; ++i) {
        sh[16 + i][lane] = feat[i];
        sh[27 + i][lane] = fb[i];
      }
      sh[26][lane] = pb;
    }
    // ---- table gradient, level by level
    const bool contributes = live && db != 0.0f;
#pragma unroll 1
    for (int l = 0; l < kPropLevels; ++l) {
      const float f0 = sh[27 + 2 * l][lane], f1 = sh[28 + 2 * l][lane];
      GridCell c;
      grid_cell(a.grid.lv[l], a.grid.smoothstep != 0, pp, c);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float w = corner_w(c, k);
        float t0 = w * f0, t1 = w * f1;
        const bool issue = wave_run_reduce(c.idx[k], contributes, t0, t1);
        scatter_stage_put(st, lane, k, issue, c.idx[k], t0, t1);
      }
      scatter_stage_flush(st, lane, a.tablebar);
    }
    // ---- weight gradients on the matrix core: lane (m = lane & 15, k = lane >> 4) takes points 16 u + 4 k + e
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's slab writes have landed (wave-private slab)
    __builtin_amdgcn_wave_barrier();
    // A rows: j = m.  B columns: n = m < 10 -> feat_n, n >= 10 -> pb (only column 10 of acc2 is used, columns >= 10 of acc1 ignored)
    const int m = lane & 15, kq = lane >> 4;
    const float w2m = a.w2[m];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 ar = *reinterpret_cast<const f32x4*>(&sh[m][16 * u + 4 * kq]);
      const f32x4 bf = *reinterpret_cast<const f32x4*>(&sh[16 + (m < 10 ? m : 10)][16 * u + 4 * kq]);
      const f32x4 pbv = *reinterpret_cast<const f32x4*>(&sh[26][16 * u + 4 * kq]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ah = ar[e] > 0.0f ? pbv[e] * w2m : 0.0f;                           // hb_j of this point
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, bf[e], acc1, 0, 0, 0);     // [j][i]  += hb_j feat_i
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[e], bf[e], acc2, 0, 0, 0);  // [j][10] += relu(hid_j) pb
      }
    }
    __builtin_amdgcn_wave_barrier();  // slab reads done before the next iteration overwrites it
  }
  // C layout of 16x16x4: col n = lane & 15, row m = 4 (lane >> 4) + reg
  {
    const int n = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * (lane >> 4) + r;
      if (n < kPropIn) red[wave][j * kPropIn + n] = acc1[r];
      if (n == 10) red[wave][160 + j] = acc2[r];
    }
  }
  __syncthreads();
  if (threadIdx.x < 176)
    a.wpartial[(size_t)blockIdx.x * 176 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Column sums of the per-block partials: out1[i] = sum_k partial[k][i] (i < n1), out2[i - n1] = sum_k partial[k][i] (i >= n1).
// Block = 32 columns x 32 row lanes (coalesced 128-byte row segments), rows strided by 32 (1024 rows: 32 dependent loads per thread, where
// 8 row lanes made 128), LDS sum over the row lanes.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ partial, const int n_rows, const int n_cols, const int row_stride,
                                                      float* __restrict__ out1, const int n1, float* __restrict__ out2) {
  __shared__ float red[32][33];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), r0 = threadIdx.x >> 5;
  float s = 0.0f;
  if (c < n_cols)
    for (int k = r0; k < n_rows; k += 32) s += partial[(size_t)k * row_stride + c];
  red[r0][threadIdx.x & 31] = s;
  __syncthreads();
  if (r0 == 0 && c < n_cols) {
#pragma unroll
    for (int r = 1; r < 32; ++r) s += red[r][threadIdx.x & 31];
    if (c < n1) out1[c] = s;
    else out2[c - n1] = s;
  }
}


// ---- numerical-gradient branch (sdf_field.py:431-453, 638-644): central differences of the six tap values
struct FdArgs {
  const float* sdf7;     // [7 P (padded)]  sdf of the points and their taps, tap-major (EncodeArgs::tap_points)
  int64_t n_points;      // P
  float delta;
  float* grad;           // [P][3]   0.5 (sdf(x + d e_k) - sdf(x - d e_k)) / d
  float* taps;           // [P][6]   sampled_sdf (sdf_field.py:644), or null
  // backward
  const float* sdf_bar;    // [P] or null
  const float* grad_bar;   // [P][3] or null   upstream d L / d gradient (eikonal loss, alpha)
  const float* csmallbar_tp;  // [T][nbs] colour backward (normal slots 30..32) or null
  const float* taps_bar;   // [P][6] or null   (curvature loss)
  int32_t nbs, pad_;
  int64_t n_padded7;
  float* sdfbar7;          // [n_padded7]
};
__global__ __launch_bounds__(256) void fd_normal_kernel(const FdArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_points) return;
  const int64_t P = a.n_points;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float tp = a.sdf7[(2 * d + 1) * P + i], tm = a.sdf7[(2 * d + 2) * P + i];
    a.grad[i * 3 + d] = 0.5f * (tp - tm) / a.delta;
    if (a.taps != nullptr) {
      a.taps[i * 6 + 2 * d] = tp;
      a.taps[i * 6 + 2 * d + 1] = tm;
    }
  }
}
// adjoint: sdfbar of the 7 P points from (sdf_bar, total d L / d gradient, taps_bar); rows beyond 7 P are zero
__global__ __launch_bounds__(256) void fd_adjoint_kernel(const FdArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n_padded7) return;
  const int64_t P = a.n_points;
  float v = 0.0f;
  if (p < 7 * P) {
    const int tap = (int)(p / P);
    const int64_t i = p - tap * P;
    if (tap == 0) {
      v = a.sdf_bar != nullptr ? a.sdf_bar[i] : 0.0f;
    } else {
      const int d = (tap - 1) >> 1;
      float g = a.grad_bar != nullptr ? a.grad_bar[i * 3 + d] : 0.0f;
      if (a.csmallbar_tp != nullptr) g += a.csmallbar_tp[tp_index(i, 30 + d, a.nbs)];
      v = ((tap - 1) & 1) ? -(0.5f * g / a.delta) : 0.5f * g / a.delta;
      if (a.taps_bar != nullptr) v += a.taps_bar[i * 6 + (tap - 1)];
    }
  }
  a.sdfbar7[p] = v;
}
