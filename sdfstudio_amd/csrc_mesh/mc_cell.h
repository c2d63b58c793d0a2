// Marching cubes (Lewiner's 33-case tables with face / interior disambiguation), per-word, per-cell and per-vertex logic of the mesh kernels.
//
// What it replaces: skimage.measure.marching_cubes as nerfstudio/utils/marching_cubes.py:125-134 calls it on every 512^3 crop
// (scikit-image's _marching_cubes_lewiner_cy, a Cython port of Lewiner's MarchingCubes.cpp) - there a serial CPU pass over a volume that
// first crosses PCIe; here one streaming pass over the volume where the SDF kernels left it (one BIT per lattice point comes out of it),
// a pass over those bits, then passes over the surface cells only (mesh_api.hip).
// The arithmetic follows oracle/marching_cubes.py line by line (that file lists what was fitted to the scikit-image binary): corner
// values and every test in double, positions rounded to float once, normals accumulated in float in scikit-image's own order.
//
// Round 6 data flow (mesh_api.hip has the kernel list):
//   point bits   bit (row, x) = volume > level, rows padded to W = ceil(nx / 64) 64-bit words: word (row * W + x / 64), bit x % 64
//   cell bits    same layout, bit set iff the cell whose corner 0 is that point exists, is unmasked and straddles the level; the order of
//                the set bits IS scikit-image's traversal order (z outermost, x innermost), so the surface-cell list is an ordered
//                compaction - no sort - and  list index of a cell = wrank[word] + popcount(bits below it)  is an O(1) rank query
//   per listed cell   tile (which triangle table, computed ONCE: every later pass reads it instead of re-running the case tests),
//                     rec (13 nibbles: the local id of the vertex this cell creates on edge e, 15 = none), cnt (block-local offsets)
//   edge -> vertex id = offset of the CREATOR cell (rank query) + its rec nibble: no per-lattice-point map.
//
// This header is plain functions: the kernels of mesh_api.hip are thin wrappers (one thread per word / cell / vertex), and
// tests/mesh_host_check.cpp compiles the SAME functions with g++ to run the passes serially against the oracle.  MC_HOST_CHECK selects
// that build; it exists for the test harness only - the library has no host path.
// No array is indexed with a run-time index (no scratch): a corner chosen by a table is re-read from the volume (an L1 hit).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(MC_HOST_CHECK)
#define MC_FN inline
#define MC_TABLE static const
#else
#define MC_FN __device__ inline
#define MC_TABLE __device__ __constant__ static const
#endif

// no fused multiply-adds anywhere below: the sign tests and the roundings are scikit-image's only without contraction
#pragma clang fp contract(off)
#pragma STDC FP_CONTRACT OFF

#include "mc_tables.h"

#define MC_EPS 2.220446049250313e-16  // the port's "FLT_EPSILON" (oracle/marching_cubes.py)

// Lewiner's corner numbering -> (dx, dy, dz); x = the fastest (last) axis of the volume
MC_TABLE signed char MC_CORNER[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
MC_TABLE signed char MC_EDGE_ENDS[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
// edge -> lower lattice point (dx, dy, dz) and axis (0 = x, 1 = y, 2 = z)
MC_TABLE signed char MC_EDGE_LO[12][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {0, 0, 1},
                                          {0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}};
MC_TABLE signed char MC_EDGE_AXIS[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
// (axis, offset of the edge's lower point inside a cell along the two other axes [second][first]) -> the cell's edge id
//   axis x: [dz][dy]; axis y: [dz][dx]; axis z: [dy][dx]
MC_TABLE signed char MC_EDGE_OF[3][2][2] = {{{0, 2}, {4, 6}}, {{3, 1}, {7, 5}}, {{8, 9}, {11, 10}}};
// test_face: face (1..6) -> corners A, B, C, D
MC_TABLE signed char MC_FACE[6][4] = {{0, 4, 5, 1}, {1, 5, 6, 2}, {2, 6, 7, 3}, {3, 7, 4, 0}, {0, 3, 2, 1}, {4, 7, 6, 5}};
// test_interior, edge-anchored form: edge -> a, b (the anchor edge), then the parallel edges B, C, D as corner pairs
MC_TABLE signed char MC_PAR[12][8] = {{0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
                                      {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2}, {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0},
                                      {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};
// the port indexes its difference table with the bitwise corner index: corners 2 <-> 3 and 6 <-> 7 trade places
MC_TABLE signed char MC_SWAP[8] = {0, 1, 3, 2, 4, 5, 7, 6};

struct McGrid {
    const float* vol;           // [nz, ny, nx], x fastest
    const unsigned char* mask;  // [nz, ny, nx] or null: cell (z, y, x) is processed iff mask[z + 1, y + 1, x + 1]
    int nz, ny, nx;
    double level;
};

typedef unsigned long long mc_u64;

// The surface-cell index the count call builds in the workspace (see the header comment)
struct McIndex {
    const mc_u64* cellbits;   // [rows * W]
    const unsigned* wrank;    // [rows * W] number of listed cells before this word
    const unsigned* list;     // [n] point index of every listed cell's corner 0, ascending
    const unsigned* tile;     // [n] MC_TILE code
    const mc_u64* rec;        // [n] 13 nibbles
    const unsigned* cnt;      // [n] block-local exclusive offsets: face indices (low 16 bits), vertices (high 16)
    const unsigned* blockoff; // [ceil(n / 256)][2] exclusive offsets of every 256-cell block: face indices, vertices
    int W;
};

#define MC_SCAN_BLOCK 256   // cells per offset block (cnt / blockoff) = threads per workgroup (mesh_api.hip)
#define MC_WORD_BLOCK 256   // words per rank block (one word per thread)

MC_FN int mc_popc64(mc_u64 v) {
#if defined(MC_HOST_CHECK)
    return __builtin_popcountll(v);
#else
    return __popcll(v);
#endif
}

// The float threshold of the streaming pass: for a float v and a double level, ((double)v - level > 0.0) <=> (v > t) with t = level
// rounded DOWN to float (a double difference of two different numbers is never zero; if level is a float, t = level; otherwise no float
// lies strictly between t and level).  NaN and the infinities come out the same on both sides.  Host side (the API computes it once).
inline float mc_float_threshold(double level) {
    float t = (float)level;
    if ((double)t > level) t = nextafterf(t, -INFINITY);
    return t;
}

MC_FN unsigned mc_point(const McGrid& g, int x, int y, int z) { return ((unsigned)z * (unsigned)g.ny + (unsigned)y) * (unsigned)g.nx + (unsigned)x; }

MC_FN void mc_point_xyz(const McGrid& g, unsigned p, int& x, int& y, int& z) {
    const unsigned row = p / (unsigned)g.nx;
    x = (int)(p - row * (unsigned)g.nx);
    z = (int)(row / (unsigned)g.ny);
    y = (int)(row - (unsigned)z * (unsigned)g.ny);
}

MC_FN bool mc_cell_exists(const McGrid& g, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= g.nx - 1 || y >= g.ny - 1 || z >= g.nz - 1) return false;
    return g.mask == nullptr || g.mask[mc_point(g, x + 1, y + 1, z + 1)] != 0;
}

// corner p (Lewiner's numbering, MC_CORNER) of the cell whose corner 0 is point `base`, minus the level, in double
MC_FN double mc_c(const McGrid& g, unsigned base, int p) {
    const unsigned dy = (unsigned)(p >> 1) & 1u, dz = (unsigned)(p >> 2) & 1u, dx = ((unsigned)p ^ dy) & 1u;
    return (double)g.vol[base + dx + dy * (unsigned)g.nx + dz * (unsigned)g.nx * (unsigned)g.ny] - g.level;
}

// all eight corners (static indices only) and the case index (bit p set iff corner p > 0)
MC_FN int mc_load_cube(const McGrid& g, unsigned base, double v[8]) {
    int idx = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        v[p] = mc_c(g, base, p);
        if (v[p] > 0.0) idx |= 1 << p;
    }
    return idx;
}

// ---- pass over the point bits: the cell word of word g (K2; one thread per word).  P: point bits, M: mask bits or null.
MC_FN mc_u64 mc_cell_word(const mc_u64* P, const mc_u64* M, unsigned g, int W, int nx, int ny, int nz) {
    const unsigned row = g / (unsigned)W;
    const int wi = (int)(g - row * (unsigned)W);
    const unsigned z = row / (unsigned)ny;
    const unsigned y = row - z * (unsigned)ny;
    if ((int)y >= ny - 1 || (int)z >= nz - 1) return 0ull;
    const int nvalid = nx - 1 - wi * 64;  // cells of this word with x <= nx - 2
    if (nvalid <= 0) return 0ull;
    const unsigned r00 = g, r10 = g + (unsigned)W, r01 = g + (unsigned)ny * (unsigned)W, r11 = r01 + (unsigned)W;
    const mc_u64 a = P[r00], b = P[r10], c = P[r01], d = P[r11];
    const mc_u64 any = a | b | c | d, all = a & b & c & d;
    mc_u64 any1 = 0ull, all1 = 0ull, m1 = 0ull;
    const bool last = wi + 1 == W;
    if (!last) {
        const mc_u64 a1 = P[r00 + 1], b1 = P[r10 + 1], c1 = P[r01 + 1], d1 = P[r11 + 1];
        any1 = a1 | b1 | c1 | d1;
        all1 = a1 & b1 & c1 & d1;
        if (M) m1 = M[r11 + 1];
    }
    mc_u64 cell = (any | (any >> 1) | (any1 << 63)) & ~(all & ((all >> 1) | (all1 << 63)));
    if (nvalid < 64) cell &= (1ull << nvalid) - 1ull;
    if (M) cell &= (M[r11] >> 1) | (m1 << 63);  // mask[z + 1, y + 1, x + 1]
    return cell;
}

// list index of the LISTED cell (x, y, z) (its bit must be set: every existing cell around a crossing edge is listed)
MC_FN unsigned mc_list_index(const McIndex& ix, const McGrid& g, int x, int y, int z) {
    const unsigned w = ((unsigned)z * (unsigned)g.ny + (unsigned)y) * (unsigned)ix.W + ((unsigned)x >> 6);
    return ix.wrank[w] + (unsigned)mc_popc64(ix.cellbits[w] & ((1ull << (x & 63)) - 1ull));
}

MC_FN bool mc_test_face(const McGrid& g, unsigned base, int face) {
    const int f = (face < 0 ? -face : face) - 1;
    const double A = mc_c(g, base, MC_FACE[f][0]), B = mc_c(g, base, MC_FACE[f][1]), C = mc_c(g, base, MC_FACE[f][2]),
                 D = mc_c(g, base, MC_FACE[f][3]);
    const double q = A * C - B * D;
    if (fabs(q) < MC_EPS) return face >= 0;
    return (double)face * A * q >= 0.0;
}

// edge: the anchor edge of the edge-anchored form (cases 6, 7, 12, 13); unused for cases 4 and 10
MC_FN bool mc_test_interior(const McGrid& g, unsigned base, int mc_case, int edge, int s) {
    double t, At, Bt, Ct, Dt;
    if (mc_case == 4 || mc_case == 10) {
        double c[8];
        mc_load_cube(g, base, c);
        const double a = (c[4] - c[0]) * (c[6] - c[2]) - (c[7] - c[3]) * (c[5] - c[1]);
        const double b = c[2] * (c[4] - c[0]) + c[0] * (c[6] - c[2]) - c[1] * (c[7] - c[3]) - c[3] * (c[5] - c[1]);
        t = -b / (2.0 * a + MC_EPS);
        if (t < 0.0 || t > 1.0) return s > 0;
        At = c[0] + (c[4] - c[0]) * t;
        Bt = c[3] + (c[7] - c[3]) * t;
        Ct = c[2] + (c[6] - c[2]) * t;
        Dt = c[1] + (c[5] - c[1]) * t;
    } else {
        const signed char* P = MC_PAR[edge];
        const double c0 = mc_c(g, base, P[0]), c1 = mc_c(g, base, P[1]), c2 = mc_c(g, base, P[2]), c3 = mc_c(g, base, P[3]),
                     c4 = mc_c(g, base, P[4]), c5 = mc_c(g, base, P[5]), c6 = mc_c(g, base, P[6]), c7 = mc_c(g, base, P[7]);
        t = c0 / (c0 - c1 + MC_EPS);
        At = 0.0;
        Bt = c2 + (c3 - c2) * t;
        Ct = c4 + (c5 - c4) * t;
        Dt = c6 + (c7 - c6) * t;
    }
    int test = 0;
    if (At >= 0.0) test += 1;
    if (Bt >= 0.0) test += 2;
    if (Ct >= 0.0) test += 4;
    if (Dt >= 0.0) test += 8;
    switch (test) {
        case 5: return (At * Ct - Bt * Dt < MC_EPS) ? (s > 0) : false;
        case 10: return (At * Ct - Bt * Dt >= MC_EPS) ? (s > 0) : false;
        case 7: case 11: case 13: case 14: case 15: return s < 0;
        default: return s > 0;  // 0, 1, 2, 3, 4, 6, 8, 9, 12
    }
}

// ---- tile codes: (table id << 16) | (element offset into the table << 4) | triangles.  Computed once per listed cell (mc_cell_classify);
// the vertex and face passes decode it instead of repeating the disambiguation tests.
enum McTable {
    MC_T1, MC_T2, MC_T3_1, MC_T3_2, MC_T4_1, MC_T4_2, MC_T5, MC_T6_1_1, MC_T6_1_2, MC_T6_2, MC_T7_1, MC_T7_2, MC_T7_3, MC_T7_4_1, MC_T7_4_2,
    MC_T8, MC_T9, MC_T10_1_1, MC_T10_1_1_, MC_T10_1_2, MC_T10_2, MC_T10_2_, MC_T11, MC_T12_1_1, MC_T12_1_1_, MC_T12_1_2, MC_T12_2, MC_T12_2_,
    MC_T13_1, MC_T13_1_, MC_T13_2, MC_T13_2_, MC_T13_3, MC_T13_3_, MC_T13_4, MC_T13_5_1, MC_T13_5_2, MC_T14
};
#define MC_TILE(id, off, ntri) (((unsigned)(id) << 16) | ((unsigned)(off) << 4) | (unsigned)(ntri))

MC_FN const signed char* mc_tile_table(int id) {
    switch (id) {
        case MC_T1: return MC_TILING1;
        case MC_T2: return MC_TILING2;
        case MC_T3_1: return MC_TILING3_1;
        case MC_T3_2: return MC_TILING3_2;
        case MC_T4_1: return MC_TILING4_1;
        case MC_T4_2: return MC_TILING4_2;
        case MC_T5: return MC_TILING5;
        case MC_T6_1_1: return MC_TILING6_1_1;
        case MC_T6_1_2: return MC_TILING6_1_2;
        case MC_T6_2: return MC_TILING6_2;
        case MC_T7_1: return MC_TILING7_1;
        case MC_T7_2: return MC_TILING7_2;
        case MC_T7_3: return MC_TILING7_3;
        case MC_T7_4_1: return MC_TILING7_4_1;
        case MC_T7_4_2: return MC_TILING7_4_2;
        case MC_T8: return MC_TILING8;
        case MC_T9: return MC_TILING9;
        case MC_T10_1_1: return MC_TILING10_1_1;
        case MC_T10_1_1_: return MC_TILING10_1_1_;
        case MC_T10_1_2: return MC_TILING10_1_2;
        case MC_T10_2: return MC_TILING10_2;
        case MC_T10_2_: return MC_TILING10_2_;
        case MC_T11: return MC_TILING11;
        case MC_T12_1_1: return MC_TILING12_1_1;
        case MC_T12_1_1_: return MC_TILING12_1_1_;
        case MC_T12_1_2: return MC_TILING12_1_2;
        case MC_T12_2: return MC_TILING12_2;
        case MC_T12_2_: return MC_TILING12_2_;
        case MC_T13_1: return MC_TILING13_1;
        case MC_T13_1_: return MC_TILING13_1_;
        case MC_T13_2: return MC_TILING13_2;
        case MC_T13_2_: return MC_TILING13_2_;
        case MC_T13_3: return MC_TILING13_3;
        case MC_T13_3_: return MC_TILING13_3_;
        case MC_T13_4: return MC_TILING13_4;
        case MC_T13_5_1: return MC_TILING13_5_1;
        case MC_T13_5_2: return MC_TILING13_5_2;
        default: return MC_TILING14;
    }
}

// the triangle list of a tile code: `ntri` triangles of three edge ids each (0..11; 12 = the cell's centre vertex)
MC_FN const signed char* mc_tile_ptr(unsigned tile, int& ntri) {
    ntri = (int)(tile & 15u);
    return mc_tile_table((int)(tile >> 16)) + ((tile >> 4) & 0xfffu);
}

// MarchingCubes.cpp::process_cube: the tile code of the cell at `base` with case index idx; 0 for the empty cases.
MC_FN unsigned mc_tiling(const McGrid& g, unsigned base, int idx) {
    const int mc_case = MC_CASES[2 * idx], cfg = MC_CASES[2 * idx + 1];
    switch (mc_case) {
        case 1: return MC_TILE(MC_T1, cfg * 3, 1);
        case 2: return MC_TILE(MC_T2, cfg * 6, 2);
        case 3:
            if (mc_test_face(g, base, MC_TEST3[cfg])) return MC_TILE(MC_T3_2, cfg * 12, 4);
            return MC_TILE(MC_T3_1, cfg * 6, 2);
        case 4:
            if (mc_test_interior(g, base, 4, 0, MC_TEST4[cfg])) return MC_TILE(MC_T4_1, cfg * 6, 2);
            return MC_TILE(MC_T4_2, cfg * 18, 6);
        case 5: return MC_TILE(MC_T5, cfg * 9, 3);
        case 6:
            if (mc_test_face(g, base, MC_TEST6[cfg * 3])) return MC_TILE(MC_T6_2, cfg * 15, 5);
            if (mc_test_interior(g, base, 6, MC_TEST6[cfg * 3 + 2], MC_TEST6[cfg * 3 + 1])) return MC_TILE(MC_T6_1_1, cfg * 9, 3);
            return MC_TILE(MC_T6_1_2, cfg * 27, 9);
        case 7: {
            int sub = 0;
            if (mc_test_face(g, base, MC_TEST7[cfg * 5])) sub += 1;
            if (mc_test_face(g, base, MC_TEST7[cfg * 5 + 1])) sub += 2;
            if (mc_test_face(g, base, MC_TEST7[cfg * 5 + 2])) sub += 4;
            switch (sub) {
                case 0: return MC_TILE(MC_T7_1, cfg * 9, 3);
                case 1: return MC_TILE(MC_T7_2, (cfg * 3 + 0) * 15, 5);
                case 2: return MC_TILE(MC_T7_2, (cfg * 3 + 1) * 15, 5);
                case 3: return MC_TILE(MC_T7_3, (cfg * 3 + 0) * 27, 9);
                case 4: return MC_TILE(MC_T7_2, (cfg * 3 + 2) * 15, 5);
                case 5: return MC_TILE(MC_T7_3, (cfg * 3 + 1) * 27, 9);
                case 6: return MC_TILE(MC_T7_3, (cfg * 3 + 2) * 27, 9);
                default:
                    if (mc_test_interior(g, base, 7, MC_TEST7[cfg * 5 + 4], MC_TEST7[cfg * 5 + 3])) return MC_TILE(MC_T7_4_2, cfg * 27, 9);
                    return MC_TILE(MC_T7_4_1, cfg * 15, 5);
            }
        }
        case 8: return MC_TILE(MC_T8, cfg * 6, 2);
        case 9: return MC_TILE(MC_T9, cfg * 12, 4);
        case 10:
            if (mc_test_face(g, base, MC_TEST10[cfg * 3])) {
                if (mc_test_face(g, base, MC_TEST10[cfg * 3 + 1])) return MC_TILE(MC_T10_1_1_, cfg * 12, 4);
                return MC_TILE(MC_T10_2, cfg * 24, 8);
            }
            if (mc_test_face(g, base, MC_TEST10[cfg * 3 + 1])) return MC_TILE(MC_T10_2_, cfg * 24, 8);
            if (mc_test_interior(g, base, 10, 0, MC_TEST10[cfg * 3 + 2])) return MC_TILE(MC_T10_1_1, cfg * 12, 4);
            return MC_TILE(MC_T10_1_2, cfg * 24, 8);
        case 11: return MC_TILE(MC_T11, cfg * 12, 4);
        case 12:
            if (mc_test_face(g, base, MC_TEST12[cfg * 4])) {
                if (mc_test_face(g, base, MC_TEST12[cfg * 4 + 1])) return MC_TILE(MC_T12_1_1_, cfg * 12, 4);
                return MC_TILE(MC_T12_2, cfg * 24, 8);
            }
            if (mc_test_face(g, base, MC_TEST12[cfg * 4 + 1])) return MC_TILE(MC_T12_2_, cfg * 24, 8);
            if (mc_test_interior(g, base, 12, MC_TEST12[cfg * 4 + 3], MC_TEST12[cfg * 4 + 2])) return MC_TILE(MC_T12_1_1, cfg * 12, 4);
            return MC_TILE(MC_T12_1_2, cfg * 24, 8);
        case 13: {
            int sub = 0;
            for (int i = 0; i < 6; ++i)
                if (mc_test_face(g, base, MC_TEST13[cfg * 7 + i])) sub += 1 << i;
            const int sc = MC_SUBCONFIG13[sub];
            if (sc <= 0) return MC_TILE(MC_T13_1, cfg * 12, 4);  // -1: unreachable combinations of the face tests (as the oracle)
            if (sc <= 6) return MC_TILE(MC_T13_2, (cfg * 6 + sc - 1) * 18, 6);
            if (sc <= 18) return MC_TILE(MC_T13_3, (cfg * 12 + sc - 7) * 30, 10);
            if (sc <= 22) return MC_TILE(MC_T13_4, (cfg * 4 + sc - 19) * 36, 12);
            if (sc <= 26) {
                const int k = sc - 23;
                const int o51 = (cfg * 4 + k) * 18;
                if (mc_test_interior(g, base, 13, MC_TILING13_5_1[o51], MC_TEST13[cfg * 7 + 6])) return MC_TILE(MC_T13_5_1, o51, 6);
                return MC_TILE(MC_T13_5_2, (cfg * 4 + k) * 30, 10);
            }
            if (sc <= 38) return MC_TILE(MC_T13_3_, (cfg * 12 + sc - 27) * 30, 10);
            if (sc <= 44) return MC_TILE(MC_T13_2_, (cfg * 6 + sc - 39) * 18, 6);
            return MC_TILE(MC_T13_1_, cfg * 12, 4);  // sc == 45 (the table holds nothing else for reachable sign patterns)
        }
        case 14: return MC_TILE(MC_T14, cfg * 12, 4);
        default: return 0u;  // case 0
    }
}

// The first cell in scikit-image's traversal order (z outermost, x innermost) that touches edge e of cell (x, y, z): it creates the
// edge's vertex (every cell that has a crossing edge references it).  Returns that cell and the edge's id INSIDE it.
MC_FN int mc_edge_creator(const McGrid& g, int x, int y, int z, int e, int& cx, int& cy, int& cz) {
    const int axis = MC_EDGE_AXIS[e];
    const int px = x + MC_EDGE_LO[e][0], py = y + MC_EDGE_LO[e][1], pz = z + MC_EDGE_LO[e][2];
    for (int hi = -1; hi <= 0; ++hi)        // the slower-varying of the two other axes
        for (int lo = -1; lo <= 0; ++lo) {  // the faster-varying one
            cx = px; cy = py; cz = pz;
            if (axis == 0) { cz += hi; cy += lo; }
            else if (axis == 1) { cz += hi; cx += lo; }
            else { cy += hi; cx += lo; }
            if (mc_cell_exists(g, cx, cy, cz)) return MC_EDGE_OF[axis][-hi][-lo];
        }
    cx = x; cy = y; cz = z;
    return e;  // unreachable: (x, y, z) itself is among the candidates
}

MC_FN bool mc_is_creator(const McGrid& g, int x, int y, int z, int e) {
    int cx, cy, cz;
    mc_edge_creator(g, x, y, z, e, cx, cy, cz);
    return cx == x && cy == y && cz == z;
}

// ---- per listed cell, once (K3): its tile code, the local ids of the vertices it creates (rec: nibble e = id, 15 = not created here; ids
// in the order of first use in the triangle list, as scikit-image hands them out), face-index and vertex counts
MC_FN void mc_cell_classify(const McGrid& g, unsigned base, unsigned& tile, mc_u64& rec, unsigned& n_face_idx, unsigned& n_new) {
    int x, y, z;
    mc_point_xyz(g, base, x, y, z);
    int idx = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p)
        if (mc_c(g, base, p) > 0.0) idx |= 1 << p;
    tile = mc_tiling(g, base, idx);
    int ntri;
    const signed char* T = mc_tile_ptr(tile, ntri);
    rec = 0xfffffffffffffull;  // 13 nibbles of 15
    n_face_idx = 3u * (unsigned)ntri;
    n_new = 0;
    unsigned seen = 0;
    for (int i = 0; i < 3 * ntri; ++i) {
        const int e = T[i];
        if (seen >> e & 1u) continue;
        seen |= 1u << e;
        if (e == 12 || mc_is_creator(g, x, y, z, e)) {
            rec = (rec & ~(0xfull << (4 * e))) | ((mc_u64)n_new << (4 * e));
            ++n_new;
        }
    }
}

MC_FN unsigned mc_off_faces(const McIndex& ix, unsigned i) { return ix.blockoff[2 * (i / MC_SCAN_BLOCK)] + (ix.cnt[i] & 0xffffu); }
MC_FN unsigned mc_off_verts(const McIndex& ix, unsigned i) { return ix.blockoff[2 * (i / MC_SCAN_BLOCK) + 1] + (ix.cnt[i] >> 16); }

// ---- E1, per listed cell: park a key in every output slot the cell owns, so that the emit passes can run one thread per VERTEX (x 4) and
// one per FACE INDEX with coalesced writes: (16 * list index + edge) in word 0 of the position of every vertex it creates, its list index in
// every face-index slot of its triangles.  The thread that fills a slot reads the key there first.
MC_FN void mc_cell_keys(const McIndex& ix, unsigned i, unsigned* verts_words, unsigned* face_words) {
    const mc_u64 rec = ix.rec[i];
    const unsigned vbase = mc_off_verts(ix, i);
#pragma unroll
    for (int e = 0; e < 13; ++e) {
        const unsigned r = (unsigned)(rec >> (4 * e)) & 15u;
        if (r != 15u) verts_words[3 * (size_t)(vbase + r)] = 16u * i + (unsigned)e;
    }
    const unsigned fbase = mc_off_faces(ix, i);
    const int nidx = 3 * (int)(ix.tile[i] & 15u);
    for (int k = 0; k < nidx; ++k) face_words[(size_t)fbase + k] = i;
}

// the port's one-sided differences of a cell: corner p (Lewiner's numbering), component k (x, y, z) -> v[a] - v[b]
MC_TABLE signed char MC_VG[8][3][2] = {{{0, 1}, {0, 3}, {0, 4}}, {{0, 1}, {1, 2}, {1, 5}}, {{3, 2}, {1, 2}, {2, 6}}, {{3, 2}, {0, 3}, {3, 7}},
                                       {{4, 5}, {4, 7}, {0, 4}}, {{4, 5}, {5, 6}, {1, 5}}, {{7, 6}, {5, 6}, {2, 6}}, {{7, 6}, {4, 7}, {3, 7}}};

MC_FN double mc_vg(const McGrid& g, unsigned base, int p, int k) { return mc_c(g, base, MC_VG[p][k][0]) - mc_c(g, base, MC_VG[p][k][1]); }

// the same differences from a cube held in registers, for COMPILE-TIME p and k only (unrolled loops: the switch folds away)
MC_FN double mc_vg_reg(const double* v, int p, int k) {
    switch (p * 3 + k) {
        case 0: case 3: return v[0] - v[1];
        case 1: case 10: return v[0] - v[3];
        case 2: case 14: return v[0] - v[4];
        case 4: case 7: return v[1] - v[2];
        case 5: case 17: return v[1] - v[5];
        case 6: case 9: return v[3] - v[2];
        case 8: case 20: return v[2] - v[6];
        case 11: case 23: return v[3] - v[7];
        case 12: case 15: return v[4] - v[5];
        case 13: case 22: return v[4] - v[7];
        case 16: case 19: return v[5] - v[6];
        default: return v[7] - v[6];  // 18, 21
    }
}

// ---- E2, per vertex: position, normal and value - what scikit-image accumulates over the cells that touch the vertex, in its traversal
// order, once per OCCURRENCE of the vertex in each cell's triangle list.  The work is cut into pieces so that the kernel can give each of
// the <= 4 cells around an edge vertex to a lane of its own (mesh_api.hip) while the host harness runs them in a loop: the float additions
// happen in ONE order either way (mc_vertex_accumulate, neighbours q = 0 .. 3).
struct McVertex {  // decoded key
    unsigned i, base;
    int e, x, y, z;
};

MC_FN McVertex mc_vertex_decode(const McGrid& g, const McIndex& ix, unsigned key) {
    McVertex V;
    V.i = key >> 4;
    V.e = (int)(key & 15u);
    V.base = ix.list[V.i];
    mc_point_xyz(g, V.base, V.x, V.y, V.z);
    return V;
}

// position in volume axis order (axis 0, 1, 2 = z, y, x), as skimage.measure.marching_cubes returns it
MC_FN void mc_vertex_position(const McGrid& g, const McVertex& V, float out[3]) {
    double fx, fy, fz, ff;
    if (V.e == 12) {
        double v[8];
        mc_load_cube(g, V.base, v);
        fx = 0.0; fy = 0.0; fz = 0.0; ff = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const double w = 1.0 / (MC_EPS + fabs(v[p]));
            fx += (double)(((p ^ (p >> 1)) & 1)) * w;
            fy += (double)((p >> 1) & 1) * w;
            fz += (double)((p >> 2) & 1) * w;
            ff += w;
        }
    } else {
        const int a = MC_EDGE_ENDS[V.e][0], b = MC_EDGE_ENDS[V.e][1];
        const double w1 = 1.0 / (MC_EPS + fabs(mc_c(g, V.base, a))), w2 = 1.0 / (MC_EPS + fabs(mc_c(g, V.base, b)));
        ff = w1 + w2;
        fx = (double)MC_CORNER[a][0] * w1 + (double)MC_CORNER[b][0] * w2;
        fy = (double)MC_CORNER[a][1] * w1 + (double)MC_CORNER[b][1] * w2;
        fz = (double)MC_CORNER[a][2] * w1 + (double)MC_CORNER[b][2] * w2;
    }
    out[0] = (float)((double)V.z + fz / ff);
    out[1] = (float)((double)V.y + fy / ff);
    out[2] = (float)((double)V.x + fx / ff);
}

// What one of the (up to) four cells around the vertex contributes: `uses` additions of (ga, then gb) to the normal and its value spread.
// Edge vertex: neighbour q = 2 * (hi + 1) + (lo + 1) in scikit-image's traversal order; centre vertex: q = 0 is the cell itself (gb = 0:
// adding +0.0f changes nothing), q > 0 contributes nothing.  uses == 0: nothing (cell missing, or the edge not in its triangles).
MC_FN void mc_vertex_neighbour(const McGrid& g, const McIndex& ix, const McVertex& V, int q, int& uses, float ga[3], float gb[3], float& spread) {
    uses = 0;
    ga[0] = ga[1] = ga[2] = gb[0] = gb[1] = gb[2] = 0.f;
    spread = 0.f;
    double v[8];
    if (V.e == 12) {
        if (q != 0) return;
        mc_load_cube(g, V.base, v);
        int ntri;
        const signed char* T = mc_tile_ptr(ix.tile[V.i], ntri);
        double gy = 0.0, gz = 0.0;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const double w = 1.0 / (MC_EPS + fabs(v[p]));
            gy += w * mc_vg_reg(v, p, 1);
            gz += w * mc_vg_reg(v, p, 2);
        }
        for (int t = 0; t < 3 * ntri; ++t) uses += T[t] == 12;
        ga[0] = (float)gz;  // the port's centre gradient lands as (zg, yg, 0)
        ga[1] = (float)gy;
    } else {
        const int axis = MC_EDGE_AXIS[V.e];
        const int hi = (q >> 1) - 1, lo = (q & 1) - 1;
        int cx = V.x + MC_EDGE_LO[V.e][0], cy = V.y + MC_EDGE_LO[V.e][1], cz = V.z + MC_EDGE_LO[V.e][2];
        if (axis == 0) { cz += hi; cy += lo; }
        else if (axis == 1) { cz += hi; cx += lo; }
        else { cy += hi; cx += lo; }
        if (!mc_cell_exists(g, cx, cy, cz)) return;
        const int ce = MC_EDGE_OF[axis][-hi][-lo];
        const unsigned cbase = mc_point(g, cx, cy, cz);
        const unsigned j = (cx == V.x && cy == V.y && cz == V.z) ? V.i : mc_list_index(ix, g, cx, cy, cz);
        int ntri;
        const signed char* T = mc_tile_ptr(ix.tile[j], ntri);
        for (int t = 0; t < 3 * ntri; ++t) uses += T[t] == ce;
        if (uses == 0) return;
        const int ca = MC_EDGE_ENDS[ce][0], cb = MC_EDGE_ENDS[ce][1];
        const float s1 = (float)(1.0 / (MC_EPS + fabs(mc_c(g, cbase, ca)))), s2 = (float)(1.0 / (MC_EPS + fabs(mc_c(g, cbase, cb))));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ga[k] = (float)(mc_vg(g, cbase, MC_SWAP[ca], k) * (double)s1);
            gb[k] = (float)(mc_vg(g, cbase, MC_SWAP[cb], k) * (double)s2);
        }
        mc_load_cube(g, cbase, v);
    }
    double vmax = v[0], vmin = v[0];
#pragma unroll
    for (int p = 1; p < 8; ++p) { vmax = v[p] > vmax ? v[p] : vmax; vmin = v[p] < vmin ? v[p] : vmin; }
    spread = (float)(vmax - vmin);
}

MC_FN void mc_vertex_accumulate(float n[3], float& value, int uses, const float ga[3], const float gb[3], float spread) {
    for (int u = 0; u < uses; ++u) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { n[k] += ga[k]; n[k] += gb[k]; }
    }
    if (uses > 0) value = spread > value ? spread : value;
}

MC_FN void mc_vertex_normal(const float n[3], float out[3]) {
    const double nx = n[0], ny = n[1], nz = n[2];
    const double len = sqrt(nx * nx + ny * ny + nz * nz);
    const double d = len > 0.0 ? len : 1.0;
    out[0] = (float)(nz / d);
    out[1] = (float)(ny / d);
    out[2] = (float)(nx / d);
}

// the whole vertex on one thread (the host harness; the kernel spreads the neighbours over four lanes)
MC_FN void mc_vertex_emit(const McGrid& g, const McIndex& ix, unsigned id, float* verts, float* normals, float* values) {
    const McVertex V = mc_vertex_decode(g, ix, ((const unsigned*)verts)[3 * (size_t)id]);
    mc_vertex_position(g, V, verts + 3 * (size_t)id);
    if (!normals) return;
    float n[3] = {0.f, 0.f, 0.f}, value = 0.f;
    for (int q = 0; q < 4; ++q) {
        int uses;
        float ga[3], gb[3], spread;
        mc_vertex_neighbour(g, ix, V, q, uses, ga, gb, spread);
        mc_vertex_accumulate(n, value, uses, ga, gb, spread);
    }
    mc_vertex_normal(n, normals + 3 * (size_t)id);
    values[id] = value;
}

// ---- E3, per FACE-INDEX SLOT s (one thread each: coalesced writes): the vertex id that belongs there.  The slot holds its cell's list
// index (mc_cell_keys).  The id of the vertex on edge e = vertex offset of the cell that CREATED it + its rec nibble.  flip: reverse every
// triangle (gradient_direction "descent", the default, is the flipped orientation of the tables: _marching_cubes_lewiner.py)
MC_FN void mc_face_slot(const McGrid& g, const McIndex& ix, unsigned s, int* faces, int flip) {
    const unsigned i = (unsigned)faces[s];
    const unsigned r = s - mc_off_faces(ix, i);
    const unsigned t = r / 3u, k = r - 3u * t;
    int ntri;
    const signed char* T = mc_tile_ptr(ix.tile[i], ntri);
    const int e = T[3u * t + (flip ? 2u - k : k)];
    const unsigned own = (unsigned)(ix.rec[i] >> (4 * e)) & 15u;
    unsigned id;
    if (own != 15u) {
        id = mc_off_verts(ix, i) + own;
    } else {
        int x, y, z, cx, cy, cz;
        mc_point_xyz(g, ix.list[i], x, y, z);
        const int ce = mc_edge_creator(g, x, y, z, e, cx, cy, cz);
        const unsigned j = mc_list_index(ix, g, cx, cy, cz);
        id = mc_off_verts(ix, j) + ((unsigned)(ix.rec[j] >> (4 * ce)) & 15u);
    }
    faces[s] = (int)id;
}
