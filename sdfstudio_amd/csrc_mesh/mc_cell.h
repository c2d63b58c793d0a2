// Marching cubes (Lewiner's 33-case tables with face / interior disambiguation), per-cell and per-vertex logic of the mesh kernels.
//
// What it replaces: skimage.measure.marching_cubes as nerfstudio/utils/marching_cubes.py:125-134 calls it on every 512^3 crop
// (scikit-image's _marching_cubes_lewiner_cy, a Cython port of Lewiner's MarchingCubes.cpp) - there a serial CPU pass over a volume that
// first crosses PCIe; here one streaming pass over the volume where the SDF kernels left it, then passes over the surface cells only
// (mesh_api.hip).
// The arithmetic follows oracle/marching_cubes.py line by line (that file lists what was fitted to the scikit-image binary): corner
// values and every test in double, positions rounded to float once, normals accumulated in float in scikit-image's own order.
//
// This header is plain functions over a McGrid: the kernels of mesh_api.hip are one-line wrappers (one thread per cell / per vertex), and
// tests/mesh_host_check.cpp compiles the SAME functions with g++ to run the passes serially against the oracle.  MC_HOST_CHECK selects
// that build; it exists for the test harness only - the library has no host path.
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

MC_FN int64_t mc_point(const McGrid& g, int x, int y, int z) { return ((int64_t)z * g.ny + y) * g.nx + x; }

MC_FN bool mc_cell_exists(const McGrid& g, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= g.nx - 1 || y >= g.ny - 1 || z >= g.nz - 1) return false;
    return g.mask == nullptr || g.mask[mc_point(g, x + 1, y + 1, z + 1)] != 0;
}

// corner values minus the level, in double; returns the case index (bit p set iff corner p > 0)
MC_FN int mc_load_cube(const McGrid& g, int x, int y, int z, double v[8]) {
    int idx = 0;
    for (int p = 0; p < 8; ++p) {
        v[p] = (double)g.vol[mc_point(g, x + MC_CORNER[p][0], y + MC_CORNER[p][1], z + MC_CORNER[p][2])] - g.level;
        if (v[p] > 0.0) idx |= 1 << p;
    }
    return idx;
}

MC_FN bool mc_test_face(const double* c, int face) {
    const int f = (face < 0 ? -face : face) - 1;
    const double A = c[MC_FACE[f][0]], B = c[MC_FACE[f][1]], C = c[MC_FACE[f][2]], D = c[MC_FACE[f][3]];
    const double q = A * C - B * D;
    if (fabs(q) < MC_EPS) return face >= 0;
    return (double)face * A * q >= 0.0;
}

// edge: the anchor edge of the edge-anchored form (cases 6, 7, 12, 13); unused for cases 4 and 10
MC_FN bool mc_test_interior(const double* c, int mc_case, int edge, int s) {
    double t, At, Bt, Ct, Dt;
    if (mc_case == 4 || mc_case == 10) {
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
        t = c[P[0]] / (c[P[0]] - c[P[1]] + MC_EPS);
        At = 0.0;
        Bt = c[P[2]] + (c[P[3]] - c[P[2]]) * t;
        Ct = c[P[4]] + (c[P[5]] - c[P[4]]) * t;
        Dt = c[P[6]] + (c[P[7]] - c[P[6]]) * t;
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

// MarchingCubes.cpp::process_cube: the triangle list of the cell with corner values c and case index idx - `ntri` triangles of three
// edge ids each (0..11; 12 = the cell's centre vertex).  nullptr / 0 for the empty cases.
MC_FN const signed char* mc_tiling(const double* c, int idx, int& ntri) {
    const int mc_case = MC_CASES[2 * idx], cfg = MC_CASES[2 * idx + 1];
    switch (mc_case) {
        case 1: ntri = 1; return MC_TILING1 + cfg * 3;
        case 2: ntri = 2; return MC_TILING2 + cfg * 6;
        case 3:
            if (mc_test_face(c, MC_TEST3[cfg])) { ntri = 4; return MC_TILING3_2 + cfg * 12; }
            ntri = 2; return MC_TILING3_1 + cfg * 6;
        case 4:
            if (mc_test_interior(c, 4, 0, MC_TEST4[cfg])) { ntri = 2; return MC_TILING4_1 + cfg * 6; }
            ntri = 6; return MC_TILING4_2 + cfg * 18;
        case 5: ntri = 3; return MC_TILING5 + cfg * 9;
        case 6:
            if (mc_test_face(c, MC_TEST6[cfg * 3])) { ntri = 5; return MC_TILING6_2 + cfg * 15; }
            if (mc_test_interior(c, 6, MC_TEST6[cfg * 3 + 2], MC_TEST6[cfg * 3 + 1])) { ntri = 3; return MC_TILING6_1_1 + cfg * 9; }
            ntri = 9; return MC_TILING6_1_2 + cfg * 27;
        case 7: {
            int sub = 0;
            if (mc_test_face(c, MC_TEST7[cfg * 5])) sub += 1;
            if (mc_test_face(c, MC_TEST7[cfg * 5 + 1])) sub += 2;
            if (mc_test_face(c, MC_TEST7[cfg * 5 + 2])) sub += 4;
            switch (sub) {
                case 0: ntri = 3; return MC_TILING7_1 + cfg * 9;
                case 1: ntri = 5; return MC_TILING7_2 + (cfg * 3 + 0) * 15;
                case 2: ntri = 5; return MC_TILING7_2 + (cfg * 3 + 1) * 15;
                case 3: ntri = 9; return MC_TILING7_3 + (cfg * 3 + 0) * 27;
                case 4: ntri = 5; return MC_TILING7_2 + (cfg * 3 + 2) * 15;
                case 5: ntri = 9; return MC_TILING7_3 + (cfg * 3 + 1) * 27;
                case 6: ntri = 9; return MC_TILING7_3 + (cfg * 3 + 2) * 27;
                default:
                    if (mc_test_interior(c, 7, MC_TEST7[cfg * 5 + 4], MC_TEST7[cfg * 5 + 3])) { ntri = 9; return MC_TILING7_4_2 + cfg * 27; }
                    ntri = 5; return MC_TILING7_4_1 + cfg * 15;
            }
        }
        case 8: ntri = 2; return MC_TILING8 + cfg * 6;
        case 9: ntri = 4; return MC_TILING9 + cfg * 12;
        case 10:
            if (mc_test_face(c, MC_TEST10[cfg * 3])) {
                if (mc_test_face(c, MC_TEST10[cfg * 3 + 1])) { ntri = 4; return MC_TILING10_1_1_ + cfg * 12; }
                ntri = 8; return MC_TILING10_2 + cfg * 24;
            }
            if (mc_test_face(c, MC_TEST10[cfg * 3 + 1])) { ntri = 8; return MC_TILING10_2_ + cfg * 24; }
            if (mc_test_interior(c, 10, 0, MC_TEST10[cfg * 3 + 2])) { ntri = 4; return MC_TILING10_1_1 + cfg * 12; }
            ntri = 8; return MC_TILING10_1_2 + cfg * 24;
        case 11: ntri = 4; return MC_TILING11 + cfg * 12;
        case 12:
            if (mc_test_face(c, MC_TEST12[cfg * 4])) {
                if (mc_test_face(c, MC_TEST12[cfg * 4 + 1])) { ntri = 4; return MC_TILING12_1_1_ + cfg * 12; }
                ntri = 8; return MC_TILING12_2 + cfg * 24;
            }
            if (mc_test_face(c, MC_TEST12[cfg * 4 + 1])) { ntri = 8; return MC_TILING12_2_ + cfg * 24; }
            if (mc_test_interior(c, 12, MC_TEST12[cfg * 4 + 3], MC_TEST12[cfg * 4 + 2])) { ntri = 4; return MC_TILING12_1_1 + cfg * 12; }
            ntri = 8; return MC_TILING12_1_2 + cfg * 24;
        case 13: {
            int sub = 0;
            for (int i = 0; i < 6; ++i)
                if (mc_test_face(c, MC_TEST13[cfg * 7 + i])) sub += 1 << i;
            const int sc = MC_SUBCONFIG13[sub];
            if (sc <= 0) { ntri = 4; return MC_TILING13_1 + cfg * 12; }  // -1: unreachable combinations of the face tests (as the oracle)
            if (sc <= 6) { ntri = 6; return MC_TILING13_2 + (cfg * 6 + sc - 1) * 18; }
            if (sc <= 18) { ntri = 10; return MC_TILING13_3 + (cfg * 12 + sc - 7) * 30; }
            if (sc <= 22) { ntri = 12; return MC_TILING13_4 + (cfg * 4 + sc - 19) * 36; }
            if (sc <= 26) {
                const int k = sc - 23;
                const signed char* t51 = MC_TILING13_5_1 + (cfg * 4 + k) * 18;
                if (mc_test_interior(c, 13, t51[0], MC_TEST13[cfg * 7 + 6])) { ntri = 6; return t51; }
                ntri = 10; return MC_TILING13_5_2 + (cfg * 4 + k) * 30;
            }
            if (sc <= 38) { ntri = 10; return MC_TILING13_3_ + (cfg * 12 + sc - 27) * 30; }
            if (sc <= 44) { ntri = 6; return MC_TILING13_2_ + (cfg * 6 + sc - 39) * 18; }
            ntri = 4; return MC_TILING13_1_ + cfg * 12;  // sc == 45 (the table holds nothing else for reachable sign patterns)
        }
        case 14: ntri = 4; return MC_TILING14 + cfg * 12;
        default: ntri = 0; return nullptr;  // case 0
    }
}

// The first cell in scikit-image's traversal order (z outermost, x innermost) that touches edge e of cell (x, y, z): it creates the
// edge's vertex (every cell that has a crossing edge references it).  Returns true iff that cell is (x, y, z) itself.
MC_FN bool mc_is_creator(const McGrid& g, int x, int y, int z, int e) {
    const int axis = MC_EDGE_AXIS[e];
    const int px = x + MC_EDGE_LO[e][0], py = y + MC_EDGE_LO[e][1], pz = z + MC_EDGE_LO[e][2];
    for (int hi = -1; hi <= 0; ++hi)        // the slower-varying of the two other axes
        for (int lo = -1; lo <= 0; ++lo) {  // the faster-varying one
            int cx = px, cy = py, cz = pz;
            if (axis == 0) { cz += hi; cy += lo; }
            else if (axis == 1) { cz += hi; cx += lo; }
            else { cy += hi; cx += lo; }
            if (mc_cell_exists(g, cx, cy, cz)) return cx == x && cy == y && cz == z;
        }
    return false;  // unreachable: (x, y, z) itself is among the candidates
}

// pass 0 (the only pass over the whole volume): does the cell exist, is it unmasked and do its corners straddle the level
MC_FN bool mc_cell_nonempty(const McGrid& g, int x, int y, int z) {
    if (!mc_cell_exists(g, x, y, z)) return false;
    bool any_in = false, any_out = false;
    for (int p = 0; p < 8; ++p) {
        const double v = (double)g.vol[mc_point(g, x + MC_CORNER[p][0], y + MC_CORNER[p][1], z + MC_CORNER[p][2])] - g.level;
        if (v > 0.0) any_in = true;
        else any_out = true;
    }
    return any_in && any_out;
}

// pass 1: the number of face INDICES (3 per triangle) and of vertices this cell creates
MC_FN void mc_cell_count(const McGrid& g, int x, int y, int z, unsigned& n_face_idx, unsigned& n_new) {
    n_face_idx = 0;
    n_new = 0;
    if (!mc_cell_exists(g, x, y, z)) return;
    double v[8];
    const int idx = mc_load_cube(g, x, y, z, v);
    if (idx == 0 || idx == 255) return;
    int ntri;
    const signed char* T = mc_tiling(v, idx, ntri);
    n_face_idx = 3u * (unsigned)ntri;
    unsigned seen = 0;
    for (int i = 0; i < 3 * ntri; ++i) {
        const int e = T[i];
        if (seen >> e & 1u) continue;
        seen |= 1u << e;
        if (e == 12 || mc_is_creator(g, x, y, z, e)) ++n_new;
    }
}

// key of a vertex: 4 * (lower lattice point of its edge) + axis; axis 3 = the centre vertex of the cell whose corner 0 is that point
MC_FN int64_t mc_edge_key(const McGrid& g, int x, int y, int z, int e) {
    if (e == 12) return 4 * mc_point(g, x, y, z) + 3;
    return 4 * mc_point(g, x + MC_EDGE_LO[e][0], y + MC_EDGE_LO[e][1], z + MC_EDGE_LO[e][2]) + MC_EDGE_AXIS[e];
}

MC_FN void mc_edge_vertex(const double* v, int x, int y, int z, int e, float out_xyz[3]) {
    const int a = MC_EDGE_ENDS[e][0], b = MC_EDGE_ENDS[e][1];
    const double w1 = 1.0 / (MC_EPS + fabs(v[a])), w2 = 1.0 / (MC_EPS + fabs(v[b]));
    const double ff = w1 + w2;
    const double fx = (double)MC_CORNER[a][0] * w1 + (double)MC_CORNER[b][0] * w2;
    const double fy = (double)MC_CORNER[a][1] * w1 + (double)MC_CORNER[b][1] * w2;
    const double fz = (double)MC_CORNER[a][2] * w1 + (double)MC_CORNER[b][2] * w2;
    out_xyz[0] = (float)((double)x + fx / ff);
    out_xyz[1] = (float)((double)y + fy / ff);
    out_xyz[2] = (float)((double)z + fz / ff);
}

MC_FN void mc_centre_vertex(const double* v, int x, int y, int z, float out_xyz[3]) {
    double fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0;
    for (int p = 0; p < 8; ++p) {
        const double w = 1.0 / (MC_EPS + fabs(v[p]));
        fx += (double)MC_CORNER[p][0] * w;
        fy += (double)MC_CORNER[p][1] * w;
        fz += (double)MC_CORNER[p][2] * w;
        ff += w;
    }
    out_xyz[0] = (float)((double)x + fx / ff);
    out_xyz[1] = (float)((double)y + fy / ff);
    out_xyz[2] = (float)((double)z + fz / ff);
}

// pass 3: the cell's triangles, fbase = its offset into the flat face-index array.  flip: reverse every triangle
// (gradient_direction "descent", the default, is the flipped orientation of the tables: _marching_cubes_lewiner.py)
MC_FN void mc_cell_faces(const McGrid& g, int x, int y, int z, unsigned fbase, const int* idmap, int* faces, int flip) {
    if (!mc_cell_exists(g, x, y, z)) return;
    double v[8];
    const int idx = mc_load_cube(g, x, y, z, v);
    if (idx == 0 || idx == 255) return;
    int ntri;
    const signed char* T = mc_tiling(v, idx, ntri);
    for (int t = 0; t < ntri; ++t)
        for (int j = 0; j < 3; ++j) {
            const int id = idmap[mc_edge_key(g, x, y, z, T[3 * t + j])];
            faces[(int64_t)fbase + 3 * t + (flip ? 2 - j : j)] = id;
        }
}

// the port's one-sided differences of a cell, corner p (Lewiner's numbering), component k (x, y, z)
MC_FN double mc_vg(const double* v, int p, int k) {
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

// normal and value of the vertex with key `key`: what scikit-image accumulates over the cells that touch the vertex, in its
// traversal order, once per OCCURRENCE of the vertex in each cell's triangle list.  normals: [V, 3] in volume axis order.
MC_FN void mc_vertex_finish(const McGrid& g, int64_t key, float* normal_out, float* value_out) {
    const int axis = (int)(key & 3);
    const int64_t pt = key >> 2;
    const int px = (int)(pt % g.nx), py = (int)((pt / g.nx) % g.ny), pz = (int)(pt / ((int64_t)g.nx * g.ny));
    float n[3] = {0.f, 0.f, 0.f};
    float value = 0.f;
    double v[8];
    if (axis == 3) {
        const int idx = mc_load_cube(g, px, py, pz, v);
        int ntri;
        const signed char* T = mc_tiling(v, idx, ntri);
        double gy = 0.0, gz = 0.0;
        for (int p = 0; p < 8; ++p) {
            const double w = 1.0 / (MC_EPS + fabs(v[p]));
            gy += w * mc_vg(v, p, 1);
            gz += w * mc_vg(v, p, 2);
        }
        double vmax = v[0], vmin = v[0];
        for (int p = 1; p < 8; ++p) { vmax = v[p] > vmax ? v[p] : vmax; vmin = v[p] < vmin ? v[p] : vmin; }
        value = (float)(vmax - vmin);
        for (int i = 0; i < 3 * ntri; ++i)
            if (T[i] == 12) {  // the port's centre gradient lands as (zg, yg, 0)
                n[0] += (float)gz;
                n[1] += (float)gy;
            }
    } else {
        for (int hi = -1; hi <= 0; ++hi)
            for (int lo = -1; lo <= 0; ++lo) {
                int cx = px, cy = py, cz = pz;
                if (axis == 0) { cz += hi; cy += lo; }
                else if (axis == 1) { cz += hi; cx += lo; }
                else { cy += hi; cx += lo; }
                if (!mc_cell_exists(g, cx, cy, cz)) continue;
                const int e = MC_EDGE_OF[axis][-hi][-lo];
                const int idx = mc_load_cube(g, cx, cy, cz, v);
                int ntri;
                const signed char* T = mc_tiling(v, idx, ntri);
                const int a = MC_EDGE_ENDS[e][0], b = MC_EDGE_ENDS[e][1];
                const float s1 = (float)(1.0 / (MC_EPS + fabs(v[a]))), s2 = (float)(1.0 / (MC_EPS + fabs(v[b])));
                float ga[3], gb[3];
                for (int k = 0; k < 3; ++k) {
                    ga[k] = (float)(mc_vg(v, MC_SWAP[a], k) * (double)s1);
                    gb[k] = (float)(mc_vg(v, MC_SWAP[b], k) * (double)s2);
                }
                bool used = false;
                for (int i = 0; i < 3 * ntri; ++i)
                    if (T[i] == e) {
                        used = true;
                        for (int k = 0; k < 3; ++k) { n[k] += ga[k]; n[k] += gb[k]; }
                    }
                if (used) {
                    double vmax = v[0], vmin = v[0];
                    for (int p = 1; p < 8; ++p) { vmax = v[p] > vmax ? v[p] : vmax; vmin = v[p] < vmin ? v[p] : vmin; }
                    const float spread = (float)(vmax - vmin);
                    value = spread > value ? spread : value;
                }
            }
    }
    const double nx = n[0], ny = n[1], nz = n[2];
    const double len = sqrt(nx * nx + ny * ny + nz * nz);
    const double d = len > 0.0 ? len : 1.0;
    normal_out[0] = (float)(nz / d);
    normal_out[1] = (float)(ny / d);
    normal_out[2] = (float)(nx / d);
    *value_out = value;
}

// pass 2: the vertices this cell creates (ids vbase, vbase + 1, ... in the order of first use): positions, the key -> id map the face
// pass reads and - when normals is not null - normals and values.  verts / normals: [V, 3] in VOLUME AXIS ORDER (axis 0, 1, 2 =
// z, y, x), as skimage.measure.marching_cubes returns them.
MC_FN void mc_cell_vertices(const McGrid& g, int x, int y, int z, unsigned vbase, float* verts, float* normals, float* values, int* idmap) {
    if (!mc_cell_exists(g, x, y, z)) return;
    double v[8];
    const int idx = mc_load_cube(g, x, y, z, v);
    if (idx == 0 || idx == 255) return;
    int ntri;
    const signed char* T = mc_tiling(v, idx, ntri);
    unsigned seen = 0, id = vbase;
    for (int i = 0; i < 3 * ntri; ++i) {
        const int e = T[i];
        if (seen >> e & 1u) continue;
        seen |= 1u << e;
        if (!(e == 12 || mc_is_creator(g, x, y, z, e))) continue;
        float p[3];
        if (e == 12) mc_centre_vertex(v, x, y, z, p);
        else mc_edge_vertex(v, x, y, z, e, p);
        verts[3 * (int64_t)id + 0] = p[2];
        verts[3 * (int64_t)id + 1] = p[1];
        verts[3 * (int64_t)id + 2] = p[0];
        const int64_t key = mc_edge_key(g, x, y, z, e);
        idmap[key] = (int)id;
        if (normals) mc_vertex_finish(g, key, normals + 3 * (int64_t)id, values + id);
        ++id;
    }
}
