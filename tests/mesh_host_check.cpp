// TEST HARNESS (tests/test_cpu_marching_cubes.py builds it with g++; nothing in the product links or runs it).
// Compiles the per-word / per-cell / per-vertex functions of sdfstudio_amd/csrc_mesh/mc_cell.h for the host and runs them in the pass
// structure of mesh_api.hip - point bits, cell words + rank blocks, the scan of the block sums, the list, classification + block scan, the
// scan of the block offsets, vertex keys, vertices (+ normals, values), faces - with a serial loop where the GPU has one thread per word /
// listed cell / vertex, so that the kernels' logic is checked against the oracle (and through it scikit-image) in a container without a GPU.
//   usage: mesh_host_check <in> <out>
//   in : int32 n0 n1 n2, float64 level, int32 has_mask, float32 volume[n0*n1*n2], uint8 mask[...] if has_mask
//   out: int64 V, int64 n_face_indices, float32 verts[V*3], int32 faces[...], float32 normals[V*3], float32 values[V]
#define MC_HOST_CHECK 1
#include "mc_cell.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t dims[3], has_mask;
    double level;
    if (fread(dims, 4, 3, f) != 3 || fread(&level, 8, 1, f) != 1 || fread(&has_mask, 4, 1, f) != 1) return 4;
    const int64_t npoints = (int64_t)dims[0] * dims[1] * dims[2];
    std::vector<float> vol(npoints);
    std::vector<unsigned char> mask(has_mask ? npoints : 0);
    if ((int64_t)fread(vol.data(), 4, npoints, f) != npoints) return 4;
    if (has_mask && (int64_t)fread(mask.data(), 1, npoints, f) != npoints) return 4;
    fclose(f);

    McGrid g{vol.data(), has_mask ? mask.data() : nullptr, dims[0], dims[1], dims[2], level};
    const unsigned nx = (unsigned)g.nx, rows = (unsigned)g.nz * (unsigned)g.ny, W = (nx + 63u) / 64u, words = rows * W;
    // K1 (mc_pointbits_kernel): one bit per point by the FLOAT threshold; the padding bits of a row stay 0.  Checked on the spot against the
    // double comparison every later pass makes (mc_float_threshold's claim).
    const float t = mc_float_threshold(level);
    std::vector<mc_u64> P(words, 0ull), M(has_mask ? words : 0, 0ull), C(words);
    for (unsigned row = 0; row < rows; ++row)
        for (unsigned x = 0; x < nx; ++x) {
            const float v = vol[(size_t)row * nx + x];
            if (((double)v - level > 0.0) != (v > t)) {
                fprintf(stderr, "float threshold disagrees with the double comparison at value %a, level %a\n", (double)v, level);
                return 5;
            }
            if (v > t) P[row * W + x / 64] |= 1ull << (x % 64);
            if (has_mask && mask[(size_t)row * nx + x]) M[row * W + x / 64] |= 1ull << (x % 64);
        }
    // K2 (mc_cellbits_kernel): cell words, ranks inside blocks of MC_WORD_BLOCK words, block sums - words visited in REVERSE order
    // (nothing may depend on the order threads run in)
    const unsigned nbw = (words + MC_WORD_BLOCK - 1) / MC_WORD_BLOCK;
    std::vector<unsigned> wrank(words), wblock(nbw, 0);
    for (unsigned gw = words; gw-- > 0;) C[gw] = mc_cell_word(P.data(), has_mask ? M.data() : nullptr, gw, (int)W, g.nx, g.ny, g.nz);
    for (unsigned b = 0; b < nbw; ++b) {
        unsigned run = 0;
        for (unsigned gw = b * MC_WORD_BLOCK; gw < words && gw < (b + 1) * MC_WORD_BLOCK; ++gw) {
            wrank[gw] = run;
            run += (unsigned)mc_popc64(C[gw]);
        }
        wblock[b] = run;
    }
    // mc_scan_words_kernel: exclusive scan of the block sums; n
    unsigned n = 0;
    for (unsigned b = 0; b < nbw; ++b) {
        const unsigned s = wblock[b];
        wblock[b] = n;
        n += s;
    }
    // mc_list_kernel: global ranks and the list (reverse word order again)
    std::vector<unsigned> list(n);
    for (unsigned gw = words; gw-- > 0;) {
        unsigned pos = wblock[gw / MC_WORD_BLOCK] + wrank[gw];
        wrank[gw] = pos;
        const unsigned row = gw / W, p0 = row * nx + (gw - row * W) * 64u;
        for (mc_u64 c = C[gw]; c; c &= c - 1ull) list[pos++] = p0 + (unsigned)__builtin_ctzll(c);
    }
    for (unsigned i = 1; i < n; ++i)
        if (list[i] <= list[i - 1]) {
            fprintf(stderr, "the list is not ascending\n");
            return 5;
        }
    // the listed cells are exactly the cells that exist and straddle the level (the round-5 per-cell test, spelled out)
    {
        size_t k = 0;
        for (int z = 0; z < g.nz - 1; ++z)
            for (int y = 0; y < g.ny - 1; ++y)
                for (int x = 0; x < g.nx - 1; ++x) {
                    bool in = false, out = false;
                    double v[8];
                    const int idx = mc_load_cube(g, mc_point(g, x, y, z), v);
                    in = idx != 0;
                    out = idx != 255;
                    const bool want = mc_cell_exists(g, x, y, z) && in && out;
                    const bool have = k < n && list[k] == mc_point(g, x, y, z);
                    if (want != have) {
                        fprintf(stderr, "cell (%d, %d, %d): listed %d, expected %d\n", x, y, z, (int)have, (int)want);
                        return 5;
                    }
                    k += have;
                }
        if (k != n) return 5;
    }
    // K3 (mc_classify_kernel) + block scan, then mc_scan_cells_kernel
    const unsigned nbc = (n + MC_SCAN_BLOCK - 1) / MC_SCAN_BLOCK;
    std::vector<unsigned> tile(n), cnt(n), blockoff(2 * (size_t)nbc + 2, 0), nfs(n), nvs(n);
    std::vector<mc_u64> rec(n);
    for (unsigned i = n; i-- > 0;) {
        mc_cell_classify(g, list[i], tile[i], rec[i], nfs[i], nvs[i]);
        if (nfs[i] == 0) {
            fprintf(stderr, "a listed cell has no triangle\n");
            return 5;
        }
    }
    for (unsigned b = 0; b < nbc; ++b) {
        unsigned rf = 0, rv = 0;
        for (unsigned i = b * MC_SCAN_BLOCK; i < n && i < (b + 1) * MC_SCAN_BLOCK; ++i) {
            cnt[i] = rf | (rv << 16);
            rf += nfs[i];
            rv += nvs[i];
        }
        blockoff[2 * b] = rf;
        blockoff[2 * b + 1] = rv;
    }
    uint64_t tf = 0, tv = 0;
    for (unsigned b = 0; b < nbc; ++b) {
        const unsigned sf = blockoff[2 * b], sv = blockoff[2 * b + 1];
        blockoff[2 * b] = (unsigned)tf;
        blockoff[2 * b + 1] = (unsigned)tv;
        tf += sf;
        tv += sv;
    }
    McIndex ix{C.data(), wrank.data(), list.data(), tile.data(), rec.data(), cnt.data(), blockoff.data(), (int)W};
    std::vector<float> verts(3 * tv), normals(3 * tv), values(tv);
    std::vector<int> faces(tf, -123456789);
    {
        const unsigned poison = 0xfffffff7u;  // the output arrive uninitialised on the GPU
        for (size_t k = 0; k < verts.size(); ++k) memcpy(&verts[k], &poison, 4);
    }
    // E1 (mc_keys_kernel), E2 (mc_vertices_kernel: four lanes per vertex on the GPU, the same pieces in a loop here), E3 (mc_faces_kernel: one
    // thread per face-index slot; flip = 1: gradient_direction "descent") - all in reverse order
    for (unsigned i = n; i-- > 0;) mc_cell_keys(ix, i, (unsigned*)verts.data(), (unsigned*)faces.data());
    for (uint64_t s = 0; s < tf; ++s)
        if ((unsigned)faces[s] >= n) {
            fprintf(stderr, "face slot %llu has no key\n", (unsigned long long)s);
            return 5;
        }
    for (uint64_t id = 0; id < tv; ++id) {
        unsigned key;
        memcpy(&key, &verts[3 * id], 4);
        if ((key >> 4) >= n || (key & 15u) > 12u) {
            fprintf(stderr, "vertex %llu has no key\n", (unsigned long long)id);
            return 5;
        }
    }
    for (uint64_t id = tv; id-- > 0;) mc_vertex_emit(g, ix, (unsigned)id, verts.data(), normals.data(), values.data());
    for (uint64_t s = tf; s-- > 0;) mc_face_slot(g, ix, (unsigned)s, faces.data(), 1);
    for (uint64_t i = 0; i < tf; ++i)
        if (faces[i] < 0 || (uint64_t)faces[i] >= tv) {
            fprintf(stderr, "face index %llu was not written or is out of range\n", (unsigned long long)i);
            return 5;
        }
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 3;
    const int64_t hdr[2] = {(int64_t)tv, (int64_t)tf};
    fwrite(hdr, 8, 2, o);
    fwrite(verts.data(), 4, verts.size(), o);
    fwrite(faces.data(), 4, faces.size(), o);
    fwrite(normals.data(), 4, normals.size(), o);
    fwrite(values.data(), 4, values.size(), o);
    fclose(o);
    return 0;
}
