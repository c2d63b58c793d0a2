// TEST HARNESS (tests/test_cpu_marching_cubes.py builds it with g++; nothing in the product links or runs it).
// Compiles the per-cell / per-vertex functions of sdfstudio_amd/csrc_mesh/mc_cell.h for the host and runs them in the pass structure of
// mesh_api.hip - classify + compact, sort, count, exclusive scans, vertices (+ normals, values), faces - with a serial loop where the GPU
// has one thread per cell / per listed cell, so that the kernels' logic is checked against the oracle (and through it scikit-image) in a container without a GPU.
//   usage: mesh_host_check <in> <out>
//   in : int32 n0 n1 n2, float64 level, int32 has_mask, float32 volume[n0*n1*n2], uint8 mask[...] if has_mask
//   out: int64 V, int64 n_face_indices, float32 verts[V*3], int32 faces[...], float32 normals[V*3], float32 values[V]
#define MC_HOST_CHECK 1
#include "mc_cell.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
    const int cx = g.nx - 1, cy = g.ny - 1, cz = g.nz - 1;
    const int64_t ncells = (int64_t)cx * cy * cz;
    auto xyz = [&](unsigned c, int& x, int& y, int& z) {  // cell_xyz of mesh_api.hip
        const unsigned row = c / (unsigned)cx;
        x = (int)(c - row * (unsigned)cx);
        z = (int)(row / (unsigned)cy);
        y = (int)(row - (unsigned)z * (unsigned)cy);
    };
    // pass 0 (mc_classify_kernel): the list of surface cells - appended in REVERSE order here (the GPU's order is whatever the
    // wavefronts' atomics make it) - then sorted ascending, as the radix sort leaves it
    std::vector<unsigned> list;
    for (int64_t c = ncells - 1; c >= 0; --c) {
        int x, y, z;
        xyz((unsigned)c, x, y, z);
        if (mc_cell_nonempty(g, x, y, z)) list.push_back((unsigned)c);
    }
    std::sort(list.begin(), list.end());
    const size_t n = list.size();
    std::vector<unsigned> cnt_f(n), cnt_v(n), off_f(n), off_v(n);
    // pass 1 (mc_count_kernel) over the list
    for (size_t i = 0; i < n; ++i) {
        int x, y, z;
        xyz(list[i], x, y, z);
        mc_cell_count(g, x, y, z, cnt_f[i], cnt_v[i]);
        if (cnt_f[i] == 0) {
            fprintf(stderr, "a listed cell has no triangle\n");
            return 5;
        }
    }
    // the two exclusive scans
    uint64_t tf = 0, tv = 0;
    for (size_t i = 0; i < n; ++i) {
        off_f[i] = (unsigned)tf;
        off_v[i] = (unsigned)tv;
        tf += cnt_f[i];
        tv += cnt_v[i];
    }
    std::vector<float> verts(3 * tv), normals(3 * tv), values(tv);
    std::vector<int> faces(tf), idmap(4 * npoints, -123456789);  // the map is uninitialised on the GPU: poison it here
    // pass 2 (mc_vertices_kernel), in REVERSE list order: nothing may depend on the order threads run in
    for (size_t i = n; i-- > 0;) {
        if (cnt_v[i] == 0) continue;
        int x, y, z;
        xyz(list[i], x, y, z);
        mc_cell_vertices(g, x, y, z, off_v[i], verts.data(), normals.data(), values.data(), idmap.data());
    }
    // pass 3 (mc_faces_kernel), reverse order as well; flip = 1 (gradient_direction "descent")
    for (size_t i = n; i-- > 0;) {
        int x, y, z;
        xyz(list[i], x, y, z);
        mc_cell_faces(g, x, y, z, off_f[i], idmap.data(), faces.data(), 1);
    }
    for (uint64_t i = 0; i < tf; ++i)
        if (faces[i] < 0 || (uint64_t)faces[i] >= tv) {
            fprintf(stderr, "face index %llu reads an unwritten map entry\n", (unsigned long long)i);
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
