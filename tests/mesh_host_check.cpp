// TEST HARNESS (tests/test_cpu_marching_cubes.py builds it with g++; nothing in the product links or runs it).
// Compiles the per-cell / per-vertex functions of sdfstudio_amd/csrc_mesh/mc_cell.h for the host and runs them in the pass structure of
// mesh_api.hip - count, exclusive scans over the cells, vertices (+ normals, values), faces - with a serial loop where the GPU has one
// thread per cell, so that the kernels' logic is checked against the oracle (and through it scikit-image) in a container without a GPU.
//   usage: mesh_host_check <in> <out>
//   in : int32 n0 n1 n2, float64 level, int32 has_mask, float32 volume[n0*n1*n2], uint8 mask[...] if has_mask
//   out: int64 V, int64 n_face_indices, float32 verts[V*3], int32 faces[...], float32 normals[V*3], float32 values[V]
#define MC_HOST_CHECK 1
#include "mc_cell.h"

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
    std::vector<unsigned> cnt_f(ncells), cnt_v(ncells), off_f(ncells), off_v(ncells);
    // pass 1 (mc_count_kernel): the same linear cell index -> (x, y, z) as cell_of_thread
    for (int64_t c = 0; c < ncells; ++c) {
        const int x = (int)(c % cx), y = (int)((c / cx) % cy), z = (int)(c / ((int64_t)cx * cy));
        mc_cell_count(g, x, y, z, cnt_f[c], cnt_v[c]);
    }
    // the two exclusive scans
    uint64_t tf = 0, tv = 0;
    for (int64_t c = 0; c < ncells; ++c) {
        off_f[c] = (unsigned)tf;
        off_v[c] = (unsigned)tv;
        tf += cnt_f[c];
        tv += cnt_v[c];
    }
    std::vector<float> verts(3 * tv), normals(3 * tv), values(tv);
    std::vector<int> faces(tf), idmap(4 * npoints, -123456789);  // the map is uninitialised on the GPU: poison it here
    // pass 2 (mc_vertices_kernel), in REVERSE cell order: nothing may depend on the order threads run in
    for (int64_t c = ncells - 1; c >= 0; --c) {
        if (cnt_v[c] == 0) continue;
        const int x = (int)(c % cx), y = (int)((c / cx) % cy), z = (int)(c / ((int64_t)cx * cy));
        mc_cell_vertices(g, x, y, z, off_v[c], verts.data(), normals.data(), values.data(), idmap.data());
    }
    // pass 3 (mc_faces_kernel), reverse order as well; flip = 1 (gradient_direction "descent")
    for (int64_t c = ncells - 1; c >= 0; --c) {
        if (cnt_f[c] == 0) continue;
        const int x = (int)(c % cx), y = (int)((c / cx) % cy), z = (int)(c / ((int64_t)cx * cy));
        mc_cell_faces(g, x, y, z, off_f[c], idmap.data(), faces.data(), 1);
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
