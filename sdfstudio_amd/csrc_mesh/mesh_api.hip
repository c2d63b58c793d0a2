// libsdfmesh.so: marching cubes on a device-resident volume (include/sdfmesh.h).  gfx950 only.
//
// Data flow (HBM-bound integer / table work: one thread per cell, x fastest so a wavefront reads four contiguous 256-byte rows):
//   mc_count_kernel      volume -> per cell: face-index count, created-vertex count            reads 4 B / point, writes 8 B / cell
//   hipcub ExclusiveSum  x 2   -> per cell: offset into the face array, first vertex id        16 B / cell
//   mc_vertices_kernel   non-empty cells: positions of the vertices the cell creates (+ their normals / values, gathered from the
//                        <= 4 cells around each edge in scikit-image's own accumulation order) and the edge -> id map
//   mc_faces_kernel      non-empty cells: triangles through the map
// Vertex ids and face offsets come from scans over the cells in scikit-image's traversal order, so the arrays come out in ITS order.
// All arithmetic is in mc_cell.h (shared with the host test harness); the kernels below only map threads to cells.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/sdfmesh.h"
#include "mc_cell.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define MESH_HIP(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(-5, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr int kBlock = 256;

struct Layout {  // carve of the caller's workspace
    int64_t ncells, npoints;
    size_t cnt_f, cnt_v, off_f, off_v, idmap, totals, scan_tmp, scan_tmp_bytes, total_bytes;
};

size_t align256(size_t n) { return (n + 255) / 256 * 256; }

bool make_layout(int n0, int n1, int n2, Layout& L) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return false;
    L.npoints = (int64_t)n0 * n1 * n2;
    if (L.npoints >= ((int64_t)1 << 31)) return false;
    L.ncells = (int64_t)(n0 - 1) * (n1 - 1) * (n2 - 1);
    size_t tmp = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, (const unsigned*)nullptr, (unsigned*)nullptr, (int)L.ncells, (hipStream_t)0) != hipSuccess)
        return false;
    L.scan_tmp_bytes = tmp;
    size_t off = 0;
    L.cnt_f = off; off += align256(sizeof(unsigned) * L.ncells);
    L.cnt_v = off; off += align256(sizeof(unsigned) * L.ncells);
    L.off_f = off; off += align256(sizeof(unsigned) * L.ncells);
    L.off_v = off; off += align256(sizeof(unsigned) * L.ncells);
    L.idmap = off; off += align256(sizeof(int) * 4 * (size_t)L.npoints);
    L.totals = off; off += 256;
    L.scan_tmp = off; off += align256(tmp);
    L.total_bytes = off;
    return true;
}

__device__ inline bool cell_of_thread(const McGrid& g, int64_t ncells, int& x, int& y, int& z, int64_t& c) {
    c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= ncells) return false;
    const int cx = g.nx - 1, cy = g.ny - 1;
    x = (int)(c % cx);
    y = (int)((c / cx) % cy);
    z = (int)(c / ((int64_t)cx * cy));
    return true;
}

__global__ __launch_bounds__(kBlock) void mc_count_kernel(McGrid g, int64_t ncells, unsigned* cnt_f, unsigned* cnt_v,
                                                          unsigned long long* totals) {
    int x, y, z;
    int64_t c;
    if (!cell_of_thread(g, ncells, x, y, z, c)) return;
    unsigned nf, nv;
    mc_cell_count(g, x, y, z, nf, nv);
    cnt_f[c] = nf;
    cnt_v[c] = nv;
    if (nf) {  // exact 64-bit totals beside the 32-bit scans: an overflow of those is detected, not wrapped
        atomicAdd(&totals[0], (unsigned long long)nf);
        atomicAdd(&totals[1], (unsigned long long)nv);
    }
}

__global__ __launch_bounds__(kBlock) void mc_vertices_kernel(McGrid g, int64_t ncells, const unsigned* cnt_v, const unsigned* off_v,
                                                             int* idmap, float* verts, float* normals, float* values) {
    int x, y, z;
    int64_t c;
    if (!cell_of_thread(g, ncells, x, y, z, c)) return;
    if (cnt_v[c] == 0) return;
    mc_cell_vertices(g, x, y, z, off_v[c], verts, normals, values, idmap);
}

__global__ __launch_bounds__(kBlock) void mc_faces_kernel(McGrid g, int64_t ncells, const unsigned* cnt_f, const unsigned* off_f,
                                                          const int* idmap, int* faces, int flip) {
    int x, y, z;
    int64_t c;
    if (!cell_of_thread(g, ncells, x, y, z, c)) return;
    if (cnt_f[c] == 0) return;
    mc_cell_faces(g, x, y, z, off_f[c], idmap, faces, flip);
}

int check_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(-2, "sdfmesh: no HIP device (the library has no CPU path)");
    return 0;
}

}  // namespace

extern "C" {

int sdfmesh_version(void) { return 100; }

const char* sdfmesh_last_error(void) { return g_err; }

size_t sdfmesh_mc_workspace_bytes(int n0, int n1, int n2) {
    Layout L;
    if (!make_layout(n0, n1, n2, L)) return 0;
    return L.total_bytes;
}

int sdfmesh_mc_count(const float* volume, const unsigned char* mask, int n0, int n1, int n2, double level, void* workspace,
                     size_t workspace_bytes, int64_t* num_vertices, int64_t* num_faces, sdfmesh_stream_t stream_) {
    if (int rc = check_device()) return rc;
    Layout L;
    if (!make_layout(n0, n1, n2, L))
        return fail(-1, "sdfmesh_mc_count: volume [%d, %d, %d] refused (every dimension >= 2, fewer than 2^31 lattice points)", n0, n1, n2);
    if (!volume || !workspace || !num_vertices || !num_faces) return fail(-1, "sdfmesh_mc_count: null argument");
    if (workspace_bytes < L.total_bytes)
        return fail(-3, "sdfmesh_mc_count: workspace of %zu bytes, %zu needed (sdfmesh_mc_workspace_bytes)", workspace_bytes, L.total_bytes);
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    unsigned* cnt_f = (unsigned*)(ws + L.cnt_f);
    unsigned* cnt_v = (unsigned*)(ws + L.cnt_v);
    unsigned* off_f = (unsigned*)(ws + L.off_f);
    unsigned* off_v = (unsigned*)(ws + L.off_v);
    unsigned long long* totals = (unsigned long long*)(ws + L.totals);
    McGrid g{volume, mask, n0, n1, n2, level};
    MESH_HIP(hipMemsetAsync(totals, 0, 2 * sizeof(unsigned long long), stream));
    const unsigned blocks = (unsigned)((L.ncells + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(mc_count_kernel, dim3(blocks), dim3(kBlock), 0, stream, g, L.ncells, cnt_f, cnt_v, totals);
    MESH_HIP(hipGetLastError());
    size_t tmp = L.scan_tmp_bytes;
    MESH_HIP(hipcub::DeviceScan::ExclusiveSum(ws + L.scan_tmp, tmp, (const unsigned*)cnt_f, off_f, (int)L.ncells, stream));
    tmp = L.scan_tmp_bytes;
    MESH_HIP(hipcub::DeviceScan::ExclusiveSum(ws + L.scan_tmp, tmp, (const unsigned*)cnt_v, off_v, (int)L.ncells, stream));
    unsigned long long host_totals[2] = {0, 0};
    MESH_HIP(hipMemcpyAsync(host_totals, totals, sizeof(host_totals), hipMemcpyDeviceToHost, stream));
    MESH_HIP(hipStreamSynchronize(stream));
    if (host_totals[0] > 0x7fffffffULL || host_totals[1] > 0x7fffffffULL)
        return fail(-4, "sdfmesh_mc_count: %llu face indices / %llu vertices do not fit 32-bit offsets (extract the mesh in smaller crops)",
                    host_totals[0], host_totals[1]);
    *num_faces = (int64_t)(host_totals[0] / 3);
    *num_vertices = (int64_t)host_totals[1];
    return 0;
}

int sdfmesh_mc_emit(const float* volume, const unsigned char* mask, int n0, int n1, int n2, double level, void* workspace,
                    size_t workspace_bytes, int64_t num_vertices, int64_t num_faces, int flip_faces, float* verts, int32_t* faces,
                    float* normals, float* values, sdfmesh_stream_t stream_) {
    if (int rc = check_device()) return rc;
    Layout L;
    if (!make_layout(n0, n1, n2, L))
        return fail(-1, "sdfmesh_mc_emit: volume [%d, %d, %d] refused (every dimension >= 2, fewer than 2^31 lattice points)", n0, n1, n2);
    if (!volume || !workspace) return fail(-1, "sdfmesh_mc_emit: null argument");
    if (workspace_bytes < L.total_bytes)
        return fail(-3, "sdfmesh_mc_emit: workspace of %zu bytes, %zu needed (sdfmesh_mc_workspace_bytes)", workspace_bytes, L.total_bytes);
    if ((normals == nullptr) != (values == nullptr)) return fail(-1, "sdfmesh_mc_emit: normals and values are both given or both NULL");
    if (num_vertices == 0 && num_faces == 0) return 0;
    if (!verts || !faces) return fail(-1, "sdfmesh_mc_emit: null output");
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    McGrid g{volume, mask, n0, n1, n2, level};
    const unsigned blocks = (unsigned)((L.ncells + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(mc_vertices_kernel, dim3(blocks), dim3(kBlock), 0, stream, g, L.ncells, (const unsigned*)(ws + L.cnt_v),
                       (const unsigned*)(ws + L.off_v), (int*)(ws + L.idmap), verts, normals, values);
    MESH_HIP(hipGetLastError());
    hipLaunchKernelGGL(mc_faces_kernel, dim3(blocks), dim3(kBlock), 0, stream, g, L.ncells, (const unsigned*)(ws + L.cnt_f),
                       (const unsigned*)(ws + L.off_f), (const int*)(ws + L.idmap), (int*)faces, flip_faces ? 1 : 0);
    MESH_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
