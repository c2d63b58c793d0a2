// libsdfmesh.so: marching cubes on a device-resident volume (include/sdfmesh.h).  gfx950 only.
//
// Data flow.  HBM-bound integer / table work; the volume is read ONCE, everything after that touches the surface cells only:
//   mc_classify_kernel   all cells, one thread each, x fastest (a wavefront reads four contiguous 256-byte rows; blocks are dealt to
//                        the 8 XCDs in contiguous runs so that the rows two neighbouring blocks share are hit in ONE L2): appends
//                        the cells whose corners straddle the level to a list (one atomic per wavefront)       4 B / lattice point read
//   hipcub radix sort    the list, ascending = scikit-image's traversal order (z outermost, x innermost)      O(surface)
//   mc_count_kernel      per listed cell: face-index count, created-vertex count
//   hipcub ExclusiveSum  x 2: offset into the face array, first vertex id - so the arrays come out in scikit-image's ORDER
//   mc_vertices_kernel   per listed cell: the vertices it creates (+ normals / values, gathered from the <= 4 cells around each edge
//                        in scikit-image's own accumulation order) and the edge -> id map
//   mc_faces_kernel      per listed cell: its triangles through the map
// Round 5's first form ran count / vertices / faces with one thread per lattice cell (1 - 2 busy lanes per wavefront on the surface,
// two 134 M-element scans): 8.6 ms per 512^3 crop on an MI355X (profiles/r5_mesh_gpu_check_v1.jsonl); this form compacts first.
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

#define MESH_HIP(expr)                                                                                             \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) return fail(-5, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr int kBlock = 256;
constexpr int kXcds = 8;
constexpr unsigned kMaxListBlocks = 16384;  // grid of the per-listed-cell kernels (grid-stride over the list)

struct Counters {  // device-side, in the workspace
    unsigned n_listed;             // cells appended by mc_classify_kernel (may exceed the capacity: then the call fails)
    unsigned pad;
    unsigned long long n_face_idx;  // 3 x triangles
    unsigned long long n_verts;
};

struct Layout {  // carve of the caller's workspace
    int64_t ncells, npoints, cap;
    size_t list_a, list_b, cnt_f, cnt_v, off_f, off_v, idmap, counters, tmp, tmp_bytes, total_bytes;
};

size_t align256(size_t n) { return (n + 255) / 256 * 256; }

// Capacity of the surface-cell list: every cell of a small volume; a quarter of the cells of a large one (a 512^3 SDF crop has < 1 %
// of its cells on the surface; white noise has ~ 100 % and is refused beyond 2^20 cells with a message).
int64_t list_capacity(int64_t ncells) {
    const int64_t small = (int64_t)1 << 20;
    if (ncells <= small) return ncells;
    return ncells / 4 > small ? ncells / 4 : small;
}

bool make_layout(int n0, int n1, int n2, Layout& L) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return false;
    L.npoints = (int64_t)n0 * n1 * n2;
    if (L.npoints >= ((int64_t)1 << 31)) return false;
    L.ncells = (int64_t)(n0 - 1) * (n1 - 1) * (n2 - 1);
    L.cap = list_capacity(L.ncells);
    size_t t_scan = 0, t_sort = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, t_scan, (const unsigned*)nullptr, (unsigned*)nullptr, (int)L.cap, (hipStream_t)0) != hipSuccess)
        return false;
    if (hipcub::DeviceRadixSort::SortKeys(nullptr, t_sort, (const unsigned*)nullptr, (unsigned*)nullptr, (int)L.cap, 0, 32, (hipStream_t)0) != hipSuccess)
        return false;
    L.tmp_bytes = t_scan > t_sort ? t_scan : t_sort;
    size_t off = 0;
    const size_t per_list = align256(sizeof(unsigned) * (size_t)L.cap);
    L.list_a = off; off += per_list;
    L.list_b = off; off += per_list;
    L.cnt_f = off; off += per_list;
    L.cnt_v = off; off += per_list;
    L.off_f = off; off += per_list;
    L.off_v = off; off += per_list;
    L.idmap = off; off += align256(sizeof(int) * 4 * (size_t)L.npoints);
    L.counters = off; off += 256;
    L.tmp = off; off += align256(L.tmp_bytes);
    L.total_bytes = off;
    return true;
}

__device__ inline void cell_xyz(const McGrid& g, unsigned c, int& x, int& y, int& z) {
    const unsigned cx = (unsigned)(g.nx - 1), cy = (unsigned)(g.ny - 1);
    const unsigned row = c / cx;
    x = (int)(c - row * cx);
    z = (int)(row / cy);
    y = (int)(row - (unsigned)z * cy);
}

// blocks_per_xcd consecutive LOGICAL blocks (a contiguous slab of cells) go to one XCD: hardware deals block b to XCD b % 8
__global__ __launch_bounds__(kBlock) void mc_classify_kernel(McGrid g, unsigned ncells, unsigned nblocks, unsigned blocks_per_xcd,
                                                             unsigned cap, unsigned* list, Counters* counters) {
    const unsigned logical = (blockIdx.x % kXcds) * blocks_per_xcd + blockIdx.x / kXcds;
    const unsigned c = logical * kBlock + threadIdx.x;
    bool nonempty = false;
    if (logical < nblocks && c < ncells) {
        int x, y, z;
        cell_xyz(g, c, x, y, z);
        nonempty = mc_cell_nonempty(g, x, y, z);
    }
    // one atomic per wavefront: the leader reserves a run of the list for the wavefront's non-empty cells (no lane has left early)
    const unsigned long long m = __ballot(nonempty);
    if (m == 0ull) return;
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(&counters->n_listed, (unsigned)__popcll(m));
    base = __shfl(base, leader, 64);
    if (nonempty) {
        const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        if (pos < cap) list[pos] = c;
    }
}

__global__ __launch_bounds__(kBlock) void mc_count_kernel(McGrid g, const unsigned* list, const Counters* counters, unsigned* cnt_f,
                                                          unsigned* cnt_v) {
    const unsigned n = counters->n_listed;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int x, y, z;
        cell_xyz(g, list[i], x, y, z);
        unsigned nf, nv;
        mc_cell_count(g, x, y, z, nf, nv);
        cnt_f[i] = nf;
        cnt_v[i] = nv;
    }
}

__global__ void mc_totals_kernel(Counters* counters, const unsigned* cnt_f, const unsigned* cnt_v, const unsigned* off_f, const unsigned* off_v) {
    const unsigned n = counters->n_listed;
    counters->n_face_idx = n ? (unsigned long long)off_f[n - 1] + cnt_f[n - 1] : 0ull;
    counters->n_verts = n ? (unsigned long long)off_v[n - 1] + cnt_v[n - 1] : 0ull;
}

__global__ __launch_bounds__(kBlock) void mc_vertices_kernel(McGrid g, const unsigned* list, const Counters* counters, const unsigned* cnt_v,
                                                             const unsigned* off_v, int* idmap, float* verts, float* normals, float* values) {
    const unsigned n = counters->n_listed;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (cnt_v[i] == 0) continue;
        int x, y, z;
        cell_xyz(g, list[i], x, y, z);
        mc_cell_vertices(g, x, y, z, off_v[i], verts, normals, values, idmap);
    }
}

__global__ __launch_bounds__(kBlock) void mc_faces_kernel(McGrid g, const unsigned* list, const Counters* counters, const unsigned* off_f,
                                                          const int* idmap, int* faces, int flip) {
    const unsigned n = counters->n_listed;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        int x, y, z;
        cell_xyz(g, list[i], x, y, z);
        mc_cell_faces(g, x, y, z, off_f[i], idmap, faces, flip);
    }
}

int check_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(-2, "sdfmesh: no HIP device (the library has no CPU path)");
    return 0;
}

unsigned list_blocks(int64_t n) {
    const int64_t b = (n + kBlock - 1) / kBlock;
    return (unsigned)(b < 1 ? 1 : (b > kMaxListBlocks ? kMaxListBlocks : b));
}

}  // namespace

extern "C" {

int sdfmesh_version(void) { return 101; }

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
    unsigned* list_a = (unsigned*)(ws + L.list_a);
    unsigned* list_b = (unsigned*)(ws + L.list_b);
    unsigned* cnt_f = (unsigned*)(ws + L.cnt_f);
    unsigned* cnt_v = (unsigned*)(ws + L.cnt_v);
    unsigned* off_f = (unsigned*)(ws + L.off_f);
    unsigned* off_v = (unsigned*)(ws + L.off_v);
    Counters* counters = (Counters*)(ws + L.counters);
    McGrid g{volume, mask, n0, n1, n2, level};
    MESH_HIP(hipMemsetAsync(counters, 0, sizeof(Counters), stream));
    const unsigned nblocks = (unsigned)((L.ncells + kBlock - 1) / kBlock);
    const unsigned blocks_per_xcd = (nblocks + kXcds - 1) / kXcds;
    hipLaunchKernelGGL(mc_classify_kernel, dim3(blocks_per_xcd * kXcds), dim3(kBlock), 0, stream, g, (unsigned)L.ncells, nblocks,
                       blocks_per_xcd, (unsigned)L.cap, list_a, counters);
    MESH_HIP(hipGetLastError());
    Counters host;
    MESH_HIP(hipMemcpyAsync(&host, counters, sizeof(Counters), hipMemcpyDeviceToHost, stream));
    MESH_HIP(hipStreamSynchronize(stream));  // the list's length sizes the sort
    const unsigned n = host.n_listed;
    if ((int64_t)n > L.cap)
        return fail(-4, "sdfmesh_mc_count: %u of %lld cells cross the level, the list holds %lld (a volume this noisy is meshed in smaller crops)",
                    n, (long long)L.ncells, (long long)L.cap);
    if ((unsigned long long)n * 36ull > 0xffffffffull)
        return fail(-4, "sdfmesh_mc_count: %u surface cells can exceed 32-bit face offsets (mesh the volume in smaller crops)", n);
    *num_vertices = 0;
    *num_faces = 0;
    if (n == 0) return 0;
    size_t tmp = L.tmp_bytes;
    MESH_HIP(hipcub::DeviceRadixSort::SortKeys(ws + L.tmp, tmp, (const unsigned*)list_a, list_b, (int)n, 0, 32, stream));
    hipLaunchKernelGGL(mc_count_kernel, dim3(list_blocks(n)), dim3(kBlock), 0, stream, g, (const unsigned*)list_b, (const Counters*)counters,
                       cnt_f, cnt_v);
    MESH_HIP(hipGetLastError());
    tmp = L.tmp_bytes;
    MESH_HIP(hipcub::DeviceScan::ExclusiveSum(ws + L.tmp, tmp, (const unsigned*)cnt_f, off_f, (int)n, stream));
    tmp = L.tmp_bytes;
    MESH_HIP(hipcub::DeviceScan::ExclusiveSum(ws + L.tmp, tmp, (const unsigned*)cnt_v, off_v, (int)n, stream));
    hipLaunchKernelGGL(mc_totals_kernel, dim3(1), dim3(1), 0, stream, counters, (const unsigned*)cnt_f, (const unsigned*)cnt_v,
                       (const unsigned*)off_f, (const unsigned*)off_v);
    MESH_HIP(hipGetLastError());
    MESH_HIP(hipMemcpyAsync(&host, counters, sizeof(Counters), hipMemcpyDeviceToHost, stream));
    MESH_HIP(hipStreamSynchronize(stream));
    if (host.n_face_idx > 0x7fffffffULL || host.n_verts > 0x7fffffffULL)
        return fail(-4, "sdfmesh_mc_count: %llu face indices / %llu vertices do not fit 32-bit indices (mesh the volume in smaller crops)",
                    host.n_face_idx, host.n_verts);
    *num_faces = (int64_t)(host.n_face_idx / 3);
    *num_vertices = (int64_t)host.n_verts;
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
    const unsigned* list = (const unsigned*)(ws + L.list_b);
    const Counters* counters = (const Counters*)(ws + L.counters);
    // the list has at most one entry per three face indices and is what the counters in the workspace say; size the grids by the mesh
    const unsigned blocks = list_blocks(num_faces + 1);
    hipLaunchKernelGGL(mc_vertices_kernel, dim3(blocks), dim3(kBlock), 0, stream, g, list, counters, (const unsigned*)(ws + L.cnt_v),
                       (const unsigned*)(ws + L.off_v), (int*)(ws + L.idmap), verts, normals, values);
    MESH_HIP(hipGetLastError());
    hipLaunchKernelGGL(mc_faces_kernel, dim3(blocks), dim3(kBlock), 0, stream, g, list, counters, (const unsigned*)(ws + L.off_f),
                       (const int*)(ws + L.idmap), (int*)faces, flip_faces ? 1 : 0);
    MESH_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
