// libsdfmesh.so: marching cubes on a device-resident volume (include/sdfmesh.h).  gfx950 only.
//
// Data flow (round 6).  HBM-bound integer / table work; the volume is read ONCE, as a pure stream, and leaves one bit per lattice point;
// everything after that touches bit arrays (1 / 32 of the volume) and the surface cells only.  No sort, no atomics, no library: the set
// bits of the cell-bit array are in scikit-image's traversal order already (z outermost, x innermost), so the surface-cell list is an
// ORDERED stream compaction (popcount -> block scan -> single-block scan of the block sums -> write), and the same scan kernels give the
// face / vertex offsets - so the arrays come out in scikit-image's order.
//   mc_pointbits_kernel  all points: bit = volume > level (float threshold, mc_float_threshold).  A wavefront owns 16 words = 16 x 64
//                        consecutive points of a row: 16 coalesced 256-byte loads in flight per lane, 16 ballots, 128 B written.
//                                                                                                        4 B / lattice point read, 1 bit written
//   mc_cellbits_kernel   per 64-bit word: the straddle test of 64 cells from 8 (+2) words of point bits, mask bits, popcount, block scan
//   mc_scan_words_kernel single workgroup, one round: exclusive scan of the per-block sums; the number of surface cells
//   mc_list_kernel       per word: its global rank (kept: the rank structure of mc_list_index) and the list entries of its set bits
//   mc_classify_kernel   per listed cell, ONCE: triangle table (the case tests in double), created-vertex record, counts + block scan
//   mc_scan_cells_kernel single workgroup, one round: chunk offsets of faces and vertices, the totals    -> the call's ONE host read
//   mc_keys_kernel       per listed cell: a key into every vertex / face-index slot it owns, so that the two emit passes can run
//   mc_vertices_kernel   FOUR lanes per VERTEX (one per cell around its edge; combined in scikit-image's order through shuffles) and
//   mc_faces_kernel      one thread per FACE INDEX (coalesced writes)                                    (sdfmesh_mc_emit)
// Round 5's form (one thread per cell with two 32-bit divisions, atomics + hipCUB radix sort + two hipCUB scans, a 16 B / lattice point
// edge map, two host reads): 2.30 ms per 512^3 crop, mc_classify_kernel 0.91 - 1.08 ms = 0.06 of HBM rate (profiles/r6_mesh_before_*).
// All arithmetic is in mc_cell.h (shared with the host test harness); the kernels below only map threads to words / cells / vertices.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
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

constexpr int kBlock = MC_SCAN_BLOCK;  // 256: threads per workgroup = words per rank block = cells per offset block
constexpr int kWordsPerWave = 16;
constexpr int kScanThreads = 1024;
constexpr unsigned kMaxListBlocks = 4096;  // grid of the per-listed-cell kernels (grid-stride over the list)
constexpr unsigned kFlagListOverflow = 1u, kFlagIndexOverflow = 2u;

struct Counters {  // device-side, in the workspace; copied to the host once per count call
    unsigned n_listed;  // surface cells (may exceed the capacity: then the call fails)
    unsigned flags;
    unsigned long long n_face_idx;  // 3 x triangles
    unsigned long long n_verts;
};

struct Layout {  // carve of the caller's workspace
    int64_t ncells, npoints, cap;
    unsigned rows, W, words, nb_words, nb_cells;
    size_t pointbits, maskbits, cellbits, wrank, wblock, list, tile, rec, cnt, blockoff, counters, total_bytes;
};

size_t align256(size_t n) { return (n + 255) / 256 * 256; }

// Capacity of the surface-cell list: every cell of a small volume; an eighth of the cells of a large one (a 512^3 SDF crop has < 1 %
// of its cells on the surface; white noise has ~ 100 % and is refused beyond 2^20 cells with a message).
int64_t list_capacity(int64_t ncells) {
    const int64_t small = (int64_t)1 << 20;
    if (ncells <= small) return ncells;
    return ncells / 8 > small ? ncells / 8 : small;
}

bool make_layout(int n0, int n1, int n2, Layout& L) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return false;
    L.npoints = (int64_t)n0 * n1 * n2;
    if (L.npoints >= ((int64_t)1 << 31)) return false;
    L.ncells = (int64_t)(n0 - 1) * (n1 - 1) * (n2 - 1);
    L.cap = list_capacity(L.ncells);
    L.rows = (unsigned)n0 * (unsigned)n1;
    L.W = ((unsigned)n2 + 63u) / 64u;
    L.words = L.rows * L.W;  // <= npoints / 64 + rows < 2^31
    L.nb_words = (L.words + MC_WORD_BLOCK - 1) / MC_WORD_BLOCK;  // MC_WORD_BLOCK == kBlock: one word per thread
    L.nb_cells = (unsigned)((L.cap + kBlock - 1) / kBlock);
    size_t off = 0;
    const size_t per_word = align256(8 * (size_t)L.words), per_cell = align256(4 * (size_t)L.cap);
    L.pointbits = off; off += per_word;
    L.maskbits = off; off += per_word;
    L.cellbits = off; off += per_word;
    L.wrank = off; off += align256(4 * (size_t)L.words);
    L.wblock = off; off += align256(4 * (size_t)L.nb_words);
    L.list = off; off += per_cell;
    L.tile = off; off += per_cell;
    L.cnt = off; off += per_cell;
    L.rec = off; off += align256(8 * (size_t)L.cap);
    L.blockoff = off; off += align256(8 * (size_t)L.nb_cells);
    L.counters = off; off += 256;
    L.total_bytes = off;
    return true;
}

// ------------------------------------------------------------------------------------------------ scans (wavefront shuffles + LDS)
template <typename T>
__device__ inline T wave_inclusive_scan(T v) {
    const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan over the workgroup's threads (NW wavefronts); `total` = the workgroup's sum on every thread.  lds: NW elements.
template <typename T, int NW>
__device__ inline T block_exclusive_scan(T v, T& total, T* lds) {
    const T inc = wave_inclusive_scan(v);
    const int lane = (int)(threadIdx.x & 63u), wid = (int)(threadIdx.x >> 6);
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    T off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const T s = lds[w];
        if (w < wid) off += s;
        tot += s;
    }
    __syncthreads();  // lds may be reused by the caller's next round
    total = tot;
    return off + inc - v;
}

// ------------------------------------------------------------------------------------------------ the pass over the volume
// One wavefront = kWordsPerWave consecutive words of the row-padded bit layout; lane l of word (row, wi) reads point x = 64 wi + l.
__global__ __launch_bounds__(kBlock) void mc_pointbits_kernel(const float* __restrict__ vol, const unsigned char* __restrict__ mask, unsigned nx,
                                                              unsigned W, unsigned words, float t, mc_u64* __restrict__ pointbits,
                                                              mc_u64* __restrict__ maskbits) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)));
    const unsigned g0 = wave * kWordsPerWave;
    if (g0 >= words) return;
    const unsigned row0 = g0 / W, wi0 = g0 - row0 * W;
    float r[kWordsPerWave];
    {
        unsigned row = row0, wi = wi0;
#pragma unroll
        for (int k = 0; k < kWordsPerWave; ++k) {
            const unsigned x = wi * 64u + lane;
            r[k] = (g0 + k < words && x < nx) ? vol[(size_t)row * nx + x] : -INFINITY;  // -inf > t is false for every t
            if (++wi == W) { wi = 0; ++row; }
        }
    }
    mc_u64 mine = 0ull;
#pragma unroll
    for (int k = 0; k < kWordsPerWave; ++k) {
        const mc_u64 b = __ballot(r[k] > t);
        if (lane == (unsigned)k) mine = b;
    }
    if (lane < kWordsPerWave && g0 + lane < words) pointbits[g0 + lane] = mine;
    if (mask != nullptr) {
        unsigned char m[kWordsPerWave];
        unsigned row = row0, wi = wi0;
#pragma unroll
        for (int k = 0; k < kWordsPerWave; ++k) {
            const unsigned x = wi * 64u + lane;
            m[k] = (g0 + k < words && x < nx) ? mask[(size_t)row * nx + x] : (unsigned char)0;
            if (++wi == W) { wi = 0; ++row; }
        }
        mine = 0ull;
#pragma unroll
        for (int k = 0; k < kWordsPerWave; ++k) {
            const mc_u64 b = __ballot(m[k] != 0);
            if (lane == (unsigned)k) mine = b;
        }
        if (lane < kWordsPerWave && g0 + lane < words) maskbits[g0 + lane] = mine;
    }
}

// ------------------------------------------------------------------------------------------------ the passes over the bits
// One word per thread; the per-workgroup sums go to wblock (scanned by mc_scan_words_kernel).
// (Folding that scan into this kernel's LAST workgroup - a ticket - was tried twice this round and lost both times: with __threadfence()
// every workgroup pays a write-back of the whole L2 on gfx950, 351 us instead of 15; with agent-scope relaxed atomics and no fence the
// serial tail's sc1 loads cost more than the 9 us launch they save, 38 us: profiles/r6_mesh_fence_lesson.txt.)
__global__ __launch_bounds__(kBlock) void mc_cellbits_kernel(const mc_u64* __restrict__ P, const mc_u64* __restrict__ M, unsigned words, int W,
                                                             int nx, int ny, int nz, mc_u64* __restrict__ cellbits,
                                                             unsigned* __restrict__ wrank, unsigned* __restrict__ wblock) {
    __shared__ unsigned lds[kBlock / 64];
    const unsigned g = blockIdx.x * kBlock + threadIdx.x;
    const mc_u64 cell = g < words ? mc_cell_word(P, M, g, W, nx, ny, nz) : 0ull;
    unsigned total;
    const unsigned excl = block_exclusive_scan<unsigned, kBlock / 64>((unsigned)__popcll(cell), total, lds);
    if (g < words) {
        cellbits[g] = cell;
        wrank[g] = excl;
    }
    if (threadIdx.x == 0) wblock[blockIdx.x] = total;
}

// single workgroup: data[0 .. n) -> its exclusive scan, in place; the total -> counters.  Every thread owns ceil(n / 1024) CONSECUTIVE
// entries (sum them, ONE workgroup scan of the sums, write them back with a running prefix): one round whatever n is.
__global__ __launch_bounds__(kScanThreads) void mc_scan_words_kernel(unsigned* data, unsigned n, unsigned cap, Counters* counters) {
    __shared__ unsigned lds[kScanThreads / 64];
    const unsigned per = (n + kScanThreads - 1) / kScanThreads;
    const unsigned a = threadIdx.x * per, b = a + per < n ? a + per : n;
    unsigned mine = 0u;
    for (unsigned i = a; i < b; ++i) mine += data[i];
    unsigned total;
    unsigned run = block_exclusive_scan<unsigned, kScanThreads / 64>(mine, total, lds);
    for (unsigned i = a; i < b; ++i) {
        const unsigned v = data[i];
        data[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) {
        counters->n_listed = total;
        counters->flags = total > cap ? kFlagListOverflow : 0u;
        counters->n_face_idx = 0ull;
        counters->n_verts = 0ull;
    }
}

__global__ __launch_bounds__(kBlock) void mc_list_kernel(const mc_u64* __restrict__ cellbits, unsigned* __restrict__ wrank,
                                                         const unsigned* __restrict__ wblock, unsigned words, unsigned W, unsigned nx, unsigned cap,
                                                         unsigned* __restrict__ list) {
    const unsigned g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= words) return;
    unsigned pos = wblock[blockIdx.x] + wrank[g];
    wrank[g] = pos;  // from here on: the number of listed cells before this word (mc_list_index)
    mc_u64 c = cellbits[g];
    if (c == 0ull) return;
    const unsigned row = g / W;
    const unsigned p0 = row * nx + (g - row * W) * 64u;
    while (c != 0ull) {
        const unsigned b = (unsigned)__ffsll((long long)c) - 1u;
        c &= c - 1ull;
        if (pos < cap) list[pos] = p0 + b;
        ++pos;
    }
}

// ------------------------------------------------------------------------------------------------ the passes over the surface cells
// Persistent workgroups over chunks of 256 listed cells.
__global__ __launch_bounds__(kBlock) void mc_classify_kernel(McGrid g, const unsigned* __restrict__ list, const Counters* __restrict__ counters,
                                                             unsigned cap, unsigned* __restrict__ tile, mc_u64* __restrict__ rec,
                                                             unsigned* __restrict__ cnt, unsigned* __restrict__ blocksum) {
    __shared__ unsigned lds[kBlock / 64];
    const unsigned n = counters->n_listed < cap ? counters->n_listed : cap;
    for (unsigned chunk = blockIdx.x; (size_t)chunk * kBlock < n; chunk += gridDim.x) {
        const unsigned i = chunk * kBlock + threadIdx.x;
        unsigned packed = 0u;
        if (i < n) {
            unsigned t, nf, nv;
            mc_u64 r;
            mc_cell_classify(g, list[i], t, r, nf, nv);
            tile[i] = t;
            rec[i] = r;
            packed = nf | (nv << 16);  // <= 36 and <= 13 per cell: the halves of a 256-cell sum cannot carry into each other
        }
        unsigned total;
        const unsigned excl = block_exclusive_scan<unsigned, kBlock / 64>(packed, total, lds);
        if (i < n) cnt[i] = excl;
        if (threadIdx.x == 0) {
            blocksum[2 * chunk] = total & 0xffffu;
            blocksum[2 * chunk + 1] = total >> 16;
        }
    }
}

// single workgroup: the [chunks][2] sums (face indices, vertices) -> exclusive chunk offsets, in place; the totals -> counters: the numbers
// the call's one host read fetches.  Same one-round scheme as mc_scan_words_kernel.
// host_out (may be null): host-mapped pinned memory the totals are ALSO stored to - visible to the host once the stream has drained, so the
// call needs no device -> host copy behind this kernel (a pageable-destination hipMemcpyAsync: a blit kernel, a staging buffer and a host
// memcpy between the last kernel and the caller).
__global__ __launch_bounds__(kScanThreads) void mc_scan_cells_kernel(unsigned* blockoff, unsigned cap, Counters* counters, Counters* host_out) {
    __shared__ unsigned long long lds[kScanThreads / 64];
    const unsigned n = counters->n_listed < cap ? counters->n_listed : cap;
    const unsigned nb = (n + kBlock - 1) / kBlock;
    const unsigned per = (nb + kScanThreads - 1) / kScanThreads;
    const unsigned a = threadIdx.x * per, b = a + per < nb ? a + per : nb;
    unsigned long long mf = 0ull, mv = 0ull;
    for (unsigned i = a; i < b; ++i) {
        mf += blockoff[2 * i];
        mv += blockoff[2 * i + 1];
    }
    unsigned long long tf, tv;
    unsigned long long rf = block_exclusive_scan<unsigned long long, kScanThreads / 64>(mf, tf, lds);
    unsigned long long rv = block_exclusive_scan<unsigned long long, kScanThreads / 64>(mv, tv, lds);
    for (unsigned i = a; i < b; ++i) {
        const unsigned f = blockoff[2 * i], v = blockoff[2 * i + 1];
        blockoff[2 * i] = (unsigned)rf;  // wraps only beyond 2^32 face indices: flagged below, the call fails
        blockoff[2 * i + 1] = (unsigned)rv;
        rf += f;
        rv += v;
    }
    if (threadIdx.x == 0) {
        counters->n_face_idx = tf;
        counters->n_verts = tv;
        if (tf > 0x7fffffffull || tv > 0x7fffffffull) counters->flags |= kFlagIndexOverflow;
        if (host_out != nullptr) {
            host_out->n_listed = counters->n_listed;
            host_out->flags = counters->flags;
            host_out->n_face_idx = tf;
            host_out->n_verts = tv;
        }
    }
}

__global__ __launch_bounds__(kBlock) void mc_keys_kernel(McIndex ix, const Counters* __restrict__ counters, unsigned cap, unsigned* verts_words,
                                                        unsigned* face_words) {
    const unsigned n = counters->n_listed < cap ? counters->n_listed : cap;
    for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) mc_cell_keys(ix, i, verts_words, face_words);
}

// FOUR lanes per vertex: lane q of the quad evaluates the q-th of the (up to) four cells around the vertex's edge - the long part: a rank
// query, a table walk, double-precision reciprocals - and lane 0 adds the four results up in scikit-image's order (the float additions
// are not associative: same order as the serial mc_vertex_emit) and writes.  The key is read by all four lanes before lane 0 overwrites it.
__global__ __launch_bounds__(kBlock) void mc_vertices_kernel(McGrid g, McIndex ix, unsigned num_vertices, float* verts, float* normals, float* values) {
    const unsigned tid = blockIdx.x * kBlock + threadIdx.x;
    const unsigned id = tid >> 2;
    const int q = (int)(tid & 3u);
    const bool live = id < num_vertices;
    McVertex V = mc_vertex_decode(g, ix, live ? ((const unsigned*)verts)[3 * (size_t)id] : 0u);
    int uses = 0;
    float ga[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f}, spread = 0.f;
    if (live && normals != nullptr) mc_vertex_neighbour(g, ix, V, q, uses, ga, gb, spread);
    float pos[3] = {0.f, 0.f, 0.f};
    if (live && q == 0) mc_vertex_position(g, V, pos);
    float n[3] = {0.f, 0.f, 0.f}, value = 0.f;
    if (normals != nullptr) {
#pragma unroll
        for (int src = 0; src < 4; ++src) {  // every lane runs the shuffles (quads are whole: the grid is a multiple of 4 threads)
            const int u = __shfl(uses, src, 4);
            float a[3], b[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a[k] = __shfl(ga[k], src, 4);
                b[k] = __shfl(gb[k], src, 4);
            }
            const float sp = __shfl(spread, src, 4);
            mc_vertex_accumulate(n, value, u, a, b, sp);
        }
    }
    if (live && q == 0) {
        verts[3 * (size_t)id + 0] = pos[0];
        verts[3 * (size_t)id + 1] = pos[1];
        verts[3 * (size_t)id + 2] = pos[2];
        if (normals != nullptr) {
            mc_vertex_normal(n, normals + 3 * (size_t)id);
            values[id] = value;
        }
    }
}

__global__ __launch_bounds__(kBlock) void mc_faces_kernel(McGrid g, McIndex ix, unsigned num_face_idx, int* faces, int flip) {
    const unsigned s = blockIdx.x * kBlock + threadIdx.x;
    if (s < num_face_idx) mc_face_slot(g, ix, s, faces, flip);
}

// One Counters block of host-mapped pinned memory per calling thread (sdfmesh_mc_count is synchronous: a thread has one call in flight),
// allocated at the thread's first call and kept; portable across devices.  null: the allocation failed, the call copies as before.
Counters* host_counters() {
    thread_local Counters* buf = nullptr;
    thread_local bool tried = false;
    if (!tried) {
        tried = true;
        void* p = nullptr;
        if (getenv("SDFMESH_NO_HOST_MAPPED") == nullptr && hipHostMalloc(&p, 256, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess)
            buf = (Counters*)p;
        else
            (void)hipGetLastError();
    }
    return buf;
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

McIndex make_index(char* ws, const Layout& L) {
    McIndex ix;
    ix.cellbits = (const mc_u64*)(ws + L.cellbits);
    ix.wrank = (const unsigned*)(ws + L.wrank);
    ix.list = (const unsigned*)(ws + L.list);
    ix.tile = (const unsigned*)(ws + L.tile);
    ix.rec = (const mc_u64*)(ws + L.rec);
    ix.cnt = (const unsigned*)(ws + L.cnt);
    ix.blockoff = (const unsigned*)(ws + L.blockoff);
    ix.W = (int)L.W;
    return ix;
}

}  // namespace

extern "C" {

int sdfmesh_version(void) { return 102; }

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
    mc_u64* pointbits = (mc_u64*)(ws + L.pointbits);
    mc_u64* maskbits = mask ? (mc_u64*)(ws + L.maskbits) : nullptr;
    mc_u64* cellbits = (mc_u64*)(ws + L.cellbits);
    unsigned* wrank = (unsigned*)(ws + L.wrank);
    unsigned* wblock = (unsigned*)(ws + L.wblock);
    unsigned* list = (unsigned*)(ws + L.list);
    Counters* counters = (Counters*)(ws + L.counters);
    McGrid g{volume, mask, n0, n1, n2, level};
    const unsigned words_per_block = (kBlock / 64) * kWordsPerWave;
    hipLaunchKernelGGL(mc_pointbits_kernel, dim3((L.words + words_per_block - 1) / words_per_block), dim3(kBlock), 0, stream, volume, mask,
                       (unsigned)n2, L.W, L.words, mc_float_threshold(level), pointbits, maskbits);
    hipLaunchKernelGGL(mc_cellbits_kernel, dim3(L.nb_words), dim3(kBlock), 0, stream, (const mc_u64*)pointbits, (const mc_u64*)maskbits, L.words,
                       (int)L.W, n2, n1, n0, cellbits, wrank, wblock);
    hipLaunchKernelGGL(mc_scan_words_kernel, dim3(1), dim3(kScanThreads), 0, stream, wblock, L.nb_words, (unsigned)L.cap, counters);
    hipLaunchKernelGGL(mc_list_kernel, dim3(L.nb_words), dim3(kBlock), 0, stream, (const mc_u64*)cellbits, wrank, (const unsigned*)wblock, L.words,
                       L.W, (unsigned)n2, (unsigned)L.cap, list);
    hipLaunchKernelGGL(mc_classify_kernel, dim3(L.nb_cells < kMaxListBlocks ? L.nb_cells : kMaxListBlocks), dim3(kBlock), 0, stream, g,
                       (const unsigned*)list, (const Counters*)counters, (unsigned)L.cap, (unsigned*)(ws + L.tile), (mc_u64*)(ws + L.rec),
                       (unsigned*)(ws + L.cnt), (unsigned*)(ws + L.blockoff));
    Counters* mapped = host_counters();
    hipLaunchKernelGGL(mc_scan_cells_kernel, dim3(1), dim3(kScanThreads), 0, stream, (unsigned*)(ws + L.blockoff), (unsigned)L.cap, counters,
                       mapped);
    MESH_HIP(hipGetLastError());
    Counters host;
    if (mapped == nullptr) MESH_HIP(hipMemcpyAsync(&host, counters, sizeof(Counters), hipMemcpyDeviceToHost, stream));
    MESH_HIP(hipStreamSynchronize(stream));  // the ONE host read: the caller has to allocate the mesh
    if (mapped != nullptr) host = *mapped;   // written by mc_scan_cells_kernel through the host mapping; the kernel has completed
    *num_vertices = 0;
    *num_faces = 0;
    if (host.flags & kFlagListOverflow)
        return fail(-4, "sdfmesh_mc_count: %u of %lld cells cross the level, the list holds %lld (a volume this noisy is meshed in smaller crops)",
                    host.n_listed, (long long)L.ncells, (long long)L.cap);
    if (host.flags & kFlagIndexOverflow)
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
    if (num_vertices < 0 || num_faces < 0 || num_vertices > 0x1fffffffLL || 3 * num_faces > 0x7fffffffLL)
        return fail(-1, "sdfmesh_mc_emit: mesh size out of range");
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    McGrid g{volume, mask, n0, n1, n2, level};
    const McIndex ix = make_index(ws, L);
    const Counters* counters = (const Counters*)(ws + L.counters);
    // the list has at most one entry per triangle and is what the counters in the workspace say; size the key pass by the mesh
    hipLaunchKernelGGL(mc_keys_kernel, dim3(list_blocks(num_faces + 1)), dim3(kBlock), 0, stream, ix, counters, (unsigned)L.cap, (unsigned*)verts,
                       (unsigned*)faces);
    if (num_vertices > 0)
        hipLaunchKernelGGL(mc_vertices_kernel, dim3((unsigned)((4 * num_vertices + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, g, ix,
                           (unsigned)num_vertices, verts, normals, values);
    if (num_faces > 0)
        hipLaunchKernelGGL(mc_faces_kernel, dim3((unsigned)((3 * num_faces + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, g, ix,
                           (unsigned)(3 * num_faces), (int*)faces, flip_faces ? 1 : 0);
    MESH_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
