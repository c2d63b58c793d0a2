/* sdfmesh — C ABI of the mesh-extraction step of the SDF path (libsdfmesh.so, gfx950): marching cubes on a device-resident volume.
 *
 * SURVEY section 8 row f4 ("dense-grid SDF evaluation for mesh extraction").  libsdfhip.so evaluates the SDF on the lattice
 * (sdfstudio_amd/utils/marching_cubes.py: sdf_on_grid / evaluate_crop_pyramid); this library turns the volume into a triangle mesh
 * WHERE IT LIES, replacing
 *     skimage.measure.marching_cubes(volume, level, spacing, mask)        nerfstudio/utils/marching_cubes.py:125-134, :224-234, :357-366
 * (scikit-image==0.19.3, pyproject.toml:41; method "lewiner", step_size 1, allow_degenerate True - the defaults the reference uses),
 * which in the reference copies every 512^3 crop (512 MB) to the host and runs a serial Cython pass over it.
 * A separate library: the SDF library's sources - and with them the digest the profiles/ evidence is tied to - do not change.
 *
 * Conventions (as include/sdfhip.h): plain C, device pointers and sizes, the HIP stream to launch on, the caller owns every buffer,
 * 0 on success / negative on error with sdfmesh_last_error() (thread-local).  No CPU fallback: without a HIP device the calls fail.
 *
 * Results are scikit-image's, bit for bit and in its array order (oracle/marching_cubes.py is pinned on the real package; the kernels
 * follow it): vertices [V,3] float32 in lattice units and VOLUME AXIS ORDER, faces [F,3] int32, unit normals [V,3] float32, values [V]
 * float32 - i.e. the four arrays skimage.measure.marching_cubes returns for spacing (1, 1, 1).  The host multiplies by the spacing in
 * double exactly as scikit-image's wrapper does (sdfstudio_amd/utils/marching_cubes.py::marching_cubes).
 */
#ifndef SDFMESH_H_
#define SDFMESH_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sdfmesh_stream_t; /* hipStream_t */

int sdfmesh_version(void);
const char* sdfmesh_last_error(void);

/* Bytes of device workspace both calls below need for an [n0, n1, n2] volume.  Three bit arrays in a row-padded layout (one bit per
 * lattice point each, rows of n2 points padded to whole 64-bit words: volume > level, mask != 0, "surface cell"), one uint32 rank per
 * 64-bit word, and 20 B per entry of the surface-cell list (point index, triangle-table code, created-vertex record, block-local
 * offsets) - capacity: every cell of a volume of up to 2^20 cells, an eighth of the cells beyond that (a 512^3 SDF crop has < 1 % of its
 * cells on the surface).  ~3 B per lattice point for a large volume: 0.39 GB for the reference's 512^3 crop (0.73 x the volume; round 5
 * needed 3.1 GB for its per-lattice-point edge map).  Nothing in it needs initialising.  0 if the shape is refused (see _count). */
size_t sdfmesh_mc_workspace_bytes(int n0, int n1, int n2);

/* The streaming pass over the volume (one bit per lattice point comes out of it), the pass over those bits that finds the surface cells
 * IN scikit-image's traversal order (an ordered compaction: no sort), their classification and the scans (the Cython routine's
 * `for z: for y: for x:` loop of _marching_cubes_lewiner_cy.marching_cubes, without emitting anything).
 * volume: [n0, n1, n2] float32, row-major; mask: [n0, n1, n2] uint8 / bool or NULL (scikit-image's `mask`: cell (i, j, k) is processed
 * iff mask[i + 1, j + 1, k + 1]); level: the iso value (the caller has checked min <= level <= max, as scikit-image's wrapper does).
 * Writes the list, its per-cell tables and offsets into the workspace and the mesh size to the two HOST integers.  This call WAITS for
 * the stream, ONCE, at its end: the size of the result is data-dependent and the caller has to allocate it (scikit-image returns fresh
 * arrays, too).  Every launch before that is sized by the shape alone.
 * Errors: a dimension < 2, n0 * n1 * n2 >= 2^31, more surface cells than the list holds (white noise beyond 2^20 cells: mesh it in smaller
 * crops), more than 2^31 - 1 vertices or face indices, workspace too small, no device. */
int sdfmesh_mc_count(const float* volume, const unsigned char* mask, int n0, int n1, int n2, double level, void* workspace,
                     size_t workspace_bytes, int64_t* num_vertices, int64_t* num_faces, sdfmesh_stream_t stream);

/* The vertex and face passes over the list _count left in the workspace (same volume, mask, level): positions, triangles, normals, values.
 * verts [num_vertices, 3], faces [num_faces, 3], normals [num_vertices, 3], values [num_vertices]; normals / values may be NULL
 * (both or neither).  flip_faces: 1 = gradient_direction "descent" (scikit-image's default, what the reference gets), 0 = "ascent".
 * num_vertices / num_faces: what _count returned.  Does not synchronise. */
int sdfmesh_mc_emit(const float* volume, const unsigned char* mask, int n0, int n1, int n2, double level, void* workspace,
                    size_t workspace_bytes, int64_t num_vertices, int64_t num_faces, int flip_faces, float* verts, int32_t* faces,
                    float* normals, float* values, sdfmesh_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SDFMESH_H_ */
