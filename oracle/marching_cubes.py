"""CPU restatement of the mesh-extraction step of SURVEY 8 row f4: ``skimage.measure.marching_cubes`` as the reference calls it.

TEST INFRASTRUCTURE: imported by tests/ only.  The product (sdfstudio_amd/utils/marching_cubes.py -> libsdfmesh.so) never touches it.

Reference call sites: nerfstudio/utils/marching_cubes.py:125-134 (get_surface_sliding: ``measure.marching_cubes(volume, level, spacing,
mask)`` per 512^3 crop), :224-234 (get_surface_occupancy), :357-366 (get_surface_sliding_with_contraction); exporter/tsdf_utils.py:123.
The algorithm lives in a third-party dependency, scikit-image (pinned ``scikit-image==0.19.3``, /root/reference/pyproject.toml:41),
default ``method="lewiner"``: Lewiner, Lopes, Vieira, Tavares, "Efficient implementation of Marching Cubes' cases with topological
guarantees", JGT 8(2) 2003, as ported to Cython in skimage/measure/_marching_cubes_lewiner_cy.pyx.  Only the compiled module of
scikit-image 0.18.3 exists in the build container (/opt/conda, python3.9; no .pyx source), so this file restates the PUBLISHED algorithm
(the structure of Lewiner's MarchingCubes.cpp::process_cube / test_face / test_interior over his LookUpTable.h) and every arithmetic
detail the port adds was fitted to that binary as a black box and is PINNED on it:

* tests/golden/mc_*.npz (tests/golden/make_golden_mc.py, run under /opt/conda/bin/python3.9) hold volumes and the real
  scikit-image outputs; tests/test_cpu_marching_cubes.py demands BIT-EXACT equality of vertices, faces (values AND order), normals and
  values, and re-runs the comparison live against scikit-image when that interpreter is present;
* per-cell triangulations were compared on 480 k random single cells at magnitudes 1e-5 .. 1e+1, and on 17 k cells of the ambiguous
  cases at 1e-17 .. 1e-8 where the port's epsilons bite (0 mismatches).

What the black box showed, beyond the paper (all reproduced here):
  - cube values are ``double(volume) - level``; a corner is inside iff its value is > 0;
  - the port's ``FLT_EPSILON`` is 2.22e-16 (the double epsilon): test_face has effectively no tie band, test_interior compares with it;
  - both interpolation parameters of test_interior have ``+ eps`` in the denominator;
  - test_interior returns False where Lewiner's falls through to ``s < 0`` (test == 5 / 10 and the product test fails) - visible in
    case 4 with a negative test value;
  - vertices: the two corner "strengths" 1 / (eps + |v|) weight the corner positions (double), rounded to float32 once; the centre
    vertex of the tunnel tilings is the strength-weighted mean of all eight corners;
  - normals: every OCCURRENCE of a vertex in a cell's triangle list adds strength * (one-sided differences of that cell) of both end
    corners, float32 accumulation of float32-rounded products with the strength itself rounded to float32; the difference table is
    built in Lewiner's corner numbering but indexed with the bitwise (dz, dy, dx) corner index, which swaps corners 2 <-> 3 and 6 <-> 7;
    the centre vertex's gradient lands as (zg, yg, 0) in (x, y, z); normalised in double;
  - values: max over the touching cells of float32(max(cube) - min(cube));
  - ``mask``: cell (z, y, x) is processed iff mask[z + 1, y + 1, x + 1];
  - the sub-cases 6.1.2, 7.4.2, 12.1.2 and 13.5.2 never occurred in 7 M random cells of those cases (nor in scikit-image's output; a
    vectorised search over 192 M case-6 cells with heavy-tailed magnitudes found no 6.1.2 either: with these tests a face that reads
    "separated" never meets an interior that reads "connected") - EXCEPT through exact ties: test_face returns `face >= 0` when
    |A C - B D| < eps, whatever the interior says.  Round 6 enumerated small-integer cells (tests/golden/find_mc_subcase_cells.py):
    6.1.2 and 7.4.2 occur, and tests/golden/mc_lewiner_subcases.npz PINS those two branches on the real scikit-image (40 cells each,
    bit-exact).  12.1.2 and 13.5.2 did not occur in 19 M enumerated tie cells of cases 12 and 13 either, and a penalty-minimising
    optimiser ends on the face-test boundary for every configuration: their branches follow the paper and remain unpinned.
Cell traversal is z (axis 0) outermost, x (axis 2) innermost; vertices are numbered by first use, so array ORDER is reproduced too.
"""
import os
from typing import Optional, Tuple

import numpy as np

_LUT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mc_luts.npz")
L = {k: v for k, v in np.load(_LUT_PATH).items()}  # tools/mint_mc_tables.py

EPS = 2.220446049250313e-16  # the port's "FLT_EPSILON"
# Lewiner's corner numbering -> (dx, dy, dz); x is the LAST numpy axis
CORNER = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
EDGE_ENDS = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
_FACE = {1: (0, 4, 5, 1), 2: (1, 5, 6, 2), 3: (2, 6, 7, 3), 4: (3, 7, 4, 0), 5: (0, 3, 2, 1), 6: (4, 7, 6, 5)}
# test_interior, edge-anchored form: edge -> (a, b, B, C, D): t from edge (a, b), the three parallel edges in Lewiner's order
_PAR = {0: (0, 1, (3, 2), (7, 6), (4, 5)), 1: (1, 2, (0, 3), (4, 7), (5, 6)), 2: (2, 3, (1, 0), (5, 4), (6, 7)),
        3: (3, 0, (2, 1), (6, 5), (7, 4)), 4: (4, 5, (7, 6), (3, 2), (0, 1)), 5: (5, 6, (4, 7), (0, 3), (1, 2)),
        6: (6, 7, (5, 4), (1, 0), (2, 3)), 7: (7, 4, (6, 5), (2, 1), (3, 0)), 8: (0, 4, (3, 7), (2, 6), (1, 5)),
        9: (1, 5, (0, 4), (3, 7), (2, 6)), 10: (2, 6, (1, 5), (0, 4), (3, 7)), 11: (3, 7, (2, 6), (1, 5), (0, 4))}
_SWAP = [0, 1, 3, 2, 4, 5, 7, 6]


def test_face(cube, face) -> bool:
    """MarchingCubes.cpp::test_face: sign of the bilinear saddle on one cube face (double arithmetic)."""
    a, b, c, d = _FACE[abs(int(face))]
    A, B, C, D = cube[a], cube[b], cube[c], cube[d]
    q = A * C - B * D
    if abs(q) < EPS:
        return face >= 0
    return face * A * q >= 0


def test_interior(cube, case: int, config: int, subconfig: int, s) -> bool:
    """MarchingCubes.cpp::test_interior (the port's deviations: module docstring)."""
    c = cube
    with np.errstate(all="ignore"):
        if case in (4, 10):
            a = (c[4] - c[0]) * (c[6] - c[2]) - (c[7] - c[3]) * (c[5] - c[1])
            b = c[2] * (c[4] - c[0]) + c[0] * (c[6] - c[2]) - c[1] * (c[7] - c[3]) - c[3] * (c[5] - c[1])
            t = np.float64(-b) / np.float64(2 * a + EPS)  # the port regularises the denominator (pinned at |v| ~ 1e-8)
            if t < 0 or t > 1:
                return s > 0
            At = c[0] + (c[4] - c[0]) * t
            Bt = c[3] + (c[7] - c[3]) * t
            Ct = c[2] + (c[6] - c[2]) * t
            Dt = c[1] + (c[5] - c[1]) * t
        else:
            edge = {6: lambda: L["TEST6"][config][2], 7: lambda: L["TEST7"][config][4], 12: lambda: L["TEST12"][config][3],
                    13: lambda: L["TILING13_5_1"][config][subconfig][0]}[case]()
            ea, eb, (b0, b1), (c0, c1), (d0, d1) = _PAR[int(edge)]
            t = np.float64(c[ea]) / np.float64(c[ea] - c[eb] + EPS)  # regularised like the other form (pinned at |v| ~ 1e-16)
            At = 0.0
            Bt = c[b0] + (c[b1] - c[b0]) * t
            Ct = c[c0] + (c[c1] - c[c0]) * t
            Dt = c[d0] + (c[d1] - c[d0]) * t
    test = (1 if At >= 0 else 0) + (2 if Bt >= 0 else 0) + (4 if Ct >= 0 else 0) + (8 if Dt >= 0 else 0)
    if test in (0, 1, 2, 3, 4, 6, 8, 9, 12):
        return s > 0
    if test == 5:
        return (s > 0) if (At * Ct - Bt * Dt < EPS) else False
    if test == 10:
        return (s > 0) if (At * Ct - Bt * Dt >= EPS) else False
    return s < 0  # 7, 11, 13, 14, 15


def cell_triangles(cube) -> Tuple[list, str]:
    """MarchingCubes.cpp::process_cube: the triangle list of one cell (edge ids 0..11, 12 = the centre vertex; 3 per triangle) and a
    tag naming the (sub)case.  cube: the 8 corner values minus the level, doubles, Lewiner's numbering."""
    idx = 0
    for p in range(8):
        if cube[p] > 0:
            idx |= 1 << p
    case, cfg = int(L["CASES"][idx][0]), int(L["CASES"][idx][1])
    tag = str(case)
    if case == 0:
        return [], "0"
    if case == 1:
        T = L["TILING1"][cfg]
    elif case == 2:
        T = L["TILING2"][cfg]
    elif case == 3:
        T, tag = (L["TILING3_2"][cfg], "3.2") if test_face(cube, L["TEST3"][cfg]) else (L["TILING3_1"][cfg], "3.1")
    elif case == 4:
        T, tag = (L["TILING4_1"][cfg], "4.1") if test_interior(cube, 4, cfg, 0, L["TEST4"][cfg]) else (L["TILING4_2"][cfg], "4.2")
    elif case == 5:
        T = L["TILING5"][cfg]
    elif case == 6:
        if test_face(cube, L["TEST6"][cfg][0]):
            T, tag = L["TILING6_2"][cfg], "6.2"
        elif test_interior(cube, 6, cfg, 0, L["TEST6"][cfg][1]):
            T, tag = L["TILING6_1_1"][cfg], "6.1.1"
        else:
            T, tag = L["TILING6_1_2"][cfg], "6.1.2"
    elif case == 7:
        sub = sum((1 << i) for i in range(3) if test_face(cube, L["TEST7"][cfg][i]))
        tag = "7.%d" % sub
        if sub == 0:
            T = L["TILING7_1"][cfg]
        elif sub in (1, 2, 4):
            T = L["TILING7_2"][cfg][{1: 0, 2: 1, 4: 2}[sub]]
        elif sub in (3, 5, 6):
            T = L["TILING7_3"][cfg][{3: 0, 5: 1, 6: 2}[sub]]
        elif test_interior(cube, 7, cfg, sub, L["TEST7"][cfg][3]):
            T, tag = L["TILING7_4_2"][cfg], "7.4.2"
        else:
            T, tag = L["TILING7_4_1"][cfg], "7.4.1"
    elif case == 8:
        T = L["TILING8"][cfg]
    elif case == 9:
        T = L["TILING9"][cfg]
    elif case in (10, 12):
        n = "10" if case == 10 else "12"
        tst = L["TEST" + n][cfg]
        if test_face(cube, tst[0]):
            T, tag = (L["TILING%s_1_1_" % n][cfg], n + ".1.1_") if test_face(cube, tst[1]) else (L["TILING%s_2" % n][cfg], n + ".2")
        elif test_face(cube, tst[1]):
            T, tag = L["TILING%s_2_" % n][cfg], n + ".2_"
        elif test_interior(cube, case, cfg, 0, tst[2]):
            T, tag = L["TILING%s_1_1" % n][cfg], n + ".1.1"
        else:
            T, tag = L["TILING%s_1_2" % n][cfg], n + ".1.2"
    elif case == 11:
        T = L["TILING11"][cfg]
    elif case == 13:
        sub = sum((1 << i) for i in range(6) if test_face(cube, L["TEST13"][cfg][i]))
        sc = int(L["SUBCONFIG13"][sub])
        tag = "13.%d" % sc
        if sc <= 0:  # -1: a combination of face tests no trilinear cell produces (never seen; Lewiner prints "impossible case"): 13.1
            T = L["TILING13_1"][cfg]
        elif sc <= 6:
            T = L["TILING13_2"][cfg][sc - 1]
        elif sc <= 18:
            T = L["TILING13_3"][cfg][sc - 7]
        elif sc <= 22:
            T = L["TILING13_4"][cfg][sc - 19]
        elif sc <= 26:
            k = sc - 23
            if test_interior(cube, 13, cfg, k, L["TEST13"][cfg][6]):
                T, tag = L["TILING13_5_1"][cfg][k], tag + ":5.1"
            else:
                T, tag = L["TILING13_5_2"][cfg][k], tag + ":5.2"
        elif sc <= 38:
            T = L["TILING13_3_"][cfg][sc - 27]
        elif sc <= 44:
            T = L["TILING13_2_"][cfg][sc - 39]
        elif sc == 45:
            T = L["TILING13_1_"][cfg]
        else:
            raise AssertionError("marching cubes: impossible case 13")
    else:  # 14
        T = L["TILING14"][cfg]
    return [int(e) for e in T], tag


def _vg_table(v):
    """The port's one-sided differences per corner (x, y, z), Lewiner's numbering."""
    v0, v1, v2, v3, v4, v5, v6, v7 = v
    return [(v0 - v1, v0 - v3, v0 - v4), (v0 - v1, v1 - v2, v1 - v5), (v3 - v2, v1 - v2, v2 - v6), (v3 - v2, v0 - v3, v3 - v7),
            (v4 - v5, v4 - v7, v0 - v4), (v4 - v5, v5 - v6, v1 - v5), (v7 - v6, v5 - v6, v2 - v6), (v7 - v6, v4 - v7, v3 - v7)]


def edge_key(x: int, y: int, z: int, e: int):
    """Global identity of edge e of cell (x, y, z): (lower lattice point, axis 0 = x / 1 = y / 2 = z); e = 12: (cell, 3)."""
    if e == 12:
        return (x, y, z), 3
    a, b = EDGE_ENDS[e]
    pa, pb = CORNER[a], CORNER[b]
    axis = 0 if pa[0] != pb[0] else (1 if pa[1] != pb[1] else 2)
    return (x + min(pa[0], pb[0]), y + min(pa[1], pb[1]), z + min(pa[2], pb[2])), axis


def marching_cubes_raw(volume: np.ndarray, level: float, mask: Optional[np.ndarray] = None):
    """_marching_cubes_lewiner_cy.marching_cubes(volume, level, luts, 1, False, mask): vertices [V,3] float32 in (x, y, z) = (axis 2, 1, 0),
    faces [3F] int32 (right-handed), unit normals [V,3] float32 in (x, y, z), values [V] float32 - in the port's own order."""
    F = np.float32
    vol = np.ascontiguousarray(volume, np.float32)
    nz, ny, nx = vol.shape
    level = float(level)
    verts, faces, normals, values, index = [], [], [], [], {}
    for z in range(nz - 1):
        for y in range(ny - 1):
            for x in range(nx - 1):
                if mask is not None and not mask[z + 1, y + 1, x + 1]:
                    continue
                v = [float(vol[z + dz, y + dy, x + dx]) - level for (dx, dy, dz) in CORNER]
                if all(t > 0 for t in v) or all(t <= 0 for t in v):
                    continue
                tris, _ = cell_triangles(v)
                vgl = _vg_table(v)
                vgs = [vgl[_SWAP[p]] for p in range(8)]
                spread = F(max(v) - min(v))
                centre = None
                for e in tris:
                    key = edge_key(x, y, z, e)
                    if e == 12:
                        if centre is None:
                            w = [1.0 / (EPS + abs(t)) for t in v]
                            fx = fy = fz = ff = 0.0
                            for p in range(8):
                                fx += CORNER[p][0] * w[p]
                                fy += CORNER[p][1] * w[p]
                                fz += CORNER[p][2] * w[p]
                                ff += w[p]
                            g = [0.0, 0.0, 0.0]
                            for k in range(3):
                                for p in range(8):
                                    g[k] += w[p] * vgl[p][k]
                            centre = ((x + fx / ff, y + fy / ff, z + fz / ff), g)
                        if key not in index:
                            index[key] = len(verts)
                            verts.append([F(t) for t in centre[0]])
                            normals.append([F(0), F(0), F(0)])
                            values.append(F(0))
                        vi = index[key]
                        normals[vi][0] = F(float(normals[vi][0]) + float(F(centre[1][2])))
                        normals[vi][1] = F(float(normals[vi][1]) + float(F(centre[1][1])))
                    else:
                        a, b = EDGE_ENDS[e]
                        w1, w2 = 1.0 / (EPS + abs(v[a])), 1.0 / (EPS + abs(v[b]))
                        if key not in index:
                            ff = w1 + w2
                            fx = CORNER[a][0] * w1 + CORNER[b][0] * w2
                            fy = CORNER[a][1] * w1 + CORNER[b][1] * w2
                            fz = CORNER[a][2] * w1 + CORNER[b][2] * w2
                            index[key] = len(verts)
                            verts.append([F(x + fx / ff), F(y + fy / ff), F(z + fz / ff)])
                            normals.append([F(0), F(0), F(0)])
                            values.append(F(0))
                        vi = index[key]
                        for (cc, ww) in ((a, w1), (b, w2)):
                            s = float(F(ww))
                            for k in range(3):
                                normals[vi][k] = F(float(normals[vi][k]) + float(F(vgs[cc][k] * s)))
                    faces.append(vi)
                    if spread > values[vi]:
                        values[vi] = spread
    verts = np.array(verts, np.float32).reshape(-1, 3)
    n64 = np.array(normals, np.float32).reshape(-1, 3).astype(np.float64)
    ln = np.sqrt(n64[:, 0] * n64[:, 0] + n64[:, 1] * n64[:, 1] + n64[:, 2] * n64[:, 2])
    normals = (n64 / np.where(ln > 0, ln, 1.0)[:, None]).astype(np.float32)
    return verts, np.array(faces, np.int32), normals, np.array(values, np.float32)


def marching_cubes(volume: np.ndarray, level: Optional[float] = None, spacing=(1.0, 1.0, 1.0), gradient_direction: str = "descent",
                   mask: Optional[np.ndarray] = None):
    """skimage.measure.marching_cubes(volume, level, spacing=..., gradient_direction=..., mask=...) with its defaults
    (step_size 1, allow_degenerate True, method "lewiner"): _marching_cubes_lewiner.py::_marching_cubes_lewiner."""
    if not isinstance(volume, np.ndarray) or volume.ndim != 3:
        raise ValueError("Input volume should be a 3D numpy array.")
    if min(volume.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    volume = np.ascontiguousarray(volume, np.float32)
    if level is None:
        level = 0.5 * (volume.min() + volume.max())
    else:
        level = float(level)
        if level < volume.min() or level > volume.max():
            raise ValueError("Surface level must be within volume data range.")
    if len(spacing) != 3:
        raise ValueError("`spacing` must consist of three floats.")
    if mask is not None and mask.shape != volume.shape:
        raise ValueError("volume and mask must have the same shape.")
    verts, faces, normals, values = marching_cubes_raw(volume, level, mask)
    if not len(verts):
        raise RuntimeError("No surface found at the given iso value.")
    verts, normals = np.fliplr(verts), np.fliplr(normals)
    faces = faces.reshape(-1, 3)
    if gradient_direction == "descent":
        faces = np.fliplr(faces)
    elif gradient_direction != "ascent":
        raise ValueError("Incorrect input %s in `gradient_direction`, see docstring." % gradient_direction)
    if not np.array_equal(spacing, (1, 1, 1)):
        verts = verts * np.r_[spacing]
    return verts, faces, normals, values
