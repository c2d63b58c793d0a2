"""Mint the marching-cubes golden vectors from the REAL scikit-image (tests/golden/mc_*.npz).

    /opt/conda/bin/python3.9 tests/golden/make_golden_mc.py        (scikit-image 0.18.3 in the build container; the reference pins 0.19.3,
                                                                    whose Lewiner implementation is the same code)

Each file holds the volume, level, mask, spacing and what ``skimage.measure.marching_cubes`` (default method, as
nerfstudio/utils/marching_cubes.py:125-134 calls it) returned, plus the RAW output of the compiled routine underneath
(_marching_cubes_lewiner_cy.marching_cubes: (x, y, z) order, unflipped faces) so that array order is pinned too.
"""
import json
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))


def volumes():
    rng = np.random.default_rng(20250923)
    out = {}
    # white noise: every case of the table, ambiguous faces and tunnels, centre vertices
    out["noise_9x8x10"] = dict(volume=rng.standard_normal((9, 8, 10)).astype(np.float32), level=0.0)
    out["noise_level_7x7x7"] = dict(volume=rng.standard_normal((7, 7, 7)).astype(np.float32), level=0.21)
    # a sphere SDF on an anisotropic lattice, the way the reference calls it: spacing from the crop, level 0
    z, y, x = np.meshgrid(np.linspace(-1, 1, 20), np.linspace(-1, 1, 18), np.linspace(-1, 1, 22), indexing="ij")
    out["sphere_20x18x22"] = dict(volume=(np.sqrt(x * x + y * y + z * z) - 0.63).astype(np.float32), level=0.0,
                                  spacing=(2.0 / 19, 2.0 / 17, 2.0 / 21))
    # two blobs whose union has saddle cells + a mask (the reference's coarse_mask path, marching_cubes.py:68-75,133)
    f = np.minimum(np.sqrt((x - 0.3) ** 2 + y * y + z * z) - 0.4, np.sqrt((x + 0.3) ** 2 + (y - 0.1) ** 2 + z * z) - 0.38)
    out["blobs_masked"] = dict(volume=(f + 0.02 * rng.standard_normal(f.shape)).astype(np.float32), level=0.0,
                               mask=rng.random(f.shape) > 0.25, spacing=(0.1, 0.2, 0.3))
    # exact zeros at lattice points (value == level), ascent orientation
    v = np.round(rng.standard_normal((6, 6, 6)) * 2).astype(np.float32) / 2
    out["ties_ascent"] = dict(volume=v, level=0.0, gradient_direction="ascent")
    # tiny magnitudes: the port's epsilons decide
    out["tiny_1e-8"] = dict(volume=(rng.standard_normal((6, 7, 6)) * 1e-8).astype(np.float32), level=0.0)
    # Lewiner's sub-cases 6.1.2 and 7.4.2 (round 6): reachable only through exact ties on the tested face (find_mc_subcase_cells.py says
    # why); 80 such cells side by side in one [2, 2, 160] volume, every second cell switched off by the mask so that no two share a face
    cells = json.load(open(os.path.join(HERE, "mc_subcase_cells.json")))
    cubes = cells["6.1.2"] + cells["7.4.2"]
    corner = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]  # Lewiner's numbering -> (dx, dy, dz)
    vol = np.zeros((2, 2, 2 * len(cubes)), np.float32)
    mask = np.zeros(vol.shape, bool)
    for k, cube in enumerate(cubes):
        for p, (dx, dy, dz) in enumerate(corner):
            vol[dz, dy, 2 * k + dx] = cube[p]
        mask[1, 1, 2 * k + 1] = True  # cell (0, 0, 2 k) is processed iff mask[1, 1, 2 k + 1]
    out["lewiner_subcases"] = dict(volume=vol, level=0.0, mask=mask)
    return out


def main():
    import skimage
    from skimage import measure
    from skimage.measure import _marching_cubes_lewiner as M

    luts = M._get_mc_luts()
    raw = M._marching_cubes_lewiner_cy.marching_cubes
    only = set(sys.argv[1:])  # names to (re)write; none: all
    for name, kw in volumes().items():
        if only and name not in only:
            continue
        args = dict(level=kw["level"], spacing=kw.get("spacing", (1.0, 1.0, 1.0)), gradient_direction=kw.get("gradient_direction", "descent"),
                    mask=kw.get("mask"))
        verts, faces, normals, values = measure.marching_cubes(kw["volume"], **args)
        rv, rf, rn, rval = raw(kw["volume"], float(kw["level"]), luts, 1, False, kw.get("mask"))
        save = dict(volume=kw["volume"], level=np.float64(kw["level"]), spacing=np.asarray(args["spacing"], np.float64),
                    ascent=np.bool_(args["gradient_direction"] == "ascent"), verts=verts, faces=faces, normals=normals, values=values,
                    raw_verts=rv, raw_faces=rf, raw_normals=rn, raw_values=rval, skimage_version=np.bytes_(skimage.__version__))
        if kw.get("mask") is not None:
            save["mask"] = kw["mask"]
        np.savez_compressed(os.path.join(HERE, "mc_%s.npz" % name), **save)
        print(name, kw["volume"].shape, "V", len(verts), "F", len(faces), verts.dtype, faces.dtype, normals.dtype)


if __name__ == "__main__":
    main()
