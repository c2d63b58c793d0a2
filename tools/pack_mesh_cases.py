"""Pack tests/golden/mc_*.npz into tests/_bin/mesh_cases.bin for tools/mesh_gpu_check.cpp: per case the volume, level, mask, the
orientation flag and the arrays libsdfmesh.so must return (scikit-image's raw arrays in the library's convention: vertices / normals in
volume axis order, lattice units; faces flipped for gradient_direction "descent")."""
import glob
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mc_*.npz")))
    os.makedirs(os.path.join(ROOT, "tests", "_bin"), exist_ok=True)
    with open(os.path.join(ROOT, "tests", "_bin", "mesh_cases.bin"), "wb") as fh:
        fh.write(struct.pack("<i", len(files)))
        for path in files:
            g = np.load(path)
            vol = np.ascontiguousarray(g["volume"], np.float32)
            flip = 0 if bool(g["ascent"]) else 1
            verts = np.ascontiguousarray(np.fliplr(g["raw_verts"]), np.float32)
            faces = g["raw_faces"].reshape(-1, 3)
            faces = np.ascontiguousarray(np.fliplr(faces) if flip else faces, np.int32)
            normals = np.ascontiguousarray(np.fliplr(g["raw_normals"]), np.float32)
            values = np.ascontiguousarray(g["raw_values"], np.float32)
            name = os.path.basename(path)[3:-4].encode()[:63]
            fh.write(name + b"\0" * (64 - len(name)))
            fh.write(struct.pack("<3i", *vol.shape))
            fh.write(struct.pack("<d", float(g["level"])))
            fh.write(struct.pack("<ii", int("mask" in g.files), flip))
            fh.write(struct.pack("<qq", len(verts), len(faces)))
            fh.write(vol.tobytes())
            if "mask" in g.files:
                fh.write(np.ascontiguousarray(g["mask"], np.uint8).tobytes())
            fh.write(verts.tobytes() + faces.tobytes() + normals.tobytes() + values.tobytes())
    print("packed", len(files), "cases")


if __name__ == "__main__":
    main()
