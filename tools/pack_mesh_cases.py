"""Pack tests/golden/mc_*.npz into tests/_bin/mesh_cases.bin for tools/mesh_gpu_check.cpp: per case the volume, level, mask, the
orientation flag and the arrays libsdfmesh.so must return (scikit-image's raw arrays in the library's convention: vertices / normals in
volume axis order, lattice units; faces flipped for gradient_direction "descent")."""
import glob
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_fnv():
    import subprocess

    src = os.path.join(ROOT, "tests", "_bin", "fnv1a.c")
    with open(src, "w") as fh:
        fh.write("#include <stdio.h>\n#include <stdint.h>\nint main(){uint64_t h=1469598103934665603ull;int c;"
                 "while((c=getchar())!=EOF){h^=(unsigned char)c;h*=1099511628211ull;}printf(\"%llu\\n\",(unsigned long long)h);return 0;}\n")
    subprocess.check_call(["gcc", "-O2", src, "-o", os.path.join(ROOT, "tests", "_bin", "fnv1a")])


def main():
    os.makedirs(os.path.join(ROOT, "tests", "_bin"), exist_ok=True)
    _build_fnv()
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
        if "--no-crop512" in sys.argv:
            print("packed", len(files), "cases (no crop512 expectation)")
            return
        # the reference's crop size: the analytic volume of mesh_gpu_check.cpp::fill_volume, made here with the same correctly rounded
        # float32 operations in the same order, meshed by the host harness (the kernels' own functions compiled by g++)
        import subprocess
        import tempfile

        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import pathlib

        import test_cpu_marching_cubes as T

        n = 512
        f32 = np.float32
        s_ = f32(2.0) / f32(n - 1)
        ax = (f32(-1) + np.arange(n, dtype=np.float32) * s_).astype(np.float32)
        fz, fy, fx = np.meshgrid(ax, ax, ax, indexing="ij")  # volume index = (z * n + y) * n + x
        a = np.sqrt((fx * fx + fy * fy) + fz * fz) - f32(0.55) - ((f32(0.2) * fx) * fy) * fz
        dx, dy, dz = fx - f32(0.8), fy - f32(0.8), fz - f32(0.8)
        b = np.sqrt((dx * dx + dy * dy) + dz * dz) - f32(0.1)
        vol = np.where(a < b, a, b).astype(np.float32)
        del a, b, dx, dy, dz, fx, fy, fz
        verts, faces, normals, values = T._run_host(vol, 0.0, None, pathlib.Path(tempfile.mkdtemp()), "crop512")

        def fnv1a(arr):
            # FNV-1a 64 over the bytes, vectorised is awkward: do it in C via a tiny helper
            return int(subprocess.check_output([os.path.join(ROOT, "tests", "_bin", "fnv1a")], input=np.ascontiguousarray(arr).tobytes()).strip())

        fh.write(struct.pack("<qq", len(verts), len(faces)))
        fh.write(struct.pack("<4Q", fnv1a(verts), fnv1a(faces), fnv1a(normals), fnv1a(values)))
        print("crop512: V", len(verts), "F", len(faces))
    print("packed", len(files), "cases")


if __name__ == "__main__":
    main()
