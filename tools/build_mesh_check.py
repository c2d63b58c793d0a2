"""Build tests/_bin/{mesh_cases.bin, mesh_gpu_check} WITH the crop-512 expectation and the content digests tests/test_gpu_zy_mesh_abi.py
looks for (so that the GPU box - and the driver's round-end run - uses these files instead of re-making a lighter set)."""
import hashlib
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_bin")


def digest(paths):
    h = hashlib.sha256()
    for q in sorted(paths):
        h.update(os.path.basename(q).encode())
        with open(q, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    os.makedirs(BIN, exist_ok=True)
    mesh_src = [q for q in glob.glob(os.path.join(ROOT, "sdfstudio_amd", "csrc_mesh", "*")) if os.path.isfile(q)] + [os.path.join(ROOT, "tests", "mesh_host_check.cpp")]
    d_cases = digest(glob.glob(os.path.join(ROOT, "tests", "golden", "mc_*.npz")) + [os.path.join(ROOT, "tools", "pack_mesh_cases.py")] + mesh_src)
    d_exe = digest([os.path.join(ROOT, "tools", "mesh_gpu_check.cpp"), os.path.join(ROOT, "include", "sdfmesh.h")])
    cases, exe = os.path.join(BIN, "mesh_cases.bin"), os.path.join(BIN, "mesh_gpu_check")

    def have(target, want):
        try:
            return os.path.exists(target) and open(target + ".digest").read().strip() == want
        except OSError:
            return False

    if not have(cases, d_cases) or "--force" in sys.argv:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pack_mesh_cases.py")])
        open(cases + ".digest", "w").write(d_cases)
    if not have(exe, d_exe) or "--force" in sys.argv:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tools", "mesh_gpu_check.cpp"), "-o", exe, "-ldl"])
        open(exe + ".digest", "w").write(d_exe)
    print("mesh check artefacts up to date:", d_cases, d_exe)


if __name__ == "__main__":
    main()
