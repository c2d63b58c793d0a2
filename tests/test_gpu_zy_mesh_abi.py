"""libsdfmesh.so through its C ABI on the GPU, with no Python between the test and the library: tools/mesh_gpu_check.cpp (the program
that proved the library on an MI355X in round 5: profiles/r5_mesh_gpu_check_v2.jsonl) run as a child process.  Every golden case of
tests/golden/mc_*.npz - the real scikit-image's arrays - must come back bit for bit; when the packed cases carry the expectation for the
reference's 512^3 crop (tools/pack_mesh_cases.py without --no-crop512), the whole crop's mesh as well.  Builds what is missing (hipcc is in
the image); unlike tests/test_gpu_zz_mesh.py this does not go through the Python binding."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_bin")


@pytest.mark.gpu
def test_mesh_library_c_abi_against_scikit_image_goldens(device, tmp_path):
    exe, cases = os.path.join(BIN, "mesh_gpu_check"), os.path.join(BIN, "mesh_cases.bin")
    os.makedirs(BIN, exist_ok=True)
    # never test with a helper / case file made from other sources than the tree's.  By CONTENT (a digest file next to each
    # artefact), not by mtime: the snapshot that travels to the GPU box does not keep modification times in order.
    import glob
    import hashlib

    def digest(paths):
        h = hashlib.sha256()
        for q in sorted(paths):
            h.update(os.path.basename(q).encode())
            with open(q, "rb") as fh:
                h.update(fh.read())
        return h.hexdigest()[:16]

    def stale(target, want):
        try:
            return not os.path.exists(target) or open(target + ".digest").read().strip() != want
        except OSError:
            return True

    mesh_src = [q for q in glob.glob(os.path.join(ROOT, "sdfstudio_amd", "csrc_mesh", "*")) if os.path.isfile(q)] + [os.path.join(ROOT, "tests", "mesh_host_check.cpp")]
    d_cases = digest(glob.glob(os.path.join(ROOT, "tests", "golden", "mc_*.npz")) + [os.path.join(ROOT, "tools", "pack_mesh_cases.py")] + mesh_src)
    d_exe = digest([os.path.join(ROOT, "tools", "mesh_gpu_check.cpp"), os.path.join(ROOT, "include", "sdfmesh.h")])
    if stale(cases, d_cases):
        # (the crop-512 expectation is the host harness' mesh of that crop: heavy - left out when the file has to be made inside a test)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pack_mesh_cases.py"), "--no-crop512"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        open(cases + ".digest", "w").write(d_cases)
    if stale(exe, d_exe):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tools", "mesh_gpu_check.cpp"), "-o", exe, "-ldl"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        open(exe + ".digest", "w").write(d_exe)
    from sdfstudio_amd import _mesh  # the library must be there (no fallback); the child dlopens the same file

    out = tmp_path / "check.jsonl"
    r = subprocess.run([exe, _mesh.LIB_PATH, cases, str(out)], capture_output=True, text=True, timeout=300)
    lines = [json.loads(ln) for ln in open(out)] if out.exists() else []
    assert r.returncode == 0, (r.returncode, lines[-3:], r.stderr[-1000:])
    per_case = [ln for ln in lines if "case" in ln]
    assert len(per_case) >= 6 and all(ln["ok"] == 1 for ln in per_case), per_case
    assert any(ln.get("all_bit_exact") == 1 for ln in lines)
    crop = [ln["crop512_vs_host_harness"] for ln in lines if "crop512_vs_host_harness" in ln]
    assert crop and (crop[0]["have_expected"] == 0 or crop[0]["bit_exact"] == 1), crop
    timing = [ln["crop512"] for ln in lines if "crop512" in ln]
    assert timing and timing[0]["rc"] == 0 and timing[0]["V"] > 100_000
    print("libsdfmesh C ABI:", json.dumps(timing[0]))
