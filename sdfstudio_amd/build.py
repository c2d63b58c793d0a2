"""Build libsdfhip.so (hand-written HIP kernels for gfx950 + the C ABI) in-tree with hipcc.

Cross-compiles without a GPU.  Objects are cached under ``sdfstudio_amd/csrc/_build`` keyed by a hash of the
sources so rebuilding after an edit only recompiles what changed.  The shared object lands next to this file
(``sdfstudio_amd/libsdfhip.so``): git-ignored, but it travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libsdfhip.so")
SOURCES = ["inst_a_inf.hip", "inst_c_inf.hip", "inst_a_bwd.hip", "inst_c_bwd.hip", "inst_a_fwd.hip", "inst_c_fwd.hip", "inst_b.hip", "api.hip", "inst_w.hip", "inst_d.hip", "inst_e.hip",
           "inst_a.hip", "inst_c.hip"]  # slowest first
# -fno-slp-vectorize: the SLP pass packs adjacent scalar fp32 adds / muls of the producers into v_pk_* instructions, which cost
# MORE issue time beside MFMAs than the scalar forms (MI355X_MICROARCH.md, "price of one filler"), and it is ~20 % of the compile time
# of the fully unrolled fused kernels (inst_a_bwd.hip: 238 s -> 190 s)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-result", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the sdfhip extension cannot be built")


def _closure(name: str, seen: set) -> None:
    """Local headers a source pulls in (transitively), so an edit only rebuilds the translation units that see it."""
    if name in seen:
        return
    seen.add(name)
    with open(os.path.join(CSRC, name), "r") as fh:
        for line in fh:
            m = re.match(r'\s*#\s*include\s+"([^"]+)"', line)
            if m and os.path.exists(os.path.join(CSRC, m.group(1))):
                _closure(os.path.normpath(m.group(1)), seen)  # also "../../include/sdfhip.h": only the units that see the ABI header


def _digest(src: str) -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    deps: set = set()
    _closure(src, deps)
    for name in sorted(deps):
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def source_digest() -> str:
    """One digest over the compile flags and EVERY source the library is built from (each translation unit's closure): the identity
    evidence files carry (profiles/*_traffic.json: `library_digest`) and bench.py compares with the library it has loaded."""
    h = hashlib.sha256()
    for src in sorted(SOURCES):
        h.update(_digest(src).encode())
    return h.hexdigest()[:16]


def built_digest() -> str:
    """The digest recorded next to the .so when it was linked ("" if none): what the LOADED library was built from."""
    try:
        with open(LIB + ".digest") as fh:
            return fh.read().strip()
    except OSError:
        return ""


def _compile(src: str) -> str:
    obj = os.path.join(BUILD, f"{os.path.splitext(src)[0]}.{_digest(src)}.o")
    if not os.path.exists(obj):
        for old in os.listdir(BUILD):
            if old.startswith(os.path.splitext(src)[0] + ".") and old.endswith(".o"):
                os.remove(os.path.join(BUILD, old))
        cmd = [_hipcc(), *FLAGS, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


# ---- libsdfmesh.so: marching cubes on the device (include/sdfmesh.h).  A separate library with its own sources: libsdfhip.so's
# digest - the identity the profiles/ evidence is tied to - does not move when the mesh code does.
MESH_CSRC = os.path.join(HERE, "csrc_mesh")
MESH_LIB = os.path.join(HERE, "libsdfmesh.so")
# -ffp-contract=off: the case tests compare products in double and the vertex arithmetic rounds where scikit-image rounds; a fused
# multiply-add in either place changes bits (mc_cell.h)
MESH_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-Wno-unused-function"]


def mesh_source_digest() -> str:
    h = hashlib.sha256()
    h.update(" ".join(MESH_FLAGS).encode())
    # the kernels' sources (csrc_mesh/*.h, *.hip); the public header holds declarations and prose only and is not part of the identity
    for name in sorted(os.listdir(MESH_CSRC)):
        path = os.path.join(MESH_CSRC, name)
        if os.path.isfile(path) and name.endswith((".h", ".hip")):
            with open(path, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def build_mesh(verbose: bool = True) -> str:
    digest = mesh_source_digest()
    stamp = MESH_LIB + ".digest"
    prev = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if prev != digest or not os.path.exists(MESH_LIB):
        cmd = [_hipcc(), *MESH_FLAGS, "-shared", "-I", MESH_CSRC, os.path.join(MESH_CSRC, "mesh_api.hip"), "-o", MESH_LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for mesh_api.hip:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as fh:
            fh.write(digest)
    if verbose:
        print(f"[sdfhip] built {MESH_LIB} ({os.path.getsize(MESH_LIB) / 1e6:.1f} MB)")
    return MESH_LIB


def build(verbose: bool = True) -> str:
    build_mesh(verbose)
    os.makedirs(BUILD, exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(SOURCES), os.cpu_count() or 1))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    stamp = os.path.join(BUILD, "link.stamp")
    key = " ".join(os.path.basename(o) for o in objs)
    prev = open(stamp).read() if os.path.exists(stamp) else ""
    if prev != key or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as fh:
            fh.write(key)
    with open(LIB + ".digest", "w") as fh:  # travels to the GPU box with the .so (git-ignored like it)
        fh.write(source_digest())
    if verbose:
        print(f"[sdfhip] built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
