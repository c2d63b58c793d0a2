"""The rules that keep the checker out of the product: oracle/ is imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
only; the host build of the mesh kernels' logic (MC_HOST_CHECK) exists for the test harness only."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _python_files(*dirs):
    for d in dirs:
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            if "__pycache__" in base or os.sep + "experiments" in base:
                continue
            for f in files:
                if f.endswith(".py"):
                    yield os.path.join(base, f)


def _imports_oracle(path):
    tree = ast.parse(open(path).read())
    hits = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Import) and any(a.name == "oracle" or a.name.startswith("oracle.") for a in node.names):
            hits.append(node.lineno)
        if isinstance(node, ast.ImportFrom) and node.module and (node.module == "oracle" or node.module.startswith("oracle.")):
            hits.append(node.lineno)
    return hits


def test_the_product_never_imports_the_oracle():
    offenders = {p: _imports_oracle(p) for p in _python_files("sdfstudio_amd")}
    assert not {p: h for p, h in offenders.items() if h}
    # the bench legs that run as child processes are product-side too
    assert not _imports_oracle(os.path.join(ROOT, "tools", "mesh_leg.py"))


def test_bench_imports_the_oracle_in_its_cpu_baseline_leg_only():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            inner = [n for n in ast.walk(node) if isinstance(n, (ast.Import, ast.ImportFrom))
                     and (getattr(n, "module", None) or "").split(".")[0] == "oracle"
                     or isinstance(n, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in n.names)]
            if inner:
                assert node.name == "cpu_baseline", node.name
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any((getattr(n, "module", None) or "").split(".")[0] == "oracle" for n in top)


def test_the_host_build_of_the_mesh_logic_is_test_infrastructure_only():
    """MC_HOST_CHECK turns csrc_mesh/mc_cell.h into plain host functions for tests/mesh_host_check.cpp; nothing the library is built from
    defines it, and the library's translation unit has no host implementation of a pass."""
    defs = []
    for base, _, files in os.walk(ROOT):
        if ".git" in base or "gpurun_out" in base or "_bin" in base:
            continue
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".py", ".sh")) and f != "test_cpu_boundaries.py":
                text = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"#\s*define\s+MC_HOST_CHECK|-DMC_HOST_CHECK", text):
                    defs.append(os.path.relpath(os.path.join(base, f), ROOT))
    assert defs == [os.path.join("tests", "mesh_host_check.cpp")], defs
    from sdfstudio_amd import build as b

    assert not any("MC_HOST_CHECK" in flag for flag in b.MESH_FLAGS)
