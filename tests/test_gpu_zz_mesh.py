"""Mesh extraction on the GPU: libsdfmesh.so through its C ABI against the golden vectors minted from the real
scikit-image, against the oracle on ragged volumes, and - at the reference's crop size - through properties no oracle run is needed for.

Hardware history.  Round 5: the C ABI proven on an MI355X by tools/mesh_gpu_check.cpp (no Python in the process); the driver's round-end
run then showed these Python-binding cases green as well (6 XPASS in GPUTEST_r05.json), so the xfail mark they carried is gone (VERDICT r5,
ADVICE r5): a broken binding fails the suite.  Round 6 rewrote the library's data flow (bit arrays, ordered compaction, no sort / hipCUB, one
host read: csrc_mesh/mesh_api.hip); the cases are unchanged - same arrays, same bits.  Every case still runs in a CHILD process (a fault
there cannot take the suite's process or its HIP context down).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CASES = ["golden", "random_vs_oracle", "errors", "crop_512_properties", "masked_and_flipped", "surface_sliding_glue"]


def _child(case: str, timeout: int = 900):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), case], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0 and "CASE-OK" in r.stdout, f"[{case}] rc {r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-4000:]}"


@pytest.mark.parametrize("case", CASES)
def test_mesh(case, device):
    _child(case)


# ------------------------------------------------------------------------------------------------ the cases (child process)

def _np(t):
    return t.detach().cpu().numpy()


def case_golden():
    """Every golden file: the four arrays of skimage.measure.marching_cubes, identical bits, identical order."""
    import glob

    import numpy as np
    import torch

    from sdfstudio_amd.utils.marching_cubes import marching_cubes

    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mc_*.npz")))
    assert len(files) >= 6
    for path in files:
        g = np.load(path)
        mask = torch.from_numpy(g["mask"]).cuda() if "mask" in g.files else None
        verts, faces, normals, values = marching_cubes(torch.from_numpy(g["volume"]).cuda(), float(g["level"]), spacing=tuple(g["spacing"]),
                                                       gradient_direction="ascent" if bool(g["ascent"]) else "descent", mask=mask)
        assert _np(verts).dtype == g["verts"].dtype, (path, verts.dtype)
        assert np.array_equal(_np(verts), g["verts"]), path
        assert np.array_equal(_np(faces), g["faces"]), path
        assert np.array_equal(_np(normals), g["normals"]), path
        assert np.array_equal(_np(values), g["values"]), path


def case_random_vs_oracle():
    """Ragged shapes from 2 x 2 x 2 up, magnitudes 1e-12 .. 1e+3, levels, masks, volumes without a surface: the oracle's arrays."""
    import numpy as np
    import torch

    from oracle import marching_cubes as OM
    from sdfstudio_amd import _mesh

    rng = np.random.default_rng(5)
    nonempty = 0
    for trial in range(60):
        shape = tuple(int(s) for s in rng.integers(2, 12, 3))
        vol = (rng.standard_normal(shape) * 10.0 ** rng.uniform(-12, 3)).astype(np.float32)
        if trial % 9 == 0:
            vol = np.abs(vol) + 1
        level = 0.0 if trial % 2 else float(np.median(vol))
        mask = (rng.random(shape) > 0.4) if trial % 3 == 0 else None
        ov, of, on, oval = OM.marching_cubes_raw(vol, level, mask)
        verts, faces, normals, values = _mesh.marching_cubes_device(torch.from_numpy(vol).cuda(), level,
                                                                    None if mask is None else torch.from_numpy(mask).cuda(), flip_faces=False)
        assert np.array_equal(_np(verts), np.fliplr(ov)), (trial, shape)
        assert np.array_equal(_np(faces), of.reshape(-1, 3)), (trial, shape)
        assert np.array_equal(_np(normals), np.fliplr(on)), (trial, shape)
        assert np.array_equal(_np(values), oval), (trial, shape)
        nonempty += len(ov) > 0
    assert nonempty >= 40


def case_errors():
    """scikit-image's checks and messages (skimage/measure/_marching_cubes_lewiner.py), and the no-CPU-path rule."""
    import torch

    from sdfstudio_amd import _mesh
    from sdfstudio_amd.utils.marching_cubes import marching_cubes

    v = torch.ones(4, 4, 4, device="cuda")
    for exc, match, call in ((ValueError, "within volume data range", lambda: marching_cubes(v, 2.0)),
                             (ValueError, "at least 2x2x2", lambda: marching_cubes(torch.ones(1, 4, 4, device="cuda"), 1.0)),
                             (RuntimeError, "No surface found", lambda: marching_cubes(v, 1.0)),
                             (ValueError, "same shape", lambda: marching_cubes(v, 1.0, mask=torch.ones(4, 4, 3, device="cuda", dtype=torch.bool))),
                             (_mesh.SdfMeshError, "no CPU path", lambda: _mesh.marching_cubes_device(torch.ones(4, 4, 4), 0.0))):
        try:
            call()
        except exc as e:  # noqa: PERF203
            assert match in str(e), (match, str(e))
        else:
            raise AssertionError(f"expected {exc.__name__} ({match})")


def _manifold_stats(faces, n_verts):
    """Directed edges of a triangle soup -> (every undirected edge is used exactly twice, once per direction; Euler characteristic)."""
    import torch

    f = faces.long()
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = e[:, 0] * n_verts + e[:, 1]
    rev = e[:, 1] * n_verts + e[:, 0]
    uniq, cnt = torch.unique(key, return_counts=True)
    directed_once = bool((cnt == 1).all())
    paired = bool(torch.isin(rev, uniq).all())
    n_edges = uniq.numel() // 2
    return directed_once and paired, n_verts - n_edges + f.shape[0]


def case_crop_512_properties():
    """The reference's crop (512^3, marching_cubes.py:31) - no oracle finishes there, the domain's invariants stand in: a closed surface
    comes out closed, consistently oriented and with Euler characteristic 2 per component; vertices sit on the iso-surface; normals point
    along the gradient; two runs are bit-identical; a sub-block of the volume meshes to the same triangles the oracle gives for it."""
    import numpy as np
    import torch

    from oracle import marching_cubes as OM
    from sdfstudio_amd.utils.marching_cubes import marching_cubes

    n = 512
    ax = torch.linspace(-1, 1, n, device="cuda")
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    # two disjoint closed surfaces: a bumpy sphere and a small sphere (Euler characteristic 2 + 2)
    r = torch.sqrt(xx * xx + yy * yy + zz * zz)
    vol = torch.minimum(r - 0.55 - 0.03 * torch.sin(9 * xx) * torch.sin(7 * yy) * torch.sin(5 * zz),
                        torch.sqrt((xx - 0.8) ** 2 + (yy - 0.8) ** 2 + (zz - 0.8) ** 2) - 0.1).contiguous()
    del r
    verts, faces, normals, values = marching_cubes(vol, 0.0)
    V, F = verts.shape[0], faces.shape[0]
    assert V > 100_000 and F > 200_000, (V, F)
    assert int(faces.min()) == 0 and int(faces.max()) == V - 1
    closed, euler = _manifold_stats(faces, V)
    assert closed, "an edge of a closed surface is not shared by exactly two consistently oriented triangles"
    assert euler == 4, euler
    # vertices are on lattice edges: two integer coordinates; trilinear interpolation of the volume at the vertex is ~ 0
    frac = verts - torch.floor(verts)
    assert float(((frac == 0).sum(1) >= 2).float().mean()) > 0.999  # (a centre vertex of a tunnel tiling has three fractional coordinates)
    p = verts / (n - 1) * 2 - 1
    pr = torch.sqrt((p * p).sum(1))
    big = (p - 0.8).pow(2).sum(1).sqrt() > 0.2
    assert float((pr[big] - 0.55).abs().max()) < 0.035
    # normals: unit length, along the volume's gradient (here: radial); verts and normals share the volume's axis order
    assert float((normals.pow(2).sum(1).sqrt() - 1).abs().max()) < 1e-5
    radial = (p[big] / pr[big, None] * normals[big]).sum(1).abs()
    assert float(radial.mean()) > 0.9
    v2, f2, n2, val2 = marching_cubes(vol, 0.0)
    assert torch.equal(verts, v2) and torch.equal(faces, f2) and torch.equal(normals, n2) and torch.equal(values, val2)
    # a sub-block through the surface against the oracle (the mesh of a sub-volume is the oracle's mesh of that sub-volume)
    sub = vol[250:262, 384:400, 250:264].contiguous()  # y = 0.503 .. 0.562 at x, z ~ 0: through r = 0.55
    sv, sf, sn, sval = marching_cubes(sub, 0.0)
    ov, of, on, oval = OM.marching_cubes(_np(sub), 0.0)
    assert np.array_equal(_np(sv), ov) and np.array_equal(_np(sf), of) and np.array_equal(_np(sn), on) and np.array_equal(_np(sval), oval)


def case_masked_and_flipped():
    """mask + both orientations on a mid-size noisy volume against the oracle (28 x 24 x 30: 1.8e4 cells, every case of the table)."""
    import numpy as np
    import torch

    from oracle import marching_cubes as OM
    from sdfstudio_amd.utils.marching_cubes import marching_cubes

    rng = np.random.default_rng(3)
    shape = (28, 24, 30)
    z, y, x = np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing="ij")
    vol = (np.sqrt(x * x + y * y + z * z) - 0.6 + 0.15 * rng.standard_normal(shape)).astype(np.float32)
    mask = rng.random(shape) > 0.2
    for direction in ("descent", "ascent"):
        for m in (None, mask):
            a = marching_cubes(torch.from_numpy(vol).cuda(), 0.05, spacing=(0.5, 1.0, 2.0), gradient_direction=direction,
                               mask=None if m is None else torch.from_numpy(m).cuda())
            b = OM.marching_cubes(vol, 0.05, spacing=(0.5, 1.0, 2.0), gradient_direction=direction, mask=m)
            for name, s, t in zip(("verts", "faces", "normals", "values"), a, b):
                assert _np(s).dtype == t.dtype and np.array_equal(_np(s), t), (direction, m is not None, name)


def case_surface_sliding_glue():
    """get_surface_sliding (marching_cubes.py:15-168) on the small golden field with 2 x 2 x 2 crops of 32^3: the concatenated mesh equals
    the oracle's marching cubes of the per-crop volumes the same call returns with return_volumes=True, offsets and spacing in double."""
    import numpy as np
    import torch

    from helpers import load_golden, product_model_from_params, small_oracle_cfg
    from oracle import marching_cubes as OM
    from sdfstudio_amd.utils.marching_cubes import get_surface_sliding

    dev = torch.device("cuda:0")
    g = load_golden("train")
    model = product_model_from_params(g["param"], small_oracle_cfg(), dev).eval()
    kw = dict(resolution=64, bounding_box_min=(-1.0, -1.0, -1.0), bounding_box_max=(1.0, 1.0, 1.0), crop=32)
    vols = get_surface_sliding(model.field, return_volumes=True, **kw)
    assert len(vols) >= 1, "the geometric-init sphere (radius 0.5) crosses every crop of [-1, 1]^3"
    mesh = get_surface_sliding(model.field, **kw)
    vs, fs, ns, off = [], [], [], 0
    for lo, hi, vol in vols:
        spacing = tuple((hi[a] - lo[a]) / 31 for a in range(3))
        v, f, nrm, _ = OM.marching_cubes(_np(vol), 0.0, spacing=spacing)
        vs.append(v + np.array(lo))
        fs.append(f.astype(np.int64) + off)
        ns.append(nrm)
        off += len(v)
    assert np.array_equal(_np(mesh[0]), np.concatenate(vs))
    assert np.array_equal(_np(mesh[1]), np.concatenate(fs))
    assert np.array_equal(_np(mesh[2]), np.concatenate(ns))


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    globals()["case_" + sys.argv[1]]()
    print("CASE-OK", sys.argv[1])
