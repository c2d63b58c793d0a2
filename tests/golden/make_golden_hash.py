"""Mint tests/golden/hash_reference.npz by running the REFERENCE's own torch hash encoding (BUILD CONTAINER ONLY).

    python tests/golden/make_golden_hash.py

nerfstudio/field_components/encodings.py holds the only hash-grid arithmetic under /root/reference that is not tiny-cuda-nn:
``HashEncoding.hash_fn`` (:338-355, the Instant-NGP spatial hash with the primes 1 / 2654435761 / 805459861) and
``HashEncoding.pytorch_fwd`` (:357-398, trilinear blend of the 8 hashed corners, level-major output).  tiny-cuda-nn's GridEncoding
uses the same hash for its hashed levels, so these two functions pin - with vectors produced by reference code - the hashed
branch of oracle/hashgrid.py::corner_index (bit exact, integer work) and the corner order / trilinear weights / table and output
layout of its hashed levels.  What they do NOT pin (tcnn only; stays "parity unpinned"): the dense-level index and its modulo, the
align-to-8 level size, the +0.5 cell offset, the per-level scale formula and Smoothstep.

Vectors: for T in 2^17, 2^19, 2^22: 4096 integer corner triples (incl. large coordinates that overflow uint32 in the products)
and hash_fn's result per level; for one small all-hashed configuration: 512 positions, the reference table and pytorch_fwd's
output.  The consuming test is tests/test_cpu_hash_pin.py (runs live against the reference when present, always on the fixture)."""
import os
import sys

sys.dont_write_bytecode = True  # /root/reference is read-only: importing it must leave no __pycache__ there

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness  # noqa: E402


def reference_hash_vectors():
    ref_harness.import_reference()
    from nerfstudio.field_components.encodings import HashEncoding

    out = {}
    gen = torch.Generator().manual_seed(5)
    for log2_t in (17, 19, 22):
        enc = HashEncoding(num_levels=2, min_res=16, max_res=32, log2_hashmap_size=log2_t, implementation="torch")
        c = torch.randint(0, 4096, (4096, 2, 3), generator=gen, dtype=torch.int32)
        c[:64] = torch.randint(0, 2 ** 20, (64, 2, 3), generator=gen, dtype=torch.int32)  # products beyond 2^32 and 2^50
        c[64:72] = torch.tensor([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1], [2047, 2047, 2047], [2048, 0, 2048], [5, 7, 11]],
                                dtype=torch.int32)[:, None, :]
        h = enc.hash_fn(c)  # [4096, 2], level offsets (l * T) included
        out[f"hash{log2_t}/coords"] = c.numpy()
        out[f"hash{log2_t}/index"] = h.numpy().astype(np.int64)
    torch.manual_seed(9)
    enc = HashEncoding(num_levels=4, min_res=16, max_res=128, log2_hashmap_size=9, features_per_level=2, hash_init_scale=1.0,
                       implementation="torch")
    x = torch.rand(512, 3, generator=gen)
    # keep every position at least 1e-3 of a cell away from a cell boundary on every level: floor() must not depend on round-off
    for _ in range(20):
        fr = (x[:, None, :] * enc.scalings.view(-1, 1))
        bad = ((fr - fr.floor()).sub(0.5).abs() > 0.499).any(dim=-1).any(dim=-1)
        if not bool(bad.any()):
            break
        x[bad] = torch.rand(int(bad.sum()), 3, generator=gen)
    with torch.no_grad():
        y = enc.pytorch_fwd(x)
    out["fwd/x"] = x.numpy()
    out["fwd/table"] = enc.hash_table.detach().numpy()
    out["fwd/scalings"] = enc.scalings.numpy().astype(np.float64)
    out["fwd/out"] = y.numpy()
    out["fwd/log2_t"] = np.int64(9)
    out.update(reference_smoothstep_vectors(gen))
    out.update(reference_level_geometry())
    return out


def reference_smoothstep_vectors(gen):
    """Smoothstep, the way the reference's OWN code applies it: PeriodicVolumeEncoding.pytorch_fwd (encodings.py:689-733, `offset *
    offset * (3 - 2 offset)` at :700-701) evaluated on a table that is LINEAR in the cell coordinates along one axis and constant along
    the others - the trilinear blend of such a table returns floor(pos_axis) + s(frac_axis), i.e. the interpolation weight itself.  (The
    class's dense periodic index is x-major and its cells sit at pos = S x: neither is tcnn's; what this pins is the weight function.)"""
    from nerfstudio.field_components.encodings import PeriodicVolumeEncoding

    out = {}
    R = 16  # periodic volume resolution 2^(12 / 3); scalings floor(4 g^l) stay below it, nothing wraps
    x = torch.rand(2048, 3, generator=gen) * 0.98 + 0.01
    for smooth in (True, False):
        enc = PeriodicVolumeEncoding(num_levels=2, min_res=4, max_res=12, log2_hashmap_size=12, features_per_level=1, hash_init_scale=1.0,
                                     smoothstep=smooth)
        coords = torch.stack(torch.meshgrid(torch.arange(R), torch.arange(R), torch.arange(R), indexing="ij"), -1).reshape(-1, 3).float()
        ys = []
        for axis in range(3):
            with torch.no_grad():
                enc.hash_table.copy_(coords[:, axis].repeat(2)[:, None])  # both levels: entry (x, y, z) holds its `axis` coordinate
                ys.append(enc.pytorch_fwd(x))  # [P, 2 levels]
        out[f"smooth/{'on' if smooth else 'off'}"] = torch.stack(ys, dim=-1).numpy()  # [P, level, axis] = floor(pos) + weight
        out["smooth/scalings"] = enc.scalings.numpy().astype(np.float64)
    out["smooth/x"] = x.numpy()
    return out


def reference_level_geometry():
    """The per-level growth factor as the reference computes it wherever it builds a multi-resolution grid (HashEncoding, encodings.py:
    296-303; SDFField hands the same expression to tcnn as per_level_scale, fields/sdf_field.py:226-238): exp((ln max - ln min) / (L - 1)),
    for BASELINE configs 2 (16 levels 16 -> 2048) and 5 (16 levels 64 -> 4096) and the proposal grids (5 levels 16 -> 64 / 256)."""
    from nerfstudio.field_components.encodings import HashEncoding

    out = {}
    for name, (L, lo, hi) in {"config2": (16, 16, 2048), "config5": (16, 64, 4096), "prop0": (5, 16, 64), "prop1": (5, 16, 256)}.items():
        enc = HashEncoding(num_levels=L, min_res=lo, max_res=hi, log2_hashmap_size=4, implementation="torch")
        # scalings = floor(min_res * growth^level): recover growth^level before the floor from the class's own expression
        growth = np.exp((np.log(hi) - np.log(lo)) / (L - 1))
        assert torch.equal(enc.scalings, torch.floor(lo * growth ** torch.arange(L))), "the reference's formula moved"
        out[f"levels/{name}/scalings"] = enc.scalings.numpy().astype(np.float64)
        out[f"levels/{name}/args"] = np.array([L, lo, hi], np.int64)
    return out


if __name__ == "__main__":
    vec = reference_hash_vectors()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hash_reference.npz")
    np.savez_compressed(path, **vec)
    print("wrote", path, {k: v.shape for k, v in vec.items()})
