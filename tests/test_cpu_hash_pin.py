"""What the reference itself holds about the hash grid, used as a pin.

The hash-grid arithmetic of the hot path lives in tiny-cuda-nn, which is absent and unbuildable here ("parity unpinned",
oracle/hashgrid.py).  The reference's own torch fallback encoding uses the same spatial hash, so its ``hash_fn`` and ``pytorch_fwd``
(field_components/encodings.py:338-398) pin the HASHED branch of the oracle: the index bit for bit, and corner order / trilinear
weights / table layout / level-major output to fp32 round-off.  Vectors: tests/golden/hash_reference.npz, minted by
tests/golden/make_golden_hash.py from reference code; with /root/reference present the same comparison also runs live."""
import os

import numpy as np
import pytest
import torch

from oracle import hashgrid

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hash_reference.npz")


def _vectors(live):
    if live:
        from oracle import ref_harness

        if not ref_harness.reference_available():
            pytest.skip("reference tree not present (GPU box)")
        import importlib.util

        spec = importlib.util.spec_from_file_location("make_golden_hash", os.path.join(os.path.dirname(GOLDEN), "make_golden_hash.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.reference_hash_vectors()
    return dict(np.load(GOLDEN))


@pytest.mark.parametrize("live", [False, True], ids=["fixture", "live-reference"])
@pytest.mark.parametrize("log2_t", [17, 19, 22])
def test_hashed_corner_index_equals_reference_hash_fn(log2_t, live):
    """Integer work: bit exact.  corner_index carries tcnn's uint32 arithmetic in int64; the reference multiplies int32 coordinates
    by the primes in int64 and reduces modulo T = 2^k - the low k bits agree for every coordinate triple."""
    v = _vectors(live)
    c = torch.from_numpy(v[f"hash{log2_t}/coords"]).to(torch.int64)
    want = torch.from_numpy(v[f"hash{log2_t}/index"])
    T = 1 << log2_t
    for lvl in range(c.shape[1]):
        got = hashgrid.corner_index(c[:, lvl, 0], c[:, lvl, 1], c[:, lvl, 2], res=0, size=T, hashed=True) + lvl * T
        assert torch.equal(got, want[:, lvl]), f"T=2^{log2_t} level {lvl}: {(got != want[:, lvl]).sum().item()} indices differ"


@pytest.mark.parametrize("live", [False, True], ids=["fixture", "live-reference"])
def test_hashed_level_lookup_equals_reference_pytorch_fwd(live):
    """The oracle's hashed-level lookup against the reference's pytorch_fwd on the same table.  The two differ in where the cell
    sits (tcnn: pos = scale x + 0.5, the reference: pos = S x with S = floor(min_res g^l)), so each level is evaluated at the
    position x' that gives the oracle the reference's pos, x' = (S x - 0.5) / scale, in fp64: everything after the position - cell
    corners and their order, the hash, the trilinear weights, the [entries, F] table with level offsets l T, the level-major
    [L F] output - must then agree to fp32 round-off of the reference's own blend."""
    v = _vectors(live)
    x = torch.from_numpy(v["fwd/x"]).double()
    table = torch.from_numpy(v["fwd/table"]).double()
    S = v["fwd/scalings"]
    want = torch.from_numpy(v["fwd/out"]).double()
    T = 1 << int(v["fwd/log2_t"])
    L, F = len(S), table.shape[1]
    outs = []
    for lvl in range(L):
        scale = float(S[lvl]) - 1.0  # any positive scale works: the position is mapped through it and back
        lv = hashgrid.GridLevels(n_levels=1, n_features=F, log2_hashmap_size=int(v["fwd/log2_t"]), base_resolution=0, per_level_scale=1.0,
                                 smoothstep=False, scale=np.array([scale], np.float32), resolution=np.array([int(S[lvl]) + 1]),
                                 size=np.array([T]), offset=np.array([0, T]), hashed=np.array([True]))
        xp = (x * float(S[lvl]) - 0.5) / float(np.float32(scale))
        outs.append(hashgrid.grid_encode(xp, table[lvl * T:(lvl + 1) * T], lv))
    got = torch.cat(outs, dim=-1)
    assert got.shape == want.shape == (x.shape[0], L * F)
    err = (got - want).abs().max().item()
    assert err <= 2e-6 * want.abs().max().item() + 1e-7, err
