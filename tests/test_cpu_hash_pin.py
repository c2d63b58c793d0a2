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


@pytest.mark.parametrize("live", [False, True], ids=["fixture", "live-reference"])
def test_smoothstep_weight_equals_reference_periodic_volume_encoding(live):
    """Smoothstep is applied by reference code too - PeriodicVolumeEncoding.pytorch_fwd (field_components/encodings.py:
    700-701).  On a table that is linear in the cell coordinate along one axis its trilinear blend returns floor(pos) + weight(frac(pos)):
    the interpolation weight itself.  oracle.hashgrid.level_cell must produce the same weight for the same fractional position, with
    Smoothstep on and off (the kernels' cell arithmetic is held to the oracle's by tests/test_gpu_hash_pin.py and every field test)."""
    v = _vectors(live)
    x = torch.from_numpy(v["smooth/x"]).double()
    S = v["smooth/scalings"]
    for name, smooth in (("on", True), ("off", False)):
        want = torch.from_numpy(v[f"smooth/{name}"]).double()  # [P, level, axis] = floor(S x) + w
        for lvl in range(len(S)):
            pos = x * float(S[lvl])
            w_ref = want[:, lvl, :] - pos.floor()
            # the oracle at the position x' that gives it the same pos = scale x' + 0.5 (any positive scale: mapped through and back)
            scale = float(S[lvl]) - 1.0
            lv = hashgrid.GridLevels(n_levels=1, n_features=1, log2_hashmap_size=12, base_resolution=0, per_level_scale=1.0, smoothstep=smooth,
                                     scale=np.array([scale], np.float32), resolution=np.array([int(S[lvl]) + 1]), size=np.array([4096]),
                                     offset=np.array([0, 4096]), hashed=np.array([False]))
            xp = (pos - 0.5) / float(np.float32(scale))
            _, w = hashgrid.level_cell(xp, lv, 0)
            err = (w - w_ref).abs().max().item()
            assert err <= 5e-6, f"smoothstep {name}, level {lvl}: weight differs from the reference's by {err:.2e}"
            if smooth:  # and it IS w^2 (3 - 2 w) of the linear weight, not the identity
                fr = pos - pos.floor()
                assert (w_ref - fr * fr * (3 - 2 * fr)).abs().max().item() <= 5e-6 and (w_ref - fr).abs().max().item() > 0.05


@pytest.mark.parametrize("live", [False, True], ids=["fixture", "live-reference"])
def test_level_growth_factor_equals_reference_formula(live):
    """The per-level scale of oracle.hashgrid.make_levels (and, bit for bit, of the library: test_level_table_of_the_library_equals_the_
    oracles) grows by the factor the reference computes: exp((ln max_res - ln min_res) / (L - 1)) (encodings.py:296-303, sdf_field.py:226),
    for the grids of BASELINE configs 2 and 5 and the proposal networks.  tcnn's scale_l = base g^l - 1 and the reference's own
    scalings_l = floor(base g^l) are different conventions of the same geometry: floor(scale_l + 1) must reproduce the reference's
    integer resolutions except where g^l is an integer and the two fp roundings straddle it."""
    from oracle import sdf_path as O

    v = _vectors(live)
    for name in ("config2", "config5", "prop0", "prop1"):
        L, lo, hi = (int(t) for t in v[f"levels/{name}/args"])
        ref = v[f"levels/{name}/scalings"]
        g = O.FieldCfg(num_levels=L, base_res=lo, max_res=hi).growth_factor()
        assert g == float(np.exp((np.log(hi) - np.log(lo)) / (L - 1)))  # the same double, not merely close
        lv = hashgrid.make_levels(L, 2, 19, lo, g, False)
        exact = lo * g ** np.arange(L)
        # tcnn receives per_level_scale as a FLOAT: g rounded to fp32 (6e-8) compounds to (L - 1) x 6e-8 at the finest level
        assert np.allclose(lv.scale.astype(np.float64) + 1.0, exact, rtol=(L - 1) * 1.2e-7, atol=0), name
        near_int = np.abs(exact - np.round(exact)) < 4e-6 * exact
        assert np.array_equal(np.floor(lv.scale.astype(np.float64) + 1.0)[~near_int], ref[~near_int]), name
        assert np.allclose(np.round(exact)[near_int], np.round(lv.scale.astype(np.float64) + 1.0)[near_int]), name
        assert lv.scale[0] == lo - 1 and abs(float(lv.scale[-1]) + 1.0 - hi) <= (L - 1) * 1.2e-7 * hi
