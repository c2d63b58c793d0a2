"""The GPU test that held the pair-wave form against the oracle and the one-wave kernel while it was wired into the library (round 5, GPU call K:
2 passed).  Not collected by the suite: the product library does not contain the kernel (tools/experiments/README.md)."""
# flake8: noqa
def test_pair_wave_sdf_forward_against_oracle_and_the_one_wave_kernel(device, monkeypatch):
    """The sdf-only forward on two waves per SIMD (csrc/pair_kernels.h) at BASELINE config 2's network: the same hidden pre-activations
    as the one-wave kernel (identical MFMA sequences per out-block), the sdf row summed per role - so within fp32 round-off of the
    one-wave kernel's values, inside the 1e-5 north-star bar against the oracle, bit-identical from call to call whatever the free
    blocks hold, on a ragged point count (padded tail) and through both entry shapes (explicit points, ray layout)."""
    from sdfstudio_amd.utils.marching_cubes import sdf_on_grid, sdf_on_points

    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.5, inside_outside=False, beta_init=0.3))
    p = _full_shape_params(cfg)
    model = product_model_from_params(p, cfg, device).eval()
    gen = torch.Generator().manual_seed(11)
    pts = (torch.rand(3000, 3, generator=gen) * 2 - 1) * 0.95  # 3000: not a multiple of 128
    ref = O.geo_network(pts, p, cfg.field)[:, 0]
    lo, hi, res = (-0.9, -0.8, -0.7), (0.9, 0.85, 0.8), (12, 11, 37)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SDFHIP_PAIR_SDF", mode)  # read by the library at every launch
        out[mode] = (sdf_on_points(model.field, pts.to(device)).clone(), sdf_on_grid(model.field, lo, hi, res, chunk_points=1500).clone())
    assert_close("pair-wave sdf vs oracle", out["1"][0], ref, rtol=0, atol=1e-5)
    assert_close("one-wave sdf vs oracle", out["0"][0], ref, rtol=0, atol=1e-5)
    assert_close("pair-wave vs one-wave (points)", out["1"][0], out["0"][0], rtol=0, atol=1e-6)
    assert_close("pair-wave vs one-wave (grid)", out["1"][1], out["0"][1], rtol=0, atol=1e-6)
    # the two forms add the sdf row's 256 terms in different orders: bit-equal results over 3000 points would mean the switch did nothing
    assert not torch.equal(out["1"][0], out["0"][0]), "SDFHIP_PAIR_SDF had no effect: both runs took the same kernel"
    monkeypatch.setenv("SDFHIP_PAIR_SDF", "1")
    for fill in (float("nan"), 1e30, 0.0):
        blocks = [torch.full((n,), fill, device=device) for n in (1 << 24, 1 << 22, 1 << 20) for _ in range(2)]
        del blocks
        for _ in range(3):
            again = sdf_on_points(model.field, pts.to(device))
            assert torch.equal(again, out["1"][0]), f"pair-wave forward is not bit-reproducible: {int((again != out['1'][0]).sum())} elements differ"
