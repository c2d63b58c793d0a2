"""Bit-reproducibility of every MFMA kernel family (-m gpu): the test class a tolerance cannot replace.

Rounds 1 - 2 carried a race in the weight-chunk wait of the fused kernels (csrc/mlp_core.h): a gemm could start on
a weight chunk whose LDS-DMA was still landing.  It showed as a 5e-5 error in one call in five of ONE shape, far inside every parity bar.
What catches that class of defect is requiring the SAME BITS from repeated calls: the fused kernels, the layer-at-a-time 512-wide kernels
and the split-K weight-gradient GEMMs have no atomics and a fixed summation order, so forward outputs and MLP weight gradients must be
torch.equal across calls, whatever the allocator's free blocks hold (NaN, huge values, zeros) and whatever ran on the CUs before.
Short steps are the dangerous ones (the race needed a gemm to start before the previous DMA had landed: few k-blocks per gemm, shallow
networks), so the shapes here are the SMALL ones: 64-wide 8 + 4, 256-wide 1 + 1 and 2 + 2, the two ReLU background networks (inst_d /
inst_e), 2 x 512 layer by layer, and config 5's 1-layer geometry network on its 7 P points.  Hash-table and embedding gradients go through
atomics: compared to round-off.  Every test poisons the allocator between model builds (NaN / 1e30 / zero fills), SDFHIP_TEST_POISON style.
"""
import math

import pytest
import torch

from helpers import load_golden, product_model_from_params, small_oracle_cfg
from oracle import sdf_path as O
from test_gpu_parity import _bundle, _full_shape_params

pytestmark = pytest.mark.gpu

FILLS = (0.0, float("nan"), 1e30, float("nan"), -3.7)
REPS = 5


def _poison(device, fill):
    """Fill and release large and small blocks of the caching allocator: the next workspace is carved out of memory holding `fill`."""
    blocks = [torch.full((n,), fill, device=device) for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16) for _ in range(3)]
    del blocks


def _same_bits(name, runs, atomic=()):
    """runs: list of {tensor name: tensor}.  Everything must equal runs[0] bit for bit, except the names in `atomic` (accumulated with
    fp32 atomics: order-dependent round-off), which must agree to 1e-4 of the maximum; everything must be finite."""
    ref = runs[0]
    for k, v in ref.items():
        assert bool(torch.isfinite(v).all()), f"{name}: {k} is not finite"
    for i, r in enumerate(runs[1:], 1):
        assert set(r) == set(ref)
        for k, v in r.items():
            if any(a in k for a in atomic):
                assert (v - ref[k]).abs().max().item() <= 1e-4 * ref[k].abs().max().item() + 1e-12, f"{name}: {k}, run {i}"
            else:
                assert torch.equal(v, ref[k]), (f"{name}: {k} differs between identical calls (run {i} vs 0: {int((v != ref[k]).sum())} elements, "
                                                f"max {float((v - ref[k]).abs().max()):.3e})")


def _field_training_runs(make_model, o, d, cam, starts, device, near=0.5, far=4.5):
    n, s = starts.shape
    gen = torch.Generator().manual_seed(n * 131 + s)
    co = [torch.randn(n, s, generator=gen).to(device), (torch.randn(n, s, 3, generator=gen) * 0.3).to(device), torch.randn(n, s, 3, generator=gen).to(device)]
    runs = []
    for fill in FILLS:
        _poison(device, fill)
        model = make_model()
        rs = _bundle(o, d, cam, near, far, device).get_ray_samples(starts.to(device), starts.to(device) + 0.05)
        for _ in range(REPS):
            model.zero_grad()
            sdf, grad, rgb, _ = model.field.forward_fused(rs)
            ((sdf * co[0]).sum() + (grad * co[1]).sum() + (rgb * co[2]).sum()).backward()
            run = {"sdf": sdf.detach().clone(), "gradient": grad.detach().clone(), "rgb": rgb.detach().clone()}
            run.update({f"grad {k}": v.grad.detach().clone() for k, v in model.field.named_parameters() if v.grad is not None})
            runs.append(run)
    return runs


@pytest.mark.parametrize("shape", [(8, 4, 64), (1, 1, 256), (2, 2, 256), (8, 4, 256)], ids=lambda t: f"{t[0]}x{t[2]}+{t[1]}x{t[2]}")
def test_fused_training_step_is_bit_reproducible(device, shape):
    """geo_fwd (forward + chain), col_fwd, col_bwd, geo_bwd (tangent + backward), the weight-gradient GEMMs and their reductions: outputs
    and every MLP weight gradient from 25 identical training calls (5 allocator fillings x 5 repetitions) must be the same bits."""
    nl, nlc, width = shape
    if width == 64:
        cfg = small_oracle_cfg()
        base = cfg.field
        cfg.field = O.FieldCfg(**{**base.__dict__, "num_layers": nl, "num_layers_color": nlc})
    else:
        cfg = O.ModelCfg(field=O.FieldCfg(num_layers=nl, num_layers_color=nlc, bias=0.5, inside_outside=False, beta_init=0.3))
    p = _full_shape_params(cfg, seed=11 + nl)
    n, s = 37, 29  # 1073 points: a padded tail, 9 workgroups
    o, d, cam = O.synthetic_rays(n, seed=3)
    starts = torch.sort(torch.rand(n, s, generator=torch.Generator().manual_seed(4)) * 4.0 + 0.5, dim=-1)[0]
    runs = _field_training_runs(lambda: product_model_from_params(p, cfg, device).train(), o, d, cam, starts, device)
    assert len(runs) == len(FILLS) * REPS and len(runs[0]) >= 3 + 3 * (nl + 1) + 3 * (nlc + 1)
    _same_bits(f"{nl}x{width}+{nlc}x{width}", runs, atomic=("encoding.params", "embedding", "deviation_network"))


def test_hidden_512_layer_by_layer_is_bit_reproducible(device):
    """csrc/wide_kernels.h (neus-facto-bigmlp's width, one launch per layer and pass) at 2 x 512: short gemm chains, one DMA per k-block."""
    cfg = O.ModelCfg(field=O.FieldCfg(num_layers=2, hidden_dim=512, num_layers_color=4, bias=0.5, inside_outside=False, beta_init=0.3))
    p = _full_shape_params(cfg, seed=33)
    n, s = 33, 24
    o, d, cam = O.synthetic_rays(n, seed=5)
    starts = torch.sort(torch.rand(n, s, generator=torch.Generator().manual_seed(6)) * 4.0 + 0.5, dim=-1)[0]
    runs = _field_training_runs(lambda: product_model_from_params(p, cfg, device).train(), o, d, cam, starts, device)
    _same_bits("2x512", runs, atomic=("encoding.params", "embedding", "deviation_network"))


@pytest.mark.parametrize("kind", ["mlp", "grid"])
def test_background_fields_are_bit_reproducible(device, kind):
    """The ReLU instantiations of the fused kernels: NeRFField ("mlp": 8 x 256 + 2 x 128, inst_d.hip) and TCNNNerfactoField ("grid": two
    64-wide bias-free MLPs on a hash grid, inst_e.hip), forward + backward."""
    from sdfstudio_amd.fields.field_heads import FieldHeadNames as H
    from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField
    from sdfstudio_amd.fields.vanilla_nerf_field import NeRFEncoding, NeRFField
    from sdfstudio_amd.models.neus_facto import SceneContraction

    n, s = 29, 11
    o, d, cam = O.synthetic_rays(n, seed=4)
    cam = cam % 9
    gen = torch.Generator().manual_seed(8)
    starts = torch.sort(torch.rand(n, s, generator=gen) * 6.0 + 0.3, dim=-1).values
    ends = starts + torch.rand(n, s, generator=gen) * 0.4 + 0.01
    co = [torch.randn(n, s, generator=gen).to(device), torch.randn(n, s, 3, generator=gen).to(device)]
    state = None
    runs = []
    for fill in FILLS:
        _poison(device, fill)
        torch.manual_seed(11)
        if kind == "grid":
            fld = TCNNNerfactoField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=9, num_levels=5, max_res=48, log2_hashmap_size=9,
                                    spatial_distortion=SceneContraction(order=float("inf")))
        else:
            fld = NeRFField(position_encoding=NeRFEncoding(in_dim=3, num_frequencies=10, min_freq_exp=0.0, max_freq_exp=9.0, include_input=True),
                            direction_encoding=NeRFEncoding(in_dim=3, num_frequencies=4, min_freq_exp=0.0, max_freq_exp=3.0, include_input=True),
                            spatial_distortion=SceneContraction(order=float("inf")))
        if state is None:
            state = {k: v.clone() for k, v in fld.state_dict().items()}
            if kind == "grid":
                state["mlp_base.table"] = (torch.rand_like(state["mlp_base.table"]) * 2 - 1) * 0.4
        fld.load_state_dict(state)
        fld = fld.to(device).train()
        rs = _bundle(o, d, cam, 0.3, 7.0, device).get_ray_samples(starts.to(device), ends.to(device))
        for _ in range(REPS):
            fld.zero_grad()
            out = fld(rs)
            ((out[H.DENSITY][..., 0] * co[0]).sum() + (out[H.RGB] * co[1]).sum()).backward()
            run = {"density": out[H.DENSITY].detach().clone(), "rgb": out[H.RGB].detach().clone()}
            run.update({f"grad {k}": v.grad.detach().clone() for k, v in fld.named_parameters() if v.grad is not None})
            runs.append(run)
    assert len(runs[0]) >= (9 if kind == "grid" else 20)
    _same_bits(f"background {kind}", runs, atomic=("table", "embedding"))


def test_config5_numerical_gradient_step_is_bit_reproducible(device):
    """Config 5's field path (sdfhip_geo_forward_n / _backward_n on the 7 P points of the numerical-gradient branch: a ONE-hidden-layer
    geometry network = the shortest gemm chain in the library; 8-feature gather / scatter kernels; colour network on the finite-difference
    normal): sdf, the six taps, rgb and every MLP weight gradient, bit for bit across calls."""
    from helpers import angelo_oracle_cfg
    from sdfstudio_amd.fields.field_heads import FieldHeadNames as H

    cfg = angelo_oracle_cfg(12)
    p = _full_shape_params(cfg, seed=13)
    p = {k: v for k, v in p.items()}
    n, s = 31, 17
    o, d, cam = O.synthetic_rays(n, seed=7)
    gen = torch.Generator().manual_seed(9)
    starts = torch.sort(torch.rand(n, s, generator=gen) * 4.0 + 0.5, dim=-1)[0]
    co = [torch.randn(n, s, generator=gen).to(device), (torch.randn(n, s, 3, generator=gen) * 0.3).to(device), torch.randn(n, s, 3, generator=gen).to(device),
          (torch.randn(n, s, 6, generator=gen) * 0.1).to(device)]
    runs = []
    for fill in FILLS:
        _poison(device, fill)
        model = product_model_from_params(p, cfg, device, field_kwargs={"use_numerical_gradients": True}).train()
        fld = model.field
        fld.update_mask(12)
        fld.set_numerical_gradients_delta(4.0 / (fld.base_res * fld.growth_factor ** 11))
        rs = _bundle(o, d, cam, 0.5, 4.5, device).get_ray_samples(starts.to(device), starts.to(device) + 0.05)
        for _ in range(REPS):
            model.zero_grad()
            out = fld(rs)
            ((out[H.SDF][..., 0] * co[0]).sum() + (out[H.GRADIENT] * co[1]).sum() + (out[H.RGB] * co[2]).sum() + (out["sampled_sdf"] * co[3]).sum()).backward()
            run = {"sdf": out[H.SDF].detach().clone(), "taps": out["sampled_sdf"].detach().clone(), "rgb": out[H.RGB].detach().clone()}
            run.update({f"grad {k}": v.grad.detach().clone() for k, v in fld.named_parameters() if v.grad is not None})
            runs.append(run)
    assert len(runs[0]) >= 3 + 6 + 15
    _same_bits("config-5 field", runs, atomic=("encoding.params", "embedding"))
    assert math.isfinite(float(runs[0]["sdf"].abs().max()))


_WREDUCE_WORKER = r"""
import hashlib, json, sys, torch
sys.path.insert(0, sys.argv[1])
import bench as B
from sdfstudio_amd.cameras.rays import RayBundle
dev = torch.device("cuda", 0)
out = {}
for name, kw in (("small", dict(small=True)), ("config2_shape", dict(small=False, samples=32))):
    torch.manual_seed(0)
    model = B.build_model(dev, **kw)
    model.train()
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    centers, rot = B.synthetic_cameras(dev)
    o, d, norm, cam = B.draw_rays(centers, rot, 512, gen)
    res = model(RayBundle(origins=o, directions=d, directions_norm=norm, camera_indices=cam[:, None]))
    loss = sum(model.get_loss_dict(res, {"image": torch.rand(512, 3, device=dev, generator=gen)}).values())
    loss.backward()
    out[name] = {k: hashlib.sha256(p.grad.detach().cpu().numpy().tobytes()).hexdigest()[:16]
                 for k, p in sorted(model.field.named_parameters()) if p.grad is not None}
    out[name]["loss"] = float(loss)
print(json.dumps(out))
"""


@pytest.mark.gpu
def test_batched_weight_gradient_reduction_is_bit_identical_to_the_per_gemm_form():
    """wreduce_batch_kernel (ONE launch reduces the split-K partials of every weight-gradient GEMM of a backward call, each GEMM with a
    partial region of its own) against SDFHIP_WREDUCE_PER_GEMM=1 (rounds 1 - 6: one shared partial buffer, a reduction behind every GEMM).
    Same partial sums added in the same order: every weight and bias gradient of the two MLPs (through the weight-norm backward) must be
    the same bits - on the small golden network and on config 2's 8 x 256 + 4 x 256 network.  (The hash table's gradient is a sum of
    atomics and is left out.)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for mode, var in (("batched", None), ("per_gemm", "SDFHIP_WREDUCE_PER_GEMM")):
        env = dict(os.environ)
        env.pop("SDFHIP_WREDUCE_PER_GEMM", None)
        if var is not None:
            env[var] = "1"
        r = subprocess.run([sys.executable, "-c", _WREDUCE_WORKER, root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (mode, r.stderr[-3000:])
        got[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for name in got["batched"]:
        ref = got["per_gemm"][name]
        mlp = [k for k in ref if k.startswith(("glin", "clin"))]
        assert len(mlp) > 10, sorted(ref)
        diff = [k for k in mlp if got["batched"][name][k] != ref[k]]
        assert not diff, (name, diff)
        assert got["batched"][name]["loss"] == ref["loss"]
