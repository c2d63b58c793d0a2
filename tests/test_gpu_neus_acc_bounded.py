"""NeuS-acc's packed path with BOUNDED arrays (-m gpu): NeuSAccSampler(bounded=True) sizes the packed arrays by a bound,
keeps the march step on the device and reads nothing back inside a step - and must compute the training step the exact-size form
computes: same samples, same rendered heads, same losses, same parameter gradients (up to the summation order of the split-K weight
gradients, which depends on the number of point tiles).  What the reference does: nerfacc returns exact-size tensors, one device -> host
read per step (model_components/ray_samplers.py:1379-1382, 1474-1484; models/neus_acc.py:88-148)."""
import pytest
import torch
import torch.nn.functional as F

from helpers import load_params, small_oracle_cfg
from oracle import sdf_path as O
from test_gpu_parity import _bundle

pytestmark = pytest.mark.gpu


def _model(device, bounded):
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_acc import NeuSAccModel, NeuSAccModelConfig
    from sdfstudio_amd.models.neus_facto import SceneBox

    cfg = small_oracle_cfg()
    fc = cfg.field
    params = O.init_field_params(fc, num_images=49, seed=3)
    g = torch.Generator().manual_seed(5)
    for k in list(params):  # perturbed: the geometric init zeroes the columns over the grid features (their table gradient would be 0)
        if k.endswith("weight_v"):
            params[k] = params[k] + 0.02 * torch.randn(params[k].shape, generator=g)
    params["deviation_network.variance"] = torch.tensor([0.5])
    fcfg = SDFFieldConfig(num_layers=fc.num_layers, hidden_dim=fc.hidden_dim, geo_feat_dim=fc.geo_feat_dim, num_layers_color=fc.num_layers_color,
                          hidden_dim_color=fc.hidden_dim_color, bias=fc.bias, inside_outside=fc.inside_outside, use_grid_feature=True,
                          beta_init=fc.beta_init, num_levels=fc.num_levels, max_res=fc.max_res, base_res=fc.base_res,
                          log2_hashmap_size=fc.log2_hashmap_size, hash_features_per_level=fc.hash_features_per_level,
                          hash_smoothstep=fc.hash_smoothstep)
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=cfg.near, far=cfg.far)
    model = NeuSAccModel(NeuSAccModelConfig(sdf_field=fcfg, num_samples=16, num_samples_importance=16, num_up_sample_steps=2, background_model="none"), box, 49)
    load_params(model, params)
    model = model.to(device).train()
    model.sampler.bounded = bounded
    model.before_train_iteration(2000)
    model.after_train_iteration(2000)  # first occupancy-grid update: the packed path runs from here on
    return model, cfg


def _step(model, cfg, rb_args, image, device, step):
    model.before_train_iteration(step)
    out = model(_bundle(*rb_args, cfg.near, cfg.far, device))
    losses = model.get_loss_dict(out, {"image": image})
    model.zero_grad()
    sum(losses.values()).backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    return out, {k: v.detach() for k, v in losses.items()}, grads


def test_bounded_packed_arrays_compute_the_exact_forms_training_step(device):
    n = 256
    o, d, cam = O.synthetic_rays(n, seed=13)
    d = F.normalize(d + 0.1 * torch.randn(n, 3), dim=-1)
    image = torch.rand(n, 3)
    exact, cfg = _model(device, bounded=False)
    bnd, _ = _model(device, bounded=True)
    assert torch.equal(exact.sampler._binary, bnd.sampler._binary)
    # step A: the bounded sampler's first packed call is an exact one (it measures the bound); step B runs on the bound, no host read
    for step in (2001, 2002):
        oe, le, ge = _step(exact, cfg, (o, d, cam), image, device, step)
        ob, lb, gb = _step(bnd, cfg, (o, d, cam), image, device, step)
        total = oe["ray_indices"].shape[0]
        if step == 2001:
            assert bnd.sampler.packed_valid is None and bnd.sampler._cap is not None and bnd.sampler._cap >= total
            assert ob["ray_indices"].shape[0] == total
            continue
        cap = bnd.sampler._cap
        assert ob["ray_indices"].shape[0] == cap > total and int(bnd.sampler.packed_valid) == total
        assert bnd.sampler._step_dev is not None  # the march step never left the device
        assert float(bnd.sampler._step_dev) == pytest.approx(exact.sampler.step_size, rel=1e-6)
        # the same samples in the same places, a filler behind them
        assert torch.equal(ob["ray_indices"][:total], oe["ray_indices"])
        assert torch.equal(ob["ray_samples"].flat_starts[:total], oe["ray_samples"].flat_starts)
        assert torch.equal(bnd.sampler.packed_info, exact.sampler.packed_info) and torch.equal(bnd.sampler.packed_counts, exact.sampler.packed_counts)
        assert float(ob["packed_weights"][total:].abs().max()) == 0.0
        for k in ("rgb", "depth", "normal", "accumulation"):
            assert torch.allclose(ob[k], oe[k], rtol=1e-6, atol=1e-7), k
        for k in le:
            assert float(lb[k]) == pytest.approx(float(le[k]), rel=2e-6, abs=1e-8), k
        assert set(gb) == set(ge)
        for k in ge:
            scale = float(ge[k].abs().max())
            assert float((gb[k] - ge[k]).abs().max()) <= 2e-5 * scale + 1e-9, (k, float((gb[k] - ge[k]).abs().max()), scale)
    # the bound is re-checked with ONE read for many steps; nothing overflowed
    bnd.sampler.check_capacity()
    assert bnd.sampler.overflowed_steps == 0
    # a bound that is too small is detected (and widened) at the next check; the step itself stays finite
    bnd.sampler._cap = 1024 if total > 1024 else max(64, total // 2)
    small = bnd.sampler._cap
    ob, lb, gb = _step(bnd, cfg, (o, d, cam), image, device, 2003)
    assert all(torch.isfinite(v).all() for v in gb.values())
    bnd.sampler.check_capacity()
    if total > small:
        assert bnd.sampler.overflowed_steps == 1 and bnd.sampler._cap > total
