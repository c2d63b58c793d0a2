"""Shared test plumbing: golden loading, oracle <-> product parameter mapping, error reporting."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle import sdf_path as O  # noqa: E402


def load_golden_file(name: str):
    z = np.load(os.path.join(GOLDEN, name))
    out = {"in": {}, "param": {}, "out": {}, "loss": {}, "grad": {}}
    for k in z.files:
        head, key = k.split("/", 1) if "/" in k else ("misc", k)
        out.setdefault(head, {})[key] = torch.from_numpy(z[k])
    return out


def load_golden(mode: str):
    return load_golden_file(f"neus_facto_small_{mode}.npz")


def small_oracle_cfg() -> O.ModelCfg:
    f = O.FieldCfg(
        num_layers=8, hidden_dim=64, geo_feat_dim=64, num_layers_color=4, hidden_dim_color=64, bias=0.5,
        inside_outside=False, use_grid_feature=True, beta_init=0.3, num_levels=8, max_res=128, base_res=4,
        log2_hashmap_size=11, hash_features_per_level=2, hash_smoothstep=True,
    )
    props = (
        O.ProposalCfg(hidden_dim=16, num_levels=5, max_res=32, base_res=4, log2_hashmap_size=9),
        O.ProposalCfg(hidden_dim=16, num_levels=5, max_res=64, base_res=4, log2_hashmap_size=9),
    )
    return O.ModelCfg(field=f, proposals=props, num_proposal_samples=(32, 24), num_neus_samples=16)


ELEM_FLOOR = 1e-2   # element-wise gate: elements with |ref| > ELEM_FLOOR * max|ref| ...
ELEM_FACTOR = 10.0  # ... must be within ELEM_FACTOR * rtol RELATIVE TO THEMSELVES (10x tighter than the max-relative bar implies there)


def report(name, got, ref, rtol, atol, elem_rtol=None):
    """Returns (ok, message) comparing got with ref under two bars:
      1. max-relative:   |d| <= atol + rtol * max|ref|                       for every element;
      2. element-wise:   |d| <= atol + elem_rtol * |ref_i|                   for every element with |ref_i| > 1e-2 max|ref|
         (elem_rtol defaults to 10 * rtol, i.e. 1e-3 for the forward heads compared at 1e-4 and 1e-2 for parameter gradients
         compared at 1e-3; no element-wise gate for pure absolute comparisons, rtol == 0).
    The second bar is what keeps "relative to the tensor maximum" from hiding a wrong mid-sized element."""
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    if got.numel() == 0:
        return True, f"{name}: empty"
    d = (got - ref).abs()
    scale = ref.abs().max().item()
    err = d.max().item()
    idx = int(d.argmax())
    ok = bool(torch.isfinite(got).all()) and err <= atol + rtol * scale
    if elem_rtol is None:
        elem_rtol = ELEM_FACTOR * rtol if rtol > 0 else None
    big = ref.abs() > ELEM_FLOOR * scale
    elem, elem_ok = 0.0, True
    if bool(big.any()):
        elem = (d[big] / ref.abs()[big]).max().item()
        if elem_rtol is not None and not os.environ.get("SDFHIP_TEST_NO_ELEM_GATE"):
            elem_ok = bool((d[big] <= atol + elem_rtol * ref.abs()[big]).all())
    gate = "no element-wise gate" if elem_rtol is None else f"element-wise gate {elem_rtol:.1e}{'' if elem_ok else ' EXCEEDED'}"
    return ok and elem_ok, (f"{name}: max|d|={err:.3e} (scale {scale:.3e}, rel-to-max {err / (scale + 1e-30):.2e}, worst element-wise rel "
                            f"{elem:.2e} over |ref| > 1e-2 scale; {gate}) at flat {idx}: "
                            f"got {got.reshape(-1)[idx].item():.8g} ref {ref.reshape(-1)[idx].item():.8g}; tol {atol + rtol * scale:.2e}")


def assert_close(name, got, ref, rtol=1e-4, atol=1e-6, elem_rtol=None):
    ok, msg = report(name, got, ref, rtol, atol, elem_rtol)
    print(("PASS " if ok else "FAIL ") + msg)
    if os.environ.get("SDFHIP_TEST_LOG"):  # every comparison of a run in one file (with KEEP_GOING: a census of what would fail)
        with open(os.environ["SDFHIP_TEST_LOG"], "a") as fh:
            fh.write(("PASS " if ok else "FAIL ") + os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0] + " :: " + msg + "\n")
    if os.environ.get("SDFHIP_TEST_KEEP_GOING"):  # debugging aid: print every comparison of a failing test
        return
    assert ok, msg


def assert_fp32_class(name, got, ref32, truth64, factor=3.0, atol=1e-6):
    """`got` (HIP, fp32) must be as close to the fp64 evaluation of the oracle as the oracle's own fp32 evaluation is,
    up to `factor`: max|got - truth| <= factor * max|ref32 - truth| + atol.  Used where fp32 round-off of the
    reference path itself exceeds a fixed tolerance (fine hash levels at scale 2e3 amplify one ulp of the position)."""
    got = got.detach().double().cpu()
    ref32 = ref32.detach().double().cpu()
    truth = truth64.detach().double().cpu()
    assert got.shape == truth.shape == ref32.shape, f"{name}: shapes {tuple(got.shape)} {tuple(ref32.shape)} {tuple(truth.shape)}"
    e_got = (got - truth).abs().max().item()
    e_ref = (ref32 - truth).abs().max().item()
    ok = bool(torch.isfinite(got).all()) and e_got <= factor * e_ref + atol
    msg = (f"{name}: |hip - fp64| = {e_got:.3e}, |oracle_fp32 - fp64| = {e_ref:.3e} (scale {truth.abs().max().item():.3e}); "
           f"bound {factor * e_ref + atol:.3e}")
    print(("PASS " if ok else "FAIL ") + msg)
    assert ok, msg


def relu_flip_basis(run_backward, margin=2e-6, max_flips=48):
    """Gradient changes caused by taking the other ReLU branch at every knife-edge pre-activation of the colour network.

    run_backward() -> {name: grad} evaluates the ORACLE (any dtype) on the test's inputs.  Pre-activations with |z| < margin
    are the (point, unit) pairs where fp32 implementations with different summation order may pick different branches;
    for each of them the oracle is re-run with that single branch flipped.  Returns (base grads, [delta grads per flip])."""
    rec = {}
    with O.relu_hook(record=rec):
        base = {k: v.detach().clone() for k, v in run_backward().items()}
    edges = [(l, i) for l, z in sorted(rec.items()) for i in torch.nonzero(z.abs() < margin).tolist()]
    assert len(edges) <= max_flips, f"{len(edges)} knife-edge ReLU pre-activations: shrink the case or the margin"
    basis = []
    for l, (row, col) in edges:
        m = torch.zeros_like(rec[l], dtype=torch.bool)
        m[row, col] = True
        with O.relu_hook(flip={l: m}):
            g = run_backward()
        basis.append({k: g[k].detach() - base[k] for k in base})
        print(f"knife-edge ReLU: colour layer {l}, point {row}, unit {col}, z = {rec[l][row, col].item():+.2e}")
    return base, basis


def assert_grads_close_mod_relu_flips(got, ref, basis, rtol, atol=1e-8):
    """Every gradient tensor must match `ref` within rtol * max|ref| once the ReLU branch choices at the knife-edge
    pre-activations (relu_flip_basis) are allowed to differ: got - ref = sum_k c_k basis_k with c_k in {-1, 0, 1}."""
    keys = [k for k in ref if k in got]
    scale = {k: ref[k].abs().max().item() + 1e-30 for k in keys}
    diff = torch.cat([((got[k].detach().cpu().double() - ref[k].double()) / scale[k]).flatten() for k in keys])
    coef = []
    if basis:
        B = torch.stack([torch.cat([(b[k].double() / scale[k]).flatten() for k in keys]) for b in basis], dim=1)
        c = torch.linalg.lstsq(B, diff[:, None]).solution[:, 0]
        coef = c.round().clamp(-1, 1)
        diff = diff - B @ coef
        print("ReLU branch differences absorbed:", [int(v) for v in coef.tolist()])
    off, bad = 0, []
    for k in keys:
        n = ref[k].numel()
        err = diff[off:off + n].abs().max().item()
        off += n
        ok = err <= rtol + atol / scale[k]
        print(("PASS " if ok else "FAIL ") + f"grad {k}: max|d| / max|ref| = {err:.3e} (tol {rtol:.1e})")
        if not ok:
            bad.append(k)
    assert not bad, f"gradients differ beyond ReLU branch ambiguity: {bad}"
    return coef


def to_double(params):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}


def product_model_from_params(params, cfg: O.ModelCfg, device, field_kwargs=None):
    """Build sdfstudio_amd's NeuSFactoModel with the given (oracle-named) parameters."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import NeuSFactoModel, NeuSFactoModelConfig, SceneBox

    fc = cfg.field
    fcfg = SDFFieldConfig(
        num_layers=fc.num_layers, hidden_dim=fc.hidden_dim, geo_feat_dim=fc.geo_feat_dim,
        num_layers_color=fc.num_layers_color, hidden_dim_color=fc.hidden_dim_color, bias=fc.bias,
        inside_outside=fc.inside_outside, use_grid_feature=True, beta_init=fc.beta_init, num_levels=fc.num_levels,
        max_res=fc.max_res, base_res=fc.base_res, log2_hashmap_size=fc.log2_hashmap_size,
        hash_features_per_level=fc.hash_features_per_level, hash_smoothstep=fc.hash_smoothstep,
        use_appearance_embedding=fc.use_appearance_embedding, use_position_encoding=fc.use_position_encoding,
        **(field_kwargs or {}),
    )
    mcfg = NeuSFactoModelConfig(
        sdf_field=fcfg, num_proposal_samples_per_ray=tuple(cfg.num_proposal_samples),
        num_neus_samples_per_ray=cfg.num_neus_samples, num_proposal_iterations=len(cfg.proposals),
        proposal_net_args_list=[
            {"hidden_dim": p.hidden_dim, "log2_hashmap_size": p.log2_hashmap_size, "num_levels": p.num_levels,
             "max_res": p.max_res, "base_res": p.base_res} for p in cfg.proposals
        ],
        eikonal_loss_mult=cfg.eikonal_loss_mult, interlevel_loss_mult=cfg.interlevel_loss_mult,
    )
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=cfg.near, far=cfg.far)
    model = NeuSFactoModel(mcfg, box, num_train_data=49)
    load_params(model, params)
    return model.to(device)


def load_params(model, params):
    sd = model.state_dict()
    for k, v in params.items():
        if k.startswith("proposal_networks."):
            i, name = k.split(".")[1:3]
            key = f"proposal_networks.{i}.mlp_base.{name}"
        else:
            key = f"field.{k}"
        assert key in sd, f"{key} missing from the product model"
        assert tuple(sd[key].shape) == tuple(v.shape), (key, tuple(sd[key].shape), tuple(v.shape))
        sd[key] = v.clone()
    model.load_state_dict(sd)


def product_grads(model):
    """Gradients keyed by oracle names."""
    out = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        if k.startswith("field."):
            out[k[len("field."):]] = p.grad.detach()
        elif k.startswith("proposal_networks."):
            i = k.split(".")[1]
            out[f"proposal_networks.{i}.{k.split('.')[-1]}"] = p.grad.detach()
    return out
