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
    if os.environ.get("SDFHIP_TEST_LOG"):
        with open(os.environ["SDFHIP_TEST_LOG"], "a") as fh:
            fh.write(("PASS " if ok else "FAIL ") + os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0] + " :: " + msg
                     + f" ratio {e_got / max(e_ref, 1e-300):.2f}\n")
    if os.environ.get("SDFHIP_TEST_KEEP_GOING"):
        return
    assert ok, msg


def relu_flip_basis(run_backward, margin=2e-6, max_flips=48, curv_margin=None):
    """Gradient changes caused by taking the other ReLU branch at every knife-edge pre-activation of the colour network (and, with
    curv_margin, the other branch of |c| at every curvature-loss element below that margin: oracle key -1, neus_facto.py:312-325 -
    the second difference divides the sdf's round-off by delta^2).

    run_backward() -> {name: grad} evaluates the ORACLE (any dtype) on the test's inputs.  Pre-activations with |z| < margin
    are the (point, unit) pairs where fp32 implementations with different summation order may pick different branches;
    for each of them the oracle is re-run with that single branch flipped.  Returns (base grads, [delta grads per flip])."""
    rec = {}
    with O.relu_hook(record=rec):
        run_backward()
    # The oracle's own fp32 evaluation is not reproducible at the last ulp from run to run (multi-threaded reductions), so every
    # evaluation below runs under an IMPOSED branch pattern: the recorded one, with a single knife-edge element toggled per flip.
    pattern = {l: z > 0 for l, z in rec.items()}
    with O.relu_hook(force=pattern):
        base = {k: v.detach().clone() for k, v in run_backward().items()}
    edges = []
    for l, z in sorted(rec.items()):
        m = curv_margin if l == -1 else margin
        if m is not None:
            edges += [(l, i) for i in torch.nonzero(z.abs() < m).tolist()]
    assert len(edges) <= max_flips, f"{len(edges)} knife-edge pre-activations: shrink the case or the margin"
    basis = []
    for l, (row, col) in edges:
        forced = dict(pattern)
        forced[l] = pattern[l].clone()
        forced[l][row, col] = ~forced[l][row, col]
        with O.relu_hook(force=forced):
            g = run_backward()
        basis.append({k: g[k].detach() - base[k] for k in base})
        what = "curvature |c|" if l == -1 else f"ReLU: colour layer {l}"
        print(f"knife edge, {what}, point {row}, unit {col}, value = {rec[l][row, col].item():+.2e}")
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
        # a flip that moves no element by a quarter of the tolerance cannot decide the comparison: left out of the fit (such columns
        # are mostly the oracle's own run-to-run round-off and make the least-squares problem rank deficient)
        live = B.abs().amax(dim=0) > 0.25 * rtol
        coef = torch.zeros(B.shape[1], dtype=B.dtype)
        if bool(live.any()):
            c = torch.linalg.lstsq(B[:, live], diff[:, None]).solution[:, 0]
            coef[live] = c.round().clamp(-1, 1)
        diff = diff - B @ coef
        print("ReLU branch differences absorbed:", [int(v) for v in coef.tolist()], f"({int(live.sum())} of {B.shape[1]} flips matter at this tolerance)")
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
        eikonal_loss_mult=cfg.eikonal_loss_mult, interlevel_loss_mult=cfg.interlevel_loss_mult, background_model="none",
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


# ------------------------------------------------------------------------------------------------ BASELINE config 5 (neus-facto-angelo)
ANGELO_GOLDEN_PROPS = [{"hidden_dim": 16, "log2_hashmap_size": 9, "num_levels": 5, "max_res": 32, "base_res": 4},
                       {"hidden_dim": 16, "log2_hashmap_size": 9, "num_levels": 5, "max_res": 64, "base_res": 4}]
ANGELO_GOLDEN_BG = dict(num_levels=6, max_res=64, log2_hashmap_size=10)  # the background field's table, shrunk through its own ctor arguments
ANGELO_GOLDEN_LOG2_T = 10
ANGELO_GOLDEN_N_FIELD = 12
ANGELO_PRESET_PROPS = [{"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 64, "base_res": 16},
                       {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256, "base_res": 16}]  # neus_facto.py:59-64
ANGELO_PRESET_BG = dict(num_levels=16, max_res=1024, log2_hashmap_size=19)  # TCNNNerfactoField's defaults (base_surface_model.py:181-187)


def angelo_oracle_cfg(log2_t=ANGELO_GOLDEN_LOG2_T, props=None, num_proposal_samples=(32, 24), n_field=ANGELO_GOLDEN_N_FIELD) -> O.ModelCfg:
    """The neus-facto-angelo preset (configs/method_configs.py:381-450) as an oracle configuration: 16 x 8 linear grid 64 -> 4096, one
    256-wide hidden layer without positional encoding, appearance embedding, near 0.01 / far 1000, eikonal 0.01."""
    props = ANGELO_GOLDEN_PROPS if props is None else props
    f = O.FieldCfg(num_layers=1, hidden_dim=256, geo_feat_dim=256, num_layers_color=4, hidden_dim_color=256, bias=0.5, inside_outside=False,
                   beta_init=0.3, use_appearance_embedding=True, use_position_encoding=False, num_levels=16, max_res=4096, base_res=64,
                   log2_hashmap_size=log2_t, hash_features_per_level=8, hash_smoothstep=False)
    pc = tuple(O.ProposalCfg(hidden_dim=a["hidden_dim"], num_levels=a["num_levels"], max_res=a["max_res"], base_res=a["base_res"],
                             log2_hashmap_size=a["log2_hashmap_size"]) for a in props)
    return O.ModelCfg(field=f, proposals=pc, num_proposal_samples=tuple(num_proposal_samples), num_neus_samples=n_field, eikonal_loss_mult=0.01,
                      near=0.01, far=1000.0)


def angelo_bg_levels(num_levels=ANGELO_GOLDEN_BG["num_levels"], max_res=ANGELO_GOLDEN_BG["max_res"],
                     log2_hashmap_size=ANGELO_GOLDEN_BG["log2_hashmap_size"]):
    """TCNNNerfactoField's grid (fields/nerfacto_field.py:99-127: base_res 16, 2 features per level, linear interpolation)."""
    from oracle import hashgrid

    growth = float(np.exp((np.log(max_res) - np.log(16)) / (num_levels - 1)))
    return hashgrid.make_levels(num_levels, 2, log2_hashmap_size, 16, growth, False)


def oracle_params_from_reference_state(state_dict):
    """Reference model.state_dict() (or a dict of its gradients) -> the oracle's flat parameter dict: field.* without the prefix; the
    tinycudann shim keeps hash tables as mlp_base.encoding.params, the oracle as proposal_networks.<i>.table /
    field_background.mlp_base.table."""
    p = {}
    for k, v in state_dict.items():
        if k == "device_indicator_param" or k.endswith(".aabb"):
            continue
        if k.startswith("field."):
            p[k[len("field."):]] = v.clone()
        elif k.startswith("proposal_networks."):
            i, name = k.split(".")[1], k.split(".")[-1]
            p[f"proposal_networks.{i}.{'table' if name == 'params' else name}"] = v.clone()
        elif k.startswith("field_background."):
            p[k.replace("mlp_base.encoding.params", "mlp_base.table")] = v.clone()
        else:
            raise KeyError(k)
    return p


def angelo_product_model(oracle_params, cfg: O.ModelCfg, device, bg=None):
    """sdfstudio_amd's NeuSFactoModel configured as the neus-facto-angelo preset (bench.py::build_model_config5, with the table sizes
    of `cfg` / `bg`), carrying the given oracle-named parameters (field.*, proposal_networks.<i>.*, field_background.*)."""
    import functools

    from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models import background as BGM
    from sdfstudio_amd.models.neus_facto import NeuSFactoModel, NeuSFactoModelConfig, SceneBox

    fc = cfg.field
    bg = ANGELO_GOLDEN_BG if bg is None else bg
    fcfg = SDFFieldConfig(use_grid_feature=True, num_layers=fc.num_layers, num_layers_color=fc.num_layers_color, hidden_dim=fc.hidden_dim,
                          hidden_dim_color=fc.hidden_dim_color, geometric_init=True, bias=fc.bias, beta_init=fc.beta_init,
                          inside_outside=fc.inside_outside, use_appearance_embedding=True, use_numerical_gradients=True, base_res=fc.base_res,
                          max_res=fc.max_res, log2_hashmap_size=fc.log2_hashmap_size, hash_features_per_level=fc.hash_features_per_level,
                          hash_smoothstep=fc.hash_smoothstep, use_position_encoding=False)
    mcfg = NeuSFactoModelConfig(
        near_plane=cfg.near, far_plane=cfg.far, overwrite_near_far_plane=True, sdf_field=fcfg, background_model="grid", level_init=8,
        eikonal_loss_mult=cfg.eikonal_loss_mult, use_anneal_beta=True, enable_progressive_hash_encoding=True,
        enable_numerical_gradients_schedule=True, enable_curvature_loss_schedule=True, curvature_loss_multi=5e-4,
        num_proposal_samples_per_ray=tuple(cfg.num_proposal_samples), num_neus_samples_per_ray=cfg.num_neus_samples,
        proposal_net_args_list=[{"hidden_dim": p.hidden_dim, "log2_hashmap_size": p.log2_hashmap_size, "num_levels": p.num_levels,
                                 "max_res": p.max_res, "base_res": p.base_res} for p in cfg.proposals])
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
    full = BGM.TCNNNerfactoField
    BGM.TCNNNerfactoField = functools.partial(TCNNNerfactoField, **bg)
    try:
        model = NeuSFactoModel(mcfg, box, num_train_data=49)
    finally:
        BGM.TCNNNerfactoField = full
    sd = model.state_dict()
    seen = set()
    for k, v in oracle_params.items():
        if k.startswith("proposal_networks."):
            i, name = k.split(".")[1:3]
            key = f"proposal_networks.{i}.mlp_base.{name}"
        elif k.startswith("field_background."):
            key = k
        else:
            key = f"field.{k}"
        assert key in sd, f"{key} missing from the product model"
        assert tuple(sd[key].shape) == tuple(v.shape), (key, tuple(sd[key].shape), tuple(v.shape))
        sd[key] = v.clone()
        seen.add(key)
    missing = [k for k in sd if k not in seen and not k.endswith("aabb")]
    assert not missing, f"product parameters without a value: {missing}"
    model.load_state_dict(sd)
    return model.to(device)


def angelo_product_grads(model):
    """Gradients of the product model keyed by oracle names (incl. field_background.*)."""
    out = product_grads(model)
    for k, p in model.named_parameters():
        if p.grad is not None and k.startswith("field_background."):
            out[k] = p.grad.detach()
    return out


def load_external_golden(path, required):
    """A vector file minted on a CUDA box (tools/mint_*_golden.py).  Absent files make their tests SKIP; a file that is present but
    unreadable or incomplete must FAIL loudly - a skipped test would read as "still unpinned" when the truth is "pinned data rejected"."""
    try:
        z = np.load(path)
        keys = set(z.files)
    except Exception as e:  # noqa: BLE001 - whatever numpy raises on a damaged archive
        raise AssertionError(f"{os.path.basename(path)} is present but cannot be read as an .npz archive: {e}") from e
    missing = [k for k in required if k not in keys]
    assert not missing, f"{os.path.basename(path)} is present but malformed: missing arrays {missing} (has {sorted(keys)})"
    return z
