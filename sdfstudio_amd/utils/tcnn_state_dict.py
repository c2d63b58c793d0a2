"""Reference checkpoints of the tiny-cuda-nn backed modules <-> this repo's parameter tensors.

The reference's `HashMLPDensityField.mlp_base` (fields/density_fields.py:75-94) and `TCNNNerfactoField.mlp_base` / `.mlp_head`
(fields/nerfacto_field.py:128-225) are tcnn modules, each with ONE flat parameter vector `params`; the mirrors here keep
`mlp_base.{table, w1, w2}` and `mlp_head.{w1, w2, w3}`.  This module maps one onto the other.

STATUS: UNVERIFIED AGAINST A REAL tiny-cuda-nn.  tcnn is an un-vendored, un-pinned dependency of the reference (README.md:48) and cannot
be built where this repo is developed (no CUDA); the layout below restates its published source:

* `NetworkWithInputEncoding::set_params_impl` (network_with_input_encoding.h): the NETWORK's parameters come first, the ENCODING's
  (the hash table, level after level, features of an entry adjacent) after them;
* `FullyFusedMLP` (fully_fused_mlp.cu): weight matrices in order (first, hidden..., last), each ROW-major `[out][in]`, the input width
  padded up to a multiple of 16 and the last layer's output rows padded up to a multiple of 16 (padding rows / columns exist in `params`;
  their values do not reach the output as long as the padded inputs are zero, which tcnn's encodings guarantee), no biases;
* the torch binding keeps `params` in fp32 (tcnn's "full-precision" master copy); an fp16 vector converts the same way.

`tools/mint_tcnn_golden.py` writes `tests/golden/tcnn_net_*.npz` (a real module's `params`, inputs and outputs) on a CUDA box that has
tcnn; `tests/test_gpu_external_goldens.py::test_converted_tcnn_params_against_real_tcnn` consumes them and is skipped until they exist.
Until then a converted checkpoint carries the same "parity unpinned" caveat as the hash grid itself.
"""
from typing import Dict, List, Sequence, Tuple

import torch


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


def mlp_matrix_shapes(d_in: int, hidden: int, n_hidden_layers: int, d_out: int) -> List[Tuple[int, int, int, int]]:
    """(rows used, cols used, rows stored, cols stored) of every FullyFusedMLP weight matrix, in `params` order.
    n_hidden_layers counts tcnn's `n_hidden_layers` (hidden activations): there are n_hidden_layers + 1 matrices."""
    if n_hidden_layers < 1:
        raise ValueError("FullyFusedMLP has at least one hidden layer")
    shapes = [(hidden, d_in, hidden, _pad16(d_in))]
    shapes += [(hidden, hidden, hidden, hidden)] * (n_hidden_layers - 1)
    shapes.append((d_out, hidden, _pad16(d_out), hidden))
    return shapes


def mlp_param_count(d_in: int, hidden: int, n_hidden_layers: int, d_out: int) -> int:
    return sum(rs * cs for _, _, rs, cs in mlp_matrix_shapes(d_in, hidden, n_hidden_layers, d_out))


def split_params(params: torch.Tensor, d_in: int, hidden: int, n_hidden_layers: int, d_out: int, n_grid: int = 0):
    """tcnn `params` -> ([W_0 [hidden, d_in], ..., W_last [d_out, hidden]], table [n_grid] or None), fp32."""
    shapes = mlp_matrix_shapes(d_in, hidden, n_hidden_layers, d_out)
    n_net = sum(rs * cs for _, _, rs, cs in shapes)
    if params.numel() != n_net + n_grid:
        raise ValueError(f"params has {params.numel()} elements, the layout needs {n_net} (network) + {n_grid} (encoding)")
    flat = params.detach().reshape(-1).float()
    mats, off = [], 0
    for r, c, rs, cs in shapes:
        mats.append(flat[off:off + rs * cs].view(rs, cs)[:r, :c].clone())
        off += rs * cs
    return mats, (flat[off:].clone() if n_grid else None)


def join_params(mats: Sequence[torch.Tensor], d_in: int, hidden: int, n_hidden_layers: int, d_out: int, table: torch.Tensor = None) -> torch.Tensor:
    """The inverse: padding rows / columns are written as zeros."""
    shapes = mlp_matrix_shapes(d_in, hidden, n_hidden_layers, d_out)
    if len(mats) != len(shapes):
        raise ValueError(f"{len(shapes)} matrices expected")
    parts = []
    for m, (r, c, rs, cs) in zip(mats, shapes):
        if tuple(m.shape) != (r, c):
            raise ValueError(f"matrix of shape {tuple(m.shape)} where {(r, c)} is expected")
        full = torch.zeros(rs, cs, dtype=torch.float32)
        full[:r, :c] = m.detach().float().cpu()
        parts.append(full.reshape(-1))
    if table is not None:
        parts.append(table.detach().reshape(-1).float().cpu())
    return torch.cat(parts)


def _module_layouts(model) -> Dict[str, Tuple[int, int, int, int, int, Tuple[str, ...]]]:
    """state_dict prefix of every tcnn-backed module of `model` -> (d_in, hidden, n_hidden_layers, d_out, n_grid, names of our tensors)."""
    from sdfstudio_amd.fields.density_fields import HashMLPDensityField
    from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField

    out = {}
    for name, mod in model.named_modules():
        pre = f"{name}." if name else ""
        if isinstance(mod, HashMLPDensityField):
            p = mod.mlp_base
            out[pre + "mlp_base."] = (p.w1.shape[1], p.w1.shape[0], 1, 1, p.table.numel(), ("w1", "w2"))
        elif isinstance(mod, TCNNNerfactoField):
            b, h = mod.mlp_base, mod.mlp_head
            out[pre + "mlp_base."] = (b.w1.shape[1], b.w1.shape[0], 1, b.w2.shape[0], b.table.numel(), ("w1", "w2"))
            out[pre + "mlp_head."] = (h.w1.shape[1], h.w1.shape[0], 2, h.w3.shape[0], 0, ("w1", "w2", "w3"))
    return out


# tcnn modules of the reference's TCNNNerfactoField that carry NO learnable state: parameter-free encodings, whose torch binding still
# registers a zero-element `params` Parameter (fields/nerfacto_field.py:127-139)
_PARAM_FREE = ("direction_encoding.", "position_encoding.")
# optional heads the surface models never switch on (nerfacto_field.py:160-223); a checkpoint that has them carries tcnn params we drop
_OPTIONAL_HEADS = ("mlp_transient.", "mlp_semantics.", "mlp_pred_normals.")


def _nerfacto_prefixes(model) -> List[str]:
    from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField

    return [f"{name}." if name else "" for name, mod in model.named_modules() if isinstance(mod, TCNNNerfactoField)]


def from_reference_state_dict(reference_sd: Dict[str, torch.Tensor], model) -> Dict[str, torch.Tensor]:
    """A reference checkpoint's state_dict -> one `model.load_state_dict(strict=True)` accepts: every `<prefix>params` of a tcnn-backed
    module is replaced by `<prefix>{w1, w2[, w3], table}`; the zero-element `params` of the reference field's parameter-free tcnn
    encodings (`direction_encoding`, `position_encoding`) are dropped - the mirror has no such modules; all other keys pass through."""
    sd = dict(reference_sd)
    for pre, (d_in, hidden, nh, d_out, n_grid, names) in _module_layouts(model).items():
        key = pre + "params"
        if key not in sd:
            continue
        mats, table = split_params(sd.pop(key), d_in, hidden, nh, d_out, n_grid)
        for n, m in zip(names, mats):
            sd[pre + n] = m
        if table is not None:
            sd[pre + "table"] = table
    for pre in _nerfacto_prefixes(model):
        for sub in _PARAM_FREE:
            t = sd.get(pre + sub + "params")
            if t is not None:
                if t.numel() != 0:
                    raise ValueError(f"{pre + sub}params has {t.numel()} elements: a parameter-free tcnn encoding is expected there")
                del sd[pre + sub + "params"]
        for sub in _OPTIONAL_HEADS:
            if pre + sub + "params" in sd:
                raise NotImplementedError(f"the checkpoint carries {pre + sub}params: the transient / semantic / predicted-normal heads of "
                                          "TCNNNerfactoField are not built (off in every surface model)")
    return sd


def to_reference_state_dict(sd: Dict[str, torch.Tensor], model) -> Dict[str, torch.Tensor]:
    """The inverse of from_reference_state_dict (for handing a checkpoint trained here to the reference's viewer / exporters): the
    reference's strict load also wants the (empty) `params` of the parameter-free encodings."""
    out = dict(sd)
    for pre, (d_in, hidden, nh, d_out, n_grid, names) in _module_layouts(model).items():
        if pre + names[0] not in out:
            continue
        mats = [out.pop(pre + n) for n in names]
        table = out.pop(pre + "table") if n_grid else None
        out[pre + "params"] = join_params(mats, d_in, hidden, nh, d_out, table)
    for pre in _nerfacto_prefixes(model):
        if pre + "mlp_base.params" in out:
            for sub in _PARAM_FREE:
                out.setdefault(pre + sub + "params", torch.zeros(0, dtype=torch.float32))
    return out
