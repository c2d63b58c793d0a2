"""Proposal density field, mirroring nerfstudio/fields/density_fields.py:40-121 (HashMLPDensityField).

The reference builds ``tcnn.NetworkWithInputEncoding(HashGrid -> FullyFusedMLP)``; here contraction, (x+2)/4, the
5-level linear hash grid, the 10->16->1 ReLU MLP (no biases, as tcnn's FullyFusedMLP) and trunc_exp are ONE HIP
kernel per direction (sdfhip_proposal_forward / _backward).  Parameters are plain fp32 tensors:
``mlp_base.table`` [entries*2], ``mlp_base.w1`` [16,10], ``mlp_base.w2`` [1,16].
"""
import math
from typing import Optional

import torch
from torch import nn

from sdfstudio_amd import _lib
from sdfstudio_amd.cameras.rays import unpack_ray_samples
from sdfstudio_amd.grad_slots import grad_target


class _ProposalDensity(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, w1, w2, grid_cfg, contract, origins, dirs, starts, ends):
        lib = _lib.load()
        if dirs is None:
            n, s = origins.shape[0], 1
        else:
            n, s = starts.shape
        density = torch.empty(n, s, device=origins.device)
        import ctypes

        _lib.check(lib.sdfhip_proposal_forward(ctypes.byref(grid_cfg), _lib.ptr(table), _lib.ptr(w1), _lib.ptr(w2),
                                               _lib.ptr(origins), _lib.ptr(dirs), _lib.ptr(starts), _lib.ptr(ends), n, s,
                                               int(contract), _lib.ptr(density), _lib.stream()), "proposal_forward")
        ctx.save_for_backward(table, w1, w2, origins, dirs, starts, ends)
        ctx.grid_cfg, ctx.contract, ctx.shape = grid_cfg, int(contract), (n, s)
        ctx.param_objs = (table, w1, w2)  # the Parameters themselves: their gradient slots (grad_slots.py) are looked up in backward
        return density

    @staticmethod
    def backward(ctx, dbar):
        import ctypes

        table, w1, w2, origins, dirs, starts, ends = ctx.saved_tensors
        lib = _lib.load()
        n, s = ctx.shape
        dev = table.device
        ws = torch.empty(lib.sdfhip_proposal_workspace_size(), dtype=torch.uint8, device=dev)
        # straight into the flat gradient buffer when a FlatGradients owns it (the table gradient is accumulated into: zero start)
        table_bar = grad_target(ctx.param_objs[0], zero_init=True)[0]
        w1_bar = grad_target(ctx.param_objs[1])[0]
        w2_bar = grad_target(ctx.param_objs[2])[0]
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_proposal_backward(ctypes.byref(ctx.grid_cfg), _lib.ptr(table), _lib.ptr(w1), _lib.ptr(w2),
                                                _lib.ptr(origins), _lib.ptr(dirs), _lib.ptr(starts), _lib.ptr(ends), n, s,
                                                ctx.contract, kp(dbar), _lib.rawptr(ws),
                                                _lib.ptr(table_bar), _lib.ptr(w1_bar), _lib.ptr(w2_bar), _lib.stream()),
                   "proposal_backward")
        del kp
        return table_bar, w1_bar, w2_bar, None, None, None, None, None, None


class _PropParams(nn.Module):
    def __init__(self, n_table: int, hidden: int, d_in: int):
        super().__init__()
        self.table = nn.Parameter((torch.rand(n_table) * 2 - 1) * 1e-4)  # tcnn grid init U(-1e-4, 1e-4)
        a1 = math.sqrt(6.0 / (d_in + hidden))
        self.w1 = nn.Parameter((torch.rand(hidden, d_in) * 2 - 1) * a1)  # tcnn xavier_uniform
        a2 = math.sqrt(6.0 / (hidden + 1))
        self.w2 = nn.Parameter((torch.rand(1, hidden) * 2 - 1) * a2)


class HashMLPDensityField(nn.Module):
    """density_fields.py:40-121.  Signature and defaults as the reference's; only the shape neus-facto's proposal networks use is built
    (5 levels x 2 features, 16 hidden, one hidden layer: neus_facto.py:59-64 passes exactly these) - the class defaults themselves
    (8 levels, 64 hidden) are refused, not silently replaced."""

    def __init__(self, aabb, num_layers: int = 2, hidden_dim: int = 64, spatial_distortion=None, use_linear=False,
                 num_levels=8, max_res=1024, base_res=16, log2_hashmap_size=18, features_per_level=2) -> None:
        super().__init__()
        if use_linear or num_layers != 2 or hidden_dim != 16 or num_levels != 5 or features_per_level != 2:
            raise NotImplementedError(
                "sdfhip builds the neus-facto proposal shape only (num_layers=2, hidden_dim=16, 5 levels x 2 features; "
                "neus_facto.py:59-64)")
        self.register_buffer("aabb", torch.as_tensor(aabb, dtype=torch.float32), persistent=False)
        self.spatial_distortion = spatial_distortion
        if spatial_distortion is None:
            raise NotImplementedError("the aabb-normalised (no contraction) proposal path is not built")
        order = getattr(spatial_distortion, "order", None)
        if order not in (float("inf"), None, 2):
            raise NotImplementedError("SceneContraction is built for order = inf and order = None / 2")
        self._contract = 1 if order == float("inf") else 2
        growth = math.exp((math.log(max_res) - math.log(base_res)) / (num_levels - 1))
        self.grid_cfg = _lib.GridCfg(num_levels, features_per_level, log2_hashmap_size, base_res, growth, 0)
        _, n_entries = _lib.grid_levels(self.grid_cfg)
        self.mlp_base = _PropParams(n_entries * features_per_level, hidden_dim, num_levels * features_per_level)

    def density_fn(self, positions) -> torch.Tensor:
        """base_field.py:48-65: densities [..., 1] at explicit positions [...,3]; given a RaySamples instead (an extension the proposal
        sampler uses) the fused kernel forms the frustum mid points itself."""
        p = self.mlp_base
        positions_or_samples = positions
        if isinstance(positions_or_samples, torch.Tensor):
            pos = positions_or_samples
            flat = pos.reshape(-1, 3).contiguous().float()
            d = _ProposalDensity.apply(p.table, p.w1, p.w2, self.grid_cfg, self._contract, flat, None, None, None)
            return d.view(*pos.shape[:-1], 1)
        o, d, st, en = unpack_ray_samples(positions_or_samples)
        dens = _ProposalDensity.apply(p.table, p.w1, p.w2, self.grid_cfg, self._contract, o, d, st, en)
        return dens[..., None]

    def get_density(self, ray_samples):
        return self.density_fn(ray_samples), None

    def forward(self, ray_samples, compute_normals: bool = False):
        if compute_normals:  # base_field.py:104-121: normals of a DENSITY field; nothing on the surface models' path asks for them
            raise NotImplementedError("HashMLPDensityField: compute_normals is not built")
        density, _ = self.get_density(ray_samples)
        return {"density": density}
