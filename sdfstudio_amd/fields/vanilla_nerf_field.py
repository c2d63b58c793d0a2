"""Background field of the surface models: nerfstudio/fields/vanilla_nerf_field.py:37-114 (NeRFField) with its components
(field_components/encodings.py:93-208 NeRFEncoding, field_components/mlp.py:25-100 MLP, field_heads.py:60-121).

``background_model="mlp"`` (the reference's default, base_surface_model.py:123,189-200) evaluates this 8x256 ReLU MLP on the
SDF samples outside the unit sphere and on ``num_samples_outside`` extra samples beyond the far plane.  Same module tree and
``state_dict`` keys as the reference (``mlp_base.layers.N``, ``mlp_head.layers.N``, ``field_output_density.net``,
``field_heads.0.net``), so reference checkpoints load unchanged.

SURVEY section 8 lists the background fields as a "next" row (f4): the positions / encodings / compositing around this field
run on the sdfhip kernels and device tensors, the MLP itself is torch matmuls here (rocBLAS), NOT a hand-written kernel -
stated plainly; it is off BASELINE config 2's path (``background_model="none"``).
"""
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from sdfstudio_amd.fields.field_heads import FieldHeadNames


class NeRFEncoding(nn.Module):
    """encodings.py:93-208 (off_axis=False, no covariances): sin of [x f | x f + pi / 2] for f = 2^linspace(min, max, n), input last."""

    def __init__(self, in_dim: int, num_frequencies: int, min_freq_exp: float, max_freq_exp: float, include_input: bool = False) -> None:
        super().__init__()
        self.in_dim, self.num_frequencies = in_dim, num_frequencies
        self.min_freq, self.max_freq, self.include_input = min_freq_exp, max_freq_exp, include_input

    def get_out_dim(self) -> int:
        return self.in_dim * self.num_frequencies * 2 + (self.in_dim if self.include_input else 0)

    def forward(self, in_tensor: torch.Tensor) -> torch.Tensor:
        freqs = 2 ** torch.linspace(self.min_freq, self.max_freq, self.num_frequencies).to(in_tensor.device)
        scaled = (in_tensor[..., None] * freqs).reshape(*in_tensor.shape[:-1], -1)
        enc = torch.sin(torch.cat([scaled, scaled + torch.pi / 2.0], dim=-1))
        return torch.cat([enc, in_tensor], dim=-1) if self.include_input else enc


class MLP(nn.Module):
    """field_components/mlp.py:25-100: ReLU MLP; a skip layer takes cat([input, x])."""

    def __init__(self, in_dim: int, num_layers: int, layer_width: int, out_dim: Optional[int] = None,
                 skip_connections: Optional[Tuple[int, ...]] = None, out_activation: Optional[nn.Module] = None) -> None:
        super().__init__()
        self.in_dim, self.num_layers, self.layer_width = in_dim, num_layers, layer_width
        self.out_dim = out_dim if out_dim is not None else layer_width
        self._skip = set(skip_connections) if skip_connections else set()
        layers = []
        if num_layers == 1:
            layers.append(nn.Linear(in_dim, self.out_dim))
        else:
            for i in range(num_layers - 1):
                if i == 0:
                    layers.append(nn.Linear(in_dim, layer_width))
                elif i in self._skip:
                    layers.append(nn.Linear(layer_width + in_dim, layer_width))
                else:
                    layers.append(nn.Linear(layer_width, layer_width))
            layers.append(nn.Linear(layer_width, self.out_dim))
        self.layers = nn.ModuleList(layers)
        self.out_activation = out_activation

    def get_out_dim(self) -> int:
        return self.out_dim

    def forward(self, in_tensor: torch.Tensor) -> torch.Tensor:
        x = in_tensor
        for i, layer in enumerate(self.layers):
            if i in self._skip:
                x = torch.cat([in_tensor, x], -1)
            x = layer(x)
            if i < len(self.layers) - 1:
                x = torch.relu(x)
        return self.out_activation(x) if self.out_activation is not None else x


class _Head(nn.Module):
    """field_heads.py:60-96 FieldHead: Linear + activation under the attribute name ``net``."""

    def __init__(self, in_dim: int, out_dim: int, activation: nn.Module) -> None:
        super().__init__()
        self.net = nn.Linear(in_dim, out_dim)
        self.activation = activation

    def forward(self, x):
        return self.activation(self.net(x))


class NeRFField(nn.Module):
    """vanilla_nerf_field.py:37-114 with the arguments base_surface_model.py:189-200 passes (10 / 4 frequency encodings)."""

    def __init__(self, position_encoding: Optional[nn.Module] = None, direction_encoding: Optional[nn.Module] = None,
                 base_mlp_num_layers: int = 8, base_mlp_layer_width: int = 256, head_mlp_num_layers: int = 2,
                 head_mlp_layer_width: int = 128, skip_connections: Tuple[int, ...] = (4,), spatial_distortion=None) -> None:
        super().__init__()
        self.position_encoding = position_encoding or NeRFEncoding(3, 10, 0.0, 9.0, include_input=True)
        self.direction_encoding = direction_encoding or NeRFEncoding(3, 4, 0.0, 3.0, include_input=True)
        self.spatial_distortion = spatial_distortion
        self.mlp_base = MLP(self.position_encoding.get_out_dim(), base_mlp_num_layers, base_mlp_layer_width,
                            skip_connections=skip_connections, out_activation=nn.ReLU())
        self.mlp_head = MLP(self.mlp_base.get_out_dim() + self.direction_encoding.get_out_dim(), head_mlp_num_layers,
                            head_mlp_layer_width, out_activation=nn.ReLU())
        self.field_output_density = _Head(self.mlp_base.get_out_dim(), 1, nn.Softplus())
        self.field_heads = nn.ModuleList([_Head(self.mlp_head.get_out_dim(), 3, nn.Sigmoid())])

    def get_density(self, ray_samples):
        positions = ray_samples.frustums.get_positions()  # frustum MID points (rays.py:46-55), unlike the SDF field's starts
        if self.spatial_distortion is not None:
            positions = self.spatial_distortion(positions)
        base = self.mlp_base(self.position_encoding(positions))
        return self.field_output_density(base), base

    def get_outputs(self, ray_samples, density_embedding: torch.Tensor) -> Dict:
        d = ray_samples.frustums.directions.expand(*density_embedding.shape[:-1], 3)
        h = self.mlp_head(torch.cat([self.direction_encoding(d), density_embedding], dim=-1))
        return {FieldHeadNames.RGB: self.field_heads[0](h)}

    def forward(self, ray_samples) -> Dict:
        """fields/base_field.py:111-126."""
        density, emb = self.get_density(ray_samples)
        out = self.get_outputs(ray_samples, density_embedding=emb)
        out[FieldHeadNames.DENSITY] = density
        return out
