"""Background field of the surface models: nerfstudio/fields/vanilla_nerf_field.py:37-114 (NeRFField) with its components
(field_components/encodings.py:93-208 NeRFEncoding, field_components/mlp.py:25-100 MLP, field_heads.py:60-121).

``background_model="mlp"`` (the reference's default, base_surface_model.py:123,189-200) evaluates this 8x256 ReLU MLP on the
SDF samples outside the unit sphere and on ``num_samples_outside`` extra samples beyond the far plane.  Same module tree and
``state_dict`` keys as the reference (``mlp_base.layers.N``, ``mlp_head.layers.N``, ``field_output_density.net``,
``field_heads.0.net``), so reference checkpoints load unchanged.

SURVEY section 8 row f4: the field runs on the SAME fused kernels as the SDF field, instantiated with a ReLU activation
(csrc/inst_d.hip; include/sdfhip.h SdfHipFieldCfg.activation / .skip_style): position encoding + the 8 x 256 base MLP with its skip
concatenation + the density row + the base-output half of the head's first layer are one ``geo_fwd_kernel`` launch
(``sdfhip_geo_forward``), the rest of the head (direction encoding columns, second head layer, rgb layer, sigmoid) one
``col_fwd_kernel`` launch (``sdfhip_color_forward``); backward = ``geo_bwd_kernel<TANGENT = false>``, ``col_bwd_kernel`` and the
split-K weight-gradient GEMMs.  No torch matmul, no rocBLAS, no CPU path: the modules below only OWN the parameters (reference
``state_dict`` layout); ``NeRFField.forward`` raises on CPU tensors.
"""
import ctypes
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from sdfstudio_amd import _lib
from sdfstudio_amd.fields.field_heads import FieldHeadNames


class NativeBackgroundNet:
    """What the first-order autograd nodes of fields/sdf_field.py (_GeoNetFunction, _ColorFunction) need from a field: the native
    handle (created on first use: it allocates device tables) and the two widths they size outputs with."""

    def __init__(self, cfg_c: "_lib.FieldCfg", geo_feat_dim: int, emb_dim: int, expect) -> None:
        self._cfg_c = cfg_c
        self.config = SimpleNamespace(geo_feat_dim=geo_feat_dim, appearance_embedding_dim=emb_dim)
        self._expect = expect  # [(out_dim, in_dim)] of every linear layer of theta, checked against the library's layout
        self._handle_v: Optional[ctypes.c_void_p] = None

    @property
    def _handle(self) -> ctypes.c_void_p:
        if self._handle_v is None:
            lib = _lib.load()
            h = ctypes.c_void_p()
            _lib.check(lib.sdfhip_field_create(ctypes.byref(self._cfg_c), ctypes.byref(h)), "sdfhip_field_create (background field)")
            n_lin = lib.sdfhip_field_num_linear(h)
            w_off, b_off = (ctypes.c_int64 * n_lin)(), (ctypes.c_int64 * n_lin)()
            od, idim = (ctypes.c_int32 * n_lin)(), (ctypes.c_int32 * n_lin)()
            lib.sdfhip_field_theta_layout(h, w_off, b_off, od, idim)
            got = [(od[i], idim[i]) for i in range(n_lin)]
            assert got == list(self._expect), (got, self._expect)
            self._handle_v = h
        return self._handle_v

    def __del__(self):
        h = getattr(self, "_handle_v", None)
        if h is not None:
            try:
                _lib.load().sdfhip_field_destroy(h)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass


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
        raise NotImplementedError("this MLP only owns parameters (reference state_dict layout); NeRFField evaluates it inside the fused "
                                  "sdfhip kernels - there is no torch path")


class _Head(nn.Module):
    """field_heads.py:60-96 FieldHead: Linear + activation under the attribute name ``net``."""

    def __init__(self, in_dim: int, out_dim: int, activation: nn.Module) -> None:
        super().__init__()
        self.net = nn.Linear(in_dim, out_dim)
        self.activation = activation

    def forward(self, x):
        raise NotImplementedError("parameter container: the head layer is evaluated inside the fused sdfhip kernels")


class NeRFField(nn.Module):
    """vanilla_nerf_field.py:37-114 with the arguments base_surface_model.py:189-200 passes (10 / 4 frequency encodings)."""

    def __init__(self, position_encoding: Optional[nn.Module] = None, direction_encoding: Optional[nn.Module] = None,
                 base_mlp_num_layers: int = 8, base_mlp_layer_width: int = 256, head_mlp_num_layers: int = 2,
                 head_mlp_layer_width: int = 128, skip_connections: Tuple[int, ...] = (4,), field_heads=None,
                 use_integrated_encoding: bool = False, spatial_distortion=None) -> None:
        super().__init__()
        # The reference's defaults are Identity encodings and an RGB head (vanilla_nerf_field.py:52-61); the surface models always pass the
        # 10- / 4-frequency encodings (base_surface_model.py:189-200), which is what the fused kernels are built for: an encoding left out
        # is refused, not replaced.
        if position_encoding is None or direction_encoding is None:
            raise NotImplementedError("NeRFField: pass position_encoding / direction_encoding (NeRFEncoding, include_input=True) as "
                                      "base_surface_model.py:189-200 does; the reference's Identity defaults are not built")
        if use_integrated_encoding or (field_heads is not None and len(field_heads) != 1):
            raise NotImplementedError("NeRFField: integrated (mip-NeRF) encodings and heads other than the RGB head are not built")
        self.position_encoding = position_encoding
        self.direction_encoding = direction_encoding
        self.spatial_distortion = spatial_distortion
        self.mlp_base = MLP(self.position_encoding.get_out_dim(), base_mlp_num_layers, base_mlp_layer_width,
                            skip_connections=skip_connections, out_activation=nn.ReLU())
        self.mlp_head = MLP(self.mlp_base.get_out_dim() + self.direction_encoding.get_out_dim(), head_mlp_num_layers,
                            head_mlp_layer_width, out_activation=nn.ReLU())
        self.field_output_density = _Head(self.mlp_base.get_out_dim(), 1, nn.Softplus())
        self.field_heads = nn.ModuleList([_Head(self.mlp_head.get_out_dim(), 3, nn.Sigmoid())])
        pe, de = self.position_encoding, self.direction_encoding
        if not (isinstance(pe, NeRFEncoding) and pe.include_input and pe.min_freq == 0.0 and pe.max_freq == pe.num_frequencies - 1 and
                isinstance(de, NeRFEncoding) and de.include_input and de.num_frequencies == 4 and de.min_freq == 0.0 and de.max_freq == 3.0):
            raise NotImplementedError("the fused background kernels take NeRFEncoding(include_input) of the position with frequencies 2^0.."
                                      "2^(n-1) and the 4-frequency direction encoding (base_surface_model.py:189-200)")
        skips = tuple(skip_connections or ())
        if len(skips) > 1 or head_mlp_num_layers < 1 or base_mlp_num_layers < 2:
            raise NotImplementedError("one skip connection, at least two base layers")
        self._nf = pe.num_frequencies
        W, WH, NL, NLC = base_mlp_layer_width, head_mlp_layer_width, base_mlp_num_layers, head_mlp_num_layers
        d0 = 3 + 6 * self._nf
        skip = skips[0] if skips else -1
        grid = _lib.GridCfg(0, 2, 4, 1, 1.0, 0)  # no grid features
        cfg_c = _lib.FieldCfg(NL, W, WH, NLC, WH, skip, self._nf, 1, 0, 0, 0.0, grid, 1, 1)
        expect = [(W, d0 if l == 0 else (W + d0 if l == skip else W)) for l in range(NL)] + [(1 + WH, W)]
        expect += [(WH, 33 + WH if l == 0 else WH) for l in range(NLC)] + [(3, WH)]
        self._native = NativeBackgroundNet(cfg_c, WH, 0, expect)
        self._widths = (W, WH, NL, NLC, d0, skip)

    # ------------------------------------------------------------------ parameters in the library's layout
    def _theta(self) -> torch.Tensor:
        """Flat parameter vector for the fused kernels (differentiable torch indexing: gradients flow back to the modules).
        Geometry-type network = mlp_base (input columns reordered from the reference's [sin | cos | x] to the kernels' [x | sin | cos]);
        its output layer = [density row ; the columns of the head's first layer that multiply the base output] (bias: the density
        bias, zeros); colour-type network = [direction columns of the head's first layer (its 27 inputs sit in the kernels' small
        -input block; position and normal slots get zero weights) | identity on the 128 partial sums] + head bias, the remaining head
        layers, the rgb layer."""
        W, WH, NL, NLC, d0, skip = self._widths
        pe = d0 - 3
        dev, dt = self.mlp_base.layers[0].weight.device, self.mlp_base.layers[0].weight.dtype
        parts = []
        for l, lin in enumerate(self.mlp_base.layers):
            w = lin.weight
            if l == 0:
                w = torch.cat([w[:, pe:pe + 3], w[:, :pe]], dim=1)
            elif l == skip:  # cat([in0, h]) (field_components/mlp.py:86-88)
                w = torch.cat([w[:, pe:pe + 3], w[:, :pe], w[:, d0:]], dim=1)
            parts += [w.reshape(-1), lin.bias]
        h0 = self.mlp_head.layers[0]
        parts += [torch.cat([self.field_output_density.net.weight, h0.weight[:, 27:]], dim=0).reshape(-1),
                  torch.cat([self.field_output_density.net.bias, torch.zeros(WH, device=dev, dtype=dt)])]
        z3 = torch.zeros(WH, 3, device=dev, dtype=dt)
        parts += [torch.cat([z3, h0.weight[:, :27], z3, torch.eye(WH, device=dev, dtype=dt)], dim=1).reshape(-1), h0.bias]
        for lin in list(self.mlp_head.layers)[1:]:
            parts += [lin.weight.reshape(-1), lin.bias]
        parts += [self.field_heads[0].net.weight.reshape(-1), self.field_heads[0].net.bias]
        return torch.cat(parts)

    def _positions(self, ray_samples) -> torch.Tensor:
        positions = ray_samples.frustums.get_positions()  # frustum MID points (rays.py:46-55), unlike the SDF field's starts
        if self.spatial_distortion is not None:
            positions = self.spatial_distortion(positions)
        return positions.reshape(-1, 3).detach().float().contiguous()

    def get_density(self, ray_samples):
        """vanilla_nerf_field.py:91-104.  Returns (density, density_embedding): the embedding handed on to get_outputs is the head's
        first-layer partial product over the base output (128 wide) rather than the 256-wide base output itself - the two methods
        are only ever chained (fields/base_field.py:111-126), and this is what the fused geometry kernel emits."""
        from sdfstudio_amd.fields.sdf_field import _GeoNetFunction

        x = self._positions(ray_samples)
        if not x.is_cuda:
            raise _lib.SdfHipError("NeRFField runs on the sdfhip kernels: HIP device tensors required (no CPU fallback)")
        shape = tuple(ray_samples.frustums.get_positions().shape[:-1])
        theta = self._theta()
        dummy = torch.zeros(8, device=x.device)
        pre, part = _GeoNetFunction.apply(theta, dummy, self._native, x, dummy)
        density = torch.nn.functional.softplus(pre).view(*shape, 1)  # DensityFieldHead: Softplus (field_heads.py:99-107)
        return density, (part, theta, x)

    def density_fn(self, positions: torch.Tensor) -> torch.Tensor:
        """Field.density_fn (fields/base_field.py:48-65): the density at explicit positions [..., 3] (zero-length frustums at the positions)."""
        from sdfstudio_amd.cameras.rays import Frustums, RaySamples

        flat = positions.reshape(-1, 1, 3)
        one = torch.ones_like(flat[..., :1])
        rs = RaySamples(frustums=Frustums(origins=flat, directions=torch.ones_like(flat), starts=torch.zeros_like(one), ends=torch.zeros_like(one),
                                          pixel_area=one))
        density, _ = self.get_density(rs)
        return density.view(*positions.shape[:-1], 1)

    def get_outputs(self, ray_samples, density_embedding=None) -> Dict:
        """vanilla_nerf_field.py:106-114."""
        from sdfstudio_amd.cameras.rays import unpack_ray_samples
        from sdfstudio_amd.fields.sdf_field import _ColorFunction

        part, theta, x = density_embedding
        _, d, st, _ = unpack_ray_samples(ray_samples)
        n, s = st.shape
        rgb = _ColorFunction.apply(theta, part, torch.zeros_like(x), None, self._native, x, d.contiguous(), n, s)
        return {FieldHeadNames.RGB: rgb.view(n, s, 3)}

    def forward(self, ray_samples, compute_normals: bool = False) -> Dict:
        """fields/base_field.py:111-126."""
        if compute_normals:
            raise NotImplementedError("compute_normals (normals of a density field, base_field.py:104-121) is not built")
        density, emb = self.get_density(ray_samples)
        out = self.get_outputs(ray_samples, density_embedding=emb)
        out[FieldHeadNames.DENSITY] = density
        return out
