"""Ray containers mirroring nerfstudio/cameras/rays.py (RayBundle :233, Frustums :30, RaySamples :109).

Same attribute names and tensor shapes as the reference (``[N,S,1]`` per-sample scalars) so host code written
against sdfstudio reads them unchanged; in addition each object carries the flat ``[N,S]`` / ``[N,3]`` tensors the HIP
kernels consume, so no broadcast views are materialised on the hot path.
"""
import ctypes
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch


class _TensorDataclass:
    """The batch operations of utils/tensor_dataclass.py:149-257 for the three ray containers: every tensor field has the batch
    dimensions in front and ONE data dimension behind (`_field_custom_dimensions` is not used by these classes); nested containers and
    dictionaries of tensors are mapped the same way.  `_BATCH_FIELDS` names the fields that take part; the kernels' flat views (ours) follow
    `to` and are dropped by everything that changes the batch shape (unpack_ray_samples then reads the frustums)."""

    _BATCH_FIELDS: tuple = ()
    _FLAT_FIELDS: tuple = ()

    def _apply(self, fn, keep_flat: bool = False):
        kw = {}
        for f in self.__dataclass_fields__:
            v = getattr(self, f)
            if f in self._BATCH_FIELDS:
                if isinstance(v, torch.Tensor):
                    v = fn(v)
                elif isinstance(v, _TensorDataclass):
                    v = v._apply(fn, keep_flat)
                elif isinstance(v, dict):
                    v = {k: (fn(t) if isinstance(t, torch.Tensor) else t) for k, t in v.items()}
            elif f in self._FLAT_FIELDS:
                v = fn(v) if (keep_flat and isinstance(v, torch.Tensor)) else None
            kw[f] = v
        return type(self)(**kw)

    @property
    def size(self) -> int:
        n = 1
        for d in self.shape:
            n *= int(d)
        return n

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def reshape(self, shape):
        """tensor_dataclass.py:197-217."""
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        return self._apply(lambda t: t.reshape((*shape, t.shape[-1])))

    def flatten(self):
        """tensor_dataclass.py:219-225."""
        return self.reshape((-1,))

    def broadcast_to(self, shape):
        """tensor_dataclass.py:227-246."""
        shape = tuple(shape)
        return self._apply(lambda t: t.broadcast_to((*shape, t.shape[-1])))

    def to(self, device):
        """tensor_dataclass.py:248-257."""
        return self._apply(lambda t: t.to(device), keep_flat=True)

    def __getitem__(self, indices):
        """tensor_dataclass.py:149-162: the same index on the batch dimensions of every field."""
        if isinstance(indices, (torch.Tensor, int, slice, type(Ellipsis))):
            indices = (indices,)
        return self._apply(lambda t: t[tuple(indices) + (slice(None),)])

    def __len__(self) -> int:
        if len(self.shape) == 0:
            raise TypeError("len() of a 0-d tensor")
        return int(self.shape[0])

    def __bool__(self) -> bool:
        """tensor_dataclass.py:172-178."""
        if len(self) == 0:
            raise ValueError(f"The truth value of {self.__class__.__name__} when `len(x) == 0` is ambiguous. Use `len(x)` or `x is not None`.")
        return True

    def __setitem__(self, indices, value):
        raise RuntimeError("Index assignment is not supported for TensorDataclass")


@dataclass
class Frustums(_TensorDataclass):
    """Region of space along a ray (rays.py:30-106). origins/directions are [N,1,3] broadcast views."""

    _BATCH_FIELDS = ("origins", "directions", "starts", "ends", "pixel_area", "offsets")

    origins: torch.Tensor
    directions: torch.Tensor
    starts: torch.Tensor  # [N,S,1]
    ends: torch.Tensor  # [N,S,1]
    pixel_area: torch.Tensor
    offsets: Optional[torch.Tensor] = None

    def get_positions(self) -> torch.Tensor:
        """Frustum mid points (rays.py:46-55)."""
        pos = self.origins + self.directions * (self.starts + self.ends) / 2
        return pos if self.offsets is None else pos + self.offsets

    def get_start_positions(self) -> torch.Tensor:
        """Frustum start points, used by the SDF field (rays.py:61-73)."""
        return self.origins + self.directions * self.starts

    def set_offsets(self, offsets):
        """rays.py:57-59: offsets the samples' positions [..., 3]."""
        self.offsets = offsets

    @classmethod
    def get_mock_frustum(cls, device="cpu") -> "Frustums":
        """rays.py:93-106: a size-1 placeholder frustum."""
        return Frustums(origins=torch.ones((1, 3)).to(device), directions=torch.ones((1, 3)).to(device), starts=torch.ones((1, 1)).to(device),
                        ends=torch.ones((1, 1)).to(device), pixel_area=torch.ones((1, 1)).to(device))

    @property
    def shape(self):
        return self.starts.shape[:-1]


@dataclass
class RaySamples(_TensorDataclass):
    """Samples along rays (rays.py:109-230)."""

    _BATCH_FIELDS = ("frustums", "camera_indices", "deltas", "spacing_starts", "spacing_ends", "metadata", "times")
    _FLAT_FIELDS = ("flat_origins", "flat_directions", "flat_starts", "flat_ends", "flat_bins", "nears", "fars")

    frustums: Frustums
    camera_indices: Optional[torch.Tensor] = None  # [N,1,1]
    deltas: Optional[torch.Tensor] = None  # [N,S,1]
    spacing_starts: Optional[torch.Tensor] = None  # [N,S,1]
    spacing_ends: Optional[torch.Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None
    metadata: Optional[Dict[str, torch.Tensor]] = None
    times: Optional[torch.Tensor] = None
    # ---- flat views for the kernels (ours)
    flat_origins: Optional[torch.Tensor] = None  # [N,3]
    flat_directions: Optional[torch.Tensor] = None  # [N,3]
    flat_starts: Optional[torch.Tensor] = None  # [N,S]
    flat_ends: Optional[torch.Tensor] = None  # [N,S]
    flat_bins: Optional[torch.Tensor] = None  # [N,S+1] spacing-domain bins
    nears: Optional[torch.Tensor] = None  # [N]
    fars: Optional[torch.Tensor] = None  # [N]

    @property
    def shape(self):
        return self.frustums.shape

    # -- compositing math (rays.py:131-230).  get_weights is one HIP kernel (density_weights_fwd / _bwd); the fused models do
    # alpha -> weights -> render in neus_render.  The remaining variants are the per-head statements of the reference as
    # small torch ops on [N,S,1] tensors: the background-model paths (base_surface_model.py:266-329) compose them.
    def get_weights(self, densities: torch.Tensor) -> torch.Tensor:
        from sdfstudio_amd.model_components.renderers import density_to_weights

        return density_to_weights(densities[..., 0], self.flat_starts, self.flat_ends)[..., None]

    def get_alphas(self, densities: torch.Tensor) -> torch.Tensor:
        """rays.py:131-144."""
        return 1 - torch.exp(-self.deltas * densities)

    def get_weights_and_transmittance(self, densities: torch.Tensor):
        """rays.py:169-192: (weights [N,S,1], transmittance [N,S,1] in front of each sample)."""
        dd = self.deltas * densities
        acc = torch.cumsum(dd[..., :-1, :], dim=-2)
        acc = torch.cat([torch.zeros((*acc.shape[:1], 1, 1), device=densities.device), acc], dim=-2)
        transmittance = torch.exp(-acc)
        return (1 - torch.exp(-dd)) * transmittance, transmittance

    def get_weights_and_transmittance_from_alphas(self, alphas: torch.Tensor):
        """rays.py:210-230: transmittance is [N,S+1,1] (its last entry is what reaches the background)."""
        transmittance = torch.cumprod(torch.cat([torch.ones((*alphas.shape[:1], 1, 1), device=alphas.device), 1.0 - alphas + 1e-7], 1), 1)
        return alphas * transmittance[:, :-1, :], transmittance

    def get_weights_from_alphas(self, alphas: torch.Tensor) -> torch.Tensor:
        """rays.py:194-208."""
        return self.get_weights_and_transmittance_from_alphas(alphas)[0]


def _lazy_deltas(self):
    """deltas = ends - starts [N,S,1] (rays.py:322), computed on first use: the fused kernels take starts and ends themselves."""
    d = self.__dict__.get("_deltas")
    if d is None and self.__dict__.get("frustums") is not None:
        d = self.frustums.ends - self.frustums.starts
        self.__dict__["_deltas"] = d
    return d


RaySamples.deltas = property(_lazy_deltas, lambda self, v: self.__dict__.__setitem__("_deltas", v))

_CONST_CACHE: Dict = {}


def constant_column(n: int, value: float, device) -> torch.Tensor:
    """A cached [n,1] tensor filled with `value` (default pixel areas, fixed near / far planes): READ-ONLY - one fill per shape instead
    of one per training step.  The rule is enforced where it can be: every hand-out checks the tensor's autograd version counter against
    the one it was created with, so an in-place write by any consumer (a collider's clamp_, a masked assignment in user code) fails
    loudly at the NEXT batch of that size instead of silently corrupting every later one."""
    key = (int(n), float(value), str(device))
    hit = _CONST_CACHE.get(key)
    if hit is not None:
        t, version = hit
        if t._version != version:
            del _CONST_CACHE[key]
            raise RuntimeError(f"a cached constant column ({n} x 1, value {value}) was written in place by one of its consumers: ray_bundle.nears / "
                               "fars / pixel_area handed out by the default paths are shared, read-only tensors - clone before modifying")
        return t
    if len(_CONST_CACHE) > 64:
        _CONST_CACHE.clear()
    t = torch.full((int(n), 1), float(value), device=device)
    _CONST_CACHE[key] = (t, t._version)
    return t


@dataclass
class RayBundle:
    """A bundle of rays (rays.py:233-339)."""

    origins: torch.Tensor  # [N,3]
    directions: torch.Tensor  # [N,3]
    pixel_area: Optional[torch.Tensor] = None  # [N,1]
    directions_norm: Optional[torch.Tensor] = None  # [N,1]
    camera_indices: Optional[torch.Tensor] = None  # [N,1]
    nears: Optional[torch.Tensor] = None  # [N,1]
    fars: Optional[torch.Tensor] = None  # [N,1]
    metadata: Optional[Dict[str, torch.Tensor]] = None
    times: Optional[torch.Tensor] = None

    def __len__(self):
        """rays.py:266-268: the number of rays whatever the batch shape ([N] for training batches, [H, W] for a camera's image)."""
        return self.origins.numel() // self.origins.shape[-1]

    @property
    def shape(self):
        return self.origins.shape[:-1]

    @property
    def size(self) -> int:
        return len(self)

    @property
    def ndim(self) -> int:
        return self.origins.dim() - 1

    def reshape(self, shape) -> "RayBundle":
        """tensor_dataclass.py:197-217."""
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        return self._map(lambda t: t.reshape((*shape, t.shape[-1])))

    def broadcast_to(self, shape) -> "RayBundle":
        """tensor_dataclass.py:227-246."""
        shape = tuple(shape)
        return self._map(lambda t: t.broadcast_to((*shape, t.shape[-1])))

    def to(self, device) -> "RayBundle":
        """tensor_dataclass.py:248-257."""
        return self._map(lambda t: t.to(device))

    def set_camera_indices(self, camera_index: int) -> None:
        """rays.py:256-262."""
        self.camera_indices = torch.ones_like(self.origins[..., 0:1]).long() * camera_index

    def sample(self, num_rays: int) -> "RayBundle":
        """rays.py:270-280: a random subset of the rays (python's `random`, as the reference)."""
        import random

        assert num_rays <= len(self)
        return self[random.sample(range(len(self)), k=num_rays)]

    _TENSOR_FIELDS = ("origins", "directions", "pixel_area", "directions_norm", "camera_indices", "nears", "fars", "times")

    def _map(self, fn) -> "RayBundle":
        kw = {k: (None if getattr(self, k) is None else fn(getattr(self, k))) for k in self._TENSOR_FIELDS}
        md = None if self.metadata is None else {k: (fn(v) if isinstance(v, torch.Tensor) else v) for k, v in self.metadata.items()}
        return RayBundle(metadata=md, **kw)

    def flatten(self) -> "RayBundle":
        """tensor_dataclass.py:159-166: batch shape [H, W] (or any) -> [H W], row major."""
        return self._map(lambda t: t.reshape(-1, t.shape[-1]))

    def __getitem__(self, idx) -> "RayBundle":
        """tensor_dataclass.py:120-140 on the batch dimensions (every field is indexed the same way; the last dimension is data)."""
        return self._map(lambda t: t[idx])

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        """rays.py:282-293: the flattened bundle's rays [start_idx, end_idx) (the chunk loop of Model.get_outputs_for_camera_ray_bundle)."""
        return self.flatten()[start_idx:end_idx]

    def get_ray_samples(self, bin_starts, bin_ends, spacing_starts=None, spacing_ends=None,
                        spacing_to_euclidean_fn=None, flat_bins=None) -> RaySamples:
        """rays.py:295-339. bin_starts / bin_ends: [N,S,1] (or [N,S])."""
        if bin_starts.dim() == 2:
            bin_starts, bin_ends = bin_starts[..., None], bin_ends[..., None]
        n = self.origins.shape[0]
        pa = self.pixel_area if self.pixel_area is not None else constant_column(n, 1.0, self.origins.device)
        fr = Frustums(
            origins=self.origins[:, None, :], directions=self.directions[:, None, :], starts=bin_starts, ends=bin_ends,
            pixel_area=pa[:, None, :],
        )
        cam = None if self.camera_indices is None else self.camera_indices[..., None]
        return RaySamples(
            frustums=fr, camera_indices=cam, deltas=None, spacing_starts=spacing_starts,  # deltas: on first use (_lazy_deltas)
            spacing_ends=spacing_ends, spacing_to_euclidean_fn=spacing_to_euclidean_fn, metadata=self.metadata,
            flat_origins=self.origins.contiguous(), flat_directions=self.directions.contiguous(),
            flat_starts=bin_starts[..., 0].contiguous(), flat_ends=bin_ends[..., 0].contiguous(), flat_bins=flat_bins,
            nears=None if self.nears is None else self.nears.reshape(-1).contiguous(),
            fars=None if self.fars is None else self.fars.reshape(-1).contiguous(),
        )


def unpack_ray_samples(rs):
    """(origins [N,3], dirs [N,3], starts [N,S], ends [N,S]) from our RaySamples OR the reference's (duck-typed)."""
    if getattr(rs, "flat_starts", None) is not None:
        return rs.flat_origins, rs.flat_directions, rs.flat_starts, rs.flat_ends
    fr = rs.frustums
    starts = fr.starts
    if starts.dim() != 3:
        raise ValueError(f"expected ray samples with batch shape [N,S], got starts {tuple(starts.shape)}")
    n, s = starts.shape[0], starts.shape[1]
    o = fr.origins.expand(n, s, 3)[:, 0, :].contiguous().float()
    d = fr.directions.expand(n, s, 3)[:, 0, :].contiguous().float()
    if not (torch.equal(fr.origins.expand(n, s, 3)[:, -1, :], fr.origins.expand(n, s, 3)[:, 0, :])):
        raise ValueError("ray samples whose origins vary along the sample axis are not supported by the fused path")
    return o, d, starts[..., 0].contiguous().float(), fr.ends[..., 0].contiguous().float()


def generate_pinhole_rays(u, centers, rot, height: int, width: int, fx: float, fy: float, cx: float, cy: float):
    """One batch of training rays from uniform draws u [n,3] in [0,1): (camera, y, x) as data/utils/pixel_samplers.py:47-50, then the
    pinhole rays of cameras/cameras.py:462-640 for those pixels (perspective camera, no distortion; pixel centres at +0.5) - the
    PixelSampler + RayGenerator arithmetic of one training iteration in ONE native launch (sdfhip_generate_rays).
    centers [C,3], rot [C,3,3] camera-to-world (columns x right, y down, z forward).  Returns (origins [n,3], directions [n,3] unit,
    directions_norm [n,1], camera index [n] int64)."""
    from sdfstudio_amd import _lib

    lib = _lib.load()
    n, dev = u.shape[0], u.device
    o, d = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
    norm = torch.empty(n, 1, device=dev)
    cam = torch.empty(n, dtype=torch.int64, device=dev)
    kp = _lib.Keep()
    _lib.check(lib.sdfhip_generate_rays(kp(u), kp(centers), kp(rot), int(centers.shape[0]), int(height), int(width), float(fx), float(fy),
                                        float(cx), float(cy), n, _lib.ptr(o), _lib.ptr(d), _lib.ptr(norm),
                                        _lib.rawptr(cam), _lib.stream()), "generate_rays")
    del kp
    return o, d, norm, cam


def generate_image_rays(center, rot, height: int, width: int, fx: float, fy: float, cx: float, cy: float, camera_index: int = 0) -> RayBundle:
    """Every pixel's ray of ONE pinhole camera as a RayBundle of batch shape [H, W] - what ``Cameras.generate_rays(camera_indices=i)``
    hands to ``Model.get_outputs_for_camera_ray_bundle`` (cameras/cameras.py:462-640 with the image's full coordinate grid: pixel
    centres at +0.5, x right, y down).  Same native launch as the training rays (sdfhip_generate_rays): the uniform draw that selects
    pixel (y, x) is ((y + 0.5) / H, (x + 0.5) / W), so the two paths cannot drift apart.  center [3], rot [3,3] camera-to-world."""
    dev = center.device
    ys = (torch.arange(height, device=dev, dtype=torch.float32) + 0.5) / height
    xs = (torch.arange(width, device=dev, dtype=torch.float32) + 0.5) / width
    u = torch.stack([torch.full((height, width), 0.5, device=dev), ys[:, None].expand(height, width), xs[None, :].expand(height, width)], -1)
    o, d, norm, _ = generate_pinhole_rays(u.reshape(-1, 3).contiguous(), center.reshape(1, 3).contiguous(), rot.reshape(1, 3, 3).contiguous(),
                                          height, width, fx, fy, cx, cy)
    cam = torch.full((height, width, 1), int(camera_index), dtype=torch.int64, device=dev)
    return RayBundle(origins=o.view(height, width, 3), directions=d.view(height, width, 3), directions_norm=norm.view(height, width, 1),
                     camera_indices=cam)
