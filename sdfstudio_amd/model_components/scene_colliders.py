"""Scene colliders: where along each ray the samplers work (model_components/scene_colliders.py).

SurfaceModel picks one from ``scene_box.collider_type`` (models/base_surface_model.py:166-172): "near_far" (DTU; BASELINE
configs 1-3, 5), "box" (the reference's indoor conversions write it, scripts/datasets/process_nerfstudio_to_sdfstudio.py:103: Replica
room0 of BASELINE config 4) and "sphere"; ``overwrite_near_far_plane`` replaces any of them by fixed planes (:175-176).  These are a
handful of elementwise operations per ray on device tensors (no kernel of their own); the per-ray nears / fars they produce are what
every sampling kernel of the library takes."""
import torch
from torch import nn


class SceneCollider(nn.Module):
    """scene_colliders.py:27-44: keeps nears / fars a bundle already carries."""

    def set_nears_and_fars(self, ray_bundle):
        raise NotImplementedError

    def forward(self, ray_bundle):
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            return ray_bundle
        return self.set_nears_and_fars(ray_bundle)


class NearFarCollider(SceneCollider):
    """scene_colliders.py:111-129."""

    def __init__(self, near_plane: float, far_plane: float) -> None:
        super().__init__()
        self.near_plane = near_plane
        self.far_plane = far_plane

    def set_nears_and_fars(self, ray_bundle):
        o = ray_bundle.origins
        if o.dim() == 2 and o.is_cuda:  # the training path: cached read-only constants instead of three launches per step
            from sdfstudio_amd.cameras.rays import constant_column

            ray_bundle.nears = constant_column(o.shape[0], self.near_plane, o.device)
            ray_bundle.fars = constant_column(o.shape[0], self.far_plane, o.device)
            return ray_bundle
        ones = torch.ones_like(o[..., 0:1])
        ray_bundle.nears = ones * self.near_plane
        ray_bundle.fars = ones * self.far_plane
        return ray_bundle


class AABBBoxCollider(SceneCollider):
    """scene_colliders.py:47-109: slab intersection with the scene box; the near plane clamps in training only."""

    def __init__(self, scene_box, near_plane: float = 0.0) -> None:
        super().__init__()
        self.scene_box = scene_box
        self.near_plane = near_plane

    def _intersect_with_aabb(self, rays_o, rays_d, aabb):
        dir_fraction = 1.0 / (rays_d + 1e-6)  # :71 "avoid divide by zero"
        lo = (aabb[0].to(rays_o) - rays_o) * dir_fraction   # t1, t3, t5
        hi = (aabb[1].to(rays_o) - rays_o) * dir_fraction   # t2, t4, t6
        nears = torch.minimum(lo, hi).max(dim=1).values
        fars = torch.maximum(lo, hi).min(dim=1).values
        near_plane = self.near_plane if self.training else 0
        nears = torch.clamp(nears, min=near_plane)
        fars = torch.maximum(fars, nears + 1e-6)
        return nears, fars

    def set_nears_and_fars(self, ray_bundle):
        nears, fars = self._intersect_with_aabb(ray_bundle.origins, ray_bundle.directions, self.scene_box.aabb)
        ray_bundle.nears = nears[..., None]
        ray_bundle.fars = fars[..., None]
        return ray_bundle


class SphereCollider(SceneCollider):
    """scene_colliders.py:132-170 (its forward ALWAYS sets the planes).  soft_intersection: the reference then replaces the
    discriminant by the radius itself, i.e. near / far = -<d, o> -+ sqrt(radius)."""

    def __init__(self, radius: float = 1.0, soft_intersection: bool = False) -> None:
        super().__init__()
        self.radius = radius
        self.soft_intersection = soft_intersection

    def forward(self, ray_bundle):
        ray_cam_dot = (ray_bundle.directions * ray_bundle.origins).sum(dim=-1, keepdim=True)
        under_sqrt = ray_cam_dot ** 2 - (ray_bundle.origins.norm(p=2, dim=-1, keepdim=True) ** 2 - self.radius ** 2)
        under_sqrt = under_sqrt.clamp_min(0.01)
        if self.soft_intersection:
            under_sqrt = torch.ones_like(under_sqrt) * self.radius
        sign = torch.tensor([-1.0, 1.0], device=under_sqrt.device, dtype=under_sqrt.dtype)
        hits = (torch.sqrt(under_sqrt) * sign - ray_cam_dot).clamp_min(0.01)
        ray_bundle.nears = hits[:, 0:1]
        ray_bundle.fars = hits[:, 1:2]
        return ray_bundle


def build_collider(scene_box, config=None):
    """models/base_surface_model.py:165-176."""
    kind = scene_box.collider_type
    if kind == "near_far":
        collider = NearFarCollider(near_plane=scene_box.near, far_plane=scene_box.far)
    elif kind == "box":
        collider = AABBBoxCollider(scene_box, near_plane=scene_box.near)
    elif kind == "sphere":
        collider = SphereCollider(radius=scene_box.radius, soft_intersection=True)
    else:
        raise NotImplementedError(f"collider_type={kind!r}")
    if config is not None and getattr(config, "overwrite_near_far_plane", False):
        collider = NearFarCollider(near_plane=config.near_plane, far_plane=config.far_plane)
    return collider
