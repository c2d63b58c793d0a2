"""VolSDF model, mirroring nerfstudio/models/volsdf.py (VolSDFModelConfig :30-40, VolSDFModel :43-92) on top of
models/base_surface_model.py: ErrorBoundedSampler (VolSDF Algorithm 1) -> SDFField -> Laplace density -> density weights ->
renderers.  BASELINE config 1 is this model with a pure-MLP field (use_grid_feature=False)."""
from dataclasses import dataclass, field
from typing import Dict, List, Type

import torch
import torch.nn.functional as F
from torch import nn

from sdfstudio_amd.cameras.rays import RayBundle
from sdfstudio_amd.fields.field_heads import FieldHeadNames
from sdfstudio_amd.model_components.ray_samplers import ErrorBoundedSampler
from sdfstudio_amd.model_components.renderers import (AccumulationRenderer, DepthRenderer, RGBRenderer, SemanticRenderer,
                                                      density_to_weights)
from sdfstudio_amd.models import background as B
from sdfstudio_amd.models.neus import NeuSModel
from sdfstudio_amd.models.neus_facto import NeuSFactoModelConfig, SceneContraction


@dataclass
class VolSDFModelConfig(NeuSFactoModelConfig):
    """models/volsdf.py:30-40 (+ the SurfaceModelConfig knobs)."""

    _target: Type = field(default_factory=lambda: VolSDFModel)
    num_samples: int = 64
    num_samples_eval: int = 128
    num_samples_extra: int = 32


class VolSDFModel(NeuSModel):
    """models/volsdf.py:43-92."""

    def populate_modules(self):
        c = self.config
        self._populate_surface_modules()
        self.sampler = ErrorBoundedSampler(num_samples=c.num_samples, num_samples_eval=c.num_samples_eval,
                                           num_samples_extra=c.num_samples_extra)
        bg = {"black": torch.zeros(3), "white": torch.ones(3)}.get(c.background_color)
        if bg is None:
            raise NotImplementedError("background_color must be black or white on this path")
        self.register_buffer("background", bg, persistent=False)
        self.anneal_end = -1

    def before_train_iteration(self, step: int):
        pass

    def sample_and_forward_field(self, ray_bundle: RayBundle) -> Dict:
        """volsdf.py:62-79."""
        ray_samples, eik_points = self.sampler(ray_bundle, density_fn=self.field.laplace_density, sdf_fn=self.field.get_sdf)
        sdf, grad, rgb, x = self.field.forward_fused(ray_samples)
        density = self.field.laplace_density(sdf)  # sdf_field.py:659
        weights = density_to_weights(density, ray_samples.flat_starts, ray_samples.flat_ends)[..., None]  # rays.py:169-192
        normals = F.normalize(grad, p=2, dim=-1)
        out_rgb = self.renderer_rgb(rgb=rgb, weights=weights)
        depth = self.renderer_depth(weights=weights, ray_samples=ray_samples)[..., 0]
        normal = self.renderer_normal(semantics=normals, weights=weights)
        acc = self.renderer_accumulation(weights=weights)[..., 0]
        field_outputs = {
            FieldHeadNames.RGB: rgb, FieldHeadNames.SDF: sdf[..., None], FieldHeadNames.GRADIENT: grad,
            FieldHeadNames.DENSITY: density[..., None], FieldHeadNames.NORMAL: normals,
            "points_norm": x.norm(dim=-1, keepdim=True), "sampled_sdf": None,
        }
        out = {"ray_samples": ray_samples, "eik_points": eik_points, "field_outputs": field_outputs, "weights": weights,
               "rendered": (out_rgb, depth, normal, acc)}
        if B.has_background(self.config):
            # volsdf.py:67-68: transmittance in front of the LAST sample (get_weights_and_transmittance's [:, -1]), i.e. without
            # the last sample's own attenuation - restated, not "fixed"
            dd = density * (ray_samples.flat_ends - ray_samples.flat_starts)
            out["bg_transmittance"] = torch.exp(-dd[:, :-1].sum(dim=1, keepdim=True))
        return out

    def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
        image = batch["image"].to(outputs["rgb"].device)
        m = {"psnr": -10.0 * torch.log10(F.mse_loss(outputs["rgb"], image))}
        if self.training:  # volsdf.py:81-88
            m["beta"] = self.field.laplace_density.get_beta().detach()
            m["alpha"] = 1.0 / self.field.laplace_density.get_beta().detach()
        return m
