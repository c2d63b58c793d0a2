"""NeuS model, mirroring nerfstudio/models/neus.py (NeuSModelConfig :34-47, NeuSModel :50-113) on top of
models/base_surface_model.py: NeuSSampler (hierarchical up-sampling driven by SDFField.get_sdf) -> SDFField -> NeuS alpha
compositing.  Host glue only; every stage is a native call (see models/neus_facto.py for the shared parts)."""
from dataclasses import dataclass, field
from typing import Dict, List, Type

import torch
from torch import nn

from sdfstudio_amd.cameras.rays import RayBundle
from sdfstudio_amd.fields.field_heads import FieldHeadNames
from sdfstudio_amd.model_components.losses import fg_mask_loss, monosdf_depth_loss, surface_losses
from sdfstudio_amd.model_components.ray_samplers import NeuSSampler
from sdfstudio_amd.model_components.renderers import neus_render
from sdfstudio_amd.models import background as B
from sdfstudio_amd.models.neus_facto import NeuSFactoModel, NeuSFactoModelConfig, SceneContraction


@dataclass
class NeuSModelConfig(NeuSFactoModelConfig):
    """models/neus.py:34-47 (+ the SurfaceModelConfig knobs inherited through NeuSFactoModelConfig)."""

    _target: Type = field(default_factory=lambda: NeuSModel)
    num_samples: int = 64
    num_samples_importance: int = 64
    num_samples_outside: int = 32
    num_up_sample_steps: int = 4
    base_variance: float = 64
    perturb: bool = True


class NeuSModel(NeuSFactoModel):
    """models/neus.py:50-113."""

    def populate_modules(self):
        """base_surface_model.py:144-233 + neus.py:61-73."""
        c = self.config
        self._populate_surface_modules()  # contraction (scene_contraction_norm honoured), SDF field, background field, renderers
        self.sampler = NeuSSampler(num_samples=c.num_samples, num_samples_importance=c.num_samples_importance,
                                   num_samples_outside=c.num_samples_outside, num_upsample_steps=c.num_up_sample_steps,
                                   base_variance=c.base_variance)
        bg = {"black": torch.zeros(3), "white": torch.ones(3)}.get(c.background_color)
        if bg is None:
            raise NotImplementedError("background_color must be black or white on the fused path")
        self.register_buffer("background", bg, persistent=False)
        self.anneal_end = 50000

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        return {"fields": list(self.field.parameters()), "field_background": self._background_params()}

    def before_train_iteration(self, step: int):
        if self.anneal_end > 0:
            self.field.set_cos_anneal_ratio(min(1.0, step / self.anneal_end))  # neus.py:80-84

    def after_train_iteration(self, step: int):
        pass

    def sample_and_forward_field(self, ray_bundle: RayBundle) -> Dict:
        """neus.py:94-104."""
        ray_samples = self.sampler(ray_bundle, sdf_fn=self.field.get_sdf)
        if B.has_background(self.config):
            # neus.py:94-104 statement by statement: per-head outputs, and the transmittance behind the last sample for the
            # background colour (the fused kernel returns weights only)
            field_outputs = self.field(ray_samples, return_alphas=True)
            weights, transmittance = ray_samples.get_weights_and_transmittance_from_alphas(field_outputs[FieldHeadNames.ALPHA])
            return {"ray_samples": ray_samples, "field_outputs": field_outputs, "weights": weights,
                    "bg_transmittance": transmittance[:, -1, :], "rendered": self._render_per_head(ray_samples, field_outputs, weights)}
        sdf, grad, rgb, x = self.field.forward_fused(ray_samples)
        bg = None if self.config.background_color == "black" else self.background
        out_rgb, depth, normal, acc, weights, alpha = neus_render(
            sdf, grad, rgb, self.field.deviation_network.variance, ray_samples.flat_directions, ray_samples.flat_starts,
            ray_samples.flat_ends, self.field._cos_anneal_ratio, bg)
        field_outputs = {
            FieldHeadNames.RGB: rgb, FieldHeadNames.SDF: sdf[..., None], FieldHeadNames.GRADIENT: grad,
            FieldHeadNames.ALPHA: alpha[..., None], "points_norm": x.norm(dim=-1, keepdim=True),
            "sampled_sdf": self.field.last_sampled_sdf if self.field.config.use_numerical_gradients else None,  # sdf_field.py:639-655
        }
        return {"ray_samples": ray_samples, "field_outputs": field_outputs, "weights": weights[..., None],
                "rendered": (out_rgb, depth, normal, acc)}

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
        """base_surface_model.py:399-416 (rgb, eikonal, fg mask)."""
        c = self.config
        image = batch["image"].to(outputs["rgb"].device)
        if not self.training:
            return {"rgb_loss": surface_losses(outputs["rgb"], image)["rgb_loss"]}
        nrm = "normal" in batch and c.mono_normal_loss_mult > 0.0  # base_surface_model.py:419-424 (mono-neus / monosdf presets)
        fo = outputs.get("field_outputs") or {}
        curv = c.curvature_loss_multi > 0.0  # neuralangelo.py:163-178 (numerical-gradient field: the six tap values are on hand)
        if curv and fo.get("sampled_sdf") is None:
            raise ValueError("curvature_loss_multi > 0 needs sdf_field.use_numerical_gradients=True")
        loss = surface_losses(outputs["rgb"], image, eik_grad=outputs["eik_grad"], eikonal_mult=c.eikonal_loss_mult,
                              sdf=fo[FieldHeadNames.SDF] if curv else None, sampled_sdf=fo["sampled_sdf"] if curv else None,
                              delta=self.field.numerical_gradients_delta,
                              curvature_mult=c.curvature_loss_multi * getattr(self, "curvature_loss_multi_factor", 1.0) if curv else 0.0,
                              normal_pred=outputs["normal"] if nrm else None, normal_gt=batch["normal"].to(image.device) if nrm else None,
                              normal_mult=c.mono_normal_loss_mult)
        if curv and "curvature_loss" not in loss:  # weight exactly 0 (step 0 of the warm-up): the reference still reports the entry
            loss["curvature_loss"] = outputs["rgb"].new_zeros(())
        if "eik_scale" in outputs:  # NeuS-acc's bounded packed arrays: the mean over all entries -> the mean over the valid ones (neus_acc.py)
            loss["eikonal_loss"] = loss["eikonal_loss"] * outputs["eik_scale"]
        if "fg_mask" in batch and c.fg_mask_loss_mult > 0.0:
            fg = batch["fg_mask"].float().to(image.device)
            loss["fg_mask_loss"] = fg_mask_loss(outputs["weights"].sum(dim=1), fg, c.fg_mask_loss_mult)  # clip + BCE + mean: one launch
        if "depth" in batch and c.mono_depth_loss_mult > 0.0:  # base_surface_model.py:427-437
            loss["depth_loss"] = monosdf_depth_loss(outputs["depth"], batch["depth"].to(image.device)[..., None]) * c.mono_depth_loss_mult
        return self.data_prior_losses(outputs, batch, loss)

    def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
        m = super().get_metrics_dict(outputs, batch)
        if self.training:  # neus.py:106-113
            m["s_val"] = self.field.deviation_network.get_variance().detach()
            m["inv_s"] = 1.0 / self.field.deviation_network.get_variance().detach()
        return m
