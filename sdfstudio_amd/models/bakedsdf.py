"""BakedSDF and BakedAngelo models, mirroring nerfstudio/models/bakedsdf.py (BakedSDFModelConfig :43-99, BakedSDFFactoModel :102-312) and
nerfstudio/models/bakedangelo.py (BakedAngeloModelConfig :43-61, BakedAngeloModel :64-180): a VolSDF-type model (Laplace density of the
sdf) sampled by the proposal networks of the NeuS-facto family, with an annealed Laplace beta, an optionally annealed / spatially varying
eikonal weight and the mip-NeRF-360 proposal loss; BakedAngelo adds the numerical-gradient field's schedules and the curvature loss.
Host glue over the native field, proposal networks and samplers; compositing per head (alpha = 1 - exp(-delta sigma), then the
alpha-weights with their 1e-7: bakedsdf.py:238-246 - not VolSDF's exp(-cumsum) form, so the fused density renderer does not apply).

The `bakedangelo` preset's field (16 x 8 x 2^22 grid, 1 x 256 + 4 x 256, numerical gradients: BASELINE config 5's shape) runs at its own
size; the `bakedsdf` presets' fields (degree-8 off-axis encoding = 371 input columns; 1024-wide) have no kernel instantiation."""
from dataclasses import dataclass, field
from typing import Dict, List, Tuple, Type

import numpy as np
import torch
from torch import nn

from sdfstudio_amd.cameras.rays import RayBundle
from sdfstudio_amd.fields.density_fields import HashMLPDensityField
from sdfstudio_amd.fields.field_heads import FieldHeadNames
from sdfstudio_amd.model_components.losses import interlevel_loss, s3im_loss, surface_losses
from sdfstudio_amd.model_components.ray_samplers import ProposalNetworkSampler
from sdfstudio_amd.models import background as B
from sdfstudio_amd.models.volsdf import VolSDFModel, VolSDFModelConfig


@dataclass
class BakedSDFModelConfig(VolSDFModelConfig):
    """models/bakedsdf.py:43-99 (same names, same defaults)."""

    _target: Type = field(default_factory=lambda: BakedSDFFactoModel)
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_neus_samples_per_ray: int = 48
    proposal_update_every: int = 5
    proposal_warmup: int = 5000
    num_proposal_iterations: int = 2
    use_same_proposal_network: bool = False
    proposal_net_args_list: List[Dict] = field(
        default_factory=lambda: [
            {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 64},
            {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256},
        ]
    )
    interlevel_loss_mult: float = 1.0
    use_proposal_weight_anneal: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    use_single_jitter: bool = True
    use_anneal_beta: bool = True
    beta_anneal_max_num_iters: int = 250000
    beta_anneal_init: float = 0.1
    beta_anneal_end: float = 0.001
    use_anneal_eikonal_weight: bool = False
    eikonal_anneal_max_num_iters: int = 250000
    use_spatial_varying_eikonal_loss: bool = False
    eikonal_loss_mult_start: float = 0.01
    eikonal_loss_mult_end: float = 0.1
    eikonal_loss_mult_slop: float = 2.0


def bakedsdf_beta(step: int, config) -> float:
    """bakedsdf.py:189-194: the Laplace beta the callback writes into laplace_density.beta."""
    frac = float(np.clip(step / config.beta_anneal_max_num_iters, 0, 1))
    return config.beta_anneal_init / (1 + (config.beta_anneal_init - config.beta_anneal_end) / config.beta_anneal_end * (frac ** 0.8))


def bakedsdf_eikonal_mult(step: int, config) -> float:
    """bakedsdf.py:207-214: the annealed eikonal weight (0.01 -> 0.1)."""
    frac = float(np.clip(step / config.eikonal_anneal_max_num_iters, 0, 1))
    w0, w1 = 0.01, 0.1
    return w1 / (1 + (w1 - w0) / w0 * ((1.0 - frac) ** 10))


def spatially_varying_eikonal_weights(points_norm: torch.Tensor, config) -> torch.Tensor:
    """bakedsdf.py:269-277: per-sample eikonal weight from the contracted position's norm (1 inside the unit sphere, growing to 2)."""
    pw = torch.where(points_norm <= 1, torch.ones_like(points_norm), points_norm)
    w0, w1 = config.eikonal_loss_mult_start, config.eikonal_loss_mult_end
    return w1 / (1 + (w1 - w0) / w0 * ((2.0 - pw) ** config.eikonal_loss_mult_slop))


class BakedSDFFactoModel(VolSDFModel):
    """models/bakedsdf.py:102-312."""

    def populate_modules(self):
        """bakedsdf.py:111-150 on top of VolSDFModel.populate_modules."""
        super().populate_modules()
        c = self.config
        self.proposal_networks = nn.ModuleList()
        n_prop = c.num_proposal_iterations
        if c.use_same_proposal_network:
            assert len(c.proposal_net_args_list) == 1, "Only one proposal network is allowed."
            net = HashMLPDensityField(self.scene_box.aabb, spatial_distortion=self.scene_contraction, **c.proposal_net_args_list[0])
            self.proposal_networks.append(net)
            self.density_fns = [net.density_fn for _ in range(n_prop)]
        else:
            for i in range(n_prop):
                args = c.proposal_net_args_list[min(i, len(c.proposal_net_args_list) - 1)]
                self.proposal_networks.append(HashMLPDensityField(self.scene_box.aabb, spatial_distortion=self.scene_contraction, **args))
            self.density_fns = [net.density_fn for net in self.proposal_networks]
        self.proposal_sampler = ProposalNetworkSampler(
            num_nerf_samples_per_ray=c.num_neus_samples_per_ray, num_proposal_samples_per_ray=c.num_proposal_samples_per_ray,
            num_proposal_network_iterations=c.num_proposal_iterations, use_uniform_sampler=False, single_jitter=c.use_single_jitter,
            update_sched=lambda step: -1)

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        """bakedsdf.py:152-168: the annealed beta is not trained."""
        fields = [p for n, p in self.field.named_parameters() if not (self.config.use_anneal_beta and "laplace_density" in n)]
        return {"fields": fields, "proposal_networks": list(self.proposal_networks.parameters()), "field_background": self._background_params()}

    def before_train_iteration(self, step: int):
        """bakedsdf.py:170-224, in the reference's callback order."""
        c = self.config
        if c.use_proposal_weight_anneal:
            frac = float(np.clip(step / c.proposal_weights_anneal_max_num_iters, 0, 1))
            b = c.proposal_weights_anneal_slope
            self.proposal_sampler.set_anneal((b * frac) / ((b - 1) * frac + 1))
        if c.use_anneal_beta:
            self.field.laplace_density.beta.data[...] = bakedsdf_beta(step, c)
        if c.use_anneal_eikonal_weight:
            c.eikonal_loss_mult = bakedsdf_eikonal_mult(step, c)

    def after_train_iteration(self, step: int):
        if self.config.use_proposal_weight_anneal:  # :179-186: the step callback is registered together with the anneal
            self.proposal_sampler.step_cb(step)

    def sample_and_forward_field(self, ray_bundle: RayBundle) -> Dict:
        """bakedsdf.py:226-255, statement by statement on the per-head operators."""
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(ray_bundle, density_fns=self.density_fns)
        field_outputs = self.field(ray_samples)
        field_outputs[FieldHeadNames.ALPHA] = ray_samples.get_alphas(field_outputs[FieldHeadNames.DENSITY])
        if B.has_background(self.config):
            field_outputs = B.forward_background_field_and_merge(self, ray_samples, field_outputs)
        weights = ray_samples.get_weights_from_alphas(field_outputs[FieldHeadNames.ALPHA])
        weights_list.append(weights)
        ray_samples_list.append(ray_samples)
        return {"ray_samples": ray_samples, "field_outputs": field_outputs, "weights": weights, "weights_list": weights_list,
                "ray_samples_list": ray_samples_list, "rendered": self._render_per_head(ray_samples, field_outputs, weights)}

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
        """bakedsdf.py:257-291: rgb, (S3IM,) eikonal - plain or spatially varying -, the mip-NeRF-360 proposal loss."""
        c = self.config
        image = batch["image"].to(outputs["rgb"].device)
        if not self.training:
            return {"rgb_loss": surface_losses(outputs["rgb"], image)["rgb_loss"]}
        grad = outputs["eik_grad"]
        if c.use_spatial_varying_eikonal_loss:
            loss = {"rgb_loss": surface_losses(outputs["rgb"], image)["rgb_loss"]}
            weights = spatially_varying_eikonal_weights(outputs["points_norm"][..., 0], c)
            loss["eikonal_loss"] = (((grad.norm(2, dim=-1) - 1) ** 2) * weights).mean()
        else:
            loss = surface_losses(outputs["rgb"], image, eik_grad=grad, eikonal_mult=c.eikonal_loss_mult)
        if c.s3im_loss_mult > 0:
            loss["s3im_loss"] = s3im_loss(image, outputs["rgb"], c.s3im_kernel_size, c.s3im_stride, c.s3im_repeat_time, c.s3im_patch_height) * c.s3im_loss_mult
        weights = [w[..., 0] for w in outputs["weights_list"]]
        bins = [rs.flat_bins if getattr(rs, "flat_bins", None) is not None else
                torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[..., -1:, 0]], dim=-1) for rs in outputs["ray_samples_list"]]
        loss["interlevel_loss"] = c.interlevel_loss_mult * interlevel_loss(weights, bins)
        return loss

    def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
        m = super().get_metrics_dict(outputs, batch)
        m["eikonal_loss_mult"] = self.config.eikonal_loss_mult  # :308-311
        return m


@dataclass
class BakedAngeloModelConfig(BakedSDFModelConfig):
    """models/bakedangelo.py:43-61."""

    _target: Type = field(default_factory=lambda: BakedAngeloModel)
    enable_progressive_hash_encoding: bool = True
    enable_numerical_gradients_schedule: bool = True
    enable_curvature_loss_schedule: bool = True
    curvature_loss_multi: float = 5e-4
    curvature_loss_warmup_steps: int = 5000
    level_init: int = 4
    steps_per_level: int = 5000


class BakedAngeloModel(BakedSDFFactoModel):
    """models/bakedangelo.py:64-180: BakedSDF on the numerical-gradient field, with the delta / level / curvature schedules in neus-facto-angelo's
    form (delta x 4, floors 1 / (4 max_res) and 1 / (10 max_res): :96-98, :137-139) and the curvature loss (:163-178)."""

    def populate_modules(self):
        super().populate_modules()
        self.curvature_loss_multi_factor = 1.0

    def before_train_iteration(self, step: int):
        super().before_train_iteration(step)
        c, f = self.config, self.field
        if c.enable_numerical_gradients_schedule:
            delta = max(1.0 / (4.0 * f.max_res), 1.0 / (f.base_res * f.growth_factor ** (step / c.steps_per_level)))
            f.set_numerical_gradients_delta(delta * 4.0)
        if c.enable_progressive_hash_encoding:
            f.update_mask(max(int(step / c.steps_per_level) + 1, c.level_init))
        if c.enable_curvature_loss_schedule:
            if step < c.curvature_loss_warmup_steps:
                self.curvature_loss_multi_factor = step / c.curvature_loss_warmup_steps
            else:
                delta = max(1.0 / (f.max_res * 10.0), 1.0 / (f.base_res * f.growth_factor ** ((step - c.curvature_loss_warmup_steps) / c.steps_per_level)))
                self.curvature_loss_multi_factor = delta / (1.0 / f.base_res)

    def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
        m = super().get_metrics_dict(outputs, batch)
        if self.training:
            m["activated_encoding"] = self.field.hash_encoding_mask.mean().item()
            m["numerical_gradients_delta"] = self.field.numerical_gradients_delta
            m["curvature_loss_multi"] = self.curvature_loss_multi_factor * self.config.curvature_loss_multi
        return m

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
        loss = super().get_loss_dict(outputs, batch, metrics_dict)
        c = self.config
        if self.training and c.curvature_loss_multi > 0.0:
            delta = self.field.numerical_gradients_delta
            fo = outputs["field_outputs"]
            centered = fo[FieldHeadNames.SDF]
            around = fo["sampled_sdf"].reshape(centered.shape[:2] + (3, 2))
            curvature = (around.sum(dim=-1) - 2 * centered) / (delta * delta)
            loss["curvature_loss"] = torch.abs(curvature).mean() * c.curvature_loss_multi * self.curvature_loss_multi_factor
        return loss
