"""NeuS-acc model, mirroring nerfstudio/models/neus_acc.py (NeuSAccModelConfig :38-44, NeuSAccModel :47-148): NeuS whose samples come
from a march through an occupancy grid that is pruned from the SDF during training ("voxel-surface guided sampling"), composited
on PACKED samples.  The reference leans on three CUDA operators of nerfacc; here they are the sdfhip march / packed-weights /
packed-accumulate kernels (csrc/packed_kernels.h).  Until the first grid update (step 2000) the model IS NeuS."""
from dataclasses import dataclass, field
from typing import Dict, Type

import torch

from sdfstudio_amd.cameras.rays import RayBundle
from sdfstudio_amd.fields.field_heads import FieldHeadNames
from sdfstudio_amd.model_components.ray_samplers import NeuSAccSampler
from sdfstudio_amd.model_components.renderers import accumulate_along_rays, render_weight_from_alpha
from sdfstudio_amd.models.neus import NeuSModel, NeuSModelConfig


@dataclass
class NeuSAccModelConfig(NeuSModelConfig):
    """models/neus_acc.py:38-44."""

    _target: Type = field(default_factory=lambda: NeuSAccModel)
    sky_loss_mult: float = 0.01


class NeuSAccModel(NeuSModel):
    """models/neus_acc.py:47-148."""

    def populate_modules(self):
        super().populate_modules()
        self.sampler = NeuSAccSampler(aabb=self.scene_box.aabb, neus_sampler=self.sampler)  # :61-62

    # the reference registers these two as training callbacks (:64-90)
    def before_train_iteration(self, step: int):
        super().before_train_iteration(step)
        self.sampler.update_step_size(step, inv_s=self.field.deviation_network.get_variance)

    def after_train_iteration(self, step: int):
        super().after_train_iteration(step)
        self.sampler.update_binary_grid(step, sdf_fn=lambda x: self.field.forward_geonetwork(x)[:, 0].contiguous(),
                                        inv_s=self.field.deviation_network.get_variance)

    def get_outputs(self, ray_bundle: RayBundle) -> Dict[str, torch.Tensor]:
        """:92-143."""
        if self.sampler.num_grid_updates() <= 0:  # bootstrap with plain NeuS (host mirror of the counter: no device sync per forward)
            return super().get_outputs(ray_bundle)
        ray_samples, ray_indices = self.sampler(ray_bundle, sdf_fn=self.field.get_sdf, alpha_fn=self.field.get_alpha)
        n_rays = len(ray_bundle)
        dev = ray_bundle.origins.device
        if ray_samples.shape[0] > 0:
            info, counts = self.sampler.packed_info, self.sampler.packed_counts
            field_outputs = self.field(ray_samples, return_alphas=True)  # [P,1,*]
            alphas = field_outputs[FieldHeadNames.ALPHA][:, 0, :]
            weights = render_weight_from_alpha(alphas, info, counts)
            # the four renderers (rgb, normal, accumulation, depth: models/neus_acc.py:106-126) as ONE segmented accumulation over
            # the packed samples: values = [rgb | normal | 1 | mid], 8 columns
            mids = (ray_samples.frustums.starts + ray_samples.frustums.ends)[:, 0, :] / 2
            vals = torch.cat([field_outputs[FieldHeadNames.RGB][:, 0, :], field_outputs[FieldHeadNames.NORMAL][:, 0, :],
                              torch.ones_like(mids), mids], dim=-1)
            acc8 = accumulate_along_rays(weights, ray_indices, vals, info, counts)
            rgb, normal, accumulation, depth = acc8[:, 0:3], acc8[:, 3:6], acc8[:, 6:7], acc8[:, 7:8]
            if ray_bundle.directions_norm is not None:
                depth = depth / ray_bundle.directions_norm  # point-to-point distance -> depth (:127-128)
            # the reference's dictionary has no "weights" entry on this path (the dense-sample losses that read it do not apply);
            # the packed weights travel under their own names
            outputs = {"rgb": rgb, "accumulation": accumulation, "depth": depth, "normal": normal, "packed_weights": weights,
                       "ray_indices": ray_indices, "ray_samples": ray_samples}
            if self.training:
                eik = field_outputs[FieldHeadNames.GRADIENT][:, 0, :]
                n_valid = self.sampler.packed_valid
                if n_valid is not None:
                    # bounded packed arrays (NeuSAccSampler(bounded=True)): the filler samples behind the last ray's segment must not enter
                    # the eikonal mean (base_surface_model.py:406: mean over ALL samples of (|grad| - 1)^2).  They are handed to the loss
                    # as unit vectors - term 0, cotangent 0 - and the mean over P entries is rescaled to the mean over the valid ones
                    p_all = eik.shape[0]
                    valid = torch.arange(p_all, device=dev) < n_valid
                    unit = torch.zeros(3, device=dev)
                    unit[0] = 1.0
                    eik = torch.where(valid[:, None], eik, unit)
                    outputs["eik_scale"] = p_all / n_valid.clamp(min=1).float()
                outputs["eik_grad"] = eik
        else:
            zeros = torch.zeros((n_rays, 3), dtype=torch.float32, device=dev)
            outputs = {"rgb": zeros, "accumulation": zeros[:, :1], "depth": zeros[:, :1], "normal": zeros}
            if self.training:
                outputs["eik_grad"] = zeros
        outputs["normal_vis"] = (outputs["normal"] + 1.0) / 2.0
        return outputs

    def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
        metrics = super().get_metrics_dict(outputs, batch)
        metrics["acc_step_size"] = self.sampler.step_size
        return metrics
