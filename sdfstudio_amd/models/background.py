"""Background models of the surface models (base_surface_model.py:181-216, 256-290, 305-329), shared by the three model mirrors.

Two mechanisms exist in the reference and both are mirrored:
  * NeuS / VolSDF (neus.py:94-104, volsdf.py:62-79): the transmittance left after the SDF samples multiplies the colour of
    ``num_samples_outside`` extra samples placed between the far plane and ``far_plane_bg`` by a LinearDisparitySampler and
    shaded by the background field (base_surface_model.py:314-329);
  * NeuS-facto (neus_facto.py:289-290): the background field is evaluated on the SDF samples themselves and replaces alpha and
    colour of the samples outside the unit sphere (forward_background_field_and_merge, base_surface_model.py:266-290).
The samplers, positions and compositing run on device tensors / sdfhip kernels; the background networks are torch ("mlp":
fields/vanilla_nerf_field.py; "grid": fields/nerfacto_field.py, whose hash-grid encoding is the sdfhip operator).
"""
from typing import Dict

import torch
from torch import nn

from sdfstudio_amd.fields.field_heads import FieldHeadNames
from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField
from sdfstudio_amd.fields.vanilla_nerf_field import NeRFEncoding, NeRFField
from sdfstudio_amd.model_components.ray_samplers import LinearDisparitySampler
from sdfstudio_amd.model_components.renderers import AccumulationRenderer, DepthRenderer, RGBRenderer, SemanticRenderer


def build_background(model, config) -> None:
    """base_surface_model.py:181-216: field_background, sampler_bg and the per-head renderers on `model`."""
    if config.background_model == "mlp":
        model.field_background = NeRFField(
            position_encoding=NeRFEncoding(in_dim=3, num_frequencies=10, min_freq_exp=0.0, max_freq_exp=9.0, include_input=True),
            direction_encoding=NeRFEncoding(in_dim=3, num_frequencies=4, min_freq_exp=0.0, max_freq_exp=3.0, include_input=True),
            spatial_distortion=model.scene_contraction)
    elif config.background_model == "grid":
        model.field_background = TCNNNerfactoField(  # :181-187
            model.scene_box.aabb, spatial_distortion=model.scene_contraction, num_images=model.num_train_data,
            use_average_appearance_embedding=getattr(config, "use_average_appearance_embedding", False))
    elif config.background_model == "none":
        model.field_background = nn.Parameter(torch.ones(1), requires_grad=False)  # the reference's dummy (:201-203)
    else:
        raise ValueError(f"background_model={config.background_model!r}: 'grid', 'mlp' or 'none' (base_surface_model.py:122-123)")
    model.sampler_bg = LinearDisparitySampler(num_samples=config.num_samples_outside)
    # "black" / "white" become colours; "random" and "last_sample" travel through to the renderer unchanged, as in the reference
    # (base_surface_model.py:206, renderers.py:81-92); the FUSED field -> render path of the models refuses those two loudly
    bc = config.background_color
    if bc not in ("black", "white", "random", "last_sample"):
        raise ValueError(f"background_color={bc!r}: 'random', 'last_sample', 'white' or 'black' (base_surface_model.py:80)")
    model.renderer_rgb = RGBRenderer(background_color=None if bc == "black" else (torch.ones(3) if bc == "white" else bc))
    model.renderer_depth = DepthRenderer(method="expected")
    model.renderer_normal = SemanticRenderer()
    model.renderer_accumulation = AccumulationRenderer()


def has_background(config) -> bool:
    return config.background_model != "none"


def foreground_mask(ray_samples) -> torch.Tensor:
    """base_surface_model.py:256-264: start position inside the unit sphere."""
    return (ray_samples.frustums.get_start_positions().norm(dim=-1, keepdim=True) < 1.0).float()


def forward_background_field_and_merge(model, ray_samples, field_outputs: Dict) -> Dict:
    """base_surface_model.py:266-290."""
    inside = foreground_mask(ray_samples)
    bg = model.field_background(ray_samples)
    bg_alpha = ray_samples.get_alphas(bg[FieldHeadNames.DENSITY])
    field_outputs[FieldHeadNames.ALPHA] = field_outputs[FieldHeadNames.ALPHA] * inside + (1.0 - inside) * bg_alpha
    field_outputs[FieldHeadNames.RGB] = field_outputs[FieldHeadNames.RGB] * inside + (1.0 - inside) * bg[FieldHeadNames.RGB]
    return field_outputs


def render_background(model, ray_bundle, bg_transmittance: torch.Tensor) -> torch.Tensor:
    """base_surface_model.py:314-329: colour arriving from beyond the far plane, already multiplied by the transmittance
    the foreground leaves.  Moves the bundle's near / far planes exactly as the reference does."""
    ray_bundle.nears = ray_bundle.fars
    ray_bundle.fars = torch.ones_like(ray_bundle.fars) * model.config.far_plane_bg
    samples_bg = model.sampler_bg(ray_bundle)
    out_bg = model.field_background(samples_bg)
    weights_bg = samples_bg.get_weights(out_bg[FieldHeadNames.DENSITY])
    return bg_transmittance * model.renderer_rgb(rgb=out_bg[FieldHeadNames.RGB], weights=weights_bg)
