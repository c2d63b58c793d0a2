"""Method presets, mirroring nerfstudio/configs/method_configs.py for the surface methods whose model is built here: the model
configuration (with its SDFFieldConfig), the optimizer dictionary, the ray batch sizes and the iteration count of each `ns-train <method>`
entry - the same values under the same names (tests/test_cpu_oracle_and_abi.py compares every entry with the reference's, field by field).
Not the trainer / pipeline / data-manager configuration around them: that is the reference's own control plane.

    from sdfstudio_amd.configs.method_configs import method_configs
    m = method_configs["neus-facto"]
    model = m.model.setup(scene_box=..., num_train_data=...)
    opts = Optimizers(m.optimizers, model.get_param_groups())

Not here: geo-neus / geo-volsdf / geo-unisurf (multi-view patch warping: the data manager's neighbouring images), bakedsdf / bakedsdf-mlp
(the model is built - models/bakedsdf.py - but the presets' field sizes, a 371-column input resp. 1024-wide layers, have no kernel
instantiation), dto, neusW."""
import dataclasses
from typing import Any, Dict

from sdfstudio_amd.engine.optimizers import (AdamOptimizerConfig, AdamWOptimizerConfig, ExponentialSchedulerConfig, MultiStepSchedulerConfig,
                                             MultiStepWarmupSchedulerConfig, NeuSSchedulerConfig)
from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
from sdfstudio_amd.models.bakedsdf import BakedAngeloModelConfig
from sdfstudio_amd.models.neuralangelo import NeuralangeloModelConfig
from sdfstudio_amd.models.neus import NeuSModelConfig
from sdfstudio_amd.models.neus_acc import NeuSAccModelConfig
from sdfstudio_amd.models.neus_facto import NeuSFactoModelConfig
from sdfstudio_amd.models.unisurf import UniSurfModelConfig
from sdfstudio_amd.models.volsdf import VolSDFModelConfig


@dataclasses.dataclass
class MethodConfig:
    """The path's share of a reference `Config` (configs/base_config.py): pipeline.model, optimizers, pipeline.datamanager.{train,eval}_num_rays_per_batch,
    trainer.max_num_iterations, trainer.mixed_precision (False in every surface method; camera_optimizer mode "off" likewise)."""

    method_name: str
    model: Any
    optimizers: Dict[str, Dict[str, Any]]
    train_num_rays_per_batch: int
    eval_num_rays_per_batch: int
    max_num_iterations: int
    mixed_precision: bool = False


def _adam(lr, scheduler, cls=AdamOptimizerConfig, weight_decay=0):
    return {"optimizer": cls(lr=lr, eps=1e-15, weight_decay=weight_decay), "scheduler": scheduler}


def _neus_groups(max_steps=300000, warm_up_end=5000):
    """fields + field_background on Adam 5e-4 with the NeuS warm-up / cosine schedule (neus, mono-neus, unisurf, mono-unisurf, neus-acc)."""
    return {k: _adam(5e-4, NeuSSchedulerConfig(warm_up_end=warm_up_end, learning_rate_alpha=0.05, max_steps=max_steps))
            for k in ("fields", "field_background")}


def _exp_groups(max_steps):
    return {k: _adam(5e-4, ExponentialSchedulerConfig(decay_rate=0.1, max_steps=max_steps)) for k in ("fields", "field_background")}


_ANGELO_FIELD = dict(use_grid_feature=True, num_layers=1, num_layers_color=4, hidden_dim=256, hidden_dim_color=256, geometric_init=True, bias=0.5,
                     beta_init=0.3, inside_outside=False, use_numerical_gradients=True, base_res=64, max_res=4096, log2_hashmap_size=22,
                     hash_features_per_level=8, hash_smoothstep=False, use_position_encoding=False)

method_configs: Dict[str, MethodConfig] = {}

# method_configs.py:452-500
method_configs["neus-facto"] = MethodConfig(
    "neus-facto",
    NeuSFactoModelConfig(sdf_field=SDFFieldConfig(use_grid_feature=True, num_layers=2, num_layers_color=2, hidden_dim=256, bias=0.5, beta_init=0.3,
                                                  use_appearance_embedding=False),
                         background_model="none", eval_num_rays_per_chunk=1024),
    {"proposal_networks": _adam(1e-2, MultiStepSchedulerConfig(max_steps=20000)),
     "fields": _adam(5e-4, NeuSSchedulerConfig(warm_up_end=500, learning_rate_alpha=0.05, max_steps=20000)),
     "field_background": _adam(5e-4, NeuSSchedulerConfig(warm_up_end=500, learning_rate_alpha=0.05, max_steps=20000))},
    train_num_rays_per_batch=2048, eval_num_rays_per_batch=1024, max_num_iterations=20001)

# :503-541 ("used in training heritage data with 8 gpus")
method_configs["neus-facto-bigmlp"] = MethodConfig(
    "neus-facto-bigmlp",
    NeuSFactoModelConfig(sdf_field=SDFFieldConfig(use_grid_feature=False, num_layers=8, num_layers_color=4, hidden_dim=512, bias=0.8, beta_init=0.1,
                                                  use_appearance_embedding=False),
                         eval_num_rays_per_chunk=1024),
    {"proposal_networks": _adam(1e-2, MultiStepSchedulerConfig(max_steps=100000)),
     "fields": _adam(1e-3, NeuSSchedulerConfig(warm_up_end=500, learning_rate_alpha=0.05, max_steps=100000)),
     "field_background": _adam(1e-2, NeuSSchedulerConfig(warm_up_end=500, learning_rate_alpha=0.05, max_steps=100000))},
    train_num_rays_per_batch=2048, eval_num_rays_per_batch=1024, max_num_iterations=100001)

# :381-450 (BASELINE config 5)
method_configs["neus-facto-angelo"] = MethodConfig(
    "neus-facto-angelo",
    NeuSFactoModelConfig(near_plane=0.01, far_plane=1000.0, overwrite_near_far_plane=True,
                         sdf_field=SDFFieldConfig(use_appearance_embedding=True, **_ANGELO_FIELD),
                         background_model="grid", eval_num_rays_per_chunk=1024, level_init=8, eikonal_loss_mult=0.01, use_anneal_beta=True,
                         enable_progressive_hash_encoding=True, enable_numerical_gradients_schedule=True, enable_curvature_loss_schedule=True,
                         curvature_loss_multi=5e-4),
    {"proposal_networks": _adam(1e-2, MultiStepSchedulerConfig(max_steps=1000000)),
     "fields": _adam(1e-3, MultiStepWarmupSchedulerConfig(warm_up_end=5000, milestones=[600_000, 800_000], gamma=0.1)),
     "field_background": _adam(1e-3, MultiStepWarmupSchedulerConfig(warm_up_end=5000, milestones=[300_000, 400_000], gamma=0.1), cls=AdamWOptimizerConfig)},
    train_num_rays_per_batch=2048, eval_num_rays_per_batch=1024, max_num_iterations=1000_001)

# :184-243.  (Its "fields" group is AdamW with weight_decay 0.01: sdfhip_adamw_step, the fused step with decoupled decay.)
method_configs["neuralangelo"] = MethodConfig(
    "neuralangelo",
    NeuralangeloModelConfig(sdf_field=SDFFieldConfig(use_appearance_embedding=False, position_encoding_max_degree=6, **_ANGELO_FIELD),
                            background_model="mlp", enable_progressive_hash_encoding=True, enable_curvature_loss_schedule=True,
                            enable_numerical_gradients_schedule=True),
    {"fields": _adam(1e-3, MultiStepWarmupSchedulerConfig(warm_up_end=5000, milestones=[300_000, 400_000], gamma=0.1), cls=AdamWOptimizerConfig,
                     weight_decay=0.01),
     "field_background": _adam(1e-3, MultiStepWarmupSchedulerConfig(warm_up_end=5000, milestones=[300_000, 400_000], gamma=0.1), cls=AdamWOptimizerConfig)},
    train_num_rays_per_batch=512, eval_num_rays_per_batch=512, max_num_iterations=500_001)

# :721-753, :686-719
method_configs["neus"] = MethodConfig("neus", NeuSModelConfig(eval_num_rays_per_chunk=1024), _neus_groups(), 1024, 1024, 100000)
method_configs["mono-neus"] = MethodConfig("mono-neus", NeuSModelConfig(mono_depth_loss_mult=0.1, mono_normal_loss_mult=0.05, eval_num_rays_per_chunk=1024),
                                           _neus_groups(), 1024, 1024, 100000)
# :616-648, :581-614 (BASELINE configs 1 and 4's alternate reading)
method_configs["volsdf"] = MethodConfig("volsdf", VolSDFModelConfig(eval_num_rays_per_chunk=1024), _exp_groups(100000), 1024, 1024, 100000)
method_configs["monosdf"] = MethodConfig("monosdf", VolSDFModelConfig(mono_depth_loss_mult=0.1, mono_normal_loss_mult=0.05, eval_num_rays_per_chunk=1024),
                                         _exp_groups(200000), 1024, 1024, 200000)
# :756-788, :791-823
method_configs["unisurf"] = MethodConfig("unisurf", UniSurfModelConfig(eval_num_rays_per_chunk=1024), _neus_groups(), 1024, 1024, 100000)
method_configs["mono-unisurf"] = MethodConfig("mono-unisurf",
                                              UniSurfModelConfig(mono_depth_loss_mult=0.1, mono_normal_loss_mult=0.05, eval_num_rays_per_chunk=1024),
                                              _neus_groups(), 1024, 1024, 100000)
# :937-970
method_configs["neus-acc"] = MethodConfig("neus-acc", NeuSAccModelConfig(eval_num_rays_per_chunk=1024), _neus_groups(max_steps=20000, warm_up_end=500),
                                          2048, 1024, 20000)

# :111-181: BakedSDF's model on the neuralangelo-type field (BASELINE config 5's field shape), AdamW with weight decay on the fields
method_configs["bakedangelo"] = MethodConfig(
    "bakedangelo",
    BakedAngeloModelConfig(near_plane=0.01, far_plane=1000.0, overwrite_near_far_plane=True,
                           sdf_field=SDFFieldConfig(use_grid_feature=True, num_layers=1, num_layers_color=4, hidden_dim=256, hidden_dim_color=256,
                                                    geometric_init=True, bias=1.5, beta_init=0.1, inside_outside=True, use_appearance_embedding=True,
                                                    use_numerical_gradients=True, base_res=64, max_res=4096, log2_hashmap_size=22,
                                                    hash_features_per_level=8, hash_smoothstep=False, use_position_encoding=False),
                           background_model="grid", eval_num_rays_per_chunk=1024, proposal_weights_anneal_max_num_iters=10000, use_anneal_beta=True,
                           beta_anneal_max_num_iters=1000000, beta_anneal_init=0.1, beta_anneal_end=0.0002, eikonal_loss_mult=0.01,
                           level_init=4, steps_per_level=10000, curvature_loss_warmup_steps=20000, curvature_loss_multi=5e-4),
    {"proposal_networks": _adam(1e-2, MultiStepSchedulerConfig(max_steps=1000000)),
     "fields": _adam(1e-3, MultiStepWarmupSchedulerConfig(warm_up_end=5000, milestones=[600_000, 800_000], gamma=0.1), cls=AdamWOptimizerConfig,
                     weight_decay=0.01),
     "field_background": _adam(1e-3, MultiStepWarmupSchedulerConfig(warm_up_end=5000, milestones=[300_000, 400_000], gamma=0.1), cls=AdamWOptimizerConfig)},
    train_num_rays_per_batch=8192, eval_num_rays_per_batch=1024, max_num_iterations=1000_001)

