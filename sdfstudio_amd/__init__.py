"""sdfstudio_amd — MI355X-native (gfx950) SDF volume-rendering hot path behind sdfstudio's plugin surface.

Python host mirrors of the reference interfaces (same names, arguments, dictionary keys):
  sdfstudio_amd.fields.sdf_field.SDFField / SDFFieldConfig          <- nerfstudio/fields/sdf_field.py
  sdfstudio_amd.fields.density_fields.HashMLPDensityField            <- nerfstudio/fields/density_fields.py
  sdfstudio_amd.model_components.ray_samplers.*                      <- nerfstudio/model_components/ray_samplers.py
  sdfstudio_amd.model_components.renderers.*                         <- nerfstudio/model_components/renderers.py
  sdfstudio_amd.cameras.rays.{RayBundle, RaySamples, Frustums}       <- nerfstudio/cameras/rays.py
  sdfstudio_amd.models.neus_facto.NeuSFactoModel                     <- nerfstudio/models/neus_facto.py
All arithmetic on the path runs in hand-written HIP kernels reached through the C ABI in include/sdfhip.h.
"""
__version__ = "0.1.0"
