"""Dictionary keys of the field boundary (mirror of nerfstudio/field_components/field_heads.py:28-44)."""
from enum import Enum


class FieldHeadNames(Enum):
    RGB = "rgb"
    SH = "sh"
    DENSITY = "density"
    NORMALS = "normals"
    PRED_NORMALS = "pred_normals"
    UNCERTAINTY = "uncertainty"
    TRANSIENT_RGB = "transient_rgb"
    TRANSIENT_DENSITY = "transient_density"
    SEMANTICS = "semantics"
    NORMAL = "normal"
    SDF = "sdf"
    ALPHA = "alpha"
    GRADIENT = "gradient"
    OCCUPANCY = "occupancy"
