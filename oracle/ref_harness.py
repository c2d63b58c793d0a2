"""Oracle tooling (test infrastructure; BUILD CONTAINER ONLY): import the reference's own Python.

``/root/reference`` exists only in the build container, never on the GPU box, so this
module is used solely by ``tests/golden/make_golden.py`` (and by the optional
``tests/test_oracle_vs_reference.py`` which skips itself when the tree is absent) to
validate the restatement in ``oracle/`` and to mint the golden vectors.

* ``install_stubs()`` registers placeholder modules for the third-party packages the
  reference imports at module scope but never exercises on the hot path (torchtyping,
  nerfacc, tyro, wandb, cv2, imageio, torchvision, torchmetrics, ...), plus
  ``torch._six`` and ``torch.utils.tensorboard``.
* ``install_tcnn_shim()`` injects a pure-PyTorch fp32 ``tinycudann`` built on
  ``oracle.hashgrid`` so ``SDFField(encoding_type="hash")`` (sdf_field.py:230) and
  ``HashMLPDensityField`` (density_fields.py:89) construct on CPU.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

# /root/reference is READ-ONLY: importing it must not leave __pycache__ directories behind.  Set before anything of
# the tree is imported; every importer of the reference goes through this module (tests/conftest.py sets it for the test processes too).
sys.dont_write_bytecode = True

import numpy as np
import torch
from torch import nn

from oracle import hashgrid

REFERENCE_ROOT = os.environ.get("SDFSTUDIO_REFERENCE", "/root/reference")

_STUB_ROOTS = (
    "torchtyping", "nerfacc", "tyro", "wandb", "cv2", "imageio", "torchvision", "torchmetrics",
    "open3d", "trimesh", "skimage", "mediapy", "pymeshlab", "zmq", "h5py", "tensorboard",
    "rawpy", "xatlas", "pyngrok", "gdown", "appdirs", "msgpack_numpy", "nuscenes", "plotly",
    "u_msgpack_python", "umsgpack", "cryptography", "aiortc", "aiohttp_cors", "av", "tornado",
    "pyquaternion", "lpips",
)


class _AnyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything

    def __getitem__(cls, item):
        return cls

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls


class _Anything(metaclass=_AnyMeta):
    """A class usable as base class, decorator, callable, subscriptable annotation."""

    def __init__(self, *a, **k):
        pass

    def __class_getitem__(cls, item):
        return cls

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "nerfstudio"))


def install_stubs():
    global _installed
    if _installed:
        return
    # only stub what is really missing
    missing = []
    for root in _STUB_ROOTS:
        try:
            __import__(root)
        except Exception:  # noqa: BLE001
            missing.append(root)
    globals()["_STUB_ROOTS"] = tuple(missing)
    sys.meta_path.insert(0, _StubFinder())
    six = types.ModuleType("torch._six")
    six.string_classes = (str, bytes)
    sys.modules["torch._six"] = six
    if "torch.utils.tensorboard" not in sys.modules:
        tb = types.ModuleType("torch.utils.tensorboard")
        tb.SummaryWriter = _Anything
        sys.modules["torch.utils.tensorboard"] = tb
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


# --------------------------------------------------------------------------- tcnn shim
class _ShimEncoding(nn.Module):
    """tcnn.Encoding(HashGrid) on CPU fp32 via oracle.hashgrid."""

    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        super().__init__()
        self.otype = encoding_config["otype"]
        if self.otype == "SphericalHarmonics":  # fields/nerfacto_field.py:128-134 (degree 4): the oracle's restatement
            assert n_input_dims == 3 and encoding_config["degree"] == 4
            self.n_output_dims = 16
            return
        if self.otype == "Frequency":  # :136-139, only consumed by the predicted-normal head (off): shape only
            self.n_output_dims = n_input_dims * 2 * encoding_config["n_frequencies"]
            return
        assert n_input_dims == 3 and encoding_config["otype"] in ("HashGrid", "Grid")
        self.levels = hashgrid.make_levels(
            n_levels=encoding_config["n_levels"],
            n_features=encoding_config["n_features_per_level"],
            log2_hashmap_size=encoding_config["log2_hashmap_size"],
            base_resolution=encoding_config["base_resolution"],
            per_level_scale=encoding_config["per_level_scale"],
            smoothstep=encoding_config.get("interpolation", "Linear") == "Smoothstep",
        )
        self.n_output_dims = self.levels.n_output_dims
        gen = torch.Generator().manual_seed(seed)
        init = (torch.rand(self.levels.n_params, generator=gen) * 2.0 - 1.0) * 1e-4  # tcnn: U(-1e-4, 1e-4)
        self.params = nn.Parameter(init)

    def forward(self, x):
        if self.otype == "SphericalHarmonics":
            from oracle import sdf_path

            return sdf_path.sh_degree4(x)
        if self.otype == "Frequency":
            raise NotImplementedError("Frequency encoding shim: shape only")
        table = self.params.view(self.levels.n_entries, self.levels.n_features)
        return hashgrid.grid_encode(x, table, self.levels)


class _ShimNetworkWithInputEncoding(nn.Module):
    """tcnn.NetworkWithInputEncoding(HashGrid -> FullyFusedMLP, no biases) on CPU fp32.

    Parameter layout (ours, documented in DESIGN.md): ``params`` = [grid table | W1[h,in] | W2[out,h]].
    """

    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337):
        super().__init__()
        self.encoding = _ShimEncoding(n_input_dims, encoding_config, seed=seed)
        h = network_config["n_neurons"]
        assert network_config["n_hidden_layers"] == 1 and network_config["activation"] == "ReLU"
        assert network_config["output_activation"] == "None"
        d_in = self.encoding.n_output_dims
        gen = torch.Generator().manual_seed(seed + 1)

        def xavier(o, i):
            a = float(np.sqrt(6.0 / (i + o)))
            return (torch.rand(o, i, generator=gen) * 2 - 1) * a

        self.w1 = nn.Parameter(xavier(h, d_in))
        self.w2 = nn.Parameter(xavier(n_output_dims, h))
        self.n_output_dims = n_output_dims

    def forward(self, x):
        f = self.encoding(x)
        return torch.relu(f @ self.w1.t()) @ self.w2.t()


class _ShimNetwork(nn.Module):
    """tcnn.Network(FullyFusedMLP: ReLU hidden layers, no biases, optional sigmoid output) on CPU fp32; weights w1 .. wK [out, in]."""

    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        h, k = network_config["n_neurons"], network_config["n_hidden_layers"]
        assert network_config["activation"] == "ReLU" and network_config["output_activation"] in ("None", "Sigmoid")
        self.sigmoid = network_config["output_activation"] == "Sigmoid"
        gen = torch.Generator().manual_seed(seed + 2)
        dims = [n_input_dims] + [h] * k + [n_output_dims]
        self.n_layers = len(dims) - 1
        for i in range(self.n_layers):
            a = float(np.sqrt(6.0 / (dims[i] + dims[i + 1])))
            setattr(self, f"w{i + 1}", nn.Parameter((torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * a))
        self.n_output_dims = n_output_dims

    def forward(self, x):
        for i in range(self.n_layers):
            x = x @ getattr(self, f"w{i + 1}").t()
            if i + 1 < self.n_layers:
                x = torch.relu(x)
        return torch.sigmoid(x) if self.sigmoid else x


def install_tcnn_shim():
    m = types.ModuleType("tinycudann")
    m.Encoding = _ShimEncoding
    m.NetworkWithInputEncoding = _ShimNetworkWithInputEncoding
    m.Network = _ShimNetwork
    sys.modules["tinycudann"] = m


def import_reference():
    """Returns a namespace with the reference modules the hot path needs."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    install_stubs()
    install_tcnn_shim()
    ns = types.SimpleNamespace()
    import nerfstudio.cameras.rays as rays
    import nerfstudio.field_components.spatial_distortions as sd
    import nerfstudio.fields.density_fields as df
    import nerfstudio.fields.sdf_field as sf
    import nerfstudio.model_components.losses as losses
    import nerfstudio.model_components.ray_samplers as rs
    import nerfstudio.model_components.renderers as rd
    from nerfstudio.field_components.field_heads import FieldHeadNames

    ns.rays, ns.sd, ns.df, ns.sf, ns.losses, ns.rs, ns.rd = rays, sd, df, sf, losses, rs, rd
    ns.FieldHeadNames = FieldHeadNames
    return ns
