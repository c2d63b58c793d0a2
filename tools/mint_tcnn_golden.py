"""Mint tests/golden/tcnn_grid_*.npz from a REAL tiny-cuda-nn (run on a CUDA box that has `tinycudann`; NOT runnable in this repo's
containers: no CUDA, no network).   python tools/mint_tcnn_golden.py [out_dir]

The hash-grid arithmetic of the hot path (fields/sdf_field.py:230-241, fields/density_fields.py:75-94) lives in tiny-cuda-nn, which the
reference installs from git master without a pin (README.md:48, Dockerfile:96); oracle/hashgrid.py restates it and says "parity
unpinned".  This script produces the vectors that pin it: for each grid configuration of the BASELINE configs it evaluates
tcnn.Encoding(HashGrid) in FULL precision (dtype float32: tcnn's default fp16 output would only pin 3 digits) on seeded positions
with a seeded table, forward and backward (d / d table for a seeded cotangent, d / d x), and writes inputs + outputs.  Commit the
files; tests/test_cpu_tcnn_golden.py (oracle) and tests/test_gpu_tcnn_golden.py (HIP kernels) pick them up when present and skip
otherwise.  Record `tinycudann` commit and GPU in the file (fields "tcnn_version", "device")."""
import math
import os
import sys

import numpy as np
import torch

CONFIGS = {
    # name: (n_levels, n_features, log2_hashmap_size, base_resolution, max_resolution, interpolation)
    "config2_field": (16, 2, 19, 16, 2048, "Smoothstep"),
    "proposal0": (5, 2, 17, 16, 64, "Linear"),
    "proposal1": (5, 2, 17, 16, 256, "Linear"),
    "config5_field": (16, 8, 22, 64, 4096, "Linear"),
    "small_golden": (8, 2, 11, 4, 128, "Smoothstep"),
}


def main():
    import tinycudann as tcnn  # noqa: PLC0415  (only exists on the CUDA box)

    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    dev = torch.device("cuda")
    for name, (L, F, log2_t, base, max_res, interp) in CONFIGS.items():
        growth = math.exp((math.log(max_res) - math.log(base)) / (L - 1))
        enc = tcnn.Encoding(n_input_dims=3, dtype=torch.float32, encoding_config={
            "otype": "HashGrid", "n_levels": L, "n_features_per_level": F, "log2_hashmap_size": log2_t, "base_resolution": base,
            "per_level_scale": growth, "interpolation": interp})
        gen = torch.Generator().manual_seed(11)
        n = enc.params.numel()
        table = (torch.rand(n, generator=gen) * 2 - 1) * 0.1
        x = torch.rand(4096, 3, generator=gen)
        x[:256] = torch.rand(256, 3, generator=gen) * 3.0 - 1.0  # outside the unit cube
        cot = torch.randn(4096, L * F, generator=gen)
        with torch.no_grad():
            enc.params.copy_(table.to(dev))
        xd = x.to(dev).requires_grad_(True)
        y = enc(xd).float()
        (y * cot.to(dev)).sum().backward()
        np.savez_compressed(os.path.join(out_dir, f"tcnn_grid_{name}.npz"), x=x.numpy(), table=table.numpy(), cot=cot.numpy(),
                            y=y.detach().cpu().numpy(), table_bar=enc.params.grad.float().cpu().numpy(), x_bar=xd.grad.float().cpu().numpy(),
                            cfg=np.array([L, F, log2_t, base, max_res, 1 if interp == "Smoothstep" else 0], np.int64),
                            growth=np.float64(growth), tcnn_version=str(getattr(tcnn, "__version__", "unknown")),
                            device=torch.cuda.get_device_name(0))
        print("wrote", name, y.shape)


def mint_networks(out_dir):
    """tests/golden/tcnn_net_*.npz: a real NetworkWithInputEncoding's flat `params`, inputs and outputs - pins the parameter LAYOUT that
    sdfstudio_amd/utils/tcnn_state_dict.py restates (network before encoding, row-major matrices, 16-padding)."""
    import tinycudann as tcnn  # noqa: PLC0415

    dev = torch.device("cuda")
    # HashMLPDensityField(num_levels=5, max_res=64, log2_hashmap_size=17, hidden_dim=16, num_layers=2) - fields/density_fields.py:75-94
    L, F, log2_t, base, max_res, hidden = 5, 2, 17, 16, 64, 16
    growth = math.exp((math.log(max_res) - math.log(base)) / (L - 1))
    net = tcnn.NetworkWithInputEncoding(n_input_dims=3, n_output_dims=1, encoding_config={
        "otype": "HashGrid", "n_levels": L, "n_features_per_level": F, "log2_hashmap_size": log2_t, "base_resolution": base,
        "per_level_scale": growth}, network_config={
        "otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": hidden, "n_hidden_layers": 1})
    gen = torch.Generator().manual_seed(5)
    params = (torch.rand(net.params.numel(), generator=gen) * 2 - 1) * 0.5
    x = 0.25 + 0.5 * torch.rand(2048, 3, generator=gen)  # inside the cube the scene contraction leaves alone
    with torch.no_grad():
        net.params.copy_(params.to(dev))
        y = net(x.to(dev)).float().cpu()
    np.savez_compressed(os.path.join(out_dir, "tcnn_net_proposal0.npz"), params=params.numpy(), x=x.numpy(), y=y.numpy(),
                        cfg=np.array([L, F, log2_t, base, max_res, hidden], np.int64), growth=np.float64(growth),
                        params_dtype=str(net.params.dtype), tcnn_version=str(getattr(tcnn, "__version__", "unknown")),
                        device=torch.cuda.get_device_name(0))
    print("wrote tcnn_net_proposal0", y.shape, net.params.dtype)


if __name__ == "__main__":
    main()
    mint_networks(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
