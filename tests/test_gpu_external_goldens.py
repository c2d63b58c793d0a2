"""HIP kernels against goldens minted from the real tiny-cuda-nn / nerfacc (-m gpu); SKIPPED until the files are committed
(tools/mint_tcnn_golden.py, tools/mint_nerfacc_golden.py; see tests/test_cpu_external_goldens.py)."""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TCNN = sorted(glob.glob(os.path.join(GOLDEN, "tcnn_grid_*.npz")))
NERFACC = sorted(glob.glob(os.path.join(GOLDEN, "nerfacc_march_*.npz")))


@pytest.mark.skipif(not TCNN, reason="no tests/golden/tcnn_grid_*.npz: hash grid stays parity-unpinned")
@pytest.mark.parametrize("path", TCNN or ["absent"])
def test_hip_hash_grid_against_real_tcnn(device, path):
    from sdfstudio_amd import _lib
    from sdfstudio_amd.fields.nerfacto_field import hash_grid_encode

    z = np.load(path)
    L, F, log2_t, base, _, smooth = [int(v) for v in z["cfg"]]
    cfg = _lib.GridCfg(L, F, log2_t, base, float(z["growth"]), smooth)
    table = torch.from_numpy(z["table"]).to(device).requires_grad_(True)
    x = torch.from_numpy(z["x"]).to(device)
    y = hash_grid_encode(table, x, cfg)
    (y * torch.from_numpy(z["cot"]).to(device)).sum().backward()
    assert (y.detach().cpu() - torch.from_numpy(z["y"])).abs().max().item() <= 2e-6 * float(np.abs(z["y"]).max()) + 1e-7
    gb = torch.from_numpy(z["table_bar"])
    assert (table.grad.cpu().view(-1) - gb.view(-1)).abs().max().item() <= 1e-4 * gb.abs().max().item()


@pytest.mark.skipif(not NERFACC, reason="no tests/golden/nerfacc_march_*.npz: march / resampling stay parity-unpinned")
@pytest.mark.parametrize("path", NERFACC or ["absent"])
def test_hip_march_and_resampling_against_real_nerfacc(device, path):
    from sdfstudio_amd.model_components.ray_samplers import march_occupancy_grid, resample_packed

    z = np.load(path)
    t = lambda k: torch.from_numpy(z[k]).to(device)  # noqa: E731
    info, counts, ray_idx, ts, te = march_occupancy_grid(t("origins"), t("dirs"), t("t_min"), t("t_max"), torch.from_numpy(z["aabb"]), t("binary"),
                                                         float(z["step"]))
    assert torch.equal(info[:, 1].cpu(), torch.from_numpy(z["packed_info"])[:, 1].long())
    assert torch.equal(ts.view(-1).cpu(), torch.from_numpy(z["t_starts"]).view(-1)) and torch.equal(te.view(-1).cpu(), torch.from_numpy(z["t_ends"]).view(-1))
    _, _, _, rs, re = resample_packed(info, counts, ts, te, t("weights"), 16)
    assert (rs.view(-1).cpu() - torch.from_numpy(z["resampled_starts"]).view(-1)).abs().max().item() <= 2e-6
    assert (re.view(-1).cpu() - torch.from_numpy(z["resampled_ends"]).view(-1)).abs().max().item() <= 2e-6
