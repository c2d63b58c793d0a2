"""Mint tests/golden/nerfacc_*.npz from a REAL nerfacc == 0.3.5 (the reference's pin, pyproject.toml:31; CUDA only - run on a CUDA box;
NOT runnable in this repo's containers).   python tools/mint_nerfacc_golden.py [out_dir]

Pins the two nerfacc CUDA operators the packed-sample path restates without being able to run them (oracle/sdf_path.py ray_marching,
ray_resampling: "parity unpinned"): nerfacc.ray_marching with an occupancy grid (models/neus_acc.py:92-143 via
model_components/ray_samplers.py:1315-1503) and nerfacc.ray_resampling (NeuSAccSampler importance sampling).  The consuming tests
(tests/test_cpu_nerfacc_golden.py: oracle; tests/test_gpu_nerfacc_golden.py: HIP kernels) skip while the files are absent."""
import os
import sys

import numpy as np
import torch


def main():
    import nerfacc  # noqa: PLC0415

    assert nerfacc.__version__.startswith("0.3.5"), nerfacc.__version__
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    dev = torch.device("cuda")
    gen = torch.Generator().manual_seed(4)
    for res, step in ((16, 0.05), (32, 0.013), (128, 0.005)):
        n = 512
        o = (torch.rand(n, 3, generator=gen) * 2 - 1) * 1.5
        d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
        aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
        binary = torch.rand(res, res, res, generator=gen) < 0.15
        grid = nerfacc.OccupancyGrid(roi_aabb=aabb, resolution=res).to(dev)
        grid._binary.copy_(binary.to(dev))
        t_min, t_max = nerfacc.ray_aabb_intersect(o.to(dev), d.to(dev), aabb.to(dev))
        ray_indices, t_starts, t_ends = nerfacc.ray_marching(o.to(dev), d.to(dev), t_min=t_min, t_max=t_max, scene_aabb=aabb.to(dev),
                                                             grid=grid, render_step_size=step)
        packed_info = nerfacc.pack_info(ray_indices, n)
        w = torch.rand(t_starts.shape[0], generator=gen).to(dev)
        rp, rs, re = nerfacc.ray_resampling(packed_info, t_starts, t_ends, w, 16)
        np.savez_compressed(os.path.join(out_dir, f"nerfacc_march_{res}.npz"), origins=o.numpy(), dirs=d.numpy(), aabb=aabb.numpy(),
                            binary=binary.numpy(), step=np.float32(step), t_min=t_min.cpu().numpy(), t_max=t_max.cpu().numpy(),
                            ray_indices=ray_indices.cpu().numpy(), t_starts=t_starts.cpu().numpy(), t_ends=t_ends.cpu().numpy(),
                            packed_info=packed_info.cpu().numpy(), weights=w.cpu().numpy(), resampled_packed_info=rp.cpu().numpy(),
                            resampled_starts=rs.cpu().numpy(), resampled_ends=re.cpu().numpy(), nerfacc_version=nerfacc.__version__,
                            device=torch.cuda.get_device_name(0))
        print("wrote march", res, int(t_starts.shape[0]))


if __name__ == "__main__":
    main()
