"""Dense-grid SDF evaluation for mesh extraction: the device side of scripts/extract_mesh.py:94-133 and
nerfstudio/utils/marching_cubes.py:15-168.

``ns-extract-mesh`` evaluates ``field.forward_geonetwork(x)[:, 0]`` on 512^3 crops of a 1024^3-4096^3 lattice, coarse to fine
(a 64^3 -> 512^3 pyramid that only refines cells whose |sdf| is below a shrinking threshold).  Here

* ``sdf_on_grid`` evaluates a whole lattice with the inference variant of the fused kernels in "ray layout": one ray per (x, y)
  column, direction +z, ``starts`` = the z coordinates, so the lattice points are formed inside the encode kernel and never exist
  as an [P, 3] tensor; only the sdf row of the network is computed (SDFHIP_MODE_SDF, no feature GEMM, nothing saved);
* ``sdf_on_points`` does the same for explicit positions (the masked levels of the pyramid);
* ``evaluate_crop_pyramid`` is the reference's coarse-to-fine loop (marching_cubes.py:77-121) on device tensors;
* ``get_surface_sliding`` strings them together per crop and hands the volume to skimage's marching cubes when it is
  installed (it is CPU post-processing, not part of the hot path; absent in this image -> the volumes are returned).
No scene contraction is applied: ``forward_geonetwork`` takes positions as given (sdf_field.py:380-410).
"""
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from sdfstudio_amd import _lib


@torch.no_grad()
def sdf_on_points(field, points: torch.Tensor, chunk: int = 1 << 22) -> torch.Tensor:
    """sdf at explicit positions [P,3] -> [P] (forward_geonetwork(x)[:, 0]; chunked so the workspace stays bounded)."""
    pts = points.reshape(-1, 3).contiguous().float()
    out = torch.empty(pts.shape[0], device=pts.device)
    for a in range(0, pts.shape[0], chunk):
        x = pts[a:a + chunk]
        sdf, _ = field._run_inference(_lib.MODE_SDF, x, None, None, x.shape[0], 1, False)
        out[a:a + chunk] = sdf
    return out


@torch.no_grad()
def sdf_on_grid(field, bounding_box_min: Sequence[float], bounding_box_max: Sequence[float], resolution: Sequence[int],
                device=None, chunk_points: int = 1 << 23) -> torch.Tensor:
    """sdf on the lattice linspace(min, max, n) per axis (indexing "ij", as marching_cubes.py:56-61) -> [nx, ny, nz]."""
    nx, ny, nz = (int(r) for r in resolution)
    dev = device if device is not None else field.encoding.params.device
    xs = torch.linspace(float(bounding_box_min[0]), float(bounding_box_max[0]), nx, device=dev)
    ys = torch.linspace(float(bounding_box_min[1]), float(bounding_box_max[1]), ny, device=dev)
    zs = torch.linspace(float(bounding_box_min[2]), float(bounding_box_max[2]), nz, device=dev)
    out = torch.empty(nx, ny, nz, device=dev)
    d = torch.tensor([0.0, 0.0, 1.0], device=dev).expand(ny, 3)
    slab = max(1, chunk_points // (ny * nz))  # x slabs of `slab` columns: one "ray" per (x, y), nz samples each
    for a in range(0, nx, slab):
        xa = xs[a:a + slab]
        n = xa.shape[0] * ny
        o = torch.stack([xa[:, None].expand(-1, ny), ys[None, :].expand(xa.shape[0], -1), torch.zeros(xa.shape[0], ny, device=dev)], -1)
        sdf, _ = field._run_inference(_lib.MODE_SDF, o.reshape(n, 3).contiguous(), d.repeat(xa.shape[0], 1).contiguous(),
                                      zs[None, :].expand(n, nz).contiguous(), n, nz, False)
        out[a:a + slab] = sdf.view(xa.shape[0], ny, nz)
    return out


_avg_pool_3d = torch.nn.AvgPool3d(2, stride=2)
_upsample = torch.nn.Upsample(scale_factor=2, mode="nearest")


@torch.no_grad()
def evaluate_crop_pyramid(sdf: Callable[[torch.Tensor], torch.Tensor], points: torch.Tensor, extent: float,
                          valid: Optional[Callable[[torch.Tensor], torch.Tensor]] = None):
    """marching_cubes.py:77-121: points [3, n, n, n] (n divisible by 8) -> (sdf volume [n^3] refined only near the surface, the
    number of network evaluations per level, the mask [n^3] of lattice points evaluated at full resolution).  ``extent`` = x_max - x_min of the crop (threshold 2 extent / n * 8, halved per level);
    ``valid(pts) -> bool [P]`` restricts the coarsest level (the reference's coarse_mask lookup)."""
    n = points.shape[-1]
    pyramid = [points]
    for _ in range(3):
        points = _avg_pool_3d(points[None])[0]
        pyramid.append(points)
    pyramid = pyramid[::-1]
    mask, pts_sdf, counts, finest = None, None, [], None
    threshold = 2 * extent / n * 8
    for pid, pts in enumerate(pyramid):
        cn = pts.shape[-1]
        pts = pts.reshape(3, -1).permute(1, 0).contiguous()
        if mask is None:
            if valid is not None:
                pts_sdf = torch.ones_like(pts[:, 1])
                vm = valid(pts)
                if vm.any():
                    pts_sdf[vm] = sdf(pts[vm].contiguous())
                counts.append(int(vm.sum()))
            else:
                pts_sdf = sdf(pts)
                counts.append(pts.shape[0])
        else:
            m = mask.reshape(-1)
            finest = m
            sel = pts[m]
            if sel.shape[0] > 0:
                pts_sdf[m] = sdf(sel.contiguous())
            counts.append(int(sel.shape[0]))
        if pid < 3:
            mask = (torch.abs(pts_sdf) < threshold).reshape(cn, cn, cn)[None, None]
            mask = _upsample(mask.float()).bool()
            pts_sdf = _upsample(pts_sdf.reshape(cn, cn, cn)[None, None]).reshape(-1)
        threshold /= 2.0
    return pts_sdf, counts, finest


@torch.no_grad()
def get_surface_sliding(field, resolution: int = 512, bounding_box_min=(-1.0, -1.0, -1.0), bounding_box_max=(1.0, 1.0, 1.0),
                        level: float = 0.0, crop: int = 512, device=None, return_volumes: bool = False):
    """marching_cubes.py:15-168 without the mask / simplification options: per crop^3 block, coarse-to-fine sdf evaluation on the
    device, then marching cubes (skimage, CPU) if available.  Returns a list of (verts, faces, normals) per block, or with
    return_volumes=True (or without skimage) a list of ((x_min, y_min, z_min), (x_max, y_max, z_max), volume [crop^3])."""
    assert resolution % crop == 0 and crop % 8 == 0
    dev = device if device is not None else field.encoding.params.device
    try:
        from skimage import measure  # type: ignore
    except Exception:  # noqa: BLE001
        measure = None
    nblk = resolution // crop
    edges = [np.linspace(bounding_box_min[a], bounding_box_max[a], nblk + 1) for a in range(3)]
    results = []
    for i in range(nblk):
        for j in range(nblk):
            for k in range(nblk):
                lo = (edges[0][i], edges[1][j], edges[2][k])
                hi = (edges[0][i + 1], edges[1][j + 1], edges[2][k + 1])
                ax = [torch.linspace(float(lo[a]), float(hi[a]), crop, device=dev) for a in range(3)]
                xx, yy, zz = torch.meshgrid(*ax, indexing="ij")
                pts = torch.stack([xx, yy, zz], 0)
                z, _, _ = evaluate_crop_pyramid(lambda p: sdf_on_points(field, p), pts, float(hi[0] - lo[0]))
                if float(z.min()) > level or float(z.max()) < level:
                    continue
                vol = z.reshape(crop, crop, crop)
                if measure is None or return_volumes:
                    results.append((lo, hi, vol))
                    continue
                spacing = tuple((hi[a] - lo[a]) / (crop - 1) for a in range(3))
                verts, faces, normals, _ = measure.marching_cubes(volume=vol.cpu().numpy().astype(np.float32), level=level, spacing=spacing)
                results.append((verts + np.array(lo), faces, normals))
    return results
