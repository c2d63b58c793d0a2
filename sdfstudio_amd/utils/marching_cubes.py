"""Dense-grid SDF evaluation for mesh extraction: the device side of scripts/extract_mesh.py:94-133 and
nerfstudio/utils/marching_cubes.py:15-168.

``ns-extract-mesh`` evaluates ``field.forward_geonetwork(x)[:, 0]`` on 512^3 crops of a 1024^3-4096^3 lattice, coarse to fine
(a 64^3 -> 512^3 pyramid that only refines cells whose |sdf| is below a shrinking threshold).  Here

* ``sdf_on_grid`` evaluates a whole lattice with the inference variant of the fused kernels in "ray layout": one ray per (x, y)
  column, direction +z, ``starts`` = the z coordinates, so the lattice points are formed inside the encode kernel and never exist
  as an [P, 3] tensor; only the sdf row of the network is computed (SDFHIP_MODE_SDF, no feature GEMM, nothing saved);
* ``sdf_on_points`` does the same for explicit positions (the masked levels of the pyramid);
* ``evaluate_crop_pyramid`` is the reference's coarse-to-fine loop (marching_cubes.py:77-121) on device tensors;
* ``marching_cubes`` is ``skimage.measure.marching_cubes`` (scikit-image==0.19.3 in the reference's pyproject; Lewiner's method, the
  default) on a volume that STAYS on the device: libsdfmesh.so (include/sdfmesh.h, csrc_mesh/) returns scikit-image's four arrays - bit
  for bit and in its array order, oracle/marching_cubes.py is pinned on the real package - without the 512 MB host copy and the serial
  Cython pass the reference pays per crop;
* ``get_surface_sliding`` strings them together per crop (marching_cubes.py:15-168: crops, coarse mask, pyramid, marching cubes, the
  crop's offset, concatenation); ``get_surface_occupancy`` is the UniSurf variant (marching_cubes.py:171-216) and
  ``get_surface_sliding_with_contraction`` the one for contracted (unbounded) scenes (marching_cubes.py:218-335).  trimesh's
  ``merge_vertices``, mesh simplification and the .ply writer (pymeshlab / trimesh) are file-format post-processing outside the path and
  are not built.
No scene contraction is applied: ``forward_geonetwork`` takes positions as given (sdf_field.py:380-410).
"""
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from sdfstudio_amd import _lib, _mesh


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
def marching_cubes(volume: torch.Tensor, level: Optional[float] = None, *, spacing: Sequence[float] = (1.0, 1.0, 1.0),
                   gradient_direction: str = "descent", step_size: int = 1, allow_degenerate: bool = True, method: str = "lewiner",
                   mask: Optional[torch.Tensor] = None):
    """skimage.measure.marching_cubes (skimage/measure/_marching_cubes_lewiner.py: marching_cubes -> _marching_cubes_lewiner) on a CUDA
    volume [M, N, P]; same arguments, same checks and error messages, same four results - as DEVICE tensors: verts [V,3] (float32 in
    lattice units for unit spacing, else float64 = float32 vertex x spacing, exactly the wrapper's ``vertices * np.r_[spacing]``),
    faces [F,3] int32, normals [V,3] float32, values [V] float32.  Only what the reference uses is built: method "lewiner",
    step_size 1, allow_degenerate True (the defaults; marching_cubes.py:125-134 passes volume, level, spacing and mask)."""
    if not isinstance(volume, torch.Tensor) or volume.dim() != 3:
        raise ValueError("Input volume should be a 3D numpy array.")
    if min(volume.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    if method != "lewiner":
        raise NotImplementedError(f"marching_cubes: method {method!r} is not built (the reference uses the default, 'lewiner')")
    if int(step_size) != 1 or not allow_degenerate:
        raise NotImplementedError("marching_cubes: step_size != 1 / allow_degenerate=False are not built (the reference uses the defaults)")
    if len(spacing) != 3:
        raise ValueError("`spacing` must consist of three floats.")
    if gradient_direction not in ("descent", "ascent"):
        raise ValueError("Incorrect input %s in `gradient_direction`, see docstring." % (gradient_direction,))
    if mask is not None and tuple(mask.shape) != tuple(volume.shape):
        raise ValueError("volume and mask must have the same shape.")
    vol = volume.contiguous().float()
    lo, hi = (float(t) for t in torch.aminmax(vol))
    if level is None:
        level = 0.5 * (lo + hi)
    else:
        level = float(level)
        if level < lo or level > hi:
            raise ValueError("Surface level must be within volume data range.")
    verts, faces, normals, values = _mesh.marching_cubes_device(vol, level, mask, flip_faces=(gradient_direction == "descent"))
    if verts.shape[0] == 0:
        raise RuntimeError("No surface found at the given iso value.")
    if tuple(float(t) for t in spacing) != (1.0, 1.0, 1.0):
        verts = verts.double() * torch.tensor([float(t) for t in spacing], dtype=torch.float64, device=verts.device)
    return verts, faces, normals, values


def concatenate_meshes(meshes):
    """trimesh.util.concatenate for (verts, faces, normals) triples: vertices stacked, face indices offset (marching_cubes.py:150)."""
    if not meshes:
        return None
    off, vs, fs, ns = 0, [], [], []
    for v, f, n in meshes:
        vs.append(v.double())
        fs.append(f.long() + off)
        ns.append(n)
        off += v.shape[0]
    return torch.cat(vs), torch.cat(fs), torch.cat(ns)


def _finish_mesh(mesh, output_path, merge: bool):
    """the reference's tail: [merge_vertices(digits_vertex=6),] export(path) - only when a path is given (the mesh is returned either way)"""
    if mesh is None or output_path is None:
        return mesh
    from sdfstudio_amd.utils import mesh_io

    verts, faces, normals = mesh
    if merge:
        verts, faces, normals = mesh_io.merge_vertices(verts, faces, normals, digits_vertex=6)
    mesh_io.export_ply(output_path, verts, faces, normals)
    return verts, faces, normals


def _coarse_mask_lookup(coarse_mask: torch.Tensor, pts: torch.Tensor) -> torch.Tensor:
    """marching_cubes.py:27-29,68-70,97-101: the scene box's coarse binary grid sampled at points [..., 3] (grid_sample's default
    bilinear lookup on the (z, y, x)-permuted grid, > 0)."""
    cm = coarse_mask.permute(2, 1, 0)[None, None].to(device=pts.device, dtype=torch.float32)
    flat = pts.reshape(1, 1, 1, -1, 3)
    return (torch.nn.functional.grid_sample(cm, flat, align_corners=False)[0, 0, 0, 0] > 0.0).reshape(pts.shape[:-1])


@torch.no_grad()
def _field_or_sdf(field, sdf, device):
    """The first argument of the reference's drivers is the sdf callable (marching_cubes.py:15-17, :218-220); here it may also be an
    SDFField, whose MODE_SDF kernels then evaluate the lattice.  Returns (sdf callable, device)."""
    if sdf is None and field is not None and not hasattr(field, "forward_geonetwork") and callable(field):
        sdf, field = field, None  # a reference-style call: get_surface_sliding(sdf, ...)
    if sdf is None and field is None:
        raise TypeError("an SDFField or an sdf(points [P, 3]) -> [P] callable is required")
    if device is None:
        device = field.encoding.params.device if field is not None else torch.device("cuda", torch.cuda.current_device())
    return (sdf if sdf is not None else (lambda p: sdf_on_points(field, p))), device


def _no_simplify(simplify_mesh: bool):
    if simplify_mesh:  # marching_cubes.py:161-167, :337-343: pymeshlab's quadric edge collapse on the written file
        raise NotImplementedError("simplify_mesh=True: pymeshlab's mesh simplification is not built (pass simplify_mesh=False)")


def get_surface_sliding(field=None, resolution: int = 512, bounding_box_min=(-1.0, -1.0, -1.0), bounding_box_max=(1.0, 1.0, 1.0),
                        return_mesh: bool = True, level: float = 0.0, coarse_mask: Optional[torch.Tensor] = None, crop: int = 512,
                        device=None, return_volumes: bool = False, sdf: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                        output_path=None, simplify_mesh: bool = False):
    """marching_cubes.py:15-168: per crop^3 block (the reference fixes crop = 512) the coarse-to-fine sdf evaluation, then marching cubes
    ON THE DEVICE (libsdfmesh.so), the crop's offset added in double as the reference adds it, the crops concatenated.
    ``field``: an SDFField (its MODE_SDF kernels evaluate the lattice) - or pass ``sdf(points [P,3]) -> [P]`` as the reference does.
    Returns (verts [V,3] float64, faces [F,3] int64, normals [V,3] float32) on the device, or None without a surface;
    return_mesh=False: the list of per-crop (verts, faces, normals); return_volumes=True: the list of (lo, hi, volume [crop^3]).
    ``level`` is overwritten with 0 as at marching_cubes.py:33.  ``output_path``: the reference's return_mesh=False branch (:156-160) -
    merge_vertices(digits_vertex=6) and the binary .ply (utils/mesh_io.py; the merged mesh is returned as well).  pymeshlab's
    simplification (:161-167) is not built."""
    assert resolution % crop == 0 and crop % 8 == 0
    _no_simplify(simplify_mesh)
    level = 0.0  # marching_cubes.py:33
    fn, dev = _field_or_sdf(field, sdf, device)
    nblk = resolution // crop
    edges = [np.linspace(bounding_box_min[a], bounding_box_max[a], nblk + 1) for a in range(3)]
    results = []
    for i in range(nblk):
        for j in range(nblk):
            for k in range(nblk):
                lo = (edges[0][i], edges[1][j], edges[2][k])
                hi = (edges[0][i + 1], edges[1][j + 1], edges[2][k + 1])
                # np.linspace in double, cast to float: the reference's lattice (marching_cubes.py:51-56)
                ax = [torch.from_numpy(np.linspace(lo[a], hi[a], crop)).float().to(dev) for a in range(3)]
                xx, yy, zz = torch.meshgrid(*ax, indexing="ij")
                pts = torch.stack([xx, yy, zz], 0)
                current_mask, valid = None, None
                if coarse_mask is not None:
                    current_mask = _coarse_mask_lookup(coarse_mask, pts.permute(1, 2, 3, 0))
                    valid = lambda p: _coarse_mask_lookup(coarse_mask, p)  # noqa: E731
                z, _, _ = evaluate_crop_pyramid(fn, pts, float(hi[0] - lo[0]), valid)
                vol = z.reshape(crop, crop, crop)
                if current_mask is not None:
                    inside = vol[current_mask]
                    if inside.numel() == 0 or float(inside.min()) > level or float(inside.max()) < level:
                        continue
                if float(z.min()) > level or float(z.max()) < level:
                    continue
                if return_volumes:
                    results.append((lo, hi, vol))
                    continue
                spacing = tuple((hi[a] - lo[a]) / (crop - 1) for a in range(3))
                verts, faces, normals, _ = marching_cubes(vol, level, spacing=spacing, mask=current_mask)
                verts = verts.double() + torch.tensor(lo, dtype=torch.float64, device=verts.device)
                results.append((verts, faces, normals))
    if return_volumes or (not return_mesh and output_path is None):
        return results
    return _finish_mesh(concatenate_meshes(results), output_path, merge=True)


@torch.no_grad()
def get_surface_occupancy(occupancy_fn: Callable[[torch.Tensor], torch.Tensor], resolution: int = 512, bounding_box_min=(-1.0, -1.0, -1.0),
                          bounding_box_max=(1.0, 1.0, 1.0), level: float = 0.5, device=None, chunk: int = 1 << 22, output_path=None,
                          return_mesh: bool = True):
    """marching_cubes.py:171-216 (UniSurf: occupancy = sigmoid(10 sdf), level 0.5): one resolution^3 lattice, marching cubes on the
    device.  Returns (verts float64, faces int32, normals) or None ("no surface skip"); ``output_path``: the .ply as at :212-213 (no merge).
    device = None (the reference's default): the current HIP device - the lattice is evaluated and meshed there (libsdfmesh.so has no host path)."""
    n = int(resolution)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    ax = [torch.from_numpy(np.linspace(bounding_box_min[a], bounding_box_max[a], n)).float().to(device) for a in range(3)]
    xx, yy, zz = torch.meshgrid(*ax, indexing="ij")
    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
    z = torch.cat([occupancy_fn(pts[a:a + chunk].contiguous()).reshape(-1) for a in range(0, pts.shape[0], chunk)])
    if float(z.min()) > level or float(z.max()) < level:
        return None
    spacing = tuple((bounding_box_max[a] - bounding_box_min[a]) / (n - 1) for a in range(3))
    verts, faces, normals, _ = marching_cubes(z.reshape(n, n, n), level, spacing=spacing)
    verts = verts.double() + torch.tensor(tuple(float(t) for t in bounding_box_min), dtype=torch.float64, device=verts.device)
    return _finish_mesh((verts, faces, normals), output_path, merge=False)


_max_pool_3d = torch.nn.MaxPool3d(3, stride=1, padding=1)


@torch.no_grad()
def get_surface_sliding_with_contraction(field=None, resolution: int = 512, bounding_box_min=(-1.0, -1.0, -1.0), bounding_box_max=(1.0, 1.0, 1.0),
                                         coarse_mask: Optional[torch.Tensor] = None, inv_contraction: Optional[Callable] = None,
                                         max_range: float = 32.0, crop: int = 512, device=None,
                                         sdf: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, output_path=None,
                                         merge: bool = True, return_mesh: bool = True, level: float = 0.0, simplify_mesh: bool = False):
    """marching_cubes.py:218-335 (scenes trained under a scene contraction; scripts/extract_mesh.py:95-107): per crop the sdf is evaluated
    only where the visibility grid ``coarse_mask`` [1, 1, D, H, W] (over the contracted cube [-2, 2]^3, grid_sample's lookup at points / 2)
    is set, everything else starts at 100 and is replaced by the 3^3 minimum of its neighbourhood ("to remove masked marching cube
    artefacts"), marching cubes runs with the crop's mask ON THE DEVICE, and the concatenated vertices go through ``inv_contraction`` and
    the clip to [-max_range, max_range] in double, as the reference applies them.  Level 0 (marching_cubes.py:235).
    Returns (verts [V,3] float64, faces [F,3] int64, normals [V,3] float32) or None.  ``merge``: the reference's
    ``combined.merge_vertices(digits_vertex=6)`` at :321, BEFORE the inverse contraction, whether or not a file is written (utils/mesh_io.py;
    merge=False returns the plain concatenation of the crops); ``output_path``: the binary .ply, as at :330-334."""
    assert resolution % crop == 0 and coarse_mask is not None
    _no_simplify(simplify_mesh)
    level = 0.0  # marching_cubes.py:235 (the argument is overwritten there as well); return_mesh: the mesh is returned either way
    fn, dev = _field_or_sdf(field, sdf, device)
    cm = coarse_mask.to(device=dev, dtype=torch.float32)
    nblk = resolution // crop
    edges = [np.linspace(bounding_box_min[a], bounding_box_max[a], nblk + 1) for a in range(3)]
    meshes = []
    for i in range(nblk):
        for j in range(nblk):
            for k in range(nblk):
                lo = (edges[0][i], edges[1][j], edges[2][k])
                hi = (edges[0][i + 1], edges[1][j + 1], edges[2][k + 1])
                ax = [torch.from_numpy(np.linspace(lo[a], hi[a], crop)).float().to(dev) for a in range(3)]
                pts = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1)  # [n, n, n, 3]
                current = torch.nn.functional.grid_sample(cm, pts[None] * 0.5, align_corners=False)  # [1, 1, n, n, n]
                valid = current.reshape(-1) > 0
                flat = pts.reshape(-1, 3)
                pts_sdf = torch.full((flat.shape[0],), 100.0, device=dev)
                if bool(valid.any()):
                    pts_sdf[valid] = fn(flat[valid].contiguous())
                vol = pts_sdf.reshape(1, 1, crop, crop, crop)
                min_sdf = _max_pool_3d(vol * -1.0) * -1.0
                inside = (current > 0.0).float()
                vol = (vol * inside + min_sdf * (1.0 - inside))[0, 0]
                cur = current[0, 0] > 0.0
                sel = vol[cur]
                if sel.numel() == 0 or float(sel.min()) > level or float(sel.max()) < level:
                    continue
                if float(vol.min()) > level or float(vol.max()) < level:
                    continue
                spacing = tuple((hi[a] - lo[a]) / (crop - 1) for a in range(3))
                verts, faces, normals, _ = marching_cubes(vol, level, spacing=spacing, mask=cur)
                meshes.append((verts.double() + torch.tensor(lo, dtype=torch.float64, device=verts.device), faces, normals))
    mesh = concatenate_meshes(meshes)
    if mesh is None:
        return None
    verts, faces, normals = mesh
    if merge:
        from sdfstudio_amd.utils import mesh_io

        verts, faces, normals = mesh_io.merge_vertices(verts, faces, normals, digits_vertex=6)
    if inv_contraction is not None:
        verts = torch.clamp(inv_contraction(verts), -max_range, max_range)
    return _finish_mesh((verts, faces, normals), output_path, merge=False)
