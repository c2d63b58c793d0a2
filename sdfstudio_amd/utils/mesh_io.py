"""The last step of mesh extraction: what the reference does with the concatenated crops before the file exists
(nerfstudio/utils/marching_cubes.py:152-160, 212-213, 320-334): ``combined.merge_vertices(digits_vertex=6)`` and ``combined.export(path)``.

Both are trimesh calls (trimesh is not vendored by the reference and is absent from this image: PARITY UNPINNED for this file - it
restates the documented behaviour of ``Trimesh.merge_vertices`` and of trimesh's binary PLY exporter and is checked by round trips and
by properties, not against trimesh):

* ``merge_vertices(verts, faces, normals, digits_vertex=6)``: vertices whose coordinates agree after rounding to ``digits_vertex``
  decimals AND whose normals agree after rounding to 2 decimals (trimesh's defaults ``merge_norm=False``, ``digits_norm=2``: a mesh
  that carries vertex normals only merges vertices with the same normal) become one vertex - the first of the group in the original
  order; faces are re-indexed; unreferenced vertices are dropped first, as trimesh does.  The seams between 512^3 crops are where this
  matters: both crops emit the vertices on their shared lattice plane.
* ``export_ply``: binary little-endian PLY, ``float`` x y z (+ nx ny nz), faces as ``list uchar int vertex_indices`` - the element and
  property layout trimesh writes, so MeshLab / pymeshlab (the reference's optional simplification step, :161-167) read it the same way.

Everything runs on the device the mesh lives on; only ``export_ply`` copies (once) to the host.  The simplification itself
(pymeshlab's quadric edge collapse) is out of scope: DESIGN.md section 1.
"""
import os
from typing import Optional, Tuple

import numpy as np
import torch


def _row_keys(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """unique rows of an integer matrix [V, C]: (inverse [V], first-occurrence index of every unique row)."""
    _, inverse = torch.unique(x, dim=0, return_inverse=True)
    n = int(inverse.max()) + 1 if inverse.numel() else 0
    first = torch.full((n,), x.shape[0], dtype=torch.long, device=x.device)
    first.scatter_reduce_(0, inverse, torch.arange(x.shape[0], device=x.device), reduce="amin")
    return inverse, first


@torch.no_grad()
def merge_vertices(verts: torch.Tensor, faces: torch.Tensor, normals: Optional[torch.Tensor] = None, digits_vertex: int = 6,
                   digits_norm: int = 2, merge_norm: bool = False):
    """See the module docstring.  verts [V,3] (any float type, kept), faces [F,3] integer, normals [V,3] or None.
    Returns (verts, faces int64, normals or None); vertex order = order of first occurrence in the input."""
    faces = faces.long()
    V = verts.shape[0]
    referenced = torch.zeros(V, dtype=torch.bool, device=verts.device)
    referenced[faces.reshape(-1)] = True
    cols = [torch.round(verts.double() * (10.0 ** digits_vertex)).long()]
    if normals is not None and not merge_norm:
        cols.append(torch.round(normals.double() * (10.0 ** digits_norm)).long())
    key = torch.cat(cols, dim=1)
    keep_idx = torch.nonzero(referenced).reshape(-1)
    inverse_ref, first_ref = _row_keys(key[keep_idx])
    first = keep_idx[first_ref]                      # original index of every merged vertex's representative
    order = torch.argsort(first)                     # representatives in input order
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.shape[0], device=order.device)
    remap = torch.full((V,), -1, dtype=torch.long, device=verts.device)
    remap[keep_idx] = rank[inverse_ref]
    sel = first[order]
    return verts[sel], remap[faces], (None if normals is None else normals[sel])


def export_ply(path, verts: torch.Tensor, faces: torch.Tensor, normals: Optional[torch.Tensor] = None) -> None:
    """Binary little-endian PLY (module docstring).  Vertices and normals are written as float32, as trimesh's exporter does."""
    v = np.ascontiguousarray(verts.detach().cpu().numpy().astype("<f4"))
    f = np.ascontiguousarray(faces.detach().cpu().numpy().astype("<i4"))
    assert v.ndim == 2 and v.shape[1] == 3 and f.ndim == 2 and f.shape[1] == 3
    header = ["ply", "format binary_little_endian 1.0", "comment sdfstudio_amd (layout of trimesh's PLY exporter)",
              f"element vertex {v.shape[0]}", "property float x", "property float y", "property float z"]
    vdt = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if normals is not None:
        n = np.ascontiguousarray(normals.detach().cpu().numpy().astype("<f4"))
        assert n.shape == v.shape
        header += ["property float nx", "property float ny", "property float nz"]
        vdt += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
    header += [f"element face {f.shape[0]}", "property list uchar int vertex_indices", "end_header"]
    vrec = np.empty(v.shape[0], dtype=vdt)
    vrec["x"], vrec["y"], vrec["z"] = v[:, 0], v[:, 1], v[:, 2]
    if normals is not None:
        vrec["nx"], vrec["ny"], vrec["nz"] = n[:, 0], n[:, 1], n[:, 2]
    frec = np.empty(f.shape[0], dtype=[("count", "u1"), ("index", "<i4", (3,))])
    frec["count"] = 3
    frec["index"] = f
    tmp = str(path) + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        fh.write(vrec.tobytes())
        fh.write(frec.tobytes())
    os.replace(tmp, str(path))


def load_ply(path):
    """Reader for the files ``export_ply`` (and trimesh's exporter, same layout) writes: (verts [V,3] f32, faces [F,3] i32, normals or None)
    as numpy arrays.  Test infrastructure for the round trip; not a general PLY parser."""
    with open(str(path), "rb") as fh:
        data = fh.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii").splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0", lines[:2]
    nv = nf = 0
    vprops, in_vertex = [], False
    for ln in lines:
        t = ln.split()
        if t[:2] == ["element", "vertex"]:
            nv, in_vertex = int(t[2]), True
        elif t[:2] == ["element", "face"]:
            nf, in_vertex = int(t[2]), False
        elif t[0] == "property" and in_vertex:
            assert t[1] == "float", ln
            vprops.append(t[2])
    vdt = np.dtype([(p, "<f4") for p in vprops])
    vrec = np.frombuffer(data, dtype=vdt, count=nv, offset=end)
    fdt = np.dtype([("count", "u1"), ("index", "<i4", (3,))])
    frec = np.frombuffer(data, dtype=fdt, count=nf, offset=end + nv * vdt.itemsize)
    assert end + nv * vdt.itemsize + nf * fdt.itemsize == len(data), "trailing or missing bytes"
    assert nf == 0 or bool((frec["count"] == 3).all())
    verts = np.stack([vrec["x"], vrec["y"], vrec["z"]], 1)
    normals = np.stack([vrec["nx"], vrec["ny"], vrec["nz"]], 1) if "nx" in vprops else None
    return verts, frec["index"].copy(), normals
