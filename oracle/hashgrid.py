"""Oracle (test infrastructure): multi-resolution hash grid, restated in PyTorch.

The arithmetic is that of ``tinycudann``'s ``GridEncoding`` (NVlabs/tiny-cuda-nn,
``include/tiny-cuda-nn/encodings/grid.h``; installed by the reference from git
master with no version pin: reference ``README.md:48``, ``Dockerfile:96``).  The
package is CUDA-only and not vendored under ``/root/reference`` so it cannot be
built or run here: **parity unpinned** at this boundary.  Call sites this
restatement serves: ``fields/sdf_field.py:230-241`` (16 levels, F=2, T=2^19,
Smoothstep) and ``fields/density_fields.py:89-94`` (5 levels, F=2, T=2^17,
Linear).

Published algorithm (restated):
  per level l:  scale_l = fp32( exp2(l * log2(per_level_scale)) * base_res - 1 )  (evaluated in double, see make_levels)
                res_l   = ceil(scale_l) + 1
                size_l  = min(next_multiple(res_l^3, 8), 2^log2_hashmap_size)
  position:     pos = fmaf(scale_l, x, 0.5) (ONE rounding, tcnn grid.h pos_fract) ; cell = floor(pos) ; w = pos - cell
                Smoothstep: w <- w^2 (3 - 2 w)
  corner index: dense  (res_l^3 <= size_l):  (cx + cy*res + cz*res^2)        mod size_l
                hashed (otherwise):          (cx*1 ^ cy*2654435761 ^ cz*805459861) mod size_l   (uint32)
  output:       [P, L*F] level-major, trilinear blend of the 8 corner feature vectors.
Autograd supplies d/dx, d/dtable and the mixed second derivative.
"""
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

PRIME_Y = 2654435761
PRIME_Z = 805459861


@dataclass
class GridLevels:
    n_levels: int
    n_features: int
    log2_hashmap_size: int
    base_resolution: int
    per_level_scale: float
    smoothstep: bool
    scale: np.ndarray  # float32 [L]
    resolution: np.ndarray  # int64 [L]
    size: np.ndarray  # int64 [L]  entries in level
    offset: np.ndarray  # int64 [L+1] entry offsets
    hashed: np.ndarray  # bool [L]

    @property
    def n_entries(self) -> int:
        return int(self.offset[-1])

    @property
    def n_params(self) -> int:
        return self.n_entries * self.n_features

    @property
    def n_output_dims(self) -> int:
        return self.n_levels * self.n_features


def make_levels(
    n_levels: int,
    n_features: int,
    log2_hashmap_size: int,
    base_resolution: int,
    per_level_scale: float,
    smoothstep: bool,
) -> GridLevels:
    """Per-level scale / resolution / table extent.  The scale is evaluated in DOUBLE from the fp32 per_level_scale and rounded
    to fp32 once: tcnn evaluates exp2f / log2 in single precision with the host libm, whose last bit is library dependent
    (glibc exp2f vs numpy's differ by up to 1.5e-3 at scale 4095, which moves the finest cells by 7e-4 of their size and the
    analytic normal of a linear 8-feature level by 0.1); one correctly rounded value is what both sides can reproduce."""
    l2 = np.log2(np.float64(np.float32(per_level_scale)))
    scale = np.zeros(n_levels, np.float32)
    res = np.zeros(n_levels, np.int64)
    size = np.zeros(n_levels, np.int64)
    hashed = np.zeros(n_levels, bool)
    offset = np.zeros(n_levels + 1, np.int64)
    for lvl in range(n_levels):
        s = np.float32(np.exp2(np.float64(lvl) * l2) * np.float64(base_resolution) - 1.0)
        r = int(np.ceil(s)) + 1
        n = r**3
        n = (n + 7) // 8 * 8
        n = min(n, 1 << log2_hashmap_size)
        scale[lvl], res[lvl], size[lvl] = s, r, n
        hashed[lvl] = r**3 > n
        offset[lvl + 1] = offset[lvl] + n
    return GridLevels(
        n_levels, n_features, log2_hashmap_size, base_resolution, float(per_level_scale), smoothstep,
        scale, res, size, offset, hashed,
    )


def corner_index(cx: torch.Tensor, cy: torch.Tensor, cz: torch.Tensor, res: int, size: int, hashed: bool):
    """uint32 index arithmetic carried in int64."""
    m = 0xFFFFFFFF
    if hashed:
        idx = (cx & m) ^ ((cy * PRIME_Y) & m) ^ ((cz * PRIME_Z) & m)
    else:
        idx = (cx + cy * res + cz * res * res) & m
    return idx % size


def _fma_half(x: torch.Tensor, scale: float) -> torch.Tensor:
    """fmaf(scale, x, 0.5): tcnn's pos_fract rounds the multiply-add ONCE.  In fp32 the exact product of two fp32 numbers fits
    a double and so does its sum with 0.5 at these magnitudes (48 significant bits), so rounding the fp64 value to fp32 IS the
    fused result; the correction is detached, the autograd path stays d pos / d x = scale.  (fp64 inputs: plain arithmetic.)"""
    pos = x * scale + 0.5
    if x.dtype == torch.float32:
        fused = (x.detach().double() * float(np.float32(scale)) + 0.5).float()
        pos = pos + (fused - pos.detach())
    return pos


def level_cell(x: torch.Tensor, lv: GridLevels, lvl: int):
    """Cell of every point on level lvl: (idx [P,8] int64 table-entry indices INCLUDING the level offset, corner k = bit 0 -> +x,
    bit 1 -> +y, bit 2 -> +z; w [P,3] interpolation weight along each axis after the optional smoothstep)."""
    scale = float(lv.scale[lvl])
    res, size, off = int(lv.resolution[lvl]), int(lv.size[lvl]), int(lv.offset[lvl])
    pos = _fma_half(x, scale)
    cell = torch.floor(pos)
    w = pos - cell
    if lv.smoothstep:
        w = w * w * (3.0 - 2.0 * w)
    c = cell.detach().to(torch.int64)
    idx = []
    for corner in range(8):
        bx, by, bz = corner & 1, (corner >> 1) & 1, (corner >> 2) & 1
        idx.append(off + corner_index(c[:, 0] + bx, c[:, 1] + by, c[:, 2] + bz, res, size, bool(lv.hashed[lvl])))
    return torch.stack(idx, dim=-1), w


def grid_encode(x: torch.Tensor, table: torch.Tensor, lv: GridLevels) -> torch.Tensor:
    """x: [P,3] in [0,1]; table: [n_entries, F]; returns [P, L*F].  The blend is the same sequence of operations for every table; how
    the corner entries are FETCHED depends on the table's size: one indexing operation per (level, corner) normally (fastest on the
    host: bench.py times this function as its CPU baseline), ONE indexing operation for all 8 L corners when the table is huge -
    autograd builds a table-sized gradient per indexing operation, and 128 of them for the 2.1 GB table of BASELINE config 5 would
    cost minutes where one costs seconds."""
    cells = [level_cell(x, lv, lvl) for lvl in range(lv.n_levels)]
    one_gather = table.numel() > (1 << 27)
    if one_gather:
        idx_all = torch.stack([c[0] for c in cells], dim=1)  # [P, L, 8]
        vals = table[idx_all.reshape(-1)].view(x.shape[0], lv.n_levels, 8, table.shape[-1])
    outs: List[torch.Tensor] = []
    for lvl in range(lv.n_levels):
        idx, w = cells[lvl]
        acc = 0.0
        for corner in range(8):
            bx, by, bz = corner & 1, (corner >> 1) & 1, (corner >> 2) & 1
            wx = w[:, 0] if bx else 1.0 - w[:, 0]
            wy = w[:, 1] if by else 1.0 - w[:, 1]
            wz = w[:, 2] if bz else 1.0 - w[:, 2]
            acc = acc + (wx * wy * wz)[:, None] * (vals[:, lvl, corner] if one_gather else table[idx[:, corner]])
        outs.append(acc)
    return torch.cat(outs, dim=-1)
