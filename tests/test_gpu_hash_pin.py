"""The kernels' table indexing against the oracle, BIT FOR BIT (-m gpu).

Every kernel of the library that touches a hash table (fused encode, grid backward, proposal fields, standalone encode) calls one
device function, grid_cell (csrc/point_kernels.h); sdfhip_grid_cell_dump exposes its result.  Integer work: the 8 corner entry
indices of every (point, level) must EQUAL oracle/hashgrid.py::level_cell's - whose hashed branch is pinned on the reference's own
HashEncoding.hash_fn by tests/test_cpu_hash_pin.py - for BASELINE config 2's grid (16 x 2 x 2^19, Smoothstep), the proposal grids
(5 levels, 2^17, Linear) and config 5's (16 x 8 x 2^22, Linear), inside the unit cube, on its faces and outside it (get_sdf takes
uncontracted positions, sdf_field.py:412-418).  The interpolation weights agree to 2 ulp (the device contracts the smoothstep
polynomial into fused multiply-adds)."""
import ctypes
import math

import pytest
import torch

from oracle import hashgrid

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,L,F,log2_t,base,max_res,smooth", [
    ("config 2 field", 16, 2, 19, 16, 2048, True),
    ("proposal 0", 5, 2, 17, 16, 64, False),
    ("proposal 1", 5, 2, 17, 16, 256, False),
    ("config 5 field", 16, 8, 22, 64, 4096, False),
    ("small golden", 8, 2, 11, 4, 128, True),
])
def test_kernel_cell_indices_equal_oracle_bit_for_bit(device, name, L, F, log2_t, base, max_res, smooth):
    from sdfstudio_amd import _lib

    growth = math.exp((math.log(max_res) - math.log(base)) / (L - 1))
    lv = hashgrid.make_levels(L, F, log2_t, base, growth, smooth)
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(20000, 3, generator=gen)
    x[:2000] = torch.rand(2000, 3, generator=gen) * 3.0 - 1.0           # outside the unit cube (wraps like tiny-cuda-nn)
    x[2000:2100] = torch.randint(0, 2, (100, 3), generator=gen).float()  # corners of the cube
    x[2100:2400, 0] = 1.0                                                # the x == 1 face (the dense levels' only wrap inside the cube)
    x[2400:2500] = torch.randint(0, 65, (100, 3), generator=gen).float() / 64.0  # exactly on coarse cell boundaries
    P = x.shape[0]
    cfg = _lib.GridCfg(L, F, log2_t, base, growth, 1 if smooth else 0)
    xd = x.to(device).contiguous()
    idx = torch.zeros(P, L, 8, dtype=torch.int32, device=device)
    w = torch.zeros(P, L, 3, dtype=torch.float32, device=device)
    lib = _lib.load()
    _lib.check(lib.sdfhip_grid_cell_dump(ctypes.byref(cfg), _lib.ptr(xd), P, ctypes.c_void_p(idx.data_ptr()), _lib.ptr(w), _lib.stream()),
               "grid_cell_dump")
    got = idx.cpu().to(torch.int64) & 0xFFFFFFFF
    gw = w.cpu()
    n_hashed = 0
    for lvl in range(L):
        want, ww = hashgrid.level_cell(x, lv, lvl)
        n_hashed += int(lv.hashed[lvl])
        bad = (got[:, lvl] != want).any(dim=-1)
        assert not bool(bad.any()), (f"{name} level {lvl} ({'hashed' if lv.hashed[lvl] else 'dense'}): {int(bad.sum())} points index differently, "
                                     f"first x = {x[bad][0].tolist()} got {got[bad][0, lvl].tolist()} want {want[bad][0].tolist()}")
        assert int(want.min()) >= int(lv.offset[lvl]) and int(want.max()) < int(lv.offset[lvl + 1])
        assert (gw[:, lvl] - ww).abs().max().item() <= 3e-7, (name, lvl, (gw[:, lvl] - ww).abs().max().item())
    assert n_hashed >= 1
