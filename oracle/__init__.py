"""CPU oracle for the SDF volume-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sdfstudio_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

The oracle is a plain PyTorch (CPU, fp32 or fp64) restatement of the reference's
algorithm for the path ``BASELINE.json:north_star`` names.  Every function
cites the reference ``file:line`` it follows (paths relative to the upstream
``nerfstudio/`` package).

Parity status
-------------
* Everything except the multi-resolution hash grid is pinned: ``tests/golden``
  holds vectors produced by importing the reference's own Python in the build
  container (``tests/golden/make_golden.py``) and the oracle reproduces them.
* The hash grid arithmetic lives in ``tinycudann`` (NVlabs/tiny-cuda-nn, git
  master, un-pinned, CUDA only, absent from the reference tree).  Its published
  algorithm is restated in ``oracle/hashgrid.py``; **parity at that boundary is
  unpinned** (no reference test, fixture or runnable binary exists for it).
"""
