"""The `mesh` leg of bench.py, run as a CHILD process (bench.py::mesh_leg): marching cubes of one 512^3 crop - the reference's crop size,
nerfstudio/utils/marching_cubes.py:31 - through the Python binding of libsdfmesh.so.  Prints ONE JSON object.

A child: its own HIP context and allocator beside the parent's; whatever happens here, the parent's bench line survives."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK_HBM_GBS = 8000.0


def _manifold_stats(faces, n_verts):
    """(every undirected edge is used exactly twice, once per direction; Euler characteristic V - E + F) of a triangle list."""
    import torch

    f = faces.long()
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = e[:, 0] * n_verts + e[:, 1]
    rev = e[:, 1] * n_verts + e[:, 0]
    uniq, cnt = torch.unique(key, return_counts=True)
    closed = bool((cnt == 1).all()) and bool(torch.isin(rev, uniq).all())
    return closed, n_verts - uniq.numel() // 2 + f.shape[0]


def main():
    import torch

    from sdfstudio_amd import _mesh
    from sdfstudio_amd import build as _build
    from sdfstudio_amd.utils.marching_cubes import marching_cubes

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    reps = 10
    dev = torch.device("cuda", 0)
    ax = torch.linspace(-1, 1, n, device=dev)
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    vol = torch.minimum(torch.sqrt(xx * xx + yy * yy + zz * zz) - 0.55 - 0.03 * torch.sin(9 * xx) * torch.sin(7 * yy) * torch.sin(5 * zz),
                        torch.sqrt((xx - 0.8) ** 2 + (yy - 0.8) ** 2 + (zz - 0.8) ** 2) - 0.1).contiguous()
    del xx, yy, zz
    spacing = (2.0 / (n - 1),) * 3
    verts, faces, normals, values = marching_cubes(vol, 0.0, spacing=spacing)  # warm-up; the call the reference makes per crop
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        v, f, nr, val = _mesh.marching_cubes_device(vol, 0.0)
    e1.record()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / reps * 1e3
    gpu_ms = e0.elapsed_time(e1) / reps
    V, F = int(v.shape[0]), int(f.shape[0])
    # sanity on the spot (the parity tests proper: tests/test_gpu_zz_mesh.py, tools/mesh_gpu_check.cpp): two closed surfaces
    closed, euler = _manifold_stats(f, V)
    again = _mesh.marching_cubes_device(vol, 0.0)
    reproducible = all(torch.equal(a, b) for a, b in zip((v, f, nr, val), again))
    P = n ** 3
    alg = 4 * P + 28 * V + 12 * F  # the volume once + the mesh written: verts 12 V, normals 12 V, values 4 V, faces 12 F
    # HBM bytes per crop from the committed rocprofv3 --pmc passes of the standalone check on the same crop size (tools/profile_mesh.sh,
    # tools/mesh_pmc_summary.py: FETCH_SIZE x 2 + WRITE_SIZE, calibrated in that run on the volume's known byte count); stale when the
    # library was built from other sources than the profiled one
    traffic = {"traffic": None}
    pm = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("r") and f.endswith("_pmc_summary.json") and "_mesh_" in f)
    if pm:
        loaded = [(f, json.load(open(os.path.join(ROOT, "profiles", f)))) for f in pm]
        same = [x for x in loaded if x[1].get("mesh_library_digest") == _build.mesh_source_digest()]
        f, j = (same or loaded)[-1]
        traffic = {"traffic": j["per_crop"]["hbm_bytes"], "traffic_source": f"profiles/{f} (sum over the library's kernels on one 512^3 crop: FETCH_SIZE x 2 + "
                                                                             "WRITE_SIZE, separate rocprofv3 --pmc passes)",
                   "traffic_over_algorithmic": round(j["per_crop"]["hbm_bytes"] / alg, 3), "traffic_library_digest": j.get("mesh_library_digest"),
                   "traffic_stale": j.get("mesh_library_digest") != _build.mesh_source_digest(),
                   "kernel_time_us_profiled": j["per_crop"]["kernel_time_us_sum_of_medians"],
                   "stream_kernel": {"kernel": "mc_pointbits_kernel", "median_us": j["kernels"]["mc_pointbits_kernel"]["median_us"],
                                     "GBps_on_volume_bytes": round(4 * 512 ** 3 / j["kernels"]["mc_pointbits_kernel"]["median_us"] / 1e3, 1),
                                     "frac_of_8TBps": round(4 * 512 ** 3 / j["kernels"]["mc_pointbits_kernel"]["median_us"] / 1e3 / PEAK_HBM_GBS, 4)}}
    out = {"workload": f"marching cubes (Lewiner, scikit-image's arrays bit for bit) of one {n}^3 crop of an analytic SDF (bumpy sphere + small "
                       "sphere), volume resident in HBM; what nerfstudio/utils/marching_cubes.py:125-134 does per crop with skimage on the CPU",
           "points": P, "vertices": V, "faces": F, "ms": round(gpu_ms, 4), "ms_wall_incl_host": round(wall_ms, 4),
           "value": round(P / (gpu_ms * 1e-3), 1), "unit": "lattice points/s",
           "closed_surface": bool(closed), "euler_characteristic": int(euler), "bit_reproducible": bool(reproducible),
           "mesh_library_digest": _build.mesh_source_digest(),
           "roofline": {"kernels": "mc_pointbits_kernel (the one pass over the volume: 1 bit per point out) + cell bits / list / classify / keys / "
                                   "vertices / faces over bit arrays and surface cells; no sort, no library",
                        "bound": "hbm", "achieved": round(alg / (gpu_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(alg / (gpu_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), **traffic,
                        "algorithmic_bytes": alg, "achieved_is": "4 B per lattice point read + 28 B per vertex + 12 B per face written, over the WHOLE "
                                                                  "call through the Python binding (count + emit, incl. its one host read and the "
                                                                  "output allocations)"},
           "round5": "2.30 ms per crop (frac 0.03): profiles/r6_mesh_before_check.jsonl, the round-5 library re-measured this round"}
    # the whole operation the reference's `ns-extract-mesh --resolution 512` performs on BASELINE config 2's field (geometric init: a sphere of
    # radius 0.5): get_surface_sliding = lattice + coarse-to-fine sdf evaluation (MODE_SDF kernels of libsdfhip.so) + marching cubes
    # (libsdfmesh.so) + crop offset.  Guarded separately: the marching-cubes figures above stand whatever happens here.
    try:
        del vol, v, f, nr, val, again, verts, faces, normals, values
        torch.cuda.empty_cache()
        import bench as B
        from sdfstudio_amd.utils.marching_cubes import evaluate_crop_pyramid, get_surface_sliding, sdf_on_points

        model = B.build_model(dev).eval()
        get_surface_sliding(model.field, resolution=128, crop=128)  # warm-up (allocator, kernels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mesh = get_surface_sliding(model.field, resolution=n, crop=n)
        torch.cuda.synchronize()
        ex_ms = (time.perf_counter() - t0) * 1e3
        ax1 = torch.linspace(-1, 1, n, device=dev)
        pts = torch.stack(torch.meshgrid(ax1, ax1, ax1, indexing="ij"), 0)
        _, counts, _ = evaluate_crop_pyramid(lambda q: sdf_on_points(model.field, q), pts, 2.0)
        del pts
        out["extract_mesh"] = {"workload": f"get_surface_sliding(resolution = crop = {n}) on BASELINE config 2's field at geometric init: what scripts/extract_mesh.py:125-133 "
                                           "runs (without its .ply export / simplification)",
                               "ms": round(ex_ms, 3), "vertices": None if mesh is None else int(mesh[0].shape[0]),
                               "faces": None if mesh is None else int(mesh[1].shape[0]),
                               "network_evaluations_per_pyramid_level": [int(c) for c in counts], "lattice_points": P,
                               "fraction_of_lattice_evaluated": round(sum(counts) / P, 5)}
    except Exception as e:  # noqa: BLE001
        out["extract_mesh"] = {"error": repr(e)[:300]}
    # the reference's own CPU step beside it, when the box has the build container's scikit-image interpreter: a bounded sample (256^3)
    py = "/opt/conda/bin/python3.9"
    if os.path.exists(py):
        code = ("import numpy as np, time, warnings\nwarnings.filterwarnings('ignore')\nfrom skimage import measure\nimport skimage\n"
                "n=256\nax=np.linspace(-1,1,n,dtype=np.float32)\nzz,yy,xx=np.meshgrid(ax,ax,ax,indexing='ij')\n"
                "vol=np.minimum(np.sqrt(xx*xx+yy*yy+zz*zz)-0.55-0.03*np.sin(9*xx)*np.sin(7*yy)*np.sin(5*zz),np.sqrt((xx-0.8)**2+(yy-0.8)**2+(zz-0.8)**2)-0.1).astype(np.float32)\n"
                "measure.marching_cubes(vol,0.0)\nt=time.perf_counter()\nv,f,_,_=measure.marching_cubes(vol,0.0)\n"
                "print(skimage.__version__, time.perf_counter()-t, len(v), len(f))")
        try:
            r = subprocess.run([py, "-c", code], capture_output=True, text=True, timeout=120)
            ver, sec, cv, cf = r.stdout.split()[-4:]
            out["cpu_baseline"] = {"kind": "reference", "what": f"skimage.measure.marching_cubes, scikit-image {ver}, 1 thread (the routine is serial)",
                                   "sample": "one 256^3 volume of the same surface", "seconds": round(float(sec), 4),
                                   "value": round(256 ** 3 / float(sec), 1), "unit": "lattice points/s", "cores": 1,
                                   "vertices": int(cv), "faces": int(cf)}
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)[:200]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
