"""Reduce the rocprofv3 passes of tools/profile_mesh.sh (gpurun_out/<tag>/{kt,pmc_fetch,pmc_write,pmc_sq}) to the evidence committed under profiles/:

    profiles/<name>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the standalone check (verbatim; ALL launches: golden cases + crop)
    profiles/<name>_pmc_summary.json   per kernel, over its launches on the 512^3 crop only (the largest grid of each kernel): launches, median /
                                       min duration [us], FETCH_SIZE and WRITE_SIZE [bytes per launch], HBM read bytes = FETCH_SIZE x 2
    usage: python tools/mesh_pmc_summary.py <tag under gpurun_out> <name>

Counter units as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE doubled on gfx950 for coalesced streams).  The
correction is CALIBRATED inside this very run on two known byte counts: mc_pointbits_kernel reads the volume exactly once (4 B x 512^3 =
536 870 912 B: FETCH_SIZE x 2 must reproduce it), fill_volume writes it exactly once (WRITE_SIZE must).  Both ratios are in the summary."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def counter(sub, cname):
    acc = collections.defaultdict(list)
    path = os.path.join(src, sub, "p_counter_collection.csv")
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == cname:
            acc[short(r["Kernel_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return acc


shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, f"{name}_kernel_stats.csv"))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_trace.csv"))):
    dur[short(r["Kernel_Name"])].append((int(r["Grid_Size_X"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
fetch, write = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
sq = {c: counter("pmc_sq", c) for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")}
mean = lambda v: sum(v) / len(v) if v else None
kernels = {}
for k, v in dur.items():
    if "rocclr" in k:
        continue
    g = max(x for x, _ in v)
    big = sorted(d for x, d in v if x == g)
    pick = lambda acc: mean([c for x, c in acc.get(k, []) if x == max(y for y, _ in acc[k])]) if acc.get(k) else None
    f_kb, w_kb = pick(fetch), pick(write)
    row = {"grid_threads": g, "launches_sampled": len(big), "median_us": round(big[len(big) // 2] / 1e3, 2), "min_us": round(big[0] / 1e3, 2),
           "FETCH_SIZE_bytes": None if f_kb is None else round(f_kb * 1024), "WRITE_SIZE_bytes": None if w_kb is None else round(w_kb * 1024),
           "hbm_read_bytes_fetch_x2": None if f_kb is None else round(2 * f_kb * 1024)}
    wave = pick(sq["SQ_WAVE_CYCLES"])
    if wave:
        for c, out in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_ACTIVE_INST_VALU", "valu_frac")):
            x = pick(sq[c])
            row[out] = None if x is None else round(x / wave, 3)
    kernels[k] = row
lib_kernels = [k for k in kernels if k.startswith("mc_")]
n_pts = 512 ** 3
summary = {
    "what": "libsdfmesh.so on one 512^3 crop (tools/mesh_gpu_check.cpp), rocprofv3: --kernel-trace --stats, then separate --pmc passes "
            "(FETCH_SIZE | WRITE_SIZE | SQ): tools/profile_mesh.sh",
    "mesh_library_digest": __import__("sdfstudio_amd.build", fromlist=["x"]).mesh_source_digest(),
    "kernels": kernels,
    "per_crop": {"kernel_time_us_sum_of_medians": round(sum(kernels[k]["median_us"] for k in lib_kernels), 2),
                 "hbm_read_bytes": sum(kernels[k]["hbm_read_bytes_fetch_x2"] or 0 for k in lib_kernels),
                 "hbm_write_bytes": sum(kernels[k]["WRITE_SIZE_bytes"] or 0 for k in lib_kernels)},
    "calibration": {"volume_bytes": 4 * n_pts,
                    "mc_pointbits_kernel_fetch_x2_over_volume": None if not kernels.get("mc_pointbits_kernel", {}).get("hbm_read_bytes_fetch_x2") else
                    round(kernels["mc_pointbits_kernel"]["hbm_read_bytes_fetch_x2"] / (4 * n_pts), 4),
                    "fill_volume_write_over_volume": None if not kernels.get("fill_volume", {}).get("WRITE_SIZE_bytes") else
                    round(kernels["fill_volume"]["WRITE_SIZE_bytes"] / (4 * n_pts), 4)},
}
summary["per_crop"]["hbm_bytes"] = summary["per_crop"]["hbm_read_bytes"] + summary["per_crop"]["hbm_write_bytes"]
json.dump(summary, open(os.path.join(dst, f"{name}_pmc_summary.json"), "w"), indent=1)
for k in sorted(lib_kernels, key=lambda k: -kernels[k]["median_us"]):
    print(k, kernels[k])
print(summary["per_crop"], summary["calibration"])
