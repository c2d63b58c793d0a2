"""Reduce the rocprofv3 CSVs of tools/profile_round.sh to the summaries committed under profiles/.

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_pmc_summary.csv    per kernel (average per launch): FETCH_SIZE / WRITE_SIZE [KB], HBM read GB (FETCH_SIZE x 2: gfx950's
                                    counter reports half of a coalesced stream, MI355X_MICROARCH.md), HBM write GB, SQ ratios, LDS bank-conflict fraction
  profiles/<tag>_pmc_traffic.json   the dominant kernel's HBM bytes per launch (read by bench.py -> roofline.traffic)
"""
import csv, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
inference = "--inference" in sys.argv[2:]  # tools/profile_eval.sh: no training steps in the passes - per-kernel tables only
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
sys.path.insert(0, root)
from sdfstudio_amd import build as _build  # noqa: E402

# identity of the library the passes ran on (the digest written next to the .so when it was linked; the sources' own digest beside it)
LIB_ID = {"library_digest": _build.built_digest(), "source_digest_at_summary": _build.source_digest()}


def short(name):
    return name.split("(")[0].replace("void ", "")


def counters(sub):
    acc = defaultdict(lambda: defaultdict(list))
    path = os.path.join(src, sub, "p_counter_collection.csv")
    if not os.path.exists(path):
        return acc
    with open(path) as fh:
        for r in csv.DictReader(fh):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


ks = os.path.join(src, "kt", "kt_kernel_stats.csv")
if os.path.exists(ks):
    shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
fetch, write, sq, lds = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_sq"), counters("pmc_lds")
mean = lambda v: sum(v) / len(v) if v else float("nan")
rows = []
names = fetch if fetch else sq  # EVAL_SQ_ONLY passes have no traffic counters
for k in sorted(names, key=lambda n: (-mean(fetch[n]["FETCH_SIZE"]) - mean(write.get(n, {}).get("WRITE_SIZE", [0.0]))) if fetch else n):
    if "rocclr" in k or k.startswith("at::") or "elementwise" in k:
        continue
    f_kb = mean(fetch[k]["FETCH_SIZE"]) if k in fetch else float("nan")
    w_kb = mean(write[k]["WRITE_SIZE"]) if k in write else float("nan")
    s = sq.get(k, {})
    wave = mean(s.get("SQ_WAVE_CYCLES", []))
    ratio = lambda c: round(mean(s.get(c, [])) / wave, 3) if s and wave == wave and wave > 0 else ""
    rnd = lambda v: round(v) if v == v else ""
    rows.append({"kernel": k, "launches_sampled": len(fetch[k]["FETCH_SIZE"]) if k in fetch else len(sq[k].get("SQ_WAVE_CYCLES", [])),
                 "FETCH_SIZE_KB": rnd(f_kb), "WRITE_SIZE_KB": rnd(w_kb),
                 "hbm_read_GB_corrected_x2": round(2 * f_kb * 1024 / 1e9, 3), "hbm_write_GB": round(w_kb * 1024 / 1e9, 3),
                 # matrix-pipe busy cycles over (cycles x 1024 SIMDs): GRBM_GUI_ACTIVE is summed over the 8 XCDs
                 "mfma_busy_frac": round(mean(s["SQ_VALU_MFMA_BUSY_CYCLES"]) / (mean(s["GRBM_GUI_ACTIVE"]) * 128), 3) if s else "",
                 "wait_any_frac": ratio("SQ_WAIT_ANY"),
                 "wait_inst_frac": ratio("SQ_WAIT_INST_ANY"), "valu_frac": ratio("SQ_ACTIVE_INST_VALU"),
                 # LDS: extra cycles lost to bank conflicts over all cycles the LDS arrays were busy (MI355X_MICROARCH.md)
                 "lds_bank_conflict_frac": (round(mean(lds[k]["SQ_LDS_BANK_CONFLICT"]) / mean(lds[k]["SQ_LDS_IDX_ACTIVE"]), 4)
                                            if k in lds and mean(lds[k].get("SQ_LDS_IDX_ACTIVE", [])) > 0 else "")})
if rows:
    with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows[:24])
    json.dump({"tag": tag, **LIB_ID, "passes": sorted(d for d in os.listdir(src) if os.path.isdir(os.path.join(src, d)))},
              open(os.path.join(dst, f"{tag}_pmc_meta.json"), "w"), indent=1)
    # the dominant kernel runs as two launches per step (tangent pass | data backward, geo_kernels.h PHASE): its traffic is their sum
    gb = [] if inference else [r for r in rows if r["kernel"].startswith("geo_bwd_kernel")]
    if gb:
        json.dump({"kernel": "geo_bwd_kernel", "launches_per_step": len(gb), **LIB_ID,
                   "hbm_read_bytes": sum(r["hbm_read_GB_corrected_x2"] for r in gb) * 1e9, "hbm_write_bytes": sum(r["hbm_write_GB"] for r in gb) * 1e9,
                   "source": f"profiles/{tag}_pmc_summary.csv: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), FETCH_SIZE "
                             "doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced stream); sum over the kernel's launches of a step"},
                  open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
# whole-step HBM bytes: every kernel's average launch traffic x its launches in the sampled steps (the PMC passes run
# bench.py --steps 2 --warmup 1 --no-forward-only: 3 training steps and nothing else)
if rows and not inference:
    tot_r = sum(r["hbm_read_GB_corrected_x2"] * r["launches_sampled"] for r in rows if r["hbm_read_GB_corrected_x2"] == r["hbm_read_GB_corrected_x2"])
    tot_w = sum(r["hbm_write_GB"] * r["launches_sampled"] for r in rows if r["hbm_write_GB"] == r["hbm_write_GB"])
    gb = [r for r in rows if r["kernel"].startswith("geo_bwd_kernel")]
    steps = gb[0]["launches_sampled"] if gb else 1  # geo_bwd runs once per training step
    json.dump({"training_steps_sampled": steps, **LIB_ID, "hbm_read_GB_all_launches": round(tot_r, 2), "hbm_write_GB_all_launches": round(tot_w, 2),
               "hbm_GB_per_training_step": round((tot_r + tot_w) / steps, 2),
               "note": "sum over all sdfhip kernels of (average bytes per launch x launches sampled), divided by the training steps sampled"},
              open(os.path.join(dst, f"{tag}_step_traffic.json"), "w"), indent=1)
for r in rows[:12]:
    print(r)
