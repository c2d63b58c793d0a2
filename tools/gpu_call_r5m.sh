#!/bin/bash
# round 5, GPU call M: where the issue slots of the INFERENCE kernels go (the counters of tools/profile_issue.sh on bench.py --only inference)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --only inference --steps 1"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY \
  --output-format csv -d $O/pmc_issue -o p -- $B > $O/pmc_issue.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $O/pmc_issue2 -o p -- $B > $O/pmc_issue2.log 2>&1
rm -f $O/*/*kernel_trace.csv
python - <<'PY'
import csv, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5m"
for sub in ("pmc_issue", "pmc_issue2"):
    path = f"{root}/{sub}/p_counter_collection.csv"
    if not os.path.exists(path):
        print(sub, "missing"); print(open(f"{root}/{sub}.log").read()[-600:]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if n.startswith("geo_fwd_kernel") or n.startswith("col_fwd"):
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, c in acc.items():
        w = sum(c["SQ_WAVE_CYCLES"]) / len(c["SQ_WAVE_CYCLES"])
        print(sub, n[:62], {k.replace("SQ_", ""): (round(sum(v) / len(v) / w, 3) if "INSTS" not in k else round(sum(v) / len(v))) for k, v in c.items() if k != "SQ_WAVE_CYCLES"}, "wave_cycles", round(w))
PY
