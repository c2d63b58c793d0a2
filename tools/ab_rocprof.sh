#!/bin/bash
# Same-box A/B by KERNEL DURATION (rocprofv3 --kernel-trace --stats), for kernels whose cost the in-bench event scopes distort:
#   tools/ab_rocprof.sh <out-dir> <kernel-name-regex> <libA> <libB> ...
OUT=$1; PAT=$2; shift 2
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  N=$(basename $L .so)
  SDFHIP_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$N -o kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-forward-only --no-kernel-table --steps 8 --warmup 2 > $OUT/$N.log 2>&1
  echo "== $N: $(grep -h '^{' $OUT/$N.log | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  python - "$OUT/$N/kt_kernel_stats.csv" "$PAT" <<'PY'
import csv,re,sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]): print("  ", r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg", round(float(r["TotalDurationNs"])/1e6,2), "ms total")
PY
  rm -f $OUT/$N/*kernel_trace.csv
done
