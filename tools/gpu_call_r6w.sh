#!/bin/bash
# GPU call W (round 6): geo_encode_kernel in XCD order (every level's table gathered by ONE XCD) against the plain grid (SDFHIP_ENCODE_PLAIN_GRID=1):
# kernel durations by rocprofv3 --kernel-trace --stats and FETCH_SIZE / WRITE_SIZE by separate --pmc passes, same box, same library; then the GPU suite.
O=gpurun_out/r6w
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
bash tools/box_class.sh $O/box_class.txt > /dev/null 2>&1
B="python $R/bench.py --no-cpu-baseline --no-config5 --no-bigmlp --no-preset --no-neus-acc --no-dense-sdf --no-mesh --no-volsdf --no-config4 --no-exchange-n1 --no-forward-only --no-kernel-table"
cd /tmp && export TMPDIR=/tmp
for V in plain xcd plain2 xcd2; do
  case $V in plain*) export SDFHIP_ENCODE_PLAIN_GRID=1;; *) unset SDFHIP_ENCODE_PLAIN_GRID;; esac
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_$V -o kt -- $B --steps 8 --warmup 2 > $R/$O/kt_$V.log 2>&1
  rm -f $R/$O/kt_$V/*kernel_trace.csv
  echo "== $V: ms_per_step $(grep -h '^{' $R/$O/kt_$V.log | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')  $(grep geo_encode_kernel $R/$O/kt_$V/kt_kernel_stats.csv | cut -d, -f1-4)"
done
for V in plain xcd; do
  case $V in plain*) export SDFHIP_ENCODE_PLAIN_GRID=1;; *) unset SDFHIP_ENCODE_PLAIN_GRID;; esac
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$O/pmc_${V}_$C -o p -- $B --steps 2 --warmup 1 > $R/$O/pmc_${V}_$C.log 2>&1
    rm -f $R/$O/pmc_${V}_$C/*kernel_trace.csv
    python - "$R/$O/pmc_${V}_$C/p_counter_collection.csv" $V $C <<'PY'
import csv,sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("geo_encode_kernel")]
print(f"   {sys.argv[2]} {sys.argv[3]}: {len(v)} launches, mean {sum(v)/max(len(v),1):.0f} KB")
PY
  done
done
unset SDFHIP_ENCODE_PLAIN_GRID
cd $R
( time timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_time.txt
echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log; tail -3 $O/pytest_time.txt
