#!/bin/bash
# GPU call Z (round 6): the evidence set on the round's FINAL libsdfhip.so (batched split-K reductions, sensor-depth kernels): smoke(), rocprofv3 kernel
# stats + PMC passes for config 2, config 5 at 8 and 16 levels and the inference legs; kernel resources; the default bench run; the whole GPU suite.
O=gpurun_out/r6z
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/box_class.sh $O/box_class.txt > /dev/null 2>&1
head -12 $O/box_class.txt | tail -4
bash tools/profile_round.sh r6 > $O/prof_train.log 2>&1; tail -3 $O/prof_train.log | cut -c1-300
bash tools/profile_eval.sh r6_eval > $O/prof_eval.log 2>&1; tail -3 $O/prof_eval.log | cut -c1-300
bash tools/profile_round.sh r6_cfg5l16 --config 5 --levels 16 > $O/prof_cfg5l16.log 2>&1; tail -2 $O/prof_cfg5l16.log | cut -c1-300
bash tools/profile_round.sh r6_cfg5 --config 5 > $O/prof_cfg5.log 2>&1; tail -2 $O/prof_cfg5.log | cut -c1-300
python tools/kernel_resources.py > $O/kernel_resources.csv
( time timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err ) 2> $O/bench_time.txt
echo "bench rc=$?"; wc -c $O/bench_line.json; tail -3 $O/bench_time.txt
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
( time timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_time.txt
echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log | cut -c1-300; tail -3 $O/pytest_time.txt
ls profiles | grep "^r6_" | wc -l
