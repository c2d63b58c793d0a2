#!/bin/bash
# GPU call K (round 6): the evidence set on the round's final libsdfhip.so - rocprofv3 kernel stats + PMC passes for config 2, config 5 at
# 8 and 16 levels and the inference legs; the box's class; kernel resources.
O=gpurun_out/r6k
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/box_class.sh $O/box_class.txt > /dev/null 2>&1
head -12 $O/box_class.txt | tail -4
bash tools/profile_round.sh r6 > $O/prof_train.log 2>&1; tail -3 $O/prof_train.log | cut -c1-300
bash tools/profile_eval.sh r6_eval > $O/prof_eval.log 2>&1; tail -3 $O/prof_eval.log | cut -c1-300
bash tools/profile_round.sh r6_cfg5l16 --config 5 --levels 16 > $O/prof_cfg5l16.log 2>&1; tail -2 $O/prof_cfg5l16.log | cut -c1-300
bash tools/profile_round.sh r6_cfg5 --config 5 > $O/prof_cfg5.log 2>&1; tail -2 $O/prof_cfg5.log | cut -c1-300
python tools/kernel_resources.py > $O/kernel_resources.csv
ls profiles | grep "^r6_" | head -60
