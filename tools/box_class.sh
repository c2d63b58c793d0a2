#!/bin/bash
# Classify the gpurun box before anything else runs: the one-wave-per-SIMD fused stack probe (tools/probe_split.hip) takes ~240 k
# cycles per workgroup (3-term) on most boxes and ~590 k on the slow class.  Also records clocks / power.
OUT=${1:-gpurun_out/box_class.log}
mkdir -p $(dirname $OUT)
{
  echo "== $(date -u +%FT%TZ) $(hostname)"
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showperflevel --showmaxpower 2>/dev/null | grep -v "^=\|^$" | head -20
  tools/_bin/probe_split
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" | head -6
} > $OUT 2>&1
CYC=$(grep "3-term" $OUT | head -1 | sed 's/.* \([0-9]*\) cyc.*/\1/')
if [ -n "$CYC" ] && [ "$CYC" -gt 400000 ]; then echo "BOX CLASS: SLOW ($CYC cyc/WG)" | tee -a $OUT; else echo "BOX CLASS: fast ($CYC cyc/WG)" | tee -a $OUT; fi
