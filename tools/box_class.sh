#!/bin/bash
# Classify the gpurun box before anything else runs: the one-wave-per-SIMD fused stack probe (tools/probe_split.hip) takes ~240 k
# cycles per workgroup (3-term) on most boxes and ~590 k on the slow class.  Also records clocks / power.
# On a SLOW box the diagnostic battery runs automatically (such boxes are rare and cannot be requested): micro-probes, the
# two-waves-per-SIMD version of the same stack, the stack without weight-fragment LDS reads, SMI state.
OUT=${1:-gpurun_out/box_class.log}
mkdir -p $(dirname $OUT)
{
  echo "== $(date -u +%FT%TZ) $(hostname)"
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showperflevel --showmaxpower 2>/dev/null | grep -v "^=\|^$" | head -20
  tools/_bin/probe_split
  [ -x tools/_bin/probe_split_looped ] && tools/_bin/probe_split_looped
  [ -x tools/_bin/probe_icache ] && tools/_bin/probe_icache
  /opt/rocm/bin/rocm-smi --showuniqueid --showvbios --showfwinfo --showdriverversion 2>/dev/null | grep -i "unique\|vbios\|SMC\|MEC \|RLC \|SOS\|driver" | head -10
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" | head -6
} > $OUT 2>&1
CYC=$(grep "3-term" $OUT | head -1 | sed 's/.* \([0-9]*\) cyc.*/\1/')
if [ -n "$CYC" ] && [ "$CYC" -gt 400000 ]; then
  echo "BOX CLASS: SLOW ($CYC cyc/WG)" | tee -a $OUT
  D=$(dirname $OUT)/slowbox_diag.log
  {
    echo "== slow-box diagnostics $(date -u +%FT%TZ)"
    /opt/rocm/bin/rocm-smi --showall 2>/dev/null | grep -v "^=\|^$" | head -120
    echo "-- probe_box"; tools/_bin/probe_box
    echo "-- two waves per SIMD, same stack (probe_pair)"; tools/_bin/probe_pair_sb1b4
    echo "-- one wave per SIMD without the weight-fragment LDS reads (ABL=4)"; tools/_bin/probe_split_abl4
    echo "-- one wave per SIMD, fp16 parts"; tools/_bin/probe_split_f16
    echo "-- one wave per SIMD, ReLU instead of softplus (no transcendental instructions)"; tools/_bin/probe_split_relu
    echo "-- two waves per SIMD without the weight DMA (ABL=1) / without production (ABL=2)"; tools/_bin/probe_pair_abl1; tools/_bin/probe_pair_abl2
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -v "^=\|^$" | head -20
  } > $D 2>&1
  echo "diagnostics -> $D"
else
  echo "BOX CLASS: fast ($CYC cyc/WG)" | tee -a $OUT
fi
