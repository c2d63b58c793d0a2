"""One-off (round 5): assemble the restructured DESIGN.md from the kept sections of the old file + the new head / tail fragments."""
import re, sys
old = open('DESIGN.md').read()
head = open('tools/_design_new_head.md').read()
tail = open('tools/_design_new_tail.md').read()
i1 = old.index("## 1. The path and its boundary") + len("## 1. The path and its boundary\n")
i4 = old.index("## 4. Kernels, bounds, algorithmic work")
mid = old[i1:i4]

STATE = """* **BASELINE config 2** (4096 rays x 128 samples, full train step incl. fused Adam): **22.0 ms per step = 23.8 M ray-samples/s** (`profiles/r5_bench_default_line.json`;
  driver-timed in round 4: 22.87 ms).  Dominant kernel `geo_bwd_kernel`: 6.97 ms, `roofline.frac` **0.19** (issued 16-bit MFMA terms over the dense
  peak), HBM-bound on its own saved tensors (34.3 GB at 92 % of the rate this pool streams that read / write mix).  Section 7 records the decision
  that this data flow is the end state of the architecture.
* **Compute-bound inference** (nothing saved, row f4): dense SDF evaluation **368 M points/s** over 2^26 lattice points, the MODE_SDF kernel at **0.51 of the
  dense 16-bit peak** as issued terms (0.45 over the whole leg); eval-mode render **5.26 ms per 4096 x 128 batch = 100 M ray-samples/s** (round 4: 6.46),
  `roofline.frac` 0.37 - this round found that no `torch.no_grad()` render had ever taken the nothing-saved kernels (section 4.1).
* **BASELINE config 5** (neus-facto-angelo at its own sizes): **7.2 ms per step with 8 of 16 levels on, 9.3 ms in the steady state** (driver-timed round 4:
  7.84 / 9.69); its steady-state parity bars are back at the reference path's own fp32 class x 2 (round 4: x 8) - measured 0.24 - 0.93 x, i.e.
  closer to fp64 than the fp32 oracle - by 24-bit products plus a compensated sdf-row sum below delta = 2e-3 (+0.18 ms per step).
* **The reference's one published operating point** (`neus-facto` preset, `README.md:83`: ~22 it/s on an RTX 3090): **401 it/s** (2.50 ms per step).
* **Packed-sample path** (NeuS-acc, row f2): 9.1 ms per step at 85 kept samples per ray.  `neus-facto-bigmlp` (8 x 512): 48.4 ms at config 2's batch = 2.20 x the
  256-wide step (round 4: 2.41 x).
* **Exchange sized for config 5** (row g3): reduce-scatter -> owned-slice fused Adam -> all-gather, the table's chunks leaving from inside the native
  backward; bit-identical to the all-reduce path under gloo at world 2 / 4; time model in section 6.  No N > 1 hardware run exists.
* Zero scratch in every kernel of the library (`profiles/r5_kernel_resources.csv`, guarded by a CPU test).
* A regression of this round's own scratch fix - every gradient at and below the skip layer zero in FIRST-ORDER backwards - was caught by the full GPU
  suite and fixed (section 4.1); GPU suite 164 + 3 skipped, CPU suite 107 + 2 skipped.
"""
head = head.replace("@@STATE@@", STATE)

KT = """| Kernel | Bound | HBM bytes per step (PMC) | Measured per step |
|---|---|---|---|
| `geo_bwd_kernel` (tangent pass + data backward, **dominant**) | **HBM** (its own saved tensors) | 34.3 GB (20.1 read + 14.3 written) | 6.97 ms -> 4.9 TB/s = 92 % of the 5.2 TB/s this mix streams; matrix pipe 21 - 26 % busy |
| `wgrad_bf16x8_kernel<NA,NB>` x 16 (+ `sdfrow_grad`, `wreduce` x 14) | HBM reads (4.6 TB/s of 5.5), matrix pipe half busy | 25.6 GB + 2.0 | 5.98 + 0.60 ms |
| `geo_fwd_kernel` (forward launch + analytic-normal chain launch) | forward: producer VALU / writes; chain: HBM | 16.4 GB (6.3 read + 10.1 written) | 4.59 ms (1.9 + 2.7) |
| `col_fwd_kernel` / `col_bwd_kernel` | MFMA / HBM | 3.0 / 5.3 GB | 0.95 / 1.26 ms |
| `grid_bwd_kernel` | memory-side atomic unit | 1.0 GB | 1.57 ms (beside the weight gradients) |
| `prop_bwd_kernel` x 2 | same | 0.5 GB | 0.75 ms |
| `geo_encode_kernel` | the part's mixed streaming rate on ITS traffic | 0.95 GB for 0.61 GB algorithmic | 0.20 ms -> **0.38 of 8 TB/s** on algorithmic bytes (`encode_roofline`; ceiling of this data flow 0.40: it also writes d feature / d p and zero-padded in0 for its consumers) |
| `adam_kernel` (12.5 M parameters, fused 1 / world scale) | HBM | 0.41 GB | 0.07 ms |
| everything else (samplers, render, interlevel, prep, assemble, pack, losses) | HBM / latency | 1.2 GB | <= 0.17 ms each; ~42 ATen / runtime launches, 0.23 ms |
| **step** | HBM at this data flow | **91.3 GB** (3.9 GB algorithmic) | **22.0 ms** -> 4.1 TB/s = 52 % of 8 TB/s, 79 % of the mix ceiling; issued MFMA terms 0.23 of 2.5 PF |

Inference launches (`profiles/r5_eval_*`; nothing saved):

| Kernel | Bound | HBM bytes per launch (PMC) | Measured |
|---|---|---|---|
| `geo_fwd_kernel<GRAD = SAVE = FEAT = false>` (MODE_SDF; dense SDF: 8 launches of 2^23 points) | **MFMA / producer VALU** | 6.1 GB (3.2 GB algorithmic: the tile-packed in0) | 20.6 ms per launch = 407 M points/s -> **0.51 of 2.5 PF** as issued terms; matrix pipe 56 %, VALU 39 %, waiting 17 % |
| `geo_fwd_kernel<GRAD, !SAVE, FEAT>` forward + chain launches (eval render) | forward: producer / writes; chain: HBM | 8.6 GB hand-over (`u_l` written, then read) + 1.2 | 3.73 ms per 524 288 points |
| `col_fwd_kernel<…, false>` | MFMA | 0.9 read + feature tiles | 0.84 ms |
| eval render as a whole | | | 5.26 ms -> (2G + C) x 3 terms / the three launches = **0.37 of 2.5 PF** |
"""
tail = tail.replace("@@KERNEL_TABLE@@", KT)
tail = tail.replace("@@SINGLE_LAUNCH_AB@@", "**slower** (5.34 vs 5.14 - 5.28 ms per batch, `profiles/r5_ab_single_launch_eval.jsonl`: 62 KB of two loops per CU pair against the 64 KB instruction cache, and the hand-over does not stay in cache - 1 MiB per workgroup x 256 workgroups in flight is the whole 256 MB Infinity Cache); not adopted")
tail = tail.replace("@@BOXNOTE@@", "box class and probe rates of the call: `profiles/r5_box_class.txt`")
NUM = """| leg | figure |
|---|---|
| config 2 train step (headline) | **22.02 ms** = 23.8 M ray-samples/s = 45.4 it/s; `roofline.frac` 0.189 (`geo_bwd` 6.97 ms); `encode_roofline.frac` 0.377; model FLOPs 190 TFLOP/s = 1.21 x the fp32-matrix peak, 0.228 of the dense 16-bit peak as issued terms |
| `forward_only` | 5.26 ms per batch = 99.7 M ray-samples/s; `frac` 0.365; `geo_fwd` 3.74, `col_fwd` 0.84, encode 0.20 ms |
| `dense_sdf` | 182.2 ms per 2^26 points = **368 M points/s**; `geo_fwd` 164.9 ms (0.51 kernel-only), encode 16.2 ms; leg `frac` 0.448 |
| `config5.levels8 / levels16` | 7.22 / 9.28 ms; encode `frac` 0.54 / 0.56; 0.77 / 1.84 GB exchanged per rank at N > 1; Adam rows 192 M / 460 M |
| config 5 steady state, 24-bit sdf rows on / off (`SDFHIP_NUMFIELD_HP`) | `geo_fwd` 0.608 / 0.433 ms per step (+0.18 ms); step 8.79 / 8.91 ms (inside run-to-run noise) (`profiles/r5_cfg5l16_hp{1,0}_timing.json`) |
| `preset` (neus-facto as shipped) | 2.495 ms per step = **401 it/s** (published: ~22 it/s, RTX 3090); host enqueue 1.85 ms vs GPU 2.26 ms per step: not host-bound |
| `neus_acc` | 9.13 ms per step, 85.2 samples kept per ray (128 dense-equivalent), 19.1 M packed ray-samples/s; host enqueue 8.5 ms vs GPU 9.1 ms: **host-bound** (a count is read back every step, as nerfacc's API does) |
| `bigmlp` preset batch / config-2 batch | 11.4 / 48.4 ms; ratio to the 256-wide step **2.20** |
| `cpu_baseline` (oracle, kind "port", 64 threads, 3 iterations of 512 x 128) | 10.3 k ray-samples/s (6.4 s per iteration); `cpu_baseline_reference` (the reference's own Python, build container, 8 vCPU): 10.5 k |
"""
tail = tail.replace("@@NUMBERS@@", NUM)
open('DESIGN.md', 'w').write(head + mid + tail)
print(len(head + mid + tail))
