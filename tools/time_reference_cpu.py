"""CPU baseline per BASELINE.md section 3: the REFERENCE's own Python (/root/reference/nerfstudio, unmodified, imported through
oracle/ref_harness.py with the documented PyTorch tinycudann shim for the hash grid) timed on this host's cores.

    python tools/time_reference_cpu.py [--rays 64] [--iters 3] [--out profiles/cpu_reference_r2.json]

One iteration = what NeuSFactoModel does per training step on BASELINE config 2's networks (16x2x2^19 smoothstep grid, 8x256
geometry MLP, 4x256 colour MLP, proposal sampler 256 / 96 -> 128 field samples per ray, two 5-level proposal fields):
ProposalNetworkSampler -> SDFField.forward (autograd normal) -> alpha -> weights -> renderers -> L1 + eikonal + interlevel loss
-> backward -> Adam step, on a BOUNDED batch of rays (the full 4096-ray batch would take minutes per step on a CPU).
Runs only where /root/reference exists (the build container); the result is committed and quoted by bench.py next to the
same-host timing of the oracle port.
"""
import argparse
import json
import os
import platform
import sys

sys.dont_write_bytecode = True  # /root/reference is read-only: importing it must leave no __pycache__ there
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle import ref_harness, sdf_path as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=64)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "cpu_reference_r2.json"))
    args = ap.parse_args()
    import make_golden as G

    torch.set_float32_matmul_precision("highest")
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    ns = ref_harness.import_reference()
    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.5, inside_outside=False, beta_init=0.3), num_neus_samples=128)
    p = O.init_field_params(cfg.field, seed=0)
    p.update(O.init_proposal_params(cfg.proposals))
    field, nets, sampler = G.build_reference(ns, cfg, p)
    params = [q for m in [field, *nets] for q in m.parameters() if q.requires_grad]
    opt = torch.optim.Adam(params, lr=5e-4, eps=1e-15)
    n = args.rays
    o, d, cam = O.synthetic_rays(n, seed=1)
    image = torch.rand(n, 3)

    def step():
        rand = [torch.rand(n, 1) for _ in range(3)]
        _, losses = G.run_reference(ns, field, nets, sampler, cfg, o, d, cam, image, rand, True, 1.0, 1.0)
        opt.zero_grad(set_to_none=True)
        sum(losses.values()).backward()
        opt.step()

    step()
    times = []
    for _ in range(args.iters):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    cpu = ""
    try:
        with open("/proc/cpuinfo") as fh:
            cpu = next(ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name"))
    except (OSError, StopIteration):
        cpu = platform.processor()
    res = {
        "kind": "reference",
        "what": "reference Python (nerfstudio, unmodified) + documented PyTorch tinycudann shim, fp32, full training step incl. Adam",
        "config": "BASELINE config 2 networks; proposal samples 256 / 96 -> 128 field samples per ray",
        "rays": n, "samples_per_ray": 128, "iters_timed": args.iters, "s_per_iter_median": round(med, 3),
        "value": round(n * 128 / med, 1), "unit": "ray-samples/s", "cores": cores, "torch_threads": torch.get_num_threads(),
        "host": f"build container: {cpu}, {cores} vCPU", "torch": torch.__version__,
    }
    print(json.dumps(res, indent=1))
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
