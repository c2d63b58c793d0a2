"""Worker of tests/test_gpu_rccl_single_rank.py (a process of its own: it owns a one-rank "nccl" = RCCL process group on cuda:0).

    python tests/rccl_single_rank_worker.py protocol | small | config5

Prints one JSON object.  SDFHIP_FORCE_EXCHANGE=1 makes a one-rank group exchange as if it had peers (sdfstudio_amd/distributed.py): the
RCCL reduce_scatter_tensor / all_gather_into_tensor / all_reduce kernels run on RCCL's stream, the waits are stream waits, the native
table-gradient callback launches its bucket from inside the field's backward - and every collective returns its input, so the training
must be the SAME training as without any exchange (reference seam: pipelines/base_pipeline.py:241-243, scripts/train.py:127-145).
"""
import functools
import json
import operator
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODES = ("none", "none_again", "shard", "allreduce")


def _set_mode(mode):
    if mode.startswith("none"):
        os.environ.pop("SDFHIP_FORCE_EXCHANGE", None)
    else:
        os.environ["SDFHIP_FORCE_EXCHANGE"] = "1"
        os.environ["SDFHIP_BENCH_EXCHANGE"] = mode


def protocol(device):
    """Deterministic gradients (no atomics anywhere): parameters AND Adam moments must be bit-identical across the three exchanges, through
    a progressive-level switch of the table (track_active), with chunked buckets, an unused parameter and a late bucket."""
    from sdfstudio_amd.distributed import FlatGradients, plan_buckets
    from sdfstudio_amd.engine.optimizers import Optimizers

    out = {}
    for mode in ("none", "shard", "allreduce"):
        _set_mode(mode)
        g = torch.Generator().manual_seed(3)
        table = torch.nn.Parameter((torch.rand(41000, generator=g) - 0.5).to(device))
        weight = torch.nn.Parameter(torch.randn(37, 83, generator=g).to(device))
        bias = torch.nn.Parameter(torch.randn(83, generator=g).to(device))
        unused = torch.nn.Parameter(torch.randn(5, generator=g).to(device))
        prop = torch.nn.Parameter(torch.randn(700, generator=g).to(device))
        groups = {"fields": [weight, table, bias, unused], "proposal_networks": [prop]}
        shard = mode != "allreduce"  # "none": the sharded LAYOUT (padding) without any collective
        if shard:
            params, buckets, late = plan_buckets(groups, big_numel=1 << 12)
            flat = FlatGradients(params, buckets=buckets, shard=True, late_buckets=late, chunk_numel=1 << 12)
        else:
            flat = FlatGradients([p for v in groups.values() for p in v], buckets=list(groups.values()), chunk_numel=1 << 12)
        assert flat.exchanging == (mode != "none")
        table0 = table.detach().clone()
        opts = Optimizers({"fields": {"lr": 1e-2, "scheduler": None}, "proposal_networks": {"lr": 3e-2, "scheduler": None}}, groups, flat_grads=flat)
        if device.type == "cpu":  # protocol_cpu (gloo; the -m "not gpu" twin of this case): the oracle's Adam formula stands in for the kernel
            from test_cpu_distributed import _oracle_adam_step_slice

            opts.adam._step_slice = _oracle_adam_step_slice(opts.adam)
        active = [16000]
        flat.track_active(table, lambda: active[0])
        colls = []
        for step in range(5):
            opts.wait_parameters()
            if step == 2:
                active[0] = 29000  # a level is switched on
            c = {id(p): (torch.rand(p.shape, generator=g) + 0.5).to(device) for p in (table, weight, bias, prop)}
            c[id(table)][active[0]:] = 0.0
            loss = functools.reduce(operator.add, [(p * c[id(p)]).sum() for p in (table, weight, bias, prop)])
            flat.zero(loss)
            loss.backward()
            opts.optimizer_step_all(grad_scale=None)
            colls.append([flat.last_collectives, flat.last_gather_collectives])
        opts.wait_parameters()
        if device.type == "cuda":
            torch.cuda.synchronize()
        opts.gather_moments()
        sd = opts.state_dict()
        out[mode] = {"table_moved": float((table.detach() != table0)[:29000].float().mean()), "table_tail_untouched": bool(torch.equal(table.detach()[29000:], table0[29000:])),
                     "params": {n: p.detach().cpu() for n, p in (("table", table), ("weight", weight), ("bias", bias), ("unused", unused), ("prop", prop))},
                     "moments": {k: (v["exp_avg"].cpu(), v["exp_avg_sq"].cpu()) for k, v in sd["groups"].items()},
                     "offsets": {n: flat._offset[id(p)] for n, p in (("table", table), ("weight", weight), ("bias", bias), ("unused", unused), ("prop", prop))},
                     "starts": {k: v["start"] for k, v in opts.adam.groups.items()}, "collectives": colls}
        flat.close()
    rep = {"collectives": {m: out[m]["collectives"] for m in out}, "params_equal": {}, "moments_equal": {}}
    for mode in ("shard", "allreduce"):
        rep["params_equal"][mode] = all(torch.equal(out[mode]["params"][n], out["none"]["params"][n]) for n in out["none"]["params"])
        ok = True
        for n, p in out["none"]["params"].items():  # per parameter: the layouts differ (padding), the moments of every element may not
            for which in (0, 1):
                def mom(m):
                    grp = "proposal_networks" if n == "prop" else "fields"
                    a = out[m]["offsets"][n] - out[m]["starts"][grp]
                    return out[m]["moments"][grp][which][a:a + p.numel()]
                ok = ok and torch.equal(mom(mode), mom("none"))
        rep["moments_equal"][mode] = bool(ok)
    rep["table_moved"] = min(out[m]["table_moved"] for m in out)
    rep["table_tail_untouched"] = all(out[m]["table_tail_untouched"] for m in out)
    return rep


def model_steps(device, config, first, steps, small=False):
    """The real training step (bench.make_job) under the three exchanges, same seeds.  Hash-table and embedding gradients are accumulated
    with fp32 atomics (order-dependent round-off), so a run is not bit-reproducible against ITSELF there: "none" runs twice, and the
    exchange runs are held to bit-identity on every parameter the two plain runs agree on bit for bit after step 1 (the MLP weights:
    fixed summation order), and to the plain runs' own spread (x 10, plus 1e-6 of the parameter's scale) on everything after `steps`."""
    import bench

    snaps = {}
    info = {}
    for mode in MODES:
        _set_mode(mode)
        job = bench.make_job(config, device, 1, 0, small=small)
        flat, opts, model = job["flat"], job["opts"], job["model"]
        names = {id(p): n for n, p in model.named_parameters()}
        plist = [(names[id(p)], p) for p in flat.params]
        snap = {}
        levels = []
        for i in range(steps):
            job["step"](first + i)
            levels.append(int(getattr(model.field, "_active_levels", 0)))
            if i == 0:
                opts.wait_parameters()
                torch.cuda.synchronize()
                snap["after1"] = {n: p.detach().clone() for n, p in plist}
        opts.wait_parameters()
        torch.cuda.synchronize()
        snap["end"] = {n: p.detach().clone() for n, p in plist}
        snaps[mode] = snap
        info[mode] = {"levels": levels, "collectives_last_step": flat.last_collectives,
                      "gather_collectives_total": flat.last_gather_collectives, "early_buckets_last_step": flat.last_early_buckets,
                      "overlapped_buckets_last_step": flat.last_overlapped_buckets, "exchanging": bool(flat.exchanging), "shard": bool(job["shard"])}
        flat.close()
        del job, flat, opts, model
        torch.cuda.empty_cache()
    a, b = snaps["none"], snaps["none_again"]
    det = [n for n in a["after1"] if torch.equal(a["after1"][n], b["after1"][n])]
    rep = {"info": info, "n_params": len(a["after1"]), "deterministic_after_step1": len(det),
           "deterministic_names_sample": det[:6], "modes": {}}
    for mode in ("shard", "allreduce"):
        s = snaps[mode]
        bad = [n for n in det if not torch.equal(s["after1"][n], a["after1"][n])]
        # Adam with eps 1e-15 turns a round-off difference of a near-zero gradient into a full +-lr step, so single elements of the
        # atomically accumulated tables may differ by ~lr between ANY two runs: the criterion is the FRACTION of elements that moved by
        # more than 1e-5 of the parameter's scale, against the same fraction between the two plain runs
        # ... POOLED over all parameters (a 64-element bias with one such element is 1.6 % "moved" on its own): elements that moved / all
        # elements, held to 10 x the plain runs' own pooled fraction or 2e-3, whichever is larger
        moved_plain = moved_mode = total = 0
        worst = [None, 0.0, 0.0]
        for n in a["end"]:
            tol = 1e-5 * float(a["end"][n].abs().max()) + 1e-12
            mp = int(((a["end"][n] - b["end"][n]).abs() > tol).sum())
            mm = int(((s["end"][n] - a["end"][n]).abs() > tol).sum())
            moved_plain, moved_mode, total = moved_plain + mp, moved_mode + mm, total + a["end"][n].numel()
            if mm / a["end"][n].numel() >= worst[1]:
                worst = [n, mm / a["end"][n].numel(), mp / a["end"][n].numel()]
        ok = moved_mode / total <= max(10.0 * moved_plain / total, 2e-3)
        worst += [moved_mode / total, moved_plain / total]
        rep["modes"][mode] = {"bit_identical_on_deterministic_params_after_step1": not bad, "mismatching": bad[:5],
                              "end_within_plain_runs_spread": ok, "worst_fraction_moved": worst,
                              "all_finite": all(bool(torch.isfinite(v).all()) for v in s["end"].values())}
    return rep


def main():
    case = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if case == "protocol_cpu":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        device = torch.device("cpu")
        dist.init_process_group("gloo", rank=0, world_size=1)
    else:
        assert torch.cuda.is_available()
        torch.cuda.set_device(0)
        device = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        if case in ("protocol", "protocol_cpu"):
            rep = protocol(device)
        elif case == "small":
            rep = model_steps(device, 2, 0, 3, small=True)
        elif case == "config5":
            rep = model_steps(device, 5, 79998, 4)  # steps 79998 .. 80001: level 8 -> 9 at step 80000 (steps_per_level 10 000)
        else:
            raise SystemExit(f"unknown case {case}")
        rep["backend"] = dist.get_backend()
    finally:
        os.environ.pop("SDFHIP_FORCE_EXCHANGE", None)
        dist.destroy_process_group()
    print(json.dumps(rep), flush=True)


if __name__ == "__main__":
    main()
