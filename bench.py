"""bench.py — NeuS-facto training throughput of the sdfhip hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full training iteration of BASELINE config 2 on every rank: draw 4096 rays, proposal sampling
(256 + 96 proposal samples, two proposal fields), 128 SDF-field samples per ray through hash grid + 8x256 geometry MLP
(+ analytic normal) + 4x256 colour MLP, NeuS alpha compositing, L1 + eikonal + interlevel losses, full backward
(including the second-order terms), one RCCL all-reduce of the flat gradient buffer when N > 1, Adam step.
Inputs are synthetic (DTU-scan65-like cameras) and already resident in HBM.

`--config 5` runs BASELINE config 5 instead (neus-facto-angelo, method_configs.py:381-450: 2048 rays x 48 samples, 16 x 8 x 2^22 linear
hash grid = 2.1 GB table, numerical SDF gradients (7 geometry evaluations per sample), progressive levels from level_init = 8, "grid"
background model, curvature loss); its roofline is the hash-grid gather of geo_encode_kernel, the HBM-bound stage SURVEY 8(d) names K1.
The default (config 2) is the headline the driver records.

Rank 0 prints ONE JSON line.  `value` = whole-job field ray-samples per second (N * 4096 * 128 / step time).
`roofline` is for the dominant kernel (geo_bwd_kernel: tangent + data backward of the geometry MLP), with its launch
time measured live by HIP events recorded on the launch stream inside the timed region (sdfhip_profile_*).  With the
matrix products on the 16-bit pipe (hi + lo parts, fp32 accumulate) that kernel is bound by the HBM traffic of the saved
per-layer tensors it reads and writes (DESIGN.md section 4 derives the 58.9 KB per ray-sample = `dataflow_bytes`;
`algorithmic_bytes` is SURVEY 8(d)'s boundary-only figure and `waste_ratio` the traffic over it).  `step_roofline` is the
whole step.  `cpu_baseline` times the CPU oracle (a port of the reference's PyTorch path) on this host for a bounded sample;
`cpu_baseline_reference` quotes the reference's own Python timed in the build container (profiles/cpu_reference_r2.json).
"""
import argparse
import functools
import json
import math
import operator
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_RAYS = 4096
N_SAMPLES = 128
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA, f32 accumulate
PEAK_HBM_GBS = 8000.0


def geo_bwd_io_bytes(nb0=3, nbf=8):
    """SURVEY section 8(d)'s notion of algorithmic bytes for geo_bwd_kernel: what crosses its boundary if nothing were saved
    between kernels - the tangent seed (nb0 blocks of 128 B), d L / d feature (nbf), d L / d sdf (4 B) in, d L / d in0 (nb0) out."""
    return 128 * (nb0 + nbf + nb0) + 4


def geo_bwd_algorithmic_bytes(nbh=8, nb0=3, nb3=8, nl=8, skip=4, nbf=8):
    """HBM bytes per ray-sample that geo_bwd_kernel's DATA FLOW moves: every tensor is tile-packed in
    blocks of 32 features x 4 B = 128 B per point.  Tangent pass: reads the seed, z_l and r_l; writes the tangent qb_l
    entering every layer and zc_l.  Data backward: reads featbar, z_l and zc_l; writes zbar_l and d L / d in0.  These saved
    per-layer tensors exist because the three sweeps over the layers alternate direction and the weight gradients are separate
    GEMMs; geo_bwd_io_bytes is the (27x smaller) figure of an ideal implementation that keeps them on chip."""
    kb = lambda l: nb0 if l == 0 else (nb3 + nb0 if l == skip else nbh)
    nbo = lambda l: nbf if l == nl else (nb3 if l + 1 == skip else nbh)
    has_skip = 0 < skip < nl
    # tangent pass: layer 0 reads the seed (and rewrites it: qb_0 == seed); layer l >= 1 reads (z, r) of the layer below and
    # writes qb_l and zc_{l-1}; the skip layer's in0 gemm reads the seed again; the epilogue reads (z, r) and writes qb_NL, zc
    rd = nb0 + sum(2 * nbo(l - 1) for l in range(1, nl)) + (nb0 if has_skip else 0) + 2 * nbo(nl - 1)
    wr = nb0 + sum(kb(l) + nbo(l - 1) for l in range(1, nl)) + 2 * nbh
    # data backward: featbar; per layer (z_l, zc_l) -> zbar_l; the skip layer makes two passes over its zbar (in0 columns, then
    # hidden columns) and parks its part of d L / d in0, which layer 0 re-reads and completes
    rd += nbf + sum(2 * nbo(l) for l in range(nl)) + ((nb0 + 2 * nbo(skip)) if has_skip else 0)
    wr += sum(nbo(l) for l in range(nl)) + nb0 * (2 if has_skip else 1) + (nbo(skip) if has_skip else 0)
    return 128 * (rd + wr)


def synthetic_cameras(device):
    """49 pinhole cameras, 384x384, fx=925.5 fy=922.6 cx=199.4 cy=198.1, centres on a sphere of radius 2.73 looking at
    the origin (docs/sdfstudio-data.md:26-88 scan65 meta)."""
    k = torch.arange(49, dtype=torch.float64)
    phi = k * 2.399963229728653
    z = 0.15 + 0.7 * (k + 0.5) / 49
    r = torch.sqrt(1 - z * z)
    centers = torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], dim=-1) * 2.73
    fwd = -centers / centers.norm(dim=-1, keepdim=True)
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64).expand_as(fwd)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    down = torch.cross(fwd, right, dim=-1)
    rot = torch.stack([right, down, fwd], dim=-1)  # camera-to-world columns (x right, y down, z forward)
    return centers.float().to(device), rot.float().to(device)


def draw_rays(centers, rot, n, gen):
    """Uniform random (camera, y, x) as pixel_samplers.py:47-50, then pinhole ray directions (49 cameras of 384 x 384 pixels, DTU-like
    intrinsics): one torch.rand + one native launch (cameras/rays.py::generate_pinhole_rays)."""
    from sdfstudio_amd.cameras.rays import generate_pinhole_rays

    u = torch.rand(n, 3, device=centers.device, generator=gen)
    return generate_pinhole_rays(u, centers, rot, 384, 384, 925.5, 922.6, 199.4, 198.1)


def draw_rays_torch(centers, rot, n, gen):
    """The same batch in plain torch (26 launches): the statement tests/test_gpu_parity.py holds the native generator to."""
    dev = centers.device
    u = torch.rand(n, 3, device=dev, generator=gen)
    cam = (u[:, 0] * 49).long().clamp_(max=48)
    y = (u[:, 1] * 384).floor() + 0.5
    x = (u[:, 2] * 384).floor() + 0.5
    d_cam = torch.stack([(x - 199.4) / 925.5, (y - 198.1) / 922.6, torch.ones_like(x)], dim=-1)
    d = (rot[cam] * d_cam[:, None, :]).sum(dim=-1)  # rot @ d_cam per ray
    norm = d.norm(dim=-1, keepdim=True)
    return centers[cam].contiguous(), (d / norm).contiguous(), norm, cam


def build_model_config5(device):
    """BASELINE config 5: the neus-facto-angelo preset (method_configs.py:381-450) at its own sizes."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import NeuSFactoModel, NeuSFactoModelConfig, SceneBox

    torch.manual_seed(0)
    fcfg = SDFFieldConfig(use_grid_feature=True, num_layers=1, num_layers_color=4, hidden_dim=256, hidden_dim_color=256, geometric_init=True,
                          bias=0.5, beta_init=0.3, inside_outside=False, use_appearance_embedding=True, use_numerical_gradients=True,
                          base_res=64, max_res=4096, log2_hashmap_size=22, hash_features_per_level=8, hash_smoothstep=False,
                          use_position_encoding=False)
    mcfg = NeuSFactoModelConfig(near_plane=0.01, far_plane=1000.0, overwrite_near_far_plane=True, sdf_field=fcfg, background_model="grid",
                                level_init=8, eikonal_loss_mult=0.01, use_anneal_beta=True, enable_progressive_hash_encoding=True,
                                enable_numerical_gradients_schedule=True, enable_curvature_loss_schedule=True, curvature_loss_multi=5e-4)
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
    return NeuSFactoModel(mcfg, box, num_train_data=49).to(device).train()


def build_model(device, small=False, hidden=256, samples=None):
    """BASELINE config 2 (default), or the small parity configuration of tests/golden (8x64 networks, 8x2x2^11 grid, 32/24
    proposal + 16 field samples): the latter only drives the N > 1 control-flow test, never a reported number.  hidden = 512: the
    geometry network of the neus-facto-bigmlp preset (method_configs.py:503-523) in config 2's model."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import NeuSFactoModel, NeuSFactoModelConfig, SceneBox

    torch.manual_seed(0)
    if small:
        fcfg = SDFFieldConfig(num_layers=8, hidden_dim=64, geo_feat_dim=64, num_layers_color=4, hidden_dim_color=64, bias=0.5,
                              inside_outside=False, use_grid_feature=True, beta_init=0.3, num_levels=8, max_res=128, base_res=4,
                              log2_hashmap_size=11, hash_features_per_level=2, hash_smoothstep=True)
        mcfg = NeuSFactoModelConfig(sdf_field=fcfg, num_proposal_samples_per_ray=(32, 24), num_neus_samples_per_ray=16,
                                    background_model="none")
        box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
        return NeuSFactoModel(mcfg, box, num_train_data=49).to(device).train()
    fcfg = SDFFieldConfig(num_layers=8, hidden_dim=hidden, geo_feat_dim=256, num_layers_color=4, hidden_dim_color=256, bias=0.5,
                          inside_outside=False, use_grid_feature=True, beta_init=0.3, num_levels=16, max_res=2048, base_res=16,
                          log2_hashmap_size=19, hash_features_per_level=2, hash_smoothstep=True)
    mcfg = NeuSFactoModelConfig(sdf_field=fcfg, num_proposal_samples_per_ray=(256, 96), num_neus_samples_per_ray=samples or N_SAMPLES,
                                background_model="none")
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
    return NeuSFactoModel(mcfg, box, num_train_data=49).to(device).train()


def flops_per_sample():
    """SURVEY.md section 8 FLOP bookkeeping (2 x MACs): geometry MLP G, colour MLP C; training step = 6G + 3C."""
    g = 2 * (71 * 256 + 2 * 256 * 256 + 256 * 185 + 4 * 256 * 256 + 256 * 257)
    c = 2 * (321 * 256 + 3 * 256 * 256 + 256 * 3)
    return g, c


def cpu_baseline():
    """The CPU oracle (oracle/sdf_path.py: PyTorch port of the reference path, pinned against the reference's own Python)
    timed on this host: same network / sampler configuration, a bounded batch of rays, full training step with Adam."""
    from oracle import sdf_path as O

    torch.set_float32_matmul_precision("highest")
    cores = min(os.cpu_count() or 1, 64)  # beyond ~64 threads the small per-ray ops of the sampler stop scaling
    torch.set_num_threads(cores)
    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.5, inside_outside=False, beta_init=0.3), num_neus_samples=N_SAMPLES)
    p = O.init_field_params(cfg.field, seed=0)
    p.update(O.init_proposal_params(cfg.proposals))
    for k, v in p.items():
        if v.is_floating_point() and k != "laplace_density.beta_min":
            v.requires_grad_(True)
    opt = torch.optim.Adam([v for v in p.values() if v.requires_grad], lr=5e-4, eps=1e-15)
    n = 512  # large enough for the 8x256 GEMMs to spread over 64 threads (128 rays left most of them idle)
    o, d, cam = O.synthetic_rays(n, seed=1)
    image = torch.rand(n, 3)

    def step():
        rand = [torch.rand(n, 1) for _ in range(3)]
        out = O.neus_facto_forward(o, d, cam, p, cfg, anneal=1.0, cos_anneal_ratio=1.0, rand=rand, training=True)
        loss = sum(O.neus_facto_loss(out, image, cfg).values())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    step()
    t0 = time.perf_counter()
    iters = 0
    while iters < 2 or (time.perf_counter() - t0 < 15.0 and iters < 50):
        step()
        iters += 1
    dt = (time.perf_counter() - t0) / iters
    return {"value": n * N_SAMPLES / dt, "unit": "ray-samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "note": "the oracle (a PyTorch port of the reference's path, pinned on the reference's own Python by the CPU tests) timed on THIS "
                    "host; the reference tree itself does not exist on the GPU box - its own Python was timed in the build container on 8 "
                    "vCPU: cpu_baseline_reference",
            "sample": f"{iters} training iterations of {n} rays x {N_SAMPLES} samples (same networks, samplers, losses, Adam); "
                      f"{dt * 1e3:.0f} ms/iter"}


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawned(local_rank, args, port):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(args.gpus),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    run(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=[2, 5], help="BASELINE config: 2 (default, the headline) or 5 (neus-facto-angelo)")
    ap.add_argument("--levels", type=int, default=8, choices=[8, 16],
                    help="--config 5 only: hash levels the progressive mask has switched on in the timed steps - 8 = the preset's level_init "
                         "(steps 0 .. 80 k of its 1 M iterations), 16 = the steady state (steps >= 150 k: 85 %% of the schedule); the timed steps "
                         "start at step 5 resp. 200 000 of the preset's schedules")
    ap.add_argument("--no-config5", action="store_true", help="default (config 2) run: skip the two short config-5 legs appended as \"config5\"")
    ap.add_argument("--no-bigmlp", action="store_true", help="default (config 2) run: skip the two short 512-wide legs appended as \"bigmlp\"")
    ap.add_argument("--no-preset", action="store_true", help="default (config 2) run: skip the neus-facto preset leg appended as \"preset\"")
    ap.add_argument("--no-neus-acc", action="store_true", help="default (config 2) run: skip the packed-sample (NeuS-acc) leg appended as \"neus_acc\"")
    ap.add_argument("--no-dense-sdf", action="store_true", help="default (config 2) run: skip the dense-SDF (mesh extraction) leg appended as \"dense_sdf\"")
    ap.add_argument("--no-volsdf", action="store_true", help="default (config 2) run: skip the VolSDF legs (BASELINE config 1's model on the GPU) appended as \"volsdf\"")
    ap.add_argument("--no-config4", action="store_true", help="default (config 2) run: skip the BASELINE config 4 leg appended as \"config4\"")
    ap.add_argument("--full-line", action="store_true", help="print the full (~25 KB) object instead of the compact line (the full object is always "
                                                              "written to gpurun_out/bench_detail.json)")
    ap.add_argument("--no-mesh", action="store_true", help="default (config 2) run: skip the marching-cubes leg appended as \"mesh\" (a child process)")
    ap.add_argument("--only", default=None, choices=["inference", "exchange", "volsdf", "config4", "neus_acc"],
                    help="inference: only the forward-only and dense-SDF legs on config 2's model (no training steps; tools/ A/B and PMC runs); "
                         "exchange: only the forced single-rank RCCL exchange legs (what the default run appends as \"exchange_at_n1\")")
    ap.add_argument("--no-exchange-n1", action="store_true", help="default (config 2) run: skip the single-rank RCCL exchange legs (a child process)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--small", action="store_true", help="small parity configuration (control-flow tests only, not a benchmark)")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip the second (untimed) pass that times every launch")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the eval-mode leg (PMC passes: per-step launch counts stay clean)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: one process per GPU, spawned here as the reference does (scripts/train.py:190-203);
        # under torch.distributed.run the ranks already exist and RANK / WORLD_SIZE come from the environment
        import torch.multiprocessing as mp

        mp.spawn(_spawned, args=(args, _free_port()), nprocs=args.gpus, join=True)
        return
    run(args)


def fence(world):
    import torch.distributed as dist

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def make_job(config, device, world, rank, small=False, hidden=256, rays=None, samples=None, model_factory=None, ray_fn=None, batch_fn=None,
             opt_config=None):
    """Model, flat gradient buffer, optimizers and the step function of one benchmark configuration (2 or 5) on this rank."""
    from sdfstudio_amd.cameras.rays import RayBundle
    from sdfstudio_amd.distributed import FlatGradients, broadcast_parameters
    from sdfstudio_amd.engine.optimizers import Optimizers, multi_step_scheduler, multi_step_warmup_scheduler, neus_scheduler

    cfg5 = config == 5
    n_rays = rays or (2048 if cfg5 else (512 if small else 4096))  # method_configs.py:396 train_num_rays_per_batch (config 5)
    model = model_factory(device) if model_factory is not None else (
        build_model_config5(device) if cfg5 else build_model(device, small=small, hidden=hidden, samples=samples))
    broadcast_parameters(model)
    groups = {k: v for k, v in model.get_param_groups().items() if v}  # "field_background" is empty with background_model="none"
    # Gradient exchange (distributed.py).  N > 1 default: SHARDED - reduce-scatter of the flat gradient, fused Adam on this rank's 1 / N
    # slice of parameters and moments, all-gather of the updated slices (the SDF table's last: it overlaps the next step's ray
    # generation and proposal sampling); the big tables are buckets of their own.  SDFHIP_BENCH_EXCHANGE=allreduce: one bucket per
    # parameter group, all-reduced (RCCL) as soon as backward has produced it, replicated Adam (rounds 1 - 4).
    from sdfstudio_amd.distributed import force_single_rank_exchange

    # SDFHIP_FORCE_EXCHANGE=1 (with a one-rank process group): the N > 1 exchange runs at N = 1 - real RCCL kernels, real stream waits
    exchanging = world > 1 or (force_single_rank_exchange() and torch.distributed.is_initialized())
    shard = exchanging and os.environ.get("SDFHIP_BENCH_EXCHANGE", "shard") == "shard"
    if shard:
        from sdfstudio_amd.distributed import plan_buckets

        params, buckets, late = plan_buckets(groups, big_numel=(1 << 14) if small else (1 << 22))
        flat = FlatGradients(params, buckets=buckets, shard=True, late_buckets=late)
        if os.environ.get("SDFHIP_BENCH_EARLY_TABLE", "1") == "1":
            # the SDF field's table leaves from INSIDE the field's backward, behind the scatter and beside the weight-gradient GEMMs
            flat.launch_from_native(model.field.encoding.params, model.field)
    else:
        flat = FlatGradients([p for g in groups.values() for p in g], buckets=list(groups.values()))
    flat.time_waits = exchanging
    if cfg5:
        # method_configs.py:434-447: Adam 1e-3 with MultiStepWarmup (fields; AdamW with weight_decay 0 = Adam for field_background),
        # Adam 1e-2 with MultiStepLR (proposal networks)
        opts = Optimizers({"fields": {"lr": 1e-3, "scheduler": multi_step_warmup_scheduler(5000, (600000, 800000), 0.1)},
                           "field_background": {"lr": 1e-3, "scheduler": multi_step_warmup_scheduler(5000, (300000, 400000), 0.1)},
                           "proposal_networks": {"lr": 1e-2, "scheduler": multi_step_scheduler(1000000)}}, groups, flat_grads=flat)
        # progressive levels: the masked levels' table rows have exactly zero gradient on every rank; zero() asks the model
        flat.track_active(model.field.encoding.params, model.active_table_floats)
    elif opt_config is not None:
        opts = Optimizers({k: v for k, v in opt_config.items() if k in groups}, groups, flat_grads=flat)
    else:
        # optimizers and schedulers as method_configs.py:485-500 (neus-facto): Adam eps 1e-15, lr 5e-4 with NeuS warm-up / cosine
        # (fields), 1e-2 with MultiStepLR (proposal networks): one fused Adam launch per group over the flat buffers
        opts = Optimizers({"fields": {"lr": 5e-4, "scheduler": neus_scheduler(500, 0.05, 20000)},
                           "proposal_networks": {"lr": 1e-2, "scheduler": multi_step_scheduler(20000)}}, groups, flat_grads=flat)
    centers, rot = synthetic_cameras(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(42 + rank)  # base_config.py:74 + scripts/train.py:86: seed + global rank

    if shard:
        model.before_field = lambda: opts.wait_parameters(late=True)

    def step(i):
        opts.wait_parameters(late=False)  # sharded exchange: everything but the SDF table has to be back before the callbacks touch parameters
        model.before_train_iteration(i)
        o, d, norm, cam = draw_rays(centers, rot, n_rays, gen) if ray_fn is None else ray_fn(n_rays, gen)
        batch = {"image": torch.rand(n_rays, 3, device=device, generator=gen)}
        if batch_fn is not None:
            batch.update(batch_fn(n_rays, gen))
        rb = RayBundle(origins=o, directions=d, directions_norm=norm, camera_indices=cam[:, None])
        out = model(rb)
        loss = functools.reduce(operator.add, model.get_loss_dict(out, batch).values())  # (sum() would start with 0 + a tensor: a launch)
        flat.zero(loss)  # the loss's graph tells the buckets which gradients to wait for (unused parameters: distributed.py)
        loss.backward()
        # closes the exchange chunk by chunk (SUM collectives; the 1 / world mean rides in the Adam read) and, sharded, sends the updated
        # slices back asynchronously
        opts.optimizer_step_all(grad_scale=None)
        opts.scheduler_step_all(i)
        model.after_train_iteration(i)
        return loss

    return {"model": model, "flat": flat, "groups": groups, "opts": opts, "step": step, "centers": centers, "rot": rot, "gen": gen,
            "n_rays": n_rays, "shard": shard}


def timed_steps(job, first, warmup, steps, dominant, world):
    """`warmup` untimed steps, then EXACTLY `steps` timed ones between fences (barrier + synchronize on both sides); the schedules'
    step counter starts at `first`.  Returns (seconds on this rank, events of the dominant kernel inside the timed region, last loss)."""
    from sdfstudio_amd import _lib

    step = job["step"]
    flat, opts = job["flat"], job["opts"]
    for i in range(warmup):
        step(first + i)
    opts.wait_parameters()
    fence(world)
    if flat.time_waits:
        flat.exposed_ms(), flat.exposed_gather_ms()  # reset: what follows belongs to the timed steps
    _lib.profile_enable_only([dominant] if isinstance(dominant, str) else list(dominant))
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(first + warmup + i)
    opts.wait_parameters()  # sharded exchange: the last step is complete when its parameters are back on every rank
    fence(world)
    dt = time.perf_counter() - t0
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    if flat.time_waits:  # GPU time the compute stream stalled on the exchange, per phase, per step (this rank)
        job["exposed"] = {"reduce_ms_per_step": sum(flat.exposed_ms()) / steps, "gather_ms_per_step": sum(flat.exposed_gather_ms()) / steps}
    return dt, prof, loss


def max_over_ranks(dt, device, world):
    import torch.distributed as dist

    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def pmc_rows(select):
    """Rows of the newest committed rocprofv3 PMC summary (profiles/<tag>_pmc_summary.csv, tools/prof_summary.py) whose file name passes
    `select`, with the identity of the library the passes ran on: (rows, file name, library digest) or (None, None, None).  rocprofv3 cannot
    run inside the bench, so `traffic` fields quote these files - and say `traffic_stale` when the loaded library is not the profiled one."""
    import csv

    from sdfstudio_amd import build as _build

    pm_dir = os.path.join(ROOT, "profiles")
    want = _build.built_digest() or None
    cands = []
    for f in sorted((f for f in os.listdir(pm_dir) if f.endswith("_pmc_summary.csv") and select(f)), reverse=True):
        digest = None
        meta = os.path.join(pm_dir, f.replace("_pmc_summary.csv", "_pmc_meta.json"))
        if os.path.exists(meta):
            with open(meta) as fh:
                digest = json.load(fh).get("library_digest")
        cands.append((digest != want, f, digest))  # the set taken on THIS library first; else the newest name (r1 < r2 < ...; stale)
    for _stale, f, digest in sorted(cands, key=lambda c: c[0]):  # stable: keeps the reverse-name order inside each class
        with open(os.path.join(pm_dir, f)) as fh:
            rows = list(csv.DictReader(fh))
        return rows, f, digest
    return None, None, None


def newest_matching_json(pm_dir, names_sorted, digest):
    """Of the evidence JSONs `names_sorted` (ascending by name = by round): the newest one taken on the library `digest`, else the newest."""
    loaded = []
    for f in names_sorted:
        with open(os.path.join(pm_dir, f)) as fh:
            loaded.append(json.load(fh))
    same = [j for j in loaded if j.get("library_digest") == digest]
    return (same or loaded)[-1]


def geo_fwd_flags(kernel_name):
    """Template arguments after the dimensions of a geo_fwd_kernel<GeoDims<..>, GRAD, SAVE, FEAT, PHASE[, NS]> instantiation name."""
    k = kernel_name.replace(" ", "")
    if not k.startswith("geo_fwd_kernel<GeoDims<") or ">," not in k:
        return []
    return k.split(">,", 1)[1].rstrip(">").split(",")


def pmc_traffic_per_launch(select, kernel_pred):
    """HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction) summed over the kernels `kernel_pred` picks."""
    from sdfstudio_amd import build as _build

    rows, f, digest = pmc_rows(select)
    if not rows:
        return None
    picked = [r for r in rows if kernel_pred(r.get("kernel", "")) and r.get("hbm_write_GB") not in (None, "", "nan")]
    if not picked:
        return None
    tot = sum((float(r["hbm_read_GB_corrected_x2"]) + float(r["hbm_write_GB"])) * 1e9 for r in picked)
    return {"traffic": tot, "traffic_kernels": [r["kernel"] for r in picked],
            "traffic_source": f"profiles/{f} (FETCH_SIZE x 2 + WRITE_SIZE per launch, separate rocprofv3 --pmc passes)",
            "traffic_library_digest": digest, "traffic_stale": digest != (_build.built_digest() or None)}


def encode_step_traffic(select):
    """HBM bytes of all geo_encode* launches of ONE training step from the newest committed PMC summary `select` accepts.  Evidence tags:
    <round>_cfg5 (8 levels), <round>_cfg5l16 (steady state); everything else is config 2.  The slot covers every encode launch of a step
    (config 5: the 8-feature kernel of the SDF grid and the 2-feature one of the background grid); the PMC passes sample whole training
    steps, geo_bwd_kernel runs once per step and phase."""
    from sdfstudio_amd import build as _build

    rows, f, digest = pmc_rows(select)
    if not rows:
        return None
    sampled = min([int(r["launches_sampled"]) for r in rows if r.get("kernel", "").startswith("geo_bwd_kernel")] or [0])
    tot = sum((float(r["hbm_read_GB_corrected_x2"]) + float(r["hbm_write_GB"])) * 1e9 * int(r["launches_sampled"])
              for r in rows if r.get("kernel", "").startswith("geo_encode") and r.get("hbm_write_GB"))
    if sampled <= 0 or tot <= 0:
        return None
    return {"traffic": tot / sampled,  # HBM bytes per training step (PMC: average launch x launches, over the steps sampled)
            "traffic_source": f"profiles/{f} (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes; all geo_encode* launches)",
            "traffic_library_digest": digest, "traffic_stale": digest != (_build.built_digest() or None)}


def encode_roofline_config5(model, prof, steps, P):
    """K1 of SURVEY 8(d) for config 5: 7 evaluations per ray-sample (centre + 6 taps); ACTIVE levels x 8 corners x 8 features x 4 B +
    position in + 6 in0 blocks out (progressive levels: the kernel skips the levels the mask has switched off: not counted either)."""
    enc_ms, enc_n = prof.get("geo_encode_kernel", (0.0, 0))
    if enc_n <= 0:
        return None
    lv_on = int(getattr(model.field, "_active_levels", model.field.num_levels))
    per_sample = 7 * (lv_on * 8 * 8 * 4 + 12 + 6 * 128)
    eb = per_sample * P
    es = enc_ms / steps * 1e-3
    return {"kernel": "geo_encode_kernel", "bound": "hbm", "achieved": round(eb / es / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(eb / es / 1e9 / PEAK_HBM_GBS, 4), "algorithmic_bytes": eb,
            "achieved_is": f"7 x ({lv_on} active levels x 256 B gather + 12 B position + 768 B tile-packed in0) per ray-sample / its time per step",
            "ms_per_step": round(enc_ms / steps, 4), "levels_active": lv_on, "traffic": None}


def config5_legs(device, world, rank, steps=10, warmup=3):
    """BASELINE config 5 (neus-facto-angelo at its own sizes) as two short legs inside the default run, so that the driver's record
    carries a config-5 figure: the preset's level_init state (8 of 16 hash levels on: steps 0 .. 80 k of its 1 M iterations) and the
    steady state (all 16 on, schedules at step 200 000: 85 % of the run) - twice the gather / scatter bytes, 2.5 x the Adam rows and
    twice the all-reduce of the first.  Same model object, same timing discipline as the headline leg."""
    job = make_job(5, device, world, rank)
    P = job["n_rays"] * 48
    out = {"workload": "BASELINE config 5: neus-facto-angelo preset (16x8x2^22 linear grid = 2.1 GB table, 1x256 geo MLP evaluated 7 x per sample, "
                       "4x256 colour MLP, 'grid' background, curvature loss), 2048 rays x 48 samples per GPU per step, full train step incl. Adam",
           "steps": steps, "warmup": warmup}
    for name, first in (("levels8", 0), ("levels16", 200000)):
        dt, prof, loss = timed_steps(job, first, warmup, steps, "geo_encode_kernel", world)
        dt = max_over_ranks(dt, device, world)
        assert math.isfinite(float(loss.detach())), f"config 5 ({name}) diverged"
        ms = dt / steps * 1e3
        flat = job["flat"]
        out[name] = {"ms_per_step": round(ms, 3), "value": round(world * P / (dt / steps), 1), "unit": "ray-samples/s",
                     "levels_active": int(job["model"].field._active_levels), "first_step": first,
                     "exchanged_bytes_per_rank": 4 * flat.exchanged_numel(),
                     "gathered_bytes_per_rank": flat.gathered_bytes(),  # sharded exchange only (N > 1)
                     "exchange_exposed_ms_rank0": job.get("exposed"),
                     "adam_rows_live": sum(b - a for a, b in flat.live_ranges()),
                     "adam_rows_visited": job["opts"].adam.last_elements_visited,  # this rank: 1 / N of the live rows when sharded
                     "roofline": encode_roofline_config5(job["model"], prof, steps, P)}
        if out[name]["roofline"] is not None:
            l16 = name == "levels16"
            out[name]["roofline"].update(encode_step_traffic(lambda f, l16=l16: "cfg5" in f and ("cfg5l16" in f) == l16) or {})
    del job
    torch.cuda.empty_cache()
    return out


def bigmlp_legs(device, world, rank, ms_config2, steps=8, warmup=3):
    """The geometry network of the reference's neus-facto-bigmlp preset (8 x 512, method_configs.py:503-523; its own batch is 2048 rays x
    48 samples) as two short legs of the default run: at the preset's batch and at config 2's (4096 x 128), the latter next to the
    256-wide step just timed.  NOT a BASELINE config.  The 512-wide network runs LAYER BY LAYER (csrc/wide_kernels.h: two accumulator
    sets of 16 blocks do not fit a wave); its step-level figure is the 16-bit MFMA terms the geometry + colour networks issue per step
    (forward G, chain G, tangent G, data backward G, two weight-gradient sets 2G; 3 terms per product) over the step time."""
    g = 2 * (71 * 512 + 2 * 512 * 512 + 512 * 441 + 4 * 512 * 512 + 512 * 257)
    _, c = flops_per_sample()
    out = {"workload": "neus-facto-bigmlp geometry network (8 x 512 softplus, skip at 4) in config 2's model (16x2x2^19 grid, 4x256 colour MLP, "
                       "256/96 proposal samples), full train step incl. Adam; layer-at-a-time kernels (csrc/wide_kernels.h)",
           "steps": steps, "warmup": warmup, "geo_flops_per_sample": g}
    for name, rays, samples in (("preset_batch", 2048, 48), ("config2_batch", N_RAYS, N_SAMPLES)):
        job = make_job(2, device, world, rank, hidden=512, rays=rays, samples=samples)
        dt, prof, loss = timed_steps(job, 0, warmup, steps, ("geo_fwd_kernel", "geo_bwd_kernel", "wgrad_kernel"), world)
        dt = max_over_ranks(dt, device, world)
        assert math.isfinite(float(loss.detach())), f"bigmlp ({name}) diverged"
        ms, P = dt / steps * 1e3, rays * samples
        issued = 3 * (6 * g + 3 * c) * P / (dt / steps) / 1e12
        out[name] = {"ms_per_step": round(ms, 3), "value": round(world * P / (dt / steps), 1), "unit": "ray-samples/s", "rays": rays,
                     "samples_per_ray": samples,
                     # HIP events on these launches inside the timed steps (the layer-at-a-time launches record under the fused kernels' slots)
                     "kernels_ms_per_step": {k: round(v[0] / steps, 3) for k, v in prof.items()},
                     "step_roofline": {"bound": "mfma", "achieved": round(issued, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(issued / PEAK_BF16_MFMA_TFLOPS, 4),
                                       "achieved_is": "3 issued 16-bit MFMA terms x (6 G512 + 3 C) flops per ray-sample / whole step time"}}
        if name == "config2_batch" and ms_config2:
            out[name]["ratio_to_256_wide_step"] = round(ms / ms_config2, 3)
        del job, loss
        torch.cuda.empty_cache()
    return out


def enqueue_vs_gpu(step_fn, first, steps, device):
    """Is the step HOST bound?  Two measurements.  (a) From an IDLE GPU (synchronised first), the host time to enqueue ONE step, median of 3:
    with no device -> host read inside the step this is pure launch overhead and must sit well below the step's GPU time; a read inside the
    step shows up as the GPU time in front of it.  `host_bound` = this time exceeds 0.9 x the GPU time of a step.  (b) The steady state:
    host time to enqueue `steps` steps back to back against their GPU time.  (b) alone cannot tell: once the GPU is the bottleneck the HIP
    runtime throttles a host that has run ahead (bounded queue of in-flight dispatches), so host ~ GPU there for every long step - config 2's
    22 ms step reads 20 ms "host" in (b) and 3 ms in (a)."""
    idle = []
    for i in range(3):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        step_fn(first + i)
        idle.append(time.perf_counter() - t0)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(first + 3 + i)
    host = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1) / steps
    one = sorted(idle)[1] * 1e3
    return {"host_enqueue_ms_one_step_from_idle": round(one, 3), "gpu_ms_per_step": round(gpu_ms, 3),
            "host_enqueue_ms_per_step": round(host / steps * 1e3, 3), "wall_ms_per_step": round(wall / steps * 1e3, 3),
            "host_bound": one > 0.9 * gpu_ms, "host_over_gpu": round(one / gpu_ms, 3)}


def preset_leg(device, world, rank, steps=30, warmup=5):
    """The reference's ONE published operating point (README.md:83: ~22 it/s on an RTX 3090): the `neus-facto` preset
    exactly as shipped (method_configs.py:452-500) - 2 x 256 geometry + 2 x 256 colour layers on the default 16 x 2 x 2^19 smoothstep
    grid, no appearance embedding, no background model, 2048 rays x 48 field samples (+ 256 / 96 proposal samples), Adam 5e-4 with the
    NeuS warm-up / cosine schedule (fields), 1e-2 with MultiStepLR (proposal networks).  ~98 k points per step: the launch / latency
    regime, which config 2 says nothing about."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import NeuSFactoModel, NeuSFactoModelConfig, SceneBox

    def build(dev):
        torch.manual_seed(0)
        fcfg = SDFFieldConfig(use_grid_feature=True, num_layers=2, num_layers_color=2, hidden_dim=256, bias=0.5, beta_init=0.3,
                              use_appearance_embedding=False)
        mcfg = NeuSFactoModelConfig(sdf_field=fcfg, background_model="none")
        box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
        return NeuSFactoModel(mcfg, box, num_train_data=49).to(dev).train()

    job = make_job(2, device, world, rank, rays=2048, model_factory=build)
    rays, samples = 2048, int(job["model"].config.num_neus_samples_per_ray)
    assert samples == 48
    dt, prof, loss = timed_steps(job, 0, warmup, steps, ("geo_fwd_kernel", "geo_bwd_kernel", "wgrad_kernel"), world)
    dt = max_over_ranks(dt, device, world)
    assert math.isfinite(float(loss.detach())), "preset leg diverged"
    ms = dt / steps * 1e3
    _lib_mod = __import__("sdfstudio_amd._lib", fromlist=["_lib"])
    # per-kernel table (untimed, instrumented pass) and the host-enqueue / GPU split (un-instrumented)
    _lib_mod.profile_enable(True)
    for i in range(10):
        job["step"](warmup + steps + i)
    job["opts"].wait_parameters()
    torch.cuda.synchronize(device)
    table = {k: round(v[0] / 10, 4) for k, v in _lib_mod.profile_collect().items()}
    _lib_mod.profile_enable(False)
    split = enqueue_vs_gpu(job["step"], warmup + steps + 10, 20, device)
    job["opts"].wait_parameters()
    out = {"workload": "the reference's neus-facto preset as shipped (method_configs.py:452-500): 2x256 geo + 2x256 colour MLP, 16x2x2^19 smoothstep "
                       f"grid, {rays} rays x {samples} samples (+256/96 proposal samples) per GPU per step, full train step incl. Adam and schedulers",
           "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3), "iters_per_sec": round(1e3 / ms, 2),
           "value": round(world * rays * samples / (dt / steps), 1), "unit": "ray-samples/s",
           "published": {"iters_per_sec": 22.0, "hardware": "RTX 3090", "source": "reference README.md:83 (BASELINE.md section 2)"},
           "vs_published": round(1e3 / ms / 22.0, 2),
           "kernels_ms_per_step": table, "native_kernel_ms_per_step": round(sum(table.values()), 3), "enqueue_vs_gpu": split}
    del job, loss
    torch.cuda.empty_cache()
    return out


def _leg_tail(job, device, first, steps, table_steps=5, split_steps=10):
    """The per-kernel table (instrumented pass) and the host-enqueue / GPU split of a leg's step, after its timed steps."""
    from sdfstudio_amd import _lib

    _lib.profile_enable(True)
    for i in range(table_steps):
        job["step"](first + i)
    job["opts"].wait_parameters()
    torch.cuda.synchronize(device)
    table = {k: round(v[0] / table_steps, 4) for k, v in _lib.profile_collect().items()}
    _lib.profile_enable(False)
    split = enqueue_vs_gpu(job["step"], first + table_steps, split_steps, device)
    job["opts"].wait_parameters()
    return table, split


def volsdf_legs(device, world, rank, steps=20, warmup=5):
    """BASELINE config 1's model ON THE GPU: VolSDF (models/volsdf.py:56-79) with the pure-MLP field (8 x 256 + 4 x 256,
    use_grid_feature = False, positional encoding), ErrorBoundedSampler (ray_samplers.py:581-702: Algorithm 1 with up to 5 outer iterations,
    each ending in the reference's own HOST decision `beta.max() > beta0`, :665), 64 + 32 samples per ray, Laplace density, L1 + eikonal,
    backward, Adam with the exponential schedule of the `volsdf` / `monosdf` presets (method_configs.py:581-614, 616-650).  Two batches:
    config 1's 512 rays and the 4096 rays of config 2; `monosdf` = the same model with the two monocular-prior losses
    (base_surface_model.py:419-437) at the preset's 1024 rays."""
    from sdfstudio_amd.engine.optimizers import exponential_decay_scheduler
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import SceneBox
    from sdfstudio_amd.models.volsdf import VolSDFModel, VolSDFModelConfig

    def build(mono):
        def f(dev):
            torch.manual_seed(0)
            fcfg = SDFFieldConfig(bias=0.5, inside_outside=False, use_grid_feature=False, beta_init=0.1)
            kw = {"mono_depth_loss_mult": 0.1, "mono_normal_loss_mult": 0.05} if mono else {}
            box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
            return VolSDFModel(VolSDFModelConfig(sdf_field=fcfg, background_model="none", **kw), box, num_train_data=49).to(dev).train()
        return f

    def mono_batch(n, gen):
        nrm = torch.randn(n, 3, device=device, generator=gen)
        return {"depth": torch.rand(n, device=device, generator=gen), "normal": nrm / nrm.norm(dim=-1, keepdim=True)}

    sched = {"fields": {"lr": 5e-4, "scheduler": exponential_decay_scheduler(0.1, 200000)},
             "field_background": {"lr": 5e-4, "scheduler": exponential_decay_scheduler(0.1, 200000)}}
    out = {"workload": "VolSDF, pure-MLP field (8x256 geo + 4x256 colour, no hash grid, PE), ErrorBoundedSampler (128 evaluation samples per outer "
                       "iteration, <= 5 iterations), 64 + 32 samples per ray, full train step incl. Adam", "steps": steps, "warmup": warmup}
    for name, rays, mono in (("config1_512rays", 512, False), ("rays4096", 4096, False), ("monosdf_preset_1024rays", 1024, True)):
        job = make_job(2, device, world, rank, rays=rays, model_factory=build(mono), batch_fn=mono_batch if mono else None, opt_config=sched)
        smp = job["model"].sampler
        dt, _prof, loss = timed_steps(job, 0, warmup, steps, ("geo_fwd_kernel", "geo_bwd_kernel"), world)
        dt = max_over_ranks(dt, device, world)
        assert math.isfinite(float(loss.detach())), f"volsdf leg {name} diverged"
        ms = dt / steps * 1e3
        table, split = _leg_tail(job, device, warmup + steps, steps)
        out[name] = {"ms_per_step": round(ms, 3), "iters_per_sec": round(1e3 / ms, 2), "rays": rays, "samples_per_ray": 96,
                     "value": round(world * rays * 96 / (dt / steps), 1), "unit": "ray-samples/s",
                     "error_bound_iterations_last_step": int(smp.last_total_iters), "host_reads_per_step": int(smp.last_total_iters),
                     "sdf_evaluations_per_ray_in_sampler": 128 * int(smp.last_total_iters),
                     "kernels_ms_per_step": table, "native_kernel_ms_per_step": round(sum(table.values()), 3), "enqueue_vs_gpu": split}
        job["flat"].close()
        del job, loss
        torch.cuda.empty_cache()
    out["host_reads_note"] = ("one device -> host read per outer iteration of Algorithm 1: the reference's own global decision `beta.max() > beta0` "
                              "(ray_samplers.py:665) - while ANY ray is above beta0, EVERY ray gets 128 more samples, so a fixed iteration count "
                              "would change the sample sets of the rays that had converged (not the reference's result)")
    return out


def config4_leg(device, world, rank, steps=15, warmup=5):
    """BASELINE config 4: config 2's NeuS-facto kernels in an INDOOR scene - inside_outside = True (sign-flipped geometric
    init, sdf_field.py:294-299), cameras INSIDE the box, near / far from the AABB box collider (scene_colliders.py:47-109), monocular depth
    (scale-and-shift invariant, losses.py:392-409) and normal (losses.py:264-275) priors with the README's multipliers (README.md:74:
    0.1 / 0.05; base_surface_model.py:419-437).  4096 rays x 128 samples (N = 0 mod 32: the depth loss reshapes to (1, 32, -1))."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.model_components.scene_colliders import build_collider
    from sdfstudio_amd.models.neus_facto import NeuSFactoModel, NeuSFactoModelConfig, SceneBox

    def build(dev):
        torch.manual_seed(0)
        fcfg = SDFFieldConfig(num_layers=8, hidden_dim=256, geo_feat_dim=256, num_layers_color=4, hidden_dim_color=256, bias=0.8,
                              inside_outside=True, use_grid_feature=True, beta_init=0.3, num_levels=16, max_res=2048, base_res=16,
                              log2_hashmap_size=19, hash_features_per_level=2, hash_smoothstep=True)
        mcfg = NeuSFactoModelConfig(sdf_field=fcfg, num_proposal_samples_per_ray=(256, 96), num_neus_samples_per_ray=N_SAMPLES,
                                    background_model="none", mono_depth_loss_mult=0.1, mono_normal_loss_mult=0.05)
        box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.05, far=4.0, collider_type="box")
        m = NeuSFactoModel(mcfg, box, num_train_data=49)
        m.collider = build_collider(m.scene_box, m.config)
        return m.to(dev).train()

    def rays(n, gen):  # camera centres inside the room, directions over the whole sphere (Replica-like)
        o = (torch.rand(n, 3, device=device, generator=gen) - 0.5) * 0.6
        d = torch.randn(n, 3, device=device, generator=gen)
        d = d / d.norm(dim=-1, keepdim=True)
        cam = (torch.rand(n, device=device, generator=gen) * 49).long().clamp_(max=48)
        return o.contiguous(), d.contiguous(), torch.ones(n, 1, device=device), cam

    def mono_batch(n, gen):
        nrm = torch.randn(n, 3, device=device, generator=gen)
        return {"depth": torch.rand(n, device=device, generator=gen), "normal": nrm / nrm.norm(dim=-1, keepdim=True)}

    job = make_job(2, device, world, rank, rays=N_RAYS, model_factory=build, ray_fn=rays, batch_fn=mono_batch)
    dt, _prof, loss = timed_steps(job, 0, warmup, steps, "geo_bwd_kernel", world)
    dt = max_over_ranks(dt, device, world)
    assert math.isfinite(float(loss.detach())), "config 4 leg diverged"
    ms = dt / steps * 1e3
    table, split = _leg_tail(job, device, warmup + steps, steps)
    out = {"workload": "BASELINE config 4: NeuS-facto (config 2's grid and networks) with inside_outside = True, cameras inside the box, AABB box "
                       f"collider, mono depth (0.1) + normal (0.05) prior losses; {N_RAYS} rays x {N_SAMPLES} samples (+256/96 proposal samples), "
                       "full train step incl. Adam", "steps": steps, "warmup": warmup,
           "ms_per_step": round(ms, 3), "iters_per_sec": round(1e3 / ms, 2), "value": round(world * N_RAYS * N_SAMPLES / (dt / steps), 1),
           "unit": "ray-samples/s", "host_reads_per_step": 0, "kernels_ms_per_step": table,
           "native_kernel_ms_per_step": round(sum(table.values()), 3), "enqueue_vs_gpu": split}
    job["flat"].close()
    del job, loss
    torch.cuda.empty_cache()
    return out


def neus_acc_leg(device, steps=20, warmup=5):
    """SURVEY row f2, the packed-sample path (ray_samplers.py:1315-1503, models/neus_acc.py:88-148): the reference's `neus-acc` preset
    (method_configs.py:937-970: NeuSAccModelConfig defaults - 8 x 256 geometry + 4 x 256 colour MLP, positional encoding only, 2048 rays
    per step) AFTER its first occupancy-grid update: the march through the pruned 128^3 grid, NeuS up-sampling inside it, packed alpha
    compositing, losses, backward, Adam.  The field is a geometric-init sphere with a trained-looking sharpness (inv_s = e^5) so that the
    pruning removes what a trained scene's would; background_model "none" (the preset's default NeRFField background is a separate
    dense path, BASELINE configs do not use it)."""
    from sdfstudio_amd import _lib
    from sdfstudio_amd.cameras.rays import RayBundle
    from sdfstudio_amd.distributed import FlatGradients
    from sdfstudio_amd.engine.optimizers import Optimizers, neus_scheduler
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_acc import NeuSAccModel, NeuSAccModelConfig
    from sdfstudio_amd.models.neus_facto import SceneBox

    torch.manual_seed(0)
    rays = 2048
    fcfg = SDFFieldConfig(bias=0.5, beta_init=0.3, inside_outside=False)  # the preset's SDFFieldConfig(): 8x256 + 4x256, no feature grid
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
    model = NeuSAccModel(NeuSAccModelConfig(sdf_field=fcfg, background_model="none"), box, 49).to(device).train()
    # the bounded packed arrays: no device -> host read inside a step - the march step stays on the device, the packed
    # arrays are sized by a bound measured after the grid update and re-checked every 50 steps (ray_samplers.NeuSAccSampler(bounded=True))
    model.sampler.bounded = os.environ.get("SDFHIP_BENCH_ACC_EXACT") != "1"
    with torch.no_grad():
        model.field.deviation_network.variance.fill_(0.5)
    groups = {k: v for k, v in model.get_param_groups().items() if v}
    flat = FlatGradients([p for g in groups.values() for p in g], buckets=list(groups.values()))
    opts = Optimizers({k: {"lr": 5e-4, "scheduler": neus_scheduler(500, 0.05, 20000)} for k in groups}, groups, flat_grads=flat)
    centers, rot = synthetic_cameras(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(42)
    model.before_train_iteration(2000)
    model.after_train_iteration(2000)  # the first occupancy-grid update (ray_samplers.py:1383-1432): from here on the packed path runs
    occupied = float(model.sampler._binary.float().mean())
    kept = []

    def step(i):
        model.before_train_iteration(i)
        o, d, norm, cam = draw_rays(centers, rot, rays, gen)
        image = torch.rand(rays, 3, device=device, generator=gen)
        out = model(RayBundle(origins=o, directions=d, directions_norm=norm, camera_indices=cam[:, None]))
        nv = model.sampler.packed_valid
        kept.append((out["ray_samples"].shape[0] if nv is None else nv) if "ray_indices" in out else -1)
        loss = functools.reduce(operator.add, model.get_loss_dict(out, {"image": image}).values())
        flat.zero(loss)
        loss.backward()
        opts.optimizer_step_all(grad_scale=None)
        opts.scheduler_step_all(i)
        return loss

    for i in range(warmup):
        step(2001 + i)
    torch.cuda.synchronize(device)
    kept.clear()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(2001 + warmup + i)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    kept = [int(k) for k in kept]  # (device scalars in the bounded form: read AFTER the timed region)
    assert math.isfinite(float(loss.detach())) and min(kept) > 0, "neus-acc leg: no packed samples / diverged"
    n_kept = sum(kept) / len(kept)
    model.sampler.check_capacity()
    _lib.profile_enable(True)
    for i in range(5):
        step(2001 + warmup + steps + i)
    torch.cuda.synchronize(device)
    table = {k: round(v[0] / 5, 4) for k, v in _lib.profile_collect().items()}
    _lib.profile_enable(False)
    split = enqueue_vs_gpu(step, 2001 + warmup + steps + 5, 10, device)
    out = {"workload": "the reference's neus-acc preset (method_configs.py:937-970: 8x256 geo + 4x256 colour MLP, PE only) after its first occupancy-grid "
                       f"update, {rays} rays per step, packed samples; full train step incl. Adam", "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt * 1e3, 3), "iters_per_sec": round(1.0 / dt, 2), "rays_per_step": rays,
           "samples_kept_per_ray": round(n_kept / rays, 2), "packed_samples_per_step": round(n_kept, 1), "occupied_voxel_fraction": round(occupied, 5),
           "march_step_size": float(model.sampler.step_size), "value": round(n_kept / dt, 1), "unit": "packed ray-samples/s",
           "dense_equivalent": "NeuS samples 64 + 64 per ray on every ray (models/neus.py:34-47): 128 samples per ray",
           "kernels_ms_per_step": table, "enqueue_vs_gpu": split,
           "bounded_packed_arrays": bool(model.sampler.bounded), "packed_capacity": model.sampler._cap,
           "fill_of_capacity": None if not model.sampler._cap else round(n_kept / model.sampler._cap, 3),
           "overflowed_steps": int(model.sampler.overflowed_steps), "host_reads_per_step": 0 if model.sampler.bounded else 2,
           "enqueue_note": "bounded form: no device -> host read inside a step (the reference's nerfacc call returns exact-size tensors: one read "
                           "for the count, one for the step size, ray_samplers.py:1379-1382,1474-1484); the bound is re-checked every 50 steps"}
    del model, flat, opts, loss
    torch.cuda.empty_cache()
    return out


def exchange_at_n1_legs(device, steps=8, warmup=3):
    """the data-parallel exchange EXECUTED ON RCCL on the one GPU a box has.  A process group of one rank, backend "nccl"
    (= RCCL), SDFHIP_FORCE_EXCHANGE=1: every bucket's reduce_scatter_tensor / all_gather_into_tensor (sharded) or all_reduce (bucketed) is
    issued on RCCL's stream exactly as at N > 1 - launched from the autograd hooks and from inside the native backward (table callback),
    waited for chunk by chunk on the compute stream, parameters gathered behind the step's Adam - and returns its input.  What it measures is
    the exchange's OVERHEAD at N = 1 (launches, stream hand-offs, the extra pass structure), config 2 and config 5, against the same steps
    with no process group in play; what it cannot measure is wire time (reference seam: pipelines/base_pipeline.py:241-243,
    scripts/train.py:127-145)."""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    out = {"backend": dist.get_backend(), "world_size": 1, "steps": steps, "warmup": warmup,
           "what": "ms per training step with the exchange forced on (RCCL collectives of one rank) vs off; same model, rays and schedules"}
    try:
        for key, cfg, first, dominant in (("config2", 2, 0, "geo_bwd_kernel"), ("config5_levels8", 5, 0, "geo_encode_kernel"),
                                          ("config5_levels16", 5, 200000, "geo_encode_kernel")):
            res = {}
            for mode in ("none", "shard", "allreduce"):
                if mode == "none":
                    os.environ.pop("SDFHIP_FORCE_EXCHANGE", None)
                else:
                    os.environ["SDFHIP_FORCE_EXCHANGE"] = "1"
                    os.environ["SDFHIP_BENCH_EXCHANGE"] = mode
                job = make_job(cfg, device, 1, 0)
                dt, _prof, loss = timed_steps(job, first, warmup, steps, dominant, 1)
                assert math.isfinite(float(loss.detach())), f"exchange leg {key}/{mode} diverged"
                flat = job["flat"]
                r = {"ms_per_step": round(dt / steps * 1e3, 3)}
                if mode != "none":
                    assert job["shard"] == (mode == "shard") and flat.exchanging
                    r.update({"collectives_per_step": flat.last_collectives,
                              "gather_collectives_per_step": flat.last_gather_collectives // max(flat._finished_steps, 1) if job["shard"] else 0,
                              "buffer_bytes_per_step": 4 * flat.exchanged_numel(), "gathered_bytes_per_step": flat.gathered_bytes(),
                              "buckets_launched_during_backward": flat.last_overlapped_buckets,
                              "buckets_launched_from_inside_the_native_backward": flat.last_early_buckets,
                              "exposed_ms_per_step": job.get("exposed")})
                res[mode] = r
                job["flat"].close()
                del job, loss, flat
                torch.cuda.empty_cache()
            for mode in ("shard", "allreduce"):
                res[mode]["overhead_ms_per_step"] = round(res[mode]["ms_per_step"] - res["none"]["ms_per_step"], 3)
            out[key] = res
    finally:
        os.environ.pop("SDFHIP_FORCE_EXCHANGE", None)
        os.environ.pop("SDFHIP_BENCH_EXCHANGE", None)
        dist.destroy_process_group()
    return out


def child_leg(argv, timeout=420, env=None):
    """A leg that runs in a child process (its own HIP context, its own process group) and prints one JSON object; never raises: a
    failure is reported in the leg's own object and the bench line survives."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, *argv], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": f"rc {r.returncode}", "stderr_tail": r.stderr[-600:]}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def collective_report(job, backend, exposed_reduce_by_rank, exposed_gather_by_rank):
    """The data-parallel exchange of the timed steps, phase by phase: bytes per step per rank and the GPU time the compute stream stalled
    on each phase (HIP events around the waits, one value per rank)."""
    flat = job["flat"]
    shard = job["shard"]
    n_in = 4 * flat.exchanged_numel()
    w = flat.world
    native = backend == "nccl"
    rep = {"backend": backend, "exchange": "sharded: reduce-scatter -> fused Adam on the owned 1 / N slice -> all-gather" if shard else
                                          "all-reduce of the flat gradient, replicated Adam",
           "buckets": len(flat._buckets), "chunk_bytes": None if flat._chunk is None else 4 * flat._chunk,
           "buckets_launched_during_backward": flat.last_overlapped_buckets,
           "buckets_launched_from_inside_the_native_backward": flat.last_early_buckets, "parameters_outside_the_graph": flat.last_unused,
           "adam_elements_visited_per_rank": job["opts"].adam.last_elements_visited,
           "phases": {
               "reduce": {"collective": ("reduce_scatter" if native else "all_reduce of the grid chunk (gloo has no reduce-scatter: same sum in the owned slice)") if shard else "all_reduce",
                          "collectives_per_step": flat.last_collectives, "buffer_bytes_per_step_per_rank": n_in,
                          # ring / direct algorithms move (W - 1) / W of the buffer out of (and into) every rank per reduce-scatter; an all-reduce twice that
                          "wire_bytes_out_per_rank": int(n_in * (w - 1) / w * (1 if shard else 2)),
                          "exposed_ms_per_step_by_rank": exposed_reduce_by_rank,
                          "overlap": "bucket collectives leave in fixed index order from post-accumulate-grad hooks during backward; exposed = GPU "
                                     "time the compute stream stalled in the chunk waits before the owned slices' Adam"},
               "gather": None if not shard else {
                   "collective": "all_gather_into_tensor" if native else "all_gather (list of views)",
                   "collectives_per_step": flat.last_gather_collectives // max(flat._finished_steps, 1),
                   "buffer_bytes_per_step_per_rank": flat.gathered_bytes(),
                   "wire_bytes_out_per_rank": int(flat.gathered_bytes() * (w - 1) / w),
                   "exposed_ms_per_step_by_rank": exposed_gather_by_rank,
                   "overlap": "issued after the step's Adam, small buckets first, the SDF table last; the next step waits for the small ones "
                              "before its callbacks and for the table after its proposal sampling has been enqueued"}}}
    return rep


def replicas_in_sync(job, device, world):
    """N > 1: every rank must hold the SAME parameters after the exchange (base_pipeline.py:241-243: DDP's invariant) - bit for bit, since
    every rank receives the same sums (all-reduce) or the same updated slices (all-gather).  A 64-bit checksum of every parameter's bit
    pattern, MIN and MAX over the ranks: equal = in sync.  Outside the timed region; two collectives of one int64."""
    import torch.distributed as dist

    job["opts"].wait_parameters()
    torch.cuda.synchronize()
    h = torch.zeros(1, dtype=torch.int64, device=device)
    n = 0
    for g in job["groups"].values():
        for i, p in enumerate(g):
            bits = p.detach().contiguous().view(torch.int32).to(torch.int64)
            h += (bits * ((n + i) % 8191 + 1)).sum()  # position-weighted: two parameters swapping their differences do not cancel
        n += len(g)
    lo, hi = h.clone(), h.clone()
    if world > 1:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return {"bit_identical_across_ranks": bool(int(lo.item()) == int(hi.item())), "parameters_checked": n}


def sdf_flops_per_point():
    """2 x MACs of the geometry network of config 2 evaluated for its sdf row alone (SDFHIP_MODE_SDF: the 256 feature rows of the output
    layer are not computed): 71->256, 2 x 256->256, 256->185, skip layer (185 + 71)->256, 3 x 256->256, 256->1."""
    return 2 * (71 * 256 + 2 * 256 * 256 + 256 * 185 + 256 * 256 + 3 * 256 * 256 + 256)


def dense_sdf_leg(model, device, resolution=(512, 512, 256), reps=3):
    """SURVEY row f4, the compute-bound inference path: `utils/marching_cubes.sdf_on_grid` (the device side of scripts/extract_mesh.py:94-133
    / utils/marching_cubes.py:15-168) over a 2^26-point lattice on config 2's network - hash-grid encode + the sdf row of the 8 x 256 geometry
    MLP, nothing saved, nothing read back but the sdf.  `roofline` is the MFMA kernel (geo_fwd_kernel, inference instantiation): algorithmic
    flops of the sdf row x the 3 split-precision terms issued per product / its launch time (HIP events on the launch stream, all launches of
    the timed repetitions), against the dense 16-bit MFMA peak."""
    from sdfstudio_amd import _lib
    from sdfstudio_amd.utils.marching_cubes import sdf_on_grid

    field = model.field
    lo, hi = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
    P = resolution[0] * resolution[1] * resolution[2]
    with torch.no_grad():
        sdf_on_grid(field, lo, hi, (64, 64, 64), device=device)  # sizes the allocator
        sdf_on_grid(field, lo, hi, resolution, device=device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            vol = sdf_on_grid(field, lo, hi, resolution, device=device)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        # a second, instrumented pass: events on the two kernels of the path
        _lib.profile_enable_only(["geo_fwd_kernel", "geo_encode_kernel"])
        for _ in range(reps):
            sdf_on_grid(field, lo, hi, resolution, device=device)
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
    assert bool(torch.isfinite(vol).all()), "dense sdf: non-finite values"
    g = sdf_flops_per_point()
    k_ms, k_n = prof.get("geo_fwd_kernel", (0.0, 0))
    e_ms, _ = prof.get("geo_encode_kernel", (0.0, 0))
    k_s = k_ms / reps * 1e-3
    issued = 3 * g * P / k_s / 1e12 if k_s > 0 else 0.0
    enc_bytes = (16 * 8 * 2 * 4 + 12 + 3 * 128) * P  # gather + position in + the tile-packed in0 (3 blocks) out
    return {"workload": f"sdf_on_grid: {resolution[0]} x {resolution[1]} x {resolution[2]} = 2^{int(math.log2(P))} lattice points, config 2's field "
                        "(16x2x2^19 smoothstep grid + 8x256 geometry MLP, sdf row only), no grad, nothing saved",
            "points": P, "ms": round(wall * 1e3, 3), "value": round(P / wall, 1), "unit": "points/s",
            "kernels_ms": {"geo_fwd_kernel": round(k_ms / reps, 3), "geo_encode_kernel": round(e_ms / reps, 3)},
            "launches": k_n // reps if reps else 0,
            "roofline": {"kernel": "geo_fwd_kernel<..., GRAD = false, SAVE = false, FEAT = false> (SDFHIP_MODE_SDF)", "bound": "mfma",
                         "achieved": round(issued, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(issued / PEAK_BF16_MFMA_TFLOPS, 4),
                         "achieved_is": f"sdf-row flops {g / 1e6:.3f} MFLOP per point x 3 issued 16-bit MFMA terms per fp32-class product x "
                                        "points / the kernel's launch time (HIP events)",
                         "flops_per_point": g, "terms_per_product": 3, "points_per_s_kernel_only": round(P / k_s, 1) if k_s > 0 else None,
                         # the same figure over the WHOLE leg (encode + host glue included), and at the yardstick VERDICT r4 used (the full
                         # geometry network's G = 1.049 MFLOP per point, although MODE_SDF does not compute the 256 feature rows)
                         "frac_whole_leg": round(3 * g * P / wall / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                         "frac_kernel_at_full_network_G": round(3 * flops_per_sample()[0] * P / k_s / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4) if k_s > 0 else None,
                         "traffic": None, "algorithmic_bytes_per_launch": (3 * 128 + 4) * P // max(1, k_n // reps if reps else 1),
                         # HBM bytes per launch of the MODE_SDF instantiation (a launch = one chunk of the lattice) from the committed PMC passes
                         **(pmc_traffic_per_launch(lambda f: "_eval" in f, lambda k: geo_fwd_flags(k)[:3] == ["false", "false", "false"]) or {})},
            "encode": {"GBps_on_gather_bytes": round(enc_bytes / (e_ms / reps * 1e-3) / 1e9, 1) if e_ms > 0 else None,
                       "bytes_per_point": enc_bytes // P}}


def mesh_leg(timeout=300):
    """SURVEY row f4, the step after the dense SDF evaluation: marching cubes of one 512^3 crop on the device (libsdfmesh.so), in a child
    process (tools/mesh_leg.py says why).  Never raises: a failure is reported in the leg's own object."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mesh_leg.py")], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": f"rc {r.returncode}", "stderr_tail": r.stderr[-600:]}
    except Exception as e:  # noqa: BLE001 - the bench line must survive this leg
        return {"error": repr(e)[:300]}


def forward_only_leg(job, device, n_rays, n_samples, reps=10):
    """the eval-mode render (no grad, nothing saved for a backward) of one batch, timed as a whole and - in a second,
    instrumented pass - per kernel.  Its MFMA roofline: forward G, analytic-normal chain G, colour C per ray-sample, 3 issued terms."""
    from sdfstudio_amd import _lib
    from sdfstudio_amd.cameras.rays import RayBundle

    model = job["model"]
    model.eval()
    o, d, norm, cam = draw_rays(job["centers"], job["rot"], n_rays, job["gen"])
    rb_eval = RayBundle(origins=o, directions=d, directions_norm=norm, camera_indices=cam[:, None])
    with torch.no_grad():
        for _ in range(2):  # the first calls size the caching allocator for the forward-only workspace
            model(rb_eval)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            model(rb_eval)
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t1) / reps * 1e3
        _lib.profile_enable(True)
        for _ in range(reps):
            model(rb_eval)
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
        # the caller of the eval path: one camera's whole 384 x 384 image through Model.get_outputs_for_camera_ray_bundle
        # (models/base_model.py:165-189) in row-major chunks of eval_num_rays_per_chunk (4096, the ModelConfig default)
        image = None
        if n_rays >= 4096:
            from sdfstudio_amd.cameras.rays import generate_image_rays

            cam_rays = generate_image_rays(job["centers"][7], job["rot"][7], 384, 384, 925.5, 922.6, 199.4, 198.1, camera_index=7)
            model.get_outputs_for_camera_ray_bundle(cam_rays)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            img = model.get_outputs_for_camera_ray_bundle(cam_rays)
            torch.cuda.synchronize()
            img_ms = (time.perf_counter() - t2) * 1e3
            image = {"ms_per_image": round(img_ms, 2), "pixels": 384 * 384, "chunks": -(-384 * 384 // int(model.config.eval_num_rays_per_chunk)),
                     "rays_per_s": round(384 * 384 / (img_ms * 1e-3), 1), "value": round(384 * 384 * n_samples / (img_ms * 1e-3), 1),
                     "unit": "ray-samples/s", "mean_accumulation": round(float(img["accumulation"].mean()), 4)}
            del img, cam_rays
    model.train()
    P = n_rays * n_samples
    g, c = flops_per_sample()
    kernels = {k: round(v[0] / reps, 4) for k, v in prof.items()}
    mfma_ms = sum(prof.get(k, (0.0, 0))[0] for k in ("geo_fwd_kernel", "col_fwd_kernel")) / reps
    issued = 3 * (2 * g + c) * P / (mfma_ms * 1e-3) / 1e12 if mfma_ms > 0 else 0.0
    # bytes of the forward -> chain hand-over: the chain runs top-down over s'(z_l) of EVERY layer, 8 x 256 floats per point, which no
    # on-chip store of a 128-point workgroup holds (1 MB): written by the forward launch, read by the chain launch
    handover = 2 * 8 * 256 * 4 * P
    return {"value": round(P / (fwd_ms * 1e-3), 1), "unit": "ray-samples/s per GPU (eval-mode render, no grad)", "ms_per_batch": round(fwd_ms, 3),
            "kernels_ms_per_batch": kernels, "image_384x384": image,
            "roofline": {"kernel": "geo_fwd_kernel (forward launch + chain launch, nothing saved) + col_fwd_kernel", "bound": "mfma",
                         "achieved": round(issued, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(issued / PEAK_BF16_MFMA_TFLOPS, 4),
                         "achieved_is": "(2G + C) flops per ray-sample x 3 issued 16-bit MFMA terms / the three launches' time (HIP events, "
                                        "instrumented pass)",
                         "mfma_kernels_ms": round(mfma_ms, 3), "traffic": None,
                         # HBM bytes of the three launches of one batch from the committed PMC passes (forward | chain | colour, nothing-saved forms)
                         **(pmc_traffic_per_launch(lambda f: "_eval" in f, lambda k: geo_fwd_flags(k)[:3] == ["true", "false", "true"] or
                                                   (k.startswith("col_fwd_kernel<") and k.replace(" ", "").endswith(",false>"))) or {}),
                         "forward_to_chain_handover_bytes": handover,
                         "handover_note": "u_l = s(z_l) of all 8 layers, 8 KiB per point: written by the forward launch, read by the chain "
                                          "launch (DESIGN.md section 4.1: why no on-chip store holds it)"}}


def _pick(d, keys):
    if d is None:
        return None
    if "error" in d:
        return {"error": str(d["error"])[:120]}
    return {k: d[k] for k in keys if k in d and d[k] is not None}


def compact_line(line):
    """The one JSON line the driver records: the contract's keys, `roofline` and `cpu_baseline` whole-but-terse, every leg as {ms, value,
    frac, ...}.  Prose notes and per-kernel tables stay in gpurun_out/bench_detail.json.  Target <= 6 KB."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "iters_per_sec", "per_gpu", "final_loss", "library_digest", "model_tflops", "mfma_kernels_ms_per_step")
    out = {k: line[k] for k in keep if k in line}
    out["dtype_note"] = "fp32 tensors + accumulators; products as 3 split 16-bit MFMA terms (hi + lo operand parts)"
    cfg = dict(line["config"])
    cfg["workload"] = cfg["workload"].split(":")[0] + ": " + ("NeuS-facto 16x2x2^19 grid + 8x256 geo + 4x256 colour MLP, 4096 rays x 128 samples, full train step incl. Adam"
                                                               if cfg["workload"].startswith("BASELINE config 2") else cfg["workload"].split(":", 1)[-1].strip()[:140])
    out["config"] = cfg
    r = line.get("roofline")
    if r is not None:
        out["roofline"] = _pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "avg_launch_ms", "launches",
                                    "ideal_io_bytes", "traffic_over_ideal", "frac_of_fp32_class_ceiling", "frac_algorithmic_of_16bit_peak",
                                    "frac_algorithmic_of_fp32_matrix_peak", "terms_per_product", "algorithmic_bytes", "ms_per_step", "levels_active"))
        if "hbm" in r:
            out["roofline"]["hbm"] = _pick(r["hbm"], ("dataflow_GBps", "dataflow_frac", "waste_ratio"))
    out["encode_roofline"] = _pick(line.get("encode_roofline"), ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "ms_per_step", "algorithmic_bytes"))
    if line.get("enqueue_vs_gpu"):
        out["enqueue_vs_gpu"] = _pick(line["enqueue_vs_gpu"], ("host_enqueue_ms_one_step_from_idle", "gpu_ms_per_step", "host_bound"))
    out["step_roofline"] = _pick(line.get("step_roofline"), ("model_tflops", "frac_of_fp32_matrix_peak", "issued_16bit_mfma_tflops", "frac_of_dense_bf16_peak",
                                                                "hbm_GB_per_step_pmc", "hbm_frac_of_8TBps", "hbm_pmc_stale"))
    legs = {}
    c5 = line.get("config5")
    if c5:
        for k in ("levels8", "levels16"):
            if k in c5:
                legs["config5_" + k] = {"ms": c5[k]["ms_per_step"], "value": c5[k]["value"], "frac": (c5[k].get("roofline") or {}).get("frac"),
                                        "encode_GBps": (c5[k].get("roofline") or {}).get("achieved"), "traffic": (c5[k].get("roofline") or {}).get("traffic")}
    bm = line.get("bigmlp")
    if bm:
        for k in ("preset_batch", "config2_batch"):
            if k in bm:
                legs["bigmlp_" + k] = {"ms": bm[k]["ms_per_step"], "value": bm[k]["value"], "frac": bm[k]["step_roofline"]["frac"],
                                       "ratio_to_256_wide_step": bm[k].get("ratio_to_256_wide_step")}
    pr = line.get("preset")
    if pr:
        legs["preset"] = _pick(pr, ("ms_per_step", "iters_per_sec", "value", "vs_published"))
        if "enqueue_vs_gpu" in pr:
            legs["preset"].update({"host_enqueue_ms": pr["enqueue_vs_gpu"]["host_enqueue_ms_one_step_from_idle"], "host_bound": pr["enqueue_vs_gpu"]["host_bound"]})
    na = line.get("neus_acc")
    if na:
        legs["neus_acc"] = _pick(na, ("ms_per_step", "value", "unit", "samples_kept_per_ray", "host_reads_per_step"))
        if "enqueue_vs_gpu" in na:
            legs["neus_acc"].update({"host_enqueue_ms": na["enqueue_vs_gpu"]["host_enqueue_ms_one_step_from_idle"], "gpu_ms": na["enqueue_vs_gpu"]["gpu_ms_per_step"],
                                     "host_bound": na["enqueue_vs_gpu"]["host_bound"]})
    vs = line.get("volsdf")
    if vs:
        if "error" in vs:
            legs["volsdf"] = _pick(vs, ())
        for k in ("config1_512rays", "rays4096", "monosdf_preset_1024rays"):
            if k in vs:
                legs["volsdf_" + k] = {"ms": vs[k]["ms_per_step"], "value": vs[k]["value"], "iterations": vs[k]["error_bound_iterations_last_step"],
                                       "host_reads": vs[k]["host_reads_per_step"], "host_enqueue_ms": vs[k]["enqueue_vs_gpu"]["host_enqueue_ms_one_step_from_idle"],
                                       "gpu_ms": vs[k]["enqueue_vs_gpu"]["gpu_ms_per_step"], "host_bound": vs[k]["enqueue_vs_gpu"]["host_bound"]}
    c4 = line.get("config4")
    if c4:
        legs["config4"] = _pick(c4, ("ms_per_step", "value", "host_reads_per_step"))
        if "enqueue_vs_gpu" in c4:
            legs["config4"].update({"host_enqueue_ms": c4["enqueue_vs_gpu"]["host_enqueue_ms_one_step_from_idle"], "host_bound": c4["enqueue_vs_gpu"]["host_bound"]})
    ex = line.get("exchange_at_n1")
    if ex:
        e = {"backend": ex.get("backend")} if "error" not in ex else _pick(ex, ())
        for k in ("config2", "config5_levels8", "config5_levels16"):
            if k in ex:
                e[k] = {"none_ms": ex[k]["none"]["ms_per_step"], "shard_ms": ex[k]["shard"]["ms_per_step"], "allreduce_ms": ex[k]["allreduce"]["ms_per_step"],
                        "shard_collectives": ex[k]["shard"]["collectives_per_step"] + ex[k]["shard"]["gather_collectives_per_step"],
                        "bytes": ex[k]["shard"]["buffer_bytes_per_step"]}
        legs["exchange_at_n1"] = e
    fo = line.get("forward_only")
    if fo:
        legs["forward_only"] = {"ms": fo["ms_per_batch"], "value": fo["value"], "frac": fo["roofline"]["frac"], "traffic": fo["roofline"].get("traffic")}
        if fo.get("image_384x384"):
            legs["forward_only"]["image_384x384_ms"] = fo["image_384x384"]["ms_per_image"]
    ds = line.get("dense_sdf")
    if ds:
        legs["dense_sdf"] = {"ms": ds["ms"], "value": ds["value"], "unit": ds["unit"], "frac": ds["roofline"]["frac"], "traffic": ds["roofline"].get("traffic")}
    me = line.get("mesh")
    if me:
        if "error" in me:
            legs["mesh"] = _pick(me, ())
        else:
            legs["mesh"] = {"ms": me["ms"], "value": me["value"], "unit": me["unit"], "frac": me["roofline"]["frac"], "traffic": me["roofline"].get("traffic"),
                            "algorithmic_bytes": me["roofline"]["algorithmic_bytes"], "vertices": me["vertices"], "faces": me["faces"],
                            "stream_kernel_frac": (me["roofline"].get("stream_kernel") or {}).get("frac_of_8TBps"),
                            "extract_mesh_ms": (me.get("extract_mesh") or {}).get("ms"),
                            "cpu_points_per_s": (me.get("cpu_baseline") or {}).get("value")}
    out["legs"] = legs
    if line.get("appended_legs"):
        out["appended_legs"] = line["appended_legs"]
    col = line.get("collective")
    if col:
        c = {k: col[k] for k in ("backend", "exchange", "buckets", "chunk_bytes", "buckets_launched_during_backward",
                                 "buckets_launched_from_inside_the_native_backward", "parameters_outside_the_graph", "adam_elements_visited_per_rank")}
        c["phases"] = {ph: (None if v is None else {k: v[k] for k in v if k != "overlap"}) for ph, v in col["phases"].items()}
        c["replicas"] = col.get("replicas")
        out["collective"] = c
    else:
        out["collective"] = None
    k = line.get("kernels") or {}
    out["kernels_ms_per_step"] = {n: v["ms_per_step"] for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]}
    if "cpu_baseline" in line:
        out["cpu_baseline"] = _pick(line["cpu_baseline"], ("value", "unit", "cores", "kind", "sample"))
    if "cpu_baseline_reference" in line:
        out["cpu_baseline_reference"] = _pick(line["cpu_baseline_reference"], ("value", "unit", "cores", "kind"))
    out["detail"] = "gpurun_out/bench_detail.json (full tables and notes; profiles/r6_bench_detail.json is a committed copy of a builder run)"
    return out


def run(args):
    global N_RAYS, N_SAMPLES
    if args.small:
        N_RAYS, N_SAMPLES = 512, 16
    cfg5 = getattr(args, "config", 2) == 5
    if cfg5:
        N_RAYS, N_SAMPLES = 2048, 48  # method_configs.py:396 train_num_rays_per_batch, neus_facto.py:51 num_neus_samples_per_ray
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the sdfhip path has no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl") over xGMI; SDFHIP_BENCH_BACKEND=gloo exercises the N > 1 control flow on a single-GPU box
        backend = os.environ.get("SDFHIP_BENCH_BACKEND", "nccl")
        # a collective that does not complete aborts the run after this long (torch's default: 10 min): a hang costs minutes, not the caller's limit
        import datetime

        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("SDFHIP_BENCH_PG_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        print(f"[bench] rank {rank}/{world}: backend {dist.get_backend()}{' (RCCL)' if dist.get_backend() == 'nccl' else ''}, "
              f"device {local_rank} {torch.cuda.get_device_name(local_rank)}", file=sys.stderr, flush=True)

    from sdfstudio_amd import _lib

    if getattr(args, "only", None) == "exchange":
        assert world == 1, "--only exchange is the N = 1 leg"
        print(json.dumps(exchange_at_n1_legs(device, steps=min(args.steps, 8), warmup=min(args.warmup, 3))), flush=True)
        return
    if getattr(args, "only", None) in ("volsdf", "config4", "neus_acc"):
        # one appended leg alone (same function as in the default run): the rocprofv3 kernel stats of BASELINE configs 1 / 4 and of the
        # packed path come from these commands (tools/gpu_call_r6p.sh -> profiles/r6_{volsdf,config4,neus_acc}_kernel_stats.csv)
        leg = {"volsdf": lambda: volsdf_legs(device, world, rank), "config4": lambda: config4_leg(device, world, rank),
               "neus_acc": lambda: neus_acc_leg(device)}[args.only]()
        if rank == 0:
            print(json.dumps({args.only: leg}), flush=True)
        return
    job = make_job(5 if cfg5 else 2, device, world, rank, small=args.small)
    model, flat, groups = job["model"], job["flat"], job["groups"]
    step = job["step"]
    if getattr(args, "only", None) == "inference":
        # the two inference legs alone (same functions, same model as the default run): same-box A/Bs of library variants and the PMC
        # passes of profiles/r5_eval_* run this, so that every launch they see belongs to an inference path
        from sdfstudio_amd import build as _build

        out = {"library_digest": _build.built_digest() or None, "library": _lib.LIB_PATH,
               "forward_only": None if args.no_forward_only else forward_only_leg(job, device, N_RAYS, N_SAMPLES),
               "dense_sdf": None if args.no_dense_sdf else dense_sdf_leg(model, device, reps=args.steps if args.steps < 20 else 3)}
        if rank == 0:
            print(json.dumps(out), flush=True)
        return
    first = (200000 if args.levels == 16 else 0) if cfg5 else 0  # config 5: where in the preset's schedules the timed steps sit
    # Timed region: HIP events on the launches of the DOMINANT kernel only (the roofline figure must come from these steps).  An event
    # pair serialises the command stream around its launch; with every launch instrumented the step measured ~1 ms longer, so the
    # per-kernel table comes from a second, untimed pass with events everywhere (its step time is reported beside the table).
    dominant = "geo_encode_kernel" if cfg5 else "geo_bwd_kernel"
    dt_local, prof_timed, loss = timed_steps(job, first, args.warmup, args.steps, dominant, world)
    dt = dt_local
    table_steps = 0 if args.no_kernel_table else min(args.steps, 10)
    prof, instrumented_ms = dict(prof_timed), None
    if table_steps:
        _lib.profile_enable(True)
        t1 = time.perf_counter()
        for i in range(table_steps):
            step(first + args.warmup + args.steps + i)
        fence(world)
        instrumented_ms = (time.perf_counter() - t1) / table_steps * 1e3
        for k, v in _lib.profile_collect().items():  # scaled to the timed region's step count: the code below divides by args.steps
            prof.setdefault(k, (v[0] * args.steps / table_steps, v[1] * args.steps / table_steps))
        _lib.profile_enable(False)
    main_split = None
    if world == 1 and not args.small and table_steps:
        main_split = enqueue_vs_gpu(step, first + args.warmup + args.steps + table_steps, 8, device)
        job["opts"].wait_parameters()
    # SDFHIP_BENCH_ALLOW_NONFINITE=1: timing ablation builds (tools/build_variant.sh -DSDFHIP_ABL_*) compute wrong numbers on purpose
    assert math.isfinite(float(loss.detach())) or os.environ.get("SDFHIP_BENCH_ALLOW_NONFINITE") == "1", "training diverged"
    final_loss = float(loss.detach())
    # forward-only leg (eval-mode render, reported separately; outside the timed training region)
    fwd_only = None
    if not args.no_forward_only:
        fwd_only = forward_only_leg(job, device, N_RAYS, N_SAMPLES)
    dense = None
    if not cfg5 and not args.small and not args.no_dense_sdf:
        dense = dense_sdf_leg(model, device)
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    exposed_by_rank = exposed_gather_by_rank = None
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ex = job.get("exposed", {"reduce_ms_per_step": 0.0, "gather_ms_per_step": 0.0})
        allr = torch.zeros(2, world, device=device, dtype=torch.float64)  # all_reduce of one-hot rows: works on RCCL and on gloo alike
        allr[0, rank], allr[1, rank] = ex["reduce_ms_per_step"], ex["gather_ms_per_step"]
        dist.all_reduce(allr)
        exposed_gather_by_rank = [round(float(v), 4) for v in allr[1].tolist()]
        allr = allr[0]
        exposed_by_rank = [round(float(v), 4) for v in allr.tolist()]
    dt = float(t.item())
    replicas = replicas_in_sync(job, device, world) if world > 1 else None
    if replicas is not None and not replicas["bit_identical_across_ranks"]:
        print(f"[bench] rank {rank}: PARAMETERS DIFFER ACROSS RANKS after the exchange", file=sys.stderr, flush=True)

    def run_appended_legs():
        nonlocal loss
        # N > 1: config 5 only (the one leg whose exchange differs in kind: 0.86 / 1.84 GB through 20 / 36 collectives, the progressive
        # prefix); the 512-wide, preset, VolSDF and config-4 legs repeat config 2's exchange on other bucket plans and each would put
        # another first-time-on-hardware collective sequence (and up to one process-group timeout) behind the line.  SDFHIP_BENCH_ALL_LEGS=1: all.
        all_legs = world == 1 or os.environ.get("SDFHIP_BENCH_ALL_LEGS") == "1"
        cfg5_extra = None
        if not cfg5 and not args.small and not args.no_config5:
            del loss  # the last step's graph (and its 25 GB field workspace) goes back to the allocator
            cfg5_extra = config5_legs(device, world, rank)
        bigmlp_extra = None
        if not cfg5 and not args.small and not args.no_bigmlp and all_legs:
            loss = None
            bigmlp_extra = bigmlp_legs(device, world, rank, dt / args.steps * 1e3)
        preset_extra = None
        if not cfg5 and not args.small and not args.no_preset and all_legs:
            loss = None
            preset_extra = preset_leg(device, world, rank)
        volsdf_extra = None
        if not cfg5 and not args.small and not args.no_volsdf and all_legs:
            loss = None
            volsdf_extra = volsdf_legs(device, world, rank)
        cfg4_extra = None
        if not cfg5 and not args.small and not args.no_config4 and all_legs:
            cfg4_extra = config4_leg(device, world, rank)
        acc_extra = None
        if not cfg5 and not args.small and not args.no_neus_acc and world == 1:
            acc_extra = neus_acc_leg(device)
        exch_extra = None
        if not cfg5 and not args.small and not args.no_exchange_n1 and world == 1 and rank == 0:
            torch.cuda.empty_cache()  # the child builds config 5's model (12 GB) beside this process
            env = dict(os.environ)
            for k in ("SDFHIP_FORCE_EXCHANGE", "SDFHIP_BENCH_EXCHANGE", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
                env.pop(k, None)
            exch_extra = child_leg([os.path.join(ROOT, "bench.py"), "--only", "exchange", "--steps", "8", "--warmup", "3"], timeout=600, env=env)
        mesh_extra = None
        if not cfg5 and not args.small and not args.no_mesh and world == 1 and rank == 0:
            torch.cuda.empty_cache()  # the child allocates its own 3.6 GB beside this process
            mesh_extra = mesh_leg()

        return (cfg5_extra, bigmlp_extra, preset_extra, volsdf_extra, cfg4_extra, acc_extra, exch_extra, mesh_extra)

    def emit_line(cfg5_extra, bigmlp_extra, preset_extra, volsdf_extra, cfg4_extra, acc_extra, exch_extra, mesh_extra, legs_pending=False):
        if rank == 0:
            from sdfstudio_amd import build as _build

            lib_digest = _build.built_digest() or None  # what the loaded libsdfhip.so was built from (sdfstudio_amd/build.py)
            ms = dt / args.steps * 1e3
            samples = world * N_RAYS * N_SAMPLES
            value = samples / (dt / args.steps)
            g, c = flops_per_sample()
            train_flops = 6 * g + 3 * c  # fwd G, analytic-normal chain G, its double backward 2G, backward 2G; colour C + 2C
            if cfg5:  # 1-hidden-layer geometry net on 167 inputs, evaluated 7 x per sample (numerical gradients), no double backward
                g = 2 * (167 * 256 + 256 * 257)
                train_flops = 7 * 3 * g + 3 * c
            P = N_RAYS * N_SAMPLES
            # dominant kernel: geo_bwd_kernel = tangent pass (G) + data backward (G) of the geometry MLP, two launches per step.
            # the MLP kernels (K3) are priced against the MATRIX roofline: algorithmic flops (2G per ray-sample for this
            # kernel) x the 3 split-precision terms actually issued per product, against the dense 16-bit MFMA peak.  The HBM view of
            # the same launches (the kernel streams saved per-layer tensors) is reported beside it under "hbm", not as `frac`.
            kt_ms, kn = prof.get("geo_bwd_kernel", (0.0, 0))
            roof = None
            pm_dir = os.path.join(ROOT, "profiles")
            if kn > 0 and not cfg5:
                avg_s = kt_ms / kn * 1e-3
                flow_bytes = geo_bwd_algorithmic_bytes() * P
                traffic, traffic_source, traffic_digest = None, None, None  # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside the bench)
                cands = sorted(f for f in os.listdir(pm_dir) if f.endswith("_pmc_traffic.json") and "cfg5" not in f)
                if cands:
                    tj = newest_matching_json(pm_dir, cands, lib_digest)  # the pass taken on THIS library, else the newest (r1 < r2 < ...)
                    traffic = tj["hbm_read_bytes"] + tj["hbm_write_bytes"]
                    traffic_source = tj["source"]
                    traffic_digest = tj.get("library_digest")
                flops = 2 * g * P
                io_bytes = geo_bwd_io_bytes() * P
                issued = 3 * flops / avg_s / 1e12
                roof = {"kernel": "geo_bwd_kernel", "bound": "mfma", "achieved": round(issued, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(issued / PEAK_BF16_MFMA_TFLOPS, 4),
                        "achieved_is": "SURVEY 8(d) algorithmic flops of the kernel (2G = 2.098 MFLOP per ray-sample: tangent pass + data backward) x 3 "
                                       "issued 16-bit MFMA terms per fp32-class product / launch time (HIP events on the launch stream)",
                        "algorithmic_tflops": round(flops / avg_s / 1e12, 1), "terms_per_product": 3,
                        "frac_algorithmic_of_16bit_peak": round(flops / avg_s / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                        # the plateau, stated.  `frac` IS the fraction of the ceiling fp32-class products can reach on the
                        # pipe the kernel runs on (dense 16-bit peak / 3 terms = 833 TFLOP/s); the kernel is HBM-bound on its own saved tensors
                        "frac_of_fp32_class_ceiling": round(issued / PEAK_BF16_MFMA_TFLOPS, 4), "fp32_class_ceiling_tflops": round(PEAK_BF16_MFMA_TFLOPS / 3, 1),
                        "ideal_io_bytes": io_bytes, "traffic_over_ideal": None if traffic is None else round(traffic / io_bytes, 1),
                        "plateau": "0.19 - 0.22 of the pipe it runs on; 1.0 - 1.17 x the fp32-matrix peak an exact-fp32 implementation is bound by; "
                                   "HBM-bound on saved per-layer tensors (DESIGN.md section 7: the end state of this data flow)",
                        "frac_algorithmic_of_fp32_matrix_peak": round(flops / avg_s / 1e12 / 157.3, 4),
                        "traffic": traffic, "traffic_unit": "HBM bytes per step of this kernel (sum of its two launches)", "traffic_source": traffic_source,
                        # the PMC passes are a separate rocprofv3 run: they describe THIS library only if it was built from the same sources
                        "traffic_library_digest": traffic_digest, "traffic_stale": traffic is not None and traffic_digest != lib_digest,
                        "avg_launch_ms": round(kt_ms / kn, 4), "launches": kn,
                        # the memory side of the same launches
                        "hbm": {"algorithmic_bytes": io_bytes, "frac_at_algorithmic_bytes": round(io_bytes / avg_s / 1e9 / PEAK_HBM_GBS, 4),
                                "waste_ratio": round((traffic if traffic else flow_bytes) / io_bytes, 1),
                                "dataflow_bytes": flow_bytes, "dataflow_GBps": round(flow_bytes / avg_s / 1e9, 1),
                                "dataflow_frac": round(flow_bytes / avg_s / 1e9 / PEAK_HBM_GBS, 4),
                                "note": "algorithmic_bytes = what must cross the kernel boundary (SURVEY 8d); dataflow_bytes = the saved per-layer "
                                        "tensors this data flow reads and writes (DESIGN.md section 4); measured streaming ceilings of this pool: "
                                        "read 5.5, write 4.0, mixed 5.2 TB/s (profiles/r3_hbm_ceiling.txt)"}}
            # K1 of SURVEY 8(d), the stage north_star asks rocprof HBM GB/s for: the hash-grid gather of geo_encode_kernel
            enc_ms, enc_n = prof.get("geo_encode_kernel", (0.0, 0))
            enc = None
            if enc_n > 0:
                if cfg5:
                    enc = encode_roofline_config5(model, prof, args.steps, P)
                else:
                    per_sample, what = 16 * 8 * 2 * 4 + 12 + 128, "1024 B gather + 12 B position + 128 B of features per ray-sample (SURVEY 8d: 1164 B)"
                    eb = per_sample * P
                    es = enc_ms / args.steps * 1e-3
                    enc = {"kernel": "geo_encode_kernel", "bound": "hbm", "achieved": round(eb / es / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": round(eb / es / 1e9 / PEAK_HBM_GBS, 4), "algorithmic_bytes": eb, "achieved_is": what + " / its time per step",
                           "ms_per_step": round(enc_ms / args.steps, 4), "traffic": None}
                enc.update(encode_step_traffic(lambda f: "_eval" not in f and ("cfg5" in f) == cfg5 and (not cfg5 or ("cfg5l16" in f) == (args.levels == 16))) or {})
                if cfg5:
                    roof = enc
            kernels = {k: {"ms_per_step": round(v[0] / args.steps, 4), "launches_per_step": v[1] / args.steps} for k, v in prof.items()}
            mfma_ms = sum(prof.get(k, (0.0, 0))[0] for k in ("geo_fwd_kernel", "geo_bwd_kernel", "col_fwd_kernel", "col_bwd_kernel",
                                                             "wgrad_kernel")) / args.steps
            line = {
                "library_digest": lib_digest,
                "metric": f"ray-samples/sec (NeuS-facto train step, {N_RAYS} rays x {N_SAMPLES} samples per GPU)",
                "value": round(value, 1), "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "dtype_note": "fp32 tensors and accumulators; matrix products as three 16-bit MFMA terms of hi + lo operand parts: fp16 parts "
                              "(22 mantissa bits, fp32-class) for everything the forward returns, bf16 parts (2^-17 per product, full exponent "
                              "range) in the backward kernels and weight-gradient GEMMs",
                "data": "synthetic", "iters_per_sec": round(1e3 / ms, 3), "per_gpu": round(value / world, 1),
                "config": {"workload": "SMALL parity configuration (control-flow test only, NOT a benchmark)" if args.small else
                                       ("BASELINE config 5: neus-facto-angelo preset - hash grid 16x8x2^22 linear (2.1 GB table), 1x256 geo MLP with "
                                        "numerical SDF gradients (7 evaluations per sample), 4x256 colour MLP, 'grid' background field, progressive "
                                        f"levels ({'all 16 on: steady state, schedules at step 200 000' if args.levels == 16 else 'level_init 8: schedules at step 0'}), "
                                        "curvature loss; 2048 rays x 48 samples (+256/96 proposal samples) per GPU per step, "
                                        "full train step incl. Adam" if cfg5 else
                                        "BASELINE config 2: NeuS-facto hash-grid 16x2x2^19 smoothstep + 8x256 geo MLP + 4x256 colour MLP, "
                                        "4096 rays x 128 samples (+256/96 proposal samples) per GPU per step, full train step incl. Adam"),
                           "rays_per_gpu": N_RAYS, "samples_per_ray": N_SAMPLES,
                           "parallelism": (f"dp{world} (flat gradient buffer; " + ("sharded exchange: reduce-scatter -> owned-slice Adam -> all-gather" if job["shard"] else
                                                                                    "bucketed all-reduce") + ", RCCL)") if world > 1 else "single GPU"},
                "roofline": roof,
                "encode_roofline": enc,
                "config5": cfg5_extra,
                "bigmlp": bigmlp_extra,
                "preset": preset_extra,
                "neus_acc": acc_extra,
                "volsdf": volsdf_extra,
                "config4": cfg4_extra,
                "exchange_at_n1": exch_extra,
                "mesh": mesh_extra,
                "appended_legs": ("run AFTER this line at N > 1 (config5; SDFHIP_BENCH_ALL_LEGS=1: bigmlp / preset / volsdf / config4 too, each with its own exchange): stderr "
                                  "'[bench] appended legs' and gpurun_out/bench_detail.json") if legs_pending else None,
                "collective": None if world == 1 else dict(collective_report(job, dist.get_backend(), exposed_by_rank, exposed_gather_by_rank), replicas=replicas),
                "forward_only": fwd_only,
                "dense_sdf": dense,
                "final_loss": float(final_loss),
                "enqueue_vs_gpu": main_split,
                "model_tflops": round(train_flops * P / (ms * 1e-3) / 1e12, 2),
                "mfma_kernels_ms_per_step": round(mfma_ms, 3),
                "kernels": kernels,
                "kernels_note": f"{dominant}: HIP events inside the timed region; every other entry: a separate untimed pass of {table_steps} "
                                "steps with events on every launch"
                                + ("" if instrumented_ms is None else f", which ran at {instrumented_ms:.3f} ms/step (the events' own cost)"),
            }
            # whole-step view: model FLOPs (6G + 3C per sample) against the fp32 matrix peak an exact-fp32 implementation would be
            # bound by, the issued 16-bit MFMA terms (3 per product in every pass) against the dense bf16 / fp16 peak, and the whole
            # step's HBM bytes from the committed PMC passes
            step_bytes, step_digest = None, None
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_step_traffic.json") and ("cfg5" in f) == cfg5
                           and (not cfg5 or ("cfg5l16" in f) == (args.levels == 16)))
            if cands:
                sj = newest_matching_json(os.path.join(ROOT, "profiles"), cands, lib_digest)
                step_bytes, step_digest = sj.get("hbm_GB_per_training_step"), sj.get("library_digest")
            model_tf = train_flops * P / (ms * 1e-3) / 1e12
            line["step_roofline"] = {
                "model_tflops": round(model_tf, 1), "fp32_matrix_peak_tflops": 157.3, "frac_of_fp32_matrix_peak": round(model_tf / 157.3, 3),
                "issued_16bit_mfma_tflops": round(3 * model_tf, 1), "frac_of_dense_bf16_peak": round(3 * model_tf / PEAK_BF16_MFMA_TFLOPS, 4),
                "hbm_GB_per_step_pmc": step_bytes, "hbm_pmc_stale": step_bytes is not None and step_digest != lib_digest,
                "hbm_GBps": None if step_bytes is None else round(step_bytes / (ms * 1e-3), 1),
                "hbm_frac_of_8TBps": None if step_bytes is None else round(step_bytes / (ms * 1e-3) / PEAK_HBM_GBS, 4),
            }
            ref_path = os.path.join(ROOT, "profiles", "cpu_reference_r2.json")
            if os.path.exists(ref_path):
                with open(ref_path) as fh:
                    line["cpu_baseline_reference"] = json.load(fh)  # the reference's own Python, timed in the build container (no GPU box has it)
            if world == 1 and not args.no_cpu_baseline and not args.small and not cfg5:
                print("[bench] GPU leg done: " + json.dumps(compact_line(line)), file=sys.stderr, flush=True)
                line["cpu_baseline"] = cpu_baseline()
            # the FULL object (per-kernel tables, prose notes, every leg in detail) goes to a file; the ONE line the driver records is its
            # compact form - every figure README / DESIGN quote survives a `tail` of the log (the full line was 23 KB)
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as fh:
                    json.dump(line, fh, indent=1)
            except OSError as e:
                print(f"[bench] could not write gpurun_out/bench_detail.json: {e}", file=sys.stderr)
            print(json.dumps(line if args.full_line else compact_line(line)), flush=True)

    if world == 1:
        emit_line(*run_appended_legs())
    else:
        # N > 1: the ONE line is complete without the appended legs and is printed FIRST - the legs' collectives have never run on N > 1
        # hardware (single-GPU boxes only), and one that hangs into the process group's timeout must not take the headline along.
        # Their results go to stderr and into gpurun_out/bench_detail.json.
        emit_line(*([None] * 8), legs_pending=True)
        names = ("config5", "bigmlp", "preset", "volsdf", "config4", "neus_acc", "exchange_at_n1", "mesh")
        legs = dict(zip(names, run_appended_legs()))
        if rank == 0:
            legs = {k: v for k, v in legs.items() if v is not None}
            print(f"[bench] appended legs at N = {world}: " + json.dumps(legs), file=sys.stderr, flush=True)
            try:
                path = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
                with open(path) as fh:
                    detail = json.load(fh)
                detail.update(legs)
                detail.pop("appended_legs", None)
                with open(path, "w") as fh:
                    json.dump(detail, fh, indent=1)
            except (OSError, ValueError) as e:
                print(f"[bench] could not extend gpurun_out/bench_detail.json: {e}", file=sys.stderr)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
