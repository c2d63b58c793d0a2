"""world_size-2 gloo test of the flat-gradient exchange (the N>1 path, runnable without GPUs)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sdfstudio_amd.distributed import FlatGradients, broadcast_parameters

    torch.manual_seed(rank)  # different init per rank -> broadcast must equalise
    lin = torch.nn.Linear(5, 3)
    extra = torch.nn.Parameter(torch.randn(7))
    mod = torch.nn.ParameterList([lin.weight, lin.bias, extra])
    broadcast_parameters(mod, src=0)
    flat = FlatGradients(list(mod))
    flat.zero()
    x = torch.full((4, 5), float(rank + 1))
    loss = lin(x).sum() + (extra * (rank + 1)).sum()
    loss.backward()  # accumulates in place into the flat views
    assert lin.weight.grad.data_ptr() == flat.flat.data_ptr()
    flat.all_reduce_mean()
    ret[rank] = (lin.weight.detach().clone(), flat.flat.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    w0, g0 = ret[0]
    w1, g1 = ret[1]
    assert torch.equal(w0, w1), "parameters were not broadcast"
    assert torch.equal(g0, g1), "gradients differ after the all-reduce"
    # d/dW of sum(lin(x)) with x = c: 4*c per entry; mean over ranks c=1,2 -> 6 ; extra grad mean -> 1.5
    assert torch.allclose(g0[:15], torch.full((15,), 6.0))
    assert torch.allclose(g0[15:18], torch.full((3,), 4.0))
    assert torch.allclose(g0[18:], torch.full((7,), 1.5))


def _bench_worker(rank, world, port, ret):
    """The parameter set, groups and exchange of bench.py's N > 1 step on the REAL NeuS-facto module tree (built on the CPU: the
    kernels need a GPU, the gradient exchange does not), with synthetic per-rank gradients pushed through autograd so the
    post-accumulate hooks, the buckets and the zero_grad recovery run exactly as in training."""
    import sys

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from sdfstudio_amd.distributed import FlatGradients, broadcast_parameters

    model = bench.build_model(torch.device("cpu"), small=True)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * rank)  # ranks start apart: the broadcast must equalise them
    broadcast_parameters(model)
    groups = {k: v for k, v in model.get_param_groups().items() if v}
    assert set(groups) == {"fields", "proposal_networks"}
    params = [p for g in groups.values() for p in g if p.requires_grad]
    flat = FlatGradients(params, buckets=list(groups.values()))
    launched = []
    orig = flat._launch
    flat._launch = lambda bi: (launched.append(bi), orig(bi))[1]

    def backward(scale):
        loss = sum((p * (scale * (i % 7 + 1))).sum() for i, p in enumerate(params))
        loss.backward()

    flat.zero()
    backward(float(rank + 1))
    assert sorted(set(launched)) == [0, 1], "every bucket must be all-reduced from the autograd hooks, before finish()"
    flat.finish()
    g_first = flat.flat.clone()
    # 2. a stray model.zero_grad() (set_to_none=True) must not leave the flat buffer stale
    model.zero_grad()
    assert params[0].grad is None
    flat.zero()
    backward(float(rank + 1))
    flat.finish()
    assert torch.equal(flat.flat, g_first), "gradients after zero_grad(set_to_none=True) differ"
    assert params[0].grad.data_ptr() == flat.flat.data_ptr()
    # 3. active prefix (progressive hash levels): only the first rows of the table travel; the rest stays local
    table = model.field.encoding.params
    flat.set_active_numel(table, 1000)
    n_all = sum(p.numel() for p in params)
    assert flat.exchanged_numel() == n_all - (table.numel() - 1000)
    flat.zero()
    backward(float(rank + 1))
    flat.finish()
    off = flat._offset[id(table)]
    ret[rank] = (torch.cat([p.detach().reshape(-1)[:50] for p in params[:4]]), g_first, flat.flat[off:off + 2000].clone(), flat.exchanged_numel())
    dist.barrier()
    dist.destroy_process_group()


def test_bench_parameter_groups_bucketed_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bench_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    p0, g0, t0, n0 = ret[0]
    p1, g1, t1, n1 = ret[1]
    assert torch.equal(p0, p1), "parameters were not broadcast"
    assert torch.equal(g0, g1), "gradients differ after the bucketed all-reduce"
    # d/dp of sum(p * s * c_i) = s c_i with s = 1, 2 on the two ranks -> mean 1.5 c_i; parameter 0 has c = 1
    assert torch.allclose(g0[:10], torch.full((10,), 1.5))
    # active prefix: the first 1000 table entries are means (equal on both ranks), the tail kept the local gradient
    assert torch.equal(t0[:1000], t1[:1000]) and not torch.equal(t0[1000:], t1[1000:])
    assert n0 == n1


def _order_worker(rank, world, port, ret):
    """the collective sequence must be the same on every rank even when the autograd graphs differ.  Rank 1 leaves the
    whole first bucket unused; rank 0 uses everything.  Buckets must leave in index order on both ranks (bucket 1 may not go out
    from a hook while bucket 0 is still pending), and the unused bucket contributes its zeros."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sdfstudio_amd.distributed import FlatGradients

    a = [torch.nn.Parameter(torch.ones(6)), torch.nn.Parameter(torch.ones(3))]
    b = [torch.nn.Parameter(torch.ones(4))]
    flat = FlatGradients(a + b, buckets=[a, b], chunk_numel=4)  # bucket 0 (9 elements) travels as 3 chunks
    order = []
    orig = flat._launch
    flat._launch = lambda bi: (order.append(bi), orig(bi))[1]
    for step in range(2):
        flat.zero()
        loss = (b[0] * 2.0).sum()
        if rank == 0:
            loss = loss + (a[0] * 3.0).sum() + (a[1] * 5.0).sum()
        loss.backward()
        assert flat.last_overlapped_buckets == (2 if rank == 0 else 0), "bucket 1 must wait for bucket 0 on the rank that never completes it"
        flat.finish()
        assert order[-2:] == [0, 1]
    # protocol violations raise instead of corrupting the buffer
    errs = []
    try:
        (b[0] * 1.0).sum().backward()  # backward without zero() after finish()
    except RuntimeError as e:
        errs.append("nozero" if "without zero" in str(e) else str(e))
    flat.zero()
    (b[0] * 1.0).sum().backward()
    try:
        (b[0] * 1.0).sum().backward()  # second backward before finish()
    except RuntimeError as e:
        errs.append("second" if "second backward" in str(e) else str(e))
    flat.finish()
    ret[rank] = (flat.flat.clone(), errs, flat.last_collectives)
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_order_is_fixed_when_one_rank_skips_a_bucket_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_order_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    (g0, e0, n0), (g1, e1, n1) = ret[0], ret[1]
    assert e0 == ["nozero", "second"] and e1 == ["nozero", "second"]
    assert torch.equal(g0, g1) and n0 == n1 == 3 + 1
    # last step: only b[0] was differentiated on both ranks (twice: the refused second backward had already accumulated when the
    # post-accumulate hook raised - the exception is the signal); a's gradients are the zeros zero() left
    assert torch.equal(g0, torch.cat([torch.zeros(9), torch.full((4,), 2.0)]))


def test_zero_discards_stray_gradients_and_finish_rearms():
    """zero() after a set_to_none zero_grad + a backward used to copy the stray gradient back into the freshly zeroed
    buffer (flat = 5 5 5 5, next step's gradient 7 instead of 2)."""
    from sdfstudio_amd.distributed import FlatGradients

    p = torch.nn.Parameter(torch.ones(4))
    flat = FlatGradients([p])
    flat.zero()
    (p * 5.0).sum().backward()
    flat.finish()
    p.grad = None
    flat.zero()
    (p * 5.0).sum().backward()   # hook folds the fresh gradient into the view
    flat.finish()
    assert torch.equal(flat.flat, torch.full((4,), 5.0)) and p.grad.data_ptr() == flat.flat.data_ptr()
    p.grad = torch.full((4,), 123.0)  # a stray tensor parked in .grad between steps is discarded, not folded in
    flat.zero()
    assert torch.equal(flat.flat, torch.zeros(4)) and p.grad is None  # zero() leaves .grad None: the first gradient is adopted, not added
    (p * 2.0).sum().backward()
    assert p.grad.data_ptr() == flat.flat.data_ptr()
    flat.finish()
    assert torch.equal(flat.flat, torch.full((4,), 2.0))
    try:
        flat.finish()
        raise AssertionError("finish() twice must raise")
    except RuntimeError:
        pass


def test_gradient_slots_are_adopted_without_an_add():
    """grad_slots.py: a native backward writes a parameter's gradient into a view of the flat buffer and returns it; with .grad None
    autograd adopts it (no accumulate kernel) and .grad aliases the flat buffer; a second producer in the same backward gets an
    ordinary tensor, which autograd adds.  Emulated on CPU with a torch.autograd.Function that uses grad_target like the kernels do."""
    from sdfstudio_amd.distributed import FlatGradients
    from sdfstudio_amd.grad_slots import grad_target

    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, p, k):
            ctx.p, ctx.k = p, k
            return (p.detach() * k).sum()

        @staticmethod
        def backward(ctx, g):
            out, is_slot = grad_target(ctx.p)
            ctx.p._test_slots = getattr(ctx.p, "_test_slots", []) + [is_slot]
            out.copy_(torch.full_like(out, ctx.k) * g)
            return out, None

    p = torch.nn.Parameter(torch.ones(2, 3))
    q = torch.nn.Parameter(torch.ones(5))
    flat = FlatGradients([p, q])
    for step in range(2):
        flat.zero()
        p._test_slots = []
        (Producer.apply(p, 2.0) + Producer.apply(p, 3.0) + (q * 4.0).sum()).backward()
        assert sorted(p._test_slots) == [False, True], "exactly one producer per backward gets the slot"
        assert p.grad.data_ptr() == flat.flat.data_ptr() and q.grad.data_ptr() == flat.flat.data_ptr() + 4 * 6
        flat.finish()
        assert torch.equal(flat.flat, torch.cat([torch.full((6,), 5.0), torch.full((5,), 4.0)]))


def test_live_ranges_skip_a_never_active_suffix():
    """Progressive hash levels (BASELINE config 5): the table rows of levels that have not been switched on never carry a gradient.
    live_ranges() - what zero() rewrites and the fused Adam step visits - leaves that suffix out while it has never been active, grows
    with it, and covers everything when the restriction arrives after unrestricted steps or after a checkpoint load."""
    from sdfstudio_amd.distributed import FlatGradients

    a, table, c = (torch.nn.Parameter(torch.randn(n)) for n in (6, 40, 5))
    flat = FlatGradients([a, table, c])
    assert flat.live_ranges() == [(0, 51)]
    flat.set_active_numel(table, 16)
    assert flat.live_ranges() == [(0, 22), (46, 51)]
    flat.flat[22:46] = 7.0  # sentinel: zero() must not touch the never-active suffix
    flat.flat[:22] = 3.0
    flat.zero()
    assert float(flat.flat[:22].abs().max()) == 0.0 and torch.equal(flat.flat[22:46], torch.full((24,), 7.0))
    flat.flat[22:46] = 0.0
    (a.sum() + (table[:16] * 2).sum() + c.sum()).backward()
    flat.finish()
    assert torch.equal(flat.flat[6:22], torch.full((16,), 2.0)) and float(flat.flat[22:46].abs().max()) == 0.0
    flat.set_active_numel(table, 24)  # a level is switched on: the range grows, never shrinks
    assert flat.live_ranges() == [(0, 30), (46, 51)]
    flat.set_active_numel(table, 8)
    assert flat.live_ranges() == [(0, 30), (46, 51)] and flat._ranges(0) == [(0, 14), (46, 51)]  # the exchange follows the current prefix
    flat.mark_all_live()
    assert flat.live_ranges() == [(0, 51)]
    # a restriction that arrives after unrestricted steps: gradients (and moments) exist beyond it, nothing is skipped
    a2, table2, c2 = (torch.nn.Parameter(torch.randn(n)) for n in (6, 40, 5))
    flat2 = FlatGradients([a2, table2, c2])
    flat2.zero()
    (a2.sum() + table2.sum() + c2.sum()).backward()
    flat2.finish()
    flat2.set_active_numel(table2, 16)
    assert flat2.live_ranges() == [(0, 51)]


def _real_model_overlap_worker(rank, world, port, ret):
    """on the REAL NeuS-facto parameter set the exchange must leave from the autograd hooks.  Which parameters a
    NeuS-facto step's graph reaches is taken from the reference's own run - tests/golden/neus_facto_small_train.npz holds a gradient for
    exactly those (no laplace_density.beta, no appearance embedding) - and the loss built here touches exactly them, so the hooks, the
    graph walk of zero(loss) and the buckets run as in training (the kernels need a GPU, the exchange does not)."""
    import sys

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d in (root, os.path.join(root, "tests")):
        if d not in sys.path:
            sys.path.insert(0, d)
    import bench
    from helpers import load_golden
    from sdfstudio_amd.distributed import FlatGradients, broadcast_parameters

    model = bench.build_model(torch.device("cpu"), small=True)
    broadcast_parameters(model)
    groups = {k: v for k, v in model.get_param_groups().items() if v}
    params = [p for g in groups.values() for p in g if p.requires_grad]
    flat = FlatGradients(params, buckets=list(groups.values()))
    touched = set(load_golden("train")["grad"])  # oracle / reference names: field.* without the prefix, proposal_networks.<i>.<name>
    named = {}
    for k, p in model.named_parameters():
        if k.startswith("field."):
            named[k[len("field."):]] = p
        elif k.startswith("proposal_networks."):
            named[f"proposal_networks.{k.split('.')[1]}.{k.split('.')[-1]}"] = p
    in_graph = [named[k] for k in touched]
    outside = [k for k, p in named.items() if p.requires_grad and k not in touched]
    assert "laplace_density.beta" in outside and len(in_graph) >= 40
    res = {}
    for mode in ("graph walk", "conservative"):
        loss = sum((p * float(rank + 1)).sum() for p in in_graph)
        flat.zero(loss if mode == "graph walk" else None)
        loss.backward()
        res[mode] = (flat.last_overlapped_buckets, flat.last_unused)
        flat.finish()
    beta = model.field.laplace_density.beta
    off = flat._offset[id(beta)]
    ret[rank] = (res, flat.flat.clone(), float(flat.flat[off]), sorted(outside))
    # a gradient for a parameter zero(loss) was told nothing about is a protocol violation
    flat.zero(sum(p.sum() for p in in_graph))
    try:
        (beta * 2.0).sum().backward()
        ret[f"err{rank}"] = "no error"
    except RuntimeError as e:
        ret[f"err{rank}"] = "refused" if "not in the graph" in str(e) else str(e)
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_overlaps_with_backward_on_the_real_model_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_real_model_overlap_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    (res0, g0, beta0, out0), (res1, g1, beta1, out1) = ret[0], ret[1]
    assert out0 == out1 and "laplace_density.beta" in out0
    for res in (res0, res1):
        # with the graph walk both buckets ("fields", "proposal_networks") leave from the hooks; counted conservatively the "fields"
        # bucket waits for laplace_density.beta forever and - fixed order - takes the proposal networks' bucket with it
        assert res["graph walk"] == (2, len(out0)), res
        assert res["conservative"][0] == 0, res
    assert torch.equal(g0, g1) and beta0 == 0.0 and beta1 == 0.0
    assert ret["err0"] == "refused" and ret["err1"] == "refused"


def test_mark_all_live_before_any_restriction_is_sticky():
    """FusedAdam.load_state_dict calls mark_all_live() right after construction, before any set_active_numel; the first
    restriction afterwards must not make zero() / Adam skip rows whose loaded moments may be non-zero."""
    from sdfstudio_amd.distributed import FlatGradients

    a, table = torch.nn.Parameter(torch.randn(6)), torch.nn.Parameter(torch.randn(40))
    flat = FlatGradients([a, table])
    flat.mark_all_live()
    flat.set_active_numel(table, 16)
    assert flat.live_ranges() == [(0, 46)] and flat._ranges(0) == [(0, 22)]  # everything is visited, the active prefix is exchanged
    # track_active: the restriction is re-read from the model at every zero()
    b, t2 = torch.nn.Parameter(torch.randn(6)), torch.nn.Parameter(torch.randn(40))
    flat2 = FlatGradients([b, t2])
    level = {"n": 16}
    flat2.track_active(t2, lambda: level["n"])
    flat2.zero()
    assert flat2.live_ranges() == [(0, 22)]
    (b.sum() + (t2[:16] * 2).sum()).backward()
    flat2.finish()
    level["n"] = 24  # a level is switched on: nobody has to remember to widen the restriction
    flat2.zero()
    assert flat2.live_ranges() == [(0, 30)] and flat2._ranges(0) == [(0, 30)]


# ------------------------------------------------------------------------------------------------ sharded exchange (round 5)
def _oracle_adam_step_slice(adam):
    """FusedAdam._step_slice on CPU tensors by the oracle's formula (oracle.sdf_path.adam_reference): the checker standing in for the
    native kernel, which needs a GPU.  The PROTOCOL under test (what travels, who owns what, what is visited) is the product's."""
    from oracle import sdf_path as O

    def step_slice(a, b, loc, lr, grad_scale):
        n = b - a
        P, G = adam.flat_params.flat, adam.flat_grads.flat
        with torch.no_grad():  # in place on views of the flat buffers, as the kernel does
            O.adam_reference(P[a:b], G[a:b], adam.exp_avg[loc:loc + n], adam.exp_avg_sq[loc:loc + n], lr, adam.betas[0], adam.betas[1], adam.eps,
                             adam.step_count, grad_scale=grad_scale)

    return step_slice


def _small_angelo_model():
    """A neus-facto-angelo-shaped model small enough for CPU processes: 8 hash levels of 8 features with a 2^10-entry table, progressive
    levels from level_init = 4, one level more every 2 steps (so that three training steps cross a level switch)."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import NeuSFactoModel, NeuSFactoModelConfig, SceneBox

    torch.manual_seed(0)
    fcfg = SDFFieldConfig(use_grid_feature=True, num_layers=1, num_layers_color=2, hidden_dim=64, hidden_dim_color=64, geo_feat_dim=64,
                          geometric_init=True, bias=0.5, beta_init=0.3, inside_outside=False, use_appearance_embedding=False,
                          use_numerical_gradients=True, num_levels=8, base_res=4, max_res=64, log2_hashmap_size=10, hash_features_per_level=8,
                          hash_smoothstep=False, use_position_encoding=False)
    mcfg = NeuSFactoModelConfig(sdf_field=fcfg, background_model="none", level_init=4, steps_per_level=2, enable_progressive_hash_encoding=True,
                                num_proposal_samples_per_ray=(32, 24), num_neus_samples_per_ray=16)
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
    return NeuSFactoModel(mcfg, box, num_train_data=4).train()


def _shard_worker(rank, world, port, ret, shard):
    import sys

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d in (root, os.path.join(root, "tests")):
        if d not in sys.path:
            sys.path.insert(0, d)
    from sdfstudio_amd.distributed import FlatGradients, broadcast_parameters, plan_buckets
    from sdfstudio_amd.engine.optimizers import Optimizers, multi_step_scheduler

    model = _small_angelo_model()
    broadcast_parameters(model)
    groups = {k: v for k, v in model.get_param_groups().items() if v}
    table = model.field.encoding.params
    params, buckets, late = plan_buckets(groups, big_numel=table.numel())  # the SDF table and the two proposal tables get buckets of their own
    assert params[0] is table and late == [0] and len(buckets) == 5
    flat = FlatGradients(params, buckets=buckets, shard=shard, late_buckets=late if shard else (), chunk_numel=1 << 12)
    flat.track_active(table, model.active_table_floats)
    opts = Optimizers({"fields": {"lr": 1e-3, "scheduler": None}, "proposal_networks": {"lr": 1e-2, "scheduler": multi_step_scheduler(4)}},
                      groups, flat_grads=flat)
    opts.adam._step_slice = _oracle_adam_step_slice(opts.adam)
    named = dict(model.named_parameters())
    unused = {id(named["field.laplace_density.beta"])}
    visited, exchanged, gathered, levels = [], [], [], []
    gen = torch.Generator().manual_seed(100 + rank)
    for step in range(6, 10):
        model.before_train_iteration(step)  # progressive levels (int(step / 2) + 1, at least level_init): 4, 4, 5 (the switch), 5
        opts.wait_parameters()
        levels.append(int(model.field._active_levels))
        n_act = model.active_table_floats()
        loss = 0.0
        for p in params:
            if id(p) in unused:
                continue
            c = torch.rand(p.shape, generator=gen) + 0.5  # a different gradient on every rank and step
            if p is table:
                c = c.clone()
                c.view(-1)[n_act:] = 0.0  # the masked levels' rows get exactly zero gradient (sdf_field.py:376-378)
            loss = loss + (p * c).sum()
        flat.zero(loss)
        loss.backward()
        opts.optimizer_step_all(grad_scale=None)
        opts.scheduler_step_all(step)
        visited.append(opts.adam.last_elements_visited)
        exchanged.append(flat.exchanged_numel())
        gathered.append(flat.gathered_bytes())
    opts.wait_parameters()
    if shard:
        try:  # state_dict() is LOCAL - it must refuse rather than start a collective the reference's rank-0-only checkpoint path would hang in
            opts.state_dict()
            raise AssertionError("sharded state_dict() without gather_moments() must raise")
        except RuntimeError as e:
            assert "gather_moments" in str(e)
    opts.gather_moments()  # the collective half (every rank); a no-op under the all-reduce exchange
    sd = opts.state_dict()  # local: no collective inside
    full_m = torch.zeros(flat.flat.numel())
    for name, g in opts.adam.groups.items():
        full_m[g["start"]:g["end"]] = sd["groups"][name]["exp_avg"]
    ret[rank] = {"params": torch.cat([p.detach().reshape(-1) for p in params]), "visited": visited, "exchanged": exchanged, "levels": levels,
                 "moments_local": int(opts.adam.exp_avg.numel()), "total": int(flat.flat.numel()), "n_params": sum(p.numel() for p in params),
                 "exp_avg": full_m, "gathered": gathered,
                 "collectives": flat.last_collectives, "gather_collectives": flat.last_gather_collectives,
                 "offsets": [flat._offset[id(p)] for p in params], "numels": [p.numel() for p in params]}
    dist.barrier()
    dist.destroy_process_group()


def _run_shard(world, shard):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, _free_port(), ret, shard), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _unpadded(res, key):
    """a per-layout vector (moments over the padded sharded layout) -> one value per parameter element, in parameter order"""
    return torch.cat([res[key][o:o + n] for o, n in zip(res["offsets"], res["numels"])])


def _check_sharded_against_allreduce(world):
    ar = _run_shard(world, shard=False)
    sh = _run_shard(world, shard=True)
    assert ar[0]["levels"] == [4, 4, 5, 5] == sh[0]["levels"], "the three steps must cross a level switch"
    for r in range(world):
        # every rank of either protocol holds the same parameters, bit for bit: x + 0 and the order of gloo's sums are the same
        assert torch.equal(ar[r]["params"], ar[0]["params"]) and torch.equal(sh[r]["params"], sh[0]["params"])
        if world == 2:  # a sum of two is commutative: whatever segments gloo cuts a collective into, the bits agree
            assert torch.equal(sh[r]["params"], ar[r]["params"]), f"world {world}: sharded parameters differ from the all-reduce path's on rank {r}"
        else:
            # W > 2: gloo's ring sums a segment's W contributions in an order that depends on where the segment lies in ITS collective, and
            # the two protocols cut the buffer differently (active ranges vs the fixed grid): the sums differ in the last bit, Adam's
            # normalised update carries that ulp through.  Bit equality ACROSS RANKS (above) is the protocol's own guarantee; against the
            # all-reduce path the bar is a few ulps.
            torch.testing.assert_close(sh[r]["params"], ar[r]["params"], rtol=2e-6, atol=1e-9)
    n_params, total = sh[0]["n_params"], sh[0]["total"]
    assert total % (64 * world) == 0 and total - n_params < 5 * 64 * world  # five buckets, each padded to the quantum
    for r in range(world):
        assert sh[r]["moments_local"] * world == total, "moments exist for the owned 1 / W of the padded buffer only"
        assert ar[r]["moments_local"] == ar[r]["total"] == n_params
    # Adam visits ~1 / W of what the all-reduce path's replica visits, summed over the ranks everything exactly once
    for step in range(4):
        full = ar[0]["visited"][step]
        per_rank = [sh[r]["visited"][step] for r in range(world)]
        assert sum(per_rank) == full, (per_rank, full)
        assert max(per_rank) <= full // world + 64 * 5 * 2 and all(a["visited"][step] == full for a in ar)
    # the never-active suffix of the table is skipped by both; the level switch widens what travels
    assert ar[0]["exchanged"][2] > ar[0]["exchanged"][1] and sh[0]["exchanged"][2] > sh[0]["exchanged"][1]
    assert sh[0]["exchanged"][0] >= ar[0]["exchanged"][0]  # whole grid chunks travel
    # the assembled moments equal the replica's moments (checkpoints are world-size independent in content)
    if world == 2:
        assert torch.equal(_unpadded(sh[0], "exp_avg"), _unpadded(ar[0], "exp_avg"))
    else:
        torch.testing.assert_close(_unpadded(sh[0], "exp_avg"), _unpadded(ar[0], "exp_avg"), rtol=2e-6, atol=1e-12)
    assert sh[0]["gather_collectives"] > 0 and ar[0]["gather_collectives"] == 0


def test_sharded_exchange_matches_allreduce_world2():
    """reduce-scatter -> owned-slice Adam -> all-gather (distributed.FlatGradients(shard=True)) on the small angelo model
    across a level switch: parameters bit-identical to the all-reduce path's after every protocol step, moments for the owned slice only,
    1 / W of the Adam work per rank."""
    _check_sharded_against_allreduce(2)


def test_sharded_exchange_matches_allreduce_world4():
    _check_sharded_against_allreduce(4)


class _FieldLikeNode(torch.autograd.Function):
    """Stands in for the SDF field's single autograd node: its backward writes the table's gradient into the gradient SLOT (grad_slots.py),
    tells the exchange "the table's gradient is in the queue" the way the native backward does through sdfhip_set_table_grad_callback,
    and only then produces the weight gradient."""

    @staticmethod
    def forward(ctx, table, weight, flat, scale):
        ctx.flat, ctx.scale, ctx.table, ctx.weight = flat, scale, table, weight
        return (table.sum() + weight.sum()) * scale

    @staticmethod
    def backward(ctx, g):
        from sdfstudio_amd.grad_slots import grad_target

        tb, is_slot = grad_target(ctx.table, zero_init=True)
        tb.add_(float(g) * ctx.scale)
        if is_slot:  # what api.hip's notify_table_grad hands to the registered callback: the pointer it wrote to (and a stream)
            ctx.flat._native_ready(tb.data_ptr(), 0)
        wb, _ = grad_target(ctx.weight, zero_init=True)
        wb.add_(float(g) * ctx.scale)
        return tb, wb, None, None


def _early_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sdfstudio_amd.distributed import FlatGradients

    table = torch.nn.Parameter(torch.zeros(4096))
    weight = torch.nn.Parameter(torch.zeros(300))
    other = torch.nn.Parameter(torch.zeros(50))
    res = {}
    for early in (True, False):
        flat = FlatGradients([table, weight, other], buckets=[[table], [weight], [other]], shard=True, late_buckets=[0], chunk_numel=1024)
        if early:
            flat._early[id(table)] = 0  # launch_from_native without registering the C callback (the node above calls _native_ready itself)
        order = []
        orig = flat._launch
        flat._launch = lambda bi, _o=orig: (order.append(bi), _o(bi))[1]
        # step 1: one producer -> the table's bucket leaves from inside the node's backward
        loss = _FieldLikeNode.apply(table, weight, flat, float(rank + 1)) + (other * 3.0).sum()
        flat.zero(loss)
        loss.backward()
        first = (flat.last_early_buckets, list(order), flat.last_overlapped_buckets)
        flat.finish()
        g1 = flat.flat.clone()
        # step 2: TWO producers of the table's gradient in the graph -> no early launch (the first producer's slot is not the whole gradient)
        order.clear()
        loss = _FieldLikeNode.apply(table, weight, flat, float(rank + 1)) + (table * 2.0).sum() + (other * 3.0).sum()
        flat.zero(loss)
        loss.backward()
        second = (flat.last_early_buckets, list(order))
        flat.finish()
        res[early] = (first, second, g1, flat.flat.clone())
        flat._early = {}  # (nothing was registered with the library)
        flat.close()
        for p in (table, weight, other):
            p.grad = None
    ret[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_table_bucket_leaves_from_inside_the_native_backward_world2():
    """"the table's chunks are launched from the grid_bwd8 completion event inside sdfhip_numfield_backward, before the
    weight-gradient GEMMs".  Protocol side (the native side is the callback of include/sdfhip.h, exercised on the GPU by
    tests/test_gpu_bench_multirank.py): with ONE producer of the table's gradient its bucket leaves from inside the node's backward, first
    and before the weight gradient exists; with two producers it waits for the hook; the reduced gradients are the same either way."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_early_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        (f_e, s_e, g1_e, g2_e), (f_n, s_n, g1_n, g2_n) = ret[r][True], ret[r][False]
        assert f_e[0] == 1 and f_e[1][0] == 0 and sorted(f_e[1]) == [0, 1, 2], f_e
        assert f_n[0] == 0 and sorted(f_n[1]) == [0, 1, 2], f_n
        assert s_e[0] == 0 and s_n[0] == 0, (s_e, s_n)
        assert torch.equal(g1_e, g1_n) and torch.equal(g2_e, g2_n)
        # sums over the two ranks: table / weight 1 + 2 = 3, other 3 + 3 = 6 (finish() averages: / 2)
        assert torch.allclose(g1_e[:4096], torch.full((4096,), 1.5)) and torch.allclose(g2_e[:4096], torch.full((4096,), 3.5))


def test_forced_single_rank_exchange_protocol_over_gloo():
    """SDFHIP_FORCE_EXCHANGE=1: a ONE-rank process group runs the whole N > 1 protocol - bucket launches from the hooks,
    chunk-wise waits, owned-slice Adam, parameter gathers - and must leave the same bits as no exchange.  This is the gloo twin of
    tests/test_gpu_rccl_single_rank.py::test_rccl_protocol_bit_identical_to_no_exchange (same worker, CPU tensors)."""
    import json
    import subprocess
    import sys

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SDFHIP_FORCE_EXCHANGE", "SDFHIP_BENCH_EXCHANGE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_worker.py"), "protocol_cpu"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rep["backend"] == "gloo"
    assert rep["params_equal"] == {"shard": True, "allreduce": True} and rep["moments_equal"] == {"shard": True, "allreduce": True}, rep
    assert rep["table_moved"] > 0.99 and rep["table_tail_untouched"]
    assert all(c == [0, 0] for c in rep["collectives"]["none"])
    sh = rep["collectives"]["shard"]
    assert sh[0][0] == 4 + 2 and sh[-1][0] == 8 + 2 and sh[-1][1] > sh[0][1] > 0, sh
