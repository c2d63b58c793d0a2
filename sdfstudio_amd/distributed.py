"""Data-parallel gradient exchange: one process per GPU, one flat fp32 buffer, bucketed RCCL all-reduces overlapped with backward.

The reference wraps the model in torch DDP with find_unused_parameters=True (pipelines/base_pipeline.py:241-243):
bucketed NCCL all-reduce of ~58 MB of gradients plus a per-step graph walk.  Every rank here owns a full replica and
its own rays (scripts/train.py:86); parameter gradients are VIEWS into a single contiguous buffer, so a bucket's
exchange is an in-place all-reduce (mean) over xGMI of one slice with no packing copies.  A bucket (one per parameter
group by default: "fields", "proposal_networks") is reduced as soon as autograd has accumulated the last of its
gradients (post-accumulate-grad hooks), i.e. the field's 58 MB travel while the proposal networks' backward still
runs; ``finish()`` waits for the outstanding collectives before the optimiser step.

Collective order.  Every rank issues the SAME sequence of all-reduces over the SAME slices: buckets are launched strictly in
index order (bucket k leaves from a hook only once buckets 0..k-1 have left; otherwise it waits for ``finish()``, which launches
what is left, again in index order), and a bucket larger than ``chunk_numel`` elements travels as fixed-size chunks in address
order.  A rank whose autograd graph misses a whole parameter group in some step (the zero-sample branch of NeuS-acc, a background
field no ray hits) therefore still matches its peers, as DDP's fixed bucket order does; its untouched gradients are the zeros
``zero()`` left.

``set_active_numel(param, n)`` restricts a parameter's exchange to its first ``n`` elements: with the progressive
level mask of neus-facto-angelo (sdf_field.py:376-378) the hash-table rows of the masked levels have exactly zero
gradient on every rank, so BASELINE config 5's 1.8 GB table moves only its active prefix.

Unused parameters.  A bucket leaves from a hook when the LAST of its gradients has arrived, so it has to know how many will arrive.
Not every parameter of a group is in every step's autograd graph: ``laplace_density.beta`` never is under the NeuS family (no Laplace
density on the path to the loss), ``deviation_network.variance`` never is under VolSDF, the appearance embedding is not when it is
switched off, a background field is not in a step whose rays stay inside the unit sphere.  Their hooks never fire; counted as pending
they keep their bucket - and, with the fixed order, every bucket behind it - from leaving before ``finish()``: no overlap at all, on
every model of the family.  ``zero(loss)`` therefore walks the graph below ``loss`` once per step, exactly what DDP does
with ``find_unused_parameters=True`` (pipelines/base_pipeline.py:242), and arms every bucket with the parameters that are REACHABLE;
the unreachable ones contribute the zeros ``zero()`` left.  (The walk visits ~10^2 nodes: the whole field is one autograd node.)

Sharded mode (``shard=True``, round 5; SURVEY 8(e): "reduce-scatter + all-gather over all 7 links"): the exchange sized for BASELINE
config 5, whose 1.8 GB table cannot hide an all-reduce behind a 10 ms step.  The flat buffer is cut into a FIXED grid of chunks (per
bucket, anchored at the bucket's start, every chunk a multiple of the world size W); rank r OWNS elements [r n / W, (r + 1) n / W) of
every chunk of n elements.  A chunk's gradients are reduce-scattered (RCCL, in place: each rank receives the sum of its own slice), the
fused Adam step runs on the owned slices only - moments exist for 1 / W of the parameters - and the updated slices are all-gathered
into every rank's parameter buffer (``gather_parameters`` / ``wait_parameters``), which may overlap the start of the next step.  Same
bytes over xGMI as the all-reduce (which IS a reduce-scatter followed by an all-gather), but the optimiser's HBM traffic and state
divide by W and the second half of the exchange moves off the critical path.  The ownership grid never depends on the active prefix:
a level that is switched on later finds its moments where they will always be.  Backends without reduce-scatter (gloo: the CPU tests
and the single-GPU control-flow runs) emulate it with an all-reduce of the chunk - the owned slice then holds the same sum.

Protocol per step: ``zero(loss)`` -> one backward of that loss -> ``finish()``.  Anything else raises: a second backward before
``finish()`` would accumulate into slices that are being reduced, a backward without ``zero()`` would mix last step's means with new
local sums.  (``zero()`` without the loss keeps the conservative count - every parameter pending - which is correct and overlaps only
when every parameter of a bucket receives a gradient.)
"""
import os
import time
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

from sdfstudio_amd.grad_slots import CLAIM_ATTR, SLOT_ATTR  # noqa: F401


_DEBUG_LIVE = os.environ.get("SDFHIP_DEBUG_GRADS") == "1"


def force_single_rank_exchange() -> bool:
    """SDFHIP_FORCE_EXCHANGE=1: a process group of ONE rank exchanges as if it had peers - every bucket's collective is issued (RCCL's
    reduce_scatter / all_gather / all_reduce kernels run on their own stream, the waits are real stream waits, the native table-gradient
    callback launches its bucket) and the result is the input.  The one way to execute the N > 1 path on the single MI355X a
    `gpurun` box has (tests/test_gpu_rccl_single_rank.py, bench.py's `exchange_at_n1`); never set in production."""
    return os.environ.get("SDFHIP_FORCE_EXCHANGE") == "1"


def _dist_on(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_single_rank_exchange())


class FlatGradients:
    def __init__(self, params: Iterable[torch.nn.Parameter], buckets: Optional[Sequence[Sequence[torch.nn.Parameter]]] = None,
                 group=None, overlap: bool = True, chunk_numel: Optional[int] = 32 * 1024 * 1024, shard: bool = False,
                 late_buckets: Sequence[int] = ()):
        """params: every parameter whose gradient lives in the flat buffer, in buffer order.  buckets: a partition of them
        into exchange units (default: one bucket); each bucket's parameters must be contiguous in `params`.  chunk_numel: a
        bucket's ranges travel in pieces of at most this many elements (128 MB by default: config 2's 50 MB field bucket is one
        collective, config 5's 1.8 GB table is 14, so the ring starts delivering finished pieces while later ones are in
        flight and no single collective monopolises the links); None: one collective per contiguous range.
        shard: reduce-scatter / owned-slice optimiser / all-gather instead of all-reduce (module docstring); the buckets must then be
        listed in buffer order, and every bucket is padded to a multiple of 64 W elements (zeros: no parameter lives there).
        late_buckets (sharded mode): indices of buckets whose parameters the next step needs LAST (the big hash table: after the proposal
        sampling): their all-gathers are issued after everybody else's and wait_parameters(late=False) leaves them in flight."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.shard = bool(shard)
        self.world = dist.get_world_size(group) if _dist_on(group) else 1
        self.rank = dist.get_rank(group) if _dist_on(group) else 0
        self.exchanging = _dist_on(group)  # collectives are issued (world > 1, or a single rank under SDFHIP_FORCE_EXCHANGE=1)
        if buckets is None:
            buckets = [self.params]
        self._buckets: List[List[torch.nn.Parameter]] = [[p for p in b if p.requires_grad] for b in buckets]
        self._buckets = [b for b in self._buckets if b]
        assert sorted(id(p) for b in self._buckets for p in b) == sorted(id(p) for p in self.params), "buckets must partition the parameters"
        if self.shard:
            assert [id(p) for b in self._buckets for p in b] == [id(p) for p in self.params], \
                "shard=True: list the buckets in buffer order (the ownership grid is anchored at every bucket's start)"
        # layout: parameters back to back; in sharded mode every bucket ends on a multiple of the quantum (64 W elements: every owned
        # slice is a whole number of 256-byte lines) so that each chunk of the grid divides evenly among the ranks
        self._quantum = 64 * self.world if self.shard else 1
        self._offset: Dict[int, int] = {}
        self._span: List[List[int]] = [[0, 0] for _ in self._buckets]  # [start, end) of every bucket incl. its padding (sharded mode)
        if self.shard:
            off = 0
            for bi, b in enumerate(self._buckets):
                self._span[bi][0] = off
                for p in b:
                    self._offset[id(p)] = off
                    off += p.numel()
                off = (off + self._quantum - 1) // self._quantum * self._quantum
                self._span[bi][1] = off
            total = off
        else:
            off = 0
            for p in self.params:
                self._offset[id(p)] = off
                off += p.numel()
            total = off
            for bi, b in enumerate(self._buckets):
                self._span[bi] = [self._offset[id(b[0])], self._offset[id(b[-1])] + b[-1].numel()]
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._active: Dict[int, int] = {}
        self._active_fn = {}  # id(param) -> (param, callable): the active size is re-read at every zero() (track_active)
        self._all_live = False  # sticky: set by mark_all_live() (optimizer state loaded): never skip a suffix again
        # high-water mark of set_active_numel per parameter: elements beyond it have NEVER carried a gradient, so they are still the
        # zeros the buffer was created with (zero() need not rewrite them, Adam need not visit them: live_ranges())
        self._hwm: Dict[int, int] = {}
        self._finished_steps = 0
        self._attach()
        for b in self._buckets:
            offs = [self._offset[id(p)] for p in b]
            assert offs == sorted(offs) and all(offs[i] + b[i].numel() == offs[i + 1] for i in range(len(b) - 1)), \
                "a bucket's parameters must be contiguous in the flat buffer"
        self._bucket_of = {id(p): bi for bi, b in enumerate(self._buckets) for p in b}
        self._pending = [0] * len(self._buckets)
        self._launched = [False] * len(self._buckets)
        self._used: Optional[set] = None
        self.last_unused = 0    # parameters the last zero(loss) found outside the graph
        self._next = 0          # first bucket that has not been launched: launches happen in index order only
        self._armed = False
        self._work = []
        self._overlap = overlap
        self._chunk = None if chunk_numel is None else max(int(chunk_numel), 1)
        if self.shard:  # the grid's chunk: a multiple of the quantum (None: one chunk per bucket)
            self._chunk = None if self._chunk is None else max(self._chunk // self._quantum, 1) * self._quantum
        self._late = set(int(i) for i in late_buckets)
        # buckets a NATIVE backward launches itself, the moment their gradient is in the queue (launch_from_native)
        self._early: Dict[int, int] = {}        # id(param) -> bucket index (single-parameter buckets only)
        self._early_fields: list = []           # fields whose native backward calls _native_ready (launch_from_native)
        self._early_done: set = set()
        self._producers: Dict[int, int] = {}    # id(param) -> autograd edges into its AccumulateGrad in this step's graph (zero(loss))
        self._cb_error: Optional[BaseException] = None
        self.last_early_buckets = 0
        self._gather_work = []   # in-flight parameter all-gathers (sharded mode): (work, a, b, late)
        self._gather_events = []
        self.last_gather_collectives = 0
        # diagnostics of the last step (bench.py prints them per rank): seconds finish() spent blocked in work.wait(), collectives
        # issued, buckets that left from the autograd hooks (i.e. overlapped with the rest of backward)
        self.last_wait_s = 0.0
        self.last_collectives = 0
        self.last_overlapped_buckets = 0
        # with time_waits = True on a CUDA buffer, every finish() brackets its waits with a pair of events on the current stream:
        # the GPU time between them is the part of the exchange that was NOT hidden behind backward (wait() itself only makes
        # the stream wait, the host returns at once); exposed_ms() reads them
        self.time_waits = False
        self._wait_events = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._arm()

    def close(self):
        """Detach from the parameters: hooks and gradient slots removed, the native callback cleared (a second FlatGradients over the same
        parameters - tests, a re-built optimiser - must not find this one's hooks still firing)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            for attr in (SLOT_ATTR, CLAIM_ATTR):
                if hasattr(p, attr):
                    delattr(p, attr)
        for fld in self._early_fields:
            try:
                from sdfstudio_amd import _lib

                _lib.field_set_table_grad_callback(fld._handle, None)
            except Exception:  # noqa: BLE001 - library not loadable here: nothing was registered either
                pass
        self._early_fields = []
        self._early = {}

    # ---- buffer ownership
    def _view(self, p):
        off = self._offset[id(p)]
        return self.flat[off:off + p.numel()].view_as(p)

    def _attach(self):
        for p in self.params:
            p.grad = self._view(p)
            # native backward kernels write this parameter's gradient straight into its slice (grad_slots.py)
            setattr(p, SLOT_ATTR, (lambda q=p: self._view(q)))
            setattr(p, CLAIM_ATTR, False)

    def _is_view(self, p) -> bool:
        return p.grad is not None and p.grad.data_ptr() == self.flat.data_ptr() + 4 * self._offset[id(p)]

    def zero(self, loss: Optional[torch.Tensor] = None):
        """Clear every gradient (the ONLY way gradients of these parameters should be cleared) and re-arm the buckets.
        loss: the tensor backward() is about to be called on - its graph tells which parameters will receive a gradient in this step, so
        that a bucket leaves as soon as the last of THOSE has arrived (module docstring: unused parameters).
        optimizer.zero_grad() / module.zero_grad() default to set_to_none=True, which drops the views (autograd would then
        allocate fresh gradients and the flat buffer would go stale): the views are re-attached here, and whatever stray
        tensor sat in .grad is DISCARDED - this call means "all gradients are zero now"."""
        if self._work:
            raise RuntimeError("FlatGradients.zero() while all-reduces of the previous backward are in flight: call finish() first")
        for prm, fn in self._active_fn.values():  # progressive hash levels: the model is asked, the caller cannot forget (track_active)
            self.set_active_numel(prm, fn())
        for a, b in self.live_ranges():  # one range (the whole buffer) unless a parameter has a never-active suffix
            self.flat[a:b].zero_()
        for p in self.params:
            # None, not a view: AccumulateGrad then ADOPTS the first incoming gradient instead of adding it into .grad - and the native
            # backward kernels hand it a view of this buffer they have already written (grad_slots.py): no launch per parameter
            p.grad = None
            setattr(p, CLAIM_ATTR, False)
        self._arm(None if loss is None else self.reachable(loss))

    zero_grad = zero

    def set_active_numel(self, param: torch.nn.Parameter, numel: Optional[int]):
        """Exchange only param.view(-1)[:numel] (None: all of it); the rest must be identically zero on every rank."""
        if numel is None or numel >= param.numel():
            self._active.pop(id(param), None)
            if id(param) in self._hwm:
                self._hwm[id(param)] = param.numel()  # everything has been live from here on (a later restriction skips nothing)
        else:
            n = max(int(numel), 0)
            self._active[id(param)] = n
            # a restriction that arrives after unrestricted steps (or after mark_all_live: loaded moments) finds gradients / Adam
            # moments beyond it: nothing to skip then
            first = param.numel() if (id(param) not in self._hwm and (self._finished_steps > 0 or self._all_live)) else 0
            self._hwm[id(param)] = max(self._hwm.get(id(param), first), n)

    def track_active(self, param: torch.nn.Parameter, numel_fn):
        """set_active_numel(param, numel_fn()) at every zero(): the model's own statement of how much of the parameter can carry a
        gradient in the coming step (NeuSFactoModel.active_table_floats under the progressive level mask).  Why this matters: zero()
        clears only live_ranges() and the native backward kernels ACCUMULATE into the slots, so a level that is switched on without
        the restriction being widened would pile stale gradient into rows that are neither exchanged nor stepped."""
        self._active_fn[id(param)] = (param, numel_fn)

    def mark_all_live(self):
        """Forget the never-active suffixes for good (e.g. after loading optimizer moments from a checkpoint: the moments beyond the
        current restriction may be non-zero).  Sticky: restrictions set later still bound the EXCHANGE, never what zero() / Adam visit."""
        self._all_live = True
        for k in list(self._hwm):
            self._hwm[k] = 1 << 62

    def live_ranges(self):
        """Contiguous [start, end) ranges of the flat buffer that may hold a non-zero gradient: everything except the suffix of a
        parameter that set_active_numel has kept inactive since the first step (progressive hash levels: the table rows of levels
        that have not been switched on yet - BASELINE config 5 starts with 8 of 16 levels of a 2.1 GB table).  zero() rewrites and
        the fused Adam step visits only these: beyond them gradient, moments and update are exactly zero."""
        out = []
        for p in self.params:
            off = self._offset[id(p)]
            n = min(self._hwm.get(id(p), p.numel()), p.numel())
            if n > 0:
                if out and out[-1][1] == off:
                    out[-1][1] = off + n
                else:
                    out.append([off, off + n])
            if n < p.numel():
                out.append([off + p.numel(), off + p.numel()])  # break the run: the next parameter starts a new range
        return [(a, b) for a, b in out if b > a]

    # ---- exchange
    def reachable(self, loss: torch.Tensor) -> set:
        """ids of the parameters whose AccumulateGrad node hangs below `loss` in the autograd graph, i.e. whose post-accumulate hook
        can fire in loss.backward() (DDP's find_unused_parameters walk, torch/nn/parallel/distributed.py; the reference enables it:
        pipelines/base_pipeline.py:242)."""
        seen, out = set(), set()
        self._producers = {}
        stack = [loss.grad_fn]
        while stack:
            fn = stack.pop()
            if fn is None or fn in seen:
                continue
            seen.add(fn)
            var = getattr(fn, "variable", None)  # AccumulateGrad
            if var is not None:
                out.add(id(var))
                continue
            for f, _ in fn.next_functions:
                v = getattr(f, "variable", None) if f is not None else None
                if v is not None:
                    self._producers[id(v)] = self._producers.get(id(v), 0) + 1  # one edge = one producer of this parameter's gradient
                stack.append(f)
        return out

    def _arm(self, used: Optional[set] = None):
        for bi, b in enumerate(self._buckets):
            self._pending[bi] = len(b) if used is None else sum(1 for p in b if id(p) in used)
            self._launched[bi] = False
        self._used = used
        self._next = 0
        self._work = []
        self._armed = True
        self.last_overlapped_buckets = 0
        self.last_collectives = 0
        self.last_early_buckets = 0
        self._early_done = set()
        if used is None:
            self._producers = {}
        self.last_unused = 0 if used is None else sum(1 for p in self.params if id(p) not in used)

    def _raw_ranges(self, bi):
        """Contiguous [start, end) element ranges of bucket bi that can carry a gradient in this step (the active prefixes)."""
        out = []
        for p in self._buckets[bi]:
            off = self._offset[id(p)]
            n = self._active.get(id(p), p.numel())
            if n <= 0:
                continue
            if out and out[-1][1] == off:
                out[-1][1] = off + n
            else:
                out.append([off, off + n])
            if n < p.numel():  # a gap follows: the next parameter starts a new range
                out.append([off + p.numel(), off + p.numel()])
        return [(a, b) for a, b in out if b > a]

    def _grid(self, bi):
        """Sharded mode: the FIXED chunks [start, end) of bucket bi (anchored at the bucket's start; every length a multiple of W)."""
        a0, b0 = self._span[bi]
        step = (b0 - a0) if self._chunk is None else self._chunk
        return [(s, min(s + step, b0)) for s in range(a0, b0, max(step, 1))]

    def _ranges(self, bi):
        """What travels for bucket bi.  All-reduce mode: the active ranges, cut into chunks of at most chunk_numel.  Sharded mode: the
        chunks of the fixed grid that intersect an active range, WHOLE (what lies beyond the active prefix inside such a chunk is zero
        on every rank; the grid - and with it the ownership of every element - never moves)."""
        raw = self._raw_ranges(bi)
        if self.shard:
            return [(a, b) for a, b in self._grid(bi) if any(ra < b and a < rb for ra, rb in raw)]
        cut = []
        for a, b in raw:
            step = (b - a) if self._chunk is None else self._chunk
            for s in range(a, b, step):
                cut.append((s, min(s + step, b)))
        return cut

    def owned(self, a: int, b: int):
        """The slice [oa, ob) of grid chunk [a, b) this rank owns (sharded mode; everything otherwise)."""
        if not self.shard or self.world == 1:
            return a, b
        n = (b - a) // self.world
        return a + self.rank * n, a + (self.rank + 1) * n

    def owned_slices(self):
        """Every slice of the flat buffer this rank owns, with its offset in the rank-local moment buffers: [(a, b, local_offset)], in
        buffer order over the WHOLE grid (independent of the active prefix).  All-reduce mode: one slice, the buffer itself."""
        if not self.shard or self.world == 1:
            return [(0, self.flat.numel(), 0)]
        out, loc = [], 0
        for bi in range(len(self._buckets)):
            for a, b in self._grid(bi):
                oa, ob = self.owned(a, b)
                out.append((oa, ob, loc))
                loc += ob - oa
        return out

    def local_numel(self) -> int:
        """Elements of optimiser state this rank keeps (sharded mode: 1 / W of the padded buffer)."""
        return sum(b - a for a, b, _ in self.owned_slices())

    def owned_live(self):
        """owned_slices() intersected with live_ranges(): what the optimiser step of THIS rank visits."""
        live = self.live_ranges()
        out = []
        for a, b, loc in self.owned_slices():
            for la, lb in live:
                x, y = max(a, la), min(b, lb)
                if y > x:
                    out.append((x, y, loc + (x - a)))
        return out

    def _use_reduce_scatter(self) -> bool:
        # gloo has no reduce_scatter_tensor: an all-reduce of the chunk leaves the same sum in the owned slice (CPU tests, and the
        # single-GPU control-flow runs of bench.py with SDFHIP_BENCH_BACKEND=gloo)
        return self.shard and dist.get_backend(self.group) == "nccl" and os.environ.get("SDFHIP_SHARD_EMULATE") != "1"

    def _launch(self, bi):
        assert bi == self._next and not self._launched[bi], "buckets leave in index order"
        self._launched[bi] = True
        self._next = bi + 1
        if not _dist_on(self.group):
            return
        rs = self._use_reduce_scatter()
        for a, b in self._ranges(bi):
            if rs:  # in place: this rank's slice of the chunk receives the sum over the ranks (ncclReduceScatter with recv = send + rank * count)
                oa, ob = self.owned(a, b)
                work = dist.reduce_scatter_tensor(self.flat[oa:ob], self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                work = dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._work.append((work, a, b))
            self.last_collectives += 1

    def _launch_ready(self, from_hook: bool):
        while self._next < len(self._buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            if from_hook:
                self.last_overlapped_buckets += 1

    def _on_grad(self, p):
        if not self._armed:
            raise RuntimeError("FlatGradients: backward without zero() since the last finish() - the flat buffer still holds the "
                               "reduced gradients of the previous step")
        bi = self._bucket_of[id(p)]
        if self._used is not None and id(p) not in self._used:
            raise RuntimeError("FlatGradients: a gradient arrived for a parameter that is not in the graph of the loss zero() was given "
                               "(zero(loss) and backward() must see the same loss)")
        self._pending[bi] -= 1
        if bi in self._early_done and self._pending[bi] == 0:
            # launched from inside the native backward that produced it (launch_from_native): the hook only confirms that autograd adopted
            # the slot the kernels wrote - anything else would mean the collective carried a partial gradient
            if not self._is_view(p):
                raise RuntimeError("FlatGradients: a bucket was launched from the native backward, but autograd delivered a gradient that is "
                                   "not the slot it wrote (another producer appeared after the graph walk)")
            if self._overlap:
                self._launch_ready(from_hook=True)
            return
        if self._pending[bi] < 0 or self._launched[bi]:
            raise RuntimeError("FlatGradients: a second backward reached a parameter before finish() - its bucket may already be "
                               "in flight (accumulate micro-batches into one loss, or call finish() / zero() between backwards)")
        if not self._is_view(p):
            # AccumulateGrad runs once per leaf and backward with the SUM of everything that reached the parameter.  It adopted an
            # ordinary tensor: a torch op produced the gradient, or several producers did and the engine summed them out of place (a
            # native kernel's slice is then one of the summands, already inside this total), or it cloned the view it was handed.
            # In every case p.grad is the whole gradient of this pass: it REPLACES the slice.
            v = self._view(p)
            v.copy_(p.grad)
            p.grad = v
        if self._overlap:
            self._launch_ready(from_hook=True)

    def launch_from_native(self, param: torch.nn.Parameter, field):
        """Let the native backward that produces `param`'s gradient start its bucket's exchange ITSELF, the moment the producing kernels
        are in the queue (include/sdfhip.h: sdfhip_field_set_table_grad_callback, registered on `field`, the SDFField that owns
        `param`: the hook is state of that field's native handle, not of the process): the SDF field is one autograd node, so the table's hook
        fires only after the whole call - scatter, THEN ~2 ms of weight-gradient GEMMs - has been enqueued, and a collective launched
        from the hook waits for all of it.  From the callback it waits for the scatter alone and travels beside the GEMMs.
        Conditions, all checked per step (else the hook launches the bucket as usual): `param` is a bucket of its own and the next to
        leave; zero(loss) found exactly ONE producer of its gradient in the graph; the kernels wrote the flat slot itself."""
        from sdfstudio_amd import _lib

        bi = self._bucket_of[id(param)]
        assert len(self._buckets[bi]) == 1, "launch_from_native: the parameter must be a bucket of its own (distributed.plan_buckets)"
        self._early[id(param)] = bi
        _lib.field_set_table_grad_callback(field._handle, self._native_ready)
        self._early_fields.append(field)

    def _native_ready(self, table_bar_ptr: int, stream_ptr: int):
        try:
            if not (self._armed and self._overlap and _dist_on(self.group)):
                return
            for pid, bi in self._early.items():
                if table_bar_ptr != self.flat.data_ptr() + 4 * self._offset[pid]:
                    continue  # another field's table, or a producer that was handed an ordinary tensor
                if bi != self._next or self._launched[bi] or self._producers.get(pid, 0) != 1:
                    return
                if self.flat.is_cuda:
                    with torch.cuda.stream(torch.cuda.ExternalStream(stream_ptr, device=self.flat.device)):
                        self._launch(bi)  # the collective orders itself behind `stream`: the scatter, not the GEMMs that follow
                else:
                    self._launch(bi)
                self._early_done.add(bi)
                self.last_overlapped_buckets += 1
                self.last_early_buckets += 1
                return
        except BaseException as e:  # noqa: BLE001 - cannot cross the C frame: finish() re-raises
            self._cb_error = e

    def finish(self, average: bool = True, wait: bool = True) -> float:
        """Wait for the outstanding bucket all-reduces (launching, in index order, every bucket that has not left yet: e.g. one
        behind a bucket with parameters unused in this step).  average=True turns the sums into means in place; average=False
        leaves the SUMS and returns the scale (1 / world_size) for the consumer to apply - the fused Adam step multiplies the
        gradient by it as it reads it (engine/optimizers.py), which saves the pass over the buffer."""
        if not self._armed:
            raise RuntimeError("FlatGradients.finish() twice without zero() + backward in between")
        if self.shard and wait and _dist_on(self.group) and self._use_reduce_scatter():
            # after a reduce-scatter only the OWNED slice of every chunk holds the sum; the rest of the buffer - most p.grad views - still
            # holds this rank's local gradient.  Returning "means" here would hand gradient clipping / grad-norm logging different numbers
            # than the all-reduce exchange does
            raise RuntimeError("FlatGradients.finish(wait=True) under the sharded exchange: p.grad is valid on owned slices only - close the "
                               "step with Optimizers.optimizer_step_all(grad_scale=None), which takes the reduce-scatters chunk by chunk")
        if self._cb_error is not None:
            err, self._cb_error = self._cb_error, None
            raise RuntimeError("FlatGradients: the native table-gradient callback failed") from err
        for p in self.params:
            if not self._is_view(p):
                if p.grad is None:
                    p.grad = self._view(p)  # unused in this step: its gradient is the zeros zero() left
                    continue
                if self._launched[self._bucket_of[id(p)]]:
                    raise RuntimeError("FlatGradients: a gradient outside the flat buffer appeared after its bucket was launched")
                v = self._view(p)  # stray gradient of a parameter whose hook never fired (grad set by hand): fold it in
                v.add_(p.grad)
                p.grad = v
        if _DEBUG_LIVE:  # SDFHIP_DEBUG_GRADS=1: nothing may have been written beyond a never-active suffix (it is neither zeroed,
            for p in self.params:  # exchanged nor stepped: a level switched on without widening the restriction would pile up there)
                n = self._hwm.get(id(p))
                if n is not None and n < p.numel():
                    off = self._offset[id(p)]
                    if float(self.flat[off + n:off + p.numel()].abs().max()) != 0.0:
                        raise RuntimeError("FlatGradients: gradient beyond the active prefix of a parameter (set_active_numel / track_active "
                                           "was not widened before a level was switched on)")
        for bi in range(self._next, len(self._buckets)):
            self._launch(bi)
        scale = 1.0
        self.last_wait_s = 0.0
        if _dist_on(self.group):
            w = dist.get_world_size(self.group)
            scale = 1.0 / w
            if not wait:
                # the consumer (FusedAdam.step in sharded mode) takes the collectives chunk by chunk: pop_work()
                assert not average, "finish(wait=False) hands over SUMS"
                self._armed = False
                self._finished_steps += 1
                return scale
            ev = None
            if self.time_waits and self.flat.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            t0 = time.perf_counter()
            for work, a, b in self._work:
                work.wait()
                if average:
                    self.flat[a:b].div_(w)
            self.last_wait_s = time.perf_counter() - t0
            if ev is not None:
                ev[1].record()
                self._wait_events.append(ev)
        self._work = []
        self._armed = False
        self._finished_steps += 1
        return 1.0 if average else scale

    def pop_work(self):
        """After finish(wait=False): the in-flight gradient collectives [(work, a, b)] in launch order; the caller waits for each before it
        reads the chunk (its owned slice, in sharded mode).  wait_chunk() does the wait with the exposed-time bookkeeping of finish()."""
        work, self._work = self._work, []
        return work

    def wait_chunk(self, work):
        ev = None
        if self.time_waits and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        t0 = time.perf_counter()
        work.wait()
        self.last_wait_s += time.perf_counter() - t0
        if ev is not None:
            ev[1].record()
            self._wait_events.append(ev)

    # ---- sharded mode: parameters travel back
    def gather_chunks(self, with_late: bool = False):
        """Grid chunks whose parameters an optimiser step may have changed on SOME rank: those that intersect live_ranges(), in buffer
        order (everything else has zero gradient and zero moments everywhere: no update, nothing to send)."""
        live = self.live_ranges()
        out = []
        for bi in range(len(self._buckets)):
            for a, b in self._grid(bi):
                if any(la < b and a < lb for la, lb in live):
                    out.append((a, b, bi in self._late) if with_late else (a, b))
        return out

    def gather_parameters(self, flat_params: torch.Tensor):
        """All-gather the owned slices of gather_chunks() of the flat PARAMETER buffer (same layout as the gradient buffer) into every
        rank's copy, asynchronously, in the order the next step needs them - the late buckets (the big table) last: wait_parameters()
        before anything reads the parameters again.  In place (RCCL: ncclAllGather with send = recv + rank * count); gloo takes a list of
        views of the chunk."""
        if not (self.shard and _dist_on(self.group)):
            return
        assert flat_params.numel() == self.flat.numel(), "the parameter buffer must mirror the gradient buffer's layout"
        native = dist.get_backend(self.group) == "nccl" and os.environ.get("SDFHIP_SHARD_EMULATE") != "1"
        chunks = self.gather_chunks(with_late=True)
        for a, b, late in sorted(chunks, key=lambda c: c[2]):  # stable: buffer order within each class
            oa, ob = self.owned(a, b)
            if native:
                work = dist.all_gather_into_tensor(flat_params[a:b], flat_params[oa:ob], group=self.group, async_op=True)
            else:
                n = ob - oa
                views = [flat_params[a + r * n:a + (r + 1) * n] for r in range(self.world)]
                work = dist.all_gather(views, flat_params[oa:ob].clone(), group=self.group, async_op=True)
            self._gather_work.append((work, a, b, late))
            self.last_gather_collectives += 1

    def wait_parameters(self, late: bool = True):
        """Wait for the in-flight parameter all-gathers: all of them, or (late=False) all but the late buckets', which stay in flight until
        a later call.  The wait is a stream wait: GPU work enqueued BEFORE this call overlaps the gathers."""
        if not self._gather_work:
            return
        keep, ev = [], None
        if self.time_waits and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for item in self._gather_work:
            if item[3] and not late:
                keep.append(item)
            else:
                item[0].wait()
        if ev is not None:
            ev[1].record()
            self._gather_events.append(ev)
        self._gather_work = keep

    def gathered_bytes(self) -> int:
        return 4 * sum(b - a for a, b in self.gather_chunks()) if self.shard else 0

    all_reduce_mean = finish

    def exposed_ms(self, reset: bool = True) -> List[float]:
        """GPU milliseconds the compute stream stalled on the gradient exchange in every wait since the last reset (time_waits): one
        entry per finish(), or per chunk wait in sharded mode (sum them per step)."""
        if self._wait_events:
            torch.cuda.synchronize(self.flat.device)
        out = [a.elapsed_time(b) for a, b in self._wait_events]
        if reset:
            self._wait_events = []
        return out

    def exposed_gather_ms(self, reset: bool = True) -> List[float]:
        """The same for the parameter all-gathers of sharded mode: one entry per wait_parameters() that had something to wait for."""
        if self._gather_events:
            torch.cuda.synchronize(self.flat.device)
        out = [a.elapsed_time(b) for a, b in self._gather_events]
        if reset:
            self._gather_events = []
        return out

    def exchanged_numel(self) -> int:
        return sum(b - a for bi in range(len(self._buckets)) for a, b in self._ranges(bi))


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """DDP's initial parameter broadcast from rank 0 (base_pipeline.py:242)."""
    if _dist_on():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


def plan_buckets(param_groups: Dict[str, Sequence[torch.nn.Parameter]], big_numel: int = 1 << 22, late_groups: Sequence[str] = ("fields",)):
    """Buffer order and buckets for the sharded exchange from the model's parameter groups: within every group the big tensors (hash
    tables: >= big_numel elements) come first, each as a bucket of its own - a table's gradient is complete long before the weight
    gradients of its group - then one bucket with the rest of the group.  The big buckets of `late_groups` (the SDF field's table: read
    only after the proposal sampling of the next step) are the ones whose parameters travel back last.
    Returns (params in buffer order, buckets, indices of the late buckets = FlatGradients(late_buckets=...))."""
    params, buckets, late = [], [], []
    for name, plist in param_groups.items():
        plist = [p for p in plist if p.requires_grad]
        big = [p for p in plist if p.numel() >= big_numel]
        rest = [p for p in plist if p.numel() < big_numel]
        for p in big:
            if name in late_groups:
                late.append(len(buckets))
            buckets.append([p])
        if rest:
            buckets.append(rest)
        params += big + rest
    return params, buckets, late
