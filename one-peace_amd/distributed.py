"""Data-parallel pieces of the pretraining step, MI355X-first (one process per GPU, RCCL over xGMI via
``torch.distributed``'s "nccl" backend).

What the reference does: ``LegacyDistributedDataParallel`` packs gradients into a flat buffer after
backward, divides by the world size and all-reduces ~15 sequential 512 MiB buckets, not overlapped with backward
(fairseq/distributed/legacy_distributed_data_parallel.py:76-165).  Here parameters and gradients LIVE in flat bf16
buffers (no pack/unpack), buckets are contiguous slices that are all-reduced asynchronously as soon as autograd has
finished the last parameter of the bucket (post-accumulate-grad hooks -> overlap with the rest of backward), and the
1/world division is folded into the fused AdamW kernel."""
import collections
import os

import torch
import torch.distributed as dist


def force_collectives():
    """ONEPEACE_FORCE_COLLECTIVES=1: run the broadcast / all-gather / bucketed all-reduce path even at world size 1 (under
    torchrun --nproc-per-node 1), so the RCCL code path can be exercised on a single-GPU box."""
    return bool(os.environ.get("ONEPEACE_FORCE_COLLECTIVES")) and "RANK" in os.environ


def init_distributed(backend=None):
    """torchrun-style env:// initialisation; returns (rank, world, local_rank).  No-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not force_collectives():
        return 0, 1, int(os.environ.get("LOCAL_RANK", "0"))
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available() and not os.environ.get("ONEPEACE_SINGLE_DEVICE_DEBUG"):
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        warm_up_collectives()
    rule = os.environ.get("ONEPEACE_SHARE_CUS")  # "1": one-tile NT launches whatever the world size; "0": never; unset: the default below
    if rule == "1" or (rule is None and SHARE_CUS_DEFAULT and world > 1 and backend == "nccl" and "ONEPEACE_TUNE_SCHED" not in os.environ):
        share_cus_with_collectives(verbose=rank == 0)
    return rank, world, local


# Default launch rule of the NT GEMMs on a multi-GPU node (see share_cus_with_collectives).  Decided by tools/cu_contention_ab.py on one GPU
# (profiles/r6_cu_contention_ab.txt); ONEPEACE_SHARE_CUS=1 / 0 overrides it either way.
SHARE_CUS_DEFAULT = True


def share_cus_with_collectives(verbose=False):
    """NT GEMM launches as one tile per workgroup (tune sched 7: gemm256v_kernel for single problems, the grouped launches as plain tile
    lists; bit-identical results: tests/test_distributed_gpu.py) while RCCL kernels share the GPU with backward.  The persistent form
    (one workgroup per CU walking tiles blockIdx, blockIdx + 256, ...) assumes every workgroup gets its CU at once; a four-wave GEMM
    workgroup owns its CU's whole register file and LDS, so the c CUs an all-reduce kernel holds are c workgroups that start only when
    others END -- the launch then takes up to twice its time, where thousands of one-tile workgroups simply flow onto the CUs that are
    free (T * 256 / (256 - c)).  Without contention the one-tile form is 0.5 ... 3.6 % behind per K = 1536 launch and level over the whole
    step (700.5 / 701.1 against 699.8 / 698.2 ms on one box, profiles/r5_bench_nt_*_samebox_1gpu.json).  The grouped weight-gradient
    launch draws tickets and needs no such switch.  Measured on ONE GPU with a stand-in for the collective's kernel on a CU-masked stream
    (tools/cu_contention_ab.py -> profiles/r6_cu_contention_ab.txt); never yet beside a real RCCL kernel (no multi-GPU node here)."""
    from . import hip
    hip.TUNE.sched = 7
    if verbose:
        import sys
        print("[one_peace_amd.distributed] NT GEMM launches: one tile per workgroup (tune sched 7) while collectives share the CUs; "
              "ONEPEACE_SHARE_CUS=0 keeps the persistent single-GPU rule", file=sys.stderr)


def warm_up_collectives():
    """One tiny launch of every collective kind the step uses, right after initialisation.  RCCL sets up its rings and
    transport buffers (device memory taken from the driver, outside torch's allocator) when a collective kind is first used;
    doing that now keeps those allocations away from the moment the step has filled the GPU with activations."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    world = dist.get_world_size()
    x = torch.ones(1024, dtype=torch.bfloat16, device=dev)
    dist.broadcast(x, src=0)
    dist.all_reduce(x, op=dist.ReduceOp.SUM, async_op=True).wait()
    gathered = torch.empty(world * x.numel(), dtype=x.dtype, device=dev)
    dist.all_gather_into_tensor(gathered, x)
    t = torch.zeros(1, dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()


class FlatParameters:
    """Re-homes a module's parameters (and their .grad) into two flat buffers of the parameter dtype.

    Parameters are laid out group by group -- a group = (lr_scale, weight decay on/off), the optimiser's param groups of
    trainer.py:265-278 / utils/layer_decay.py:34-77 -- so that every group is ONE contiguous range the fused AdamW kernel
    addresses through a small table.  Groups appear in the order their first parameter appears in REVERSE registration order
    (roughly the order backward finishes them: with layer decay the groups are the layers, last layer first), and inside a
    group the reverse registration order is kept; each parameter start is aligned to 8 elements (16 bytes for bf16) so the
    vector kernels can run over any sub-range.

    no_decay(name, p) -> bool: parameters without weight decay.  lr_scale(name, p) -> float (optional): layer-wise lr decay.
    ``groups``: [(start, end, lr_scale, decays)]; ``decay_range`` / ``no_decay_range`` are kept for the two-group layout."""

    ALIGN = 8

    def __init__(self, module, no_decay=lambda name, p: p.dim() <= 1, lr_scale=None):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        named.reverse()
        if lr_scale is None:  # two groups, all decayed parameters first (the layout the bucketed reducer was tuned on)
            keyed = [((1.0, not no_decay(n, p)), n, p) for n, p in named]
            order = [(1.0, True), (1.0, False)]
        else:
            keyed = [((float(lr_scale(n, p)), not no_decay(n, p)), n, p) for n, p in named]
            order = []
            for k, _, _ in keyed:
                if k not in order:
                    order.append(k)
        self.entries, self.groups = [], []
        off = 0
        for key in order:
            start = off
            for k, n, p in keyed:
                if k == key:
                    self.entries.append((n, p, off, p.numel()))
                    off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            if off > start or lr_scale is None:
                self.groups.append((start, off, key[0], key[1]))
        if lr_scale is None:
            self.decay_range, self.no_decay_range = self.groups[0][:2], self.groups[1][:2]
        self.numel = off
        p0 = named[0][1]
        self.params = torch.zeros(off, dtype=p0.dtype, device=p0.device)
        self.grads = torch.zeros(off, dtype=p0.dtype, device=p0.device)
        with torch.no_grad():
            for n, p, o, k in self.entries:
                self.params[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.params[o:o + k].view(p.shape)
                p.grad = self.grads[o:o + k].view(p.shape)
                p._op_flat = True      # ops.AttnBranchFn / FfnBranchFn accumulate this gradient in place (see ops._direct_grad)
                p._op_pending = 0

    def zero_grad(self):
        from . import ops
        ops.reset_wgrads()  # problems queued by a backward pass that raised (OOM handling of the trainer) die with its gradients
        self.grads.zero_()
        for _, p, _, _ in self.entries:
            p._op_pending = 0


class BucketedGradReducer:
    """Asynchronous SUM all-reduce of contiguous gradient buckets, each launched when its last gradient is final.

    Protocol per optimiser step:  ``reset()`` -> forward -> backward -> ``finish()``.
    A parameter's gradient is final when BOTH hold:
      * autograd's post-accumulate-grad hook of the parameter has fired -- the engine runs a leaf's AccumulateGrad node only
        after EVERY backward function that consumes the parameter has returned, whether those functions hand their
        gradient to autograd (torch-op passes) or return None and accumulate into the flat gradient view in place (the
        fused HIP layers; PyTorch >= 2.1 runs the hook for such undefined gradients too), and
      * no in-place contribution is outstanding (``_op_pending == 0``; ``ops._direct_grad_done`` reports here when the
        last fused kernel writing the view has been enqueued -- on PyTorch versions that skip the hook for undefined
        gradients the parameter then simply stays open and its bucket goes out in ``finish()``).
    So a parameter completes exactly once per backward pass also when fused and torch-op passes share it.  A bucket is
    all-reduced as soon as all of its parameters are complete (overlapping the rest of backward); ``finish()`` reduces what
    is left; parameters that took no part in the first step (unused ones, e.g. mask embeddings in a contrastive-only step) are
    remembered and no longer hold their bucket back in later steps (static graph; ``forget_unused()`` re-learns).  Gradient accumulation: wrap all micro-steps but the last in ``no_sync()``; a second
    completion of a parameter whose bucket is already in flight raises instead of corrupting the sum."""

    def __init__(self, flat: FlatParameters, bucket_bytes=256 << 20, process_group=None):
        self.flat, self.pg = flat, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (dist.is_initialized() and force_collectives())
        cap = max(1, bucket_bytes // flat.grads.element_size())
        self.buckets = []  # (start, end, [param indices])
        cur_start, cur_items = 0, []
        for idx, (n, p, o, k) in enumerate(flat.entries):
            cur_items.append(idx)
            end = o + (k + flat.ALIGN - 1) // flat.ALIGN * flat.ALIGN
            if end - cur_start >= cap:
                self.buckets.append((cur_start, end, cur_items))
                cur_start, cur_items = end, []
        if cur_items:
            self.buckets.append((cur_start, flat.numel, cur_items))
        self._bucket_of = {}
        for b, (_, _, items) in enumerate(self.buckets):
            for idx in items:
                self._bucket_of[idx] = b
        self._hooks = []
        self._sync = True
        self._unused = None  # indices of parameters that took no part in the first step (learned in its finish())
        self.stats = {"steps": 0, "buckets": len(self.buckets), "launched_in_backward": 0, "launched_in_finish": 0}
        self.launch_order = []  # bucket indices in the order their all-reduce was issued, all steps (see launch_order_digest)
        self._exposed = collections.deque(maxlen=64)  # (event before the waits, event after) of the last steps (nccl only)
        if self.active:
            for idx, (n, p, o, k) in enumerate(flat.entries):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(idx, from_autograd=True)))
                p._op_on_final = self._make_hook(idx, from_autograd=False)
        self.reset()

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        self.launch_order.append(b)
        if len(self.launch_order) > 1 << 16:  # bounded: the digest below is for first-contact checks, not for long runs
            del self.launch_order[: 1 << 15]
        self._handles.append(dist.all_reduce(self.flat.grads[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _make_hook(self, idx, from_autograd):
        def hook(param):
            if not self._sync:
                return
            b = self._bucket_of[idx]
            if from_autograd:
                if self._unused and idx in self._unused and self._launched[b]:
                    raise RuntimeError(
                        "BucketedGradReducer: parameter %r took no part in the first step (its bucket no longer waits for it) but "
                        "received a gradient now; call forget_unused() when the set of used parameters changes"
                        % self.flat.entries[idx][0])
                if self._hooked[idx] and self._launched[b]:
                    raise RuntimeError(
                        "BucketedGradReducer: parameter %r finished a second backward pass after its bucket was all-reduced "
                        "(several backward passes per step: wrap the accumulation micro-steps in no_sync() and call reset() "
                        "before the last one)" % self.flat.entries[idx][0])
                self._hooked[idx] = True
            if self._done[idx] or not self._hooked[idx] or getattr(param, "_op_pending", 0) > 0:
                return
            self._done[idx] = True
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)
                self.stats["launched_in_backward"] += 1
        return hook

    def no_sync(self):
        """Context manager for gradient-accumulation micro-steps: backward passes inside it only accumulate locally."""
        reducer = self

        class _NoSync:
            def __enter__(self_inner):
                reducer._sync = False

            def __exit__(self_inner, *exc):
                reducer._sync = True
        return _NoSync()

    def reset(self):
        """Call before each (final) backward pass."""
        from . import ops
        ops.reset_wgrads()
        self._pending = [len(items) for (_, _, items) in self.buckets]
        self._done = [False] * len(self.flat.entries)
        self._hooked = [False] * len(self.flat.entries)
        self._launched = [False] * len(self.buckets)
        self._handles = []
        for idx in (self._unused or ()):  # static graph: parameters without a gradient in step 1 (e.g. the mask embeddings in a
            self._done[idx] = True        # contrastive-only step) do not hold their bucket back until finish()
            self._pending[self._bucket_of[idx]] -= 1

    def forget_unused(self):
        """Re-learn which parameters take part in a step (after switching objective / modalities)."""
        self._unused = None

    def finish(self):
        """Reduces every bucket that has not gone out yet (unused parameters) and waits for all of them."""
        if not self.active:
            return
        self.stats["steps"] += 1
        if self._unused is None and self._sync:
            self._unused = {i for i, d in enumerate(self._done) if not d}
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
                self.stats["launched_in_finish"] += 1
        timed = self.flat.grads.is_cuda and dist.get_backend(self.pg) == "nccl"
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in self._handles:
            h.wait()
        if timed:
            e1.record()
            self._exposed.append((e0, e1))
        self._handles = []

    def launch_order_digest(self):
        """A 63-bit digest of the order in which this rank issued its bucket all-reduces so far.  Collectives match up across
        ranks by ISSUE ORDER, not by bucket: ranks that issue the same buckets in different orders all-reduce unrelated slices
        into each other (or deadlock when the sizes differ).  The order follows from autograd's execution order, which is the same
        on every rank for the same graph -- `bench.py --check-replicas` all-gathers this digest and fails loudly if it is not."""
        h = 1469598103934665603
        for b in self.launch_order:
            h = ((h ^ (b + 1)) * 1099511628211) & ((1 << 63) - 1)
        return h

    def overlap_report(self):
        """After a synchronize: how the buckets went out and how long the compute stream sat in finish() waiting for RCCL
        (the part of the gradient all-reduce that backward did NOT hide), averaged per step."""
        rep = dict(self.stats)
        if self._exposed:
            rep["exposed_allreduce_ms_per_step"] = sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)
        self._exposed.clear()
        return rep
