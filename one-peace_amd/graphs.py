"""hipGraph capture of forward-only feature extraction (the launch-bound end of BASELINE configs[1]: batch 1-8 through 40
layers is ~9 kernels per layer of a few microseconds each, so the host's launch path, not the GPU, sets the latency).

Every kernel of the HIP path is enqueued on torch's current stream and the C-ABI never synchronises or allocates
(include/onepeace_hip.h), so a whole ``model(...)`` call records into one graph; replay is a single hipGraphLaunch.  The
captured call owns static input / output buffers: ``__call__`` copies the new inputs in, replays, and returns the outputs
(clones by default -- the static buffers are overwritten by the next replay).  One graph per input-shape signature.

``TrainStepGraph`` does the same for the launch-bound end of TRAINING (small per-GPU batches: 7 000 launches per step cost
the host ~30 us each, more than the kernels take below ~32 tuples): gradient zeroing, the forwards, the loss and the whole
backward -- autograd included -- record into one graph; the optimiser step stays outside (its bias-correction step count is a
host-side kernel argument) and so do the gradient collectives of a multi-rank run."""
import torch


def _signature(inputs):
    return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(inputs.items()) if torch.is_tensor(v))


class GraphedCall:
    def __init__(self, fn, example_inputs, warmup=2):
        tensors = {k: v for k, v in example_inputs.items() if torch.is_tensor(v)}
        if not tensors or not all(v.is_cuda for v in tensors.values()):
            raise RuntimeError("GraphedCall: inputs must be device tensors (hipGraph capture has no CPU path)")
        self.signature = _signature(example_inputs)
        self.static_in = {k: v.clone() for k, v in tensors.items()}
        self.consts = {k: v for k, v in example_inputs.items() if not torch.is_tensor(v)}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():  # allocator / workspace / derived-buffer warm-up outside the capture
            for _ in range(warmup):
                fn(**self.static_in, **self.consts)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = fn(**self.static_in, **self.consts)

    def __call__(self, clone=True, **inputs):
        if _signature(inputs) != self.signature:
            raise RuntimeError("GraphedCall: input shapes/dtypes differ from the captured ones: %s vs %s"
                               % (_signature(inputs), self.signature))
        for k, buf in self.static_in.items():
            buf.copy_(inputs[k])
        self.graph.replay()
        out = self.static_out
        if not clone:
            return out
        return out.clone() if torch.is_tensor(out) else type(out)(o.clone() if torch.is_tensor(o) else o for o in out)


class GraphCache:
    """fn + one GraphedCall per input signature (feature extraction sees a handful of batch shapes)."""

    def __init__(self, fn, max_graphs=8):
        self.fn, self.max_graphs, self.graphs = fn, max_graphs, {}

    def __call__(self, **inputs):
        sig = _signature(inputs)
        g = self.graphs.get(sig)
        if g is None:
            if len(self.graphs) >= self.max_graphs:  # bounded: each graph pins its activations' memory pool
                self.graphs.pop(next(iter(self.graphs)))
            g = self.graphs[sig] = GraphedCall(self.fn, inputs)
        return g(**inputs)


class TrainStepGraph:
    """One hipGraph of ``fwd_bwd()`` = zero the gradients, forward(s), loss, backward, on a STATIC batch.

    ``fwd_bwd`` must not synchronise, must keep every gradient buffer at a fixed address (``FlatParameters`` does: ``.grad`` of
    every parameter is a view of one flat buffer that the kernels and autograd accumulate into in place) and returns the tensors
    to read after a replay (the loss).  Random draws of torch ops inside it (drop-path masks) are replay-safe: torch registers
    its Philox state with the graph and advances it per replay.  New data: ``copy_`` into the tensors the captured call read.
    Build it BEFORE any eager backward of the model: autograd creates a parameter's AccumulateGrad node on the stream of the first
    backward that reaches it, and nodes living on the default stream cannot take part in a capture (the warm-up here runs on a
    side stream for that reason)."""

    def __init__(self, fwd_bwd, warmup=2):
        if not torch.cuda.is_available():
            raise RuntimeError("TrainStepGraph: hipGraph capture has no CPU path")
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # the bucketed gradient reducer works through Python hooks that only run at capture time: a replay would launch no
            # all-reduce (or the captured ones, out of step with reset() / finish()) -- data-parallel steps stay eager
            raise RuntimeError("TrainStepGraph: single-process only (world size %d): the gradient reducer's hooks do not run on "
                               "replay" % dist.get_world_size())
        from . import ops
        if ops.FP8_FFN:
            # the fp8 copies of the FFN weights are re-quantised by refresh_weight_cache() after every optimiser step, in place,
            # so a replay reads current weights; the capture below must therefore see them already cached (the warm-up does that)
            pass
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # autograd's stream bookkeeping, workspaces and derived buffers settle on a side stream
            for _ in range(warmup):
                fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fwd_bwd()

    def replay(self):
        self.graph.replay()
        return self.out
