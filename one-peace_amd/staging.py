"""Input staging: what the reference does in ``Trainer._prepare_sample``
(fairseq/fairseq/trainer.py:1297-1336: ``utils.move_to_cuda(sample)`` then ``_fp_convert_sample``: fp32 -> bf16 for every
floating tensor of the nested sample dict) -- but off the compute stream.

``SamplePrefetcher`` wraps an iterator of (nested) sample dicts of CPU tensors.  Each sample is copied into one of two sets
of PINNED host buffers, sent to the device with non-blocking copies on a dedicated HIP stream, and converted to the model
dtype there; the compute stream only waits for the copy event of the sample it is about to use.  While step k runs, the
sample of step k+1 is already crossing PCIe.  Integer and bool tensors (token ids, masks, preserve ids) keep their dtype,
exactly like the reference."""
import torch


def _map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(v, fn) for v in obj)
    return obj


class SamplePrefetcher:
    def __init__(self, iterable, device, dtype=torch.bfloat16, depth=2):
        self.it = iter(iterable)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SamplePrefetcher stages host samples onto an MI355X; got device %s" % device)
        self.dtype = dtype
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.depth = depth
        self._pinned = [dict() for _ in range(depth)]  # slot -> {path: pinned host tensor}
        self._slot = 0
        self._queue = []                               # [(device sample, event)]
        self.bytes_staged = 0
        for _ in range(depth):
            self._issue()

    def _pin(self, slot, path, t):
        buf = self._pinned[slot].get(path)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._pinned[slot][path] = buf
        buf.copy_(t)
        return buf

    def _issue(self):
        try:
            sample = next(self.it)
        except StopIteration:
            return
        slot, self._slot = self._slot, (self._slot + 1) % self.depth
        counter = [0]

        def stage(t):
            path = counter[0]
            counter[0] += 1
            if t.is_cuda:
                return t
            host = self._pin(slot, path, t)
            self.bytes_staged += host.numel() * host.element_size()
            d = host.to(self.device, non_blocking=True)
            # trainer.py:1321-1336 (_fp_convert_sample via apply_half / apply_bfloat16): ONLY float32 tensors are converted
            return d.to(self.dtype) if d.dtype == torch.float32 and self.dtype != torch.float32 else d

        # the pinned slot is reused every `depth` samples: its previous H2D copies were enqueued on copy_stream before
        # this one, and the host-side buf.copy_ above may only overwrite it once they are done
        self.copy_stream.synchronize() if self._pinned[slot] else None
        with torch.cuda.stream(self.copy_stream):
            dev = _map(sample, stage)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._queue.append((dev, ev))

    def __iter__(self):
        return self

    def __next__(self):
        if not self._queue:
            raise StopIteration
        dev, ev = self._queue.pop(0)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        _map(dev, lambda t: t.record_stream(cur) if t.is_cuda else None)
        self._issue()
        return dev
