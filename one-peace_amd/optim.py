"""Fused AdamW over flat bf16 parameters.

Update rule = the reference's in-repo ``Adam`` (one_peace/optim/adam.py:186-253; what runs when apex is absent):
fp32 moments, fp32 math on bf16 parameters (the ``MemoryEfficientFP16Optimizer`` arrangement: no fp32 master copy),
decoupled weight decay applied before the Adam update, eps added to sqrt(v).  One HIP launch per decay group over the
flat buffers of ``distributed.FlatParameters``: 22 bytes/parameter of HBM traffic per step."""
import torch

from . import hip, ops
from .distributed import FlatParameters


class FusedAdamW:
    def __init__(self, flat: FlatParameters, lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05):
        if not flat.params.is_cuda or flat.params.dtype != torch.bfloat16:
            raise RuntimeError("FusedAdamW needs bf16 parameters on an MI355X (the HIP path has no CPU fallback)")
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros(flat.numel, dtype=torch.float32, device=flat.params.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.step_count = 0

    def step(self, grad_scale=1.0, clip_norm=0.0):
        """grad_scale multiplies the gradient inside the kernel (1/world_size after a SUM all-reduce, trainer.py:917-923).
        clip_norm > 0: the reference's global-norm clipping (trainer.py:929 -> fairseq/utils.py:349-397) -- the norm is
        one extra pass over the flat gradient buffer, the clip coefficient is derived on the device inside the update
        kernel.  Returns the (unclipped, scaled) gradient norm as a device scalar when clipping is on."""
        self.step_count += 1
        f = self.flat
        sq = hip.sqnorm(f.grads) if clip_norm > 0 else None
        for (s, e), wd in ((f.decay_range, self.weight_decay), (f.no_decay_range, 0.0)):
            if e > s:
                hip.adamw_step(f.params[s:e], f.grads[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e], self.lr, self.betas[0],
                               self.betas[1], self.eps, wd, self.step_count, grad_scale, sq, clip_norm)
        ops.refresh_weight_cache()  # the raw-pointer update does not bump _version: refresh the dgrad copies in one launch
        return sq.sqrt() * abs(grad_scale) if sq is not None else None

    def zero_grad(self):
        self.flat.zero_grad()
