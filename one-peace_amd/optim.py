"""Fused AdamW over flat bf16 parameters.

Update rule = the reference's in-repo ``Adam`` (one_peace/optim/adam.py:186-253; what runs when apex is absent):
fp32 moments, fp32 math on bf16 parameters (the ``MemoryEfficientFP16Optimizer`` arrangement: no fp32 master copy),
decoupled weight decay applied before the Adam update, eps added to sqrt(v).  Param groups = the reference's
(trainer.py:265-278 -> utils/layer_decay.py:34-77): no weight decay for ``ndim <= 1`` / ``.bias`` / ``no_weight_decay()`` names,
and -- with ``layer_decay < 1`` -- the lr of layer id i scaled by ``layer_decay ** (L + 1 - i)`` (optim/base_optimizer.py:8-14).
ONE HIP launch over the flat buffers of ``distributed.FlatParameters`` per step, whatever the number of groups (a device table
maps ranges to their lr scale / weight decay): 22 bytes/parameter of HBM traffic.

Deviation from the reference, stated: global-norm clipping multiplies the gradient by the clip coefficient in fp32 inside the
update kernel; the reference scales its bf16 gradients in place first (fairseq/utils.py:393-397: one extra bf16 rounding)."""
import re

import torch

from . import hip, ops
from .distributed import FlatParameters


def layer_id_of(var_name, num_max_layer):
    """one_peace/utils/layer_decay.py:8-21 (get_num_layer): adapters' embeddings -> 0, adapter rel_pos_table k and encoder
    layer k -> k + 1, everything else (projections, logit scale, final norms) -> the last id."""
    for ad in ("text_adapter", "image_adapter", "audio_adapter"):
        if var_name.startswith(ad):
            rest = var_name[len(ad) + 1:]
            return int(rest.split(".")[1]) + 1 if rest.startswith("rel_pos_table") else 0
    if var_name.startswith("fusion_model.layers"):
        return int(var_name.split(".")[2]) + 1
    return num_max_layer - 1


def reference_param_groups(model, num_layers=None, layer_decay=1.0):
    """(no_decay(name, p), lr_scale(name, p)) callables for ``FlatParameters`` that reproduce trainer.py:265-278."""
    skip = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()

    def no_decay(name, p):
        name = re.sub("^module.module.", "", name)
        return p.ndim <= 1 or name.endswith(".bias") or name in skip

    if num_layers is None or layer_decay >= 1.0:
        return no_decay, None
    values = [layer_decay ** (num_layers + 1 - i) for i in range(num_layers + 2)]

    def lr_scale(name, p):
        name = re.sub("^module.module.", "", name)
        return values[layer_id_of(re.sub("^encoder_wrapper.", "", name), len(values))]
    return no_decay, lr_scale


class FusedAdamW:
    def __init__(self, flat: FlatParameters, lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05):
        if not flat.params.is_cuda or flat.params.dtype != torch.bfloat16:
            raise RuntimeError("FusedAdamW needs bf16 parameters on an MI355X (the HIP path has no CPU fallback)")
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros(flat.numel, dtype=torch.float32, device=flat.params.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.step_count = 0
        groups = [g for g in flat.groups if g[1] > g[0]]
        dev = flat.params.device
        self._end8 = torch.tensor([g[1] // 8 for g in groups], dtype=torch.int64, device=dev)
        self._scale = torch.tensor([g[2] for g in groups], dtype=torch.float32, device=dev)
        self._wd = torch.tensor([weight_decay if g[3] else 0.0 for g in groups], dtype=torch.float32, device=dev)
        assert groups and groups[0][0] == 0 and groups[-1][1] == flat.numel and all(a[1] == b[0] for a, b in zip(groups, groups[1:]))

    def set_lr(self, lr):
        """optim/base_optimizer.py:8-14: the scheduled lr; group g runs at lr * lr_scale_g."""
        self.lr = lr

    def step(self, grad_scale=1.0, clip_norm=0.0):
        """grad_scale multiplies the gradient inside the kernel (1/world_size after a SUM all-reduce, trainer.py:917-923).
        clip_norm > 0: the reference's global-norm clipping (trainer.py:929 -> fairseq/utils.py:349-397) -- the norm is
        one extra pass over the flat gradient buffer, the clip coefficient is derived on the device inside the update
        kernel.  Returns the (unclipped, scaled) gradient norm as a device scalar when clipping is on."""
        self.step_count += 1
        f = self.flat
        sq = hip.sqnorm(f.grads) if clip_norm > 0 else None
        hip.adamw_step_groups(f.params, f.grads, self.exp_avg, self.exp_avg_sq, self._end8, self._scale, self._wd, self.lr,
                              self.betas[0], self.betas[1], self.eps, self.step_count, grad_scale, sq, clip_norm)
        ops.refresh_weight_cache()  # the raw-pointer update does not bump _version: refresh the dgrad copies in one launch
        return sq.sqrt() * abs(grad_scale) if sq is not None else None

    def zero_grad(self):
        self.flat.zero_grad()


class TorchAdamW:
    """The same update over ``FlatParameters`` written with torch ops (fp32 math on the flat buffers, any device / dtype): the
    optimiser of the CPU control-flow runs of the data-parallel step (``bench.py --debug-cpu-micro``, world-size-4 gloo test) and
    a readable statement of what the fused kernel computes.  Same interface as ``FusedAdamW``."""

    def __init__(self, flat: FlatParameters, lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros(flat.numel, dtype=torch.float32, device=flat.params.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.step_count = 0

    def set_lr(self, lr):
        self.lr = lr

    @torch.no_grad()
    def step(self, grad_scale=1.0, clip_norm=0.0):
        self.step_count += 1
        f, (b1, b2) = self.flat, self.betas
        g = f.grads.float() * grad_scale
        norm = g.norm() if clip_norm > 0 else None
        if norm is not None:
            g = g * (clip_norm / (norm + 1e-6)).clamp(max=1.0)  # fairseq/utils.py:349-397
        self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        update = self.exp_avg / (self.exp_avg_sq.sqrt() + self.eps)
        bias = (1 - b2 ** self.step_count) ** 0.5 / (1 - b1 ** self.step_count)
        for start, end, scale, decays in f.groups:
            if end <= start:
                continue
            pf = f.params[start:end].float()
            if decays and self.weight_decay != 0:
                pf = pf + pf * (-self.weight_decay * self.lr * scale)
            f.params[start:end].copy_(pf - self.lr * scale * bias * update[start:end])
        if f.params.is_cuda:
            ops.refresh_weight_cache()
        return norm

    def zero_grad(self):
        self.flat.zero_grad()
