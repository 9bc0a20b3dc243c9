"""How well-conditioned is each parameter gradient of the micro fixture model under bf16?  (CPU, torch ops only.)

The GPU parity tests accept  err_hip <= max(2 * err_torch_bf16, floor)  per parameter, err = relative Frobenius distance to
the fp32 gradient.  That presumes err_torch_bf16 is a stable yardstick.  For gradients with heavy cancellation it is not:
this script re-runs the bf16 torch path with +-1 bf16 ulp of multiplicative noise on the audio input and prints the spread
of err_torch_bf16 per parameter.  Round-1 finding: the audio relative-position table (|g| = 8.6e-3, 5x smaller than the
text table's) spans 0.06 ... 0.15 over six draws, every other parameter stays within +-25 % of its mean -- hence the wider
floor for small-norm gradients in tests/test_model_gpu.py::test_micro_model_forward_backward.

    python tools/grad_conditioning.py [draws]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.model_util import build_retrieval, load_synth  # noqa: E402
from tests.util import rel_fro  # noqa: E402


def grads(fx, dtype, noise_seed=None):
    from one_peace_amd.criterions.contrastive import TriModalContrastiveCriterion
    m = load_synth(build_retrieval(fx["cfg"], fx["vocab"]), fx["shapes"]).to(dtype).eval()
    inp = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in fx["inputs"].items()}
    if noise_seed is not None:
        g = torch.Generator().manual_seed(noise_seed)
        a = inp["src_audios"].float()
        inp["src_audios"] = (a * (1 + (torch.rand(a.shape, generator=g) - 0.5) * 2 ** -5)).to(dtype)
    loss, _, _ = TriModalContrastiveCriterion(None, 0.0)(m, {"net_input": inp, "nsentences": 4})
    m.zero_grad()
    loss.backward()
    return {n: p.grad.float() for n, p in m.named_parameters() if p.grad is not None}


def main():
    draws = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "micro_retrieval.pt"), weights_only=False)
    ref = grads(fx, torch.float32)
    errs = {}
    for s in [None] + list(range(1, draws)):
        for n, g in grads(fx, torch.bfloat16, s).items():
            if n.startswith("encoder_wrapper.audio_adapter") or "fusion_model" in n:
                errs.setdefault(n, []).append(rel_fro(g, ref[n]))
    rows = sorted(errs.items(), key=lambda kv: -max(kv[1]) / max(min(kv[1]), 1e-12))
    print("%-78s %9s %9s %9s" % ("parameter (bf16 torch path vs fp32, %d draws)" % draws, "|g| fp32", "min err", "max err"))
    for n, e in rows[:12]:
        print("%-78s %9.2e %9.2e %9.2e" % (n, float(ref[n].norm()), min(e), max(e)))


if __name__ == "__main__":
    main()
