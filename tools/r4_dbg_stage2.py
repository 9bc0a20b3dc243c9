"""debug: which loss term differs between the stage-2 (frozen) and the fully trainable audio-language pretraining model?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_model_cpu import _build_pretrain
from one_peace_amd.criterions.pretrain import AudioTextPretrainLossCriterion
from one_peace_amd import ops, hip
fx = torch.load("tests/golden/micro_pretrain_al_stage2.pt", weights_only=False)
ni = {k: (v.cuda().to(torch.bfloat16) if v.is_floating_point() else v.cuda()) for k, v in fx["net_input"].items()}
crit = AudioTextPretrainLossCriterion(None, 1.0, 0.5, 0.5, 2.5, label_smoothing=0.0)
calls = []
orig = ops._ffn_forward
def logged(x_mid, P, S, ps2, keep, want_y=True, grad=None):
    calls.append((tuple(x_mid.shape), keep, want_y, grad))
    return orig(x_mid, P, S, ps2, keep, want_y, grad)
ops._ffn_forward = logged
orig_g = hip.gemm_nt
gl = []
def gemm_logged(A, Bs, *a, **k):
    gl.append((tuple(A.shape), len([b for b in Bs if b is not None]), k.get("epilogue", 0), k.get("h0") is not None, k.get("rowscale") is not None))
    return orig_g(A, Bs, *a, **k)
hip.gemm_nt = gemm_logged
res = {}
for stage2 in (True, False):
    calls.clear(); gl.clear()
    m = _build_pretrain(fx, audio_language=True, stage2=stage2).cuda().to(torch.bfloat16).eval()
    loss, _, log = crit(m, {"net_input": ni, "nsentences": 4})
    res[stage2] = (list(calls), list(gl))
    print("stage2", stage2, float(loss))
a, b = res[True], res[False]
print("ffn calls", len(a[0]), len(b[0]))
for x, y in zip(a[0], b[0]):
    if x != y: print("FFN DIFF", x, y)
print("gemm calls", len(a[1]), len(b[1]))
for i, (x, y) in enumerate(zip(a[1], b[1])):
    if x != y: print("GEMM DIFF", i, x, y)
