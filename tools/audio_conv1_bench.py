"""Round 5: the fused first block of the audio feature extractor (op_audio_conv1_ln_gelu_fwd / _bwd) against the GEMM + LayerNorm form it
replaces, at the headline size (128 x 16 000 rows of 512 channels)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip, ops, audio_ops  # noqa: E402
from tools.bench_ops import timeit  # noqa: E402

hip.lib()
B, Tp, C = 128, 80000, 512
rows = B * Tp // 5
bf = dict(dtype=torch.bfloat16, device="cuda")
wav = torch.randn(B * Tp + 16, **bf)
w = (torch.randn(C, 1, 10, **bf) * 0.3).requires_grad_(True)
lw, lb = torch.ones(C, **bf).requires_grad_(True), torch.zeros(C, **bf).requires_grad_(True)
dy = torch.randn(rows, C, **bf)
w0 = w.detach().reshape(C, 10).contiguous()
y, mean, rstd = hip.audio_conv1_ln_gelu_fwd(wav, 5, w0, None, lw.detach(), lb.detach(), rows, 1e-5)
tf = timeit(lambda: hip.audio_conv1_ln_gelu_fwd(wav, 5, w0, None, lw.detach(), lb.detach(), rows, 1e-5), iters=5)
tb = timeit(lambda: hip.audio_conv1_ln_gelu_bwd(dy, wav, 5, w0, None, lw.detach(), lb.detach(), mean, rstd), iters=5)
print("fused: forward %.3f ms (%.0f GB/s of the 2.1 GB it writes)   backward %.3f ms" % (tf, rows * C * 2 / tf / 1e6, tb), flush=True)


def unfused_fwd():
    a0 = audio_ops._first_layer_rows(wav, rows)
    x = ops.linear(a0, torch.nn.functional.pad(w.reshape(C, 10), (0, 54)), None)
    return ops.layer_norm(x, lw, lb, 1e-5, gelu=True)


out = unfused_fwd()
tf2 = timeit(lambda: unfused_fwd(), iters=3)
def fb():
    o = unfused_fwd()
    o.backward(dy)
tfb = timeit(fb, iters=3)
print("GEMM + LayerNorm form (incl. the im2col copy): forward %.3f ms   forward + backward %.3f ms" % (tf2, tfb), flush=True)
