"""Measure the lane <-> element map and the scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 operands)."""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

torch.manual_seed(0)
A = (torch.randn(16, 128) * 2).to(torch.float8_e4m3fn)   # A[row][k]
B = (torch.randn(16, 128) * 2).to(torch.float8_e4m3fn)   # B[col][k]  (B operand = columns of the second matrix)
ref = A.float() @ B.float().t()                           # D[row][col]


def pack(Mx, kmap):
    """lane (g,t) holds row t, 32 fp8: element j <- k = kmap(g, j)."""
    raw = Mx.view(torch.uint8)
    out = torch.zeros(64, 32, dtype=torch.uint8)
    for l in range(64):
        g, t = l >> 4, l & 15
        for j in range(32):
            out[l, j] = raw[t, kmap(g, j)]
    return out.view(torch.int32).reshape(64, 8)


kmaps = {"k=g*32+j": lambda g, j: g * 32 + j,
         "k=j*4+g": lambda g, j: j * 4 + g,
         "k=(j//16)*64+g*16+j%16": lambda g, j: (j // 16) * 64 + g * 16 + (j % 16),
         "k=(j//8)*32+g*8+j%8": lambda g, j: (j // 8) * 32 + g * 8 + (j % 8)}
one = torch.full((64,), 127, dtype=torch.int32)
for name, km in kmaps.items():
    a, b = pack(A, km).cuda(), pack(B, km).cuda()
    d = torch.zeros(64, 4, device="cuda")
    hip._check_probe(hip.probe_lib().op_probe_mfma_f8(hip.ptr(a), hip.ptr(b), hip.ptr(one.cuda()), hip.ptr(one.cuda()), hip.ptr(d), 1, hip.stream()), "probe")
    d = d.cpu()
    got = torch.zeros(16, 16)
    for l in range(64):
        for r in range(4):
            got[(l >> 4) * 4 + r, l & 15] = d[l, r]
    err = (got - ref).abs().max().item()
    print("%-28s max err vs A@B^T (rows=A rows, D[row=g*4+r][col=t]): %.4g   (|ref| max %.1f)" % (name, err, ref.abs().max()))
# scale semantics with the natural map: per-lane scale byte 0 = E8M0 of that lane's 32-element block
km = kmaps["k=g*32+j"]
a, b = pack(A, km).cuda(), pack(B, km).cuda()
sa = torch.full((64,), 127, dtype=torch.int32)
sb = torch.full((64,), 127, dtype=torch.int32)
for l in range(64):
    sa[l] = 127 + ((l >> 4) % 3) - 1 + ((l & 15) % 2)      # depends on (g, t): block scale of A[row t][k block g]
    sb[l] = 127 - ((l >> 4) % 2) + ((l & 15) % 3)
d = torch.zeros(64, 4, device="cuda")
hip._check_probe(hip.probe_lib().op_probe_mfma_f8(hip.ptr(a), hip.ptr(b), hip.ptr(sa.cuda()), hip.ptr(sb.cuda()), hip.ptr(d), 1, hip.stream()), "probe")
d = d.cpu()
got = torch.zeros(16, 16)
for l in range(64):
    for r in range(4):
        got[(l >> 4) * 4 + r, l & 15] = d[l, r]
Af, Bf = A.float().clone(), B.float().clone()
for g in range(4):
    for t in range(16):
        Af[t, g * 32:(g + 1) * 32] *= 2.0 ** (int(sa[g * 16 + t]) - 127)
        Bf[t, g * 32:(g + 1) * 32] *= 2.0 ** (int(sb[g * 16 + t]) - 127)
print("block-scale semantics (lane (g,t) scale byte0 scales A[row t][k in block g]): max err %.4g" % (got - Af @ Bf.t()).abs().max().item())
# fp8 conversion check: v_cvt_pk_fp8_f32 must produce OCP e4m3fn
