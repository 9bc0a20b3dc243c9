"""Round 4: the non-temporal instantiations of ln_fwd / ln_bwd (training pass over >= 64 MiB) against the default ones (same arithmetic,
other cache policy: results must be BIT-identical) and against fp32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from one_peace_amd import hip  # noqa: E402

torch.manual_seed(0)
rows, cols = 40000, 1536          # 117 MiB: the non-temporal instantiation; the two 20000-row halves: 58.6 MiB each: the default one
x = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
w, b = torch.randn(cols, device="cuda").to(torch.bfloat16), torch.randn(cols, device="cuda").to(torch.bfloat16)
dy = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
y, mean, rstd = hip.layernorm_fwd(x, w, b, want_stats=True)
ys = [hip.layernorm_fwd(x[i * 20000:(i + 1) * 20000].contiguous(), w, b, want_stats=True) for i in range(2)]
assert torch.equal(y, torch.cat([t[0] for t in ys])) and torch.equal(mean, torch.cat([t[1] for t in ys])) and torch.equal(rstd, torch.cat([t[2] for t in ys]))
ref = torch.nn.functional.layer_norm(x.float(), (cols,), w.float(), b.float(), 1e-5)
print("ln_fwd non-temporal == default: True; rel error vs fp32 %.2e" % float((y.float() - ref).norm() / ref.norm()))
dx, dw, db = hip.layernorm_bwd(dy, x, w, b, mean, rstd, add=dy)
dxs = [hip.layernorm_bwd(dy[i * 20000:(i + 1) * 20000].contiguous(), x[i * 20000:(i + 1) * 20000].contiguous(), w, b, mean[i * 20000:(i + 1) * 20000].contiguous(),
                         rstd[i * 20000:(i + 1) * 20000].contiguous(), add=dy[i * 20000:(i + 1) * 20000].contiguous()) for i in range(2)]
assert torch.equal(dx, torch.cat([t[0] for t in dxs])), "ln_bwd dx differs between the two cache policies"
print("ln_bwd non-temporal == default: True; dw finite %s" % bool(dw.float().isfinite().all()))
