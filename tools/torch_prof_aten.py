"""Which torch-native (aten) ops with GPU time are left in one bench step, by call site: runs tools/torch_prof.py's step at two depths and
prints the aten ops whose launch count grows with depth (per-layer ops) next to those that do not (adapters, head, optimiser)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from one_peace_amd.criterions.contrastive import TriModalContrastiveCriterion
from one_peace_amd.distributed import BucketedGradReducer, FlatParameters
from one_peace_amd.optim import FusedAdamW
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
res = {}
for layers in (2, 6):
    model = bench.build_model(layers, dev)
    flat = FlatParameters(model, no_decay=lambda n, p: p.dim() <= 1)
    red = BucketedGradReducer(flat)
    opt = FusedAdamW(flat)
    crit = TriModalContrastiveCriterion(None, 0.0)
    batch, _ = bench.synthetic_batch(B, dev, 1, audio_seconds=5.0)
    sample = {"net_input": batch, "nsentences": B}

    def step():
        opt.zero_grad(); red.reset()
        loss, _, _ = crit(model, sample)
        loss.backward(); red.finish(); opt.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = {}
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=4):
        if not e.key.startswith("aten::") or e.self_device_time_total <= 0:
            continue
        site = next((s for s in e.stack if "one-peace_amd" in s or "bench.py" in s), e.stack[0] if e.stack else "?")
        k = (e.key, str(e.input_shapes)[:70], site[-70:])
        c = agg.setdefault(k, [0, 0.0])
        c[0] += e.count
        c[1] += e.self_device_time_total
    res[layers] = agg
    del model, flat, red, opt
    torch.cuda.empty_cache()
print("aten ops with GPU time: per-layer (count grows with depth) first; columns: calls@2 calls@6 us@6 | op | shapes | site")
rows = []
for k, (c6, t6) in res[6].items():
    c2 = res[2].get(k, [0, 0.0])[0]
    rows.append((c6 - c2, t6, c2, c6, k))
for d, t6, c2, c6, k in sorted(rows, key=lambda r: (-(r[0] > 0), -r[1]))[:70]:
    print("%4d %4d %9.1f | %-28s | %-70s | %s" % (c2, c6, t6, k[0], k[1], k[2]))
