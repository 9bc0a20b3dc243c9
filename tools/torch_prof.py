"""torch.profiler view of one bench step at reduced depth: which torch ops (copies, adds, ...) sit between the HIP kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from one_peace_amd.criterions.contrastive import TriModalContrastiveCriterion
from one_peace_amd.distributed import BucketedGradReducer, FlatParameters
from one_peace_amd.optim import FusedAdamW
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
model = bench.build_model(4, dev)
flat = FlatParameters(model, no_decay=lambda n, p: p.dim() <= 1)
red = BucketedGradReducer(flat)
opt = FusedAdamW(flat)
crit = TriModalContrastiveCriterion(None, 0.0)
batch, _ = bench.synthetic_batch(64, dev, 1, audio_seconds=5.0)
sample = {"net_input": batch, "nsentences": 64}

def step():
    opt.zero_grad(); red.reset()
    loss, _, _ = crit(model, sample)
    loss.backward(); red.finish(); opt.step()

for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60,
                                                         max_shapes_column_width=90))
