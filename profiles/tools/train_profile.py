"""Where one training step (zero_grad + forward + BPR loss + backward + Adam) spends its GPU time: torch profiler table
over 3 steps of the default bench workload.  Run from the repo root on the GPU box."""
import sys, torch
sys.path.insert(0, '.')
import bench
from graph_recsys_benchmark_amd.utils.synthetic import SyntheticHIN
dev = torch.device('cuda', 0)
ds = SyntheticHIN('ml25m_shaped', seed=2019)
model = bench.build_model(ds, sys.argv[1] if len(sys.argv) > 1 else 'gat', dev); model.train()
batch = torch.from_numpy(ds.bpr_batch()).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3)
def step():
    opt.zero_grad(); l = model.loss(batch); l.backward(); opt.step(); return l
for _ in range(3): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print('ms/step', (time.perf_counter() - t0) / 5 * 1e3)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=40, max_name_column_width=60))
