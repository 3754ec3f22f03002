import sys, ctypes as C, torch
sys.path.insert(0, '.')
import bench
from graph_recsys_benchmark_amd import _lib
from graph_recsys_benchmark_amd.utils.synthetic import SyntheticHIN
w = int(sys.argv[1])
dev = torch.device('cuda', 0)
ds = SyntheticHIN('ml25m_shaped', seed=2019)
model = bench.build_model(ds, 'gat', dev); model.train()
batch = torch.from_numpy(ds.bpr_batch()).to(dev)
if w > 1:
    model.shard(0, w); model._get_engine().plan.layout.dry = True
import os
if os.environ.get("LD_EAGER"): 
    from graph_recsys_benchmark_amd import engine as _e; _e.GRAPHS_ENABLED = False
lib = _lib.load()
with torch.no_grad():
    for _ in range(30): model.loss(batch)
    torch.cuda.synchronize()
    lib.pea_profile_enable(1)
    model.loss(batch)
    torch.cuda.synchronize()
    lib.pea_profile_enable(0)
cap = 1 << 12
names = C.create_string_buffer(cap * 32); ms = (C.c_float * cap)(); units = (C.c_double * cap)(); cnt = C.c_int()
lib.pea_profile_read(cap, names, ms, units, C.byref(cnt))
tot = 0
for i in range(cnt.value):
    nm = names.raw[i * 32:(i + 1) * 32].split(b'\0')[0].decode()
    print('%-28s %8.4f ms  %10.3f MB' % (nm, ms[i], units[i] / 1e6)); tot += ms[i]
print('total', tot)
import time
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): model.loss(batch)
    torch.cuda.synchronize(); print('ms/step', (time.perf_counter() - t0) / 50 * 1e3)
    # host-only time (no sync)
    t0 = time.perf_counter()
    for _ in range(50): model.loss(batch)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print('host enqueue ms/step', (t1 - t0) / 50 * 1e3)
