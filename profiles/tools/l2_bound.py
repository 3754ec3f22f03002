"""Upper bound of what ANY row order could return on the dominant gather, measured instead of argued:
python profiles/tools/l2_bound.py [K ...]
The layer-2 launch of the default workload (`agg_rows_g32`: user rows gathering the 448-byte T_1 rows of the items they rated,
7 channels at once) misses an XCD's 4 MiB L2 on 37 % of its row requests.  A row order can at best make every request hit.
This tool measures that limit directly: the item ids of `user2item` are folded onto K distinct items (id mod K; popularity
is a random permutation of the ids, so the fold keeps the Zipf shape) -- the same destination rows, the same number of
messages per row, the same kernel, but a gather table of K x 448 B.  K = 8000 is one L2's worth (3.6 MB): hit rate ~1.
Prints the average launch time of the kernel per K (HIP events around the launch: csrc ProfScope)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from graph_recsys_benchmark_amd import _lib, engine as _eng  # noqa: E402
from graph_recsys_benchmark_amd.utils import SyntheticHIN  # noqa: E402


def run(k_items):
    dev = torch.device('cuda', 0)
    ds = SyntheticHIN('ml25m_shaped', seed=2019)
    if k_items:
        ei = ds.edge_index_nps['user2item']
        i0 = ds.type_accs['iid']
        ei[1] = i0 + np.mod(ei[1] - i0, k_items)
    model = bench.build_model(ds, 'gat', dev)
    model.train()          # the training-mode loss runs the full-graph forward (reference models/base.py:43-48)
    batch = torch.from_numpy(ds.bpr_batch()).to(dev)
    lib = _lib.load()
    acc, steps = {}, 20
    with torch.no_grad():
        for _ in range(10):
            model.loss(batch)
        torch.cuda.synchronize()
        _eng.GRAPHS_ENABLED = False          # HIP events around every launch need eager launches
        for _ in range(steps):
            lib.pea_profile_enable(1)
            model.loss(batch)
            torch.cuda.synchronize()
            lib.pea_profile_enable(0)
            cap = 1 << 10
            names = C.create_string_buffer(cap * 32)
            ms = (C.c_float * cap)()
            units = (C.c_double * cap)()
            cnt = C.c_int()
            lib.pea_profile_read(cap, names, ms, units, C.byref(cnt))
            for i in range(cnt.value):
                nm = names.raw[i * 32:(i + 1) * 32].split(b'\0')[0].decode()
                acc[nm] = acc.get(nm, 0.0) + ms[i]
    _eng.GRAPHS_ENABLED = True
    return {k: v / steps for k, v in acc.items()}


if __name__ == '__main__':
    ks = [int(a) for a in sys.argv[1:]] or [0, 32000, 16000, 8000, 4000]
    for k in ks:
        r = run(k)
        tot = sum(r.values())
        print('K = %6s items (table %5.1f MB): agg_rows_g32 %.4f ms, all kernels %.4f ms' % (
            k or 'all', (k or 59047) * 448 / 1e6, r.get('agg_rows_g32_gat', float('nan')), tot), flush=True)
