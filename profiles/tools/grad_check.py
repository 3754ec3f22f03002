"""python profiles/tools/grad_check.py KIND [TRIPLES]: bench.gradient_check of one model kind at full size, all details."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from graph_recsys_benchmark_amd.utils import SyntheticHIN  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'gat'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ds = SyntheticHIN('ml25m_shaped', seed=2019)
model = bench.build_model(ds, kind, torch.device('cuda', 0))
model.train()
batch = torch.from_numpy(ds.bpr_batch()).cuda()[:n]
print(json.dumps(bench.gradient_check(ds, model, batch, kind), indent=1))
