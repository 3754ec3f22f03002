"""Where does the fp32 error of the dense transforms come from?  python profiles/tools/dense_error.py
out = a @ w (K = 64, 128) on the library's f32-input MFMA kernel against float64, next to (1) a strictly sequential fp32
fma chain in k order (one rounding per step: what a v_mfma_f32 k-chain should give if the matrix pipe rounds like v_fma_f32),
(2) the fp32 CPU oracle's product (oracle/pea_oracle.c: gcc-vectorised loop, 8-16 partial sums), (3) torch CPU fp32 (MKL / blocked).
Errors in units of eps x max |out|."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_recsys_benchmark_amd import engine  # noqa: E402
from oracle import oracle as orc  # noqa: E402

EPS = 2.0 ** -24
for k, n_out in ((64, 64), (128, 128), (128, 16)):
    g = torch.Generator().manual_seed(k + n_out)
    n = 4096
    a = torch.randn(n, k, generator=g)
    w = torch.randn(k, n_out, generator=g) * 0.2
    out = torch.empty((n, n_out), device='cuda')
    engine.dense_batch([(a.cuda(), w.cuda(), out)])
    want = (a.double() @ w.double()).numpy()
    scale = np.abs(want).max()
    seq = np.zeros((n, n_out), np.float32)
    a64, w64 = a.double().numpy(), w.double().numpy()
    for kk in range(k):            # fused multiply-add: exact product, one rounding of the sum
        seq = (seq.astype(np.float64) + a64[:, kk:kk + 1] * w64[kk:kk + 1, :]).astype(np.float32)
    tcpu = (a @ w).numpy()
    # the oracle's own product: a GCN layer without edges and bias 0 is x @ W scaled by dinv^2 = 1 (self loop only)
    lp = {'weight': w.numpy(), 'bias': np.zeros(n_out, np.float32)}
    o32 = orc.conv('gcn', a.numpy(), np.zeros((2, 0), np.int64), lp, 1)
    def st(x):
        e = np.abs(x.astype(np.float64) - want) / (EPS * scale)
        return 'mean %.2f p99 %.2f max %.2f' % (e.mean(), np.percentile(e, 99), e.max())
    print('K %3d n_out %3d | hip mfma: %s | sequential fma chain: %s | cpu oracle: %s | torch cpu: %s'
          % (k, n_out, st(out.cpu().numpy()), st(seq), st(o32), st(tcpu)), flush=True)
