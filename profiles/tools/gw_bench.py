"""Micro-benchmark of pea_grad_weight_rows on the shapes of the training step (GPU box):
python profiles/tools/gw_bench.py [N] [LIVE] [CHANNELS] [WIDTH] [OUT]
Level-0 jobs (dZ_0^T A_0: WIDTH x WIDTH per channel, masked dual-source operand) and level-1 jobs (dT_1^T H: OUT x WIDTH)
over a sorted list of LIVE of N rows, tables laid out like the workspace ([N, CHANNELS * WIDTH]).  Prints ms per call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from graph_recsys_benchmark_amd.engine import RowSet, grad_weight  # noqa: E402


def timed(fn, iters=20):
    """ms per call of the library's launches (HIP events around every launch: csrc ProfScope; the Python side of a call --
    nine output allocations, the job table -- takes longer than the small cases' kernels and would hide them)."""
    import ctypes as C
    from graph_recsys_benchmark_amd import _lib
    lib = _lib.load()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    lib.pea_profile_enable(1)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    lib.pea_profile_enable(0)
    cap = 1 << 12
    names = C.create_string_buffer(cap * 32)
    ms = (C.c_float * cap)()
    units = (C.c_double * cap)()
    cnt = C.c_int()
    lib.pea_profile_read(cap, names, ms, units, C.byref(cnt))
    tot = {}
    for i in range(cnt.value):
        nm = names.raw[i * 32:(i + 1) * 32].split(b'\0')[0].decode()
        tot[nm] = tot.get(nm, 0.0) + ms[i]
    return tot.get('grad_weight', 0.0) / iters, tot.get('grad_weight_sum', 0.0) / iters


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 273744
    live_n = int(sys.argv[2]) if len(sys.argv) > 2 else 72401
    p = int(sys.argv[3]) if len(sys.argv) > 3 else 9
    w = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    out = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cpu').manual_seed(1)
    dz = torch.randn(n, p * w, device=dev)
    a0 = torch.randn(n, p * w, device=dev)
    x = torch.randn(n, w, device=dev)
    dt1 = torch.randn(n, p * out, device=dev)
    mask = (torch.rand(n, generator=g) < 0.3).to(torch.uint8).to(dev)
    mark = torch.zeros(n, 4, device=dev)
    ids = torch.randperm(n, generator=g)[:live_n].to(dev)
    mark[ids, 0] = 1.0
    live = RowSet(n, dev).fill_from(mark, 4)
    lvl0 = [(dz[:, c * w:(c + 1) * w], a0[:, c * w:(c + 1) * w], mask, x) for c in range(p)]
    lvl0_sage = lvl0 + [(dz[:, c * w:(c + 1) * w], x) for c in range(p)]
    lvl1 = [(dt1[:, c * out:(c + 1) * out], dz[:, c * w:(c + 1) * w]) for c in range(p)]
    das = torch.randn(n, 12, device=dev)
    att = [(das, x), (das, x)]
    for name, pairs in (('att (2 x [12, %d], all rows are live)' % w, att), ('level 0 (%d x [%d, %d], masked operand)' % (p, w, w), lvl0),
                        ('level 0 SAGE (%d x [%d, %d])' % (2 * p, w, w), lvl0_sage),
                        ('level 1 (%d x [%d, %d])' % (p, out, w), lvl1), ('level 1 SAGE (%d x [%d, %d])' % (2 * p, out, w), lvl1 + lvl1)):
        t_list, s_list = timed(lambda: grad_weight(pairs, rows=live))
        t_all, s_all = timed(lambda: grad_weight(pairs))
        byt = sum(4.0 * (a.shape[1] + b.shape[1]) for a, b, *_ in pairs)
        print('%-44s listed rows %.4f + %.4f ms (%.2f TB/s of operand bytes)   all rows %.4f + %.4f ms (%.2f TB/s)' % (
            name, t_list, s_list, byt * live_n / t_list / 1e9, t_all, s_all, byt * n / t_all / 1e9), flush=True)


if __name__ == '__main__':
    main()
