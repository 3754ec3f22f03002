"""Host-side cost of one sharded forward + loss step (GPU box): python profiles/tools/host_profile.py [WORLD] [STEPS] [train]
One rank of WORLD on the one GPU, collectives skipped (bench.py --emulate-world): the time the Python / ctypes path needs to
ENQUEUE a step (loop time before the final synchronize) against the time the GPU needs to run it, and cProfile's top
entries of the enqueue loop.  A step is host-bound when the first exceeds the second."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from graph_recsys_benchmark_amd.utils import SyntheticHIN  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    train = len(sys.argv) > 3 and sys.argv[3] == 'train'      # the whole training step (zero_grad, loss, backward, Adam)
    dev = torch.device('cuda', 0)
    ds = SyntheticHIN('ml25m_shaped', seed=2019)
    model = bench.build_model(ds, 'gat', dev)
    model.train()
    batch = torch.from_numpy(ds.bpr_batch()).to(dev)
    if world > 1:
        model.shard(0, world)
        model._get_engine().plan.layout.dry = True

    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-3, fused=True) if train else None

    def step():
        if train:
            opt.zero_grad()
            loss = model.loss(batch)
            loss.backward()
            opt.step()
            return loss
        with torch.no_grad():
            return model.loss(batch)

    for _ in range(50):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('world %d: enqueue %.1f us/step, enqueue + drain %.1f us/step' % (world, t_enq / steps * 1e6, t_all / steps * 1e6))
    # the loop above runs into the runtime's queue limit (the host is throttled to the GPU's pace once a few hundred launches
    # are outstanding): the host's OWN cost per step is what a short burst into an empty queue takes
    burst, best = (1 if train else 8), 1e9        # a training step is ~150 launches: one step per burst stays under the queue limit
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(burst):
            step()
        best = min(best, (time.perf_counter() - t0) / burst)
        torch.cuda.synchronize()
    print('world %d: host cost %.1f us/step (best of 20 bursts of %d steps into an empty queue)' % (world, best * 1e6, burst))
    if train:     # run the backward on this thread so that cProfile sees PEALossFunction.backward and what it calls
        torch.autograd.set_multithreading_enabled(False)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(60 if train else 22)


if __name__ == '__main__':
    main()
