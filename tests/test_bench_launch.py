"""bench.py --gpus N self-launch supervision (CPU): a rank that exits non-zero takes the job down at once, with its
exit code, instead of leaving its peers in a collective until a watchdog fires."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _job(code):
    return ([sys.executable, '-c', code], dict(os.environ))


def test_failed_rank_takes_the_job_down_quickly(tmp_path):
    marks = [str(tmp_path / ('alive%d' % r)) for r in range(3)]
    hang = "import sys,time,pathlib; pathlib.Path(sys.argv[1]).write_text('up'); time.sleep(120)"
    jobs = [([sys.executable, '-c', hang, marks[0]], dict(os.environ)),
            _job('import time,sys; time.sleep(0.5); sys.exit(7)'),          # rank 1 dies; ranks 0 and 2 "sit in a collective"
            ([sys.executable, '-c', hang, marks[2]], dict(os.environ))]
    t0 = time.time()
    rc = bench.launch_ranks(jobs, grace_s=2.0)
    dt = time.time() - t0
    assert rc == 7
    assert dt < 20, 'supervision waited %.1f s for peers of a dead rank' % dt
    assert os.path.exists(marks[0]) and os.path.exists(marks[2])       # the peers had really started


def test_signalled_rank_reports_nonzero():
    rc = bench.launch_ranks([_job('import os,signal; os.kill(os.getpid(), signal.SIGKILL)'), _job('import time; time.sleep(60)')],
                            grace_s=2.0)
    assert rc != 0


def test_all_ranks_ok():
    assert bench.launch_ranks([_job('pass'), _job('import time; time.sleep(0.2)')]) == 0


def test_metric_string_follows_the_workload():
    a = bench.parse([])
    assert bench.metric_name(a) == 'BPR-scored edges/sec, PEAGAT MovieLens-25m, emb_dim=64, 9 metapaths'
    b = bench.parse(['--preset', 'yelp_shaped', '--kind', 'sage'])
    assert 'yelp_shaped' in bench.metric_name(b) and 'PEASAGE' in bench.metric_name(b)
    assert bench.metric_name(b) != bench.metric_name(a)
