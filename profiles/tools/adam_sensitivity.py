"""How far do two Adam trajectories of the SAME model drift apart when only the fp32 summation order of the weight-gradient
reduction changes?  python profiles/tools/adam_sensitivity.py KIND   (run under different PEA_GW_PARTS values)
Prints the loss of the first 8 steps of tests/test_gpu_training_converges.py's setup.  Adam divides by sqrt(v): an entry whose
gradient is rounding noise still moves by ~lr per step, so last-bit differences grow geometrically over the steps -- the
evidence behind that test's per-step bound."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import test_gpu_training_converges as T  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'gcn'
model, batch = T._setup(kind, 'mean' if kind == 'sage' else 'att')
print(kind, 'PEA_GW_PARTS=%s' % os.environ.get('PEA_GW_PARTS'), ' '.join('%.5f' % v for v in T._train(model, batch, 8)), flush=True)
