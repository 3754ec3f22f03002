"""Shared test helpers: golden-fixture loading and oracle-side model evaluation."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'pea_*.npz')))


class GoldenCase:
    """One fixture written by oracle/make_golden.py (inputs + outputs of the reference's models/base.py)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + '.npz'))
        self.name = name
        self.meta = json.loads(bytes(z['meta']).decode())
        self.kind = self.meta['kind']
        self.steps = self.meta['steps']
        self.P = len(self.steps)
        self.heads = self.meta['heads']
        self.state_dict = {k[len('param/'):]: z[k] for k in z.files if k.startswith('param/')}
        self.edges = [[z['edge/%d/%d' % (p, s)] for s in range(self.steps[p])] for p in range(self.P)]
        self.out = {k[len('out/'):]: z[k] for k in z.files if k.startswith('out/')}
        self.batch = z['in/batch']
        self.batch9 = z['in/batch9'] if 'in/batch9' in z.files else None

    def layer_params(self, p, s):
        pre = 'pea_channels.%d.gnn_layers.%d.' % (p, s)
        return {k[len(pre):]: v for k, v in self.state_dict.items() if k.startswith(pre)}

    def channel_params(self):
        return [[self.layer_params(p, s) for s in range(self.steps[p])] for p in range(self.P)]

    def heads_lists(self):
        """PEAGATChannel: every layer uses num_heads except the last of a multi-step channel,
        which always uses heads=1 (models/peagat.py:16-21)."""
        out = []
        for p in range(self.P):
            S = self.steps[p]
            if self.kind != 'gat':
                out.append([1] * S)
            elif S == 1:
                out.append([self.heads])
            else:
                out.append([self.heads] * (S - 1) + [1])
        return out
