"""CPU: state_dict compatibility with the six checkpoints the reference ships
(experiments/checkpoint/weights/Movielenslatest-small/{PEAGAT,PEAGCN,PEASage}/BPR/*/run_1/latest.pkl;
format utils/general_utils.py:40-53).  The manifest (keys, shapes, dtypes) is a committed fixture; when the
reference tree is present the real files are also loaded (weights_only=True) with strict=True."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, build_model

REF_ROOT = '/root/reference/experiments/checkpoint/weights/Movielenslatest-small'
KIND = {'PEAGAT': 'gat', 'PEAGCN': 'gcn', 'PEASage': 'sage'}
LEGACY = (('mpagcn_channels', 'pea_channels'), ('gcn_layers', 'gnn_layers'))


def _model(kind):
    empty = np.zeros((2, 0), np.int64)
    return build_model(kind, 2933, [[empty, empty]] * 9, [2] * 9, 64, 64, 16, device='cpu')


def _remap(k):
    for a, b in LEGACY:
        k = k.replace(a, b)
    return k


with open(os.path.join(GOLDEN, 'checkpoint_manifest.json')) as f:
    MANIFEST = json.load(f)


@pytest.mark.parametrize('entry', sorted(MANIFEST))
def test_state_dict_layout_matches_reference_checkpoints(entry):
    model = _model(KIND[entry.split('/')[0]])
    ours = {k: (list(v.shape), str(v.dtype)) for k, v in model.state_dict().items()}
    theirs = {_remap(k): (v['shape'], v['dtype']) for k, v in MANIFEST[entry]['keys'].items()}
    assert ours == theirs


@pytest.mark.skipif(not os.path.isdir(REF_ROOT), reason='reference tree not present (GPU box)')
@pytest.mark.parametrize('name', sorted(KIND))
def test_reference_checkpoints_load_strict(name):
    import numpy._core.multiarray as ncm
    allow = [(ncm._reconstruct, 'numpy.core.multiarray._reconstruct'), (ncm.scalar, 'numpy.core.multiarray.scalar'),
             np.ndarray, np.dtype, type(np.dtype(np.float64))]
    base = os.path.join(REF_ROOT, name, 'BPR')
    for d in sorted(os.listdir(base)):
        with torch.serialization.safe_globals(allow):
            ck = torch.load(os.path.join(base, d, 'run_1', 'latest.pkl'), map_location='cpu', weights_only=True)
        sd = {_remap(k): v for k, v in ck['model_states']['model'].items()}
        model = _model(KIND[name])
        model.load_state_dict(sd, strict=True)
        assert int(ck['epoch']) == 30
        n_params = sum(p.numel() for p in model.parameters())
        assert n_params == {'PEAGAT': 236641, 'PEAGCN': 235201, 'PEASage': 281281}[name]
