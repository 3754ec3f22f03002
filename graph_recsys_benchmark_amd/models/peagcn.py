"""PEAGCN channel / model with the reference's constructor logic
(graph_recsys_benchmark/models/peagcn.py:8-29): emb -> hidden (x heads) -> ... -> repr."""
import torch

from ..nn import GCNConv
from .base import PEABaseChannel, PEABaseRecsysModel


class PEAGCNChannel(PEABaseChannel):
    def __init__(self, **kwargs):
        super().__init__()
        self.num_steps = kwargs['num_steps']
        self.num_nodes = kwargs['num_nodes']
        self.dropout = kwargs['dropout']
        widths = [kwargs['emb_dim']] + [kwargs['hidden_size']] * (self.num_steps - 1) + [kwargs['repr_dim']]
        self.gnn_layers = torch.nn.ModuleList(self._make_layers(widths, kwargs))
        self.reset_parameters()

    @staticmethod
    def _make_layers(widths, kwargs):
        deg = kwargs.get('gcn_deg_from', 'row')
        return [GCNConv(widths[s], widths[s + 1], gcn_deg_from=deg) for s in range(len(widths) - 1)]


class PEAGCNRecsysModel(PEABaseRecsysModel):
    kind = 'gcn'

    def __init__(self, **kwargs):
        kwargs['channel_class'] = PEAGCNChannel
        super().__init__(**kwargs)
