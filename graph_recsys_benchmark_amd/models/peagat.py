"""PEAGAT channel / model with the reference's constructor logic
(graph_recsys_benchmark/models/peagat.py:8-29): emb -> hidden (x heads) -> ... -> repr."""
import torch

from ..nn import GATConv
from .base import PEABaseChannel, PEABaseRecsysModel


class PEAGATChannel(PEABaseChannel):
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
        heads, p = kwargs['num_heads'], kwargs['dropout']
        n = len(widths) - 1
        layers = []
        for s in range(n):
            last_of_many = n > 1 and s == n - 1
            in_w = widths[s] * (heads if s > 0 else 1)
            layers.append(GATConv(in_w, widths[s + 1], heads=1 if last_of_many else heads, dropout=p))
        return layers


class PEAGATRecsysModel(PEABaseRecsysModel):
    kind = 'gat'

    def __init__(self, **kwargs):
        kwargs['channel_class'] = PEAGATChannel
        super().__init__(**kwargs)
