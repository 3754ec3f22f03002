"""PEASAGE channel / model with the reference's constructor logic
(graph_recsys_benchmark/models/peasage.py:8-29): emb -> hidden (x heads) -> ... -> repr."""
import torch

from ..nn import SAGEConv
from .base import PEABaseChannel, PEABaseRecsysModel


class PEASageChannel(PEABaseChannel):
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
        return [SAGEConv(widths[s], widths[s + 1]) for s in range(len(widths) - 1)]


class PEASageRecsysModel(PEABaseRecsysModel):
    kind = 'sage'

    def __init__(self, **kwargs):
        kwargs['channel_class'] = PEASageChannel
        super().__init__(**kwargs)
