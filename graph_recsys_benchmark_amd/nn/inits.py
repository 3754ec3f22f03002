"""Parameter initialisers with the semantics of torch_geometric.nn.inits (release 1.5.0), which the
reference imports at graph_recsys_benchmark/models/base.py:5 and uses at :181-189."""
import math


def glorot(tensor):
    if tensor is not None:
        stdv = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        tensor.data.uniform_(-stdv, stdv)


def zeros(tensor):
    if tensor is not None:
        tensor.data.fill_(0)
