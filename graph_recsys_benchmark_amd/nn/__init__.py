from .conv import GATConv, GCNConv, SAGEConv
from .inits import glorot, zeros

__all__ = ['GATConv', 'GCNConv', 'SAGEConv', 'glorot', 'zeros']
