from .conv import GATConv, GCNConv, SAGEConv
from .inits import glorot, zeros
from .kg_conv import KGATConv, KGCNConv, NGCFConv, weighted_aggregate

__all__ = ['GATConv', 'GCNConv', 'SAGEConv', 'KGATConv', 'KGCNConv', 'NGCFConv', 'weighted_aggregate', 'glorot', 'zeros']
