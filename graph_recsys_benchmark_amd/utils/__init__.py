from .graph_input import metapath_table, update_pea_graph_input
from .synthetic import PRESETS, SyntheticHIN

__all__ = ['metapath_table', 'update_pea_graph_input', 'PRESETS', 'SyntheticHIN']
