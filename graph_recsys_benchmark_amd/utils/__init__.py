from .graph_input import metapath_table, update_pea_graph_input
from .sampling import cf_negative_sampling, device_negative_sampling, entity_aware_row, generate_candidates
from .synthetic import PRESETS, SyntheticHIN

__all__ = ['metapath_table', 'update_pea_graph_input', 'PRESETS', 'SyntheticHIN', 'cf_negative_sampling',
           'device_negative_sampling', 'entity_aware_row', 'generate_candidates']
