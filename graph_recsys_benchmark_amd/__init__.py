"""graph_recsys_benchmark_amd -- MI355X-native metapath-GNN aggregation path behind the PEAGNN module surface.

Only the hot path of ecml-peagnn/graph_recsys_benchmark is here: conv layers, the PEA
channel loop + fusion, BPR scoring.  Everything computes in csrc/libpeahip.so (hand-written HIP for gfx950).
"""
from . import _lib  # noqa: F401

__version__ = '0.1.0'
