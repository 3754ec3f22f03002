from .base import GraphRecsysModel, PEABaseChannel, PEABaseRecsysModel
from .peagat import PEAGATChannel, PEAGATRecsysModel
from .peagcn import PEAGCNChannel, PEAGCNRecsysModel
from .peasage import PEASageChannel, PEASageRecsysModel

__all__ = ['GraphRecsysModel', 'PEABaseChannel', 'PEABaseRecsysModel', 'PEAGATChannel', 'PEAGATRecsysModel',
           'PEAGCNChannel', 'PEAGCNRecsysModel', 'PEASageChannel', 'PEASageRecsysModel']
