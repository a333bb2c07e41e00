from .feed_forward import Attention   # noqa: F401
from .combination import FlatMultiAttention, HierarchicalMultiAttention   # noqa: F401
