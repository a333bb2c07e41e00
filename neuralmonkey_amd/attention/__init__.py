from .feed_forward import Attention   # noqa: F401
from .coverage import CoverageAttention   # noqa: F401
from .combination import FlatMultiAttention, HierarchicalMultiAttention   # noqa: F401
from .scaled_dot_product import MultiHeadAttention, ScaledDotProdAttention   # noqa: F401
from .stateful_context import StatefulContext   # noqa: F401
