from .recurrent import FactoredEncoder, RecurrentEncoder, SentenceEncoder   # noqa: F401
from .numpy_stateful_filler import SpatialFiller, StatefulFiller          # noqa: F401
from .transformer import TransformerEncoder             # noqa: F401
