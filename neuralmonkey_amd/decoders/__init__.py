from .decoder import Decoder                             # noqa: F401
from .beam_search_decoder import BeamSearchDecoder       # noqa: F401
from .transformer import TransformerDecoder             # noqa: F401
