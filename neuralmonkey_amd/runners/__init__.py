from .runner import GreedyRunner                                                # noqa: F401
from .beamsearch_runner import BeamSearchRunner, beam_search_runner_range       # noqa: F401
from .plain_runner import PlainRunner                                           # noqa: F401
from .xent_runner import XentRunner                                             # noqa: F401
from .tensor_runner import RepresentationRunner, TensorRunner                   # noqa: F401
