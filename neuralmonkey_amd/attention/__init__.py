from .feed_forward import Attention   # noqa: F401
