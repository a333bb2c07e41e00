from .recurrent import RecurrentEncoder, SentenceEncoder   # noqa: F401
