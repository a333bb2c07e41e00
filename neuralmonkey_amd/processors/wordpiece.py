"""Word pieces in the tensor2tensor fashion (mirror of neuralmonkey/processors/wordpiece.py:
``WordpiecePreprocessor(vocabulary)`` makes a sentence-level preprocessor, ``WordpiecePostprocessor``
is the batch-level inverse; both are used with ``from_t2t_vocabulary`` and ``T2TReader`` in the
reference's Transformer configurations).

A token is escaped, closed with ``_`` and cut greedily into the longest pieces the vocabulary holds,
left to right.  Characters outside the vocabulary's alphabet are written as ``\\<code point>;``.
The reference prepares ``\\\\`` / ``\\u`` escapes for backslashes and underscores but then builds the
escaped token from the unescaped characters (wordpiece.py:31-38), so in-alphabet backslashes and
underscores go through verbatim; this module does what the reference does, not what it prepares.
"""
import re
from typing import Callable, List, Set

from ..checking import check_argument_types
from ..vocabulary import Vocabulary

_ESCAPE = re.compile(r"\\u|\\\\|\\([0-9]+);")
_UNDEFINED = "〓"


def escape_token(token: str, alphabet: Set[str]) -> str:
    return "".join(c if c in alphabet and c != "\n" else "\\{};".format(ord(c)) for c in token) + "_"


def _unescape(match) -> str:
    code = match.group(1)
    if code is None:
        return "_" if match.group(0) == "\\u" else "\\"
    try:
        return chr(int(code))
    except (ValueError, OverflowError):
        return _UNDEFINED


def unescape_token(escaped_token: str) -> str:
    body = escaped_token[:-1] if escaped_token.endswith("_") else escaped_token
    return _ESCAPE.sub(_unescape, body)


def _longest_pieces(escaped: str, vocabulary: Vocabulary) -> List[str]:
    pieces, start = [], 0
    while start < len(escaped):
        end = next((e for e in range(len(escaped), start, -1) if escaped[start:e] in vocabulary), None)
        if end is None:
            raise AssertionError("No token substring found in the vocab ({}).".format(escaped[start:]))
        pieces.append(escaped[start:end])
        start = end
    return pieces


def wordpiece_encode(sentence: List[str], vocabulary: Vocabulary) -> List[str]:
    alphabet = vocabulary.alphabet
    return [piece for token in sentence for piece in _longest_pieces(escape_token(token, alphabet), vocabulary)]


def wordpiece_decode(sentence: List[str]) -> List[str]:
    """Glue the pieces, cut at the underscores, unescape; empty tokens vanish."""
    tokens = (unescape_token(chunk) for chunk in "".join(sentence).split("_") if chunk)
    return [token for token in tokens if token]


def wordpiece_decode_batch(sentences: List[List[str]]) -> List[List[str]]:
    return [wordpiece_decode(sentence) for sentence in sentences]


def get_wordpiece_preprocessor(vocabulary: Vocabulary) -> Callable[[List[str]], List[str]]:
    check_argument_types()
    return lambda sentence: wordpiece_encode(sentence, vocabulary)


# pylint: disable=invalid-name
WordpiecePreprocessor = get_wordpiece_preprocessor
WordpiecePostprocessor = wordpiece_decode_batch
# pylint: enable=invalid-name
