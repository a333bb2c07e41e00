"""INI value grammar of Neural Monkey configs (semantics of
neuralmonkey/config/parsing.py:14-32,152-252), written as a small
recursive-descent reader instead of a regex table.

Values: integers, floats (with exponents), "strings" with ``{var}``
substitution, ``$var`` references, ``<object.attr>`` references, dotted class
names, ``[lists]``, ``(tuples)``, True / False / None.  A ``[vars]`` section is
read first; undefined variables fall back to the environment; ``section.key=value``
overrides are applied before parsing; ``{TIME}`` is predefined."""
import configparser
import os
import re
import time
from collections import OrderedDict
from typing import Any, Dict, Iterable, List, Optional, Tuple

from .exceptions import ParseError

_INT = re.compile(r"-?[0-9]+\Z")
_FLOAT = re.compile(r"-?[0-9]*\.[0-9]*(e[+-]?[0-9]+)?\Z|-?[0-9]+e[+-]?[0-9]+\Z")
_IDENT = r"[a-zA-Z][a-zA-Z0-9_]*"
_VAR = re.compile(r"\$(" + _IDENT + r")\Z")
_OBJ = re.compile(r"<(" + _IDENT + r"(\." + _IDENT + r")*)>\Z")
_CLASS = re.compile(r"_*[a-zA-Z][a-zA-Z0-9_]*(\._*[a-zA-Z][a-zA-Z0-9_]*)+\Z")
_KEYWORDS = {"False": False, "True": True, "None": None}


class ClassSymbol:
    """A dotted name to be imported by the builder."""

    def __init__(self, string: str) -> None:
        self.clazz = string

    def __repr__(self):
        return "ClassSymbol({})".format(self.clazz)


class ObjectRef:
    """``<section>`` or ``<section.attr.chain>``."""

    def __init__(self, expression: str) -> None:
        self.expression = expression
        parts = expression.split(".")
        self.name, self.attr_chain = parts[0], parts[1:]
        self._obj = None

    def bind(self, value: Any) -> None:
        self._obj = value

    @property
    def target(self) -> Any:
        value = self._obj
        for attr in self.attr_chain:
            value = getattr(value, attr)
        return value

    def __repr__(self):
        return "ObjectRef({})".format(self.expression)


class VarsDict(OrderedDict):
    def __missing__(self, key):
        if key in os.environ:
            raw = os.environ[key]
            try:
                return parse_value(raw, self)
            except ParseError:
                return raw
        raise ParseError("Undefined variable: {}".format(key))


def _split_top_level(body: str) -> List[str]:
    """Split on commas that are not nested in brackets / parentheses / quotes."""
    items, depth, cur, quoted = [], [], [], False
    for pos, ch in enumerate(body):
        if ch == '"':
            quoted = not quoted
        if not quoted:
            if ch in "([":
                depth.append(ch)
            elif ch in ")]":
                want = "(" if ch == ")" else "["
                if not depth or depth.pop() != want:
                    raise ParseError("Invalid bracket end '{}', col {}.".format(ch, pos))
            elif ch == "," and not depth:
                piece = "".join(cur).strip()
                if piece:
                    items.append(piece)
                cur = []
                continue
        cur.append(ch)
    if depth:
        raise ParseError("Unclosed bracket in '{}'".format(body))
    piece = "".join(cur).strip()
    if piece:
        items.append(piece)
    return items


def parse_value(text: str, variables: VarsDict) -> Any:
    text = text.strip()
    if text in _KEYWORDS:
        return _KEYWORDS[text]
    if _INT.match(text):
        return int(text)
    if _FLOAT.match(text):
        return float(text)
    if len(text) >= 2 and text[0] == '"' and text[-1] == '"':
        return text[1:-1].format_map(variables)
    found = _VAR.match(text)
    if found:
        return variables[found.group(1)]
    found = _OBJ.match(text)
    if found:
        return ObjectRef(found.group(1))
    if _CLASS.match(text):
        return ClassSymbol(text)
    if text.startswith("[") and text.endswith("]"):
        return [parse_value(item, variables) for item in _split_top_level(text[1:-1])]
    if text.startswith("(") and text.endswith(")"):
        return tuple(parse_value(item, variables) for item in _split_top_level(text[1:-1]))
    raise ParseError("Cannot parse value: '{}'.".format(text))


# the names under which the reference's own unit tests reach the value grammar (neuralmonkey/tests/test_config.py)
_parse_value = parse_value
_split_on_commas = _split_top_level


def _read_ini(lines: Iterable[str]) -> "OrderedDict[str, OrderedDict]":
    """Sections -> key -> (line number, raw text).  configparser joins
    continuation lines; the line number of a key is that of its first line."""
    numbered = []
    for num, line in enumerate(lines):
        stripped = line.strip()
        numbered.append("{} \x00{}".format(stripped, num + 1) if stripped else "")
    parser = configparser.ConfigParser(interpolation=None)
    parser.optionxform = str.lower
    parser.read_file(numbered)
    out: "OrderedDict[str, OrderedDict]" = OrderedDict()
    for section in parser.sections():
        out[section] = OrderedDict()
        for key, raw in parser[section].items():
            pieces = raw.split("\n")
            first_no = None
            cleaned = []
            for piece in pieces:
                if "\x00" in piece:
                    body, num = piece.rsplit("\x00", 1)
                    first_no = first_no or num
                    cleaned.append(body.rstrip())
                else:
                    cleaned.append(piece)
            out[section][key] = (first_no, "\n".join(cleaned).strip())
    return out


def _apply_change(config: Dict[str, Any], setting: str) -> None:
    if "=" not in setting:
        raise ParseError("Invalid setting '{}'".format(setting))
    key, value = (s.strip() for s in setting.split("=", 1))
    section, option = key.split(".", 1) if "." in key else ("main", key)
    config.setdefault(section, OrderedDict())[option] = (None, value)


def parse_file(config_file: Iterable[str], changes: Optional[Iterable[str]] = None
               ) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    """Returns (raw strings per section, parsed values per section)."""
    config = _read_ini(config_file)
    for change in changes or []:
        _apply_change(config, change)
    variables = VarsDict()
    variables["TIME"] = time.strftime("%Y-%m-%d-%H-%M-%S")

    def parse_section(section: str, into: Dict[str, Any]) -> None:
        for key, (lineno, raw) in config[section].items():
            try:
                into[key] = parse_value(raw, variables)
            except ParseError as exc:
                exc.set_line(lineno)
                raise

    if "vars" in config:
        parse_section("vars", variables)
    parsed: "OrderedDict[str, Any]" = OrderedDict()
    for section in config:
        if section != "vars":
            parsed[section] = OrderedDict()
            parse_section(section, parsed[section])
    raw = OrderedDict((name, OrderedDict((k, v) for k, (_, v) in sec.items()))
                      for name, sec in config.items())
    return raw, parsed
