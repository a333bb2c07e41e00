"""Exceptions of config parsing and loading (mirror of neuralmonkey/config/exceptions.py: same classes, same
attributes, same texts -- the text of a ParseError is pinned by the reference-executed fixture "ini_grammar" under tests/golden)."""
import traceback
from typing import Any


class ParseError(Exception):
    """A syntax error in an INI file (config/exceptions.py:7-23)."""

    def __init__(self, message: str, line: int = None) -> None:
        super().__init__()
        self.message = message
        self.line = line

    def set_line(self, line: int) -> None:
        self.line = line

    def __str__(self) -> str:
        if self.line is not None:
            return "INI error on line {}: {}".format(self.line, self.message)
        return "INI parsing error: {}".format(self.message)


class ConfigInvalidValueException(Exception):
    """config/exceptions.py:26-42."""

    def __init__(self, value: Any, message: str) -> None:
        super().__init__()
        self.value = value
        self.message = message

    def __str__(self) -> str:
        return "Error in configuration of {}: {}".format(self.value, self.message)


class ConfigBuildException(Exception):
    """An object of the configuration failed to build (config/exceptions.py:45-66)."""

    def __init__(self, object_name: str, original_exception: Exception) -> None:
        super().__init__()
        self.object_name = object_name
        self.original_exception = original_exception

    def __str__(self) -> str:
        trc = "".join(traceback.format_list(traceback.extract_tb(self.original_exception.__traceback__)))
        return "Error while loading '{}': {}\nTraceback: {}".format(self.object_name, self.original_exception, trc)
