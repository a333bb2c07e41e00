"""Configuration errors (same names as neuralmonkey/config/exceptions.py)."""


class ParseError(Exception):
    def __init__(self, message: str, line: int = None) -> None:
        super().__init__(message)
        self.message = message
        self.line = line

    def set_line(self, line) -> None:
        self.line = line

    def __str__(self) -> str:
        if self.line is not None:
            return "line {}: {}".format(self.line, self.message)
        return self.message


class ConfigInvalidValueException(Exception):
    def __init__(self, value, message) -> None:
        super().__init__("Error in configuration of {}: {}".format(value, message))


class ConfigBuildException(Exception):
    def __init__(self, object_name, original_exception) -> None:
        super().__init__("Error while building object \"{}\": {}: {}".format(
            object_name, type(original_exception).__name__, original_exception))
        self.original_exception = original_exception
