"""What configuration parsing and loading raise: the attributes and the texts a caller of the reference sees
(neuralmonkey/config/exceptions.py); the texts are pinned by the reference-executed fixtures "ini_grammar" and
"config_builder" under tests/golden."""
import traceback


class ParseError(Exception):
    """A line of an INI file that the value grammar does not accept; ``line`` is filled in by the file parser."""

    def __init__(self, message, line=None) -> None:
        super().__init__()
        self.message = message
        self.line = line

    def set_line(self, line) -> None:
        self.line = line

    def __str__(self) -> str:
        where = "parsing error" if self.line is None else "error on line {}".format(self.line)
        return "INI {}: {}".format(where, self.message)


class ConfigInvalidValueException(Exception):
    """A section that cannot be turned into an object (undefined, without a class, not callable)."""

    def __init__(self, value, message) -> None:
        super().__init__()
        self.value = value
        self.message = message

    def __str__(self) -> str:
        return "Error in configuration of {}: {}".format(self.value, self.message)


class ConfigBuildException(Exception):
    """Whatever went wrong while the object named ``object_name`` was being built, kept in ``original_exception``."""

    def __init__(self, object_name, original_exception=None) -> None:
        super().__init__()
        self.object_name = object_name
        self.original_exception = original_exception

    def __str__(self) -> str:
        # (a value that is not an exception, or none at all, has no traceback: the text must still print)
        frames = traceback.extract_tb(getattr(self.original_exception, "__traceback__", None))
        return "Error while loading '{}': {}\nTraceback: {}".format(
            self.object_name, self.original_exception, "".join(traceback.format_list(frames)))
