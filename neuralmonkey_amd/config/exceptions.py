"""What configuration parsing and loading raise.  The three public classes carry the attributes and print the texts
a caller of the reference sees (neuralmonkey/config/exceptions.py); the texts are pinned by the reference-executed
fixtures "ini_grammar" and "config_builder" under tests/golden."""
import traceback


class _ConfigError(Exception):
    """Named fields, kept as attributes, and one line of text made from them on demand."""
    fields = ()

    def __init__(self, *values) -> None:
        Exception.__init__(self)
        values = values + (None,) * (len(self.fields) - len(values))
        for field, value in zip(self.fields, values):
            setattr(self, field, value)

    def describe(self) -> str:
        raise NotImplementedError

    def __str__(self) -> str:
        return self.describe()


class ParseError(_ConfigError):
    """A line of an INI file that the value grammar does not accept; ``line`` is filled in by the file parser."""
    fields = ("message", "line")

    def set_line(self, line) -> None:
        self.line = line

    def describe(self) -> str:
        where = "parsing error" if self.line is None else "error on line {}".format(self.line)
        return "INI {}: {}".format(where, self.message)


class ConfigInvalidValueException(_ConfigError):
    """A section that cannot be turned into an object (undefined, without a class, not callable)."""
    fields = ("value", "message")

    def describe(self) -> str:
        return "Error in configuration of {0.value}: {0.message}".format(self)


class ConfigBuildException(_ConfigError):
    """Whatever went wrong while the object named ``object_name`` was being built, kept in ``original_exception``."""
    fields = ("object_name", "original_exception")

    def describe(self) -> str:
        frames = traceback.extract_tb(self.original_exception.__traceback__)
        return "Error while loading '{}': {}\nTraceback: {}".format(
            self.object_name, self.original_exception, "".join(traceback.format_list(frames)))
