"""Argument type checks of the plugin constructors.

The reference's model parts start their ``__init__`` with typeguard's ``check_argument_types()``
(e.g. neuralmonkey/encoders/recurrent.py:121, decoders/decoder.py:122, attention/base_attention.py:159),
so a configuration that hands a string where an integer is expected dies when the object is
built, with a TypeError, not in the middle of the first batch.  neuralmonkey/tests/test_encoders_init.py
tabulates that behaviour.  typeguard is not a dependency here; this module checks the annotations
the constructors of this package use: plain classes, Optional / Union, List / Tuple / Dict /
Set / Sequence / Iterable of those, Callable, Any and forward references (by class name).

Like PEP 484 (and typeguard) an int is accepted where a float is annotated; unlike isinstance a
bool is NOT accepted for an int or float (the INI grammar has separate literals for them).
"""
import collections.abc
import gc
import inspect
import numbers
import sys
import typing
from typing import Any, Optional


def _name_of(hint: Any) -> str:
    if typing.get_origin(hint) is not None:
        return str(hint).replace("typing.", "")
    return getattr(hint, "__name__", None) or str(hint).replace("typing.", "")


def _matches_forward(value: Any, name: str) -> bool:
    name = name.split(".")[-1].strip("'\"")
    return any(cls.__name__ == name for cls in type(value).__mro__)


# pylint: disable=too-many-return-statements,too-many-branches
def matches(value: Any, hint: Any) -> bool:
    """True when ``value`` is acceptable for the annotation ``hint``."""
    if hint is Any or hint is inspect.Parameter.empty or hint is None and value is None:
        return True
    if hint is None or hint is type(None):
        return value is None
    if isinstance(hint, str):
        return _matches_forward(value, hint)
    if isinstance(hint, typing.ForwardRef):
        return _matches_forward(value, hint.__forward_arg__)
    if isinstance(hint, typing.TypeVar):
        bound = hint.__bound__
        return True if bound is None else matches(value, bound)
    origin = typing.get_origin(hint)
    args = typing.get_args(hint)
    if origin is typing.Union:
        return any(matches(value, a) for a in args)
    if origin is collections.abc.Callable or hint is typing.Callable:
        return callable(value)
    if origin is type:
        return inspect.isclass(value) and (not args or isinstance(args[0], typing.TypeVar) or args[0] is Any
                                           or not inspect.isclass(args[0]) or issubclass(value, args[0]))
    if origin is tuple:
        if not isinstance(value, tuple):
            return False
        if not args:
            return True
        if len(args) == 2 and args[1] is Ellipsis:
            return all(matches(v, args[0]) for v in value)
        return len(value) == len(args) and all(matches(v, a) for v, a in zip(value, args))
    if origin in (list, set, frozenset, collections.abc.Sequence, collections.abc.Iterable,
                  collections.abc.Collection, collections.abc.MutableSequence, collections.abc.Set):
        if isinstance(value, (str, bytes)) and origin not in (collections.abc.Iterable, collections.abc.Sequence):
            return False
        if not isinstance(value, origin):
            return False
        if isinstance(value, (str, bytes)) or not args or not isinstance(value, collections.abc.Sized):
            return True                                                   # a generator is not consumed
        return all(matches(v, args[0]) for v in value)
    if origin in (dict, collections.abc.Mapping, collections.abc.MutableMapping):
        if not isinstance(value, origin):
            return False
        return not args or all(matches(k, args[0]) and matches(v, args[1]) for k, v in value.items())
    if origin is not None:                                                # some other generic: check the container only
        return isinstance(value, origin) if inspect.isclass(origin) else True
    if hint is float:                                                     # numpy scalars count as numbers
        return isinstance(value, numbers.Real) and not isinstance(value, bool)
    if hint is int:
        return isinstance(value, numbers.Integral) and not isinstance(value, bool)
    if hint is complex:
        return isinstance(value, (int, float, complex)) and not isinstance(value, bool)
    if inspect.isclass(hint):
        return isinstance(value, hint)
    if hasattr(hint, "__supertype__"):                                    # typing.NewType
        return matches(value, hint.__supertype__)
    return True


def _function_of(frame) -> Optional[Any]:
    """The function object whose body ``frame`` executes: a method of the first argument's class,
    or a module-level function."""
    code = frame.f_code
    if code.co_argcount:
        first = frame.f_locals.get(code.co_varnames[0])
        owner = first if inspect.isclass(first) else type(first)
        for cls in owner.__mro__:
            cand = cls.__dict__.get(code.co_name)
            cand = getattr(cand, "__func__", cand)
            if getattr(cand, "__code__", None) is code:
                return cand
    cand = frame.f_globals.get(code.co_name)
    if getattr(cand, "__code__", None) is code:
        return cand
    for cand in gc.get_referrers(code):                                   # a nested function or a static method
        if inspect.isfunction(cand) and cand.__code__ is code:
            return cand
    return None


def _hints(func) -> dict:
    try:
        return typing.get_type_hints(func)
    except Exception:                                                     # pylint: disable=broad-except
        # an unresolvable forward reference (a class imported under TYPE_CHECKING only): fall back
        # to the raw annotations, strings are then matched by class name
        module = sys.modules.get(func.__module__)
        scope = dict(vars(typing), **(vars(module) if module else {}))
        hints = {}
        for key, raw in getattr(func, "__annotations__", {}).items():
            if isinstance(raw, str):
                try:
                    raw = eval(raw, scope)                                # pylint: disable=eval-used
                except Exception:                                         # pylint: disable=broad-except
                    pass
            hints[key] = raw
        return hints


def _check_frame(frame, func) -> None:
    hints = _hints(func)
    code = frame.f_code
    nargs = code.co_argcount + code.co_kwonlyargcount
    defaults = {k: v.default for k, v in inspect.signature(func).parameters.items()}
    for argname in code.co_varnames[:nargs]:
        if argname not in hints or argname not in frame.f_locals:
            continue
        value = frame.f_locals[argname]
        if value is None and defaults.get(argname, inspect.Parameter.empty) is None:
            continue                                                      # ``x: int = None`` means Optional[int]
        if not matches(value, hints[argname]):
            raise TypeError("type of argument \"{}\" must be {}; got {} instead"
                            .format(argname, _name_of(hints[argname]), type(value).__qualname__))


def check_constructor_chain(obj: Any) -> None:
    """Check the arguments of every ``__init__`` of ``obj`` that is on the call stack right now.

    The base classes of the plugins (``Parameterized``, ``GraphExecutor``) call this first thing, so
    each subclass constructor -- whose first statement is its parent's ``__init__`` -- is checked
    without repeating the call in forty places.  The innermost frame checked is the caller's."""
    frame = inspect.currentframe().f_back
    try:
        while frame is not None and frame.f_code.co_name == "__init__" and frame.f_code.co_argcount \
                and frame.f_locals.get(frame.f_code.co_varnames[0]) is obj:
            func = _function_of(frame)
            if func is not None:
                _check_frame(frame, func)
            frame = frame.f_back
    finally:
        del frame


def check_argument_types() -> bool:
    """Check the arguments of the calling function against its annotations; TypeError on the
    first mismatch (typeguard's message format)."""
    frame = inspect.currentframe().f_back
    try:
        func = _function_of(frame)
        if func is not None:
            _check_frame(frame, func)
        return True
    finally:
        del frame
