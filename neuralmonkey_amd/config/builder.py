"""Object builder for parsed configs (semantics of neuralmonkey/config/builder.py).

``class=`` names resolve like the reference (builder.py:25-58): first as an
importable module path, then ``tf.*`` against the TensorFlow look-alike
namespace ``neuralmonkey_amd.tf_shim``, then relative to this package (the
reference prefixes ``neuralmonkey.``; that prefix is accepted too).  A missing
``name`` argument defaults to the section name when the callable declares
``name: str`` (builder.py:170-185)."""
import collections
import collections.abc
import importlib
from argparse import Namespace
from inspect import isclass, isfunction, signature
from typing import Any, Dict, Set, Tuple

from .exceptions import ConfigBuildException, ConfigInvalidValueException
from .parsing import ClassSymbol, ObjectRef

PACKAGE = __name__.rsplit(".", 2)[0]          # "neuralmonkey_amd"

# [main] keys whose object graphs belong to the reference's host control plane (out of
# scope: evaluators, series post-processing).  An INI written for the reference names classes there that
# this package does not ship; such a class -- and only under these keys -- builds into an
# ``OutOfScope`` placeholder so that the file loads unmodified (experiment.py:60-120 lists the keys).
SOFT_MAIN_KEYS = frozenset(["evaluation", "postprocess"])


class OutOfScope:
    """Placeholder for an object of the reference's control plane (evaluator, postprocessor) that an
    unmodified INI file names.  Remembers what was asked for; calling it is an error."""

    def __init__(self, symbol: str, arguments: Dict[str, Any] = None) -> None:
        self.symbol = symbol
        self.arguments = dict(arguments or {})
        self.name = self.arguments.get("name", symbol.rsplit(".", 1)[-1])

    def __call__(self, *args, **kwargs):
        raise NotImplementedError("{} belongs to the reference's host control plane and is not part of "
                                  "the MI355X engine".format(self.symbol))

    def __repr__(self):
        return "OutOfScope({})".format(self.symbol)


class _Soft:
    """Build context of one [main] key: ``on`` while unresolvable classes may become placeholders."""
    on = False


# namespaces of the reference's host control plane: only names in here may turn into placeholders
SOFT_NAMESPACES = frozenset(["evaluators", "processors"])


class SymbolNotShipped(Exception):
    """The dotted name points at a module or attribute this package does not have (as opposed to a shipped
    module that failed while being imported)."""


def resolve_soft(dotted: str) -> Any:
    """``resolve_symbol`` under a SOFT_MAIN_KEYS key: a class of the reference's evaluators / processors that
    is not shipped becomes an ``OutOfScope`` placeholder (with a warning naming it); everything else -- other
    namespaces, import errors inside a shipped module -- raises as usual."""
    parts = dotted.split(".")
    namespace = parts[1] if parts[0] == "neuralmonkey" and len(parts) > 1 else parts[0]
    try:
        return resolve_symbol(dotted)
    except SymbolNotShipped as exc:
        if namespace not in SOFT_NAMESPACES:
            raise
        import warnings
        warnings.warn("{} is not part of this engine (host control plane of the reference): placeholder built "
                      "({})".format(dotted, exc), stacklevel=2)
        return None


def resolve_symbol(dotted: str) -> Any:
    parts = dotted.split(".")
    attr, module_path = parts[-1], ".".join(parts[:-1])
    candidates = [module_path]
    if parts[0] == "tf":
        candidates = [PACKAGE + ".tf_shim"]
    else:
        if parts[0] == "neuralmonkey":
            candidates.append(".".join([PACKAGE] + parts[1:-1]))
        candidates.append(".".join([PACKAGE] + parts[:-1]))
    last_exc = None
    for cand in candidates:
        try:
            module = importlib.import_module(cand)
        except ModuleNotFoundError as exc:
            if exc.name is None or not (cand == exc.name or cand.startswith(exc.name + ".")):
                raise                          # the module exists, something IT imports is missing
            last_exc = exc
            continue
        if parts[0] == "tf":
            obj = module
            for piece in parts[1:]:
                obj = getattr(obj, piece)
            return obj
        try:
            return getattr(module, attr)
        except AttributeError as exc:
            raise SymbolNotShipped("Interpretation '{}' as type name, class '{}' does not exist. "
                                   "Did you mean file './{}'? \n{}".format(dotted, attr, dotted, exc))
    raise SymbolNotShipped("Cannot import module {} ({})".format(module_path, last_exc))


def build_object(value: Any, all_dicts: Dict[str, Any], existing: Dict[str, Any], depth: int) -> Any:
    if depth > 20:
        raise AssertionError("Config recursion should not be deeper that 20.")
    if isinstance(value, tuple):
        return tuple(build_object(v, all_dicts, existing, depth + 1) for v in value)
    if isinstance(value, collections.abc.Iterable) and not isinstance(value, (str, bytes, dict)):
        return [build_object(v, all_dicts, existing, depth + 1) for v in value]
    if isinstance(value, ObjectRef):
        if value.name not in existing:
            existing[value.name] = instantiate_class(value.name, all_dicts, existing, depth)
        value.bind(existing[value.name])
        return value.target
    if isinstance(value, ClassSymbol):
        if _Soft.on:
            obj = resolve_soft(value.clazz)
            return OutOfScope(value.clazz) if obj is None else obj
        return resolve_symbol(value.clazz)
    return value


def instantiate_class(name: str, all_dicts: Dict[str, Any], existing: Dict[str, Any], depth: int) -> Any:
    if name not in all_dicts:
        raise ConfigInvalidValueException(name, "Undefined object")
    section = all_dicts[name]
    if "class" not in section:
        raise ConfigInvalidValueException(name, "Undefined object type")
    if _Soft.on:
        clazz = resolve_soft(section["class"].clazz)
        if clazz is None:
            return OutOfScope(section["class"].clazz,
                              {k: v for k, v in section.items() if k != "class" and isinstance(v, (str, int, float))})
    else:
        clazz = resolve_symbol(section["class"].clazz)
    if not isclass(clazz) and not isfunction(clazz):
        raise ConfigInvalidValueException(name, "Cannot instantiate object with '{}'".format(clazz))
    arguments = {key: build_object(val, all_dicts, existing, depth + 1)
                 for key, val in section.items() if key != "class"}
    sig = signature(clazz)
    if "name" in sig.parameters and "name" not in arguments:
        if sig.parameters["name"].annotation in (str, "str"):
            arguments["name"] = name
    try:
        bound = sig.bind(**arguments)
    except TypeError as exc:
        raise ConfigBuildException(clazz, exc)
    return clazz(*bound.args, **bound.kwargs)


def build_config(config_dicts: Dict[str, Any], ignore_names: Set[str],
                 warn_unused: bool = False) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    """builder.py:207-249: build every key of [main] (sorted, tf_manager last)."""
    if "main" not in config_dicts:
        raise Exception("Configuration does not contain the main block.")
    existing: Dict[str, Any] = collections.OrderedDict()
    main = config_dicts["main"]
    existing["main"] = Namespace(**main)
    configuration: Dict[str, Any] = collections.OrderedDict()
    for key in sorted(main, key=lambda k: "zzz" if k == "tf_manager" else k):
        if key in ignore_names:
            continue
        _Soft.on = key in SOFT_MAIN_KEYS
        try:
            configuration[key] = build_object(main[key], config_dicts, existing, 0)
        except Exception as exc:
            raise ConfigBuildException(key, exc) from None
        finally:
            _Soft.on = False
    return configuration, existing
