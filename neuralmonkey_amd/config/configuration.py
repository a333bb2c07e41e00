"""Loading an experiment INI into built objects (subset of
neuralmonkey/config/configuration.py + experiment.py:176-227: parse, build the
model, give every feedable its inputs, initialise the sessions)."""
from argparse import Namespace
from typing import Any, Dict, Iterable, List, Optional

from .builder import build_config
from .parsing import parse_file


class Configuration:
    def __init__(self) -> None:
        self.args = Namespace()
        self.model: Optional[Namespace] = None
        self.config_dict: Dict[str, Any] = {}
        self.raw_config: Dict[str, Any] = {}
        self.defaults: Dict[str, Any] = {}
        self.ignored: set = set()
        self.objects: Dict[str, Any] = {}

    def add_argument(self, name: str, required: bool = False, default: Any = None) -> None:
        self.defaults[name] = default

    def ignore_argument(self, name: str) -> None:
        self.ignored.add(name)

    def load_file(self, path: str, changes: Optional[List[str]] = None) -> None:
        with open(path, "r", encoding="utf-8") as handle:
            self.raw_config, self.config_dict = parse_file(handle, changes)
        main = self.config_dict.get("main", {})
        self.args = Namespace(**{**self.defaults, **{k: v for k, v in main.items()}})

    def build_model(self, warn_unused: bool = False) -> Namespace:
        from .. import dataset
        dataset._MAIN_BATCH_SIZE.append(getattr(self.args, "batch_size", None))       # dataset.load's default scheme
        try:
            built, objects = build_config(self.config_dict, self.ignored, warn_unused)
        finally:
            dataset._MAIN_BATCH_SIZE.pop()
        self.objects = objects
        self.model = Namespace(**{**self.defaults, **built})
        return self.model


def load_experiment(path: str, changes: Optional[List[str]] = None, initialize: bool = True,
                    device: Optional[str] = None, seed: Optional[int] = None) -> Namespace:
    """INI file -> namespace of built objects with initialised variables.

    Mirrors Experiment.build_model (experiment.py:176-227): build the [main]
    objects, default the ``tf_manager`` (config/normalize.py:33-34), make
    ``runners`` / ``trainers`` lists, then create and initialise all variables."""
    from ..runtime import reset_registry
    from ..tf_manager import TensorFlowManager
    reset_registry()
    cfg = Configuration()
    cfg.load_file(path, changes)
    model = cfg.build_model()
    if getattr(model, "tf_manager", None) is None:
        model.tf_manager = TensorFlowManager(num_sessions=1, num_threads=4, device=device, seed=seed)
    trainer = getattr(model, "trainer", None)
    model.trainers = trainer if isinstance(trainer, list) else ([trainer] if trainer is not None else [])
    runners = getattr(model, "runners", None) or []
    flat: List[Any] = []
    for r in runners:
        flat.extend(r if isinstance(r, list) else [r])
    model.runners = flat
    if initialize:
        model.tf_manager.initialize_sessions()
        if model.trainers or model.runners:
            model.tf_manager.initialize_model_parts(model.runners + model.trainers)
    return model
