"""The drop-in boundary of SURVEY 8(b)1, checked against the reference's sources: every class and function an INI file
(or user code) of the hot path names has the reference's parameter names, in the reference's order, required where
the reference requires them -- read from the reference's files with ``ast`` (nothing of it is imported) and from the
product with ``inspect``.  Two documented additions, both optional and at the end: ``TensorFlowManager(device, seed)``
and ``Dataset(series=)``; one relaxation: ``Dataset``'s ``iterators`` and ``batching`` have defaults.  The reference tree does not travel to the GPU box: skipped there."""
import ast
import importlib
import inspect
import os

import pytest

REF = "/root/reference/neuralmonkey/"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="no reference tree on this machine")

BOUNDARY = {
    "encoders/recurrent.py": ["RecurrentEncoder", "SentenceEncoder", "FactoredEncoder"],
    "encoders/transformer.py": ["TransformerEncoder"],
    "encoders/numpy_stateful_filler.py": ["SpatialFiller", "StatefulFiller"],
    "attention/feed_forward.py": ["Attention"],
    "attention/scaled_dot_product.py": ["MultiHeadAttention", "ScaledDotProdAttention"],
    "attention/combination.py": ["FlatMultiAttention", "HierarchicalMultiAttention"],
    "attention/coverage.py": ["CoverageAttention"],
    "attention/stateful_context.py": ["StatefulContext"],
    "decoders/decoder.py": ["Decoder"],
    "decoders/transformer.py": ["TransformerDecoder"],
    "decoders/beam_search_decoder.py": ["BeamSearchDecoder"],
    "model/sequence.py": ["EmbeddedFactorSequence", "EmbeddedSequence"],
    "runners/runner.py": ["GreedyRunner"],
    "runners/beamsearch_runner.py": ["BeamSearchRunner", "beam_search_runner_range"],
    "runners/plain_runner.py": ["PlainRunner"],
    "runners/xent_runner.py": ["XentRunner"],
    "runners/tensor_runner.py": ["TensorRunner", "RepresentationRunner"],
    "trainers/generic_trainer.py": ["GenericTrainer"],
    "trainers/cross_entropy_trainer.py": ["CrossEntropyTrainer"],
    "trainers/delayed_update_trainer.py": ["DelayedUpdateTrainer"],
    "trainers/multitask_trainer.py": ["MultitaskTrainer"],
    "trainers/objective.py": ["CostObjective"],
    "tf_manager.py": ["TensorFlowManager"],
    "dataset.py": ["Dataset", "BatchingScheme", "load"],
    "vocabulary.py": ["Vocabulary", "from_wordlist", "from_t2t_vocabulary", "from_nematus_json"],
    "processors/editops.py": ["Preprocess", "Postprocess"],
    "processors/bpe.py": ["BPEPreprocessor", "BPEPostprocessor"],
    "readers/numpy_reader.py": ["from_file_list"],
    "functions.py": ["noam_decay", "inverse_sigmoid_decay", "piecewise_function"],
}
ADDITIONS = {"TensorFlowManager": ["device", "seed"], "Dataset": ["series"]}
# required in the reference, optional here (every call the reference accepts is accepted): a Dataset may be given
# ``series=`` lists instead of ``iterators``, and may leave its batching scheme to ``batches(scheme)``
RELAXED = {"Dataset": ["iterators", "batching"]}


def _listed(fn, method):
    """[(name, has a default)] of a function definition node."""
    args = fn.args
    names = [a.arg for a in args.args]
    optional = [False] * (len(names) - len(args.defaults)) + [True] * len(args.defaults)
    listed = list(zip(names, optional))[1 if method else 0:]
    return listed + [(a.arg, d is not None) for a, d in zip(args.kwonlyargs, args.kw_defaults)]


def reference_parameters(path, name):
    with open(os.path.join(REF, path), encoding="utf-8") as handle:
        tree = ast.parse(handle.read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == name:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == "__init__":
                    return _listed(sub, True)
        if isinstance(node, ast.FunctionDef) and node.name == name:
            return _listed(node, False)
    raise AssertionError("{} not found in the reference's {}".format(name, path))


def product_parameters(path, name):
    module = importlib.import_module("neuralmonkey_amd." + path[:-3].replace("/", "."))
    target = getattr(module, name)
    signature = inspect.signature(target.__init__ if inspect.isclass(target) else target)
    return [(p.name, p.default is not inspect.Parameter.empty) for p in signature.parameters.values()
            if p.name != "self"]


@pytest.mark.parametrize("path", sorted(BOUNDARY))
def test_parameters_are_the_references(path):
    for name in BOUNDARY[path]:
        want, got = reference_parameters(path, name), product_parameters(path, name)
        extra = ADDITIONS.get(name, [])
        if extra:
            assert [n for n, _ in got[len(want):]] == extra and all(optional for _, optional in got[len(want):]), name
            got = got[:len(want)]
        relaxed = RELAXED.get(name, [])
        got = [(n, optional and n not in relaxed) for n, optional in got]
        assert got == want, "{}.{}:\n  product   {}\n  reference {}".format(path, name, got, want)
