"""The reference's OWN unit tests, run against the product: the test modules under /root/reference/neuralmonkey/tests
are loaded unmodified with ``neuralmonkey`` resolving to ``neuralmonkey_amd`` (an import alias that lives for the
duration of a test) and their unittest suites must pass -- "the parity tests read like the reference's own tests",
literally.  Covered: the modules that need no TensorFlow session -- ``test_dataset`` (lazy and shuffled datasets,
bucketing, globs: dataset.py), ``test_readers``, ``test_wordpiece``, ``test_config`` (the value grammar through the
reference's private names), ``test_decoder`` (constructor checks; its only use of TensorFlow, clearing the default
graph, maps to clearing the model-part registry) and the ``SentenceEncoder`` constructor table of
``test_encoders_init`` (its other test is for the sentence CNN encoder, which is outside the hot path: a placeholder
module lets the file import).  The reference tree does not travel to the GPU box: skipped there."""
import importlib
import importlib.abc
import importlib.util
import io
import os
import sys
import types
import unittest

import pytest

REF = "/root/reference"
TESTS = os.path.join(REF, "neuralmonkey", "tests")
pytestmark = pytest.mark.skipif(not os.path.isdir(TESTS), reason="no reference tree on this machine")

# modules of the reference that a test file imports and the product does not have (outside the hot path)
PLACEHOLDERS = {"neuralmonkey.encoders.sentence_cnn_encoder": ["SentenceCNNEncoder"]}


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name == "neuralmonkey" or name.startswith("neuralmonkey."):
            return importlib.util.spec_from_loader(name, self)
        return None

    def create_module(self, spec):
        if spec.name in PLACEHOLDERS:
            module = types.ModuleType(spec.name)
            for attribute in PLACEHOLDERS[spec.name]:
                setattr(module, attribute, type(attribute, (), {}))
            return module
        return importlib.import_module("neuralmonkey_amd" + spec.name[len("neuralmonkey"):])

    def exec_module(self, module):
        pass


def run_reference_tests(module_name, only=None):
    finder = _Alias()
    cwd = os.getcwd()
    sys.meta_path.insert(0, finder)
    if "tensorflow" not in sys.modules:             # test_decoder.py clears the default graph around its tests
        from neuralmonkey_amd.runtime import reset_registry
        stand_in = types.ModuleType("tensorflow")
        stand_in.reset_default_graph = reset_registry
        stand_in.__nm_test_stand_in__ = True
        sys.modules["tensorflow"] = stand_in
    os.chdir(REF)                                   # the tests name their data relative to the repository root
    try:
        spec = importlib.util.spec_from_file_location("reference_" + module_name,
                                                      os.path.join(TESTS, module_name + ".py"))
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        suite = unittest.defaultTestLoader.loadTestsFromModule(module)
        if only is not None:
            picked = unittest.TestSuite()
            for case in suite:
                for test in case:
                    if test.id().rsplit(".", 1)[-1] in only:
                        picked.addTest(test)
            suite = picked
        stream = io.StringIO()
        result = unittest.TextTestRunner(stream=stream, verbosity=2).run(suite)
        return result, stream.getvalue()
    finally:
        os.chdir(cwd)
        sys.meta_path.remove(finder)
        for name in [n for n in sys.modules if n == "neuralmonkey" or n.startswith("neuralmonkey.")]:
            del sys.modules[name]
        if getattr(sys.modules.get("tensorflow"), "__nm_test_stand_in__", False):
            del sys.modules["tensorflow"]


@pytest.mark.parametrize("module_name,ran,only", [
    ("test_dataset", 10, None),
    ("test_readers", 3, None),
    ("test_wordpiece", 7, None),
    ("test_config", 4, None),
    ("test_encoders_init", 1, ("test_sentence_encoder",)),
    ("test_decoder", 5, None),
])
def test_the_references_own_unit_tests_pass_on_the_product(module_name, ran, only):
    from neuralmonkey_amd.runtime import reset_registry
    reset_registry()
    result, log = run_reference_tests(module_name, only)
    assert result.testsRun == ran, log
    assert not result.failures and not result.errors, log
