"""The reference's own acceptance configs load UNMODIFIED (north_star: "existing configs
are drop-in").  Files: tests/{small,beamsearch,bahdanau,transformer,flat-multiattention,factored,
beamsearch_ensembles}.ini of /root/reference, byte for byte, read either from the reference tree (build
container) or from the committed bundle tests/golden/reference_tests.tar.gz (GPU box), with the working
directory at the root of that tree so that the relative data paths inside the files resolve
(config/configuration.py:60-120, experiment.py:176-227).  [main] keys of the host control plane
(evaluation / postprocess) build into ``OutOfScope`` placeholders.

CPU part: every file parses and its object graph builds (no variables are created).  GPU part
(test_reference_inis_gpu.py): variables initialise, three optimizer steps run, the runners decode.
"""
import os
import tarfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BUNDLE = os.path.join(HERE, "golden", "reference_tests.tar.gz")
BUNDLE_MORE = os.path.join(HERE, "golden", "reference_tests_more.tar.gz")     # post-edit.ini + its data, nematus.ini
REF = "/root/reference"
INIS = ["small", "beamsearch", "bahdanau", "transformer", "flat-multiattention", "factored",
        "beamsearch_ensembles"]


@pytest.fixture(scope="module")
def ref_root(tmp_path_factory):
    """Root of a tree that holds tests/<name>.ini and tests/data: the committed bundle, extracted."""
    root = tmp_path_factory.mktemp("reference_tests")
    for bundle in (BUNDLE, BUNDLE_MORE):
        with tarfile.open(bundle) as tar:
            tar.extractall(root)
    return str(root)


def load_verbatim(root, name, **kw):
    from neuralmonkey_amd.config.configuration import load_experiment
    cwd = os.getcwd()
    old = os.environ.get("NM_EXPERIMENT_NAME")
    os.environ["NM_EXPERIMENT_NAME"] = "small"          # tests/tests_run.sh:33 (variable substitution test)
    os.chdir(root)
    try:
        return load_experiment("tests/{}.ini".format(name), **kw)
    finally:
        os.chdir(cwd)
        if old is None:
            del os.environ["NM_EXPERIMENT_NAME"]
        else:
            os.environ["NM_EXPERIMENT_NAME"] = old


def test_bundle_is_the_reference_byte_for_byte(ref_root):
    if not os.path.isdir(REF):
        pytest.skip("no reference tree on this machine")
    names = []
    for bundle in (BUNDLE, BUNDLE_MORE):
        with tarfile.open(bundle) as tar:
            names += [m.name for m in tar.getmembers()]
    assert {"tests/{}.ini".format(n) for n in INIS + ["post-edit", "nematus"]} <= set(names)
    for rel in names:
        with open(os.path.join(REF, rel), "rb") as a, open(os.path.join(ref_root, rel), "rb") as b:
            assert a.read() == b.read(), rel


@pytest.mark.parametrize("name", INIS)
def test_reference_ini_builds_unmodified(ref_root, name):
    from neuralmonkey_amd.config.builder import OutOfScope
    model = load_verbatim(ref_root, name, initialize=False, device="cpu")
    assert model.trainers and model.runners
    # (tests/small.ini and tests/beamsearch*.ini give their training data a ``buffer_size``: lazy datasets, whose
    # length the reference refuses to tell -- dataset.py:407-419 -- and so does the product)
    if name in ("small", "beamsearch", "beamsearch_ensembles"):
        assert model.train_dataset.lazy
        with pytest.raises(NotImplementedError, match="Querying the len of a lazy dataset."):
            len(model.train_dataset)
    else:
        assert len(model.train_dataset) > 0
    assert len(next(iter(model.train_dataset.batches()))) > 0 and model.val_dataset is not None
    # the evaluation list survives with its series names; the evaluators are placeholders
    assert model.evaluation and all(isinstance(item[-1], OutOfScope) for item in model.evaluation)
    assert model.batch_size > 0 and model.epochs > 0 and isinstance(model.output, str)
    if name == "small":
        assert model.output == "tests/outputs/small"                 # {parent_dir}/{NM_EXPERIMENT_NAME}
        assert model.name.endswith("with 0.50 dropout")              # "{dropout:.2f}" from [vars]
    if name == "beamsearch_ensembles":
        assert len(model.tf_manager.sessions) == 4


@pytest.mark.parametrize("name", INIS)
def test_reference_ini_builds_from_the_reference_tree(name):
    if not os.path.isdir(REF):
        pytest.skip("no reference tree on this machine")
    model = load_verbatim(REF, name, initialize=False, device="cpu")
    assert model.trainers and model.runners


def test_coverage_attention_is_a_drop_in_section(ref_root):
    """tests/bahdanau.ini with its [attention] section switched to attention.CoverageAttention
    (attention/coverage.py:19-36: same arguments plus max_fertility) builds like the original."""
    from neuralmonkey_amd.attention import CoverageAttention
    src = os.path.join(ref_root, "tests", "bahdanau.ini")
    with open(src) as fh:
        text = fh.read()
    assert "class=attention.Attention" in text
    with open(os.path.join(ref_root, "tests", "bahdanau_coverage.ini"), "w") as fh:
        fh.write(text.replace("class=attention.Attention", "class=attention.CoverageAttention\nmax_fertility=4"))
    model = load_verbatim(ref_root, "bahdanau_coverage", initialize=False, device="cpu")
    atts = model.runners[0].decoder.attentions
    assert atts and isinstance(atts[0], CoverageAttention) and atts[0].max_fertility == 4
    assert atts[0].name == "attention_sentence_encoder"
    # the variables the reference's graph would hold for this part (coverage.py:38-46; no attn_bias: the override of
    # get_energies never touches bias_term), with TensorFlow's shapes
    model = load_verbatim(ref_root, "bahdanau_coverage", device="cpu")
    store = model.tf_manager.sessions[0].store
    mine = {n.split("/", 1)[1]: tuple(store[n].shape) for n in store.names() if n.startswith("attention_sentence_encoder/")}
    width = atts[0].context_vector_size
    assert mine == {"Attention/attn_query_projection": (8, width), "attn_key_projection": (width, width),
                    "attn_similarity_v": (width,), "attn_projection_bias": (width,),
                    "coverage_matrix": (1, 1, 1, width), "fertility_matrix": (1, 1, width)}


def test_first_batch_of_bahdanau_ini_is_the_batch_the_reference_built(ref_root):
    """The PRODUCT, loading tests/bahdanau.ini from the bundle, reads the same files through the same bucketed
    batching scheme: its first training batch is, sentence for sentence, the batch the REFERENCE'S own parser,
    builder and ``Dataset.batches`` produced for the fixture ``tests/golden/ref_exec/ini_bahdanau.npz``."""
    import numpy as np
    fixture = np.load(os.path.join(HERE, "golden", "ref_exec", "ini_bahdanau.npz"))
    model = load_verbatim(ref_root, "bahdanau", initialize=False, device="cpu")
    batch = next(iter(model.train_dataset.batches()))
    want_src = [[str(t) for t in row if str(t) != "<pad>"] for row in fixture["in/src_tokens"]]
    got_src = [list(s)[:10] for s in batch.get_series("source")]             # max_input_len=10 of the encoder section
    assert got_src == want_src
    want_tgt = [[str(t) for t in row if str(t) not in ("<pad>", "</s>")] for row in fixture["in/tgt_tokens"]]
    got_tgt = [list(s)[:len(w)] for s, w in zip(batch.get_series("target"), want_tgt)]
    assert got_tgt == want_tgt
    # and the vocabularies the INI's word lists give
    runner = model.runners[0]
    assert list(runner.decoder.vocabulary.index_to_word) == [str(w) for w in fixture["in/tgt_vocabulary"]]
    enc = runner.decoder.encoders[0]
    assert list(enc.input_sequence.vocabularies[0].index_to_word) == [str(w) for w in fixture["in/src_vocabulary"]]


VARIABLE_INIS = ["small", "bahdanau", "factored", "post-edit", "beamsearch", "transformer",
                 "flat-multiattention"]


def reference_variables(name):
    import json
    import numpy as np
    z = np.load(os.path.join(HERE, "golden", "ref_exec", "ini_variables.npz"))
    return json.loads(str(z["out/" + name]))


@pytest.mark.parametrize("name", VARIABLE_INIS)
def test_checkpoint_variables_are_the_ones_the_reference_creates(ref_root, name):
    """SURVEY 8(f)1, the variable-name contract: the REFERENCE built each acceptance configuration (parser, builder,
    model parts, runners -- tests/golden/make_reference_exec_golden.py: ini_variables) and listed the variables its
    lazily built model creates; the product, given the same file, declares variables of the same names and shapes --
    so a checkpoint of either loads into the other (parameterized.py:68-125, tf_manager.py:274-277)."""
    from neuralmonkey_amd.runtime import registered_parts
    from neuralmonkey_amd.variables import VariableStore
    load_verbatim(ref_root, name, initialize=False, device="cpu")
    store = VariableStore("cpu")
    for part in registered_parts():
        part.declare_variables(store)
    # what a checkpoint holds: the variables of the flat store in TensorFlow's shapes (tf_bundle.tf_shape: scalars,
    # [1, 1, n] vectors, 1x1 convolution filters) + the variables TensorFlow creates and nothing reads
    from neuralmonkey_amd.tf_bundle import tf_shape
    mine = {n: list(tf_shape(n, s.shape)) for n, s in store.specs.items()}
    unread = {n: list(s.shape) for n, s in store.checkpoint_only.items()}
    assert not set(mine) & set(unread)
    if name == "small":              # NematusGRU: GRUCell.build's four variables per cell, four cells
        assert len(unread) == 16 and all("nematus_gru_cell" in n or "cond_gru_2_cell" in n for n in unread)
    else:
        assert not unread
    mine.update(unread)
    theirs = {n: s for n, s in reference_variables(name)["variables"]}
    missing = sorted(set(theirs) - set(mine))
    extra = sorted(set(mine) - set(theirs))
    assert not missing and not extra, (missing, extra)
    assert {n: mine[n] for n in theirs} == theirs


def test_the_file_the_reference_refuses_is_refused_with_its_words(ref_root):
    """tests/nematus.ini does not build in the reference at this commit (a residual encoder whose layers differ in
    size, encoders/recurrent.py:163-168; the file is not part of tests/tests_run.sh): same error, same text."""
    want = reference_variables("nematus")["error"]
    with pytest.raises(Exception) as info:
        load_verbatim(ref_root, "nematus", initialize=False, device="cpu")
    inner = getattr(info.value, "original_exception", info.value)
    assert type(inner).__name__ == want["type"] and str(inner) == want["text"]


def test_post_edit_ini_builds_and_its_scripts_are_the_references(ref_root):
    """tests/post-edit.ini (tests/tests_run.sh:12): two encoders, dot-product attentions on the RNN decoder, a
    dataset-level preprocessor (``processors.editops.Preprocess`` as an element of ``data``), a [main] postprocessor,
    and datasets that batch by [main] batch_size (dataset.py:237-246)."""
    from neuralmonkey_amd.processors.editops import Postprocess, convert_to_edits
    model = load_verbatim(ref_root, "post-edit", initialize=False, device="cpu")
    assert model.batch_size == 2 and model.train_dataset.batching.batch_size == 2
    batches = list(model.train_dataset.batches())
    assert [len(b) for b in batches] == [2] * (len(model.train_dataset) // 2)
    first = batches[0]
    mt, pe, edits = (list(first.get_series(key)) for key in ("translated", "target", "edits"))
    assert edits == [convert_to_edits(before, after) for before, after in zip(mt, pe)]
    assert any(op == "<keep>" for script in edits for op in script)
    (series, post), = model.postprocess
    assert series == "target" and isinstance(post, Postprocess)
    assert post({"translated": mt}, {"edits": edits}) == pe
    # ... and it is the batch the REFERENCE'S pipeline made of the file (fixture ini_postedit: the model's inputs
    # after its length limit of 5)
    import numpy as np
    z = np.load(os.path.join(HERE, "golden", "ref_exec", "ini_postedit.npz"))
    assert [" ".join(sentence) for sentence in mt] == z["in/translated"].tolist()
    assert [sentence[:5] for sentence in first.get_series("source")] == z["in/src_tokens"].tolist()
    assert [script[:5] for script in edits] == [[t for t in row if t != "<pad>"] for row in z["in/tgt_tokens"].tolist()]


def test_a_dataset_without_batching_needs_main_batch_size(ref_root, tmp_path):
    """dataset.py:243-245: the reference's error, raised while the configuration is being built."""
    from neuralmonkey_amd.config.configuration import load_experiment
    with open(os.path.join(ref_root, "tests", "post-edit.ini"), encoding="utf-8") as handle:
        text = handle.read()
    assert "\nbatch_size=2\n" in text
    path = tmp_path / "no_batch_size.ini"
    path.write_text(text.replace("\nbatch_size=2\n", "\n"), encoding="utf-8")
    cwd = os.getcwd()
    os.chdir(ref_root)
    try:
        with pytest.raises(Exception) as info:
            load_experiment(str(path), initialize=False, device="cpu")
    finally:
        os.chdir(cwd)
    inner = getattr(info.value, "original_exception", info.value)
    assert isinstance(inner, ValueError)
    assert str(inner) == "Argument main.batch_size is not specified, cannot use default batching scheme."


def _product_trainers(model, name):
    if name == "bahdanau":                     # trainer=[<mt_trainer>, <greedy_trainer>], mt_trainer over 1, 1, 2
        multitask, greedy = model.trainers
        return {"trainer1": multitask.trainers[0], "trainer2": multitask.trainers[2], "greedy_trainer": greedy}
    return {"trainer": model.trainers[0]}


@pytest.mark.parametrize("name", ["small", "bahdanau", "post-edit"])
def test_what_the_trainers_minimise_is_what_the_reference_trainers_minimise(ref_root, name):
    """trainers/generic_trainer.py:84-134, evaluated by the REFERENCE on its own configurations (fixture
    ``ini_trainer_objectives``): the variables under the regulariser, the trainers' variable lists, their weights and
    clipping thresholds -- the product's trainers, built from the same files, agree -- and the L1 / L2 sums and the
    minimised sum loss + l1_weight L1 + l2_weight L2, recomputed here from the fixture's variables over the PRODUCT'S
    choice of variables.  (tests/bahdanau.ini trains with ``supress_unk``: -1e9 on the <unk> logit makes the loss of
    a batch with unknown target words 3.8e8 -- in the reference, hence here.)"""
    import json
    import numpy as np
    from neuralmonkey_amd.runtime import registered_parts
    from neuralmonkey_amd.variables import VariableStore
    z = np.load(os.path.join(HERE, "golden", "ref_exec", "ini_trainer_objectives.npz"))
    want = json.loads(str(z["out/" + name]))
    strip = lambda names: [n[:-2] if n.endswith(":0") else n for n in names]
    model = load_verbatim(ref_root, name, initialize=False, device="cpu")
    store = VariableStore("cpu")
    for part in registered_parts():
        part.declare_variables(store)
    # variables TensorFlow creates and nothing reads (GRUCell.build under NematusGRUCell) are trainable variables of the
    # reference's graph: its regulariser sums them too.  The engine keeps them out of the flat buffers, so its REPORTED
    # L1 / L2 lack their share (they stay at their N(0, 0.001) initial values: nothing but the regulariser's own
    # gradient ever moves them); no gradient of a variable that is read is affected.  (The fixture's variables carry
    # the generator's seeded test values, not their initial ones: the size of that share cannot be read off it.)
    unread = [n for n in store.checkpoint_only if "ias" not in n]
    assert bool(unread) == (name == "small")
    for section, trainer in _product_trainers(model, name).items():
        ref = want[section]
        assert sorted(trainer.regularizable(store) + unread) == sorted(strip(want["_regularizable"]))
        assert sorted(trainer.var_list(store) + list(store.checkpoint_only)) == sorted(strip(ref["var_list"]))
        assert (trainer.l1_weight, trainer.l2_weight, trainer.clip_norm) == \
            (ref["l1_weight"], ref["l2_weight"], ref["clip_norm"])
        assert [obj.name for obj in trainer.objectives] == ref["objective_names"]
        assert [obj.weight for obj in trainer.objectives] == ref["objective_weights"]
        values = {n: z["vars/{}/{}".format(name, n)].astype(np.float64) for n in trainer.regularizable(store) + unread}
        l1 = sum(np.abs(v).sum() for v in values.values())
        l2 = sum((v ** 2).sum() for v in values.values())
        loss, ref_l1, ref_l2 = ref["objective_values"]
        assert abs(l1 - ref_l1) <= 2e-6 * ref_l1 and abs(l2 - ref_l2) <= 2e-6 * ref_l2
        total = np.float32(loss) + np.float32(ref["l1_weight"]) * np.float32(ref_l1) \
            + np.float32(ref["l2_weight"]) * np.float32(ref_l2)
        assert abs(float(total) - ref["differentiable_loss_sum"]) <= 2e-7 * abs(ref["differentiable_loss_sum"])
