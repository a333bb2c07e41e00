"""Generate ``tests/golden/ref_exec/*.npz``: numbers produced by the REFERENCE'S OWN Python.

Runs only in the build container (needs ``/root/reference``; nothing at test time does).  It installs the NumPy-eager
TensorFlow stand-in (``tests/ref_exec/tf_eager.py``) as ``sys.modules["tensorflow"]``, imports the model parts from
``/root/reference/neuralmonkey`` UNMODIFIED and runs them -- constructors, ``feed_dict``, the lazy ``@tensor``
properties, the ``tf.while_loop`` bodies of ``AutoregressiveDecoder`` / ``BeamSearchDecoder``, ``BeamSearchRunner``'s
executable -- on seeded inputs.  What TensorFlow itself would compute inside its ops (GRUCell, LSTMCell, dynamic_rnn,
dense, softmax, top_k, sequence_loss) is the stand-in's restatement; every line between those ops is the reference's.

A fixture holds: ``cfg`` (JSON: how the model was built), ``p/<variable name>`` (every variable the reference created,
under the name TensorFlow's scoping rules give it, in creation order in ``var_order``), ``in/*`` (token strings and
what ``Vocabulary.strings_to_indices`` made of them), ``out/*`` (what the reference computed).

    python tests/golden/make_reference_exec_golden.py            # all cases
    python tests/golden/make_reference_exec_golden.py rnn_gru    # one case

``tests/test_reference_exec.py`` asserts that ``oracle/`` reproduces every fixture; ``tests/test_reference_exec_gpu.py``
runs the engine on the same strings and variables.
"""
import collections
import collections.abc
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get("NM_REF_EXEC_OUT") or os.path.join(HERE, "ref_exec")
REFERENCE = "/root/reference"

sys.dont_write_bytecode = True                      # /root/reference is read-only
collections.Sized = collections.abc.Sized           # the reference targets Python <= 3.7 (vocabulary.py:175)
sys.path.insert(0, REPO)
from tests.ref_exec import tf_eager                 # noqa: E402

tf = tf_eager.install()
sys.path.insert(0, REFERENCE)


# ----------------------------------------------------------------------------------------------------------------
# variables: values are a function of the variable's NAME (the oracle / the engine receive the same arrays by name)
# ----------------------------------------------------------------------------------------------------------------
SCALE = {"default": 0.35}


def variable_factory(name, shape, np_dtype, initializer):
    if np.dtype(np_dtype).kind != "f":
        return np.zeros(shape, np_dtype)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    scale = SCALE["default"]
    if name.endswith("gamma"):                           # layer-norm gains around one
        return (1.0 + rng.normal(0, 0.2, shape)).astype(np_dtype)
    if "word_embeddings" in name or "embedding_matrix" in name:
        scale = 0.6
    if name.endswith("state_to_word_W"):
        scale = 0.9
    return rng.normal(0, scale, shape).astype(np_dtype)


tf_eager.VARIABLE_FACTORY = variable_factory


def fresh_graph():
    tf_eager.reset_default_graph()


def joined(sentence):
    """A decoded sentence as one string.  NB ``BeamSearchRunner.prepare_results`` (beamsearch_runner.py:88-99) assigns
    ``decoded_tokens[i] = decoded`` INSIDE its token loop, after the ``break`` on ``</s>``: a hypothesis whose first
    token is ``</s>`` keeps its raw id array instead of becoming ``[]``; such entries come out as digits here."""
    return " ".join(str(t) for t in sentence)


def to_numpy(struct):
    return tf_eager.nest.map_structure(
        lambda t: t.numpy() if isinstance(t, tf_eager.Tensor) else t, struct)


def variables():
    order = [n for n, _ in tf_eager.CREATION_LOG]
    return order, {n: tf_eager._STORE.vars[n].numpy().copy() for n in order}


def save(case, cfg, arrays):
    os.makedirs(OUT, exist_ok=True)
    order, params = variables()
    blob = {"cfg": np.asarray(json.dumps(cfg, sort_keys=True)), "var_order": np.asarray(order)}
    for n, v in params.items():
        blob["p/" + n] = v
    for k, v in arrays.items():
        v = np.asarray(v)
        if v.dtype == object:
            v = v.astype(str)
        blob[k] = v
    path = os.path.join(OUT, case + ".npz")
    np.savez_compressed(path, **blob)
    print("{:28s} {:4d} variables {:4d} arrays {:8d} bytes".format(
        case, len(order), len(arrays), os.path.getsize(path)))


# ----------------------------------------------------------------------------------------------------------------
# data
# ----------------------------------------------------------------------------------------------------------------
def make_vocab(n):
    from neuralmonkey.vocabulary import Vocabulary
    return Vocabulary(["w{}".format(i) for i in range(n)])


def sentences(rng, count, vocab_words, min_len, max_len, oov_every=0):
    out = []
    for i in range(count):
        n = int(rng.integers(min_len, max_len + 1))
        sent = ["w{}".format(int(rng.integers(0, vocab_words))) for _ in range(n)]
        if oov_every and n and i % oov_every == 0:
            sent[int(rng.integers(0, n))] = "never-seen"
        out.append(sent)
    return out


def dataset(series):
    from neuralmonkey.dataset import Dataset
    return Dataset("fixture", {k: (lambda v=v: iter(v)) for k, v in series.items()}, batching=None)


def feed(parts, ds, train, inputs):
    for p in parts:
        p.register_input(inputs)
    fd = {}
    for p in parts:
        fd.update(p.feed_dict(ds, train=train))
    return fd


def string_inputs(*names):
    return {n: tf.placeholder(tf.string, [None, None], n) for n in names}


# ----------------------------------------------------------------------------------------------------------------
# RNN encoder-decoder family
# ----------------------------------------------------------------------------------------------------------------
RNN_DEFAULT = dict(
    src_vocab=17, tgt_vocab=13, emb=6, enc_layers=[[5, "bidirectional", "GRU"]], sentence_encoder=True,
    add_layer_norm=False, add_residual=False, enc_keep=1.0, att_keep=1.0, att_state=None,
    dec_cell="GRU", rnn_size=6, conditional_gru=False, attention_on_input=False, dec_keep=1.0,
    output_projection=["nonlinear", "tanh"], encoder_projection="linear", tie_embeddings=False, supress_unk=False,
    max_output_len=7, max_input_len=None, beam=[3, 6, 0.6], seed=1, batch=5, spatial=None)


def build_rnn(cfg):
    from neuralmonkey.encoders.recurrent import SentenceEncoder, RecurrentEncoder
    from neuralmonkey.model.sequence import EmbeddedSequence
    from neuralmonkey.attention.feed_forward import Attention
    from neuralmonkey.decoders.decoder import Decoder
    from neuralmonkey.decoders import output_projection as OP
    from neuralmonkey.decoders import encoder_projection as EP
    sv, tv = make_vocab(cfg["src_vocab"]), make_vocab(cfg["tgt_vocab"])
    parts = []
    if cfg["spatial"] is not None:
        from neuralmonkey.encoders.numpy_stateful_filler import SpatialFiller
        h, w, c, ff_dim, proj_dim = cfg["spatial"]
        enc = SpatialFiller(name="encoder", input_shape=[h, w, c], data_id="maps",
                            projection_dim=proj_dim, ff_hidden_dim=ff_dim)
        parts.append(enc)
    elif cfg["sentence_encoder"]:
        size, direction, cell = cfg["enc_layers"][0]
        enc = SentenceEncoder(name="encoder", vocabulary=sv, data_id="source", embedding_size=cfg["emb"],
                              rnn_size=size, rnn_cell=cell, rnn_direction=direction,
                              add_residual=cfg["add_residual"], add_layer_norm=cfg["add_layer_norm"],
                              max_input_len=cfg["max_input_len"], dropout_keep_prob=cfg["enc_keep"])
        parts += [enc, enc.input_sequence]
    else:
        seq = EmbeddedSequence(name="encoder_input", vocabulary=sv, data_id="source", embedding_size=cfg["emb"],
                               max_length=cfg["max_input_len"])
        enc = RecurrentEncoder(name="encoder", input_sequence=seq,
                               rnn_layers=[tuple(layer) for layer in cfg["enc_layers"]],
                               add_residual=cfg["add_residual"], add_layer_norm=cfg["add_layer_norm"],
                               dropout_keep_prob=cfg["enc_keep"])
        parts += [enc, seq]
    att = Attention(name="attention", encoder=enc, dropout_keep_prob=cfg["att_keep"], state_size=cfg["att_state"])
    kind = cfg["output_projection"][0]
    act = {"tanh": tf.tanh, "relu": tf.nn.relu}
    if kind == "nonlinear":
        op = OP.nonlinear_output(cfg["rnn_size"], act[cfg["output_projection"][1]], cfg["dec_keep"])
    elif kind == "nematus":
        op = OP.nematus_output(cfg["rnn_size"], act[cfg["output_projection"][1]], cfg["dec_keep"])
    elif kind == "maxout":
        op = OP.maxout_output(cfg["rnn_size"], cfg["dec_keep"])
    elif kind == "mlp":
        op = OP.mlp_output(list(cfg["output_projection"][1]), act[cfg["output_projection"][2]], cfg["dec_keep"])
    elif kind == "default":
        op = None
    else:
        raise ValueError(kind)
    ep = {"linear": None, "nematus": EP.nematus_projection(cfg["dec_keep"]),
          "concat": None, "empty": EP.empty_initial_state}[cfg["encoder_projection"]]
    dec = Decoder(encoders=[enc], vocabulary=tv, data_id="target", name="decoder",
                  max_output_len=cfg["max_output_len"], dropout_keep_prob=cfg["dec_keep"],
                  embedding_size=cfg["rnn_size"], rnn_size=None if cfg["encoder_projection"] == "concat" else cfg["rnn_size"],
                  output_projection=op, encoder_projection=ep, attentions=[att],
                  attention_on_input=cfg["attention_on_input"], rnn_cell=cfg["dec_cell"],
                  conditional_gru=cfg["conditional_gru"], tie_embeddings=cfg["tie_embeddings"],
                  supress_unk=cfg["supress_unk"])
    parts += [att, dec]
    return enc, att, dec, parts


def rnn_series(cfg):
    rng = np.random.default_rng(cfg["seed"])
    bsz = cfg["batch"]
    tgt = sentences(rng, bsz, cfg["tgt_vocab"], 1, cfg["max_output_len"] + 2, oov_every=3)
    if cfg["spatial"] is not None:
        h, w, c = cfg["spatial"][:3]
        maps = np.maximum(rng.normal(0, 1, (bsz, h, w, c)), 0).astype(np.float32)
        return {"maps": list(maps), "target": tgt}
    src = sentences(rng, bsz, cfg["src_vocab"], 2, 7, oov_every=2)
    src[-1] = src[-1][:1]                                        # a one-word sentence
    if bsz > 2:
        src[1] = (src[1] * 7)[:7]                                 # the longest one
    return {"source": src, "target": tgt}


def run_rnn(case, **overrides):
    cfg = dict(RNN_DEFAULT, **overrides)
    series = rnn_series(cfg)
    out = {}
    src_key = "maps" if cfg["spatial"] is not None else "source"

    # -- teacher-forced pass and greedy decoding on the whole batch ------------------------------------------------
    fresh_graph()
    enc, att, dec, parts = build_rnn(cfg)
    inputs = string_inputs("source", "target")
    if cfg["spatial"] is not None:
        inputs["maps"] = tf.placeholder(tf.float32, [None] + list(cfg["spatial"][:3]), "maps")
    ds = dataset(series)
    fd = feed(parts, ds, False, inputs)
    with tf_eager.feeding(fd):
        if cfg["spatial"] is None:
            out["in/src_tokens"] = enc.input_sequence.input_factors[0].numpy()
            out["in/src_ids"] = enc.input_sequence.inputs.numpy()
            out["out/enc_input"] = enc.input_sequence.temporal_states.numpy()
            out["out/enc_states"] = enc.temporal_states.numpy()
            out["out/enc_mask"] = enc.temporal_mask.numpy()
        else:
            out["in/maps"] = np.stack(series["maps"])
            out["out/enc_states"] = enc.spatial_states.numpy()
            out["out/att_states"] = att.attention_states.numpy()
        out["out/enc_output"] = enc.output.numpy()
        out["in/tgt_tokens"] = dec.train_tokens.numpy()
        out["in/tgt_ids"] = dec.train_inputs.numpy()                     # time-major [T,B]
        out["out/train_mask"] = dec.train_mask.numpy()
        out["out/hidden_features"] = att.hidden_features.numpy()
        out["out/initial_state"] = dec.initial_state.numpy()
        out["out/train_logits"] = dec.train_logits.numpy()
        out["out/train_output_states"] = dec.train_output_states.numpy()
        out["out/train_xents"] = dec.train_xents.numpy()
        out["out/train_loss"] = dec.train_loss.numpy()
        tr = dec.train_loop_result
        out["out/train_rnn_outputs"] = tr.histories.other.rnn_outputs.numpy()
        out["out/train_att_weights"] = att.histories["decoder_train"].numpy()
        out["out/train_contexts"] = tr.histories.other.attention_histories[0].contexts.numpy()
        rr = dec.runtime_loop_result
        out["out/runtime_logits"] = dec.runtime_logits.numpy()
        out["out/runtime_symbols"] = rr.histories.output_symbols.numpy()
        out["out/runtime_mask"] = dec.runtime_mask.numpy()
        out["out/runtime_steps"] = rr.feedables.step.numpy()
        out["out/runtime_finished"] = rr.feedables.finished.numpy()
        out["out/runtime_att_weights"] = att.histories["decoder_run"].numpy()
        out["out/runtime_xents"] = dec.runtime_xents.numpy()
        out["out/runtime_loss"] = dec.runtime_loss.numpy()
        out["out/runtime_logprobs"] = dec.runtime_logprobs.numpy()
        out["out/decoded"] = dec.decoded.numpy()
        # GreedyRunner's post-processing (runners/runner.py:35-63 -> vocabulary.py:257-288)
        amax = np.argmax(out["out/runtime_logprobs"], axis=2)
        sents = dec.vocabulary.vectors_to_sentences(amax)
        out["out/greedy_sentences"] = np.asarray([" ".join(s) for s in sents])
        # the trainer's objective (trainers/generic_trainer.py:84-135, cross_entropy_trainer.py:21-53): which variables
        # the L1 / L2 terms cover (name regex), the terms, the weighted sum.  (Gradients / Adam are TensorFlow's.)
        if case == "rnn_gru":
            from neuralmonkey.trainers.cross_entropy_trainer import CrossEntropyTrainer
            trainer = CrossEntropyTrainer(decoders=[dec], l1_weight=0.3, l2_weight=0.02, clip_norm=1.0)
            l1, l2 = trainer.regularization_losses
            out["out/trainer_l1"], out["out/trainer_l2"] = l1.numpy(), l2.numpy()
            out["out/trainer_loss_sum"] = trainer.differentiable_loss_sum.numpy()
            out["out/trainer_objective_values"] = np.stack([np.asarray(v.numpy(), np.float32)
                                                            for v in trainer.objective_values])
            out["out/trainer_loss_names"] = np.asarray([o.name for o in trainer.objectives] + ["L1", "L2"])
    order_full, _ = variables()

    # -- beam search, one sentence at a time (the reference does not tile Bahdanau keys: batch 1 only) ---------------
    k, max_steps, alpha = cfg["beam"]
    from neuralmonkey.decoders.beam_search_decoder import BeamSearchDecoder
    from neuralmonkey.runners.beamsearch_runner import BeamSearchRunner
    for i in range(cfg["batch"]):
        fresh_graph()
        enc, att, dec, parts = build_rnn(cfg)
        bs = BeamSearchDecoder(name="beam", parent_decoder=dec, beam_size=k, max_steps=max_steps,
                               length_normalization=alpha)
        runner = BeamSearchRunner(output_series="hyp", decoder=bs, rank=1)
        one = {name: [vals[i]] for name, vals in series.items()}
        fd = feed(parts + [bs], dataset(one), False, inputs)
        with tf_eager.feeding(fd):
            ex = runner.get_executable(compute_losses=False, summaries=False, num_sessions=1)
            fetches, _ = ex.next_to_execute()
            res = to_numpy(fetches)
            ex.collect_results([res])
            bo = res["bs_outputs"]
            pre = "out/beam{}_".format(i)
            out[pre + "scores"] = bo.last_search_step_output.scores
            out[pre + "token_ids"] = bo.last_search_step_output.token_ids
            out[pre + "logprob_sum"] = bo.last_search_state.logprob_sum
            out[pre + "lengths"] = bo.last_search_state.lengths
            out[pre + "finished"] = bo.last_search_state.finished
            out[pre + "prev_logprobs"] = bo.last_search_state.prev_logprobs
            out[pre + "dec_step"] = bo.last_dec_loop_state.feedables.step
            out[pre + "sentence"] = np.asarray(joined(ex.result.outputs["hyp"][0]))
            out[pre + "loss"] = np.asarray(ex.result.losses["hyp/beam_search_score"])
        order_beam, _ = variables()
        missing = [n for n in order_beam if n not in order_full]
        assert not missing, missing
    # leave the store holding the full-batch graph's variables for save()
    fresh_graph()
    enc, att, dec, parts = build_rnn(cfg)
    with tf_eager.feeding(feed(parts, ds, False, inputs)):
        dec.train_loss.numpy()
        dec.runtime_logits.numpy()
    save(case, cfg, out)


# ----------------------------------------------------------------------------------------------------------------
# Transformer (rows a19-a21)
# ----------------------------------------------------------------------------------------------------------------
TR_DEFAULT = dict(src_vocab=19, tgt_vocab=19, dim=8, ff=12, depth=2, heads=2, heads_self=2, heads_enc=2,
                  max_output_len=7, tie_embeddings=True, use_att_transform_bias=False, target_space_id=None,
                  shared_embeddings=False, scale_embeddings=False,
                  second_encoder=False, strategy="serial", heads_hier=None,
                  beam=[3, 6, 0.6], seed=3, batch=4)


def build_transformer(cfg):
    from neuralmonkey.model.sequence import EmbeddedSequence
    from neuralmonkey.encoders.transformer import TransformerEncoder
    from neuralmonkey.decoders.transformer import TransformerDecoder
    sv = make_vocab(cfg["src_vocab"])
    tv = sv if cfg["shared_embeddings"] else make_vocab(cfg["tgt_vocab"])
    seq = EmbeddedSequence(name="encoder_input", vocabulary=sv, data_id="source", embedding_size=cfg["dim"],
                           scale_embeddings_by_depth=cfg["scale_embeddings"])
    enc = TransformerEncoder(name="encoder", input_sequence=seq, ff_hidden_size=cfg["ff"], depth=cfg["depth"],
                             n_heads=cfg["heads"], target_space_id=cfg["target_space_id"],
                             use_att_transform_bias=cfg["use_att_transform_bias"])
    encoders, parts = [enc], [seq, enc]
    if cfg["second_encoder"]:      # attention/transformer_cross_layer.py:68-268: the four combination strategies
        seq2 = EmbeddedSequence(name="encoder2_input", vocabulary=sv, data_id="source2", embedding_size=cfg["dim"])
        enc2 = TransformerEncoder(name="encoder2", input_sequence=seq2, ff_hidden_size=cfg["ff"], depth=cfg["depth"],
                                  n_heads=cfg["heads"])
        encoders.append(enc2)
        parts += [seq2, enc2]
    dec = TransformerDecoder(name="decoder", encoders=encoders, vocabulary=tv, data_id="target",
                             ff_hidden_size=cfg["ff"], n_heads_self=cfg["heads_self"], n_heads_enc=cfg["heads_enc"],
                             depth=cfg["depth"], max_output_len=cfg["max_output_len"],
                             embedding_size=None if cfg["shared_embeddings"] else cfg["dim"],
                             embeddings_source=seq if cfg["shared_embeddings"] else None,
                             tie_embeddings=cfg["tie_embeddings"],
                             use_att_transform_bias=cfg["use_att_transform_bias"],
                             attention_combination_strategy=cfg["strategy"], n_heads_hier=cfg["heads_hier"])
    return seq, enc, dec, parts + [dec]


def run_transformer(case, **overrides):
    cfg = dict(TR_DEFAULT, **overrides)
    rng = np.random.default_rng(cfg["seed"])
    bsz = cfg["batch"]
    src = sentences(rng, bsz, cfg["src_vocab"], 1, 6, oov_every=2)
    tgt = sentences(rng, bsz, cfg["tgt_vocab"], 1, cfg["max_output_len"] + 1, oov_every=3)
    src[-1] = src[-1][:1]
    series = {"source": src, "target": tgt}
    if cfg["second_encoder"]:
        series["source2"] = sentences(rng, bsz, cfg["src_vocab"], 1, 5, oov_every=4)
    out = {}
    fresh_graph()
    seq, enc, dec, parts = build_transformer(cfg)
    inputs = string_inputs("source", "target", "source2")
    ds = dataset(series)
    with tf_eager.feeding(feed(parts, ds, False, inputs)):
        out["in/src_tokens"] = seq.input_factors[0].numpy()
        out["in/src_ids"] = seq.inputs.numpy()
        if cfg["second_encoder"]:
            out["in/src2_tokens"] = parts[2].input_factors[0].numpy()
            out["in/src2_ids"] = parts[2].inputs.numpy()
            out["out/enc2_states"] = parts[3].temporal_states.numpy()
        out["in/tgt_tokens"] = dec.train_tokens.numpy()
        out["in/tgt_ids"] = dec.train_inputs.numpy()
        out["out/encoder_inputs"] = enc.encoder_inputs.numpy()
        out["out/enc_states"] = enc.temporal_states.numpy()
        out["out/enc_mask"] = enc.temporal_mask.numpy()
        out["out/enc_output"] = enc.output.numpy()
        with enc.use_scope():       # ``layer`` is a plain method: its callers are inside the part's scope
            for lvl in range(cfg["depth"] + 1):
                out["out/enc_layer{}".format(lvl)] = enc.layer(lvl).temporal_states.numpy()
        out["out/train_input_symbols"] = dec.train_input_symbols.numpy()
        out["out/train_logits"] = dec.train_logits.numpy()
        out["out/train_output_states"] = dec.train_output_states.numpy()
        out["out/train_xents"] = dec.train_xents.numpy()
        out["out/train_loss"] = dec.train_loss.numpy()
        rr = dec.runtime_loop_result
        out["out/runtime_logits"] = dec.runtime_logits.numpy()
        out["out/runtime_symbols"] = rr.histories.output_symbols.numpy()
        out["out/runtime_mask"] = dec.runtime_mask.numpy()
        out["out/runtime_loss"] = dec.runtime_loss.numpy()
        out["out/decoded"] = dec.decoded.numpy()
    # beam search on the whole batch (encoder states are tiled to the beam: beam_search_decoder.py:166-178)
    from neuralmonkey.decoders.beam_search_decoder import BeamSearchDecoder
    from neuralmonkey.runners.beamsearch_runner import BeamSearchRunner
    k, max_steps, alpha = cfg["beam"]
    fresh_graph()
    seq, enc, dec, parts = build_transformer(cfg)
    bs = BeamSearchDecoder(name="beam", parent_decoder=dec, beam_size=k, max_steps=max_steps,
                           length_normalization=alpha)
    runner = BeamSearchRunner(output_series="hyp", decoder=bs, rank=1)
    with tf_eager.feeding(feed(parts + [bs], ds, False, inputs)):
        ex = runner.get_executable(compute_losses=False, summaries=False, num_sessions=1)
        fetches, _ = ex.next_to_execute()
        res = to_numpy(fetches)
        ex.collect_results([res])
        bo = res["bs_outputs"]
        out["out/beam_scores"] = bo.last_search_step_output.scores
        out["out/beam_token_ids"] = bo.last_search_step_output.token_ids
        out["out/beam_logprob_sum"] = bo.last_search_state.logprob_sum
        out["out/beam_lengths"] = bo.last_search_state.lengths
        out["out/beam_finished"] = bo.last_search_state.finished
        out["out/beam_sentences"] = np.asarray([joined(s) for s in ex.result.outputs["hyp"]])
        out["out/beam_loss"] = np.asarray(ex.result.losses["hyp/beam_search_score"])
    save(case, cfg, out)


# ----------------------------------------------------------------------------------------------------------------
# gradients of the reference's loss, by central differences of the reference's own forward pass (row a22)
# ----------------------------------------------------------------------------------------------------------------
def run_fd_gradients(case, family, per_variable=4, h=5e-3, **overrides):
    """``tf.gradients`` is TensorFlow's (trainers/generic_trainer.py:136-195) and the stand-in has no autodiff -- but
    it can evaluate the reference's ``train_loss`` (decoders/autoregressive.py:289-316) at perturbed variables.  For
    ``per_variable`` coordinates of every trainable variable: (loss(theta + h e_i) - loss(theta - h e_i)) / 2h, the
    graph rebuilt for each evaluation.  float32 forward passes: about 1e-4 of absolute noise on each derivative,
    enough to tell a wrong gradient (a missing term, a transposed product, a wrong mask) from a right one."""
    if family == "rnn":
        cfg = dict(RNN_DEFAULT, **overrides)
        series = rnn_series(cfg)
        inputs = string_inputs("source", "target")
        if cfg["spatial"] is not None:
            inputs["maps"] = tf.placeholder(tf.float32, [None] + list(cfg["spatial"][:3]), "maps")

        def build():
            enc, att, dec, parts = build_rnn(cfg)
            return dec, parts
    elif family == "ms":
        cfg = dict(MS_DEFAULT, **overrides)
        rng = np.random.default_rng(cfg["seed"])
        bsz = cfg["batch"]
        src = sentences(rng, bsz, cfg["src_vocab"], 2, 6, oov_every=2)
        series = {"source": src,
                  "target": sentences(rng, bsz, cfg["tgt_vocab"], 1, cfg["max_output_len"] + 1, oov_every=3)}
        inputs = string_inputs("source", "target", "tags")
        if cfg["kind"] in ("flat", "hier"):
            hh, ww, cc = cfg["image"][:3]
            series["maps"] = list(np.maximum(rng.normal(0, 1, (bsz, hh, ww, cc)), 0).astype(np.float32))
            inputs["maps"] = tf.placeholder(tf.float32, [None, hh, ww, cc], "maps")

        def build():
            enc, att, dec, parts = build_multisource(cfg)
            return dec, parts
    else:
        cfg = dict(TR_DEFAULT, **overrides)
        rng = np.random.default_rng(cfg["seed"])
        src = sentences(rng, cfg["batch"], cfg["src_vocab"], 1, 6, oov_every=2)
        tgt = sentences(rng, cfg["batch"], cfg["tgt_vocab"], 1, cfg["max_output_len"] + 1, oov_every=3)
        series = {"source": src, "target": tgt}
        if cfg["second_encoder"]:
            series["source2"] = sentences(rng, cfg["batch"], cfg["src_vocab"], 1, 5, oov_every=4)
        inputs = string_inputs("source", "target", "source2")

        def build():
            seq, enc, dec, parts = build_transformer(cfg)
            return dec, parts
    ds = dataset(series)
    bump = {}

    def factory(name, shape, np_dtype, initializer):
        value = variable_factory(name, shape, np_dtype, initializer)
        if name in bump:
            idx, delta = bump[name]
            value = value.copy()
            value.reshape(-1)[idx] += np.asarray(delta, value.dtype)
        return value

    def loss():
        fresh_graph()
        dec, parts = build()
        with tf_eager.feeding(feed(parts, ds, False, inputs)):
            return float(dec.train_loss.numpy()), dec, parts
    tf_eager.VARIABLE_FACTORY = factory
    try:
        base, dec, parts = loss()
        out = {"out/train_loss": np.asarray(base, np.float32)}
        with tf_eager.feeding(feed(parts, ds, False, inputs)):
            out["in/tgt_ids"] = dec.train_inputs.numpy()
            if family == "rnn" and cfg["spatial"] is not None:
                out["in/maps"] = np.stack(series["maps"])
            elif family == "rnn":
                out["in/src_ids"] = parts[1].inputs.numpy()
            elif family == "ms":
                out["in/src_ids"] = parts[1].input_factor_indices[0].numpy()
                if "maps" in series:
                    out["in/maps"] = np.stack(series["maps"])
            else:
                out["in/src_ids"] = parts[0].inputs.numpy()
                if cfg["second_encoder"]:
                    out["in/src2_ids"] = parts[2].inputs.numpy()
        order, params = variables()
        rng = np.random.default_rng(zlib.crc32(case.encode()))
        names, index, value = [], [], []
        for name in order:
            v = params[name]
            if v.dtype.kind != "f" or v.size == 0:
                continue
            picks = rng.choice(v.size, size=min(per_variable, v.size), replace=False)
            for i in picks:
                bump.clear()
                bump[name] = (int(i), +h)
                up = loss()[0]
                bump[name] = (int(i), -h)
                down = loss()[0]
                names.append(name)
                index.append(int(i))
                value.append((up - down) / (2.0 * h))
        bump.clear()
        loss()                                   # leave the unperturbed variables in the store for save()
        out["fd/names"] = np.asarray(names)
        out["fd/index"] = np.asarray(index, np.int64)
        out["fd/value"] = np.asarray(value, np.float64)
        out["fd/h"] = np.asarray(h)
    finally:
        tf_eager.VARIABLE_FACTORY = variable_factory
    save(case, dict(cfg, family=family), out)


# ----------------------------------------------------------------------------------------------------------------
# function-level cases with hand-made edge inputs
# ----------------------------------------------------------------------------------------------------------------
def run_functions(case):
    """Reference functions called directly: layer_norm, position_signal, the scaled-dot-product helpers, the
    Bahdanau step with an all-padding row, pad_batch, _length_penalty, the beam body on hand-made distributions with
    exact ties and finished hypotheses."""
    from neuralmonkey import tf_utils
    from neuralmonkey.encoders.transformer import position_signal
    from neuralmonkey.attention import scaled_dot_product as sdp
    from neuralmonkey.vocabulary import pad_batch
    rng = np.random.default_rng(11)
    out = {}
    fresh_graph()
    # tf_utils.layer_norm (:189-219)
    x = rng.normal(0, 2, (3, 4, 10)).astype(np.float32)
    x[0, 0] = 3.0                                              # a constant row: variance 0, eps decides
    with tf.variable_scope("ln_case", reuse=tf.AUTO_REUSE):
        out["in/ln_x"] = x
        out["out/ln_y"] = tf_utils.layer_norm(tf.constant(x)).numpy()
    # position_signal (encoders/transformer.py:23-45), even and odd dimension
    out["out/pos_8_5"] = position_signal(8, tf.constant(5)).numpy()
    out["out/pos_7_4"] = position_signal(7, tf.constant(4)).numpy()
    # split_for_heads / mask_energies / mask_future (scaled_dot_product.py:24-93)
    q = rng.normal(0, 1, (2, 3, 8)).astype(np.float32)
    out["in/heads_x"] = q
    out["out/heads_y"] = sdp.split_for_heads(tf.constant(q), 4, 2).numpy()
    e = rng.normal(0, 1, (2, 2, 3, 3)).astype(np.float32)
    m = np.asarray([[1, 1, 0], [0, 0, 0]], np.float32)          # second row: every key masked
    out["in/energies"], out["in/key_mask"] = e, m
    out["out/mask_energies"] = sdp.mask_energies(tf.constant(e), tf.constant(m)).numpy()
    out["out/mask_future"] = sdp.mask_future(tf.constant(e)).numpy()
    out["out/mask_future_then_keys"] = sdp.mask_energies(sdp.mask_future(tf.constant(e)), tf.constant(m)).numpy()
    # attention() without and with head projections, masked, all keys of a row masked (:98-226)
    keys = rng.normal(0, 1, (2, 3, 8)).astype(np.float32)
    out["in/sdp_q"], out["in/sdp_k"] = q, keys
    with tf.variable_scope("sdp1", reuse=tf.AUTO_REUSE):
        ctx, w = sdp.attention(tf.constant(q), tf.constant(keys), tf.constant(keys), tf.constant(m), 1,
                               lambda t: t, masked=True)
        out["out/sdp1_ctx"], out["out/sdp1_w"] = ctx.numpy(), w.numpy()
    with tf.variable_scope("sdp4", reuse=tf.AUTO_REUSE):
        ctx, w = sdp.attention(tf.constant(q), tf.constant(keys), tf.constant(keys), tf.constant(m), 4,
                               lambda t: t, masked=False, use_bias=True)
        out["out/sdp4_ctx"], out["out/sdp4_w"] = ctx.numpy(), w.numpy()
    # pad_batch (vocabulary.py:331-354)
    sents = [["a", "b", "c"], [], ["d"] * 6]
    for tag, kw in (("plain", {}), ("max4", {"max_length": 4}), ("end", {"add_end_symbol": True}),
                    ("end_max4", {"max_length": 4, "add_end_symbol": True}),
                    ("start_end_max4", {"max_length": 4, "add_start_symbol": True, "add_end_symbol": True})):
        out["out/pad_" + tag] = np.asarray(pad_batch(sents, **kw))
    save(case, {"kind": "functions"}, out)


def run_beam_body(case, vsz=7, k=3, max_steps=5, alpha=0.6, seed=5, rank=2):
    """``BeamSearchDecoder`` (beam_search_decoder.py:218-596) over a hand-made parent decoder whose step
    distributions are read from a table indexed by (step, previous symbol): exact score ties (TopK must take the
    lower flat index), hypotheses that finish at different steps, a sentence whose whole beam finishes early, the
    max_steps stop.  Only the beam logic is exercised, so every number is exact in float32."""
    from neuralmonkey.decoders.autoregressive import AutoregressiveDecoder
    from neuralmonkey.decoders.beam_search_decoder import BeamSearchDecoder
    from neuralmonkey.runners.beamsearch_runner import BeamSearchRunner
    bsz = 3
    rng = np.random.default_rng(seed)
    # logits[sentence, step, prev_symbol, :]: small multiples of 1/4 so that sums and ties are exact
    table = (rng.integers(-8, 9, (bsz, max_steps + 2, vsz, vsz)) / 4.0).astype(np.float32)
    first = np.full(vsz, -3.0, np.float32)
    first[3:6], first[6] = 1.0, 0.5
    table[0, 0, 1] = first                                       # sentence 0, step 0 from <s>: three-way exact tie
    table[0, 1, 3] = table[0, 1, 4]                              # identical continuations -> tied scores next step
    table[0, 1, 5, 2] = 6.0                                      # a hypothesis that ends at once
    table[1, 1:, :, 2] += 9.0                                    # sentence 1: the whole beam ends early
    table[2, :, :, 2] = -9.0                                     # sentence 2: never ends, runs into max_steps

    class TableDecoder(AutoregressiveDecoder):
        """Parent decoder whose logits are table[sentence, step, previous symbol]; the sentence index of a row
        travels in ``feedables.other`` (tiled by expand_to_beam, reordered by gather_flat like any state)."""

        def __init__(self):
            AutoregressiveDecoder.__init__(self, name="table", vocabulary=make_vocab(vsz - 4), data_id="target",
                                           max_output_len=max_steps + 1, embedding_size=vsz)

        @property
        def output_dimension(self):
            return vsz

        # identity "embedding" and identity output projection
        @property
        def embedding_matrix(self):
            return tf.constant(np.eye(vsz, dtype=np.float32))

        @property
        def decoding_w(self):
            return tf.constant(np.eye(vsz, dtype=np.float32))

        @property
        def decoding_b(self):
            return tf.zeros([vsz])

        def get_initial_histories(self):
            h = AutoregressiveDecoder.get_initial_histories(self)
            return h._replace(other=tf.zeros([]))

        def get_initial_feedables(self):
            f = AutoregressiveDecoder.get_initial_feedables(self)
            return f._replace(other=tf.reshape(tf.range(self.batch_size), [-1, 1]))

        def next_state(self, loop_state):
            step = loop_state.feedables.step
            sent = loop_state.feedables.other[:, 0]
            prev = tf.to_int32(tf.argmax(loop_state.feedables.embedded_input, axis=1))
            idx = tf.stack([sent, tf.fill(tf.shape(sent), step), prev], axis=1)
            logits = tf.gather_nd(tf.constant(table), idx)
            return logits, loop_state.feedables.other, loop_state.histories.other

    out = {"in/table": table}
    fresh_graph()
    dec = TableDecoder()
    bs = BeamSearchDecoder(name="beam", parent_decoder=dec, beam_size=k, max_steps=max_steps,
                           length_normalization=alpha)
    runner = BeamSearchRunner(output_series="hyp", decoder=bs, rank=rank)
    series = {"target": [["w0"]] * bsz}
    inputs = string_inputs("target")
    with tf_eager.feeding(feed([dec, bs], dataset(series), False, inputs)):
        out["out/length_penalty"] = bs._length_penalty(tf.constant(np.arange(0, 12, dtype=np.int32))).numpy()
        ex = runner.get_executable(compute_losses=False, summaries=False, num_sessions=1)
        fetches, _ = ex.next_to_execute()
        res = to_numpy(fetches)
        ex.collect_results([res])
        bo = res["bs_outputs"]
        out["out/scores"] = bo.last_search_step_output.scores
        out["out/token_ids"] = bo.last_search_step_output.token_ids
        out["out/logprob_sum"] = bo.last_search_state.logprob_sum
        out["out/lengths"] = bo.last_search_state.lengths
        out["out/finished"] = bo.last_search_state.finished
        out["out/dec_step"] = bo.last_dec_loop_state.feedables.step
        out["out/rank2_sentences"] = np.asarray([joined(s) for s in ex.result.outputs["hyp"]])
        out["out/rank2_loss"] = np.asarray(ex.result.losses["hyp/beam_search_score"])
    save(case, {"kind": "beam_body", "vocab": vsz, "beam": [k, max_steps, alpha], "batch": bsz, "rank": rank}, out)


# ----------------------------------------------------------------------------------------------------------------
# attention variants on the RNN decoder (row f3): combinations over two encoders, dot-product attention
# ----------------------------------------------------------------------------------------------------------------
MS_DEFAULT = dict(kind="flat", state_size=5, share=False, sentinel=False, image=[2, 3, 7, None, None],
                  src_vocab=17, tgt_vocab=13, emb=6, enc_size=4, rnn_size=6, dec_cell="GRU", conditional_gru=False,
                  max_output_len=6, seed=21, batch=4, heads=None, factored=False, label_smoothing=None)


def build_multisource(cfg):
    from neuralmonkey.encoders.recurrent import SentenceEncoder, FactoredEncoder
    from neuralmonkey.encoders.numpy_stateful_filler import SpatialFiller
    from neuralmonkey.attention.feed_forward import Attention
    from neuralmonkey.attention.combination import FlatMultiAttention, HierarchicalMultiAttention
    from neuralmonkey.attention.scaled_dot_product import MultiHeadAttention
    from neuralmonkey.decoders.decoder import Decoder
    sv, tv = make_vocab(cfg["src_vocab"]), make_vocab(cfg["tgt_vocab"])
    if cfg["factored"]:
        enc = FactoredEncoder(name="encoder", vocabularies=[sv, make_vocab(5)], data_ids=["source", "tags"],
                              embedding_sizes=[cfg["emb"], 3], rnn_size=cfg["enc_size"])
    else:
        enc = SentenceEncoder(name="encoder", vocabulary=sv, data_id="source", embedding_size=cfg["emb"],
                              rnn_size=cfg["enc_size"])
    parts = [enc, enc.input_sequence]
    encoders = [enc]
    if cfg["kind"] in ("flat", "hier"):
        h, w, c, ff_dim, proj_dim = cfg["image"]
        img = SpatialFiller(name="imagenet", input_shape=[h, w, c], data_id="maps", projection_dim=proj_dim,
                            ff_hidden_dim=ff_dim)
        parts.append(img)
        encoders.append(img)
    if cfg["kind"] == "flat":
        att = FlatMultiAttention(name="wrapper", encoders=encoders, attention_state_size=cfg["state_size"],
                                 share_attn_projections=cfg["share"], use_sentinels=cfg["sentinel"])
        parts.append(att)
    elif cfg["kind"] == "hier":
        children = [Attention(name="att_text", encoder=enc), Attention(name="att_image", encoder=img, state_size=7)]
        att = HierarchicalMultiAttention(name="wrapper", attentions=children, attention_state_size=cfg["state_size"],
                                         use_sentinels=cfg["sentinel"], share_attn_projections=cfg["share"])
        parts += children + [att]
    elif cfg["kind"] == "dotprod":
        att = MultiHeadAttention(name="attention", n_heads=cfg["heads"], keys_encoder=enc)
        parts.append(att)
    elif cfg["kind"] == "stateful":
        from neuralmonkey.attention.stateful_context import StatefulContext
        att = StatefulContext(name="attention", encoder=enc)
        parts.append(att)
    else:
        att = Attention(name="attention", encoder=enc)
        parts.append(att)
    dec = Decoder(encoders=encoders, vocabulary=tv, data_id="target", name="decoder",
                  max_output_len=cfg["max_output_len"], embedding_size=cfg["rnn_size"], rnn_size=cfg["rnn_size"],
                  attentions=[att], rnn_cell=cfg["dec_cell"], conditional_gru=cfg["conditional_gru"],
                  label_smoothing=cfg["label_smoothing"])
    parts.append(dec)
    return enc, att, dec, parts


def run_multisource(case, **overrides):
    cfg = dict(MS_DEFAULT, **overrides)
    rng = np.random.default_rng(cfg["seed"])
    bsz = cfg["batch"]
    src = sentences(rng, bsz, cfg["src_vocab"], 2, 6, oov_every=2)
    src[-1] = src[-1][:1]
    series = {"source": src, "target": sentences(rng, bsz, cfg["tgt_vocab"], 1, cfg["max_output_len"] + 1, oov_every=3)}
    inputs = string_inputs("source", "target", "tags")
    if cfg["factored"]:
        series["tags"] = [["w{}".format(int(rng.integers(0, 5))) for _ in s] for s in src]
    if cfg["kind"] in ("flat", "hier"):
        h, w, c = cfg["image"][:3]
        series["maps"] = list(np.maximum(rng.normal(0, 1, (bsz, h, w, c)), 0).astype(np.float32))
        inputs["maps"] = tf.placeholder(tf.float32, [None, h, w, c], "maps")
    out = {}
    fresh_graph()
    enc, att, dec, parts = build_multisource(cfg)
    with tf_eager.feeding(feed(parts, dataset(series), False, inputs)):
        out["in/src_ids"] = enc.input_sequence.input_factor_indices[0].numpy()
        if cfg["factored"]:
            out["in/tag_ids"] = enc.input_sequence.input_factor_indices[1].numpy()
        if "maps" in series:
            out["in/maps"] = np.stack(series["maps"])
        out["in/tgt_ids"] = dec.train_inputs.numpy()
        out["out/enc_states"] = enc.temporal_states.numpy()
        out["out/enc_output"] = enc.output.numpy()
        out["out/train_logits"] = dec.train_logits.numpy()
        out["out/train_xents"] = dec.train_xents.numpy()
        out["out/train_loss"] = dec.train_loss.numpy()
        rr = dec.runtime_loop_result
        out["out/runtime_logits"] = dec.runtime_logits.numpy()
        out["out/runtime_symbols"] = rr.histories.output_symbols.numpy()
        out["out/runtime_mask"] = dec.runtime_mask.numpy()
    save(case, cfg, out)


def run_ensemble(case, family="transformer"):
    """BeamSearchRunner over several sessions (runners/beamsearch_runner.py:38-82): every session advances its own
    model by ONE beam body per call, the step distributions are averaged in log space on the host (scipy logsumexp)
    and fed back -- driven here exactly as tf_manager.execute drives it (next_to_execute -> run -> collect_results
    until the executable has a result), on three sets of variables of the ``transformer`` fixture's model.  (Over an
    RNN ``Decoder`` the reference cannot run this protocol at this commit: ``fd = {self.decoder.decoder_state: ...}``
    (beamsearch_runner.py:70-75) uses a LoopState that holds LISTS (RNNFeedables.prev_contexts,
    RNNHistories.attention_histories) as a dictionary key -> ``TypeError: unhashable type: 'list'``; recorded by the
    ``defects`` case.)"""
    from neuralmonkey.decoders.beam_search_decoder import BeamSearchDecoder
    from neuralmonkey.runners.beamsearch_runner import BeamSearchRunner
    if family == "transformer":
        cfg = dict(TR_DEFAULT, seed=31, batch=3, beam=[3, 5, 0.6])
        rng = np.random.default_rng(cfg["seed"])
        series = {"source": sentences(rng, cfg["batch"], cfg["src_vocab"], 2, 6, oov_every=2),
                  "target": sentences(rng, cfg["batch"], cfg["tgt_vocab"], 1, 4)}

        def builder(c):
            seq, enc, dec, parts = build_transformer(c)
            return enc, None, dec, parts
    else:
        cfg = dict(RNN_DEFAULT, seed=31, batch=3, beam=[3, 5, 0.6])
        series = rnn_series(cfg)
        builder = build_rnn
    k, max_steps, alpha = cfg["beam"]
    n_models = 3
    out = {}
    inputs = string_inputs("source", "target")
    stores = []
    base_factory = tf_eager.VARIABLE_FACTORY
    for m in range(n_models):                       # the variables of "session" m: another seed per model
        tf_eager.VARIABLE_FACTORY = lambda name, shape, dt, init, m=m: base_factory("model{}/".format(m) + name,
                                                                                    shape, dt, init)
        fresh_graph()
        enc, att, dec, parts = builder(cfg)
        with tf_eager.feeding(feed(parts, dataset(series), False, inputs)):
            dec.runtime_logits.numpy()
        order, params = variables()
        stores.append(params)
        for n in order:
            out["p{}/{}".format(m, n)] = params[n]
    groups = [list(range(cfg["batch"]))] if family == "transformer" else [[i] for i in range(cfg["batch"])]
    for i, rows in enumerate(groups):
        one = {name: [vals[r] for r in rows] for name, vals in series.items()}
        # one graph per session, as TensorFlowManager holds one tf.Session (= one set of variable values) each
        sessions = []
        for m in range(n_models):
            factory = lambda name, shape, dt, init, m=m: stores[m][name]
            with tf_eager.session_store(None, factory) as store:      # variables are created lazily, at first use
                enc, att, dec, parts = builder(cfg)
                bs = BeamSearchDecoder(name="beam", parent_decoder=dec, beam_size=k, max_steps=max_steps,
                                       length_normalization=alpha)
                fd = feed(parts + [bs], dataset(one), False, inputs)
            sessions.append((bs, fd, store, factory))
        runner = BeamSearchRunner(output_series="hyp", decoder=sessions[0][0], rank=1)
        ex = runner.get_executable(compute_losses=False, summaries=False, num_sessions=n_models)
        first = sessions[0][0]
        calls = 0
        while ex.result is None:
            # (graph construction in TF; in the eager stand-in the first access of ``outputs`` COMPUTES: it has to
            # happen under the feeds of the first call -- max_steps 0: the initial loop state, no beam body)
            zero = dict(sessions[0][1])
            zero[first.max_steps] = 0
            with tf_eager.session_store(sessions[0][2], sessions[0][3]), tf_eager.feeding(zero):
                _, extra_feeds = ex.next_to_execute()
            results = []
            for (bs, fd, store, factory), extra in zip(sessions, extra_feeds):
                # the runner's feeds name tensors of ITS decoder (session 0's graph); every session's graph has the
                # same structure, fed by position
                mapped = {}
                for key, val in extra.items():
                    if key is first.max_steps:
                        mapped[bs.max_steps] = val
                    else:
                        with tf_eager.session_store(store, factory), tf_eager.feeding(fd):
                            mine = {id(first.search_state): lambda: bs.search_state,
                                    id(first.search_results): lambda: bs.search_results,
                                    id(first.decoder_state): lambda: bs.decoder_state}[id(key)]()
                        mapped[mine] = val
                base = dict(fd)
                base[bs.max_steps] = mapped.pop(bs.max_steps)
                with tf_eager.session_store(store, factory), tf_eager.feeding(base):
                    with tf_eager.feeding(mapped):
                        if calls == 0:
                            res = bs.outputs                # builds the initial loop state (one parent step) + loop
                        else:
                            # session.run again: the same loop from the FED state; the encoder states the parent
                            # attends to are tiled to the beam as ``outputs`` tiles them (beam_search_decoder.py:166-178)
                            par = bs.parent_decoder
                            es, em = par.encoder_states, par.encoder_masks
                            par.encoder_states = lambda es=es: [bs.expand_to_beam(x) for x in es()]
                            par.encoder_masks = lambda em=em: [bs.expand_to_beam(x) for x in em()]
                            try:
                                with bs.use_scope():
                                    res = bs.decoding_loop()
                            finally:
                                par.encoder_states, par.encoder_masks = es, em
                        results.append(to_numpy({"bs_outputs": res}))
            ex.collect_results(results)
            calls += 1
        pre = "out/ens{}_".format(i)
        out[pre + "calls"] = np.asarray(calls)
        out[pre + "sentence"] = np.asarray([joined(sent) for sent in ex.result.outputs["hyp"]])
        out[pre + "loss"] = np.asarray(ex.result.losses["hyp/beam_search_score"])
        last = results[0]["bs_outputs"]
        out[pre + "token_ids"] = last.last_search_step_output.token_ids
        out[pre + "scores"] = last.last_search_step_output.scores
    tf_eager.VARIABLE_FACTORY = base_factory
    fresh_graph()
    out["in/src_ids_note"] = np.asarray("sources / targets are those of rnn_series(cfg) with cfg in the 'cfg' entry")
    with tf_eager.feeding({}):
        pass
    # source ids for the oracle
    fresh_graph()
    enc, att, dec, parts = builder(cfg)
    with tf_eager.feeding(feed(parts, dataset(series), False, inputs)):
        out["in/src_ids"] = enc.input_sequence.inputs.numpy()
        out["out/enc_mask"] = enc.temporal_mask.numpy()
    fresh_graph()
    save(case, dict(cfg, n_models=n_models, family=family), out)


def run_greedy_runner_ensemble(case):
    """``GreedyRunner.Executable.collect_results`` (runners/runner.py:35-63) over several sessions: the [T,B,V]
    log-probabilities of the sessions are combined step by step with ``np.logaddexp`` into a list as long as
    SESSION 0'S loop -- a session that stopped earlier contributes to its own steps only, one that ran longer makes
    the list assignment fail.  Plain NumPy in the reference: called here on hand-made session results."""
    from neuralmonkey.runners.runner import GreedyRunner
    cfg = dict(RNN_DEFAULT)
    fresh_graph()
    enc, att, dec, parts = build_rnn(cfg)
    runner = GreedyRunner(output_series="target", decoder=dec)
    rng = np.random.default_rng(41)
    bsz, vsz = 4, cfg["tgt_vocab"] + 4

    def session(steps, xents):
        lg = rng.normal(0, 2.0, (steps, bsz, vsz)).astype(np.float32)
        lp = lg - np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1, keepdims=True)) - lg.max(-1, keepdims=True)
        return {"decoded_logprobs": lp.astype(np.float32), "train_xent": np.float32(xents[0]),
                "runtime_xent": np.float32(xents[1])}
    out = {}
    for tag, lengths in (("equal", (5, 5, 5)), ("shorter", (5, 3, 4)), ("longer", (3, 5, 3))):
        results = [session(n, (1.25 + i, 2.5 + i)) for i, n in enumerate(lengths)]
        for i, res in enumerate(results):
            out["in/{}_logprobs{}".format(tag, i)] = res["decoded_logprobs"]
            out["in/{}_xents{}".format(tag, i)] = np.asarray([res["train_xent"], res["runtime_xent"]])
        ex = runner.get_executable(compute_losses=True, summaries=False, num_sessions=len(results))
        try:
            ex.collect_results(results)
            out["out/{}_sentences".format(tag)] = np.asarray([joined(s) for s in ex.result.outputs["target"]])
            out["out/{}_losses".format(tag)] = np.asarray([ex.result.losses["target/train_xent"],
                                                           ex.result.losses["target/runtime_xent"]], np.float32)
            out["out/{}_error".format(tag)] = np.asarray("")
        except Exception as exc:        # noqa: BLE001 -- whatever it raises is the reference's behaviour
            out["out/{}_error".format(tag)] = np.asarray("{}: {}".format(type(exc).__name__, exc))
    save(case, {"kind": "greedy_runner_ensemble", "tgt_vocab": cfg["tgt_vocab"], "batch": bsz}, out)


def canonical_rows(rows):
    """TensorRunner outputs as JSON-able structures: arrays -> nested lists, dicts keep their key order."""
    def one(v):
        if isinstance(v, dict):
            return {"dict": [[k, one(x)] for k, x in v.items()]}
        if isinstance(v, (list, tuple)):
            return {type(v).__name__: [one(x) for x in v]}
        return {"array": np.asarray(v).tolist()}
    return one(rows)


def run_tensor_runner(case):
    """``TensorRunner.Executable.collect_results`` (runners/tensor_runner.py:24-55, plain NumPy in the reference) on
    hand-made session results: the batch axis of every fetched tensor moved to the front and cut into one entry per
    example, as dictionaries or -- ``single_tensor`` -- bare arrays; several sessions zipped example by example, or,
    with ``select_session`` set, SESSION 0'S RESULTS whatever the number says (:27-34).  The runner's constructor
    checks (:105-118) with their texts; ``RepresentationRunner`` is a single-tensor TensorRunner."""
    from neuralmonkey.runners.tensor_runner import RepresentationRunner, TensorRunner
    cfg = dict(RNN_DEFAULT)
    fresh_graph()
    enc, att, dec, parts = build_rnn(cfg)
    rng = np.random.default_rng(47)
    sessions = [{"a": rng.normal(size=(3, 2, 4)).astype(np.float32), "b": rng.normal(size=(5, 3)).astype(np.float32)}
                for _ in range(3)]
    out = {}
    for i, res in enumerate(sessions):
        out["in/a{}".format(i)], out["in/b{}".format(i)] = res["a"], res["b"]
    settings = collections.OrderedDict([
        ("one_session", dict(n=1, select=None, single=False)), ("three_sessions", dict(n=3, select=None, single=False)),
        ("three_sessions_select_2", dict(n=3, select=2, single=False)),
        ("single_tensor", dict(n=1, select=None, single=True)),
        ("single_tensor_three_sessions", dict(n=3, select=None, single=True))])
    record = {}
    for tag, st in settings.items():
        names = ["a"] if st["single"] else ["a", "b"]
        runner = TensorRunner(output_series="dbg", modelparts=[enc] * len(names),
                              tensors=["temporal_states"] * len(names), batch_dims=[0] * len(names),
                              tensors_by_name=[], batch_dims_by_name=[], select_session=st["select"],
                              single_tensor=st["single"])
        runner.batch_ids = {"a": 0, "b": 1}                 # (what ``fetches`` would have filled in: a's batch axis 0, b's 1)
        ex = runner.get_executable(compute_losses=False, summaries=False, num_sessions=st["n"])
        ex.collect_results([{k: res[k] for k in names} for res in sessions[:st["n"]]])
        record[tag] = {"outputs": canonical_rows(ex.result.outputs["dbg"]), "losses": dict(ex.result.losses)}
    errors = {}
    for tag, kw in (("no_parts", dict(modelparts=[], tensors=[], batch_dims=[])),
                    ("lengths", dict(modelparts=[enc], tensors=["output", "temporal_states"], batch_dims=[0, 0])),
                    ("single_of_two", dict(modelparts=[enc, enc], tensors=["output", "temporal_states"],
                                           batch_dims=[0, 0], single_tensor=True))):
        try:
            TensorRunner(output_series="dbg", tensors_by_name=[], batch_dims_by_name=[], **kw)
            errors[tag] = ""
        except Exception as exc:        # noqa: BLE001
            errors[tag] = "{}: {}".format(type(exc).__name__, exc)
    rep = RepresentationRunner(output_series="encoded", encoder=enc)
    record["representation"] = {"single_tensor": rep.single_tensor, "batch_dims": rep.batch_dims,
                                "tensors": rep._tensors, "loss_names": rep.loss_names}
    out["out/record"] = np.asarray(json.dumps(record))
    out["out/errors"] = np.asarray(json.dumps(errors))
    save(case, {"kind": "tensor_runner"}, out)


LAZY_SHUFFLE_SETTINGS = collections.OrderedDict([
    ("eager_shuffled", dict(scheme=dict(batch_size=4), buffer=None, shuffled=True)),
    ("eager_shuffled_buckets", dict(scheme=dict(bucket_boundaries=[3, 6], bucket_batch_sizes=[3, 2, 4]), buffer=None,
                                    shuffled=True)),
    ("lazy", dict(scheme=dict(batch_size=4), buffer=[6, 10], shuffled=False)),
    ("lazy_buckets_drop", dict(scheme=dict(bucket_boundaries=[3, 6], bucket_batch_sizes=[3, 2, 4], drop_remainder=True),
                               buffer=[6, 10], shuffled=False)),
    ("lazy_shuffled", dict(scheme=dict(batch_size=4), buffer=[6, 10], shuffled=True)),
    ("lazy_shuffled_buckets", dict(scheme=dict(bucket_boundaries=[3, 6], bucket_batch_sizes=[3, 2, 4]), buffer=[5, 9],
                                   shuffled=True)),
    ("lazy_small_low_mark", dict(scheme=dict(batch_size=4), buffer=[2, 7], shuffled=True)),
])


def run_dataset_lazy_shuffle(case):
    """``Dataset`` (dataset.py:335-640; no TensorFlow) as a LAZY and as a SHUFFLED dataset: rows drawn into a buffer
    of ``buffer_size`` that is topped up when fewer than ``buffer_min_size`` are left, ``random.shuffle`` of all rows
    (eager) or of the buffer at every top-up (lazy), two passes after one ``random.seed(11)`` each; ``len()`` of a
    lazy dataset, ``subset`` (rows 5..16, passed once), and how often the series' factories are opened."""
    fresh_graph()                 # (no variables of an earlier case in this fixture)
    import random
    from neuralmonkey.dataset import BatchingScheme, Dataset
    rng = np.random.default_rng(53)
    rows = 23
    src = [["s{}".format(i)] + ["x"] * int(rng.integers(0, 8)) for i in range(rows)]
    tgt = [["t{}".format(i)] + ["y"] * int(rng.integers(0, 7)) for i in range(rows)]
    out = {"in/source_lengths": np.asarray([len(s) for s in src]), "in/target_lengths": np.asarray([len(t) for t in tgt])}
    record = {}
    for tag, st in LAZY_SHUFFLE_SETTINGS.items():
        opened = {"source": 0, "target": 0}

        def factory(key, items):
            def open_series():
                opened[key] += 1
                return iter(items)
            return open_series
        ds = Dataset("data", {"source": factory("source", src), "target": factory("target", tgt)},
                     BatchingScheme(**st["scheme"]), None, None if st["buffer"] is None else tuple(st["buffer"]),
                     st["shuffled"])
        after_init = dict(opened)
        random.seed(11)
        passes, names = [], []
        for _ in range(2):
            batches = []
            for b in ds.batches():
                ids = [int(row[0][1:]) for row in b.get_series("source")]
                assert ids == [int(row[0][1:]) for row in b.get_series("target")]
                batches.append(ids)
                names.append(b.name)
            passes.append(batches)
        try:
            length = len(ds)
        except NotImplementedError as exc:
            length = "NotImplementedError: {}".format(exc)
        random.seed(11)
        sub = ds.subset(5, 11)
        record[tag] = {"passes": passes, "names": names[:3], "len": length, "opened_at_init": after_init,
                       "opened_after_two_passes": dict(opened), "series": ds.series, "lazy": ds.lazy,
                       "subset_name": sub.name, "subset_lazy": sub.lazy,
                       "subset_batches": [[int(row[0][1:]) for row in b.get_series("source")] for b in sub.batches()]}
    out["out/record"] = np.asarray(json.dumps(record))
    save(case, {"kind": "dataset_lazy_shuffle", "rows": rows, "settings": LAZY_SHUFFLE_SETTINGS}, out)


def run_dataset_batching(case):
    """``Dataset.batches`` (dataset.py:467-579): fixed-size batches and length buckets (a row goes to the TIGHTEST
    bucket that fits the longest of its series, to the last one when none does), with and without the remainder.
    No TensorFlow involved: the reference's own host code on seeded sentences."""
    fresh_graph()                 # (no variables of an earlier case in this fixture)
    from neuralmonkey.dataset import BatchingScheme, Dataset
    rng = np.random.default_rng(43)
    rows = 41
    src = [["s{}".format(i)] + ["x"] * int(rng.integers(0, 11)) for i in range(rows)]
    tgt = [["t{}".format(i)] + ["y"] * int(rng.integers(0, 9)) for i in range(rows)]
    schemes = {
        "fixed": dict(batch_size=7),
        "fixed_drop": dict(batch_size=7, drop_remainder=True),
        "buckets": dict(bucket_boundaries=[3, 6, 9], bucket_batch_sizes=[4, 3, 2, 5]),
        "buckets_drop": dict(bucket_boundaries=[3, 6, 9], bucket_batch_sizes=[4, 3, 2, 5], drop_remainder=True),
        "buckets_unsorted": dict(bucket_boundaries=[6, 3, 9], bucket_batch_sizes=[3, 4, 2, 5]),
        # ``ignore_series`` is stored and never read at this commit (dataset.py:521 "TODO: use only specific series"):
        # the longest of ALL series decides
        "buckets_ignore": dict(bucket_boundaries=[3, 6, 9], bucket_batch_sizes=[4, 3, 2, 5], ignore_series=["target"]),
    }
    out = {"in/source_lengths": np.asarray([len(s) for s in src]), "in/target_lengths": np.asarray([len(t) for t in tgt])}
    for tag, kw in schemes.items():
        ds = Dataset("data", {"source": lambda: iter(src), "target": lambda: iter(tgt)}, BatchingScheme(**kw))
        batches = []
        for b in ds.batches():
            ids = [int(row[0][1:]) for row in b.get_series("source")]
            assert ids == [int(row[0][1:]) for row in b.get_series("target")]
            batches.append(ids)
        out["out/{}_batch_of_row".format(tag)] = np.asarray(
            [next((j for j, ids in enumerate(batches) if i in ids), -1) for i in range(rows)])
        out["out/{}_order".format(tag)] = np.asarray([i for ids in batches for i in ids])
        out["out/{}_sizes".format(tag)] = np.asarray([len(ids) for ids in batches])
    save(case, {"kind": "dataset_batching", "rows": rows, "schemes": schemes}, out)


VOCAB_FILES = {
    # wordlist with header and frequencies (vocabulary.py:32-99), the format save_wordlist writes
    "wordlist_header": "word\tcount\nthe\t120\ncat\t7\nsat\t3\n\u010de\u0161tina\t1\n",
    # bare wordlist: no header, no second column
    "wordlist_plain": "alpha\nbeta\ngamma\n",
    # tensor2tensor vocabulary (vocabulary.py:102-134): quoted tokens, <pad> and <EOS> first
    "t2t": "'<pad>'\n'<EOS>'\n'hello_'\n'wor'\n'ld_'\n'\\u;_'\n",
    # Nematus JSON (vocabulary.py:137-187): word -> index, eos / UNK at 0 / 1
    "nematus": '{"eos": 0, "UNK": 1, "der": 2, "die": 3, "das": 4, "und": 5, "ist": 6}',
}


def run_vocabulary_formats(case):
    """The vocabulary loaders and ``vectors_to_sentences`` of the reference (vocabulary.py, no TensorFlow involved) on
    small files whose text is kept in the fixture: the product's loaders are run on the same text by the CPU test."""
    fresh_graph()                 # (no variables of an earlier case in this fixture)
    import tempfile
    from neuralmonkey import vocabulary as V
    out = {}
    loaded = {}
    with tempfile.TemporaryDirectory() as tmp:
        paths = {}
        for name, text in VOCAB_FILES.items():
            paths[name] = os.path.join(tmp, name)
            with open(paths[name], "w", encoding="utf-8") as handle:
                handle.write(text)
        loaded["wordlist_header"] = V.from_wordlist(paths["wordlist_header"])
        loaded["wordlist_plain"] = V.from_wordlist(paths["wordlist_plain"], contains_header=False,
                                                   contains_frequencies=False)
        loaded["t2t"] = V.from_t2t_vocabulary(paths["t2t"])
        loaded["nematus"] = V.from_nematus_json(paths["nematus"])
        loaded["nematus_max5"] = V.from_nematus_json(paths["nematus"], max_size=5)
        loaded["nematus_pad9"] = V.from_nematus_json(paths["nematus"], max_size=9, pad_to_max_size=True)
        # save_wordlist (:290-320): what it writes, that it refuses to overwrite, and that its output loads back
        saved = os.path.join(tmp, "saved.tsv")
        loaded["wordlist_header"].save_wordlist(saved)
        with open(saved, encoding="utf-8") as handle:
            out["out/saved_wordlist"] = np.asarray(handle.read())
        try:
            loaded["wordlist_header"].save_wordlist(saved)
            out["out/save_again_error"] = np.asarray("")
        except Exception as exc:        # noqa: BLE001
            out["out/save_again_error"] = np.asarray("{}: {}".format(type(exc).__name__, str(exc).replace(tmp, "<dir>")))
        loaded["wordlist_header"].save_wordlist(saved, overwrite=True)
        out["out/saved_reloaded_words"] = np.asarray(list(V.from_wordlist(
            saved, contains_header=True, contains_frequencies=False).index_to_word))
    for name, vocab in loaded.items():
        out["out/{}_words".format(name)] = np.asarray(list(vocab.index_to_word))
    # vectors_to_sentences (:257-288): time-major ids, every sentence cut at its first </s>, <pad> kept as a word
    vocab = loaded["wordlist_header"]
    rng = np.random.default_rng(47)
    ids = rng.integers(0, len(vocab), (6, 5))
    ids[2, 1] = 2                # </s> in the middle
    ids[0, 3] = 2                # </s> first: an empty sentence
    ids[:, 4] = [4, 0, 5, 0, 2, 6]
    out["in/time_major_ids"] = ids
    out["out/sentences_array"] = np.asarray([" ".join(s) for s in vocab.vectors_to_sentences(ids)])
    out["out/sentences_list"] = np.asarray([" ".join(s) for s in vocab.vectors_to_sentences([row for row in ids])])
    save(case, {"kind": "vocabulary_formats", "files": VOCAB_FILES}, out)


TEXT_FILES = {
    "plain.txt": "the  cat sat\n\n  leading and trailing  \nÜnïcödé wörds ok\n",
    "t2t.txt": "Hello, world! It's 3.5 (approx.)\nDon't  stop\u2014now\n",
    "table.tsv": "a b\tfirst col\t1\nc d e\tsecond \"quoted\" col\t2\n",
    "table.csv": "a b,\"x, y\",1\nc,\"he said \"\"hi\"\"\",2\n",
    "vectors.txt": "1 2.5 -3\n0.125 4e2 5\n",
}


def edit_pairs():
    """Seeded sentence pairs over a five-word alphabet (many equally cheap alignments), with the empty sentence, equal
    sentences, disjoint ones, and the two special tokens occurring as ordinary words."""
    rng = np.random.default_rng(77)
    words = ["a", "b", "c", "d", "e"]
    pairs = [([], []), ([], ["a", "b"]), (["a", "b"], []), (["a"], ["a"]), (["a", "b", "c"], ["a", "b", "c"]),
             (["a", "b"], ["c", "d"]), (["a", "a", "a"], ["a"]), (["a"], ["a", "a", "a"]),
             (["<keep>", "a"], ["a", "<delete>"]), (["a", "b", "a", "b"], ["b", "a", "b", "a"])]
    for _ in range(150):
        pairs.append(([words[i] for i in rng.integers(0, 5, rng.integers(0, 9))],
                      [words[i] for i in rng.integers(0, 5, rng.integers(0, 9))]))
    return pairs


def run_editops(case):
    """processors/editops.py (the post-editing scripts of tests/post-edit.ini; no TensorFlow): ``convert_to_edits``
    on seeded pairs, ``reconstruct`` on those scripts, on scripts cut short (the decoder's length limit) and on
    scripts that keep / delete beyond the end of the sentence; ``Preprocess`` / ``Postprocess`` as the dataset and
    [main] call them, and the errors of ``Postprocess``."""
    fresh_graph()                 # (no variables of an earlier case in this fixture)
    from neuralmonkey.processors import editops as E
    pairs = edit_pairs()
    join = lambda rows: np.asarray(["\x1f".join(r) for r in rows])
    scripts = [E.convert_to_edits(list(a), list(b)) for a, b in pairs]
    out = {"in/source": join([a for a, _ in pairs]), "in/target": join([b for _, b in pairs]),
           "out/scripts": join(scripts),
           "out/rebuilt": join([E.reconstruct(list(a), list(sc)) for (a, _), sc in zip(pairs, scripts)]),
           "out/rebuilt_cut": join([E.reconstruct(list(a), list(sc[:len(sc) // 2])) for (a, _), sc in zip(pairs, scripts)]),
           "out/rebuilt_long": join([E.reconstruct(list(a), list(sc) + ["<keep>", "z", "<delete>", "<keep>"])
                                     for (a, _), sc in zip(pairs, scripts)])}
    series = {"mt": lambda: iter([list(a) for a, _ in pairs]), "pe": lambda: iter([list(b) for _, b in pairs])}
    out["out/preprocess"] = join(list(E.Preprocess("mt", "pe")(series)))
    post = E.Postprocess("mt", "edits")
    out["out/postprocess"] = join(post({"mt": [list(a) for a, _ in pairs]}, {"edits": scripts}))
    errors = []
    for dataset, generated in (({}, {"edits": []}), ({"mt": []}, {})):
        try:
            post(dataset, generated)
            errors.append("")
        except Exception as exc:        # noqa: BLE001
            errors.append("{}: {}".format(type(exc).__name__, exc))
    out["out/errors"] = np.asarray(errors)
    save(case, {"kind": "editops"}, out)


def run_host_text_pipeline(case):
    """Readers and string processors of the reference, none of which touches TensorFlow: plain_text_reader.py:23-134
    (whitespace tokens, the tensor2tensor tokenizer, column readers), string_vector_reader.py:6-40,
    processors/helpers.py:5-52, processors/wordpiece.py:22-130 -- on files whose text the fixture carries."""
    fresh_graph()                 # (no variables of an earlier case in this fixture)
    import tempfile
    from neuralmonkey.readers import plain_text_reader as R
    from neuralmonkey.readers.string_vector_reader import get_string_vector_reader
    from neuralmonkey.processors import helpers as H
    from neuralmonkey.processors import wordpiece as W
    from neuralmonkey.vocabulary import Vocabulary
    out = {}
    join = lambda rows: np.asarray(["\x1f".join(r) for r in rows])          # unit separator: tokens may hold spaces
    with tempfile.TemporaryDirectory() as tmp:
        paths = {}
        for name, text in TEXT_FILES.items():
            paths[name] = os.path.join(tmp, name)
            with open(paths[name], "w", encoding="utf-8") as handle:
                handle.write(text)
        out["out/string_reader"] = np.asarray(list(R.string_reader()([paths["plain.txt"]])))
        out["out/tokenized"] = join(R.tokenized_text_reader()([paths["plain.txt"], paths["t2t.txt"]]))
        out["out/t2t_tokenized"] = join(R.t2t_tokenized_text_reader()([paths["t2t.txt"]]))
        out["out/tsv_col1"] = join(R.tsv_reader(1)([paths["table.tsv"]]))
        out["out/tsv_col2"] = join(R.tsv_reader(2)([paths["table.tsv"]]))
        out["out/csv_col2"] = join(R.csv_reader(2)([paths["table.csv"]]))
        out["out/vectors"] = np.stack(list(get_string_vector_reader()([paths["vectors.txt"]])))
    sents = [["the", "cat"], ["Ünï", "x"], [], ["a"]]
    out["out/char_based"] = join([H.preprocess_char_based(s) for s in sents])
    out["out/char_based_back"] = join(H.postprocess_char_based([H.preprocess_char_based(s) for s in sents]))
    out["out/untruecase"] = join(list(H.untruecase([["hello", "World"], ["x"], []])))
    out["out/pipeline"] = join([H.pipeline([H.preprocess_char_based, lambda s: s[::-1]])(["ab", "c"])])
    # wordpieces over a vocabulary of pieces (the tensor2tensor scheme: underscore ends a word, escapes)
    pieces = ["the_", "c", "a", "t", "t_", "at_", "s", "sa", "_", "\\", "u", "9", "5", ";", "x", "1", "2", "3", "4",
              "6", "7", "8", "0"]
    vocab = Vocabulary(pieces)
    enc = [W.wordpiece_encode(s, vocab) for s in (["the", "cat"], ["sat", "cat"], ["ta_t"], ["\u00e9x"])]
    out["in/wordpiece_vocab"] = np.asarray(pieces)
    out["out/wordpiece_encoded"] = join(enc)
    out["out/wordpiece_decoded"] = join([W.wordpiece_decode(e) for e in enc])
    out["out/escape"] = np.asarray([W.escape_token(t, set("abc_\\;0123456789u")) for t in ("abc", "a_b", "a\\b", "\u00e9")])
    out["out/unescape"] = np.asarray([W.unescape_token(t) for t in ("abc_", "a\\ub_", "a\\\\b_", "\\233;_", "\\x;_")])
    save(case, {"kind": "host_text_pipeline", "files": TEXT_FILES}, out)


def run_schedules(case):
    """functions.py:9-80 (the schedules INI files name for learning rates and sampling probabilities), evaluated by the
    reference against the graph's global step."""
    from neuralmonkey import functions as F
    fresh_graph()
    gs = tf.train.get_or_create_global_step()
    steps = [0, 1, 2, 5, 99, 100, 101, 3999, 4000, 4001, 5000, 10000, 100000]
    rows = {"noam_512_4000": [], "noam_64_10": [], "inverse_sigmoid_300": [], "inverse_sigmoid_2_scaled": [],
            "piecewise": []}
    for step in steps:
        gs.assign(step)
        x = tf.to_float(gs)
        rows["noam_512_4000"].append(F.noam_decay(0.2, 512, 4000).numpy())
        rows["noam_64_10"].append(F.noam_decay(1.0, 64, 10).numpy())
        rows["inverse_sigmoid_300"].append(F.inverse_sigmoid_decay(x, 300.0).numpy())
        rows["inverse_sigmoid_2_scaled"].append(F.inverse_sigmoid_decay(x / 1000.0, 2.0, 0.1, 0.9).numpy())
        rows["piecewise"].append(F.piecewise_function(x, [1.0, 0.5, 0.1], [100, 5000]).numpy())
    out = {"in/steps": np.asarray(steps, np.int64)}
    for name, vals in rows.items():
        out["out/" + name] = np.asarray(vals, np.float32)
    try:
        F.piecewise_function(tf.to_float(gs), [1.0, 0.5], [1, 2])
        out["out/piecewise_error"] = np.asarray("")
    except ValueError as exc:
        out["out/piecewise_error"] = np.asarray(str(exc))
    fresh_graph()
    save(case, {"kind": "schedules"}, out)


def canonical_ini_value(value):
    """A parsed INI value as JSON-able structure: class symbols and object references by what they name."""
    kind = type(value).__name__
    if kind == "ClassSymbol":
        return {"class": value.clazz}
    if kind == "ObjectRef":
        return {"object": value.expression}
    if isinstance(value, (list, tuple)):
        return {"list" if isinstance(value, list) else "tuple": [canonical_ini_value(v) for v in value]}
    if isinstance(value, (bool, int, float, str)) or value is None:
        return {type(value).__name__: value}
    raise TypeError("unexpected parsed value {!r}".format(value))


def run_ini_grammar(case):
    """``config/parsing.py:parse_file`` (the INI value grammar: numbers, strings with $variables, lists, tuples, class
    symbols, <object.attribute> references, the [vars] section) on every configuration file of the reference's own
    test suite -- no TensorFlow involved.  The files' text is kept in the fixture; $TIME is pinned."""
    fresh_graph()                 # (no variables of an earlier case in this fixture)
    import glob
    import time as time_module
    from neuralmonkey.config import parsing
    real = time_module.strftime
    time_module.strftime = lambda fmt, *a: "TIME"
    os.environ["NM_EXPERIMENT_NAME"] = "exp-7"          # a [vars]-less variable some files take from the environment
    out, files = {}, {}
    try:
        for path in sorted(glob.glob(os.path.join(REFERENCE, "tests", "*.ini"))):
            name = os.path.basename(path)
            with open(path, encoding="utf-8") as handle:
                text = handle.read()
            files[name] = text
            raw, parsed = parsing.parse_file(text.splitlines(True))
            out["out/" + name] = np.asarray(json.dumps(
                {sec: {k: canonical_ini_value(v) for k, v in body.items()} for sec, body in parsed.items()},
                sort_keys=True))
            out["raw/" + name] = np.asarray(json.dumps(raw, sort_keys=True))
        # the value grammar probe by probe (each in a file of its own): what it parses to, or the error it gives
        probes = ['1', '-4', '2.5e-3', '1e3', '-.5', '.5', '1.', '"s t"', '"pre-$x-{x}"', '"{nowhere_defined}"', '$x',
                  '[1, 2]', '[]', '()', '(3,)', '(1, 2)', '[ [1,2] , (3, 4) ]', '["s, t"]', '[1, 2', 'tf.nn.relu',
                  'a.b', 'a', '<obj>', '<obj.attr.b>', 'None', 'True', 'False', 'true', '1 2', '"unterminated',
                  '[1,,2]', '[1, 2,]', '0x10', '1_000', '-', '"a" "b"', "'single'"]
        files["_probes"] = json.dumps(probes)
        results = []
        for probe in probes:
            text = "[vars]\nx=3\n[main]\nv={}\n".format(probe)
            try:
                _, parsed = parsing.parse_file(text.splitlines(True))
                results.append({"value": canonical_ini_value(parsed["main"]["v"])})
            except Exception as exc:        # noqa: BLE001
                results.append({"error": "{}: {}".format(type(exc).__name__, exc)})
        out["out/_probes"] = np.asarray(json.dumps(results, sort_keys=True))
    finally:
        time_module.strftime = real
    save(case, {"kind": "ini_grammar", "files": files}, out)


BUILDER_INI = """[vars]
n=3
[main]
pair=(<frac>, <frac.numerator>)
items=[<frac>, <other>, $n, "s"]
tf_manager=<first>
zeta=<other>
alpha=<third.tag>
[frac]
class=fractions.Fraction
numerator=3
denominator=4
[other]
class=argparse.Namespace
a=<frac>
b=[1, <frac.denominator>]
[first]
class=argparse.Namespace
tag="first"
[third]
class=argparse.Namespace
tag=fractions.Fraction
[unused]
class=argparse.Namespace
"""

BUILDER_ERRORS = {
    "undefined_object": "[main]\na=<nope>\n",
    "no_class": "[main]\na=<sec>\n[sec]\nx=1\n",
    "not_callable": "[main]\na=<sec>\n[sec]\nclass=math.pi\n",
    "bad_kwargs": "[main]\na=<sec>\n[sec]\nclass=fractions.Fraction\nnumerator=1\nno_such_argument=2\n",
    "cycle": "[main]\na=<x>\n[x]\nclass=argparse.Namespace\ny=<y>\n[y]\nclass=argparse.Namespace\nx=<x>\n",
    "constructor_raises": "[main]\na=<sec>\n[sec]\nclass=fractions.Fraction\nnumerator=1\ndenominator=0\n",
    "no_main": "[other]\nclass=argparse.Namespace\n",
}


def describe_built(value):
    """A built configuration value in words: type name + content, recursively (objects by their repr)."""
    if isinstance(value, (list, tuple)):
        return {type(value).__name__: [describe_built(v) for v in value]}
    if isinstance(value, type):
        return {"class": "{}.{}".format(value.__module__, value.__qualname__)}
    if type(value).__name__ == "Namespace":
        return {"Namespace": {k: describe_built(v) for k, v in sorted(vars(value).items())}}
    return {type(value).__name__: repr(value)}


def run_config_builder(case):
    """``config/builder.py:build_config`` (object references resolved recursively and once, attribute chains,
    class symbols, ``tf_manager`` built last, the errors of bad configurations) on a configuration that names only
    standard-library callables, so that the product's builder can be run on the same text."""
    fresh_graph()                 # (no variables of an earlier case in this fixture)
    import collections.abc
    collections.Iterable = collections.abc.Iterable          # (builder.py:114 uses the pre-3.10 alias)
    from neuralmonkey.config import parsing
    from neuralmonkey.config.builder import build_config
    out = {}
    _, parsed = parsing.parse_file(BUILDER_INI.splitlines(True))
    configuration, existing = build_config(parsed, ignore_names=set(), warn_unused=True)
    out["out/configuration"] = np.asarray(json.dumps({k: describe_built(v) for k, v in configuration.items()}))
    out["out/configuration_order"] = np.asarray(list(configuration))
    out["out/construction_order"] = np.asarray(list(existing))
    out["out/shared_identity"] = np.asarray([configuration["pair"][0] is configuration["items"][0],
                                             configuration["items"][1] is configuration["zeta"],
                                             configuration["items"][1].a is configuration["pair"][0]])
    _, parsed = parsing.parse_file(BUILDER_INI.splitlines(True))
    ignored, _ = build_config(parsed, ignore_names={"zeta", "items"}, warn_unused=False)
    out["out/ignored_order"] = np.asarray(list(ignored))
    errors = {}
    for tag, text in BUILDER_ERRORS.items():
        _, parsed = parsing.parse_file(text.splitlines(True))
        try:
            build_config(parsed, ignore_names=set())
            errors[tag] = ""
        except BaseException as exc:        # noqa: BLE001
            inner = getattr(exc, "original_exception", None)
            errors[tag] = {"type": type(exc).__name__, "object_name": str(getattr(exc, "object_name", "")),
                           "inner_type": type(inner).__name__ if inner is not None else "",
                           # (a nested ConfigBuildException prints a traceback with this machine's paths: cut)
                           "inner_text": (str(inner) if inner is not None else str(exc)).split("\nTraceback")[0]}
    out["out/errors"] = np.asarray(json.dumps(errors, sort_keys=True))
    save(case, {"kind": "config_builder", "ini": BUILDER_INI, "errors": BUILDER_ERRORS}, out)


LOAD_FILES = {
    "train.a.src": "the cat\nsat on\n",
    "train.b.src": "a mat\n",
    "train.tgt": "le chat\nassis sur\nun tapis\n",
    "short.tgt": "le chat\n",
}


def lengths_of(iterators):
    """A dataset-level preprocessor: called with {series: () -> iterator} of the series read so far."""
    return (len(s) + len(t) for s, t in zip(iterators["source"](), iterators["target"]()))


def run_dataset_loading(case):
    """``dataset.load`` (dataset.py:207-333): series read from files (a glob over two files, a (files, reader) pair),
    a series-level preprocessor, a dataset-level preprocessor -- and the errors of bad specifications."""
    fresh_graph()                 # (no variables of an earlier case in this fixture)
    import tempfile
    from neuralmonkey.dataset import BatchingScheme, load
    from neuralmonkey.processors.helpers import preprocess_char_based
    from neuralmonkey.readers.plain_text_reader import tokenized_text_reader
    out = {}
    scheme = BatchingScheme(batch_size=2)
    join = lambda rows: np.asarray(["\x1f".join(str(t) for t in r) if isinstance(r, (list, tuple)) else str(r)
                                    for r in rows])
    with tempfile.TemporaryDirectory() as tmp:
        for name, text in LOAD_FILES.items():
            with open(os.path.join(tmp, name), "w", encoding="utf-8") as handle:
                handle.write(text)
        at = lambda name: os.path.join(tmp, name)
        ds = load("data", ["source", "target", "chars", "lens"],
                  [at("train.*.src"), (at("train.tgt"), tokenized_text_reader()),
                   (preprocess_char_based, "source"), lengths_of], scheme)
        for sid in ("source", "target", "chars", "lens"):
            out["out/" + sid] = join(list(ds.get_series(sid)))
        out["out/length"] = np.asarray(len(ds))
        out["out/batches"] = np.asarray([len(list(b.get_series("source"))) for b in ds.batches()])
        probes = {
            "count_mismatch": lambda: load("d", ["source", "target"], [at("train.tgt")], scheme),
            "duplicates": lambda: load("d", ["source", "source"], [at("train.tgt"), at("train.tgt")], scheme),
            "missing_file": lambda: load("d", ["source"], [at("nowhere.txt")], scheme),
            "no_file_series": lambda: load("d", ["chars"], [(preprocess_char_based, "source")], scheme),
            "no_series": lambda: load("d", [], [], scheme),
            "unknown_source": lambda: load("d", ["source", "chars"], [at("train.tgt"), (preprocess_char_based, "nope")],
                                           scheme),
            "unequal_lengths": lambda: load("d", ["source", "target"], [at("train.tgt"), at("short.tgt")], scheme),
            "multiple_outputs": lambda: load("d", ["source"], [at("train.tgt")], scheme,
                                             outputs=[("source", "a.txt"), ("source", "b.txt")]),
        }
        errors = {}
        for tag, probe in probes.items():
            try:
                probe()
                errors[tag] = ""
            except Exception as exc:        # noqa: BLE001
                errors[tag] = "{}: {}".format(type(exc).__name__, str(exc).replace(tmp, "<dir>"))
    out["out/errors"] = np.asarray(json.dumps(errors, sort_keys=True))
    save(case, {"kind": "dataset_loading", "files": LOAD_FILES}, out)


def experiment_stand_in(batch_size):
    """What ``Experiment.get_current()`` answers while a configuration is being built: the reference's own
    ``_DummyExperiment`` (the initializer registry of model parts created outside a run, experiment.py:494-523) plus
    the one number of [main] that a dataset section without ``batching`` reads (dataset.py:237-246).  The Experiment
    proper -- output directories, logging, the training loop -- is host control plane."""
    from argparse import Namespace
    from neuralmonkey.experiment import _DummyExperiment
    stand_in = _DummyExperiment()
    stand_in.config = Namespace(args=Namespace(batch_size=batch_size))
    return stand_in


def run_ini(case, ini_name, wanted, decoder_key="decoder", encoder_key="encoder", attention_key="attention",
            runner_key="runner", dataset_key="train_data"):
    """One of the reference's own acceptance configurations (tests/<ini_name>.ini), built by the reference's parser
    and builder from the file as it is: vocabularies and the training data from tests/data, the model parts and the
    runner from their sections (the trainers, the TensorFlow manager and the evaluators are TensorFlow's / the host
    control plane and stay unbuilt).  The first batch the file's batching scheme yields is fed with train_mode False
    (the file's dropout is then the identity) and the model's numbers are recorded under the variables' own names."""
    import collections.abc
    collections.Iterable = collections.abc.Iterable
    from neuralmonkey.config import parsing
    from neuralmonkey.config.builder import ObjectRef, build_config
    cwd = os.getcwd()
    os.chdir(REFERENCE)                          # the file's data paths are relative to the repository root
    os.environ.setdefault("NM_EXPERIMENT_NAME", "small")       # tests/tests_run.sh:33 (tests/small.ini reads it)
    try:
        fresh_graph()
        with open(os.path.join("tests", ini_name + ".ini"), encoding="utf-8") as handle:
            _, parsed = parsing.parse_file(handle.read().splitlines(True))
        # a dataset section without ``batching`` takes main.batch_size from the experiment being built
        # (dataset.py:237-246); the Experiment itself is host control plane, so only that one number stands in
        from neuralmonkey.experiment import Experiment
        Experiment._current_experiment = experiment_stand_in(parsed["main"].get("batch_size"))
        parsed["main"] = collections.OrderedDict((key, ObjectRef(section)) for key, section in wanted.items())
        try:
            built, _ = build_config(parsed, ignore_names=set())
        finally:
            Experiment._current_experiment = None
        enc, att, dec = built[encoder_key], built[attention_key], built[decoder_key]
        runner, data = built[runner_key], built[dataset_key]
        batch = next(iter(data.batches()))
        parts = [enc, enc.input_sequence, att, dec]
        inputs = string_inputs("source", "target")
        out = {}
        with tf_eager.feeding(feed(parts, batch, False, inputs)):
            out["in/src_tokens"] = enc.input_sequence.input_factors[0].numpy()
            out["in/src_ids"] = enc.input_sequence.inputs.numpy()
            out["in/tgt_tokens"] = dec.train_tokens.numpy()
            out["in/tgt_ids"] = dec.train_inputs.numpy()
            out["out/enc_states"] = enc.temporal_states.numpy()
            out["out/enc_mask"] = enc.temporal_mask.numpy()
            out["out/enc_output"] = enc.output.numpy()
            out["out/train_logits"] = dec.train_logits.numpy()
            out["out/train_xents"] = dec.train_xents.numpy()
            out["out/train_loss"] = dec.train_loss.numpy()
            out["out/runtime_logits"] = dec.runtime_logits.numpy()
            out["out/runtime_symbols"] = dec.runtime_loop_result.histories.output_symbols.numpy()
            out["out/runtime_mask"] = dec.runtime_mask.numpy()
            out["out/runtime_loss"] = dec.runtime_loss.numpy()
            ex = runner.get_executable(compute_losses=True, summaries=False, num_sessions=1)
            fetches, _ = ex.next_to_execute()
            ex.collect_results([to_numpy(fetches)])
            out["out/runner_sentences"] = np.asarray([joined(sent) for sent in ex.result.outputs[runner.output_series]])
            out["out/runner_losses"] = np.asarray([ex.result.losses["{}/{}".format(runner.output_series, name)]
                                                   for name in runner.loss_names], np.float32)
        out["in/src_vocabulary"] = np.asarray(list(enc.input_sequence.vocabularies[0].index_to_word))
        out["in/tgt_vocabulary"] = np.asarray(list(dec.vocabulary.index_to_word))
    finally:
        os.chdir(cwd)
    save(case, {"kind": "ini", "ini": ini_name, "batch": int(out["in/src_ids"].shape[0])}, out)


INI_VARIABLE_FILES = ["small", "bahdanau", "factored", "post-edit", "beamsearch", "transformer", "flat-multiattention",
                      "nematus"]


def run_ini_variables(case):
    """The variable-name contract of a checkpoint (parameterized.py:68-125, tf_manager.py:274-277): for
    each of the reference's acceptance configurations, the runners of [main] and the training data are built by the
    reference's parser and builder from the file as it is, every runner's fetches and every decoder's training loss
    are evaluated once on the first batch -- which is when the reference's lazily built model creates its variables
    -- and the names and shapes of all variables are recorded (the optimizer's slots belong to TensorFlow's trainers
    and are not built).  A file the reference itself refuses to build is recorded with its error."""
    import collections.abc
    collections.Iterable = collections.abc.Iterable
    from neuralmonkey.config import parsing
    from neuralmonkey.config.builder import build_config
    from neuralmonkey.experiment import Experiment
    cwd = os.getcwd()
    os.chdir(REFERENCE)
    os.environ.setdefault("NM_EXPERIMENT_NAME", "small")
    out = {}
    try:
        for ini in INI_VARIABLE_FILES:
            fresh_graph()
            with open(os.path.join("tests", ini + ".ini"), encoding="utf-8") as handle:
                _, parsed = parsing.parse_file(handle.read().splitlines(True))
            main = parsed["main"]
            Experiment._current_experiment = experiment_stand_in(main.get("batch_size"))
            parsed["main"] = collections.OrderedDict([("runners", main["runners"]),
                                                      ("train_dataset", main["train_dataset"])])
            try:
                built, _ = build_config(parsed, ignore_names=set())
            except Exception as exc:        # noqa: BLE001
                inner = getattr(exc, "original_exception", exc)
                out["out/" + ini] = np.asarray(json.dumps({"error": {
                    "type": type(inner).__name__, "text": str(inner), "section": str(getattr(exc, "object_name", ""))}}))
                continue
            finally:
                Experiment._current_experiment = None
            runners = built["runners"]
            batch = next(iter(built["train_dataset"].batches()))
            feedables = sorted(set.union(*[set(r.feedables) for r in runners]), key=lambda f: str(getattr(f, "name", "")))
            inputs = {}
            for part in feedables:
                for series, dtype in part.input_types.items():
                    if series not in inputs:
                        inputs[series] = tf.placeholder(dtype, part.input_shapes[series], series)
            with tf_eager.feeding(feed(feedables, batch, False, inputs)):
                for runner in runners:
                    ex = runner.get_executable(compute_losses=True, summaries=False, num_sessions=1)
                    ex.next_to_execute()
                    decoder = getattr(runner, "decoder", None)
                    for part in (decoder, getattr(decoder, "parent_decoder", None)):
                        if part is not None and hasattr(type(part), "train_loss"):
                            part.train_loss      # noqa: B018  (the trainer's fetch: creates what only training reads)
            order, params = variables()
            out["out/" + ini] = np.asarray(json.dumps({"variables": [[n, list(params[n].shape)] for n in order]}))
    finally:
        os.chdir(cwd)
    fresh_graph()
    save(case, {"kind": "ini_variables", "files": INI_VARIABLE_FILES}, out)


def run_ini_beamsearch(case, rows=6):
    """tests/beamsearch.ini (TransformerEncoder / TransformerDecoder of dimension 6 + BeamSearchDecoder, beam 3,
    length normalisation 0.6, 10 steps) built by the reference's parser and builder from the file as it is.  Its
    datasets rely on ``main.batch_size`` through the running Experiment (host control plane), so the batch here is
    the first ``rows`` sentence pairs of the file's training data, read by the reference's reader."""
    import collections.abc
    collections.Iterable = collections.abc.Iterable
    from neuralmonkey.config import parsing
    from neuralmonkey.config.builder import ObjectRef, build_config
    from neuralmonkey.readers.plain_text_reader import UtfPlainTextReader
    from neuralmonkey.runners.beamsearch_runner import BeamSearchRunner
    cwd = os.getcwd()
    os.chdir(REFERENCE)
    try:
        fresh_graph()
        with open(os.path.join("tests", "beamsearch.ini"), encoding="utf-8") as handle:
            _, parsed = parsing.parse_file(handle.read().splitlines(True))
        files = parsed["train_data"]["data"][:2]
        parsed["main"] = collections.OrderedDict((key, ObjectRef(key)) for key in ("inpseq", "encoder", "decoder",
                                                                                   "bs_decoder"))
        built, _ = build_config(parsed, ignore_names=set())
        seq, enc, dec, bs = built["inpseq"], built["encoder"], built["decoder"], built["bs_decoder"]
        series = {"source": list(UtfPlainTextReader([files[0]]))[:rows], "target": list(UtfPlainTextReader([files[1]]))[:rows]}
        ds = dataset(series)
        inputs = string_inputs("source", "target")
        out = {}
        with tf_eager.feeding(feed([seq, enc, dec], ds, False, inputs)):
            out["in/src_tokens"] = seq.input_factors[0].numpy()
            out["in/src_ids"] = seq.inputs.numpy()
            out["in/tgt_tokens"] = dec.train_tokens.numpy()
            out["in/tgt_ids"] = dec.train_inputs.numpy()
            out["out/enc_states"] = enc.temporal_states.numpy()
            out["out/enc_mask"] = enc.temporal_mask.numpy()
            out["out/train_logits"] = dec.train_logits.numpy()
            out["out/train_loss"] = dec.train_loss.numpy()
            out["out/runtime_logits"] = dec.runtime_logits.numpy()
            out["out/runtime_symbols"] = dec.runtime_loop_result.histories.output_symbols.numpy()
            out["out/runtime_mask"] = dec.runtime_mask.numpy()
        order_full, _ = variables()
        # the beam search of the file's [bs_decoder], through a rank-1 and a rank-2 runner as its [bs_runners] makes
        for rank in (1, 2):
            fresh_graph()
            with open(os.path.join("tests", "beamsearch.ini"), encoding="utf-8") as handle:
                _, parsed = parsing.parse_file(handle.read().splitlines(True))
            parsed["main"] = collections.OrderedDict((key, ObjectRef(key)) for key in ("inpseq", "encoder", "decoder",
                                                                                       "bs_decoder"))
            built, _ = build_config(parsed, ignore_names=set())
            seq, enc, dec, bs = built["inpseq"], built["encoder"], built["decoder"], built["bs_decoder"]
            runner = BeamSearchRunner(output_series="target_beam", decoder=bs, rank=rank)
            with tf_eager.feeding(feed([seq, enc, dec, bs], ds, False, inputs)):
                ex = runner.get_executable(compute_losses=False, summaries=False, num_sessions=1)
                fetches, _ = ex.next_to_execute()
                res = to_numpy(fetches)
                ex.collect_results([res])
                bo = res["bs_outputs"]
                if rank == 1:
                    out["out/beam_scores"] = bo.last_search_step_output.scores
                    out["out/beam_token_ids"] = bo.last_search_step_output.token_ids
                out["out/rank{}_sentences".format(rank)] = np.asarray([joined(t) for t in ex.result.outputs["target_beam"]])
                out["out/rank{}_loss".format(rank)] = np.asarray(ex.result.losses["target_beam/beam_search_score"])
        out["in/src_vocabulary"] = np.asarray(list(seq.vocabularies[0].index_to_word))
        out["in/tgt_vocabulary"] = np.asarray(list(dec.vocabulary.index_to_word))
        out["cfg/beam"] = np.asarray([bs.beam_size, bs.max_steps_int, 0], np.int64)
    finally:
        os.chdir(cwd)
    # leave the store holding the full model's variables for save()
    os.chdir(REFERENCE)
    try:
        fresh_graph()
        with open(os.path.join("tests", "beamsearch.ini"), encoding="utf-8") as handle:
            _, parsed = parsing.parse_file(handle.read().splitlines(True))
        parsed["main"] = collections.OrderedDict((key, ObjectRef(key)) for key in ("inpseq", "encoder", "decoder"))
        built, _ = build_config(parsed, ignore_names=set())
        with tf_eager.feeding(feed([built["inpseq"], built["encoder"], built["decoder"]], ds, False, inputs)):
            built["decoder"].train_loss.numpy()
            built["decoder"].runtime_logits.numpy()
    finally:
        os.chdir(cwd)
    save(case, {"kind": "ini", "ini": "beamsearch", "batch": rows}, out)


def run_ini_factored(case, rows=6):
    """tests/factored.ini (FactoredEncoder over word forms and tags, ScaledDotProdAttention with one head, GRU decoder
    of size 32) built by the reference's parser and builder from the file as it is; like tests/beamsearch.ini its
    datasets lean on ``main.batch_size``, so the batch is the first ``rows`` lines of the file's training data."""
    import collections.abc
    collections.Iterable = collections.abc.Iterable
    from neuralmonkey.config import parsing
    from neuralmonkey.config.builder import ObjectRef, build_config
    from neuralmonkey.readers.plain_text_reader import UtfPlainTextReader
    cwd = os.getcwd()
    os.chdir(REFERENCE)
    try:
        fresh_graph()
        with open(os.path.join("tests", "factored.ini"), encoding="utf-8") as handle:
            _, parsed = parsing.parse_file(handle.read().splitlines(True))
        names, files = parsed["train_data"]["series"], parsed["train_data"]["data"]
        parsed["main"] = collections.OrderedDict((key, ObjectRef(key)) for key in ("encoder", "attention", "decoder",
                                                                                   "runner"))
        built, _ = build_config(parsed, ignore_names=set())
        enc, att, dec, runner = built["encoder"], built["attention"], built["decoder"], built["runner"]
        series = {name: list(UtfPlainTextReader([path]))[:rows] for name, path in zip(names, files)}
        ds = dataset(series)
        inputs = string_inputs("source", "tags", "target")
        out = {}
        with tf_eager.feeding(feed([enc, enc.input_sequence, att, dec], ds, False, inputs)):
            out["in/src_tokens"] = enc.input_sequence.input_factors[0].numpy()
            out["in/tag_tokens"] = enc.input_sequence.input_factors[1].numpy()
            out["in/src_ids"] = enc.input_sequence.input_factor_indices[0].numpy()
            out["in/tag_ids"] = enc.input_sequence.input_factor_indices[1].numpy()
            out["in/tgt_tokens"] = dec.train_tokens.numpy()
            out["in/tgt_ids"] = dec.train_inputs.numpy()
            out["out/enc_states"] = enc.temporal_states.numpy()
            out["out/enc_output"] = enc.output.numpy()
            out["out/train_logits"] = dec.train_logits.numpy()
            out["out/train_loss"] = dec.train_loss.numpy()
            out["out/runtime_logits"] = dec.runtime_logits.numpy()
            out["out/runtime_symbols"] = dec.runtime_loop_result.histories.output_symbols.numpy()
            out["out/runtime_mask"] = dec.runtime_mask.numpy()
            ex = runner.get_executable(compute_losses=True, summaries=False, num_sessions=1)
            fetches, _ = ex.next_to_execute()
            ex.collect_results([to_numpy(fetches)])
            out["out/runner_sentences"] = np.asarray([joined(sent) for sent in ex.result.outputs["target"]])
            out["out/runner_losses"] = np.asarray([ex.result.losses["target/train_xent"],
                                                   ex.result.losses["target/runtime_xent"]], np.float32)
        out["in/tgt_vocabulary"] = np.asarray(list(dec.vocabulary.index_to_word))
    finally:
        os.chdir(cwd)
    save(case, {"kind": "ini", "ini": "factored", "batch": rows}, out)


def run_ini_postedit(case):
    """tests/post-edit.ini (tests/tests_run.sh:12) built by the reference's parser and builder from the file as it is:
    a GRU ``SentenceEncoder`` over the source, an LSTM ``RecurrentEncoder`` over the machine translation, an RNN
    decoder of the edit scripts (``processors.editops.Preprocess`` makes that series while the data are loaded) that
    borrows the translation's embeddings and attends with a three-head ``MultiHeadAttention`` (keys: source encoder,
    values: translation encoder) and a ``ScaledDotProdAttention`` over the source.  First batch of the training data
    (two sentences, [main] batch_size), train_mode False."""
    import collections.abc
    collections.Iterable = collections.abc.Iterable
    from neuralmonkey.config import parsing
    from neuralmonkey.config.builder import ObjectRef, build_config
    from neuralmonkey.experiment import Experiment
    cwd = os.getcwd()
    os.chdir(REFERENCE)
    try:
        fresh_graph()
        with open(os.path.join("tests", "post-edit.ini"), encoding="utf-8") as handle:
            _, parsed = parsing.parse_file(handle.read().splitlines(True))
        Experiment._current_experiment = experiment_stand_in(parsed["main"].get("batch_size"))
        parsed["main"] = collections.OrderedDict((key, ObjectRef(key)) for key in (
            "src_encoder", "trans_encoder", "trans_embedded_input", "src_attention", "trans_attention", "decoder",
            "runner", "train_dataset", "postprocess"))
        try:
            built, _ = build_config(parsed, ignore_names=set())
        finally:
            Experiment._current_experiment = None
        src, trans, seq = built["src_encoder"], built["trans_encoder"], built["trans_embedded_input"]
        dec, runner = built["decoder"], built["runner"]
        batch = next(iter(built["train_dataset"].batches()))
        parts = [src, src.input_sequence, trans, seq, built["src_attention"], built["trans_attention"], dec]
        inputs = string_inputs("source", "translated", "edits")
        out = {}
        with tf_eager.feeding(feed(parts, batch, False, inputs)):
            out["in/src_tokens"] = src.input_sequence.input_factors[0].numpy()
            out["in/src_ids"] = src.input_sequence.inputs.numpy()
            out["in/mt_tokens"] = seq.input_factors[0].numpy()
            out["in/mt_ids"] = seq.inputs.numpy()
            out["in/tgt_tokens"] = dec.train_tokens.numpy()
            out["in/tgt_ids"] = dec.train_inputs.numpy()
            out["out/src_states"] = src.temporal_states.numpy()
            out["out/src_mask"] = src.temporal_mask.numpy()
            out["out/src_output"] = src.output.numpy()
            out["out/mt_states"] = trans.temporal_states.numpy()
            out["out/mt_mask"] = trans.temporal_mask.numpy()
            out["out/mt_output"] = trans.output.numpy()
            out["out/train_logits"] = dec.train_logits.numpy()
            out["out/train_xents"] = dec.train_xents.numpy()
            out["out/train_loss"] = dec.train_loss.numpy()
            out["out/runtime_logits"] = dec.runtime_logits.numpy()
            out["out/runtime_symbols"] = dec.runtime_loop_result.histories.output_symbols.numpy()
            out["out/runtime_mask"] = dec.runtime_mask.numpy()
            ex = runner.get_executable(compute_losses=True, summaries=False, num_sessions=1)
            fetches, _ = ex.next_to_execute()
            ex.collect_results([to_numpy(fetches)])
            scripts = ex.result.outputs[runner.output_series]
            out["out/runner_sentences"] = np.asarray([joined(sent) for sent in scripts])
            out["out/runner_losses"] = np.asarray([ex.result.losses["{}/{}".format(runner.output_series, name)]
                                                   for name in runner.loss_names], np.float32)
        # [main] postprocess=[("target", <postprocess>)]: the generated scripts applied to the batch's translations
        translated = list(batch.get_series("translated"))
        rebuilt = built["postprocess"]({"translated": translated}, {"edits": scripts})
        out["out/postprocessed"] = np.asarray([joined(sent) for sent in rebuilt])
        out["in/translated"] = np.asarray([joined(sent) for sent in translated])
        out["in/src_vocabulary"] = np.asarray(list(src.input_sequence.vocabularies[0].index_to_word))
        out["in/tgt_vocabulary"] = np.asarray(list(dec.vocabulary.index_to_word))
    finally:
        os.chdir(cwd)
    save(case, {"kind": "ini", "ini": "post-edit", "batch": int(out["in/src_ids"].shape[0])}, out)


FLAT_DECODERS = [("flat_noshare_nosentinel", "wrapper_fnn"), ("flat_share_nosentinel", "wrapper_fsn"),
                 ("flat_share_sentinel", "wrapper_fss"), ("flat_noshare_sentinel", "wrapper_fns")]


def run_ini_flat(case):
    """tests/flat-multiattention.ini (tests/tests_run.sh:28) built by the reference's parser and builder from the file
    as it is: a ``SpatialFiller`` over pre-extracted 8x8x2048 feature maps (read through ``numpy_reader.from_file_list``)
    and a ``SentenceEncoder``, four RNN decoders over both, each with its own ``FlatMultiAttention`` (projections
    shared or not, sentinel or not), a ``GreedyRunner`` per decoder, and the RNN beam search (beam 2, length
    normalisation 1.0, 3 steps, ``BeamSearchRunner(rank=2)``) over the shared-projections-with-sentinel decoder.  First
    batch of the training data ([main] batch_size 1: the reference's beam search over an RNN decoder is confined to
    it), train_mode False."""
    import collections.abc
    collections.Iterable = collections.abc.Iterable
    from neuralmonkey.config import parsing
    from neuralmonkey.config.builder import build_config
    from neuralmonkey.experiment import Experiment
    cwd = os.getcwd()
    os.chdir(REFERENCE)
    try:
        fresh_graph()
        with open(os.path.join("tests", "flat-multiattention.ini"), encoding="utf-8") as handle:
            _, parsed = parsing.parse_file(handle.read().splitlines(True))
        main = parsed["main"]
        Experiment._current_experiment = experiment_stand_in(main.get("batch_size"))
        parsed["main"] = collections.OrderedDict([("runners", main["runners"]), ("train_dataset", main["train_dataset"])])
        try:
            built, _ = build_config(parsed, ignore_names=set())
        finally:
            Experiment._current_experiment = None
        runners = built["runners"]
        batch = next(iter(built["train_dataset"].batches()))
        feedables = sorted(set.union(*[set(r.feedables) for r in runners]), key=lambda f: str(getattr(f, "name", "")))
        inputs = {}
        for part in feedables:
            for series, dtype in part.input_types.items():
                if series not in inputs:
                    inputs[series] = tf.placeholder(dtype, part.input_shapes[series], series)
        out = {}
        with tf_eager.feeding(feed(feedables, batch, False, inputs)):
            first = runners[0].decoder
            enc, img = first.encoders
            out["in/src_tokens"] = enc.input_sequence.input_factors[0].numpy()
            out["in/src_ids"] = enc.input_sequence.inputs.numpy()
            out["in/maps"] = img.spatial_states.numpy()
            out["in/tgt_tokens"] = first.train_tokens.numpy()
            out["in/tgt_ids"] = first.train_inputs.numpy()
            out["out/enc_states"] = enc.temporal_states.numpy()
            out["out/enc_output"] = enc.output.numpy()
            for runner, (tag, _) in zip(runners[:4], FLAT_DECODERS):
                dec = runner.decoder
                assert dec.name == "decoder_" + tag, dec.name
                out["out/{}/train_logits".format(tag)] = dec.train_logits.numpy()
                out["out/{}/train_loss".format(tag)] = dec.train_loss.numpy()
                out["out/{}/runtime_logits".format(tag)] = dec.runtime_logits.numpy()
                out["out/{}/runtime_symbols".format(tag)] = dec.runtime_loop_result.histories.output_symbols.numpy()
                out["out/{}/runtime_mask".format(tag)] = dec.runtime_mask.numpy()
                ex = runner.get_executable(compute_losses=True, summaries=False, num_sessions=1)
                fetches, _ = ex.next_to_execute()
                ex.collect_results([to_numpy(fetches)])
                out["out/{}/runner_sentences".format(tag)] = np.asarray(
                    [joined(sent) for sent in ex.result.outputs[runner.output_series]])
                out["out/{}/runner_losses".format(tag)] = np.asarray(
                    [ex.result.losses["{}/{}".format(runner.output_series, name)] for name in runner.loss_names],
                    np.float32)
            beam_runner = runners[4]
            bs = beam_runner.decoder
            ex = beam_runner.get_executable(compute_losses=True, summaries=False, num_sessions=1)
            fetches, _ = ex.next_to_execute()
            res = to_numpy(fetches)
            ex.collect_results([res])
            bo = res["bs_outputs"]
            out["out/beam_scores"] = bo.last_search_step_output.scores
            out["out/beam_token_ids"] = bo.last_search_step_output.token_ids
            out["out/beam_runner_sentences"] = np.asarray([joined(sent) for sent in
                                                           ex.result.outputs[beam_runner.output_series]])
            out["out/beam_runner_loss"] = np.asarray(
                ex.result.losses["{}/beam_search_score".format(beam_runner.output_series)], np.float32)
        out["in/tgt_vocabulary"] = np.asarray(list(first.vocabulary.index_to_word))
        out["cfg/beam"] = np.asarray([bs.beam_size, bs.max_steps_int, beam_runner.rank], np.int64)
    finally:
        os.chdir(cwd)
    save(case, {"kind": "ini", "ini": "flat-multiattention", "batch": int(out["in/src_ids"].shape[0])}, out)


TRAINER_INIS = collections.OrderedDict([("small", ["trainer"]), ("bahdanau", ["trainer1", "trainer2", "greedy_trainer"]),
                                        ("post-edit", ["trainer"])])


def run_ini_trainer_objectives(case):
    """trainers/generic_trainer.py:84-134 (what a training step minimises) for the trainers of three acceptance
    configurations, built by the reference's parser and builder from the files as they are: which variables the
    regulariser covers (trainable, no ``[Bb]ias`` in the name, :87-91), the L1 and L2 sums over them,
    ``objective_values`` (losses, L1, L2) and ``differentiable_loss_sum`` (objective weights, ``l1_weight``,
    ``l2_weight``) on the first training batch with train_mode False, and the trainer's ``var_list``.  The optimizer
    and ``tf.gradients`` are TensorFlow's and are not run."""
    import collections.abc
    collections.Iterable = collections.abc.Iterable
    import re
    from neuralmonkey.config import parsing
    from neuralmonkey.config.builder import ObjectRef, build_config
    from neuralmonkey.experiment import Experiment
    bias = re.compile(r"[Bb]ias")
    cwd = os.getcwd()
    os.chdir(REFERENCE)
    os.environ.setdefault("NM_EXPERIMENT_NAME", "small")
    out = {}
    try:
        for ini, sections in TRAINER_INIS.items():
            fresh_graph()
            with open(os.path.join("tests", ini + ".ini"), encoding="utf-8") as handle:
                _, parsed = parsing.parse_file(handle.read().splitlines(True))
            main = parsed["main"]
            Experiment._current_experiment = experiment_stand_in(main.get("batch_size"))
            parsed["main"] = collections.OrderedDict([(sec, ObjectRef(sec)) for sec in sections]
                                                     + [("train_dataset", main["train_dataset"])])
            try:
                built, _ = build_config(parsed, ignore_names=set())
            finally:
                Experiment._current_experiment = None
            trainers = [built[sec] for sec in sections]
            batch = next(iter(built["train_dataset"].batches()))
            feedables = sorted(set.union(*[set(t.feedables) | {t} for t in trainers]),
                               key=lambda f: str(getattr(f, "name", type(f).__name__)))
            inputs = {}
            for part in feedables:
                for series, dtype in part.input_types.items():
                    if series not in inputs:
                        inputs[series] = tf.placeholder(dtype, part.input_shapes[series], series)
            record = collections.OrderedDict()
            with tf_eager.feeding(feed(feedables, batch, False, inputs)):
                for sec, trainer in zip(sections, trainers):
                    values = [float(v.numpy()) for v in trainer.objective_values]
                    record[sec] = {"objective_values": values,
                                   "differentiable_loss_sum": float(trainer.differentiable_loss_sum.numpy()),
                                   "l1_weight": trainer.l1_weight, "l2_weight": trainer.l2_weight,
                                   "clip_norm": trainer.clip_norm,
                                   "objective_names": [obj.name for obj in trainer.objectives],
                                   "objective_weights": [obj.weight for obj in trainer.objectives],
                                   "var_list": [v.name for v in trainer.var_list]}
            order, params = variables()
            trainable = [v.name for v in tf.trainable_variables()]
            record["_regularizable"] = [n for n in trainable if not bias.findall(n)]
            record["_trainable"] = trainable
            out["out/" + ini] = np.asarray(json.dumps(record))
            for name in order:
                out["vars/{}/{}".format(ini, name)] = params[name]
    finally:
        os.chdir(cwd)
    fresh_graph()
    save(case, {"kind": "ini_trainer_objectives", "files": list(TRAINER_INIS)}, out)


def run_defects(case):
    """Configurations the reference cannot execute at this commit: the exception IS the reference behaviour."""
    import traceback
    found = {}
    # Decoder(attention_on_input=True): decoder.py:270-273 reads ``feedables.prev_contexts`` where the field lives in
    # ``feedables.other`` (RNNFeedables) -- AttributeError as soon as the loop body is traced
    cfg = dict(RNN_DEFAULT, attention_on_input=True)
    fresh_graph()
    enc, att, dec, parts = build_rnn(cfg)
    inputs = string_inputs("source", "target")
    try:
        with tf_eager.feeding(feed(parts, dataset(rnn_series(cfg)), False, inputs)):
            dec.train_logits.numpy()
        found["attention_on_input"] = ""
    except AttributeError as exc:
        tb = traceback.extract_tb(exc.__traceback__)[-1]
        found["attention_on_input"] = "{}:{} {}".format(os.path.relpath(tb.filename, REFERENCE), tb.lineno, exc)
    # CoverageAttention: coverage.py:52 calls ``.size()`` on a tf.Tensor
    from neuralmonkey.attention.coverage import CoverageAttention
    fresh_graph()
    cfg = dict(RNN_DEFAULT)
    enc, att, dec, parts = build_rnn(cfg)
    cov = CoverageAttention(name="coverage", encoder=enc)
    try:
        with tf_eager.feeding(feed(parts + [cov], dataset(rnn_series(cfg)), False, inputs)):
            cov.attention(tf.zeros([cfg["batch"], cfg["rnn_size"]]), None, None, cov.initial_loop_state())
        found["coverage"] = ""
    except Exception as exc:        # noqa: BLE001 -- whatever it raises is what we record
        tb = traceback.extract_tb(exc.__traceback__)
        here = [t for t in tb if t.filename.startswith(REFERENCE)][-1]
        found["coverage"] = "{}:{} {}: {}".format(os.path.relpath(here.filename, REFERENCE), here.lineno,
                                                  type(exc).__name__, exc)
    # BeamSearchRunner over several sessions of an RNN decoder: an unhashable feed key (see run_ensemble)
    try:
        run_ensemble("_rnn_ensemble_probe", family="rnn")
        found["rnn_ensemble"] = ""
    except TypeError as exc:
        tb = [t for t in traceback.extract_tb(exc.__traceback__) if t.filename.startswith(REFERENCE)][-1]
        found["rnn_ensemble"] = "{}:{} {}: {}".format(os.path.relpath(tb.filename, REFERENCE), tb.lineno,
                                                      type(exc).__name__, exc)
    finally:
        tf_eager.VARIABLE_FACTORY = variable_factory
    fresh_graph()
    print(json.dumps(found, indent=1))
    save(case, found, {})


CASES = collections.OrderedDict([
    ("functions", lambda: run_functions("functions")),
    ("beam_body", lambda: run_beam_body("beam_body")),
    # no length normalisation (alpha 0: scores are the plain log-probability sums) with a beam as wide as the live
    # vocabulary allows; alpha 1 with a longer search
    ("beam_body_k5_alpha0", lambda: run_beam_body("beam_body_k5_alpha0", vsz=9, k=5, max_steps=6, alpha=0.0, seed=6,
                                                  rank=3)),
    ("beam_body_k4_alpha1", lambda: run_beam_body("beam_body_k4_alpha1", vsz=8, k=4, max_steps=8, alpha=1.0, seed=7,
                                                  rank=1)),
    ("rnn_gru", lambda: run_rnn("rnn_gru")),
    ("rnn_gru_supress_unk", lambda: run_rnn("rnn_gru_supress_unk", supress_unk=True, seed=2, max_input_len=5,
                                            att_state=7)),
    ("rnn_nematus_cgru", lambda: run_rnn(
        "rnn_nematus_cgru", enc_layers=[[5, "bidirectional", "NematusGRU"]], dec_cell="NematusGRU",
        conditional_gru=True, output_projection=["nematus", "tanh"], encoder_projection="nematus",
        enc_keep=0.5, att_keep=0.5, dec_keep=0.5, seed=4)),
    ("defects", lambda: run_defects("defects")),
    ("rnn_lstm", lambda: run_rnn("rnn_lstm", enc_layers=[[5, "bidirectional", "LSTM"]], dec_cell="LSTM", seed=6,
                                 output_projection=["maxout"])),
    ("rnn_stacked", lambda: run_rnn(
        "rnn_stacked", sentence_encoder=False, enc_layers=[[3, "bidirectional", "GRU"], [6, "forward", "GRU"],
                                                           [6, "backward", "LSTM"]],
        add_layer_norm=True, add_residual=True, output_projection=["mlp", [7, 6], "relu"],
        encoder_projection="concat", seed=7)),
    ("rnn_tied", lambda: run_rnn("rnn_tied", tie_embeddings=True, output_projection=["default"],
                                 encoder_projection="empty", seed=8)),
    ("captioning", lambda: run_rnn("captioning", spatial=[3, 3, 10, None, None], seed=9)),
    ("captioning_projected", lambda: run_rnn("captioning_projected", spatial=[2, 3, 10, 9, 8], seed=10)),
    ("ms_flat", lambda: run_multisource("ms_flat")),
    ("ms_flat_share_sentinel", lambda: run_multisource("ms_flat_share_sentinel", share=True, sentinel=True, seed=22,
                                                       dec_cell="NematusGRU", conditional_gru=True)),
    ("ms_flat_projected_sentinel", lambda: run_multisource("ms_flat_projected_sentinel", sentinel=True, seed=23,
                                                           image=[2, 3, 7, None, 8])),
    ("ms_hier", lambda: run_multisource("ms_hier", kind="hier", seed=24)),
    ("ms_hier_share_sentinel", lambda: run_multisource("ms_hier_share_sentinel", kind="hier", share=True,
                                                       sentinel=True, state_size=6, seed=25)),
    ("dotprod_heads2", lambda: run_multisource("dotprod_heads2", kind="dotprod", heads=2, enc_size=3, seed=26)),
    ("dotprod_heads1", lambda: run_multisource("dotprod_heads1", kind="dotprod", heads=1, enc_size=3, seed=27,
                                               dec_cell="LSTM")),
    ("factored_smoothing", lambda: run_multisource("factored_smoothing", kind="plain", factored=True,
                                                   label_smoothing=0.1, seed=28)),
    ("stateful_context", lambda: run_multisource("stateful_context", kind="stateful", seed=30,
                                                 conditional_gru=True)),
    ("fd_gradients_rnn_gru", lambda: run_fd_gradients("fd_gradients_rnn_gru", "rnn")),
    ("fd_gradients_rnn_nematus_lstm", lambda: run_fd_gradients(
        "fd_gradients_rnn_nematus_lstm", "rnn", enc_layers=[[5, "bidirectional", "NematusGRU"]], dec_cell="LSTM",
        output_projection=["nematus", "tanh"], encoder_projection="nematus", seed=16)),
    ("fd_gradients_transformer", lambda: run_fd_gradients("fd_gradients_transformer", "transformer")),
    ("fd_gradients_ms_hier", lambda: run_fd_gradients("fd_gradients_ms_hier", "ms", kind="hier", share=True,
                                                      sentinel=True, state_size=6, seed=36, per_variable=3)),
    ("fd_gradients_ms_flat", lambda: run_fd_gradients("fd_gradients_ms_flat", "ms", kind="flat", sentinel=True,
                                                      image=[2, 3, 7, None, 8], seed=37, per_variable=3)),
    ("fd_gradients_dotprod", lambda: run_fd_gradients("fd_gradients_dotprod", "ms", kind="dotprod", heads=2,
                                                      enc_size=3, seed=38, per_variable=3)),
    ("fd_gradients_captioning", lambda: run_fd_gradients("fd_gradients_captioning", "rnn", spatial=[2, 3, 10, 9, 8],
                                                         seed=17)),
    ("fd_gradients_transformer_ms_hier", lambda: run_fd_gradients(
        "fd_gradients_transformer_ms_hier", "transformer", second_encoder=True, strategy="hierarchical",
        heads_hier=4, seed=35, per_variable=2)),
    ("vocabulary_formats", lambda: run_vocabulary_formats("vocabulary_formats")),
    ("host_text_pipeline", lambda: run_host_text_pipeline("host_text_pipeline")),
    ("editops", lambda: run_editops("editops")),
    ("ini_bahdanau", lambda: run_ini("ini_bahdanau", "bahdanau", collections.OrderedDict(
        [("encoder", "encoder"), ("attention", "attention"), ("decoder", "decoder"), ("runner", "runner"),
         ("train_data", "train_data")]))),
    ("ini_beamsearch", lambda: run_ini_beamsearch("ini_beamsearch")),
    ("ini_factored", lambda: run_ini_factored("ini_factored")),
    ("ini_small", lambda: run_ini("ini_small", "small", collections.OrderedDict(
        [("encoder", "my_encoder"), ("attention", "my_attention"), ("decoder", "my_decoder"), ("runner", "runner"),
         ("data", "val_data")]), dataset_key="data")),
    ("ini_variables", lambda: run_ini_variables("ini_variables")),
    ("ini_postedit", lambda: run_ini_postedit("ini_postedit")),
    ("ini_flat", lambda: run_ini_flat("ini_flat")),
    ("ini_trainer_objectives", lambda: run_ini_trainer_objectives("ini_trainer_objectives")),
    ("schedules", lambda: run_schedules("schedules")),
    ("ini_grammar", lambda: run_ini_grammar("ini_grammar")),
    ("config_builder", lambda: run_config_builder("config_builder")),
    ("dataset_loading", lambda: run_dataset_loading("dataset_loading")),
    ("dataset_batching", lambda: run_dataset_batching("dataset_batching")),
    ("dataset_lazy_shuffle", lambda: run_dataset_lazy_shuffle("dataset_lazy_shuffle")),
    ("greedy_runner_ensemble", lambda: run_greedy_runner_ensemble("greedy_runner_ensemble")),
    ("tensor_runner", lambda: run_tensor_runner("tensor_runner")),
    ("ensemble", lambda: run_ensemble("ensemble")),
    ("transformer", lambda: run_transformer("transformer")),
    ("transformer_bias_untied", lambda: run_transformer(
        "transformer_bias_untied", tie_embeddings=False, use_att_transform_bias=True, heads=4, heads_self=1,
        heads_enc=2, depth=3, target_space_id=5, seed=12)),
    ("transformer_shared", lambda: run_transformer("transformer_shared", shared_embeddings=True,
                                                   scale_embeddings=True, seed=13)),
    ("transformer_ms_serial", lambda: run_transformer("transformer_ms_serial", second_encoder=True, seed=31)),
    ("transformer_ms_parallel", lambda: run_transformer("transformer_ms_parallel", second_encoder=True,
                                                        strategy="parallel", seed=32)),
    ("transformer_ms_flat", lambda: run_transformer("transformer_ms_flat", second_encoder=True, strategy="flat",
                                                    seed=33)),
    ("transformer_ms_hier", lambda: run_transformer("transformer_ms_hier", second_encoder=True,
                                                    strategy="hierarchical", heads_hier=4, seed=34)),
])


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for name in names:
        CASES[name]()
