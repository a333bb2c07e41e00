"""End-to-end parity of the plugin surface (SentenceEncoder -> Attention ->
Decoder -> GreedyRunner / BeamSearchRunner through TensorFlowManager.execute)
against the CPU oracle on the same weights and the same seeded batch."""
import numpy as np
import pytest
import torch

from oracle import nm_oracle as O

pytestmark = pytest.mark.gpu


def _setup(dev, vocab, emb, rnn, batch, slen, tlen, ragged, beam=3, max_steps=None, seed=7, **kw):
    from neuralmonkey_amd import synthetic
    std = kw.pop("std", 0.08)
    model = synthetic.build_translation_model(
        vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn, max_len=max(slen, tlen), beam_size=beam,
        max_steps=max_steps or tlen, with_trainer=False, device=str(dev), **kw)
    params = O.init_params(seed=seed, vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn, std=std)
    model.tf_manager.sessions[0].store.load_state_dict(params)
    ds = synthetic.synthetic_dataset(seed=seed + 1, batch=batch, src_len=slen, tgt_len=tlen, vocab=vocab,
                                     ragged=ragged)
    src = O.pad_ids([list(s) for s in ds.get_series("source")], max(slen, tlen))
    tgt = O.pad_ids([list(s) for s in ds.get_series("target")], max(slen, tlen), add_end_symbol=True)
    return model, params, ds, src, np.ascontiguousarray(tgt.T)


def _ids_to_words(vocab, sents):
    return [[vocab.index_to_word[i] for i in s] for s in sents]


@pytest.mark.parametrize("vocab,emb,rnn,batch,slen,tlen,ragged", [
    (64, 12, 12, 5, 7, 6, True),
    (500, 32, 32, 16, 20, 12, True),
    (2000, 64, 64, 32, 30, 25, False),
])
def test_encoder_and_greedy_match_oracle(dev, vocab, emb, rnn, batch, slen, tlen, ragged):
    model, params, ds, src, tgt = _setup(dev, vocab, emb, rnn, batch, slen, tlen, ragged)
    enc = O.sentence_encoder(params, src)
    spec = O.DecoderSpec(max_output_len=max(slen, tlen))
    ref = O.decoding_loop(params, spec, enc, None, False)
    ref_train = O.decoding_loop(params, spec, enc, tgt, True)

    sess = model.tf_manager.sessions[0]
    fd = {}
    for f in model.greedy_runner.feedables:
        fd.update(f.feed_dict(ds, train=False))
    got = sess.run({"states": model.encoder.temporal_states, "final": model.encoder.output,
                    "sym": model.decoder.decoded_symbols, "logits": model.decoder.runtime_logits,
                    "train_loss": model.decoder.train_loss, "runtime_loss": model.decoder.runtime_loss,
                    "w": model.attention.hidden_features}, fd)
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))
    assert rel(got["states"], enc.temporal_states) < 1e-4
    assert rel(got["final"], enc.output) < 1e-4
    assert np.array_equal(got["sym"], ref.symbols.astype(np.int32)), "greedy symbols differ"
    assert rel(got["logits"], ref.logits) < 1e-4
    assert abs(float(got["train_loss"]) - float(O.train_loss(ref_train, tgt))) < 1e-4 * abs(float(O.train_loss(ref_train, tgt)))
    assert abs(float(got["runtime_loss"]) - float(O.runtime_loss(ref, tgt))) < 1e-4 * abs(float(O.runtime_loss(ref, tgt)))

    res = model.tf_manager.execute(ds, model.greedy_runner.feedables, [model.greedy_runner])[0]
    want = _ids_to_words(model.tgt_vocab, O.greedy_tokens(ref))
    assert res.outputs["target"] == want
    assert set(res.losses) == {"target/train_xent", "target/runtime_xent"}
    assert res.size == batch


@pytest.mark.parametrize("vocab,emb,rnn,batch,slen,tlen,beam,alpha", [
    (64, 12, 12, 4, 7, 6, 3, 0.6),
    (500, 32, 32, 1, 20, 12, 5, 1.0),     # the reference's own (batch-1) regime
    (500, 32, 32, 9, 20, 12, 5, 0.6),
    (2000, 64, 64, 16, 30, 20, 4, 0.0),
    (500, 32, 32, 6, 20, 12, 12, 0.6),    # beams wider than 8: the 16-wide top-k instances, two attention query groups
    (300, 16, 16, 3, 9, 8, 16, 1.0),
])
def test_beam_search_matches_oracle(dev, vocab, emb, rnn, batch, slen, tlen, beam, alpha):
    model, params, ds, src, tgt = _setup(dev, vocab, emb, rnn, batch, slen, tlen, True, beam=beam,
                                         max_steps=tlen, length_normalization=alpha)
    enc = O.sentence_encoder(params, src)
    spec = O.DecoderSpec(max_output_len=max(slen, tlen))
    ref = O.beam_search(params, spec, enc, beam, tlen, alpha)
    sess = model.tf_manager.sessions[0]
    fd = {}
    for f in model.beam_runner.feedables:
        fd.update(f.feed_dict(ds, train=False))
    out = sess.run({"bs": model.beam_decoder.outputs}, fd)["bs"]
    tok = out.last_search_step_output.token_ids
    if ref.min_gap > 1e-5:
        assert tok.shape == ref.token_ids.shape
        assert np.array_equal(tok, ref.token_ids.astype(np.int32)), "beam token ids differ"
        assert np.array_equal(out.last_search_state.lengths, ref.lengths)
        assert np.array_equal(out.last_search_state.finished.astype(bool), ref.finished)
        rel = np.abs(out.last_search_step_output.scores - ref.scores).max() / np.abs(ref.scores).max()
        assert rel < 1e-4
    else:       # the oracle itself saw a near-tie: report instead of hiding it
        same = np.mean(tok[:min(len(tok), len(ref.token_ids))] ==
                       ref.token_ids[:min(len(tok), len(ref.token_ids))])
        assert same > 0.9, "near-tie reported by the oracle (gap {:.2e}) but outputs diverge widely".format(ref.min_gap)
    res = model.tf_manager.execute(ds, model.beam_runner.feedables, [model.beam_runner])[0]
    want, want_loss = O.beam_tokens(ref, 1)
    if ref.min_gap > 1e-5:
        assert res.outputs["target_beam"] == _ids_to_words(model.tgt_vocab, want)
        assert abs(res.losses["target_beam/beam_search_score"] - want_loss) < 1e-3 * abs(want_loss)


def test_beam_early_stop_and_greedy_early_stop(dev):
    """A model that emits </s> immediately: loops stop after the first step and
    histories are cropped exactly like the reference's while-loop."""
    model, params, ds, src, tgt = _setup(dev, 64, 12, 12, 4, 7, 6, True, beam=3)
    params = dict(params)
    b = params["decoder/state_to_word_b"].copy()
    b[O.END] = 50.0
    params["decoder/state_to_word_b"] = b
    model.tf_manager.sessions[0].store.load_state_dict(params)
    enc = O.sentence_encoder(params, src)
    spec = O.DecoderSpec(max_output_len=7)
    ref = O.decoding_loop(params, spec, enc, None, False)
    refb = O.beam_search(params, spec, enc, 3, 6, 0.6)
    sess = model.tf_manager.sessions[0]
    fd = {}
    for f in model.beam_runner.feedables | model.greedy_runner.feedables:
        fd.update(f.feed_dict(ds, train=False))
    got = sess.run({"sym": model.decoder.decoded_symbols, "bs": model.beam_decoder.outputs}, fd)
    assert got["sym"].shape == ref.symbols.shape == (1, 4)
    assert np.array_equal(got["sym"], ref.symbols)
    assert got["bs"].last_search_step_output.token_ids.shape == refb.token_ids.shape
    assert np.array_equal(got["bs"].last_search_step_output.token_ids, refb.token_ids)


@pytest.mark.parametrize("case", ["tiny", "mid"])
def test_committed_golden_vectors(dev, case):
    """HIP engine vs the committed fixtures (tests/golden/*.npz, generated by
    tests/golden/make_golden.py from the oracle) -- no oracle call at run time."""
    import os
    from neuralmonkey_amd import synthetic
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", case + ".npz"))
    vocab, dim, batch, slen, tlen, ragged, beam, seed = [int(x) for x in gold["meta"]]
    alpha = float(gold["alpha"])
    model = synthetic.build_translation_model(vocab_src=vocab, vocab_tgt=vocab, emb=dim, rnn=dim,
                                              max_len=max(slen, tlen), beam_size=beam, max_steps=tlen,
                                              length_normalization=alpha, device=str(dev))
    params = O.init_params(seed=seed, vocab_src=vocab, vocab_tgt=vocab, emb=dim, rnn=dim, std=0.1)
    store = model.tf_manager.sessions[0].store
    store.load_state_dict(params)
    src, tgt = gold["src"], gold["tgt"].T
    ds = Dataset("gold", {"source": [row[row != 0] for row in src],
                          "target": [row[(row != 0) & (row != O.END)] for row in tgt]},
                 BatchingScheme(batch_size=batch))
    fd = {}
    for f in model.greedy_runner.feedables | model.beam_runner.feedables:
        fd.update(f.feed_dict(ds, train=False))
    sess = model.tf_manager.sessions[0]
    got = sess.run({"states": model.encoder.temporal_states, "final": model.encoder.output,
                    "sym": model.decoder.decoded_symbols, "logits": model.decoder.runtime_logits,
                    "train_loss": model.decoder.train_loss, "runtime_loss": model.decoder.runtime_loss,
                    "bs": model.beam_decoder.outputs}, fd)
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))
    assert rel(got["states"], gold["enc_states"]) < 1e-4 and rel(got["final"], gold["enc_final"]) < 1e-4
    assert np.array_equal(got["sym"], gold["greedy_symbols"])
    assert rel(got["logits"], gold["greedy_logits"]) < 1e-4
    assert abs(float(got["train_loss"]) - float(gold["train_loss"])) < 1e-4 * float(gold["train_loss"])
    assert abs(float(got["runtime_loss"]) - float(gold["runtime_loss"])) < 1e-4 * float(gold["runtime_loss"])
    if float(gold["beam_min_gap"]) > 1e-5:
        assert np.array_equal(got["bs"].last_search_step_output.token_ids, gold["beam_token_ids"])
        assert rel(got["bs"].last_search_step_output.scores, gold["beam_scores"]) < 1e-4
    res = model.tf_manager.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
    assert abs(res.losses["decoder - cost"] - float(gold["torch_loss"])) < 1e-4 * float(gold["torch_loss"])
    assert abs(res.losses["L2"] - float(gold["l2"])) < 1e-4 * float(gold["l2"])
    for name, key in (("decoder/state_to_word_W", "grad_logit_w"), ("attention/attn_similarity_v", "grad_attn_v"),
                      ("encoder_input/embedding_matrix_0", "grad_enc_emb")):
        assert rel(store.g(name).cpu().numpy(), gold[key]) < 1e-3, name


def test_ini_experiment_trains_and_decodes(dev, tmp_path):
    """An INI config (the reference's plugin surface) builds, trains for a few
    steps with falling loss and decodes with both runners on the GPU."""
    from tests.test_host import write_ini
    from neuralmonkey_amd.config.configuration import load_experiment
    model = load_experiment(write_ini(tmp_path), changes=['tf_manager.device="{}"'.format(dev)])
    tfm = model.tf_manager
    batch = next(model.train_dataset.batches())
    feedables = set.union(*[r.feedables for r in model.runners + model.trainers])
    losses = []
    for _ in range(30):
        out = tfm.execute(batch, feedables, model.trainers, train=True)
        losses.append(out[0].losses["decoder - cost"])
    assert losses[-1] < losses[0] - 0.02
    results = tfm.execute(batch, feedables, model.runners)
    assert [len(r.outputs[s]) for r, s in zip(results, ("target", "target_beam.rank001", "target_beam.rank002"))] \
        == [4, 4, 4]
    assert all("<unk>" not in sent for sent in results[0].outputs["target"])       # supress_unk


def test_decode_graph_replay_matches_eager_on_fresh_batches(dev):
    """Greedy and beam decoding capture their step chunks into HIP graphs (eager -> capture ->
    replay).  Five different batches of the same shape: every pass, replayed ones included, must
    reproduce the oracle on its own batch (no stale pointers baked into the graphs)."""
    from neuralmonkey_amd import synthetic
    vocab, emb, rnn, batch, slen = 120, 16, 16, 6, 9
    model = synthetic.build_translation_model(vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn, max_len=12,
                                              beam_size=3, max_steps=12, device=str(dev), with_trainer=False)
    params = O.init_params(seed=21, vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn, std=0.3)
    sess = model.tf_manager.sessions[0]
    sess.store.load_state_dict(params)
    assert sess.use_graphs
    spec = O.DecoderSpec(max_output_len=12)
    for i in range(5):
        ds = synthetic.synthetic_dataset(seed=100 + i, batch=batch, src_len=slen, tgt_len=slen, vocab=vocab,
                                         ragged=True, with_target=False)
        src = O.pad_ids([list(s) for s in ds.get_series("source")], 12)
        enc = O.sentence_encoder(params, src)
        want = O.greedy_tokens(O.decoding_loop(params, spec, enc, None, False))
        res = model.tf_manager.execute(ds, model.greedy_runner.feedables | model.beam_runner.feedables,
                                       [model.greedy_runner, model.beam_runner], compute_losses=False)
        w2i = model.tgt_vocab._word_to_index
        got = [[w2i[w] for w in sent] for sent in res[0].outputs["target"]]
        assert got == want, "greedy mismatch on pass {}".format(i)
        bres = O.beam_search(params, spec, enc, 3, 12, 0.6)
        want_beam, _ = O.beam_tokens(bres, 1)
        got_beam = [[w2i[w] for w in sent] for sent in res[1].outputs["target_beam"]]
        if bres.min_gap > 1e-5:
            assert got_beam == want_beam, "beam mismatch on pass {}".format(i)


def test_greedy_chunks_captured_at_different_times_keep_the_state(dev):
    """the greedy loop runs in chunks of 8 steps, each captured as a HIP graph the second time
    it is reached.  Batches of one shape whose outputs end inside the first chunk (eager, capture, replay) are
    followed by outputs that run on: the later chunks are then launched from Python while chunk 0 is a replayed
    graph whose body no longer runs -- the stepper must not fall back to the initial state it was started
    with.  Every pass is compared with the oracle, fused step on and off."""
    import os
    from neuralmonkey_amd import synthetic
    vocab, emb, rnn, batch, slen, tmax = 120, 16, 16, 6, 9, 24
    params = O.init_params(seed=33, vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn, std=0.3)
    spec = O.DecoderSpec(max_output_len=tmax)
    w2i = None
    for fused in (True, False):
        old = os.environ.pop("NM_NO_FUSED_STEP", None)
        if not fused:
            os.environ["NM_NO_FUSED_STEP"] = "1"
        try:
            model = synthetic.build_translation_model(vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn,
                                                      max_len=tmax, beam_size=3, max_steps=tmax, device=str(dev),
                                                      with_trainer=False)
            sess = model.tf_manager.sessions[0]
            assert sess.use_graphs
            w2i = model.tgt_vocab._word_to_index
            lengths = []
            for i, end_bias in enumerate((50.0, 50.0, 50.0, -1e9, -1e9, 50.0, -1e9)):
                p = dict(params)
                b = p["decoder/state_to_word_b"].copy()
                b[O.END] = end_bias
                p["decoder/state_to_word_b"] = b
                sess.store.load_state_dict(p)
                ds = synthetic.synthetic_dataset(seed=300 + i, batch=batch, src_len=slen, tgt_len=slen, vocab=vocab,
                                                 ragged=True, with_target=False)
                src = O.pad_ids([list(s) for s in ds.get_series("source")], tmax)
                want = O.greedy_tokens(O.decoding_loop(p, spec, O.sentence_encoder(p, src), None, False))
                res = model.tf_manager.execute(ds, model.greedy_runner.feedables, [model.greedy_runner],
                                               compute_losses=False)[0]
                got = [[w2i[w] for w in sent] for sent in res.outputs["target"]]
                assert got == want, "greedy mismatch on pass {} (fused step {})".format(i, fused)
                lengths.append(max(len(s) for s in want))
            assert min(lengths) <= 1 and max(lengths) == tmax
        finally:
            os.environ.pop("NM_NO_FUSED_STEP", None)
            if old is not None:
                os.environ["NM_NO_FUSED_STEP"] = old


def test_lookahead_encodes_the_next_batch_on_a_second_stream(dev):
    """``TensorFlowManager.execute(..., lookahead=next_batch)``: the next batch's encoder states, attention keys and
    initial decoder state are evaluated on a second stream, in the other buffer slot, while the current batch
    decodes.  Seven batches (the slots alternate; eager, capture and replay passes in both), one announced batch
    that never comes, one batch of another shape in between: every greedy and beam result is the oracle's."""
    from neuralmonkey_amd import synthetic
    vocab, emb, rnn, slen = 120, 16, 16, 9
    model = synthetic.build_translation_model(vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn, max_len=12,
                                              beam_size=3, max_steps=12, device=str(dev), with_trainer=False)
    params = O.init_params(seed=21, vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn, std=0.3)
    sess = model.tf_manager.sessions[0]
    sess.store.load_state_dict(params)
    spec = O.DecoderSpec(max_output_len=12)
    w2i = model.tgt_vocab._word_to_index
    sizes = [6, 6, 6, 4, 6, 6, 6]
    batches = [synthetic.synthetic_dataset(seed=500 + i, batch=b, src_len=slen, tgt_len=slen, vocab=vocab, ragged=True,
                                           with_target=False) for i, b in enumerate(sizes)]
    never = synthetic.synthetic_dataset(seed=999, batch=6, src_len=slen, tgt_len=slen, vocab=vocab, ragged=True,
                                        with_target=False)
    runners = [model.greedy_runner, model.beam_runner]
    feedables = model.greedy_runner.feedables | model.beam_runner.feedables
    slots = []
    for i, ds in enumerate(batches):
        nxt = batches[i + 1] if i + 1 < len(batches) else None
        if i == 4:
            nxt = never                                   # announced, never executed: dropped at the next run
        res = model.tf_manager.execute(ds, feedables, runners, compute_losses=False, lookahead=nxt)
        slots.append(sess.slot)
        src = O.pad_ids([list(s) for s in ds.get_series("source")], 12)
        enc = O.sentence_encoder(params, src)
        want = O.greedy_tokens(O.decoding_loop(params, spec, enc, None, False))
        got = [[w2i[w] for w in sent] for sent in res[0].outputs["target"]]
        assert got == want, "greedy mismatch on pass {}".format(i)
        bres = O.beam_search(params, spec, enc, 3, 12, 0.6)
        if bres.min_gap > 1e-5:
            want_beam, _ = O.beam_tokens(bres, 1)
            assert [[w2i[w] for w in sent] for sent in res[1].outputs["target_beam"]] == want_beam, i
    assert slots[:5] == [0, 1, 0, 1, 0] and slots[5] == 0 and slots[6] == 1      # batch 5 was not announced
    assert not sess._ahead
    # variables rewritten between the announcement and the run: what was encoded ahead is stale and must be dropped
    model.tf_manager.execute(batches[0], feedables, runners, compute_losses=False, lookahead=batches[1])
    assert sess._ahead
    name = model.encoder.input_sequence.embedding_matrix_name
    sess.store[name].mul_(-1.0)
    p2 = dict(params)
    p2[name] = -params[name]
    res = model.tf_manager.execute(batches[1], feedables, runners, compute_losses=False)
    src = O.pad_ids([list(s) for s in batches[1].get_series("source")], 12)
    want = O.greedy_tokens(O.decoding_loop(p2, spec, O.sentence_encoder(p2, src), None, False))
    assert [[w2i[w] for w in sent] for sent in res[0].outputs["target"]] == want


def test_input_tables_follow_the_variables(dev):
    """The decoder step reads E.Wg_x / E.Wc_x / E.Wo_e from a table indexed by the input symbol
    (decoder_general.input_table).  The table must follow the CONTENTS of the variables: decode, change the
    embedding of one frequent word in place (no official mutation path involved), decode again -- both results are
    the oracle's on the respective weights, with tables on and off."""
    import os
    from neuralmonkey_amd import synthetic
    vocab, emb, rnn, batch, slen = 120, 16, 16, 6, 9
    params = O.init_params(seed=41, vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn, std=0.3)
    spec = O.DecoderSpec(max_output_len=12)
    ds = synthetic.synthetic_dataset(seed=77, batch=batch, src_len=slen, tgt_len=slen, vocab=vocab, ragged=True,
                                     with_target=False)
    src = O.pad_ids([list(s) for s in ds.get_series("source")], 12)
    for tables in ("1", "0"):
        os.environ["NM_STEP_TABLES"] = tables
        try:
            model = synthetic.build_translation_model(vocab_src=vocab, vocab_tgt=vocab, emb=emb, rnn=rnn, max_len=12,
                                                      beam_size=3, max_steps=12, device=str(dev), with_trainer=False)
            sess = model.tf_manager.sessions[0]
            sess.store.load_state_dict(params)
            w2i = model.tgt_vocab._word_to_index
            p = dict(params)
            for round_ in range(3):
                if round_ == 1:                     # poke the embeddings of the decoder in place
                    name = model.decoder.embedding_matrix_name
                    first = O.greedy_tokens(O.decoding_loop(p, spec, O.sentence_encoder(p, src), None, False))
                    word = max(set(sum(first, [])), key=sum(first, []).count) if sum(first, []) else 5
                    sess.store[name][word].mul_(-2.0)
                    p = dict(p)
                    e2 = p[name].copy()
                    e2[word] *= -2.0
                    p[name] = e2
                enc = O.sentence_encoder(p, src)
                want = O.greedy_tokens(O.decoding_loop(p, spec, enc, None, False))
                res = model.tf_manager.execute(ds, model.greedy_runner.feedables | model.beam_runner.feedables,
                                               [model.greedy_runner, model.beam_runner], compute_losses=False)
                got = [[w2i[w] for w in sent] for sent in res[0].outputs["target"]]
                assert got == want, "greedy mismatch in round {} (tables {})".format(round_, tables)
                bres = O.beam_search(p, spec, enc, 3, 12, 0.6)
                if bres.min_gap > 1e-5:
                    want_beam, _ = O.beam_tokens(bres, 1)
                    assert [[w2i[w] for w in sent] for sent in res[1].outputs["target_beam"]] == want_beam
            if tables == "1":
                assert sess.__dict__.get("_input_tables"), "the fused stepper did not build its input table"
        finally:
            os.environ.pop("NM_STEP_TABLES", None)


@pytest.mark.parametrize("seed,end_bias,expect", [(7, 3.0, "early"), (8, 1.5, "mid"), (26, 2.0, "mid"),
                                                  (7, -50.0, "full")])
def test_loops_that_finish_early_mid_and_never_equal_the_chunk_by_chunk_path(dev, monkeypatch, seed, end_bias,
                                                                             expect):
    """The decoding loops enqueue one chunk of steps AHEAD of the finished-flag read-back (Session.decode_chunks);
    whatever the step at which a batch finishes, symbols / masks / beam histories equal the path that reads every
    chunk's flags before enqueueing the next one, and the oracle's while-loop (autoregressive.py:425-437,
    beam_search_decoder.py:330-355)."""
    tmax = 50
    model, params, ds, src, tgt = _setup(dev, 96, 16, 16, 6, 9, tmax, True, beam=3, max_steps=tmax, std=0.25,
                                         seed=seed)
    params = dict(params)
    b = params["decoder/state_to_word_b"].copy()
    b[O.END] = end_bias
    params["decoder/state_to_word_b"] = b
    sess = model.tf_manager.sessions[0]
    sess.store.load_state_dict(params)
    enc = O.sentence_encoder(params, src)
    spec = O.DecoderSpec(max_output_len=tmax)
    ref = O.decoding_loop(params, spec, enc, None, False)
    refb = O.beam_search(params, spec, enc, 3, tmax, 0.6)
    steps = ref.symbols.shape[0]
    assert {"early": steps <= 6, "mid": 8 < steps < 40, "full": steps == tmax}[expect], steps     # 4 / 10 / 15 / 50
    fd = {}
    for f in model.beam_runner.feedables | model.greedy_runner.feedables:
        fd.update(f.feed_dict(ds, train=False))
    fetch = {"sym": model.decoder.decoded_symbols, "mask": model.decoder.runtime_mask,
             "bs": model.beam_decoder.outputs}
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NM_DECODE_RUN_AHEAD", mode)
        for _ in range(2):                      # second run: every chunk replays its captured graph
            got = sess.run(fetch, fd)
        outs[mode] = (np.asarray(got["sym"]).copy(), np.asarray(got["mask"]).copy(),
                      np.asarray(got["bs"].last_search_step_output.token_ids).copy(),
                      np.asarray(got["bs"].last_search_step_output.scores).copy())
    for a, b_ in zip(outs["1"], outs["0"]):
        assert a.shape == b_.shape and np.array_equal(a, b_)
    sym, mask, tok, _ = outs["1"]
    assert sym.shape == ref.symbols.shape and np.array_equal(sym, ref.symbols)
    assert np.array_equal(mask.astype(bool), ref.mask)
    assert tok.shape == refb.token_ids.shape
    if refb.min_gap > 1e-5:
        assert np.array_equal(tok[1:], refb.token_ids[1:])


def test_stateful_filler_under_a_decoder(dev):
    """``StatefulFiller`` (encoders/numpy_stateful_filler.py:16-72) with its dense projection as the only encoder of an
    RNN decoder without attention: the decoder's initial state comes from the projected vectors, and a training step
    gives the projection's gradients that float64 autograd gives (``initial_state = dense(output)``,
    decoders/encoder_projection.py:47-73)."""
    import torch
    from neuralmonkey_amd.dataset import BatchingScheme, Dataset
    from neuralmonkey_amd.decoders import Decoder
    from neuralmonkey_amd.encoders.numpy_stateful_filler import StatefulFiller
    from neuralmonkey_amd.runtime import reset_registry
    from neuralmonkey_amd.tf_manager import TensorFlowManager
    from neuralmonkey_amd.trainers.cross_entropy_trainer import CrossEntropyTrainer
    from neuralmonkey_amd.vocabulary import Vocabulary
    reset_registry()
    rng = np.random.default_rng(5)
    vocab = Vocabulary(["w{}".format(i) for i in range(11)])
    filler = StatefulFiller("vec", 7, "vectors", output_shape=5)
    dec = Decoder(encoders=[filler], vocabulary=vocab, data_id="target", name="decoder", max_output_len=4,
                  embedding_size=6, rnn_size=6, dropout_keep_prob=1.0)
    trainer = CrossEntropyTrainer(decoders=[dec])
    tfm = TensorFlowManager(num_sessions=1, num_threads=1, device=str(dev), seed=3)
    tfm.initialize_sessions()
    store = tfm.sessions[0].store
    vectors = [rng.normal(size=7).astype(np.float32) for _ in range(3)]
    ds = Dataset("d", {"vectors": vectors, "target": [["w1", "w2"], ["w3"], ["w4", "w5", "w6"]]},
                 BatchingScheme(batch_size=3))
    fd = {}
    for part in trainer.feedables:
        fd.update(part.feed_dict(ds, train=False))
    out = tfm.sessions[0].run({"enc": filler.output}, fd)["enc"]
    w, b = store["vec/dense/kernel"].cpu().numpy(), store["vec/dense/bias"].cpu().numpy()
    assert np.allclose(np.asarray(out), np.stack(vectors) @ w + b, atol=1e-5)
    before = w.copy()
    tfm.execute(ds, trainer.feedables, [trainer], train=True)
    grad = store.g("vec/dense/kernel").cpu().numpy()
    assert np.abs(grad).max() > 0 and not np.array_equal(store["vec/dense/kernel"].cpu().numpy(), before)
    # d loss / d kernel = vectors^T . d loss / d output: rank <= batch size, rows in the span of the fed vectors
    assert np.linalg.matrix_rank(grad.astype(np.float64), tol=1e-6 * np.abs(grad).max()) <= 3


def test_train_logprobs_is_the_log_softmax_of_the_train_logits(dev):
    """``AutoregressiveDecoder.train_logprobs`` (autoregressive.py:288-290; new here, for runners that fetch it by
    name): tf.nn.log_softmax of the teacher-forced logits."""
    import torch
    from neuralmonkey_amd import synthetic
    model = synthetic.build_translation_model(vocab_src=300, vocab_tgt=300, emb=32, rnn=32, max_len=12, beam_size=0,
                                              device=str(dev))
    ds = synthetic.synthetic_dataset(seed=4, batch=6, src_len=9, tgt_len=8, vocab=300, ragged=True)
    dec = model.decoder
    fd = {}
    for part in model.trainer.feedables:
        fd.update(part.feed_dict(ds, train=False))
    out = model.tf_manager.sessions[0].run({"logits": dec.train_logits, "logprobs": dec.train_logprobs}, fd)
    want = torch.log_softmax(torch.as_tensor(np.asarray(out["logits"])).double(), -1).numpy()
    assert np.abs(np.asarray(out["logprobs"]) - want).max() < 1e-5


def test_a_set_device_error_word_reaches_the_caller(dev):
    """The kernels' only way to say "my results are garbage" is the session's device error word (cluster time loops
    whose hand-offs timed out: csrc/nm_gru_cluster.hip, ``sticky_error``).  It travels to the host with a training
    step's losses and is read after every inference batch.  The first time, the session falls back to the per-step
    path and runs the work again (tests/test_cluster_recovery_gpu.py); a word raised with the loops already off is
    the fallback's own failure and reaches the caller as an exception.  (The word is set by hand here.)"""
    import warnings
    from neuralmonkey_amd import synthetic
    model = synthetic.build_translation_model(vocab_src=300, vocab_tgt=300, emb=32, rnn=32, max_len=12, beam_size=0,
                                              device=str(dev))
    ds = synthetic.synthetic_dataset(seed=4, batch=6, src_len=9, tgt_len=8, vocab=300, ragged=True)
    tfm = model.tf_manager
    sess = tfm.sessions[0]
    res = tfm.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
    assert res.losses["decoder - cost"] > 0                       # a clean word: the losses read as ever
    tfm.execute(ds, model.greedy_runner.feedables, [model.greedy_runner])
    want = tfm.execute(ds, model.greedy_runner.feedables, [model.greedy_runner])[0].outputs["target"]
    was_on = sess.use_cluster_loops
    sess.error_word().fill_(1)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = tfm.execute(ds, model.greedy_runner.feedables, [model.greedy_runner])[0].outputs["target"]
    if was_on:
        assert got == want and not sess.use_cluster_loops
        assert len([w for w in caught if "gave up waiting" in str(w.message)]) == 1
        sess.error_word().fill_(1)
    # the loops are off now: a set word is the fallback's failure
    res = tfm.execute(ds, model.trainer.feedables, [model.trainer], train=True)[0]
    with pytest.raises(RuntimeError, match="gave up waiting"):
        res.losses["decoder - cost"]
    sess.error_word().fill_(1)
    with pytest.raises(RuntimeError, match="gave up waiting"):
        tfm.execute(ds, model.greedy_runner.feedables, [model.greedy_runner], lookahead=ds)
    sess.error_word().fill_(1)
    with pytest.raises(RuntimeError, match="gave up waiting"):
        tfm.execute(ds, model.greedy_runner.feedables, [model.greedy_runner])
