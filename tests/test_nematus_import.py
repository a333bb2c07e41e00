"""Nematus model import (scripts/import_nematus.py of the reference): a synthetic
Nematus parameter file + JSON dictionaries go through ``nematus_import`` into a model built from the
generated INI sections.  Host-side only (CPU device: variables are created, no kernel runs)."""
import json

import numpy as np
import pytest

DIM_WORD, DIM, N_SRC, N_TGT = 6, 4, 9, 8           # Nematus sizes: vocabularies count eos + UNK + words


def _nematus_arrays(rng):
    r = lambda *shape: rng.standard_normal(shape).astype(np.float32)
    ctx = 2 * DIM
    arrays = {"Wemb": r(N_SRC, DIM_WORD), "Wemb_dec": r(N_TGT, DIM_WORD),
              "ff_logit_W": r(DIM_WORD, N_TGT), "ff_logit_b": r(N_TGT),
              "ff_state_W": r(ctx, DIM), "ff_state_b": r(DIM),
              "decoder_Wc_att": r(ctx, ctx), "decoder_b_att": r(ctx), "decoder_W_comb_att": r(DIM, ctx),
              "decoder_U_att": r(ctx, 1), "decoder_c_tt": r(1),
              "decoder_U_nl": r(DIM, 2 * DIM), "decoder_Wc": r(ctx, 2 * DIM), "decoder_b_nl": r(2 * DIM),
              "decoder_Ux_nl": r(DIM, DIM), "decoder_Wcx": r(ctx, DIM), "decoder_bx_nl": r(DIM),
              "ff_logit_lstm_W": r(DIM, DIM_WORD), "ff_logit_lstm_b": r(DIM_WORD),
              "ff_logit_prev_W": r(DIM_WORD, DIM_WORD), "ff_logit_prev_b": r(DIM_WORD),
              "ff_logit_ctx_W": r(ctx, DIM_WORD), "ff_logit_ctx_b": r(DIM_WORD)}
    for prefix, d_in in (("encoder_", DIM_WORD), ("encoder_r_", DIM_WORD), ("decoder_", DIM_WORD)):
        arrays.update({prefix + "U": r(DIM, 2 * DIM), prefix + "W": r(d_in, 2 * DIM), prefix + "b": r(2 * DIM),
                       prefix + "Ux": r(DIM, DIM), prefix + "Wx": r(d_in, DIM), prefix + "bx": r(DIM)})
    return arrays


def _write_model(root, rng):
    words = ["w{}".format(i) for i in range(20)]
    src = {"eos": 0, "UNK": 1, **{w: i + 2 for i, w in enumerate(words[:N_SRC - 2])}}
    tgt = {"eos": 0, "UNK": 1, **{w: i + 2 for i, w in enumerate(words[5:5 + N_TGT - 2])}}
    (root / "src.json").write_text(json.dumps(src))
    (root / "tgt.json").write_text(json.dumps(tgt))
    arrays = _nematus_arrays(rng)
    np.savez(root / "model.npz", **arrays)
    (root / "model.npz.json").write_text(json.dumps({
        "encoder": "gru", "decoder": "gru_cond", "n_words_src": N_SRC, "n_words": N_TGT,
        "saveto": "model.npz", "dim": DIM, "dim_word": DIM_WORD, "dictionaries": ["src.json", "tgt.json"],
        "maxlen": 12}))
    return arrays


def test_nematus_model_lands_in_the_engine_variables(tmp_path):
    from neuralmonkey_amd import nematus_import as N
    from neuralmonkey_amd.config.configuration import load_experiment
    rng = np.random.default_rng(0)
    arrays = _write_model(tmp_path, rng)
    config = N.load_nematus_json(str(tmp_path / "model.npz.json"))
    assert config["rnn_size"] == DIM and config["embedding_size"] == DIM_WORD
    ini = tmp_path / "imported.ini"
    ini.write_text("[main]\nname=\"imported\"\ndecoder=<decoder>\n\n" + N.experiment_ini(config))
    model = load_experiment(str(ini), device="cpu", seed=1)
    store = model.tf_manager.sessions[0].store
    assert len(model.decoder.vocabulary) == N_TGT + 2            # <pad>, <s> in front of eos / UNK / words
    untouched = N.import_model(str(tmp_path / "model.npz.json"), store)
    assert untouched == []
    get = lambda name: store[name].cpu().numpy()
    emb = get("encoder_input/embedding_matrix_0")
    assert emb.shape == (N_SRC + 2, DIM_WORD) and np.all(emb[:2] == 0) and np.array_equal(emb[2:], arrays["Wemb"])
    w = get("decoder/state_to_word_W")
    assert w.shape == (DIM_WORD, N_TGT + 2) and np.all(w[:, :2] == 0) and np.array_equal(w[:, 2:], arrays["ff_logit_W"])
    b = get("decoder/state_to_word_b")
    assert np.all(b[:2] == 0) and np.array_equal(b[2:], arrays["ff_logit_b"])
    assert np.array_equal(get("attention/attn_similarity_v"), arrays["decoder_U_att"].reshape(-1))
    assert np.array_equal(get("attention/attn_bias"), arrays["decoder_c_tt"].reshape(-1))
    assert np.array_equal(get("encoder/rnn_0_bidirectional/bidirectional_rnn/bw/nematus_gru_cell/candidate/"
                              "input_proj/kernel"), arrays["encoder_r_Wx"])
    assert np.array_equal(get("decoder/attention_decoder/cond_gru_2_cell/gates/state_proj/bias"),
                          arrays["decoder_b_nl"])
    assert np.array_equal(get("decoder/attention_decoder/context/kernel"), arrays["ff_logit_ctx_W"])
    assert np.array_equal(get("decoder/initial_state/encoders_projection/kernel"), arrays["ff_state_W"])


def test_nematus_import_refuses_misfits(tmp_path):
    from neuralmonkey_amd import nematus_import as N
    rng = np.random.default_rng(1)
    _write_model(tmp_path, rng)
    bad = json.loads((tmp_path / "model.npz.json").read_text())
    bad["decoder"] = "gru"
    (tmp_path / "bad.json").write_text(json.dumps(bad))
    with pytest.raises(ValueError, match="Unsupported decoder type"):
        N.load_nematus_json(str(tmp_path / "bad.json"))
    bad["decoder"], bad["dictionaries"] = "gru_cond", ["missing.json", "tgt.json"]
    (tmp_path / "bad.json").write_text(json.dumps(bad))
    with pytest.raises(FileNotFoundError):
        N.load_nematus_json(str(tmp_path / "bad.json"))
    assert np.array_equal(N.prepend_special_slots(np.ones((2, 3)), 1), [[0, 0, 1, 1, 1], [0, 0, 1, 1, 1]])

    class Store:                                           # a variable whose shape cannot take the array
        def names(self):
            return ["attention/attn_similarity_v"]

        def __getitem__(self, name):
            return np.zeros(5, np.float32)
    with pytest.raises(ValueError, match="does not fit"):
        N.import_variables(Store(), {"decoder_U_att": np.ones((8, 1), np.float32)})
