"""Import of a Nematus model (``model.npz`` + ``model.npz.json`` + JSON dictionaries) into the
engine's variable store (mirror of the reference's scripts/import_nematus.py; SURVEY 8f row 1: the
variable-name contract).  Host-side only: arrays are read with NumPy, re-laid-out and handed to
``VariableStore.load_state_dict``; the model they land in is the ``tests/nematus.ini`` family --
NematusGRU encoder and conditional-GRU decoder, ``nematus_projection`` initial state,
``nematus_output`` projection, Bahdanau attention.

Layout differences covered (scripts/import_nematus.py:31-130):

* Nematus knows two special symbols (``eos`` = 0, ``UNK`` = 1), this vocabulary four (``<pad>``,
  ``<s>``, ``</s>``, ``<unk>``): every tensor with a vocabulary axis gets two zero slices in front
  (``emb_fix``), and the dictionaries are read with ``vocabulary.from_nematus_json``.
* ``decoder_U_att`` [A,1] and ``decoder_c_tt`` [1] are the similarity vector / scalar bias.
* The reference script still names encoder variables as its older one-layer encoder did
  (``encoder/bidirectional_rnn/fw/...``); the encoder of this commit scopes them per layer
  (``encoder/rnn_0_bidirectional/bidirectional_rnn/fw/...``, encoders/recurrent.py:71-110).  Both
  spellings are accepted as targets.
"""
import json
import os
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

Transform = Optional[Callable[[np.ndarray], np.ndarray]]


def prepend_special_slots(array: np.ndarray, axis: int = 0) -> np.ndarray:
    """Two zero slices for <pad> and <s> in front of the vocabulary axis (``emb_fix``, :34-63)."""
    array = np.asarray(array)
    shape = list(array.shape)
    shape[axis] = 2
    return np.concatenate([np.zeros(shape, dtype=array.dtype), array], axis=axis)


def _vocab_rows(a):
    return prepend_special_slots(a, 0)


def _vocab_cols(a):
    return prepend_special_slots(a, 1)


def _squeeze(a):
    return np.asarray(a).reshape(-1)


def variable_map(encoder: str = "encoder", decoder: str = "decoder", attention: str = "attention") \
        -> Dict[str, Tuple[str, Transform]]:
    """engine variable name -> (Nematus array name, re-layout) (scripts/import_nematus.py:88-130)."""
    table: Dict[str, Tuple[str, Transform]] = {
        "{}_input/embedding_matrix_0".format(encoder): ("Wemb", _vocab_rows),
        "{}/word_embeddings".format(decoder): ("Wemb_dec", _vocab_rows),
        "{}/state_to_word_W".format(decoder): ("ff_logit_W", _vocab_cols),
        "{}/state_to_word_b".format(decoder): ("ff_logit_b", _vocab_rows),
        "{}/initial_state/encoders_projection/kernel".format(decoder): ("ff_state_W", None),
        "{}/initial_state/encoders_projection/bias".format(decoder): ("ff_state_b", None),
        "{}/attn_key_projection".format(attention): ("decoder_Wc_att", None),
        "{}/attn_projection_bias".format(attention): ("decoder_b_att", None),
        "{}/Attention/attn_query_projection".format(attention): ("decoder_W_comb_att", None),
        "{}/attn_similarity_v".format(attention): ("decoder_U_att", _squeeze),
        "{}/attn_bias".format(attention): ("decoder_c_tt", _squeeze),
    }
    cell = {"gates/state_proj/kernel": "U", "gates/input_proj/kernel": "W", "gates/input_proj/bias": "b",
            "candidate/state_proj/kernel": "Ux", "candidate/input_proj/kernel": "Wx",
            "candidate/input_proj/bias": "bx"}
    for direction, prefix in (("fw", "encoder_"), ("bw", "encoder_r_")):
        for local, suffix in cell.items():
            scope = "{}/rnn_0_bidirectional/bidirectional_rnn/{}/nematus_gru_cell".format(encoder, direction)
            table["{}/{}".format(scope, local)] = (prefix + suffix, None)
    for local, suffix in cell.items():
        table["{}/attention_decoder/nematus_gru_cell/{}".format(decoder, local)] = ("decoder_" + suffix, None)
    cond = {"gates/state_proj/kernel": "decoder_U_nl", "gates/input_proj/kernel": "decoder_Wc",
            "gates/state_proj/bias": "decoder_b_nl", "candidate/state_proj/kernel": "decoder_Ux_nl",
            "candidate/input_proj/kernel": "decoder_Wcx", "candidate/state_proj/bias": "decoder_bx_nl"}
    for local, name in cond.items():
        table["{}/attention_decoder/cond_gru_2_cell/{}".format(decoder, local)] = (name, None)
    for layer, stem in (("rnn_state", "ff_logit_lstm"), ("prev_out", "ff_logit_prev"), ("context", "ff_logit_ctx")):
        table["{}/attention_decoder/{}/kernel".format(decoder, layer)] = (stem + "_W", None)
        table["{}/attention_decoder/{}/bias".format(decoder, layer)] = (stem + "_b", None)
    return table


def load_nematus_json(path: str) -> Dict:
    """``model.npz.json`` -> the fields the importer needs (scripts/import_nematus.py:137-176)."""
    with open(path, "r", encoding="utf-8") as handle:
        contents = json.load(handle)
    prefix = os.path.realpath(os.path.dirname(path))
    config = {"encoder_type": contents["encoder"], "decoder_type": contents["decoder"],
              "n_words_src": contents["n_words_src"], "n_words_tgt": contents["n_words"],
              "variables_file": contents["saveto"], "rnn_size": contents["dim"],
              "embedding_size": contents["dim_word"],
              "src_vocabulary": os.path.join(prefix, contents["dictionaries"][0]),
              "tgt_vocabulary": os.path.join(prefix, contents["dictionaries"][1]),
              "max_length": contents["maxlen"]}
    if config["encoder_type"] != "gru":
        raise ValueError("Unsupported encoder type: {}".format(config["encoder_type"]))
    if config["decoder_type"] != "gru_cond":
        raise ValueError("Unsupported decoder type: {}".format(config["decoder_type"]))
    for key in ("src_vocabulary", "tgt_vocabulary"):
        if not os.path.isfile(config[key]):
            raise FileNotFoundError("Vocabulary file not found: {}".format(config[key]))
    return config


def experiment_ini(config: Dict, encoder: str = "encoder", decoder: str = "decoder",
                   attention: str = "attention") -> str:
    """Model sections of an experiment that receives the imported variables (the templates of
    scripts/import_nematus.py:178-303 in this commit's constructor signatures)."""
    # Nematus' n_words counts eos and UNK, which map onto </s> and <unk>: the vocabulary holds the
    # n_words - 2 words of index 2.. behind the four special tokens, so that row i + 2 of the padded
    # embedding matrix is Nematus row i.  (The reference template passes max_size=n_words with
    # pad_to_max_size=True, which yields n_words + 6 entries against n_words + 2 matrix rows.)
    config = dict(config, src_words=config["n_words_src"] - 2, tgt_words=config["n_words_tgt"] - 2)
    return """\
[vocabulary_src]
class=vocabulary.from_nematus_json
path="{src_vocabulary}"
max_size={src_words}
pad_to_max_size=False

[vocabulary_tgt]
class=vocabulary.from_nematus_json
path="{tgt_vocabulary}"
max_size={tgt_words}
pad_to_max_size=False

[input_sequence]
class=model.sequence.EmbeddedSequence
name="{encoder}_input"
vocabulary=<vocabulary_src>
data_id="source"
embedding_size={embedding_size}
max_length={max_length}
add_end_symbol=True

[encoder]
class=encoders.RecurrentEncoder
name="{encoder}"
input_sequence=<input_sequence>
rnn_layers=[({rnn_size}, "bidirectional", "NematusGRU")]
include_final_layer_norm=False
dropout_keep_prob=1.0

[attention]
class=attention.Attention
name="{attention}"
encoder=<encoder>
dropout_keep_prob=1.0

[nematus_nonlinear]
class=decoders.output_projection.nematus_output
output_size={embedding_size}
dropout_keep_prob=1.0

[nematus_mean]
class=decoders.encoder_projection.nematus_projection
dropout_keep_prob=1.0

[decoder]
class=decoders.Decoder
name="{decoder}"
vocabulary=<vocabulary_tgt>
data_id="target"
embedding_size={embedding_size}
rnn_size={rnn_size}
max_output_len={max_length}
encoders=[<encoder>]
encoder_projection=<nematus_mean>
attentions=[<attention>]
attention_on_input=False
conditional_gru=True
output_projection=<nematus_nonlinear>
rnn_cell="NematusGRU"
dropout_keep_prob=1.0
""".format(encoder=encoder, decoder=decoder, attention=attention, **config)


def import_variables(store, nematus: Dict[str, np.ndarray], encoder: str = "encoder", decoder: str = "decoder",
                     attention: str = "attention") -> List[str]:
    """Copy the arrays of a Nematus model into ``store``; returns the engine variables left untouched.
    Shape mismatches raise (the reference only logs them and lets the assign fail, :26-28, :333-341)."""
    values: Dict[str, np.ndarray] = {}
    shapes = {name: tuple(store[name].shape) for name in store.names()}
    for target, (source, transform) in variable_map(encoder, decoder, attention).items():
        names = [target, target.replace("/rnn_0_bidirectional/", "/")]        # current and legacy encoder scopes
        found = next((n for n in names if n in shapes), None)
        if found is None or source not in nematus:
            continue
        array = np.asarray(nematus[source], dtype=np.float32)
        if transform is not None:
            array = transform(array)
        if int(np.prod(array.shape)) != int(np.prod(shapes[found])) or \
                (array.ndim == len(shapes[found]) and tuple(array.shape) != shapes[found]):
            raise ValueError("Nematus array '{}' of shape {} does not fit variable '{}' of shape {}"
                             .format(source, tuple(array.shape), found, shapes[found]))
        values[found] = array.reshape(shapes[found])
    store.load_state_dict(values, strict=False)
    return sorted(set(shapes) - set(values))


def import_model(json_path: str, store, encoder: str = "encoder", decoder: str = "decoder",
                 attention: str = "attention") -> List[str]:
    """``model.npz.json`` next to its ``.npz``: read both and fill ``store``."""
    config = load_nematus_json(json_path)
    variables_file = config["variables_file"]
    if not os.path.isabs(variables_file):
        variables_file = os.path.join(os.path.dirname(os.path.realpath(json_path)), os.path.basename(variables_file))
    with np.load(variables_file) as data:
        arrays = {name: data[name] for name in data.files}
    return import_variables(store, arrays, encoder, decoder, attention)
