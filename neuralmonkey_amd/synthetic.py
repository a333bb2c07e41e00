"""Synthetic translation.ini-shape model and data.

Used by bench.py, ``__graft_entry__.smoke()`` and the parity tests: the model is
assembled from the same plugin classes an INI file names
(examples/translation.ini:90-134 topology: SentenceEncoder -> Attention ->
Decoder -> CrossEntropyTrainer / GreedyRunner / BeamSearchRunner)."""
from typing import NamedTuple, Optional

import numpy as np

from .attention import Attention
from .dataset import BatchingScheme, Dataset
from .decoders import BeamSearchDecoder, Decoder
from .encoders import SentenceEncoder
from .runners import BeamSearchRunner, GreedyRunner
from .runtime import reset_registry
from .tf_manager import TensorFlowManager
from .vocabulary import Vocabulary


class TranslationModel(NamedTuple):
    encoder: SentenceEncoder
    attention: Attention
    decoder: Decoder
    beam_decoder: Optional[BeamSearchDecoder]
    greedy_runner: GreedyRunner
    beam_runner: Optional[BeamSearchRunner]
    trainer: object
    tf_manager: TensorFlowManager
    src_vocab: Vocabulary
    tgt_vocab: Vocabulary


def synthetic_vocabulary(size: int) -> Vocabulary:
    return Vocabulary(["w{}".format(i) for i in range(size - 4)])


def build_translation_model(vocab_src=32000, vocab_tgt=32000, emb=512, rnn=512, dec_rnn=None,
                            dec_emb=None, att_size=None, max_len=50, beam_size=5, max_steps=50,
                            length_normalization=0.6, l2_weight=1e-8, clip_norm=1.0,
                            with_trainer=True, supress_unk=False, device=None, seed=1234,
                            num_sessions=1) -> TranslationModel:
    reset_registry()
    src_vocab, tgt_vocab = synthetic_vocabulary(vocab_src), synthetic_vocabulary(vocab_tgt)
    encoder = SentenceEncoder(name="encoder", vocabulary=src_vocab, data_id="source",
                              embedding_size=emb, rnn_size=rnn, max_input_len=max_len)
    attention = Attention(name="attention", encoder=encoder, state_size=att_size)
    decoder = Decoder(encoders=[encoder], vocabulary=tgt_vocab, data_id="target", name="decoder",
                      max_output_len=max_len, embedding_size=dec_emb or emb, rnn_size=dec_rnn or rnn,
                      attentions=[attention], supress_unk=supress_unk)
    beam_decoder = beam_runner = None
    if beam_size:
        beam_decoder = BeamSearchDecoder(name="beam_decoder", parent_decoder=decoder, beam_size=beam_size,
                                         max_steps=max_steps, length_normalization=length_normalization)
        beam_runner = BeamSearchRunner(output_series="target_beam", decoder=beam_decoder, rank=1)
    greedy_runner = GreedyRunner(output_series="target", decoder=decoder)
    trainer = None
    if with_trainer:
        from .trainers import CrossEntropyTrainer
        trainer = CrossEntropyTrainer(decoders=[decoder], l2_weight=l2_weight, clip_norm=clip_norm)
    tf_manager = TensorFlowManager(num_sessions=num_sessions, num_threads=4, device=device, seed=seed)
    tf_manager.initialize_sessions()
    return TranslationModel(encoder, attention, decoder, beam_decoder, greedy_runner, beam_runner,
                            trainer, tf_manager, src_vocab, tgt_vocab)


def synthetic_ids(seed=1234, batch=128, src_len=50, tgt_len=50, vocab=32000, ragged=False):
    """ids uniform in [4,V); </s> closes every target (added by the decoder's
    feed_dict, so target sentences here hold len-1 tokens)."""
    rng = np.random.default_rng(seed)
    src = rng.integers(4, vocab, size=(batch, src_len)).astype(np.int32)
    tgt = rng.integers(4, vocab, size=(batch, tgt_len)).astype(np.int32)
    if ragged:
        sl = rng.integers(max(1, src_len // 2), src_len + 1, size=batch)
        tl = rng.integers(max(1, tgt_len // 2), tgt_len + 1, size=batch)
    else:
        sl = np.full(batch, src_len)
        tl = np.full(batch, tgt_len)
    src_sents = [src[b, :sl[b]] for b in range(batch)]
    tgt_sents = [tgt[b, :tl[b] - 1] for b in range(batch)]
    return src_sents, tgt_sents


def synthetic_dataset(seed=1234, batch=128, src_len=50, tgt_len=50, vocab=32000, ragged=False,
                      with_target=True) -> Dataset:
    src, tgt = synthetic_ids(seed, batch, src_len, tgt_len, vocab, ragged)
    series = {"source": src}
    if with_target:
        series["target"] = tgt
    return Dataset("synthetic", series, BatchingScheme(batch_size=batch))


def load_baseline_weights(store, seed: int = 1234, std: float = 0.05) -> None:
    """Random-init weights of BASELINE.md section 3 for the benchmark model: dense / embedding matrices and
    the attention vector ~ N(0, std); the recurrent H x H blocks of the GRU kernels orthogonal
    (tf.orthogonal_initializer, nn/ortho_gru_cell.py:20-41) with N(0, std) input rows; GRU gate bias 1,
    every other bias 0; LayerNorm gamma 1, beta 0.  (The model parts' own default initialisers are
    N(0, 0.001)-class and make attention and beam search degenerate.)"""
    from .variables import orthogonal_initializer
    rng = np.random.default_rng(seed)
    ortho = orthogonal_initializer()
    values = {}
    for name in store.names():
        shape = tuple(store[name].shape)
        leaf = name.rsplit("/", 1)[-1]
        if "OrthoGRUCell" in name and leaf == "kernel":
            rows, cols = shape
            hsz = cols // 2 if "/gates/" in name else cols
            top = (rng.standard_normal((rows - hsz, cols)) * std).astype(np.float32)
            rec = np.concatenate([ortho(rng, (hsz, hsz)) for _ in range(cols // hsz)], 1)
            values[name] = np.concatenate([top, rec], 0)
        elif "OrthoGRUCell" in name and "/gates/" in name and leaf == "bias":
            values[name] = np.ones(shape, np.float32)
        elif leaf == "gamma":
            values[name] = np.ones(shape, np.float32)
        elif len(shape) >= 2 or leaf == "attn_similarity_v":
            values[name] = (rng.standard_normal(shape) * std).astype(np.float32)
        else:
            values[name] = np.zeros(shape, np.float32)
    store.load_state_dict(values)


class TransformerModel(NamedTuple):
    input_sequence: object
    encoder: object
    decoder: object
    beam_decoder: Optional[BeamSearchDecoder]
    greedy_runner: GreedyRunner
    beam_runner: Optional[BeamSearchRunner]
    trainer: object
    tf_manager: TensorFlowManager
    vocab: Vocabulary


def build_transformer_model(vocab=32000, dim=512, ff=2048, depth=6, heads=8, max_len=50, beam_size=5, max_steps=50,
                            length_normalization=0.6, l2_weight=1e-8, clip_norm=1.0, with_trainer=True, device=None,
                            seed=1234) -> TransformerModel:
    """BASELINE configs[4] (tests/transformer.ini topology at the Transformer-base size): EmbeddedSequence ->
    TransformerEncoder -> TransformerDecoder with tied embeddings, shared source / target vocabulary."""
    from .decoders import TransformerDecoder
    from .encoders import TransformerEncoder
    from .model.sequence import EmbeddedSequence
    reset_registry()
    voc = synthetic_vocabulary(vocab)
    seq = EmbeddedSequence(name="encoder_input", vocabulary=voc, data_id="source", embedding_size=dim,
                           max_length=max_len, scale_embeddings_by_depth=True)
    enc = TransformerEncoder(name="encoder", input_sequence=seq, ff_hidden_size=ff, depth=depth, n_heads=heads)
    dec = TransformerDecoder(name="decoder", encoders=[enc], vocabulary=voc, data_id="target", ff_hidden_size=ff,
                             n_heads_self=heads, n_heads_enc=heads, depth=depth, max_output_len=max_len,
                             embedding_size=dim)
    bdec = brun = None
    if beam_size:
        bdec = BeamSearchDecoder(name="beam_decoder", parent_decoder=dec, beam_size=beam_size, max_steps=max_steps,
                                 length_normalization=length_normalization)
        brun = BeamSearchRunner(output_series="target_beam", decoder=bdec, rank=1)
    grun = GreedyRunner(output_series="target", decoder=dec)
    trainer = None
    if with_trainer:
        from .trainers import CrossEntropyTrainer
        trainer = CrossEntropyTrainer(decoders=[dec], l2_weight=l2_weight, clip_norm=clip_norm)
    tfm = TensorFlowManager(num_sessions=1, num_threads=4, device=device, seed=seed)
    tfm.initialize_sessions()
    return TransformerModel(seq, enc, dec, bdec, grun, brun, trainer, tfm, voc)


def transformer_train_flops(batch, length, vocab=32000, dim=512, ff=2048, depth=6) -> float:
    """Multiply-add flops (x2) of one training step of the model above, forward + backward = 3 x forward: per layer
    the four attention projections (8 N d^2), the attention core (4 N T d), the feed-forward block (4 N d ff); the
    decoder has two attention blocks per layer; the tied vocabulary projection 2 N d V."""
    n = float(batch * length)
    att = 8 * n * dim * dim + 4 * n * length * dim
    ffn = 4 * n * dim * ff
    fwd = depth * (att + ffn) + depth * (2 * att + ffn) + 2 * n * dim * vocab
    return 3.0 * fwd


def transformer_decode_step_flops(rows, vocab=32000, dim=512, ff=2048, depth=6, cached_len=25, src_len=50) -> float:
    """Multiply-add flops (x2) of ONE cached decoding step of the same model for ``rows`` rows (sentences, or
    hypotheses of a beam): per decoder layer the self-attention's q / k / v / output projections (8 d^2), the
    cross-attention's query and output projections over cached encoder keys / values (4 d^2), the two attention cores
    over ``cached_len`` own and ``src_len`` encoder positions (4 d each per position), the feed-forward block
    (4 d ff); the tied vocabulary projection 2 d V.  38.4 M multiply-adds per row at the base size."""
    per_row = depth * (12 * dim * dim + 4 * dim * (cached_len + src_len) + 4 * dim * ff) + 2 * dim * vocab
    return float(rows) * per_row


def translation_train_flops(batch, length, vocab=32000, dim=512, att=None) -> float:
    """Multiply-add flops (x2) of the DENSE products of one training step of ``build_translation_model`` (emb = rnn =
    ``dim``, attention state size ``att`` = 2 dim by default), forward + backward = 3 x forward (every product has an
    input gradient and a weight gradient of its own size).  Per source position: both directions' gate and candidate
    kernels ([emb + rnn] x 3 rnn each) and the attention keys (2 rnn x att); per target position: the decoder cell
    ([emb + rnn] x 3 rnn), the attention query (rnn x att), the output projection ([rnn + emb + 2 rnn] x rnn) and the
    logits (rnn x vocab).  The Bahdanau energies / contexts (2 x att and 2 x 2 rnn flops per position pair) are
    elementwise-and-reduce work on the vector units and are NOT counted."""
    att = 2 * dim if att is None else att
    n = float(batch * length)
    src = 2 * (2 * dim) * (3 * dim) + (2 * dim) * att
    tgt = (2 * dim) * (3 * dim) + dim * att + (4 * dim) * dim + dim * vocab
    return 3.0 * 2.0 * n * (src + tgt)


class CaptioningModel(NamedTuple):
    encoder: object
    attention: Attention
    decoder: Decoder
    beam_decoder: Optional[BeamSearchDecoder]
    greedy_runner: GreedyRunner
    beam_runner: Optional[BeamSearchRunner]
    trainer: object
    tf_manager: TensorFlowManager
    vocab: Vocabulary


def build_captioning_model(vocab=32000, shape=(8, 8, 2048), att_size=512, emb=512, rnn=512, max_len=50, beam_size=5,
                           max_steps=50, length_normalization=0.6, l2_weight=1e-8, clip_norm=1.0, with_trainer=True,
                           device=None, seed=1234) -> CaptioningModel:
    """BASELINE configs[3] (tests/captioning.ini topology): pre-extracted convolutional maps -> SpatialFiller ->
    Bahdanau attention over the 64 positions -> attention GRU decoder."""
    from .encoders import SpatialFiller
    reset_registry()
    voc = synthetic_vocabulary(vocab)
    enc = SpatialFiller(name="image_encoder", input_shape=list(shape), data_id="images")
    att = Attention(name="attention", encoder=enc, state_size=att_size)
    dec = Decoder(encoders=[enc], vocabulary=voc, data_id="target", name="decoder", max_output_len=max_len,
                  embedding_size=emb, rnn_size=rnn, attentions=[att])
    bdec = brun = None
    if beam_size:
        bdec = BeamSearchDecoder(name="beam_decoder", parent_decoder=dec, beam_size=beam_size, max_steps=max_steps,
                                 length_normalization=length_normalization)
        brun = BeamSearchRunner(output_series="target_beam", decoder=bdec, rank=1)
    grun = GreedyRunner(output_series="target", decoder=dec)
    trainer = None
    if with_trainer:
        from .trainers import CrossEntropyTrainer
        trainer = CrossEntropyTrainer(decoders=[dec], l2_weight=l2_weight, clip_norm=clip_norm)
    tfm = TensorFlowManager(num_sessions=1, num_threads=4, device=device, seed=seed)
    tfm.initialize_sessions()
    return CaptioningModel(enc, att, dec, bdec, grun, brun, trainer, tfm, voc)


def synthetic_captioning_dataset(seed=1234, batch=128, shape=(8, 8, 2048), tgt_len=50, vocab=32000,
                                 with_target=True) -> Dataset:
    """Maps N(0,1) clipped at zero; targets as in ``synthetic_ids``."""
    rng = np.random.default_rng(seed)
    maps = np.maximum(rng.standard_normal((batch,) + tuple(shape), dtype=np.float32), 0.0)
    series = {"images": list(maps)}
    if with_target:
        tgt = rng.integers(4, vocab, size=(batch, tgt_len - 1)).astype(np.int32)
        series["target"] = list(tgt)
    return Dataset("synthetic_captions", series, BatchingScheme(batch_size=batch))
