"""Drop-in acceptance: the model sections of the reference's
tests/flat-multiattention.ini and tests/hier-multiattention.ini, built from INI
text through the plugin surface (same class paths, argument names and object graph: several
decoders over two shared encoders, child attentions shared by several wrappers, RNN beam search with
``rank=2``), trained with dropout on the tape and decoded.  Data, evaluation and logging sections
belong to the control plane; the data here are synthetic files of the same formats (token lines,
``.npz`` feature maps named by a list file)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HEAD = """
[main]
name="{name}"
batch_size=1
epochs=1
train_dataset=<train_data>
trainer=<trainer>
runners=[{runners}]
[batching]
class=dataset.BatchingScheme
batch_size={batch}
[numpy_reader]
class=readers.numpy_reader.from_file_list
prefix="{root}"
shape=[8, 8, 20]
[train_data]
class=dataset.load
series=["source", "target", "images"]
data=["{root}/train.en", "{root}/train.de", ("{root}/train_images.txt", <numpy_reader>)]
batching=<batching>
[imagenet]
class=encoders.numpy_stateful_filler.SpatialFiller
input_shape=[8, 8, 20]
data_id="images"
[encoder_vocabulary]
class=vocabulary.from_wordlist
path="{root}/vocab.tsv"
[encoder]
class=encoders.recurrent.SentenceEncoder
name="sentence_encoder"
rnn_size=4
max_input_len=3
embedding_size=2
dropout_keep_prob=0.5
data_id="source"
vocabulary=<encoder_vocabulary>
[decoder_vocabulary]
class=vocabulary.from_wordlist
path="{root}/vocab.tsv"
"""

FLAT_VARIANTS = [("flat_noshare_nosentinel", "fnn", False, False), ("flat_share_nosentinel", "fsn", True, False),
                 ("flat_share_sentinel", "fss", True, True), ("flat_noshare_sentinel", "fns", False, True)]


def _flat_ini(root, batch):
    body, runners = "", []
    for name, short, share, sentinel in FLAT_VARIANTS:
        body += """
[{name}]
class=attention.combination.FlatMultiAttention
name="wrapper_{short}"
encoders=[<encoder>, <imagenet>]
attention_state_size=5
use_sentinels={sentinel}
share_attn_projections={share}
[decoder_{name}]
class=decoders.decoder.Decoder
name="decoder_{name}"
attentions=[<{name}>]
encoders=[<encoder>, <imagenet>]
rnn_size=3
embedding_size=3
dropout_keep_prob=0.5
data_id="target"
max_output_len=3
vocabulary=<decoder_vocabulary>
[runner_{name}]
class=runners.GreedyRunner
decoder=<decoder_{name}>
output_series="target_{name}"
""".format(name=name, short=short, share=share, sentinel=sentinel)
        runners.append("<runner_{}>".format(name))
    body += """
[beam_decoder_fss]
class=decoders.beam_search_decoder.BeamSearchDecoder
name="beam_search_decoder"
parent_decoder=<decoder_flat_share_sentinel>
beam_size=2
length_normalization=1.0
max_steps=3
[beam_runner_fss]
class=runners.BeamSearchRunner
decoder=<beam_decoder_fss>
output_series="target_beam"
rank=2
[trainer]
class=trainers.cross_entropy_trainer.CrossEntropyTrainer
decoders=[{decs}]
l2_weight=1.0e-8
clip_norm=1.0
optimizer=<optimizer>
[optimizer]
class=tf.train.AdamOptimizer
learning_rate=0.02
""".format(decs=",".join("<decoder_{}>".format(v[0]) for v in FLAT_VARIANTS))
    runners.append("<beam_runner_fss>")
    return HEAD.format(name="flat multi-attention", runners=", ".join(runners), root=root, batch=batch) + body


HIER_VARIANTS = [("hier_noshare_nosentinel", "hnn", False, False), ("hier_share_nosentinel", "hsn", True, False),
                 ("hier_share_sentinel", "hss", True, True), ("hier_noshare_sentinel", "hns", False, True)]


def _hier_ini(root, batch):
    body = """
[enc_attention]
class=attention.Attention
name="enc_attention"
state_size=3
encoder=<encoder>
[img_attention]
class=attention.Attention
name="img_attention"
state_size=2
encoder=<imagenet>
"""
    runners = []
    for name, short, share, sentinel in HIER_VARIANTS:
        body += """
[{name}]
class=attention.combination.HierarchicalMultiAttention
name="wrapper_{short}"
attentions=[<enc_attention>, <img_attention>]
attention_state_size=5
use_sentinels={sentinel}
share_attn_projections={share}
[decoder_{name}]
class=decoders.decoder.Decoder
name="decoder_{name}"
encoders=[<encoder>, <imagenet>]
attentions=[<{name}>]
rnn_size=3
embedding_size=3
dropout_keep_prob=0.5
data_id="target"
max_output_len=3
vocabulary=<decoder_vocabulary>
[runner_{name}]
class=runners.GreedyRunner
decoder=<decoder_{name}>
output_series="target_{name}"
""".format(name=name, short=short, share=share, sentinel=sentinel)
        runners.append("<runner_{}>".format(name))
    body += """
[trainer]
class=trainers.cross_entropy_trainer.CrossEntropyTrainer
decoders=[{decs}]
l2_weight=1.0e-8
clip_norm=1.0
optimizer=<optimizer>
[optimizer]
class=tf.train.AdamOptimizer
learning_rate=0.02
""".format(decs=",".join("<decoder_{}>".format(v[0]) for v in HIER_VARIANTS))
    return HEAD.format(name="hierarchical multi-attention", runners=", ".join(runners), root=root,
                       batch=batch) + body


def _write_data(root, n=6):
    rng = np.random.default_rng(0)
    words = ["a", "b", "c", "d", "x", "y", "z"]
    (root / "vocab.tsv").write_text("Word\tCount\n<pad>\t1\n<s>\t1\n</s>\t1\n<unk>\t1\n"
                                    + "".join("{}\t{}\n".format(w, 9 - i) for i, w in enumerate(words)))
    line = lambda: " ".join(rng.choice(words, size=int(rng.integers(1, 4))))
    (root / "train.en").write_text("".join(line() + "\n" for _ in range(n)))
    (root / "train.de").write_text("".join(line() + "\n" for _ in range(n)))
    names = []
    for i in range(n):
        np.savez(root / "img{}.npz".format(i), np.maximum(rng.standard_normal((8, 8, 20)), 0).astype(np.float32))
        names.append("img{}.npz".format(i))
    (root / "train_images.txt").write_text("\n".join(names) + "\n")


@pytest.mark.parametrize("kind,batch", [("flat", 1), ("flat", 3), ("hier", 2)])
def test_multiattention_ini_experiment(dev, tmp_path, kind, batch):
    from neuralmonkey_amd.config.configuration import load_experiment
    _write_data(tmp_path)
    path = tmp_path / "{}.ini".format(kind)
    path.write_text((_flat_ini if kind == "flat" else _hier_ini)(tmp_path, batch))
    model = load_experiment(str(path), device=str(dev), seed=1234)
    trainer = model.trainers[0]
    assert len(trainer.objectives) == 4
    store = model.tf_manager.sessions[0].store
    if kind == "flat":
        for n in ("wrapper_fns/context_projections/proj_matrix_1", "wrapper_fss/attn_v",
                  "decoder_flat_share_sentinel/attention_decoder/attention_wrapper_fss/sentinel/dense/kernel"):
            assert n in store, n
        assert "wrapper_fss/context_projections/proj_matrix_0" not in store        # shared projections
    else:
        for n in ("enc_attention/attn_key_projection", "wrapper_hns/attn_v",
                  "decoder_hier_noshare_sentinel/attention_decoder/attention_wrapper_hns/proj_attn_enc_attention/kernel",
                  "decoder_hier_noshare_sentinel/attention_decoder/attention_wrapper_hns/proj_sentinel/kernel",
                  "decoder_hier_share_sentinel/attention_decoder/attention_wrapper_hss/img_attention_logit/vector_bias"):
            assert n in store, n
    feedables = set.union(*[r.feedables for r in model.runners + model.trainers])
    batches = list(model.train_dataset.batches())
    assert len(batches[0]) == batch
    first, last = [], []
    for epoch in range(30):
        for b in batches:
            res = model.tf_manager.execute(b, feedables, model.trainers, train=True)[0]
            (first if epoch < 3 else last if epoch >= 27 else []).append(sum(
                v for k, v in res.losses.items() if k.endswith(" - cost")))
    assert np.isfinite(last).all() and np.mean(last) < np.mean(first) - 0.5, (np.mean(first), np.mean(last))
    outs = model.tf_manager.execute(batches[0], feedables, model.runners)
    assert len(outs) == len(model.runners)
    for out in outs:
        (series, sents), = out.outputs.items()
        assert len(sents) == batch and all(len(s) <= 3 for s in sents), series
