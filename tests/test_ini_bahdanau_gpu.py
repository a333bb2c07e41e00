"""Drop-in acceptance: the model, trainer and runner sections of the reference's
tests/bahdanau.ini -- SentenceEncoder (GRU 7: an odd size, so the taped path), Attention, Decoder with
``maxout_output(9)`` / ``supress_unk`` / dropout 0.5, a MultitaskTrainer over three CrossEntropyTrainers
PLUS a second trainer in the same ``trainer=[...]`` list (two optimizer updates per batch, each with
its own Adam slots), GreedyRunner, RepresentationRunner and a TensorRunner over three tensors --
built from INI text, trained on bucketed batches and run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INI = """
[main]
name="translation bahdanau style"
batch_size=16
epochs=2
train_dataset=<train_data>
trainer=[<mt_trainer>, <greedy_trainer>]
runners=[<runner>, <representation_runner>, <debug_runner>]
[batching]
class=dataset.BatchingScheme
bucket_boundaries=[2, 4]
bucket_batch_sizes=[4, 3, 2]
[train_data]
class=dataset.load
series=["source", "target"]
data=["{root}/train.en", "{root}/train.de"]
batching=<batching>
[encoder_vocabulary]
class=vocabulary.from_wordlist
path="{root}/vocab.tsv"
[encoder]
class=encoders.recurrent.SentenceEncoder
name="sentence_encoder"
rnn_size=7
max_input_len=10
embedding_size=11
data_id="source"
vocabulary=<encoder_vocabulary>
[attention]
class=attention.Attention
name="attention_sentence_encoder"
encoder=<encoder>
[decoder_vocabulary]
class=vocabulary.from_wordlist
path="{root}/vocab.tsv"
[decoder]
class=decoders.decoder.Decoder
name="bahdanau_decoder"
encoders=[<encoder>]
rnn_size=8
embedding_size=9
attentions=[<attention>]
output_projection=<dec_maxout_output>
dropout_keep_prob=0.5
data_id="target"
max_output_len=10
vocabulary=<decoder_vocabulary>
supress_unk=True
[dec_maxout_output]
class=decoders.output_projection.maxout_output
maxout_size=9
[fast_adam]
class=tf.train.AdamOptimizer
learning_rate=0.01
[trainer1]
class=trainers.cross_entropy_trainer.CrossEntropyTrainer
decoders=[<decoder>]
l2_weight=1.0e-8
clip_norm=1.0
optimizer=<fast_adam>
[trainer2]
class=trainers.cross_entropy_trainer.CrossEntropyTrainer
decoders=[<decoder>]
optimizer=<fast_adam>
[greedy_trainer]
class=trainers.cross_entropy_trainer.CrossEntropyTrainer
decoders=[<decoder>]
clip_norm=10
l1_weight=0.0001
optimizer=<fast_adam>
[mt_trainer]
class=trainers.multitask_trainer.MultitaskTrainer
trainers=[<trainer1>, <trainer1>, <trainer2>]
[runner]
class=runners.GreedyRunner
output_series="target"
decoder=<decoder>
[representation_runner]
class=runners.tensor_runner.RepresentationRunner
encoder=<encoder>
output_series="encoded"
[debug_runner]
class=runners.tensor_runner.TensorRunner
modelparts=[<encoder>, <encoder>, <decoder>]
tensors=["output", "temporal_states", "runtime_logits"]
batch_dims=[0, 0, 1]
tensors_by_name=[]
batch_dims_by_name=[]
output_series="debugtensors"
"""


def test_bahdanau_ini_experiment(dev, tmp_path):
    from neuralmonkey_amd.config.configuration import load_experiment
    rng = np.random.default_rng(0)
    words = ["a", "b", "c", "d", "x", "y", "z"]
    (tmp_path / "vocab.tsv").write_text("Word\tCount\n<pad>\t1\n<s>\t1\n</s>\t1\n<unk>\t1\n"
                                        + "".join("{}\t{}\n".format(w, 9 - i) for i, w in enumerate(words)))
    line = lambda: " ".join(rng.choice(words, size=int(rng.integers(1, 7))))
    (tmp_path / "train.en").write_text("".join(line() + "\n" for _ in range(14)))
    (tmp_path / "train.de").write_text("".join(line() + "\n" for _ in range(14)))
    path = tmp_path / "bahdanau.ini"
    path.write_text(INI.format(root=tmp_path))
    model = load_experiment(str(path), device=str(dev), seed=7)
    assert len(model.trainers) == 2 and len(model.runners) == 3
    mt, greedy_trainer = model.trainers
    dec = greedy_trainer.objectives[0].decoder
    assert dec.uses_general_path(True)                     # GRU 7 / rnn 8 with dropout: the taped path
    sess = model.tf_manager.sessions[0]
    feedables = set.union(*[r.feedables for r in model.runners + model.trainers])
    batches = list(model.train_dataset.batches())
    assert len({len(b) for b in batches}) > 1              # bucketed: several batch sizes
    first, last = [], []
    for epoch in range(12):
        for b in batches:
            res = model.tf_manager.execute(b, feedables, model.trainers, train=True)
            assert len(res) == 2 and all("bahdanau_decoder - cost" in r.losses for r in res)
            (first if epoch == 0 else last if epoch == 11 else []).append(res[0].losses["bahdanau_decoder - cost"])
    steps = 12 * len(batches)
    assert sess.global_step == 2 * steps                   # both trainers of the list update on every batch
    assert mt.trainer_idx == steps % 3
    # each optimizer keeps its own Adam slots and update count (TF: slots and beta powers per optimizer)
    states = [t._adam[id(sess.store)] for t in (mt.trainers[0], mt.trainers[2], greedy_trainer)]
    assert sum(s["applied"] for s in states[:2]) == steps and states[2]["applied"] == steps
    assert len({s["m"].data_ptr() for s in states}) == 3
    assert np.mean(last) < np.mean(first) - 0.3, (np.mean(first), np.mean(last))
    out = model.tf_manager.execute(batches[0], feedables, model.runners)
    n = len(batches[0])
    assert len(out[0].outputs["target"]) == n
    assert len(out[1].outputs["encoded"]) == n and out[1].outputs["encoded"][0].shape == (14,)
    row = out[2].outputs["debugtensors"][0]
    assert set(row) == {"sentence_encoder/output", "sentence_encoder/temporal_states",
                        "bahdanau_decoder/runtime_logits"}
    assert row["bahdanau_decoder/runtime_logits"].shape[1] == len(dec.vocabulary)
    assert np.all(row["bahdanau_decoder/runtime_logits"][:, 3] < -1e8)      # supress_unk
