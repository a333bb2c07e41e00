"""Vocabulary file formats beyond the word list (vocabulary.py:102-173 of the reference):
tensor2tensor subword vocabularies and Nematus JSON dictionaries.  Host-side only."""
import json

from neuralmonkey_amd.vocabulary import (PAD_TOKEN_INDEX, UNK_TOKEN_INDEX, from_nematus_json,
                                         from_t2t_vocabulary)


def test_t2t_vocabulary_strips_quotes_and_reserved_entries(tmp_path):
    path = tmp_path / "vocab.t2t"
    path.write_text("'<pad>'\n'<EOS>'\n'the_'\n\"it's_\"\nplain\n'a'\n", encoding="utf-8")
    vocab = from_t2t_vocabulary(str(path))
    assert vocab.index_to_word[:4] == ["<pad>", "<s>", "</s>", "<unk>"]
    assert vocab.index_to_word[4:] == ["the_", "it's_", "plain", "a"]
    assert vocab.strings_to_indices([["the_", "nope", "<pad>"]]).tolist() == [[4, UNK_TOKEN_INDEX, PAD_TOKEN_INDEX]]


def test_nematus_json_order_truncation_and_padding(tmp_path):
    path = tmp_path / "vocab.json"
    path.write_text(json.dumps({"eos": 0, "UNK": 1, "dog": 3, "cat": 2, "emu": 5, "bee": 4}), encoding="utf-8")
    assert from_nematus_json(str(path)).index_to_word[4:] == ["cat", "dog", "bee", "emu"]
    assert from_nematus_json(str(path), max_size=2).index_to_word[4:] == ["cat", "dog"]
    padded = from_nematus_json(str(path), max_size=6, pad_to_max_size=True)
    # six real slots + the two Nematus slots that map onto </s> and <unk>: the matrix keeps its row count
    assert padded.index_to_word[4:] == ["cat", "dog", "bee", "emu", "<pad_0>", "<pad_1>", "<pad_2>", "<pad_3>"]
    assert len(from_nematus_json(str(path), pad_to_max_size=True)) == 4 + 4        # nothing to pad without max_size
