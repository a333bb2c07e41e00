"""TensorFlow tensor-bundle checkpoints (neuralmonkey_amd/tf_bundle.py): known answers of the
published building blocks (CRC-32C, leveldb CRC masking, varints, protobuf encodings, SSTable footer)
and write -> read round trips.  Host-only: runs in the CPU suite."""
import struct

import numpy as np
import pytest

from neuralmonkey_amd import tf_bundle as TB


def test_crc32c_known_answers():
    # RFC 3720 / leveldb crc32c_test.cc vectors
    assert TB.crc32c(b"123456789") == 0xE3069283
    assert TB.crc32c(bytes(32)) == 0x8A9136AA
    assert TB.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert TB.crc32c(bytes(range(32))) == 0x46DD794E
    assert TB.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    # chaining and ndarray input
    assert TB.crc32c(b"6789", TB.crc32c(b"12345")) == 0xE3069283
    arr = np.arange(1000, dtype=np.float32)
    assert TB.crc32c(arr) == TB.crc32c(arr.tobytes())
    # leveldb mask: unmask(mask(x)) == x, and masking changes the value
    crc = TB.crc32c(b"foo")
    assert TB.unmask_crc(TB.mask_crc(crc)) == crc and TB.mask_crc(crc) != crc
    assert TB.mask_crc(TB.mask_crc(crc)) != crc


def test_varint_and_protobuf_encodings():
    for value, enc in ((0, b"\x00"), (1, b"\x01"), (127, b"\x7f"), (128, b"\x80\x01"), (300, b"\xac\x02"),
                       (2 ** 32, b"\x80\x80\x80\x80\x10")):
        assert TB.put_varint(value) == enc
        assert TB.get_varint(enc + b"\xff", 0) == (value, len(enc))
    # BundleHeaderProto {num_shards: 1, version {producer: 1}}
    assert TB.HEADER == b"\x08\x01\x1a\x02\x08\x01"
    # BundleEntryProto {dtype: DT_FLOAT, shape {dim {size: 3} dim {size: 4}}, offset: 48, size: 48, crc32c}
    enc = TB.encode_entry(1, (3, 4), 48, 48, 0xDEADBEEF)
    assert enc == b"\x08\x01" + b"\x12\x08" + b"\x12\x02\x08\x03" + b"\x12\x02\x08\x04" + b"\x20\x30" + b"\x28\x30" \
        + b"\x35" + struct.pack("<I", 0xDEADBEEF)
    dec = TB.decode_entry(enc)
    assert (dec["dtype"], dec["shape"], dec["offset"], dec["size"], dec["crc32c"]) == (1, [3, 4], 48, 48, 0xDEADBEEF)
    scalar = TB.decode_entry(TB.encode_entry(9, (), 0, 8, 1))
    assert scalar["shape"] == [] and scalar["offset"] == 0 and scalar["dtype"] == 9


def test_sstable_round_trip_and_layout():
    rng = np.random.default_rng(0)
    items = [(b"", b"header")]
    for i in range(700):        # several 4 KB blocks, long shared prefixes
        items.append(("decoder/attention_decoder/layer_{:04d}/kernel".format(i).encode(),
                      bytes(rng.integers(0, 256, size=int(rng.integers(1, 60)), dtype=np.uint8))))
    table = TB.write_table(items)
    assert struct.unpack("<Q", table[-8:])[0] == 0xDB4775248B80FB57 and len(table[-48:]) == 48
    assert TB.read_table(table) == sorted(items)
    # a flipped payload byte is caught by the block checksum
    bad = bytearray(table)
    bad[100] ^= 0x40
    with pytest.raises(ValueError):
        TB.read_table(bytes(bad))
    with pytest.raises(ValueError):
        TB.read_table(table[:-1] + b"\x00")


def test_bundle_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {"encoder/rnn/kernel": rng.standard_normal((37, 12)).astype(np.float32),
               "attention/attn_key_projection": rng.standard_normal((1, 1, 8, 6)).astype(np.float32),
               "attention/attn_bias": np.float32(0.25),
               "global_step": np.int64(1234),
               "lengths": np.arange(7, dtype=np.int32)}
    prefix = str(tmp_path / "variables.data")
    TB.write_bundle(prefix, tensors)
    got = TB.read_bundle(prefix)
    assert sorted(got) == sorted(tensors)
    for name, arr in tensors.items():
        assert got[name].dtype == np.asarray(arr).dtype and got[name].shape == np.asarray(arr).shape
        assert np.array_equal(got[name], arr)
    # corrupt one tensor byte: the per-tensor CRC-32C catches it
    data_file = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data_file, "rb").read())
    raw[5] ^= 1
    open(data_file, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        TB.read_bundle(prefix)


def test_store_export_import(tmp_path):
    import torch
    from neuralmonkey_amd.variables import VariableStore, random_normal_initializer, zeros_initializer
    store = VariableStore("cpu", seed=3)
    store.declare("attention/attn_key_projection", (8, 6), random_normal_initializer(stddev=0.5))
    store.declare("attention/attn_bias", (1,), random_normal_initializer(stddev=0.5))
    store.declare("encoder/conv2d/kernel", (5, 4), random_normal_initializer(stddev=0.5))
    store.declare("decoder/state_to_word_b", (9,), zeros_initializer())
    # a variable of the reference's graph that nothing reads (GRUCell.build's kernel under NematusGRUCell)
    store.declare_checkpoint_only("encoder/nematus_gru_cell/gates/kernel", (3, 4), random_normal_initializer(stddev=0.5))
    store.finalize()
    assert "encoder/nematus_gru_cell/gates/kernel" not in store.names() and store.total == 48 + 4 + 20 + 12
    m, v = store.ensure_adam()
    m.copy_(torch.arange(store.total, dtype=torch.float32))
    v.copy_(torch.arange(store.total, dtype=torch.float32) * 2)
    prefix = str(tmp_path / "variables.data")
    TB.export_store(store, prefix, global_step=17, with_adam=True)
    raw = TB.read_bundle(prefix)
    # TensorFlow's shapes on disk: tf.layers.conv2d filters [1,1,in,out], scalar attention bias; the attention's key
    # projection is a matrix (expanded in the graph: attention/feed_forward.py:77-82) -- the shapes the reference
    # itself creates, tests/test_reference_inis.py
    assert raw["attention/attn_key_projection"].shape == (8, 6)
    assert raw["encoder/conv2d/kernel"].shape == (1, 1, 5, 4)
    assert raw["attention/attn_bias"].shape == ()
    assert int(raw["global_step"]) == 17 and "decoder/state_to_word_b/Adam_1" in raw
    other = VariableStore("cpu", seed=99)
    for name, spec in store.specs.items():
        other.declare(name, spec.shape, zeros_initializer())
    other.finalize()
    # the unread variable travels with its (zero) Adam slots; a store that does not know it reports it unused
    extra = raw["encoder/nematus_gru_cell/gates/kernel"]
    assert extra.shape == (3, 4) and np.abs(extra).max() > 0 and not raw["encoder/nematus_gru_cell/gates/kernel/Adam"].any()
    info = TB.import_store(other, prefix)
    assert info == {"missing": [], "global_step": 17, "unused": [
        "encoder/nematus_gru_cell/gates/kernel" + s for s in ("", "/Adam", "/Adam_1")]}
    knows = VariableStore("cpu", seed=1)
    for name, spec in store.specs.items():
        knows.declare(name, spec.shape, zeros_initializer())
    knows.declare_checkpoint_only("encoder/nematus_gru_cell/gates/kernel", (3, 4), zeros_initializer())
    knows.finalize()
    assert TB.import_store(knows, prefix)["unused"] == []
    assert np.array_equal(knows.checkpoint_only_values()["encoder/nematus_gru_cell/gates/kernel"], extra)   # carried on
    for name in store.names():
        assert torch.equal(other[name], store[name])
    for spec in store.specs.values():          # (alignment padding between variables is not stored)
        sl = slice(spec.offset, spec.offset + spec.size)
        assert torch.equal(other.adam_m[sl], store.adam_m[sl]) and torch.equal(other.adam_v[sl], store.adam_v[sl])
    # VariableStore.load picks the bundle up by its prefix; save(fmt="tf") writes one
    third = VariableStore("cpu", seed=5)
    for name, spec in store.specs.items():
        third.declare(name, spec.shape, zeros_initializer())
    third.finalize()
    third.load(prefix)
    assert all(torch.equal(third[n], store[n]) for n in store.names())
    third.save(str(tmp_path / "again"), fmt="tf")
    again = TB.read_bundle(str(tmp_path / "again"))         # the restored Adam slots travel on (Saver: all globals)
    assert set(store.names()) <= set(again) and "decoder/state_to_word_b/Adam" in again and "global_step" not in again
    # a variable the checkpoint lacks
    third2 = VariableStore("cpu", seed=5)
    third2.declare("new/variable", (3,), zeros_initializer())
    third2.finalize()
    with pytest.raises(KeyError):
        TB.import_store(third2, prefix)
    assert TB.import_store(third2, prefix, strict=False)["missing"] == ["new/variable"]


# ---------------------------------------------------------------------------------------------------------
# cross-check against an INDEPENDENT implementation of the format (tests/golden/make_tf_bundle_fixture.py:
# written from the TensorFlow 1.12 on-disk format, no code shared with neuralmonkey_amd/tf_bundle.py)
# ---------------------------------------------------------------------------------------------------------
def _indep():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_tf_bundle_fixture.py")
    spec = importlib.util.spec_from_file_location("make_tf_bundle_fixture", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, os.path.join(os.path.dirname(path), "tf_bundle")


@pytest.mark.parametrize("which", ["big_blocks", "small_blocks"])
def test_reader_reads_the_independently_written_bundle(which):
    """The committed fixtures (one data block with TF's 256 KB block size; many 512-byte blocks with
    shortened separator keys in the index block) decode to exactly the tensors their generator wrote."""
    import os
    from neuralmonkey_amd import tf_bundle
    mod, root = _indep()
    want = mod.fixture_tensors()
    got = tf_bundle.read_bundle(os.path.join(root, which))
    assert set(got) == set(want)
    for name, arr in want.items():
        assert got[name].dtype == np.asarray(arr).dtype and got[name].shape == np.asarray(arr).shape, name
        assert np.array_equal(got[name], arr), name
    assert got["attention/attn_bias"].shape == () and got["attention/attn_key_projection"].shape == (1, 1, 8, 8)
    assert int(got["global_step"]) == 7


def test_the_committed_fixture_is_what_the_generator_writes(tmp_path):
    import os
    mod, root = _indep()
    for which, block in (("big_blocks", 262144), ("small_blocks", 512)):
        mod.write_fixture(str(tmp_path / which), block)
        for ext in (".index", ".data-00000-of-00001"):
            with open(os.path.join(root, which + ext), "rb") as a, open(str(tmp_path / which) + ext, "rb") as b:
                assert a.read() == b.read(), which + ext


def test_independent_reader_reads_the_product_writer(tmp_path):
    """... and the other way round: what tf_bundle.write_bundle writes (several blocks forced) is parsed by
    the independent reader, checksums, separator keys and protobuf field numbers included."""
    from neuralmonkey_amd import tf_bundle
    mod, _ = _indep()
    tensors = mod.fixture_tensors()
    prefix = str(tmp_path / "ours")
    tf_bundle.write_bundle(prefix, tensors)
    got = mod.parse_bundle(prefix)
    assert set(got) == set(tensors)
    for name, arr in tensors.items():
        assert np.array_equal(got[name], arr) and got[name].shape == np.asarray(arr).shape, name
    items = [(b"k%04d" % i, bytes([i % 251]) * (i % 37)) for i in range(400)]
    assert mod.parse_table(tf_bundle.write_table(items, block_bytes=300)) == sorted(items)
    assert tf_bundle.read_table(mod.build_table(items, 300)) == sorted(items)


def test_store_import_of_the_independent_bundle():
    """A checkpoint written the TensorFlow way restores a model of the same shapes: variables (TF conv-filter /
    scalar shapes folded), Adam slots and global_step."""
    import os
    from neuralmonkey_amd import synthetic, tf_bundle
    mod, root = _indep()
    model = synthetic.build_translation_model(vocab_src=11, vocab_tgt=11, emb=4, rnn=4, max_len=5, beam_size=0,
                                              with_trainer=False, device="cpu")
    store = model.tf_manager.sessions[0].store
    info = tf_bundle.import_store(store, os.path.join(root, "small_blocks"))
    want = mod.fixture_tensors()
    assert not info["missing"] and not info["unused"] and info["global_step"] == 7
    for name in store.names():
        assert np.array_equal(store[name].numpy().reshape(-1), want[name].reshape(-1)), name
    m, v = store.ensure_adam()
    for name, spec in store.specs.items():
        assert np.array_equal(m[spec.offset:spec.offset + spec.size].numpy(), want[name + "/Adam"].reshape(-1))
        assert np.array_equal(v[spec.offset:spec.offset + spec.size].numpy(), want[name + "/Adam_1"].reshape(-1))


@pytest.mark.parametrize("fmt", ["tf", "npz"])
def test_slots_travel_under_the_optimizers_name(tmp_path, fmt):
    """TensorFlow names an optimizer's slot variables after the optimizer: ``<var>/Adam`` and ``<var>/Adam_1``, or --
    tests/bpe.ini:102-108, ``name="adadelta"`` -- ``<var>/adadelta`` and ``<var>/adadelta_1`` (no beta powers there).  A
    store writes its two slots under ``slot_suffixes`` and a reader that has not been told the names (a fresh process
    restores before its trainer has taken a step) finds the pair every variable carries."""
    import torch
    from neuralmonkey_amd.optimizers import AdadeltaOptimizer
    from neuralmonkey_amd.variables import VariableStore, find_slot_suffixes, random_normal_initializer
    store = VariableStore("cpu", seed=3)
    store.declare("decoder/state_to_word_W", (4, 6), random_normal_initializer(stddev=0.5))
    store.declare("decoder/state_to_word_b", (6,), random_normal_initializer(stddev=0.5))
    store.finalize()
    accum, accum_update = store.ensure_adam()
    accum.copy_(torch.arange(store.total, dtype=torch.float32) + 1)
    accum_update.copy_(torch.arange(store.total, dtype=torch.float32) * 3 + 2)
    store.slot_suffixes = tuple(AdadeltaOptimizer(name="adadelta").slot_suffixes)
    path = str(tmp_path / ("variables.data" if fmt == "tf" else "variables.npz"))
    store.save(path, fmt=fmt, global_step=5)
    if fmt == "tf":
        keys = set(TB.read_bundle(path))
    else:
        with np.load(path) as data:
            keys = {k.replace("|", "/") for k in data.files}
    assert {"decoder/state_to_word_W/adadelta", "decoder/state_to_word_b/adadelta_1", "global_step"} <= keys
    assert "beta1_power" not in keys and not any(k.endswith(("/Adam", "/Adam_1")) for k in keys)
    fresh = VariableStore("cpu", seed=9)
    for name, spec in store.specs.items():
        fresh.declare(name, spec.shape, random_normal_initializer(stddev=0.5))
    fresh.finalize()
    assert fresh.slot_suffixes == ("/Adam", "/Adam_1")
    info = fresh.load(path)
    assert info["global_step"] == 5 and fresh.slot_suffixes == ("/adadelta", "/adadelta_1")
    assert torch.equal(fresh.theta, store.theta)
    for spec in store.specs.values():                 # (the flat buffers carry alignment padding no file holds)
        span = slice(spec.offset, spec.offset + spec.size)
        assert torch.equal(fresh.adam_m[span], accum[span]) and torch.equal(fresh.adam_v[span], accum_update[span])
    # the helper on its own: the preferred pair when present, else the pair all variables carry, else the preferred
    names = ["a/w", "b"]
    assert find_slot_suffixes({"a/w", "b", "a/w/Adam", "a/w/Adam_1", "b/Adam", "b/Adam_1"}, names,
                              ("/Adam", "/Adam_1")) == ("/Adam", "/Adam_1")
    assert find_slot_suffixes({"a/w", "b", "a/w/opt", "a/w/opt_1", "b/opt", "b/opt_1"}, names,
                              ("/Adam", "/Adam_1")) == ("/opt", "/opt_1")
    assert find_slot_suffixes({"a/w", "b", "a/w/opt", "a/w/opt_1"}, names, ("/Adam", "/Adam_1")) == ("/Adam", "/Adam_1")

