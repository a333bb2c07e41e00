"""An INDEPENDENT writer (and reader) of TensorFlow tensor bundles ("checkpoint V2"), written from the
published on-disk format of TensorFlow 1.12 -- it shares no code with neuralmonkey_amd/tf_bundle.py and
imports nothing from the package:

    python tests/golden/make_tf_bundle_fixture.py      -> tests/golden/tf_bundle/{big,small}_blocks.*

What tf.train.Saver (tf_manager.py:274-277 of the reference) leaves on disk, restated:

  <prefix>.data-00000-of-00001   the tensors' little-endian bytes back to back, in key order
  <prefix>.index                 a leveldb-format table (tensorflow/core/lib/io/table_builder.cc):
      data blocks     entries "varint shared | varint non_shared | varint value_len | key suffix | value",
                      a restart point (shared = 0) every 16 entries, then the uint32 restart offsets and
                      their count; a block is closed once its estimated size reaches block_size
                      (262144 in TF's table::Options)
      block trailer   1 byte compression type (0 = none) + uint32 masked CRC-32C of block + type
                      (mask: rotate right by 15, add 0xa282ead8)
      metaindex block (empty), index block (one entry per data block: a SHORTENED separator key >= the
                      block's last key and < the next block's first key -- BytewiseComparator::
                      FindShortestSeparator / FindShortSuccessor -- and the BlockHandle "varint offset |
                      varint size"; restart interval 1)
      footer          metaindex handle, index handle, zero padding to 40 bytes, magic 0xdb4775248b80fb57 (LE)
      key ""          BundleHeaderProto  {1: num_shards = 1, 2: endianness (LITTLE = 0, omitted),
                                          3: VersionDef {1: producer = 1}}
      key <name>      BundleEntryProto   {1: dtype, 2: TensorShapeProto {2: Dim {1: size}}, 3: shard_id,
                                          4: offset, 5: size, 6: fixed32 masked CRC-32C of the tensor bytes};
                      proto3 omits zero scalars (shard_id, offset of the first tensor)

Two fixtures: TF's own block size (one data block, several restart groups) and a 512-byte block size (many
data blocks, shortened separator keys) -- the structure a large real checkpoint has.  tests/test_tf_bundle.py
reads both with the product reader and parses the product WRITER's output with ``parse_table`` below.
"""
import os
import struct

import numpy as np

MAGIC = 0xDB4775248B80FB57


# ---- CRC-32C (Castagnoli), reflected polynomial 0x82f63b78, table driven --------------------------------
def _make_table():
    table = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)
    return table


_TABLE = _make_table()


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for byte in data:
        c = _TABLE[(c ^ byte) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


assert crc32c(b"123456789") == 0xE3069283                # the standard check value of CRC-32C


def varint(n: int) -> bytes:
    out = bytearray()
    while True:
        low = n & 0x7F
        n >>= 7
        if n:
            out.append(low | 0x80)
        else:
            out.append(low)
            return bytes(out)


# ---- protobufs ------------------------------------------------------------------------------------------
DTYPES = {"float32": 1, "float64": 2, "int32": 3, "int64": 9}      # tensorflow/core/framework/types.proto


def header_proto() -> bytes:
    version = b"\x08" + varint(1)                                   # VersionDef.producer = 1
    return b"\x08" + varint(1) + b"\x1a" + varint(len(version)) + version


def entry_proto(dtype: int, shape, offset: int, size: int, crc_masked: int) -> bytes:
    dims = b""
    for extent in shape:
        dim = b"\x08" + varint(extent)
        dims += b"\x12" + varint(len(dim)) + dim
    msg = b"\x08" + varint(dtype) + b"\x12" + varint(len(dims)) + dims
    if offset:
        msg += b"\x20" + varint(offset)
    if size:
        msg += b"\x28" + varint(size)
    return msg + b"\x35" + struct.pack("<I", crc_masked)


# ---- leveldb table builder ------------------------------------------------------------------------------
def shortest_separator(start: bytes, limit: bytes) -> bytes:
    n = 0
    while n < min(len(start), len(limit)) and start[n] == limit[n]:
        n += 1
    if n < min(len(start), len(limit)) and start[n] < 0xFF and start[n] + 1 < limit[n]:
        return start[:n] + bytes([start[n] + 1])
    return start


def short_successor(key: bytes) -> bytes:
    for i, byte in enumerate(key):
        if byte != 0xFF:
            return key[:i] + bytes([byte + 1])
    return key


class BlockBuilder:
    def __init__(self, restart_interval: int):
        self.interval, self.buf, self.restarts, self.count, self.last = restart_interval, bytearray(), [0], 0, b""

    def add(self, key: bytes, value: bytes) -> None:
        shared = 0
        if self.count < self.interval:
            while shared < min(len(self.last), len(key)) and self.last[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def estimate(self) -> int:
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + \
            struct.pack("<I", len(self.restarts))

    @property
    def empty(self) -> bool:
        return not self.buf


def build_table(items, block_size: int) -> bytes:
    out = bytearray()
    index = BlockBuilder(1)
    data = BlockBuilder(16)
    pending = None                       # (last key of the closed block, its handle)

    def write_block(raw: bytes) -> bytes:
        handle = varint(len(out)) + varint(len(raw))
        out.extend(raw + b"\x00" + struct.pack("<I", masked(crc32c(raw + b"\x00"))))
        return handle
    last_key = b""
    for key, value in sorted(items):
        if pending is not None:
            index.add(shortest_separator(pending[0], key), pending[1])
            pending = None
        data.add(key, value)
        last_key = key
        if data.estimate() >= block_size:
            pending = (last_key, write_block(data.finish()))
            data = BlockBuilder(16)
    if not data.empty:
        pending = (last_key, write_block(data.finish()))
    if pending is not None:
        index.add(short_successor(pending[0]), pending[1])
    meta_handle = write_block(BlockBuilder(16).finish())
    index_handle = write_block(index.finish())
    footer = meta_handle + index_handle
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
    return bytes(out)


# ---- an equally independent reader (checks the PRODUCT writer's output) -----------------------------------
def _get_varint(buf, pos):
    shift = value = 0
    while True:
        byte = buf[pos]
        pos += 1
        value |= (byte & 0x7F) << shift
        if not byte & 0x80:
            return value, pos
        shift += 7


def _entries(table: bytes, offset: int, size: int):
    raw = table[offset:offset + size]
    assert table[offset + size] == 0, "compressed block"
    stored = struct.unpack("<I", table[offset + size + 1:offset + size + 5])[0]
    assert stored == masked(crc32c(raw + b"\x00")), "block checksum"
    count = struct.unpack("<I", raw[-4:])[0]
    end, pos, key = len(raw) - 4 - 4 * count, 0, b""
    while pos < end:
        shared, pos = _get_varint(raw, pos)
        rest, pos = _get_varint(raw, pos)
        vlen, pos = _get_varint(raw, pos)
        key = key[:shared] + raw[pos:pos + rest]
        pos += rest
        yield key, raw[pos:pos + vlen]
        pos += vlen


def parse_table(table: bytes):
    assert struct.unpack("<Q", table[-8:])[0] == MAGIC
    footer = table[-48:]
    _, pos = _get_varint(footer, 0)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    items, prev_sep = [], None
    for sep, handle in _entries(table, ioff, isize):
        boff, p = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, p)
        block = list(_entries(table, boff, bsize))
        assert block and block[-1][0] <= sep and (prev_sep is None or block[0][0] > prev_sep)
        prev_sep = sep
        items.extend(block)
    assert [k for k, _ in items] == sorted(k for k, _ in items)
    return items


def parse_bundle(prefix: str):
    """{name: array} of a bundle, by this module's own decoding (field numbers as in the docstring)."""
    with open(prefix + ".index", "rb") as fh:
        items = parse_table(fh.read())
    with open(prefix + ".data-00000-of-00001", "rb") as fh:
        blob = fh.read()
    codes = {v: k for k, v in DTYPES.items()}
    out = {}
    for key, value in items:
        if key == b"":
            assert value == header_proto()
            continue
        pos, fields = 0, {}
        while pos < len(value):
            tag, pos = _get_varint(value, pos)
            if tag & 7 == 0:
                fields[tag >> 3], pos = _get_varint(value, pos)
            elif tag & 7 == 2:
                n, pos = _get_varint(value, pos)
                fields[tag >> 3], pos = value[pos:pos + n], pos + n
            else:
                assert tag & 7 == 5
                fields[tag >> 3], pos = struct.unpack("<I", value[pos:pos + 4])[0], pos + 4
        shape, spos, sbuf = [], 0, fields.get(2, b"")
        while spos < len(sbuf):
            tag, spos = _get_varint(sbuf, spos)
            n, spos = _get_varint(sbuf, spos)
            dim = sbuf[spos:spos + n]
            spos += n
            assert tag == 0x12 and (dim[:1] == b"\x08" or dim == b"")
            shape.append(_get_varint(dim, 1)[0] if dim else 0)
        raw = blob[fields.get(4, 0):fields.get(4, 0) + fields.get(5, 0)]
        assert fields[6] == masked(crc32c(raw)), key
        out[key.decode()] = np.frombuffer(raw, dtype=np.dtype(codes[fields[1]]).newbyteorder("<")).reshape(shape)
    return out


# ---- the fixture ---------------------------------------------------------------------------------------
def fixture_tensors():
    """Variables of a small Neural Monkey model under their TensorFlow names and TF shapes, with Adam slots
    and the optimizer scalars a tf.train.Saver over all global variables writes."""
    rng = np.random.default_rng(20260925)
    e, h, v = 4, 4, 11                    # (the default output projection needs embedding size == rnn size)
    c = 2 * h
    shapes = {
        "encoder_input/embedding_matrix_0": (v, e),
        "encoder/rnn_0_bidirectional/bidirectional_rnn/fw/OrthoGRUCell/gates/kernel": (e + h, 2 * h),
        "encoder/rnn_0_bidirectional/bidirectional_rnn/fw/OrthoGRUCell/gates/bias": (2 * h,),
        "encoder/rnn_0_bidirectional/bidirectional_rnn/fw/OrthoGRUCell/candidate/kernel": (e + h, h),
        "encoder/rnn_0_bidirectional/bidirectional_rnn/fw/OrthoGRUCell/candidate/bias": (h,),
        "encoder/rnn_0_bidirectional/bidirectional_rnn/bw/OrthoGRUCell/gates/kernel": (e + h, 2 * h),
        "encoder/rnn_0_bidirectional/bidirectional_rnn/bw/OrthoGRUCell/gates/bias": (2 * h,),
        "encoder/rnn_0_bidirectional/bidirectional_rnn/bw/OrthoGRUCell/candidate/kernel": (e + h, h),
        "encoder/rnn_0_bidirectional/bidirectional_rnn/bw/OrthoGRUCell/candidate/bias": (h,),
        "encoder/LayerNorm/gamma": (c,),
        "encoder/LayerNorm/beta": (c,),
        "attention/attn_key_projection": (1, 1, c, c),               # a 1x1 convolution filter in TF
        "attention/Attention/attn_query_projection": (h, c),
        "attention/attn_projection_bias": (c,),
        "attention/attn_similarity_v": (c,),
        "attention/attn_bias": (),                                   # a scalar in TF
        "decoder/word_embeddings": (v, e),
        "decoder/initial_state/encoders_projection/kernel": (c, h),
        "decoder/initial_state/encoders_projection/bias": (h,),
        "decoder/attention_decoder/OrthoGRUCell/gates/kernel": (e + h, 2 * h),
        "decoder/attention_decoder/OrthoGRUCell/gates/bias": (2 * h,),
        "decoder/attention_decoder/OrthoGRUCell/candidate/kernel": (e + h, h),
        "decoder/attention_decoder/OrthoGRUCell/candidate/bias": (h,),
        "decoder/attention_decoder/dense/kernel": (h + e + c, e),
        "decoder/attention_decoder/dense/bias": (e,),
        "decoder/state_to_word_W": (e, v),
        "decoder/state_to_word_b": (v,),
    }
    tensors = {}
    for name, shape in shapes.items():
        tensors[name] = rng.standard_normal(shape).astype(np.float32)
        tensors[name + "/Adam"] = (rng.standard_normal(shape) * 0.01).astype(np.float32)
        tensors[name + "/Adam_1"] = (rng.random(shape) * 1e-4).astype(np.float32)
    tensors["beta1_power"] = np.float32(0.9 ** 8)
    tensors["beta2_power"] = np.float32(0.999 ** 8)
    tensors["global_step"] = np.int64(7)
    return tensors


def write_fixture(prefix: str, block_size: int) -> None:
    tensors = fixture_tensors()
    items, blob = [(b"", header_proto())], bytearray()
    for name in sorted(tensors):
        arr = np.asarray(tensors[name])
        raw = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
        items.append((name.encode(), entry_proto(DTYPES[arr.dtype.name], arr.shape, len(blob), len(raw),
                                                 masked(crc32c(raw)))))
        blob += raw
    with open(prefix + ".index", "wb") as fh:
        fh.write(build_table(items, block_size))
    with open(prefix + ".data-00000-of-00001", "wb") as fh:
        fh.write(bytes(blob))


if __name__ == "__main__":
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf_bundle")
    os.makedirs(here, exist_ok=True)
    write_fixture(os.path.join(here, "big_blocks"), 262144)
    write_fixture(os.path.join(here, "small_blocks"), 512)
    for name in ("big_blocks", "small_blocks"):
        got = parse_bundle(os.path.join(here, name))
        assert set(got) == set(fixture_tensors())
        print(name, os.path.getsize(os.path.join(here, name + ".index")), "bytes of index,", len(got), "tensors")
