"""TensorFlow tensor-bundle ("checkpoint V2") import / export for a ``VariableStore``.

The reference saves and restores through ``tf.train.Saver`` (tf_manager.py:227-288,
model/parameterized.py:101-125): ``<prefix>.index`` + ``<prefix>.data-00000-of-00001``.  Variables
here already carry the TF names, so real Neural Monkey checkpoints map onto the
flat parameter buffer name by name.  TensorFlow itself is not needed (and not installable here):
the two files are a leveldb-format sorted string table of hand-encoded protobufs plus raw
little-endian tensor bytes, restated below from the published format:

  index    SSTable: data blocks of prefix-compressed (key, value) entries with restart points, each
           followed by a 1-byte compression tag (0) and a masked CRC-32C; an (empty) meta-index block;
           an index block mapping the last key of each data block to its BlockHandle (offset, size);
           a 48-byte footer (two BlockHandles, padding, magic 0xdb4775248b80fb57).
           key ""   -> BundleHeaderProto {num_shards = 1, endianness = LITTLE, version {producer = 1}}
           key name -> BundleEntryProto {dtype, shape, shard_id = 0, offset, size, crc32c (masked)}
  data     the tensors back to back at the recorded offsets.

Validation status: CRC-32C and the varint / protobuf / SSTable encoders are checked against
published known answers and by write -> read round trips (tests/test_host.py); no TensorFlow-written
file exists in the reference repository to read, so interoperability with TF itself is untested.

Shape conventions that differ from the flat store: SpatialFiller's ``conv2d*/kernel`` are [1,1,in,out]
conv filters in TF ([in,out] here), ``attn_bias`` / ``attn_bias_<i>`` / ``vector_bias`` are scalars ([1] here),
``attn_v`` is [1,1,n] ([n] here) -- pinned to the variables the reference itself creates for its acceptance
configurations (tests/test_reference_inis.py).  Import accepts any TF shape with the same element order (extra unit dimensions);
export writes the TF shapes.  Adam slots (``<var>/Adam``, ``<var>/Adam_1``), ``beta1_power``,
``beta2_power`` and ``global_step`` are carried when present / requested.
"""
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

MAGIC = 0xDB4775248B80FB57
MASK_DELTA = 0xA282EAD8
DT = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8")}
DT_CODE = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9}


# -- checksums ---------------------------------------------------------------------------------------
def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C through libnmhip's host helper (a pure-Python loop over 200 MB would take minutes)."""
    import ctypes
    from . import _lib
    if isinstance(data, np.ndarray):
        arr = np.ascontiguousarray(data)
        return int(_lib.load().nm_crc32c(crc, arr.ctypes.data, arr.nbytes))
    data = bytes(data)
    ptr = ctypes.cast(ctypes.c_char_p(data), ctypes.c_void_p)          # no copy of the payload
    return int(_lib.load().nm_crc32c(crc, ptr, len(data)))


def mask_crc(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked: int) -> int:
    rot = (masked - MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# -- varints / protobuf wire format ------------------------------------------------------------------
def put_varint(value: int) -> bytes:
    out = bytearray()
    value &= (1 << 64) - 1
    while value >= 0x80:
        out.append((value & 0x7F) | 0x80)
        value >>= 7
    out.append(value)
    return bytes(out)


def get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        byte = buf[pos]
        pos += 1
        result |= (byte & 0x7F) << shift
        if byte < 0x80:
            return result, pos
        shift += 7


def _pb_fields(buf: bytes) -> List[Tuple[int, int, object]]:
    """[(field number, wire type, value)] of one message; unknown fields pass through."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = get_varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = get_varint(buf, pos)
        elif wire == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wire == 2:
            size, pos = get_varint(buf, pos)
            val, pos = buf[pos:pos + size], pos + size
        elif wire == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type {}".format(wire))
        out.append((field, wire, val))
    return out


def encode_entry(dtype_code: int, shape: Iterable[int], offset: int, size: int, crc_masked: int) -> bytes:
    """BundleEntryProto."""
    dims = b"".join(b"\x12" + put_varint(len(d)) + d for d in (b"\x08" + put_varint(s) for s in shape))
    msg = b"\x08" + put_varint(dtype_code) + b"\x12" + put_varint(len(dims)) + dims
    if offset:
        msg += b"\x20" + put_varint(offset)
    msg += b"\x28" + put_varint(size) + b"\x35" + struct.pack("<I", crc_masked)
    return msg


def decode_entry(buf: bytes) -> Dict[str, object]:
    entry = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None}
    for field, _, val in _pb_fields(buf):
        if field == 1:
            entry["dtype"] = val
        elif field == 2:
            for f2, _, v2 in _pb_fields(val):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = v3
                    entry["shape"].append(size)
        elif field == 3:
            entry["shard_id"] = val
        elif field == 4:
            entry["offset"] = val
        elif field == 5:
            entry["size"] = val
        elif field == 6:
            entry["crc32c"] = struct.unpack("<I", val)[0]
        elif field == 7:
            raise NotImplementedError("sliced (partitioned) variables are not supported")
    return entry


HEADER = b"\x08\x01\x1a\x02\x08\x01"      # num_shards = 1, (endianness LITTLE = default), version {producer = 1}


# -- SSTable -----------------------------------------------------------------------------------------
def _block(entries: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out, restarts, prev = bytearray(), [], b""
    for i, (key, value) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            limit = min(len(prev), len(key))
            while shared < limit and prev[shared] == key[shared]:
                shared += 1
        out += put_varint(shared) + put_varint(len(key) - shared) + put_varint(len(value))
        out += key[shared:] + value
        prev = key
    if not restarts:
        restarts = [0]
    out += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return bytes(out)


def _with_trailer(block: bytes) -> bytes:
    return block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00")))


def _handle(offset: int, size: int) -> bytes:
    return put_varint(offset) + put_varint(size)


def write_table(items: List[Tuple[bytes, bytes]], block_bytes: int = 4096) -> bytes:
    """Sorted (key, value) pairs -> SSTable bytes."""
    items = sorted(items)
    out, index_entries, pending = bytearray(), [], []
    size = 0

    def flush():
        nonlocal pending, size
        if not pending:
            return
        block = _block(pending)
        index_entries.append((pending[-1][0], _handle(len(out), len(block))))
        out.extend(_with_trailer(block))
        pending, size = [], 0
    for key, value in items:
        pending.append((key, value))
        size += len(key) + len(value) + 8
        if size >= block_bytes:
            flush()
    flush()
    meta = _block([])
    meta_handle = _handle(len(out), len(meta))
    out.extend(_with_trailer(meta))
    index = _block(index_entries, restart_interval=1)
    index_handle = _handle(len(out), len(index))
    out.extend(_with_trailer(index))
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    out.extend(footer)
    return bytes(out)


def _read_block(table: bytes, offset: int, size: int, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    block = table[offset:offset + size]
    kind = table[offset + size]
    if kind != 0:
        raise NotImplementedError("compressed SSTable blocks (type {}) are not supported".format(kind))
    if verify:
        stored = struct.unpack("<I", table[offset + size + 1:offset + size + 5])[0]
        if unmask_crc(stored) != crc32c(block + b"\x00"):
            raise ValueError("SSTable block checksum mismatch at offset {}".format(offset))
    nrestarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = get_varint(block, pos)
        non_shared, pos = get_varint(block, pos)
        vlen, pos = get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_table(table: bytes) -> List[Tuple[bytes, bytes]]:
    if len(table) < 48 or struct.unpack("<Q", table[-8:])[0] != MAGIC:
        raise ValueError("not an SSTable (bad magic)")
    footer = table[-48:]
    _, pos = get_varint(footer, 0)
    _, pos = get_varint(footer, pos)
    ioff, pos = get_varint(footer, pos)
    isize, pos = get_varint(footer, pos)
    items = []
    for _, handle in _read_block(table, ioff, isize):
        boff, p = get_varint(handle, 0)
        bsize, _ = get_varint(handle, p)
        items.extend(_read_block(table, boff, bsize))
    return items


# -- bundles <-> named arrays ----------------------------------------------------------------------------
def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    items, offset = [(b"", HEADER)], 0
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        for name in sorted(tensors):
            arr = np.asarray(tensors[name])
            if arr.ndim and not arr.flags.c_contiguous:     # (ascontiguousarray would turn scalars into [1])
                arr = np.ascontiguousarray(arr)
            if arr.dtype not in DT_CODE:
                raise TypeError("dtype {} of '{}' is not supported".format(arr.dtype, name))
            raw = arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
            items.append((name.encode(), encode_entry(DT_CODE[arr.dtype], arr.shape, offset, len(raw),
                                                      mask_crc(crc32c(raw)))))
            data.write(raw)
            offset += len(raw)
    with open(prefix + ".index", "wb") as index:
        index.write(write_table(items))


def read_bundle(prefix: str, verify: bool = True) -> Dict[str, np.ndarray]:
    with open(prefix + ".index", "rb") as handle:
        items = read_table(handle.read())
    out: Dict[str, np.ndarray] = {}
    with open(prefix + ".data-00000-of-00001", "rb") as data:
        for key, value in items:
            if key == b"":
                for field, _, val in _pb_fields(value):
                    if field == 1 and val != 1:
                        raise NotImplementedError("sharded bundles ({} shards) are not supported".format(val))
                    if field == 2 and val != 0:
                        raise NotImplementedError("big-endian bundles are not supported")
                continue
            entry = decode_entry(value)
            if entry["dtype"] not in DT:
                continue                      # strings etc.: nothing of the model
            data.seek(entry["offset"])
            raw = data.read(entry["size"])
            if verify and entry["crc32c"] is not None and unmask_crc(entry["crc32c"]) != crc32c(raw):
                raise ValueError("checksum mismatch in tensor '{}'".format(key.decode()))
            out[key.decode()] = np.frombuffer(raw, dtype=DT[entry["dtype"]]).reshape(entry["shape"]).copy()
    return out


# -- VariableStore bridge ------------------------------------------------------------------------------
def tf_shape(name: str, shape: Tuple[int, ...]) -> Tuple[int, ...]:
    """The shape TensorFlow gives the variable ``name`` (see the module docstring)."""
    parts = name.split("/")
    leaf = parts[-1]
    # scalars: Attention's attn_bias, FlatMultiAttention's attn_bias_<i>, the vector_bias of _vector_logit
    scalar = leaf == "attn_bias" or leaf == "vector_bias" or \
        (leaf.startswith("attn_bias_") and leaf[len("attn_bias_"):].isdigit())
    if scalar and tuple(shape) == (1,):
        return ()
    if leaf == "attn_v" and len(shape) == 1:          # combination.py:66-70: [1, 1, attention_state_size]
        return (1, 1) + tuple(shape)
    # tf.layers.conv2d filters of SpatialFiller's 1x1 projections.  (Attention's attn_key_projection is created
    # [in, out] and expanded inside the graph: attention/feed_forward.py:77-82,108-113.)
    conv_kernel = len(parts) >= 2 and parts[-1] == "kernel" and parts[-2].startswith("conv2d")
    if len(shape) == 2 and conv_kernel:
        return (1, 1) + tuple(shape)
    return tuple(shape)


def export_store(store, prefix: str, global_step: Optional[int] = None, with_adam: bool = False,
                 beta1: float = 0.9, beta2: float = 0.999) -> None:
    tensors = {}
    for name, arr in store.state_dict().items():
        tensors[name] = np.asarray(arr, np.float32).reshape(tf_shape(name, arr.shape))
    # variables of the reference's graph that nothing reads (VariableStore.declare_checkpoint_only): a Saver over all
    # global variables expects their keys -- and, being trainable, their Adam slots
    extras = store.checkpoint_only_values()
    tensors.update(extras)
    s0, s1 = store.slot_suffixes
    if with_adam and store.adam_m is not None:
        for name, arr in extras.items():
            tensors[name + s0] = np.zeros_like(arr)
            tensors[name + s1] = np.zeros_like(arr)
        m, v = store.adam_m.cpu().numpy(), store.adam_v.cpu().numpy()
        for name, spec in store.specs.items():
            shape = tf_shape(name, spec.shape)
            tensors[name + s0] = m[spec.offset:spec.offset + spec.size].reshape(shape)
            tensors[name + s1] = v[spec.offset:spec.offset + spec.size].reshape(shape)
        if s0 == "/Adam":                  # Adam's two non-slot variables; Adadelta has none
            step = global_step or 0
            tensors["beta1_power"] = np.float32(beta1 ** (step + 1))
            tensors["beta2_power"] = np.float32(beta2 ** (step + 1))
    if global_step is not None:
        tensors["global_step"] = np.int64(global_step)
    write_bundle(prefix, tensors)


def import_store(store, prefix: str, strict: bool = True) -> Dict[str, object]:
    """Load every variable of ``store`` found in the bundle (unit dimensions ignored).  Returns
    {"missing": [...], "unused": [...], "global_step": int or None}."""
    bundle = read_bundle(prefix)
    values, missing = {}, []
    for name, spec in store.specs.items():
        arr = bundle.get(name)
        if arr is None:
            missing.append(name)
            continue
        if [d for d in arr.shape if d != 1] != [d for d in spec.shape if d != 1]:
            raise ValueError("shape of '{}' in the checkpoint {} does not match {}".format(name, arr.shape, spec.shape))
        values[name] = np.asarray(arr, np.float32).reshape(spec.shape)
    if strict and missing:
        raise KeyError("variables missing from the checkpoint: {}".format(missing[:8]))
    store.load_state_dict(values, strict=False)
    from .variables import find_slot_suffixes
    s0, s1 = store.slot_suffixes = find_slot_suffixes(bundle, values, store.slot_suffixes)
    if all(n + s0 in bundle and n + s1 in bundle for n in values) and values:
        import torch
        m, v = store.ensure_adam()
        for name in values:
            spec = store.specs[name]
            m[spec.offset:spec.offset + spec.size] = torch.from_numpy(
                np.asarray(bundle[name + s0], np.float32).reshape(-1)).to(m.device)
            v[spec.offset:spec.offset + spec.size] = torch.from_numpy(
                np.asarray(bundle[name + s1], np.float32).reshape(-1)).to(v.device)
    extras = store.take_checkpoint_only(bundle)
    known = set(values) | set(extras)
    known |= {n + s for n in known for s in (s0, s1)}
    step = bundle.get("global_step")
    return {"missing": missing, "unused": sorted(set(bundle) - known - {"global_step", "beta1_power", "beta2_power"}),
            "global_step": None if step is None else int(step)}
