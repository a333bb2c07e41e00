"""Registers, LDS and scratch of the kernels inside neuralmonkey_amd/libnmhip.so, read from the code objects' own
metadata (no GPU, no disassembly):

    python tools/kernel_resources.py                    # every kernel, widest first
    python tools/kernel_resources.py gru_cluster        # the kernels whose name contains the substring

The shared library carries one clang offload bundle per translation unit in its `.hip_fatbin` section; every bundle
holds a gfx950 ELF whose NT_AMDGPU_METADATA note (a MessagePack map, `amdhsa.kernels`) records what the loader needs
to place a workgroup: `.vgpr_count`, `.agpr_count`, `.sgpr_count`, `.group_segment_fixed_size`,
`.private_segment_fixed_size` (scratch = spills), `.max_flat_workgroup_size`.  The budget that matters on gfx950 is
512 vector registers per lane slot of a SIMD (MI355X_MICROARCH.md): a wave of v registers (`.vgpr_count` is the
unified VGPR + AGPR total, allocated in blocks of 8) lets floor(512 / v) waves share a SIMD, and the training step's
schedule depends on two of these numbers staying where they are
(tests/test_abi.py::test_register_budgets_of_the_overlapped_kernels).  Dynamic LDS is a launch argument and not in
the metadata."""
import os
import struct
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBRARY = os.path.join(ROOT, "neuralmonkey_amd", "libnmhip.so")
BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
NT_AMDGPU_METADATA = 32


def _sections(blob):
    """(name, offset, size, type) of an ELF64 little-endian image's sections."""
    if blob[:4] != b"\x7fELF" or blob[4] != 2 or blob[5] != 1:
        raise ValueError("not a little-endian ELF64 image")
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    raw = []
    for i in range(shnum):
        name, kind, _flags, _addr, offset, size = struct.unpack_from("<IIQQQQ", blob, shoff + i * shentsize)
        raw.append((name, offset, size, kind))
    strings = blob[raw[shstrndx][1]:raw[shstrndx][1] + raw[shstrndx][2]]
    return [(strings[n:strings.index(b"\0", n)].decode(), off, size, kind) for n, off, size, kind in raw]


def _bundled_code_objects(fatbin):
    """The device ELF images of every offload bundle in a `.hip_fatbin` section."""
    at = fatbin.find(BUNDLE_MAGIC)
    while at >= 0:
        count, = struct.unpack_from("<Q", fatbin, at + len(BUNDLE_MAGIC))
        cursor = at + len(BUNDLE_MAGIC) + 8
        for _ in range(count):
            offset, size, triple_len = struct.unpack_from("<QQQ", fatbin, cursor)
            triple = fatbin[cursor + 24:cursor + 24 + triple_len].decode()
            cursor += 24 + triple_len
            if "amdgcn" in triple and size:
                yield triple, fatbin[at + offset:at + offset + size]
        at = fatbin.find(BUNDLE_MAGIC, cursor)


def _kernel_records(code_object):
    for name, offset, size, kind in _sections(code_object):
        if kind != 7:                                   # SHT_NOTE
            continue
        cursor, end = offset, offset + size
        while cursor + 12 <= end:
            namesz, descsz, note_type = struct.unpack_from("<III", code_object, cursor)
            desc_at = cursor + 12 + (namesz + 3) // 4 * 4
            if note_type == NT_AMDGPU_METADATA and code_object[cursor + 12:cursor + 12 + namesz].rstrip(b"\0") == b"AMDGPU":
                meta = msgpack.unpackb(code_object[desc_at:desc_at + descsz], raw=False, strict_map_key=False)
                for kernel in meta.get("amdhsa.kernels", []):
                    yield kernel
            cursor = desc_at + (descsz + 3) // 4 * 4


def kernel_resources(library=LIBRARY):
    """{mangled kernel name: dict(vgpr, agpr, sgpr, lds, scratch, max_threads, arch_vgprs)} of a built library;
    ``arch_vgprs`` is what one wave takes of a SIMD's 512: the unified count (``vgpr``, which includes ``agpr``) rounded
    up to the allocation block of 8."""
    with open(library, "rb") as fh:
        image = fh.read()
    fatbins = [(off, size) for name, off, size, _ in _sections(image) if name == ".hip_fatbin"]
    if not fatbins:
        raise ValueError("{} has no .hip_fatbin section".format(library))
    out = {}
    for off, size in fatbins:
        for triple, code_object in _bundled_code_objects(image[off:off + size]):
            if "gfx950" not in triple:
                raise ValueError("code object for {} in a gfx950-only library".format(triple))
            for k in _kernel_records(code_object):
                vgpr, agpr = int(k.get(".vgpr_count", 0)), int(k.get(".agpr_count", 0))
                out[k[".name"]] = dict(vgpr=vgpr, agpr=agpr, sgpr=int(k.get(".sgpr_count", 0)),
                                       lds=int(k.get(".group_segment_fixed_size", 0)),
                                       scratch=int(k.get(".private_segment_fixed_size", 0)),
                                       max_threads=int(k.get(".max_flat_workgroup_size", 0)),
                                       arch_vgprs=(vgpr + 7) // 8 * 8)
    return out


def main():
    needle = sys.argv[1] if len(sys.argv) > 1 else ""
    table = kernel_resources()
    rows = sorted(((v["arch_vgprs"], name, v) for name, v in table.items() if needle in name), reverse=True)
    print("{:>5} {:>5} {:>5} {:>7} {:>7} {:>7} {:>9}  kernel".format("regs", "agpr", "sgpr", "lds", "scratch", "threads",
                                                                   "waves/SIMD"))
    for regs, name, v in rows:
        print("{:>5} {:>5} {:>5} {:>7} {:>7} {:>7} {:>9}  {}".format(regs, v["agpr"], v["sgpr"], v["lds"], v["scratch"],
                                                                     v["max_threads"], min(8, 512 // max(regs, 1)), name))
    print("{} kernels".format(len(rows)))


if __name__ == "__main__":
    main()
