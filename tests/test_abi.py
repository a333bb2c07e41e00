"""The C-ABI library loads and exports exactly what include/nmhip.h declares
(no compute calls: there is no GPU in this container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from neuralmonkey_amd import build
    build.build(verbose=False)           # hipcc cross-compiles gfx950 without a GPU
    from neuralmonkey_amd import _lib
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "nmhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(nm_[a-z0-9_]+)\s*\(", text))


def test_header_matches_binding_table(lib):
    from neuralmonkey_amd import _lib
    assert header_symbols() == set(_lib.SIGNATURES), (
        header_symbols() ^ set(_lib.SIGNATURES))


def test_every_declared_symbol_is_exported(lib):
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_error_reporting_without_gpu(lib):
    assert lib.nm_version() >= 1
    # argument validation happens before any launch: a null operand is an error, not a crash
    rc = lib.nm_gemm_f32(None, 0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, 1, 0, 0, 0, 0, None, 0)
    assert rc < 0 and b"null" in lib.nm_last_error()
    # energies + split-S partial contexts + partial statistics + one arrival counter per key batch
    assert lib.nm_attn_workspace_bytes(128, 50, 1024) == 4 * (6400 + 128 * 5 * 1024 + 128 * 5 * 4 + 128)


def test_round5_entry_points_validate_before_any_launch(lib):
    """The cluster time loops, the split projection and Adadelta: sizes are plain arithmetic, a shape no device here
    can take is "not supported" (never an exception), null operands are errors with a text, not crashes."""
    assert lib.nm_gru_seq_supported(128, 512, 2) == 0             # no device: nothing is supported, callers step
    assert lib.nm_gru_seq_workspace_bytes(128, 512, 2) == 256     # ... and the workspace is its header alone
    assert lib.nm_proj_split_bytes(32000, 512) == 3 * 2 * 32000 * 512          # three bf16 planes
    assert lib.nm_proj_split_forget(None) == 0                    # forgetting everything with nothing registered
    rc = lib.nm_gru_seq_fwd(None, None, 5, 0, 0, 0, 0, None, 0, 0, None, 0, 0, None, 0, None)
    assert rc < 0 and b"nm_gru_seq_fwd" in lib.nm_last_error()
    rc = lib.nm_gru_seq_bwd(None, None, 5, 0, 0, None, 0, 0, None, 0, 0, None, 0, None)
    assert rc < 0 and b"nm_gru_seq_bwd" in lib.nm_last_error()
    rc = lib.nm_proj_split_prepare(None, None, 0, 0, 32000, 512, None, 0)
    assert rc < 0 and b"nm_proj_split_prepare" in lib.nm_last_error()
    rc = lib.nm_optim_clip_adadelta(None, None, None, None, None, None, None, None, None, None, None, 0, 0,
                                    1.0, 1.0, 0.95, 1e-6, None, 0)
    assert rc < 0 and b"null" in lib.nm_last_error()


def test_round6_entry_points_validate_before_any_launch(lib):
    """The ranged optimizer passes (sharded optimizer, the skip word of the cluster-loop recovery), the conditional
    zeroing of a gradient and the cluster loops' test hooks: arguments are checked before anything is launched."""
    none9 = [None] * 9
    rc = lib.nm_optim_partials(None, *none9[:8], 4, 2, 0.0, 0.0, 0, 4, None, 0)
    assert rc < 0 and b"nm_optim_partials" in lib.nm_last_error()
    rc = lib.nm_optim_segments(None, *none9[:6], 4, 2, None, None, 0)
    assert rc < 0 and b"nm_optim_segments" in lib.nm_last_error()
    buf = (ctypes.c_float * 64)()
    tabs = [ctypes.cast(buf, ctypes.c_void_p)] * 6
    # a chunk range that leaves the table is refused (the operands are never touched: no device here)
    rc = lib.nm_optim_apply(None, 0, buf, buf, buf, buf, *tabs, 4, 2, 1.0, 1e-3, 0.9, 0.999, 1e-8, 3, 9, None, buf, 4096)
    assert rc < 0 and b"chunk range" in lib.nm_last_error()
    rc = lib.nm_optim_apply(None, 7, buf, buf, buf, buf, *tabs, 4, 2, 1.0, 1e-3, 0.9, 0.999, 1e-8, 0, 4, None, buf, 4096)
    assert rc < 0 and b"kind" in lib.nm_last_error()
    # an empty range is a no-op, not a launch
    assert lib.nm_optim_apply(None, 0, buf, buf, buf, buf, *tabs, 4, 2, 1.0, 1e-3, 0.9, 0.999, 1e-8, 2, 2, None, buf,
                              4096) == 0
    rc = lib.nm_zero_if(None, None, None, 16)
    assert rc < 0 and b"nm_zero_if" in lib.nm_last_error()
    assert lib.nm_zero_if(None, buf, buf, 0) == 0
    assert lib.nm_gru_seq_force_give_up(3) == 0 and lib.nm_gru_seq_force_give_up(0) == 3     # a counter, nothing else
    rc = lib.nm_gru_seq_test_hog(None, 0, 1024, 10)
    assert rc < 0 and b"nm_gru_seq_test_hog" in lib.nm_last_error()
    # the fused layer-norm backward, the grouped weight-gradient product and the NematusGRU loops
    assert lib.nm_layer_norm_bwd_params_workspace_bytes(512) > 0
    rc = lib.nm_layer_norm_bwd_params(None, buf, buf, buf, buf, buf, buf, 4, 6, buf, buf, 0, buf, 1 << 20)
    assert rc < 0 and b"multiple of 4" in lib.nm_last_error()
    rc = lib.nm_layer_norm_bwd_params(None, buf, buf, buf, buf, buf, buf, 4, 16, buf, buf, 0, buf, 8)
    assert rc < 0 and b"workspace too small" in lib.nm_last_error()
    rc = lib.nm_gemm_f32_group(None, 1, 0, 64, 64, 64, None, 64, 64, 64, 1, 3)
    assert rc < 0 and b"nm_gemm_f32_group" in lib.nm_last_error()
    rc = lib.nm_gemm_f32_group(None, 1, 0, 64, 64, 64, buf, 64, 62, 64, 1, 3)
    assert rc < 0 and b"nm_gemm_f32_group" in lib.nm_last_error()
    from neuralmonkey_amd import _lib
    epi = _lib.GruEpilogue()
    rc = lib.nm_nematus_seq_fwd(None, ctypes.byref(epi), 3, 0, 0, 0, 0, None, 512, 0, None, 256, 0, None, None, None, 0,
                                None)
    assert rc < 0 and b"nm_nematus_seq_fwd" in lib.nm_last_error()
    epi.R, epi.H, epi.ndir = 16, 256, 1
    rc = lib.nm_nematus_seq_bwd(None, ctypes.byref(epi), 3, 0, 0, 0, buf, 512, 0, buf, 256, 0, buf, 1 << 20, None)
    assert rc < 0 and b"nm_nematus_seq_bwd: missing operand" in lib.nm_last_error()
    # (no device here: no shape is "supported", the workspace is the bare header)
    assert lib.nm_nematus_seq_workspace_bytes(16, 100, 1) == lib.nm_gru_seq_workspace_bytes(16, 100, 1)
    assert lib.nm_dec_step_cluster_supported(128, 100, 512, 512) == 0
    # ... the LSTM loops, the chained products / column sums / rank-1 updates of taped time loops, the chunk-list optimizer
    # passes, fills and copies, the fused NematusGRU cell
    rc = lib.nm_lstm_seq_fwd(None, ctypes.byref(epi), 3, 0, 0, 0, None, 1024, 0, 1.0, None, 0, None)
    assert rc < 0 and b"nm_lstm_seq_fwd" in lib.nm_last_error()
    rc = lib.nm_lstm_seq_bwd(None, ctypes.byref(epi), 3, 0, 0, buf, 1024, 0, buf, 1 << 20, None)
    assert rc < 0 and b"nm_lstm_seq_bwd: missing operand" in lib.nm_last_error()
    rc = lib.nm_gemm_f32_chain(None, 64, 64, 24, 3, buf, 64, 64, buf, 64, 1, None, 0)
    assert rc < 0 and b"multiple of 16" in lib.nm_last_error()
    rc = lib.nm_colsum_chain(None, buf, 1, 0, 3, 16, 62, 64, buf, 1, buf, 1 << 20)
    assert rc < 0 and b"nm_colsum_chain" in lib.nm_last_error()
    rc = lib.nm_outer_chain(None, buf, 65, 4, 8, 16, 8, 16, buf, 1)
    assert rc < 0 and b"at most 64 steps" in lib.nm_last_error()
    rc = lib.nm_optim_partials_list(None, buf, buf, *tabs, 4, 2, 0.0, 0.0, None, 3, buf, 4096)
    assert rc < 0 and b"bad chunk list" in lib.nm_last_error()
    assert lib.nm_optim_apply_list(None, 0, buf, buf, buf, buf, *tabs, 4, 2, 1.0, 1e-3, 0.9, 0.999, 1e-8, None, 0, None, buf,
                                   4096) == 0                                  # an empty list: no launch
    rc = lib.nm_fill_u32(None, None, 16, 0)
    assert rc < 0 and b"nm_fill_u32" in lib.nm_last_error()
    assert lib.nm_fill_u32(None, buf, 0, 7) == 0 and lib.nm_copy_d2d(None, buf, buf, 256) == 0       # nothing to do
    rc = lib.nm_nematus_cell_fwd(None, buf, 8, buf, 8, buf, 8, buf, 8, buf, 8, None, None, None, 0, 4, 8)
    assert rc < 0 and b"nm_nematus_cell_fwd: bad shape" in lib.nm_last_error()       # the gates are 2H wide
    rc = lib.nm_add_layer_norm_stats_fwd(None, buf, buf, buf, buf, buf, buf, buf, buf, 4, 6, 1e-6)
    assert rc < 0 and b"multiple of 4" in lib.nm_last_error()
    rc = lib.nm_test_xcc_ids(None, None, 8, 64)
    assert rc < 0 and b"nm_test_xcc_ids" in lib.nm_last_error()
    # a NematusGRUCell step's state half and point-wise part in one launch: H in steps of 8, never in place
    rc = lib.nm_nematus_state_step(None, buf, 12, buf, 36, None, buf, 36, buf, 12, None, None, None, 0, 4, 12)
    assert rc < 0 and b"nm_nematus_state_step: bad shape" in lib.nm_last_error()
    rc = lib.nm_nematus_state_step(None, buf, 8, buf, 24, None, buf, 24, buf, 8, None, None, None, 0, 4, 8)
    assert rc < 0 and b"may not overwrite" in lib.nm_last_error()
    rc = lib.nm_nematus_full_step(None, buf, 8, buf, 24, None, buf, 12, buf, 24, None, buf, 8, None, None, None, 0, 4, 8, 12)
    assert rc < 0 and b"nm_nematus_full_step: bad shape" in lib.nm_last_error()
    # one taped step's attention backward: operands, then shapes (C in float4 steps)
    rc = lib.nm_attn_step_bwd(None, buf, 8, buf, buf, None, buf, None, 8, buf, buf, buf, 8, 2, 4, 8, 8)
    assert rc < 0 and b"nm_attn_step_bwd: null pointer" in lib.nm_last_error()
    rc = lib.nm_attn_step_bwd(None, buf, 6, buf, buf, None, buf, buf, 8, buf, buf, buf, 8, 2, 4, 6, 8)
    assert rc < 0 and b"nm_attn_step_bwd: bad shape" in lib.nm_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from neuralmonkey_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.NMHipError, match="no CPU fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "neuralmonkey_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                code = re.sub(r"#.*", "", src).replace('"""', "")
                assert "oracle" not in code, os.path.join(dirpath, f)
                # the NumPy-eager TensorFlow stand-in that executes the reference's files is test infrastructure too
                assert "tf_eager" not in code and "ref_exec" not in code, os.path.join(dirpath, f)


def _header_struct(name):
    """[(C type, field)] of ``typedef struct <name> { ... } <name>;`` in include/nmhip.h, in declaration order."""
    text = open(os.path.join(ROOT, "include", "nmhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = re.match(r"((?:const\s+)?\w+\s*\*?)\s*(.*)", decl, flags=re.S).groups()
        for item in names.split(","):
            item = item.strip()
            fields.append((ctype.strip() + ("*" if item.startswith("*") else ""), item.lstrip("* ")))
    return fields


@pytest.mark.parametrize("cname,pyname", [("nm_step_problem", "StepProblem"), ("nm_decoder_step", "DecoderStep"),
                                          ("nm_gru_epilogue", "GruEpilogue")])
def test_descriptor_structs_match_the_header(cname, pyname):
    """The ctypes mirrors of the by-pointer descriptors have the header's fields, order and widths (a drift here is a
    silent ABI break: every field after it would be read from the wrong offset)."""
    from neuralmonkey_amd import _lib
    want = _header_struct(cname)
    got = getattr(_lib, pyname)._fields_
    assert [n for _, n in want] == [n for n, _ in got]
    for (ctype, field), (_, pytype) in zip(want, got):
        if "*" in ctype:
            assert pytype is ctypes.c_void_p, field
        elif ctype.endswith("int64_t"):
            assert ctypes.sizeof(pytype) == 8, field
        elif ctype.endswith("int32_t"):
            assert ctypes.sizeof(pytype) == 4, field
        else:
            raise AssertionError("unexpected field type {} {}".format(ctype, field))


@pytest.mark.parametrize("cname", ["nm_step_problem", "nm_decoder_step"])
def test_kernel_side_structs_match_the_header(cname):
    """csrc/nm_step.hip re-declares the descriptors it receives by pointer: field for field the header's."""
    text = open(os.path.join(ROOT, "neuralmonkey_amd", "csrc", "nm_step.hip")).read()
    text = re.sub(r"//[^\n]*", "", text)
    body = re.search(r"struct %s \{(.*?)\};" % cname, text, flags=re.S).group(1)
    mine = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = re.match(r"((?:const\s+)?\w+\s*\*?)\s*(.*)", decl, flags=re.S).groups()
        for item in names.split(","):
            item = item.strip()
            mine.append((ctype.strip() + ("*" if item.startswith("*") else ""), item.lstrip("* ")))
    norm = lambda fields: [(t.replace(" ", ""), n) for t, n in fields]
    assert norm(mine) == norm(_header_struct(cname))


def test_contexts_own_their_switches_and_timers(lib, monkeypatch):
    """nm_create reads the NM_* switches ONCE; a later change of the environment is seen by the next context
    only.  Timer state is per context; the default context cannot be destroyed; a destroyed handle is refused."""
    def switch(ctx, name):
        out = ctypes.c_int(12345)
        assert lib.nm_ctx_switch(ctx, name.encode(), ctypes.byref(out)) == 0, lib.nm_last_error()
        return out.value

    monkeypatch.delenv("NM_ATTN_WHOLE", raising=False)
    monkeypatch.delenv("NM_GEMM_SK", raising=False)
    a = ctypes.c_void_p()
    assert lib.nm_create(-1, ctypes.byref(a)) == 0 and a.value
    monkeypatch.setenv("NM_ATTN_WHOLE", "0")
    monkeypatch.setenv("NM_GEMM_SK", "4")
    b = ctypes.c_void_p()
    assert lib.nm_create(0, ctypes.byref(b)) == 0 and b.value and b.value != a.value
    assert switch(a, "attn_whole") == -1 and switch(a, "gemm_sk") == 0          # read before the change
    assert switch(b, "attn_whole") == 0 and switch(b, "gemm_sk") == 4
    assert lib.nm_ctx_device(b) == 0
    out = ctypes.c_int()
    assert lib.nm_ctx_switch(a, b"no_such_switch", ctypes.byref(out)) < 0

    default = lib.nm_ctx_current()
    assert default and default not in (a.value, b.value)
    assert lib.nm_ctx_bind(b) == 0 and lib.nm_ctx_current() == b.value
    assert switch(None, "gemm_sk") == 4                                        # NULL = the bound context
    assert lib.nm_ctx_bind(None) == 0 and lib.nm_ctx_current() == default

    # timers: enabling one context's recorder leaves the other's alone (no GPU: nothing is recorded, counts are 0)
    tot, cnt = ctypes.c_double(-1.0), ctypes.c_int64(-1)
    assert lib.nm_prof_enable(a, 1) == 0
    assert lib.nm_prof_attn_step(b, ctypes.byref(tot), ctypes.byref(cnt)) == 0 and cnt.value == 0
    assert lib.nm_prof_attn_step(a, ctypes.byref(tot), ctypes.byref(cnt)) == 0 and cnt.value == 0
    assert lib.nm_prof_enable(a, 0) == 0

    assert lib.nm_destroy(ctypes.c_void_p(default)) < 0 and b"default" in lib.nm_last_error()
    assert lib.nm_ctx_bind(a) == 0
    assert lib.nm_destroy(a) == 0                      # unbinds itself
    assert lib.nm_ctx_current() == default
    assert lib.nm_destroy(b) == 0


def test_gradient_exchange_entry_points_refuse_without_a_device_or_a_communicator():
    """nm_allreduce_* (RCCL resolved at run time, csrc/nm_comm.hip): no compute here -- the argument checks, and that
    a box without a GPU gets an error code and a message instead of a crash or a load failure of the library."""
    import ctypes
    from neuralmonkey_amd import _lib
    lib = _lib.load()
    assert lib.nm_allreduce_unique_id(None, 128) != 0 and b"128 bytes" in lib.nm_last_error()
    uid = ctypes.create_string_buffer(128)
    comm = ctypes.c_void_p()
    assert lib.nm_allreduce_init(3, 2, uid, ctypes.byref(comm)) != 0 and b"rank 3 of 2" in lib.nm_last_error()
    assert lib.nm_allreduce_bucket(None, None, None, 0) != 0 and b"not a communicator" in lib.nm_last_error()
    assert lib.nm_allreduce_wait(None, None) != 0 and lib.nm_allreduce_destroy(None) != 0
    import torch
    if not torch.cuda.is_available():
        assert lib.nm_allreduce_init(0, 1, uid, ctypes.byref(comm)) != 0
        assert b"nm_allreduce_init" in lib.nm_last_error()


def test_register_budgets_of_the_overlapped_kernels(lib):
    """What the training step's schedule rests on, read from the code objects inside the built library
    (tools/kernel_resources.py; no GPU): a cluster time loop (neuralmonkey_amd/csrc/nm_gru_cluster.hip, 8 waves per
    workgroup = 2 per SIMD) has to fit on a CU BESIDE one residency-capped leaf-GEMM workgroup (nm_gemm_f32 algo 4:
    gemm_tiled<4,2,1,2,..,16,false,1,1>, 8 waves = 2 per SIMD) within a SIMD's 512 vector registers per lane --
    measured when it did not: loops ten times slower (profiles/r05_cluster_loops.md).  And nothing on a hot path may
    spill: only the wide-beam variants of the first-step row scan (13 launches per beam batch) carry scratch."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from kernel_resources import kernel_resources
    finally:
        sys.path.pop(0)
    table = kernel_resources()
    assert len(table) > 300
    loops = {k: v for k, v in table.items() if "gru_cluster_" in k or "nematus_cluster_" in k}
    assert len(loops) == 8                                   # TF GRU / NematusGRU x forward / backward x one / two row tiles
    capped = {k: v for k, v in table.items()
              if re.match(r"_Z10gemm_tiledILi4ELi2ELi1ELi2ELb[01]ELb[01]ELb[01]ELi16ELb0ELi1ELi1ELb0EE", k)}
    assert len(capped) == 8
    for v in list(loops.values()) + list(capped.values()):
        assert v["max_threads"] == 512 and v["scratch"] == 0
    widest_loop = max(v["arch_vgprs"] for v in loops.values())
    widest_gemm = max(v["arch_vgprs"] for v in capped.values())
    assert widest_loop <= 160 and widest_gemm <= 80
    assert 2 * widest_loop + 2 * widest_gemm <= 512
    spilling = sorted(k for k, v in table.items() if v["scratch"])
    assert all("row_scan_kernel" in k for k in spilling), spilling

