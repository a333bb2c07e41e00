"""Thin Python wrappers over the C-ABI: torch tensors in, HIP kernels out.

torch is used only as the device allocator and stream owner; every arithmetic
op on the hot path is a libnmhip kernel launched on torch's current stream.
"""
import ctypes
import struct
import os
import threading
from typing import Optional

import numpy as np
import torch

from . import _lib

ACT = {None: 0, "none": 0, "tanh": 1, "relu": 2}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda, (t.dtype, t.device)
    return t


def _i32(t):
    assert t.dtype == torch.int32 and t.is_cuda, (t.dtype, t.device)
    return t


def add_layer_norm_fwd(a, x, gamma, beta, sum_out, out, eps=1e-6):
    """sum_out = a + x ; out = layer_norm(sum_out) over the last dimension (contiguous [rows, D] operands)."""
    lib = _lib.load()
    d = x.shape[-1]
    rows = x.numel() // d
    assert a.is_contiguous() and x.is_contiguous() and sum_out.is_contiguous() and out.is_contiguous()
    assert a.numel() == x.numel() == sum_out.numel() == out.numel()
    _lib.check(lib.nm_add_layer_norm_fwd(_stream(), a.data_ptr(), d, x.data_ptr(), d, gamma.data_ptr(),
                                         beta.data_ptr(), sum_out.data_ptr(), d, out.data_ptr(), d, rows, d, float(eps)),
               "nm_add_layer_norm_fwd")
    return sum_out, out


def add_layer_norm_stats_ok(a, x, gamma, beta) -> bool:
    """Can ``add_layer_norm_stats_fwd`` take these operands (contiguous rows of D <= 2048 floats, D % 4 == 0, aligned)?"""
    d = x.shape[-1]
    return (x.is_cuda and d % 4 == 0 and d <= 2048 and a.is_contiguous() and x.is_contiguous() and a.shape == x.shape
            and all(t.data_ptr() % 16 == 0 for t in (a, x, gamma, beta)))


def add_layer_norm_stats_fwd(a, x, gamma, beta, sum_out, out, mean, rstd, eps=1e-6):
    """sum_out = a + x ; out = layer_norm(sum_out), with the row statistics the backward pass reads
    (nm_add_layer_norm_stats_fwd: one wave per row)."""
    d = x.shape[-1]
    rows = x.numel() // d
    assert sum_out.is_contiguous() and out.is_contiguous()
    _lib.check(_lib.load().nm_add_layer_norm_stats_fwd(_stream(), a.data_ptr(), x.data_ptr(), gamma.data_ptr(),
                                                       beta.data_ptr(), sum_out.data_ptr(), out.data_ptr(),
                                                       mean.data_ptr(), rstd.data_ptr(), rows, d, float(eps)),
               "nm_add_layer_norm_stats_fwd")
    return sum_out, out


GEMM_BACKGROUND = 4     # nm_gemm_f32 algo: residency-capped 128x128 tiles for a long leaf GEMM on a side stream


def gemm(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, bias=None, act=None,
         trans_a=False, trans_b=False, accumulate=False, algo=0):
    """out[M,N] = act(op(a) @ op(b) + bias (+ out)).  2-D (or batched 3-D with
    equal leading dim) row-major views with unit inner stride."""
    lib = _lib.load()
    _f32(a), _f32(b)
    batched = a.dim() == 3
    if batched:
        assert b.dim() == 3 and a.shape[0] == b.shape[0]
        nb = a.shape[0]
        a2, b2 = a[0], b[0]
        s_a, s_b = a.stride(0), b.stride(0)
    else:
        nb, a2, b2, s_a, s_b = 1, a, b, 0, 0
    assert a2.stride(1) == 1 and b2.stride(1) == 1, "inner stride must be 1"
    m, k = (a2.shape[1], a2.shape[0]) if trans_a else (a2.shape[0], a2.shape[1])
    kb, n = (b2.shape[1], b2.shape[0]) if trans_b else (b2.shape[0], b2.shape[1])
    assert k == kb, f"inner dims differ: {k} vs {kb}"
    if out is None:
        assert not accumulate
        out = torch.empty((nb, m, n) if batched else (m, n), dtype=torch.float32, device=a.device)
    o2 = out[0] if batched else out
    assert tuple(o2.shape) == (m, n) and o2.stride(1) == 1
    s_c = out.stride(0) if batched else 0
    ws = _gemm_workspace(a.device) if (nb == 1 and k >= 1024) else None
    _lib.check(lib.nm_gemm_f32(_stream(), int(trans_a), int(trans_b), m, n, k,
                               a.data_ptr(), a2.stride(0), b.data_ptr(), b2.stride(0),
                               out.data_ptr(), o2.stride(0), _p(bias), ACT[act], int(accumulate),
                               nb, s_a, s_b, s_c, algo, _p(ws), ws.numel() * 4 if ws is not None else 0),
               "nm_gemm_f32")
    return out


_GROUP_TABLES = {}


def gemm_group(items, trans_a=False, trans_b=False, accumulate=True):
    """``items`` = [(a, b, out)] of ONE shape: out_i (+)= op(a_i) @ op(b_i) in one launch (nm_gemm_f32_group).  The
    device table of operand pointers is built once per distinct list of pointers and kept (a training step names the
    same persistent buffers every time; a captured step replays the launch with its table)."""
    lib = _lib.load()
    a0, b0, o0 = items[0]
    m, k = (a0.shape[1], a0.shape[0]) if trans_a else (a0.shape[0], a0.shape[1])
    n = b0.shape[0] if trans_b else b0.shape[1]
    lda, ldb, ldc = a0.stride(0), b0.stride(0), o0.stride(0)
    ptrs = []
    for a, b, o in items:
        assert a.shape == a0.shape and b.shape == b0.shape and o.shape == o0.shape
        assert (a.stride(0), b.stride(0), o.stride(0)) == (lda, ldb, ldc) and a.stride(1) == b.stride(1) == o.stride(1) == 1
        assert a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and o.data_ptr() % 16 == 0
        ptrs += [a.data_ptr(), b.data_ptr(), o.data_ptr()]
    assert len({p for p in ptrs[2::3]}) == len(items), "two products of a group share their output"
    table = _pointer_table(a0.device, ptrs)
    _lib.check(lib.nm_gemm_f32_group(_stream(), int(trans_a), int(trans_b), m, n, k, table.data_ptr(), lda, ldb, ldc,
                                     int(accumulate), len(items)), "nm_gemm_f32_group")


def _pointer_table(device, ptrs):
    key = (device, tuple(ptrs))
    table = _GROUP_TABLES.get(key)
    if table is None:
        if len(_GROUP_TABLES) > 1024:
            _GROUP_TABLES.clear()
        table = _GROUP_TABLES[key] = torch.tensor(ptrs, dtype=torch.int64, device=device)
    return table


def gemm_chain(members, out, accumulate=True):
    """``out (+)= sum_i a_i^T @ b_i`` over ``members`` = [(a_i, b_i)] of one shape ([rows, M] and [rows, N]) as ONE product
    whose K dimension is the chain of the members (nm_gemm_f32_chain): the weight gradient of a taped time loop."""
    lib = _lib.load()
    a0, b0 = members[0]
    rows, m, n = a0.shape[0], a0.shape[1], b0.shape[1]
    lda, ldb = a0.stride(0), b0.stride(0)
    ptrs = []
    for a, b in members:
        assert a.shape == a0.shape and b.shape == b0.shape and (a.stride(0), b.stride(0)) == (lda, ldb)
        assert a.stride(1) == 1 and b.stride(1) == 1 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
        ptrs += [a.data_ptr(), b.data_ptr(), 0]
    assert out.shape == (m, n) and out.stride(1) == 1
    table = _pointer_table(a0.device, ptrs)
    ws = _gemm_workspace(a0.device)
    _lib.check(lib.nm_gemm_f32_chain(_stream(), m, n, rows, len(members), table.data_ptr(), lda, ldb, out.data_ptr(),
                                     out.stride(0), int(accumulate), ws.data_ptr(), ws.numel() * 4), "nm_gemm_f32_chain")
    return out


OUTER_CHAIN_MAX = 64


def outer_chain(members, out, accumulate=True):
    """``out[b, s, c] (+)= sum_t w_t[b, s] * d_t[b, c]`` over ``members`` = [(w_t [B, >=S], d_t [B, C])] (nm_outer_chain);
    ``out`` [B, S, C] contiguous."""
    w0, d0 = members[0]
    bsz, s, c = out.shape
    assert out.is_contiguous() and len(members) <= OUTER_CHAIN_MAX
    ptrs = []
    for w, d in members:
        assert w.shape[0] == bsz and d.shape == (bsz, c) and w.stride(1) == 1 and d.stride(1) == 1
        assert w.stride(0) == w0.stride(0) and d.stride(0) == d0.stride(0) and w.shape[1] >= s
        ptrs += [w.data_ptr(), d.data_ptr()]
    table = _pointer_table(out.device, ptrs)
    _lib.check(_lib.load().nm_outer_chain(_stream(), table.data_ptr(), len(members), bsz, s, c, w0.stride(0), d0.stride(0),
                                          out.data_ptr(), int(accumulate)), "nm_outer_chain")
    return out


def colsum_chain(members, out, accumulate=True):
    """``out[c] (+)= sum_i sum_r x_i[r, c]`` over tensors of one shape (nm_colsum_chain): a bias gradient of a taped time
    loop in one launch."""
    lib = _lib.load()
    x0 = members[0]
    rows, cols, ld = x0.shape[0], x0.shape[1], x0.stride(0)
    for x in members:
        assert x.shape == x0.shape and x.stride(0) == ld and x.stride(1) == 1 and x.data_ptr() % 16 == 0
    table = _pointer_table(x0.device, [x.data_ptr() for x in members])
    ws = _colsum_workspace(x0.device, cols)
    _lib.check(lib.nm_colsum_chain(_stream(), table.data_ptr(), 1, 0, len(members), rows, ld, cols, out.data_ptr(),
                                   int(accumulate), ws.data_ptr(), ws.numel() * 4), "nm_colsum_chain")
    return out


_GEMM_WS = {}
GEMM_WORKSPACE_BYTES = 128 << 20
# Scratch that kernels re-use from call to call (split-K slabs, column-sum partials + tickets) is keyed by the stream
# the launch goes to -- but every torch.cuda.graph capture uses torch's ONE capture stream, so two captured graphs
# would share a workspace and may later replay on different streams at the same time (a look-ahead encoder graph
# next to the running batch's decoding graphs).  Whoever captures work for a second stream sets a tag
# (runtime.Session._run_ahead): tagged launches get workspaces of their own.  The tag is per THREAD, like the library's
# context (nm_cur()): another session launching from another thread (the input pipeline's prefetcher, an ensemble's
# sessions) keeps its own.
_TLS = threading.local()


def workspace_tag():
    return getattr(_TLS, "workspace_tag", None)


def set_workspace_tag(tag):
    """Sets this thread's tag; returns the previous one (to be restored by the caller)."""
    old = workspace_tag()
    _TLS.workspace_tag = tag
    return old


def _gemm_workspace(device):
    """Persistent split-K slab buffer, one per (device, stream, tag): kernels of one
    stream are ordered, kernels of different streams must not share slabs."""
    key = (device, _stream(), workspace_tag())
    ws = _GEMM_WS.get(key)
    if ws is None:
        ws = torch.empty(GEMM_WORKSPACE_BYTES // 4, dtype=torch.float32, device=device)
        _GEMM_WS[key] = ws
    return ws


def embedding_gather(table, ids, out=None, mask_pad=False, scale=1.0):
    lib = _lib.load()
    _f32(table), _i32(ids)
    n = ids.numel()
    e = table.shape[1]
    if out is None:
        out = torch.empty(tuple(ids.shape) + (e,), dtype=torch.float32, device=table.device)
    ldo = out.stride(-2) if out.dim() >= 2 else e
    _lib.check(lib.nm_embedding_gather(_stream(), table.data_ptr(), table.shape[0], e,
                                       ids.data_ptr(), n, out.data_ptr(), ldo, int(mask_pad),
                                       float(scale)), "nm_embedding_gather")
    return out


def _rev_mask(ndir, reverse_dir0):
    """bit d set -> direction d walks its sequence backwards (reverse_sequence)."""
    return 1 if reverse_dir0 else (2 if ndir == 2 else 0)


def gru_gates_fwd(xp, x_dir_off, x_row_stride, x_time_stride, hg, h, ru, rh, lengths, t, ndir, rows, hsz,
                  reverse_dir0=False):
    lib = _lib.load()
    _lib.check(lib.nm_gru_gates_fwd(_stream(), xp.data_ptr(), x_dir_off, x_row_stride, x_time_stride,
                                    hg.data_ptr(), h.data_ptr(), ru.data_ptr(), rh.data_ptr(),
                                    _p(lengths), t, _rev_mask(ndir, reverse_dir0), ndir, rows, hsz),
               "nm_gru_gates_fwd")


def gru_blend_fwd(xp, x_dir_off, x_row_stride, x_time_stride, hc, ru, h_in, h_out, c_save, out,
                  out_dir_off, out_row_stride, out_time_stride, lengths, t, ndir, rows, hsz,
                  reverse_dir0=False):
    lib = _lib.load()
    _lib.check(lib.nm_gru_blend_fwd(_stream(), xp.data_ptr(), x_dir_off, x_row_stride, x_time_stride,
                                    hc.data_ptr(), ru.data_ptr(), h_in.data_ptr(), h_out.data_ptr(),
                                    _p(c_save), _p(out), out_dir_off, out_row_stride, out_time_stride,
                                    _p(lengths), t, _rev_mask(ndir, reverse_dir0), ndir, rows, hsz),
               "nm_gru_blend_fwd")


def layer_norm_fwd(x, gamma, beta, out=None, mean=None, rstd=None, eps=1e-6):
    lib = _lib.load()
    d = x.shape[-1]
    rows = x.numel() // d
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.nm_layer_norm_fwd(_stream(), x.data_ptr(), d, gamma.data_ptr(), beta.data_ptr(),
                                     out.data_ptr(), d, _p(mean), _p(rstd), rows, d, float(eps)),
               "nm_layer_norm_fwd")
    return out


def copy_cols(src, dst):
    """dst[:, :w] = src (2-D views, inner stride 1)."""
    lib = _lib.load()
    assert src.dim() == 2 and dst.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1
    _lib.check(lib.nm_copy_cols(_stream(), src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0),
                                src.shape[0], src.shape[1]), "nm_copy_cols")


def reduce_sum(x, out):
    lib = _lib.load()
    assert x.is_contiguous()
    _lib.check(lib.nm_reduce_sum(_stream(), x.data_ptr(), x.numel(), out.data_ptr()), "nm_reduce_sum")
    return out


def log_softmax_from_stats(x, rmax, rlse, out):
    lib = _lib.load()
    assert x.dim() == 2 and out.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1
    _lib.check(lib.nm_log_softmax(_stream(), x.data_ptr(), x.stride(0), rmax.data_ptr(), rlse.data_ptr(),
                                  out.data_ptr(), out.stride(0), x.shape[0], x.shape[1]), "nm_log_softmax")
    return out


def attn_workspace(rows, s, c, device):
    lib = _lib.load()
    nbytes = lib.nm_attn_workspace_bytes(rows, s, c)
    # zeroed ONCE: the tail holds the arrival counters of the in-kernel merge, which the kernels leave at zero
    return zero(torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device))


def gru_rh_seq(ru_all, hprev, out, lengths, ndir, hsz, reverse_dir0=False):
    lib = _lib.load()
    b, s = hprev.shape[0], hprev.shape[1]
    _lib.check(lib.nm_gru_rh_seq(_stream(), ru_all.data_ptr(), hprev.data_ptr(), out.data_ptr(),
                                 _p(lengths), _rev_mask(ndir, reverse_dir0), b, s, ndir, hsz),
               "nm_gru_rh_seq")


def attn_fwd(y, hf, states, mask, v, bias, rows_per_key, ctx, weights, workspace, energies_out=None):
    """Fused Bahdanau step.  y [R,A]; hf [Bk,S,A]; states [Bk,S,C]; mask [Bk,S]."""
    lib = _lib.load()
    r, a = y.shape
    _, s, c = states.shape
    assert hf.is_contiguous() and states.is_contiguous() and y.is_contiguous()
    assert ctx.stride(1) == 1
    _lib.check(lib.nm_attn_fwd(_stream(), y.data_ptr(), hf.data_ptr(), states.data_ptr(), _p(mask),
                               v.data_ptr(), _p(bias), r, rows_per_key, s, a, c, ctx.data_ptr(),
                               ctx.stride(0), _p(weights), workspace.data_ptr(),
                               workspace.numel() * 4, _p(energies_out)), "nm_attn_fwd")


def attn_fwd_time_major(y_all, hf, states, mask, v, bias, ctx_all, w_all, workspace, energies_out=None):
    """All T teacher-forced attention steps in one launch.  y_all [T,B,A] (time-major query
    projections), ctx_all [T,B,C], w_all / energies_out [T,B,S]; keys are read once per sentence."""
    lib = _lib.load()
    t, b, a = y_all.shape
    _, s, c = states.shape
    assert y_all.is_contiguous() and ctx_all.is_contiguous() and hf.is_contiguous() and states.is_contiguous()
    _lib.check(lib.nm_attn_fwd_multi(_stream(), y_all.data_ptr(), hf.data_ptr(), states.data_ptr(), _p(mask),
                                     v.data_ptr(), _p(bias), b, t, 1, b, s, a, c, ctx_all.data_ptr(), c,
                                     _p(w_all), workspace.data_ptr(), workspace.numel() * 4,
                                     _p(energies_out)), "nm_attn_fwd_multi")


def row_stats(x, rmax=None, lse=None, argmax=None):
    lib = _lib.load()
    assert x.dim() == 2 and x.stride(1) == 1
    _lib.check(lib.nm_row_stats(_stream(), x.data_ptr(), x.stride(0), x.shape[0], x.shape[1],
                                _p(rmax), _p(lse), _p(argmax)), "nm_row_stats")


def gumbel_argmax(x, salt: int, out):
    """One categorical draw per row from softmax(x) (tf.multinomial): argmax of x + Gumbel noise of (salt, row, col)."""
    lib = _lib.load()
    assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.int32
    _lib.check(lib.nm_gumbel_argmax(_stream(), x.data_ptr(), x.stride(0), x.shape[0], x.shape[1],
                                    int(salt) & 0xFFFFFFFF, out.data_ptr()), "nm_gumbel_argmax")


def greedy_update(argmax, finished, sym_out, mask_out, end_id, all_finished=None):
    lib = _lib.load()
    _lib.check(lib.nm_greedy_update(_stream(), argmax.data_ptr(), finished.data_ptr(),
                                    sym_out.data_ptr(), _p(mask_out), argmax.numel(), end_id,
                                    _p(all_finished)), "nm_greedy_update")


def xent(logits, targets, weights, loss_rows, grad_scale=None, write_grad=False, label_smoothing=0.0):
    lib = _lib.load()
    assert logits.dim() == 2 and logits.stride(1) == 1
    _lib.check(lib.nm_xent(_stream(), logits.data_ptr(), logits.stride(0), logits.shape[0],
                           logits.shape[1], targets.data_ptr(), _p(weights), _p(loss_rows),
                           _p(grad_scale), int(write_grad), float(label_smoothing)), "nm_xent")


XENT_COLSUM_ROWS = 256        # one 1024-thread workgroup per CU


def xent_colsum_ok(logits) -> bool:
    v = logits.shape[1]
    return (logits.is_cuda and v % 4 == 0 and v <= 32768 and logits.stride(0) % 4 == 0 and
            logits.data_ptr() % 16 == 0 and os.environ.get("NM_XENT_COLSUM", "1") != "0")


def xent_colsum(logits, targets, weights, loss_rows, grad_scale, label_smoothing, partial):
    """Cross entropy + in-place gradient as ``xent(..., write_grad=True)``; ``partial`` [G, V] receives per-workgroup
    column sums of the gradient (G <= rows): ``colsum(partial, bias_grad)`` is the projection's bias gradient."""
    lib = _lib.load()
    assert logits.dim() == 2 and logits.stride(1) == 1 and partial.is_contiguous()
    assert partial.shape[1] == logits.shape[1] and 1 <= partial.shape[0] <= logits.shape[0]
    _lib.check(lib.nm_xent_colsum(_stream(), logits.data_ptr(), logits.stride(0), logits.shape[0], logits.shape[1],
                                  targets.data_ptr(), _p(weights), _p(loss_rows), _p(grad_scale),
                                  float(label_smoothing), partial.data_ptr(), partial.shape[0]), "nm_xent_colsum")


def beam_workspace(b, k, v, device):
    lib = _lib.load()
    return torch.empty((lib.nm_beam_workspace_bytes(b, k, v) + 3) // 4, dtype=torch.float32, device=device)


def beam_topk_step(logits, b, k, rmax, rlse, logprob_sum, lengths, finished, penalty, end_id,
                   out_score, out_word, out_beam, out_logprob_sum, out_lengths, out_finished,
                   out_src_row, workspace, all_finished=None):
    lib = _lib.load()
    assert logits.dim() == 2 and logits.stride(1) == 1 and logits.shape[0] == b * k
    _lib.check(lib.nm_beam_topk_step(_stream(), logits.data_ptr(), logits.stride(0), b, k,
                                     logits.shape[1], rmax.data_ptr(), rlse.data_ptr(),
                                     logprob_sum.data_ptr(), lengths.data_ptr(), finished.data_ptr(),
                                     penalty.data_ptr(), end_id, out_score.data_ptr(),
                                     out_word.data_ptr(), out_beam.data_ptr(),
                                     out_logprob_sum.data_ptr(), out_lengths.data_ptr(),
                                     out_finished.data_ptr(), out_src_row.data_ptr(),
                                     workspace.data_ptr(), workspace.numel() * 4, _p(all_finished)),
               "nm_beam_topk_step")


def beam_topk_step_fused(logits, b, k, logprob_sum, lengths, finished, penalty, end_id, out_score, out_word,
                         out_beam, out_logprob_sum, out_lengths, out_finished, out_src_row, workspace, rmax, rlse,
                         all_finished=None):
    """One beam body straight from the raw parent logits (row statistics fused into the scan)."""
    lib = _lib.load()
    assert logits.dim() == 2 and logits.stride(1) == 1 and logits.shape[0] == b * k
    _lib.check(lib.nm_beam_topk_step_fused(
        _stream(), logits.data_ptr(), logits.stride(0), b, k, logits.shape[1], logprob_sum.data_ptr(),
        lengths.data_ptr(), finished.data_ptr(), penalty.data_ptr(), end_id, out_score.data_ptr(),
        out_word.data_ptr(), out_beam.data_ptr(), out_logprob_sum.data_ptr(), out_lengths.data_ptr(),
        out_finished.data_ptr(), out_src_row.data_ptr(), workspace.data_ptr(), workspace.numel() * 4,
        _p(all_finished), rmax.data_ptr(), rlse.data_ptr()), "nm_beam_topk_step_fused")


def logits_stats_numel(rows, vocab):
    """Floats of the per-tile row statistics ``logits_stats_gemm`` writes for [rows, vocab] logits."""
    return _lib.load().nm_logits_stats_bytes(rows, vocab) // 4


def logits_stats_buffer(rows, vocab, device):
    return torch.empty(logits_stats_numel(rows, vocab), dtype=torch.float32, device=device)


def logits_stats_gemm(state, w, bias, stats, out=None, trans_b=False):
    """logits = state . W + b with {max, sum exp, argmax} of every 128-column tile of every row written to
    ``stats``; the logits are stored only when ``out`` is given."""
    lib = _lib.load()
    _f32(state), _f32(w)
    m, k = state.shape
    n = w.shape[0] if trans_b else w.shape[1]
    assert state.stride(1) == 1 and w.stride(1) == 1 and (out is None or (out.stride(1) == 1 and out.shape == (m, n)))
    _lib.check(lib.nm_logits_stats_gemm(_stream(), int(trans_b), m, n, k, state.data_ptr(), state.stride(0),
                                        w.data_ptr(), w.stride(0), _p(bias), _p(out),
                                        out.stride(0) if out is not None else 0, stats.data_ptr(),
                                        stats.numel() * 4), "nm_logits_stats_gemm")


PROJ_SPLIT = os.environ.get("NM_PROJ_SPLIT", "0") == "1"


def proj_split_prepare(w, trans_b=False, planes=None):
    """OPT-IN (NM_PROJ_SPLIT=1): split the vocabulary projection's weights ``w`` ([K,N]; [N,K] with ``trans_b``) into
    three bf16 planes and register them: ``logits_stats_gemm`` with this ``w`` then runs six bf16 matrix-core products
    instead of the exact-fp32 kernel (nm_proj_split_prepare).  Returns the planes (keep them alive while ``w`` is
    registered; call again when ``w`` changes)."""
    lib = _lib.load()
    _f32(w)
    assert w.dim() == 2 and w.stride(1) == 1
    n, k = (w.shape[0], w.shape[1]) if trans_b else (w.shape[1], w.shape[0])
    nbytes = lib.nm_proj_split_bytes(n, k)
    if nbytes <= 0:
        return None
    if planes is None or planes.numel() * 2 < nbytes:
        planes = torch.empty(nbytes // 2, dtype=torch.int16, device=w.device)
    _lib.check(lib.nm_proj_split_prepare(_stream(), w.data_ptr(), w.stride(0), int(trans_b), n, k, planes.data_ptr(),
                                         planes.numel() * 2), "nm_proj_split_prepare")
    return planes


def proj_split_forget(w=None):
    _lib.check(_lib.load().nm_proj_split_forget(_p(w)), "nm_proj_split_forget")


def greedy_finish(stats, vocab, finished, sym_out, mask_out, end_id, all_finished=None, table=None, emb_out=None,
                  argmax_out=None, max_out=None, lse_out=None):
    """Greedy step tail from the tile statistics: argmax, symbol / finished update, next input embedding."""
    lib = _lib.load()
    rows = sym_out.numel()
    tile = lib.nm_logits_stats_tile(rows)
    ntiles = (vocab + tile - 1) // tile
    if emb_out is not None:
        assert emb_out.dim() == 2 and emb_out.stride(1) == 1 and emb_out.shape[0] == rows
    _lib.check(lib.nm_greedy_finish(_stream(), stats.data_ptr(), ntiles, rows, finished.data_ptr(),
                                    sym_out.data_ptr(), _p(mask_out), end_id, _p(all_finished), _p(table),
                                    table.shape[0] if table is not None else 0,
                                    table.shape[1] if table is not None else 0, _p(emb_out),
                                    emb_out.stride(0) if emb_out is not None else 0, _p(argmax_out), _p(max_out),
                                    _p(lse_out)), "nm_greedy_finish")


def beam_topk_step_tiles(logits, stats, b, k, logprob_sum, lengths, finished, penalty, end_id, out_score, out_word,
                         out_beam, out_logprob_sum, out_lengths, out_finished, out_src_row, workspace, rmax, rlse,
                         all_finished=None):
    """One beam body from logits whose tile statistics ``logits_stats_gemm`` left in ``stats``."""
    lib = _lib.load()
    assert logits.dim() == 2 and logits.stride(1) == 1 and logits.shape[0] == b * k
    tile = lib.nm_logits_stats_tile(b * k)
    v = logits.shape[1]
    _lib.check(lib.nm_beam_topk_step_tiles(
        _stream(), logits.data_ptr(), logits.stride(0), stats.data_ptr(), tile, b, k, v,
        logprob_sum.data_ptr(), lengths.data_ptr(), finished.data_ptr(), penalty.data_ptr(), end_id,
        out_score.data_ptr(), out_word.data_ptr(), out_beam.data_ptr(), out_logprob_sum.data_ptr(),
        out_lengths.data_ptr(), out_finished.data_ptr(), out_src_row.data_ptr(), workspace.data_ptr(),
        workspace.numel() * 4, _p(all_finished), rmax.data_ptr(), rlse.data_ptr()), "nm_beam_topk_step_tiles")


def attn_partials_layout(rows, s, a, c):
    """(nchunk, pctx offset, pstat offset) in floats inside the attention workspace, or None when the shape is
    served by the any-shape kernel (no split-S partials)."""
    lib = _lib.load()
    n, po, so = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    if lib.nm_attn_partials_layout(rows, s, a, c, ctypes.byref(n), ctypes.byref(po), ctypes.byref(so)) != 0:
        return None
    return int(n.value), int(po.value), int(so.value)


def attn_fwd_partials(y, hf, states, mask, v, bias, rows_per_key, workspace):
    """The attention step without its combine launch: energies + split-S partials stay in ``workspace``."""
    lib = _lib.load()
    r, a = y.shape
    _, s, c = states.shape
    assert hf.is_contiguous() and states.is_contiguous() and y.is_contiguous()
    _lib.check(lib.nm_attn_fwd_partials(_stream(), y.data_ptr(), hf.data_ptr(), states.data_ptr(), _p(mask),
                                        v.data_ptr(), _p(bias), r, rows_per_key, s, a, c, workspace.data_ptr(),
                                        workspace.numel() * 4), "nm_attn_fwd_partials")


class StepGroup:
    """A launch of ``nm_step_group``: problem descriptors are built once (all pointers are persistent buffers)
    and re-launched every step; ``patch`` changes the per-step pointers (history rows)."""

    def __init__(self, rows, problems):
        lib = _lib.load()
        self.rows = rows
        self.keep = []                       # tensors the descriptors point into
        self.arr = (_lib.StepProblem * len(problems))()
        for slot, spec in zip(self.arr, problems):
            for name, val in spec.items():
                if isinstance(val, torch.Tensor):
                    self.keep.append(val)
                    val = val.data_ptr()
                setattr(slot, name, val)
        self._fn = lib.nm_step_group

    def patch(self, index, **fields):
        for name, val in fields.items():
            setattr(self.arr[index], name, val.data_ptr() if isinstance(val, torch.Tensor) else (val or 0))

    def launch(self):
        _lib.check(self._fn(_stream(), self.rows, self.arr, len(self.arr)), "nm_step_group")


class DecoderStepCall:
    """``nm_decoder_step_fused``: the descriptor of a whole decoder step, built once from persistent buffers;
    ``launch`` patches the per-step pointers (history rows, outputs) and makes the one call."""

    def __init__(self, fields):
        lib = _lib.load()
        self.keep = {}
        self.desc = _lib.DecoderStep()
        self._set(fields)
        self._fn = lib.nm_decoder_step_fused

    def _set(self, fields):
        for name, val in fields.items():
            if isinstance(val, torch.Tensor):
                self.keep[name] = val
                val = val.data_ptr()
            setattr(self.desc, name, val or 0)

    def launch(self, **fields):
        self._set(fields)
        _lib.check(self._fn(_stream(), ctypes.byref(self.desc)), "nm_decoder_step_fused")


def gather_rows(src, idx, dst):
    lib = _lib.load()
    assert src.dim() == 2 and dst.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1
    _lib.check(lib.nm_gather_rows_f32(_stream(), src.data_ptr(), src.stride(0), idx.data_ptr(),
                                      dst.data_ptr(), dst.stride(0), dst.shape[0], dst.shape[1]),
               "nm_gather_rows_f32")


def beam_reorder_tokens(src, src_row, word, dst, steps, rows):
    lib = _lib.load()
    _lib.check(lib.nm_beam_reorder_tokens(_stream(), src.data_ptr(), src_row.data_ptr(),
                                          word.data_ptr(), dst.data_ptr(), steps, rows),
               "nm_beam_reorder_tokens")


def beam_backtrace(src_hist, word_hist, first, out, steps):
    """Token histories [steps+1, R] from the per-step (source row, word) records of a beam search."""
    lib = _lib.load()
    rows = first.numel()
    assert src_hist.is_contiguous() and word_hist.is_contiguous() and out.is_contiguous()
    assert src_hist.shape[1] == rows and word_hist.shape[1] == rows and out.shape[1] == rows and out.shape[0] > steps
    _lib.check(lib.nm_beam_backtrace(_stream(), src_hist.data_ptr(), word_hist.data_ptr(), first.data_ptr(),
                                     out.data_ptr(), steps, rows), "nm_beam_backtrace")


def length_penalty_table(max_len: int, alpha: float, device) -> torch.Tensor:
    """((5+len)/6)**alpha for len in [0, max_len], evaluated in fp32 on the host
    the way the reference's tf.pow sees it (beam_search_decoder.py:561-573)."""
    lens = np.arange(max_len + 1, dtype=np.float32)
    tab = ((np.float32(5.0) + lens) / np.float32(6.0)) ** np.float32(alpha)
    return torch.from_numpy(tab.astype(np.float32)).to(device)


# ---- backward / trainer kernels ------------------------------------------------
def tanh_bwd(dy, y):
    lib = _lib.load()
    assert dy.is_contiguous() and y.is_contiguous() and dy.numel() == y.numel()
    _lib.check(lib.nm_tanh_bwd(_stream(), dy.data_ptr(), y.data_ptr(), dy.numel()), "nm_tanh_bwd")
    return dy


_COLSUM_WS = {}


def _colsum_workspace(device, cols):
    key = (device, cols, _stream(), workspace_tag())
    ws = _COLSUM_WS.get(key)
    if ws is None:
        # zeroed ONCE: the tail holds the arrival counters of the in-kernel final pass, which the kernel leaves at zero
        # (created inside a stream capture when the capture stream is new to this cache: the fill below is then a node
        # of that graph, replayed with it -- the library's fill, not a tensor library's)
        ws = zero(torch.empty(_lib.load().nm_colsum_workspace_bytes(cols) // 4, dtype=torch.float32, device=device))
        _COLSUM_WS[key] = ws
    return ws


def colsum(x, out, accumulate=False):
    """out[c] (+)= sum_r x[r,c] (bias gradients); deterministic."""
    lib = _lib.load()
    assert x.dim() == 2 and x.stride(1) == 1
    cols = x.shape[1]
    ws = _colsum_workspace(x.device, cols)
    # (on a side stream -- runtime.Session.side -- the launch runs beside a time loop: the low-pressure kernel)
    _lib.check(lib.nm_colsum_algo(_stream(), x.data_ptr(), x.stride(0), x.shape[0], cols, out.data_ptr(),
                                  int(accumulate), ws.data_ptr(), ws.numel() * 4,
                                  1 if getattr(_TLS, "on_side", False) else 0), "nm_colsum")
    return out


def embedding_scatter_add(dtable, ids, d, skip_pad=False):
    lib = _lib.load()
    assert d.dim() == 2 and d.stride(1) == 1 and dtable.is_contiguous()
    _lib.check(lib.nm_embedding_scatter_add(_stream(), dtable.data_ptr(), dtable.shape[0], dtable.shape[1],
                                            ids.data_ptr(), ids.numel(), d.data_ptr(), d.stride(0),
                                            int(skip_pad)), "nm_embedding_scatter_add")


_LNB_WS = {}


def layer_norm_bwd_params(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, accumulate=True, accumulate_dx=False):
    """dx of a layer norm and its parameter gradients in one call (nm_layer_norm_bwd_params); ``accumulate_dx``: dx is
    added to what ``dx`` holds."""
    lib = _lib.load()
    d = x.shape[-1]
    rows = x.numel() // d
    assert dy.is_contiguous() and x.is_contiguous() and dx.is_contiguous()
    key = (x.device, d)
    ws = _LNB_WS.get(key)
    if ws is None:
        ws = _LNB_WS[key] = torch.empty(lib.nm_layer_norm_bwd_params_workspace_bytes(d) // 4, dtype=torch.float32,
                                        device=x.device)
    _lib.check(lib.nm_layer_norm_bwd_params(_stream(), dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                            gamma.data_ptr(), dx.data_ptr(), rows, d, dgamma.data_ptr(),
                                            dbeta.data_ptr(), int(bool(accumulate)) | (2 if accumulate_dx else 0),
                                            ws.data_ptr(), ws.numel() * 4), "nm_layer_norm_bwd_params")


def layer_norm_bwd(dy, x, mean, rstd, gamma, dx, dyx):
    lib = _lib.load()
    d = x.shape[-1]
    rows = x.numel() // d
    assert dy.is_contiguous() and x.is_contiguous() and dx.is_contiguous() and dyx.is_contiguous()
    _lib.check(lib.nm_layer_norm_bwd(_stream(), dy.data_ptr(), x.data_ptr(), mean.data_ptr(),
                                     rstd.data_ptr(), gamma.data_ptr(), dx.data_ptr(), dyx.data_ptr(),
                                     rows, d), "nm_layer_norm_bwd")


def gru_step_bwd(phase, dh, dout, dout_strides, ru, c, h0, hseq, hseq_strides, dxp, dxp_strides, dgpre,
                 dcpre, drh, lengths, t, ndir, rows, hsz, reverse_dir0=False):
    """phase 0 = blend backward, phase 1 = gates backward (nm_backward.hip)."""
    lib = _lib.load()
    do = dout_strides or (0, 0, 0)
    _lib.check(lib.nm_gru_step_bwd(_stream(), phase, dh.data_ptr(), _p(dout), do[0], do[1], do[2],
                                   ru.data_ptr(), _p(c), _p(h0), hseq.data_ptr(), hseq_strides[0],
                                   hseq_strides[1], hseq_strides[2], dxp.data_ptr(), dxp_strides[0],
                                   dxp_strides[1], dxp_strides[2], dgpre.data_ptr(), _p(dcpre), _p(drh),
                                   _p(lengths), t, _rev_mask(ndir, reverse_dir0), ndir, rows, hsz),
               "nm_gru_step_bwd")


def gru_seq_shift(seq, out, lengths, ndir, hsz, reverse_dir0=False):
    lib = _lib.load()
    b, s = seq.shape[0], seq.shape[1]
    _lib.check(lib.nm_gru_seq_shift(_stream(), seq.data_ptr(), out.data_ptr(), _p(lengths),
                                    _rev_mask(ndir, reverse_dir0), b, s, ndir, hsz), "nm_gru_seq_shift")


def attn_softmax_bwd(dw, e, mask, de, bsz):
    lib = _lib.load()
    s = dw.shape[-1]
    rows = dw.numel() // s
    _lib.check(lib.nm_attn_softmax_bwd(_stream(), dw.data_ptr(), e.data_ptr(), _p(mask), de.data_ptr(),
                                       rows, bsz, s), "nm_attn_softmax_bwd")


def attn_step_bwd(dctx, states, e, mask, hf, y, v, de, dy):
    """One step's attention backward up to the query in one launch: dctx [B,C] (rows may be strided), states [B,S,C],
    e [B,S] the step's energies, hf [B,S,A], y [B,A] -> de [B,S] (written), dy [B,A] (written)."""
    lib = _lib.load()
    b, s, c = states.shape
    a = hf.shape[2]
    assert states.is_contiguous() and hf.is_contiguous() and e.is_contiguous() and de.is_contiguous()
    assert dctx.stride(1) == 1 and y.stride(1) == 1 and dy.stride(1) == 1
    _lib.check(lib.nm_attn_step_bwd(_stream(), dctx.data_ptr(), dctx.stride(0), states.data_ptr(), e.data_ptr(), _p(mask),
                                    hf.data_ptr(), y.data_ptr(), y.stride(0), v.data_ptr(), de.data_ptr(), dy.data_ptr(),
                                    dy.stride(0), b, s, c, a), "nm_attn_step_bwd")


def attn_step_bwd_ok(dctx, c) -> bool:
    return dctx.is_cuda and c % 4 == 0 and dctx.stride(0) % 4 == 0 and dctx.data_ptr() % 16 == 0


def attn_softmax_fwd(e, mask, w, bsz, rows_per_key=1):
    """w = renorm(softmax(e) * mask) of contiguous [R,S] energies assembled by the caller."""
    lib = _lib.load()
    assert e.is_contiguous() and w.is_contiguous() and e.shape == w.shape
    s = e.shape[-1]
    rows = e.numel() // s
    _lib.check(lib.nm_attn_softmax_fwd(_stream(), e.data_ptr(), _p(mask), w.data_ptr(), rows, bsz, s,
                                       rows_per_key), "nm_attn_softmax_fwd")
    return w


def attn_energy_bwd(de, hf, y, v, dhf, dv_partial, dy, accumulate=False):
    """Energies backward of T queries per sentence (nm_attn_energy_bwd); ``dhf`` and ``dv_partial`` both None: the query
    gradients ``dy`` alone."""
    lib = _lib.load()
    t, b, s = de.shape
    a = hf.shape[-1]
    assert de.is_contiguous() and y.is_contiguous() and dy.is_contiguous()
    _lib.check(lib.nm_attn_energy_bwd(_stream(), de.data_ptr(), hf.data_ptr(), y.data_ptr(), v.data_ptr(),
                                      _p(dhf), _p(dv_partial), dy.data_ptr(), t, b, s, a,
                                      int(accumulate)), "nm_attn_energy_bwd")


# ---- strided element-wise primitives (general / taped path) -------------------------------------
EW = {"copy": 0, "add": 1, "sub": 2, "mul": 3, "scale": 4, "sigmoid": 5, "tanh": 6, "relu": 7,
      "sigmoid_bwd": 8, "tanh_bwd": 9, "relu_bwd": 10, "logaddexp": 11, "add_scalar": 12, "rowscale": 13, "div": 14}


def _rc(t):
    """(rows, cols, ld) of a 1-D / 2-D / contiguous n-D float tensor with unit inner stride."""
    _f32(t)
    if t.dim() == 2:
        assert t.stride(1) == 1 or t.shape[1] == 1
        return t.shape[0], t.shape[1], t.stride(0)
    assert t.is_contiguous(), "n-D operands of element-wise kernels must be contiguous"
    return 1, t.numel(), t.numel()


def zero_if(word, x):
    """``x[:] = 0`` when the int32 device word is not zero (nm_zero_if): the gradient of a step whose time loop gave up
    must not reach an accumulation buffer or a collective as NaNs."""
    _lib.check(_lib.load().nm_zero_if(_stream(), word.data_ptr(), x.data_ptr(), x.numel()), "nm_zero_if")


def fill(x, value=0):
    """``x[:] = value`` for a contiguous float32 / int32 buffer with the library's fill kernel (nm_fill_u32).  Other
    tensors (host tensors, other dtypes, strided views) take torch's fill."""
    if not x.is_cuda or not x.is_contiguous() or x.dtype not in (torch.float32, torch.int32):
        return x.fill_(value)
    if x.dtype == torch.float32:
        pattern = struct.unpack("<I", struct.pack("<f", float(value)))[0]
    else:
        pattern = int(value) & 0xFFFFFFFF
    _lib.check(_lib.load().nm_fill_u32(_stream(), x.data_ptr(), x.numel(), pattern), "nm_fill_u32")
    return x


def zero(x):
    return fill(x, 0)


def copy(dst, src):
    """``dst[:] = src`` for two contiguous device buffers of one dtype and size as a runtime device-to-device copy
    (nm_copy_d2d: a memcpy node inside a captured graph); anything else takes torch's copy_."""
    if (dst.is_cuda and src.is_cuda and dst.dtype == src.dtype and dst.numel() == src.numel() and dst.is_contiguous()
            and src.is_contiguous() and dst.device == src.device):
        _lib.check(_lib.load().nm_copy_d2d(_stream(), dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size()),
                   "nm_copy_d2d")
        return dst
    return dst.copy_(src)


def ew(op, a, b, out, alpha=0.0, accumulate=False):
    """out (+)= op(a, b) element-wise; 2-D operands may be column slices (row stride = ld)."""
    lib = _lib.load()
    rows, cols, lda = _rc(a)
    ro, co, ldo = _rc(out)
    assert (rows, cols) == (ro, co), (a.shape, out.shape)
    ldb = 0
    if b is not None:
        rb, cb, ldb = _rc(b)
        assert (rb, cb) == ((rows, 1) if op == "rowscale" else (rows, cols)), (a.shape, b.shape)
    _lib.check(lib.nm_ew(_stream(), EW[op], a.data_ptr(), lda, _p(b), ldb, out.data_ptr(), ldo, rows, cols,
                         float(alpha), int(accumulate)), "nm_ew")
    return out


def blend_fwd(u, h, c, out):
    lib = _lib.load()
    rows, cols, ldu = _rc(u)
    _lib.check(lib.nm_blend_fwd(_stream(), u.data_ptr(), ldu, h.data_ptr(), _rc(h)[2], c.data_ptr(), _rc(c)[2],
                                out.data_ptr(), _rc(out)[2], rows, cols), "nm_blend_fwd")
    return out


def lstm_cell_fwd(z, c_prev, c_new, h_new, gates=None, forget_bias=1.0):
    """One LSTMCell step after its products: z [R,4H] (i, j, f, o) -> c', h' (+ the activated gates)."""
    lib = _lib.load()
    rows, h = c_prev.shape
    assert z.shape == (rows, 4 * h) and z.stride(1) == 1
    _lib.check(lib.nm_lstm_cell_fwd(_stream(), z.data_ptr(), z.stride(0), c_prev.data_ptr(), _rc(c_prev)[2],
                                    c_new.data_ptr(), _rc(c_new)[2], h_new.data_ptr(), _rc(h_new)[2], _p(gates),
                                    0 if gates is None else gates.stride(0), rows, h, float(forget_bias)),
               "nm_lstm_cell_fwd")


def lstm_cell_bwd(dh, dc_new, gates, c_prev, c_new, dz, dc_prev, accumulate_dz=False, accumulate_dc_prev=False):
    lib = _lib.load()
    rows, h = c_prev.shape
    ld = lambda t: 0 if t is None else _rc(t)[2]
    _lib.check(lib.nm_lstm_cell_bwd(_stream(), _p(dh), ld(dh), _p(dc_new), ld(dc_new), gates.data_ptr(),
                                    gates.stride(0), c_prev.data_ptr(), ld(c_prev), c_new.data_ptr(), ld(c_new),
                                    dz.data_ptr(), dz.stride(0), _p(dc_prev), ld(dc_prev), rows, h,
                                    int(accumulate_dz), int(accumulate_dc_prev)), "nm_lstm_cell_bwd")


def nematus_state_step_ok(h_prev, w_st, h_new) -> bool:
    """The one-launch step takes it, and pays: up to 512 tiles of 16 rows x 16 units (two per compute unit).  Beyond
    that -- 640 beam rows x 512 units are 1280 -- the product on large tiles and the point-wise launch are faster
    (beam-5 of the general path 24.6 against 25.0 ms per batch)."""
    rows, h = h_prev.shape
    if ((rows + 15) // 16) * ((h + 15) // 16) > 512 or h % 8:
        return False
    return (h_prev.is_cuda and h_prev.stride(1) == 1 and h_prev.stride(0) % 4 == 0 and h_prev.data_ptr() % 16 == 0
            and w_st.stride(1) == 1 and h_new.stride(1) == 1 and h_new.data_ptr() != h_prev.data_ptr())


def nematus_state_step(h_prev, w_st, b_st, x_all, h_new, ru=None, c_out=None, sc_out=None):
    """h_new = NematusGRUCell step from the previous state [R,H], the state kernels [H,3H] = [U_g | U_c] (+ bias [3H]) and
    the projected input half x_all [R,3H]: product and point-wise part in ONE launch (nm_nematus_state_step).  ru [R,2H],
    c_out [R,H] (contiguous) and sc_out [R,H] (row stride free) keep what the backward pass reads."""
    lib = _lib.load()
    rows, h = h_prev.shape
    assert w_st.shape == (h, 3 * h) and x_all.shape == (rows, 3 * h) and h_new.shape == (rows, h) and x_all.stride(1) == 1
    assert ru is None or ru.is_contiguous()
    assert c_out is None or c_out.is_contiguous()
    assert sc_out is None or sc_out.stride(1) == 1
    _lib.check(lib.nm_nematus_state_step(_stream(), h_prev.data_ptr(), h_prev.stride(0), w_st.data_ptr(), w_st.stride(0),
                                         _p(b_st), x_all.data_ptr(), x_all.stride(0), h_new.data_ptr(), h_new.stride(0),
                                         _p(ru), _p(c_out), _p(sc_out), 0 if sc_out is None else sc_out.stride(0), rows, h),
               "nm_nematus_state_step")


def nematus_full_step_ok(x, w_in) -> bool:
    """(beside nematus_state_step_ok) the step's input rows can be multiplied in the same launch"""
    d = x.shape[1]
    return (d % 8 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and w_in.stride(1) == 1)


def nematus_full_step(h_prev, w_st, b_st, x, w_in, b_in, h_new, ru=None, c_out=None, sc_out=None):
    """nematus_state_step with the input half x [R,D] . w_in [D,3H] (+ b_in) computed in the same launch."""
    lib = _lib.load()
    rows, h = h_prev.shape
    d = x.shape[1]
    assert w_st.shape == (h, 3 * h) and w_in.shape == (d, 3 * h) and x.shape[0] == rows and h_new.shape == (rows, h)
    assert ru is None or ru.is_contiguous()
    assert c_out is None or c_out.is_contiguous()
    assert sc_out is None or sc_out.stride(1) == 1
    _lib.check(lib.nm_nematus_full_step(_stream(), h_prev.data_ptr(), h_prev.stride(0), w_st.data_ptr(), w_st.stride(0),
                                        _p(b_st), x.data_ptr(), x.stride(0), w_in.data_ptr(), w_in.stride(0), _p(b_in),
                                        h_new.data_ptr(), h_new.stride(0), _p(ru), _p(c_out), _p(sc_out),
                                        0 if sc_out is None else sc_out.stride(0), rows, h, d), "nm_nematus_full_step")


def nematus_cell_fwd(g_pre, sc, ci, h_prev, h_new, ru=None, c_out=None, g2=None):
    """h' of one NematusGRUCell step from its products (nm_nematus_cell_fwd); ``ru`` [R,2H] / ``c_out`` [R,H]: contiguous
    buffers for the backward call; ``g2``: a second gate operand added to ``g_pre``."""
    rows, h = h_prev.shape
    assert g_pre.shape == (rows, 2 * h) and (ru is None or ru.is_contiguous()) and (c_out is None or c_out.is_contiguous())
    _lib.check(_lib.load().nm_nematus_cell_fwd(_stream(), g_pre.data_ptr(), _rc(g_pre)[2], sc.data_ptr(), _rc(sc)[2],
                                               ci.data_ptr(), _rc(ci)[2], h_prev.data_ptr(), _rc(h_prev)[2],
                                               h_new.data_ptr(), _rc(h_new)[2], _p(ru), _p(c_out), _p(g2),
                                               0 if g2 is None else _rc(g2)[2], rows, h), "nm_nematus_cell_fwd")
    return h_new


def nematus_cell_bwd(dh, ru, c, sc, h_prev, dg, dci, dsc, dh_prev, acc_dg=False, acc_dci=False, acc_dsc=False,
                     acc_dh_prev=False, dg2=None):
    rows, h = h_prev.shape
    ld = lambda t: 0 if t is None else _rc(t)[2]
    _lib.check(_lib.load().nm_nematus_cell_bwd(_stream(), dh.data_ptr(), ld(dh), ru.data_ptr(), c.data_ptr(), sc.data_ptr(),
                                               ld(sc), h_prev.data_ptr(), ld(h_prev), dg.data_ptr(), ld(dg), _p(dci), ld(dci),
                                               _p(dsc), ld(dsc), _p(dh_prev), ld(dh_prev), _p(dg2), ld(dg2), rows, h, int(acc_dg),
                                               int(acc_dci), int(acc_dsc), int(acc_dh_prev)), "nm_nematus_cell_bwd")


def blend_bwd(dy, u, h, c, du, dh, dc):
    lib = _lib.load()
    rows, cols, ldu = _rc(u)
    ld = lambda t: 0 if t is None else _rc(t)[2]
    _lib.check(lib.nm_blend_bwd(_stream(), dy.data_ptr(), _rc(dy)[2], u.data_ptr(), ldu, h.data_ptr(), ld(h),
                                c.data_ptr(), ld(c), _p(du), ld(du), _p(dh), ld(dh), _p(dc), ld(dc), rows, cols),
               "nm_blend_bwd")


def dropout_salt(*parts) -> int:
    """32-bit salt of a dropout call site: crc32 of the "/"-joined parts (restated by the CPU checker in tests)."""
    import zlib
    return zlib.crc32("/".join(str(p) for p in parts).encode()) & 0xFFFFFFFF


def dropout(x, out, keep_prob, salt, accumulate=False, step=None):
    """``step``: optional int32 device scalar (global step) that advances the salt on the device."""
    lib = _lib.load()
    rows, cols, ldx = _rc(x)
    _lib.check(lib.nm_dropout(_stream(), x.data_ptr(), ldx, out.data_ptr(), _rc(out)[2], rows, cols,
                              float(keep_prob), int(salt) & 0xFFFFFFFF, _p(step), int(accumulate)), "nm_dropout")
    return out


def rnn_select_fwd(h_new, h_prev, lengths, t, h_out, y_out):
    lib = _lib.load()
    rows, cols, ldn = _rc(h_new)
    _lib.check(lib.nm_rnn_select_fwd(_stream(), h_new.data_ptr(), ldn, h_prev.data_ptr(), _rc(h_prev)[2],
                                     _p(lengths), t, h_out.data_ptr(), _rc(h_out)[2], _p(y_out),
                                     0 if y_out is None else _rc(y_out)[2], rows, cols), "nm_rnn_select_fwd")


def rnn_select_bwd(dh, dy, lengths, t, d_new, d_prev):
    lib = _lib.load()
    rows, cols, ldn = _rc(d_new)
    ld = lambda x: 0 if x is None else _rc(x)[2]
    _lib.check(lib.nm_rnn_select_bwd(_stream(), _p(dh), ld(dh), _p(dy), ld(dy), _p(lengths), t, d_new.data_ptr(),
                                     ldn, _p(d_prev), ld(d_prev), rows, cols), "nm_rnn_select_bwd")


def reverse_sequence(x, out, lengths, accumulate=False):
    lib = _lib.load()
    b, s, d = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.shape == x.shape
    _lib.check(lib.nm_reverse_sequence(_stream(), x.data_ptr(), out.data_ptr(), _i32(lengths).data_ptr(), b, s, d,
                                       int(accumulate)), "nm_reverse_sequence")
    return out


def maxout_fwd(x, out, argmax, pool=2):
    lib = _lib.load()
    rows, cols, ldx = _rc(x)
    groups = cols // pool
    assert groups * pool == cols and out.shape[1] == groups
    _lib.check(lib.nm_maxout_fwd(_stream(), x.data_ptr(), ldx, out.data_ptr(), _rc(out)[2], _p(argmax), rows,
                                 groups, pool), "nm_maxout_fwd")
    return out


def _bs(t):
    """Batch stride (floats) of a [B, T, D] tensor whose rows are contiguous."""
    assert t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == t.shape[2], (t.shape, t.stride())
    return t.stride(0)


def sdp_attn_fwd(q, k, v, key_mask, heads, ctx, weights=None, causal=False, rows_per_key=1, keep_prob=1.0,
                 salt=0, step=None):
    """Multi-head scaled dot-product attention.  q/ctx [Bq,Tq,D], k/v [Bk,Tk,D] (Bq = Bk*rows_per_key;
    batch strides may exceed T*D: a key/value cache), key_mask [Bk,Tk] or None, weights [Bq,H,Tq,Tk]."""
    lib = _lib.load()
    bq, tq, d = q.shape
    bk, tk, _ = k.shape
    assert d % heads == 0 and bq == bk * rows_per_key and v.shape[1] == tk
    mask_bs = 0
    if key_mask is not None:
        assert key_mask.dim() == 2 and key_mask.stride(1) == 1 and key_mask.shape[1] >= tk
        mask_bs = key_mask.stride(0)
    if weights is not None:
        assert weights.is_contiguous() and weights.numel() == bq * heads * tq * tk
    _lib.check(lib.nm_sdp_attn_fwd(_stream(), q.data_ptr(), _bs(q), k.data_ptr(), _bs(k), v.data_ptr(), _bs(v),
                                   _p(key_mask), mask_bs, bq, rows_per_key, tq, tk, heads, d // heads, int(causal),
                                   float(keep_prob), int(salt) & 0xFFFFFFFF, _p(step), ctx.data_ptr(), _bs(ctx),
                                   _p(weights)), "nm_sdp_attn_fwd")
    return ctx


def sdp_attn_step(q, k, v, key_mask, heads, ancestors, ctx, weights=None):
    """One cached decoding step: q/ctx [R,1,D]; k/v [R,Tk,D] views of the caches; position j of row r is read from
    cache row ``ancestors[r, j]`` (int32 [R, >=Tk])."""
    lib = _lib.load()
    rows, tq, d = q.shape
    tk = k.shape[1]
    assert tq == 1 and d % heads == 0 and ancestors.dtype == torch.int32 and ancestors.stride(1) == 1
    assert ancestors.shape[0] == rows and ancestors.shape[1] >= tk
    mask_bs = 0
    if key_mask is not None:
        assert key_mask.dim() == 2 and key_mask.stride(1) == 1 and key_mask.shape[1] >= tk
        mask_bs = key_mask.stride(0)
    _lib.check(lib.nm_sdp_attn_step(_stream(), q.data_ptr(), _bs(q), k.data_ptr(), _bs(k), v.data_ptr(), _bs(v),
                                    _p(key_mask), mask_bs, rows, tk, heads, d // heads, ancestors.data_ptr(),
                                    ancestors.stride(0), ctx.data_ptr(), _bs(ctx), _p(weights)), "nm_sdp_attn_step")
    return ctx


def sdp_attn_bwd(q, k, v, key_mask, weights, dctx, heads, dq, dk, dv, de_ws, causal=False, keep_prob=1.0, salt=0,
                 accumulate=False, step=None):
    lib = _lib.load()
    b, tq, d = q.shape
    tk = k.shape[1]
    mask_bs = 0 if key_mask is None else key_mask.stride(0)
    assert de_ws.numel() >= b * heads * tq * tk
    _lib.check(lib.nm_sdp_attn_bwd(_stream(), q.data_ptr(), _bs(q), k.data_ptr(), _bs(k), v.data_ptr(), _bs(v),
                                   _p(key_mask), mask_bs, weights.data_ptr(), dctx.data_ptr(), _bs(dctx), b, tq, tk,
                                   heads, d // heads, int(causal), float(keep_prob), int(salt) & 0xFFFFFFFF,
                                   _p(step), dq.data_ptr(), _bs(dq), dk.data_ptr(), _bs(dk), dv.data_ptr(), _bs(dv),
                                   de_ws.data_ptr(), int(accumulate)), "nm_sdp_attn_bwd")


def add_position(x, signal, out, t0=0):
    """out[b,t,:] = x[b,t,:] + signal[t0+t,:]; x [B,T,D] contiguous, signal [Tmax,D]."""
    lib = _lib.load()
    b, t, d = x.shape
    assert x.is_contiguous() and out.is_contiguous() and signal.is_contiguous()
    assert signal.shape[1] == d and signal.shape[0] >= t0 + t
    _lib.check(lib.nm_add_position(_stream(), x.data_ptr(), signal.data_ptr(), out.data_ptr(), b, t, d, t0),
               "nm_add_position")
    return out


def unfinished_mask(finished, out_col):
    """out_col[r] = 0.0 where finished[r] else 1.0; ``out_col`` may be a strided column view."""
    lib = _lib.load()
    n = finished.numel()
    assert out_col.dim() == 1 and out_col.numel() == n
    _lib.check(lib.nm_unfinished_mask(_stream(), _i32(finished).data_ptr(), out_col.data_ptr(),
                                      out_col.stride(0) if n > 1 else 1, n), "nm_unfinished_mask")


def time_sum(x, out):
    lib = _lib.load()
    b, t, d = x.shape
    assert x.is_contiguous() and out.is_contiguous()
    _lib.check(lib.nm_time_sum(_stream(), x.data_ptr(), out.data_ptr(), b, t, d), "nm_time_sum")
    return out


def time_bcast_add(dy, dx):
    lib = _lib.load()
    b, t, d = dx.shape
    assert dy.is_contiguous() and dx.is_contiguous()
    _lib.check(lib.nm_time_bcast_add(_stream(), dy.data_ptr(), dx.data_ptr(), b, t, d), "nm_time_bcast_add")


def maxout_bwd(dy, argmax, dx, pool=2):
    lib = _lib.load()
    rows, groups, lddy = _rc(dy)
    _lib.check(lib.nm_maxout_bwd(_stream(), dy.data_ptr(), lddy, argmax.data_ptr(), dx.data_ptr(), _rc(dx)[2],
                                 rows, groups, pool), "nm_maxout_bwd")


def gru_gemm(mode, a, b, trans_b, t, ndir, rows, hsz, lengths=None, reverse_dir0=False, xp=None,
             x_strides=(0, 0, 0), h_in=None, h_out=None, ru=None, rh=None, c_save=None, out=None,
             out_strides=(0, 0, 0), dh=None, dout=None, dout_strides=(0, 0, 0), c=None, h0=None, hseq=None,
             hseq_strides=(0, 0, 0), dxp=None, dxp_strides=(0, 0, 0), dgpre=None, dcpre=None):
    """Recurrent GEMM of one GRU step with the fused epilogue ``mode`` (nm_gru_gemm).
    a: [ndir,R,K] (or [R,K]); b: [ndir,K,N] / [ndir,N,K] when trans_b (or 2-D for ndir == 1)."""
    lib = _lib.load()
    e = _lib.GruEpilogue()
    e.mode, e.t, e.rev_mask, e.ndir, e.R, e.H = mode, t, _rev_mask(ndir, reverse_dir0), ndir, rows, hsz
    e.lengths = _p(lengths)
    e.xp = _p(xp)
    e.x_dir, e.x_row, e.x_time = x_strides
    e.h_in, e.h_out, e.ru, e.rh, e.c_save, e.out = _p(h_in), _p(h_out), _p(ru), _p(rh), _p(c_save), _p(out)
    e.o_dir, e.o_row, e.o_time = out_strides
    e.dh, e.dout = _p(dh), _p(dout)
    e.do_dir, e.do_row, e.do_time = dout_strides
    e.c, e.h0, e.hseq = _p(c), _p(h0), _p(hseq)
    e.hs_dir, e.hs_row, e.hs_time = hseq_strides
    e.dxp = _p(dxp)
    e.dx_dir, e.dx_row, e.dx_time = dxp_strides
    e.dgpre, e.dcpre = _p(dgpre), _p(dcpre)
    a2 = a[0] if a.dim() == 3 else a
    b2 = b[0] if b.dim() == 3 else b
    assert a2.stride(1) == 1 and b2.stride(1) == 1
    k = a2.shape[1]
    s_a = a.stride(0) if a.dim() == 3 else 0
    s_b = b.stride(0) if b.dim() == 3 else 0
    _lib.check(lib.nm_gru_gemm(_stream(), ctypes.byref(e), int(trans_b), k, a.data_ptr(), a2.stride(0), s_a,
                               b.data_ptr(), b2.stride(0), s_b), "nm_gru_gemm")


def gru_seq_supported(rows, hsz, ndir) -> bool:
    """Can the time loops of this shape run as one cluster launch each (nm_gru_seq_fwd / nm_gru_seq_bwd)?"""
    return bool(_lib.load().nm_gru_seq_supported(rows, hsz, ndir))


def gru_seq_workspace_floats(rows, hsz, ndir) -> int:
    return _lib.load().nm_gru_seq_workspace_bytes(rows, hsz, ndir) // 4


def gru_seq_workspace(rows, hsz, ndir, device) -> torch.Tensor:
    """Header + granule buffers of one cluster loop (the call zeroes what it needs)."""
    return torch.empty(gru_seq_workspace_floats(rows, hsz, ndir), dtype=torch.float32, device=device)


def gru_seq_force_give_up(launches: int) -> int:
    """Test hook (nm_gru_seq_force_give_up): the next ``launches`` cluster loops behave like loops whose hand-offs
    timed out."""
    return int(_lib.load().nm_gru_seq_force_give_up(int(launches)))


def gru_seq_test_hog(blocks: int, lds_bytes: int, microseconds: int, stream=None) -> None:
    """Test utility (nm_gru_seq_test_hog): ``blocks`` workgroups that hold ``lds_bytes`` of LDS each for a while."""
    st = stream.cuda_stream if stream is not None else _stream()
    _lib.check(_lib.load().nm_gru_seq_test_hog(st, int(blocks), int(lds_bytes), int(microseconds)), "nm_gru_seq_test_hog")


def gru_seq_failed(workspace) -> bool:
    """After a synchronisation: did a cluster loop that used ``workspace`` give up waiting for a hand-off?"""
    return _lib.load().nm_gru_seq_failed(workspace.data_ptr()) != 0


# One cluster loop at a time per device.  A launch is ``ncu`` workgroups that must ALL be resident before they agree on
# their roles (csrc/nm_gru_cluster.hip); by registers exactly one such workgroup fits a CU (8 waves x 160 VGPRs = 2 per
# SIMD x 160 of 512), so two loops launched on two streams interleave over the CUs and neither becomes resident until
# both time out.  Every launch therefore waits for the previous one's event, whatever its stream, and leaves its own.
# Inside a stream capture the order is the captured stream's (a graph holds one loop after the other anyway).
_LAST_CLUSTER_LAUNCH = {}          # device index -> (event, stream it was recorded on)


def _cluster_serialize_before():
    if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
        return None
    stream = torch.cuda.current_stream()
    last = _LAST_CLUSTER_LAUNCH.get(stream.device.index)
    if last is not None and last[1] != stream.cuda_stream:
        stream.wait_event(last[0])
    return stream


def _cluster_serialize_after(stream):
    if stream is None:
        return
    last = _LAST_CLUSTER_LAUNCH.get(stream.device.index)
    event = last[0] if last is not None else torch.cuda.Event()
    event.record(stream)
    _LAST_CLUSTER_LAUNCH[stream.device.index] = (event, stream.cuda_stream)


def gru_seq_fwd(steps, ndir, rows, hsz, xp, x_strides, h_in0, h_out0, h_step, ru0, ru_step, rh0, rh_step, c0,
                c_step, wgh, wch, workspace, lengths=None, reverse_dir0=False, out=None, out_strides=(0, 0, 0),
                sticky=None):
    """All ``steps`` forward GRU steps in one launch (nm_gru_seq_fwd: workgroup clusters, csrc/nm_gru_cluster.hip).
    Tensors of step t: h_out0 + t*h_step etc. (element strides); ``rh0`` may be None; wgh [ndir,H,2H], wch [ndir,H,H]
    (2-D accepted for ndir 1); ``workspace`` from ``gru_seq_workspace``; ``sticky``: an int32 device word that a launch
    which gave up waiting sets to 1 (runtime.Session.error_word)."""
    lib = _lib.load()
    e = _lib.GruEpilogue()
    e.mode, e.t, e.rev_mask, e.ndir, e.R, e.H = 1, 0, _rev_mask(ndir, reverse_dir0), ndir, rows, hsz
    e.lengths = _p(lengths)
    e.xp = xp.data_ptr()
    e.x_dir, e.x_row, e.x_time = x_strides
    e.h_in, e.h_out, e.ru, e.rh, e.c_save, e.out = (h_in0.data_ptr(), h_out0.data_ptr(), ru0.data_ptr(),
                                                   _p(rh0), _p(c0), _p(out))
    e.o_dir, e.o_row, e.o_time = out_strides
    g2 = wgh[0] if wgh.dim() == 3 else wgh
    c2 = wch[0] if wch.dim() == 3 else wch
    assert g2.stride(1) == 1 and c2.stride(1) == 1
    serial = _cluster_serialize_before()
    _lib.check(lib.nm_gru_seq_fwd(_stream(), ctypes.byref(e), steps, h_step, ru_step, rh_step, c_step,
                                  wgh.data_ptr(), g2.stride(0), wgh.stride(0) if wgh.dim() == 3 else 0,
                                  wch.data_ptr(), c2.stride(0), wch.stride(0) if wch.dim() == 3 else 0,
                                  workspace.data_ptr(), workspace.numel() * workspace.element_size(), _p(sticky)),
               "nm_gru_seq_fwd")
    _cluster_serialize_after(serial)


def gru_seq_bwd(steps, ndir, rows, hsz, dh, dout, dout_strides, ru0, ru_step, c0, c_step, h0, hseq, hseq_strides,
                dxp, dxp_strides, wgh, wch, workspace, lengths=None, reverse_dir0=False, sticky=None):
    """The whole BPTT loop in one launch (nm_gru_seq_bwd): ``dh`` [ndir,R,H] holds dL/dh after the last step on entry
    and dL/dh_0 on exit; ``ru0`` / ``c0`` are the gates / candidates of step 0 (step t at + t*step elements);
    pre-activation gradients land in ``dxp``; wgh [ndir,H,2H], wch [ndir,H,H] (2-D accepted for ndir 1)."""
    lib = _lib.load()
    e = _lib.GruEpilogue()
    e.mode, e.t, e.rev_mask, e.ndir, e.R, e.H = 4, 0, _rev_mask(ndir, reverse_dir0), ndir, rows, hsz
    e.lengths = _p(lengths)
    e.dh, e.dout = dh.data_ptr(), _p(dout)
    e.do_dir, e.do_row, e.do_time = dout_strides or (0, 0, 0)
    e.ru, e.c, e.h0, e.hseq = ru0.data_ptr(), c0.data_ptr(), _p(h0), hseq.data_ptr()
    e.hs_dir, e.hs_row, e.hs_time = hseq_strides
    e.dxp = dxp.data_ptr()
    e.dx_dir, e.dx_row, e.dx_time = dxp_strides
    g2 = wgh[0] if wgh.dim() == 3 else wgh
    c2 = wch[0] if wch.dim() == 3 else wch
    assert g2.stride(1) == 1 and c2.stride(1) == 1
    serial = _cluster_serialize_before()
    _lib.check(lib.nm_gru_seq_bwd(_stream(), ctypes.byref(e), steps, ru_step, c_step,
                                  wgh.data_ptr(), g2.stride(0), wgh.stride(0) if wgh.dim() == 3 else 0,
                                  wch.data_ptr(), c2.stride(0), wch.stride(0) if wch.dim() == 3 else 0,
                                  workspace.data_ptr(), workspace.numel() * workspace.element_size(), _p(sticky)),
               "nm_gru_seq_bwd")
    _cluster_serialize_after(serial)


def lstm_seq_workspace_floats(rows, hsz, ndir) -> int:
    return _lib.load().nm_lstm_seq_workspace_bytes(rows, hsz, ndir) // 4


def lstm_seq_fwd(steps, ndir, rows, hsz, xp, x_strides, h_in0, h_out0, h_step, gates0, g_step, c0, c_step, wh, workspace,
                 forget_bias=1.0, lengths=None, reverse_dir0=False, out=None, out_strides=(0, 0, 0), sticky=None):
    """All ``steps`` forward steps of an LSTMCell layer in one launch (nm_lstm_seq_fwd).  ``xp`` 4H wide per direction
    (i | j | f | o), ``wh`` [ndir,H,4H]: the state half of the kernel, ``gates0`` / ``c0``: where step 0's activated gates
    and cell state are saved (step t at + t * step elements); zero initial cell state."""
    lib = _lib.load()
    e = _lib.GruEpilogue()
    e.mode, e.t, e.rev_mask, e.ndir, e.R, e.H = 1, 0, _rev_mask(ndir, reverse_dir0), ndir, rows, hsz
    e.lengths = _p(lengths)
    e.xp = xp.data_ptr()
    e.x_dir, e.x_row, e.x_time = x_strides
    e.h_in, e.h_out, e.ru, e.c_save, e.out = h_in0.data_ptr(), h_out0.data_ptr(), gates0.data_ptr(), c0.data_ptr(), _p(out)
    e.o_dir, e.o_row, e.o_time = out_strides
    ld_w, s_w = _dir_strides(wh)
    serial = _cluster_serialize_before()
    _lib.check(lib.nm_lstm_seq_fwd(_stream(), ctypes.byref(e), steps, h_step, g_step, c_step, wh.data_ptr(), ld_w, s_w,
                                   float(forget_bias), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                   _p(sticky)), "nm_lstm_seq_fwd")
    _cluster_serialize_after(serial)


def lstm_seq_bwd(steps, ndir, rows, hsz, dh, dout, dout_strides, gates0, g_step, c0, c_step, dxp, dxp_strides, wh,
                 workspace, lengths=None, reverse_dir0=False, sticky=None):
    """The BPTT loop of an LSTMCell layer in one launch (nm_lstm_seq_bwd): ``dh`` [ndir,R,H] holds dL/dh after the last
    step on entry and dL/dh_0 on exit; ``dxp`` (4H wide per direction) receives the pre-activation gradients."""
    lib = _lib.load()
    e = _lib.GruEpilogue()
    e.mode, e.t, e.rev_mask, e.ndir, e.R, e.H = 4, 0, _rev_mask(ndir, reverse_dir0), ndir, rows, hsz
    e.lengths = _p(lengths)
    e.dh, e.dout = dh.data_ptr(), _p(dout)
    e.do_dir, e.do_row, e.do_time = dout_strides or (0, 0, 0)
    e.ru, e.c = gates0.data_ptr(), c0.data_ptr()
    e.dxp = dxp.data_ptr()
    e.dx_dir, e.dx_row, e.dx_time = dxp_strides
    ld_w, s_w = _dir_strides(wh)
    serial = _cluster_serialize_before()
    _lib.check(lib.nm_lstm_seq_bwd(_stream(), ctypes.byref(e), steps, g_step, c_step, wh.data_ptr(), ld_w, s_w,
                                   workspace.data_ptr(), workspace.numel() * workspace.element_size(), _p(sticky)),
               "nm_lstm_seq_bwd")
    _cluster_serialize_after(serial)


def nematus_seq_workspace_floats(rows, hsz, ndir) -> int:
    return _lib.load().nm_nematus_seq_workspace_bytes(rows, hsz, ndir) // 4


def _dir_strides(w):
    w2 = w[0] if w.dim() == 3 else w
    assert w2.stride(1) == 1
    return w2.stride(0), (w.stride(0) if w.dim() == 3 else 0)


def nematus_seq_fwd(steps, ndir, rows, hsz, xp, x_strides, h_in0, h_out0, h_step, ru0, ru_step, sc0, sc_step, c0,
                    c_step, ug, uc, workspace, bgs=None, bcs=None, lengths=None, reverse_dir0=False, out=None,
                    out_strides=(0, 0, 0), sticky=None):
    """All ``steps`` forward steps of a NematusGRUCell layer in one launch (nm_nematus_seq_fwd).  As ``gru_seq_fwd``;
    ``ug`` [ndir,H,2H] / ``uc`` [ndir,H,H]: the state projections, ``bgs`` [ndir,2H] / ``bcs`` [ndir,H] their biases,
    ``sc0``: where step 0's h.U_c + b_cs is saved."""
    lib = _lib.load()
    e = _lib.GruEpilogue()
    e.mode, e.t, e.rev_mask, e.ndir, e.R, e.H = 1, 0, _rev_mask(ndir, reverse_dir0), ndir, rows, hsz
    e.lengths = _p(lengths)
    e.xp = xp.data_ptr()
    e.x_dir, e.x_row, e.x_time = x_strides
    e.h_in, e.h_out, e.ru, e.rh, e.c_save, e.out = (h_in0.data_ptr(), h_out0.data_ptr(), ru0.data_ptr(),
                                                   sc0.data_ptr(), c0.data_ptr(), _p(out))
    e.o_dir, e.o_row, e.o_time = out_strides
    ld_g, s_g = _dir_strides(ug)
    ld_c, s_c = _dir_strides(uc)
    serial = _cluster_serialize_before()
    _lib.check(lib.nm_nematus_seq_fwd(_stream(), ctypes.byref(e), steps, h_step, ru_step, sc_step, c_step,
                                      ug.data_ptr(), ld_g, s_g, uc.data_ptr(), ld_c, s_c, _p(bgs), _p(bcs),
                                      workspace.data_ptr(), workspace.numel() * workspace.element_size(), _p(sticky)),
               "nm_nematus_seq_fwd")
    _cluster_serialize_after(serial)


def nematus_seq_bwd(steps, ndir, rows, hsz, dh, dout, dout_strides, ru0, ru_step, sc0, sc_step, c0, c_step, h0, hseq,
                    hseq_strides, dxp, dxp_strides, ug, uc, workspace, lengths=None, reverse_dir0=False, sticky=None):
    """The BPTT loop of a NematusGRUCell layer in one launch (nm_nematus_seq_bwd); ``dxp`` is 4H wide per direction:
    [dr' | du' | dc' | dsc]."""
    lib = _lib.load()
    e = _lib.GruEpilogue()
    e.mode, e.t, e.rev_mask, e.ndir, e.R, e.H = 4, 0, _rev_mask(ndir, reverse_dir0), ndir, rows, hsz
    e.lengths = _p(lengths)
    e.dh, e.dout = dh.data_ptr(), _p(dout)
    e.do_dir, e.do_row, e.do_time = dout_strides or (0, 0, 0)
    e.ru, e.rh, e.c, e.h0, e.hseq = ru0.data_ptr(), sc0.data_ptr(), c0.data_ptr(), _p(h0), hseq.data_ptr()
    e.hs_dir, e.hs_row, e.hs_time = hseq_strides
    e.dxp = dxp.data_ptr()
    e.dx_dir, e.dx_row, e.dx_time = dxp_strides
    ld_g, s_g = _dir_strides(ug)
    ld_c, s_c = _dir_strides(uc)
    serial = _cluster_serialize_before()
    _lib.check(lib.nm_nematus_seq_bwd(_stream(), ctypes.byref(e), steps, ru_step, sc_step, c_step,
                                      ug.data_ptr(), ld_g, s_g, uc.data_ptr(), ld_c, s_c,
                                      workspace.data_ptr(), workspace.numel() * workspace.element_size(), _p(sticky)),
               "nm_nematus_seq_bwd")
    _cluster_serialize_after(serial)


def optimizer_chunk_table(store, regularizable, trainable, cuts=(), chunk=65536):
    """The host side of the flat optimizer kernels' tables: every variable ("segment") is cut into chunks of at most
    ``chunk`` elements -- and at every flat offset in ``cuts`` (the slice boundaries of a sharded optimizer,
    distributed.ShardPlan.cuts: a chunk never straddles one, so every rank owns whole chunks).  The table -- and with it
    the order of every reduction -- depends on the cuts only, not on who owns which slice: one process and N ranks that
    use the same cuts compute bit-identical norms.  Returns (starts, lens, segs, seg_first, seg_count, seg_flags)."""
    import bisect
    cuts = sorted(set(int(c) for c in cuts))
    starts, lens, segs, first, count, flags = [], [], [], [], [], []
    for si, (name, spec) in enumerate(store.specs.items()):
        first.append(len(starts))
        off = 0
        while off < spec.size:
            n = min(chunk, spec.size - off)
            if cuts:
                k = bisect.bisect_right(cuts, spec.offset + off)
                if k < len(cuts) and cuts[k] < spec.offset + off + n:
                    n = cuts[k] - (spec.offset + off)
            starts.append(spec.offset + off)
            lens.append(n)
            segs.append(si)
            off += n
        count.append(len(starts) - first[-1])
        flags.append((1 if name in regularizable else 0) | (2 if name in trainable else 0))
    return starts, lens, segs, first, count, flags


def chunk_range_of(starts, lens, lo, hi):
    """The chunks whose elements lie in [lo, hi) of the flat buffer: (begin, end).  No chunk may straddle either end
    (the table was cut there: ``optimizer_chunk_table(..., cuts)``); the ends themselves may fall into the padding
    between two variables, which belongs to no chunk."""
    import bisect
    b, e = bisect.bisect_left(starts, lo), bisect.bisect_left(starts, hi)
    assert (b == 0 or starts[b - 1] + lens[b - 1] <= lo) and (e == 0 or starts[e - 1] + lens[e - 1] <= hi), \
        "a chunk straddles the end of a slice: the table was not cut there"
    return b, e


class OptimizerTables:
    """Chunk / segment tables of a VariableStore for the flat optimizer kernels."""
    CHUNK = 65536

    def __init__(self, store, regularizable, trainable, cuts=()):
        lib = _lib.load()
        self.names = list(store.specs)
        starts, lens, segs, first, count, flags = optimizer_chunk_table(store, regularizable, trainable, cuts, self.CHUNK)
        dev = store.device
        self._starts_host, self._lens_host = list(starts), list(lens)
        self.chunk_start = torch.tensor(starts, dtype=torch.int64, device=dev)
        self.chunk_len = torch.tensor(lens, dtype=torch.int32, device=dev)
        self.chunk_seg = torch.tensor(segs, dtype=torch.int32, device=dev)
        self.seg_first = torch.tensor(first, dtype=torch.int32, device=dev)
        self.seg_count = torch.tensor(count, dtype=torch.int32, device=dev)
        self.seg_flags = torch.tensor(flags, dtype=torch.int32, device=dev)
        self.nchunk, self.nseg = len(starts), len(first)
        self.workspace = torch.empty(lib.nm_optim_workspace_bytes(self.nchunk, self.nseg) // 4,
                                     dtype=torch.float32, device=dev)
        self.l1l2 = torch.zeros(2, dtype=torch.float32, device=dev)

    def _tabs(self):
        return (self.chunk_start.data_ptr(), self.chunk_len.data_ptr(), self.chunk_seg.data_ptr(),
                self.seg_first.data_ptr(), self.seg_count.data_ptr(), self.seg_flags.data_ptr(),
                self.nchunk, self.nseg)

    def regularize_and_norms(self, theta, grad, l1_weight, l2_weight):
        """grad += d(l1*L1 + l2*L2); per-variable ||grad||^2; returns device [L1, L2]."""
        lib = _lib.load()
        _lib.check(lib.nm_optim_regularize_norms(_stream(), theta.data_ptr(), grad.data_ptr(), *self._tabs(),
                                                 float(l1_weight), float(l2_weight), self.l1l2.data_ptr(),
                                                 self.workspace.data_ptr(), self.workspace.numel() * 4),
                   "nm_optim_regularize_norms")
        return self.l1l2

    def clip_adam(self, theta, grad, m, v, clip_norm, lr_t, beta1, beta2, epsilon, skip=None, chunks=None):
        """Per-tensor clip + Adam.  ``skip``: an int32 device word that, when not zero, turns the launch into a no-op
        (the session's error word: the update of a step whose time loop gave up is never applied); ``chunks``: the
        (begin, end) range of chunks a rank owns under the sharded optimizer (default: all)."""
        self.apply(0, theta, grad, m, v, clip_norm, (lr_t, beta1, beta2, epsilon), skip=skip, chunks=chunks)

    def clip_adadelta(self, theta, grad, accum, accum_update, clip_norm, lr, rho, epsilon, skip=None, chunks=None):
        self.apply(1, theta, grad, accum, accum_update, clip_norm, (lr, rho, epsilon, 0.0), skip=skip, chunks=chunks)

    def chunk_list(self, ranges):
        """Device int32 array of the chunk indices in the (begin, end) ``ranges`` (kept: a rank's ranges never change)."""
        key = tuple((int(b), int(e)) for b, e in ranges)
        lists = self.__dict__.setdefault("_chunk_lists", {})
        hit = lists.get(key)
        if hit is None:
            idx = [c for b, e in key for c in range(b, e)]
            hit = lists[key] = torch.tensor(idx, dtype=torch.int32, device=self.workspace.device)
        return hit

    def apply(self, kind, theta, grad, slot0, slot1, clip_norm, params, skip=None, chunks=None, chunk_list=None):
        """nm_optim_apply: kind 0 Adam (lr_t, beta1, beta2, epsilon), 1 Adadelta (lr, rho, epsilon, -); ``chunks``: a
        (begin, end) range, ``chunk_list``: a device list of chunk indices (``chunk_list()``) in one launch."""
        lib = _lib.load()
        p0, p1, p2, p3 = (float(x) for x in params)
        if chunk_list is not None:
            _lib.check(lib.nm_optim_apply_list(_stream(), int(kind), theta.data_ptr(), grad.data_ptr(), slot0.data_ptr(),
                                               slot1.data_ptr(), *self._tabs(), float(clip_norm or 0.0), p0, p1, p2, p3,
                                               chunk_list.data_ptr(), chunk_list.numel(), _p(skip),
                                               self.workspace.data_ptr(), self.workspace.numel() * 4),
                       "nm_optim_apply_list")
            return
        c0, c1 = chunks if chunks is not None else (0, self.nchunk)
        _lib.check(lib.nm_optim_apply(_stream(), int(kind), theta.data_ptr(), grad.data_ptr(), slot0.data_ptr(),
                                      slot1.data_ptr(), *self._tabs(), float(clip_norm or 0.0), p0, p1, p2, p3,
                                      int(c0), int(c1), _p(skip), self.workspace.data_ptr(),
                                      self.workspace.numel() * 4), "nm_optim_apply")

    def partials(self, theta, grad, l1_weight, l2_weight, chunks, chunk_list=None):
        """Pass 1 over the chunks [begin, end) -- or over a device list of chunks: regulariser terms into ``grad``,
        partial sums into the workspace."""
        lib = _lib.load()
        if chunk_list is not None:
            _lib.check(lib.nm_optim_partials_list(_stream(), theta.data_ptr(), grad.data_ptr(), *self._tabs(),
                                                  float(l1_weight), float(l2_weight), chunk_list.data_ptr(),
                                                  chunk_list.numel(), self.workspace.data_ptr(),
                                                  self.workspace.numel() * 4), "nm_optim_partials_list")
            return
        _lib.check(lib.nm_optim_partials(_stream(), theta.data_ptr(), grad.data_ptr(), *self._tabs(),
                                         float(l1_weight), float(l2_weight), int(chunks[0]), int(chunks[1]),
                                         self.workspace.data_ptr(), self.workspace.numel() * 4), "nm_optim_partials")

    def partial_vector(self):
        """The 3 floats per chunk that ``partials`` writes (a view of the workspace)."""
        return self.workspace[:3 * self.nchunk]

    def segments(self):
        """Pass 2: per-variable squared norms + [L1, L2] from the whole partial vector, in a fixed order."""
        lib = _lib.load()
        _lib.check(lib.nm_optim_segments(_stream(), *self._tabs(), self.l1l2.data_ptr(), self.workspace.data_ptr(),
                                         self.workspace.numel() * 4), "nm_optim_segments")
        return self.l1l2

    def chunk_range(self, lo, hi):
        """The chunks whose elements lie in [lo, hi) of the flat buffer: (begin, end)."""
        return chunk_range_of(self._starts_host, self._lens_host, lo, hi)
