"""Recurrent sentence encoder (mirror of neuralmonkey/encoders/recurrent.py).

RecurrentEncoder.rnn (recurrent.py:179-217) = dropout(input) -> [LN] -> rnn_layer
-> dropout -> (residual) -> final layer_norm on states and final state.
rnn_layer (recurrent.py:71-110) = tf.nn.(bidirectional_)dynamic_rnn with
sequence_length masking and reverse_sequence for the backward direction.

MI355X mapping: the input half of both GRU kernels is hoisted out of the time
loop into MFMA GEMMs over all B*S positions; the loop runs the two recurrent
GEMMs (both directions batched in one launch) with fused gate / blend
epilogue kernels that implement the length masking and the reversed indexing
of the backward direction in-kernel, so there is no reverse_sequence copy.
"""
import os
from typing import Callable, List, NamedTuple, Optional, Tuple, Union

import torch

from .. import autodiff as F
from .. import ops
from ..model.model_part import InitializerSpecs, ModelPart
from ..model.sequence import EmbeddedSequence
from ..model.stateful import TemporalStateful, TemporalStatefulWithOutput
from ..nn import gru
from ..nn.cells import make_cell
from ..nn.dropout import dropout
from ..runtime import tensor
from ..variables import ones_initializer, orthogonal_initializer, zeros_initializer, constant_initializer
from ..vocabulary import Vocabulary

RNN_CELL_TYPES = ("NematusGRU", "GRU", "LSTM")
RNN_DIRECTIONS = ["forward", "backward", "bidirectional"]

RNNSpecTuple = Union[Tuple[int], Tuple[int, str], Tuple[int, str, str]]


class RNNSpec(NamedTuple):
    size: int
    direction: str
    cell_type: str


def _make_rnn_spec(size: int, direction: str = "bidirectional", cell_type: str = "GRU") -> RNNSpec:
    """recurrent.py:42-63."""
    if size <= 0:
        raise ValueError("RNN size must be a positive integer. {} given.".format(size))
    if direction not in RNN_DIRECTIONS:
        raise ValueError("RNN direction must be one of {}. {} given.".format(str(RNN_DIRECTIONS), direction))
    if cell_type not in RNN_CELL_TYPES:
        raise ValueError("RNN cell type must be one of {}. {} given.".format(str(RNN_CELL_TYPES), cell_type))
    return RNNSpec(size, direction, cell_type)


class EncoderActivations(NamedTuple):
    states: torch.Tensor        # [B,S,C] (after the final layer norm)
    final: torch.Tensor         # [B,C]
    saved: dict                 # activations kept for the backward pass


class RecurrentEncoder(ModelPart, TemporalStatefulWithOutput):
    has_time_loop = True      # its backward pass is a latency-bound BPTT loop: leaf GEMMs of other parts hide under it

    # pylint: disable=too-many-arguments
    def __init__(self, name: str, input_sequence: TemporalStateful, rnn_layers: List[RNNSpecTuple],
                 add_residual: bool = False, add_layer_norm: bool = False,
                 include_final_layer_norm: bool = True, dropout_keep_prob: float = 1.0,
                 reuse: ModelPart = None, save_checkpoint: str = None, load_checkpoint: str = None,
                 initializers: InitializerSpecs = None) -> None:
        ModelPart.__init__(self, name, reuse, save_checkpoint, load_checkpoint, initializers)
        self.input_sequence = input_sequence
        self.dropout_keep_prob = dropout_keep_prob
        self.rnn_specs = [_make_rnn_spec(*r) for r in rnn_layers]
        self.add_residual = add_residual
        self.add_layer_norm = add_layer_norm
        self.include_final_layer_norm = include_final_layer_norm
        if self.dropout_keep_prob <= 0.0 or self.dropout_keep_prob > 1.0:
            raise ValueError("Dropout keep prob must be inside (0,1].")
        layer_sizes = [2 * l.size if l.direction == "bidirectional" else l.size for l in self.rnn_specs]
        if add_residual and len(set(layer_sizes)) > 1:
            raise ValueError("When using residual connectiong, all layers must have the same size, "
                             "but are {}.".format(layer_sizes))
        self._layer_sizes = layer_sizes
        # one cell object per (layer, direction): variables + the taped step of the general path
        self._cells = []
        d_in = input_sequence.dimension
        for i, spec in enumerate(self.rnn_specs):
            self._cells.append([make_cell(spec.cell_type, self, "rnn_{}_{}/{}".format(i, spec.direction, d),
                                          d_in, spec.size) for d in self._dirs(spec)])
            d_in = layer_sizes[i]

    def uses_general_path(self, train_mode: bool) -> bool:
        """The hand-scheduled path covers one TF-GRU layer without layer norm / residual and
        without dropout; everything else runs on the autodiff tape."""
        if len(self.rnn_specs) != 1 or self.rnn_specs[0].cell_type != "GRU":
            return True
        if self.add_layer_norm or self.add_residual:
            return True
        if self.rnn_specs[0].size % 4 != 0 or self.input_sequence.dimension % 4 != 0:
            return True                  # the fused gate kernels move float4s (tests/bahdanau.ini: GRU 7)
        return train_mode and self.dropout_keep_prob != 1.0

    def graph_safe_training(self, train_mode: bool) -> bool:
        """The taped path is pure kernel launches on persistent buffers (see Decoder.graph_safe_training)."""
        return self.uses_general_path(train_mode)

    # -- static sizes ------------------------------------------------------------
    @property
    def dimension(self) -> int:
        return self._layer_sizes[-1]

    @property
    def output_size(self) -> int:
        return self._layer_sizes[-1]

    def _dirs(self, spec: RNNSpec) -> List[str]:
        if spec.direction == "bidirectional":
            return ["bidirectional_rnn/fw", "bidirectional_rnn/bw"]
        return ["rnn"]

    def declare_variables(self, store) -> None:
        d_in = self.input_sequence.dimension
        for i, spec in enumerate(self.rnn_specs):
            for cell in self._cells[i]:
                cell.declare_variables(store)          # GRU: <scope>/OrthoGRUCell/{gates,candidate}/...
            if self.add_layer_norm:
                pre = "rnn_{}_{}/LayerNorm".format(i, spec.direction)
                self.declare(store, pre + "/gamma", (d_in,), ones_initializer())
                self.declare(store, pre + "/beta", (d_in,), zeros_initializer())
            d_in = self._layer_sizes[i]
        if self.include_final_layer_norm:
            self.declare(store, "LayerNorm/gamma", (self._layer_sizes[-1],), ones_initializer())
            self.declare(store, "LayerNorm/beta", (self._layer_sizes[-1],), zeros_initializer())

    # -- weight views for the kernels -----------------------------------------------
    def _cell_views(self, ctx, layer: int):
        spec = self.rnn_specs[layer]
        store = ctx.store
        views = []
        for d in self._dirs(spec):
            pre = "rnn_{}_{}/{}/OrthoGRUCell".format(layer, spec.direction, d)
            views.append({k: self.var(ctx, pre + k) for k in
                          ("/gates/kernel", "/gates/bias", "/candidate/kernel", "/candidate/bias")})
            views[-1]["off_g"] = store.offset(self.var_name(pre + "/gates/kernel"))
            views[-1]["off_c"] = store.offset(self.var_name(pre + "/candidate/kernel"))
        return views

    @tensor
    def rnn_input(self, ctx) -> torch.Tensor:
        return dropout(ctx, self.input_sequence.temporal_states(ctx), self.dropout_keep_prob,
                       ctx.fed(self.train_mode))

    @tensor
    def rnn(self, ctx) -> EncoderActivations:
        """One (bi)directional GRU layer + final layer norm."""
        if self.uses_general_path(bool(ctx.fed(self.train_mode))):
            return self._general_rnn(ctx)
        x = self.rnn_input(ctx)                                     # [B,S,E]
        lengths = self.input_sequence.lengths(ctx)                  # int32 [B]
        spec = self.rnn_specs[0]
        bsz, slen, e = x.shape
        h = spec.size
        cells = self._cell_views(ctx, 0)
        ndir = len(cells)
        c_out = ndir * h
        key = id(self)
        theta = ctx.store.theta

        # ---- hoisted input projection: xp[b,s, d*3H : (d+1)*3H] = x.[Wg_x | Wc_x] + [bg | bc]
        xp = ctx.buffer((key, "xp"), (bsz * slen, ndir * 3 * h))
        x2 = x.view(bsz * slen, e)
        for d, cv in enumerate(cells):
            ops.gemm(x2, cv["/gates/kernel"][:e], out=xp[:, d * 3 * h:d * 3 * h + 2 * h],
                     bias=cv["/gates/bias"])
            ops.gemm(x2, cv["/candidate/kernel"][:e], out=xp[:, d * 3 * h + 2 * h:(d + 1) * 3 * h],
                     bias=cv["/candidate/bias"])

        # ---- recurrent weights of both directions as one strided batch
        def dir_batch(off_key, ncols):
            base = cells[0][off_key] + e * ncols
            stride = (cells[1][off_key] - cells[0][off_key]) if ndir == 2 else 0
            return theta.as_strided((ndir, h, ncols), (stride, ncols, 1), base)
        wgh = dir_batch("off_g", 2 * h)
        wch = dir_batch("off_c", h)

        train = ctx.wants_backward(bool(ctx.fed(self.train_mode)))
        states_raw = ctx.buffer((key, "states_raw"), (bsz, slen, c_out), zero=True)
        hcur = ctx.buffer((key, "hcur"), (ndir, bsz, h), zero=True)
        hg = ctx.buffer((key, "hg"), (ndir, bsz, 2 * h))
        hc = ctx.buffer((key, "hc"), (ndir, bsz, h))
        rh = ctx.buffer((key, "rh"), (ndir, bsz, h))
        if train:       # keep per-step gates / candidates for the backward pass
            ru_all = ctx.buffer((key, "ru_all"), (slen, ndir, bsz, 2 * h))
            c_all = ctx.buffer((key, "c_all"), (slen, ndir, bsz, h))
        else:
            ru_all = ctx.buffer((key, "ru_one"), (1, ndir, bsz, 2 * h))
            c_all = None
        reverse_only = spec.direction == "backward"
        len_arg = lengths
        xrs, xts = slen * ndir * 3 * h, ndir * 3 * h
        ors, ots = slen * c_out, c_out
        # (transposed [N,K] copies of the recurrent kernels were measured SLOWER here, 17.4 vs 16.7 us per step:
        # a lane's 16-byte fragment then comes from its own 2 KB row, while the [K,N] layout serves 16 lanes from
        # one 64-byte segment -- tools/gru_loop_bench.py)
        fused = False
        wg_f, wc_f = wgh, wch

        def time_loop():
            ops.zero(states_raw)
            ops.zero(hcur)
            for t in range(slen):
                ru = ru_all[t] if train else ru_all[0]
                gru.step_fwd(xp, (3 * h, xrs, xts), hcur, hcur, wg_f, wc_f, ru, rh, c_all[t] if train else None,
                             states_raw, (h, ors, ots), len_arg, t, ndir, bsz, h, reverse_only, hg, hc,
                             transposed=fused)
        loop_h = gru.seq_mode(ctx.session, bsz, h, ndir, wgh, wch)
        if loop_h:
            # both directions, all positions: ONE launch (csrc/nm_gru_cluster.hip); a hidden size the kernels do not
            # take (300) runs at the next one they do, on zero-padded operands (nn/gru.py: seq_fwd)
            ctx.session.start_deferred_side()      # work that waits for a time loop to hide under (Session.defer_side)
            ops.zero(states_raw)
            ops.zero(hcur)
            gru.seq_fwd(ctx, key, loop_h, slen, ndir, bsz, h, xp, (3 * h, xrs, xts), hcur, hcur, 0, ru_all[0],
                        ndir * bsz * 2 * h if train else 0, None, 0, c_all[0] if train else None, ndir * bsz * h, wgh,
                        wch, lengths=len_arg, reverse_dir0=reverse_only, out=states_raw, out_strides=(h, ors, ots))
        else:
            ctx.session.start_deferred_side()
            ctx.session.graphed((key, "fwd_loop", bsz, slen, train), time_loop)
        final_raw = ctx.buffer((key, "final_raw"), (bsz, c_out))
        for d in range(ndir):
            ops.copy_cols(hcur[d], final_raw[:, d * h:(d + 1) * h])

        saved = {"x": x, "xp": xp, "ru_all": ru_all, "c_all": c_all, "states_raw": states_raw,
                 "final_raw": final_raw, "lengths": lengths, "wgh": wgh, "wch": wch, "cells": cells,
                 "ndir": ndir, "h": h, "reverse_only": reverse_only}
        if not self.include_final_layer_norm:
            return EncoderActivations(states_raw, final_raw, saved)
        gamma, beta = self.var(ctx, "LayerNorm/gamma"), self.var(ctx, "LayerNorm/beta")
        states = ctx.buffer((key, "states"), (bsz, slen, c_out))
        final = ctx.buffer((key, "final"), (bsz, c_out))
        st_mean = ctx.buffer((key, "st_mean"), (bsz * slen,))
        st_rstd = ctx.buffer((key, "st_rstd"), (bsz * slen,))
        fi_mean = ctx.buffer((key, "fi_mean"), (bsz,))
        fi_rstd = ctx.buffer((key, "fi_rstd"), (bsz,))
        ops.layer_norm_fwd(states_raw, gamma, beta, out=states, mean=st_mean, rstd=st_rstd)
        ops.layer_norm_fwd(final_raw, gamma, beta, out=final, mean=fi_mean, rstd=fi_rstd)
        saved.update(st_mean=st_mean, st_rstd=st_rstd, fi_mean=fi_mean, fi_rstd=fi_rstd)
        return EncoderActivations(states, final, saved)

    def _backward_loop(self, ctx, sv, d_states, d_final) -> torch.Tensor:
        """Final layer norm's backward, then the BPTT loop (a HIP graph); returns dxp [B*S, ndir*3H]."""
        store = ctx.store
        x, lengths = sv["x"], sv["lengths"]
        bsz, slen, _ = x.shape
        ndir, h, rev0 = sv["ndir"], sv["h"], sv["reverse_only"]
        c_out = ndir * h
        key = (id(self), "bwd")
        states_raw, final_raw = sv["states_raw"], sv["final_raw"]

        # ---- final layer norm (shared gamma/beta for states and final state)
        if self.include_final_layer_norm:
            gamma = self.var(ctx, "LayerNorm/gamma")
            g_gamma, g_beta = store.g(self.var_name("LayerNorm/gamma")), store.g(self.var_name("LayerNorm/beta"))
            d_states_raw = d_final_raw = None
            if d_states is not None:
                d_states_raw = ctx.buffer(key + ("d_states_raw",), (bsz, slen, c_out))
                tmp = ctx.buffer(key + ("ln_tmp",), (bsz * slen, c_out))
                ops.layer_norm_bwd(d_states, states_raw, sv["st_mean"], sv["st_rstd"], gamma, d_states_raw, tmp)
                ops.colsum(tmp, g_gamma, accumulate=True)
                ops.colsum(d_states.view(bsz * slen, c_out), g_beta, accumulate=True)
            if d_final is not None:
                d_final_raw = ctx.buffer(key + ("d_final_raw",), (bsz, c_out))
                tmp2 = ctx.buffer(key + ("ln_tmp2",), (bsz, c_out))
                ops.layer_norm_bwd(d_final, final_raw, sv["fi_mean"], sv["fi_rstd"], gamma, d_final_raw, tmp2)
                ops.colsum(tmp2, g_gamma, accumulate=True)
                ops.colsum(d_final, g_beta, accumulate=True)
        else:
            d_states_raw, d_final_raw = d_states, d_final

        # ---- BPTT over the (bi)directional GRU
        dh = ctx.buffer(key + ("dh",), (ndir, bsz, h), zero=True)
        if d_final_raw is not None:
            for d in range(ndir):
                ops.copy_cols(d_final_raw[:, d * h:(d + 1) * h], dh[d])
        dxp = ctx.buffer(key + ("dxp",), (bsz * slen, ndir * 3 * h), zero=True)
        dgpre = ctx.buffer(key + ("dgpre",), (2, ndir, bsz, 2 * h))
        dcpre = ctx.buffer(key + ("dcpre",), (ndir, bsz, h))
        drh = ctx.buffer(key + ("drh",), (ndir, bsz, h))
        seq_strides = (h, slen * c_out, c_out)
        dxp_strides = (3 * h, slen * ndir * 3 * h, ndir * 3 * h)
        wgh, wch = sv["wgh"], sv["wch"]
        loop_h = gru.seq_mode(ctx.session, bsz, h, ndir, wgh, wch)
        if loop_h:
            gru.seq_bwd(ctx, id(self), loop_h, slen, ndir, bsz, h, dh, d_states_raw,
                        seq_strides if d_states_raw is not None else None, sv["ru_all"][0], ndir * bsz * 2 * h,
                        sv["c_all"][0], ndir * bsz * h, None, states_raw, seq_strides, dxp, dxp_strides, wgh, wch,
                        lengths=lengths, reverse_dir0=rev0)
            from .. import distributed
            if distributed.current() is not None:
                distributed.current().after_time_loops()
            return dxp

        def bptt_loop():
            gru.bptt(slen, dh, d_states_raw, seq_strides if d_states_raw is not None else None, sv["ru_all"],
                     sv["c_all"], None, states_raw, seq_strides, dxp, dxp_strides, wgh, wch, lengths, ndir, bsz,
                     h, rev0, dgpre, dcpre, drh)
        ctx.session.graphed((id(self), "bwd_loop", bsz, slen, d_states_raw is not None), bptt_loop)
        return dxp

    def backward(self, ctx, d_states: Optional[torch.Tensor], d_final: Optional[torch.Tensor]) -> None:
        """dL/d(temporal_states) [B,S,C] and dL/d(output) [B,C] -> variable
        gradients of this encoder and of its input sequence."""
        store = ctx.store
        act = self.rnn(ctx)
        sv = act.saved
        if "tape" in sv:
            return self._general_backward(ctx, sv, d_states, d_final)
        x, xp, lengths = sv["x"], sv["xp"], sv["lengths"]
        bsz, slen, e = x.shape
        ndir, h, rev0 = sv["ndir"], sv["h"], sv["reverse_only"]
        c_out = ndir * h
        key = (id(self), "bwd")
        states_raw, final_raw = sv["states_raw"], sv["final_raw"]

        dxp = self._backward_loop(ctx, sv, d_states, d_final)

        # ---- weight gradients, batched over all positions
        hprev = ctx.buffer(key + ("hprev",), (bsz, slen, ndir, h))
        rh_seq = ctx.buffer(key + ("rh_seq",), (bsz, slen, ndir, h))
        ops.gru_seq_shift(states_raw, hprev, lengths, ndir, h, reverse_dir0=rev0)
        ops.gru_rh_seq(sv["ru_all"], hprev, rh_seq, lengths, ndir, h, reverse_dir0=rev0)
        x2 = x.view(bsz * slen, e)
        hp2, rh2 = hprev.view(bsz * slen, c_out), rh_seq.view(bsz * slen, c_out)
        dx = ctx.buffer(key + ("dx",), (bsz * slen, e))
        spec = self.rnn_specs[0]
        first = True
        for d, (dname, cv) in enumerate(zip(self._dirs(spec), sv["cells"])):
            pre = "rnn_0_{}/{}/OrthoGRUCell".format(spec.direction, dname)
            g_wg = store.g(self.var_name(pre + "/gates/kernel"))
            g_wc = store.g(self.var_name(pre + "/candidate/kernel"))
            dg = dxp[:, d * 3 * h:d * 3 * h + 2 * h]
            dc = dxp[:, d * 3 * h + 2 * h:(d + 1) * 3 * h]
            acc = self.shares_variables          # encoders sharing this scope (reuse=) add their gradients up
            ops.gemm(x2, dg, out=g_wg[:e], trans_a=True, accumulate=acc)
            ops.gemm(hp2[:, d * h:(d + 1) * h], dg, out=g_wg[e:], trans_a=True, accumulate=acc)
            ops.gemm(x2, dc, out=g_wc[:e], trans_a=True, accumulate=acc)
            ops.gemm(rh2[:, d * h:(d + 1) * h], dc, out=g_wc[e:], trans_a=True, accumulate=acc)
            ops.colsum(dg, store.g(self.var_name(pre + "/gates/bias")), accumulate=acc)
            ops.colsum(dc, store.g(self.var_name(pre + "/candidate/bias")), accumulate=acc)
            ops.gemm(dg, cv["/gates/kernel"][:e], out=dx, trans_b=True, accumulate=not first)
            ops.gemm(dc, cv["/candidate/kernel"][:e], out=dx, trans_b=True, accumulate=True)
            first = False
        self.input_sequence.backward(ctx, dx.view(bsz, slen, e))

    # -- general (taped) path: any cell, stacked layers, layer norm, residual, dropout ------------
    def _nematus_cluster_layer(self, tape, x, bsz: int, slen: int, lengths, layer: int):
        """One NematusGRUCell layer (both directions) as ONE tape operation: the input projections of all positions
        as products over [B*S] rows, the time loop as one cluster launch (ops.nematus_seq_fwd), and on the way back one
        launch for the BPTT loop, then the weight gradients as products over all positions -- the schedule of the
        hand-written GRU path above instead of 8 launches per step and direction each way.  None: this layer stays on
        the step-by-step tape (shape the cluster kernels do not take, cluster loops switched off, unaligned kernels)."""
        spec = self.rnn_specs[layer]
        cells = self._cells[layer]
        ctx = tape.ctx
        ndir, h = len(cells), spec.size
        d_in = x.shape[1]
        if spec.cell_type != "NematusGRU" or not ctx.session.use_cluster_loops or d_in % 4 or not x.data.is_cuda:
            return None
        store = ctx.store
        theta = store.theta
        names = {k: [self.var_name(c._n(k)) for c in cells]
                 for k in ("gates/state_proj/kernel", "candidate/state_proj/kernel")}

        def dir_batch(k, ncols):
            base = store.offset(names[k][0])
            stride = store.offset(names[k][1]) - base if ndir == 2 else 0
            return theta.as_strided((ndir, h, ncols), (stride, ncols, 1), base)
        ug, uc = dir_batch("gates/state_proj/kernel", 2 * h), dir_batch("candidate/state_proj/kernel", h)
        q = gru.seq_mode(ctx.session, bsz, h, ndir, ug, uc)      # the hidden size the loops run at: h, or h padded
        if not q:
            return None
        padded = q != h         # blocks of width h at offsets that are multiples of q; the rest zero (nn/gru.py)
        use_sb, use_ib = cells[0].use_state_bias, cells[0].use_input_bias
        rev0 = spec.direction == "backward"
        width = ndir * h

        def p(cell, name):
            return tape.param(self, cell._n(name))
        w_gi = [p(c, "gates/input_proj/kernel") for c in cells]
        w_ci = [p(c, "candidate/input_proj/kernel") for c in cells]
        w_gs = [p(c, "gates/state_proj/kernel") for c in cells]
        w_cs = [p(c, "candidate/state_proj/kernel") for c in cells]
        b_gi = [p(c, "gates/input_proj/bias") if use_ib else None for c in cells]
        b_ci = [p(c, "candidate/input_proj/bias") if use_ib else None for c in cells]
        b_gs = [p(c, "gates/state_proj/bias") if use_sb else None for c in cells]
        b_cs = [p(c, "candidate/state_proj/bias") if use_sb else None for c in cells]
        key = (id(self), "nematus", layer)
        if padded:
            ug, uc = gru._pad_weights(ctx, key, ug, uc, h, q)       # pylint: disable=protected-access
        # the reset / update halves of the gates: one block [0, 2h) when nothing is padded, else two of width h
        gate_cols = [(0, 2 * h, 0)] if not padded else [(0, h, 0), (h, 2 * h, q)]
        bgs = bcs = None
        if use_sb:                                  # the loops read the state biases as [ndir, 2q] / [ndir, q]
            bgs, bcs = tape.buf((ndir, 2 * q), zero=padded), tape.buf((ndir, q), zero=padded)
            for d in range(ndir):
                for lo, hi, off in gate_cols:
                    ops.copy_cols(b_gs[d].data[lo:hi].view(1, -1), bgs[d:d + 1, off:off + hi - lo])
                ops.copy_cols(b_cs[d].data.view(1, -1), bcs[d:d + 1, :h])

        xp = tape.buf((bsz * slen, ndir * 3 * q), zero=padded)
        for d in range(ndir):
            for lo, hi, off in gate_cols:
                ops.gemm(x.data, w_gi[d].data[:, lo:hi], out=xp[:, d * 3 * q + off:d * 3 * q + off + hi - lo],
                         bias=None if b_gi[d] is None else b_gi[d].data[lo:hi])
            ops.gemm(x.data, w_ci[d].data, out=xp[:, d * 3 * q + 2 * q:d * 3 * q + 2 * q + h],
                     bias=None if b_ci[d] is None else b_ci[d].data)
        out, final = tape.new((bsz * slen, width)), tape.new((bsz, width))
        out_q = tape.buf((bsz * slen, ndir * q)) if padded else out.data
        hcur, hzero = tape.buf((ndir, bsz, q)), tape.buf((ndir, bsz, q), zero=True)
        rec = tape.recording
        nsave = slen if rec else 1
        ru_all, c_all, sc_all = (tape.buf((nsave, ndir, bsz, 2 * q)), tape.buf((nsave, ndir, bsz, q)),
                                 tape.buf((nsave, ndir, bsz, q)))
        ws = ctx.buffer(key + ("ws",), (ops.nematus_seq_workspace_floats(bsz, q, ndir),))
        ops.zero(out_q)
        xrs, xts = slen * ndir * 3 * q, ndir * 3 * q
        seq_strides = (q, slen * ndir * q, ndir * q)
        ops.nematus_seq_fwd(slen, ndir, bsz, q, xp, (3 * q, xrs, xts), hzero, hcur, 0, ru_all[0],
                            ndir * bsz * 2 * q if rec else 0, sc_all[0], ndir * bsz * q if rec else 0, c_all[0],
                            ndir * bsz * q if rec else 0, ug, uc, ws, bgs=bgs, bcs=bcs, lengths=lengths,
                            reverse_dir0=rev0, out=out_q, out_strides=seq_strides, sticky=ctx.session.error_word())
        if padded:
            gru._blocks(out.data, out_q, ndir, 1, h, q, back=True)       # pylint: disable=protected-access
        for d in range(ndir):
            ops.copy_cols(hcur[d][:, :h], final.data[:, d * h:(d + 1) * h])

        def bwd():
            if out.grad is None and final.grad is None:
                return
            dh = tape.buf((ndir, bsz, q), zero=padded or final.grad is None)
            if final.grad is not None:
                for d in range(ndir):
                    ops.copy_cols(final.grad[:, d * h:(d + 1) * h], dh[d][:, :h])
            dout = out.grad
            if padded and dout is not None:
                dout = tape.buf((bsz * slen, ndir * q), zero=True)
                gru._blocks(out.grad, dout, ndir, 1, h, q)                # pylint: disable=protected-access
            dxp = tape.buf((bsz * slen, ndir * 4 * q), zero=True)
            ops.nematus_seq_bwd(slen, ndir, bsz, q, dh, dout, seq_strides if dout is not None else None,
                                ru_all[0], ndir * bsz * 2 * q, sc_all[0], ndir * bsz * q, c_all[0], ndir * bsz * q,
                                None, out_q, seq_strides, dxp, (4 * q, slen * ndir * 4 * q, ndir * 4 * q), ug, uc,
                                ws, lengths=lengths, reverse_dir0=rev0, sticky=ctx.session.error_word())
            hprev = tape.buf((bsz, slen, ndir, q))
            ops.gru_seq_shift(out_q.view(bsz, slen, ndir * q), hprev, lengths, ndir, q, reverse_dir0=rev0)
            hp2 = hprev.view(bsz * slen, ndir * q)
            gx, acc = tape.grad_slot(x) if x.needs_grad else (None, False)
            for d in range(ndir):
                base = d * 4 * q
                dc = dxp[:, base + 2 * q:base + 2 * q + h]
                dsc = dxp[:, base + 3 * q:base + 3 * q + h]
                hp = hp2[:, d * q:d * q + h]
                for lo, hi, off in gate_cols:
                    dg = dxp[:, base + off:base + off + hi - lo]
                    ops.gemm(x.data, dg, out=tape.grad(w_gi[d])[:, lo:hi], trans_a=True, accumulate=True)
                    ops.gemm(hp, dg, out=tape.grad(w_gs[d])[:, lo:hi], trans_a=True, accumulate=True)
                    if use_ib:
                        ops.colsum(dg, tape.grad(b_gi[d])[lo:hi], accumulate=True)
                    if use_sb:
                        ops.colsum(dg, tape.grad(b_gs[d])[lo:hi], accumulate=True)
                    if gx is not None:
                        ops.gemm(dg, w_gi[d].data[:, lo:hi], out=gx, trans_b=True, accumulate=acc)
                        acc = True
                ops.gemm(x.data, dc, out=tape.grad(w_ci[d]), trans_a=True, accumulate=True)
                ops.gemm(hp, dsc, out=tape.grad(w_cs[d]), trans_a=True, accumulate=True)
                if use_ib:
                    ops.colsum(dc, tape.grad(b_ci[d]), accumulate=True)
                if use_sb:
                    ops.colsum(dsc, tape.grad(b_cs[d]), accumulate=True)
                if gx is not None:
                    ops.gemm(dc, w_ci[d].data, out=gx, trans_b=True, accumulate=True)
        tape.record(bwd)
        return out, final

    def _gru_cluster_layer(self, tape, x, bsz: int, slen: int, lengths, layer: int):
        """One TF-GRUCell layer of the GENERAL path (stacked / layer-normed / residual / dropout encoders) as one tape
        operation: the schedule of the hand-written fast path above -- input halves of both kernels hoisted to products
        over all positions, the time loop as one cluster launch each way (nn/gru.py: seq_fwd / seq_bwd, padded where the
        kernels do not take the hidden size), weight gradients as products over all positions."""
        spec = self.rnn_specs[layer]
        cells = self._cells[layer]
        ctx = tape.ctx
        ndir, h = len(cells), spec.size
        e = x.shape[1]
        if spec.cell_type != "GRU" or not ctx.session.use_cluster_loops or e % 4 or h % 4 or not x.data.is_cuda:
            return None
        store = ctx.store
        kn = {k: [self.var_name(c._n(k)) for c in cells] for k in ("gates/kernel", "candidate/kernel")}    # pylint: disable=protected-access

        def dir_batch(k, ncols):
            base = store.offset(kn[k][0]) + e * ncols
            stride = store.offset(kn[k][1]) - store.offset(kn[k][0]) if ndir == 2 else 0
            return store.theta.as_strided((ndir, h, ncols), (stride, ncols, 1), base)
        wgh, wch = dir_batch("gates/kernel", 2 * h), dir_batch("candidate/kernel", h)
        loop_h = gru.seq_mode(ctx.session, bsz, h, ndir, wgh, wch)
        if not loop_h:
            return None
        rev0 = spec.direction == "backward"
        width = ndir * h
        par = lambda c, n: tape.param(self, c._n(n))                         # pylint: disable=protected-access
        wg, bg = [par(c, "gates/kernel") for c in cells], [par(c, "gates/bias") for c in cells]
        wc, bc = [par(c, "candidate/kernel") for c in cells], [par(c, "candidate/bias") for c in cells]
        xp = tape.buf((bsz * slen, ndir * 3 * h))
        for d in range(ndir):
            ops.gemm(x.data, wg[d].data[:e], out=xp[:, d * 3 * h:d * 3 * h + 2 * h], bias=bg[d].data)
            ops.gemm(x.data, wc[d].data[:e], out=xp[:, d * 3 * h + 2 * h:(d + 1) * 3 * h], bias=bc[d].data)
        out, final = tape.new((bsz * slen, width)), tape.new((bsz, width))
        hcur = tape.buf((ndir, bsz, h), zero=True)
        rec = tape.recording
        nsave = slen if rec else 1
        ru_all, c_all = tape.buf((nsave, ndir, bsz, 2 * h)), (tape.buf((nsave, ndir, bsz, h)) if rec else None)
        key = (id(self), "gru_layer", layer)
        ops.zero(out.data)
        seq_strides = (h, slen * width, width)
        x_strides = (3 * h, slen * ndir * 3 * h, ndir * 3 * h)
        gru.seq_fwd(ctx, key, loop_h, slen, ndir, bsz, h, xp, x_strides, hcur, hcur, 0, ru_all[0],
                    ndir * bsz * 2 * h if rec else 0, None, 0, c_all[0] if rec else None, ndir * bsz * h, wgh, wch,
                    lengths=lengths, reverse_dir0=rev0, out=out.data, out_strides=seq_strides)
        for d in range(ndir):
            ops.copy_cols(hcur[d], final.data[:, d * h:(d + 1) * h])

        def bwd():
            if out.grad is None and final.grad is None:
                return
            dh = tape.buf((ndir, bsz, h), zero=final.grad is None)
            if final.grad is not None:
                for d in range(ndir):
                    ops.copy_cols(final.grad[:, d * h:(d + 1) * h], dh[d])
            dxp = tape.buf((bsz * slen, ndir * 3 * h), zero=True)
            gru.seq_bwd(ctx, key, loop_h, slen, ndir, bsz, h, dh, out.grad, seq_strides if out.grad is not None else None,
                        ru_all[0], ndir * bsz * 2 * h, c_all[0], ndir * bsz * h, None, out.data, seq_strides, dxp,
                        x_strides, wgh, wch, lengths=lengths, reverse_dir0=rev0)
            states = out.data.view(bsz, slen, width)
            hprev, rh_seq = tape.buf((bsz, slen, ndir, h)), tape.buf((bsz, slen, ndir, h))
            ops.gru_seq_shift(states, hprev, lengths, ndir, h, reverse_dir0=rev0)
            ops.gru_rh_seq(ru_all, hprev, rh_seq, lengths, ndir, h, reverse_dir0=rev0)
            hp2, rh2 = hprev.view(bsz * slen, width), rh_seq.view(bsz * slen, width)
            gx, acc = tape.grad_slot(x) if x.needs_grad else (None, False)
            for d in range(ndir):
                dg = dxp[:, d * 3 * h:d * 3 * h + 2 * h]
                dc = dxp[:, d * 3 * h + 2 * h:(d + 1) * 3 * h]
                g_wg, g_wc = tape.grad(wg[d]), tape.grad(wc[d])
                ops.gemm(x.data, dg, out=g_wg[:e], trans_a=True, accumulate=True)
                ops.gemm(hp2[:, d * h:(d + 1) * h], dg, out=g_wg[e:], trans_a=True, accumulate=True)
                ops.gemm(x.data, dc, out=g_wc[:e], trans_a=True, accumulate=True)
                ops.gemm(rh2[:, d * h:(d + 1) * h], dc, out=g_wc[e:], trans_a=True, accumulate=True)
                ops.colsum(dg, tape.grad(bg[d]), accumulate=True)
                ops.colsum(dc, tape.grad(bc[d]), accumulate=True)
                if gx is not None:
                    ops.gemm(dg, wg[d].data[:e], out=gx, trans_b=True, accumulate=acc)
                    ops.gemm(dc, wc[d].data[:e], out=gx, trans_b=True, accumulate=True)
                    acc = True
        tape.record(bwd)
        return out, final

    def _lstm_cluster_layer(self, tape, x, bsz: int, slen: int, lengths, layer: int):
        """One LSTMCell layer (both directions) as ONE tape operation, like ``_nematus_cluster_layer``: z = x.W_x + b of
        all positions as one product per direction, the time loop as one cluster launch each way (ops.lstm_seq_fwd /
        lstm_seq_bwd), the kernel's gradient as two products over all positions.  None: the layer stays on the
        step-by-step tape."""
        spec = self.rnn_specs[layer]
        cells = self._cells[layer]
        ctx = tape.ctx
        ndir, h = len(cells), spec.size
        d_in = x.shape[1]
        if spec.cell_type != "LSTM" or not ctx.session.use_cluster_loops or d_in % 4 or h % 4 or not x.data.is_cuda:
            return None
        store = ctx.store
        names = [self.var_name(c._n("kernel")) for c in cells]               # pylint: disable=protected-access
        base = store.offset(names[0]) + d_in * 4 * h                         # rows d_in .. d_in + h: the state half
        stride = store.offset(names[1]) - store.offset(names[0]) if ndir == 2 else 0
        wh = store.theta.as_strided((ndir, h, 4 * h), (stride, 4 * h, 1), base)
        if not gru.cluster_ok(ctx.session, bsz, h, ndir, wh, wh):
            return None
        rev0 = spec.direction == "backward"
        width = ndir * h
        w = [tape.param(self, c._n("kernel")) for c in cells]                # pylint: disable=protected-access
        b = [tape.param(self, c._n("bias")) for c in cells]                  # pylint: disable=protected-access
        xp = tape.buf((bsz * slen, ndir * 4 * h))
        for d in range(ndir):
            ops.gemm(x.data, w[d].data[:d_in], out=xp[:, d * 4 * h:(d + 1) * 4 * h], bias=b[d].data)
        out, final = tape.new((bsz * slen, width)), tape.new((bsz, width))
        hcur, hzero = tape.buf((ndir, bsz, h)), tape.buf((ndir, bsz, h), zero=True)
        rec = tape.recording
        nsave = slen if rec else 1
        gates, c_all = tape.buf((nsave, ndir, bsz, 4 * h)), tape.buf((nsave, ndir, bsz, h))
        ws = ctx.buffer((id(self), "lstm_ws", layer), (ops.lstm_seq_workspace_floats(bsz, h, ndir),))
        ops.zero(out.data)
        seq_strides = (h, slen * width, width)
        x_strides = (4 * h, slen * ndir * 4 * h, ndir * 4 * h)
        ops.lstm_seq_fwd(slen, ndir, bsz, h, xp, x_strides, hzero, hcur, 0, gates[0], ndir * bsz * 4 * h if rec else 0,
                         c_all[0], ndir * bsz * h if rec else 0, wh, ws, forget_bias=1.0, lengths=lengths,
                         reverse_dir0=rev0, out=out.data, out_strides=seq_strides, sticky=ctx.session.error_word())
        for d in range(ndir):
            ops.copy_cols(hcur[d], final.data[:, d * h:(d + 1) * h])

        def bwd():
            if out.grad is None and final.grad is None:
                return
            dh = tape.buf((ndir, bsz, h), zero=final.grad is None)
            if final.grad is not None:
                for d in range(ndir):
                    ops.copy_cols(final.grad[:, d * h:(d + 1) * h], dh[d])
            dxp = tape.buf((bsz * slen, ndir * 4 * h), zero=True)
            ops.lstm_seq_bwd(slen, ndir, bsz, h, dh, out.grad, seq_strides if out.grad is not None else None, gates[0],
                             ndir * bsz * 4 * h, c_all[0], ndir * bsz * h, dxp, x_strides, wh, ws, lengths=lengths,
                             reverse_dir0=rev0, sticky=ctx.session.error_word())
            hprev = tape.buf((bsz, slen, ndir, h))
            ops.gru_seq_shift(out.data.view(bsz, slen, width), hprev, lengths, ndir, h, reverse_dir0=rev0)
            hp2 = hprev.view(bsz * slen, width)
            gx, acc = tape.grad_slot(x) if x.needs_grad else (None, False)
            for d in range(ndir):
                dz = dxp[:, d * 4 * h:(d + 1) * 4 * h]
                gw = tape.grad(w[d])
                ops.gemm(x.data, dz, out=gw[:d_in], trans_a=True, accumulate=True)
                ops.gemm(hp2[:, d * h:(d + 1) * h], dz, out=gw[d_in:], trans_a=True, accumulate=True)
                ops.colsum(dz, tape.grad(b[d]), accumulate=True)
                if gx is not None:
                    ops.gemm(dz, w[d].data[:d_in], out=gx, trans_b=True, accumulate=acc)
                    acc = True
        tape.record(bwd)
        return out, final

    def _general_layer(self, tape, x, bsz: int, slen: int, lengths, layer: int, train: bool):
        """rnn_layer (recurrent.py:71-110) on the tape.  x: Var [B*S, D] -> (outputs Var
        [B*S, ndir*H], final Var [B, ndir*H])."""
        spec = self.rnn_specs[layer]
        cells = self._cells[layer]
        ndir, h = len(cells), spec.size
        width = ndir * h
        if os.environ.get("NM_NEMATUS_CLUSTER", "1") != "0":      # (read per call: tests compare both schedules)
            fused = self._nematus_cluster_layer(tape, x, bsz, slen, lengths, layer)
            if fused is None and os.environ.get("NM_LSTM_CLUSTER", "1") != "0":
                fused = self._lstm_cluster_layer(tape, x, bsz, slen, lengths, layer)
            if fused is None and os.environ.get("NM_GRU_LAYER_CLUSTER", "1") != "0":
                fused = self._gru_cluster_layer(tape, x, bsz, slen, lengths, layer)
            if fused is not None:
                return fused
        out = tape.new((bsz * slen, width))
        final = tape.new((bsz, width))
        for d, cell in enumerate(cells):
            reverse = spec.direction == "backward" or (spec.direction == "bidirectional" and d == 1)
            d_in = x.shape[1]
            if reverse:
                src = F.reverse_sequence(tape, tape.view(x, lambda t: t.view(bsz, slen, d_in)), lengths)
                src = tape.view(src, lambda t: t.view(bsz * slen, d_in))
                dst = tape.new((bsz * slen, h))
                lo = 0
            else:
                src, dst, lo = x, out, d * h
            state = tuple(tape.leaf(tape.buf((bsz, h), zero=True)) for _ in range(cell.state_count))
            for t in range(slen):
                x_t = tape.view(src, lambda v, t=t: v.view(bsz, slen, -1)[:, t])
                y_t = tape.view(dst, lambda v, t=t: v.view(bsz, slen, -1)[:, t, lo:lo + h])
                _, new_state = cell.step(tape, x_t, state)
                # dynamic_rnn: beyond the sentence length the state is carried and the output is 0
                carried = [F.rnn_select(tape, n, p, lengths, t, None) for n, p in zip(new_state[:-1], state[:-1])]
                carried.append(F.rnn_select(tape, new_state[-1], state[-1], lengths, t, y_t))
                state = tuple(carried)
            if reverse:
                back = F.reverse_sequence(tape, tape.view(dst, lambda t: t.view(bsz, slen, h)), lengths)
                F.copy(tape, tape.view(back, lambda t: t.view(bsz * slen, h)), out=tape.cols(out, d * h, (d + 1) * h))
            F.copy(tape, state[-1], out=tape.cols(final, d * h, (d + 1) * h))      # LSTM: .h
        return out, final

    def _general_rnn(self, ctx) -> EncoderActivations:
        """RecurrentEncoder.rnn (recurrent.py:179-217) on the autodiff tape."""
        train = bool(ctx.fed(self.train_mode))
        keep = self.dropout_keep_prob
        tape = F.Tape(ctx, (id(self), "genc"), recording=ctx.wants_backward(train))
        x_raw = self.input_sequence.temporal_states(ctx)
        lengths = self.input_sequence.lengths(ctx)
        bsz, slen, e = x_raw.shape
        x_in = tape.leaf(x_raw.reshape(bsz * slen, e), needs_grad=True)
        layer_input = F.dropout(tape, x_in, keep, train, ctx.salt(self.name, "rnn_input"))
        layer_final = tape.view(layer_input, lambda t: t.view(bsz, slen, -1)[:, slen - 1])
        for i, spec in enumerate(self.rnn_specs):
            if self.add_layer_norm:
                pre = "rnn_{}_{}/LayerNorm".format(i, spec.direction)
                layer_input = F.layer_norm(tape, layer_input, tape.param(self, pre + "/gamma"),
                                           tape.param(self, pre + "/beta"))
            out, fin = self._general_layer(tape, layer_input, bsz, slen, lengths, i, train)
            out = F.dropout(tape, out, keep, train, ctx.salt(self.name, "layer_output", i))
            fin = F.dropout(tape, fin, keep, train, ctx.salt(self.name, "layer_final", i))
            if self.add_residual and layer_input.shape[1] == out.shape[1]:
                layer_input = F.add(tape, layer_input, out)
                layer_final = F.add(tape, layer_final, fin)
            else:
                layer_input, layer_final = out, fin
        if self.include_final_layer_norm:
            gamma, beta = tape.param(self, "LayerNorm/gamma"), tape.param(self, "LayerNorm/beta")
            layer_input = F.layer_norm(tape, layer_input, gamma, beta)
            layer_final = F.layer_norm(tape, layer_final, gamma, beta)
        saved = {"tape": tape, "x_in": x_in, "states": layer_input, "final": layer_final,
                 "shape": (bsz, slen, e)}
        return EncoderActivations(layer_input.data.view(bsz, slen, -1), layer_final.data, saved)

    def _general_backward(self, ctx, sv, d_states, d_final) -> None:
        tape = sv["tape"]
        bsz, slen, e = sv["shape"]
        if d_states is not None:
            sv["states"].grad = d_states.reshape(bsz * slen, -1)
        if d_final is not None:
            sv["final"].grad = d_final
        tape.backward()
        if sv["x_in"].grad is not None:
            self.input_sequence.backward(ctx, sv["x_in"].grad.view(bsz, slen, e))

    @tensor
    def temporal_states(self, ctx) -> torch.Tensor:
        return self.rnn(ctx).states

    @tensor
    def temporal_mask(self, ctx) -> torch.Tensor:
        return self.input_sequence.temporal_mask(ctx)

    @tensor
    def output(self, ctx) -> torch.Tensor:
        return self.rnn(ctx).final


class SentenceEncoder(RecurrentEncoder):
    # pylint: disable=too-many-arguments,too-many-locals
    def __init__(self, name: str, vocabulary: Vocabulary, data_id: str, embedding_size: int,
                 rnn_size: int, rnn_cell: str = "GRU", rnn_direction: str = "bidirectional",
                 add_residual: bool = False, add_layer_norm: bool = False, max_input_len: int = None,
                 dropout_keep_prob: float = 1.0, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None,
                 embedding_initializer: Callable = None) -> None:
        """recurrent.py:236-314: an EmbeddedSequence named ``<name>_input`` feeding one RNN layer."""
        s_ckp = "input_{}".format(save_checkpoint) if save_checkpoint else None
        l_ckp = "input_{}".format(load_checkpoint) if load_checkpoint else None
        input_initializers = []
        if embedding_initializer is not None:
            input_initializers.append(("embedding_matrix_0", embedding_initializer))
        self.data_id = data_id
        input_sequence = EmbeddedSequence(
            name="{}_input".format(name), vocabulary=vocabulary, data_id=data_id,
            embedding_size=embedding_size, max_length=max_input_len, save_checkpoint=s_ckp,
            load_checkpoint=l_ckp, initializers=input_initializers)
        RecurrentEncoder.__init__(
            self, name=name, input_sequence=input_sequence,
            rnn_layers=[(rnn_size, rnn_direction, rnn_cell)], add_residual=add_residual,
            add_layer_norm=add_layer_norm, dropout_keep_prob=dropout_keep_prob, reuse=reuse,
            save_checkpoint=save_checkpoint, load_checkpoint=load_checkpoint, initializers=initializers)


class FactoredEncoder(RecurrentEncoder):
    # pylint: disable=too-many-arguments,too-many-locals
    def __init__(self, name: str, vocabularies: List[Vocabulary], data_ids: List[str], embedding_sizes: List[int],
                 rnn_size: int, rnn_cell: str = "GRU", rnn_direction: str = "bidirectional",
                 add_residual: bool = False, add_layer_norm: bool = False, max_input_len: int = None,
                 dropout_keep_prob: float = 1.0, reuse: ModelPart = None, save_checkpoint: str = None,
                 load_checkpoint: str = None, initializers: InitializerSpecs = None,
                 input_initializers: InitializerSpecs = None) -> None:
        """recurrent.py:317-391: an EmbeddedFactorSequence named ``<name>_input`` (one embedding matrix per
        factor, concatenated along the feature axis) feeding one RNN layer (tests/factored.ini)."""
        from ..model.sequence import EmbeddedFactorSequence
        s_ckp = "input_{}".format(save_checkpoint) if save_checkpoint else None
        l_ckp = "input_{}".format(load_checkpoint) if load_checkpoint else None
        input_sequence = EmbeddedFactorSequence(
            name="{}_input".format(name), vocabularies=vocabularies, data_ids=data_ids,
            embedding_sizes=embedding_sizes, max_length=max_input_len, save_checkpoint=s_ckp,
            load_checkpoint=l_ckp, initializers=input_initializers)
        RecurrentEncoder.__init__(
            self, name=name, input_sequence=input_sequence,
            rnn_layers=[(rnn_size, rnn_direction, rnn_cell)], add_residual=add_residual,
            add_layer_norm=add_layer_norm, dropout_keep_prob=dropout_keep_prob, reuse=reuse,
            save_checkpoint=save_checkpoint, load_checkpoint=load_checkpoint, initializers=initializers)
