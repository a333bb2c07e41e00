"""Initial-state projections (mirror of neuralmonkey/decoders/encoder_projection.py).

``apply`` / ``backward`` serve the hand-scheduled fast path of the decoder,
``apply_var`` is the same arithmetic on an autodiff tape (general path)."""
from .. import autodiff as F
from .. import ops
from ..nn.dropout import dropout
from ..variables import zeros_initializer


class EncoderProjection:
    dropout_keep_prob = 1.0

    def output_size(self, rnn_size, encoders) -> int:
        raise NotImplementedError

    def declare_variables(self, decoder, store, rnn_size, encoders) -> None:
        pass

    def apply(self, ctx, decoder, rnn_size, encoders, out, train_mode):
        raise NotImplementedError

    def backward(self, ctx, decoder, rnn_size, encoders, d_state):
        """Accumulate variable gradients; return dL/d(encoder.output) per encoder."""
        return [None for _ in encoders]

    def apply_var(self, tape, decoder, rnn_size, enc_outputs, bsz: int, train_mode: bool):
        """enc_outputs: one Var [B,D] per encoder -> initial state Var [B,rnn_size]."""
        raise NotImplementedError


class _Empty(EncoderProjection):
    """empty_initial_state (encoder_projection.py:37-44): zeros, tiled to the batch."""

    def output_size(self, rnn_size, encoders):
        if rnn_size is None:
            raise ValueError("You must supply rnn_size for this type of encoder projection")
        return rnn_size

    def apply(self, ctx, decoder, rnn_size, encoders, out, train_mode):
        ops.zero(out)
        return out

    def apply_var(self, tape, decoder, rnn_size, enc_outputs, bsz, train_mode):
        return tape.leaf(tape.buf((bsz, rnn_size), zero=True))


class _Concat(EncoderProjection):
    """concat_encoder_projection (encoder_projection.py:76-96)."""

    def output_size(self, rnn_size, encoders):
        if not encoders:
            raise ValueError("There must be at least one encoder for this type of encoder projection")
        total = sum(e.output_size for e in encoders)
        if rnn_size is not None and rnn_size != total:
            raise ValueError("RNN size supplied for concat projection ({}) does not match the size of "
                             "the concatenated vectors ({}).".format(rnn_size, total))
        return total

    def apply(self, ctx, decoder, rnn_size, encoders, out, train_mode):
        col = 0
        for enc in encoders:
            val = enc.output(ctx)
            ops.copy_cols(val, out[:, col:col + val.shape[1]])
            col += val.shape[1]
        return out

    def backward(self, ctx, decoder, rnn_size, encoders, d_state):
        grads, col = [], 0
        for enc in encoders:
            sz = enc.output_size
            grads.append(d_state[:, col:col + sz].contiguous())
            col += sz
        return grads

    def apply_var(self, tape, decoder, rnn_size, enc_outputs, bsz, train_mode):
        if len(enc_outputs) == 1:          # a copy, so that later in-place use cannot alias the encoder
            return F.copy(tape, enc_outputs[0])
        return F.concat(tape, enc_outputs)


class _Linear(EncoderProjection):
    """linear_encoder_projection (encoder_projection.py:47-73):
    dropout(dense(concat(encoder outputs), rnn_size, name="encoders_projection"))."""

    def __init__(self, dropout_keep_prob: float):
        self.dropout_keep_prob = dropout_keep_prob

    def output_size(self, rnn_size, encoders):
        if rnn_size is None:
            raise ValueError("You must supply rnn_size for this type of encoder projection")
        return rnn_size

    def declare_variables(self, decoder, store, rnn_size, encoders):
        total = sum(e.output_size for e in encoders)
        decoder.declare(store, "initial_state/encoders_projection/kernel", (total, rnn_size))
        decoder.declare(store, "initial_state/encoders_projection/bias", (rnn_size,), zeros_initializer())

    def apply(self, ctx, decoder, rnn_size, encoders, out, train_mode):
        w = decoder.var(ctx, "initial_state/encoders_projection/kernel")
        b = decoder.var(ctx, "initial_state/encoders_projection/bias")
        row = 0
        for i, enc in enumerate(encoders):
            val = enc.output(ctx)
            last = i == len(encoders) - 1
            ops.gemm(val, w[row:row + val.shape[1]], out=out, accumulate=i > 0, bias=b if last else None)
            row += val.shape[1]
        return dropout(ctx, out, self.dropout_keep_prob, train_mode)

    def backward(self, ctx, decoder, rnn_size, encoders, d_state):
        store = ctx.store
        w = decoder.var(ctx, "initial_state/encoders_projection/kernel")
        g_w = store.g(decoder.var_name("initial_state/encoders_projection/kernel"))
        acc = decoder.shares_variables         # decoders sharing the scope (reuse=) add their gradients up
        ops.colsum(d_state, store.g(decoder.var_name("initial_state/encoders_projection/bias")), accumulate=acc)
        grads, row = [], 0
        for i, enc in enumerate(encoders):
            val = enc.output(ctx)
            sz = val.shape[1]
            ops.gemm(val, d_state, out=g_w[row:row + sz], trans_a=True, accumulate=acc)
            d_val = ctx.buffer((id(self), "d_enc", i), tuple(val.shape))
            ops.gemm(d_state, w[row:row + sz], out=d_val, trans_b=True)
            grads.append(d_val)
            row += sz
        return grads

    def apply_var(self, tape, decoder, rnn_size, enc_outputs, bsz, train_mode):
        w = tape.param(decoder, "initial_state/encoders_projection/kernel")
        b = tape.param(decoder, "initial_state/encoders_projection/bias")
        out, row = None, 0
        for val in enc_outputs:
            sz = val.shape[1]
            out = F.linear(tape, val, tape.rows(w, row, row + sz), b if out is None else None, out=out,
                           accumulate=out is not None)
            row += sz
        return F.dropout(tape, out, self.dropout_keep_prob, train_mode,
                         tape.ctx.salt(decoder.name, "encoders_projection"))


empty_initial_state = _Empty()
concat_encoder_projection = _Concat()


def linear_encoder_projection(dropout_keep_prob: float) -> EncoderProjection:
    return _Linear(dropout_keep_prob)


class _Nematus(EncoderProjection):
    """nematus_projection (encoder_projection.py:99-145): tanh(dense(dropout(mean over the valid
    positions of the encoder states))).  The masked mean is a batched [1,S] x [S,D] GEMM with the
    host-built weights mask/length; training runs on the tape (no hand-scheduled backward)."""
    fast_path = False

    def __init__(self, dropout_keep_prob: float):
        self.dropout_keep_prob = dropout_keep_prob

    def output_size(self, rnn_size, encoders):
        if rnn_size is None:
            raise ValueError("You must supply rnn_size for this type of encoder projection")
        return rnn_size

    @staticmethod
    def _encoder(encoders):
        if len(encoders) != 1:
            raise ValueError("Exactly one encoder required for this type of projection. {} given."
                             .format(len(encoders)))
        return encoders[0]

    def declare_variables(self, decoder, store, rnn_size, encoders):
        from ..variables import orthogonal_initializer
        dim = self._encoder(encoders).dimension
        decoder.declare(store, "initial_state/encoders_projection/kernel", (dim, rnn_size),
                        orthogonal_initializer() if dim == rnn_size else None)
        decoder.declare(store, "initial_state/encoders_projection/bias", (rnn_size,), zeros_initializer())

    @staticmethod
    def _mean_weights(ctx, encoder):
        """[B,1,S] = mask / length (host arithmetic on the fed mask, then resident on the device)."""
        mask = encoder.temporal_mask(ctx)
        lens = mask.sum(1, keepdim=True)                 # plumbing: B scalars
        return (mask / lens).unsqueeze(1).contiguous()

    def apply(self, ctx, decoder, rnn_size, encoders, out, train_mode):
        enc = self._encoder(encoders)
        states = enc.temporal_states(ctx)
        bsz, slen, dim = states.shape
        means = ctx.buffer((id(self), "means", bsz), (bsz, 1, dim))
        ops.gemm(self._mean_weights(ctx, enc), states, out=means)
        means2 = dropout(ctx, means.view(bsz, dim), self.dropout_keep_prob, train_mode)
        return ops.gemm(means2, decoder.var(ctx, "initial_state/encoders_projection/kernel"), out=out,
                        bias=decoder.var(ctx, "initial_state/encoders_projection/bias"), act="tanh")

    def apply_var(self, tape, decoder, rnn_size, enc_outputs, bsz, train_mode):
        ctx = tape.ctx
        enc = self._encoder(decoder.encoders)
        states = enc.temporal_states(ctx)
        _, slen, dim = states.shape
        weights = self._mean_weights(ctx, enc)
        st_var = tape.leaf(states.reshape(bsz * slen, dim), needs_grad=True)
        self.states_var = st_var                          # the decoder adds its gradient to d(encoder states)
        means = tape.new((bsz, dim))
        ops.gemm(weights, states, out=means.data.view(bsz, 1, dim))

        def bwd():
            if means.grad is not None:
                ops.gemm(weights, means.grad.view(bsz, 1, dim), out=tape.grad(st_var).view(bsz, slen, dim),
                         trans_a=True, accumulate=True)
        tape.record(bwd)
        dropped = F.dropout(tape, means, self.dropout_keep_prob, train_mode,
                            ctx.salt(decoder.name, "encoders_projection"))      # (bwd closes over ``means``)
        pre = F.linear(tape, dropped, tape.param(decoder, "initial_state/encoders_projection/kernel"),
                       tape.param(decoder, "initial_state/encoders_projection/bias"))
        return F.tanh(tape, pre)


def nematus_projection(dropout_keep_prob: float = 1.0) -> EncoderProjection:
    return _Nematus(dropout_keep_prob)
