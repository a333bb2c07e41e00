"""Building blocks shared by the Transformer encoder and decoder
(encoders/transformer.py:199-288, decoders/transformer.py:270-390,
attention/transformer_cross_layer.py:12-103) on the autodiff tape.

Every dense layer is an MFMA GEMM over all B*T rows; the attention core is the fused
``nm_sdp_attn_fwd`` kernel.  Variable names follow the TF scopes of the reference
(``layer_<i>/self_attention/query_proj/kernel`` ...), so TF checkpoints map one to one."""
import math
import os
from typing import Optional

import numpy as np
import torch

from .. import autodiff as F
from ..variables import ones_initializer, zeros_initializer


FF_RELU_FUSED = os.environ.get("NM_FF_RELU_FUSED", "1") != "0"


def position_signal(dimension: int, length: int) -> np.ndarray:
    """encoders/transformer.py:23-45 (tensor2tensor timing signal): [length, dimension] float32."""
    positions = np.arange(length, dtype=np.float32)
    num_timescales = dimension // 2
    log_timescale_increment = math.log(1.0e4) / (num_timescales - 1)
    inv_timescales = np.exp(np.arange(num_timescales, dtype=np.float32) * np.float32(-log_timescale_increment))
    scaled_time = positions[:, None] * inv_timescales[None, :]
    signal = np.concatenate([np.sin(scaled_time), np.cos(scaled_time)], axis=1).astype(np.float32)
    if dimension % 2:
        signal = np.pad(signal, [[0, 0], [0, 1]])
    return signal


def signal_table(ctx, dimension: int, length: int) -> torch.Tensor:
    """Device-resident position signal, cached per session (grown geometrically)."""
    cache = ctx.session.__dict__.setdefault("_const", {})
    key = ("position_signal", dimension)
    tab = cache.get(key)
    if tab is None or tab.shape[0] < length:
        size = max(64, 2 * length)
        tab = torch.from_numpy(position_signal(dimension, size)).to(ctx.device)
        cache[key] = tab
    return tab


def declare_layer_norm(part, store, scope: str, dim: int) -> None:
    part.declare(store, scope + "/LayerNorm/gamma", (dim,), ones_initializer())
    part.declare(store, scope + "/LayerNorm/beta", (dim,), zeros_initializer())


def layer_norm(tape: F.Tape, part, scope: str, x: F.Var) -> F.Var:
    return F.layer_norm(tape, x, tape.param(part, scope + "/LayerNorm/gamma"),
                        tape.param(part, scope + "/LayerNorm/beta"))


PROJECTIONS = ("query_proj", "keys_proj", "vals_proj", "output_proj")


def declare_attention(part, store, scope: str, dim: int, heads: int, use_bias: bool) -> None:
    """attention() creates its four dense layers only for num_heads > 1 (scaled_dot_product.py:170-177,217-223)."""
    if heads <= 0:
        raise ValueError("Number of heads must be greater than zero.")
    if dim % heads != 0:
        raise ValueError("Last dimension of the query ({}) should be divisible by the number of heads ({})"
                         .format(dim, heads))
    if heads == 1:
        return
    for proj in PROJECTIONS:
        part.declare(store, "{}/{}/kernel".format(scope, proj), (dim, dim))
        if use_bias:
            part.declare(store, "{}/{}/bias".format(scope, proj), (dim,), zeros_initializer())


def project(tape: F.Tape, part, scope: str, proj: str, x: F.Var, heads: int, use_bias: bool) -> F.Var:
    if heads == 1:
        return x
    bias = tape.param(part, "{}/{}/bias".format(scope, proj)) if use_bias else None
    return F.linear(tape, x, tape.param(part, "{}/{}/kernel".format(scope, proj)), bias)


def multihead_attention(tape: F.Tape, part, scope: str, queries: F.Var, keys: F.Var, key_mask: Optional[torch.Tensor],
                        heads: int, bq: int, tq: int, bk: int, tk: int, causal: bool, keep_prob: float,
                        train: bool, salt: int, use_bias: bool) -> F.Var:
    """attention() of scaled_dot_product.py:98-226 with keys == values."""
    if heads > 1 and not use_bias and queries.shape[0] >= 1024:
        # training / encoding shapes: projections that read the same rows go out as one batched product
        kern = lambda proj: tape.param(part, "{}/{}/kernel".format(scope, proj))
        if queries is keys:
            q, k, v = F.linear_multi(tape, queries, [kern("query_proj"), kern("keys_proj"), kern("vals_proj")])
        else:
            q = project(tape, part, scope, "query_proj", queries, heads, use_bias)
            k, v = F.linear_multi(tape, keys, [kern("keys_proj"), kern("vals_proj")])
    else:
        q = project(tape, part, scope, "query_proj", queries, heads, use_bias)
        k = project(tape, part, scope, "keys_proj", keys, heads, use_bias)
        v = project(tape, part, scope, "vals_proj", keys, heads, use_bias)
    ctx = F.sdp_attention(tape, q, k, v, key_mask, heads, bq, tq, bk, tk, causal,
                          keep_prob if train else 1.0, salt)
    return project(tape, part, scope, "output_proj", ctx, heads, use_bias)


def declare_feedforward(part, store, scope: str, dim: int, hidden: int) -> None:
    declare_layer_norm(part, store, scope, dim)
    part.declare(store, scope + "/hidden_state/kernel", (dim, hidden))
    part.declare(store, scope + "/hidden_state/bias", (hidden,), zeros_initializer())
    part.declare(store, scope + "/output/kernel", (hidden, dim))
    part.declare(store, scope + "/output/bias", (dim,), zeros_initializer())


def feedforward_sublayer(tape: F.Tape, part, scope: str, x: F.Var, keep: float, train: bool, site) -> F.Var:
    """encoders/transformer.py:262-288 == decoders/transformer.py:334-358: pre-LN, ReLU hidden layer,
    dropout on the hidden activations and on the output, residual."""
    ctx = tape.ctx
    normed = layer_norm(tape, part, scope, x)
    # (the ReLU runs in the product's epilogue; NM_FF_RELU_FUSED=0: a launch of its own as in rounds 1-5)
    w_h, b_h = tape.param(part, scope + "/hidden_state/kernel"), tape.param(part, scope + "/hidden_state/bias")
    if FF_RELU_FUSED and normed.data.is_cuda:
        hidden = F.linear(tape, normed, w_h, b_h, act="relu")
    else:
        hidden = F.relu(tape, F.linear(tape, normed, w_h, b_h))
    hidden = F.dropout(tape, hidden, keep, train, ctx.salt(*site, "ff_hidden"))
    out = F.linear(tape, hidden, tape.param(part, scope + "/output/kernel"), tape.param(part, scope + "/output/bias"))
    out = F.dropout(tape, out, keep, train, ctx.salt(*site, "ff_output"))
    return F.add(tape, out, x)
