"""torch-CPU restatement of the training step (forward + autograd backward).

TEST INFRASTRUCTURE ONLY (see oracle/nm_oracle.py header: parity pinned to the reference's own code;
this module is held to ``nm_oracle`` and, through the ``fd_gradients_*`` fixtures, to finite differences of the
reference's loss).
It restates the same reference call sites as ``nm_oracle`` but in torch ops so
that (a) gradients come from autograd of the restated forward -- the
reference gets them from ``tf.gradients`` (trainers/generic_trainer.py:136-142)
-- and (b) ``bench.py``'s ``cpu_baseline`` leg can time the reference's op
granularity (per-step cell / attention / projection / logits, no fusion) on the
host cores.  ``tests/test_oracle.py`` checks this forward against the NumPy
oracle to ~1e-6.
"""
from typing import Dict, Optional

import numpy as np
import torch

from . import nm_oracle as O


def to_torch(params: Dict[str, np.ndarray], dtype=torch.float32, requires_grad=True):
    return {k: torch.tensor(np.asarray(v), dtype=dtype).requires_grad_(requires_grad)
            for k, v in params.items()}


def layer_norm(x, g, b, eps=1e-6):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * g + b


def gru_cell(x, h, wg, bg, wc, bc):
    hsz = h.shape[-1]
    g = torch.sigmoid(torch.cat([x, h], -1) @ wg + bg)
    r, u = g[..., :hsz], g[..., hsz:]
    c = torch.tanh(torch.cat([x, r * h], -1) @ wc + bc)
    return u * h + (1 - u) * c


def dynamic_rnn(x, lengths, wg, bg, wc, bc):
    bsz, steps, _ = x.shape
    h = x.new_zeros(bsz, bc.shape[0])
    outs = []
    for t in range(steps):
        nh = gru_cell(x[:, t], h, wg, bg, wc, bc)
        live = (t < lengths)[:, None]
        h = torch.where(live, nh, h)
        outs.append(torch.where(live, nh, torch.zeros_like(nh)))
    return torch.stack(outs, 1), h


def reverse_sequence(x, lengths):
    idx = torch.arange(x.shape[1])[None, :].repeat(x.shape[0], 1)
    rev = lengths[:, None] - 1 - idx
    idx = torch.where(rev >= 0, rev, idx)
    return torch.gather(x, 1, idx[:, :, None].expand_as(x))


def encoder(p, src_ids, name="encoder"):
    ids = torch.as_tensor(src_ids, dtype=torch.long)
    emb = p[f"{name}_input/embedding_matrix_0"]
    mask = (ids != O.PAD).to(emb.dtype)
    x = emb[ids] * mask[..., None]
    lengths = mask.sum(1).to(torch.long)
    pre = f"{name}/rnn_0_bidirectional/bidirectional_rnn"
    cp = lambda d: (p[f"{pre}/{d}/OrthoGRUCell/gates/kernel"], p[f"{pre}/{d}/OrthoGRUCell/gates/bias"],
                    p[f"{pre}/{d}/OrthoGRUCell/candidate/kernel"], p[f"{pre}/{d}/OrthoGRUCell/candidate/bias"])
    of, ff = dynamic_rnn(x, lengths, *cp("fw"))
    obr, fb = dynamic_rnn(reverse_sequence(x, lengths), lengths, *cp("bw"))
    ob = reverse_sequence(obr, lengths)
    states, final = torch.cat([of, ob], 2), torch.cat([ff, fb], 1)
    g, b = p[f"{name}/LayerNorm/gamma"], p[f"{name}/LayerNorm/beta"]
    return layer_norm(states, g, b), mask, layer_norm(final, g, b)


def attention_step(q, hf, states, mask, wq, bq, v, bias):
    y = q @ wq + bq
    e = (v * torch.tanh(hf + y[:, None, :])).sum(-1) + bias
    w_all = torch.softmax(e, -1) * mask
    w = w_all / (w_all.sum(1, keepdim=True) + 1e-8)
    return (w[:, :, None] * states).sum(1), w


def train_forward(p, src_ids, tgt_ids_tb, enc_name="encoder", dec_name="decoder",
                  att_name="attention", hoist_logits=False):
    """Teacher-forced decoder loss = sum(xent)/sum(mask) (autoregressive.py:292-316).

    The loop runs over all T rows of ``tgt_ids_tb``: the reference's while-loop
    stops when every sentence has produced </s>, which for pad_batch'ed targets
    is exactly T steps (vocabulary.py:331-354, autoregressive.py:425-437).
    """
    states, mask, final = encoder(p, src_ids, enc_name)
    tgt = torch.as_tensor(tgt_ids_tb, dtype=torch.long)
    tsteps, bsz = tgt.shape
    a, n = att_name, dec_name
    hf = states @ p[f"{a}/attn_key_projection"]
    h = final @ p[f"{n}/initial_state/encoders_projection/kernel"] + \
        p[f"{n}/initial_state/encoders_projection/bias"]
    emb = p[f"{n}/word_embeddings"]
    cell = f"{n}/attention_decoder/OrthoGRUCell"
    cw = (p[f"{cell}/gates/kernel"], p[f"{cell}/gates/bias"],
          p[f"{cell}/candidate/kernel"], p[f"{cell}/candidate/bias"])
    aw = (p[f"{a}/Attention/attn_query_projection"], p[f"{a}/attn_projection_bias"],
          p[f"{a}/attn_similarity_v"], p[f"{a}/attn_bias"])
    ow, ob = p[f"{n}/attention_decoder/dense/kernel"], p[f"{n}/attention_decoder/dense/bias"]
    lw, lb = p[f"{n}/state_to_word_W"], p[f"{n}/state_to_word_b"]
    x = emb[torch.full((bsz,), O.START, dtype=torch.long)]
    tmask = (tgt != O.PAD).to(emb.dtype)
    total = emb.new_zeros(())
    outs = []
    for t in range(tsteps):
        h = gru_cell(x, h, *cw)
        ctx, _ = attention_step(h, hf, states, mask, *aw)
        out = torch.tanh(torch.cat([h, x, ctx], 1) @ ow + ob)
        if hoist_logits:
            outs.append(out)
        else:
            logits = out @ lw + lb
            lp = torch.log_softmax(logits, -1)
            total = total - (lp.gather(1, tgt[t][:, None])[:, 0] * tmask[t]).sum()
        x = emb[tgt[t]]
    if hoist_logits:
        logits = torch.stack(outs) @ lw + lb
        lp = torch.log_softmax(logits, -1)
        total = -(lp.gather(2, tgt[:, :, None])[:, :, 0] * tmask).sum()
    return total / tmask.sum()


def regularizers(p):
    names = O.regularizable(sorted(p))
    l1 = sum(p[n].abs().sum() for n in names)
    l2 = sum((p[n] ** 2).sum() for n in names)
    return l1, l2


def train_step_grads(p, src_ids, tgt_ids_tb, l1_weight=0.0, l2_weight=0.0, **kw):
    """differentiable_loss_sum and its gradients (generic_trainer.py:118-142)."""
    for v in p.values():
        v.grad = None
    loss = train_forward(p, src_ids, tgt_ids_tb, **kw)
    l1, l2 = regularizers(p)
    total = loss + l1_weight * l1 + l2_weight * l2
    total.backward()
    grads = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v))
             for k, v in p.items()}
    return loss.detach(), l1.detach(), l2.detach(), grads


def clip_and_adam(p, grads, m, v, t, clip_norm: Optional[float], lr=1e-4,
                  b1=0.9, b2=0.999, eps=1e-8):
    """Per-tensor clip_by_norm (generic_trainer.py:179-186) then Adam
    (:55-57); mutates p/m/v in place (no_grad)."""
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    with torch.no_grad():
        for k in p:
            g = grads[k]
            if clip_norm:
                g = g * (clip_norm / max(float(g.norm()), clip_norm))
            m[k].mul_(b1).add_(g, alpha=1 - b1)
            v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            p[k].sub_(lr_t * m[k] / (v[k].sqrt() + eps))


def clip_and_adadelta(p, grads, accum, accum_update, clip_norm: Optional[float], lr=1e-3, rho=0.95, eps=1e-8):
    """Per-tensor clip_by_norm (generic_trainer.py:179-186) then tf.train.AdadeltaOptimizer as the reference's
    configs set it up (tests/bpe.ini:102-108: learning_rate 1e-4, epsilon 1e-6, rho 0.95).  The update rule is
    TensorFlow 1.12's ``ApplyAdadelta`` (tensorflow/core/kernels/training_ops.cc, a dependency that is not part of
    /root/reference -- restated from its published definition, UNPINNED by any reference-executed fixture):
        accum        = rho * accum + (1 - rho) * g^2
        update       = sqrt(accum_update + eps) * rsqrt(accum + eps) * g
        var         -= lr * update
        accum_update = rho * accum_update + (1 - rho) * update^2
    Mutates p / accum / accum_update in place."""
    with torch.no_grad():
        for k in p:
            g = grads[k]
            if clip_norm:
                g = g * (clip_norm / max(float(g.norm()), clip_norm))
            accum[k].mul_(rho).addcmul_(g, g, value=1 - rho)
            update = (accum_update[k] + eps).sqrt() * (accum[k] + eps).rsqrt() * g
            p[k].sub_(lr * update)
            accum_update[k].mul_(rho).addcmul_(update, update, value=1 - rho)
