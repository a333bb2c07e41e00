"""Generate the committed golden vectors from the CPU oracle.

The reference holds no golden vectors for this path, so these freeze the ORACLE'S outputs against silent
drift.  (Since round 4 the oracle itself is pinned to the reference's own code: see oracle/nm_oracle.py and
tests/golden/make_reference_exec_golden.py.)  Re-run only when the oracle is deliberately changed:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nm_oracle as O          # noqa: E402
from oracle import torch_ref as TR         # noqa: E402

CASES = {
    # name: (vocab, dim, batch, src_len, tgt_len, ragged, beam, alpha, seed)
    "tiny": (64, 12, 5, 7, 6, True, 3, 0.6, 21),
    "mid": (400, 32, 8, 14, 10, True, 5, 1.0, 22),
}


def build(name):
    vocab, dim, batch, slen, tlen, ragged, beam, alpha, seed = CASES[name]
    params = O.init_params(seed=seed, vocab_src=vocab, vocab_tgt=vocab, emb=dim, rnn=dim, std=0.1)
    src, tgt = O.synthetic_batch(seed=seed + 1, batch=batch, src_len=slen, tgt_len=tlen, vocab=vocab,
                                 ragged=ragged)
    enc = O.sentence_encoder(params, src)
    spec = O.DecoderSpec(max_output_len=max(slen, tlen))
    greedy = O.decoding_loop(params, spec, enc, None, False)
    train = O.decoding_loop(params, spec, enc, tgt, True)
    beam_res = O.beam_search(params, spec, enc, beam, tlen, alpha)
    loss, l1, l2, grads = TR.train_step_grads(TR.to_torch(params), src, tgt, l1_weight=0.0, l2_weight=1e-8)
    out = {"src": src, "tgt": tgt, "enc_states": enc.temporal_states, "enc_final": enc.output,
           "greedy_symbols": greedy.symbols, "greedy_logits": greedy.logits,
           "train_loss": np.float32(O.train_loss(train, tgt)),
           "runtime_loss": np.float32(O.runtime_loss(greedy, tgt)),
           "beam_token_ids": beam_res.token_ids, "beam_scores": beam_res.scores,
           "beam_min_gap": np.float64(beam_res.min_gap), "torch_loss": np.float32(loss),
           "l2": np.float32(l2),
           "grad_logit_w": grads["decoder/state_to_word_W"].numpy(),
           "grad_attn_v": grads["attention/attn_similarity_v"].numpy(),
           "grad_enc_emb": grads["encoder_input/embedding_matrix_0"].numpy(),
           "meta": np.array([vocab, dim, batch, slen, tlen, int(ragged), beam, seed], dtype=np.int64),
           "alpha": np.float64(alpha)}
    return params, out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for case in CASES:
        _, out = build(case)
        np.savez_compressed(os.path.join(here, "{}.npz".format(case)), **out)
        print(case, {k: getattr(v, "shape", v) for k, v in out.items() if k != "meta"})
