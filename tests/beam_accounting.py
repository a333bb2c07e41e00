"""Accounting for EVERY selection of a beam search against an oracle that follows the engine's picks
(``oracle.nm_oracle.beam_search_core(follow=...)``); shared by the full-size parity tests."""
import numpy as np


def account_for_every_selection(ref, sel_beam, sel_word, tok, out, k, vocab, near_tie=1e-5):
    """``ref`` = the oracle made to FOLLOW the engine's selections: at every step it scored all k*V candidates from
    its own state, which is the state the engine's earlier picks lead to.  Every one of the steps x B x k selections is
    accounted for: the engine's k picks are the oracle's own top k in the same order (exact), or -- where the
    oracle's candidates around the k-th place lie within the near-tie margin of each other, so that two fp32
    implementations may order them differently -- they form a legitimate top k of the oracle's scores up to that
    margin: every pick scores within the margin of the oracle's k-th best or better, no candidate that was left out
    scores more than the margin above the worst pick, and the picks are in descending order up to the margin.
    Because the oracle continues from the ENGINE's picks, the comparison goes on to the last step for every
    sentence, and the final histories / lengths / flags must agree for all of them.

    Returns (share of exact (step, sentence) top-k lists, number of legitimately re-ordered lists, share of
    sentences that never met a near-tie)."""
    f = ref.follow
    steps, bsz = f["best_other"].shape
    given = sel_beam[:steps].astype(np.int64) * vocab + sel_word[:steps].astype(np.int64)          # [steps,B,k]
    exact = (f["own_idx"][:, :, :k] == given).all(-1)
    kth = f["own_scores"][:, :, k - 1]
    tol = 2 * near_tie * np.abs(kth)
    inside = (f["given_scores"] >= (kth - tol)[..., None]).all(-1)
    nothing_better_left_out = f["best_other"] <= f["given_scores"].min(-1) + tol
    ordered = (np.diff(f["given_scores"], axis=-1) <= tol[..., None]).all(-1)
    legit = inside & nothing_better_left_out & ordered
    bad = np.argwhere(~(exact | legit))
    assert bad.size == 0, ("selections neither exact nor a legitimate top-k of the oracle's scores at (step, "
                           "sentence) {}".format(bad[:8].tolist()))
    # the oracle walked the engine's path: the end state is the same search
    assert np.array_equal(tok, ref.token_ids.astype(np.int32)), "beam token ids differ"
    assert np.array_equal(np.asarray(out.last_search_state.lengths), ref.lengths)
    assert np.array_equal(np.asarray(out.last_search_state.finished).astype(bool), ref.finished)
    sc = np.asarray(out.last_search_step_output.scores)
    assert np.abs(sc - ref.scores).max() <= 1e-4 * np.abs(ref.scores).max()
    lps = np.asarray(out.last_search_state.logprob_sum)
    assert np.abs(lps - ref.logprob_sum).max() <= 1e-4 * np.abs(ref.logprob_sum).max()
    return exact.mean(), int((~exact).sum()), exact.all(axis=0).mean()
