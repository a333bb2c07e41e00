"""nn/utils.py:6-22 on the hand-scheduled (untaped) fast paths: identity at keep_prob 1 or outside training.

Training-mode dropout itself IS implemented -- ``autodiff.dropout`` (counter-based masks of ``nm_dropout``, the same
mask re-derived in the backward pass; TF's Philox stream is not reproducible, so the masks differ from the
reference's while the distribution is the same).  Parts configured with dropout_keep_prob < 1 are routed to the taped
general path, which calls that one; this function is what the fast paths call, and they refuse loudly instead of
silently training a different model if such a configuration ever reaches them."""


def dropout(ctx, variable, keep_prob: float, train_mode: bool):
    if keep_prob <= 0.0 or keep_prob > 1.0:
        raise ValueError("keep_prob must be a scalar tensor or a float in the range (0, 1], got {}"
                         .format(keep_prob))
    if keep_prob == 1.0 or not train_mode:
        return variable
    raise NotImplementedError(
        "dropout_keep_prob < 1 in training mode reached a hand-scheduled fast path; such models train on the "
        "taped general path (autodiff.dropout)")
