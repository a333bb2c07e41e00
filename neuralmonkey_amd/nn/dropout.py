"""nn/utils.py:6-22.  Dropout cannot reproduce TF's Philox stream, so parity
holds for keep_prob == 1 or train=False; with keep_prob < 1 in train mode the
HIP engine refuses loudly instead of silently training a different model."""


def dropout(ctx, variable, keep_prob: float, train_mode: bool):
    if keep_prob <= 0.0 or keep_prob > 1.0:
        raise ValueError("keep_prob must be a scalar tensor or a float in the range (0, 1], got {}"
                         .format(keep_prob))
    if keep_prob == 1.0 or not train_mode:
        return variable
    raise NotImplementedError(
        "dropout_keep_prob < 1 in training mode is not implemented in the HIP engine yet "
        "(set dropout_keep_prob=1.0)")
