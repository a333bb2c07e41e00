"""nm_nematus_state_step -- the state product of a NematusGRUCell step and its point-wise part in one launch -- against
float64 arithmetic of the reference's cell (nn/ortho_gru_cell.py:73-105: the reset gate multiplies the state projection
of the candidate) and against the two launches it replaces (nm_gemm_f32 + nm_nematus_cell_fwd), over ragged row counts,
unit counts off the 16-unit tiles, strided operands as the decoder hands them over (rows of all-steps buffers), with
and without the bias and the outputs the backward pass keeps.  Tolerance 2e-6 of each output's largest magnitude."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,h", [(1, 8), (5, 24), (16, 16), (37, 264), (128, 512), (64, 384), (130, 40), (640, 512)])
@pytest.mark.parametrize("bias,keep", [(True, True), (False, False)])
def test_state_step_matches_float64_and_the_two_launches(dev, rows, h, bias, keep):
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(rows * 1000 + h)
    mk = lambda *shape, s=1.0: torch.tensor((rng.standard_normal(shape) * s).astype(np.float32), device=dev)
    h_all = mk(2, rows, h)                                     # previous / new state: rows of one buffer
    h_prev, h_new = h_all[0], h_all[1]
    w_st = mk(h, 3 * h, s=1.0 / np.sqrt(h))
    b_st = mk(3 * h) if bias else None
    x_wide = mk(rows, 3 * h + 4)
    x_all = x_wide[:, :3 * h]                                  # strided rows
    ru = torch.full((rows, 2 * h), 9.0, device=dev) if keep else None
    c = torch.full((rows, h), 9.0, device=dev) if keep else None
    s_all = torch.full((rows, 3 * h), 9.0, device=dev)
    sc = s_all[:, 2 * h:] if keep else None
    # (the kernel takes any size; the engine asks for it up to 512 tiles of 16 rows x 16 units)
    assert ops.nematus_state_step_ok(h_prev, w_st, h_new) == (rows < 640)
    ops.nematus_state_step(h_prev, w_st, b_st, x_all, h_new, ru, c, sc)

    s64 = h_prev.double() @ w_st.double() + (b_st.double() if bias else 0.0)
    x64 = x_all.double()
    r64 = torch.sigmoid(x64[:, :h] + s64[:, :h])
    u64 = torch.sigmoid(x64[:, h:2 * h] + s64[:, h:2 * h])
    c64 = torch.tanh(x64[:, 2 * h:] + r64 * s64[:, 2 * h:])
    hn64 = u64 * h_prev.double() + (1.0 - u64) * c64

    def close(got, want, tol=2e-6):
        scale = max(float(want.abs().max()), 1.0)
        assert float((got.double() - want).abs().max()) <= tol * scale

    close(h_new, hn64)
    if keep:
        close(ru[:, :h], r64)
        close(ru[:, h:], u64)
        close(c, c64)
        close(sc, s64[:, 2 * h:])
        assert float(s_all[:, :2 * h].min()) == 9.0            # the gates' columns of the state projection are not kept
    # the two launches this replaces
    s2 = torch.empty(rows, 3 * h, device=dev)
    hn2 = torch.empty(rows, h, device=dev)
    ops.gemm(h_prev, w_st, out=s2, bias=b_st)
    ops.nematus_cell_fwd(s2[:, :2 * h], s2[:, 2 * h:], x_all[:, 2 * h:], h_prev, hn2, None, None, g2=x_all[:, :2 * h])
    close(h_new, hn2.double())


def test_state_step_refuses_what_it_cannot_take(dev):
    from neuralmonkey_amd import ops
    h_prev, h_new = torch.zeros(4, 12, device=dev), torch.zeros(4, 12, device=dev)
    assert not ops.nematus_state_step_ok(h_prev, torch.zeros(12, 36, device=dev), h_new)
    h8 = torch.zeros(4, 8, device=dev)
    assert not ops.nematus_state_step_ok(h8, torch.zeros(8, 24, device=dev), h8)   # in place


@pytest.mark.parametrize("rows,h,d", [(1, 8, 8), (5, 24, 40), (37, 264, 136), (128, 512, 1024), (64, 384, 768), (130, 40, 24)])
@pytest.mark.parametrize("bias,keep", [(True, True), (False, False)])
def test_full_step_matches_float64(dev, rows, h, d, bias, keep):
    """nm_nematus_full_step: the same with the input half x . [W_g | W_c] computed in the launch (the second cell of a
    conditional decoder, whose input is the step's own attention context)."""
    from neuralmonkey_amd import ops
    rng = np.random.default_rng(rows * 1000 + h + d)
    mk = lambda *shape, s=1.0: torch.tensor((rng.standard_normal(shape) * s).astype(np.float32), device=dev)
    h_all = mk(2, rows, h)
    h_prev, h_new = h_all[0], h_all[1]
    w_st, w_in = mk(h, 3 * h, s=1.0 / np.sqrt(h)), mk(d, 3 * h, s=1.0 / np.sqrt(d))
    b_st, b_in = (mk(3 * h), mk(3 * h)) if bias else (None, None)
    x = mk(rows, d + 4)[:, :d]                                 # strided rows
    ru = torch.full((rows, 2 * h), 9.0, device=dev) if keep else None
    c = torch.full((rows, h), 9.0, device=dev) if keep else None
    s_all = torch.full((rows, 3 * h), 9.0, device=dev)
    sc = s_all[:, 2 * h:] if keep else None
    assert ops.nematus_state_step_ok(h_prev, w_st, h_new) and ops.nematus_full_step_ok(x, w_in)
    ops.nematus_full_step(h_prev, w_st, b_st, x, w_in, b_in, h_new, ru, c, sc)

    s64 = h_prev.double() @ w_st.double() + (b_st.double() if bias else 0.0)
    x64 = x.double() @ w_in.double() + (b_in.double() if bias else 0.0)
    r64 = torch.sigmoid(x64[:, :h] + s64[:, :h])
    u64 = torch.sigmoid(x64[:, h:2 * h] + s64[:, h:2 * h])
    c64 = torch.tanh(x64[:, 2 * h:] + r64 * s64[:, 2 * h:])
    hn64 = u64 * h_prev.double() + (1.0 - u64) * c64

    def close(got, want, tol=3e-6):
        scale = max(float(want.abs().max()), 1.0)
        assert float((got.double() - want).abs().max()) <= tol * scale

    close(h_new, hn64)
    if keep:
        close(ru[:, :h], r64)
        close(ru[:, h:], u64)
        close(c, c64)
        close(sc, s64[:, 2 * h:])
