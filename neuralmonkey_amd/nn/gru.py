"""GRU time-step drivers shared by the encoder and the decoder.

TF GRUCell (nn/ortho_gru_cell.py:44-53) split so both matrix products run on
MFMA: the input half is hoisted (xp), a step is the two recurrent GEMMs.  When
the shapes allow it (K % 8 == 0) the gate / blend arithmetic is fused into the
GEMM kernels as epilogues (nm_gru_gemm): 2 launches per step forward and
backward; otherwise GEMM + stand-alone epilogue kernels (4 launches)."""
from .. import ops


def fused_ok(rows: int, hsz: int) -> bool:
    return hsz % 8 == 0 and rows <= 1024


_CLUSTER_OK = {}


def cluster_ok(session, rows: int, hsz: int, ndir: int, wgh, wch) -> bool:
    """Does this time loop run as one cluster launch (ops.gru_seq_fwd / gru_seq_bwd)?  The shape must be one the
    kernels take on this device, the recurrent kernels 16-byte aligned row by row (their fragments are 16-byte loads)."""
    if not session.use_cluster_loops:
        return False
    for w in (wgh, wch):
        if w.data_ptr() % 16 or any(st % 4 for st in w.stride()[:-1]) or w.stride(-1) != 1:
            return False
    key = (rows, hsz, ndir)
    if key not in _CLUSTER_OK:
        _CLUSTER_OK[key] = ops.gru_seq_supported(rows, hsz, ndir)
    return _CLUSTER_OK[key]


def cluster_workspace(ctx, key, rows: int, hsz: int, ndir: int):
    """Hand-off buffers of one module's cluster loops (forward and backward run one after the other)."""
    return ctx.buffer((key, "cluster_ws"), (ops.gru_seq_workspace_floats(rows, hsz, ndir),))


def transposed_weights(ctx, key, wgh, wch):
    """[ndir,H,2H] / [ndir,H,H] recurrent kernels -> persistent transposed copies [ndir,2H,H] / [ndir,H,H]
    ([N,K]: both MFMA fragments of a wave of the skinny kernel are then single 16-byte loads and the kernel
    stays within 64 VGPRs, i.e. every workgroup of a step is resident at once).  Refreshed by the caller once
    per time loop -- the weights change with every optimizer step."""
    g3 = wgh if wgh.dim() == 3 else wgh.unsqueeze(0)
    c3 = wch if wch.dim() == 3 else wch.unsqueeze(0)
    wg_t = ctx.buffer((key, "wgh_t"), (g3.shape[0], g3.shape[2], g3.shape[1]))
    wc_t = ctx.buffer((key, "wch_t"), (c3.shape[0], c3.shape[2], c3.shape[1]))
    ops.copy(wg_t, g3.transpose(1, 2))
    ops.copy(wc_t, c3.transpose(1, 2))
    return (wg_t, wc_t) if wgh.dim() == 3 else (wg_t[0], wc_t[0])


def step_fwd(xp, x_strides, h_in, h_out, wgh, wch, ru, rh, c_save, out, out_strides, lengths, t, ndir, rows,
             hsz, rev0, hg, hc, transposed=False):
    """h_out = GRU(x_t, h_in) for ``ndir`` directions; tensors are [ndir,R,*]
    (2-D accepted when ndir == 1).  ``transposed``: wgh / wch are the [N,K] copies of ``transposed_weights``."""
    if fused_ok(rows, hsz):
        ops.gru_gemm(1, h_in, wgh, transposed, t, ndir, rows, hsz, lengths, rev0, xp=xp, x_strides=x_strides,
                     h_in=h_in, ru=ru, rh=rh)
        ops.gru_gemm(2, rh, wch, transposed, t, ndir, rows, hsz, lengths, rev0, xp=xp, x_strides=x_strides,
                     h_in=h_in, h_out=h_out, ru=ru, c_save=c_save, out=out, out_strides=out_strides)
        return
    assert not transposed
    ops.gemm(h_in, wgh, out=hg)
    ops.gru_gates_fwd(xp, x_strides[0], x_strides[1], x_strides[2], hg, h_in, ru, rh, lengths, t, ndir, rows,
                      hsz, reverse_dir0=rev0)
    ops.gemm(rh, wch, out=hc)
    ops.gru_blend_fwd(xp, x_strides[0], x_strides[1], x_strides[2], hc, ru, h_in, h_out, c_save, out,
                      out_strides[0], out_strides[1], out_strides[2], lengths, t, ndir, rows, hsz,
                      reverse_dir0=rev0)


def bptt(steps, dh, dout, dout_strides, ru_all, c_all, h0, hseq, hseq_strides, dxp, dxp_strides, wgh, wch,
         lengths, ndir, rows, hsz, rev0, dgpre2, dcpre, drh):
    """Back-propagation through ``steps`` GRU steps (last to first).  ``dh``
    [ndir,R,H] holds dL/dh after the last step on entry and dL/dh_0 on exit;
    pre-activation gradients land in ``dxp``.  ``dgpre2`` is a [2,ndir,R,2H]
    ping-pong pair (the fused blend epilogue of step t-1 writes one while the
    GEMM of step t still reads the other)."""
    wgh2 = wgh if wgh.dim() == 3 else wgh
    if not fused_ok(rows, hsz):
        dg = dgpre2[0]
        for t in range(steps - 1, -1, -1):
            ops.gru_step_bwd(0, dh, dout, dout_strides, ru_all[t], c_all[t], h0, hseq, hseq_strides, dxp,
                             dxp_strides, dg, dcpre, None, lengths, t, ndir, rows, hsz, reverse_dir0=rev0)
            ops.gemm(_b(dcpre, ndir), wch, out=_b(drh, ndir), trans_b=True)
            ops.gru_step_bwd(1, dh, None, None, ru_all[t], None, h0, hseq, hseq_strides, dxp, dxp_strides, dg,
                             None, drh, lengths, t, ndir, rows, hsz, reverse_dir0=rev0)
            ops.gemm(_b(dg, ndir), wgh2, out=_b(dh, ndir), trans_b=True, accumulate=True)
        return
    cur = 0
    t_last = steps - 1
    ops.gru_step_bwd(0, dh, dout, dout_strides, ru_all[t_last], c_all[t_last], h0, hseq, hseq_strides, dxp,
                     dxp_strides, dgpre2[cur], dcpre, None, lengths, t_last, ndir, rows, hsz, reverse_dir0=rev0)
    for t in range(steps - 1, -1, -1):
        ops.gru_gemm(3, _b(dcpre, ndir), wch, True, t, ndir, rows, hsz, lengths, rev0, dh=dh, ru=ru_all[t], h0=h0,
                     hseq=hseq, hseq_strides=hseq_strides, dxp=dxp, dxp_strides=dxp_strides, dgpre=dgpre2[cur])
        if t > 0:
            ops.gru_gemm(4, _b(dgpre2[cur], ndir), wgh2, True, t - 1, ndir, rows, hsz, lengths, rev0, dh=dh,
                         dout=dout, dout_strides=dout_strides or (0, 0, 0), ru=ru_all[t - 1], c=c_all[t - 1],
                         h0=h0, hseq=hseq, hseq_strides=hseq_strides, dxp=dxp, dxp_strides=dxp_strides,
                         dgpre=dgpre2[cur ^ 1], dcpre=dcpre)
        else:
            ops.gemm(_b(dgpre2[cur], ndir), wgh2, out=_b(dh, ndir), trans_b=True, accumulate=True)
        cur ^= 1


def _b(x, ndir):
    """[1,R,*] -> [R,*] when the weights are a plain 2-D view (single direction)."""
    return x
