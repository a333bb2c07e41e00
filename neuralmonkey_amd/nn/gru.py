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


# ---- hidden sizes the cluster kernels do not take (H = 300 of the reference's examples/translation.ini): the loops run at
# the next size they do take, on zero-padded copies of their operands.  A padded unit has zero weights and zero input
# projections: r = u = 1/2, c = tanh(0) = 0, h' = h / 2 -- it starts at zero and stays there, forward and backward, and
# contributes nothing to the real units (its column of every recurrent kernel is zero).  Copies in and out are ~15
# strided-copy launches and ~100 MB per loop (0.1 ms) against 50 x 6 us of launches saved.
PAD_LOOPS = __import__("os").environ.get("NM_CLUSTER_PAD", "1") != "0"


def padded_size(hsz: int) -> int:
    for hp in (256, 384, 512):
        if hsz <= hp:
            return hp
    return 0


def seq_mode(session, rows: int, hsz: int, ndir: int, wgh, wch) -> int:
    """0: no cluster loop for this shape; otherwise the hidden size the loops run at (= hsz, or the padded size)."""
    if cluster_ok(session, rows, hsz, ndir, wgh, wch):
        return hsz
    if not (PAD_LOOPS and session.use_cluster_loops and hsz % 4 == 0 and wgh.is_cuda):
        return 0
    hp = padded_size(hsz)
    if not hp or hp == hsz:
        return 0
    key = (rows, hp, ndir)
    if key not in _CLUSTER_OK:
        _CLUSTER_OK[key] = ops.gru_seq_supported(rows, hp, ndir)
    return hp if _CLUSTER_OK[key] else 0


def _blocks(src, dst, ndir, nblk, h, hp, back=False):
    """Copy the [.., ndir x nblk blocks of width h] columns of ``src`` into the blocks of width hp of ``dst`` (or back)."""
    s2, d2 = src.reshape(-1, ndir * nblk * h), dst.view(-1, ndir * nblk * hp)
    for i in range(ndir * nblk):
        a, b = s2[:, i * h:(i + 1) * h], d2[:, i * hp:i * hp + h]
        if back:
            ops.copy_cols(b, a)
        else:
            ops.copy_cols(a, b)


def _pad_weights(ctx, key, wgh, wch, h, hp):
    g3 = wgh if wgh.dim() == 3 else wgh.unsqueeze(0)
    c3 = wch if wch.dim() == 3 else wch.unsqueeze(0)
    ndir = g3.shape[0]
    wg_p = ctx.buffer((key, "pad_wgh"), (ndir, hp, 2 * hp), zero_init=True)
    wc_p = ctx.buffer((key, "pad_wch"), (ndir, hp, hp), zero_init=True)
    for d in range(ndir):
        ops.copy_cols(g3[d][:, :h], wg_p[d][:h, :h])
        ops.copy_cols(g3[d][:, h:], wg_p[d][:h, hp:hp + h])
        ops.copy_cols(c3[d], wc_p[d][:h, :h])
    return wg_p, wc_p


def seq_fwd(ctx, key, hp, steps, ndir, rows, h, xp, x_strides, h_in0, h_out0, h_step, ru0, ru_step, rh0, rh_step, c0,
            c_step, wgh, wch, lengths=None, reverse_dir0=False, out=None, out_strides=(0, 0, 0)):
    """ops.gru_seq_fwd at hidden size ``hp`` (``seq_mode``): directly when hp == h, else on padded copies whose results
    are copied back into the caller's buffers (the padded gates / candidates / states stay for ``seq_bwd``)."""
    sticky = ctx.session.error_word()
    if hp == h:
        ops.gru_seq_fwd(steps, ndir, rows, h, xp, x_strides, h_in0, h_out0, h_step, ru0, ru_step, rh0, rh_step, c0, c_step,
                        wgh, wch, cluster_workspace(ctx, key, rows, h, ndir), lengths=lengths, reverse_dir0=reverse_dir0,
                        out=out, out_strides=out_strides, sticky=sticky)
        return
    sc = lambda st: st // h * hp
    buf = lambda name, shape, **kw: ctx.buffer((key, "pad", name), shape, **kw)
    wg_p, wc_p = _pad_weights(ctx, key, wgh, wch, h, hp)
    n_x = xp.numel() // (ndir * 3 * h)
    xp_p = buf("xp", (n_x, ndir * 3 * hp), zero_init=True)
    _blocks(xp, xp_p, ndir, 3, h, hp)
    hin_p = buf("h_in", (ndir, rows, hp), zero_init=True)
    _blocks(h_in0, hin_p, 1, 1, h, hp)
    nst = steps if h_step else 1
    hout_p = buf("h_out", (nst, ndir, rows, hp), zero_init=True)
    nru = steps if ru_step else 1
    ru_p = buf("ru", (nru, ndir, rows, 2 * hp))
    nc = steps if (c0 is not None and c_step) else 1
    c_p = buf("c", (nc, ndir, rows, hp)) if c0 is not None else None
    nrh = steps if (rh0 is not None and rh_step) else 1
    rh_p = buf("rh", (nrh, ndir, rows, hp)) if rh0 is not None else None
    out_p = None
    if out is not None:
        out_p = buf("out", (out.numel() // (ndir * h), ndir * hp))
        ops.zero(out_p)
    ops.gru_seq_fwd(steps, ndir, rows, hp, xp_p, tuple(sc(v) for v in x_strides), hin_p, hout_p[0],
                    ndir * rows * hp if h_step else 0, ru_p[0], ndir * rows * 2 * hp if ru_step else 0,
                    None if rh_p is None else rh_p[0], ndir * rows * hp if (rh_p is not None and rh_step) else 0,
                    None if c_p is None else c_p[0], ndir * rows * hp if (c_p is not None and c_step) else 0, wg_p, wc_p,
                    cluster_workspace(ctx, key, rows, hp, ndir), lengths=lengths, reverse_dir0=reverse_dir0, out=out_p,
                    out_strides=tuple(sc(v) for v in out_strides), sticky=sticky)
    # back into the caller's buffers (what the rest of the step reads)
    span = lambda t0, n, w: t0.as_strided((n * ndir * rows, w), (w, 1))          # step t at t * ndir * rows rows
    _blocks(span(h_out0, nst, h), hout_p, 1, 1, h, hp, back=True)
    _blocks(span(ru0, nru, 2 * h), ru_p, 1, 2, h, hp, back=True)
    if c_p is not None:
        _blocks(span(c0, nc, h), c_p, 1, 1, h, hp, back=True)
    if rh_p is not None:
        _blocks(span(rh0, nrh, h), rh_p, 1, 1, h, hp, back=True)
    if out is not None:
        _blocks(out, out_p, ndir, 1, h, hp, back=True)
    ctx.memo[(key, "padded_loop")] = {"ru": ru_p, "c": c_p, "h_in": hin_p, "h_out": hout_p, "out": out_p, "wg": wg_p,
                                      "wc": wc_p, "out_ptr": None if out is None else out.data_ptr(),
                                      "h_out_ptr": h_out0.data_ptr(), "steps_saved": (nru, nc, nst)}


def seq_bwd(ctx, key, hp, steps, ndir, rows, h, dh, dout, dout_strides, ru0, ru_step, c0, c_step, h0, hseq, hseq_strides,
            dxp, dxp_strides, wgh, wch, lengths=None, reverse_dir0=False):
    """ops.gru_seq_bwd at hidden size ``hp``; padded: on the gates / candidates / states ``seq_fwd`` kept."""
    sticky = ctx.session.error_word()
    if hp == h:
        ops.gru_seq_bwd(steps, ndir, rows, h, dh, dout, dout_strides, ru0, ru_step, c0, c_step, h0, hseq, hseq_strides, dxp,
                        dxp_strides, wgh, wch, cluster_workspace(ctx, key, rows, h, ndir), lengths=lengths,
                        reverse_dir0=reverse_dir0, sticky=sticky)
        return
    kept = ctx.memo[(key, "padded_loop")]
    nru, nc, nst = kept["steps_saved"]
    assert nru == steps and nc == steps, "the forward loop did not keep its gates (inference pass?)"
    sc = lambda st: st // h * hp
    buf = lambda name, shape, **kw: ctx.buffer((key, "pad", name), shape, **kw)
    if hseq.data_ptr() == kept["out_ptr"]:
        hseq_p = kept["out"]
    else:
        assert hseq.data_ptr() == kept["h_out_ptr"] and nst == steps, "hseq must be a buffer the forward loop wrote"
        hseq_p = kept["h_out"]
    dh_p = buf("dh", (ndir, rows, hp), zero_init=True)
    _blocks(dh, dh_p, 1, 1, h, hp)
    dout_p = None
    if dout is not None:
        dout_p = buf("dout", (dout.numel() // (ndir * h), ndir * hp), zero_init=True)
        _blocks(dout, dout_p, ndir, 1, h, hp)
    n_x = dxp.numel() // (ndir * 3 * h)
    dxp_p = buf("dxp", (n_x, ndir * 3 * hp))
    ops.zero(dxp_p)
    ops.gru_seq_bwd(steps, ndir, rows, hp, dh_p, dout_p, None if dout is None else tuple(sc(v) for v in dout_strides),
                    kept["ru"][0], ndir * rows * 2 * hp, kept["c"][0], ndir * rows * hp, None if h0 is None else kept["h_in"],
                    hseq_p, tuple(sc(v) for v in hseq_strides), dxp_p, tuple(sc(v) for v in dxp_strides), kept["wg"],
                    kept["wc"], cluster_workspace(ctx, key, rows, hp, ndir), lengths=lengths, reverse_dir0=reverse_dir0,
                    sticky=sticky)
    _blocks(dxp, dxp_p, ndir, 3, h, hp, back=True)
    _blocks(dh, dh_p, 1, 1, h, hp, back=True)


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
