"""ctypes binding of libnmhip.so -- the C-ABI drop-in boundary (include/nmhip.h).

The product path fails loudly when the HIP extension is missing: there is no
CPU fallback anywhere in this package.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

# torch must be imported first: it ships its own libamdhip64.so, and libnmhip has to
# bind to that same HIP runtime instance (streams and device pointers come from torch).
import torch  # noqa: F401  pylint: disable=unused-import

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnmhip.so")

_lib = None

P = c_void_p
I = c_int
L = c_int64
F = c_float

# name -> (restype, argtypes); must list every symbol include/nmhip.h declares.
SIGNATURES = {
    "nm_last_error": (c_char_p, []),
    "nm_version": (I, []),
    "nm_crc32c": (ctypes.c_uint32, [ctypes.c_uint32, P, L]),
    "nm_gemm_f32": (I, [P, I, I, L, L, L, P, L, P, L, P, L, P, I, I, L, L, L, L, I, P, L]),
    "nm_gemm_f32_group": (I, [P, I, I, L, L, L, P, L, L, L, I, L]),
    "nm_gemm_f32_chain": (I, [P, L, L, L, L, P, L, L, P, L, I, P, L]),
    "nm_colsum_chain": (I, [P, P, ctypes.c_int32, ctypes.c_int32, L, L, L, L, P, I, P, L]),
    "nm_outer_chain": (I, [P, P, L, L, L, L, L, L, P, I]),
    "nm_embedding_gather": (I, [P, P, L, L, P, L, P, L, I, F]),
    "nm_gru_gates_fwd": (I, [P, P, L, L, L, P, P, P, P, P, I, I, I, L, L]),
    "nm_gru_blend_fwd": (I, [P, P, L, L, L, P, P, P, P, P, P, L, L, L, P, I, I, I, L, L]),
    "nm_layer_norm_fwd": (I, [P, P, L, P, P, P, L, P, P, L, L, F]),
    "nm_add_layer_norm_fwd": (I, [P, P, L, P, L, P, P, P, L, P, L, L, L, F]),
    "nm_add_layer_norm_stats_fwd": (I, [P, P, P, P, P, P, P, P, P, L, L, F]),
    "nm_copy_cols": (I, [P, P, L, P, L, L, L]),
    "nm_reduce_sum": (I, [P, P, L, P]),
    "nm_log_softmax": (I, [P, P, L, P, P, P, L, L, L]),
    "nm_attn_workspace_bytes": (L, [L, L, L]),
    "nm_attn_fwd": (I, [P, P, P, P, P, P, P, L, L, L, L, L, P, L, P, P, L, P]),
    "nm_attn_fwd_multi": (I, [P, P, P, P, P, P, P, L, L, L, L, L, L, L, P, L, P, P, L, P]),
    "nm_gru_rh_seq": (I, [P, P, P, P, P, I, L, L, I, L]),
    "nm_gru_gemm": (I, [P, P, I, L, P, L, L, P, L, L]),
    "nm_gru_seq_fwd": (I, [P, P, ctypes.c_int32, L, L, L, L, P, L, L, P, L, L, P, L, P]),
    "nm_gru_seq_bwd": (I, [P, P, ctypes.c_int32, L, L, P, L, L, P, L, L, P, L, P]),
    "nm_lstm_seq_workspace_bytes": (L, [L, L, ctypes.c_int32]),
    "nm_lstm_seq_fwd": (I, [P, P, ctypes.c_int32, L, L, L, P, L, L, F, P, L, P]),
    "nm_lstm_seq_bwd": (I, [P, P, ctypes.c_int32, L, L, P, L, L, P, L, P]),
    "nm_nematus_seq_workspace_bytes": (L, [L, L, ctypes.c_int32]),
    "nm_nematus_seq_fwd": (I, [P, P, ctypes.c_int32, L, L, L, L, P, L, L, P, L, L, P, P, P, L, P]),
    "nm_nematus_seq_bwd": (I, [P, P, ctypes.c_int32, L, L, L, P, L, L, P, L, L, P, L, P]),
    "nm_proj_split_bytes": (L, [L, L]),
    "nm_proj_split_prepare": (I, [P, P, L, I, L, L, P, L]),
    "nm_proj_split_forget": (I, [P]),
    "nm_gru_seq_supported": (I, [L, L, ctypes.c_int32]),
    "nm_gru_seq_workspace_bytes": (L, [L, L, ctypes.c_int32]),
    "nm_gru_seq_failed": (I, [P]),
    "nm_gru_seq_force_give_up": (I, [ctypes.c_int32]),
    "nm_gru_seq_test_hog": (I, [P, ctypes.c_int32, L, L]),
    "nm_test_xcc_ids": (I, [P, P, ctypes.c_int32, ctypes.c_int32]),
    "nm_create": (I, [I, P]),
    "nm_destroy": (I, [P]),
    "nm_ctx_bind": (I, [P]),
    "nm_ctx_current": (P, []),
    "nm_ctx_device": (I, [P]),
    "nm_ctx_switch": (I, [P, c_char_p, P]),
    "nm_ctx_set_background": (I, [P, I]),
    "nm_allreduce_unique_id": (I, [P, L]),
    "nm_allreduce_init": (I, [I, I, P, P]),
    "nm_allreduce_bucket": (I, [P, P, P, L]),
    "nm_allreduce_wait": (I, [P, P]),
    "nm_allreduce_destroy": (I, [P]),
    "nm_prof_enable": (I, [P, I]),
    "nm_prof_attn_step": (I, [P, P, P]),
    "nm_row_stats": (I, [P, P, L, L, L, P, P, P]),
    "nm_gumbel_argmax": (I, [P, P, L, L, L, ctypes.c_uint32, P]),
    "nm_greedy_update": (I, [P, P, P, P, P, L, I, P]),
    "nm_xent": (I, [P, P, L, L, L, P, P, P, P, I, F]),
    "nm_xent_colsum": (I, [P, P, L, L, L, P, P, P, P, F, P, L]),
    "nm_beam_workspace_bytes": (L, [L, L, L]),
    "nm_beam_topk_step": (I, [P, P, L, L, L, L, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, L, P]),
    "nm_beam_topk_step_fused": (I, [P, P, L, L, L, L, P, P, P, P, I, P, P, P, P, P, P, P, P, L, P, P, P]),
    "nm_tanh_bwd": (I, [P, P, P, L]),
    "nm_colsum_workspace_bytes": (L, [L]),
    "nm_colsum": (I, [P, P, L, L, L, P, I, P, L]),
    "nm_colsum_algo": (I, [P, P, L, L, L, P, I, P, L, I]),
    "nm_embedding_scatter_add": (I, [P, P, L, L, P, L, P, L, I]),
    "nm_layer_norm_bwd": (I, [P, P, P, P, P, P, P, P, L, L]),
    "nm_layer_norm_bwd_params_workspace_bytes": (L, [L]),
    "nm_layer_norm_bwd_params": (I, [P, P, P, P, P, P, P, L, L, P, P, I, P, L]),
    "nm_gru_step_bwd": (I, [P, I, P, P, L, L, L, P, P, P, P, L, L, L, P, L, L, L, P, P, P, P, I, I, I, L, L]),
    "nm_gru_seq_shift": (I, [P, P, P, P, I, L, L, I, L]),
    "nm_attn_softmax_bwd": (I, [P, P, P, P, P, L, L, L]),
    "nm_attn_energy_bwd": (I, [P, P, P, P, P, P, P, P, L, L, L, L, I]),
    "nm_attn_step_bwd": (I, [P, P, L, P, P, P, P, P, L, P, P, P, L, L, L, L, L]),
    "nm_attn_softmax_fwd": (I, [P, P, P, P, L, L, L, L]),
    "nm_ew": (I, [P, I, P, L, P, L, P, L, L, L, F, I]),
    "nm_blend_fwd": (I, [P, P, L, P, L, P, L, P, L, L, L]),
    "nm_blend_bwd": (I, [P, P, L, P, L, P, L, P, L, P, L, P, L, P, L, L, L]),
    "nm_gemm_bf16x3_nt": (I, [P, L, L, L, P, L, P, L, P, L, I, I]),
    "nm_lstm_cell_fwd": (I, [P, P, L, P, L, P, L, P, L, P, L, L, L, F]),
    "nm_lstm_cell_bwd": (I, [P, P, L, P, L, P, L, P, L, P, L, P, L, P, L, L, L, I, I]),
    "nm_nematus_cell_fwd": (I, [P, P, L, P, L, P, L, P, L, P, L, P, P, P, L, L, L]),
    "nm_nematus_state_step": (I, [P, P, L, P, L, P, P, L, P, L, P, P, P, L, L, L]),
    "nm_nematus_full_step": (I, [P, P, L, P, L, P, P, L, P, L, P, P, L, P, P, P, L, L, L, L]),
    "nm_nematus_cell_bwd": (I, [P, P, L, P, P, P, L, P, L, P, L, P, L, P, L, P, L, P, L, L, L, I, I, I, I]),
    "nm_dropout": (I, [P, P, L, P, L, L, L, F, ctypes.c_uint32, P, I]),
    "nm_rnn_select_fwd": (I, [P, P, L, P, L, P, I, P, L, P, L, L, L]),
    "nm_rnn_select_bwd": (I, [P, P, L, P, L, P, I, P, L, P, L, L, L]),
    "nm_reverse_sequence": (I, [P, P, P, P, L, L, L, I]),
    "nm_maxout_fwd": (I, [P, P, L, P, L, P, L, L, L]),
    "nm_maxout_bwd": (I, [P, P, L, P, P, L, L, L, L]),
    "nm_sdp_attn_fwd": (I, [P, P, L, P, L, P, L, P, L, L, L, L, L, L, L, I, F, ctypes.c_uint32, P, P, L, P]),
    "nm_sdp_attn_step": (I, [P, P, L, P, L, P, L, P, L, L, L, L, L, P, L, P, L, P]),
    "nm_sdp_attn_bwd": (I, [P, P, L, P, L, P, L, P, L, P, P, L, L, L, L, L, L, I, F, ctypes.c_uint32, P,
                            P, L, P, L, P, L, P, I]),
    "nm_add_position": (I, [P, P, P, P, L, L, L, L]),
    "nm_unfinished_mask": (I, [P, P, P, L, L]),
    "nm_time_sum": (I, [P, P, P, L, L, L]),
    "nm_time_bcast_add": (I, [P, P, P, L, L, L]),
    "nm_optim_workspace_bytes": (L, [L, L]),
    "nm_optim_regularize_norms": (I, [P, P, P, P, P, P, P, P, P, L, L, F, F, P, P, L]),
    "nm_optim_clip_adam": (I, [P, P, P, P, P, P, P, P, P, P, P, L, L, F, F, F, F, F, P, L]),
    "nm_optim_clip_adadelta": (I, [P, P, P, P, P, P, P, P, P, P, P, L, L, F, F, F, F, P, L]),
    "nm_optim_partials": (I, [P, P, P, P, P, P, P, P, P, L, L, F, F, L, L, P, L]),
    "nm_optim_partials_list": (I, [P, P, P, P, P, P, P, P, P, L, L, F, F, P, L, P, L]),
    "nm_optim_apply_list": (I, [P, I, P, P, P, P, P, P, P, P, P, P, L, L, F, F, F, F, F, P, L, P, P, L]),
    "nm_optim_segments": (I, [P, P, P, P, P, P, P, L, L, P, P, L]),
    "nm_optim_apply": (I, [P, I, P, P, P, P, P, P, P, P, P, P, L, L, F, F, F, F, F, L, L, P, P, L]),
    "nm_zero_if": (I, [P, P, P, L]),
    "nm_fill_u32": (I, [P, P, L, ctypes.c_uint32]),
    "nm_copy_d2d": (I, [P, P, P, L]),
    "nm_gather_rows_f32": (I, [P, P, L, P, P, L, L, L]),
    "nm_beam_reorder_tokens": (I, [P, P, P, P, P, L, L]),
    "nm_beam_backtrace": (I, [P, P, P, P, P, L, L]),
    "nm_logits_stats_tile": (L, [L]),
    "nm_logits_stats_bytes": (L, [L, L]),
    "nm_logits_stats_gemm": (I, [P, I, L, L, L, P, L, P, L, P, P, L, P, L]),
    "nm_greedy_finish": (I, [P, P, L, L, P, P, P, I, P, P, L, L, P, L, P, P, P]),
    "nm_attn_partials_layout": (I, [L, L, L, L, P, P, P]),
    "nm_attn_fwd_partials": (I, [P, P, P, P, P, P, P, L, L, L, L, L, P, L]),
    "nm_step_group": (I, [P, L, P, ctypes.c_int32]),
    "nm_decoder_step_fused": (I, [P, P]),
    "nm_dec_step_cluster_supported": (I, [L, L, L, L]),
    "nm_dec_step_cluster_workspace_bytes": (L, [L, L]),
    "nm_prof_stream_read": (I, [P, P, L, P]),
    "nm_beam_topk_step_tiles": (I, [P, P, L, P, L, L, L, L, P, P, P, P, I, P, P, P, P, P, P, P, P, L, P, P, P]),
}


class NMHipError(RuntimeError):
    pass


class GruEpilogue(ctypes.Structure):
    """``nm_gru_epilogue`` of include/nmhip.h."""
    _fields_ = [("mode", ctypes.c_int32), ("t", ctypes.c_int32), ("rev_mask", ctypes.c_int32),
                ("ndir", ctypes.c_int32), ("R", L), ("H", L), ("lengths", P),
                ("xp", P), ("x_dir", L), ("x_row", L), ("x_time", L),
                ("h_in", P), ("h_out", P), ("ru", P), ("rh", P), ("c_save", P),
                ("out", P), ("o_dir", L), ("o_row", L), ("o_time", L),
                ("dh", P), ("dout", P), ("do_dir", L), ("do_row", L), ("do_time", L),
                ("c", P), ("h0", P), ("hseq", P), ("hs_dir", L), ("hs_row", L), ("hs_time", L),
                ("dxp", P), ("dx_dir", L), ("dx_row", L), ("dx_time", L),
                ("dgpre", P), ("dcpre", P)]


class StepProblem(ctypes.Structure):
    """``nm_step_problem`` of include/nmhip.h."""
    _fields_ = [("A", P), ("lda", L), ("Bt", P), ("ldb", L), ("N", L), ("K", L),
                ("a_kind", ctypes.c_int32), ("epilogue", ctypes.c_int32), ("act", ctypes.c_int32),
                ("nchunk", ctypes.c_int32),
                ("bias", P), ("add", P), ("ldadd", L), ("C", P), ("ldc", L),
                ("pctx", P), ("pstat", P), ("energies", P), ("mask", P), ("weights", P),
                ("S", L), ("mask_div", L), ("mask_mod", L),
                ("h", P), ("ldh", L), ("ru", P), ("rh", P), ("xc", P), ("ldxc", L),
                ("h_out", P), ("ldho", L), ("h_out2", P), ("ldho2", L), ("add_ids", P), ("xc_ids", P)]


class DecoderStep(ctypes.Structure):
    """``nm_decoder_step`` of include/nmhip.h."""
    _fields_ = ([(n, L) for n in ("rows", "emb", "rnn", "attn_state", "ctx_width", "out", "vocab", "src_len",
                                  "rows_per_key")] +
                [("cat", P), ("h_copy", P), ("ld_h_copy", L), ("out_state", P), ("ld_out_state", L),
                 ("attn_weights", P), ("logits", P), ("ld_logits", L), ("stats", P), ("stats_bytes", L)] +
                [(n, P) for n in ("ru", "rh", "xc", "y", "pre_e", "pre", "ctx")] +
                [("attn_workspace", P), ("attn_workspace_bytes", L)] +
                [(n, P) for n in ("wg_t", "bg", "wcx_t", "wch_t", "bc", "wq_t", "bq", "keys", "values", "mask", "v",
                                  "attn_bias", "wo_h_t", "wo_e_t", "wo_c_t", "bo", "w_vocab")] +
                [("ld_w_vocab", L), ("b_vocab", P), ("out_act", ctypes.c_int32), ("vocab_trans_b", ctypes.c_int32)] +
                [(n, L) for n in ("ld_cat", "ld_ctx", "ld_wg", "ld_wcx", "ld_wch", "ld_wq", "ld_wo_h", "ld_wo_e",
                                  "ld_wo_c")] +
                [("in_table", P), ("ld_table", L), ("in_ids", P)] +
                [("cluster_ws", P), ("cluster_ws_bytes", L), ("sticky_error", P)])


def load():
    """Load libnmhip.so; raise (never fall back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NMHipError(
            f"{LIB_PATH} is missing: build it with `python -m neuralmonkey_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().nm_last_error()
        raise NMHipError(f"{what} failed ({code}): {msg.decode() if msg else '?'}")
