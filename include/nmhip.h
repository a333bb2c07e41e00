/* libnmhip -- C ABI of the MI355X (gfx950) attention-decoder hot path.
 *
 * The reference (ufal/neuralmonkey) has no FFI: its arithmetic is issued as
 * TensorFlow-1.12 ops from Python.  This header is the boundary a maintainer
 * binds instead (ctypes stub in INTEGRATION.md); each entry point cites the
 * reference call site (file:line under /root/reference) whose TF op(s) it
 * replaces.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; nm_last_error() gives
 *     the thread-local message;
 *   - all tensor pointers are DEVICE pointers owned by the caller (row-major
 *     fp32 / int32); the library never allocates or frees caller memory;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it;
 *   - sizes and strides are int64_t element counts; "ld*" = leading dimension.
 */
#ifndef NMHIP_H
#define NMHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* nm_last_error(void);
int nm_version(void);

/* ---- library contexts (nm_create / nm_destroy; the reference's counterpart is the tf.Session a
 * TensorFlowManager owns, tf_manager.py:78-79) -----------------------------------------------------------------
 * A context owns everything the library keeps between calls: the NM_* A/B and tuning switches (read from the
 * environment once, in nm_create), the HIP-event pool of the live attention-step timer, the device it was made
 * for.  nm_ctx_bind makes a context current for the CALLING THREAD; entry points called from that thread use
 * it.  A thread that never bound one uses the process default context (created on first use, never destroyed).
 * Handles are opaque; device = -1 takes the current HIP device. */
typedef void nm_ctx;
int nm_create(int device, nm_ctx** out_ctx);
int nm_destroy(nm_ctx* ctx);
int nm_ctx_bind(nm_ctx* ctx /* NULL: back to the default context */);
nm_ctx* nm_ctx_current(void);
int nm_ctx_device(nm_ctx* ctx /* NULL: the calling thread's */);
/* value of a switch as the context read it at creation: NM_ATTN_WHOLE -> "attn_whole" (-1 = unset) ... */
int nm_ctx_switch(nm_ctx* ctx, const char* name, int* value);
/* background mode: the launches that follow run beside a latency-bound loop of another stream (the reference
 * evaluates one batch at a time, tf_manager.py:226-262; here the encoder of batch n+1 runs under the decoding loop of
 * batch n).  Automatic GEMMs and recurrent step kernels cap their residency per CU (algo 4 of nm_gemm_f32) and
 * leave the wave priority to the foreground loop. */
int nm_ctx_set_background(nm_ctx* ctx /* NULL: the calling thread's */, int on);
/* host utility: CRC-32C (Castagnoli) of a HOST buffer, chained through `crc` (start with 0) -- the
 * checksum of TensorFlow tensor-bundle checkpoints (tf_manager.py:274-288 -> tf.train.Saver) */
uint32_t nm_crc32c(uint32_t crc, const void* data, int64_t n);

/* ---- dense projections: tf.matmul / tf.layers.dense / 1x1 tf.nn.conv2d -------------
 * attention/feed_forward.py:111-118 (keys), :130-132 (query);
 * decoders/output_projection.py:115-130; decoders/autoregressive.py:450-459 (logits);
 * decoders/encoder_projection.py:47-73; nn/ortho_gru_cell.py:44-53 (GRUCell kernels);
 * and their tf.gradients transposes.
 * C[M,N] = act(op(A).op(B) + bias (+ C)); transA: A stored [K,M]; transB: B stored [N,K];
 * act 0 none / 1 tanh / 2 relu; batch > 1 strides the three operands;
 * algo 0 auto / 1 tiled-128 / 2 tiled-64 / 3 skinny / 4 background: 128x128 tiles, at most NM_GEMM_BG_WGS (1)
 * workgroups resident per CU, for a long leaf GEMM that runs beside another stream's latency-bound launches.
 * fp32 MFMA (exact f32).
 * workspace (optional, device): split-K slabs for deep-K / few-tile shapes (weight gradients);
 * slabs are summed in a fixed order, results stay deterministic. */
int nm_gemm_f32(void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                const float* bias, int act, int accumulate, int64_t batch, int64_t strideA,
                int64_t strideB, int64_t strideC, int algo, void* workspace, int64_t workspace_bytes);
/* `count` independent products of ONE shape in one launch: C_i (+)= op(A_i) . op(B_i); `pointer_table` is a DEVICE
 * array of 3 * count pointers {A_0, B_0, C_0, A_1, ...} (16-byte aligned operands).  The weight gradients of a
 * backward pass (tf.gradients of tf.layers.dense kernels, one tf.matmul each in the reference's graph): their K -- the
 * rows of the batch -- is deep and their output tiles few, so alone each has to split K and reduce slabs; together
 * they fill the chip with whole-K workgroups.  No two products of a launch may share their C. */
int nm_gemm_f32_group(void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K,
                      const void* pointer_table, int64_t lda, int64_t ldb, int64_t ldc, int accumulate,
                      int64_t count);
/* (what tf.gradients -- trainers/generic_trainer.py:84-195 -- adds up for a tf.layers.dense kernel used in every step of a
 * tf.while_loop / unrolled decoder: decoders/decoder.py:226-325, nn/ortho_gru_cell.py:73-105)
 * C (+)= sum_i A_i^T . B_i over `count` members of `rows` rows each (A_i [rows, M], B_i [rows, N]): ONE product whose
 * K dimension is the chain of the members; pointer_table: device array [count][3] of {A_i, B_i, unused}.  The weight
 * gradients of a taped time loop (x_t^T . dy_t of every step) as one launch per kernel instead of one per step.
 * rows % 16 == 0; M, N, lda, ldb multiples of 4, members 16-byte aligned; K is split over workgroups like nm_gemm_f32
 * (workspace: split-K slabs, may be null). */
int nm_gemm_f32_chain(void* stream, int64_t M, int64_t N, int64_t rows, int64_t count, const void* pointer_table,
                      int64_t lda, int64_t ldb, float* C, int64_t ldc, int accumulate, void* workspace,
                      int64_t workspace_bytes);
/* out[c] (+)= sum_i sum_r x_i[r, c]: the column sums (bias gradients) of the same kind of chain; member i is
 * table[i * table_stride + table_offset].  workspace: nm_colsum_workspace_bytes(cols), zero-initialised once. */
int nm_colsum_chain(void* stream, const void* pointer_table, int32_t table_stride, int32_t table_offset, int64_t count,
                    int64_t rows, int64_t ldx, int64_t cols, float* out, int accumulate, void* workspace,
                    int64_t workspace_bytes);
/* (the gradient of attention/feed_forward.py:158-166, context = reduce_sum(weights * attention_states), w.r.t. the states)
 * out[b, s, c] (+)= sum_t w_t[b, s] * d_t[b, c] over `count` (<= 64) steps named by a device table [count][2] of
 * {w_t, d_t}: the gradient of the attended states [B, S, C] through the context sums of a taped time loop
 * (ctx_t = sum_s w_t[., s] states[., s, :]), one launch instead of one batched rank-1 product per step. */
int nm_outer_chain(void* stream, const void* pointer_table, int64_t count, int64_t B, int64_t S, int64_t C, int64_t ldw,
                   int64_t ldd, float* out, int accumulate);

/* ---- embedding lookup: model/sequence.py:170-194, decoders/autoregressive.py:269-272 --
 * out[i,:] = table[ids[i],:] * scale * (mask_pad ? ids[i] != 0 : 1) */
int nm_embedding_gather(void* stream, const float* table, int64_t V, int64_t E, const int32_t* ids,
                        int64_t n, float* out, int64_t ldo, int mask_pad, float scale);
/* gradient of the lookup (tf.gather grad); skip_pad drops id 0 rows (the mask multiply) */
int nm_embedding_scatter_add(void* stream, float* dtable, int64_t V, int64_t E, const int32_t* ids,
                             int64_t n, const float* d, int64_t ldd, int skip_pad);

/* ---- GRU cell epilogues: TF GRUCell via OrthoGRUCell, nn/ortho_gru_cell.py:44-53;
 * length masking / reverse_sequence of (bidirectional_)dynamic_rnn, encoders/recurrent.py:86-102.
 * xp = x.[Wg_x|Wc_x]+[bg|bc] addressed xp + d*x_dir_off + r*x_row_stride + pos*x_time_stride;
 * rev_mask bit d: direction d walks its sequence backwards; rows with t >= lengths[r] are dead. */
int nm_gru_gates_fwd(void* stream, const float* xp, int64_t x_dir_off, int64_t x_row_stride,
                     int64_t x_time_stride, const float* hg, const float* h, float* ru, float* rh,
                     const int32_t* lengths, int t, int rev_mask, int ndir, int64_t R, int64_t H);
int nm_gru_blend_fwd(void* stream, const float* xp, int64_t x_dir_off, int64_t x_row_stride,
                     int64_t x_time_stride, const float* hc, const float* ru, const float* h_in,
                     float* h_out, float* c_save, float* out, int64_t out_dir_off,
                     int64_t out_row_stride, int64_t out_time_stride, const int32_t* lengths, int t,
                     int rev_mask, int ndir, int64_t R, int64_t H);
/* one BPTT step of the cell: phase 0 = blend backward, phase 1 = gates backward */
int nm_gru_step_bwd(void* stream, int phase, float* dh, const float* dout, int64_t do_dir,
                    int64_t do_row, int64_t do_time, const float* ru, const float* c, const float* h0,
                    const float* hseq, int64_t hs_dir, int64_t hs_row, int64_t hs_time, float* dxp,
                    int64_t dx_dir, int64_t dx_row, int64_t dx_time, float* dgpre, float* dcpre,
                    const float* drh, const int32_t* lengths, int t, int rev_mask, int ndir, int64_t R,
                    int64_t H);
/* One recurrent GEMM of a GRU step with its epilogue fused into the GEMM kernel, so a step is
 * two launches (gates, candidate) instead of four.  mode 1: C = h.Wg_h -> r,u,r*h;  2: C =
 * (r*h).Wc_h -> c, h';  3: C = dc_pre.Wc_h^T -> dr_pre, dh += ..;  4: dh += dg_pre.Wg_h^T, then the
 * blend backward of step `t` on the completed dh.  M = R, N = 2H (mode 1) or H, batch = ndir. */
typedef struct nm_gru_epilogue {
    int32_t mode, t, rev_mask, ndir;
    int64_t R, H;
    const int32_t* lengths;
    const float* xp; int64_t x_dir, x_row, x_time;
    const float* h_in; float* h_out; float* ru; float* rh; float* c_save;
    float* out; int64_t o_dir, o_row, o_time;
    float* dh; const float* dout; int64_t do_dir, do_row, do_time;
    const float* c; const float* h0; const float* hseq; int64_t hs_dir, hs_row, hs_time;
    float* dxp; int64_t dx_dir, dx_row, dx_time;
    float* dgpre; float* dcpre;
} nm_gru_epilogue;
int nm_gru_gemm(void* stream, const nm_gru_epilogue* epi, int transB, int64_t K, const float* A,
                int64_t lda, int64_t strideA, const float* B, int64_t ldb, int64_t strideB);
/* The time loops of a GRU layer (both directions) as ONE launch each (csrc/nm_gru_cluster.hip): the chip is cut into
 * clusters of workgroups = (direction, block of 16 or 32 rows); a workgroup keeps its 16 hidden units' slices of the
 * recurrent kernels in registers for all steps and hands its stage outputs to the rest of its cluster as tagged 8-byte
 * granules -- no launch, no barrier and no cache invalidate between steps.  Products, accumulation order and
 * epilogues are those of nm_gru_gemm (forward: modes 1, 2; backward: modes 4, 3).  `e` holds the step-0 pointers.
 *   supported        1 when (R, H, ndir) can run this way on the current device (H % 128 == 0, 256 <= H <= 512, the
 *                    clusters -- ndir * ceil(R / 16), or ndir * ceil(R / 32) at H = 512 -- fit the XCDs with their
 *                    H/16 workgroups each: ceil(clusters / 8) * H / 16 <= CUs / 8), else the caller steps with
 *                    nm_gru_gemm;
 *   workspace_bytes  device memory a call needs (header + granules; zeroed by the call, 16-byte aligned); a workspace
 *                    belongs to ONE loop in flight (the next call on the same stream may reuse it);
 *   failed           after a synchronisation: 1 when a loop that used `workspace` gave up waiting for a hand-off
 *                    (0.2 s without progress; its results are garbage), 0 otherwise; a launch that gave up also
 *                    sets the caller's device word `sticky_error` (null: none) to 1 and never clears it;
 *   fwd              step t writes h_out + t*h_step, ru + t*ru_step, rh + t*rh_step (rh may be null), c_save +
 *                    t*c_step and `out` at the step's position; h_in is read once;
 *   bwd              dh holds dL/dh after the last step on entry and dL/dh_0 on exit; step t (last first) reads ru +
 *                    t*ru_step, c + t*c_step, h_prev through hseq / h0 and dout at the step's position, and writes the
 *                    three pre-activation gradients of that position into dxp (dead positions are left alone). */
int nm_gru_seq_supported(int64_t R, int64_t H, int32_t ndir);
int64_t nm_gru_seq_workspace_bytes(int64_t R, int64_t H, int32_t ndir);
int nm_gru_seq_failed(const void* workspace);
/* Test hook: the next `launches` cluster loops of this process raise their error word at once (what a launch does
 * after 0.2 s without progress when something else holds compute units): the recovery paths of the host side
 * (runtime.Session.recover_training, TensorFlowManager.execute) are exercised on a healthy device.  Returns the
 * number of forced launches that were still pending. */
int nm_gru_seq_force_give_up(int32_t launches);
/* Test utility: `blocks` workgroups that each hold `lds_bytes` of LDS on a CU for `microseconds` (sleeping): what a
 * long-running kernel of another stream or process does to a cluster loop launched meanwhile. */
int nm_gru_seq_test_hog(void* stream, int32_t blocks, int64_t lds_bytes, int64_t microseconds);
/* Test utility: the XCD (0..7) every workgroup of a `blocks` x `threads` launch landed on, into xcc_out[block].  The
 * wide attention step and the GEMM tile order rest their speed (not their results) on round-robin dispatch. */
int nm_test_xcc_ids(void* stream, int32_t* xcc_out, int32_t blocks, int32_t threads);
int nm_gru_seq_fwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t h_step, int64_t ru_step,
                   int64_t rh_step, int64_t c_step, const float* wgh, int64_t ld_g, int64_t stride_g,
                   const float* wch, int64_t ld_c, int64_t stride_c, void* workspace,
                   int64_t workspace_bytes, uint32_t* sticky_error);
int nm_gru_seq_bwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t ru_step, int64_t c_step,
                   const float* wgh, int64_t ld_g, int64_t stride_g, const float* wch, int64_t ld_c,
                   int64_t stride_c, void* workspace, int64_t workspace_bytes, uint32_t* sticky_error);
/* The same for NematusGRUCell (nn/ortho_gru_cell.py:73-105: the reset gate multiplies the state projection AFTER the
 * product, c = tanh(x_c + r * (h.U_c + b_cs))): both recurrent products read h only, so a step is ONE product, one
 * element-wise stage and one hand-off.  Shapes, workspace ownership, give-up behaviour and `sticky_error` as
 * nm_gru_seq_*; ug [ndir][H][2H] and uc [ndir][H][H] are the STATE projections (input projections and their biases are
 * in xp, 3H wide per direction: r | u | c), bgs / bcs their optional biases ([ndir][2H] / [ndir][H]).
 *   fwd   step t also writes sc = h.U_c + b_cs to e->rh + t*sc_step (required: the backward loop reads it); h_in and
 *         h_out must be DIFFERENT buffers (one stage per step: a workgroup may write step 0's state while another
 *         still reads the initial one);
 *   bwd   dxp is 4H wide per direction: [dr' | du' | dc' | dsc] of the step's position -- columns [0, 3H) are the
 *         gradient of xp, columns [0, 2H) and [3H, 4H) those of the state projections' outputs. */
int64_t nm_nematus_seq_workspace_bytes(int64_t R, int64_t H, int32_t ndir);
int nm_nematus_seq_fwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t h_step, int64_t ru_step,
                       int64_t sc_step, int64_t c_step, const float* ug, int64_t ld_g, int64_t stride_g,
                       const float* uc, int64_t ld_c, int64_t stride_c, const float* bgs, const float* bcs,
                       void* workspace, int64_t workspace_bytes, uint32_t* sticky_error);
int nm_nematus_seq_bwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t ru_step, int64_t sc_step,
                       int64_t c_step, const float* ug, int64_t ld_g, int64_t stride_g, const float* uc,
                       int64_t ld_c, int64_t stride_c, void* workspace, int64_t workspace_bytes,
                       uint32_t* sticky_error);
/* ... and for LSTMCell (tf.nn.rnn_cell.LSTMCell as the reference builds it: gate order i, j, f, o, forget_bias added to f,
 * state (c, h), zero initial cell state): one product per step over wh [ndir][H][4H], the state half of the cell's
 * kernel; xp and dxp are 4H wide per direction; e->ru holds the ACTIVATED gates [i | j | f | o] of every step (g_step
 * apart), e->c_save / e->c the cell state after every step.  Shapes, workspace ownership and give-up behaviour as
 * nm_gru_seq_*; h_in and h_out must be different buffers. */
int64_t nm_lstm_seq_workspace_bytes(int64_t R, int64_t H, int32_t ndir);
int nm_lstm_seq_fwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t h_step, int64_t g_step, int64_t c_step,
                    const float* wh, int64_t ld_w, int64_t stride_w, float forget_bias, void* workspace,
                    int64_t workspace_bytes, uint32_t* sticky_error);
int nm_lstm_seq_bwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t g_step, int64_t c_step, const float* wh,
                    int64_t ld_w, int64_t stride_w, void* workspace, int64_t workspace_bytes, uint32_t* sticky_error);
int nm_gru_seq_shift(void* stream, const float* seq, float* out, const int32_t* lengths, int rev_mask,
                     int64_t B, int64_t S, int ndir, int64_t H);
int nm_gru_rh_seq(void* stream, const float* ru_all, const float* hprev, float* out,
                  const int32_t* lengths, int rev_mask, int64_t B, int64_t S, int ndir, int64_t H);

/* ---- layer norm: tf_utils.py:189-219 (eps inside rsqrt, biased variance) ------------------ */
/* sum_out = a + x ; y = layer_norm(sum_out): the residual connection that ends a Transformer sub-layer and the
 * pre-norm that starts the next (decoders/transformer.py:270-358, tf_utils.py:189-219) in one pass. */
int nm_add_layer_norm_fwd(void* stream, const float* a, int64_t lda, const float* x, int64_t ldx,
                          const float* gamma, const float* beta, float* sum_out, int64_t lds, float* y, int64_t ldy,
                          int64_t rows, int64_t D, float eps);
/* ... with the row statistics the backward pass reads (contiguous rows, D a multiple of 4 up to 2048, 16-byte aligned). */
int nm_add_layer_norm_stats_fwd(void* stream, const float* a, const float* x, const float* gamma, const float* beta,
                                float* sum_out, float* y, float* mean_out, float* rstd_out, int64_t rows, int64_t D,
                                float eps);
int nm_layer_norm_fwd(void* stream, const float* x, int64_t ldx, const float* gamma, const float* beta,
                      float* y, int64_t ldy, float* mean_out, float* rstd_out, int64_t rows, int64_t D,
                      float eps);
int nm_layer_norm_bwd(void* stream, const float* dy, const float* x, const float* mean,
                      const float* rstd, const float* gamma, float* dx, float* dyx, int64_t rows,
                      int64_t D);
/* The same with the parameter gradients in the call: dx as above, dgamma = sum over rows of dy * xhat and dbeta = sum
 * over rows of dy (tf_utils.py:189-219 differentiated), added to what is there when bit 0 of `accumulate` is set; bit 1:
 * dx is ADDED to what `dx` holds (the gradient a residual connection already left there).  Two launches (row
 * pass with per-workgroup partial sums, a fixed-order reduction) instead of nm_layer_norm_bwd + two nm_colsum and a
 * [rows, D] buffer of dy * xhat.  workspace: nm_layer_norm_bwd_params_workspace_bytes(D). */
int64_t nm_layer_norm_bwd_params_workspace_bytes(int64_t D);
int nm_layer_norm_bwd_params(void* stream, const float* dy, const float* x, const float* mean, const float* rstd,
                             const float* gamma, float* dx, int64_t rows, int64_t D, float* dgamma, float* dbeta,
                             int accumulate, void* workspace, int64_t workspace_bytes);

/* ---- Bahdanau attention step: Attention.attention, attention/feed_forward.py:120-166 ------
 * energies + softmax + mask renormalisation (+1e-8) + context, fused; keys of row r are
 * those of sentence r / rows_per_key (beam search without tiling the keys).
 * Dispatch by shape.  One query per sentence, >= 96 sentences, 40..52 positions (the headline decoding step): one
 * 1024-thread workgroup per sentence does the whole step, nothing is merged afterwards.  Otherwise split-S: every
 * sentence is scored by several chunk workgroups; with one query per sentence and <= 8 chunks the workgroup that
 * arrives LAST merges the partials itself (write-through hand-off + one arrival counter per sentence), else a combine
 * kernel follows (several queries per sentence: beam search).  The workspace therefore ends with arrival counters
 * that must be ZERO at launch: zero a workspace once after allocating it -- the kernels leave the counters at zero. */
int64_t nm_attn_workspace_bytes(int64_t R, int64_t S, int64_t C);
int nm_attn_fwd(void* stream, const float* y, const float* hf, const float* states, const float* mask,
                const float* v, const float* bias, int64_t R, int64_t rows_per_key, int64_t S,
                int64_t A, int64_t C, float* ctx, int64_t ldctx, float* weights, void* workspace,
                int64_t workspace_bytes, float* energies_out);
/* nq queries per key batch in one launch; query q of key batch b is row b*q_stride_b + q*q_stride_q
 * of y / ctx / weights: [Bk,nq]-major (beam hypotheses) or [nq,Bk]-major (all T teacher-forced
 * steps of training at once -- the keys are then read from HBM once for the whole target sentence). */
int nm_attn_fwd_multi(void* stream, const float* y, const float* hf, const float* states,
                      const float* mask, const float* v, const float* bias, int64_t Bk, int64_t nq,
                      int64_t q_stride_b, int64_t q_stride_q, int64_t S, int64_t A, int64_t C, float* ctx,
                      int64_t ldctx, float* weights, void* workspace, int64_t workspace_bytes,
                      float* energies_out);
/* backward of T steps at once: softmax/renorm part, then the tanh energies part */
int nm_attn_softmax_bwd(void* stream, const float* dw, const float* e, const float* mask, float* de,
                        int64_t rows, int64_t B, int64_t S);
int nm_attn_energy_bwd(void* stream, const float* de, const float* hf, const float* y, const float* v,
                       float* dhf, float* dv_partial, float* dy, int64_t T, int64_t B, int64_t S,
                       int64_t A, int accumulate /* dhf, dv_partial += (per-step backward) */);
/* ONE step's backward up to the query in one launch (a taped decoder step, decoders/decoder.py:303-325 with the cells
 * of any configuration around the attention): dw = <dctx, states> (feed_forward.py:146-149), the softmax/renorm
 * backward (:139-144) -> de [B,S] (written: the key-side sums over all steps are taken later from the stacked de),
 * dy [B,A] = v * sum_s de (1 - tanh^2(hf + y)) (:120-123).  One query per sentence; C and lddctx multiples of 4. */
int nm_attn_step_bwd(void* stream, const float* dctx, int64_t lddctx, const float* states, const float* e,
                     const float* mask /* [B,S] or NULL */, const float* hf, const float* y, int64_t ldy,
                     const float* v, float* de, float* dy, int64_t lddy, int64_t B, int64_t S, int64_t C, int64_t A);
/* (dhf and dv_partial both null: the query gradients dy alone -- what a step of a taped loop needs at once; the key-side
 * sums are then one call over all steps when the backward pass has been through them) */
/* the distribution alone, from energies assembled by the caller (several encoders + a sentinel):
 * attention/combination.py:301-307 (FlatMultiAttention._renorm_softmax), :421 (hierarchical softmax,
 * mask == NULL).  Mask row of query row r: (r / rows_per_key) % B. */
int nm_attn_softmax_fwd(void* stream, const float* e, const float* mask, float* w, int64_t rows, int64_t B,
                        int64_t S, int64_t rows_per_key);
/* live HIP-event timing of one attention step = everything nm_attn_fwd launches (split-S partial kernel +
 * combine), events recorded on the launch stream (bench.py roofline); the recorder belongs to the context: launches
 * made by threads bound to other contexts are not seen */
int nm_prof_enable(nm_ctx* ctx /* NULL: the calling thread's context */, int on);
int nm_prof_attn_step(nm_ctx* ctx, double* total_ms, int64_t* count);
/* yardstick for that timing: a plain streaming read (float4 loads, one partial sum per workgroup into
 * sink[0..2047]) of `bytes` bytes, timed by the same event pool when profiling is enabled */
int nm_prof_stream_read(void* stream, const void* src, int64_t bytes, float* sink);

/* The attention step WITHOUT its combine launch: energies [R,S] (workspace offset 0) and the split-S partials
 * stay in the workspace for a consumer that merges them while it loads them (nm_step_group, a_kind 1).
 * nm_attn_partials_layout: float offsets of the partial contexts [R,nchunk,C] and statistics [R,nchunk,4] =
 * {max, sum exp, sum exp*mask, -} inside that workspace; <0 when the shape takes the any-shape kernel. */
int nm_attn_partials_layout(int64_t R, int64_t S, int64_t A, int64_t C, int64_t* nchunk, int64_t* pctx_off,
                            int64_t* pstat_off);
int nm_attn_fwd_partials(void* stream, const float* y, const float* hf, const float* states, const float* mask,
                         const float* v, const float* bias, int64_t R, int64_t rows_per_key, int64_t S, int64_t A,
                         int64_t C, void* workspace, int64_t workspace_bytes);

/* ---- one inference step of the RNN attention decoder as groups of skinny GEMMs: Decoder.next_state,
 * decoders/decoder.py:279-358 (GRUCell nn/ortho_gru_cell.py:44-53, query projection
 * attention/feed_forward.py:130-132, output projection decoders/output_projection.py:115-130).
 * Every problem of a group is C[M,N] = epilogue(A[M,K] . Bt[N,K]^T) with M common to the group; problems of
 * one group do not depend on each other and share a launch.  Weights are passed transposed ([N,K]).
 *   a_kind   0: A as stored.  1: A[r,:] = sum_i f_i pctx[r,i,:] / den -- the merge of the split-S attention
 *               partials (softmax -> mask -> renormalise +1e-8, feed_forward.py:139-154) done in the operand
 *               loader (K = C); with `weights` != NULL the normalised distribution [M,S] is written too
 *               (mask row of query row r: (r / mask_div) % mask_mod).
 *   epilogue 0: C = act(s + bias + add), act 0 none / 1 tanh.
 *            1: GRU gates, N = 2H: ru = sigmoid(s + bias), rh = r * h.
 *            2: GRU candidate + blend, N = H: c = tanh(xc + s), h' = u*h + (1-u)*c -> h_out (and h_out2). */
typedef struct nm_step_problem {
    const float* A; int64_t lda;
    const float* Bt; int64_t ldb;
    int64_t N, K;
    int32_t a_kind, epilogue, act, nchunk;
    const float* bias; const float* add; int64_t ldadd;
    float* C; int64_t ldc;
    const float* pctx; const float* pstat;
    const float* energies; const float* mask; float* weights; int64_t S, mask_div, mask_mod;
    const float* h; int64_t ldh;
    float* ru; float* rh;
    const float* xc; int64_t ldxc;
    float* h_out; int64_t ldho; float* h_out2; int64_t ldho2;
    /* optional row indirection of the epilogue operands: `add` (epilogues 0 and 1: added to the sum before the
     * activation / the sigmoid) is read at row add_ids[row], `xc` (epilogue 2) at row xc_ids[row] -- rows of a
     * table indexed by the step's input symbols (nm_decoder_step.in_table) */
    const int32_t* add_ids; const int32_t* xc_ids;
} nm_step_problem;
int nm_step_group(void* stream, int64_t M, const nm_step_problem* problems, int32_t nproblems);

/* ---- COSTING ONLY (not called by the product path): C[M,N] = A[M,K] . B[N,K]^T with the fp32 operands emulated by
 * three bf16 matrix-core products (split-bf16: hi.hi + hi.lo + lo.hi, fp32 accumulate; terms = 1: plain bf16).  The
 * shapes of tf.matmul(state, decoding_w) with tied embeddings (decoders/autoregressive.py:226-251,450-459) and of
 * its input gradient.  K % 4 == 0, rows 16-byte aligned.  See csrc/nm_gemm_bf16x3.hip, tools/gemm_bf16x3_cost.py. */
int nm_gemm_bf16x3_nt(void* stream, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, int terms, int variant /* tile / k-depth choice, 0..3 */);

/* ---- OPT-IN: the decoding steps' vocabulary projection (tf.matmul(state, decoding_w) + bias,
 * decoders/autoregressive.py:450-459) on the bf16 matrix cores with fp32-class accuracy (csrc/nm_gemm_bf16x3.hip):
 * both operands split three ways into bf16 (24 mantissa bits), six products, fp32 accumulate.  The weights are
 * split once: `prepare` fills `planes` (split_bytes(N, K) bytes, 16-byte aligned; K % 16 == 0; W is [K][N], or
 * [N][K] with trans_b) and registers W -- from then on nm_logits_stats_gemm called with this B pointer (same N, K,
 * trans_b; 128-column statistics tiles) takes the split kernel; `forget` (NULL: every matrix) returns it to the
 * exact-fp32 kernel, which stays the default.  The caller re-prepares when W changes. */
int64_t nm_proj_split_bytes(int64_t N, int64_t K);
int nm_proj_split_prepare(void* stream, const float* W, int64_t ldw, int trans_b, int64_t N, int64_t K, void* planes,
                          int64_t planes_bytes);
int nm_proj_split_forget(const float* W);

/* ---- the WHOLE inference step of the headline decoder behind one call (SURVEY 8(b)4 nm_decoder_step_fused):
 * Decoder.next_state, decoders/decoder.py:279-358 (plain GRUCell nn/ortho_gru_cell.py:44-53, ONE Bahdanau
 * attention attention/feed_forward.py:120-166, nonlinear output projection decoders/output_projection.py:115-130)
 * followed by the vocabulary projection of get_body, decoders/autoregressive.py:450-459.
 * Seven launches on `stream`: nm_step_group x3, nm_attn_fwd, nm_step_group, nm_logits_stats_gemm (plain
 * nm_gemm_f32 when stats == NULL); a caller that replays the step captures this call in a HIP graph.
 *   cat        [rows, emb+rnn] = [embedded input symbol | state]: the persistent input row.  The state half is
 *              REPLACED by h'; the caller embeds the next symbols into the left half (nm_greedy_finish emb_out,
 *              ld_emb = emb+rnn) and -- beam search -- gathers the surviving states into the right half.
 *   h_copy     optional second copy of h' (ld_h_copy), the state history of a beam search
 *   out_state  [rows, out] the projected output the logits are computed from
 *   attn_weights [rows, src_len] the distribution of this step, or NULL
 *   logits / stats   as nm_logits_stats_gemm (logits may be NULL with stats: greedy decoding)
 *   ru rh xc y pre_e pre ctx   dense scratch [rows, 2*rnn | rnn | rnn | attn_state | out | out | ctx_width]
 *   attn_workspace   nm_attn_workspace_bytes(rows, src_len, ctx_width), zeroed once after allocation
 *   *_t        parameters TRANSPOSED to [N,K]: wg_t [2*rnn, emb+rnn] (gates kernel), wcx_t [rnn, emb] / wch_t
 *              [rnn, rnn] (candidate kernel rows of the input / of the state), wq_t [attn_state, rnn] (query
 *              projection), wo_h_t / wo_e_t / wo_c_t [out, rnn | emb | ctx_width] (output projection kernel rows
 *              [state | embedded input | context]); biases bg [2*rnn], bc [rnn], bq [attn_state] or NULL, bo [out]
 *   keys [Bk,src_len,attn_state], values [Bk,src_len,ctx_width], mask [Bk,src_len], v [attn_state], attn_bias [1]
 *              with Bk = rows / rows_per_key
 *   w_vocab    [out, vocab] (vocab_trans_b: [vocab, out], tied embeddings), ld_w_vocab; b_vocab [vocab] or NULL
 *   out_act    0 none / 1 tanh.  emb, rnn, ctx_width multiples of 16; all pointers 16-byte aligned. */
typedef struct nm_decoder_step {
    int64_t rows, emb, rnn, attn_state, ctx_width, out, vocab, src_len, rows_per_key;
    float* cat;
    float* h_copy; int64_t ld_h_copy;
    float* out_state; int64_t ld_out_state;
    float* attn_weights;
    float* logits; int64_t ld_logits;
    float* stats; int64_t stats_bytes;
    float* ru; float* rh; float* xc; float* y; float* pre_e; float* pre; float* ctx;
    void* attn_workspace; int64_t attn_workspace_bytes;
    const float* wg_t; const float* bg; const float* wcx_t; const float* wch_t; const float* bc;
    const float* wq_t; const float* bq; const float* keys; const float* values; const float* mask;
    const float* v; const float* attn_bias;
    const float* wo_h_t; const float* wo_e_t; const float* wo_c_t; const float* bo;
    const float* w_vocab; int64_t ld_w_vocab; const float* b_vocab;
    int32_t out_act, vocab_trans_b;
    /* leading dimensions in floats, 0 = dense (emb + rnn for cat and wg_t, the K of the product for the other
     * transposed weights, ctx_width for ctx).  Power-of-two row strides (4 KB at the benchmark shape) put every
     * row of an operand tile on the same L2 channel; a caller that pads its rows by 128 bytes spreads them. */
    int64_t ld_cat, ld_ctx, ld_wg, ld_wcx, ld_wch, ld_wq, ld_wo_h, ld_wo_e, ld_wo_c;
    /* Input tables (optional).  Everything the step computes from the EMBEDDED INPUT SYMBOL alone is a function of
     * the symbol: in_table [V, 2*rnn + rnn + out] = [E.Wg_x | E.Wc_x + bc | E.Wo_e] (E = the embedding matrix; one
     * GEMM per set of weights).  With in_table and in_ids (the rows' input symbols) the embedding half of group 1
     * becomes a gather in the epilogues: gates = sigmoid(h.Wg_h + in_table[id, :2*rnn] + bg), xc = in_table[id,
     * 2*rnn:3*rnn], the output projection adds in_table[id, 3*rnn:]; the left half of `cat` is then not read. */
    const float* in_table; int64_t ld_table; const int32_t* in_ids;
    /* Optional, with input tables: a ZERO-initialised workspace of nm_dec_step_cluster_workspace_bytes(rows, rnn)
     * bytes that belongs to this decoder alone.  When the shape is taken (nm_dec_step_cluster_supported: greedy-sized
     * steps, rnn 256 / 384 / 512) the gates, the candidate + blend, the attention query and the state part of the
     * output projection (decoders/decoder.py:279-325) run as ONE launch of workgroup clusters with tagged hand-offs
     * instead of three dependent launches; `sticky_error` is the error word of nm_gru_seq_fwd: set when the hand-offs
     * timed out (the step's results are garbage then; the caller runs the batch again without the workspace). */
    void* cluster_ws; int64_t cluster_ws_bytes; uint32_t* sticky_error;
} nm_decoder_step;
int nm_decoder_step_fused(void* stream, const nm_decoder_step* step);
int nm_dec_step_cluster_supported(int64_t rows, int64_t rnn, int64_t attn_state, int64_t out);
int64_t nm_dec_step_cluster_workspace_bytes(int64_t rows, int64_t rnn);

/* ---- vocabulary-axis rows: tf.argmax / tf.nn.log_softmax / sequence_loss ----------------------
 * decoders/autoregressive.py:470 (argmax, first max wins), :289-316,351-375 (xent, log-probs) */
int nm_row_stats(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t V, float* max_out,
                 float* lse_out, int32_t* argmax_out);
/* one draw per row from softmax(x): tf.multinomial(logits, 1) of the sampling decoder body
 * (decoders/autoregressive.py:470-473), as argmax(x + Gumbel noise) with counter-based noise of (salt, row, column) */
int nm_gumbel_argmax(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t V, uint32_t salt,
                     int32_t* out);
int nm_log_softmax(void* stream, const float* x, int64_t ldx, const float* rmax, const float* rlse,
                   float* out, int64_t ldo, int64_t rows, int64_t V);
int nm_greedy_update(void* stream, const int32_t* argmax, int32_t* finished, int32_t* sym_out,
                     int32_t* mask_out, int64_t n, int end_id, int32_t* all_finished);
/* label_smoothing eps (decoders/autoregressive.py:292-299, tf.losses.softmax_cross_entropy): the
 * target distribution is (1-eps)*onehot + eps/V */
int nm_xent(void* stream, float* logits, int64_t ldx, int64_t rows, int64_t V, const int32_t* targets,
            const float* weights, float* loss_rows, const float* grad_scale, int write_grad,
            float label_smoothing);
/* The same with the column sums of the gradient -- the bias gradient of the vocabulary projection
 * (decoders/autoregressive.py:450-459 under tf.gradients) -- accumulated on the way: partial[g, :] receives the sums
 * over logits rows g, g + partial_rows, ...; nm_colsum over `partial` finishes the reduction.  V % 4 == 0, V <= 32768. */
int nm_xent_colsum(void* stream, float* logits, int64_t ldx, int64_t rows, int64_t V, const int32_t* targets,
                   const float* weights, float* loss_rows, const float* grad_scale, float label_smoothing,
                   float* partial, int64_t partial_rows);

/* ---- beam search step: decoders/beam_search_decoder.py:440-501 (mask, + logprob_sum, length
 * penalty, tf.nn.top_k over [B,k*V] with lower-index-first ties, div/mod, gathers) and the
 * state / history reordering :503-551 (tf_utils.py:106-131 gather_flat) */
int64_t nm_beam_workspace_bytes(int64_t B, int64_t k, int64_t V);
int nm_beam_topk_step(void* stream, const float* logits, int64_t ldx, int64_t B, int64_t k, int64_t V,
                      const float* rmax, const float* rlse, const float* logprob_sum,
                      const int32_t* lengths, const int32_t* finished, const float* penalty, int end_id,
                      float* out_score, int32_t* out_word, int32_t* out_beam, float* out_logprob_sum,
                      int32_t* out_lengths, int32_t* out_finished, int32_t* out_src_row, void* workspace,
                      int64_t workspace_bytes, int32_t* all_finished);
/* the same step on the raw logits of the previous parent step: max / lse / candidates of every row come
 * from ONE register-resident scan of the row (no separate nm_row_stats pass); rmax_out / rlse_out
 * [B*k] receive the row statistics */
int nm_beam_topk_step_fused(void* stream, const float* logits, int64_t ldx, int64_t B, int64_t k, int64_t V,
                            const float* logprob_sum, const int32_t* lengths, const int32_t* finished,
                            const float* penalty, int end_id, float* out_score, int32_t* out_word,
                            int32_t* out_beam, float* out_logprob_sum, int32_t* out_lengths,
                            int32_t* out_finished, int32_t* out_src_row, void* workspace,
                            int64_t workspace_bytes, int32_t* all_finished, float* rmax_out, float* rlse_out);
/* ---- the vocabulary projection with its row statistics in the GEMM epilogue ----------------------------
 * decoders/autoregressive.py:450-459 (logits = state.W + b) fused with what the decoding loops do to the
 * logits next: tf.argmax (:470) and tf.nn.log_softmax (beam_search_decoder.py:537-543).  Every 128-column
 * tile of a row leaves {max, sum exp(x - max), first argmax (int bits), -} in stats[row][tile]; the logits
 * themselves are written only when C != NULL (greedy decoding never reads them back: nm_greedy_finish works
 * on the statistics; a beam step reads back only the tiles that can hold a top-k candidate).
 * transB: B stored [N,K] (tied embeddings).  A, B 16-byte aligned, K, N, lda, ldb multiples of 4. */
int64_t nm_logits_stats_tile(int64_t M);     /* columns per statistics tile: 64 for M <= 256 rows, else 128 */
int64_t nm_logits_stats_bytes(int64_t M, int64_t N);
int nm_logits_stats_gemm(void* stream, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                         const float* B, int64_t ldb, const float* bias, float* C /* or NULL */, int64_t ldc,
                         float* stats, int64_t stats_bytes);
/* greedy step tail on the statistics (decoders/autoregressive.py:461-480 + the embedding lookup of the next
 * input, :269-272): sym = finished ? 0 : argmax; finished |= sym == </s>; mask = !finished;
 * emb_out[r,:] = table[sym[r],:] (skipped when emb_out == NULL).  argmax / max / lse outputs optional. */
int nm_greedy_finish(void* stream, const float* stats, int64_t ntiles, int64_t R, int32_t* finished,
                     int32_t* sym_out, int32_t* mask_out, int end_id, int32_t* all_finished, const float* table,
                     int64_t V, int64_t E, float* emb_out, int64_t ld_emb, int32_t* argmax_out, float* max_out,
                     float* lse_out);
/* nm_beam_topk_step_fused on logits whose tile statistics are known: max / lse from the merged tiles, the
 * exact top-k from the few tiles whose maximum can reach it (same scores, same tie order) */
int nm_beam_topk_step_tiles(void* stream, const float* logits, int64_t ldx, const float* stats, int64_t tile_w,
                            int64_t B, int64_t k, int64_t V, const float* logprob_sum, const int32_t* lengths,
                            const int32_t* finished, const float* penalty, int end_id, float* out_score,
                            int32_t* out_word, int32_t* out_beam, float* out_logprob_sum, int32_t* out_lengths,
                            int32_t* out_finished, int32_t* out_src_row, void* workspace, int64_t workspace_bytes,
                            int32_t* all_finished, float* rmax_out, float* rlse_out);
int nm_gather_rows_f32(void* stream, const float* src, int64_t ld_src, const int32_t* idx, float* dst,
                       int64_t ld_dst, int64_t rows, int64_t width);
int nm_beam_reorder_tokens(void* stream, const int32_t* src, const int32_t* src_row, const int32_t* word,
                           int32_t* dst, int64_t steps, int64_t R);
/* the same histories from back-pointers, once per search instead of once per step: src_row / word [steps,R] are
 * the out_src_row / out_word of every beam body, first [R] the parent's initial symbols;
 * out[t+1,r] = word[t, ancestor_t(r)], out[0,r] = first[ancestor_0(r)]  (beam_search_decoder.py:546-551) */
int nm_beam_backtrace(void* stream, const int32_t* src_row, const int32_t* word, const int32_t* first,
                      int32_t* out, int64_t steps, int64_t R);

/* ---- small utilities -------------------------------------------------------------------------- */
int nm_copy_cols(void* stream, const float* src, int64_t ld_src, float* dst, int64_t ld_dst,
                 int64_t rows, int64_t width);
int nm_reduce_sum(void* stream, const float* x, int64_t n, float* out);
int nm_tanh_bwd(void* stream, float* dy, const float* y, int64_t n);
/* out[c] (+)= sum_r x[r,c] (bias / LayerNorm gradients: tf.gradients of a broadcast add), ONE launch, sums in a
 * fixed order.  The workspace (device) must be ZERO when first used and belongs to one stream at a time: its tail
 * holds arrival counters that the kernel itself puts back to zero. */
int64_t nm_colsum_workspace_bytes(int64_t cols);
int nm_colsum(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t cols, float* out,
              int accumulate, void* workspace, int64_t workspace_bytes);
/* The same with the kernel named: algo 0 = nm_colsum's choice, 1 = the low-pressure kernel (one float per thread and
 * row) for launches that share the chip with a cluster time loop of another stream. */
int nm_colsum_algo(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t cols, float* out, int accumulate,
                   void* workspace, int64_t workspace_bytes, int algo);

/* ---- strided element-wise primitives of the general (taped) path -----------------------------
 * Cells other than the fused TF GRU (NematusGRUCell nn/ortho_gru_cell.py:57-105, LSTMCell
 * decoders/decoder.py:309-325), conditional GRU (decoders/decoder.py:303-307), attention on input
 * (:264-277), output-projection variants (decoders/output_projection.py:35-188) and the
 * Transformer blocks are compositions of MFMA GEMMs and these one-pass kernels.
 * op codes: 0 copy, 1 a+b, 2 a-b, 3 a*b, 4 alpha*a, 5 sigmoid(a+alpha), 6 tanh(a), 7 relu(a),
 *           8 b*a*(1-a), 9 b*(1-a^2), 10 b*(a>0)   (8-10: a = forward output, b = upstream grad),
 *           11 log(exp(a)+exp(b)) (ensemble mean in log space, runners/beamsearch_runner.py:50-55),
 *           12 a+alpha, 13 a[r,c]*b[r,0] (per-row scalar), 14 a/b (attention/coverage.py:57) */
int nm_ew(void* stream, int op, const float* a, int64_t lda, const float* b, int64_t ldb, float* out,
          int64_t ldo, int64_t rows, int64_t cols, float alpha, int accumulate);
/* h' = u*h + (1-u)*c  and its gradient (du, dh, dc accumulate; any may be NULL) */
int nm_blend_fwd(void* stream, const float* u, int64_t ldu, const float* h, int64_t ldh, const float* c,
                 int64_t ldc, float* out, int64_t ldo, int64_t rows, int64_t cols);
int nm_blend_bwd(void* stream, const float* dy, int64_t lddy, const float* u, int64_t ldu, const float* h,
                 int64_t ldh, const float* c, int64_t ldc, float* du, int64_t lddu, float* dh,
                 int64_t lddh, float* dc, int64_t lddc, int64_t rows, int64_t cols);
/* The point-wise part of one LSTMCell step (SURVEY 8(b)4 nm_lstm_cell_{fwd,bwd}; tf.nn.rnn_cell.LSTMCell as the
 * reference uses it: decoders/decoder.py:29,309-325, encoders/recurrent.py:21).  z [rows, 4H] = [x, h].W + b from the
 * caller's products, gate order i, j, f, o:  c' = sigmoid(f + forget_bias) c + sigmoid(i) tanh(j),
 * h' = sigmoid(o) tanh(c').  gates (optional) receives the ACTIVATED gates [rows, 4H] for the backward.
 * bwd: dh / dc_new may be NULL (no gradient from that side); dz is written (accumulate_dz 0) or added to;
 * dc_prev (optional) likewise. */
int nm_lstm_cell_fwd(void* stream, const float* z, int64_t ldz, const float* c_prev, int64_t ldc, float* c_new,
                     int64_t ldcn, float* h_new, int64_t ldh, float* gates, int64_t ldg, int64_t rows, int64_t H,
                     float forget_bias);
int nm_lstm_cell_bwd(void* stream, const float* dh, int64_t lddh, const float* dc_new, int64_t lddc,
                     const float* gates, int64_t ldg, const float* c_prev, int64_t ldc, const float* c_new,
                     int64_t ldcn, float* dz, int64_t lddz, float* dc_prev, int64_t lddcp, int64_t rows, int64_t H,
                     int accumulate_dz, int accumulate_dc_prev);
/* The point-wise part of one NematusGRUCell step (nn/ortho_gru_cell.py:73-105) after its four products:
 * [r | u] = sigmoid(g_pre), c = tanh(ci + sc * r), h' = u h + (1 - u) c; g_pre [rows, 2H] = input + state gate
 * projections, sc / ci = state / input candidate projections.  ru [rows, 2H] and c_out [rows, H] (contiguous, may be
 * null) keep what the backward call reads.  bwd: dg = [dr' | du'], dci, dsc, dh_prev (each written or added to per its
 * flag; dci / dsc / dh_prev may be null).  g2 (may be null): a second gate operand added to g_pre before the sigmoid
 * -- the input half when the state and input projections of the step are two products [gates | candidate] instead of
 * four; dg2 (may be null): a second destination that receives dg as well (the gradient of that other product). */
int nm_nematus_cell_fwd(void* stream, const float* g_pre, int64_t ldg, const float* sc, int64_t ldsc, const float* ci,
                        int64_t ldci, const float* h_prev, int64_t ldh, float* h_new, int64_t ldhn, float* ru,
                        float* c_out, const float* g2, int64_t ldg2, int64_t rows, int64_t H);
int nm_nematus_cell_bwd(void* stream, const float* dh, int64_t lddh, const float* ru, const float* c, const float* sc,
                        int64_t ldsc, const float* h_prev, int64_t ldh, float* dg, int64_t lddg, float* dci,
                        int64_t lddci, float* dsc, int64_t lddsc, float* dh_prev, int64_t lddhp, float* dg2, int64_t lddg2,
                        int64_t rows, int64_t H, int accumulate_dg, int accumulate_dci, int accumulate_dsc,
                        int accumulate_dh_prev);
/* The state half of a NematusGRUCell step AND its point-wise part in one launch (nn/ortho_gru_cell.py:73-105; the
 * reset gate multiplies the state projection, so nothing of the step waits for a second product): s = h_prev . w_st
 * (+ b_st), w_st [H, 3H] = [U_g | U_c]; r, u = sigmoid(x_all[:, :2H] + s[:, :2H]); c = tanh(x_all[:, 2H:] + r * s[:, 2H:]);
 * h_new = u h_prev + (1 - u) c.  x_all [rows, 3H]: the input half x . [W_g | W_c] + biases.  ru [rows, 2H], c_out
 * [rows, H] (contiguous) and sc_out [rows, H] = s[:, 2H:] keep what nm_nematus_cell_bwd reads; all three may be null.
 * H in steps of 8; h_new may not alias h_prev. */
int nm_nematus_state_step(void* stream, const float* h_prev, int64_t ldh, const float* w_st, int64_t ldw,
                          const float* b_st, const float* x_all, int64_t ldx, float* h_new, int64_t ldhn, float* ru,
                          float* c_out, float* sc_out, int64_t ldsc, int64_t rows, int64_t H);
/* ... and with the step's input half in the same launch: x [rows, D] . w_in [D, 3H] = [W_g | W_c] (+ b_in) takes the
 * place of x_all (the second cell of a conditional decoder, decoders/decoder.py:303-325: its input is the step's own
 * attention context).  H and D in steps of 8. */
int nm_nematus_full_step(void* stream, const float* h_prev, int64_t ldh, const float* w_st, int64_t ldw,
                         const float* b_st, const float* x, int64_t ldx, const float* w_in, int64_t ldwi,
                         const float* b_in, float* h_new, int64_t ldhn, float* ru, float* c_out, float* sc_out,
                         int64_t ldsc, int64_t rows, int64_t H, int64_t D);
/* nn/utils.py:6-22 (tf.nn.dropout): keep iff floor(keep_prob + u_i) == 1, scale 1/keep_prob;
 * u_i = hash(salt, i) (counter based: the backward pass and the CPU oracle regenerate the mask) */
int nm_dropout(void* stream, const float* x, int64_t ldx, float* out, int64_t ldo, int64_t rows,
               int64_t cols, float keep_prob, uint32_t salt, const uint32_t* step /* optional device
               scalar (the global step): salt += step * 0x9E3779B9, so a HIP-graph replay of a training
               step draws fresh masks */, int accumulate);
/* tf.nn.dynamic_rnn(sequence_length) step t (encoders/recurrent.py:86-110): rows with
 * t >= lengths[r] carry h_prev through and emit zeros */
int nm_rnn_select_fwd(void* stream, const float* h_new, int64_t ld_new, const float* h_prev,
                      int64_t ld_prev, const int32_t* lengths, int t, float* h_out, int64_t ld_h,
                      float* y_out, int64_t ld_y, int64_t rows, int64_t cols);
int nm_rnn_select_bwd(void* stream, const float* dh, int64_t ld_dh, const float* dy, int64_t ld_dy,
                      const int32_t* lengths, int t, float* d_new, int64_t ld_dnew, float* d_prev,
                      int64_t ld_dprev, int64_t rows, int64_t cols);
/* tf.reverse_sequence(x [B,S,D], lengths, seq_axis=1); self-inverse */
int nm_reverse_sequence(void* stream, const float* x, float* out, const int32_t* lengths, int64_t B,
                        int64_t S, int64_t D, int accumulate);
/* nn/projection.py:7-35 maxout: out[r,g] = max_p x[r, p*groups + g] */
int nm_maxout_fwd(void* stream, const float* x, int64_t ldx, float* out, int64_t ldo, int32_t* argmax,
                  int64_t rows, int64_t groups, int64_t pool);
int nm_maxout_bwd(void* stream, const float* dy, int64_t lddy, const int32_t* argmax, float* dx,
                  int64_t lddx, int64_t rows, int64_t groups, int64_t pool);

/* ---- Transformer blocks: attention/scaled_dot_product.py:98-226, encoders/transformer.py,
 * decoders/transformer.py ------------------------------------------------------------------------
 * Multi-head scaled dot-product attention over [B, T, H*dh] tensors (heads = column blocks):
 * energies = (q/sqrt(dh)).k^T, optional future mask (where(tril, e, -1e9)), optional key mask
 * (e*m + (1-m)*-1e9), softmax, dropout on the weights (counter-based mask over the flattened
 * [Bq,H,Tq,Tk] tensor), context = w.v.  *_bs are batch strides in floats; query row r reads key
 * batch r / rows_per_key (a beam shares its encoder keys; Tq = 1 against a key/value cache is a
 * decoding step).  `weights` [Bq,H,Tq,Tk] receives the softmax output (needed by the backward). */
int nm_sdp_attn_fwd(void* stream, const float* q, int64_t q_bs, const float* k, int64_t k_bs,
                    const float* v, int64_t v_bs, const float* key_mask, int64_t mask_bs, int64_t Bq,
                    int64_t rows_per_key, int64_t Tq, int64_t Tk, int64_t H, int64_t dh, int causal,
                    float keep_prob, uint32_t salt, const uint32_t* step /* as nm_dropout */, float* ctx,
                    int64_t ctx_bs, float* weights);
/* One decoding step (Tq = 1) against a key/value cache addressed through an ancestor table: position j of query
 * row b is read from cache row ancestors[b * anc_ld + j].  Beam search (beam_search_decoder.py:218-330 gathers the
 * decoder's loop state, i.e. every layer's cached keys and values, at every step) re-points rows at their
 * ancestors instead of copying the caches. */
int nm_sdp_attn_step(void* stream, const float* q, int64_t q_bs, const float* k, int64_t k_bs, const float* v,
                     int64_t v_bs, const float* key_mask, int64_t mask_bs, int64_t Bq, int64_t Tk, int64_t H,
                     int64_t dh, const int32_t* ancestors, int64_t anc_ld, float* ctx, int64_t ctx_bs,
                     float* weights);
int nm_sdp_attn_bwd(void* stream, const float* q, int64_t q_bs, const float* k, int64_t k_bs,
                    const float* v, int64_t v_bs, const float* key_mask, int64_t mask_bs,
                    const float* weights, const float* dctx, int64_t dctx_bs, int64_t B, int64_t Tq,
                    int64_t Tk, int64_t H, int64_t dh, int causal, float keep_prob, uint32_t salt,
                    const uint32_t* step, float* dq, int64_t dq_bs, float* dk, int64_t dk_bs, float* dv, int64_t dv_bs,
                    float* de_workspace /* [B,H,Tq,Tk] */, int accumulate);
/* out[b,t,:] = x[b,t,:] + signal[t0+t,:]  (position_signal, encoders/transformer.py:23-45) */
int nm_add_position(void* stream, const float* x, const float* signal, float* out, int64_t B, int64_t T,
                    int64_t D, int64_t t0);
/* out[r*ld_out] = finished[r] ? 0 : 1: the key-mask column of a new decoding position
 * (decoders/transformer.py:493-497) */
int nm_unfinished_mask(void* stream, const int32_t* finished, float* out, int64_t ld_out, int64_t n);
/* TransformerEncoder.output = sum over time (encoders/transformer.py:170-172) and its gradient */
int nm_time_sum(void* stream, const float* x, float* out, int64_t B, int64_t T, int64_t D);
int nm_time_bcast_add(void* stream, const float* dy, float* dx, int64_t B, int64_t T, int64_t D);

/* ---- trainer arithmetic over the flat parameter buffer: trainers/generic_trainer.py:84-195
 * (L1/L2 over non-bias variables, per-tensor tf.clip_by_norm, tf.train.AdamOptimizer) ----------- */
int64_t nm_optim_workspace_bytes(int64_t nchunk, int64_t nseg);
int nm_optim_regularize_norms(void* stream, const float* theta, float* grad, const int64_t* chunk_start,
                              const int32_t* chunk_len, const int32_t* chunk_seg, const int32_t* seg_first,
                              const int32_t* seg_count, const int32_t* seg_flags, int64_t nchunk,
                              int64_t nseg, float l1_weight, float l2_weight, float* l1l2_out,
                              void* workspace, int64_t workspace_bytes);
int nm_optim_clip_adam(void* stream, float* theta, const float* grad, float* m, float* v,
                       const int64_t* chunk_start, const int32_t* chunk_len, const int32_t* chunk_seg,
                       const int32_t* seg_first, const int32_t* seg_count, const int32_t* seg_flags,
                       int64_t nchunk, int64_t nseg, float clip_norm, float lr_t, float beta1,
                       float beta2, float epsilon, void* workspace, int64_t workspace_bytes);
/* tf.train.AdadeltaOptimizer (tests/bpe.ini:102-108, tests/str.ini:100-106; TF 1.12 ApplyAdadelta) behind the same
 * per-tensor clip: accum / accum_update are the optimizer's two slots, `lr` the plain learning rate. */
int nm_optim_clip_adadelta(void* stream, float* theta, const float* grad, float* accum, float* accum_update,
                           const int64_t* chunk_start, const int32_t* chunk_len, const int32_t* chunk_seg,
                           const int32_t* seg_first, const int32_t* seg_count, const int32_t* seg_flags,
                           int64_t nchunk, int64_t nseg, float clip_norm, float lr, float rho, float epsilon,
                           void* workspace, int64_t workspace_bytes);

/* The same three passes over a RANGE of chunks, for a rank that owns a slice of the flat buffers (sharded optimizer,
 * reduce-scatter -> Adam on the rank's slice -> all-gather; the reference applies the identical
 * update on one device, trainers/generic_trainer.py:179-195).  nm_optim_partials: regulariser terms into `grad` and
 * the per-chunk partial sums into the workspace for chunks [chunk_begin, chunk_end); the partial vector (the first
 * 3 * nchunk floats of the workspace, zero where no rank wrote) is summed over ranks -- one non-zero contributor per
 * entry, so the sum is exact; nm_optim_segments: per-tensor squared norms + global L1 / L2 from the whole partial
 * vector in a fixed order; nm_optim_apply: per-tensor clip + update of chunks [chunk_begin, chunk_end), kind 0 = Adam
 * (p0..p3 = lr_t, beta1, beta2, epsilon), kind 1 = Adadelta (lr, rho, epsilon, -).  `skip_word` (may be null): a
 * device word that turns the launch into a no-op when it is not zero -- the error word of nm_gru_seq_fwd / _bwd: the
 * update of a step whose time loop gave up is never applied, the caller runs the step again. */
int nm_optim_partials(void* stream, const float* theta, float* grad, const int64_t* chunk_start,
                      const int32_t* chunk_len, const int32_t* chunk_seg, const int32_t* seg_first,
                      const int32_t* seg_count, const int32_t* seg_flags, int64_t nchunk, int64_t nseg,
                      float l1_weight, float l2_weight, int64_t chunk_begin, int64_t chunk_end, void* workspace,
                      int64_t workspace_bytes);
int nm_optim_segments(void* stream, const int64_t* chunk_start, const int32_t* chunk_len, const int32_t* chunk_seg,
                      const int32_t* seg_first, const int32_t* seg_count, const int32_t* seg_flags, int64_t nchunk,
                      int64_t nseg, float* l1l2_out, void* workspace, int64_t workspace_bytes);
int nm_optim_apply(void* stream, int32_t kind, float* theta, const float* grad, float* slot0, float* slot1,
                   const int64_t* chunk_start, const int32_t* chunk_len, const int32_t* chunk_seg,
                   const int32_t* seg_first, const int32_t* seg_count, const int32_t* seg_flags, int64_t nchunk,
                   int64_t nseg, float clip_norm, float p0, float p1, float p2, float p3, int64_t chunk_begin,
                   int64_t chunk_end, const int32_t* skip_word, void* workspace, int64_t workspace_bytes);
/* nm_optim_partials / nm_optim_apply over a LIST of chunks (device array of `count` chunk indices, any order): all the
 * chunks a rank of the sharded optimizer owns -- one slice per bucket plus the shared tails -- in one launch each. */
int nm_optim_partials_list(void* stream, const float* theta, float* grad, const int64_t* chunk_start,
                           const int32_t* chunk_len, const int32_t* chunk_seg, const int32_t* seg_first,
                           const int32_t* seg_count, const int32_t* seg_flags, int64_t nchunk, int64_t nseg,
                           float l1_weight, float l2_weight, const int32_t* chunk_list, int64_t count, void* workspace,
                           int64_t workspace_bytes);
int nm_optim_apply_list(void* stream, int32_t kind, float* theta, const float* grad, float* slot0, float* slot1,
                        const int64_t* chunk_start, const int32_t* chunk_len, const int32_t* chunk_seg,
                        const int32_t* seg_first, const int32_t* seg_count, const int32_t* seg_flags, int64_t nchunk,
                        int64_t nseg, float clip_norm, float p0, float p1, float p2, float p3, const int32_t* chunk_list,
                        int64_t count, const int32_t* skip_word, void* workspace, int64_t workspace_bytes);
/* x[0..n) = 0 when *word != 0: the gradient a given-up time loop left behind must not reach an accumulation buffer
 * (trainers/delayed_update_trainer.py:146-150) or a collective as NaNs */
int nm_zero_if(void* stream, const int32_t* word, float* x, int64_t n);
/* Whole-buffer fills (a 4-byte pattern: float and int32 buffers alike; a kernel of this library) and device-to-device
 * copies (the runtime's: a memcpy node inside a captured graph) instead of a tensor library's fill / copy kernels
 * (the reference's tf.zeros / tf.assign of its state variables are graph nodes the same way). */
int nm_fill_u32(void* stream, void* x, int64_t count, uint32_t pattern);
int nm_copy_d2d(void* stream, void* dst, const void* src, int64_t bytes);

/* ---- data-parallel gradient exchange (the reference is single-device, tf_manager.py:62-100): the
 * in-place sum over ranks of slices of the flat gradient buffer on RCCL, ordered against HIP streams only.  RCCL is
 * resolved at run time (the copy already in the process, else librccl.so.1): no link-time dependency.
 *   unique_id  rank 0 fills 128 bytes, the caller distributes them;  init  one communicator (+ its own stream) per
 *   process on the current device;  bucket  buf[0:count] <- sum over ranks, after everything enqueued on `stream` so
 *   far, running beside what `stream` enqueues next;  wait  `stream` waits on the device for all buckets so far. */
typedef void nm_comm;
int nm_allreduce_unique_id(void* out, int64_t bytes /* >= 128 */);
int nm_allreduce_init(int rank, int world, const void* unique_id, nm_comm** out_comm);
int nm_allreduce_bucket(nm_comm* comm, void* stream, float* buf, int64_t count);
int nm_allreduce_wait(nm_comm* comm, void* stream);
int nm_allreduce_destroy(nm_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* NMHIP_H */
