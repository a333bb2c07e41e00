// libnmhip: strided element-wise primitives, dropout masks and sequence utilities used by the
// general (taped) model path -- NematusGRU / LSTM cells, conditional GRU, attention on input,
// dropout, the output-projection variants and the Transformer blocks.  All of this is
// HBM-bound glue between the MFMA GEMMs: one coalesced pass per operand, no LDS.
//
//   nm_ew               out[r,c] (+)= f(a[r,c], b[r,c])      f selected by an op code
//   nm_blend_fwd/bwd    h' = u*h + (1-u)*c                   (last line of every GRU variant)
//   nm_dropout          tf.nn.dropout with a counter-based mask (nn/utils.py:6-22)
//   nm_rnn_select_*     dynamic_rnn length masking           (encoders/recurrent.py:86-110)
//   nm_reverse_sequence tf.reverse_sequence by lengths       (bidirectional_dynamic_rnn)
//   nm_maxout_fwd/bwd   max-pool of a dense output           (nn/projection.py:7-35)
#include "nm_common.h"

// ---------------------------------------------------------------------------------------------
// op codes of nm_ew (mirrored in neuralmonkey_amd/ops.py)
// ---------------------------------------------------------------------------------------------
enum {
    NM_EW_COPY = 0,         // a
    NM_EW_ADD = 1,          // a + b
    NM_EW_SUB = 2,          // a - b
    NM_EW_MUL = 3,          // a * b
    NM_EW_SCALE = 4,        // alpha * a
    NM_EW_SIGMOID = 5,      // sigmoid(a + alpha)
    NM_EW_TANH = 6,         // tanh(a)
    NM_EW_RELU = 7,         // max(a, 0)
    NM_EW_SIGMOID_BWD = 8,  // b * a * (1 - a)        a = forward output, b = upstream gradient
    NM_EW_TANH_BWD = 9,     // b * (1 - a^2)
    NM_EW_RELU_BWD = 10,    // b * (a > 0)
    NM_EW_LOGADDEXP = 11,   // log(exp(a) + exp(b))   (ensemble mean of probabilities in log space)
    NM_EW_ADD_SCALAR = 12,  // a + alpha
    NM_EW_ROWSCALE = 13,    // a[r,c] * b[r,0]        (a per-row scalar: attention weight of a single vector)
    NM_EW_DIV = 14,         // a / b                  (coverage / fertility, attention/coverage.py:57)
    NM_EW_OPS = 15
};

template <int OP>
__device__ __forceinline__ float ew_apply(float a, float b, float alpha) {
    if (OP == NM_EW_COPY) return a;
    if (OP == NM_EW_ADD) return a + b;
    if (OP == NM_EW_SUB) return a - b;
    if (OP == NM_EW_MUL) return a * b;
    if (OP == NM_EW_SCALE) return alpha * a;
    if (OP == NM_EW_SIGMOID) return nm_sigmoid(a + alpha);
    if (OP == NM_EW_TANH) return nm_tanh(a);
    if (OP == NM_EW_RELU) return fmaxf(a, 0.0f);
    if (OP == NM_EW_SIGMOID_BWD) return b * a * (1.0f - a);
    if (OP == NM_EW_TANH_BWD) return b * (1.0f - a * a);
    if (OP == NM_EW_RELU_BWD) return a > 0.0f ? b : 0.0f;
    if (OP == NM_EW_LOGADDEXP) {
        const float hi = fmaxf(a, b), lo = fminf(a, b);
        return hi == -INFINITY ? -INFINITY : hi + log1pf(expf(lo - hi));
    }
    if (OP == NM_EW_ADD_SCALAR) return a + alpha;
    if (OP == NM_EW_ROWSCALE) return a * b;
    if (OP == NM_EW_DIV) return a / b;
    return 0.0f;
}

template <int OP, bool BIN>
__global__ void ew_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                          float* __restrict__ out, long ldo, long rows, int cols, float alpha, int acc) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const float av = a[r * lda + c];
        const float bv = BIN ? b[r * ldb + (OP == NM_EW_ROWSCALE ? 0 : c)] : 0.0f;
        float v = ew_apply<OP>(av, bv, alpha);
        if (acc) v += out[r * ldo + c];
        out[r * ldo + c] = v;
    }
}

// contiguous, 16-byte aligned fast path
template <int OP, bool BIN>
__global__ void ew_kernel_vec(const float4* __restrict__ a, const float4* __restrict__ b,
                              float4* __restrict__ out, long n4, float alpha, int acc) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 av = a[i];
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BIN) bv = b[i];
        float4 v;
        v.x = ew_apply<OP>(av.x, bv.x, alpha);
        v.y = ew_apply<OP>(av.y, bv.y, alpha);
        v.z = ew_apply<OP>(av.z, bv.z, alpha);
        v.w = ew_apply<OP>(av.w, bv.w, alpha);
        if (acc) {
            const float4 o = out[i];
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        out[i] = v;
    }
}

template <int OP, bool BIN>
static void ew_launch(hipStream_t st, const float* a, long lda, const float* b, long ldb, float* out, long ldo,
                      long rows, long cols, float alpha, int acc) {
    const long total = rows * cols;
    const bool contiguous = OP != NM_EW_ROWSCALE &&
                            ((rows == 1) || (lda == cols && ldo == cols && (!BIN || ldb == cols)));
    if (contiguous && total % 4 == 0 && nm_aligned16(a) && nm_aligned16(out) && (!BIN || nm_aligned16(b))) {
        const long n4 = total / 4;
        const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
        hipLaunchKernelGGL((ew_kernel_vec<OP, BIN>), dim3(blocks), dim3(256), 0, st, (const float4*)a,
                           (const float4*)b, (float4*)out, n4, alpha, acc);
        return;
    }
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((ew_kernel<OP, BIN>), dim3(blocks), dim3(256), 0, st, a, lda, b, ldb, out, ldo, rows,
                       (int)cols, alpha, acc);
}

extern "C" int nm_ew(void* stream, int op, const float* a, int64_t lda, const float* b, int64_t ldb, float* out,
                     int64_t ldo, int64_t rows, int64_t cols, float alpha, int accumulate) {
    NM_REQUIRE(op >= 0 && op < NM_EW_OPS, "nm_ew: unknown op %d", op);
    NM_REQUIRE(a && out && rows >= 0 && cols >= 0 && cols < (1LL << 31), "nm_ew: bad args");
    const bool binary = op == NM_EW_ADD || op == NM_EW_SUB || op == NM_EW_MUL || op == NM_EW_ROWSCALE || op == NM_EW_DIV ||
                        (op >= NM_EW_SIGMOID_BWD && op <= NM_EW_LOGADDEXP);
    NM_REQUIRE(!binary || b, "nm_ew: op %d needs a second operand", op);
    if (rows == 0 || cols == 0) return NM_OK;
    hipStream_t st = nm_stream(stream);
#define NM_EW_CASE(OP, BIN) \
    case OP: ew_launch<OP, BIN>(st, a, lda, b, ldb, out, ldo, rows, cols, alpha, accumulate); break;
    switch (op) {
        NM_EW_CASE(NM_EW_COPY, false)
        NM_EW_CASE(NM_EW_ADD, true)
        NM_EW_CASE(NM_EW_SUB, true)
        NM_EW_CASE(NM_EW_MUL, true)
        NM_EW_CASE(NM_EW_SCALE, false)
        NM_EW_CASE(NM_EW_SIGMOID, false)
        NM_EW_CASE(NM_EW_TANH, false)
        NM_EW_CASE(NM_EW_RELU, false)
        NM_EW_CASE(NM_EW_SIGMOID_BWD, true)
        NM_EW_CASE(NM_EW_TANH_BWD, true)
        NM_EW_CASE(NM_EW_RELU_BWD, true)
        NM_EW_CASE(NM_EW_LOGADDEXP, true)
        NM_EW_CASE(NM_EW_ADD_SCALAR, false)
        NM_EW_CASE(NM_EW_ROWSCALE, true)
        NM_EW_CASE(NM_EW_DIV, true)
        default: break;
    }
#undef NM_EW_CASE
    NM_LAUNCH_CHECK("nm_ew");
}

// ---------------------------------------------------------------------------------------------
// h' = u*h + (1-u)*c   (TF GRUCell / NematusGRUCell last line, nn/ortho_gru_cell.py:104)
// ---------------------------------------------------------------------------------------------
__global__ void blend_fwd_kernel(const float* __restrict__ u, long ldu, const float* __restrict__ h, long ldh,
                                 const float* __restrict__ c, long ldc, float* __restrict__ out, long ldo,
                                 long rows, int cols) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int k = (int)(i - r * cols);
        const float uu = u[r * ldu + k];
        out[r * ldo + k] = uu * h[r * ldh + k] + (1.0f - uu) * c[r * ldc + k];
    }
}

// du += dy*(h-c) ; dh += dy*u ; dc += dy*(1-u)
__global__ void blend_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ u, long ldu,
                                 const float* __restrict__ h, long ldh, const float* __restrict__ c, long ldc,
                                 float* __restrict__ du, long lddu, float* __restrict__ dh, long lddh,
                                 float* __restrict__ dc, long lddc, long rows, int cols) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int k = (int)(i - r * cols);
        const float g = dy[r * lddy + k];
        const float uu = u[r * ldu + k];
        if (du) du[r * lddu + k] += g * (h[r * ldh + k] - c[r * ldc + k]);
        if (dh) dh[r * lddh + k] += g * uu;
        if (dc) dc[r * lddc + k] += g * (1.0f - uu);
    }
}

static inline int ew_blocks(long total) { return (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096); }

extern "C" int nm_blend_fwd(void* stream, const float* u, int64_t ldu, const float* h, int64_t ldh,
                            const float* c, int64_t ldc, float* out, int64_t ldo, int64_t rows, int64_t cols) {
    NM_REQUIRE(u && h && c && out && rows >= 0 && cols >= 0, "nm_blend_fwd: bad args");
    if (rows * cols == 0) return NM_OK;
    hipLaunchKernelGGL(blend_fwd_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0, nm_stream(stream), u, ldu, h,
                       ldh, c, ldc, out, ldo, rows, (int)cols);
    NM_LAUNCH_CHECK("nm_blend_fwd");
}

extern "C" int nm_blend_bwd(void* stream, const float* dy, int64_t lddy, const float* u, int64_t ldu,
                            const float* h, int64_t ldh, const float* c, int64_t ldc, float* du, int64_t lddu,
                            float* dh, int64_t lddh, float* dc, int64_t lddc, int64_t rows, int64_t cols) {
    NM_REQUIRE(dy && u && h && c && rows >= 0 && cols >= 0, "nm_blend_bwd: bad args");
    if (rows * cols == 0) return NM_OK;
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0, nm_stream(stream), dy, lddy,
                       u, ldu, h, ldh, c, ldc, du, lddu, dh, lddh, dc, lddc, rows, (int)cols);
    NM_LAUNCH_CHECK("nm_blend_bwd");
}

// ---------------------------------------------------------------------------------------------
// Dropout (nn/utils.py:6-22 -> tf.nn.dropout): keep element i iff floor(keep_prob + u_i) == 1 and
// scale the kept ones by 1/keep_prob.  TF draws u from its Philox stream, which no other
// implementation can replay; here u_i is a counter-based hash of (salt, i) so that the CPU oracle
// restates the identical mask and the backward pass regenerates it instead of storing it.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t nm_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}

// ``step`` (optional device scalar: the optimizer's global step) advances the salt on the device,
// salt_eff = salt + step * 0x9E3779B9, so that a training step captured once into a HIP graph draws
// fresh masks at every replay.
// ---------------------------------------------------------------------------
// LSTMCell point-wise part (tf.nn.rnn_cell.LSTMCell without peepholes / projection: decoders/decoder.py:29,309-325,
// encoders/recurrent.py:21): z = [x, h].W + b comes from the caller's products, gate order i, j, f, o;
//   c' = sigmoid(f + forget_bias) c + sigmoid(i) tanh(j) ;  h' = sigmoid(o) tanh(c')
// One launch instead of four activations + four element-wise products; the activated gates are kept for the backward.
// ---------------------------------------------------------------------------
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ z, long ldz, const float* __restrict__ c_prev, long ldc,
                                     float* __restrict__ c_new, long ldcn, float* __restrict__ h_new, long ldh,
                                     float* __restrict__ gates, long ldg, long rows, int H, float forget_bias) {
    const long total = rows * H;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / H;
        const int k = (int)(idx - r * H);
        const float* zr = z + r * ldz;
        const float gi = nm_sigmoid(zr[k]);
        const float gj = nm_tanh(zr[H + k]);
        const float gf = nm_sigmoid(zr[2 * H + k] + forget_bias);
        const float go = nm_sigmoid(zr[3 * H + k]);
        const float c = gf * c_prev[r * ldc + k] + gi * gj;
        c_new[r * ldcn + k] = c;
        h_new[r * ldh + k] = go * nm_tanh(c);
        if (gates) {
            float* gr = gates + r * ldg;
            gr[k] = gi; gr[H + k] = gj; gr[2 * H + k] = gf; gr[3 * H + k] = go;
        }
    }
}

// dz = [di i (1-i), dj (1-j^2), df f (1-f), do o (1-o)] with tc = tanh(c'), do = dh tc, dc = dc' + dh o (1 - tc^2),
// di = dc j, dj = dc i, df = dc c, dc_prev (+)= dc f
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dh, long lddh, const float* __restrict__ dc_new, long lddc,
                                     const float* __restrict__ gates, long ldg, const float* __restrict__ c_prev,
                                     long ldc, const float* __restrict__ c_new, long ldcn, float* __restrict__ dz,
                                     long lddz, float* __restrict__ dc_prev, long lddcp, long rows, int H, int acc_dz,
                                     int acc_dcp) {
    const long total = rows * H;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / H;
        const int k = (int)(idx - r * H);
        const float* gr = gates + r * ldg;
        const float gi = gr[k], gj = gr[H + k], gf = gr[2 * H + k], go = gr[3 * H + k];
        const float tc = nm_tanh(c_new[r * ldcn + k]);
        const float g_h = dh ? dh[r * lddh + k] : 0.0f;
        const float dc = (dc_new ? dc_new[r * lddc + k] : 0.0f) + g_h * go * (1.0f - tc * tc);
        float* dr = dz + r * lddz;
        const float d_i = dc * gj * gi * (1.0f - gi), d_j = dc * gi * (1.0f - gj * gj);
        const float d_f = dc * c_prev[r * ldc + k] * gf * (1.0f - gf), d_o = g_h * tc * go * (1.0f - go);
        if (acc_dz) { dr[k] += d_i; dr[H + k] += d_j; dr[2 * H + k] += d_f; dr[3 * H + k] += d_o; }
        else { dr[k] = d_i; dr[H + k] = d_j; dr[2 * H + k] = d_f; dr[3 * H + k] = d_o; }
        if (dc_prev) {
            if (acc_dcp) dc_prev[r * lddcp + k] += dc * gf;
            else dc_prev[r * lddcp + k] = dc * gf;
        }
    }
}

extern "C" int nm_lstm_cell_fwd(void* stream, const float* z, int64_t ldz, const float* c_prev, int64_t ldc,
                                float* c_new, int64_t ldcn, float* h_new, int64_t ldh, float* gates, int64_t ldg,
                                int64_t rows, int64_t H, float forget_bias) {
    NM_REQUIRE(z && c_prev && c_new && h_new, "nm_lstm_cell_fwd: null pointer");
    NM_REQUIRE(rows >= 0 && H > 0 && ldz >= 4 * H && ldc >= H && ldcn >= H && ldh >= H && (!gates || ldg >= 4 * H),
               "nm_lstm_cell_fwd: bad shape rows=%ld H=%ld", (long)rows, (long)H);
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(ew_blocks(rows * H)), dim3(256), 0, nm_stream(stream), z, (long)ldz,
                       c_prev, (long)ldc, c_new, (long)ldcn, h_new, (long)ldh, gates, (long)ldg, (long)rows, (int)H,
                       forget_bias);
    NM_LAUNCH_CHECK("nm_lstm_cell_fwd");
}

extern "C" int nm_lstm_cell_bwd(void* stream, const float* dh, int64_t lddh, const float* dc_new, int64_t lddc,
                                const float* gates, int64_t ldg, const float* c_prev, int64_t ldc, const float* c_new,
                                int64_t ldcn, float* dz, int64_t lddz, float* dc_prev, int64_t lddcp, int64_t rows,
                                int64_t H, int accumulate_dz, int accumulate_dc_prev) {
    NM_REQUIRE(gates && c_prev && c_new && dz, "nm_lstm_cell_bwd: null pointer");
    NM_REQUIRE(rows >= 0 && H > 0 && ldg >= 4 * H && lddz >= 4 * H && ldc >= H && ldcn >= H,
               "nm_lstm_cell_bwd: bad shape rows=%ld H=%ld", (long)rows, (long)H);
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ew_blocks(rows * H)), dim3(256), 0, nm_stream(stream), dh, (long)lddh,
                       dc_new, (long)lddc, gates, (long)ldg, c_prev, (long)ldc, c_new, (long)ldcn, dz, (long)lddz,
                       dc_prev, (long)lddcp, (long)rows, (int)H, accumulate_dz, accumulate_dc_prev);
    NM_LAUNCH_CHECK("nm_lstm_cell_bwd");
}

// The point-wise part of one NematusGRUCell step (nn/ortho_gru_cell.py:73-105) after its four products:
//   g = [r | u] = sigmoid(g_pre),  c = tanh(ci + sc * r),  h' = u h + (1 - u) c
// with g_pre = x.W_g + h.U_g (+ biases) [R, 2H], sc = h.U_c (+ bias), ci = x.W_c (+ bias) -- one launch where the tape
// ran sigmoid, mul, tanh and blend (and on the way back blend', tanh', the two halves of mul' and sigmoid': the general
// path's conditional decoder went through ~1100 element-wise launches of 4.4 us per training step at the headline size).
__global__ void nematus_cell_fwd_kernel(const float* __restrict__ g_pre, long ldg, const float* __restrict__ sc, long ldsc,
                                        const float* __restrict__ ci, long ldci, const float* __restrict__ h_prev, long ldh,
                                        float* __restrict__ h_new, long ldhn, float* __restrict__ ru, float* __restrict__ c_out,
                                        const float* __restrict__ g2, long ldg2, long rows, int H) {
    const long total = rows * H;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r_ = idx / H;
        const int k = (int)(idx - r_ * H);
        float pr = g_pre[r_ * ldg + k], pu = g_pre[r_ * ldg + H + k];
        if (g2) { pr += g2[r_ * ldg2 + k]; pu += g2[r_ * ldg2 + H + k]; }      // state half + input half of the gates
        const float r = nm_sigmoid(pr);
        const float u = nm_sigmoid(pu);
        const float c = nm_tanh(sc[r_ * ldsc + k] * r + ci[r_ * ldci + k]);
        h_new[r_ * ldhn + k] = u * h_prev[r_ * ldh + k] + (1.0f - u) * c;
        if (ru) { ru[r_ * 2 * H + k] = r; ru[r_ * 2 * H + H + k] = u; }
        if (c_out) c_out[r_ * H + k] = c;
    }
}

// dc' = dh (1-u)(1-c^2);  du' = dh (h - c) u (1-u);  dsc = dc' r;  dr' = dc' sc r (1-r);  dh_prev = dh u
__global__ void nematus_cell_bwd_kernel(const float* __restrict__ dh, long lddh, const float* __restrict__ ru,
                                        const float* __restrict__ c, const float* __restrict__ sc, long ldsc,
                                        const float* __restrict__ h_prev, long ldh, float* __restrict__ dg, long lddg,
                                        float* __restrict__ dci, long lddci, float* __restrict__ dsc, long lddsc,
                                        float* __restrict__ dhp, long lddhp, float* __restrict__ dg2, long lddg2, long rows,
                                        int H, int acc_dg, int acc_dci, int acc_dsc, int acc_dhp) {
    const long total = rows * H;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r_ = idx / H;
        const int k = (int)(idx - r_ * H);
        const float r = ru[r_ * 2 * H + k], u = ru[r_ * 2 * H + H + k], cv = c[r_ * H + k];
        const float d = dh[r_ * lddh + k];
        const float dcp = d * (1.0f - u) * (1.0f - cv * cv);
        const float dup = d * (h_prev[r_ * ldh + k] - cv) * u * (1.0f - u);
        const float s = sc[r_ * ldsc + k];
        const float drp = dcp * s * r * (1.0f - r);
        float* g = dg + r_ * lddg;
        if (acc_dg) { g[k] += drp; g[H + k] += dup; } else { g[k] = drp; g[H + k] = dup; }
        if (dg2) { dg2[r_ * lddg2 + k] = drp; dg2[r_ * lddg2 + H + k] = dup; }       // the same gradient for the second product
        if (dci) { float* p = dci + r_ * lddci + k; *p = acc_dci ? *p + dcp : dcp; }
        if (dsc) { float* p = dsc + r_ * lddsc + k; *p = acc_dsc ? *p + dcp * r : dcp * r; }
        if (dhp) { float* p = dhp + r_ * lddhp + k; *p = acc_dhp ? *p + d * u : d * u; }
    }
}

extern "C" int nm_nematus_cell_fwd(void* stream, const float* g_pre, int64_t ldg, const float* sc, int64_t ldsc,
                                   const float* ci, int64_t ldci, const float* h_prev, int64_t ldh, float* h_new,
                                   int64_t ldhn, float* ru, float* c_out, const float* g2, int64_t ldg2, int64_t rows,
                                   int64_t H) {
    NM_REQUIRE(g_pre && sc && ci && h_prev && h_new, "nm_nematus_cell_fwd: null pointer");
    NM_REQUIRE(rows >= 0 && H > 0 && ldg >= 2 * H && ldsc >= H && ldci >= H && ldh >= H && ldhn >= H && (!g2 || ldg2 >= 2 * H),
               "nm_nematus_cell_fwd: bad shape rows=%ld H=%ld", (long)rows, (long)H);
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(nematus_cell_fwd_kernel, dim3(ew_blocks(rows * H)), dim3(256), 0, nm_stream(stream), g_pre, (long)ldg,
                       sc, (long)ldsc, ci, (long)ldci, h_prev, (long)ldh, h_new, (long)ldhn, ru, c_out, g2, (long)ldg2,
                       (long)rows, (int)H);
    NM_LAUNCH_CHECK("nm_nematus_cell_fwd");
}

extern "C" int nm_nematus_cell_bwd(void* stream, const float* dh, int64_t lddh, const float* ru, const float* c,
                                   const float* sc, int64_t ldsc, const float* h_prev, int64_t ldh, float* dg,
                                   int64_t lddg, float* dci, int64_t lddci, float* dsc, int64_t lddsc, float* dh_prev,
                                   int64_t lddhp, float* dg2, int64_t lddg2, int64_t rows, int64_t H, int accumulate_dg,
                                   int accumulate_dci, int accumulate_dsc, int accumulate_dh_prev) {
    NM_REQUIRE(dh && ru && c && sc && h_prev && dg, "nm_nematus_cell_bwd: null pointer");
    NM_REQUIRE(rows >= 0 && H > 0 && lddh >= H && ldsc >= H && ldh >= H && lddg >= 2 * H && (!dci || lddci >= H) &&
                   (!dsc || lddsc >= H) && (!dh_prev || lddhp >= H) && (!dg2 || lddg2 >= 2 * H),
               "nm_nematus_cell_bwd: bad shape rows=%ld H=%ld", (long)rows, (long)H);
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(nematus_cell_bwd_kernel, dim3(ew_blocks(rows * H)), dim3(256), 0, nm_stream(stream), dh, (long)lddh,
                       ru, c, sc, (long)ldsc, h_prev, (long)ldh, dg, (long)lddg, dci, (long)lddci, dsc, (long)lddsc, dh_prev,
                       (long)lddhp, dg2, (long)lddg2, (long)rows, (int)H, accumulate_dg, accumulate_dci, accumulate_dsc,
                       accumulate_dh_prev);
    NM_LAUNCH_CHECK("nm_nematus_cell_bwd");
}

__global__ void dropout_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out, long ldo,
                               long rows, int cols, float keep_prob, float inv_keep, uint32_t salt,
                               const uint32_t* __restrict__ step, int acc) {
    const long total = rows * cols;
    if (step) salt += step[0] * 0x9E3779B9u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const uint32_t bits = nm_mix32((uint32_t)i * 0x9E3779B1u + salt);
        const float uni = (float)(bits >> 8) * (1.0f / 16777216.0f);
        const float keep = (keep_prob + uni >= 1.0f) ? inv_keep : 0.0f;
        float v = x[r * ldx + c] * keep;
        if (acc) v += out[r * ldo + c];
        out[r * ldo + c] = v;
    }
}

extern "C" int nm_dropout(void* stream, const float* x, int64_t ldx, float* out, int64_t ldo, int64_t rows,
                          int64_t cols, float keep_prob, uint32_t salt, const uint32_t* step, int accumulate) {
    NM_REQUIRE(x && out && rows >= 0 && cols >= 0, "nm_dropout: bad args");
    NM_REQUIRE(keep_prob > 0.0f && keep_prob <= 1.0f, "nm_dropout: keep_prob %g outside (0,1]", keep_prob);
    NM_REQUIRE(rows * cols < (1LL << 32), "nm_dropout: more than 2^32 elements in one mask");
    if (rows * cols == 0) return NM_OK;
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0, nm_stream(stream), x, ldx, out,
                       ldo, rows, (int)cols, keep_prob, 1.0f / keep_prob, salt, step, accumulate);
    NM_LAUNCH_CHECK("nm_dropout");
}

// ---------------------------------------------------------------------------------------------
// dynamic_rnn(sequence_length=L) step t: rows with t >= L[b] copy the state through and emit 0.
//   h_out = live ? h_new : h_prev ;  y_out = live ? h_new : 0
// ---------------------------------------------------------------------------------------------
__global__ void rnn_select_fwd_kernel(const float* __restrict__ hnew, long ldn, const float* __restrict__ hprev,
                                      long ldp, const int* __restrict__ lengths, int t, float* __restrict__ hout,
                                      long ldh, float* __restrict__ yout, long ldy, long rows, int cols) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const bool live = !lengths || t < lengths[r];
        const float n = hnew[r * ldn + c];
        hout[r * ldh + c] = live ? n : hprev[r * ldp + c];
        if (yout) yout[r * ldy + c] = live ? n : 0.0f;
    }
}

// dnew += live*(dh + dy) ; dprev += (1-live)*dh
__global__ void rnn_select_bwd_kernel(const float* __restrict__ dh, long lddh, const float* __restrict__ dy,
                                      long lddy, const int* __restrict__ lengths, int t,
                                      float* __restrict__ dnew, long lddn, float* __restrict__ dprev, long lddp,
                                      long rows, int cols) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const bool live = !lengths || t < lengths[r];
        const float g = dh ? dh[r * lddh + c] : 0.0f;
        if (live) {
            dnew[r * lddn + c] += g + (dy ? dy[r * lddy + c] : 0.0f);
        } else if (dprev) {
            dprev[r * lddp + c] += g;
        }
    }
}

extern "C" int nm_rnn_select_fwd(void* stream, const float* h_new, int64_t ld_new, const float* h_prev,
                                 int64_t ld_prev, const int32_t* lengths, int t, float* h_out, int64_t ld_h,
                                 float* y_out, int64_t ld_y, int64_t rows, int64_t cols) {
    NM_REQUIRE(h_new && h_prev && h_out && rows >= 0 && cols >= 0, "nm_rnn_select_fwd: bad args");
    if (rows * cols == 0) return NM_OK;
    hipLaunchKernelGGL(rnn_select_fwd_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0, nm_stream(stream), h_new,
                       ld_new, h_prev, ld_prev, lengths, t, h_out, ld_h, y_out, ld_y, rows, (int)cols);
    NM_LAUNCH_CHECK("nm_rnn_select_fwd");
}

extern "C" int nm_rnn_select_bwd(void* stream, const float* dh, int64_t ld_dh, const float* dy, int64_t ld_dy,
                                 const int32_t* lengths, int t, float* d_new, int64_t ld_dnew, float* d_prev,
                                 int64_t ld_dprev, int64_t rows, int64_t cols) {
    NM_REQUIRE(d_new && rows >= 0 && cols >= 0, "nm_rnn_select_bwd: bad args");
    if (rows * cols == 0) return NM_OK;
    hipLaunchKernelGGL(rnn_select_bwd_kernel, dim3(ew_blocks(rows * cols)), dim3(256), 0, nm_stream(stream), dh,
                       ld_dh, dy, ld_dy, lengths, t, d_new, ld_dnew, d_prev, ld_dprev, rows, (int)cols);
    NM_LAUNCH_CHECK("nm_rnn_select_bwd");
}

// ---------------------------------------------------------------------------------------------
// tf.reverse_sequence(x [B,S,D], lengths, seq_axis=1): the first L[b] positions are reversed, the
// rest copied.  The op is its own inverse, so its backward is the same call with accumulate=1.
// ---------------------------------------------------------------------------------------------
__global__ void reverse_sequence_kernel(const float* __restrict__ x, float* __restrict__ out,
                                        const int* __restrict__ lengths, int S, int D, long total, int acc) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const long bs = i / D;
        const int s = (int)(bs % S);
        const long b = bs / S;
        const int len = lengths[b] < S ? lengths[b] : S;
        const int src = s < len ? len - 1 - s : s;
        float v = x[(b * S + src) * D + d];
        if (acc) v += out[i];
        out[i] = v;
    }
}

extern "C" int nm_reverse_sequence(void* stream, const float* x, float* out, const int32_t* lengths, int64_t B,
                                   int64_t S, int64_t D, int accumulate) {
    NM_REQUIRE(x && out && lengths && x != out && B >= 0 && S >= 0 && D >= 0, "nm_reverse_sequence: bad args");
    NM_REQUIRE(S < (1LL << 31) && D < (1LL << 31), "nm_reverse_sequence: shape too large");
    const long total = B * S * D;
    if (total == 0) return NM_OK;
    hipLaunchKernelGGL(reverse_sequence_kernel, dim3(ew_blocks(total)), dim3(256), 0, nm_stream(stream), x, out,
                       lengths, (int)S, (int)D, total, accumulate);
    NM_LAUNCH_CHECK("nm_reverse_sequence");
}

// ---------------------------------------------------------------------------------------------
// maxout (nn/projection.py:7-35): the dense output x [R, P*G] is reshaped to [R,1,P,G] and
// max-pooled over P, i.e. out[r,g] = max_p x[r, p*G + g] (pool members are G apart, not
// adjacent; the reference uses P = 2).  The backward routes the gradient to the first maximal
// element, as MaxPoolGrad does.
// ---------------------------------------------------------------------------------------------
__global__ void maxout_fwd_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out, long ldo,
                                  int* __restrict__ arg, long rows, int groups, int pool) {
    const long total = rows * groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / groups;
        const int g = (int)(i - r * groups);
        const float* p = x + r * ldx + g;
        float best = p[0];
        int bi = 0;
        for (int k = 1; k < pool; ++k)
            if (p[(long)k * groups] > best) { best = p[(long)k * groups]; bi = k; }
        out[r * ldo + g] = best;
        if (arg) arg[i] = bi;
    }
}

__global__ void maxout_bwd_kernel(const float* __restrict__ dy, long lddy, const int* __restrict__ arg,
                                  float* __restrict__ dx, long lddx, long rows, int groups, int pool) {
    const long total = rows * groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / groups;
        const int g = (int)(i - r * groups);
        dx[r * lddx + (long)arg[i] * groups + g] += dy[r * lddy + g];
    }
}

extern "C" int nm_maxout_fwd(void* stream, const float* x, int64_t ldx, float* out, int64_t ldo, int32_t* argmax,
                             int64_t rows, int64_t groups, int64_t pool) {
    NM_REQUIRE(x && out && rows >= 0 && groups > 0 && pool > 0, "nm_maxout_fwd: bad args");
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(maxout_fwd_kernel, dim3(ew_blocks(rows * groups)), dim3(256), 0, nm_stream(stream), x, ldx,
                       out, ldo, argmax, rows, (int)groups, (int)pool);
    NM_LAUNCH_CHECK("nm_maxout_fwd");
}

extern "C" int nm_maxout_bwd(void* stream, const float* dy, int64_t lddy, const int32_t* argmax, float* dx,
                             int64_t lddx, int64_t rows, int64_t groups, int64_t pool) {
    NM_REQUIRE(dy && argmax && dx && rows >= 0 && groups > 0 && pool > 0, "nm_maxout_bwd: bad args");
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(maxout_bwd_kernel, dim3(ew_blocks(rows * groups)), dim3(256), 0, nm_stream(stream), dy,
                       lddy, argmax, dx, lddx, rows, (int)groups, (int)pool);
    NM_LAUNCH_CHECK("nm_maxout_bwd");
}
