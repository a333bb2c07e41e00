// Bandwidth-bound row kernels of the attention-decoder path (forward):
// embedding gather, GRU gate/blend epilogues, layer norm, small utilities.
// Reference call sites:
//   model/sequence.py:170-194            embedding lookup * mask
//   decoders/autoregressive.py:269-272   decoder embedding lookup
//   nn/ortho_gru_cell.py:44-53           TF GRUCell gates / candidate / blend
//   encoders/recurrent.py:86-102         dynamic_rnn length masking + reverse_sequence
//   tf_utils.py:189-219                  layer_norm
#include "nm_common.h"

thread_local char nm_err_buf[512] = {0};
extern "C" const char* nm_last_error(void) { return nm_err_buf; }
extern "C" int nm_version(void) { return 1; }

// ---------------------------------------------------------------------------
// embedding gather: out[i,:] = table[ids[i],:] * scale * (mask_pad ? ids[i]!=0 : 1)
// ---------------------------------------------------------------------------
__global__ void embedding_gather_kernel(const float* __restrict__ table, long V, int E,
                                        const int* __restrict__ ids, long n,
                                        float* __restrict__ out, long ldo, int mask_pad, float scale,
                                        int vec) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    int id = ids[row];
    float s = scale;
    if (mask_pad && id == 0) s = 0.0f;
    if (id < 0 || id >= V) { id = 0; s = 0.0f; }
    const float* src = table + (long)id * E;
    float* dst = out + row * ldo;
    if (vec) {
        for (int c = lane * 4; c < E; c += 256) {
            float4 v = *reinterpret_cast<const float4*>(src + c);
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
            *reinterpret_cast<float4*>(dst + c) = v;
        }
    } else {
        for (int c = lane; c < E; c += 64) dst[c] = src[c] * s;
    }
}

extern "C" int nm_embedding_gather(void* stream, const float* table, int64_t V, int64_t E,
                                   const int32_t* ids, int64_t n, float* out, int64_t ldo,
                                   int mask_pad, float scale) {
    NM_REQUIRE(table && ids && out, "nm_embedding_gather: null pointer");
    NM_REQUIRE(V > 0 && E > 0 && n >= 0 && ldo >= E, "nm_embedding_gather: bad shape");
    if (n == 0) return NM_OK;
    // float4 rows when everything is 16-byte aligned; factored inputs (embedding sizes 5 + 3 side by side,
    // tests/factored.ini) take the scalar path
    const int vec = (E % 4 == 0 && ldo % 4 == 0 && nm_aligned16(table) && nm_aligned16(out)) ? 1 : 0;
    hipLaunchKernelGGL(embedding_gather_kernel, dim3(nm_cdiv(n, 4)), dim3(256), 0, nm_stream(stream),
                       table, (long)V, (int)E, ids, (long)n, out, (long)ldo, mask_pad, scale, vec);
    NM_LAUNCH_CHECK("nm_embedding_gather");
}

// ---------------------------------------------------------------------------
// GRU step epilogues.  The cell is split so both GEMMs run on MFMA:
//   xp  = x . [Wg_x | Wc_x] + [bg | bc]       (hoisted over all time steps)
//   hg  = h . Wg_h                  -> gates:  r,u = sigmoid(xp[:, :2H] + hg), rh = r*h
//   hc  = (r*h) . Wc_h              -> blend:  c = tanh(xp[:, 2H:] + hc), h' = u*h + (1-u)*c
// xp is addressed as  xp + d*x_dir_off + r*x_row_stride + pos*x_time_stride  where
// pos = t, or for the backward direction of a length-masked encoder L[r]-1-t
// (tf.reverse_sequence semantics).  Rows with t >= L[r] are dead: state copied
// through, outputs left untouched (pre-zeroed by the caller).
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool gru_pos(const int* lengths, int r, int d, int t, int rev_mask, int& pos) {
    pos = t;
    if (!lengths) return true;
    const int len = lengths[r];
    if (t >= len) return false;
    if ((rev_mask >> d) & 1) pos = len - 1 - t;      // tf.reverse_sequence indexing
    return true;
}

__global__ void gru_gates_fwd_kernel(const float* __restrict__ xp, long x_dir_off, long x_row_stride,
                                     long x_time_stride, const float* __restrict__ hg,
                                     const float* __restrict__ h, float* __restrict__ ru,
                                     float* __restrict__ rh, const int* __restrict__ lengths, int t,
                                     int rev_mask, long R, int H) {
    const int d = blockIdx.z;
    const long r = blockIdx.y;
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (j >= H) return;
    int pos;
    const bool live = gru_pos(lengths, (int)r, d, t, rev_mask, pos);
    const long ro = ((long)d * R + r);
    float4 rr = make_float4(0, 0, 0, 0), uu = rr, rhv = rr;
    if (live) {
        const float* x = xp + d * x_dir_off + r * x_row_stride + (long)pos * x_time_stride;
        const float4 xr = *reinterpret_cast<const float4*>(x + j);
        const float4 xu = *reinterpret_cast<const float4*>(x + H + j);
        const float4 gr = *reinterpret_cast<const float4*>(hg + ro * 2 * H + j);
        const float4 gu = *reinterpret_cast<const float4*>(hg + ro * 2 * H + H + j);
        const float4 hv = *reinterpret_cast<const float4*>(h + ro * H + j);
        rr = make_float4(nm_sigmoid(xr.x + gr.x), nm_sigmoid(xr.y + gr.y), nm_sigmoid(xr.z + gr.z),
                         nm_sigmoid(xr.w + gr.w));
        uu = make_float4(nm_sigmoid(xu.x + gu.x), nm_sigmoid(xu.y + gu.y), nm_sigmoid(xu.z + gu.z),
                         nm_sigmoid(xu.w + gu.w));
        rhv = make_float4(rr.x * hv.x, rr.y * hv.y, rr.z * hv.z, rr.w * hv.w);
    }
    *reinterpret_cast<float4*>(ru + ro * 2 * H + j) = rr;
    *reinterpret_cast<float4*>(ru + ro * 2 * H + H + j) = uu;
    *reinterpret_cast<float4*>(rh + ro * H + j) = rhv;
}

extern "C" int nm_gru_gates_fwd(void* stream, const float* xp, int64_t x_dir_off, int64_t x_row_stride,
                                int64_t x_time_stride, const float* hg, const float* h, float* ru,
                                float* rh, const int32_t* lengths, int t, int rev_mask, int ndir,
                                int64_t R, int64_t H) {
    NM_REQUIRE(xp && hg && h && ru && rh, "nm_gru_gates_fwd: null pointer");
    NM_REQUIRE(H > 0 && H % 4 == 0 && R > 0 && ndir >= 1 && ndir <= 2, "nm_gru_gates_fwd: bad shape");
    NM_REQUIRE(x_dir_off % 4 == 0 && x_row_stride % 4 == 0 && x_time_stride % 4 == 0 &&
                   nm_aligned16(xp) && nm_aligned16(hg) && nm_aligned16(h) && nm_aligned16(ru) &&
                   nm_aligned16(rh),
               "nm_gru_gates_fwd: unaligned");
    const int tpb = 128;
    dim3 grid(nm_cdiv(H, 4 * tpb), (unsigned)R, ndir);
    hipLaunchKernelGGL(gru_gates_fwd_kernel, grid, dim3(tpb), 0, nm_stream(stream), xp, (long)x_dir_off,
                       (long)x_row_stride, (long)x_time_stride, hg, h, ru, rh, lengths, t, rev_mask, (long)R,
                       (int)H);
    NM_LAUNCH_CHECK("nm_gru_gates_fwd");
}

__global__ void gru_blend_fwd_kernel(const float* __restrict__ xp, long x_dir_off, long x_row_stride,
                                     long x_time_stride, const float* __restrict__ hc,
                                     const float* __restrict__ ru, const float* h_in,
                                     float* h_out, float* __restrict__ c_save, float* __restrict__ out,
                                     long out_dir_off, long out_row_stride, long out_time_stride,
                                     const int* __restrict__ lengths, int t, int rev_mask, long R,
                                     int H) {
    const int d = blockIdx.z;
    const long r = blockIdx.y;
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (j >= H) return;
    int pos;
    const bool live = gru_pos(lengths, (int)r, d, t, rev_mask, pos);
    const long ro = ((long)d * R + r);
    const float4 hv = *reinterpret_cast<const float4*>(h_in + ro * H + j);
    if (!live) {
        if (h_out != h_in) *reinterpret_cast<float4*>(h_out + ro * H + j) = hv;
        if (c_save) *reinterpret_cast<float4*>(c_save + ro * H + j) = make_float4(0, 0, 0, 0);
        return;
    }
    const float* x = xp + d * x_dir_off + r * x_row_stride + (long)pos * x_time_stride + 2 * H;
    const float4 xc = *reinterpret_cast<const float4*>(x + j);
    const float4 cc = *reinterpret_cast<const float4*>(hc + ro * H + j);
    const float4 uu = *reinterpret_cast<const float4*>(ru + ro * 2 * H + H + j);
    float4 c = make_float4(nm_tanh(xc.x + cc.x), nm_tanh(xc.y + cc.y), nm_tanh(xc.z + cc.z),
                           nm_tanh(xc.w + cc.w));
    float4 hn = make_float4(uu.x * hv.x + (1.0f - uu.x) * c.x, uu.y * hv.y + (1.0f - uu.y) * c.y,
                            uu.z * hv.z + (1.0f - uu.z) * c.z, uu.w * hv.w + (1.0f - uu.w) * c.w);
    *reinterpret_cast<float4*>(h_out + ro * H + j) = hn;
    if (c_save) *reinterpret_cast<float4*>(c_save + ro * H + j) = c;
    if (out) {
        float* o = out + d * out_dir_off + r * out_row_stride + (long)pos * out_time_stride;
        *reinterpret_cast<float4*>(o + j) = hn;
    }
}

extern "C" int nm_gru_blend_fwd(void* stream, const float* xp, int64_t x_dir_off, int64_t x_row_stride,
                                int64_t x_time_stride, const float* hc, const float* ru,
                                const float* h_in, float* h_out, float* c_save, float* out,
                                int64_t out_dir_off, int64_t out_row_stride, int64_t out_time_stride,
                                const int32_t* lengths, int t, int rev_mask, int ndir, int64_t R,
                                int64_t H) {
    NM_REQUIRE(xp && hc && ru && h_in && h_out, "nm_gru_blend_fwd: null pointer");
    NM_REQUIRE(H > 0 && H % 4 == 0 && R > 0 && ndir >= 1 && ndir <= 2, "nm_gru_blend_fwd: bad shape");
    NM_REQUIRE(x_dir_off % 4 == 0 && x_row_stride % 4 == 0 && x_time_stride % 4 == 0 &&
                   out_dir_off % 4 == 0 && out_row_stride % 4 == 0 && out_time_stride % 4 == 0,
               "nm_gru_blend_fwd: strides must be multiples of 4");
    const int tpb = 128;
    dim3 grid(nm_cdiv(H, 4 * tpb), (unsigned)R, ndir);
    hipLaunchKernelGGL(gru_blend_fwd_kernel, grid, dim3(tpb), 0, nm_stream(stream), xp, (long)x_dir_off,
                       (long)x_row_stride, (long)x_time_stride, hc, ru, h_in, h_out, c_save, out,
                       (long)out_dir_off, (long)out_row_stride, (long)out_time_stride, lengths, t,
                       rev_mask, (long)R, (int)H);
    NM_LAUNCH_CHECK("nm_gru_blend_fwd");
}

// ---------------------------------------------------------------------------
// layer norm forward: y = (x-mean)*rsqrt(var+eps)*gamma+beta, biased variance.
// One 256-thread block per row; saves mean / rstd for the backward pass.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = nm_wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void layer_norm_fwd_kernel(const float* __restrict__ x, long ldx,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ y, long ldy,
                                                             float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out, int D,
                                                             float eps) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    const float* xr = x + row * ldx;
    float s = 0.0f;
    for (int c = threadIdx.x; c < D; c += 256) s += xr[c];
    const float mean = block_sum_256(s, sh) / (float)D;
    float q = 0.0f;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float dlt = xr[c] - mean;
        q += dlt * dlt;
    }
    const float var = block_sum_256(q, sh) / (float)D;
    const float rstd = rsqrtf(var + eps);
    float* yr = y + row * ldy;
    for (int c = threadIdx.x; c < D; c += 256) yr[c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
    if (threadIdx.x == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// Layer norm forward, one WAVE per row (round 6): a row of D <= 2048 floats is NCH float4s per lane, read once, held in
// registers for the mean, the variance (around the mean, as above) and the output; the two sums are DPP wave
// reductions -- no LDS, no barrier; RPI rows of a wave are in flight together.  The workgroup-per-row kernel above
// spends two block reductions on every 2 KB row: 12.6 us for a [6400, 512] operand (2 TB/s), 32 of them per
// Transformer-base training step.  ADD: the row is a + x (a residual connection), written to sum_out as well -- what
// nm_add_layer_norm_fwd does for the decoding steps, here with the statistics the backward pass needs.
template <int NCH, bool ADD>
__global__ __launch_bounds__(256) void layer_norm_fwd_wave_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  float* __restrict__ sum_out, float* __restrict__ y,
                                                                  float* __restrict__ mean_out,
                                                                  float* __restrict__ rstd_out, long rows, int D, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RPI = NCH <= 2 ? 4 : (NCH <= 4 ? 2 : 1);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gm[NCH], bt[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 256 * c + 4 * lane;
        gm[c] = col < D ? *reinterpret_cast<const float4*>(gamma + col) : z4;
        bt[c] = col < D ? *reinterpret_cast<const float4*>(beta + col) : z4;
    }
    const float invd = 1.0f / (float)D;
    const long stride = (long)gridDim.x * 4;
    for (long row0 = (long)blockIdx.x * 4 + wave; row0 < rows; row0 += stride * RPI) {
        float4 v[RPI][NCH];
#pragma unroll
        for (int r = 0; r < RPI; ++r) {
            const long row = row0 + r * stride;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = 256 * c + 4 * lane;
                v[r][c] = z4;
                if (row < rows && col < D) {
                    v[r][c] = *reinterpret_cast<const float4*>(x + row * D + col);
                    if constexpr (ADD) {
                        const float4 w = *reinterpret_cast<const float4*>(a + row * D + col);
                        v[r][c].x += w.x; v[r][c].y += w.y; v[r][c].z += w.z; v[r][c].w += w.w;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPI; ++r) {
            const long row = row0 + r * stride;
            if (row >= rows) break;
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) s += (v[r][c].x + v[r][c].y) + (v[r][c].z + v[r][c].w);
            const float mean = nm_wave_sum_dpp(s) * invd;
            float q = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (256 * c + 4 * lane < D) {
                    const float d0 = v[r][c].x - mean, d1 = v[r][c].y - mean, d2 = v[r][c].z - mean, d3 = v[r][c].w - mean;
                    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
            const float rstd = rsqrtf(nm_wave_sum_dpp(q) * invd + eps);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = 256 * c + 4 * lane;
                if (col < D) {
                    if constexpr (ADD) *reinterpret_cast<float4*>(sum_out + row * D + col) = v[r][c];
                    float4 o;
                    o.x = (v[r][c].x - mean) * rstd * gm[c].x + bt[c].x;
                    o.y = (v[r][c].y - mean) * rstd * gm[c].y + bt[c].y;
                    o.z = (v[r][c].z - mean) * rstd * gm[c].z + bt[c].z;
                    o.w = (v[r][c].w - mean) * rstd * gm[c].w + bt[c].w;
                    *reinterpret_cast<float4*>(y + row * D + col) = o;
                }
            }
            if (lane == 0) {
                if (mean_out) mean_out[row] = mean;
                if (rstd_out) rstd_out[row] = rstd;
            }
        }
    }
}

// the wave kernel's conditions: contiguous rows of D <= 2048 floats, D % 4 == 0, 16-byte aligned operands
static bool ln_wave_ok(const float* a, const float* x, int64_t lda, int64_t ldx, const float* gamma, const float* beta,
                       const float* sum_out, int64_t lds, const float* y, int64_t ldy, int64_t D, int64_t rows) {
    static const bool on = !(getenv("NM_LN_FWD_WAVE") && atoi(getenv("NM_LN_FWD_WAVE")) == 0);
    // (a decoding step's 128 or 640 rows are a handful of workgroups here and a workgroup each above: measured 2 us per
    // call slower -- 2.6 ms per Transformer-base greedy batch; the wave kernel is for the B x T rows of training)
    return on && rows >= 1024 && D % 4 == 0 && D <= 2048 && ldx == D && ldy == D && (!a || lda == D) && (!sum_out || lds == D) &&
           nm_aligned16(x) && nm_aligned16(y) && nm_aligned16(gamma) && nm_aligned16(beta) && (!a || nm_aligned16(a)) &&
           (!sum_out || nm_aligned16(sum_out));
}

template <bool ADD>
static void ln_wave_launch(hipStream_t st, const float* a, const float* x, const float* gamma, const float* beta,
                           float* sum_out, float* y, float* mean_out, float* rstd_out, int64_t rows, int64_t D, float eps) {
    const int nch = (int)((D + 255) / 256);
    const int rpi = nch <= 2 ? 4 : (nch <= 4 ? 2 : 1);
    long G = (rows + 4L * rpi - 1) / (4L * rpi);
    if (G > 2048) G = 2048;
    if (G < 1) G = 1;
#define NM_LNF(N_) hipLaunchKernelGGL((layer_norm_fwd_wave_kernel<N_, ADD>), dim3((unsigned)G), dim3(256), 0, st, a, x, gamma, \
                                      beta, sum_out, y, mean_out, rstd_out, (long)rows, (int)D, eps)
    if (nch <= 1) NM_LNF(1);
    else if (nch == 2) NM_LNF(2);
    else if (nch <= 4) NM_LNF(4);
    else NM_LNF(8);
#undef NM_LNF
}

// s = a + x ; y = layer_norm(s): a residual connection and the pre-norm of the next sub-layer in one pass
// (decoders/transformer.py:270-358: every sub-layer ends in `+ x` and the next one starts with layer_norm).  The
// arithmetic is ew "add" followed by layer_norm_fwd_kernel's, element for element.
__global__ __launch_bounds__(256) void add_layer_norm_fwd_kernel(const float* __restrict__ a, long lda,
                                                                 const float* __restrict__ x, long ldx,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 float* __restrict__ sum_out, long lds,
                                                                 float* __restrict__ y, long ldy, int D, float eps) {
    __shared__ float sh[4];
    extern __shared__ float srow[];               // [D] the summed row
    const long row = blockIdx.x;
    const float* ar = a + row * lda;
    const float* xr = x + row * ldx;
    float* sr = sum_out + row * lds;
    float s = 0.0f;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float v = ar[c] + xr[c];
        srow[c] = v;
        sr[c] = v;
        s += v;
    }
    const float mean = block_sum_256(s, sh) / (float)D;
    float q = 0.0f;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float dlt = srow[c] - mean;
        q += dlt * dlt;
    }
    const float var = block_sum_256(q, sh) / (float)D;
    const float rstd = rsqrtf(var + eps);
    float* yr = y + row * ldy;
    for (int c = threadIdx.x; c < D; c += 256) yr[c] = (srow[c] - mean) * rstd * gamma[c] + beta[c];
}

extern "C" int nm_add_layer_norm_fwd(void* stream, const float* a, int64_t lda, const float* x, int64_t ldx,
                                     const float* gamma, const float* beta, float* sum_out, int64_t lds, float* y,
                                     int64_t ldy, int64_t rows, int64_t D, float eps) {
    NM_REQUIRE(a && x && gamma && beta && sum_out && y, "nm_add_layer_norm_fwd: null pointer");
    NM_REQUIRE(rows >= 0 && D > 0 && D <= 16384, "nm_add_layer_norm_fwd: bad shape rows=%ld D=%ld", (long)rows, (long)D);
    if (rows == 0) return NM_OK;
    if (ln_wave_ok(a, x, lda, ldx, gamma, beta, sum_out, lds, y, ldy, D, rows)) {
        ln_wave_launch<true>(nm_stream(stream), a, x, gamma, beta, sum_out, y, nullptr, nullptr, rows, D, eps);
        NM_LAUNCH_CHECK("nm_add_layer_norm_fwd");
    }
    hipLaunchKernelGGL(add_layer_norm_fwd_kernel, dim3((unsigned)rows), dim3(256), (size_t)D * sizeof(float),
                       nm_stream(stream), a, (long)lda, x, (long)ldx, gamma, beta, sum_out, (long)lds, y, (long)ldy,
                       (int)D, eps);
    NM_LAUNCH_CHECK("nm_add_layer_norm_fwd");
}

// The same with the row statistics the backward pass reads (training tapes: autodiff.layer_norm on a pending sum).
// Contiguous rows only (the wave kernel's conditions); returns NM_ERR_ARG otherwise -- the caller adds, then norms.
extern "C" int nm_add_layer_norm_stats_fwd(void* stream, const float* a, const float* x, const float* gamma,
                                           const float* beta, float* sum_out, float* y, float* mean_out, float* rstd_out,
                                           int64_t rows, int64_t D, float eps) {
    NM_REQUIRE(a && x && gamma && beta && sum_out && y && mean_out && rstd_out, "nm_add_layer_norm_stats_fwd: null pointer");
    NM_REQUIRE(rows >= 0 && D > 0, "nm_add_layer_norm_stats_fwd: bad shape");
    NM_REQUIRE(ln_wave_ok(a, x, D, D, gamma, beta, sum_out, D, y, D, D, 1 << 20),
               "nm_add_layer_norm_stats_fwd: D = %ld must be a multiple of 4 up to 2048, operands 16-byte aligned", (long)D);
    if (rows == 0) return NM_OK;
    ln_wave_launch<true>(nm_stream(stream), a, x, gamma, beta, sum_out, y, mean_out, rstd_out, rows, D, eps);
    NM_LAUNCH_CHECK("nm_add_layer_norm_stats_fwd");
}

extern "C" int nm_layer_norm_fwd(void* stream, const float* x, int64_t ldx, const float* gamma,
                                 const float* beta, float* y, int64_t ldy, float* mean_out,
                                 float* rstd_out, int64_t rows, int64_t D, float eps) {
    NM_REQUIRE(x && gamma && beta && y, "nm_layer_norm_fwd: null pointer");
    NM_REQUIRE(rows >= 0 && D > 0, "nm_layer_norm_fwd: bad shape");
    if (rows == 0) return NM_OK;
    if (ln_wave_ok(nullptr, x, 0, ldx, gamma, beta, nullptr, 0, y, ldy, D, rows)) {
        ln_wave_launch<false>(nm_stream(stream), nullptr, x, gamma, beta, nullptr, y, mean_out, rstd_out, rows, D, eps);
        NM_LAUNCH_CHECK("nm_layer_norm_fwd");
    }
    hipLaunchKernelGGL(layer_norm_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, nm_stream(stream), x,
                       (long)ldx, gamma, beta, y, (long)ldy, mean_out, rstd_out, (int)D, eps);
    NM_LAUNCH_CHECK("nm_layer_norm_fwd");
}

// ---------------------------------------------------------------------------
// concat helper: dst[r, off:off+w] = src[r, :w]   (builds [h | emb | ctx] rows)
// ---------------------------------------------------------------------------
__global__ void copy_cols_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst,
                                 long ldd, long rows, int w) {
    const long r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < w) dst[r * ldd + c] = src[r * lds_ + c];
}

extern "C" int nm_copy_cols(void* stream, const float* src, int64_t ld_src, float* dst, int64_t ld_dst,
                            int64_t rows, int64_t width) {
    NM_REQUIRE(src && dst && rows >= 0 && width >= 0, "nm_copy_cols: bad args");
    if (rows == 0 || width == 0) return NM_OK;
    hipLaunchKernelGGL(copy_cols_kernel, dim3(nm_cdiv(width, 256), (unsigned)rows), dim3(256), 0,
                       nm_stream(stream), src, (long)ld_src, dst, (long)ld_dst, (long)rows, (int)width);
    NM_LAUNCH_CHECK("nm_copy_cols");
}

// ---------------------------------------------------------------------------
// deterministic sum of n floats -> out[0] (single block, fixed reduction tree)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void reduce_sum_kernel(const float* __restrict__ x, long n,
                                                          float* __restrict__ out) {
    __shared__ float sh[16];
    float s = 0.0f;
    for (long i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = nm_wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int k = 0; k < 16; ++k) t += sh[k];
        out[0] = t;
    }
}

// two-stage variant for long vectors: 64 fixed slices -> library-owned partials -> final sum
__device__ float g_reduce_partials[64];
__global__ __launch_bounds__(1024) void reduce_sum_slices_kernel(const float* __restrict__ x, long n) {
    __shared__ float sh[16];
    const long per = (n + 63) / 64;
    const long beg = blockIdx.x * per, end = min(n, beg + per);
    float s = 0.0f;
    for (long i = beg + threadIdx.x; i < end; i += 1024) s += x[i];
    s = nm_wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int k = 0; k < 16; ++k) t += sh[k];
        g_reduce_partials[blockIdx.x] = t;
    }
}
__global__ void reduce_sum_final_kernel(float* __restrict__ out) {
    float t = 0.0f;
    for (int k = 0; k < 64; ++k) t += g_reduce_partials[k];
    out[0] = t;
}

extern "C" int nm_reduce_sum(void* stream, const float* x, int64_t n, float* out) {
    NM_REQUIRE(x && out && n >= 0, "nm_reduce_sum: bad args");
    if (n > 65536) {
        hipLaunchKernelGGL(reduce_sum_slices_kernel, dim3(64), dim3(1024), 0, nm_stream(stream), x, (long)n);
        hipLaunchKernelGGL(reduce_sum_final_kernel, dim3(1), dim3(1), 0, nm_stream(stream), out);
    } else {
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, nm_stream(stream), x, (long)n, out);
    }
    NM_LAUNCH_CHECK("nm_reduce_sum");
}

// ---------------------------------------------------------------------------
// log-softmax from row statistics: out = (x - max) - lse   (runtime_logprobs,
// decoders/autoregressive.py:373-375; only the ensemble path materialises it)
// ---------------------------------------------------------------------------
__global__ void log_softmax_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ rmax,
                                   const float* __restrict__ rlse, float* __restrict__ out, long ldo,
                                   int V) {
    const long r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < V) out[r * ldo + c] = (x[r * ldx + c] - rmax[r]) - rlse[r];
}

extern "C" int nm_log_softmax(void* stream, const float* x, int64_t ldx, const float* rmax,
                              const float* rlse, float* out, int64_t ldo, int64_t rows, int64_t V) {
    NM_REQUIRE(x && rmax && rlse && out && rows >= 0 && V > 0, "nm_log_softmax: bad args");
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(log_softmax_kernel, dim3(nm_cdiv(V, 256), (unsigned)rows), dim3(256), 0,
                       nm_stream(stream), x, (long)ldx, rmax, rlse, out, (long)ldo, (int)V);
    NM_LAUNCH_CHECK("nm_log_softmax");
}
