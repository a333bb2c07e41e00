// Row kernels over the vocabulary axis: max / argmax / log-sum-exp, masked
// cross entropy (+ its gradient), and the beam-search top-k step.
// Reference call sites:
//   decoders/autoregressive.py:470          tf.argmax(logits, axis=1)  (first max wins)
//   decoders/autoregressive.py:289-316      log_softmax / sequence_loss * mask
//   decoders/beam_search_decoder.py:440-501 masking, +logprob_sum, length penalty,
//                                           tf.nn.top_k over [B, k*V], div/mod, gathers
#include "nm_common.h"
#include <stdlib.h>

#define NM_NEG_INF_F (-1e9f)   // the reference's INF (beam_search_decoder.py:42)

// ---------------------------------------------------------------------------
// row statistics: max, first argmax, lse = log(sum(exp(x - max)))
// two passes over the row (second pass is L2-resident), matching
// tf.nn.log_softmax = x - max - log(sum(exp(x - max))).
// ---------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void block_argmax(float& v, int& i, float* shv, int* shi) {
    // reduce (value desc, index asc) over the block
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(i, off, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { shv[w] = v; shi[w] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bv = shv[0];
        int bi = shi[0];
        for (int k = 1; k < NT / 64; ++k)
            if (shv[k] > bv || (shv[k] == bv && shi[k] < bi)) { bv = shv[k]; bi = shi[k]; }
        shv[0] = bv;
        shi[0] = bi;
    }
    __syncthreads();
    v = shv[0];
    i = shi[0];
    __syncthreads();
}

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = nm_wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float s = 0.0f;
    if (threadIdx.x == 0) {
        for (int k = 0; k < NT / 64; ++k) s += sh[k];
        sh[0] = s;
    }
    __syncthreads();
    s = sh[0];
    __syncthreads();
    return s;
}

template <int NT>
__global__ __launch_bounds__(NT) void row_stats_kernel(const float* __restrict__ x, long ldx, int V,
                                                       float* __restrict__ max_out,
                                                       float* __restrict__ lse_out,
                                                       int* __restrict__ argmax_out) {
    __shared__ float shv[NT / 64];
    __shared__ int shi[NT / 64];
    const long row = blockIdx.x;
    const float* xr = x + row * ldx;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < V; c += NT) {
        const float v = xr[c];
        if (v > bv) { bv = v; bi = c; }     // strided ascending: first max kept
    }
    block_argmax<NT>(bv, bi, shv, shi);
    float s = 0.0f;
    if (lse_out) {
        for (int c = threadIdx.x; c < V; c += NT) s += expf(xr[c] - bv);
        s = block_sum<NT>(s, shv);
    }
    if (threadIdx.x == 0) {
        if (max_out) max_out[row] = bv;
        if (lse_out) lse_out[row] = logf(s);
        if (argmax_out) argmax_out[row] = bi;
    }
}

// ---------------------------------------------------------------------------
// greedy symbol update (autoregressive.py:461-480):
//   sym = argmax * !finished ; finished |= (sym == </s>) ; mask = !finished
// ---------------------------------------------------------------------------
__global__ void greedy_update_kernel(const int* __restrict__ argmax, int* __restrict__ finished,
                                     int* __restrict__ sym_out, int* __restrict__ mask_out, int n,
                                     int end_id, int* __restrict__ all_finished) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int f = finished[i];
    const int s = f ? 0 : argmax[i];
    f = f | (s == end_id);
    finished[i] = f;
    sym_out[i] = s;
    if (mask_out) mask_out[i] = !f;
    if (all_finished && !f) *all_finished = 0;       // (a plain store: every writer writes the same 0 --
                                                       //  640 atomics on one word were ~8 us of a beam step)
}

extern "C" int nm_greedy_update(void* stream, const int32_t* argmax, int32_t* finished, int32_t* sym_out,
                                int32_t* mask_out, int64_t n, int end_id, int32_t* all_finished) {
    NM_REQUIRE(argmax && finished && sym_out && n >= 0, "nm_greedy_update: bad args");
    if (n == 0) return NM_OK;
    hipLaunchKernelGGL(greedy_update_kernel, dim3(nm_cdiv(n, 256)), dim3(256), 0, nm_stream(stream),
                       argmax, finished, sym_out, mask_out, (int)n, end_id, all_finished);
    NM_LAUNCH_CHECK("nm_greedy_update");
}

// ---------------------------------------------------------------------------
// masked cross entropy over [rows, V] logits (+ optional in-place gradient):
//   loss[r] = -(x[t] - max - lse) * w[r]
//   dx      = (softmax(x) - onehot(t)) * w[r] * grad_scale        (if write_grad)
// ---------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void xent_kernel(float* __restrict__ x, long ldx, int V,
                                                  const int* __restrict__ targets,
                                                  const float* __restrict__ weights,
                                                  float* __restrict__ loss_rows,
                                                  const float* __restrict__ grad_scale,
                                                  int write_grad, float smoothing) {
    __shared__ float shm[NT / 64], shs[NT / 64], shx[NT / 64];
    const long row = blockIdx.x;
    float* xr = x + row * ldx;
    // one pass over the row: running (max, sum exp(x - max)) per thread, merged pairwise;
    // label smoothing also needs the plain sum of the logits
    float bv = -INFINITY, s = 0.0f, sx = 0.0f;
    for (int c = threadIdx.x; c < V; c += NT) {
        const float v = xr[c];
        sx += v;
        if (v > bv) { s = s * expf(bv - v) + 1.0f; bv = v; }
        else s += expf(v - bv);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(bv, off, 64), os = __shfl_xor(s, off, 64);
        const float mm = fmaxf(bv, om);
        s = (bv == -INFINITY ? 0.0f : s * expf(bv - mm)) + (om == -INFINITY ? 0.0f : os * expf(om - mm));
        bv = mm;
        sx += __shfl_xor(sx, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { shm[threadIdx.x >> 6] = bv; shs[threadIdx.x >> 6] = s; shx[threadIdx.x >> 6] = sx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mm = shm[0], ss = shs[0], xx = shx[0];
        for (int k = 1; k < NT / 64; ++k) {
            const float om = shm[k], os = shs[k], nm = fmaxf(mm, om);
            ss = (mm == -INFINITY ? 0.0f : ss * expf(mm - nm)) + (om == -INFINITY ? 0.0f : os * expf(om - nm));
            mm = nm;
            xx += shx[k];
        }
        shm[0] = mm;
        shs[0] = ss;
        shx[0] = xx;
    }
    __syncthreads();
    bv = shm[0];
    s = shs[0];
    sx = shx[0];
    const float lse = logf(s);
    const int t = targets[row];
    const float w = weights ? weights[row] : 1.0f;
    if (threadIdx.x == 0 && loss_rows) {
        // -sum_v q_v log p_v with q = (1-eps)*onehot + eps/V  (tf.losses.softmax_cross_entropy label_smoothing)
        const float nll = (t >= 0 && t < V) ? -(xr[t] - bv - lse) : 0.0f;
        const float uniform = (bv + lse) - sx / (float)V;
        loss_rows[row] = ((1.0f - smoothing) * nll + smoothing * uniform) * w;
    }
    if (write_grad) {
        __syncthreads();   // loss read of xr[t] done before overwrite
        const float gs = w * (grad_scale ? grad_scale[0] : 1.0f);
        const float inv = 1.0f / s;
        const float qu = smoothing / (float)V;
        for (int c = threadIdx.x; c < V; c += NT) {
            float p = expf(xr[c] - bv) * inv - qu;
            if (c == t) p -= 1.0f - smoothing;
            xr[c] = p * gs;
        }
    }
}

// Register-resident variant: the row is read from HBM ONCE (NV float4 per thread, all loads in flight
// together), max / sum exp / sum x are reduced from registers and the gradient is written straight
// from them -- 1 read + 1 write of the [rows, V] logits instead of 2 reads + 1 write.
template <int NV>
__global__ __launch_bounds__(1024) void xent_regs_kernel(float* __restrict__ x, long ldx, int V,
                                                         const int* __restrict__ targets,
                                                         const float* __restrict__ weights,
                                                         float* __restrict__ loss_rows,
                                                         const float* __restrict__ grad_scale,
                                                         int write_grad, float smoothing) {
    __shared__ float sh[16];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    float4* x4 = reinterpret_cast<float4*>(x + row * ldx);
    const int V4 = V >> 2;
    float4 xv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = tid + i * 1024;
        xv[i] = q < V4 ? x4[q] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    float bv = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) bv = fmaxf(bv, fmaxf(fmaxf(xv[i].x, xv[i].y), fmaxf(xv[i].z, xv[i].w)));
    bv = nm_wave_max(bv);
    if ((tid & 63) == 0) sh[tid >> 6] = bv;
    __syncthreads();
    bv = sh[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) bv = fmaxf(bv, sh[w]);
    __syncthreads();
    float s = 0.0f, sx = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (tid + i * 1024 < V4) {
            s += expf(xv[i].x - bv) + expf(xv[i].y - bv) + expf(xv[i].z - bv) + expf(xv[i].w - bv);
            sx += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
        }
    }
    s = block_sum<1024>(s, sh);
    __syncthreads();
    if (smoothing != 0.0f) { sx = block_sum<1024>(sx, sh); __syncthreads(); }
    const float lse = logf(s);
    const int t = targets[row];
    const float w = weights ? weights[row] : 1.0f;
    if (tid == 0 && loss_rows) {
        const float nll = (t >= 0 && t < V) ? -(x[row * ldx + t] - bv - lse) : 0.0f;    // not yet overwritten: see barrier
        const float uniform = (bv + lse) - sx / (float)V;
        loss_rows[row] = ((1.0f - smoothing) * nll + smoothing * uniform) * w;
    }
    if (!write_grad) return;
    __syncthreads();                       // the loss read of x[t] precedes every store of this row
    const float gs = w * (grad_scale ? grad_scale[0] : 1.0f);
    const float inv = 1.0f / s;
    const float qu = smoothing / (float)V;
    const float hot = 1.0f - smoothing;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = tid + i * 1024;
        if (q < V4) {
            const int e = q * 4;
            float4 g;
            g.x = (expf(xv[i].x - bv) * inv - qu - (e == t ? hot : 0.0f)) * gs;
            g.y = (expf(xv[i].y - bv) * inv - qu - (e + 1 == t ? hot : 0.0f)) * gs;
            g.z = (expf(xv[i].z - bv) * inv - qu - (e + 2 == t ? hot : 0.0f)) * gs;
            g.w = (expf(xv[i].w - bv) * inv - qu - (e + 3 == t ? hot : 0.0f)) * gs;
            x4[q] = g;
        }
    }
}

// The same cross entropy + in-place gradient with the COLUMN SUMS of the gradient (the bias gradient of the
// vocabulary projection, decoders/autoregressive.py:450-459 under tf.gradients) as a by-product: a workgroup walks
// rows blockIdx.x, blockIdx.x + G, ... with the same thread -> column mapping for every row and adds what it stores
// to a [V] accumulator in LDS (each thread only ever touches its own entries); the G partial vectors are summed by
// nm_colsum afterwards.  The separate column-sum pass this replaces re-read the 819 MB gradient (0.27 ms at the
// benchmark shape) on the side stream, right under the first launches of the BPTT loop, which waited for it.
// Per-row arithmetic is xent_regs_kernel's.
#define XC_NT 1024
typedef float xc_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 xc_load_nt(const float4* p) {
    const xc_f4 v = __builtin_nontemporal_load(reinterpret_cast<const xc_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int NV>
__global__ __launch_bounds__(XC_NT) void xent_cols_kernel(float* __restrict__ x, long ldx, int V, int rows,
                                                          const int* __restrict__ targets,
                                                          const float* __restrict__ weights,
                                                          float* __restrict__ loss_rows,
                                                          const float* __restrict__ grad_scale, float smoothing,
                                                          float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float4 cs4[];      // [V / 4]
    __shared__ float sh[XC_NT / 64];
    const int tid = threadIdx.x;
    const int V4 = V >> 2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (tid + i * XC_NT < V4) cs4[tid + i * XC_NT] = zero4;
    const float4 ninf4 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    const float gsc = grad_scale ? grad_scale[0] : 1.0f;
    const float qu = smoothing / (float)V;
    const float hot = 1.0f - smoothing;
    const long G = gridDim.x;

    // (non-temporal builtin: a plain conditional float4 load is split into four guarded dword loads)
    auto load_row = [&](float4 (&r)[NV], long row) {
        const float4* r4 = reinterpret_cast<const float4*>(x + row * ldx);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            r[i] = ninf4;
            if (tid + i * XC_NT < V4) r[i] = xc_load_nt(&r4[tid + i * XC_NT]);
        }
    };
    // one row from registers: statistics, loss, gradient stored in place and added to the column accumulator.
    // `xt` = the target's logit, read by the caller before the row is overwritten
    auto process = [&](float4 (&xv)[NV], long row, int t, float xt) {
        float4* x4 = reinterpret_cast<float4*>(x + row * ldx);
        float bv = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) bv = fmaxf(bv, fmaxf(fmaxf(xv[i].x, xv[i].y), fmaxf(xv[i].z, xv[i].w)));
        bv = nm_wave_max(bv);
        if ((tid & 63) == 0) sh[tid >> 6] = bv;
        __syncthreads();
        bv = sh[0];
#pragma unroll
        for (int w = 1; w < XC_NT / 64; ++w) bv = fmaxf(bv, sh[w]);
        __syncthreads();
        float s = 0.0f, sx = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (tid + i * XC_NT < V4) {
                sx += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
                xv[i].x = expf(xv[i].x - bv); xv[i].y = expf(xv[i].y - bv);      // kept for the gradient
                xv[i].z = expf(xv[i].z - bv); xv[i].w = expf(xv[i].w - bv);
                s += xv[i].x + xv[i].y + xv[i].z + xv[i].w;
            }
        }
        s = block_sum<XC_NT>(s, sh);
        __syncthreads();
        if (smoothing != 0.0f) { sx = block_sum<XC_NT>(sx, sh); __syncthreads(); }
        const float lse = logf(s);
        const float w = weights ? weights[row] : 1.0f;
        if (tid == 0 && loss_rows) {
            const float nll = (t >= 0 && t < V) ? -(xt - bv - lse) : 0.0f;
            const float uniform = (bv + lse) - sx / (float)V;
            loss_rows[row] = ((1.0f - smoothing) * nll + smoothing * uniform) * w;
        }
        const float gs = w * gsc;
        const float inv = 1.0f / s;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = tid + i * XC_NT;
            if (q < V4) {
                const int e = q * 4;
                float4 g;
                g.x = (xv[i].x * inv - qu - (e == t ? hot : 0.0f)) * gs;
                g.y = (xv[i].y * inv - qu - (e + 1 == t ? hot : 0.0f)) * gs;
                g.z = (xv[i].z * inv - qu - (e + 2 == t ? hot : 0.0f)) * gs;
                g.w = (xv[i].w * inv - qu - (e + 3 == t ? hot : 0.0f)) * gs;
                x4[q] = g;
                float4 c = cs4[q];
                c.x += g.x; c.y += g.y; c.z += g.z; c.w += g.w;
                cs4[q] = c;
            }
        }
    };
    auto target_logit = [&](long row, int t) {
        return (tid == 0 && t >= 0 && t < V) ? x[row * ldx + t] : 0.0f;
    };

    // (Prefetching the next row into a second register set was tried three ways -- copied sets, ping-pong sets, 512
    // threads x 16 float4: hipcc either waits for the prefetch right behind its issue (vmcnt counts in order across
    // the loop's back edge) or spills 28-101 registers; the plain loop runs at 3.9 TB/s.)
    float4 xv[NV];
    for (long row = blockIdx.x; row < rows; row += G) {
        const int t = targets[row];
        const float xt = target_logit(row, t);
        load_row(xv, row);
        process(xv, row, t, xt);
    }
    float4* out = reinterpret_cast<float4*>(partial + (long)blockIdx.x * V);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (tid + i * XC_NT < V4) out[tid + i * XC_NT] = cs4[tid + i * XC_NT];
}

// partial [partial_rows, V]: row g receives the column sums of logits rows g, g + partial_rows, ...
extern "C" int nm_xent_colsum(void* stream, float* logits, int64_t ldx, int64_t rows, int64_t V,
                              const int32_t* targets, const float* weights, float* loss_rows,
                              const float* grad_scale, float label_smoothing, float* partial, int64_t partial_rows) {
    NM_REQUIRE(logits && targets && partial && rows > 0 && V > 0 && ldx >= V, "nm_xent_colsum: bad args");
    NM_REQUIRE(label_smoothing >= 0.0f && label_smoothing < 1.0f, "nm_xent_colsum: label_smoothing %g outside [0,1)",
               label_smoothing);
    NM_REQUIRE(partial_rows >= 1 && partial_rows <= rows && partial_rows < (1 << 30), "nm_xent_colsum: partial_rows %ld",
               (long)partial_rows);
    NM_REQUIRE(V % 4 == 0 && ldx % 4 == 0 && nm_aligned16(logits) && nm_aligned16(partial) && V / 4 <= 8 * XC_NT &&
                   V * 4 <= 144 * 1024,
               "nm_xent_colsum: V=%ld must be a multiple of 4, at most 32768, rows 16-byte aligned", (long)V);
    const int nv = (int)((V / 4 + XC_NT - 1) / XC_NT);
#define NM_XC(NV_)                                                                                                    \
    do {                                                                                                              \
        static std::atomic<unsigned> attr_devs{0}; /* per device */                                                               \
        const unsigned attr_bit = 1u << (nm_cur()->device & 31);                                                      \
        if (!(attr_devs.load(std::memory_order_relaxed) & attr_bit)) {                                                                                \
            (void)hipFuncSetAttribute((const void*)xent_cols_kernel<NV_>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                      144 * 1024);                                                                    \
            attr_devs.fetch_or(attr_bit, std::memory_order_relaxed);                                                                                    \
        }                                                                                                             \
        hipLaunchKernelGGL((xent_cols_kernel<NV_>), dim3((unsigned)partial_rows), dim3(XC_NT), (size_t)V * 4,           \
                           nm_stream(stream), logits, (long)ldx, (int)V, (int)rows, targets, weights, loss_rows,      \
                           grad_scale, label_smoothing, partial);                                                     \
    } while (0)
    if (nv <= 2) NM_XC(2);
    else if (nv <= 4) NM_XC(4);
    else NM_XC(8);
#undef NM_XC
    NM_LAUNCH_CHECK("nm_xent_colsum");
}

extern "C" int nm_xent(void* stream, float* logits, int64_t ldx, int64_t rows, int64_t V,
                       const int32_t* targets, const float* weights, float* loss_rows,
                       const float* grad_scale, int write_grad, float label_smoothing) {
    NM_REQUIRE(logits && targets && rows >= 0 && V > 0 && ldx >= V, "nm_xent: bad args");
    NM_REQUIRE(label_smoothing >= 0.0f && label_smoothing < 1.0f, "nm_xent: label_smoothing %g outside [0,1)",
               label_smoothing);
    if (rows == 0) return NM_OK;
    hipStream_t st = nm_stream(stream);
    const bool vec = (V % 4 == 0) && (ldx % 4 == 0) && nm_aligned16(logits);
    const int64_t v4 = V / 4;
#define NM_XR(NV_)                                                                                               \
    hipLaunchKernelGGL((xent_regs_kernel<NV_>), dim3((unsigned)rows), dim3(1024), 0, st, logits, (long)ldx, (int)V, \
                       targets, weights, loss_rows, grad_scale, write_grad, label_smoothing)
    if (vec && v4 <= 8 * 1024) NM_XR(8);
    else if (vec && v4 <= 16 * 1024) NM_XR(16);
    else
        hipLaunchKernelGGL((xent_kernel<1024>), dim3((unsigned)rows), dim3(1024), 0, st, logits, (long)ldx, (int)V,
                           targets, weights, loss_rows, grad_scale, write_grad, label_smoothing);
#undef NM_XR
    NM_LAUNCH_CHECK("nm_xent");
}

// ---------------------------------------------------------------------------
// beam search step (decoders/beam_search_decoder.py:440-501).
// Inputs per hypothesis row r = b*k + j: logits [R,V] of the previous parent
// step with their (max, lse) row statistics; logprob_sum, lengths, finished.
//   lp        = finished ? (v==PAD ? 0 : -1e9) : (x - max - lse)
//   hyp       = logprob_sum + lp                       (un-normalised, carried)
//   score     = hyp / penalty[len + 1 - finished]      (penalty table from host,
//               ((5+len)/6)^alpha evaluated in fp32 exactly as the oracle does)
//   top-k over the k*V candidates of sentence b, ties -> lower flat index.
// Stage 1: each block scans a slice of the k*V candidates of one sentence and
// keeps its best k; stage 2 merges the slices and emits word / beam ids and the
// gathered search state.
// ---------------------------------------------------------------------------
#define BEAM_MAX_K 16

struct Cand { float score; int idx; };

__device__ __forceinline__ bool cand_better(float sa, int ia, float sb, int ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// insert into a descending sorted list of length K (registers)
template <int K>
__device__ __forceinline__ void topk_insert(float (&s)[K], int (&ix)[K], float v, int i) {
    if (!cand_better(v, i, s[K - 1], ix[K - 1])) return;
    s[K - 1] = v; ix[K - 1] = i;
#pragma unroll
    for (int p = K - 1; p > 0; --p) {
        if (cand_better(s[p], ix[p], s[p - 1], ix[p - 1])) {
            const float ts = s[p]; s[p] = s[p - 1]; s[p - 1] = ts;
            const int ti = ix[p]; ix[p] = ix[p - 1]; ix[p - 1] = ti;
        }
    }
}

__device__ __forceinline__ float beam_score(const float* __restrict__ logits, long ldx, int V, int k,
                                            int b, int flat, const float* rmax, const float* rlse,
                                            const float* logprob_sum, const int* lengths,
                                            const int* finished, const float* penalty, float* hyp_out) {
    const int j = flat / V, v = flat - j * V;
    const int r = b * k + j;
    const int fin = finished[r];
    float lp;
    if (fin) lp = (v == 0) ? 0.0f : NM_NEG_INF_F;
    else lp = (logits[(long)r * ldx + v] - rmax[r]) - rlse[r];
    const float hyp = logprob_sum[r] + lp;
    const int len = lengths[r] + 1 - (fin ? 1 : 0);
    if (hyp_out) *hyp_out = hyp;
    return hyp / penalty[len];
}

// top K of the union of two descending lists held in registers: C[p] = better(A[p], B[K-1-p]) is a
// bitonic sequence holding the K best, log2(K) compare-exchange stages sort it.  All indices are
// compile-time constants, so the lists stay in VGPRs.
template <int K>
__device__ __forceinline__ void topk_merge_regs(float (&s)[K], int (&ix)[K], const float (&os)[K], const int (&oi)[K]) {
#pragma unroll
    for (int p = 0; p < K; ++p) {
        if (cand_better(os[K - 1 - p], oi[K - 1 - p], s[p], ix[p])) { s[p] = os[K - 1 - p]; ix[p] = oi[K - 1 - p]; }
    }
#pragma unroll
    for (int stride = K / 2; stride >= 1; stride /= 2) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            if ((p & stride) == 0 && cand_better(s[p + stride], ix[p + stride], s[p], ix[p])) {
                const float ts = s[p]; s[p] = s[p + stride]; s[p + stride] = ts;
                const int ti = ix[p]; ix[p] = ix[p + stride]; ix[p + stride] = ti;
            }
        }
    }
}

// butterfly over the 64 lanes: afterwards every lane holds the wave's K best
template <int K>
__device__ __forceinline__ void topk_wave_merge(float (&s)[K], int (&ix)[K]) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float os[K];
        int oi[K];
#pragma unroll
        for (int p = 0; p < K; ++p) { os[p] = __shfl_xor(s[p], off, 64); oi[p] = __shfl_xor(ix[p], off, 64); }
        topk_merge_regs<K>(s, ix, os, oi);
    }
}

// Stage 1: workgroup (slice, r) scans a slice of ONE hypothesis row, so the row constants (finished,
// max, lse, logprob_sum, penalty) live in registers and the logits stream in as float4.  The score of
// every candidate is computed exactly as the reference does (subtract, subtract, add, divide): the
// selection is over the same fp32 values the oracle ranks.
template <int K>
__global__ __launch_bounds__(256) void beam_topk_partial(const float* __restrict__ logits, long ldx,
                                                         int V, int k, const float* __restrict__ rmax,
                                                         const float* __restrict__ rlse,
                                                         const float* __restrict__ logprob_sum,
                                                         const int* __restrict__ lengths,
                                                         const int* __restrict__ finished,
                                                         const float* __restrict__ penalty,
                                                         float* __restrict__ part_score,
                                                         int* __restrict__ part_idx, int ns, int per) {
    __shared__ float shs[4 * K];
    __shared__ int shi[4 * K];
    const int r = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = r % k;
    const int fin = finished[r];
    const float mx = rmax[r], lse = rlse[r], lps = logprob_sum[r];
    const float pen = penalty[lengths[r] + 1 - (fin ? 1 : 0)];
    const int beg = slice * per, end = min(V, beg + per);
    const float* row = logits + (long)r * ldx;
    const int base = j * V;
    float s[K];
    int ix[K];
#pragma unroll
    for (int p = 0; p < K; ++p) { s[p] = -INFINITY; ix[p] = 0x7fffffff; }
    if (fin) {
        // finished hypothesis: lp = 0 for <pad>, -1e9 otherwise (:444-456); equal scores -> lowest ids win
        const int v = beg + tid;
        if (tid < K && v < end) {
            const float lp = (v == 0) ? 0.0f : NM_NEG_INF_F;
            topk_insert<K>(s, ix, (lps + lp) / pen, base + v);
        }
    } else if ((ldx & 3) == 0 && nm_aligned16_dev(row)) {
        const int end4 = beg + ((end - beg) & ~3);
        for (int v = beg + tid * 4; v < end4; v += 1024) {
            const float4 x = *reinterpret_cast<const float4*>(row + v);
            topk_insert<K>(s, ix, (lps + ((x.x - mx) - lse)) / pen, base + v);
            topk_insert<K>(s, ix, (lps + ((x.y - mx) - lse)) / pen, base + v + 1);
            topk_insert<K>(s, ix, (lps + ((x.z - mx) - lse)) / pen, base + v + 2);
            topk_insert<K>(s, ix, (lps + ((x.w - mx) - lse)) / pen, base + v + 3);
        }
        for (int v = end4 + tid; v < end; v += 256)
            topk_insert<K>(s, ix, (lps + ((row[v] - mx) - lse)) / pen, base + v);
    } else {
        for (int v = beg + tid; v < end; v += 256)
            topk_insert<K>(s, ix, (lps + ((row[v] - mx) - lse)) / pen, base + v);
    }
    topk_wave_merge<K>(s, ix);
    if (lane == 0) {
#pragma unroll
        for (int p = 0; p < K; ++p) { shs[wave * K + p] = s[p]; shi[wave * K + p] = ix[p]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            s[p] = lane < 4 ? shs[lane * K + p] : -INFINITY;
            ix[p] = lane < 4 ? shi[lane * K + p] : 0x7fffffff;
        }
        topk_wave_merge<K>(s, ix);
        if (lane == 0) {
#pragma unroll
            for (int p = 0; p < K; ++p) {
                part_score[((long)r * ns + slice) * K + p] = s[p];
                part_idx[((long)r * ns + slice) * K + p] = ix[p];
            }
        }
    }
}

// Stage 2: one wave per sentence merges its k*ns slice lists (lanes take lists round-robin, butterfly
// merge), then lanes 0..k-1 emit word / beam ids and the gathered search state.
template <int K>
__global__ __launch_bounds__(64) void beam_topk_final(const float* __restrict__ logits, long ldx, int V, int k,
                                const float* __restrict__ rmax, const float* __restrict__ rlse,
                                const float* __restrict__ logprob_sum, const int* __restrict__ lengths,
                                const int* __restrict__ finished, const float* __restrict__ penalty,
                                const float* __restrict__ part_score, const int* __restrict__ part_idx,
                                int nslice, int B, int end_id, float* __restrict__ out_score,
                                int* __restrict__ out_word, int* __restrict__ out_beam,
                                float* __restrict__ out_logprob_sum, int* __restrict__ out_lengths,
                                int* __restrict__ out_finished, int* __restrict__ out_src_row,
                                int* __restrict__ all_finished) {
    __shared__ float fs[K];
    __shared__ int fi[K];
    __shared__ float cs[256];
    __shared__ int ci[256];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int ncand = nslice * K;
    if (ncand <= 256) {
        // few lists (the tile-scan path leaves k of them): no merge tree -- every candidate counts how many of the
        // others beat it ((score desc, flat index asc) is a strict order over distinct indices) and drops itself
        // into that slot.  A few hundred broadcast LDS reads instead of six dependent butterfly rounds.
        for (int c = lane; c < ncand; c += 64) {
            cs[c] = part_score[(long)b * ncand + c];
            ci[c] = part_idx[(long)b * ncand + c];
        }
        __syncthreads();
        for (int c = lane; c < ncand; c += 64) {
            const float sc = cs[c];
            const int ic = ci[c];
            int rank = 0;
            for (int j = 0; j < ncand; ++j) rank += cand_better(cs[j], ci[j], sc, ic) ? 1 : 0;
            if (rank < K && ic != 0x7fffffff) { fs[rank] = sc; fi[rank] = ic; }
        }
        __syncthreads();
    } else {
    float s[K];
    int ix[K];
#pragma unroll
    for (int p = 0; p < K; ++p) { s[p] = -INFINITY; ix[p] = 0x7fffffff; }
    for (int sl = lane; sl < nslice; sl += 64) {
        float os[K];
        int oi[K];
#pragma unroll
        for (int p = 0; p < K; ++p) {
            os[p] = part_score[((long)b * nslice + sl) * K + p];
            oi[p] = part_idx[((long)b * nslice + sl) * K + p];
        }
        topk_merge_regs<K>(s, ix, os, oi);
    }
    topk_wave_merge<K>(s, ix);
    if (lane == 0) {
#pragma unroll
        for (int p = 0; p < K; ++p) { fs[p] = s[p]; fi[p] = ix[p]; }
    }
    __syncthreads();
    }
    if (lane < k) {
        const int p = lane;
        const int flat = fi[p];
        const int j = flat / V, v = flat - j * V;
        const int r = b * k + j;
        float hyp;
        beam_score(logits, ldx, V, k, b, flat, rmax, rlse, logprob_sum, lengths, finished, penalty, &hyp);
        const int fin = finished[r];
        const int o = b * k + p;
        out_score[o] = fs[p];
        out_word[o] = v;
        out_beam[o] = j;
        out_logprob_sum[o] = hyp;
        out_lengths[o] = lengths[r] + 1 - (fin ? 1 : 0);
        const int nf = fin | (v == end_id);
        out_finished[o] = nf;
        out_src_row[o] = r;
        if (all_finished && !nf) *all_finished = 0;     // plain store, see nm_greedy_update
    }
}

// ---------------------------------------------------------------------------------------------
// Register-resident row scan: one 1024-thread workgroup per row loads the whole row ONCE (up to NV
// float4 per thread, all loads issued before the first use), and from registers derives max /
// first argmax / lse and -- for the beam -- the row's K best candidates with the exact scores.
// One HBM pass instead of the three of row_stats (2) + top-k (1): at B*k = 640 rows of 32000 logits
// that is 82 MB read once.
// ---------------------------------------------------------------------------------------------
#define ROW_SCAN_CAND 256
template <int K, int NV, bool TOPK>
__global__ __launch_bounds__(1024) void row_scan_kernel(const float* __restrict__ x, long ldx, int V,
                                                        float* __restrict__ max_out, float* __restrict__ lse_out,
                                                        int* __restrict__ argmax_out, int k,
                                                        const float* __restrict__ logprob_sum,
                                                        const int* __restrict__ lengths,
                                                        const int* __restrict__ finished,
                                                        const float* __restrict__ penalty,
                                                        float* __restrict__ part_score, int* __restrict__ part_idx) {
    __shared__ float shv[16];
    __shared__ int shi[16];
    __shared__ float shs[16 * K];
    __shared__ int shx[16 * K];
    __shared__ float cand_s[ROW_SCAN_CAND];
    __shared__ int cand_i[ROW_SCAN_CAND];
    __shared__ int cand_n;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = TOPK ? (r % k) * V : 0;
    if (TOPK && finished[r]) {
        // finished hypothesis: lp = 0 for <pad>, -1e9 otherwise (:444-456): the K lowest ids win, no scan
        if (tid < K) {
            const float pen = penalty[lengths[r]];
            const float lp = (tid == 0) ? 0.0f : NM_NEG_INF_F;
            part_score[(long)r * K + tid] = tid < V ? (logprob_sum[r] + lp) / pen : -INFINITY;
            part_idx[(long)r * K + tid] = tid < V ? base + tid : 0x7fffffff;
        }
        return;
    }
    const float4* x4 = reinterpret_cast<const float4*>(x + (long)r * ldx);
    const int V4 = V >> 2;
    float4 xv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = tid + i * 1024;
        xv[i] = q < V4 ? x4[q] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NV; ++i) {          // ascending element order within the thread: first max kept
        const int e = (tid + i * 1024) * 4;
        if (xv[i].x > bv) { bv = xv[i].x; bi = e; }
        if (xv[i].y > bv) { bv = xv[i].y; bi = e + 1; }
        if (xv[i].z > bv) { bv = xv[i].z; bi = e + 2; }
        if (xv[i].w > bv) { bv = xv[i].w; bi = e + 3; }
    }
    block_argmax<1024>(bv, bi, shv, shi);
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        sum += expf(xv[i].x - bv) + expf(xv[i].y - bv) + expf(xv[i].z - bv) + expf(xv[i].w - bv);
    sum = block_sum<1024>(sum, shv);
    const float lse = logf(sum);
    if (tid == 0) {
        if (max_out) max_out[r] = bv;
        if (lse_out) lse_out[r] = lse;
        if (argmax_out) argmax_out[r] = bi;
    }
    if (!TOPK) return;
    const float lps = logprob_sum[r];
    const float pen = penalty[lengths[r] + 1];
    float s[K];
    int ix[K];
#pragma unroll
    for (int p = 0; p < K; ++p) { s[p] = -INFINITY; ix[p] = 0x7fffffff; }

    // ---- candidate filter.  The score (lps + ((x - max) - lse)) / pen is monotone in x, so the row's K
    // best candidates are among its largest logits.  tau = the K-th largest of the 16 wave maxima: at
    // least K elements are >= tau, hence every element of the exact top K scores at least score(tau).
    // Elements below tau - margin score STRICTLY less (margin = 2^-18 of the magnitudes entering the
    // score, far above the few ulps within which distinct logits can round to equal scores) and
    // cannot be in the top K whatever the index tie-break says.  Typically 5-10 of the 32000 logits
    // survive; they are scored exactly and ranked by the same (score, flat index) order as before.  A
    // row whose survivors do not fit the list (many equal logits, or |logprob_sum| so large that all
    // scores collapse: the -1e9 of the not-yet-live beams of the first step) takes the full
    // insertion path below.
    float tmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) tmax = fmaxf(tmax, fmaxf(fmaxf(xv[i].x, xv[i].y), fmaxf(xv[i].z, xv[i].w)));
    tmax = nm_wave_max(tmax);
    __syncthreads();                       // shv was used by the block reductions above
    if (lane == 0) shv[wave] = tmax;
    if (tid == 0) cand_n = 0;
    __syncthreads();
    float tau = -INFINITY;
    {
        float wm = lane < 16 ? shv[lane] : -INFINITY;      // every wave derives tau itself (no second barrier)
#pragma unroll
        for (int p = 0; p < K; ++p) {
            tau = nm_wave_max(wm);
            const unsigned long long hit = __ballot(wm == tau);
            if (lane == __ffsll((long long)hit) - 1) wm = -INFINITY;     // drop ONE instance of the maximum
        }
    }
    const float margin = (fabsf(lse) + fabsf(lps) + (bv - tau) + 1.0f) * (1.0f / 262144.0f);
    const float cut = tau - margin;
    if (cut > -INFINITY) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = tid + i * 1024;
            const float xe[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (xe[c] >= cut) {                                       // (padding lanes hold -inf)
                    const int slot = atomicAdd(&cand_n, 1);
                    if (slot < ROW_SCAN_CAND) {
                        cand_s[slot] = (lps + ((xe[c] - bv) - lse)) / pen;
                        cand_i[slot] = base + q * 4 + c;
                    }
                }
            }
        }
    }
    __syncthreads();
    const int ncand = cand_n;
    if (cut > -INFINITY && ncand <= ROW_SCAN_CAND) {
        if (wave == 0) {
            for (int c = lane; c < ncand; c += 64) topk_insert<K>(s, ix, cand_s[c], cand_i[c]);
            topk_wave_merge<K>(s, ix);
            if (lane == 0) {
#pragma unroll
                for (int p = 0; p < K; ++p) { part_score[(long)r * K + p] = s[p]; part_idx[(long)r * K + p] = ix[p]; }
            }
        }
        return;
    }
    // ---- full path: every element through the per-thread insertion lists and the merge tree
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = tid + i * 1024;
        if (q < V4) {
            const int e = base + q * 4;
            topk_insert<K>(s, ix, (lps + ((xv[i].x - bv) - lse)) / pen, e);
            topk_insert<K>(s, ix, (lps + ((xv[i].y - bv) - lse)) / pen, e + 1);
            topk_insert<K>(s, ix, (lps + ((xv[i].z - bv) - lse)) / pen, e + 2);
            topk_insert<K>(s, ix, (lps + ((xv[i].w - bv) - lse)) / pen, e + 3);
        }
    }
    topk_wave_merge<K>(s, ix);
    if (lane == 0) {
#pragma unroll
        for (int p = 0; p < K; ++p) { shs[wave * K + p] = s[p]; shx[wave * K + p] = ix[p]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            s[p] = lane < 16 ? shs[lane * K + p] : -INFINITY;
            ix[p] = lane < 16 ? shx[lane * K + p] : 0x7fffffff;
        }
        topk_wave_merge<K>(s, ix);
        if (lane == 0) {
#pragma unroll
            for (int p = 0; p < K; ++p) { part_score[(long)r * K + p] = s[p]; part_idx[(long)r * K + p] = ix[p]; }
        }
    }
}

// rows whose length fits the register file of one workgroup and whose float4 view is aligned
static inline int row_scan_nv(const float* x, int64_t ldx, int64_t V) {
    if ((V & 3) || (ldx & 3) || !nm_aligned16(x)) return 0;
    const int64_t v4 = V >> 2;
    if (v4 <= 8 * 1024) return 8;
    if (v4 <= 16 * 1024) return 16;
    if (v4 <= 32 * 1024) return 32;
    return 0;
}

extern "C" int nm_row_stats(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t V,
                            float* max_out, float* lse_out, int32_t* argmax_out) {
    NM_REQUIRE(x && rows >= 0 && V > 0 && ldx >= V, "nm_row_stats: bad args");
    if (rows == 0) return NM_OK;
    const int nv = row_scan_nv(x, ldx, V);
#define NM_RS(NV_)                                                                                         \
    hipLaunchKernelGGL((row_scan_kernel<4, NV_, false>), dim3((unsigned)rows), dim3(1024), 0, nm_stream(stream), x, \
                       (long)ldx, (int)V, max_out, lse_out, argmax_out, 1, (const float*)nullptr,             \
                       (const int*)nullptr, (const int*)nullptr, (const float*)nullptr, (float*)nullptr,      \
                       (int*)nullptr)
    if (nv == 8) NM_RS(8);
    else if (nv == 16) NM_RS(16);
    else if (nv == 32) NM_RS(32);
    else
        hipLaunchKernelGGL((row_stats_kernel<1024>), dim3((unsigned)rows), dim3(1024), 0, nm_stream(stream),
                           x, (long)ldx, (int)V, max_out, lse_out, argmax_out);
#undef NM_RS
    NM_LAUNCH_CHECK("nm_row_stats");
}

// ---------------------------------------------------------------------------
// Sampling step (decoders/autoregressive.py:470-473: tf.multinomial(logits, 1) when ``sample``): one draw per row
// from softmax(x), as argmax_c (x[r,c] + g[r,c]) with Gumbel noise g = -log(-log(u)).  TF's Philox stream cannot be
// replayed by anyone; u is a counter-based hash of (salt, row, column) as for dropout (nm_eltwise.hip), so that the
// CPU checker restates the draw (oracle/nm_oracle.py:gumbel_noise):
//   key = mix32(salt + row * 0x85EBCA6B),  bits = mix32(column * 0x9E3779B1 + key),  u = ((bits >> 9) + 0.5) / 2^23
// (23 bits + 0.5 is exact in fp32, so u stays strictly inside (0, 1) and the noise finite for every bit pattern).
// The caller folds the step (and whatever else distinguishes two draws) into ``salt``.  Ties: the first maximum.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sample_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}

__global__ __launch_bounds__(256) void gumbel_argmax_kernel(const float* __restrict__ x, long ldx, int V, uint32_t salt,
                                                            int* __restrict__ out) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int r = blockIdx.x;
    const float* row = x + (long)r * ldx;
    const uint32_t key = sample_mix32(salt + (uint32_t)r * 0x85EBCA6Bu);
    float best = -INFINITY;
    int bi = 0;                                       // a row of NaN / -inf logits still yields a valid symbol
    for (int c = threadIdx.x; c < V; c += 256) {
        const uint32_t bits = sample_mix32((uint32_t)c * 0x9E3779B1u + key);
        const float u = ((float)(bits >> 9) + 0.5f) * (1.0f / 8388608.0f);
        const float v = row[c] - logf(-logf(u));
        if (v > best) { best = v; bi = c; }           // ascending columns per thread: its first maximum
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        out[r] = bi;
    }
}

extern "C" int nm_gumbel_argmax(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t V, uint32_t salt,
                                int32_t* out) {
    NM_REQUIRE(x && out && rows >= 0 && V > 0 && ldx >= V && V < (1 << 30), "nm_gumbel_argmax: bad args");
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(gumbel_argmax_kernel, dim3((unsigned)rows), dim3(256), 0, nm_stream(stream), x, (long)ldx,
                       (int)V, salt, out);
    NM_LAUNCH_CHECK("nm_gumbel_argmax");
}

extern "C" int64_t nm_beam_workspace_bytes(int64_t B, int64_t k, int64_t V) {
    (void)k; (void)V;
    return B * 64 * BEAM_MAX_K * 8 + 256;
}

extern "C" int nm_beam_topk_step(void* stream, const float* logits, int64_t ldx, int64_t B, int64_t k,
                                 int64_t V, const float* rmax, const float* rlse,
                                 const float* logprob_sum, const int32_t* lengths,
                                 const int32_t* finished, const float* penalty, int end_id,
                                 float* out_score, int32_t* out_word, int32_t* out_beam,
                                 float* out_logprob_sum, int32_t* out_lengths, int32_t* out_finished,
                                 int32_t* out_src_row, void* workspace, int64_t workspace_bytes,
                                 int32_t* all_finished) {
    NM_REQUIRE(logits && rmax && rlse && logprob_sum && lengths && finished && penalty && out_score &&
                   out_word && out_beam && out_logprob_sum && out_lengths && out_finished &&
                   out_src_row && workspace,
               "nm_beam_topk_step: null pointer");
    NM_REQUIRE(B > 0 && k >= 1 && k <= BEAM_MAX_K && V > 0 && k * V < (1L << 31),
               "nm_beam_topk_step: bad shape B=%ld k=%ld V=%ld", (long)B, (long)k, (long)V);
    NM_REQUIRE(k * V >= k, "nm_beam_topk_step: fewer candidates than beam");
    NM_REQUIRE(workspace_bytes >= nm_beam_workspace_bytes(B, k, V), "nm_beam_topk_step: workspace too small");
    // ns slices per hypothesis row (>= 4096 candidates each, k*ns <= 64 lists per sentence)
    const int ns_env = nm_cur()->sw.beam_ns;                                            // tuning override
    int ns = ns_env > 0 ? ns_env : (int)((V + 4095) / 4096);
    if (ns < 1) ns = 1;
    if (ns > 64 / k) ns = (int)(64 / k);
    const int per = (int)((((V + ns - 1) / ns) + 3) & ~3L);          // multiple of 4: float4 loads stay aligned
    const int nslice = (int)(k * ns);                                 // lists per sentence, contiguous
    NM_REQUIRE(B * k <= 65535, "nm_beam_topk_step: too many hypothesis rows");
    float* ps = reinterpret_cast<float*>(workspace);
    int* pi = reinterpret_cast<int*>(ps + B * 64 * BEAM_MAX_K);
    hipStream_t st = nm_stream(stream);
    dim3 grid(ns, (unsigned)(B * k));
#define NM_BK(K_)                                                                                   \
    do {                                                                                            \
        hipLaunchKernelGGL((beam_topk_partial<K_>), grid, dim3(256), 0, st, logits, (long)ldx, (int)V, \
                           (int)k, rmax, rlse, logprob_sum, lengths, finished, penalty, ps, pi, ns, per); \
        hipLaunchKernelGGL((beam_topk_final<K_>), dim3((unsigned)B), dim3(64), 0, st, logits,         \
                           (long)ldx, (int)V, (int)k, rmax, rlse, logprob_sum, lengths, finished,    \
                           penalty, ps, pi, nslice, (int)B, end_id, out_score, out_word, out_beam,   \
                           out_logprob_sum, out_lengths, out_finished, out_src_row, all_finished);   \
    } while (0)
    if (k <= 4) NM_BK(4);
    else if (k <= 8) NM_BK(8);
    else NM_BK(16);
#undef NM_BK
    NM_LAUNCH_CHECK("nm_beam_topk_step");
}

// The same beam step reading the raw logits of the previous parent step directly: max / lse and the
// per-row candidates come from one register-resident scan of each row (row_scan_kernel), the second
// stage merges the k row lists of a sentence.  rmax_out / rlse_out / argmax_out are optional.
extern "C" int nm_beam_topk_step_fused(void* stream, const float* logits, int64_t ldx, int64_t B, int64_t k,
                                       int64_t V, const float* logprob_sum, const int32_t* lengths,
                                       const int32_t* finished, const float* penalty, int end_id,
                                       float* out_score, int32_t* out_word, int32_t* out_beam,
                                       float* out_logprob_sum, int32_t* out_lengths, int32_t* out_finished,
                                       int32_t* out_src_row, void* workspace, int64_t workspace_bytes,
                                       int32_t* all_finished, float* rmax_out, float* rlse_out) {
    NM_REQUIRE(logits && logprob_sum && lengths && finished && penalty && out_score && out_word && out_beam &&
                   out_logprob_sum && out_lengths && out_finished && out_src_row && workspace && rmax_out && rlse_out,
               "nm_beam_topk_step_fused: null pointer");
    NM_REQUIRE(B > 0 && k >= 1 && k <= BEAM_MAX_K && V > 0 && k * V < (1L << 31),
               "nm_beam_topk_step_fused: bad shape B=%ld k=%ld V=%ld", (long)B, (long)k, (long)V);
    NM_REQUIRE(workspace_bytes >= nm_beam_workspace_bytes(B, k, V), "nm_beam_topk_step_fused: workspace too small");
    const int nv = row_scan_nv(logits, ldx, V);
    hipStream_t st = nm_stream(stream);
    if (nv == 0 || k > 8) {   // unaligned or very long rows, or a beam wider than the register-resident scan's lists
                              // (its 16-wide instances spill): statistics pass + sliced top-k pass
        hipLaunchKernelGGL((row_stats_kernel<1024>), dim3((unsigned)(B * k)), dim3(1024), 0, st, logits, (long)ldx,
                           (int)V, rmax_out, rlse_out, (int*)nullptr);
        return nm_beam_topk_step(stream, logits, ldx, B, k, V, rmax_out, rlse_out, logprob_sum, lengths, finished,
                                 penalty, end_id, out_score, out_word, out_beam, out_logprob_sum, out_lengths,
                                 out_finished, out_src_row, workspace, workspace_bytes, all_finished);
    }
    float* ps = reinterpret_cast<float*>(workspace);
    int* pi = reinterpret_cast<int*>(ps + B * 64 * BEAM_MAX_K);
    const unsigned rows = (unsigned)(B * k);
#define NM_SCAN(K_, NV_)                                                                                       \
    hipLaunchKernelGGL((row_scan_kernel<K_, NV_, true>), dim3(rows), dim3(1024), 0, st, logits, (long)ldx, (int)V, \
                       rmax_out, rlse_out, (int*)nullptr, (int)k, logprob_sum, lengths, finished, penalty, ps, pi)
#define NM_FUSED(K_)                                                                                           \
    do {                                                                                                       \
        if (nv == 8) NM_SCAN(K_, 8);                                                                           \
        else if (nv == 16) NM_SCAN(K_, 16);                                                                    \
        else NM_SCAN(K_, 32);                                                                                  \
        hipLaunchKernelGGL((beam_topk_final<K_>), dim3((unsigned)B), dim3(64), 0, st, logits, (long)ldx, (int)V, \
                           (int)k, rmax_out, rlse_out, logprob_sum, lengths, finished, penalty, ps, pi, (int)k,  \
                           (int)B, end_id, out_score, out_word, out_beam, out_logprob_sum, out_lengths,         \
                           out_finished, out_src_row, all_finished);                                           \
    } while (0)
    if (k <= 4) NM_FUSED(4);
    else NM_FUSED(8);
#undef NM_FUSED
#undef NM_SCAN
    NM_LAUNCH_CHECK("nm_beam_topk_step_fused");
}

// ---------------------------------------------------------------------------------------------
// Consumers of the per-tile row statistics that nm_logits_stats_gemm leaves behind
// (stats[row][tile] = {max, sum exp(x - max), argmax bits, -}, tiles of `tile_w` columns: nm_logits_stats_tile(rows)).
// ---------------------------------------------------------------------------------------------

// merge the tiles of one row: global max, first argmax, lse = log(sum exp(x - max)); every thread of the
// block returns the same values.  `sh` needs 3 * (NT/64) words.
template <int NT>
__device__ __forceinline__ void merge_row_tiles(const float4* __restrict__ st, int ntiles, float& M, int& A, float& lse,
                                                float* shf, int* shi) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int t = threadIdx.x; t < ntiles; t += NT) {
        const float4 r = st[t];
        const int a = __float_as_int(r.z);
        if (r.x > bv || (r.x == bv && a < bi)) { bv = r.x; bi = a; }
    }
    block_argmax<NT>(bv, bi, shf, shi);
    float s = 0.0f;
    for (int t = threadIdx.x; t < ntiles; t += NT) {
        const float4 r = st[t];
        s += r.y * expf(r.x - bv);
    }
    s = block_sum<NT>(s, shf);
    M = bv; A = bi; lse = logf(s);
}

// greedy step tail (decoders/autoregressive.py:461-480) on the tile statistics: argmax over the
// vocabulary, symbol zeroing / finished update (nm_greedy_update) and the embedding lookup of the new
// symbol (autoregressive.py:269-272), one workgroup per sentence.
__global__ __launch_bounds__(256) void greedy_finish_kernel(const float4* __restrict__ stats, int ntiles,
                                                            int* __restrict__ finished, int* __restrict__ sym_out,
                                                            int* __restrict__ mask_out, int end_id,
                                                            int* __restrict__ all_finished,
                                                            const float* __restrict__ table, int V, int E,
                                                            float* __restrict__ emb_out, long ld_emb,
                                                            int* __restrict__ argmax_out, float* __restrict__ max_out,
                                                            float* __restrict__ lse_out) {
    __shared__ float shf[4];
    __shared__ int shi[4];
    __shared__ int sym_s;
    const int r = blockIdx.x;
    float M, lse;
    int A;
    merge_row_tiles<256>(stats + (long)r * ntiles, ntiles, M, A, lse, shf, shi);
    if (threadIdx.x == 0) {
        int f = finished[r];
        const int s = f ? 0 : A;
        f = f | (s == end_id);
        finished[r] = f;
        sym_out[r] = s;
        if (mask_out) mask_out[r] = !f;
        if (all_finished && !f) *all_finished = 0;       // (a plain store: every writer writes the same 0 --
                                                       //  640 atomics on one word were ~8 us of a beam step)
        if (argmax_out) argmax_out[r] = A;
        if (max_out) max_out[r] = M;
        if (lse_out) lse_out[r] = lse;
        sym_s = s;
    }
    __syncthreads();
    if (!emb_out) return;
    int s = sym_s;
    s = s < 0 ? 0 : (s >= V ? V - 1 : s);
    const float* src = table + (long)s * E;
    float* dst = emb_out + (long)r * ld_emb;
    if ((E & 3) == 0 && (ld_emb & 3) == 0 && nm_aligned16_dev(src) && nm_aligned16_dev(dst)) {
        for (int c = threadIdx.x * 4; c < E; c += 1024)
            *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
    } else {
        for (int c = threadIdx.x; c < E; c += 256) dst[c] = src[c];
    }
}

extern "C" int nm_greedy_finish(void* stream, const float* stats, int64_t ntiles, int64_t R, int32_t* finished,
                                int32_t* sym_out, int32_t* mask_out, int end_id, int32_t* all_finished,
                                const float* table, int64_t V, int64_t E, float* emb_out, int64_t ld_emb,
                                int32_t* argmax_out, float* max_out, float* lse_out) {
    NM_REQUIRE(stats && finished && sym_out && R >= 0 && ntiles > 0, "nm_greedy_finish: bad args");
    NM_REQUIRE(nm_aligned16(stats), "nm_greedy_finish: statistics must be 16-byte aligned");
    NM_REQUIRE(!emb_out || (table && V > 0 && E > 0 && ld_emb >= E), "nm_greedy_finish: bad embedding arguments");
    if (R == 0) return NM_OK;
    hipLaunchKernelGGL(greedy_finish_kernel, dim3((unsigned)R), dim3(256), 0, nm_stream(stream),
                       reinterpret_cast<const float4*>(stats), (int)ntiles, finished, sym_out, mask_out, end_id,
                       all_finished, table, (int)V, (int)E, emb_out, (long)ld_emb, argmax_out, max_out, lse_out);
    NM_LAUNCH_CHECK("nm_greedy_finish");
}

// Beam step, first stage, on the tile statistics: max / lse of the row come from the merged tiles; the
// candidate filter of row_scan_kernel works on TILE maxima (tau = the K-th largest tile maximum: at least K
// elements of the row are >= tau, so the exact top K all score at least score(tau); elements below
// tau - margin score strictly less), so only the handful of tiles whose maximum reaches tau - margin are
// read back from the logits: ~K x 512 B per row instead of the whole 128 KB row.  Exact scores and the
// (score, flat index) order are those of row_scan_kernel / beam_topk_partial; a row whose survivors do not
// fit the list takes the full insertion path over the whole row.
#define TILE_SCAN_TILES 64
template <int K>
__global__ __launch_bounds__(256) void beam_tile_scan_kernel(const float* __restrict__ x, long ldx, int V,
                                                             const float4* __restrict__ stats, int ntiles, int tile_w,
                                                             float* __restrict__ max_out, float* __restrict__ lse_out,
                                                             int k, const float* __restrict__ logprob_sum,
                                                             const int* __restrict__ lengths,
                                                             const int* __restrict__ finished,
                                                             const float* __restrict__ penalty,
                                                             float* __restrict__ part_score, int* __restrict__ part_idx) {
    __shared__ float shf[4];
    __shared__ int shi[4];
    __shared__ float shs[4 * K];
    __shared__ int shx[4 * K];
    __shared__ float cand_s[ROW_SCAN_CAND];
    __shared__ int cand_i[ROW_SCAN_CAND];
    __shared__ int cand_n, tile_n;
    __shared__ int tile_list[TILE_SCAN_TILES];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = (r % k) * V;
    if (finished[r]) {
        // finished hypothesis: lp = 0 for <pad>, -1e9 otherwise (:444-456): the K lowest ids win, no scan
        if (tid < K) {
            const float pen = penalty[lengths[r]];
            const float lp = (tid == 0) ? 0.0f : NM_NEG_INF_F;
            part_score[(long)r * K + tid] = tid < V ? (logprob_sum[r] + lp) / pen : -INFINITY;
            part_idx[(long)r * K + tid] = tid < V ? base + tid : 0x7fffffff;
        }
        return;
    }
    const float4* st = stats + (long)r * ntiles;
    float bv, lse;
    int bi;
    merge_row_tiles<256>(st, ntiles, bv, bi, lse, shf, shi);
    if (tid == 0) {
        if (max_out) max_out[r] = bv;
        if (lse_out) lse_out[r] = lse;
        cand_n = 0;
        tile_n = 0;
    }
    const float lps = logprob_sum[r];
    const float pen = penalty[lengths[r] + 1];
    // tau = K-th largest tile maximum (K knock-out rounds over the block; ntiles is a few hundred)
    float tau = -INFINITY;
    {
        float mine[4];                                      // this thread's tile maxima (ntiles <= 1024)
#pragma unroll
        for (int q = 0; q < 4; ++q) mine[q] = (tid + q * 256 < ntiles) ? st[tid + q * 256].x : -INFINITY;
        for (int p = 0; p < K; ++p) {
            float v = fmaxf(fmaxf(mine[0], mine[1]), fmaxf(mine[2], mine[3]));
            int who = tid;
            block_argmax<256>(v, who, shf, shi);            // (value desc, thread id asc): a unique owner
            tau = v;
            if (who == tid) {                               // drop ONE instance of the maximum
                bool done = false;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (!done && mine[q] == v) { mine[q] = -INFINITY; done = true; }
            }
        }
    }
    const float margin = (fabsf(lse) + fabsf(lps) + (bv - tau) + 1.0f) * (1.0f / 262144.0f);
    const float cut = tau - margin;
    __syncthreads();
    bool overflow = !(cut > -INFINITY) || ntiles > 1024;
    if (!overflow) {
        for (int t = tid; t < ntiles; t += 256) {
            if (st[t].x >= cut) {
                const int slot = atomicAdd(&tile_n, 1);
                if (slot < TILE_SCAN_TILES) tile_list[slot] = t;
            }
        }
        __syncthreads();
        const int nt = tile_n;
        if (nt > TILE_SCAN_TILES) overflow = true;
        else {
            const float* row = x + (long)r * ldx;
            for (int i = tid; i < nt * tile_w; i += 256) {
                const int col = tile_list[i / tile_w] * tile_w + (i % tile_w);
                if (col < V) {
                    const float xe = row[col];
                    if (xe >= cut) {
                        const int slot = atomicAdd(&cand_n, 1);
                        if (slot < ROW_SCAN_CAND) {
                            cand_s[slot] = (lps + ((xe - bv) - lse)) / pen;
                            cand_i[slot] = base + col;
                        }
                    }
                }
            }
            __syncthreads();
            if (cand_n > ROW_SCAN_CAND) overflow = true;
        }
    }
    float s[K];
    int ix[K];
#pragma unroll
    for (int p = 0; p < K; ++p) { s[p] = -INFINITY; ix[p] = 0x7fffffff; }
    if (!overflow) {
        // the survivors (<= 256, one per thread) rank themselves: the slot of a candidate is the number of
        // candidates that beat it; slots beyond the survivors are padded by the threads that hold none
        const int ncand = cand_n;
        if (tid < ncand) {
            const float sc = cand_s[tid];
            const int ic = cand_i[tid];
            int rank = 0;
            for (int j = 0; j < ncand; ++j) rank += cand_better(cand_s[j], cand_i[j], sc, ic) ? 1 : 0;
            if (rank < K) { part_score[(long)r * K + rank] = sc; part_idx[(long)r * K + rank] = ic; }
        } else if (tid < K) {
            part_score[(long)r * K + tid] = -INFINITY;
            part_idx[(long)r * K + tid] = 0x7fffffff;
        }
        return;
    }
    // ---- full path: every element of the row through the per-thread insertion lists and the merge tree
    {
        const float* row = x + (long)r * ldx;
        for (int v = tid; v < V; v += 256) topk_insert<K>(s, ix, (lps + ((row[v] - bv) - lse)) / pen, base + v);
    }
    topk_wave_merge<K>(s, ix);
    if (lane == 0) {
#pragma unroll
        for (int p = 0; p < K; ++p) { shs[wave * K + p] = s[p]; shx[wave * K + p] = ix[p]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            s[p] = lane < 4 ? shs[lane * K + p] : -INFINITY;
            ix[p] = lane < 4 ? shx[lane * K + p] : 0x7fffffff;
        }
        topk_wave_merge<K>(s, ix);
        if (lane == 0) {
#pragma unroll
            for (int p = 0; p < K; ++p) { part_score[(long)r * K + p] = s[p]; part_idx[(long)r * K + p] = ix[p]; }
        }
    }
}

// nm_beam_topk_step_fused on logits whose per-tile statistics are known (nm_logits_stats_gemm)
extern "C" int nm_beam_topk_step_tiles(void* stream, const float* logits, int64_t ldx, const float* stats,
                                       int64_t tile_w, int64_t B, int64_t k, int64_t V, const float* logprob_sum,
                                       const int32_t* lengths, const int32_t* finished, const float* penalty,
                                       int end_id, float* out_score, int32_t* out_word, int32_t* out_beam,
                                       float* out_logprob_sum, int32_t* out_lengths, int32_t* out_finished,
                                       int32_t* out_src_row, void* workspace, int64_t workspace_bytes,
                                       int32_t* all_finished, float* rmax_out, float* rlse_out) {
    NM_REQUIRE(logits && stats && logprob_sum && lengths && finished && penalty && out_score && out_word && out_beam &&
                   out_logprob_sum && out_lengths && out_finished && out_src_row && workspace && rmax_out && rlse_out,
               "nm_beam_topk_step_tiles: null pointer");
    NM_REQUIRE(B > 0 && k >= 1 && k <= BEAM_MAX_K && V > 0 && k * V < (1L << 31) && ldx >= V,
               "nm_beam_topk_step_tiles: bad shape B=%ld k=%ld V=%ld", (long)B, (long)k, (long)V);
    NM_REQUIRE((tile_w == 64 || tile_w == 128) && nm_aligned16(stats),
               "nm_beam_topk_step_tiles: tile width %ld (nm_logits_stats_tile gives 64 or 128)", (long)tile_w);
    const int64_t ntiles = (V + tile_w - 1) / tile_w;
    NM_REQUIRE(workspace_bytes >= nm_beam_workspace_bytes(B, k, V), "nm_beam_topk_step_tiles: workspace too small");
    float* ps = reinterpret_cast<float*>(workspace);
    int* pi = reinterpret_cast<int*>(ps + B * 64 * BEAM_MAX_K);
    const unsigned rows = (unsigned)(B * k);
    hipStream_t st = nm_stream(stream);
    const float4* st4 = reinterpret_cast<const float4*>(stats);
#define NM_TILES(K_)                                                                                              \
    do {                                                                                                          \
        hipLaunchKernelGGL((beam_tile_scan_kernel<K_>), dim3(rows), dim3(256), 0, st, logits, (long)ldx, (int)V, st4, \
                           (int)ntiles, (int)tile_w, rmax_out, rlse_out, (int)k, logprob_sum, lengths, finished, penalty, \
                           ps, pi);                                                                               \
        hipLaunchKernelGGL((beam_topk_final<K_>), dim3((unsigned)B), dim3(64), 0, st, logits, (long)ldx, (int)V,   \
                           (int)k, rmax_out, rlse_out, logprob_sum, lengths, finished, penalty, ps, pi, (int)k,    \
                           (int)B, end_id, out_score, out_word, out_beam, out_logprob_sum, out_lengths,           \
                           out_finished, out_src_row, all_finished);                                             \
    } while (0)
    if (k <= 4) NM_TILES(4);
    else if (k <= 8) NM_TILES(8);
    else NM_TILES(16);
#undef NM_TILES
    NM_LAUNCH_CHECK("nm_beam_topk_step_tiles");
}

// ---------------------------------------------------------------------------
// row gather: dst[r,:] = src[idx[r],:]   (beam reorder of decoder state,
// beam_search_decoder.py:503-532 / tf_utils.py:106-131) and the int32 variant
// for token histories [steps, R].
// ---------------------------------------------------------------------------
__global__ void gather_rows_f32_kernel(const float* __restrict__ src, long lds_, const int* __restrict__ idx,
                                       float* __restrict__ dst, long ldd, long rows, int w) {
    const long r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < w) dst[r * ldd + c] = src[(long)idx[r] * lds_ + c];
}

extern "C" int nm_gather_rows_f32(void* stream, const float* src, int64_t ld_src, const int32_t* idx,
                                  float* dst, int64_t ld_dst, int64_t rows, int64_t width) {
    NM_REQUIRE(src && idx && dst && rows >= 0 && width >= 0 && src != dst, "nm_gather_rows_f32: bad args");
    if (rows == 0 || width == 0) return NM_OK;
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(nm_cdiv(width, 256), (unsigned)rows), dim3(256), 0,
                       nm_stream(stream), src, (long)ld_src, idx, dst, (long)ld_dst, (long)rows, (int)width);
    NM_LAUNCH_CHECK("nm_gather_rows_f32");
}

// token history reorder: dst[t, r] = src[t, idx[r]] for t < steps; dst[steps, r] = word[r]
__global__ void beam_tokens_kernel(const int* __restrict__ src, const int* __restrict__ idx,
                                   const int* __restrict__ word, int* __restrict__ dst, int steps, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int sr = idx[r];
    for (int t = 0; t < steps; ++t) dst[(long)t * R + r] = src[(long)t * R + sr];
    dst[(long)steps * R + r] = word[r];
}

extern "C" int nm_beam_reorder_tokens(void* stream, const int32_t* src, const int32_t* src_row,
                                      const int32_t* word, int32_t* dst, int64_t steps, int64_t R) {
    NM_REQUIRE(src && src_row && word && dst && steps >= 0 && R > 0 && src != dst,
               "nm_beam_reorder_tokens: bad args");
    hipLaunchKernelGGL(beam_tokens_kernel, dim3(nm_cdiv(R, 256)), dim3(256), 0, nm_stream(stream), src,
                       src_row, word, dst, (int)steps, (int)R);
    NM_LAUNCH_CHECK("nm_beam_reorder_tokens");
}


// ---------------------------------------------------------------------------------------------
// Token histories of a finished beam search from back-pointers.  The reference re-gathers the whole
// [steps, B, k] token history by the surviving beams at EVERY step (beam_search_decoder.py:546-551), O(T^2)
// over a search; keeping each step's (source row, word) and walking the parents once at the end gives the same
// histories: out[t+1, r] = word[t, a_t(r)], out[0, r] = first[a_0(r)] with a_t(r) the ancestor of final row r.
// ---------------------------------------------------------------------------------------------
__global__ void beam_backtrace_kernel(const int* __restrict__ parent, const int* __restrict__ word,
                                      const int* __restrict__ first, int* __restrict__ out, int steps, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    int cur = r;
    for (int t = steps - 1; t >= 0; --t) {
        out[(long)(t + 1) * R + r] = word[(long)t * R + cur];
        cur = parent[(long)t * R + cur];
    }
    out[r] = first[cur];
}

extern "C" int nm_beam_backtrace(void* stream, const int32_t* src_row, const int32_t* word, const int32_t* first,
                                 int32_t* out, int64_t steps, int64_t R) {
    NM_REQUIRE(src_row && word && first && out && steps >= 0 && R > 0 && R < (1LL << 31), "nm_beam_backtrace: bad args");
    hipLaunchKernelGGL(beam_backtrace_kernel, dim3(nm_cdiv(R, 64)), dim3(64), 0, nm_stream(stream), src_row, word,
                       first, out, (int)steps, (int)R);
    NM_LAUNCH_CHECK("nm_beam_backtrace");
}
