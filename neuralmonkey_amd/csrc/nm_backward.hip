// Hand-written backward kernels of the attention-decoder path.  The reference
// obtains these from tf.gradients (trainers/generic_trainer.py:136-142) over
// the forward ops cited in nm_elementwise.hip / nm_attention.hip.
//
// Only the GRU state feeds back through time, so everything else (attention,
// projections, logits) is differentiated batched over all T steps; the
// sequential part is two skinny NT GEMMs + the two epilogues below per step.
#include "nm_common.h"

// ---------------------------------------------------------------------------
// dpre = dy * (1 - y^2)  in place (tanh epilogue of the output projection)
// ---------------------------------------------------------------------------
__global__ void tanh_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 d = *reinterpret_cast<float4*>(dy + i);
        const float4 v = *reinterpret_cast<const float4*>(y + i);
        d.x *= 1.0f - v.x * v.x; d.y *= 1.0f - v.y * v.y;
        d.z *= 1.0f - v.z * v.z; d.w *= 1.0f - v.w * v.w;
        *reinterpret_cast<float4*>(dy + i) = d;
    } else {
        for (long k = i; k < n; ++k) dy[k] *= 1.0f - y[k] * y[k];
    }
}

extern "C" int nm_tanh_bwd(void* stream, float* dy, const float* y, int64_t n) {
    NM_REQUIRE(dy && y && n >= 0 && nm_aligned16(dy) && nm_aligned16(y), "nm_tanh_bwd: bad args");
    if (n == 0) return NM_OK;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(nm_cdiv(n, 1024)), dim3(256), 0, nm_stream(stream), dy, y,
                       (long)n);
    NM_LAUNCH_CHECK("nm_tanh_bwd");
}

// ---------------------------------------------------------------------------
// column sums (bias gradients): out[c] (+)= sum_r x[r,c].  Deterministic two
// stage: RSPLIT row slices -> partial[RSPLIT][cols] -> fixed-order sum.
// ---------------------------------------------------------------------------
// A block is 64 columns x 4 row lanes over one slice of <= 32 rows (8 rows per thread, all
// loads independent), so even a [6400, 512] operand launches 1600 workgroups.
#define COLSUM_MAX_SPLIT 256
// One launch: every (column tile, row slice) workgroup leaves its partial sums in the workspace; the workgroup that
// arrives LAST for a column tile adds the slices in a fixed order and writes the result -- deterministic, and half
// the launches of the two-kernel version (a Transformer-base training step has 88 bias / LayerNorm gradients:
// 2 x 88 launches of ~12 us were 2.1 of its 33 ms).  Hand-off as in nm_attention.hip (CDNA4: a CU's L1 is never
// refreshed by other CUs' stores): partials stored write-through, every storing wave drains, one lane takes a ticket
// with an agent-scope atomic, the last workgroup reads the partials with L1-bypassing loads and puts the ticket back
// to zero for the next launch.
__device__ __forceinline__ void colsum_st_wt(float* p, float a) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float colsum_ld_wt(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT));
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ldx, long rows, int cols,
                                                     int nsplit, float* __restrict__ part,
                                                     unsigned* __restrict__ tickets, float* __restrict__ out,
                                                     int accumulate) {
    __shared__ float sh[4][64];
    __shared__ int s_last;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int sl = blockIdx.y;
    const long per = (rows + nsplit - 1) / nsplit;
    const long r0 = sl * per, r1 = min(rows, r0 + per);
    float s = 0.0f;
    if (c < cols)
        for (long r = r0 + ry; r < r1; r += 4) s += x[r * ldx + c];
    sh[ry][cx] = s;
    __syncthreads();
    const float mine = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
    if (nsplit == 1) {
        if (ry == 0 && c < cols) out[c] = accumulate ? out[c] + mine : mine;
        return;
    }
    if (ry == 0 && c < cols) colsum_st_wt(part + (long)sl * cols + c, mine);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(nsplit - 1));
        if (last) __hip_atomic_store(tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    float f = 0.0f;
    if (c < cols)
        for (int k = ry; k < nsplit; k += 4) f += colsum_ld_wt(part + (long)k * cols + c);
    __syncthreads();
    sh[ry][cx] = f;
    __syncthreads();
    if (ry == 0 && c < cols) {
        const float tot = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
        out[c] = accumulate ? out[c] + tot : tot;
    }
}

// The same for float4-addressable operands (16-byte aligned, ldx and cols multiples of 4 -- every bias gradient of the
// models here): a workgroup is 16 float4 column groups x 16 row lanes, four independent 16-byte loads in flight per
// thread, where the kernel above reads one float per thread and row (18.7 us for a [6400, 512] operand = 0.7 TB/s;
// 88 such launches were 1.65 of Transformer-base's 30.7 ms per step, 15 of them 0.46 of the headline step's 10.3).
// Sums in a fixed order: rows r0 + ty, + 16, ... per thread (four interleaved partial sums), the 16 row lanes in order,
// then the slices in order by the workgroup that arrives last.
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ x, long ldx, long rows, int cols,
                                                         int nsplit, float* __restrict__ part,
                                                         unsigned* __restrict__ tickets, float* __restrict__ out,
                                                         int accumulate) {
    __shared__ float sh[16][65];
    __shared__ int s_last;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c4 = blockIdx.x * 64 + 4 * tx;
    const int sl = blockIdx.y;
    const long per = (rows + nsplit - 1) / nsplit;
    const long r0 = sl * per, r1 = min(rows, r0 + per);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (c4 < cols) {
        const float* p = x + c4;
        long r = r0 + ty;
        for (; r + 48 < r1; r += 64) {
            const float4 a = *reinterpret_cast<const float4*>(p + r * ldx);
            const float4 b = *reinterpret_cast<const float4*>(p + (r + 16) * ldx);
            const float4 c = *reinterpret_cast<const float4*>(p + (r + 32) * ldx);
            const float4 d = *reinterpret_cast<const float4*>(p + (r + 48) * ldx);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
            s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
        }
        for (; r < r1; r += 16) {
            const float4 a = *reinterpret_cast<const float4*>(p + r * ldx);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    sh[ty][4 * tx] = (s0.x + s1.x) + (s2.x + s3.x);
    sh[ty][4 * tx + 1] = (s0.y + s1.y) + (s2.y + s3.y);
    sh[ty][4 * tx + 2] = (s0.z + s1.z) + (s2.z + s3.z);
    sh[ty][4 * tx + 3] = (s0.w + s1.w) + (s2.w + s3.w);
    __syncthreads();
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float mine = 0.0f;
    if (ry == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) mine += sh[k][cx];
    }
    if (nsplit == 1) {
        if (ry == 0 && c < cols) out[c] = accumulate ? out[c] + mine : mine;
        return;
    }
    if (ry == 0 && c < cols) colsum_st_wt(part + (long)sl * cols + c, mine);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(nsplit - 1));
        if (last) __hip_atomic_store(tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    float f = 0.0f;
    if (c < cols)
        for (int k = ry; k < nsplit; k += 4) f += colsum_ld_wt(part + (long)k * cols + c);
    __syncthreads();
    sh[ry][cx] = f;
    __syncthreads();
    if (ry == 0 && c < cols) {
        const float tot = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
        out[c] = accumulate ? out[c] + tot : tot;
    }
}

// The column sums of a CHAIN of operands -- out[c] (+)= sum_i sum_r x_i[r, c], ``count`` members of ``rows`` rows each,
// named by a device table of pointers (entry i at table[i * table_stride + table_offset]: the {A, B, -} table of
// nm_gemm_f32_chain serves both) -- as one launch: the bias gradients of a taped time loop, one colsum_kernel launch
// per step and bias before (307 launches of 6.5 us per training step of the general-path model at the headline size).
// Order: as colsum_vec_kernel over the concatenated rows.
__global__ __launch_bounds__(256) void colsum_chain_kernel(const float* const* __restrict__ table, int table_stride,
                                                           int table_offset, int mrows, long ldx, long rows, int cols,
                                                           int nsplit, float* __restrict__ part,
                                                           unsigned* __restrict__ tickets, float* __restrict__ out,
                                                           int accumulate) {
    __shared__ float sh[16][65];
    __shared__ int s_last;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c4 = blockIdx.x * 64 + 4 * tx;
    const int sl = blockIdx.y;
    const long per = (rows + nsplit - 1) / nsplit;
    const long r0 = sl * per, r1 = min(rows, r0 + per);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (c4 < cols) {
        auto at = [&](long r) {
            const int mem = (int)(r / mrows);
            return *reinterpret_cast<const float4*>(table[(long)mem * table_stride + table_offset] + (r - (long)mem * mrows) * ldx + c4);
        };
        long r = r0 + ty;
        for (; r + 16 < r1; r += 32) {
            const float4 a = at(r), b = at(r + 16);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (r < r1) {
            const float4 a = at(r);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    sh[ty][4 * tx] = s0.x + s1.x;
    sh[ty][4 * tx + 1] = s0.y + s1.y;
    sh[ty][4 * tx + 2] = s0.z + s1.z;
    sh[ty][4 * tx + 3] = s0.w + s1.w;
    __syncthreads();
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float mine = 0.0f;
    if (ry == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) mine += sh[k][cx];
    }
    if (nsplit == 1) {
        if (ry == 0 && c < cols) out[c] = accumulate ? out[c] + mine : mine;
        return;
    }
    if (ry == 0 && c < cols) colsum_st_wt(part + (long)sl * cols + c, mine);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(nsplit - 1));
        if (last) __hip_atomic_store(tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    float f = 0.0f;
    if (c < cols)
        for (int k = ry; k < nsplit; k += 4) f += colsum_ld_wt(part + (long)k * cols + c);
    __syncthreads();
    sh[ry][cx] = f;
    __syncthreads();
    if (ry == 0 && c < cols) {
        const float tot = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
        out[c] = accumulate ? out[c] + tot : tot;
    }
}

// partial sums [COLSUM_MAX_SPLIT][cols] + one arrival counter per 64 columns.  The counters must be ZERO when a
// launch starts (the kernel leaves them zero): allocate the workspace zero-initialised.
extern "C" int64_t nm_colsum_workspace_bytes(int64_t cols) {
    return (cols * COLSUM_MAX_SPLIT + ((cols + 63) / 64 + 3) / 4 * 4) * 4;
}

extern "C" int nm_colsum_algo(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t cols, float* out,
                              int accumulate, void* workspace, int64_t workspace_bytes, int algo);

extern "C" int nm_colsum(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t cols, float* out,
                         int accumulate, void* workspace, int64_t workspace_bytes) {
    return nm_colsum_algo(stream, x, ldx, rows, cols, out, accumulate, workspace, workspace_bytes, 0);
}

// algo 0: the float4 kernel when the operand allows it; 1: the one-float-per-thread kernel whatever the operand -- for
// launches that run on a side stream BESIDE a cluster time loop: the float4 kernel's 768 workgroups with four 16-byte
// loads in flight each took the headline training step from 10.59 to 10.73 ms although they are 35 % shorter
// themselves (profiles/r06_colsum_beside_loops.txt: the BPTT loops next to them slow down).
extern "C" int nm_colsum_algo(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t cols, float* out,
                              int accumulate, void* workspace, int64_t workspace_bytes, int algo) {
    NM_REQUIRE(x && out && workspace && rows >= 0 && cols > 0 && algo >= 0 && algo <= 1, "nm_colsum: bad args");
    NM_REQUIRE(workspace_bytes >= nm_colsum_workspace_bytes(cols), "nm_colsum: workspace too small");
    hipStream_t st = nm_stream(stream);
    float* part = reinterpret_cast<float*>(workspace);
    unsigned* tickets = reinterpret_cast<unsigned*>(part + cols * COLSUM_MAX_SPLIT);
    // enough row slices to fill the chip (cols/64 * nsplit >= ~1024 workgroups), few enough that
    // the fixed-order final pass (nsplit / 4 sequential adds per thread) stays short
    int nsplit = (int)((1024 * 64 + cols - 1) / cols);
    if (nsplit > (int)((rows + 31) / 32)) nsplit = (int)((rows + 31) / 32);
    if (nsplit > 48) nsplit = 48;
    if (nsplit < 1) nsplit = 1;
    static const bool vec_on = !(getenv("NM_COLSUM_VEC") && atoi(getenv("NM_COLSUM_VEC")) == 0);
    if (vec_on && algo == 0 && nm_aligned16(x) && ldx % 4 == 0 && cols % 4 == 0 && rows >= 64) {
        // slices of at least 64 rows (a thread's four loads in flight), enough of them for ~768 workgroups
        int ns = (int)((768 * 64 + cols - 1) / cols);
        if (ns > (int)(rows / 64)) ns = (int)(rows / 64);
        if (ns > 48) ns = 48;
        if (ns < 1) ns = 1;
        hipLaunchKernelGGL(colsum_vec_kernel, dim3(nm_cdiv(cols, 64), ns), dim3(256), 0, st, x, (long)ldx, (long)rows,
                           (int)cols, ns, part, tickets, out, accumulate);
        NM_LAUNCH_CHECK("nm_colsum");
    }
    hipLaunchKernelGGL(colsum_kernel, dim3(nm_cdiv(cols, 64), nsplit), dim3(256), 0, st, x, (long)ldx, (long)rows,
                       (int)cols, nsplit, part, tickets, out, accumulate);
    NM_LAUNCH_CHECK("nm_colsum");
}

// out[b, s, c] (+)= sum_t w_t[b, s] * d_t[b, c]: the gradient of the attended states through the context sums of a taped
// time loop, ctx_t[b, :] = sum_s w_t[b, s] states[b, s, :] (autodiff.weighted_sum).  Step by step that was one batched
// rank-1 product per step on 128x128 tiles (M = S = 50, K = 1: 40 us, 50 per training step of the general-path model =
// 2.0 ms); here the steps are the K dimension: a workgroup owns (sentence, 256 value columns), keeps every step's weights
// of the sentence in LDS and its column of every d_t in registers.  table: [count][2] device pointers {w_t, d_t}.
#define OUTER_MAX_T 64
__global__ __launch_bounds__(256) void outer_chain_kernel(const float* const* __restrict__ table, int count, int S, int C,
                                                          long ldw, long ldd, float* __restrict__ out, int accumulate) {
    extern __shared__ float wsh[];                      // [count][S]
    const int b = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < count * S; i += 256) {
        const int t = i / S, s = i - t * S;
        wsh[i] = table[2 * t][(long)b * ldw + s];
    }
    float d[OUTER_MAX_T];
#pragma unroll
    for (int t = 0; t < OUTER_MAX_T; ++t) d[t] = (t < count && c < C) ? table[2 * t + 1][(long)b * ldd + c] : 0.0f;
    __syncthreads();
    if (c >= C) return;
    float* o = out + ((long)b * S) * C + c;
    for (int s = 0; s < S; ++s) {
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < OUTER_MAX_T; ++t)
            if (t < count) acc += wsh[t * S + s] * d[t];
        o[(long)s * C] = accumulate ? o[(long)s * C] + acc : acc;
    }
}

extern "C" int nm_outer_chain(void* stream, const void* pointer_table, int64_t count, int64_t B, int64_t S, int64_t C,
                              int64_t ldw, int64_t ldd, float* out, int accumulate) {
    NM_REQUIRE(pointer_table && out && count >= 1 && count <= OUTER_MAX_T && B > 0 && S > 0 && C > 0 && ldw >= S && ldd >= C,
               "nm_outer_chain: bad arguments (count %ld of at most %d steps)", (long)count, OUTER_MAX_T);
    NM_REQUIRE(count * S * 4 <= 64 * 1024 && B < 65536, "nm_outer_chain: %ld steps x %ld positions do not fit LDS", (long)count, (long)S);
    hipLaunchKernelGGL(outer_chain_kernel, dim3(nm_cdiv(C, 256), (unsigned)B), dim3(256), (size_t)(count * S * 4),
                       nm_stream(stream), reinterpret_cast<const float* const*>(pointer_table), (int)count, (int)S, (int)C,
                       (long)ldw, (long)ldd, out, accumulate);
    NM_LAUNCH_CHECK("nm_outer_chain");
}

// workspace: that of nm_colsum for ``cols`` (zero-initialised once; the kernel leaves the counters at zero)
extern "C" int nm_colsum_chain(void* stream, const void* pointer_table, int32_t table_stride, int32_t table_offset,
                               int64_t count, int64_t rows, int64_t ldx, int64_t cols, float* out, int accumulate,
                               void* workspace, int64_t workspace_bytes) {
    NM_REQUIRE(pointer_table && out && workspace && count >= 1 && rows > 0 && cols > 0 && table_stride >= 1 &&
                   table_offset >= 0 && table_offset < table_stride, "nm_colsum_chain: bad arguments");
    NM_REQUIRE(ldx % 4 == 0 && cols % 4 == 0, "nm_colsum_chain: ld and cols must be multiples of 4 (members 16-byte aligned)");
    NM_REQUIRE(workspace_bytes >= nm_colsum_workspace_bytes(cols), "nm_colsum_chain: workspace too small");
    float* part = reinterpret_cast<float*>(workspace);
    unsigned* tickets = reinterpret_cast<unsigned*>(part + cols * COLSUM_MAX_SPLIT);
    const long total = rows * count;
    int ns = (int)((768 * 64 + cols - 1) / cols);
    if (ns > (int)(total / 32)) ns = (int)(total / 32);
    if (ns > 48) ns = 48;
    if (ns < 1) ns = 1;
    hipLaunchKernelGGL(colsum_chain_kernel, dim3(nm_cdiv(cols, 64), ns), dim3(256), 0, nm_stream(stream),
                       reinterpret_cast<const float* const*>(pointer_table), (int)table_stride, (int)table_offset, (int)rows,
                       (long)ldx, total, (int)cols, ns, part, tickets, out, accumulate);
    NM_LAUNCH_CHECK("nm_colsum_chain");
}

// ---------------------------------------------------------------------------
// embedding gradient: dtable[ids[i],:] += d[i,:]   (tf.gather gradient;
// skip_pad drops rows with id 0 = the mask multiply of model/sequence.py:191)
// ---------------------------------------------------------------------------
__global__ void embedding_scatter_kernel(float* __restrict__ dtable, long V, int E,
                                         const int* __restrict__ ids, long n, const float* __restrict__ d,
                                         long ldd, int skip_pad) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    const int id = ids[row];
    if (id < 0 || id >= V || (skip_pad && id == 0)) return;
    float* dst = dtable + (long)id * E;
    const float* src = d + row * ldd;
    for (int c = lane; c < E; c += 64) atomicAdd(dst + c, src[c]);
}

extern "C" int nm_embedding_scatter_add(void* stream, float* dtable, int64_t V, int64_t E,
                                        const int32_t* ids, int64_t n, const float* d, int64_t ldd,
                                        int skip_pad) {
    NM_REQUIRE(dtable && ids && d && V > 0 && E > 0 && n >= 0, "nm_embedding_scatter_add: bad args");
    if (n == 0) return NM_OK;
    hipLaunchKernelGGL(embedding_scatter_kernel, dim3(nm_cdiv(n, 4)), dim3(256), 0, nm_stream(stream),
                       dtable, (long)V, (int)E, ids, (long)n, d, (long)ldd, skip_pad);
    NM_LAUNCH_CHECK("nm_embedding_scatter_add");
}

// ---------------------------------------------------------------------------
// layer norm backward (tf_utils.py:189-219):
//   xhat = (x-mean)*rstd ; g = dy*gamma
//   dx   = rstd * (g - mean(g) - xhat*mean(g*xhat))
//   dyx  = dy*xhat   (column-summed by the caller into dgamma; dbeta = colsum(dy))
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(const float* __restrict__ dy,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             float* __restrict__ dx,
                                                             float* __restrict__ dyx, int D) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + row * D;
    const float* dr = dy + row * D;
    float s1 = 0.0f, s2 = 0.0f;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float xh = (xr[c] - mu) * rs;
        const float g = dr[c] * gamma[c];
        s1 += g;
        s2 += g * xh;
    }
    auto bsum = [&](float v) {
        v = nm_wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        return sh[0] + sh[1] + sh[2] + sh[3];
    };
    const float m1 = bsum(s1) / (float)D;
    const float m2 = bsum(s2) / (float)D;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float xh = (xr[c] - mu) * rs;
        const float g = dr[c] * gamma[c];
        dx[row * D + c] = rs * (g - m1 - xh * m2);
        dyx[row * D + c] = dr[c] * xh;
    }
}

extern "C" int nm_layer_norm_bwd(void* stream, const float* dy, const float* x, const float* mean,
                                 const float* rstd, const float* gamma, float* dx, float* dyx,
                                 int64_t rows, int64_t D) {
    NM_REQUIRE(dy && x && mean && rstd && gamma && dx && dyx && rows >= 0 && D > 0,
               "nm_layer_norm_bwd: bad args");
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, nm_stream(stream), dy, x,
                       mean, rstd, gamma, dx, dyx, (int)D);
    NM_LAUNCH_CHECK("nm_layer_norm_bwd");
}

// Layer norm backward with its parameter gradients (round 6).  nm_layer_norm_bwd leaves dy * xhat behind and the
// caller column-sums it and dy for dgamma / dbeta: three launches and a [rows, D] round trip through HBM per layer
// norm -- 64 of the 88 column sums of a Transformer-base training step (1.65 ms of 30.7,
// profiles/r05_transformer_train_kernel_stats.csv).  Here a workgroup walks rows blockIdx.x, + gridDim.x, ... : dx as
// above, and every lane keeps the running column sums of its own columns in registers; the workgroups' partial sums
// [G][2][D] are added by a second, tiny launch in a fixed order (deterministic; G <= 256).  (A first version with one
// 256-thread workgroup per row group and two workgroup reductions per row was SLOWER than the three launches it
// replaced -- Transformer-base 32.6 against 30.7 ms per step: 25 rows one after the other, four barriers each.)
#define LNB_MAX_CH 8                   // 256-column chunks per row: D <= 2048
// One WAVE per row (a lane owns 4 consecutive columns of every 256-column chunk: 16-byte loads, the two row sums are
// DPP wave reductions, no workgroup barrier inside the row loop); a workgroup's four waves walk rows
// 4 blockIdx.x + wave, + 4 gridDim.x, ... and add their column sums through LDS at the end.
template <int NCH, bool ACCDX>
__global__ __launch_bounds__(256) void layer_norm_bwd_params_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd,
                                                                    const float* __restrict__ gamma, float* __restrict__ dx,
                                                                    float* __restrict__ part, long rows, int D) {
    __shared__ float4 sh[3][NCH][2][64];           // waves 1..3: [chunk][dgamma | dbeta][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 sg[NCH], sb[NCH], gm[NCH];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        sg[c] = z4; sb[c] = z4;
        const int col = 256 * c + 4 * lane;
        gm[c] = col < D ? *reinterpret_cast<const float4*>(gamma + col) : z4;
    }
    const float invd = 1.0f / (float)D;
    // RPI rows per trip: their loads are all in flight before the first reduction (a wave that walks its rows one
    // after the other pays a memory round trip per row: measured slower than the three launches this replaces)
    constexpr int RPI = NCH <= 2 ? 4 : (NCH <= 4 ? 2 : 1);
    const long stride = (long)gridDim.x * 4;
    for (long row0 = (long)blockIdx.x * 4 + wave; row0 < rows; row0 += stride * RPI) {
        float4 xh[RPI][NCH], d[RPI][NCH], was[ACCDX ? RPI : 1][ACCDX ? NCH : 1];     // was: what dx holds (ACCDX: dx += ...)
        float mu[RPI], rs[RPI];
#pragma unroll
        for (int r = 0; r < RPI; ++r) {
            const long row = row0 + r * stride;
            const bool live = row < rows;
            mu[r] = live ? mean[row] : 0.0f;
            rs[r] = live ? rstd[row] : 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = 256 * c + 4 * lane;
                xh[r][c] = z4; d[r][c] = z4;
                if (live && col < D) {
                    xh[r][c] = *reinterpret_cast<const float4*>(x + row * D + col);
                    d[r][c] = *reinterpret_cast<const float4*>(dy + row * D + col);
                    if constexpr (ACCDX) was[r][c] = *reinterpret_cast<const float4*>(dx + row * D + col);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPI; ++r) {
            const long row = row0 + r * stride;
            if (row >= rows) break;
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const float4 xv = xh[r][c];
                xh[r][c] = make_float4((xv.x - mu[r]) * rs[r], (xv.y - mu[r]) * rs[r], (xv.z - mu[r]) * rs[r],
                                       (xv.w - mu[r]) * rs[r]);
                if (256 * c + 4 * lane >= D) xh[r][c] = z4;
                const float4 g = make_float4(d[r][c].x * gm[c].x, d[r][c].y * gm[c].y, d[r][c].z * gm[c].z,
                                             d[r][c].w * gm[c].w);
                s1 += (g.x + g.y) + (g.z + g.w);
                s2 += (g.x * xh[r][c].x + g.y * xh[r][c].y) + (g.z * xh[r][c].z + g.w * xh[r][c].w);
            }
            const float m1 = nm_wave_sum_dpp(s1) * invd, m2 = nm_wave_sum_dpp(s2) * invd;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int col = 256 * c + 4 * lane;
                if (col < D) {
                    const float4 dd = d[r][c], hh = xh[r][c];
                    float4 o;
                    o.x = rs[r] * (dd.x * gm[c].x - m1 - hh.x * m2);
                    o.y = rs[r] * (dd.y * gm[c].y - m1 - hh.y * m2);
                    o.z = rs[r] * (dd.z * gm[c].z - m1 - hh.z * m2);
                    o.w = rs[r] * (dd.w * gm[c].w - m1 - hh.w * m2);
                    if constexpr (ACCDX) { o.x += was[r][c].x; o.y += was[r][c].y; o.z += was[r][c].z; o.w += was[r][c].w; }
                    *reinterpret_cast<float4*>(dx + row * D + col) = o;
                    sg[c].x += dd.x * hh.x; sg[c].y += dd.y * hh.y; sg[c].z += dd.z * hh.z; sg[c].w += dd.w * hh.w;
                    sb[c].x += dd.x; sb[c].y += dd.y; sb[c].z += dd.z; sb[c].w += dd.w;
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) { sh[wave - 1][c][0][lane] = sg[c]; sh[wave - 1][c][1][lane] = sb[c]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = 256 * c + 4 * lane;
            if (col >= D) continue;
            float4 a = sg[c], b = sb[c];
#pragma unroll
            for (int w = 0; w < 3; ++w) {                       // waves in order: a fixed sum
                const float4 pa = sh[w][c][0][lane], pb = sh[w][c][1][lane];
                a.x += pa.x; a.y += pa.y; a.z += pa.z; a.w += pa.w;
                b.x += pb.x; b.y += pb.y; b.z += pb.z; b.w += pb.w;
            }
            *reinterpret_cast<float4*>(part + ((long)blockIdx.x * 2) * D + col) = a;
            *reinterpret_cast<float4*>(part + ((long)blockIdx.x * 2 + 1) * D + col) = b;
        }
    }
}

// 32 columns x 8 row groups per workgroup: a thread adds every eighth partial row of its column (four loads in flight),
// the eight groups meet in LDS in a fixed order.  (One thread per column walking all 256 partial rows took 60 us: a
// chain of dependent loads on four workgroups.)
__global__ __launch_bounds__(256) void layer_norm_bwd_reduce_kernel(const float* __restrict__ part, int G, int D,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                    int accumulate) {
    __shared__ float sh[8][32];
    const int cx = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    const bool ok = c < 2 * D;
    const int which = ok ? c / D : 0, col = ok ? c - which * D : 0;
    const float* p = part + (long)which * D + col;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int g = q;
    for (; g + 24 < G; g += 32) {
        s0 += p[(long)g * 2 * D];
        s1 += p[(long)(g + 8) * 2 * D];
        s2 += p[(long)(g + 16) * 2 * D];
        s3 += p[(long)(g + 24) * 2 * D];
    }
    for (; g < G; g += 8) s0 += p[(long)g * 2 * D];
    sh[q][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && ok) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += sh[k][cx];
        float* out = which ? dbeta : dgamma;
        out[col] = accumulate ? out[col] + s : s;
    }
}

extern "C" int64_t nm_layer_norm_bwd_params_workspace_bytes(int64_t D) { return 256 * 2 * D * 4; }

extern "C" int nm_layer_norm_bwd_params(void* stream, const float* dy, const float* x, const float* mean,
                                        const float* rstd, const float* gamma, float* dx, int64_t rows, int64_t D,
                                        float* dgamma, float* dbeta, int accumulate, void* workspace,
                                        int64_t workspace_bytes) {
    NM_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && workspace && rows >= 0 && D > 0,
               "nm_layer_norm_bwd_params: bad args");
    NM_REQUIRE(D <= 256 * LNB_MAX_CH && D % 4 == 0, "nm_layer_norm_bwd_params: D = %ld (a multiple of 4, at most %d)",
               (long)D, 256 * LNB_MAX_CH);
    NM_REQUIRE(nm_aligned16(dy) && nm_aligned16(x) && nm_aligned16(dx) && nm_aligned16(gamma) && nm_aligned16(workspace),
               "nm_layer_norm_bwd_params: operands must be 16-byte aligned");
    NM_REQUIRE(workspace_bytes >= nm_layer_norm_bwd_params_workspace_bytes(D), "nm_layer_norm_bwd_params: workspace too small");
    if (rows == 0) return NM_OK;
    const int G = (int)((rows + 3) / 4 < 256 ? (rows + 3) / 4 : 256);
    hipStream_t st = nm_stream(stream);
    float* part = reinterpret_cast<float*>(workspace);
    const int nch = (int)((D + 255) / 256);
    const bool acc_dx = (accumulate & 2) != 0;
    accumulate &= 1;
#define NM_LNB(N_)                                                                                                      \
    do {                                                                                                                \
        if (acc_dx) hipLaunchKernelGGL((layer_norm_bwd_params_kernel<N_, true>), dim3(G), dim3(256), 0, st, dy, x, mean, \
                                       rstd, gamma, dx, part, (long)rows, (int)D);                                      \
        else hipLaunchKernelGGL((layer_norm_bwd_params_kernel<N_, false>), dim3(G), dim3(256), 0, st, dy, x, mean, rstd, \
                                gamma, dx, part, (long)rows, (int)D);                                                   \
    } while (0)
    if (nch <= 1) NM_LNB(1);
    else if (nch == 2) NM_LNB(2);
    else if (nch <= 4) NM_LNB(4);
    else NM_LNB(8);
#undef NM_LNB
    hipLaunchKernelGGL(layer_norm_bwd_reduce_kernel, dim3((unsigned)((2 * D + 31) / 32)), dim3(256), 0, st, part, G,
                       (int)D, dgamma, dbeta, accumulate);
    NM_LAUNCH_CHECK("nm_layer_norm_bwd_params");
}

// ---------------------------------------------------------------------------
// GRU step backward epilogues (see nm_elementwise.hip for the forward split).
//   h' = u*h + (1-u)*c ; c = tanh(xc + (r*h).Wc_h) ; [r,u] = sigmoid(xg + h.Wg_h)
// Step t, given dh = dL/dh' (carried) + dout (gradient arriving at this
// step's output):
//   blend_bwd : dc_pre, du_pre, dh <- dh*u          (then drh = dc_pre . Wc_h^T by GEMM)
//   gates_bwd : dr_pre, dh += drh*r                  (then dh += [dr_pre,du_pre] . Wg_h^T by GEMM)
// The pre-activation gradients are written both to dense [ndir,R,*] GEMM
// operands and, at the step's sequence position, into dxp (same layout as xp)
// from which the weight / input gradients are formed after the loop.
// h_prev(t) is h0 (or zero) at t == 0, else the sequence output one step back
// along the direction of travel.
// ---------------------------------------------------------------------------
struct GruBwdArgs {
    float* dh;                 // [ndir,R,H] in/out
    const float* dout;         // sequence-addressed, may be null
    long do_dir, do_row, do_time;
    const float* ru;           // [ndir,R,2H] gates of step t
    const float* c;            // [ndir,R,H]  candidate of step t
    const float* h0;           // [ndir,R,H] or null (zeros)
    const float* hseq;         // sequence outputs (h of every step)
    long hs_dir, hs_row, hs_time;
    float* dxp;                // sequence-addressed [.., 3H]
    long dx_dir, dx_row, dx_time;
    float* dgpre;              // [ndir,R,2H]
    float* dcpre;              // [ndir,R,H]
    const float* drh;          // [ndir,R,H]   (gates_bwd only)
    const int* lengths;
    int t, rev_mask;
    long R;
    int H;
};

__device__ __forceinline__ bool gru_bwd_pos(const GruBwdArgs& a, int r, int d, int& pos, int& ppos,
                                            bool& first) {
    pos = a.t;
    const bool rev = (a.rev_mask >> d) & 1;
    if (a.lengths) {
        const int len = a.lengths[r];
        if (a.t >= len) return false;
        if (rev) pos = len - 1 - a.t;
    }
    ppos = rev ? pos + 1 : pos - 1;
    first = (a.t == 0);
    return true;
}

__device__ __forceinline__ float4 gru_hprev(const GruBwdArgs& a, long ro, int d, long r, int ppos,
                                            bool first, int j) {
    if (first) {
        if (a.h0) return *reinterpret_cast<const float4*>(a.h0 + ro * a.H + j);
        return make_float4(0, 0, 0, 0);
    }
    return *reinterpret_cast<const float4*>(a.hseq + d * a.hs_dir + r * a.hs_row + (long)ppos * a.hs_time + j);
}

__global__ void gru_blend_bwd_kernel(GruBwdArgs a) {
    const int d = blockIdx.z;
    const long r = blockIdx.y;
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (j >= a.H) return;
    const long ro = (long)d * a.R + r;
    int pos, ppos;
    bool first;
    const bool live = gru_bwd_pos(a, (int)r, d, pos, ppos, first);
    const float4 z = make_float4(0, 0, 0, 0);
    if (!live) {                       // state was copied through: dh unchanged, no parameter gradient
        *reinterpret_cast<float4*>(a.dcpre + ro * a.H + j) = z;
        *reinterpret_cast<float4*>(a.dgpre + ro * 2 * a.H + a.H + j) = z;
        return;
    }
    float4 dh = *reinterpret_cast<float4*>(a.dh + ro * a.H + j);
    if (a.dout) {
        const float4 o = *reinterpret_cast<const float4*>(a.dout + d * a.do_dir + r * a.do_row +
                                                          (long)pos * a.do_time + j);
        dh.x += o.x; dh.y += o.y; dh.z += o.z; dh.w += o.w;
    }
    const float4 u = *reinterpret_cast<const float4*>(a.ru + ro * 2 * a.H + a.H + j);
    const float4 c = *reinterpret_cast<const float4*>(a.c + ro * a.H + j);
    const float4 hp = gru_hprev(a, ro, d, r, ppos, first, j);
    float4 dcp, dup, dhd;
#define NM_BL(f)                                           \
    {                                                      \
        const float dc = dh.f * (1.0f - u.f);              \
        const float du = dh.f * (hp.f - c.f);              \
        dcp.f = dc * (1.0f - c.f * c.f);                   \
        dup.f = du * u.f * (1.0f - u.f);                   \
        dhd.f = dh.f * u.f;                                \
    }
    NM_BL(x) NM_BL(y) NM_BL(z) NM_BL(w)
#undef NM_BL
    *reinterpret_cast<float4*>(a.dh + ro * a.H + j) = dhd;
    *reinterpret_cast<float4*>(a.dcpre + ro * a.H + j) = dcp;
    *reinterpret_cast<float4*>(a.dgpre + ro * 2 * a.H + a.H + j) = dup;
    float* dx = a.dxp + d * a.dx_dir + r * a.dx_row + (long)pos * a.dx_time;
    *reinterpret_cast<float4*>(dx + a.H + j) = dup;
    *reinterpret_cast<float4*>(dx + 2 * a.H + j) = dcp;
}

__global__ void gru_gates_bwd_kernel(GruBwdArgs a) {
    const int d = blockIdx.z;
    const long r = blockIdx.y;
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (j >= a.H) return;
    const long ro = (long)d * a.R + r;
    int pos, ppos;
    bool first;
    const bool live = gru_bwd_pos(a, (int)r, d, pos, ppos, first);
    if (!live) {
        *reinterpret_cast<float4*>(a.dgpre + ro * 2 * a.H + j) = make_float4(0, 0, 0, 0);
        return;
    }
    const float4 rr = *reinterpret_cast<const float4*>(a.ru + ro * 2 * a.H + j);
    const float4 hp = gru_hprev(a, ro, d, r, ppos, first, j);
    const float4 drh = *reinterpret_cast<const float4*>(a.drh + ro * a.H + j);
    float4 dh = *reinterpret_cast<float4*>(a.dh + ro * a.H + j);
    float4 drp;
#define NM_GT(f)                                           \
    {                                                      \
        const float dr = drh.f * hp.f;                     \
        drp.f = dr * rr.f * (1.0f - rr.f);                 \
        dh.f += drh.f * rr.f;                              \
    }
    NM_GT(x) NM_GT(y) NM_GT(z) NM_GT(w)
#undef NM_GT
    *reinterpret_cast<float4*>(a.dh + ro * a.H + j) = dh;
    *reinterpret_cast<float4*>(a.dgpre + ro * 2 * a.H + j) = drp;
    float* dx = a.dxp + d * a.dx_dir + r * a.dx_row + (long)pos * a.dx_time;
    *reinterpret_cast<float4*>(dx + j) = drp;
}

extern "C" int nm_gru_step_bwd(void* stream, int phase, float* dh, const float* dout, int64_t do_dir,
                               int64_t do_row, int64_t do_time, const float* ru, const float* c,
                               const float* h0, const float* hseq, int64_t hs_dir, int64_t hs_row,
                               int64_t hs_time, float* dxp, int64_t dx_dir, int64_t dx_row,
                               int64_t dx_time, float* dgpre, float* dcpre, const float* drh,
                               const int32_t* lengths, int t, int rev_mask, int ndir, int64_t R,
                               int64_t H) {
    NM_REQUIRE(dh && ru && hseq && dxp && dgpre, "nm_gru_step_bwd: null pointer");
    NM_REQUIRE(phase == 0 || phase == 1, "nm_gru_step_bwd: phase must be 0 (blend) or 1 (gates)");
    NM_REQUIRE((phase == 0 && c && dcpre) || (phase == 1 && drh), "nm_gru_step_bwd: missing operand");
    NM_REQUIRE(H > 0 && H % 4 == 0 && R > 0 && ndir >= 1 && ndir <= 2, "nm_gru_step_bwd: bad shape");
    NM_REQUIRE(do_dir % 4 == 0 && do_row % 4 == 0 && do_time % 4 == 0 && hs_dir % 4 == 0 &&
                   hs_row % 4 == 0 && hs_time % 4 == 0 && dx_dir % 4 == 0 && dx_row % 4 == 0 &&
                   dx_time % 4 == 0,
               "nm_gru_step_bwd: strides must be multiples of 4");
    GruBwdArgs a;
    a.dh = dh; a.dout = dout; a.do_dir = do_dir; a.do_row = do_row; a.do_time = do_time;
    a.ru = ru; a.c = c; a.h0 = h0; a.hseq = hseq; a.hs_dir = hs_dir; a.hs_row = hs_row; a.hs_time = hs_time;
    a.dxp = dxp; a.dx_dir = dx_dir; a.dx_row = dx_row; a.dx_time = dx_time;
    a.dgpre = dgpre; a.dcpre = dcpre; a.drh = drh; a.lengths = lengths; a.t = t; a.rev_mask = rev_mask;
    a.R = R; a.H = (int)H;
    const int tpb = 128;
    dim3 grid(nm_cdiv(H, 4 * tpb), (unsigned)R, ndir);
    if (phase == 0) hipLaunchKernelGGL(gru_blend_bwd_kernel, grid, dim3(tpb), 0, nm_stream(stream), a);
    else hipLaunchKernelGGL(gru_gates_bwd_kernel, grid, dim3(tpb), 0, nm_stream(stream), a);
    NM_LAUNCH_CHECK("nm_gru_step_bwd");
}

// h_prev of every sequence position as a dense [B,S,ndir,H] tensor (operand of
// the recurrent-kernel weight gradient h_prev^T . dpre): position p of
// direction d holds the output at p-1 (forward) / p+1 (reversed), zero at the
// start of travel and at dead positions.
__global__ void gru_seq_shift_kernel(const float* __restrict__ seq, float* __restrict__ out,
                                     const int* __restrict__ lengths, int rev_mask, int S, int ndir, int H) {
    const int b = blockIdx.z, p = blockIdx.y;
    const int col = (blockIdx.x * blockDim.x + threadIdx.x) * 4;   // over ndir*H
    if (col >= ndir * H) return;
    const int d = col / H;
    const int len = lengths ? lengths[b] : S;
    const bool rev = (rev_mask >> d) & 1;
    const int pp = rev ? p + 1 : p - 1;
    float4 v = make_float4(0, 0, 0, 0);
    if (p < len && pp >= 0 && pp < len)
        v = *reinterpret_cast<const float4*>(seq + ((long)b * S + pp) * ndir * H + col);
    *reinterpret_cast<float4*>(out + ((long)b * S + p) * ndir * H + col) = v;
}

extern "C" int nm_gru_seq_shift(void* stream, const float* seq, float* out, const int32_t* lengths,
                                int rev_mask, int64_t B, int64_t S, int ndir, int64_t H) {
    NM_REQUIRE(seq && out && B > 0 && S > 0 && H % 4 == 0, "nm_gru_seq_shift: bad args");
    hipLaunchKernelGGL(gru_seq_shift_kernel, dim3(nm_cdiv(ndir * H, 1024), (unsigned)S, (unsigned)B),
                       dim3(256), 0, nm_stream(stream), seq, out, lengths, rev_mask, (int)S, ndir, (int)H);
    NM_LAUNCH_CHECK("nm_gru_seq_shift");
}

// ---------------------------------------------------------------------------
// attention backward, batched over all T decoder steps (rows = (t,b)).
// (1) softmax + mask-renorm backward (attention/feed_forward.py:139-144):
//     p = softmax(e); N = sum(p*m) + 1e-8; w = p*m/N
//     dp = m/N * (dw - sum(dw*w)) ; de = p * (dp - sum(dp*p))
// One wave per row.
// ---------------------------------------------------------------------------
__global__ void attn_softmax_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ e,
                                        const float* __restrict__ mask, float* __restrict__ de,
                                        long rows, int B, int S) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = (int)(row % B);
    const float* er = e + row * S;
    const float* dwr = dw + row * S;
    const float* mr = mask ? mask + (long)b * S : nullptr;
    float mx = -INFINITY;
    for (int s = lane; s < S; s += 64) mx = fmaxf(mx, er[s]);
    mx = nm_wave_max(mx);
    float se = 0.0f, sm = 0.0f;
    for (int s = lane; s < S; s += 64) {
        const float x = expf(er[s] - mx);
        se += x;
        sm += x * (mr ? mr[s] : 1.0f);
    }
    se = nm_wave_sum(se);
    sm = nm_wave_sum(sm);
    const float inv_se = 1.0f / se;
    const float N = sm * inv_se + 1e-8f;
    const float invN = 1.0f / N;
    float sdw = 0.0f;
    for (int s = lane; s < S; s += 64) {
        const float p = expf(er[s] - mx) * inv_se;
        const float w = p * (mr ? mr[s] : 1.0f) * invN;
        sdw += dwr[s] * w;
    }
    sdw = nm_wave_sum(sdw);
    float sdp = 0.0f;
    for (int s = lane; s < S; s += 64) {
        const float p = expf(er[s] - mx) * inv_se;
        const float dp = (mr ? mr[s] : 1.0f) * invN * (dwr[s] - sdw);
        sdp += dp * p;
    }
    sdp = nm_wave_sum(sdp);
    for (int s = lane; s < S; s += 64) {
        const float p = expf(er[s] - mx) * inv_se;
        const float dp = (mr ? mr[s] : 1.0f) * invN * (dwr[s] - sdw);
        de[row * S + s] = p * (dp - sdp);
    }
}

extern "C" int nm_attn_softmax_bwd(void* stream, const float* dw, const float* e, const float* mask,
                                   float* de, int64_t rows, int64_t B, int64_t S) {
    NM_REQUIRE(dw && e && de && rows >= 0 && B > 0 && S > 0, "nm_attn_softmax_bwd: bad args");
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3(nm_cdiv(rows, 4)), dim3(256), 0, nm_stream(stream), dw,
                       e, mask, de, (long)rows, (int)B, (int)S);
    NM_LAUNCH_CHECK("nm_attn_softmax_bwd");
}

// The same distribution computed forward from ready-made energies: softmax over all S, mask,
// renormalise with +1e-8 (feed_forward.py:139-144, combination.py:301-307).  Used where the energies
// are assembled from several sources (FlatMultiAttention: encoders + sentinel), one wave per row;
// the mask row of query row r is (r / rows_per_key) % B.
__global__ void attn_softmax_fwd_kernel(const float* __restrict__ e, const float* __restrict__ mask,
                                        float* __restrict__ w, long rows, int B, int S, int rpk) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = (int)((row / rpk) % B);
    const float* er = e + row * S;
    const float* mr = mask ? mask + (long)b * S : nullptr;
    float mx = -INFINITY;
    for (int s = lane; s < S; s += 64) mx = fmaxf(mx, er[s]);
    mx = nm_wave_max(mx);
    float se = 0.0f, sm = 0.0f;
    for (int s = lane; s < S; s += 64) {
        const float x = expf(er[s] - mx);
        se += x;
        sm += x * (mr ? mr[s] : 1.0f);
    }
    se = nm_wave_sum(se);
    sm = nm_wave_sum(sm);
    const float inv_se = 1.0f / se;
    const float invN = 1.0f / (sm * inv_se + 1e-8f);
    for (int s = lane; s < S; s += 64)
        w[row * S + s] = expf(er[s] - mx) * inv_se * (mr ? mr[s] : 1.0f) * invN;
}

extern "C" int nm_attn_softmax_fwd(void* stream, const float* e, const float* mask, float* w, int64_t rows,
                                   int64_t B, int64_t S, int64_t rows_per_key) {
    NM_REQUIRE(e && w && rows >= 0 && B > 0 && S > 0 && rows_per_key >= 1, "nm_attn_softmax_fwd: bad args");
    if (rows == 0) return NM_OK;
    hipLaunchKernelGGL(attn_softmax_fwd_kernel, dim3(nm_cdiv(rows, 4)), dim3(256), 0, nm_stream(stream), e, mask,
                       w, (long)rows, (int)B, (int)S, (int)rows_per_key);
    NM_LAUNCH_CHECK("nm_attn_softmax_fwd");
}

// (2) energies backward (feed_forward.py:120-123), tanh recomputed ONCE per (t, b, s, a):
//     z = tanh(hf[b,s,a] + y[t,b,a]) ; g = de[t,b,s] * (1 - z^2)
//     dhf[b,s,a] = v[a] * sum_t g        dvp[(b,s),a] = sum_t de * z
//     dy[t,b,a]  = v[a] * sum_s g
// One block owns (sentence b, 128 feature columns); a thread owns one column a.  Positions go
// through registers SCH at a time: hf[s0..s0+SCH) and the two position-indexed accumulators stay in
// VGPRs while the loop runs over all T queries; the query-indexed sum dy[t] is carried through global
// memory between position chunks (the block is the only writer of its (t, b, a) elements, so plain
// read-modify-write).  de[t,b,s] is block-uniform: scalar loads.  T*B*S*A tanh in total (328 M at
// the benchmark shape) against twice that in the two-kernel version this replaces
// (attn_dhf + attn_dy: 0.37 + 1.50 ms per training step on MI355X).
template <int SCH>
__global__ __launch_bounds__(128) void attn_energy_bwd_kernel(
    const float* __restrict__ de, const float* __restrict__ hf, const float* __restrict__ y,
    const float* __restrict__ v, float* __restrict__ dhf, float* __restrict__ dvp, float* __restrict__ dy,
    int T, int B, int S, int A, int accumulate) {
    const int a = blockIdx.x * 128 + threadIdx.x;
    const int b = blockIdx.y;
    if (a >= A) return;
    const float va = v[a];
    for (int s0 = 0; s0 < S; s0 += SCH) {
        // every key element h[i] meets all T queries: exp(2 h) once per element, exp(2 y) once per query, one
        // v_rcp_f32 per tanh (nm_tanh_prod); beyond that form's exact range (|.| > NM_EXP2X_MAX) the thread takes
        // nm_tanh on the sums for the affected queries
        float h[SCH], eh[SCH], g_acc[SCH], z_acc[SCH];
        float hmax = 0.0f;
#pragma unroll
        for (int i = 0; i < SCH; ++i) {
            h[i] = (s0 + i < S) ? hf[((long)b * S + s0 + i) * A + a] : 0.0f;
            eh[i] = nm_exp2x(h[i]);
            hmax = fmaxf(hmax, fabsf(h[i]));
            g_acc[i] = 0.0f;
            z_acc[i] = 0.0f;
        }
        float yy_next = y[(long)b * A + a];                   // query slice of t = 0; the next one is requested a step ahead
        for (int t = 0; t < T; ++t) {
            const long row = (long)t * B + b;
            const float yy = yy_next;
            if (t + 1 < T) yy_next = y[(row + B) * A + a];
            const float* der = de + row * S + s0;
            float dsum = 0.0f;
            // (two loops, not a select per element: hipcc if-converts `exact ? nm_tanh : nm_tanh_prod` and then
            // evaluates BOTH -- three transcendentals per element, measured 0.49 ms against 0.39 before the change)
            if (fmaxf(hmax, fabsf(yy)) > NM_EXP2X_MAX) {
#pragma unroll
                for (int i = 0; i < SCH; ++i) {
                    const float d = (s0 + i < S) ? der[i] : 0.0f;
                    const float z = nm_tanh(h[i] + yy);
                    const float g = d * (1.0f - z * z);
                    g_acc[i] += g;
                    z_acc[i] += d * z;
                    dsum += g;
                }
            } else {
                const float ey = nm_exp2x(yy);
#pragma unroll
                for (int i = 0; i < SCH; ++i) {
                    const float d = (s0 + i < S) ? der[i] : 0.0f;
                    const float z = nm_tanh_prod(eh[i], ey);
                    const float g = d * (1.0f - z * z);
                    g_acc[i] += g;
                    z_acc[i] += d * z;
                    dsum += g;
                }
            }
            float* dyp = dy + row * A + a;
            *dyp = (s0 == 0) ? va * dsum : *dyp + va * dsum;
        }
        if (!dhf) continue;             // query gradients only (a taped step: the key-side sums come later, over all steps)
#pragma unroll
        for (int i = 0; i < SCH; ++i) {
            if (s0 + i >= S) break;
            const long o = ((long)b * S + s0 + i) * A + a;
            if (accumulate) {
                dhf[o] += va * g_acc[i];
                dvp[o] += z_acc[i];
            } else {
                dhf[o] = va * g_acc[i];
                dvp[o] = z_acc[i];
            }
        }
    }
}

extern "C" int nm_attn_energy_bwd(void* stream, const float* de, const float* hf, const float* y,
                                  const float* v, float* dhf, float* dv_partial, float* dy, int64_t T,
                                  int64_t B, int64_t S, int64_t A, int accumulate) {
    NM_REQUIRE(de && hf && y && v && dy && ((dhf == nullptr) == (dv_partial == nullptr)), "nm_attn_energy_bwd: null pointer");
    NM_REQUIRE(T > 0 && B > 0 && S > 0 && A > 0 && S < 65536 && B < 65536 && T < 65536,
               "nm_attn_energy_bwd: bad shape");
    hipStream_t st = nm_stream(stream);
    const dim3 grid(nm_cdiv(A, 128), (unsigned)B);
    // positions per register chunk: the smallest chunk that covers S in ceil(S/16) passes (S=50 -> 4 x 13)
    // ... and up to 32 positions per pass while two waves per SIMD still fit their registers (<= 176 VGPRs, no
    // scratch; 40 and more spill): S = 50 runs 2 x 25 instead of 4 x 13 -- every query row y[t] is loaded twice
    // instead of four times, dy[t] is read-modified-written once instead of three times, and 25 independent tanh
    // per query hide the next query's load
    const bool wide_off = nm_cur()->sw.aeb_wide_off;                    // NM_AEB_WIDE=0
    int sch;
    if (S <= 8) sch = 8;
    else if (wide_off || S <= 16) sch = (int)nm_cdiv(S, nm_cdiv(S, 16));
    else {
        const int want = (int)nm_cdiv(S, nm_cdiv(S, 32));
        sch = want <= 20 ? 20 : want <= 24 ? 24 : want <= 26 ? 26 : want <= 28 ? 28 : 32;
    }
#define NM_AEB(SCH_)                                                                                       \
    case SCH_:                                                                                             \
        hipLaunchKernelGGL(attn_energy_bwd_kernel<SCH_>, grid, dim3(128), 0, st, de, hf, y, v, dhf,        \
                           dv_partial, dy, (int)T, (int)B, (int)S, (int)A, accumulate);                    \
        break;
    switch (sch) {
        NM_AEB(8) NM_AEB(9) NM_AEB(10) NM_AEB(11) NM_AEB(12) NM_AEB(13) NM_AEB(14) NM_AEB(15)
        NM_AEB(20) NM_AEB(24) NM_AEB(26) NM_AEB(28) NM_AEB(32)
        default:
            hipLaunchKernelGGL(attn_energy_bwd_kernel<16>, grid, dim3(128), 0, st, de, hf, y, v, dhf, dv_partial,
                               dy, (int)T, (int)B, (int)S, (int)A, accumulate);
    }
#undef NM_AEB
    NM_LAUNCH_CHECK("nm_attn_energy_bwd");
}

// ---------------------------------------------------------------------------
// ONE step's attention backward up to the query (a taped decoder step: the cells around the attention differ per
// configuration, so the step's gradient has to come back through the attention before the previous step can start):
//     dw[s] = <dctx[b,:], states[b,s,:]>          (the context sum's weights, feed_forward.py:146-149)
//     de    = softmax/renorm backward of dw       (the arithmetic of attn_softmax_bwd_kernel, feed_forward.py:139-144)
//     dy[a] = v[a] * sum_s de[s] (1 - tanh^2(hf[b,s,a] + y[b,a]))        (feed_forward.py:120-123)
// -- three launches (a batched M = 1 product, the softmax kernel, the energies kernel in its query-only mode: 11 + 5 +
// 13 us at B = 64, S = 50) in one: a workgroup owns a sentence, its waves share the S dot products, wave 0 does the
// softmax part on the S values in LDS, then a thread owns feature columns.  de goes out too: the key-side sums over all
// the steps are taken later from the stacked de (nm_attn_energy_bwd over T steps).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void attn_step_bwd_kernel(
    const float* __restrict__ dctx, long ldd, const float* __restrict__ states, const float* __restrict__ e,
    const float* __restrict__ mask, const float* __restrict__ hf, const float* __restrict__ y, long ldy,
    const float* __restrict__ v, float* __restrict__ de, float* __restrict__ dy, long lddy, int S, int C, int A) {
    extern __shared__ float sh[];                 // dw[S], then de[S] in place; pt[blockDim]: partial sums of dy
    float* pt = sh + S;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    // (1) the S dot products: a wave takes four positions at a time (sixteen 16-byte loads in flight per lane at
    // C = 1024 -- a sentence's 200 KB of states are fetched by ONE compute unit, latency is what there is to hide)
    const float* dc = dctx + (long)b * ldd;
    for (int s0 = wave; s0 < S; s0 += 4 * nw) {
        const float* sr[4];
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 4; ++i) sr[i] = states + ((long)b * S + min(s0 + i * nw, S - 1)) * C;
        // (1024 columns per pass, every load of a pass requested before the first is used: clamped addresses and zeroed
        // terms instead of an early exit, which would keep the compiler from hoisting the loads)
        for (int c0 = 0; c0 < C; c0 += 1024) {
            float4 d[4], x[4][4];
            bool ok[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j * 256 + lane * 4;
                ok[j] = c < C;
                const int cc = ok[j] ? c : 0;
                d[j] = *reinterpret_cast<const float4*>(dc + cc);
#pragma unroll
                for (int i = 0; i < 4; ++i) x[j][i] = *reinterpret_cast<const float4*>(sr[i] + cc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!ok[j]) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] += x[j][i].x * d[j].x + x[j][i].y * d[j].y + x[j][i].z * d[j].z + x[j][i].w * d[j].w;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = nm_wave_sum(acc[i]);
            if (lane == 0 && s0 + i * nw < S) sh[s0 + i * nw] = t;
        }
    }
    __syncthreads();
    // (2) softmax + mask + renormalisation backward on the S values (attn_softmax_bwd_kernel's arithmetic)
    if (wave == 0) {
        const float* er = e + (long)b * S;
        const float* mr = mask ? mask + (long)b * S : nullptr;
        float mx = -INFINITY;
        for (int s = lane; s < S; s += 64) mx = fmaxf(mx, er[s]);
        mx = nm_wave_max(mx);
        float se = 0.0f, sm = 0.0f;
        for (int s = lane; s < S; s += 64) {
            const float x = expf(er[s] - mx);
            se += x;
            sm += x * (mr ? mr[s] : 1.0f);
        }
        se = nm_wave_sum(se);
        sm = nm_wave_sum(sm);
        const float inv_se = 1.0f / se;
        const float invN = 1.0f / (sm * inv_se + 1e-8f);
        float sdw = 0.0f;
        for (int s = lane; s < S; s += 64) sdw += sh[s] * (expf(er[s] - mx) * inv_se * (mr ? mr[s] : 1.0f) * invN);
        sdw = nm_wave_sum(sdw);
        float sdp = 0.0f;
        for (int s = lane; s < S; s += 64)
            sdp += (mr ? mr[s] : 1.0f) * invN * (sh[s] - sdw) * (expf(er[s] - mx) * inv_se);
        sdp = nm_wave_sum(sdp);
        for (int s = lane; s < S; s += 64) {
            const float p = expf(er[s] - mx) * inv_se;
            const float dp = (mr ? mr[s] : 1.0f) * invN * (sh[s] - sdw);
            const float d = p * (dp - sdp);
            sh[s] = d;
            de[(long)b * S + s] = d;
        }
    }
    __syncthreads();
    // (3) the query gradient: a wave owns 64 feature columns; with fewer column chunks than waves the positions are
    // cut into parts (A = 512, 16 waves: 2 parts of 25 positions) whose sums meet in LDS
    const float* hb = hf + (long)b * S * A;
    const int nchunks = (A + 63) >> 6;
    const bool cut = nchunks <= nw;
    const int parts = cut ? nw / nchunks : 1;
    const int part = cut ? wave / nchunks : 0;
    const int per = (S + parts - 1) / parts;
    const int sb = part * per, se_ = min(S, sb + per);
    float total = 0.0f;
    for (int ch = cut ? wave % nchunks : wave; part < parts && ch < nchunks; ch += cut ? nchunks : nw) {
        const int a = ch * 64 + lane;
        float g = 0.0f;
        if (a < A) {
            const float ya = y[(long)b * ldy + a];
            for (int s = sb; s < se_; s += 13) {                  // (13 keys in flight per thread: 25 positions = two passes)
                float hv[13];
#pragma unroll
                for (int i = 0; i < 13; ++i) hv[i] = hb[(long)min(s + i, S - 1) * A + a];
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    const float z = nm_tanh(hv[i] + ya);
                    if (s + i < se_) g += sh[s + i] * (1.0f - z * z);
                }
            }
            if (!cut) dy[(long)b * lddy + a] = v[a] * g;
        }
        total = g;
    }
    if (!cut) return;                              // (block-uniform)
    pt[tid] = total;                               // wave = part * nchunks + chunk: pt[part][chunk * 64 + lane]
    __syncthreads();
    if (tid < A) {
        float g = 0.0f;
        for (int p = 0; p < parts; ++p) g += pt[p * nchunks * 64 + tid];
        dy[(long)b * lddy + tid] = v[tid] * g;
    }
}

extern "C" int nm_attn_step_bwd(void* stream, const float* dctx, int64_t lddctx, const float* states, const float* e,
                                const float* mask, const float* hf, const float* y, int64_t ldy, const float* v,
                                float* de, float* dy, int64_t lddy, int64_t B, int64_t S, int64_t C, int64_t A) {
    NM_REQUIRE(dctx && states && e && hf && y && v && de && dy, "nm_attn_step_bwd: null pointer");
    NM_REQUIRE(B > 0 && S > 0 && C > 0 && A > 0 && S <= 8192 && C % 4 == 0 && lddctx % 4 == 0 && lddctx >= C &&
                   ldy >= A && lddy >= A,
               "nm_attn_step_bwd: bad shape");
    NM_REQUIRE(((uintptr_t)dctx | (uintptr_t)states) % 16 == 0, "nm_attn_step_bwd: dctx / states not 16-byte aligned");
    hipLaunchKernelGGL(attn_step_bwd_kernel, dim3((unsigned)B), dim3(1024), (size_t)(S + 1024) * sizeof(float), nm_stream(stream),
                       dctx, (long)lddctx, states, e, mask, hf, y, (long)ldy, v, de, dy, (long)lddy, (int)S, (int)C,
                       (int)A);
    NM_LAUNCH_CHECK("nm_attn_step_bwd");
}

// r*h_prev of every sequence position, position-major [B,S,ndir,H] (operand of
// the candidate-kernel weight gradient (r*h_prev)^T . dc_pre).  ru_all is
// step-major [S,ndir,B,2H]; the step that visited position p is p (forward) or
// L-1-p (reversed).
__global__ void gru_rh_seq_kernel(const float* __restrict__ ru_all, const float* __restrict__ hprev,
                                  float* __restrict__ out, const int* __restrict__ lengths, int rev_mask,
                                  int B, int S, int ndir, int H) {
    const int b = blockIdx.z, p = blockIdx.y;
    const int col = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (col >= ndir * H) return;
    const int d = col / H, j = col - d * H;
    const int len = lengths ? lengths[b] : S;
    float4 v = make_float4(0, 0, 0, 0);
    if (p < len) {
        const int t = ((rev_mask >> d) & 1) ? len - 1 - p : p;
        const float4 r = *reinterpret_cast<const float4*>(ru_all + (((long)t * ndir + d) * B + b) * 2 * H + j);
        const float4 h = *reinterpret_cast<const float4*>(hprev + ((long)b * S + p) * ndir * H + col);
        v = make_float4(r.x * h.x, r.y * h.y, r.z * h.z, r.w * h.w);
    }
    *reinterpret_cast<float4*>(out + ((long)b * S + p) * ndir * H + col) = v;
}

extern "C" int nm_gru_rh_seq(void* stream, const float* ru_all, const float* hprev, float* out,
                             const int32_t* lengths, int rev_mask, int64_t B, int64_t S, int ndir,
                             int64_t H) {
    NM_REQUIRE(ru_all && hprev && out && B > 0 && S > 0 && H % 4 == 0, "nm_gru_rh_seq: bad args");
    hipLaunchKernelGGL(gru_rh_seq_kernel, dim3(nm_cdiv(ndir * H, 1024), (unsigned)S, (unsigned)B), dim3(256),
                       0, nm_stream(stream), ru_all, hprev, out, lengths, rev_mask, (int)B, (int)S, ndir,
                       (int)H);
    NM_LAUNCH_CHECK("nm_gru_rh_seq");
}
