// Fused Bahdanau attention step (score + softmax + mask-renorm + context).
// Reference: Attention.attention / get_energies, attention/feed_forward.py:120-166:
//   e[r,s]  = sum_a v[a] * tanh(hf[b,s,a] + y[r,a]) + bias
//   w       = softmax(e) * mask ; w /= (sum(w) + 1e-8)          (:139-144)
//   ctx[r]  = sum_s w[r,s] * states[b,s,:]                      (:151-154)
// with b = r / rows_per_key: the reference cannot tile the keys to a beam
// (SURVEY 3.3); indexing keys by row/k gives the batch-1 broadcast result for
// any batch.
//
// HBM-bound: one step streams hf [Bk,S,A] and states [Bk,S,C] once
// (53.5 MB at B=128,S=50,A=C=1024).  Layout: row-major, innermost a/c, so a
// wave reads one 4 KB row with 16 B per lane, fully coalesced.
//
// Split-S (flash-decoding style) so the grid has Bk*nchunk >> 256 workgroups:
//   attn_partial : block (chunk, b): energies of its SCH rows (one wave per
//                  row, shuffle reduction over a), chunk-local max / sums, then
//                  the partial context of the same rows (each wave owns a
//                  256-column slice, no cross-wave reduction).  All QPK queries
//                  of one key batch share a single read of hf / states.
//   attn_combine : per query row, merge the nchunk partials, write ctx and
//                  the normalised weights.
// The softmax-then-mask-renorm of the reference is carried exactly:
//   w_s = exp(e_s-M) m_s / (sum_j exp(e_j-M) m_j + 1e-8 * sum_j exp(e_j-M)).
#include "nm_common.h"

#include <stdlib.h>
#include <utility>
#include <vector>

#define ATT_MAX_SCH 16

// Yardstick for the timing above: the same number of bytes read by the simplest possible kernel (16-byte loads,
// four in flight per lane, 2048 workgroups), timed through the same event pool.
__global__ __launch_bounds__(256) void stream_read_kernel(const float4* __restrict__ src, long n4, float* __restrict__ sink) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
    }
    for (; i < n4; i += stride) {
        const float4 a = src[i];
        acc += a.x + a.y + a.z + a.w;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

extern "C" int nm_prof_stream_read(void* stream, const void* src, int64_t bytes, float* sink) {
    NM_REQUIRE(src && sink && bytes >= 16 && nm_aligned16(src), "nm_prof_stream_read: bad arguments");
    hipStream_t st = nm_stream(stream);
    NmCtx* const nmc = nm_cur();
    std::pair<hipEvent_t, hipEvent_t>* prof = nmc->prof_on ? nm_prof_next_pair(nmc) : nullptr;
    if (prof) (void)hipEventRecord(prof->first, st);
    const long n4 = (long)(bytes / 16);
    const unsigned blocks = (unsigned)(n4 >= 2048L * 256 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(stream_read_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)src, n4, sink);
    if (prof) (void)hipEventRecord(prof->second, st);
    NM_LAUNCH_CHECK("nm_prof_stream_read");
}

struct AttnArgs {
    const float* y;       // [R,A]
    const float* hf;      // [Bk,S,A]
    const float* states;  // [Bk,S,C]
    const float* mask;    // [Bk,S] or null
    const float* v;       // [A]
    const float* bias;    // [1] device scalar or null
    float* energies;      // [R,S]   raw energies (workspace)
    float* pctx;          // [R,nchunk,C]
    float* pstat;         // [R,nchunk,4]  (m, l_all, l_masked, -)
    int R, S, A, C, nchunk, sch;
    // query q (0 <= q < nq) of key batch b is row b*qsb + q*qsq of y / ctx / weights:
    // beam layout (qsb = nq, qsq = 1) or time-major layout (qsb = 1, qsq = Bk)
    int qsb, qsq, nq;
    // in-kernel merge of the split-S partials (fast kernels): the chunk workgroup that arrives LAST for its key
    // batch merges all partials and writes ctx / weights -- no combine launch
    unsigned* tickets;    // [Bk] arrival counters, zero between launches (the merging workgroup restores the zero)
    float* ctx; long ldctx;
    float* weights;       // [R,S] or null
    int merge, Bk;
};

__device__ __forceinline__ long attn_qrow(const AttnArgs& p, int b, int qi) {
    return (long)b * p.qsb + (long)min(qi, p.nq - 1) * p.qsq;
}


// ---- in-kernel hand-off of the partials to the merging workgroup (MI355X: per-XCD L2s are not coherent, a CU's
// L1 is never refreshed by other CUs' stores).  Producer side of the guide's recipe R1: the payload is stored
// WRITE-THROUGH (agent-scope relaxed stores lower to `global_store ... sc1`), every storing wave drains its
// stores (s_waitcnt vmcnt(0)), the workgroup meets, ONE lane takes a ticket with an agent-scope atomic.  The
// workgroup that draws the last ticket is the consumer: it reads the partials with L1-bypassing (sc1) loads
// (measured the same as the recipe's agent-scope acquire + plain loads: 17.8 us cold either way).
__device__ __forceinline__ void st_wt2(float* p, float a, float b) {
    const unsigned long long x = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_wt4(float* p, float4 v) { st_wt2(p, v.x, v.y); st_wt2(p + 2, v.z, v.w); }
__device__ __forceinline__ void st_wt1(float* p, float a) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// consumer side: agent-scope relaxed loads (`global_load ... sc1`) bypass this CU's L1, which may still hold the
// previous step's lines of the same buffers; they are valid without an acquire fence because the producers
// stored sc1 (write-through) and drained before taking their tickets
__device__ __forceinline__ float4 ld_wt4(const float* p) {
    const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                       __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}
__device__ __forceinline__ float ld_wt1(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT));
}

// true for every thread of the workgroup that arrived last for key batch b
__device__ __forceinline__ bool attn_arrive_last(const AttnArgs& p, int b, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.tickets + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == (unsigned)(p.nchunk - 1));
        if (last)      // ready for the next launch; (no acquire fence: the merge reads with sc1 loads, see ld_wt4)
            __hip_atomic_store(p.tickets + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0;
}

// merge of the nchunk partials of query row qr (key batch b) by the 256 threads of the last-arriving workgroup:
// attn_combine's arithmetic (feed_forward.py:139-154 carried through the chunk statistics)
#define ATT_MERGE_MAXCH 8
__device__ __forceinline__ void attn_merge_row(const AttnArgs& p, long qr, int b) {
    const int tid = threadIdx.x;
    const float* pc = p.pctx + qr * p.nchunk * p.C;
    const float4* st = reinterpret_cast<const float4*>(p.pstat) + qr * p.nchunk;
    float4 x[ATT_MERGE_MAXCH];
    {
        const int c = tid * 4;
#pragma unroll
        for (int i = 0; i < ATT_MERGE_MAXCH; ++i)
            x[i] = (c < p.C && i < p.nchunk) ? ld_wt4(pc + (long)i * p.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float e_s = 0.0f, mk = 1.0f;
    const bool w_ok = p.weights && tid < p.S;
    if (w_ok) {
        e_s = ld_wt1(p.energies + qr * p.S + tid);
        if (p.mask) mk = p.mask[(long)b * p.S + tid];
    }
    float4 sv[ATT_MERGE_MAXCH];
    float M = -INFINITY;
#pragma unroll
    for (int i = 0; i < ATT_MERGE_MAXCH; ++i) {
        sv[i] = i < p.nchunk ? ld_wt4(reinterpret_cast<const float*>(st + i)) : make_float4(-INFINITY, 0.f, 0.f, 0.f);
        M = fmaxf(M, sv[i].x);
    }
    float la = 0.0f, lm = 0.0f, f[ATT_MERGE_MAXCH];
#pragma unroll
    for (int i = 0; i < ATT_MERGE_MAXCH; ++i) {
        f[i] = i < p.nchunk ? expf(sv[i].x - M) : 0.0f;
        la += f[i] * sv[i].y;
        lm += f[i] * sv[i].z;
    }
    const float inv = 1.0f / (lm + 1e-8f * la);
    for (int c = tid * 4; c < p.C; c += 1024) {
        if (c >= 1024) {                               // second 1024-column group (C up to 2048): one more round trip
#pragma unroll
            for (int i = 0; i < ATT_MERGE_MAXCH; ++i)
                x[i] = i < p.nchunk ? ld_wt4(pc + (long)i * p.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < ATT_MERGE_MAXCH; ++i) {
            a.x += f[i] * x[i].x; a.y += f[i] * x[i].y; a.z += f[i] * x[i].z; a.w += f[i] * x[i].w;
        }
        a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
        *reinterpret_cast<float4*>(p.ctx + qr * p.ldctx + c) = a;
    }
    if (w_ok) p.weights[qr * p.S + tid] = expf(e_s - M) * mk * inv;
    if (p.weights)
        for (int s2 = tid + 256; s2 < p.S; s2 += 256) {
            const float m2 = p.mask ? p.mask[(long)b * p.S + s2] : 1.0f;
            p.weights[qr * p.S + s2] = expf(ld_wt1(p.energies + qr * p.S + s2) - M) * m2 * inv;
        }
}

template <int QPK, int NCG>
__global__ __launch_bounds__(256) void attn_partial(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ys = smem;                         // [QPK][A]
    float* vs = ys + QPK * p.A;               // [A]
    float* es = vs + p.A;                     // [QPK][ATT_MAX_SCH]  energies -> p
    float* ms = es + QPK * ATT_MAX_SCH;       // [ATT_MAX_SCH] mask
    float* pe = ms + ATT_MAX_SCH;             // [4][QPK][ATT_MAX_SCH] per-wave energy partials

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int s0 = chunk * p.sch;
    const int ns = min(p.sch, p.S - s0);
    const int q0 = blockIdx.z * QPK;            // first query of this block's group

    // the queries are staged as exp(2 y): every key element is read by QPK queries, so exp(2 hf) is evaluated once
    // per element and a tanh costs one v_rcp_f32 (nm_tanh_prod); queries or keys beyond the exact range of that
    // form (|.| > NM_EXP2X_MAX) send their slice through nm_tanh on the original values
    __shared__ int y_wide;
    if (tid == 0) y_wide = 0;
    __syncthreads();
    bool wide = false;
#pragma unroll
    for (int q = 0; q < QPK; ++q)
        for (int i = tid * 4; i < p.A; i += 1024) {
            const float4 y4 = *reinterpret_cast<const float4*>(p.y + attn_qrow(p, b, q0 + q) * p.A + i);
            wide |= fmaxf(fmaxf(fabsf(y4.x), fabsf(y4.y)), fmaxf(fabsf(y4.z), fabsf(y4.w))) > NM_EXP2X_MAX;
            *reinterpret_cast<float4*>(ys + q * p.A + i) =
                make_float4(nm_exp2x(y4.x), nm_exp2x(y4.y), nm_exp2x(y4.z), nm_exp2x(y4.w));
        }
    if (wide) y_wide = 1;
    for (int i = tid * 4; i < p.A; i += 1024)
        *reinterpret_cast<float4*>(vs + i) = *reinterpret_cast<const float4*>(p.v + i);
    if (tid < ATT_MAX_SCH)
        ms[tid] = (tid < ns) ? (p.mask ? p.mask[(long)b * p.S + s0 + tid] : 1.0f) : 0.0f;
    __syncthreads();
    const bool y_exact = y_wide != 0;

    const float bias = p.bias ? p.bias[0] : 0.0f;

    // ---- phase 1: energies.  Every wave owns 256-column slices of ALL rows of the
    // chunk (same access pattern as phase 3): 4 rows x 16 B per lane in flight, perfectly
    // balanced across waves; per-row partial sums meet in LDS.
    {
        const float* hbase = p.hf + ((long)b * p.S + s0) * p.A;
        for (int s4 = 0; s4 < ns; s4 += 4) {
            float acc[QPK][4];
#pragma unroll
            for (int q = 0; q < QPK; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[q][i] = 0.0f;
            for (int a = wave * 256 + lane * 4; a < p.A; a += 1024) {
                float4 h4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int sl = min(s4 + i, ns - 1);          // clamped duplicates are dropped below
                    h4[i] = *reinterpret_cast<const float4*>(hbase + (long)sl * p.A + a);
                }
                const float4 v4 = *reinterpret_cast<const float4*>(vs + a);
                float hmax = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    hmax = fmaxf(fmaxf(hmax, fmaxf(fabsf(h4[i].x), fabsf(h4[i].y))), fmaxf(fabsf(h4[i].z), fabsf(h4[i].w)));
                if (y_exact || hmax > NM_EXP2X_MAX) {            // outside the product form's exact range (rare)
#pragma unroll
                    for (int q = 0; q < QPK; ++q) {
                        const float4 y4 = *reinterpret_cast<const float4*>(p.y + attn_qrow(p, b, q0 + q) * p.A + a);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[q][i] += v4.x * nm_tanh(h4[i].x + y4.x) + v4.y * nm_tanh(h4[i].y + y4.y) +
                                         v4.z * nm_tanh(h4[i].z + y4.z) + v4.w * nm_tanh(h4[i].w + y4.w);
                    }
                } else {
                    float4 eh[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        eh[i] = make_float4(nm_exp2x(h4[i].x), nm_exp2x(h4[i].y), nm_exp2x(h4[i].z), nm_exp2x(h4[i].w));
#pragma unroll
                    for (int q = 0; q < QPK; ++q) {
                        const float4 ey = *reinterpret_cast<const float4*>(ys + q * p.A + a);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[q][i] += v4.x * nm_tanh_prod(eh[i].x, ey.x) + v4.y * nm_tanh_prod(eh[i].y, ey.y) +
                                         v4.z * nm_tanh_prod(eh[i].z, ey.z) + v4.w * nm_tanh_prod(eh[i].w, ey.w);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < QPK; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r = nm_wave_sum_dpp(acc[q][i]);
                    if (lane == 0 && s4 + i < ns) pe[(wave * QPK + q) * ATT_MAX_SCH + s4 + i] = r;
                }
        }
    }
    __syncthreads();
    if (tid < QPK * ns) {
        const int q = tid / ns, sl = tid - q * ns;
        const float e = ((pe[(0 * QPK + q) * ATT_MAX_SCH + sl] + pe[(1 * QPK + q) * ATT_MAX_SCH + sl]) +
                         (pe[(2 * QPK + q) * ATT_MAX_SCH + sl] + pe[(3 * QPK + q) * ATT_MAX_SCH + sl])) + bias;
        es[q * ATT_MAX_SCH + sl] = e;
        if (q0 + q < p.nq) p.energies[attn_qrow(p, b, q0 + q) * p.S + s0 + sl] = e;
    }
    __syncthreads();

    // ---- phase 2: chunk-local softmax statistics ---------------------------
    if (tid < QPK) {
        const int q = tid;
        float m = -INFINITY;
        for (int s = 0; s < ns; ++s) m = fmaxf(m, es[q * ATT_MAX_SCH + s]);
        float la = 0.0f, lm = 0.0f;
        for (int s = 0; s < ns; ++s) {
            const float e = expf(es[q * ATT_MAX_SCH + s] - m);
            la += e;
            const float em = e * ms[s];
            lm += em;
            es[q * ATT_MAX_SCH + s] = em;     // masked, un-normalised weight
        }
        if (q0 + q < p.nq) {
            float* st = p.pstat + (attn_qrow(p, b, q0 + q) * p.nchunk + chunk) * 4;
            st[0] = m; st[1] = la; st[2] = lm; st[3] = 0.0f;
        }
    }
    __syncthreads();

    // ---- phase 3: partial context, wave owns 256-column slices -------------
    float4 acc[QPK][NCG];
#pragma unroll
    for (int q = 0; q < QPK; ++q)
#pragma unroll
        for (int g = 0; g < NCG; ++g) acc[q][g] = make_float4(0, 0, 0, 0);
    const float* sbase = p.states + ((long)b * p.S + s0) * p.C;
#pragma unroll 4
    for (int sl = 0; sl < ns; ++sl) {
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            const int col = g * 1024 + wave * 256 + lane * 4;
            if (col < p.C) {
                const float4 x = *reinterpret_cast<const float4*>(sbase + (long)sl * p.C + col);
#pragma unroll
                for (int q = 0; q < QPK; ++q) {
                    const float w = es[q * ATT_MAX_SCH + sl];
                    acc[q][g].x += w * x.x; acc[q][g].y += w * x.y;
                    acc[q][g].z += w * x.z; acc[q][g].w += w * x.w;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < QPK; ++q)
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            const int col = g * 1024 + wave * 256 + lane * 4;
            if (col < p.C && q0 + q < p.nq)
                *reinterpret_cast<float4*>(p.pctx + (attn_qrow(p, b, q0 + q) * p.nchunk + chunk) * p.C + col) =
                    acc[q][g];
        }
}

// ---------------------------------------------------------------------------
// Fast path: one query per key batch (greedy decoding and every training
// step), A <= 1024, C <= 1024.  Every HBM load of the block -- its rows of hf
// AND of states, 2 x 12 x 16 B per lane -- is issued before any arithmetic, so
// the block pays one memory latency instead of one per row group per phase;
// y / v slices live in registers (no LDS staging), one __syncthreads, and the
// chunk softmax is evaluated redundantly per thread instead of serially.
// ---------------------------------------------------------------------------
#define ATT_FAST_ROWS 14
// NCG = value column groups of 1024 per lane: 1 for C <= 1024, 2 for C <= 2048 (config 4: 8x8x2048 maps attended with
// state 512 -- each lane then streams its 16-byte key slice and TWO 16-byte value slices of every row of the chunk)
template <int ROWS, int NCG = 1>
__global__ __launch_bounds__(256) void attn_partial_fast(AttnArgs p) {
    __shared__ float pe[4][ATT_MAX_SCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int s0 = chunk * p.sch;
    const int ns = min(p.sch, p.S - s0);
    const int col = wave * 256 + lane * 4;
    const bool a_ok = col < p.A;
    bool c_ok[NCG];
#pragma unroll
    for (int g = 0; g < NCG; ++g) c_ok[g] = col + g * 1024 < p.C;
    // out-of-range lanes / rows read a valid address and are weighted by zero below:
    // no branch sits between the loads, so all (1 + NCG) * ROWS of them are in flight together
    const float* hbase = p.hf + ((long)b * p.S + s0) * p.A + (a_ok ? col : 0);
    const float* sbase = p.states + ((long)b * p.S + s0) * p.C;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // y / v before the rows: vmcnt counts in issue order, a tanh of row s then only waits for rows 0..s
    const float4 y4 = *reinterpret_cast<const float4*>(p.y + (long)b * p.A + (a_ok ? col : 0));
    float4 v4 = *reinterpret_cast<const float4*>(p.v + (a_ok ? col : 0));
    if (!a_ok) v4 = zero4;
    const float bias = p.bias ? p.bias[0] : 0.0f;
    float4 hfr[ROWS], str[NCG][ROWS];
#pragma unroll
    for (int s = 0; s < ROWS; ++s)
        hfr[s] = *reinterpret_cast<const float4*>(hbase + (long)min(s, ns - 1) * p.A);
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int s = 0; s < ROWS; ++s)
            str[g][s] = *reinterpret_cast<const float4*>(sbase + (long)min(s, ns - 1) * p.C +
                                                         (c_ok[g] ? col + g * 1024 : 0));

#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
        float part = v4.x * nm_tanh(hfr[s].x + y4.x) + v4.y * nm_tanh(hfr[s].y + y4.y) +
                     v4.z * nm_tanh(hfr[s].z + y4.z) + v4.w * nm_tanh(hfr[s].w + y4.w);
        part = nm_wave_sum_dpp(part);
        if (lane == 0) pe[wave][s] = part;
    }
    __syncthreads();

    float e[ROWS];
    float m = -INFINITY;
#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
        e[s] = ((pe[0][s] + pe[1][s]) + (pe[2][s] + pe[3][s])) + bias;
        if (s < ns) m = fmaxf(m, e[s]);
    }
    float la = 0.0f, lm = 0.0f;
    float4 acc[NCG];
#pragma unroll
    for (int g = 0; g < NCG; ++g) acc[g] = zero4;
#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
        const bool ok = s < ns;
        const float ex = ok ? expf(e[s] - m) : 0.0f;
        const float mk = (ok && p.mask) ? p.mask[(long)b * p.S + s0 + s] : 1.0f;
        const float em = ex * mk;
        la += ex;
        lm += em;
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            acc[g].x += em * str[g][s].x; acc[g].y += em * str[g][s].y;
            acc[g].z += em * str[g][s].z; acc[g].w += em * str[g][s].w;
        }
    }
    const float e_own = tid < ns ? ((pe[0][tid] + pe[1][tid]) + (pe[2][tid] + pe[3][tid])) + bias : 0.0f;
    float* st = p.pstat + ((long)b * p.nchunk + chunk) * 4;
    float* pc = p.pctx + ((long)b * p.nchunk + chunk) * p.C + col;
    if (!p.merge) {
        if (tid < ns) p.energies[(long)b * p.S + s0 + tid] = e_own;
        if (tid == 0) { st[0] = m; st[1] = la; st[2] = lm; st[3] = 0.0f; }
#pragma unroll
        for (int g = 0; g < NCG; ++g)
            if (c_ok[g]) *reinterpret_cast<float4*>(pc + g * 1024) = acc[g];
        return;
    }
    // write-through hand-off to the workgroup that merges this sentence's partials (the last one to arrive)
    if (tid < ns) st_wt1(p.energies + (long)b * p.S + s0 + tid, e_own);
    if (tid == 0) st_wt4(st, make_float4(m, la, lm, 0.0f));
#pragma unroll
    for (int g = 0; g < NCG; ++g)
        if (c_ok[g]) st_wt4(pc + g * 1024, acc[g]);
    __shared__ int s_last;
    if (!attn_arrive_last(p, b, &s_last)) return;
    attn_merge_row(p, b, b);
}

// ---------------------------------------------------------------------------
// One query per sentence, short sources (S <= 4 * ATT_WHOLE_ROWS): ONE 1024-thread workgroup per sentence does
// the whole step -- no split-S partials, no cross-workgroup hand-off, no merge.  16 waves = 4 row groups x 4
// column waves: every lane requests its 16-byte slice of the keys AND values of its group's rows up front (all
// of the sentence's 300 KB in flight at once), the energies of the 4 column waves meet in LDS, wave 0 evaluates
// softmax -> mask -> renormalise (+1e-8) with one source position per lane, the 4 row groups' partial contexts
// meet in LDS.  Half the CUs stay idle at 128 sentences, but the launch no longer ends with write-through
// stores + a ticket + a second pass over the partials by the last-arriving workgroup.  (Measured and not kept:
// two workgroups per sentence split by COLUMNS that exchange their partial energies as 8-byte {value, tag}
// granules -- all 256 CUs busy, nothing merged afterwards, yet 21.9 / 17.3 us cold / warm against 20.6 / 16.6
// here: the exchange waits in the consumer CU's memory queue behind its own streaming loads,
// profiles/r02_attn_pair_vs_whole.txt.  Round 3, the same question split by POSITIONS: two 1024-thread workgroups
// per sentence with half of the rows each, write-through partials, the half that arrives second merges -- 17.7 /
// 14.9 us cold / warm against 17.2 / 13.7 for this kernel in the same run, greedy batch 5.92 vs 5.84 ms: the
// hand-off costs more than the second set of CUs brings.  Round 5, no hand-off at all: two workgroups per sentence on
// one XCD that each score the sentence by themselves and blend half of the value columns -- the split that took the
// captioning step from 30 to 16 us (attn_whole_wide below) -- 13.7 / 11.4 us cold / warm against 13.5 / 10.9 for this
// kernel (rocprofv3, same run): at this size the step is not bound by what one CU can stream.  And the beam step's five
// queries per sentence on such workgroup pairs (no split-S partials, no combine launch): 31.6 us per step against
// 24.9 + 5.6 for attn_partial_fastq + attn_combine -- every workgroup of a pair evaluates all 5 x 51 200 tanh, and at
// one quarter-rate v_rcp_f32 each the transcendental pipes, not the loads, set the time.)
// ---------------------------------------------------------------------------
#define ATT_WHOLE_ROWS 13
template <int ROWS>
__global__ __launch_bounds__(1024) void attn_whole_fast(AttnArgs p) {
    __shared__ float pe[4][4 * ROWS];          // [column wave][row group * ROWS + row]
    __shared__ float wsh[4 * ROWS];            // normalised weights, same indexing
    __shared__ float4 red[3][256];             // partial contexts of row groups 1..3
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, cw = wave & 3;
    const int b = blockIdx.x;
    const int rpg = (p.S + 3) >> 2;            // rows per row group
    const int s0 = grp * rpg;
    const int ns = max(0, min(rpg, p.S - s0));
    const int col = cw * 256 + lane * 4;
    const bool a_ok = col < p.A, c_ok = col < p.C;
    const float* hbase = p.hf + (long)b * p.S * p.A + (a_ok ? col : 0);
    const float* sbase = p.states + (long)b * p.S * p.C + (c_ok ? col : 0);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // the query slice and v FIRST: vmcnt counts loads in issue order, so a tanh of key row s may start as soon as
    // rows 0..s have landed only if nothing it needs was requested after the rows (with y / v requested last the
    // compiler had to wait for all 26 row loads -- s_waitcnt vmcnt(0) -- before the first tanh)
    const float4 y4 = *reinterpret_cast<const float4*>(p.y + (long)b * p.A + (a_ok ? col : 0));
    float4 v4 = *reinterpret_cast<const float4*>(p.v + (a_ok ? col : 0));
    if (!a_ok) v4 = zero4;
    const float bias = p.bias ? p.bias[0] : 0.0f;
    float4 hfr[ROWS], str[ROWS];
#pragma unroll
    for (int s = 0; s < ROWS; ++s)            // clamped rows are weighted by zero below
        hfr[s] = *reinterpret_cast<const float4*>(hbase + (long)min(s0 + min(s, max(ns - 1, 0)), p.S - 1) * p.A);
#pragma unroll
    for (int s = 0; s < ROWS; ++s)
        str[s] = *reinterpret_cast<const float4*>(sbase + (long)min(s0 + min(s, max(ns - 1, 0)), p.S - 1) * p.C);

#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
        float part = v4.x * nm_tanh(hfr[s].x + y4.x) + v4.y * nm_tanh(hfr[s].y + y4.y) +
                     v4.z * nm_tanh(hfr[s].z + y4.z) + v4.w * nm_tanh(hfr[s].w + y4.w);
        part = nm_wave_sum_dpp(part);
        if (lane == 0) pe[cw][grp * ROWS + s] = part;
    }
    __syncthreads();

    if (wave == 0) {                           // one source position per lane (S <= 52)
        const bool ok = lane < p.S;
        const int g = ok ? lane / rpg : 0, idx = g * ROWS + (ok ? lane - g * rpg : 0);
        const float e = ok ? ((pe[0][idx] + pe[1][idx]) + (pe[2][idx] + pe[3][idx])) + bias : -INFINITY;
        const float m = nm_wave_max_dpp(e);
        const float ex = ok ? expf(e - m) : 0.0f;
        const float mk = (ok && p.mask) ? p.mask[(long)b * p.S + lane] : 1.0f;
        const float em = ex * mk;
        const float la = nm_wave_sum_dpp(ex), lm = nm_wave_sum_dpp(em);
        const float w = em * (1.0f / (lm + 1e-8f * la));
        if (ok) {
            wsh[idx] = w;
            p.energies[(long)b * p.S + lane] = e;
            if (p.weights) p.weights[(long)b * p.S + lane] = w;
        }
    }
    __syncthreads();

    float4 acc = zero4;
#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
        const float w = s < ns ? wsh[grp * ROWS + s] : 0.0f;
        acc.x += w * str[s].x; acc.y += w * str[s].y;
        acc.z += w * str[s].z; acc.w += w * str[s].w;
    }
    if (grp > 0) red[grp - 1][cw * 64 + lane] = acc;
    __syncthreads();
    if (grp == 0 && c_ok) {
        const float4 r0 = red[0][cw * 64 + lane], r1 = red[1][cw * 64 + lane], r2 = red[2][cw * 64 + lane];
        acc.x += (r0.x + r1.x) + r2.x; acc.y += (r0.y + r1.y) + r2.y;
        acc.z += (r0.z + r1.z) + r2.z; acc.w += (r0.w + r1.w) + r2.w;
        *reinterpret_cast<float4*>(p.ctx + (long)b * p.ldctx + col) = acc;
    }
}

// ---------------------------------------------------------------------------
// Wide values, narrow keys (config 4, captioning: 8x8 maps of 2048 channels under a 512-wide attention, S = 64,
// encoders/numpy_stateful_filler.py:209-245 -> attention/feed_forward.py:125-166): TWO 1024-thread workgroups per
// sentence, split by VALUE columns.  Each scores the sentence by itself -- the keys are a fifth of the bytes and the
// second reader finds them in its XCD's L2 -- so nothing crosses workgroups: no split-S partial contexts, no
// ticket, no merge (the split-S kernels wrote 6.3 MB of partials and read them back: 99.6 MB of HBM traffic for 85.3 MB
// of operands, profiles/r04_attn_cap_pmc_cold.json), and all 256 CUs stream (one workgroup per sentence would leave
// half of them idle, each of the others filling its L1 with 655 KB).  Per thread: 8 key slices (a key row is 128
// 16-byte slices = 32 lanes of each of the 4 column waves: a wave loads TWO rows at a time) and 16 value slices,
// all 96 registers of loads in flight before the first tanh; softmax and the cross-group sum as in attn_whole_fast.
// ---------------------------------------------------------------------------
#define ATT_WIDE_ROWS 16
__device__ __forceinline__ void nm_half_sums_dpp(float v, float& lo, float& hi) {      // sums of lanes 0..31 / 32..63
    v += nm_dpp<0xB1, 0xf>(0.0f, v);            // quad_perm [1,0,3,2]
    v += nm_dpp<0x4E, 0xf>(0.0f, v);            // quad_perm [2,3,0,1]
    v += nm_dpp<0x141, 0xf>(0.0f, v);           // row_half_mirror
    v += nm_dpp<0x140, 0xf>(0.0f, v);           // row_mirror
    v += nm_dpp<0x142, 0xa>(0.0f, v);           // row_bcast:15 -> rows 1, 3
    lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__global__ __launch_bounds__(1024) void attn_whole_wide(AttnArgs p) {
    constexpr int ROWS = ATT_WIDE_ROWS;
    __shared__ float pe[4][4 * ROWS];          // [column wave][row group * ROWS + row]
    __shared__ float wsh[4 * ROWS];
    __shared__ float4 red[3][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, cw = wave & 3;
    // the two halves of a sentence are blocks i and i + 8: the same XCD under round-robin dispatch, so the second
    // reader of the keys finds them in that XCD's L2 (speed only: nothing is exchanged)
    const int b = ((int)blockIdx.x >> 4) * 8 + ((int)blockIdx.x & 7), half = ((int)blockIdx.x >> 3) & 1;
    if (b >= p.Bk) return;
    const int rpg = (p.S + 3) >> 2;            // rows per row group (<= 16)
    const int s0 = grp * rpg;
    const int ns = max(0, min(rpg, p.S - s0));
    const int hl = lane & 31, hr = lane >> 5;
    const int acol = cw * 128 + hl * 4;        // key columns: 512 = 4 column waves x 32 lanes x 4
    const int col = half * 1024 + cw * 256 + lane * 4;
    const bool a_ok = acol < p.A, c_ok = col < p.C;
    const float* hbase = p.hf + (long)b * p.S * p.A + (a_ok ? acol : 0);
    const float* sbase = p.states + (long)b * p.S * p.C + (c_ok ? col : 0);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    const float4 y4 = *reinterpret_cast<const float4*>(p.y + (long)b * p.A + (a_ok ? acol : 0));
    float4 v4 = *reinterpret_cast<const float4*>(p.v + (a_ok ? acol : 0));
    if (!a_ok) v4 = zero4;
    const float bias = p.bias ? p.bias[0] : 0.0f;
    float4 hfr[ROWS / 2], str[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS / 2; ++i)        // lanes 0..31: row 2i, lanes 32..63: row 2i + 1 (clamped rows weigh zero)
        hfr[i] = *reinterpret_cast<const float4*>(hbase + (long)min(s0 + min(2 * i + hr, max(ns - 1, 0)), p.S - 1) * p.A);
#pragma unroll
    for (int s = 0; s < ROWS; ++s)
        str[s] = *reinterpret_cast<const float4*>(sbase + (long)min(s0 + min(s, max(ns - 1, 0)), p.S - 1) * p.C);

#pragma unroll
    for (int i = 0; i < ROWS / 2; ++i) {
        const float part = v4.x * nm_tanh(hfr[i].x + y4.x) + v4.y * nm_tanh(hfr[i].y + y4.y) +
                           v4.z * nm_tanh(hfr[i].z + y4.z) + v4.w * nm_tanh(hfr[i].w + y4.w);
        float lo, hi;
        nm_half_sums_dpp(part, lo, hi);
        if (lane == 0) {
            pe[cw][grp * ROWS + 2 * i] = lo;
            pe[cw][grp * ROWS + 2 * i + 1] = hi;
        }
    }
    __syncthreads();

    if (wave == 0) {                           // one source position per lane (S <= 64)
        const bool ok = lane < p.S;
        const int g = ok ? lane / rpg : 0, idx = g * ROWS + (ok ? lane - g * rpg : 0);
        const float e = ok ? ((pe[0][idx] + pe[1][idx]) + (pe[2][idx] + pe[3][idx])) + bias : -INFINITY;
        const float m = nm_wave_max_dpp(e);
        const float ex = ok ? expf(e - m) : 0.0f;
        const float mk = (ok && p.mask) ? p.mask[(long)b * p.S + lane] : 1.0f;
        const float em = ex * mk;
        const float la = nm_wave_sum_dpp(ex), lm = nm_wave_sum_dpp(em);
        const float w = em * (1.0f / (lm + 1e-8f * la));
        if (ok) {
            wsh[idx] = w;
            if (half == 0) {                   // (both halves compute the same numbers; one of them stores them)
                p.energies[(long)b * p.S + lane] = e;
                if (p.weights) p.weights[(long)b * p.S + lane] = w;
            }
        }
    }
    __syncthreads();

    float4 acc = zero4;
#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
        const float w = s < ns ? wsh[grp * ROWS + s] : 0.0f;
        acc.x += w * str[s].x; acc.y += w * str[s].y;
        acc.z += w * str[s].z; acc.w += w * str[s].w;
    }
    if (grp > 0) red[grp - 1][cw * 64 + lane] = acc;
    __syncthreads();
    if (grp == 0 && c_ok) {
        const float4 r0 = red[0][cw * 64 + lane], r1 = red[1][cw * 64 + lane], r2 = red[2][cw * 64 + lane];
        acc.x += (r0.x + r1.x) + r2.x; acc.y += (r0.y + r1.y) + r2.y;
        acc.z += (r0.z + r1.z) + r2.z; acc.w += (r0.w + r1.w) + r2.w;
        *reinterpret_cast<float4*>(p.ctx + (long)b * p.ldctx + col) = acc;
    }
}

// ---------------------------------------------------------------------------
// The same idea for a handful of queries per key batch (beam search: the k hypotheses of a
// sentence share its keys): every HBM load of the block is in flight before the first tanh, the
// query slices live in registers, the NQ x ROWS chunk-local softmax is evaluated by NQ threads and
// handed to the context phase through LDS.  Queries beyond nq are clamped duplicates whose results
// are not stored (nq = 3 runs the NQ = 4 instance).
// ---------------------------------------------------------------------------
template <int ROWS, int NQ>
__global__ __launch_bounds__(256) void attn_partial_fastq(AttnArgs p) {
    __shared__ float pe[4][NQ][ATT_MAX_SCH];
    __shared__ float wq[NQ][ATT_MAX_SCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int s0 = chunk * p.sch;
    const int ns = min(p.sch, p.S - s0);
    const int col = wave * 256 + lane * 4;
    const bool a_ok = col < p.A, c_ok = col < p.C;
    const float* hbase = p.hf + ((long)b * p.S + s0) * p.A + (a_ok ? col : 0);
    const float* sbase = p.states + ((long)b * p.S + s0) * p.C + (c_ok ? col : 0);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // queries and v before the rows (vmcnt counts in issue order), and rows outermost below: every query's tanh of
    // row s runs as soon as rows 0..s have landed
    float4 hfr[ROWS], str[ROWS], y4[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        y4[q] = *reinterpret_cast<const float4*>(p.y + attn_qrow(p, b, q) * p.A + (a_ok ? col : 0));
    float4 v4 = *reinterpret_cast<const float4*>(p.v + (a_ok ? col : 0));
    if (!a_ok) v4 = zero4;
    const float bias = p.bias ? p.bias[0] : 0.0f;
#pragma unroll
    for (int s = 0; s < ROWS; ++s)
        hfr[s] = *reinterpret_cast<const float4*>(hbase + (long)min(s, ns - 1) * p.A);
#pragma unroll
    for (int s = 0; s < ROWS; ++s)
        str[s] = *reinterpret_cast<const float4*>(sbase + (long)min(s, ns - 1) * p.C);

    // NQ queries read every key element: exp(2 hf) once per element, exp(2 y) once per query column, one v_rcp_f32
    // per tanh (nm_tanh_prod); a wave whose queries or keys leave that form's exact range takes nm_tanh instead
    float ymax = 0.0f;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(y4[q].x), fabsf(y4[q].y))), fmaxf(fabsf(y4[q].z), fabsf(y4[q].w)));
    float4 ey[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        ey[q] = make_float4(nm_exp2x(y4[q].x), nm_exp2x(y4[q].y), nm_exp2x(y4[q].z), nm_exp2x(y4[q].w));
#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
        const float hmax = fmaxf(fmaxf(fabsf(hfr[s].x), fabsf(hfr[s].y)), fmaxf(fabsf(hfr[s].z), fabsf(hfr[s].w)));
        if (__any(fmaxf(hmax, ymax) > NM_EXP2X_MAX)) {           // wave-uniform: the DPP sums below need all lanes
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float part = v4.x * nm_tanh(hfr[s].x + y4[q].x) + v4.y * nm_tanh(hfr[s].y + y4[q].y) +
                             v4.z * nm_tanh(hfr[s].z + y4[q].z) + v4.w * nm_tanh(hfr[s].w + y4[q].w);
                part = nm_wave_sum_dpp(part);
                if (lane == 0) pe[wave][q][s] = part;
            }
        } else {
            const float4 eh = make_float4(nm_exp2x(hfr[s].x), nm_exp2x(hfr[s].y), nm_exp2x(hfr[s].z), nm_exp2x(hfr[s].w));
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float part = v4.x * nm_tanh_prod(eh.x, ey[q].x) + v4.y * nm_tanh_prod(eh.y, ey[q].y) +
                             v4.z * nm_tanh_prod(eh.z, ey[q].z) + v4.w * nm_tanh_prod(eh.w, ey[q].w);
                part = nm_wave_sum_dpp(part);
                if (lane == 0) pe[wave][q][s] = part;
            }
        }
    }
    __syncthreads();
    if (tid < NQ) {                                   // chunk-local softmax statistics of query tid
        const int q = tid;
        const bool live = q < p.nq;
        const long qr = attn_qrow(p, b, q);
        float e[ROWS];
        float m = -INFINITY;
#pragma unroll
        for (int s = 0; s < ROWS; ++s) {
            e[s] = ((pe[0][q][s] + pe[1][q][s]) + (pe[2][q][s] + pe[3][q][s])) + bias;
            if (s < ns) {
                m = fmaxf(m, e[s]);
                if (live) {
                    if (p.merge) st_wt1(p.energies + qr * p.S + s0 + s, e[s]);
                    else p.energies[qr * p.S + s0 + s] = e[s];
                }
            }
        }
        float la = 0.0f, lm = 0.0f;
#pragma unroll
        for (int s = 0; s < ROWS; ++s) {
            const bool ok = s < ns;
            const float ex = ok ? expf(e[s] - m) : 0.0f;
            const float mk = (ok && p.mask) ? p.mask[(long)b * p.S + s0 + s] : 1.0f;
            const float em = ex * mk;
            la += ex;
            lm += em;
            wq[q][s] = em;
        }
        if (live) {
            float* st = p.pstat + (qr * p.nchunk + chunk) * 4;
            if (p.merge) st_wt4(st, make_float4(m, la, lm, 0.0f));
            else { st[0] = m; st[1] = la; st[2] = lm; st[3] = 0.0f; }
        }
    }
    __syncthreads();
    float4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = zero4;
#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float w = wq[q][s];                  // zero beyond ns
            acc[q].x += w * str[s].x; acc[q].y += w * str[s].y;
            acc[q].z += w * str[s].z; acc[q].w += w * str[s].w;
        }
    }
    if (c_ok) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (q < p.nq) {
                float* pc = p.pctx + (attn_qrow(p, b, q) * p.nchunk + chunk) * p.C + col;
                if (p.merge) st_wt4(pc, acc[q]);
                else *reinterpret_cast<float4*>(pc) = acc[q];
            }
    }
    if (!p.merge) return;
    __shared__ int s_last;
    if (!attn_arrive_last(p, b, &s_last)) return;
    for (int q = 0; q < p.nq; ++q) attn_merge_row(p, attn_qrow(p, b, q), b);
}

// Merge of the split-S partials of one query row.  Latency-bound (2.6 MB in, 0.5 MB out over 128 rows), so
// nothing waits on anything it does not need: every thread issues its loads of the partial contexts first,
// derives the chunk scales from the (broadcast) statistics by itself -- no LDS, no barrier -- and the
// first S threads write the normalised weights.
#define ATT_COMBINE_MAXCH 8
__global__ __launch_bounds__(256) void attn_combine(const float* __restrict__ pctx,
                                                    const float* __restrict__ pstat,
                                                    const float* __restrict__ energies,
                                                    const float* __restrict__ mask,
                                                    float* __restrict__ ctx, long ldctx,
                                                    float* __restrict__ weights, int S, int C,
                                                    int nchunk, int mask_div, int mask_mod) {
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* pc = pctx + (long)r * nchunk * C;
    const float4* st = reinterpret_cast<const float4*>(pstat) + (long)r * nchunk;
    if (nchunk <= ATT_COMBINE_MAXCH && C <= 1024) {
        const int c = tid * 4;
        const bool c_ok = c < C;
        float4 x[ATT_COMBINE_MAXCH];
#pragma unroll
        for (int i = 0; i < ATT_COMBINE_MAXCH; ++i)
            x[i] = (c_ok && i < nchunk) ? *reinterpret_cast<const float4*>(pc + (long)i * C + c)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        const int b = (r / mask_div) % mask_mod;      // key batch of row r in either query layout
        float e_s = 0.0f, mk = 1.0f;
        const bool w_ok = weights && tid < S;
        if (w_ok) {
            e_s = energies[(long)r * S + tid];
            if (mask) mk = mask[(long)b * S + tid];
        }
        float4 sv[ATT_COMBINE_MAXCH];
        float M = -INFINITY;
#pragma unroll
        for (int i = 0; i < ATT_COMBINE_MAXCH; ++i) {
            sv[i] = i < nchunk ? st[i] : make_float4(-INFINITY, 0.f, 0.f, 0.f);
            M = fmaxf(M, sv[i].x);
        }
        float la = 0.0f, lm = 0.0f;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < ATT_COMBINE_MAXCH; ++i) {
            const float f = i < nchunk ? expf(sv[i].x - M) : 0.0f;
            la += f * sv[i].y;
            lm += f * sv[i].z;
            a.x += f * x[i].x; a.y += f * x[i].y; a.z += f * x[i].z; a.w += f * x[i].w;
        }
        const float inv = 1.0f / (lm + 1e-8f * la);
        if (c_ok) {
            a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
            *reinterpret_cast<float4*>(ctx + (long)r * ldctx + c) = a;
        }
        if (w_ok) weights[(long)r * S + tid] = expf(e_s - M) * mk * inv;
        if (weights)
            for (int s = tid + 256; s < S; s += 256) {
                const float m2 = mask ? mask[(long)b * S + s] : 1.0f;
                weights[(long)r * S + s] = expf(energies[(long)r * S + s] - M) * m2 * inv;
            }
        return;
    }
    __shared__ float sc[64];
    __shared__ float sden, smax;
    if (tid == 0) {
        float M = -INFINITY;
        for (int i = 0; i < nchunk; ++i) M = fmaxf(M, pstat[((long)r * nchunk + i) * 4]);
        float la = 0.0f, lm = 0.0f;
        for (int i = 0; i < nchunk; ++i) {
            const float* s4 = pstat + ((long)r * nchunk + i) * 4;
            const float f = expf(s4[0] - M);
            sc[i] = f;
            la += f * s4[1];
            lm += f * s4[2];
        }
        sden = lm + 1e-8f * la;
        smax = M;
    }
    __syncthreads();
    const float inv = 1.0f / sden;
    for (int c = tid * 4; c < C; c += 1024) {
        float4 a = make_float4(0, 0, 0, 0);
        for (int i = 0; i < nchunk; ++i) {
            const float4 x = *reinterpret_cast<const float4*>(pctx + ((long)r * nchunk + i) * C + c);
            const float f = sc[i];
            a.x += f * x.x; a.y += f * x.y; a.z += f * x.z; a.w += f * x.w;
        }
        a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
        *reinterpret_cast<float4*>(ctx + (long)r * ldctx + c) = a;
    }
    if (weights) {
        const int b = (r / mask_div) % mask_mod;      // key batch of row r in either query layout
        for (int s = tid; s < S; s += 256) {
            const float mk = mask ? mask[(long)b * S + s] : 1.0f;
            weights[(long)r * S + s] = expf(energies[(long)r * S + s] - smax) * mk * inv;
        }
    }
}

static int attn_max_rows() { return nm_cur()->sw.attn_maxrows; }      // NM_ATTN_MAXROWS, read when the context was made

static void attn_chunking(int64_t S, int* sch, int* nchunk) {
    const int64_t mr = attn_max_rows();
    const int64_t n0 = (S + mr - 1) / mr;             // rows per chunk <= NM_ATTN_MAXROWS (12)
    *sch = (int)((S + n0 - 1) / n0);
    *nchunk = (int)((S + *sch - 1) / *sch);
}

extern "C" int64_t nm_attn_workspace_bytes(int64_t R, int64_t S, int64_t C) {
    if (R <= 0 || S <= 0 || C <= 0) return 0;
    int sch, nchunk;
    attn_chunking(S, &sch, &nchunk);
    const int64_t e = ((R * S + 3) / 4) * 4;
    // energies | partial contexts | partial statistics | one arrival counter per key batch (<= R of them).  The
    // counters must be ZERO when a step is launched; the kernels leave them zero, so a workspace is zeroed once.
    return (int64_t)sizeof(float) * (e + R * nchunk * C + R * nchunk * 4 + ((R + 3) / 4) * 4);
}

// ---------------------------------------------------------------------------------------------
// Any-shape fallback: A, C or the context row stride not a multiple of 4 (e.g. the reference's
// tests/small.ini: rnn_size 7 -> C = 14), S beyond the chunked kernels' limit, or C > 2048.  One
// workgroup per query row, scalar loads, the whole row of energies in LDS.  Same arithmetic
// (feed_forward.py:120-166: softmax over all S, then mask, then renormalise with +1e-8).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_generic(AttnArgs p, float* __restrict__ ctx, long ldctx,
                                                    float* __restrict__ weights, float* __restrict__ energies,
                                                    int Bk) {
    extern __shared__ float gsm[];          // [S] energies -> weights, then [4] reduction slots
    float* es = gsm;
    float* red = gsm + p.S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x % p.nq, b = blockIdx.x / p.nq;
    const long row = (long)b * p.qsb + (long)q * p.qsq;
    const float* yr = p.y + row * p.A;
    const float* hfb = p.hf + (long)b * p.S * p.A;
    const float bias = p.bias ? p.bias[0] : 0.0f;
    for (int s = wave; s < p.S; s += 4) {
        const float* h = hfb + (long)s * p.A;
        float acc = 0.0f;
        for (int a = lane; a < p.A; a += 64) acc += p.v[a] * nm_tanh(h[a] + yr[a]);
        acc = nm_wave_sum(acc);
        if (lane == 0) es[s] = acc + bias;
    }
    __syncthreads();
    if (energies)
        for (int s = tid; s < p.S; s += 256) energies[row * p.S + s] = es[s];
    float mx = -INFINITY;
    for (int s = tid; s < p.S; s += 256) mx = fmaxf(mx, es[s]);
    mx = nm_wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float se = 0.0f, sm = 0.0f;
    for (int s = tid; s < p.S; s += 256) {
        const float e = expf(es[s] - mx);
        se += e;
        sm += e * (p.mask ? p.mask[(long)b * p.S + s] : 1.0f);
    }
    se = nm_wave_sum(se);
    sm = nm_wave_sum(sm);
    if (lane == 0) { red[wave] = se; red[4 + wave] = sm; }
    __syncthreads();
    se = red[0] + red[1] + red[2] + red[3];
    sm = red[4] + red[5] + red[6] + red[7];
    const float inv_se = 1.0f / se;
    const float inv_n = 1.0f / (sm * inv_se + 1e-8f);
    __syncthreads();
    for (int s = tid; s < p.S; s += 256) {
        const float w = expf(es[s] - mx) * inv_se * (p.mask ? p.mask[(long)b * p.S + s] : 1.0f) * inv_n;
        es[s] = w;
        if (weights) weights[row * p.S + s] = w;
    }
    __syncthreads();
    const float* stb = p.states + (long)b * p.S * p.C;
    for (int c = tid; c < p.C; c += 256) {
        float acc = 0.0f;
        for (int s = 0; s < p.S; ++s) acc += es[s] * stb[(long)s * p.C + c];
        ctx[row * ldctx + c] = acc;
    }
}

static bool attn_vector_shape(int64_t S, int64_t A, int64_t C, int* sch, int* nchunk) {
    attn_chunking(S, sch, nchunk);
    return A % 4 == 0 && C % 4 == 0 && C <= 2048 && *sch <= ATT_MAX_SCH && *nchunk <= 64;
}

// do_combine == 0: only the split-S partial kernel is launched; the caller merges the partials (pctx / pstat
// in the workspace, nm_attn_partials_layout) inside its own consumer -- nm_step_group's a_kind 1 operand loader
static int attn_fwd_impl(void* stream, const float* y, const float* hf, const float* states,
                         const float* mask, const float* v, const float* bias, int64_t Bk,
                         int64_t nq, int64_t q_stride_b, int64_t q_stride_q, int64_t S, int64_t A,
                         int64_t C, float* ctx, int64_t ldctx, float* weights, void* workspace,
                         int64_t workspace_bytes, float* energies_out, int do_combine) {
    NM_REQUIRE(y && hf && states && v && (ctx || !do_combine) && workspace, "nm_attn_fwd: null pointer");
    NM_REQUIRE(Bk > 0 && nq > 0 && S > 0 && A > 0 && C > 0, "nm_attn_fwd: bad shape Bk=%ld nq=%ld S=%ld",
               (long)Bk, (long)nq, (long)S);
    const bool beam_layout = (q_stride_b == nq && q_stride_q == 1);
    const bool time_layout = (q_stride_b == 1 && q_stride_q == Bk);
    NM_REQUIRE(beam_layout || time_layout, "nm_attn_fwd: query rows must be [Bk,nq] or [nq,Bk] major");
    const int64_t R = Bk * nq;
    int sch, nchunk;
    attn_chunking(S, &sch, &nchunk);
    const bool vector_ok = A % 4 == 0 && C % 4 == 0 && ldctx % 4 == 0 && C <= 2048 && sch <= ATT_MAX_SCH &&
                           nchunk <= 64 && nm_aligned16(y) && nm_aligned16(hf) && nm_aligned16(states) &&
                           nm_aligned16(v) && nm_aligned16(ctx) && nm_aligned16(workspace);
    NM_REQUIRE(vector_ok || do_combine, "nm_attn_fwd_partials: shape / alignment outside the split-S kernels");
    if (!vector_ok) {
        NM_REQUIRE(S <= 16000, "nm_attn_fwd: S=%ld too long for the any-shape kernel", (long)S);
        NM_REQUIRE(R < (1LL << 31), "nm_attn_fwd: too many query rows");
        AttnArgs g;
        g.y = y; g.hf = hf; g.states = states; g.mask = mask; g.v = v; g.bias = bias;
        g.energies = nullptr; g.pctx = nullptr; g.pstat = nullptr;
        g.R = (int)R; g.S = (int)S; g.A = (int)A; g.C = (int)C; g.nchunk = 1; g.sch = (int)S;
        g.qsb = (int)q_stride_b; g.qsq = (int)q_stride_q; g.nq = (int)nq;
        hipLaunchKernelGGL(attn_generic, dim3((unsigned)R), dim3(256), sizeof(float) * (S + 8), nm_stream(stream),
                           g, ctx, (long)ldctx, weights, energies_out, (int)Bk);
        NM_LAUNCH_CHECK("nm_attn_fwd");
    }
    NM_REQUIRE(workspace_bytes >= nm_attn_workspace_bytes(R, S, C), "nm_attn_fwd: workspace too small");
    const int qpk = (int)(nq < 8 ? nq : 8);
    const int groups = (int)((nq + qpk - 1) / qpk);
    NM_REQUIRE(groups <= 65535 && Bk <= 65535, "nm_attn_fwd: grid too large");
    float* ws = reinterpret_cast<float*>(workspace);
    AttnArgs p;
    p.y = y; p.hf = hf; p.states = states; p.mask = mask; p.v = v; p.bias = bias;
    p.energies = energies_out ? energies_out : ws;   // kept by the caller for the backward pass
    p.pctx = ws + ((R * S + 3) / 4) * 4;
    p.pstat = p.pctx + R * nchunk * C;
    p.R = (int)R; p.S = (int)S; p.A = (int)A; p.C = (int)C; p.nchunk = nchunk; p.sch = sch;
    p.qsb = (int)q_stride_b; p.qsq = (int)q_stride_q; p.nq = (int)nq;
    p.tickets = reinterpret_cast<unsigned*>(p.pstat + R * nchunk * 4);
    p.ctx = ctx; p.ldctx = (long)ldctx; p.weights = weights; p.merge = 0; p.Bk = (int)Bk;
    const bool no_merge = nm_cur()->sw.attn_nomerge;                        // A/B switch: separate combine launch
    const bool may_merge = do_combine && !no_merge && nchunk <= ATT_MERGE_MAXCH;
    const size_t shm = sizeof(float) * ((size_t)qpk * A + A + (size_t)qpk * ATT_MAX_SCH + ATT_MAX_SCH +
                                        (size_t)4 * qpk * ATT_MAX_SCH);
    NM_REQUIRE(shm <= 160 * 1024, "nm_attn_fwd: A too large for LDS staging");
    hipStream_t st = nm_stream(stream);
    dim3 grid(nchunk, (unsigned)Bk, groups), block(256);
    const int ncg = C > 1024 ? 2 : 1;
#define NM_AT(Q_)                                                                         \
    do {                                                                                  \
        if (ncg == 1) hipLaunchKernelGGL((attn_partial<Q_, 1>), grid, block, shm, st, p); \
        else hipLaunchKernelGGL((attn_partial<Q_, 2>), grid, block, shm, st, p);          \
    } while (0)
    NmCtx* const nmc = nm_cur();
    std::pair<hipEvent_t, hipEvent_t>* prof = nmc->prof_on ? nm_prof_next_pair(nmc) : nullptr;
    if (prof) (void)hipEventRecord(prof->first, st);
    const bool no_fast = nmc->sw.attn_nofast;                              // A/B switch for tuning
    // Whole-sentence workgroups where they were measured faster than split-S + in-kernel merge (HIP events, cold /
    // warm, profiles/r02_attn_whole_vs_split.txt): 128 sentences x 50 positions 20.4 / 16.6 us against 21.4 / 18.7;
    // slower at 64 sentences (13.7 vs 13.2 warm), at 16 (12.8 vs 11.5) and at 30 positions (13.9 vs 13.4), where
    // too few CUs get a workgroup or a workgroup too little to stream.
    const int whole_sw = nmc->sw.attn_whole;                              // A/B switch: 0 off, 1 whenever possible, -1 unset
    const bool whole = do_combine && nq == 1 && A <= 1024 && C <= 1024 && S <= 4 * ATT_WHOLE_ROWS && !no_fast &&
                       (whole_sw >= 0 ? whole_sw != 0 : (Bk >= 96 && S >= 40));
    if (whole) {
        hipLaunchKernelGGL(attn_whole_fast<ATT_WHOLE_ROWS>, dim3((unsigned)Bk), dim3(1024), 0, st, p);
        if (prof) (void)hipEventRecord(prof->second, st);
        NM_LAUNCH_CHECK("nm_attn_fwd");
    }
    // wide values under narrow keys (captioning): two whole-sentence workgroups per sentence, split by value columns
    const bool wide = do_combine && nq == 1 && A <= 512 && C > 1024 && C <= 2048 && S <= 4 * ATT_WIDE_ROWS && !no_fast &&
                      (whole_sw >= 0 ? whole_sw != 0 : Bk >= 64);
    if (wide) {
        hipLaunchKernelGGL(attn_whole_wide, dim3((unsigned)(((Bk + 7) / 8) * 16)), dim3(1024), 0, st, p);
        if (prof) (void)hipEventRecord(prof->second, st);
        NM_LAUNCH_CHECK("nm_attn_fwd");
    }
    if (nq == 1 && A <= 1024 && C <= 1024 && sch <= ATT_FAST_ROWS && !no_fast) {
        p.merge = may_merge;
        if (sch <= 8) hipLaunchKernelGGL(attn_partial_fast<8>, grid, block, 0, st, p);
        else if (sch <= 10) hipLaunchKernelGGL(attn_partial_fast<10>, grid, block, 0, st, p);
        else if (sch <= 12) hipLaunchKernelGGL(attn_partial_fast<12>, grid, block, 0, st, p);
        else hipLaunchKernelGGL(attn_partial_fast<14>, grid, block, 0, st, p);
    } else if (nq == 1 && A <= 1024 && C <= 2048 && sch <= 12 && !no_fast) {
        // wide values (config 4: C = 2048): the same kernel with two value slices per lane, merged in the launch
        p.merge = may_merge;
        if (sch <= 8) hipLaunchKernelGGL((attn_partial_fast<8, 2>), grid, block, 0, st, p);
        else if (sch <= 10) hipLaunchKernelGGL((attn_partial_fast<10, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((attn_partial_fast<12, 2>), grid, block, 0, st, p);
    } else if (nq <= 8 && groups == 1 && A <= 1024 && C <= 1024 && sch <= ATT_FAST_ROWS && !no_fast) {
        // a few queries per key batch (beam search): all loads in flight first, queries in registers.  (No
        // in-kernel merge here: one workgroup merging the k rows of a sentence one after the other was measured
        // 10 us slower than the separate combine launch it saves, 40.6 vs 30.5 + 5.3 us at k = 5.)
#define NM_AQ(R_)                                                                                  \
        do {                                                                                       \
            if (nq <= 2) hipLaunchKernelGGL((attn_partial_fastq<R_, 2>), grid, block, 0, st, p);      \
            else if (nq <= 4) hipLaunchKernelGGL((attn_partial_fastq<R_, 4>), grid, block, 0, st, p); \
            else if (nq == 5) hipLaunchKernelGGL((attn_partial_fastq<R_, 5>), grid, block, 0, st, p); \
            else hipLaunchKernelGGL((attn_partial_fastq<R_, 8>), grid, block, 0, st, p);              \
        } while (0)
        if (sch <= 10) NM_AQ(10);
        else if (sch <= 12) NM_AQ(12);
        else NM_AQ(14);
#undef NM_AQ
    } else
    switch (qpk) {
        case 1: NM_AT(1); break;
        case 2: NM_AT(2); break;
        case 3: NM_AT(3); break;
        case 4: NM_AT(4); break;
        case 5: NM_AT(5); break;
        case 6: NM_AT(6); break;
        case 7: NM_AT(7); break;
        default: NM_AT(8); break;
    }
#undef NM_AT
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) NM_FAIL(NM_ERR_HIP, "nm_attn_fwd: partial launch failed: %s", hipGetErrorString(e));
    if (do_combine && !p.merge)
        hipLaunchKernelGGL(attn_combine, dim3((unsigned)R), dim3(256), 0, st, p.pctx, p.pstat, p.energies,
                           mask, ctx, (long)ldctx, weights, (int)S, (int)C, nchunk,
                           beam_layout ? (int)nq : 1, (int)Bk);
    if (prof) (void)hipEventRecord(prof->second, st);
    NM_LAUNCH_CHECK("nm_attn_fwd");
}

extern "C" int nm_attn_fwd_multi(void* stream, const float* y, const float* hf, const float* states,
                                 const float* mask, const float* v, const float* bias, int64_t Bk,
                                 int64_t nq, int64_t q_stride_b, int64_t q_stride_q, int64_t S, int64_t A,
                                 int64_t C, float* ctx, int64_t ldctx, float* weights, void* workspace,
                                 int64_t workspace_bytes, float* energies_out) {
    return attn_fwd_impl(stream, y, hf, states, mask, v, bias, Bk, nq, q_stride_b, q_stride_q, S, A, C, ctx, ldctx,
                         weights, workspace, workspace_bytes, energies_out, 1);
}

// Where the split-S partials of R query rows live inside the attention workspace (float offsets): energies
// [R,S] at 0, partial contexts [R,nchunk,C] at *pctx_off, statistics [R,nchunk,4] = {max, sum exp, sum exp*mask,
// -} at *pstat_off.  Returns <0 when the shape is served by the any-shape kernel (no partials exist).
extern "C" int nm_attn_partials_layout(int64_t R, int64_t S, int64_t A, int64_t C, int64_t* nchunk,
                                       int64_t* pctx_off, int64_t* pstat_off) {
    NM_REQUIRE(R > 0 && S > 0 && A > 0 && C > 0 && nchunk && pctx_off && pstat_off, "nm_attn_partials_layout: bad args");
    int sch, nch;
    if (!attn_vector_shape(S, A, C, &sch, &nch)) NM_FAIL(NM_ERR_ARG, "nm_attn_partials_layout: any-shape kernel");
    *nchunk = nch;
    *pctx_off = ((R * S + 3) / 4) * 4;
    *pstat_off = *pctx_off + R * nch * C;
    return NM_OK;
}

// The attention step WITHOUT its combine launch: energies (workspace offset 0) and the split-S partials are
// left in the workspace for a consumer that merges them on the fly (nm_step_group).  rows_per_key as nm_attn_fwd.
extern "C" int nm_attn_fwd_partials(void* stream, const float* y, const float* hf, const float* states,
                                    const float* mask, const float* v, const float* bias, int64_t R,
                                    int64_t rows_per_key, int64_t S, int64_t A, int64_t C, void* workspace,
                                    int64_t workspace_bytes) {
    NM_REQUIRE(R > 0 && rows_per_key >= 1 && rows_per_key <= 64 && R % rows_per_key == 0,
               "nm_attn_fwd_partials: bad shape R=%ld k=%ld", (long)R, (long)rows_per_key);
    return attn_fwd_impl(stream, y, hf, states, mask, v, bias, R / rows_per_key, rows_per_key, rows_per_key, 1, S, A,
                         C, nullptr, 4, nullptr, workspace, workspace_bytes, nullptr, 0);
}

extern "C" int nm_attn_fwd(void* stream, const float* y, const float* hf, const float* states,
                           const float* mask, const float* v, const float* bias, int64_t R,
                           int64_t rows_per_key, int64_t S, int64_t A, int64_t C, float* ctx,
                           int64_t ldctx, float* weights, void* workspace, int64_t workspace_bytes,
                           float* energies_out) {
    NM_REQUIRE(R > 0 && rows_per_key >= 1 && rows_per_key <= 64 && R % rows_per_key == 0,
               "nm_attn_fwd: bad shape R=%ld k=%ld", (long)R, (long)rows_per_key);
    return nm_attn_fwd_multi(stream, y, hf, states, mask, v, bias, R / rows_per_key, rows_per_key,
                             rows_per_key, 1, S, A, C, ctx, ldctx, weights, workspace, workspace_bytes,
                             energies_out);
}
