// The vocabulary projection of a DECODING step with its row statistics (decoders/autoregressive.py:450-459 +
// the argmax / log-softmax that follow it, :461-480; beam_search_decoder.py:537-543):
//     logits[M, N] = state[M, K] . W[K, N] + b,      M = sentences or hypotheses of a step (128 .. 640),
//     stats[row][tile] = {max, sum exp(x - max), first argmax}   per 128-column tile,
// as an ACTIVATION-STATIONARY stream over the weight matrix -- the shape gemm_tiled handles worst (one row of
// 128x128 tiles: every workgroup alone on its CU, a barrier and an LDS staging pass per 16 k, both operands through
// LDS: 46.6 us at 128 rows against 27.3 us of fp32 matrix-core time, 200 us against 133 at 640 rows).
//
//   * A workgroup = 8 waves = 128 rows; wave w owns rows 16 w .. 16 w + 15 and keeps them IN REGISTERS for the whole
//     launch (K / 4 registers per lane: the A operand of v_mfma_f32_16x16x4_f32 is one register per 4 k).  The state
//     matrix is read once per workgroup instead of once per column tile.
//   * The weights stream HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) in
//     chunks of CK k x 64 columns (CK = 64: 16 KB), three buffers, issued two chunks ahead by all eight waves.
//     The LDS image is lane-linear = [k][64 columns]: a B fragment for FOUR column blocks (columns n0 + 4 n + q) is
//     one ds_read_b128, four rows of 256 contiguous bytes -- conflict-free.
//   * One barrier per chunk (4096 matrix-pipe cycles at CK = 64), none inside it; no K split, no cross-wave
//     reduction: every output element is one chain of K / 4 four-product MFMA steps.
//   * A workgroup walks a RANGE of column tiles of its row tile (persistent: 256 workgroups for any M), so 640 rows
//     are 2500 block units over 255 workgroups = 10 each (136 us of matrix time) instead of 2.44 rounds of 128x128
//     tiles.
//   * Epilogue from the accumulators: a lane holds 4 consecutive columns of 4 rows -> float4 stores of the logits
//     (256 contiguous bytes per row and instruction); statistics by an online max / sum-exp per lane over the tile's
//     two 64-column blocks, merged over the 16 lanes of a row on the DPP crossbar.  The bias sits in LDS, so the loop
//     has no ordinary global load whose s_waitcnt would drain the LDS-DMA queue.
//
// Arithmetic: exact fp32 products, fp32 accumulation (the reference's tf.matmul on the CPU accumulates in fp32 in
// its own order); the order differs from gemm_tiled's 32x32x2 chain, so logits agree with nm_gemm_f32 to rounding
// (tests/test_logits_stats_gpu.py: both against float64), not bit for bit.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "nm_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PJ_NBUF 3
#define PJ_BIAS_MAX 4096                  // columns of one workgroup's range (bias staged in LDS)

struct ProjArgs {
    const float* A; long lda;
    const float* W; long ldw;
    const float* bias;
    float* C; long ldc;
    float* stats;                         // [M][ntile][4]
    int M, N;
    int tiles_m, ntile, ncr, tpr;         // row tiles; 128-column tiles; column ranges; tiles per range
    int swz;                              // grid % 8 == 0: workgroups of one XCD take consecutive (row tile, range) units
    long* dbg;                            // NM_PROJ_ASTAT_DBG_PTR: per-chunk clock stamps of two workgroups (tools/proj_astat_probe.py stamps)
    int ablate;                           // timing ablations (NM_PROJ_ASTAT_ABLATE): 1 no weight stream, 2 no matrix work,
                                          // 8 no per-chunk wait + barrier, 16 no statistics arithmetic, 32 no sum exp
};

// one LDS-DMA piece: 64 lanes x 16 bytes -> 1 KB at the wave-uniform LDS byte address ``dst``
__device__ __forceinline__ void pj_dma16(const float* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// (value desc, index asc) maximum of the 16 lanes of a DPP row; every lane gets the result
template <int CTRL>
__device__ __forceinline__ void pj_argmax_step(float& v, int& i) {
    const float ov = nm_dpp<CTRL, 0xf>(v, v);
    const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false);
    const bool take = ov > v || (ov == v && oi < i);
    v = take ? ov : v;
    i = take ? oi : i;
}
__device__ __forceinline__ float pj_row16_sum(float v) {
    v += nm_dpp<0xB1, 0xf>(0.0f, v);
    v += nm_dpp<0x4E, 0xf>(0.0f, v);
    v += nm_dpp<0x141, 0xf>(0.0f, v);
    v += nm_dpp<0x140, 0xf>(0.0f, v);
    return v;
}

template <typename F, int... Cs>
__device__ __forceinline__ void pj_for_each(F& f, std::integer_sequence<int, Cs...>) {
    (f(std::integral_constant<int, Cs>{}), ...);
}

// KC: K / 128; CK: k-rows per chunk (64: a 16 KB image; 128 measured no faster)
//
// What was measured on the way (profiles/r06_proj_astat_*.txt): the two waves that share a SIMD share its matrix pipe
// AND its vector issue, so everything a wave does besides MFMAs -- LDS-DMA issue (~150 cycles a piece), the
// statistics arithmetic -- comes out of the pair's matrix time wherever it is placed: doing the epilogue of the two
// waves half a chunk apart, or row by row during the next block from an LDS copy of the accumulators, moved the
// time around and added instructions (19.6 -> 20.2 us per 64-column block).  The shader clock under this mix of
// MFMA + LDS + HBM traffic is 2.12 GHz (2.40 in a bare MFMA loop): the matrix time of a block is 15.4 us, not 13.65.
// So: the simplest schedule, and as few vector instructions per output as the statistics allow.
template <int KC, int CK>
__global__ __launch_bounds__(512, 2) void proj_astat_kernel(ProjArgs g) {
    constexpr int BUF = CK * 64;                 // floats of one chunk image: CK k-rows x 64 columns
    constexpr int NCB = 128 * KC / CK;           // chunks per 64-column block
    constexpr int RPW = CK / 8;                  // k-rows of a chunk that one wave fetches
    constexpr int S = CK / 4;                    // MFMA steps per chunk
    constexpr int D = 2;                         // B fragments in flight (S % D == 0: the ring runs on across chunks)
    static_assert(S % D == 0, "the fragment ring must close over a chunk");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const long t_start = g.dbg ? (long)wall_clock64() : 0;
    const long c_start = g.dbg ? (long)clock64() : 0;             // shader clock: (c_end - c_start) / wall time = the frequency
    float* bias_s = lds + PJ_NBUF * BUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;

    int v = (int)blockIdx.x;
    if (g.swz) v = (v & 7) * ((int)gridDim.x >> 3) + (v >> 3);
    const int rt = v % g.tiles_m, cr = v / g.tiles_m;
    const int tile0 = cr * g.tpr, tile1 = min(g.ntile, tile0 + g.tpr);
    if (tile0 >= tile1) return;
    const int m0 = rt * 128;
    const int nblk = 2 * (tile1 - tile0);
    const int nchunk = nblk * NCB;
    const int col_first = tile0 * 128;

    // LDS-DMA source of piece u of chunk gi: rows RPW wave + 4 u + kq of the chunk, columns 4 n .. 4 n + 3 of the block
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;   // LDS byte address
    auto issue = [&](int gi) {
        if (gi >= nchunk || (g.ablate & 1)) return;
        const int blk = gi / NCB, c = gi - blk * NCB;
        const int nb = col_first + 64 * blk;
        const int col = min(nb + 4 * n, g.N - 4);                 // (the tail block: clamped, masked in the epilogue)
        const float* src = g.W + (long)(CK * c + RPW * wave + kq) * g.ldw + col;
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            lds_base + (unsigned)((gi % PJ_NBUF) * BUF + (RPW * wave) * 64) * 4u);
#pragma unroll
        for (int u = 0; u < RPW / 4; ++u) pj_dma16(src + (long)(4 * u) * g.ldw, dst + (unsigned)(4 * u * 64 * 4));
    };
    issue(0);
    issue(1);

    // the bias of this range, pre-multiplied copies are not needed: it is added to the logits as it is
    {
        const int ncol = min(g.N, tile1 * 128) - col_first;
        for (int i = tid; i < ncol; i += 512) bias_s[i] = g.bias ? g.bias[col_first + i] : 0.0f;
    }
    // this wave's 16 rows, for the whole launch: a[4 j + q] = A[row][16 j + 4 kq + q].  They arrive chunk by chunk
    // while the first block is computed (the k-slice of chunk c + 2 is requested when chunk c starts): a CU ingests
    // its rows (256 KB at K = 512) at the rate it fills its L1 -- ~10 us if the matrix work had to wait for all of it.
    // (Rows past M re-read row M - 1: their outputs are never stored and rows do not mix.)
    float a[32 * KC];
    const float* a_src = g.A + (long)min(m0 + 16 * wave + n, g.M - 1) * g.lda + 4 * kq;
    auto load_a = [&](int c) {                                    // the registers chunk c multiplies
#pragma unroll
        for (int jj = 0; jj < CK / 16; ++jj) {
            const int j = (CK / 16) * c + jj;
            const float4 x = *reinterpret_cast<const float4*>(a_src + 16 * j);
            a[4 * j] = x.x; a[4 * j + 1] = x.y; a[4 * j + 2] = x.z; a[4 * j + 3] = x.w;
        }
    };
#pragma unroll
    for (int j = 0; j < 32 * KC; ++j) a[j] = 0.0f;
    load_a(0);
    if (NCB > 1) load_a(1);

    f32x4 acc[4];
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = zero;
    // running statistics of the current 128-column tile, per row r of this lane (rows 4 kq + r): the lane's maximum so
    // far (first occurrence) and sum 2^((x - that maximum) log2 e)
    constexpr float L2E = 1.4426950408889634f;
    float rs[4], bv[4];
    int bi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { rs[r] = 0.0f; bv[r] = -INFINITY; bi[r] = 0x7fffffff; }

    const int row_lane0 = m0 + 16 * wave + 4 * kq;                // first of this lane's four rows
    auto epilogue = [&](int blk) {
        const int nb = col_first + 64 * blk;
        const int col = nb + 4 * n;
        const bool okc = col < g.N;                                // N % 4 == 0: the whole float4 or nothing
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (okc) b4 = *reinterpret_cast<const float4*>(bias_s + (col - col_first));
        if (!okc) b4.x = b4.y = b4.z = b4.w = -INFINITY;           // columns past N: never a maximum, exp -> 0
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x0 = acc[0][r] + b4.x, x1 = acc[1][r] + b4.y, x2 = acc[2][r] + b4.z, x3 = acc[3][r] + b4.w;
            const int row = row_lane0 + r;
            if (g.C && okc && row < g.M)
                *reinterpret_cast<float4*>(g.C + (long)row * g.ldc + col) = make_float4(x0, x1, x2, x3);
            if (g.ablate & 16) continue;
            // first maximum of the four (ascending columns), then against the running one (strict >: earlier wins)
            const float m01 = fmaxf(x0, x1), m23 = fmaxf(x2, x3), m4 = fmaxf(m01, m23);
            const int i4 = col + (x0 == m4 ? 0 : (x1 == m4 ? 1 : (x2 == m4 ? 2 : 3)));
            const float before = bv[r];
            const bool up = m4 > before;
            bv[r] = up ? m4 : before;
            bi[r] = up ? i4 : bi[r];
            if (g.ablate & 32) continue;
            // sum exp(x - max) in base 2: one fma + one v_exp_f32 per value.  The fma subtracts ROUNDED max * log2(e)
            // from the exact product, so a lane's sum is 2^lo times the true one, lo = max * log2(e) - rounded(...) (up
            // to 2.4e-6 at |max| = 40: it showed in the log-sum-exp); the re-scaling below keeps that statement true
            // when the maximum moves (a difference of two ROUNDED products) and the tile merge divides 2^lo out.  (max
            // = -inf only while every column so far was past N: the products below are NaN then and the sum is
            // re-started by the select)
            const float ml = __fmul_rn(bv[r], L2E);
            float sum = rs[r] * __builtin_amdgcn_exp2f(__fsub_rn(__fmul_rn(before, L2E), ml));
            sum += __builtin_amdgcn_exp2f(fmaf(x0, L2E, -ml));
            sum += __builtin_amdgcn_exp2f(fmaf(x1, L2E, -ml));
            sum += __builtin_amdgcn_exp2f(fmaf(x2, L2E, -ml));
            sum += __builtin_amdgcn_exp2f(fmaf(x3, L2E, -ml));
            rs[r] = bv[r] > -INFINITY ? sum : 0.0f;
        }
        if (blk & 1) {                       // the tile is complete: merge the 16 lanes of each row, write, reset
            const int tile = tile0 + (blk >> 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float mv = bv[r];
                int mi = bi[r];
                pj_argmax_step<0xB1>(mv, mi);
                pj_argmax_step<0x4E>(mv, mi);
                pj_argmax_step<0x141>(mv, mi);
                pj_argmax_step<0x140>(mv, mi);
                const float lo = fmaf(bv[r], L2E, -__fmul_rn(bv[r], L2E));        // exact: what the rounding dropped
                const float mine = bv[r] > -INFINITY ? rs[r] * __builtin_amdgcn_exp2f((bv[r] - mv) * L2E - lo) : 0.0f;
                const float tot = pj_row16_sum(mine);
                const int row = row_lane0 + r;
                if (n == 0 && row < g.M) {
                    float4 rec;
                    rec.x = mv; rec.y = tot; rec.z = __int_as_float(mi); rec.w = 0.0f;
                    *reinterpret_cast<float4*>(g.stats + ((long)row * g.ntile + tile) * 4) = rec;
                }
                rs[r] = 0.0f; bv[r] = -INFINITY; bi[r] = 0x7fffffff;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = zero;
    };

    // One continuous stream of MFMA steps over the chunks; the B fragment of a step is read from LDS D steps ahead,
    // across chunk boundaries too: chunk gi + 1 is complete in LDS from the barrier that opens chunk gi (every wave
    // waited for its pieces before it), so the tail of chunk gi reads the head of chunk gi + 1 and the barrier
    // between them costs no LDS round trip.
    const float* bl = lds + (4 * kq) * 64 + 4 * n;                 // this lane's corner of a chunk image
    auto frag = [&](const float* bp, int s) { return *reinterpret_cast<const float4*>(bp + (16 * (s >> 2) + (s & 3)) * 64); };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // chunks 0, 1 (this wave's pieces), the bias, the rows
    __builtin_amdgcn_s_barrier();
    float4 ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = frag(bl, d);
    int gi = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        // (a fold over the chunk index, not a loop: hipcc gives up unrolling the loop at 8 chunks, and the row
        // registers a[] must be indexed by constants)
        auto chunk = [&](auto c_tag) {
            constexpr int c = decltype(c_tag)::value;
            if (gi > 0 && !(g.ablate & 8)) {
                // every DMA piece this wave issued (chunks <= gi + 1, the youngest a whole chunk of matrix work ago)
                // and every logits store has landed; behind the barrier that holds for all waves, and everybody is
                // done reading chunk gi - 1, whose buffer chunk gi + 2 overwrites
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (g.dbg && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 101) && gi < 60)
                g.dbg[(blockIdx.x ? 64 : 0) + gi] = (long)wall_clock64();
            if (blk == 0 && c + 2 < NCB) load_a(c + 2);          // (the first block: the rows are still arriving)
            issue(gi + 2);
            if (c == 0 && blk > 0) epilogue(blk - 1);
            const float* bp = bl + (gi % PJ_NBUF) * BUF;
            const float* bn = bl + ((gi + 1) % PJ_NBUF) * BUF;
            if (g.ablate & 2) { ++gi; return; }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int jj = s >> 2, q = s & 3;
                const float4 b = ring[s % D];
                ring[s % D] = (s + D < S) ? frag(bp, s + D) : frag(bn, s + D - S);
                const float av = a[4 * ((CK / 16) * c + jj) + q];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b.w, acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // the read of step s + D ...
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // ... then this step's four MFMAs
            }
            ++gi;
        };
        pj_for_each(chunk, std::make_integer_sequence<int, NCB>{});
    }
    epilogue(nblk - 1);
    if (g.dbg && tid == 0 && (blockIdx.x == 0 || blockIdx.x == 101)) {
        g.dbg[(blockIdx.x ? 64 : 0) + 60] = (long)wall_clock64();
        g.dbg[(blockIdx.x ? 64 : 0) + 61] = t_start;
        g.dbg[(blockIdx.x ? 64 : 0) + 62] = nchunk;
        g.dbg[(blockIdx.x ? 64 : 0) + 63] = (long)clock64() - c_start;
    }
}

static int pj_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}

template <typename Kern>
static bool pj_launch(Kern kern, int slot, int dev, unsigned grid, size_t lds, hipStream_t st, const ProjArgs& g) {
    static std::atomic<unsigned> done[8];
    const unsigned bit = 1u << (dev & 15);
    if (!(done[slot].load(std::memory_order_relaxed) & bit)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        done[slot].fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, g);
    return true;
}

// true when the product was launched.  Taken for W stored [K, N] (n contiguous), K = 128, 256, 384 or 512, 128-column
// statistics tiles, and few enough row tiles that every one gets column ranges of its own.  NM_PROJ_ASTAT=0: never.
bool nm_proj_astat_try(hipStream_t st, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc, float* stats,
                       int stats_tile) {
    static const int on = pj_env("NM_PROJ_ASTAT", 1), ablate = pj_env("NM_PROJ_ASTAT_ABLATE", 0);
    if (!on || trans_b || stats_tile != 128) return false;
    if (K % 128 != 0 || K < 128 || K > 512 || N % 4 != 0 || N < 64) return false;
    if (!nm_aligned16(A) || lda % 4 != 0 || !nm_aligned16(W) || ldw % 4 != 0) return false;
    if (C && (!nm_aligned16(C) || ldc % 4 != 0)) return false;
    int ncu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) return false;
    ProjArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.stats = stats;
    g.M = (int)M; g.N = (int)N;
    g.tiles_m = nm_cdiv(M, 128);
    g.ntile = nm_cdiv(N, 128);
    if (g.tiles_m > ncu / 4) return false;                        // hundreds of row tiles: the tiled kernels' shape
    const int want = ncu / g.tiles_m;                              // column ranges per row tile
    g.tpr = nm_cdiv(g.ntile, want < g.ntile ? want : g.ntile);
    if (g.tpr * 128 > PJ_BIAS_MAX) return false;
    g.ncr = nm_cdiv(g.ntile, g.tpr);
    const unsigned grid = (unsigned)(g.tiles_m * g.ncr);
    g.swz = (grid % 8 == 0) ? 1 : 0;
    g.ablate = ablate;
    g.dbg = nullptr;
    if (const char* e = getenv("NM_PROJ_ASTAT_DBG_PTR")) g.dbg = reinterpret_cast<long*>(strtoull(e, nullptr, 0));
    const int kc = (int)(K / 128);
    const size_t lds = (size_t)(PJ_NBUF * 64 * 64 + PJ_BIAS_MAX) * sizeof(float);   // chunk images + the bias of the range
    switch (kc) {
        case 1: return pj_launch(proj_astat_kernel<1, 64>, 0, dev, grid, lds, st, g);
        case 2: return pj_launch(proj_astat_kernel<2, 64>, 1, dev, grid, lds, st, g);
        case 3: return pj_launch(proj_astat_kernel<3, 64>, 2, dev, grid, lds, st, g);
        default: return pj_launch(proj_astat_kernel<4, 64>, 3, dev, grid, lds, st, g);
    }
}
