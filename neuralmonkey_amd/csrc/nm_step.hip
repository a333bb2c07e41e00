// Decoder-step GEMM groups: the dense products of ONE inference step of the RNN attention decoder
// (Decoder.next_state, decoders/decoder.py:279-358) launched as a few groups of skinny GEMMs whose
// epilogues / operand loaders absorb everything that is not a matrix product:
//
//   group 1  [emb | h] . Wg + bg -> r, u, r*h        (GRUCell gates, nn/ortho_gru_cell.py:44-49: ONE product
//            emb . Wc_x + bc     -> xc                over the concatenated [inputs, state], as TF does)
//   group 2  (r*h) . Wc_h + xc   -> c, h'            (candidate + blend, :50-53)
//   group 3  h' . Wq + bq        -> y                 (attention query projection, feed_forward.py:130-132)
//            [emb | h'] . Wo_eh  -> P                 (the part of the output projection that does not wait
//                                                      for the context, output_projection.py:115-130)
//   (nm_attn_fwd_partials: split-S attention partial kernel, no combine launch)
//   group 4  ctx . Wo_c + P + bo -> tanh -> out       with ctx assembled WHILE the A operand is loaded:
//            ctx[r,:] = sum_i f_i pctx[r,i,:] / den   (the merge of the split-S partials, attn_combine's
//            arithmetic); the same workgroups write the step's attention weights.
//
// Every product is M x N x K with M = rows of the step (128 sentences, 640 beam hypotheses): too small to
// fill the chip with 128x128 tiles and latency-bound, so -- like gemm_skinny16 -- a workgroup owns one
// 16x16 output tile, its 16 waves split K (v_mfma_f32_16x16x4_f32, exact f32), partial sums meet in LDS.
// Weights are passed TRANSPOSED ([N,K], k contiguous; the stepper transposes them once per decoding run), so
// both fragments of a wave are single 16-byte loads and all loads of a wave are in flight before its first
// MFMA.  Several products that do not depend on each other share one launch (a "group").
#include "nm_common.h"

#include <atomic>
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NM_STEP_MAX_PROB 3
#define NM_STEP_MAX_CHUNK 8
#define NM_Z4 make_float4(0.f, 0.f, 0.f, 0.f)

struct StepProb {           // mirrors nm_step_problem (include/nmhip.h)
    const float* A; long lda;
    const float* Bt; long ldb;
    long N, K;
    int a_kind, epilogue, act, nchunk;
    const float* bias; const float* add; long ldadd;
    float* C; long ldc;
    const float* pctx; const float* pstat;
    const float* energies; const float* mask; float* weights; long S, mask_div, mask_mod;
    const float* h; long ldh;
    float* ru; float* rh;
    const float* xc; long ldxc;
    float* h_out; long ldho; float* h_out2; long ldho2;
    const int* add_ids; const int* xc_ids;
};

struct StepGroup {
    StepProb p[NM_STEP_MAX_PROB];
    int begin[NM_STEP_MAX_PROB + 1];      // first workgroup of every problem
    int nprob, M, tiles_m;
    int wblocks;                          // leading workgroups that write the attention weights (a_kind 1)
};

// scale of partial i of query row `row`: f_i / den with f_i = exp(m_i - M), den = sum f_i lm_i + 1e-8 sum f_i la_i
// (attention/feed_forward.py:139-144 carried through the split-S statistics, as attn_combine does)
__device__ __forceinline__ void step_row_scales(const StepProb& p, int row, float (&sc)[NM_STEP_MAX_CHUNK],
                                                float& M, float& inv) {
    const float4* st = reinterpret_cast<const float4*>(p.pstat) + (long)row * p.nchunk;
    float4 sv[NM_STEP_MAX_CHUNK];
    M = -INFINITY;
#pragma unroll
    for (int i = 0; i < NM_STEP_MAX_CHUNK; ++i) {
        sv[i] = i < p.nchunk ? st[i] : make_float4(-INFINITY, 0.f, 0.f, 0.f);
        M = fmaxf(M, sv[i].x);
    }
    float la = 0.0f, lm = 0.0f;
#pragma unroll
    for (int i = 0; i < NM_STEP_MAX_CHUNK; ++i) {
        sc[i] = i < p.nchunk ? __expf(sv[i].x - M) : 0.0f;
        la += sc[i] * sv[i].y;
        lm += sc[i] * sv[i].z;
    }
    inv = 1.0f / (lm + 1e-8f * la);
}

// KS waves split K; TM 16-row MFMA tiles per workgroup (they share the weight fragments); AKIND: operand loader
// of the group (all problems of a launch share it).  These launches are pure latency: every workgroup of a
// group should be resident at once and every wave should pay ONE memory round trip, so the plain loader is held
// to 64 VGPRs (8 waves per SIMD, two 1024-thread workgroups per CU) and the host picks the tile height so that
// a group has at most 512 workgroups.
template <int KS, int TM, int AKIND>
__global__ __launch_bounds__(KS * 64, AKIND == 0 ? 8 : 4) void step_group_kernel(StepGroup g) {
    __shared__ float red[KS][TM * 4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = (int)blockIdx.x;
    if constexpr (AKIND == 1) {
        // the step's attention distribution (feed_forward.py:139-144): the FIRST workgroups of the grid, 16 query
        // rows each -- short, dispatched first, out of the way of the GEMM tiles
        const StepProb& q = g.p[0];
        if (bid < g.wblocks) {
            const int S = (int)q.S;
            for (int idx = tid; idx < 16 * S; idx += KS * 64) {
                const int row = bid * 16 + idx / S, sidx = idx % S;
                if (row >= g.M) break;
                float sc[NM_STEP_MAX_CHUNK], M, inv;
                step_row_scales(q, row, sc, M, inv);
                const long b = (row / q.mask_div) % q.mask_mod;
                const float mk = q.mask ? q.mask[b * S + sidx] : 1.0f;
                q.weights[(long)row * S + sidx] = __expf(q.energies[(long)row * S + sidx] - M) * mk * inv;
            }
            return;
        }
        bid -= g.wblocks;
    }
    int pi = 0;
#pragma unroll
    for (int i = 1; i < NM_STEP_MAX_PROB; ++i)
        if (i < g.nprob && bid >= g.begin[i]) pi = i;
    const StepProb& p = g.p[pi];
    const int tile = bid - g.begin[pi];
    const int bm = tile % g.tiles_m, bn = tile / g.tiles_m;
    const int m0 = bm * 16 * TM, n0 = bn * 16;
    const int N = (int)p.N, K = (int)p.K;

    const int i16 = lane & 15, kq = lane >> 4;
    const int nn = min(n0 + i16, N - 1);
    const int kper = ((K / 16 + KS - 1) / KS) * 16;
    const int kbeg = wave * kper, kend = min(K, kbeg + kper);

    // epilogue operands of this thread's output element are requested before the operand fragments, so their
    // latency hides under the main loop instead of following the LDS reduction
    const int e_t = tid >> 8;                                           // row tile of this thread's output element
    const int e_col = n0 + (tid & 15), e_row = m0 + e_t * 16 + 4 * ((tid & 63) >> 4) + ((tid >> 6) & 3);
    const bool e_ok = tid < 256 * TM && e_row < g.M && e_col < N;
    float e_bias = 0.0f, e_x = 0.0f, e_h = 0.0f, e_u = 0.0f;
    if (e_ok) {
        if (p.epilogue == 0) {
            if (p.bias) e_bias = p.bias[e_col];
            if (p.add) e_x = p.add[(long)(p.add_ids ? p.add_ids[e_row] : e_row) * p.ldadd + e_col];
        } else if (p.epilogue == 1) {
            e_bias = p.bias[e_col];
            if (p.add) e_x = p.add[(long)(p.add_ids ? p.add_ids[e_row] : e_row) * p.ldadd + e_col];
            if (e_col < (N >> 1)) e_h = p.h[(long)e_row * p.ldh + e_col];
        } else {
            e_x = p.xc[(long)(p.xc_ids ? p.xc_ids[e_row] : e_row) * p.ldxc + e_col];
            e_u = p.ru[(long)e_row * 2 * N + N + e_col];
            e_h = p.h[(long)e_row * p.ldh + e_col];
        }
    }

    f32x4 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = 0.0f;
    const float* bp = p.Bt + (long)nn * p.ldb + 4 * kq;

    if constexpr (AKIND == 0) {
        static_assert(TM <= 2, "one or two row tiles per workgroup");
        const float* ap0 = p.A + (long)min(m0 + i16, g.M - 1) * p.lda + 4 * kq;
        const float* ap1 = p.A + (long)min(m0 + (TM - 1) * 16 + i16, g.M - 1) * p.lda + 4 * kq;
        constexpr int CPI = TM == 1 ? 4 : 2;          // 16-deep chunks in flight per trip: what 64 VGPRs hold
        for (int k0 = kbeg; k0 < kend; k0 += 16 * CPI) {
            float4 av[TM][CPI], bv[CPI];
#pragma unroll
            for (int c = 0; c < CPI; ++c) {
                const int k = k0 + 16 * c;
                const bool ok = k + 4 * kq < kend;
                bv[c] = ok ? *reinterpret_cast<const float4*>(bp + k) : NM_Z4;
#pragma unroll
                for (int t = 0; t < TM; ++t)
                    av[t][c] = ok ? *reinterpret_cast<const float4*>((t == 0 ? ap0 : ap1) + k) : NM_Z4;
            }
#pragma unroll
            for (int c = 0; c < CPI; ++c) {
                if (k0 + 16 * c >= kend) break;
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][c].x, bv[c].x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][c].y, bv[c].y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][c].z, bv[c].z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][c].w, bv[c].w, acc[t], 0, 0, 0);
                }
            }
        }
    } else {
        // A[row, k] = sum_i scale_i * pctx[row, i, k]: the merge of the split-S attention partials happens in
        // the operand loader, the context vector itself is never written.  TM == 1; up to 5 partials per row (the
        // usual S <= 60) or up to 8, two 16-deep chunks of the wave's K slice in flight per trip.
        static_assert(AKIND == 0 || TM == 1, "the partial-merging loader owns one row tile");
        const int mm = min(m0 + i16, g.M - 1);
        float sc[NM_STEP_MAX_CHUNK], M, inv;
        step_row_scales(p, mm, sc, M, inv);
#pragma unroll
        for (int i = 0; i < NM_STEP_MAX_CHUNK; ++i) sc[i] *= inv;
        const float* pp = p.pctx + (long)mm * p.nchunk * K + 4 * kq;
        auto run = [&](auto cpi_tag, auto np_tag) {
            constexpr int CPI = decltype(cpi_tag)::value, NP = decltype(np_tag)::value;
            for (int k0 = kbeg; k0 < kend; k0 += 16 * CPI) {
                float4 pv[CPI][NP], bv[CPI];
#pragma unroll
                for (int c = 0; c < CPI; ++c) {
                    const int k = k0 + 16 * c;
                    const bool ok = k + 4 * kq < kend;
                    bv[c] = ok ? *reinterpret_cast<const float4*>(bp + k) : NM_Z4;
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        pv[c][i] = (ok && i < p.nchunk) ? *reinterpret_cast<const float4*>(pp + (long)i * K + k) : NM_Z4;
                }
#pragma unroll
                for (int c = 0; c < CPI; ++c) {
                    if (k0 + 16 * c >= kend) break;
                    float4 a = NM_Z4;
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        a.x += sc[i] * pv[c][i].x; a.y += sc[i] * pv[c][i].y;
                        a.z += sc[i] * pv[c][i].z; a.w += sc[i] * pv[c][i].w;
                    }
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv[c].x, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv[c].y, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bv[c].z, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bv[c].w, acc[0], 0, 0, 0);
                }
            }
        };
        if (p.nchunk <= 5) run(std::integral_constant<int, 2>{}, std::integral_constant<int, 5>{});
        else run(std::integral_constant<int, 2>{}, std::integral_constant<int, NM_STEP_MAX_CHUNK>{});
    }
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][t * 4 + i][lane] = acc[t][i];
    __syncthreads();
    if (e_ok) {
        const int reg = e_t * 4 + ((tid >> 6) & 3), ln = tid & 63;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < KS; ++w) s += red[w][reg][ln];
        const int col = e_col, row = e_row;
        if (p.epilogue == 0) {
            float v = s + e_bias + e_x;
            if (p.act == 1) v = nm_tanh(v);
            p.C[(long)row * p.ldc + col] = v;
        } else if (p.epilogue == 1) {              // N = 2H: r | u = sigmoid(. + bg); rh = r * h
            const int H = N >> 1;
            const float gate = nm_sigmoid(s + e_bias + e_x);
            p.ru[(long)row * N + col] = gate;
            if (col < H) p.rh[(long)row * H + col] = gate * e_h;
        } else {                                   // N = H: c = tanh(xc + .); h' = u*h + (1-u)*c
            const float c = nm_tanh(e_x + s);
            const float hn = e_u * e_h + (1.0f - e_u) * c;
            p.h_out[(long)row * p.ldho + col] = hn;
            if (p.h_out2) p.h_out2[(long)row * p.ldho2 + col] = hn;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Medium M (beam search: batch x beam = 640 hypotheses per step).  At a few hundred rows the 16-row tiles above
// re-read every weight M/16 times and a group needs thousands of 1024-thread workgroups (measured 52 us per
// group at 640 rows), while the LDS-tiled GEMMs put ONE 4-wave workgroup on a CU with nothing to hide its
// global -> LDS -> MFMA chain behind (64x64 tiles: 23.5 us for a 1 GFLOP product = 3.4x its MFMA time).  Here a
// workgroup owns one 32x32 output tile (v_mfma_f32_32x32x2_f32, exact f32), its KS waves split K, both fragments
// of a wave arrive as coalesced 128-byte lines through a wave-private LDS tile (no workgroup barrier in the loop),
// the next trip's lines are requested before the current trip's MFMAs, partial sums meet in LDS.  A group of the
// 640-row step is 640..1280 workgroups of 4 waves (32 KB LDS each); weights are read M/32 times from L2.  Same
// problems, epilogues and results as step_group_kernel (a_kind 0).
// What bounds these kernels is the rate at which a CU fills its L1 from L2 -- ~10 B / clk / CU, the figure the CDNA4
// guide gives for streaming loads (outstanding misses x 128 B / latency) -- not the matrix pipes and not L2 bandwidth:
// at 8 flop per operand byte (a 32x32 tile per K-split wave set) that is 80 flop / clk / CU = 31 % of the fp32 MFMA
// rate, which is what every variant measured (group 1 of the 640-row step, 2 GFLOP: 39-40 us; PMC passes in
// profiles/r03_step_group_medium_pmc.txt: matrix pipes 31 % busy, waves stalled, TCP pending-miss stalls): lane-per-
// row loads (45 us: additionally one address-unit pass per lane), coalesced lines through wave-private LDS tiles,
// 4 or 5 workgroups per CU, an explicit three-stage software pipeline (193 VGPRs, 39.1 us), operand rows padded by 128
// bytes against L2-channel aliasing (no change).  The lever is operand re-use INSIDE a CU: TR x TC sub-tiles per
// workgroup share rows through L1 hits (64x64: 32.0 us); an LDS-shared 64x64 tile would not go below ~25 us either
// ((64 + 64) rows x K x 4 B per CU at 10 B / clk), and 128x128 tiles leave too few workgroups at 640 rows.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// TR x TC sub-tiles of 32x32 per workgroup (each with its own KS waves): the same arithmetic in fewer, fatter
// workgroups.
// BT: the second operand is stored [N, K] (transposed weights of the decoder step); !BT: [K, N] as nm_gemm_f32 takes it
// (its tile is then fetched 8 k-rows x 128 bytes per instruction and read from LDS one float per MFMA).
template <int KS, int TR, int TC, bool BT = true>
__global__ __launch_bounds__(KS * TR * TC * 64, (KS * TR * TC == 4) ? (BT ? 5 : 4) : (KS * TR * TC == 8 ? 2 : 1))
void step_group_medium_kernel(StepGroup g) {
    constexpr int NW = KS * TR * TC;
    // Tile rows are exactly one 128-byte line (no padding: 8 KB per wave, 32 KB per 4-wave workgroup = FIVE
    // workgroups per CU, which is what lets the 1280 workgroups of group 1 at 640 rows run as one balanced round).
    // Bank conflicts are avoided by storing 16-byte chunk q of row r at chunk position q ^ ((r >> 1) & 7): the 16
    // lanes of a ds_read_b128 group hold 16 different values of r & 15, i.e. 16 different (row parity, chunk
    // position) pairs = all 64 banks; the 8 lanes of a ds_write_b128 group write the 8 chunks of one row.
    constexpr int LDT = 32;
    __shared__ __attribute__((aligned(16))) float lds[NW * 2 * 32 * LDT];
    static_assert(NW * 2 * 32 * LDT >= NW * 16 * 64, "the K reduction re-uses the operand tiles");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = wave / KS, ks = wave - sub * KS;          // sub-tile of this wave, its K slice
    const int sr = sub % TR, sc = sub / TR;
    const int bid = (int)blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < NM_STEP_MAX_PROB; ++i)
        if (i < g.nprob && bid >= g.begin[i]) pi = i;
    const StepProb& p = g.p[pi];
    const int tile = bid - g.begin[pi];
    const int bm = tile % g.tiles_m, bn = tile / g.tiles_m;         // consecutive workgroups share a weight tile
    const int m0 = (bm * TR + sr) * 32, n0 = (bn * TC + sc) * 32;
    const int N = (int)p.N, K = (int)p.K;
    // A trip = 32 consecutive k of the wave's K slice = one 128-byte line of each of the 32 + 32 operand rows.  The
    // lines are fetched COALESCED (8 lanes x 16 bytes per row, 8 rows per instruction) and turned into MFMA fragments
    // through a wave-private LDS tile: letting every lane fetch its own row (lane = row, as the 16-row kernels above
    // do) costs one address-unit pass per LANE -- measured 45 us for group 1 of the 640-row step against 13 us of
    // MFMA time, with either 16 or 64 contiguous bytes per lane and trip.
    const int kper = (((K + 31) / 32 + KS - 1) / KS) * 32;        // (K may end on a half trip: K % 32 == 16)
    const int kbeg = ks * kper, kend = min(K, kbeg + kper);
    const int lr = lane >> 3, lq = lane & 7;
    long aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aoff[i] = (long)min(m0 + lr + 8 * i, g.M - 1) * p.lda + 4 * lq;
        boff[i] = BT ? (long)min(n0 + lr + 8 * i, N - 1) * p.ldb + 4 * lq
                     : (long)(lr + 8 * i) * p.ldb + min(n0 + 4 * lq, N - 4);     // row = k, 16-byte chunk of 4 columns
    }
    float* as = lds + wave * (2 * 32 * LDT);
    float* bs = as + 32 * LDT;
    const int m = lane & 31, half = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    float4 ra[4], rb[4];
    auto fetch = [&](int k0) {
        const bool ok = k0 + 4 * lq < kend;         // K is a multiple of 16: a slice may end in the middle of a trip
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = ok ? *reinterpret_cast<const float4*>(p.A + aoff[i] + k0) : NM_Z4;
            if (BT) rb[i] = ok ? *reinterpret_cast<const float4*>(p.Bt + boff[i] + k0) : NM_Z4;
            else rb[i] = (k0 + lr + 8 * i < kend) ? *reinterpret_cast<const float4*>(p.Bt + boff[i] + (long)k0 * p.ldb) : NM_Z4;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lr + 8 * i;
            const int pos = 4 * (lq ^ ((r >> 1) & 7));
            *reinterpret_cast<float4*>(as + r * LDT + pos) = ra[i];
            *reinterpret_cast<float4*>(bs + r * LDT + (BT ? pos : 4 * lq)) = rb[i];       // !BT: row = k, plain
        }
        __builtin_amdgcn_wave_barrier();             // (LDS operations of one wave complete in order)
        if (k0 + 32 < kend) fetch(k0 + 32);          // the next trip's lines travel under this trip's MFMAs
        float4 av[4], bv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {                // lane (m, half): k = k0 + 16 half + 4 c + j
            const int pos = 4 * ((4 * half + c) ^ ((m >> 1) & 7));
            av[c] = *reinterpret_cast<const float4*>(as + m * LDT + pos);
            if (BT) bv[c] = *reinterpret_cast<const float4*>(bs + m * LDT + pos);
            else {                                   // column m of k-rows 16 half + 4 c .. + 3: 32 lanes = 32 banks
                const float* col = bs + (16 * half + 4 * c) * LDT + m;
                bv[c] = make_float4(col[0], col[LDT], col[2 * LDT], col[3 * LDT]);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 4; ++c) {                // MFMA (c, j): k-slot `half` <-> the same k on both operands
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].x, bv[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].y, bv[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].z, bv[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c].w, bv[c].w, acc, 0, 0, 0);
        }
    }
    __syncthreads();                                 // every wave is done with its tiles: they become the reduction buffer
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(lds);
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    // C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); thread (wq, lane)
    // finishes registers 4 wq .. 4 wq + 3 of that lane
    const int wq = ks;                               // the first four K-slice waves of a sub-tile finish it
    const int col = n0 + (lane & 31);
    if (wq >= 4 || col >= N) return;                 // (KS > 4: the extra waves only contributed partial sums)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int reg = 4 * wq + q;
        const int row = m0 + q + 8 * wq + 4 * (lane >> 5);
        if (row >= g.M) continue;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < KS; ++w) s += red[sub * KS + w][reg][lane];
        const float addv = p.add ? p.add[(long)(p.add_ids ? p.add_ids[row] : row) * p.ldadd + col] : 0.0f;
        if (p.epilogue == 0) {
            float v = s + (p.bias ? p.bias[col] : 0.0f) + addv;
            if (p.act == 1) v = nm_tanh(v);
            else if (p.act == 2) v = fmaxf(v, 0.0f);
            p.C[(long)row * p.ldc + col] = v;
        } else if (p.epilogue == 1) {              // N = 2H: r | u = sigmoid(. + bg); rh = r * h
            const int H = N >> 1;
            const float gate = nm_sigmoid(s + p.bias[col] + addv);
            p.ru[(long)row * N + col] = gate;
            if (col < H) p.rh[(long)row * H + col] = gate * p.h[(long)row * p.ldh + col];
        } else {                                   // N = H: c = tanh(xc + .); h' = u*h + (1-u)*c
            const float c = nm_tanh(p.xc[(long)(p.xc_ids ? p.xc_ids[row] : row) * p.ldxc + col] + s);
            const float u = p.ru[(long)row * 2 * N + N + col];
            const float hn = u * p.h[(long)row * p.ldh + col] + (1.0f - u) * c;
            p.h_out[(long)row * p.ldho + col] = hn;
            if (p.h_out2) p.h_out2[(long)row * p.ldho2 + col] = hn;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Medium M through LDS-DMA (round 6).  step_group_medium_kernel above is bound by how fast a CU fills its L1 (a 32x32
// tile per K-split wave set re-reads every operand row M / 32 or N / 32 times: 82 MB for group 1 of a 640-row step)
// and by the registers its loads in flight cost.  Here a 4-wave workgroup owns a 64x64 output tile (41 MB for the same
// group), both operands arrive as 64-k chunks by global_load_lds_dwordx4 -- no staging registers, no ds_write pass,
// three chunks in flight ahead of the one being multiplied -- into four 32 KB LDS images.
// Both operands are stored k-contiguous ([M, K] activations, [N, K] transposed weights), so a DMA piece is 4 rows x 256
// bytes and the LDS image is [row][64 k]; an MFMA fragment (lane = row, 4 consecutive k = one ds_read_b128) would hit
// one bank 16 times over, so the 16-byte unit u of row r is FETCHED into position u ^ (r & 15) (LDS-DMA writes lane-
// linear: the swizzle goes on the source address) and read back from there: 16 rows -> 16 different units.
// v_mfma_f32_32x32x2_f32: lane (m = l & 31, half = l >> 5); the four MFMAs fed by one 16-byte unit U = 2 i + half use
// k = 8 i + 4 half + j on BOTH operands.  Same problems, epilogues and results as step_group_kernel (a_kind 0); the
// products of a row are added in another order than there (chunks of 64 k, two k per MFMA), to rounding the same.
#define SGD_CK 64                          // k per chunk
#define SGD_IMG (64 * SGD_CK)              // floats of one operand image: 64 rows x 64 k

__device__ __forceinline__ void sgd_dma16(const float* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

__global__ __launch_bounds__(256, 1) void step_group_dma_kernel(StepGroup g) {
    extern __shared__ __attribute__((aligned(16))) float sgd_lds[];          // [4 buffers][A image | B image]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = (int)blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < NM_STEP_MAX_PROB; ++i)
        if (i < g.nprob && bid >= g.begin[i]) pi = i;
    const StepProb& p = g.p[pi];
    const int tile = bid - g.begin[pi];
    const int bm = tile % g.tiles_m, bn = tile / g.tiles_m;         // consecutive workgroups share a weight tile
    const int m0 = bm * 64, n0 = bn * 64;
    const int N = (int)p.N, K = (int)p.K;
    const int nchunk = K / SGD_CK;

    // DMA: 32 pieces of 1 KB per chunk (16 of A, 16 of B), eight per wave: wave w fetches rows 16 w .. 16 w + 15 of
    // both images, piece u its rows 4 u .. 4 u + 3; lane L -> row 4 u + (L >> 4), position L & 15 <- unit (L & 15) ^ (row & 15)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)sgd_lds;
    const int prow = lane >> 4, ppos = lane & 15;
    const float* asrc[4];
    const float* bsrc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = 16 * wave + 4 * u + prow;                      // row of the image
        const int unit = ppos ^ (r & 15);
        asrc[u] = p.A + (long)min(m0 + r, g.M - 1) * p.lda + 4 * unit;
        bsrc[u] = p.Bt + (long)min(n0 + r, N - 1) * p.ldb + 4 * unit;
    }
    auto issue = [&](int c) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((c & 3) * 2 * SGD_IMG + 16 * wave * SGD_CK) * 4u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sgd_dma16(asrc[u] + c * SGD_CK, dst + (unsigned)(4 * u * SGD_CK * 4));
            sgd_dma16(bsrc[u] + c * SGD_CK, dst + (unsigned)((SGD_IMG + 4 * u * SGD_CK) * 4));
        }
    };

    const int wr = wave >> 1, wc = wave & 1;                          // this wave's 32x32 quadrant of the tile
    const int m = lane & 31, half = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const int arow = 32 * wr + m, brow = 32 * wc + m;
    // three chunks in flight ahead of the one being multiplied (four LDS images): a piece is ~2 us on its way, a
    // chunk is ~1 us of matrix work -- with one chunk ahead every chunk waited for its pieces (26.7 us per group of a
    // 640-row step against 19 for the register-staged kernel)
    issue(0);
    if (nchunk > 1) issue(1);
    if (nchunk > 2) issue(2);
    for (int c = 0; c < nchunk; ++c) {
        // this wave's pieces of chunk c have landed (the younger chunks stay in flight) ...
        const int younger = min(nchunk - 1 - c, 2);
        if (younger == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... behind the barrier everybody else's too, and everybody is done with chunk c - 1, whose image chunk
        // c + 3 overwrites
        __builtin_amdgcn_s_barrier();
        if (c + 3 < nchunk) issue(c + 3);
        const float* as = sgd_lds + (c & 3) * 2 * SGD_IMG + arow * SGD_CK;
        const float* bs = sgd_lds + (c & 3) * 2 * SGD_IMG + SGD_IMG + brow * SGD_CK;
        float4 av[8], bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int unit = (2 * i + half) ^ (m & 15);
            av[i] = *reinterpret_cast<const float4*>(as + 4 * unit);
            bv[i] = *reinterpret_cast<const float4*>(bs + 4 * unit);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[i].w, acc, 0, 0, 0);
        }
    }
    // C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int col = n0 + 32 * wc + (lane & 31);
    if (col >= N) return;
    const float bias = (p.bias && p.epilogue != 2) ? p.bias[col] : 0.0f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = m0 + 32 * wr + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row >= g.M) continue;
        const float s = acc[reg];
        if (p.epilogue == 0) {
            const float addv = p.add ? p.add[(long)(p.add_ids ? p.add_ids[row] : row) * p.ldadd + col] : 0.0f;
            float v = s + bias + addv;
            if (p.act == 1) v = nm_tanh(v);
            else if (p.act == 2) v = fmaxf(v, 0.0f);
            p.C[(long)row * p.ldc + col] = v;
        } else if (p.epilogue == 1) {              // N = 2H: r | u = sigmoid(. + bg); rh = r * h
            const int H = N >> 1;
            const float addv = p.add ? p.add[(long)(p.add_ids ? p.add_ids[row] : row) * p.ldadd + col] : 0.0f;
            const float gate = nm_sigmoid(s + bias + addv);
            p.ru[(long)row * N + col] = gate;
            if (col < H) p.rh[(long)row * H + col] = gate * p.h[(long)row * p.ldh + col];
        } else {                                   // N = H: c = tanh(xc + .); h' = u*h + (1-u)*c
            const float c = nm_tanh(p.xc[(long)(p.xc_ids ? p.xc_ids[row] : row) * p.ldxc + col] + s);
            const float u = p.ru[(long)row * 2 * N + N + col];
            const float hn = u * p.h[(long)row * p.ldh + col] + (1.0f - u) * c;
            p.h_out[(long)row * p.ldho + col] = hn;
            if (p.h_out2) p.h_out2[(long)row * p.ldho2 + col] = hn;
        }
    }
}

static bool sgd_prepare() {
    static std::atomic<unsigned> devs{0};
    const unsigned ok_bit = 1u << (nm_cur()->device & 15), bad_bit = ok_bit << 16;
    unsigned seen = devs.load(std::memory_order_relaxed);
    if (!(seen & (ok_bit | bad_bit))) {
        const bool ok = hipFuncSetAttribute((const void*)step_group_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            8 * SGD_IMG * 4) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        seen = devs.fetch_or(ok ? ok_bit : bad_bit, std::memory_order_relaxed) | (ok ? ok_bit : bad_bit);
    }
    return (seen & ok_bit) != 0;
}

struct nm_step_problem {          // mirrors include/nmhip.h
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
    const int32_t* add_ids; const int32_t* xc_ids;
};

// OPT-IN (NM_STEP_DMA=1).  Measured per group of a 640-row step, alone and warm (tools/step_group_probe.py,
// profiles/r06_step_group_probe.txt): 16.7 / 16.0 / 18.0 / 25.5 us against the register-staged kernels' 15.9 / 10.5 /
// 19.1 / 16.5, and 26.7 us per group inside a beam step with one chunk in flight (profiles/
// r06_decode_beam_kernel_stats_v1.csv).  These groups are bound by the ~25 GB/s at which a CU fills its L1, whatever
// carries the bytes: halving the bytes per tile (64x64 against 32x32) leaves 160-240 workgroups of 256 KB each for 256
// CUs, and every LDS-DMA piece costs its wave ~150 issue cycles that the register-staged waves spend on MFMAs.
static bool sgd_switch() {            // (read per call: a test flips it within one process; a getenv is nanoseconds)
    const char* e = getenv("NM_STEP_DMA");
    return e && e[0] == '1';
}

extern "C" int nm_step_group(void* stream, int64_t M, const nm_step_problem* probs, int32_t nprob) {
    NM_REQUIRE(probs && nprob >= 1 && nprob <= NM_STEP_MAX_PROB, "nm_step_group: 1..%d problems", NM_STEP_MAX_PROB);
    NM_REQUIRE(M > 0 && M < (1 << 24), "nm_step_group: bad M %ld", (long)M);
    StepGroup g;
    g.nprob = nprob;
    g.M = (int)M;
    // tile height: 16 rows per workgroup while the whole group stays within 512 workgroups (two per CU), else 32
    long tiles16 = 0;
    for (int i = 0; i < nprob; ++i) tiles16 += (long)nm_cdiv(M, 16) * nm_cdiv(probs[i].N, 16);
    // medium M (beam search): 32x32 tiles, 4 waves split K (step_group_medium_kernel); NM_STEP_MEDIUM=0 keeps the
    // 16-row tiles
    const int medium_sw = nm_cur()->sw.medium_m;        // 0 off; 1 = sub-tiles per workgroup by the rule below; 2 = 32x32 only
    const bool medium = probs[0].a_kind == 0 && M > 256 && medium_sw != 0;
    // Sub-tiles per workgroup.  These products are bound by how fast a CU can FILL its L1 from L2 (~10 B / clk / CU:
    // outstanding misses x line size / latency), not by the matrix pipes: a 32x32 tile per wave set is 8 flop per
    // operand byte = 80 flop / clk / CU = 31 % of the fp32 MFMA rate -- exactly what the PMC pass of the 32x32
    // version shows.  Sub-tiles of one workgroup that share operand rows hit each other's lines in L1, so 64x64 per
    // workgroup halves the fill traffic (16 flop / byte): group 1 of the 640-row step 39.4 -> 32.0 us.  Fat
    // workgroups only while >= 256 of them remain (320 at 640 rows: some CUs then run two in a row, which is what
    // is left of the 32 us); 64x32 measured no better than 32x32 (17.9 / 38.3 us against 17.4 / 39.4).
    // LDS-DMA tiles (step_group_dma_kernel, opt-in: NM_STEP_DMA=1) when every product of the group has whole 64-k chunks
    bool dma = medium && medium_sw == 1 && sgd_switch();
    for (int i = 0; i < nprob && dma; ++i)
        dma = probs[i].K % SGD_CK == 0 && probs[i].K >= SGD_CK && probs[i].a_kind == 0 && probs[i].lda % 4 == 0 &&
              probs[i].ldb % 4 == 0;
    if (dma) dma = sgd_prepare();
    int tr = 1, tc = 1;
    if (dma) { tr = 2; tc = 2; }
    else if (medium && medium_sw == 1) {
        long t44 = 0;
        for (int i = 0; i < nprob; ++i) t44 += (long)nm_cdiv(M, 64) * nm_cdiv(probs[i].N, 64);
        if (t44 >= 256) { tr = 2; tc = 2; }
    }
    const int tm = medium ? 2 * tr : ((probs[0].a_kind == 0 && tiles16 > 512) ? 2 : 1);
    const int tile_n = medium ? 32 * tc : 16;
    g.tiles_m = nm_cdiv(M, 16 * tm);
    g.wblocks = (probs[0].a_kind == 1 && probs[0].weights) ? nm_cdiv(M, 16) : 0;
    int next = 0;
    for (int i = 0; i < nprob; ++i) {
        const nm_step_problem& q = probs[i];
        StepProb& p = g.p[i];
        NM_REQUIRE(q.Bt && q.N > 0 && q.K > 0 && q.K % 16 == 0 && q.N < (1 << 24) && q.K < (1 << 24),
                   "nm_step_group[%d]: bad shape N=%ld K=%ld (K must be a multiple of 16)", i, (long)q.N, (long)q.K);
        NM_REQUIRE(nm_aligned16(q.Bt) && q.ldb % 4 == 0 && q.ldb >= q.K, "nm_step_group[%d]: weights must be [N,K], "
                   "16-byte aligned rows", i);
        NM_REQUIRE(q.a_kind == 0 || q.a_kind == 1, "nm_step_group[%d]: bad a_kind", i);
        if (q.a_kind == 0)
            NM_REQUIRE(q.A && nm_aligned16(q.A) && q.lda % 4 == 0 && q.lda >= q.K, "nm_step_group[%d]: A must be "
                       "[M,K] with 16-byte aligned rows", i);
        else
            NM_REQUIRE(q.pctx && q.pstat && nm_aligned16(q.pctx) && nm_aligned16(q.pstat) && q.nchunk >= 1 &&
                       q.nchunk <= NM_STEP_MAX_CHUNK && (!q.weights || (q.energies && q.S > 0 && q.mask_div >= 1 &&
                                                                         q.mask_mod >= 1)),
                       "nm_step_group[%d]: bad attention partials (1..%d chunks)", i, NM_STEP_MAX_CHUNK);
        NM_REQUIRE(q.epilogue >= 0 && q.epilogue <= 2, "nm_step_group[%d]: bad epilogue", i);
        if (q.epilogue == 0) NM_REQUIRE(q.C && q.ldc >= q.N && (!q.add || q.ldadd >= q.N) && (q.act == 0 || q.act == 1),
                                        "nm_step_group[%d]: bad plain epilogue", i);
        if (q.epilogue == 1) NM_REQUIRE(q.bias && q.ru && q.rh && q.h && q.N % 2 == 0 && q.ldh >= q.N / 2 &&
                                            (!q.add || q.ldadd >= q.N), "nm_step_group[%d]: bad gates epilogue", i);
        NM_REQUIRE((!q.add_ids || q.add) && (!q.xc_ids || q.epilogue == 2), "nm_step_group[%d]: row ids without the "
                   "operand they index", i);
        if (q.epilogue == 2) NM_REQUIRE(q.xc && q.ru && q.h && q.h_out && q.ldxc >= q.N && q.ldh >= q.N &&
                                        q.ldho >= q.N && (!q.h_out2 || q.ldho2 >= q.N),
                                        "nm_step_group[%d]: bad candidate epilogue", i);
        p.A = q.A; p.lda = q.lda; p.Bt = q.Bt; p.ldb = q.ldb; p.N = q.N; p.K = q.K;
        p.a_kind = q.a_kind; p.epilogue = q.epilogue; p.act = q.act; p.nchunk = q.nchunk;
        p.bias = q.bias; p.add = q.add; p.ldadd = q.ldadd; p.C = q.C; p.ldc = q.ldc;
        p.pctx = q.pctx; p.pstat = q.pstat; p.energies = q.energies; p.mask = q.mask; p.weights = q.weights;
        p.S = q.S; p.mask_div = q.mask_div; p.mask_mod = q.mask_mod;
        p.h = q.h; p.ldh = q.ldh; p.ru = q.ru; p.rh = q.rh; p.xc = q.xc; p.ldxc = q.ldxc;
        p.h_out = q.h_out; p.ldho = q.ldho; p.h_out2 = q.h_out2; p.ldho2 = q.ldho2;
        p.add_ids = q.add_ids; p.xc_ids = q.xc_ids;
        g.begin[i] = next;
        next += g.tiles_m * nm_cdiv(q.N, tile_n);
    }
    for (int i = nprob; i <= NM_STEP_MAX_PROB; ++i) g.begin[i] = next;
    for (int i = nprob; i < NM_STEP_MAX_PROB; ++i) g.p[i] = g.p[0];
    for (int i = 1; i < nprob; ++i)
        NM_REQUIRE(probs[i].a_kind == probs[0].a_kind, "nm_step_group: the problems of a group share one operand loader");
    hipStream_t st = nm_stream(stream);
    const unsigned grid = (unsigned)(next + g.wblocks);
    // few tiles (a single 512-wide product at 640 rows: 320 tiles on 256 CUs): 8 waves split K, so that the one or
    // two workgroups a CU gets are half as long
    if (dma)
        hipLaunchKernelGGL(step_group_dma_kernel, dim3(grid), dim3(256), 8 * SGD_IMG * 4, st, g);
    else if (medium && tr == 2 && tc == 2)
        hipLaunchKernelGGL((step_group_medium_kernel<4, 2, 2>), dim3(grid), dim3(1024), 0, st, g);
    else if (medium && grid <= 640)
        hipLaunchKernelGGL((step_group_medium_kernel<8, 1, 1>), dim3(grid), dim3(512), 0, st, g);
    else if (medium) hipLaunchKernelGGL((step_group_medium_kernel<4, 1, 1>), dim3(grid), dim3(256), 0, st, g);
    else if (probs[0].a_kind == 1) hipLaunchKernelGGL((step_group_kernel<16, 1, 1>), dim3(grid), dim3(1024), 0, st, g);
    else if (tm == 2) hipLaunchKernelGGL((step_group_kernel<16, 2, 0>), dim3(grid), dim3(1024), 0, st, g);
    else hipLaunchKernelGGL((step_group_kernel<16, 1, 0>), dim3(grid), dim3(1024), 0, st, g);
    NM_LAUNCH_CHECK("nm_step_group");
}

// nm_gemm_f32's medium-M route: C[M,N] = act(A[M,K] . B + bias (+ C)) for a few hundred rows, where 128x128 / 64x64
// tiles leave most CUs without a workgroup (a 640 x 512 x 512 product of a Transformer beam step: 80 tiles of 64x64,
// 26.8 us = 12.5 TFLOP/s) -- the 32x32 K-split tiles of the decoder-step groups fill the chip (~11 us).  Returns false
// when the shape is not taken.
bool nm_medium_gemm(hipStream_t st, int transB, long M, long N, long K, const float* A, long lda, const float* B,
                    long ldb, float* C, long ldc, const float* bias, int act, int accumulate) {
    if (M <= 256 || M > 2048 || nm_cur()->sw.medium_m == 0) return false;
    // (a 32x32 tile walks ALL of K: made for K ~ 512..2048.  The input gradient of the vocabulary projection at a few
    // hundred rows -- K = 32000 -- took 401 us here against ~250 us on the split-K tiled kernels)
    if (K > 4096) return false;
    if (K % 16 || N % 32 || lda % 4 || ldb % 4 || !nm_aligned16(A) || !nm_aligned16(B)) return false;
    if (act < 0 || act > 2 || M * lda >= (1L << 31) || (transB ? N : K) * ldb >= (1L << 31)) return false;
    const long t64 = (long)nm_cdiv(M, 64) * nm_cdiv(N, 64);
    if (t64 >= 256) return false;                                    // the tiled kernels fill the chip themselves
    StepGroup g;
    memset(&g, 0, sizeof(g));
    g.nprob = 1; g.M = (int)M; g.wblocks = 0;
    StepProb& p = g.p[0];
    p.A = A; p.lda = lda; p.Bt = B; p.ldb = ldb; p.N = N; p.K = K; p.epilogue = 0; p.act = act;
    p.bias = bias; p.add = accumulate ? C : nullptr; p.ldadd = ldc; p.C = C; p.ldc = ldc;
    for (int i = 1; i < NM_STEP_MAX_PROB; ++i) g.p[i] = g.p[0];
    g.tiles_m = nm_cdiv(M, 32);
    const long tiles = (long)g.tiles_m * nm_cdiv(N, 32);
    g.begin[0] = 0;
    for (int i = 1; i <= NM_STEP_MAX_PROB; ++i) g.begin[i] = (int)tiles;
    const unsigned grid = (unsigned)tiles;
    if (transB) {
        if (tiles <= 640) hipLaunchKernelGGL((step_group_medium_kernel<8, 1, 1, true>), dim3(grid), dim3(512), 0, st, g);
        else hipLaunchKernelGGL((step_group_medium_kernel<4, 1, 1, true>), dim3(grid), dim3(256), 0, st, g);
    } else {
        if (tiles <= 640) hipLaunchKernelGGL((step_group_medium_kernel<8, 1, 1, false>), dim3(grid), dim3(512), 0, st, g);
        else hipLaunchKernelGGL((step_group_medium_kernel<4, 1, 1, false>), dim3(grid), dim3(256), 0, st, g);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// The whole step behind ONE call: Decoder.next_state (decoders/decoder.py:279-358, plain GRUCell, one Bahdanau
// attention, nonlinear output projection) followed by the vocabulary projection of get_body
// (decoders/autoregressive.py:450-459).  Host sequencing only -- the seven launches are the ones documented at
// the top of this file; a caller that replays the step (HIP graph) captures this call.
extern "C" int nm_attn_fwd(void* stream, const float* y, const float* hf, const float* states, const float* mask,
                           const float* v, const float* bias, int64_t R, int64_t rows_per_key, int64_t S, int64_t A,
                           int64_t C, float* ctx, int64_t ldctx, float* weights, void* workspace,
                           int64_t workspace_bytes, float* energies_out);
extern "C" int nm_logits_stats_gemm(void* stream, int transB, int64_t M, int64_t N, int64_t K, const float* A,
                                    int64_t lda, const float* B, int64_t ldb, const float* bias, float* C, int64_t ldc,
                                    float* stats, int64_t stats_bytes);
extern "C" int nm_gemm_f32(void* stream, int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A,
                           int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act,
                           int accumulate, int64_t batch, int64_t strideA, int64_t strideB, int64_t strideC, int algo,
                           void* workspace, int64_t workspace_bytes);

struct nm_decoder_step {          // mirrors include/nmhip.h
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
    int64_t ld_cat, ld_ctx, ld_wg, ld_wcx, ld_wch, ld_wq, ld_wo_h, ld_wo_e, ld_wo_c;
    const float* in_table; int64_t ld_table; const int32_t* in_ids;
    void* cluster_ws; int64_t cluster_ws_bytes; uint32_t* sticky_error;
};

// nm_gru_cluster.hip: groups 1-3 of a step with input tables as ONE cluster launch; false when the shape is not taken
bool nm_dec_step_cluster_try(hipStream_t st, int64_t R, int64_t H, int64_t A, int64_t O, const float* h_in, int64_t ld_h,
                             const float* table, int64_t ld_table, const int32_t* ids, const float* bg, const float* bq,
                             const float* bo, const float* wg_t, int64_t ld_wg, const float* wc_t, int64_t ld_wc,
                             const float* wq_t, int64_t ld_wq, const float* wo_t, int64_t ld_wo, float* h_out,
                             int64_t ld_ho, float* h_out2, int64_t ld_ho2, float* y, int64_t ld_y, float* pre,
                             int64_t ld_pre, void* workspace, int64_t workspace_bytes, uint32_t* sticky);

extern "C" int nm_decoder_step_fused(void* stream, const nm_decoder_step* d) {
    NM_REQUIRE(d, "nm_decoder_step_fused: null descriptor");
    const int64_t M = d->rows, E = d->emb, H = d->rnn, A = d->attn_state, C = d->ctx_width, O = d->out;
    NM_REQUIRE(M > 0 && E > 0 && H > 0 && A > 0 && C > 0 && O > 0 && d->vocab > 0 && d->src_len > 0 &&
               d->rows_per_key >= 1 && M % d->rows_per_key == 0, "nm_decoder_step_fused: bad sizes");
    NM_REQUIRE(E % 16 == 0 && H % 16 == 0 && C % 16 == 0, "nm_decoder_step_fused: emb, rnn and context widths must be "
               "multiples of 16 (the K of the step GEMMs)");
    NM_REQUIRE(d->cat && d->out_state && d->ru && d->rh && d->xc && d->y && d->pre_e && d->pre && d->ctx &&
               d->attn_workspace, "nm_decoder_step_fused: null step buffer");
    NM_REQUIRE(d->wg_t && d->bg && d->wcx_t && d->wch_t && d->bc && d->wq_t && d->keys && d->values && d->v &&
               d->wo_h_t && d->wo_e_t && d->wo_c_t && d->w_vocab, "nm_decoder_step_fused: null parameter");
    NM_REQUIRE(d->logits || d->stats, "nm_decoder_step_fused: neither logits nor tile statistics requested");
    const int64_t ld = d->ld_cat ? d->ld_cat : E + H;
    const int64_t ld_ctx = d->ld_ctx ? d->ld_ctx : C;
    const int64_t ld_wg = d->ld_wg ? d->ld_wg : E + H, ld_wcx = d->ld_wcx ? d->ld_wcx : E,
                  ld_wch = d->ld_wch ? d->ld_wch : H, ld_wq = d->ld_wq ? d->ld_wq : H,
                  ld_wo_h = d->ld_wo_h ? d->ld_wo_h : H, ld_wo_e = d->ld_wo_e ? d->ld_wo_e : E,
                  ld_wo_c = d->ld_wo_c ? d->ld_wo_c : C;
    NM_REQUIRE(ld >= E + H && ld_ctx >= C && ld % 4 == 0 && ld_ctx % 4 == 0, "nm_decoder_step_fused: bad leading "
               "dimensions of the input row / context buffer");
    float* h = d->cat + E;                       // the state half of the input row
    nm_step_problem p[3];
    int rc;
    const bool tables = d->in_table != nullptr;
    NM_REQUIRE(!tables || (d->in_ids && d->ld_table >= 3 * H + O), "nm_decoder_step_fused: input tables need the rows' "
               "symbols and [V, 2*rnn + rnn + out] columns");
    // greedy-sized steps with input tables: gates, candidate + blend, query and the state part of the output projection
    // in ONE launch of workgroup clusters (dec_step_cluster_kernel) instead of the three dependent groups below
    const bool clustered = tables && d->cluster_ws &&
        nm_dec_step_cluster_try(nm_stream(stream), M, H, A, O, h, ld, d->in_table, d->ld_table, d->in_ids, d->bg, d->bq,
                                d->bo, d->wg_t + E, ld_wg, d->wch_t, ld_wch, d->wq_t, ld_wq, d->wo_h_t, ld_wo_h, h, ld,
                                d->h_copy, d->h_copy ? d->ld_h_copy : 0, d->y, A, d->pre, O, d->cluster_ws,
                                d->cluster_ws_bytes, d->sticky_error);
    if (clustered) {
    } else if (tables) {
        // group 1 with input tables: only the state half of the gates product is left -- h . Wg_h (the state rows of
        // the gates kernel = columns emb.. of wg_t) + in_table[id, :2H] + bg
        memset(p, 0, sizeof(p));
        p[0].A = h; p[0].lda = ld; p[0].Bt = d->wg_t + E; p[0].ldb = ld_wg; p[0].N = 2 * H; p[0].K = H; p[0].epilogue = 1;
        p[0].bias = d->bg; p[0].h = h; p[0].ldh = ld; p[0].ru = d->ru; p[0].rh = d->rh;
        p[0].add = d->in_table; p[0].ldadd = d->ld_table; p[0].add_ids = d->in_ids;
        if ((rc = nm_step_group(stream, M, p, 1)) != 0) return rc;
    } else {
    // group 1: gates over [emb | h]; the two products that only need the embedded input
    memset(p, 0, sizeof(p));
    p[0].A = d->cat; p[0].lda = ld; p[0].Bt = d->wg_t; p[0].ldb = ld_wg; p[0].N = 2 * H; p[0].K = E + H; p[0].epilogue = 1;
    p[0].bias = d->bg; p[0].h = h; p[0].ldh = ld; p[0].ru = d->ru; p[0].rh = d->rh;
    p[1].A = d->cat; p[1].lda = ld; p[1].Bt = d->wcx_t; p[1].ldb = ld_wcx; p[1].N = H; p[1].K = E; p[1].bias = d->bc;
    p[1].C = d->xc; p[1].ldc = H;
    p[2].A = d->cat; p[2].lda = ld; p[2].Bt = d->wo_e_t; p[2].ldb = ld_wo_e; p[2].N = O; p[2].K = E;
    p[2].C = d->pre_e; p[2].ldc = O;
    if ((rc = nm_step_group(stream, M, p, 3)) != 0) return rc;
    }
    if (!clustered) {
    // group 2: candidate + blend, h' in place (and into the caller's history row)
    memset(p, 0, sizeof(p));
    p[0].A = d->rh; p[0].lda = H; p[0].Bt = d->wch_t; p[0].ldb = ld_wch; p[0].N = H; p[0].K = H; p[0].epilogue = 2;
    p[0].xc = d->xc; p[0].ldxc = H; p[0].ru = d->ru; p[0].h = h; p[0].ldh = ld; p[0].h_out = h; p[0].ldho = ld;
    if (tables) { p[0].xc = d->in_table + 2 * H; p[0].ldxc = d->ld_table; p[0].xc_ids = d->in_ids; }
    p[0].h_out2 = d->h_copy; p[0].ldho2 = d->h_copy ? d->ld_h_copy : 0;
    if ((rc = nm_step_group(stream, M, p, 1)) != 0) return rc;
    // group 3: attention query; the state part of the output projection
    memset(p, 0, sizeof(p));
    p[0].A = h; p[0].lda = ld; p[0].Bt = d->wq_t; p[0].ldb = ld_wq; p[0].N = A; p[0].K = H; p[0].bias = d->bq;
    p[0].C = d->y; p[0].ldc = A;
    p[1].A = h; p[1].lda = ld; p[1].Bt = d->wo_h_t; p[1].ldb = ld_wo_h; p[1].N = O; p[1].K = H; p[1].bias = d->bo;
    p[1].add = d->pre_e; p[1].ldadd = O; p[1].C = d->pre; p[1].ldc = O;
    if (tables) { p[1].add = d->in_table + 3 * H; p[1].ldadd = d->ld_table; p[1].add_ids = d->in_ids; }
    if ((rc = nm_step_group(stream, M, p, 2)) != 0) return rc;
    }
    // attention: one launch
    if ((rc = nm_attn_fwd(stream, d->y, d->keys, d->values, d->mask, d->v, d->attn_bias, M, d->rows_per_key,
                          d->src_len, A, C, d->ctx, ld_ctx, d->attn_weights, d->attn_workspace, d->attn_workspace_bytes,
                          nullptr)) != 0) return rc;
    // group 4: the context part of the output projection + activation
    memset(p, 0, sizeof(p));
    p[0].A = d->ctx; p[0].lda = ld_ctx; p[0].Bt = d->wo_c_t; p[0].ldb = ld_wo_c; p[0].N = O; p[0].K = C; p[0].act = d->out_act;
    p[0].add = d->pre; p[0].ldadd = O; p[0].C = d->out_state; p[0].ldc = d->ld_out_state;
    if ((rc = nm_step_group(stream, M, p, 1)) != 0) return rc;
    // vocabulary projection
    if (d->stats)
        return nm_logits_stats_gemm(stream, d->vocab_trans_b, M, d->vocab, O, d->out_state, d->ld_out_state, d->w_vocab,
                                    d->ld_w_vocab, d->b_vocab, d->logits, d->logits ? d->ld_logits : 0, d->stats,
                                    d->stats_bytes);
    return nm_gemm_f32(stream, 0, d->vocab_trans_b, M, d->vocab, O, d->out_state, d->ld_out_state, d->w_vocab,
                       d->ld_w_vocab, d->logits, d->ld_logits, d->b_vocab, 0, 0, 1, 0, 0, 0, 0, nullptr, 0);
}
