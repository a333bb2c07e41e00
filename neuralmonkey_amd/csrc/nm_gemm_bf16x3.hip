// COSTING ONLY -- not on any product path.  C = A . B^T with fp32 operands emulated by THREE bf16 matrix-core products
// (split-bf16: x = hi + lo, hi = bf16(x), lo = bf16(x - hi); A.B ~ hi.hi + hi.lo + lo.hi, fp32 accumulate).
//
// Why it is worth costing: gfx950 runs exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) at 1/16 of
// the bf16 rate, and the three logits-sized products are 43 % of the headline training step, the vocabulary
// projection 55 % of a beam step.  Three bf16 instructions per 16 k (3 x 32 cycles) replace eight fp32 ones
// (8 x 64): 5.3x less matrix time -- if the operand feed keeps up.  The dropped lo.lo term is 2^-16 of a term,
// each retained term is exact in fp32, so a dot product carries ~2^-17 relative error per product before the
// usual accumulation error: ~20x the exact-fp32 kernel's, 50x inside north_star's 1e-4.
//
// Shape served: the "NT" form with both operands k-contiguous -- C[M,N] = A[M,K] . B[N,K]^T -- which is
//   * dlogits . W^T   (M=6400, N=512,   K=32000)  the input gradient of the vocabulary projection, and
//   * states . E^T    (M=6400, N=32000, K=512)    the projection itself with tied embeddings (Transformer).
// 128x128 tile, 8 waves, BK = 16 = one v_mfma_f32_32x32x16_bf16 per operand pair; fp32 tiles are split when they
// are staged: LDS holds [row][k] bf16 hi and lo planes (32-byte rows: a wave's fragment read is one linear 2 KB
// ds_read_b128 sweep).  The kernel exists to put a NUMBER on the emulation (tools/gemm_bf16x3_cost.py,
// bench.py `configs.logits_gemm_bf16x3`); the shipped path stays exact fp32.
#include "nm_common.h"

#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Bf3Args {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    int terms;      // 3: hi.hi + hi.lo + lo.hi (fp32-class), 1: hi.hi only (plain bf16, for reference)
};

// round-to-nearest-even bf16 of x, as the upper 16 bits of a float
__device__ __forceinline__ unsigned bf3_hi_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}

// two floats -> (hi, hi) and (lo, lo) packed pairs of bf16
__device__ __forceinline__ void bf3_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const unsigned h0 = bf3_hi_bits(x0), h1 = bf3_hi_bits(x1);
    const unsigned l0 = bf3_hi_bits(x0 - __uint_as_float(h0)), l1 = bf3_hi_bits(x1 - __uint_as_float(h1));
    hi = (h0 >> 16) | h1;
    lo = (l0 >> 16) | l1;
}

template <int WM, int WN, int TM, int TN, int BK>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16x3_nt(Bf3Args g, int tiles_m) {
    constexpr int NT = WM * WN * 64, KQ = BK / 4;
    constexpr int BM = WM * 32 * TM, BN = WN * 32 * TN;
    constexpr int LA = BM * BK / 4 / NT, LB = BN * BK / 4 / NT;       // float4 loads per thread
    static_assert(LA >= 1 && LB >= 1 && LA * NT * 4 == BM * BK && LB * NT * 4 == BN * BK, "tile / thread mismatch");
    // planes [buf][hi|lo][row][BK] of bf16: BK * 2 = 32 bytes per row
    __shared__ __attribute__((aligned(16))) unsigned short As[2][2][BM][BK];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][2][BN][BK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm = (int)blockIdx.x % tiles_m, bn = (int)blockIdx.x / tiles_m;
    const int m0 = bm * BM, n0 = bn * BN;
    const float* __restrict__ A = g.A;
    const float* __restrict__ B = g.B;

    float4 ra[LA], rb[LB];
    auto load = [&](int k0) {
#pragma unroll
        for (int it = 0; it < LA; ++it) {
            const int idx = tid + it * NT, m = idx / KQ, k = (idx % KQ) * 4;
            const int gm = m0 + m, gk = k0 + k;
            ra[it] = (gm < g.M && gk < g.K) ? *reinterpret_cast<const float4*>(A + (long)gm * g.lda + gk)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < LB; ++it) {
            const int idx = tid + it * NT, n = idx / KQ, k = (idx % KQ) * 4;
            const int gn = n0 + n, gk = k0 + k;
            rb[it] = (gn < g.N && gk < g.K) ? *reinterpret_cast<const float4*>(B + (long)gn * g.ldb + gk)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int it = 0; it < LA; ++it) {
            const int idx = tid + it * NT, m = idx / KQ, k = (idx % KQ) * 4;
            uint2 hi, lo;
            bf3_split2(ra[it].x, ra[it].y, hi.x, lo.x);
            bf3_split2(ra[it].z, ra[it].w, hi.y, lo.y);
            *reinterpret_cast<uint2*>(&As[buf][0][m][k]) = hi;
            *reinterpret_cast<uint2*>(&As[buf][1][m][k]) = lo;
        }
#pragma unroll
        for (int it = 0; it < LB; ++it) {
            const int idx = tid + it * NT, n = idx / KQ, k = (idx % KQ) * 4;
            uint2 hi, lo;
            bf3_split2(rb[it].x, rb[it].y, hi.x, lo.x);
            bf3_split2(rb[it].z, rb[it].w, hi.y, lo.y);
            *reinterpret_cast<uint2*>(&Bs[buf][0][n][k]) = hi;
            *reinterpret_cast<uint2*>(&Bs[buf][1][n][k]) = lo;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int wm = (wave / WN) * 32 * TM, wn = (wave % WN) * 32 * TN;
    const int row = lane & 31, kh = (lane >> 5) * 8;      // fragment: 8 consecutive k of one row / column
    const int nkt = (g.K + BK - 1) / BK;

    load(0);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(&As[cur][0][wm + i * 32 + row][ks + kh]);
                al[i] = *reinterpret_cast<const bf16x8*>(&As[cur][1][wm + i * 32 + row][ks + kh]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(&Bs[cur][0][wn + j * 32 + row][ks + kh]);
                bl[j] = *reinterpret_cast<const bf16x8*>(&Bs[cur][1][wn + j * 32 + row][ks + kh]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (g.terms == 3) {     // the two cross terms first: small contributions before the large one
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nkt) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // C/D layout of the 32x32 MFMAs: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn + j * 32 + (lane & 31);
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (rr < g.M) g.C[(long)rr * g.ldc + col] = acc[i][j][r];
            }
        }
}

// C[M,N] = A[M,K] . B[N,K]^T, fp32 in / fp32 out, bf16 matrix cores inside.  terms: 3 (split-bf16, fp32-class
// accuracy) or 1 (plain bf16 operands).  K % 4 == 0, rows 16-byte aligned.  COSTING ONLY (see the file header).
extern "C" int nm_gemm_bf16x3_nt(void* stream, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                 const float* B, int64_t ldb, float* C, int64_t ldc, int terms, int variant) {
    NM_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "nm_gemm_bf16x3_nt: bad arguments");
    NM_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && nm_aligned16(A) && nm_aligned16(B),
               "nm_gemm_bf16x3_nt: K and the row strides must be multiples of 4, operands 16-byte aligned");
    NM_REQUIRE(terms == 1 || terms == 3, "nm_gemm_bf16x3_nt: terms must be 1 or 3");
    NM_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "nm_gemm_bf16x3_nt: dim too large");
    Bf3Args g{A, B, C, (int)M, (int)N, (int)K, (long)lda, (long)ldb, (long)ldc, terms};
    // variant: 0 = 128x128 tile, BK 16 (the exact-fp32 kernel's tiling); 1 = 128x128, BK 32; 2 = 256x128, BK 16;
    // 3 = 256x128, BK 32 -- more work per staged byte once the matrix time is 5x shorter
    hipStream_t st = nm_stream(stream);
    if (variant == 1) {
        const int tm = nm_cdiv(M, 128), tn = nm_cdiv(N, 128);
        hipLaunchKernelGGL((gemm_bf16x3_nt<4, 2, 1, 2, 32>), dim3(tm * tn), dim3(512), 0, st, g, tm);
    } else if (variant == 2) {
        const int tm = nm_cdiv(M, 256), tn = nm_cdiv(N, 128);
        hipLaunchKernelGGL((gemm_bf16x3_nt<4, 2, 2, 2, 16>), dim3(tm * tn), dim3(512), 0, st, g, tm);
    } else if (variant == 3) {
        const int tm = nm_cdiv(M, 256), tn = nm_cdiv(N, 128);
        hipLaunchKernelGGL((gemm_bf16x3_nt<4, 2, 2, 2, 32>), dim3(tm * tn), dim3(512), 0, st, g, tm);
    } else {
        const int tm = nm_cdiv(M, 128), tn = nm_cdiv(N, 128);
        hipLaunchKernelGGL((gemm_bf16x3_nt<4, 2, 1, 2, 16>), dim3(tm * tn), dim3(512), 0, st, g, tm);
    }
    NM_LAUNCH_CHECK("nm_gemm_bf16x3_nt");
}

// =====================================================================================================================
// OPT-IN product path (NM_PROJ_SPLIT=1, its own dtype in the bench line): the vocabulary projection of the DECODING
// steps -- tf.matmul(state, decoding_w) + bias, decoders/autoregressive.py:450-459, with the row statistics of
// nm_logits_stats_gemm -- on the bf16 matrix cores with fp32-class accuracy.  Every fp32 operand is split THREE ways,
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 24 mantissa bits, all of fp32's;
// six products are kept (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi: everything down to 2^-24 of a term), summed in
// fp32 from the smallest to the largest.  6 x 32 cycles of v_mfma_f32_32x32x16_bf16 per 16 k against 8 x 64 of
// v_mfma_f32_32x32x2_f32.  The weights do not change while a model decodes: they are split ONCE
// (nm_proj_split_prepare) into planes laid out tile by tile -- [column tile of 128][k tile of 16][hi|mid|lo][128][16]
// bf16 -- so that what a workgroup stages per k step is 12 KB of consecutive bytes; the activations are split while
// they are staged.  The exact-fp32 kernel stays the default and the number of record.
// =====================================================================================================================
__device__ __forceinline__ void split3_2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned h0 = bf3_hi_bits(x0), h1 = bf3_hi_bits(x1);
    const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
    const unsigned m0 = bf3_hi_bits(r0), m1 = bf3_hi_bits(r1);
    const unsigned l0 = bf3_hi_bits(r0 - __uint_as_float(m0)), l1 = bf3_hi_bits(r1 - __uint_as_float(m1));
    hi = (h0 >> 16) | h1;
    mid = (m0 >> 16) | m1;
    lo = (l0 >> 16) | l1;
}

// planes[((bn * KT + kt) * 3 + plane) * 2048 + n_local * 16 + k_local] of W: [K][N] (trans_b 0) or [N][K] (trans_b 1)
__global__ __launch_bounds__(256) void split3_planes_kernel(const float* __restrict__ W, long ldw, int trans_b, int N, int K,
                                                           unsigned short* __restrict__ planes) {
    const int KT = K / 16;
    const int bn = blockIdx.x / KT, kt = blockIdx.x % KT;
    unsigned short* dst = planes + (long)(bn * KT + kt) * 3 * 2048;
    for (int e = threadIdx.x; e < 2048; e += 256) {
        const int nl = trans_b ? e / 16 : e % 128, kl = trans_b ? e % 16 : e / 128;      // read along the contiguous axis
        const int n = bn * 128 + nl, k = kt * 16 + kl;
        float x = 0.0f;
        if (n < N) x = trans_b ? W[(long)n * ldw + k] : W[(long)k * ldw + n];
        const unsigned h = bf3_hi_bits(x);
        const float r = x - __uint_as_float(h);
        const unsigned m = bf3_hi_bits(r);
        const unsigned l = bf3_hi_bits(r - __uint_as_float(m));
        dst[nl * 16 + kl] = (unsigned short)(h >> 16);
        dst[2048 + nl * 16 + kl] = (unsigned short)(m >> 16);
        dst[4096 + nl * 16 + kl] = (unsigned short)(l >> 16);
    }
}

struct Split6Args {
    const float* A; long lda;                // [M][K] fp32
    const unsigned short* planes;            // split weights, tiled (see above)
    const float* bias;                       // [N] or null
    float* C; long ldc;                      // [M][N] or null: the logits are stored only when somebody reads them
    float* stats;                            // [M][tiles_n][4] = {max, sum exp(x - max), first argmax, -}
    int M, N, K;
};

__global__ __launch_bounds__(512) void gemm_split6_stats(Split6Args g, int tiles_m) {
    constexpr int BM = 128, BN = 128, BK = 16, NT = 512, TN = 2;
    constexpr int HR = 64, TS = BN + 1;                       // statistics epilogue: rows per pass, odd row stride
    // operand tiles [buf][plane][row][16] bf16, re-used by the epilogue as [HR][129] floats + 3 x 512 scratch
    __shared__ __attribute__((aligned(16))) unsigned short lds_raw[2 * 2 * 3 * 128 * 16];      // 48 KB
    unsigned short (*As)[3][BM][BK] = reinterpret_cast<unsigned short (*)[3][BM][BK]>(lds_raw);
    unsigned short (*Bs)[3][BN][BK] = reinterpret_cast<unsigned short (*)[3][BN][BK]>(lds_raw + 2 * 3 * 128 * 16);
    static_assert(HR * TS + 3 * NT <= 2 * 2 * 3 * 128 * 16 / 2, "the epilogue fits the operand buffers");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm = (int)blockIdx.x % tiles_m, bn = (int)blockIdx.x / tiles_m;
    const int m0 = bm * BM, n0 = bn * BN;
    const int KT = g.K / BK;
    const float* __restrict__ A = g.A;
    const uint4* __restrict__ Bt = reinterpret_cast<const uint4*>(g.planes + (long)bn * KT * 3 * 2048);   // 384 uint4 per k tile... x2

    // staging: A one float4 per thread (128 rows x 4 quads); B 768 16-byte chunks (3 planes x 128 rows x 2 halves)
    const int am = tid >> 2, ak = (tid & 3) * 4;
    float4 ra;
    uint4 rb0, rb1;
    auto load = [&](int kt) {
        const int gm = m0 + am;
        ra = gm < g.M ? *reinterpret_cast<const float4*>(A + (long)gm * g.lda + kt * BK + ak) : make_float4(0.f, 0.f, 0.f, 0.f);
        const uint4* src = Bt + (long)kt * 768;
        rb0 = src[tid];
        if (tid < 256) rb1 = src[512 + tid];
    };
    auto stage = [&](int buf) {
        uint2 hi, mid, lo;
        split3_2(ra.x, ra.y, hi.x, mid.x, lo.x);
        split3_2(ra.z, ra.w, hi.y, mid.y, lo.y);
        *reinterpret_cast<uint2*>(&As[buf][0][am][ak]) = hi;
        *reinterpret_cast<uint2*>(&As[buf][1][am][ak]) = mid;
        *reinterpret_cast<uint2*>(&As[buf][2][am][ak]) = lo;
        uint4* dst = reinterpret_cast<uint4*>(&Bs[buf][0][0][0]);
        dst[tid] = rb0;
        if (tid < 256) dst[512 + tid] = rb1;
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;
    const int row = lane & 31, kh = (lane >> 5) * 8;

    load(0);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) load(kt + 1);
        bf16x8 a3[3], b3[TN][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            a3[pl] = *reinterpret_cast<const bf16x8*>(&As[cur][pl][wm + row][kh]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b3[j][pl] = *reinterpret_cast<const bf16x8*>(&Bs[cur][pl][wn + j * 32 + row][kh]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {       // smallest contributions first
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[2], b3[j][0], acc[j], 0, 0, 0);     // lo . hi
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[j][2], acc[j], 0, 0, 0);     // hi . lo
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[j][1], acc[j], 0, 0, 0);     // mid . mid
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[j][0], acc[j], 0, 0, 0);     // mid . hi
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[j][1], acc[j], 0, 0, 0);     // hi . mid
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[j][0], acc[j], 0, 0, 0);     // hi . hi
        }
        if (kt + 1 < KT) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: bias, per-tile row statistics (the arithmetic and the layout of nm_gemm.hip's STATS epilogue:
    // first maximum of ascending columns, sum exp(x - max) with expf), logits stored when asked for
    float* smem = reinterpret_cast<float*>(lds_raw);
    float* T = smem;
    float* pmax = smem + HR * TS;
    int* parg = reinterpret_cast<int*>(pmax + NT);
    float* psum = pmax + 2 * NT;
    constexpr int TPR = NT / HR, CW = BN / TPR;               // 8 threads per row, 16 columns each
    bool okj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn + j * 32 + (lane & 31);
        okj[j] = col < g.N;
        const float bv = (okj[j] && g.bias) ? g.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] += bv;
    }
    const int tiles_n = (int)gridDim.x / tiles_m;
    const int rr = tid % HR, q = tid / HR;
    for (int h = 0; h < BM / HR; ++h) {
        __syncthreads();
        const int rbase = wm - h * HR;                        // this wave's 32 rows inside the pass (wave-uniform)
        if (rbase >= 0 && rbase < HR) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    T[(rbase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TS + wn + j * 32 + (lane & 31)] =
                        okj[j] ? acc[j][r] : -INFINITY;
        }
        __syncthreads();
        const float* trow = T + rr * TS + q * CW;
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const float v = trow[c];
            if (v > best) { best = v; bi = n0 + q * CW + c; }
        }
        pmax[q * HR + rr] = best;
        parg[q * HR + rr] = bi;
        __syncthreads();
        float m = pmax[rr];
#pragma unroll
        for (int w = 1; w < TPR; ++w) m = fmaxf(m, pmax[w * HR + rr]);
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < CW; ++c) sum += expf(trow[c] - m);
        psum[q * HR + rr] = sum;
        __syncthreads();
        const int orow = m0 + h * HR + tid;
        if (tid < HR && orow < g.M) {
            float tot = psum[tid];
            int a = parg[tid];
#pragma unroll
            for (int w = 1; w < TPR; ++w) tot += psum[w * HR + tid];
            float bmx = pmax[tid];
#pragma unroll
            for (int w = 1; w < TPR; ++w)
                if (pmax[w * HR + tid] > bmx) { bmx = pmax[w * HR + tid]; a = parg[w * HR + tid]; }
            float4 rec;
            rec.x = bmx; rec.y = tot; rec.z = __int_as_float(a); rec.w = 0.0f;
            *reinterpret_cast<float4*>(g.stats + ((long)orow * tiles_n + bn) * 4) = rec;
        }
    }
    if (!g.C) return;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn + j * 32 + (lane & 31);
        if (col >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (orow < g.M) g.C[(long)orow * g.ldc + col] = acc[j][r];
        }
    }
}

// ---- the registry: which weight matrices of this process have split planes (per device pointer).  A handful of
// entries (one per decoding model); nm_logits_stats_gemm looks its B operand up.
namespace {
struct SplitEntry { const float* w; const unsigned short* planes; long N, K; int trans_b; };
std::mutex g_split_mu;
std::vector<SplitEntry> g_split;
}

extern "C" int64_t nm_proj_split_bytes(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0 || K % 16) return 0;
    return ((N + 127) / 128) * (K / 16) * 3 * 2048 * (int64_t)sizeof(unsigned short);
}

// planes <- the three bf16 planes of W ([K][N], or [N][K] when trans_b), and W is registered: from now on
// nm_logits_stats_gemm with this B pointer (same N, K, trans_b, 128-column statistics tiles) runs on the bf16 cores.
extern "C" int nm_proj_split_prepare(void* stream, const float* W, int64_t ldw, int trans_b, int64_t N, int64_t K,
                                     void* planes, int64_t planes_bytes) {
    NM_REQUIRE(W && planes && N > 0 && K > 0 && K % 16 == 0, "nm_proj_split_prepare: bad arguments (K %% 16 == 0)");
    NM_REQUIRE(planes_bytes >= nm_proj_split_bytes(N, K) && nm_aligned16(planes), "nm_proj_split_prepare: planes too small");
    const int blocks = (int)(((N + 127) / 128) * (K / 16));
    hipLaunchKernelGGL(split3_planes_kernel, dim3(blocks), dim3(256), 0, nm_stream(stream), W, (long)ldw, trans_b,
                       (int)N, (int)K, reinterpret_cast<unsigned short*>(planes));
    {
        std::lock_guard<std::mutex> lock(g_split_mu);
        bool found = false;
        for (auto& e : g_split)
            if (e.w == W) { e = SplitEntry{W, reinterpret_cast<const unsigned short*>(planes), (long)N, (long)K, trans_b}; found = true; }
        if (!found) g_split.push_back(SplitEntry{W, reinterpret_cast<const unsigned short*>(planes), (long)N, (long)K, trans_b});
    }
    NM_LAUNCH_CHECK("nm_proj_split_prepare");
}

extern "C" int nm_proj_split_forget(const float* W) {       // W == NULL: forget every matrix
    std::lock_guard<std::mutex> lock(g_split_mu);
    if (!W) g_split.clear();
    else
        for (size_t i = 0; i < g_split.size(); ++i)
            if (g_split[i].w == W) { g_split.erase(g_split.begin() + i); break; }
    return NM_OK;
}

// called by nm_logits_stats_gemm: true when the product was launched here
bool nm_proj_split_try(hipStream_t st, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       const float* B, const float* bias, float* C, int64_t ldc, float* stats, int stats_tile) {
    const unsigned short* planes = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_split_mu);
        for (const auto& e : g_split)
            if (e.w == B && e.N == N && e.K == K && e.trans_b == trans_b) planes = e.planes;
    }
    if (!planes || stats_tile != 128 || K % 16 || lda % 4 || !nm_aligned16(A)) return false;
    Split6Args g{A, (long)lda, planes, bias, C, (long)ldc, stats, (int)M, (int)N, (int)K};
    const int tm = nm_cdiv(M, 128), tn = nm_cdiv(N, 128);
    hipLaunchKernelGGL(gemm_split6_stats, dim3(tm * tn), dim3(512), 0, st, g, tm);
    return true;
}

