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
