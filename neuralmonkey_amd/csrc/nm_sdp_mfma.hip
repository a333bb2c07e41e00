// libnmhip: multi-head scaled dot-product attention (attention/scaled_dot_product.py:98-226) on the gfx950 matrix
// cores -- the training / encoding shapes of the Transformer (Tq, Tk <= 128 positions, head width 16..128).  Decoding
// steps (one query against a key/value cache) and anything else stay on the wave-per-query kernels of
// nm_sdp_attention.hip; both produce the same tensors and are tested against the same oracle.
//
// v_mfma_f32_16x16x4_f32 (exact f32, a k-ordered fma chain): D[m][n] += sum_k A[m][k] B[k][n] with
//     A: lane l holds A[m = l & 15][k = l >> 4]      B: lane l holds B[k = l >> 4][n = l & 15]
//     D: lane l, register r holds D[m = 4 (l >> 4) + r][n = l & 15]
// Everything is computed TRANSPOSED so that the accumulator layout of one product is the B-operand layout of the next
// and the softmax matrix never leaves the registers:
//     forward, a wave owns 16 queries i (n = i):
//         S^T[j][i]  = sum_c K[j][c] Qs[i][c]        A = K tile (LDS), B = the wave's scaled queries (registers)
//         P = softmax over j: 16 values in the lane + the 4 lane groups (two cross-lane steps)
//         C^T[c][i]  = sum_j V[j][c] Pd[i][j]        A = V (LDS), B = the S^T accumulators themselves: lane (i, g),
//                                                    register (t, r) holds key j = 16 t + 4 g + r, so product step
//                                                    (t, r) simply uses k-slot g <-> that key on both operands
//     backward phase 1, a wave owns 16 queries:
//         dWd^T[j][i] = sum_c V[j][c] dO[i][c] ; dE = W (dW - sum_j dW W) (registers) ; dQ^T[c][i] = sum_j K[j][c] dE[i][j]
//     backward phase 2, a wave owns 16 keys j (n = j); dE went through LDS (the contraction index changes sides):
//         dK^T[c][j] = sum_i Qs[i][c] dE[i][j] ; dV^T[c][j] = sum_i dO[i][c] Wd[i][j]
// Sums over c take the head width in the order c = g dh/4 + kk (a lane reads dh/4 contiguous floats), sums over keys in
// the (t, r) order above: fp32 results differ from the wave-per-query kernels by summation order only.
//
// LDS strides: a tile that is read "by columns" (16 rows at one channel per instruction) has an odd row stride, a tile
// read "by rows" (16 channels of 4 rows) has a stride = 4 mod 8 -- both conflict-free for ds_read_b32 at dh = 64.
#include "nm_sdp.h"

#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float sdp_xor16(float v) { return __shfl_xor(v, 16, 64); }
__device__ __forceinline__ float sdp_xor32(float v) { return __shfl_xor(v, 32, 64); }

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// grid (Bq * H, ceil(Tq / 64)); 4 waves, each owning 16 queries.  NKT = key tiles of 16 (Tk <= 16 NKT),
// NDT = head width / 16.
template <int NKT, int NDT>
__global__ __launch_bounds__(256) void sdp_fwd_mfma_kernel(SdpArgs p) {
    constexpr int DH = 16 * NDT, KQ = 4 * NDT, ROWS = 16 * NKT;
    constexpr int LDK = DH + 1, LDV = DH + 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* vs = sm;                        // [ROWS][LDV] values, read by rows
    float* ks = vs + ROWS * LDV;           // [ROWS][LDK] keys, read by columns
    float* ms = ks + ROWS * LDK;           // [ROWS] key mask

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
    const int kb = b / p.rpk;
    const int d = p.H * DH;
    const float* kg = p.k + (long)kb * p.k_bs + (long)h * DH;
    const float* vg = p.v + (long)kb * p.v_bs + (long)h * DH;
    for (int idx = tid; idx < ROWS * NDT * 4; idx += 256) {
        const int j = idx / (NDT * 4), c = (idx - j * (NDT * 4)) * 4;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (j < p.Tk) {
            kv = *reinterpret_cast<const float4*>(kg + (long)j * d + c);
            vv = *reinterpret_cast<const float4*>(vg + (long)j * d + c);
        }
        *reinterpret_cast<float4*>(vs + j * LDV + c) = vv;
        float* kd = ks + j * LDK + c;
        kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
    }
    for (int j = tid; j < ROWS; j += 256) ms[j] = (p.mask && j < p.Tk) ? p.mask[(long)kb * p.mask_bs + j] : 1.0f;
    __syncthreads();

    const int i0 = (int)blockIdx.y * 64 + wave * 16;
    if (i0 >= p.Tq) return;                          // no barrier below
    const int l15 = lane & 15, g = lane >> 4;
    const int i = i0 + l15;
    const bool iok = i < p.Tq;

    float qf[KQ];
    {
        const float* qg = p.q + (long)b * p.q_bs + (long)(iok ? i : p.Tq - 1) * d + (long)h * DH + g * KQ;
#pragma unroll
        for (int c4 = 0; c4 < NDT; ++c4) {
            const float4 t = *reinterpret_cast<const float4*>(qg + 4 * c4);
            qf[4 * c4 + 0] = t.x * p.scale; qf[4 * c4 + 1] = t.y * p.scale;
            qf[4 * c4 + 2] = t.z * p.scale; qf[4 * c4 + 3] = t.w * p.scale;
        }
    }
    f32x4 acc[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const float* kr = ks + l15 * LDK + g * KQ;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk)
#pragma unroll
            for (int t = 0; t < NKT; ++t) acc[t] = NM_MFMA16(kr[t * 16 * LDK + kk], qf[kk], acc[t]);
    }
    // energies -> softmax over the keys of query i: lane (i, g), register (t, r) <-> key 16 t + 4 g + r
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * g + r;
            const float e = j < p.Tk ? sdp_masked_energy(p, acc[t][r], i, j, ms[j]) : -INFINITY;
            acc[t][r] = e;
            mx = fmaxf(mx, e);
        }
    mx = fmaxf(mx, sdp_xor16(mx));
    mx = fmaxf(mx, sdp_xor32(mx));
    float se = 0.0f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ex = expf(acc[t][r] - mx);          // exact: softmax weights to 1 ulp
            acc[t][r] = ex;
            se += ex;
        }
    se += sdp_xor16(se);
    se += sdp_xor32(se);
    const float inv = 1.0f / se;
    float* wg = (p.weights && iok) ? p.weights + (((long)b * p.H + h) * p.Tq + i) * p.Tk : nullptr;
    const bool pairs = ((p.Tk & 1) == 0) && ((reinterpret_cast<uintptr_t>(p.weights) & 7u) == 0);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = acc[t][r] * inv;
        const int j0 = 16 * t + 4 * g;
        if (wg) {
            if (pairs) {                              // Tk even: a row starts on 8 bytes, (j, j+1) never straddles Tk
                if (j0 < p.Tk) *reinterpret_cast<float2*>(wg + j0) = make_float2(w[0], w[1]);
                if (j0 + 2 < p.Tk) *reinterpret_cast<float2*>(wg + j0 + 2) = make_float2(w[2], w[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j0 + r < p.Tk) wg[j0 + r] = w[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = w[r] * sdp_keep(p, b, h, i, j0 + r);
    }
    // context
    f32x4 cacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) cacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vr = vs + (16 * t + 4 * g + r) * LDV + l15;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) cacc[dt] = NM_MFMA16(vr[16 * dt], acc[t][r], cacc[dt]);
        }
    if (iok) {
        float* cg = p.ctx + (long)b * p.ctx_bs + (long)i * d + (long)h * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
            *reinterpret_cast<float4*>(cg + 16 * dt) = make_float4(cacc[dt][0], cacc[dt][1], cacc[dt][2], cacc[dt][3]);
    }
}

template <int NKT, int NDT>
static size_t sdp_fwd_mfma_lds() { return sizeof(float) * 16 * NKT * ((16 * NDT + 1) + (16 * NDT + 4) + 1); }

template <int NKT, int NDT>
static void sdp_fwd_mfma_launch(const SdpArgs& p, hipStream_t stream) {
    // the attribute is per DEVICE (library contexts exist per device: nm_create): one bit per device id
    static std::atomic<unsigned> attr_devs{0};
    const unsigned attr_bit = 1u << (nm_cur()->device & 31);
    const size_t lds = sdp_fwd_mfma_lds<NKT, NDT>();
    if (!(attr_devs.load(std::memory_order_relaxed) & attr_bit)) {
        (void)hipFuncSetAttribute((const void*)sdp_fwd_mfma_kernel<NKT, NDT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds);
        attr_devs.fetch_or(attr_bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((sdp_fwd_mfma_kernel<NKT, NDT>), dim3((unsigned)(p.Bq * p.H), (unsigned)((p.Tq + 63) / 64)),
                       dim3(256), lds, stream, p);
}

static bool sdp_mfma_enabled() { return nm_cur()->sw.sdp_mfma; }       // NM_SDP_MFMA=0: wave-per-query kernels only

static bool sdp_vec_ok(const void* ptr, long bs, long d) { return nm_aligned16(ptr) && (bs & 3) == 0 && (d & 3) == 0; }

template <int NDT>
static bool sdp_fwd_mfma_keys(const SdpArgs& p, hipStream_t stream) {
    if (p.Tk <= 32) sdp_fwd_mfma_launch<2, NDT>(p, stream);
    else if (p.Tk <= 64) sdp_fwd_mfma_launch<4, NDT>(p, stream);
    else if (p.Tk <= 128) sdp_fwd_mfma_launch<8, NDT>(p, stream);
    else return false;
    return true;
}

bool nm_sdp_mfma_fwd(const SdpArgs& p, hipStream_t stream) {
    if (!sdp_mfma_enabled()) return false;
    if (p.Tq < 8 || p.Tk > 128) return false;        // a decoding step is a matrix-vector product: wave-per-query kernel
    const long d = (long)p.H * p.dh;
    if (!sdp_vec_ok(p.q, p.q_bs, d) || !sdp_vec_ok(p.k, p.k_bs, d) || !sdp_vec_ok(p.v, p.v_bs, d) ||
        !sdp_vec_ok(p.ctx, p.ctx_bs, d))
        return false;
    switch (p.dh) {
        case 16: return sdp_fwd_mfma_keys<1>(p, stream);
        case 32: return sdp_fwd_mfma_keys<2>(p, stream);
        case 64: return sdp_fwd_mfma_keys<4>(p, stream);
        case 128: return sdp_fwd_mfma_keys<8>(p, stream);
        default: return false;
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// grid Bq * H; 4 waves.  NT = tiles of 16 positions (Tq, Tk <= 16 NT), NDT = head width / 16.
template <int NT, int NDT>
__global__ __launch_bounds__(256, (NT <= 4 && NDT <= 4 ? 3 : 1)) void sdp_bwd_mfma_kernel(SdpBwdArgs a) {
    constexpr int DH = 16 * NDT, KQ = 4 * NDT, ROWS = 16 * NT;
    constexpr int LDR = DH + 4, LDC = DH + 1, LDE = ROWS + 4;
    const SdpArgs& p = a.f;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* ra = sm;                        // phase 1: K by rows [ROWS][LDR]    phase 2: scaled Q by rows
    float* rb = ra + ROWS * LDR;           // phase 1: V by columns [ROWS][LDC] phase 2: dO by rows [ROWS][LDR]
    float* des = rb + ROWS * LDR;          // [ROWS queries][LDE] energy gradients
    float* ms = des + ROWS * LDE;          // [ROWS] key mask

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
    const int d = p.H * DH;
    const float* kg = p.k + (long)b * p.k_bs + (long)h * DH;
    const float* vg = p.v + (long)b * p.v_bs + (long)h * DH;
    const float* qg = p.q + (long)b * p.q_bs + (long)h * DH;
    const float* gg = a.dctx + (long)b * a.dctx_bs + (long)h * DH;
    const float* wg = p.weights + ((long)b * p.H + h) * p.Tq * p.Tk;
    for (int idx = tid; idx < ROWS * NDT * 4; idx += 256) {
        const int j = idx / (NDT * 4), c = (idx - j * (NDT * 4)) * 4;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (j < p.Tk) {
            kv = *reinterpret_cast<const float4*>(kg + (long)j * d + c);
            vv = *reinterpret_cast<const float4*>(vg + (long)j * d + c);
        }
        *reinterpret_cast<float4*>(ra + j * LDR + c) = kv;
        float* vd = rb + j * LDC + c;
        vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
    }
    for (int idx = tid; idx < ROWS * LDE / 4; idx += 256)
        reinterpret_cast<float4*>(des)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = tid; j < ROWS; j += 256) ms[j] = (p.mask && j < p.Tk) ? p.mask[(long)b * p.mask_bs + j] : 1.0f;
    __syncthreads();

    // ---- phase 1: query tiles
    const int nqt = (p.Tq + 15) / 16, nkt = (p.Tk + 15) / 16;
    for (int it = wave; it < nqt; it += 4) {
        const int i = 16 * it + l15;
        const bool iok = i < p.Tq;
        float dof[KQ];
        {
            const float* gr = gg + (long)(iok ? i : p.Tq - 1) * d + g * KQ;
#pragma unroll
            for (int c4 = 0; c4 < NDT; ++c4) {
                const float4 t = *reinterpret_cast<const float4*>(gr + 4 * c4);
                dof[4 * c4 + 0] = t.x; dof[4 * c4 + 1] = t.y; dof[4 * c4 + 2] = t.z; dof[4 * c4 + 3] = t.w;
            }
        }
        // saved softmax output of the wave's queries: lane (i, g), register (t, r) <-> key 16 t + 4 g + r
        float w[NT][4];
        {
            const float* wr = wg + (long)i * p.Tk;
            const bool pairs = ((p.Tk & 1) == 0) && ((reinterpret_cast<uintptr_t>(p.weights) & 7u) == 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int j0 = 16 * t + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[t][r] = 0.0f;
                if (iok) {
                    if (pairs) {
                        if (j0 < p.Tk) {
                            const float2 x = *reinterpret_cast<const float2*>(wr + j0);
                            w[t][0] = x.x; w[t][1] = x.y;
                        }
                        if (j0 + 2 < p.Tk) {
                            const float2 x = *reinterpret_cast<const float2*>(wr + j0 + 2);
                            w[t][2] = x.x; w[t][3] = x.y;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (j0 + r < p.Tk) w[t][r] = wr[j0 + r];
                    }
                }
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const float* vr = rb + l15 * LDC + g * KQ;
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = NM_MFMA16(vr[t * 16 * LDC + kk], dof[kk], acc[t]);
        }
        float dot = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dw = acc[t][r] * sdp_keep(p, b, h, i, 16 * t + 4 * g + r);    // through the weight dropout
                acc[t][r] = dw;
                dot += dw * w[t][r];
            }
        dot += sdp_xor16(dot);
        dot += sdp_xor32(dot);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * t + 4 * g + r;
                float de = w[t][r] * (acc[t][r] - dot);                 // softmax backward (0 where w was not loaded)
                if (p.causal && j > i + p.Tk - p.Tq) de = 0.0f;         // tf.where passes no gradient
                de *= ms[j];                                            // e*m + const
                acc[t][r] = de;
            }
            *reinterpret_cast<float4*>(des + i * LDE + 16 * t + 4 * g) =
                make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        }
        f32x4 cacc[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) cacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* kr = ra + (16 * t + 4 * g + r) * LDR + l15;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) cacc[dt] = NM_MFMA16(kr[16 * dt], acc[t][r], cacc[dt]);
            }
        if (iok) {
            float* dqg = a.dq + (long)b * a.dq_bs + (long)i * d + (long)h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                float4 o = make_float4(cacc[dt][0] * p.scale, cacc[dt][1] * p.scale, cacc[dt][2] * p.scale,
                                       cacc[dt][3] * p.scale);
                float4* dst = reinterpret_cast<float4*>(dqg + 16 * dt);
                if (a.accumulate) {
                    const float4 old = *dst;
                    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                }
                *dst = o;
            }
        }
    }
    __syncthreads();                      // K / V are dead, every energy gradient is in LDS
    for (int idx = tid; idx < ROWS * NDT * 4; idx += 256) {
        const int i = idx / (NDT * 4), c = (idx - i * (NDT * 4)) * 4;
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
        if (i < p.Tq) {
            qv = *reinterpret_cast<const float4*>(qg + (long)i * d + c);
            gv = *reinterpret_cast<const float4*>(gg + (long)i * d + c);
            qv.x *= p.scale; qv.y *= p.scale; qv.z *= p.scale; qv.w *= p.scale;
        }
        *reinterpret_cast<float4*>(ra + i * LDR + c) = qv;
        *reinterpret_cast<float4*>(rb + i * LDR + c) = gv;
    }
    __syncthreads();

    // ---- phase 2: key tiles
    for (int jt = wave; jt < nkt; jt += 4) {
        const int j = 16 * jt + l15;
        const bool jok = j < p.Tk;
        // dropped weights of key j for the queries 16 it + 4 g + r (second read of the saved softmax output: L2)
        float wd[NT][4];
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * g + r;
                wd[it][r] = (jok && i < p.Tq) ? wg[(long)i * p.Tk + j] : 0.0f;
            }
        if (p.keep_prob < 1.0f) {
#pragma unroll
            for (int it = 0; it < NT; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) wd[it][r] *= sdp_keep(p, b, h, 16 * it + 4 * g + r, j);
        }
        f32x4 dk[NDT], dv[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < NT; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * it + 4 * g + r;
                const float e = des[i * LDE + j];
                const float* qr = ra + i * LDR + l15;
                const float* gr = rb + i * LDR + l15;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    dk[dt] = NM_MFMA16(qr[16 * dt], e, dk[dt]);
                    dv[dt] = NM_MFMA16(gr[16 * dt], wd[it][r], dv[dt]);
                }
            }
            // without this fence hipcc hoists the LDS reads of ALL query tiles above the first MFMA (16 NT (1 + 2 NDT)
            // live values: 224 registers + spills at NT = NDT = 4); one tile of reads ahead is enough
            __builtin_amdgcn_sched_barrier(0);
        }
        if (jok) {
            float* dkg = a.dk + (long)b * a.dk_bs + (long)j * d + (long)h * DH + 4 * g;
            float* dvg = a.dv + (long)b * a.dv_bs + (long)j * d + (long)h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                float4 ok = make_float4(dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]);
                float4 ov = make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
                float4* pk = reinterpret_cast<float4*>(dkg + 16 * dt);
                float4* pv = reinterpret_cast<float4*>(dvg + 16 * dt);
                if (a.accumulate) {
                    const float4 a0 = *pk, a1 = *pv;
                    ok.x += a0.x; ok.y += a0.y; ok.z += a0.z; ok.w += a0.w;
                    ov.x += a1.x; ov.y += a1.y; ov.z += a1.z; ov.w += a1.w;
                }
                *pk = ok;
                *pv = ov;
            }
        }
    }
}

template <int NT, int NDT>
static size_t sdp_bwd_mfma_lds() {
    return sizeof(float) * 16 * NT * (2 * (16 * NDT + 4) + (16 * NT + 4) + 1);
}

template <int NT, int NDT>
static bool sdp_bwd_mfma_launch(const SdpBwdArgs& a, hipStream_t stream) {
    static std::atomic<unsigned> attr_devs{0};                     // per device, as in sdp_fwd_mfma_launch
    const unsigned attr_bit = 1u << (nm_cur()->device & 31);
    const size_t lds = sdp_bwd_mfma_lds<NT, NDT>();
    if (lds > 160 * 1024) return false;
    if (!(attr_devs.load(std::memory_order_relaxed) & attr_bit)) {
        (void)hipFuncSetAttribute((const void*)sdp_bwd_mfma_kernel<NT, NDT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds);
        attr_devs.fetch_or(attr_bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((sdp_bwd_mfma_kernel<NT, NDT>), dim3((unsigned)(a.f.Bq * a.f.H)), dim3(256), lds, stream, a);
    return true;
}

template <int NDT>
static bool sdp_bwd_mfma_tiles(const SdpBwdArgs& a, hipStream_t stream) {
    const int t = a.f.Tq > a.f.Tk ? a.f.Tq : a.f.Tk;
    if (t <= 32) return sdp_bwd_mfma_launch<2, NDT>(a, stream);
    if (t <= 64) return sdp_bwd_mfma_launch<4, NDT>(a, stream);
    if (t <= 128) return sdp_bwd_mfma_launch<8, NDT>(a, stream);
    return false;
}

bool nm_sdp_mfma_bwd(const SdpBwdArgs& a, hipStream_t stream) {
    const SdpArgs& p = a.f;
    if (!sdp_mfma_enabled()) return false;
    if (p.rpk != 1 || p.Tq < 8 || p.Tq > 128 || p.Tk > 128) return false;
    const long d = (long)p.H * p.dh;
    if (!sdp_vec_ok(p.q, p.q_bs, d) || !sdp_vec_ok(p.k, p.k_bs, d) || !sdp_vec_ok(p.v, p.v_bs, d) ||
        !sdp_vec_ok(a.dctx, a.dctx_bs, d) || !sdp_vec_ok(a.dq, a.dq_bs, d) || !sdp_vec_ok(a.dk, a.dk_bs, d) ||
        !sdp_vec_ok(a.dv, a.dv_bs, d))
        return false;
    switch (p.dh) {
        case 16: return sdp_bwd_mfma_tiles<1>(a, stream);
        case 32: return sdp_bwd_mfma_tiles<2>(a, stream);
        case 64: return sdp_bwd_mfma_tiles<4>(a, stream);
        case 128: return sdp_bwd_mfma_tiles<8>(a, stream);
        default: return false;
    }
}
