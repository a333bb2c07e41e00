// Shared argument blocks and element-wise pieces of the scaled dot-product attention kernels
// (nm_sdp_attention.hip: wave-per-query VALU kernels; nm_sdp_mfma.hip: matrix-core kernels).
#pragma once
#include "nm_common.h"

struct SdpArgs {
    const float* q; long q_bs;        // [Bq, Tq, H*dh], batch stride in floats
    const float* k; long k_bs;        // [Bk, Tk, H*dh]
    const float* v; long v_bs;
    const float* mask; long mask_bs;  // [Bk, Tk] float 0/1 or null
    float* ctx; long ctx_bs;          // [Bq, Tq, H*dh]
    float* weights;                   // [Bq, H, Tq, Tk] softmax output (before dropout) or null
    int Bq, rpk, Tq, Tk, H, dh, causal;
    float scale, keep_prob, inv_keep;
    uint32_t salt;
    const uint32_t* step;             // optional device scalar: salt += step * 0x9E3779B9 (graph replays)
};

__device__ __forceinline__ uint32_t sdp_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21f0aaadu;
    x ^= x >> 15;
    x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}

// dropout factor of weight element (b,h,i,j): same counter-based mask as nm_dropout over the
// flattened [Bq,H,Tq,Tk] tensor
__device__ __forceinline__ float sdp_keep(const SdpArgs& p, int b, int h, int i, int j) {
    if (p.keep_prob >= 1.0f) return 1.0f;
    const uint32_t idx = (uint32_t)((((long)b * p.H + h) * p.Tq + i) * p.Tk + j);
    const uint32_t salt = p.salt + (p.step ? p.step[0] * 0x9E3779B9u : 0u);
    const uint32_t bits = sdp_mix32(idx * 0x9E3779B1u + salt);
    const float uni = (float)(bits >> 8) * (1.0f / 16777216.0f);
    return (p.keep_prob + uni >= 1.0f) ? p.inv_keep : 0.0f;
}

__device__ __forceinline__ float sdp_masked_energy(const SdpArgs& p, float e, int i, int j, float m) {
    if (p.causal && j > i + p.Tk - p.Tq) e = -1e9f;
    if (p.mask) e = e * m + (1.0f - m) * -1e9f;
    return e;
}

struct SdpBwdArgs {
    SdpArgs f;                 // forward arguments (q, k, v, mask, weights = saved softmax output)
    const float* dctx; long dctx_bs;
    float* dq; long dq_bs;     // written
    float* dk; long dk_bs;     // written (or accumulated when f.rpk > 1 is not supported: rpk must be 1)
    float* dv; long dv_bs;
    float* de;                 // workspace [Bq, H, Tq, Tk]: energy gradients
    int accumulate;            // dq/dk/dv += instead of =
};

// Matrix-core paths (nm_sdp_mfma.hip).  Return true when the shape was taken (launch issued), false when the caller
// has to fall back to the wave-per-query kernels.
bool nm_sdp_mfma_fwd(const SdpArgs& p, hipStream_t stream);
bool nm_sdp_mfma_bwd(const SdpBwdArgs& a, hipStream_t stream);
