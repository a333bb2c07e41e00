// Shared helpers for libnmhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define NM_OK 0
#define NM_ERR_ARG -1
#define NM_ERR_HIP -2
#define NM_ERR_WORKSPACE -3

// thread-local last-error text, exposed through nm_last_error()
extern thread_local char nm_err_buf[512];

#define NM_FAIL(code, ...)                                   \
    do {                                                     \
        snprintf(nm_err_buf, sizeof(nm_err_buf), __VA_ARGS__); \
        return (code);                                       \
    } while (0)

#define NM_REQUIRE(cond, ...)                                \
    do {                                                     \
        if (!(cond)) NM_FAIL(NM_ERR_ARG, __VA_ARGS__);       \
    } while (0)

#define NM_LAUNCH_CHECK(name)                                              \
    do {                                                                   \
        hipError_t e_ = hipGetLastError();                                 \
        if (e_ != hipSuccess)                                              \
            NM_FAIL(NM_ERR_HIP, "%s: launch failed: %s", name, hipGetErrorString(e_)); \
        return NM_OK;                                                      \
    } while (0)

static inline hipStream_t nm_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- library context (nm_create / nm_destroy / nm_ctx_bind, nm_host.hip) -----------------------------------
// Everything the library keeps between calls lives in a context: the A/B and tuning switches (read from the
// environment ONCE, when the context is created -- never into function-local statics), the HIP-event pool of the
// live attention-step timer and the device the context was made for.  Entry points take the context that is
// bound to the calling thread (nm_ctx_bind); a thread that never bound one uses the process default context,
// created on first use.
#include <atomic>
#include <utility>
#include <vector>

struct NmSwitches {
    int attn_maxrows;      // NM_ATTN_MAXROWS   rows per split-S chunk, 1..16 (12)
    bool attn_nomerge;     // NM_ATTN_NOMERGE   separate combine launch instead of the in-kernel merge
    bool attn_nofast;      // NM_ATTN_NOFAST    vectorised attention kernels off
    int attn_whole;        // NM_ATTN_WHOLE     -1 dispatch by measured crossover, 0 never, 1 whenever possible
    bool aeb_wide_off;     // NM_AEB_WIDE=0     attention energies backward in passes of <= 16 positions
    bool gemm_no16;        // NM_GEMM_NO16      16x16 skinny tiles off
    int gemm_swz;          // NM_GEMM_SWZ       XCD-aware tile order (1)
    bool gemm_nostore;     // NM_GEMM_NOSTORE   timing ablation: results are NOT written
    int gemm_sk;           // NM_GEMM_SK        split-K override (0 = makespan model)
    int gemm_cfg;          // NM_GEMM_CFG       tile configuration of the large GEMMs (1)
    int gemm_chains;       // NM_GEMM_CHAINS    interleaved accumulation chains of the 64x64 tiles: 1, 2 or 4 (1)
    int gemm_cfg64;        // NM_GEMM_CFG64     tile of the products too small for 512 tiles of 128x128: 0 = 64x64 (4 waves),
                           //                   1 = 128x64, 2 = 64x128, 3 = 128x128 with 4 waves (tools/gemm_mid_sweep.py)
    int gemm_bg_wgs;       // NM_GEMM_BG_WGS    workgroups per CU of a background GEMM (algo 4): 1..3, 0 = uncapped (1)
    int gemm_bg_cfg;       // NM_GEMM_BG_CFG    tile of a background GEMM: 1 = 128x128, 2 = 256x128 (1)
    int background;        // (nm_ctx_set_background, not an environment switch) launches run beside a foreground loop
    int step_prio;         // NM_STEP_PRIO      skinny (time-loop) GEMM kernels raise their wave priority (1)
    int stats_cfg;         // NM_STATS_CFG      statistics-GEMM tile / prefetch bits (3)
    bool stats_ablate;     // NM_STATS_ABLATE   timing ablation: statistics epilogue skipped
    int beam_ns;           // NM_BEAM_NS        slices per hypothesis row override (0 = by vocabulary size)
    bool sdp_mfma;         // NM_SDP_MFMA=0     matrix-core attention kernels off
    int medium_m;          // NM_STEP_MEDIUM    medium-M (beam search) step groups: 1 on, 0 off
    bool sdp_decode;       // NM_SDP_DECODE=0   wave-per-(row, head) kernel of cached decoding steps off
};

struct NmCtx {
    uint32_t magic;
    int device;
    NmSwitches sw;
    bool prof_on;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
    size_t prof_used;
};

// the context of the calling thread (never null)
NmCtx* nm_cur();
// next free event pair of the context's timer (null if events cannot be created)
std::pair<hipEvent_t, hipEvent_t>* nm_prof_next_pair(NmCtx* c);

static inline bool nm_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int nm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ bool nm_aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ float nm_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float nm_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// Wave reductions on the DPP crossbar instead of ds_bpermute (what __shfl_xor lowers to: an LDS-pipe round trip per
// step, six dependent ones per reduction).  quad_perm x2 -> row_half_mirror -> row_mirror leave every lane with the
// sum of its row of 16; row_bcast:15 / row_bcast:31 carry the rows' sums up to lane 63; v_readlane hands the total
// to every lane.  ONLY for call sites where all 64 lanes are active (EXEC full).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float nm_dpp(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float nm_wave_sum_dpp(float v) {
    v += nm_dpp<0xB1, 0xf>(0.0f, v);            // quad_perm [1,0,3,2]
    v += nm_dpp<0x4E, 0xf>(0.0f, v);            // quad_perm [2,3,0,1]
    v += nm_dpp<0x141, 0xf>(0.0f, v);           // row_half_mirror
    v += nm_dpp<0x140, 0xf>(0.0f, v);           // row_mirror
    v += nm_dpp<0x142, 0xa>(0.0f, v);           // row_bcast:15 -> rows 1, 3
    v += nm_dpp<0x143, 0xc>(0.0f, v);           // row_bcast:31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float nm_wave_max_dpp(float v) {
    v = fmaxf(v, nm_dpp<0xB1, 0xf>(-INFINITY, v));
    v = fmaxf(v, nm_dpp<0x4E, 0xf>(-INFINITY, v));
    v = fmaxf(v, nm_dpp<0x141, 0xf>(-INFINITY, v));
    v = fmaxf(v, nm_dpp<0x140, 0xf>(-INFINITY, v));
    v = fmaxf(v, nm_dpp<0x142, 0xa>(-INFINITY, v));
    v = fmaxf(v, nm_dpp<0x143, 0xc>(-INFINITY, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// tanh with ~1e-7 absolute error: 1 - 2/(1+exp(2|x|)), sign restored.
__device__ __forceinline__ float nm_tanh(float x) {
    float ax = fabsf(x);
    float e = __expf(2.0f * ax);
    float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);   // v_rcp_f32: 1 ulp
    return copysignf(t, x);
}
__device__ __forceinline__ float nm_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh(a + b) = 1 - 2 / (1 + exp(2a) exp(2b)): when one of the two exponentials is shared by many elements (a key row
// that several queries read, a query that many key rows read) an element costs ONE quarter-rate transcendental
// (v_rcp_f32) instead of nm_tanh's two (v_exp_f32 + v_rcp_f32).  ~1e-7 absolute error, like nm_tanh.  A product that
// overflows gives +1, one that underflows gives -1; nm_exp2x clamps its argument to NM_EXP2X_MAX so that neither
// factor is ever inf or 0 (no inf * 0), which is exact as long as |a|, |b| <= NM_EXP2X_MAX: callers check that
// (max |.| of what they exponentiate) and take nm_tanh(a + b) otherwise.
#define NM_EXP2X_MAX 43.0f
__device__ __forceinline__ float nm_exp2x(float x) { return __expf(2.0f * fminf(fmaxf(x, -NM_EXP2X_MAX), NM_EXP2X_MAX)); }
__device__ __forceinline__ float nm_tanh_prod(float ea, float eb) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(fmaf(ea, eb, 1.0f));
}

// nm_gemm_bf16x3.hip: the vocabulary projection on the bf16 matrix cores for weights registered with
// nm_proj_split_prepare (opt-in); true when the product was launched
bool nm_proj_split_try(hipStream_t st, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       const float* B, const float* bias, float* C, int64_t ldc, float* stats, int stats_tile);


// nm_proj.hip: the decoding steps' vocabulary projection as an activation-stationary stream over the weights (W stored
// [K, N], K a multiple of 128 up to 512, a few row tiles); true when the product was launched
bool nm_proj_astat_try(hipStream_t st, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc, float* stats,
                       int stats_tile);
