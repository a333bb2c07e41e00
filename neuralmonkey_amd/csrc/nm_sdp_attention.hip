// libnmhip: multi-head scaled dot-product attention (attention/scaled_dot_product.py:98-226) for the
// Transformer encoder / decoder (encoders/transformer.py:199-233, decoders/transformer.py:270-330).
//
//   energies = (q / sqrt(dh)) . k^T          per head, heads are column blocks of the model dim
//   [masked]   where(j <= i + Tk - Tq, e, -1e9)                       mask_future   (:72-93)
//   [key mask] e * m + (1 - m) * -1e9                                 mask_energies (:45-69)
//   w = softmax(e) ; w = dropout(w) ; ctx = w . v
//
// One workgroup per (query batch row, head): the K and V tiles of that head ([Tk, dh], padded rows)
// are staged once in LDS and reused by all Tq queries; a wave owns one query at a time
// (lane = key for the dot products and the softmax reductions, lane = channel for w.V).  At the
// Transformer-base shape (B=128, H=8, T=50, dh=64) that is 1024 workgroups and 26 KB of LDS each; the
// arithmetic is ~1 % of the layer's GEMM flops, so the kernel is sized for occupancy, not MFMA.
// The same kernel serves cached decoding (Tq = 1 against a [R, Tmax, d] key/value cache) and a
// beam sharing its encoder keys (query row r reads key batch r / rows_per_key).
//
// The backward kernel keeps Q, K, V and dctx tiles in LDS: phase 1 (wave per query) produces the
// energy gradients and dQ, phase 2 (wave per key) reduces dK and dV over the queries -- no atomics,
// deterministic.
#include "nm_sdp.h"

__global__ __launch_bounds__(256) void sdp_fwd_kernel(SdpArgs p) {
    extern __shared__ float sm[];
    const int ldt = p.dh + 1;                       // padded tile rows: conflict-free column walks
    float* ks = sm;                                 // [Tk][dh+1]
    float* vs = ks + (long)p.Tk * ldt;              // [Tk][dh+1]
    float* ms = vs + (long)p.Tk * ldt;              // [Tk] key mask
    float* qs = ms + p.Tk;                          // [4 waves][dh] scaled query of the wave
    float* ws = qs + 4 * p.dh;                      // [4 waves][Tk] weights of the wave's query

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
    const int kb = b / p.rpk;
    const int d = p.H * p.dh;
    const float* kg = p.k + (long)kb * p.k_bs + (long)h * p.dh;
    const float* vg = p.v + (long)kb * p.v_bs + (long)h * p.dh;
    for (int idx = tid; idx < p.Tk * p.dh; idx += 256) {
        const int j = idx / p.dh, c = idx - j * p.dh;
        ks[j * ldt + c] = kg[(long)j * d + c];
        vs[j * ldt + c] = vg[(long)j * d + c];
    }
    for (int j = tid; j < p.Tk; j += 256) ms[j] = p.mask ? p.mask[(long)kb * p.mask_bs + j] : 1.0f;
    __syncthreads();

    float* qw = qs + wave * p.dh;
    float* ww = ws + (long)wave * p.Tk;
    // every wave runs the same number of rounds so that the workgroup barriers below are uniform
    for (int i0 = 0; i0 < p.Tq; i0 += 4) {
        const int i = i0 + wave;
        const bool active = i < p.Tq;
        const float* qg = p.q + (long)b * p.q_bs + (long)i * d + (long)h * p.dh;
        if (active)
            for (int c = lane; c < p.dh; c += 64) qw[c] = qg[c] * p.scale;
        __syncthreads();
        // energies: lane = key
        float mx = -INFINITY;
        for (int j0 = 0; active && j0 < p.Tk; j0 += 64) {
            const int j = j0 + lane;
            float e = -INFINITY;
            if (j < p.Tk) {
                float acc = 0.0f;
                const float* kr = ks + j * ldt;
                for (int c = 0; c < p.dh; ++c) acc += qw[c] * kr[c];
                e = sdp_masked_energy(p, acc, i, j, ms[j]);
                ww[j] = e;
            }
            mx = fmaxf(mx, e);
        }
        mx = nm_wave_max(mx);
        // each lane only revisits the entries it wrote itself until the barrier
        float se = 0.0f;
        for (int j = lane; active && j < p.Tk; j += 64) {
            const float ex = expf(ww[j] - mx);
            ww[j] = ex;
            se += ex;
        }
        se = nm_wave_sum(se);
        const float inv = 1.0f / se;
        float* wg = (p.weights && active) ? p.weights + (((long)b * p.H + h) * p.Tq + i) * p.Tk : nullptr;
        for (int j = lane; active && j < p.Tk; j += 64) {
            const float w = ww[j] * inv;
            if (wg) wg[j] = w;
            ww[j] = w * sdp_keep(p, b, h, i, j);
        }
        __syncthreads();
        // context: lane = channel
        float* cg = p.ctx + (long)b * p.ctx_bs + (long)i * d + (long)h * p.dh;
        for (int c = lane; active && c < p.dh; c += 64) {
            float acc = 0.0f;
            for (int j = 0; j < p.Tk; ++j) acc += ww[j] * vs[j * ldt + c];
            cg[c] = acc;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// One query per row (a cached decoding step, Tq == 1): staging the head's K / V tiles in LDS for a single query
// leaves three of the four waves idle and walks the tiles with 4-byte loads (46.7 us per call at 640 rows x 8 heads
// against the ~20 us the bytes take).  Here a wave owns one (row, head): LPK = dh / 4 lanes x 16 bytes cover the
// head's channels of one key row, so every load instruction of the wave fetches 64 / LPK whole key rows, coalesced;
// energies are reduced inside the lane group, softmax is accumulated online per lane group (running max, sum and
// weighted value sum) with the key AND value loads of a pass in flight together, and the lane groups merge at the
// end.  `anc` (optional, [Bq][anc_ld] int32): the cache row that holds position j of row b's history -- beam search
// then re-points its hypotheses at their ancestors' keys and values instead of copying the cached prefixes of every
// layer at every step (decoders/transformer.py: TransformerStepper.reorder).
// ---------------------------------------------------------------------------------------------
template <int LPK>
__global__ __launch_bounds__(1024) void sdp_decode_kernel(SdpArgs p, const int* __restrict__ anc, long anc_ld) {
    constexpr int KPP = 64 / LPK;                   // keys per pass of the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int sub = lane % LPK, grp = lane / LPK;
    const int b = blockIdx.x, kb = b / p.rpk;
    const int d = p.H * p.dh;
    const int* arow = anc ? anc + (long)b * anc_ld : nullptr;
    for (int h = wave; h < p.H; h += nwaves) {
        const long col = (long)h * p.dh + 4 * sub;
        float4 q4 = *reinterpret_cast<const float4*>(p.q + (long)b * p.q_bs + col);
        q4.x *= p.scale; q4.y *= p.scale; q4.z *= p.scale; q4.w *= p.scale;
        float* wg = p.weights ? p.weights + ((long)b * p.H + h) * p.Tk : nullptr;
        float m = -INFINITY, ssum = 0.0f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int UN = 4;
        for (int j0 = grp; j0 < p.Tk; j0 += KPP * UN) {
            float4 k4[UN], v4[UN];
            float mk[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int j = min(j0 + u * KPP, p.Tk - 1);               // clamped duplicates are dropped below
                const long row = arow ? (long)arow[j] : (long)kb;
                k4[u] = *reinterpret_cast<const float4*>(p.k + row * p.k_bs + (long)j * d + col);
                v4[u] = *reinterpret_cast<const float4*>(p.v + row * p.v_bs + (long)j * d + col);
                mk[u] = p.mask ? p.mask[(long)kb * p.mask_bs + j] : 1.0f;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int j = j0 + u * KPP;
                float e = (q4.x * k4[u].x + q4.y * k4[u].y) + (q4.z * k4[u].z + q4.w * k4[u].w);
#pragma unroll
                for (int off = 1; off < LPK; off <<= 1) e += __shfl_xor(e, off, 64);
                if (j < p.Tk) {                                             // (uniform inside a lane group)
                    e = sdp_masked_energy(p, e, 0, j, mk[u]);
                    if (wg && sub == 0) wg[j] = e;                          // normalised below, by the same lane
                    const float mn = fmaxf(m, e);
                    const float sc = expf(m - mn), ex = expf(e - mn);       // exp(-inf) = 0 on the first key
                    const float w = ex * sdp_keep(p, b, h, 0, j);
                    ssum = ssum * sc + ex;
                    acc.x = acc.x * sc + w * v4[u].x; acc.y = acc.y * sc + w * v4[u].y;
                    acc.z = acc.z * sc + w * v4[u].z; acc.w = acc.w * sc + w * v4[u].w;
                    m = mn;
                }
            }
        }
        // merge the lane groups (a group that saw no key carries m = -inf, sum 0)
#pragma unroll
        for (int off = LPK; off < 64; off <<= 1) {
            const float mo = __shfl_xor(m, off, 64), so = __shfl_xor(ssum, off, 64);
            float4 ao;
            ao.x = __shfl_xor(acc.x, off, 64); ao.y = __shfl_xor(acc.y, off, 64);
            ao.z = __shfl_xor(acc.z, off, 64); ao.w = __shfl_xor(acc.w, off, 64);
            const float mn = fmaxf(m, mo);
            const float s1 = (m == -INFINITY) ? 0.0f : expf(m - mn), s2 = (mo == -INFINITY) ? 0.0f : expf(mo - mn);
            ssum = ssum * s1 + so * s2;
            acc.x = acc.x * s1 + ao.x * s2; acc.y = acc.y * s1 + ao.y * s2;
            acc.z = acc.z * s1 + ao.z * s2; acc.w = acc.w * s1 + ao.w * s2;
            m = mn;
        }
        const float inv = 1.0f / ssum;
        if (grp == 0) {
            acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
            *reinterpret_cast<float4*>(p.ctx + (long)b * p.ctx_bs + col) = acc;
        }
        if (wg && sub == 0)
            for (int j = grp; j < p.Tk; j += KPP) wg[j] = expf(wg[j] - m) * inv;
    }
}

// Tq == 1 dispatch.  Returns false when the shape is not taken (odd head widths, unaligned operands).
static bool sdp_decode_launch(const SdpArgs& p, const int* anc, long anc_ld, hipStream_t st) {
    if (p.Tq != 1 || p.dh % 4 || 64 % (p.dh / 4) || p.dh > 256) return false;
    const int d = p.H * p.dh;
    if (d % 4 || p.q_bs % 4 || p.k_bs % 4 || p.v_bs % 4 || p.ctx_bs % 4 || !nm_aligned16(p.q) || !nm_aligned16(p.k) ||
        !nm_aligned16(p.v) || !nm_aligned16(p.ctx))
        return false;
    const int waves = p.H < 16 ? p.H : 16;
    dim3 grid((unsigned)p.Bq), block(64 * waves);
    switch (p.dh / 4) {
        case 1: hipLaunchKernelGGL(sdp_decode_kernel<1>, grid, block, 0, st, p, anc, anc_ld); break;
        case 2: hipLaunchKernelGGL(sdp_decode_kernel<2>, grid, block, 0, st, p, anc, anc_ld); break;
        case 4: hipLaunchKernelGGL(sdp_decode_kernel<4>, grid, block, 0, st, p, anc, anc_ld); break;
        case 8: hipLaunchKernelGGL(sdp_decode_kernel<8>, grid, block, 0, st, p, anc, anc_ld); break;
        case 16: hipLaunchKernelGGL(sdp_decode_kernel<16>, grid, block, 0, st, p, anc, anc_ld); break;
        case 32: hipLaunchKernelGGL(sdp_decode_kernel<32>, grid, block, 0, st, p, anc, anc_ld); break;
        default: hipLaunchKernelGGL(sdp_decode_kernel<64>, grid, block, 0, st, p, anc, anc_ld); break;
    }
    return true;
}

static size_t sdp_fwd_lds(long Tk, long dh) { return sizeof(float) * (2 * Tk * (dh + 1) + Tk + 4 * dh + 4 * Tk); }

extern "C" int nm_sdp_attn_fwd(void* stream, const float* q, int64_t q_bs, const float* k, int64_t k_bs,
                               const float* v, int64_t v_bs, const float* key_mask, int64_t mask_bs, int64_t Bq,
                               int64_t rows_per_key, int64_t Tq, int64_t Tk, int64_t H, int64_t dh, int causal,
                               float keep_prob, uint32_t salt, const uint32_t* step, float* ctx, int64_t ctx_bs,
                               float* weights) {
    NM_REQUIRE(q && k && v && ctx, "nm_sdp_attn_fwd: null pointer");
    NM_REQUIRE(Bq > 0 && rows_per_key >= 1 && Bq % rows_per_key == 0 && Tq > 0 && Tk > 0 && H > 0 && dh > 0,
               "nm_sdp_attn_fwd: bad shape Bq=%ld Tq=%ld Tk=%ld H=%ld dh=%ld", (long)Bq, (long)Tq, (long)Tk, (long)H,
               (long)dh);
    NM_REQUIRE(keep_prob > 0.0f && keep_prob <= 1.0f, "nm_sdp_attn_fwd: keep_prob %g outside (0,1]", keep_prob);
    NM_REQUIRE(Bq * H < (1LL << 31), "nm_sdp_attn_fwd: grid too large");
    NM_REQUIRE(Bq * H * Tq * Tk < (1LL << 32) || keep_prob >= 1.0f, "nm_sdp_attn_fwd: dropout mask index overflow");
    const size_t lds = sdp_fwd_lds(Tk, dh);
    NM_REQUIRE(lds <= 160 * 1024, "nm_sdp_attn_fwd: Tk=%ld x dh=%ld key/value tile does not fit LDS", (long)Tk,
               (long)dh);
    SdpArgs p;
    p.q = q; p.q_bs = q_bs; p.k = k; p.k_bs = k_bs; p.v = v; p.v_bs = v_bs;
    p.mask = key_mask; p.mask_bs = mask_bs; p.ctx = ctx; p.ctx_bs = ctx_bs; p.weights = weights;
    p.Bq = (int)Bq; p.rpk = (int)rows_per_key; p.Tq = (int)Tq; p.Tk = (int)Tk; p.H = (int)H; p.dh = (int)dh;
    p.causal = causal;
    p.scale = 1.0f / sqrtf((float)dh);
    p.keep_prob = keep_prob; p.inv_keep = 1.0f / keep_prob; p.salt = salt; p.step = step;
    if (nm_sdp_mfma_fwd(p, nm_stream(stream))) {      // training / encoding shapes: matrix cores (nm_sdp_mfma.hip)
        NM_LAUNCH_CHECK("nm_sdp_attn_fwd (mfma)");
    }
    if (nm_cur()->sw.sdp_decode && sdp_decode_launch(p, nullptr, 0, nm_stream(stream))) {     // one query per row
        NM_LAUNCH_CHECK("nm_sdp_attn_fwd (decode)");
    }
    static std::atomic<unsigned> attr_devs{0};                     // the attribute is per device: one bit per device id
    const unsigned attr_bit = 1u << (nm_cur()->device & 31);
    if (!(attr_devs.load(std::memory_order_relaxed) & attr_bit)) {
        (void)hipFuncSetAttribute((const void*)sdp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_devs.fetch_or(attr_bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(sdp_fwd_kernel, dim3((unsigned)(Bq * H)), dim3(256), lds, nm_stream(stream), p);
    NM_LAUNCH_CHECK("nm_sdp_attn_fwd");
}

// One decoding step against a key/value cache whose rows are addressed through an ancestor table:
// position j of query row b lives in cache row ancestors[b * anc_ld + j] (decoders/transformer.py:493-516 with the
// beam search's gather of the cached prefixes, beam_search_decoder.py:218-330, folded into the addressing).
extern "C" int nm_sdp_attn_step(void* stream, const float* q, int64_t q_bs, const float* k, int64_t k_bs,
                                const float* v, int64_t v_bs, const float* key_mask, int64_t mask_bs, int64_t Bq,
                                int64_t Tk, int64_t H, int64_t dh, const int32_t* ancestors, int64_t anc_ld,
                                float* ctx, int64_t ctx_bs, float* weights) {
    NM_REQUIRE(q && k && v && ctx && ancestors, "nm_sdp_attn_step: null pointer");
    NM_REQUIRE(Bq > 0 && Tk > 0 && H > 0 && dh > 0 && anc_ld >= Tk && Bq < (1LL << 31),
               "nm_sdp_attn_step: bad shape Bq=%ld Tk=%ld H=%ld dh=%ld anc_ld=%ld", (long)Bq, (long)Tk, (long)H,
               (long)dh, (long)anc_ld);
    SdpArgs p;
    p.q = q; p.q_bs = q_bs; p.k = k; p.k_bs = k_bs; p.v = v; p.v_bs = v_bs;
    p.mask = key_mask; p.mask_bs = mask_bs; p.ctx = ctx; p.ctx_bs = ctx_bs; p.weights = weights;
    p.Bq = (int)Bq; p.rpk = 1; p.Tq = 1; p.Tk = (int)Tk; p.H = (int)H; p.dh = (int)dh; p.causal = 0;
    p.scale = 1.0f / sqrtf((float)dh);
    p.keep_prob = 1.0f; p.inv_keep = 1.0f; p.salt = 0; p.step = nullptr;
    NM_REQUIRE(sdp_decode_launch(p, ancestors, anc_ld, nm_stream(stream)),
               "nm_sdp_attn_step: dh=%ld must be 4, 8, 16, 32, 64, 128 or 256 and the operands 16-byte aligned",
               (long)dh);
    NM_LAUNCH_CHECK("nm_sdp_attn_step");
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// WLDS: the [Tq,Tk] energy-gradient and dropped-weight matrices of the head stay in LDS between the
// two phases (phase 2 walks them by columns); otherwise they go through global memory.
template <bool WLDS>
__global__ __launch_bounds__(256) void sdp_bwd_kernel(SdpBwdArgs a) {
    extern __shared__ float sm[];
    const SdpArgs& p = a.f;
    const int ldt = p.dh + 1;
    const int ldw = p.Tk + 1;
    float* ks = sm;                                 // [Tk][dh+1]
    float* vs = ks + (long)p.Tk * ldt;              // [Tk][dh+1]
    float* qs = vs + (long)p.Tk * ldt;              // [Tq][dh+1] scaled queries
    float* gs = qs + (long)p.Tq * ldt;              // [Tq][dh+1] dctx
    float* ms = gs + (long)p.Tq * ldt;              // [Tk]
    float* rw = ms + p.Tk;                          // [4 waves][Tk] row scratch
    float* des = rw + 4 * p.Tk;                     // [Tq][Tk+1] energy gradients        (WLDS)
    float* wds = des + (WLDS ? (long)p.Tq * ldw : 0);   // [Tq][Tk+1] dropout(weights)    (WLDS)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
    const int d = p.H * p.dh;
    const float* kg = p.k + (long)b * p.k_bs + (long)h * p.dh;
    const float* vg = p.v + (long)b * p.v_bs + (long)h * p.dh;
    const float* qg = p.q + (long)b * p.q_bs + (long)h * p.dh;
    const float* gg = a.dctx + (long)b * a.dctx_bs + (long)h * p.dh;
    for (int idx = tid; idx < p.Tk * p.dh; idx += 256) {
        const int j = idx / p.dh, c = idx - j * p.dh;
        ks[j * ldt + c] = kg[(long)j * d + c];
        vs[j * ldt + c] = vg[(long)j * d + c];
    }
    for (int idx = tid; idx < p.Tq * p.dh; idx += 256) {
        const int i = idx / p.dh, c = idx - i * p.dh;
        qs[i * ldt + c] = qg[(long)i * d + c] * p.scale;
        gs[i * ldt + c] = gg[(long)i * d + c];
    }
    for (int j = tid; j < p.Tk; j += 256) ms[j] = p.mask ? p.mask[(long)b * p.mask_bs + j] : 1.0f;
    __syncthreads();

    // phase 1: wave per query row -- dW, softmax gradient, energy gradient (to global), dQ
    float* row = rw + (long)wave * p.Tk;
    for (int i0 = 0; i0 < p.Tq; i0 += 4) {
        const int i = i0 + wave;
        const bool active = i < p.Tq;
        const long wbase = (((long)b * p.H + h) * p.Tq + (active ? i : 0)) * p.Tk;
        const float* wg = p.weights + wbase;
        float dot = 0.0f;
        for (int j = lane; active && j < p.Tk; j += 64) {
            float acc = 0.0f;
            const float* gr = gs + i * ldt;
            const float* vr = vs + j * ldt;
            for (int c = 0; c < p.dh; ++c) acc += gr[c] * vr[c];
            const float keep = sdp_keep(p, b, h, i, j);
            const float dw = acc * keep;                           // through the weight dropout
            row[j] = dw;
            dot += dw * wg[j];
            if (WLDS) wds[i * ldw + j] = wg[j] * keep;
        }
        dot = nm_wave_sum(dot);
        for (int j = lane; active && j < p.Tk; j += 64) {
            float de = wg[j] * (row[j] - dot);                     // softmax backward
            if (p.causal && j > i + p.Tk - p.Tq) de = 0.0f;        // tf.where passes no gradient
            de *= ms[j];                                           // e*m + const
            row[j] = de;
            if (WLDS) des[i * ldw + j] = de;
            else a.de[wbase + j] = de;
        }
        __syncthreads();
        float* dqg = a.dq + (long)b * a.dq_bs + (long)i * d + (long)h * p.dh;
        for (int c = lane; active && c < p.dh; c += 64) {
            float acc = 0.0f;
            for (int j = 0; j < p.Tk; ++j) acc += row[j] * ks[j * ldt + c];
            acc *= p.scale;
            dqg[c] = a.accumulate ? dqg[c] + acc : acc;
        }
        __syncthreads();
    }
    __threadfence_block();       // phase 2 reads the energy gradients other waves stored to global
    __syncthreads();
    // phase 2: wave per key row -- dK = de^T . q_scaled, dV = dropout(w)^T . dctx
    for (int j = wave; j < p.Tk; j += 4) {
        float* dkg = a.dk + (long)b * a.dk_bs + (long)j * d + (long)h * p.dh;
        float* dvg = a.dv + (long)b * a.dv_bs + (long)j * d + (long)h * p.dh;
        for (int c = lane; c < p.dh; c += 64) {
            float acck = 0.0f, accv = 0.0f;
            if (WLDS) {
                for (int i = 0; i < p.Tq; ++i) {
                    acck += des[i * ldw + j] * qs[i * ldt + c];
                    accv += wds[i * ldw + j] * gs[i * ldt + c];
                }
            } else {
                for (int i = 0; i < p.Tq; ++i) {
                    const long wi = (((long)b * p.H + h) * p.Tq + i) * p.Tk + j;
                    acck += a.de[wi] * qs[i * ldt + c];
                    accv += p.weights[wi] * sdp_keep(p, b, h, i, j) * gs[i * ldt + c];
                }
            }
            dkg[c] = a.accumulate ? dkg[c] + acck : acck;
            dvg[c] = a.accumulate ? dvg[c] + accv : accv;
        }
    }
}

extern "C" int nm_sdp_attn_bwd(void* stream, const float* q, int64_t q_bs, const float* k, int64_t k_bs,
                               const float* v, int64_t v_bs, const float* key_mask, int64_t mask_bs,
                               const float* weights, const float* dctx, int64_t dctx_bs, int64_t B, int64_t Tq,
                               int64_t Tk, int64_t H, int64_t dh, int causal, float keep_prob, uint32_t salt,
                               const uint32_t* step, float* dq, int64_t dq_bs, float* dk, int64_t dk_bs, float* dv, int64_t dv_bs,
                               float* de_workspace, int accumulate) {
    NM_REQUIRE(q && k && v && weights && dctx && dq && dk && dv && de_workspace, "nm_sdp_attn_bwd: null pointer");
    NM_REQUIRE(B > 0 && Tq > 0 && Tk > 0 && H > 0 && dh > 0, "nm_sdp_attn_bwd: bad shape");
    NM_REQUIRE(keep_prob > 0.0f && keep_prob <= 1.0f, "nm_sdp_attn_bwd: keep_prob %g outside (0,1]", keep_prob);
    NM_REQUIRE(B * H < (1LL << 31), "nm_sdp_attn_bwd: grid too large");
    size_t lds = sizeof(float) * (2 * (Tk + Tq) * (dh + 1) + Tk + 4 * Tk);
    NM_REQUIRE(lds <= 160 * 1024, "nm_sdp_attn_bwd: Tq=%ld Tk=%ld dh=%ld tiles do not fit LDS", (long)Tq, (long)Tk,
               (long)dh);
    const size_t lds_w = lds + sizeof(float) * 2 * Tq * (Tk + 1);
    const bool wlds = lds_w <= 160 * 1024;          // base shape (T=50, dh=64): 72 KB
    if (wlds) lds = lds_w;
    SdpBwdArgs a;
    SdpArgs& p = a.f;
    p.q = q; p.q_bs = q_bs; p.k = k; p.k_bs = k_bs; p.v = v; p.v_bs = v_bs;
    p.mask = key_mask; p.mask_bs = mask_bs; p.ctx = nullptr; p.ctx_bs = 0;
    p.weights = const_cast<float*>(weights);
    p.Bq = (int)B; p.rpk = 1; p.Tq = (int)Tq; p.Tk = (int)Tk; p.H = (int)H; p.dh = (int)dh; p.causal = causal;
    p.scale = 1.0f / sqrtf((float)dh);
    p.keep_prob = keep_prob; p.inv_keep = 1.0f / keep_prob; p.salt = salt; p.step = step;
    a.dctx = dctx; a.dctx_bs = dctx_bs; a.dq = dq; a.dq_bs = dq_bs; a.dk = dk; a.dk_bs = dk_bs;
    a.dv = dv; a.dv_bs = dv_bs; a.de = de_workspace; a.accumulate = accumulate;
    if (nm_sdp_mfma_bwd(a, nm_stream(stream))) {
        NM_LAUNCH_CHECK("nm_sdp_attn_bwd (mfma)");
    }
    static std::atomic<unsigned> attr_devs{0};                     // per device, as above
    const unsigned attr_bit = 1u << (nm_cur()->device & 31);
    if (!(attr_devs.load(std::memory_order_relaxed) & attr_bit)) {
        (void)hipFuncSetAttribute((const void*)sdp_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024);
        (void)hipFuncSetAttribute((const void*)sdp_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024);
        attr_devs.fetch_or(attr_bit, std::memory_order_relaxed);
    }
    if (wlds) hipLaunchKernelGGL(sdp_bwd_kernel<true>, dim3((unsigned)(B * H)), dim3(256), lds, nm_stream(stream), a);
    else hipLaunchKernelGGL(sdp_bwd_kernel<false>, dim3((unsigned)(B * H)), dim3(256), lds, nm_stream(stream), a);
    NM_LAUNCH_CHECK("nm_sdp_attn_bwd");
}

// ---------------------------------------------------------------------------------------------
// small utilities of the Transformer blocks
// ---------------------------------------------------------------------------------------------
// out[b,t,:] = x[b,t,:] + signal[t0 + t,:]   (position_signal, encoders/transformer.py:23-45, added to
// the embedded inputs :187-189) ; signal is a host-computed [Tmax, D] table
__global__ void add_position_kernel(const float* __restrict__ x, const float* __restrict__ signal,
                                    float* __restrict__ out, int T, int D, int t0, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int t = (int)((i / D) % T);
        out[i] = x[i] + signal[(long)(t0 + t) * D + c];
    }
}

extern "C" int nm_add_position(void* stream, const float* x, const float* signal, float* out, int64_t B, int64_t T,
                               int64_t D, int64_t t0) {
    NM_REQUIRE(x && signal && out && B >= 0 && T >= 0 && D > 0 && t0 >= 0, "nm_add_position: bad args");
    const long total = B * T * D;
    if (total == 0) return NM_OK;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_position_kernel, dim3(blocks), dim3(256), 0, nm_stream(stream), x, signal, out, (int)T,
                       (int)D, (int)t0, total);
    NM_LAUNCH_CHECK("nm_add_position");
}

// out[r*ld] = finished[r] ? 0 : 1 : the key-mask column a decoding step appends for its new position
// (decoders/transformer.py:493-497: input_mask grows by to_float(not finished))
__global__ void unfinished_mask_kernel(const int* __restrict__ finished, float* __restrict__ out, long ld, long n) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r * ld] = finished[r] ? 0.0f : 1.0f;
}

extern "C" int nm_unfinished_mask(void* stream, const int32_t* finished, float* out, int64_t ld_out, int64_t n) {
    NM_REQUIRE(finished && out && n >= 0 && ld_out >= 1, "nm_unfinished_mask: bad args");
    if (n == 0) return NM_OK;
    hipLaunchKernelGGL(unfinished_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nm_stream(stream),
                       finished, out, (long)ld_out, (long)n);
    NM_LAUNCH_CHECK("nm_unfinished_mask");
}

// out[b,:] = sum_t x[b,t,:]   (TransformerEncoder.output, encoders/transformer.py:170-172)
__global__ void time_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int D, long BD) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < BD; i += (long)gridDim.x * blockDim.x) {
        const long b = i / D;
        const int c = (int)(i - b * D);
        float acc = 0.0f;
        for (int t = 0; t < T; ++t) acc += x[(b * T + t) * D + c];
        out[i] = acc;
    }
}

extern "C" int nm_time_sum(void* stream, const float* x, float* out, int64_t B, int64_t T, int64_t D) {
    NM_REQUIRE(x && out && B >= 0 && T >= 0 && D > 0, "nm_time_sum: bad args");
    const long bd = B * D;
    if (bd == 0) return NM_OK;
    const int blocks = (int)((bd + 255) / 256 < 4096 ? (bd + 255) / 256 : 4096);
    hipLaunchKernelGGL(time_sum_kernel, dim3(blocks), dim3(256), 0, nm_stream(stream), x, out, (int)T, (int)D, bd);
    NM_LAUNCH_CHECK("nm_time_sum");
}

// dx[b,t,:] += dy[b,:]  (gradient of the time sum)
__global__ void time_bcast_add_kernel(const float* __restrict__ dy, float* __restrict__ dx, int T, int D, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const long b = i / ((long)T * D);
        dx[i] += dy[b * D + c];
    }
}

extern "C" int nm_time_bcast_add(void* stream, const float* dy, float* dx, int64_t B, int64_t T, int64_t D) {
    NM_REQUIRE(dy && dx && B >= 0 && T >= 0 && D > 0, "nm_time_bcast_add: bad args");
    const long total = B * T * D;
    if (total == 0) return NM_OK;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(time_bcast_add_kernel, dim3(blocks), dim3(256), 0, nm_stream(stream), dy, dx, (int)T, (int)D,
                       total);
    NM_LAUNCH_CHECK("nm_time_bcast_add");
}
