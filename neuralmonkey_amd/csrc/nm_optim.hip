// Trainer arithmetic over the flat parameter / gradient buffers
// (trainers/generic_trainer.py:84-195):
//   L1 = sum |v|, L2 = sum v^2 over non-bias variables            (:84-105)
//   grad += l1_weight*sign(v) + 2*l2_weight*v                     (d/dv of :118-134)
//   per-tensor tf.clip_by_norm: g * c / max(||g||, c)              (:179-186)
//   Adam (tf.train.AdamOptimizer, :55-57): lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
//   Adadelta (tf.train.AdadeltaOptimizer, tests/bpe.ini:102-108): TF 1.12 ApplyAdadelta
// All variables live in one flat buffer (variables.py); a host-built chunk
// table maps fixed-size chunks to variables ("segments") so that three launches
// cover every tensor and all reductions have a fixed order (deterministic).
#include "nm_common.h"

struct OptChunks {
    const long* chunk_start;   // [nchunk] flat offset
    const int* chunk_len;      // [nchunk]
    const int* chunk_seg;      // [nchunk]
    const int* seg_first;      // [nseg] first chunk
    const int* seg_count;      // [nseg] chunks
    const int* seg_flags;      // [nseg] bit0 regularizable, bit1 trainable
    int nchunk, nseg;
};

// pass 1: regulariser terms + squared gradient norm partials per chunk
__global__ __launch_bounds__(256) void opt_reg_sumsq_kernel(OptChunks t, const float* __restrict__ theta,
                                                            float* __restrict__ grad, float l1w, float l2w,
                                                            float* __restrict__ partial, int c0,
                                                            const int* __restrict__ list) {
    __shared__ float sh[3][4];
    const int c = list ? list[blockIdx.x] : c0 + (int)blockIdx.x;     // (list: the chunks a rank owns, in any order)
    const long base = t.chunk_start[c];
    const int len = t.chunk_len[c];
    const int flags = t.seg_flags[t.chunk_seg[c]];
    const bool reg = flags & 1;
    float gs = 0.0f, a1 = 0.0f, a2 = 0.0f;
    auto one = [&](float th, float g) -> float {
        if (reg) {
            a1 += fabsf(th);
            a2 += th * th;
            const float sg = (th > 0.0f) ? 1.0f : ((th < 0.0f) ? -1.0f : 0.0f);
            g += l1w * sg + 2.0f * l2w * th;
        }
        gs += g * g;
        return g;
    };
    int done = 0;
    if ((base & 3) == 0) {                 // 16-byte rows: the pass is a pure stream over theta and grad
        const int len4 = len >> 2;
        const float4* th4 = reinterpret_cast<const float4*>(theta + base);
        float4* g4 = reinterpret_cast<float4*>(grad + base);
        for (int i = threadIdx.x; i < len4; i += 256) {
            const float4 th = th4[i];
            float4 g = g4[i];
            g.x = one(th.x, g.x); g.y = one(th.y, g.y); g.z = one(th.z, g.z); g.w = one(th.w, g.w);
            if (reg) g4[i] = g;
        }
        done = len4 << 2;
    }
    for (int i = done + threadIdx.x; i < len; i += 256) {
        const float g = one(theta[base + i], grad[base + i]);
        if (reg) grad[base + i] = g;
    }
    gs = nm_wave_sum(gs); a1 = nm_wave_sum(a1); a2 = nm_wave_sum(a2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = gs; sh[1][w] = a1; sh[2][w] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[c * 3 + 0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        partial[c * 3 + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        partial[c * 3 + 2] = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
    }
}

// pass 2: per-segment gradient norms; global L1 / L2 (fixed order)
__global__ void opt_seg_reduce_kernel(OptChunks t, const float* __restrict__ partial,
                                      float* __restrict__ seg_norm2, float* __restrict__ l1l2) {
    // one thread per segment (fixed chunk order), then thread 0 adds the segments in order
    extern __shared__ float seg_l[];          // [2][nseg]
    for (int s = threadIdx.x; s < t.nseg; s += blockDim.x) {
        float gs = 0.0f, a1 = 0.0f, a2 = 0.0f;
        for (int c = t.seg_first[s]; c < t.seg_first[s] + t.seg_count[s]; ++c) {
            gs += partial[c * 3 + 0];
            a1 += partial[c * 3 + 1];
            a2 += partial[c * 3 + 2];
        }
        seg_norm2[s] = gs;
        seg_l[s] = a1;
        seg_l[t.nseg + s] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l1 = 0.0f, l2 = 0.0f;
        for (int s = 0; s < t.nseg; ++s) { l1 += seg_l[s]; l2 += seg_l[t.nseg + s]; }
        l1l2[0] = l1;
        l1l2[1] = l2;
    }
}

// pass 3: clip per tensor + Adam
__global__ __launch_bounds__(256) void opt_adam_kernel(OptChunks t, float* __restrict__ theta,
                                                       const float* __restrict__ grad,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       const float* __restrict__ seg_norm2, float clip,
                                                       float lr_t, float b1, float b2, float eps, int c0,
                                                       const int* __restrict__ skip, const int* __restrict__ list) {
    if (skip && *skip != 0) return;          // the step's gradient is garbage (a time loop gave up): nothing is applied
    const int c = list ? list[blockIdx.x] : c0 + (int)blockIdx.x;
    const int seg = t.chunk_seg[c];
    if (!(t.seg_flags[seg] & 2)) return;
    const long base = t.chunk_start[c];
    const int len = t.chunk_len[c];
    float scale = 1.0f;
    if (clip > 0.0f) scale = clip / fmaxf(sqrtf(seg_norm2[seg]), clip);
    for (int i = threadIdx.x; i < len; i += 256) {
        const float g = grad[base + i] * scale;
        const float mm = b1 * m[base + i] + (1.0f - b1) * g;
        const float vv = b2 * v[base + i] + (1.0f - b2) * g * g;
        m[base + i] = mm;
        v[base + i] = vv;
        theta[base + i] -= lr_t * mm / (sqrtf(vv) + eps);
    }
}

// pass 3, Adadelta: accum <- rho*accum + (1-rho)*g^2; update = sqrt(accum_update + eps) * rsqrt(accum + eps) * g;
// var -= lr*update; accum_update <- rho*accum_update + (1-rho)*update^2   (the order TF's kernel evaluates them in)
__global__ __launch_bounds__(256) void opt_adadelta_kernel(OptChunks t, float* __restrict__ theta,
                                                           const float* __restrict__ grad,
                                                           float* __restrict__ accum, float* __restrict__ accum_update,
                                                           const float* __restrict__ seg_norm2, float clip,
                                                           float lr, float rho, float eps, int c0,
                                                           const int* __restrict__ skip, const int* __restrict__ list) {
    if (skip && *skip != 0) return;
    const int c = list ? list[blockIdx.x] : c0 + (int)blockIdx.x;
    const int seg = t.chunk_seg[c];
    if (!(t.seg_flags[seg] & 2)) return;
    const long base = t.chunk_start[c];
    const int len = t.chunk_len[c];
    float scale = 1.0f;
    if (clip > 0.0f) scale = clip / fmaxf(sqrtf(seg_norm2[seg]), clip);
    for (int i = threadIdx.x; i < len; i += 256) {
        const float g = grad[base + i] * scale;
        const float a = rho * accum[base + i] + (1.0f - rho) * g * g;
        const float au = accum_update[base + i];
        const float upd = sqrtf(au + eps) * (1.0f / sqrtf(a + eps)) * g;
        accum[base + i] = a;
        theta[base + i] -= lr * upd;
        accum_update[base + i] = rho * au + (1.0f - rho) * upd * upd;
    }
}

static OptChunks make_chunks(const int64_t* chunk_start, const int32_t* chunk_len, const int32_t* chunk_seg,
                             const int32_t* seg_first, const int32_t* seg_count, const int32_t* seg_flags,
                             int64_t nchunk, int64_t nseg) {
    OptChunks t;
    t.chunk_start = reinterpret_cast<const long*>(chunk_start);
    t.chunk_len = chunk_len; t.chunk_seg = chunk_seg; t.seg_first = seg_first; t.seg_count = seg_count;
    t.seg_flags = seg_flags; t.nchunk = (int)nchunk; t.nseg = (int)nseg;
    return t;
}

// workspace: partial[nchunk*3] + seg_norm2[nseg]  (floats)
extern "C" int64_t nm_optim_workspace_bytes(int64_t nchunk, int64_t nseg) { return (nchunk * 3 + nseg) * 4; }

// pass 1 over the chunks [chunk_begin, chunk_end): regulariser terms into the gradient, the chunks' partial sums into
// the workspace.  A rank that owns a slice of the flat buffers (sharded optimizer, distributed.py) runs its own chunks
// only; the partial vector (3 floats per chunk, zero where nobody wrote) is then summed over ranks -- every entry has
// exactly one non-zero contributor, so the sum is exact and nm_optim_segments sees the same numbers on every rank as
// one process would.
extern "C" int nm_optim_partials(void* stream, const float* theta, float* grad, const int64_t* chunk_start,
                                 const int32_t* chunk_len, const int32_t* chunk_seg, const int32_t* seg_first,
                                 const int32_t* seg_count, const int32_t* seg_flags, int64_t nchunk, int64_t nseg,
                                 float l1_weight, float l2_weight, int64_t chunk_begin, int64_t chunk_end,
                                 void* workspace, int64_t workspace_bytes) {
    NM_REQUIRE(theta && grad && chunk_start && chunk_len && chunk_seg && seg_first && seg_count && seg_flags && workspace,
               "nm_optim_partials: null pointer");
    NM_REQUIRE(nchunk > 0 && nseg > 0 && workspace_bytes >= nm_optim_workspace_bytes(nchunk, nseg),
               "nm_optim_partials: bad sizes");
    NM_REQUIRE(chunk_begin >= 0 && chunk_begin <= chunk_end && chunk_end <= nchunk,
               "nm_optim_partials: bad chunk range [%ld, %ld) of %ld", (long)chunk_begin, (long)chunk_end, (long)nchunk);
    if (chunk_begin == chunk_end) return NM_OK;
    OptChunks t = make_chunks(chunk_start, chunk_len, chunk_seg, seg_first, seg_count, seg_flags, nchunk, nseg);
    hipLaunchKernelGGL(opt_reg_sumsq_kernel, dim3((unsigned)(chunk_end - chunk_begin)), dim3(256), 0, nm_stream(stream),
                       t, theta, grad, l1_weight, l2_weight, reinterpret_cast<float*>(workspace), (int)chunk_begin,
                       (const int*)nullptr);
    NM_LAUNCH_CHECK("nm_optim_partials");
}

// The same over a LIST of chunks (a device array of ``count`` chunk indices): everything a rank of the sharded optimizer
// owns -- one slice per bucket plus the shared tails -- in ONE launch.  Range by range (14 launches of ~60 chunks each
// on the headline model) a launch is as long as its slowest workgroup and the chip is a quarter full: 0.98 instead of
// 0.14 ms for this pass, 1.20 instead of 0.35 ms for the update (profiles/r06_dp_force_kernel_stats_sharded1.csv).
extern "C" int nm_optim_partials_list(void* stream, const float* theta, float* grad, const int64_t* chunk_start,
                                      const int32_t* chunk_len, const int32_t* chunk_seg, const int32_t* seg_first,
                                      const int32_t* seg_count, const int32_t* seg_flags, int64_t nchunk, int64_t nseg,
                                      float l1_weight, float l2_weight, const int32_t* chunk_list, int64_t count,
                                      void* workspace, int64_t workspace_bytes) {
    NM_REQUIRE(theta && grad && chunk_start && chunk_len && chunk_seg && seg_first && seg_count && seg_flags && workspace,
               "nm_optim_partials_list: null pointer");
    NM_REQUIRE(nchunk > 0 && nseg > 0 && workspace_bytes >= nm_optim_workspace_bytes(nchunk, nseg),
               "nm_optim_partials_list: bad sizes");
    NM_REQUIRE(count >= 0 && count <= nchunk && (count == 0 || chunk_list), "nm_optim_partials_list: bad chunk list (%ld of %ld)",
               (long)count, (long)nchunk);
    if (count == 0) return NM_OK;
    OptChunks t = make_chunks(chunk_start, chunk_len, chunk_seg, seg_first, seg_count, seg_flags, nchunk, nseg);
    hipLaunchKernelGGL(opt_reg_sumsq_kernel, dim3((unsigned)count), dim3(256), 0, nm_stream(stream), t, theta, grad,
                       l1_weight, l2_weight, reinterpret_cast<float*>(workspace), 0, chunk_list);
    NM_LAUNCH_CHECK("nm_optim_partials_list");
}

// pass 2: per-variable squared gradient norms and the global L1 / L2 terms from the partial vector, in a fixed order
extern "C" int nm_optim_segments(void* stream, const int64_t* chunk_start, const int32_t* chunk_len,
                                 const int32_t* chunk_seg, const int32_t* seg_first, const int32_t* seg_count,
                                 const int32_t* seg_flags, int64_t nchunk, int64_t nseg, float* l1l2_out,
                                 void* workspace, int64_t workspace_bytes) {
    NM_REQUIRE(chunk_start && chunk_len && chunk_seg && seg_first && seg_count && seg_flags && l1l2_out && workspace,
               "nm_optim_segments: null pointer");
    NM_REQUIRE(nchunk > 0 && nseg > 0 && nseg <= 8192 && workspace_bytes >= nm_optim_workspace_bytes(nchunk, nseg),
               "nm_optim_segments: bad sizes");
    OptChunks t = make_chunks(chunk_start, chunk_len, chunk_seg, seg_first, seg_count, seg_flags, nchunk, nseg);
    float* partial = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(opt_seg_reduce_kernel, dim3(1), dim3(256), (size_t)nseg * 2 * sizeof(float), nm_stream(stream), t,
                       partial, partial + nchunk * 3, l1l2_out);
    NM_LAUNCH_CHECK("nm_optim_segments");
}

// pass 3 over the chunks [chunk_begin, chunk_end): per-tensor clip + the optimizer's update.  kind 0: Adam
// (p = lr_t, beta1, beta2, epsilon), kind 1: Adadelta (p = lr, rho, epsilon, -).  ``skip_word`` (may be null): a device
// word that, when not zero, turns the launch into a no-op -- the session's error word (a GRU time loop gave up: the
// gradient is garbage and the step will be run again, runtime.Session.recover_training).
extern "C" int nm_optim_apply(void* stream, int32_t kind, float* theta, const float* grad, float* slot0, float* slot1,
                              const int64_t* chunk_start, const int32_t* chunk_len, const int32_t* chunk_seg,
                              const int32_t* seg_first, const int32_t* seg_count, const int32_t* seg_flags,
                              int64_t nchunk, int64_t nseg, float clip_norm, float p0, float p1, float p2, float p3,
                              int64_t chunk_begin, int64_t chunk_end, const int32_t* skip_word, void* workspace,
                              int64_t workspace_bytes) {
    NM_REQUIRE(theta && grad && slot0 && slot1 && workspace, "nm_optim_apply: null pointer");
    NM_REQUIRE(kind == 0 || kind == 1, "nm_optim_apply: kind %d (0 Adam, 1 Adadelta)", (int)kind);
    NM_REQUIRE(nchunk > 0 && nseg > 0 && workspace_bytes >= nm_optim_workspace_bytes(nchunk, nseg),
               "nm_optim_apply: bad sizes");
    NM_REQUIRE(chunk_begin >= 0 && chunk_begin <= chunk_end && chunk_end <= nchunk,
               "nm_optim_apply: bad chunk range [%ld, %ld) of %ld", (long)chunk_begin, (long)chunk_end, (long)nchunk);
    if (chunk_begin == chunk_end) return NM_OK;
    OptChunks t = make_chunks(chunk_start, chunk_len, chunk_seg, seg_first, seg_count, seg_flags, nchunk, nseg);
    const float* seg_norm2 = reinterpret_cast<const float*>(workspace) + nchunk * 3;
    const unsigned grid = (unsigned)(chunk_end - chunk_begin);
    if (kind == 0)
        hipLaunchKernelGGL(opt_adam_kernel, dim3(grid), dim3(256), 0, nm_stream(stream), t, theta, grad, slot0, slot1,
                           seg_norm2, clip_norm, p0, p1, p2, p3, (int)chunk_begin, skip_word, (const int*)nullptr);
    else
        hipLaunchKernelGGL(opt_adadelta_kernel, dim3(grid), dim3(256), 0, nm_stream(stream), t, theta, grad, slot0, slot1,
                           seg_norm2, clip_norm, p0, p1, p2, (int)chunk_begin, skip_word, (const int*)nullptr);
    NM_LAUNCH_CHECK("nm_optim_apply");
}

// ... over a list of chunks (see nm_optim_partials_list)
extern "C" int nm_optim_apply_list(void* stream, int32_t kind, float* theta, const float* grad, float* slot0, float* slot1,
                                   const int64_t* chunk_start, const int32_t* chunk_len, const int32_t* chunk_seg,
                                   const int32_t* seg_first, const int32_t* seg_count, const int32_t* seg_flags,
                                   int64_t nchunk, int64_t nseg, float clip_norm, float p0, float p1, float p2, float p3,
                                   const int32_t* chunk_list, int64_t count, const int32_t* skip_word, void* workspace,
                                   int64_t workspace_bytes) {
    NM_REQUIRE(theta && grad && slot0 && slot1 && workspace, "nm_optim_apply_list: null pointer");
    NM_REQUIRE(kind == 0 || kind == 1, "nm_optim_apply_list: kind %d (0 Adam, 1 Adadelta)", (int)kind);
    NM_REQUIRE(nchunk > 0 && nseg > 0 && workspace_bytes >= nm_optim_workspace_bytes(nchunk, nseg),
               "nm_optim_apply_list: bad sizes");
    NM_REQUIRE(count >= 0 && count <= nchunk && (count == 0 || chunk_list), "nm_optim_apply_list: bad chunk list (%ld of %ld)",
               (long)count, (long)nchunk);
    if (count == 0) return NM_OK;
    OptChunks t = make_chunks(chunk_start, chunk_len, chunk_seg, seg_first, seg_count, seg_flags, nchunk, nseg);
    const float* seg_norm2 = reinterpret_cast<const float*>(workspace) + nchunk * 3;
    if (kind == 0)
        hipLaunchKernelGGL(opt_adam_kernel, dim3((unsigned)count), dim3(256), 0, nm_stream(stream), t, theta, grad, slot0,
                           slot1, seg_norm2, clip_norm, p0, p1, p2, p3, 0, skip_word, chunk_list);
    else
        hipLaunchKernelGGL(opt_adadelta_kernel, dim3((unsigned)count), dim3(256), 0, nm_stream(stream), t, theta, grad, slot0,
                           slot1, seg_norm2, clip_norm, p0, p1, p2, 0, skip_word, chunk_list);
    NM_LAUNCH_CHECK("nm_optim_apply_list");
}

// x[0..n) = 0 when *word != 0 (the session's error word): a gradient that a given-up time loop left behind must not
// reach an accumulation buffer or a collective as NaNs
__global__ __launch_bounds__(256) void zero_if_kernel(const int* __restrict__ word, float* __restrict__ x, long n) {
    if (*word == 0) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] = 0.0f;
}

extern "C" int nm_zero_if(void* stream, const int32_t* word, float* x, int64_t n) {
    NM_REQUIRE(word && x && n >= 0, "nm_zero_if: bad arguments");
    if (n == 0) return NM_OK;
    const unsigned grid = (unsigned)(n / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(zero_if_kernel, dim3(grid), dim3(256), 0, nm_stream(stream), word, x, (long)n);
    NM_LAUNCH_CHECK("nm_zero_if");
}

// Fills of whole buffers with a 4-byte pattern (float and int32 buffers alike) and device-to-device copies: the
// engine's steps carried ~25 torch fill / copy kernels (tensor.zero_(), fill_(), copy_()) that these replace.  The fill is
// a kernel of this library (16-byte stores; a kernel node inside a captured graph -- hipMemsetD32Async nodes were tried
// and a decoding graph with them did not come back on this stack), the copy is the runtime's (a memcpy node).
__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ x, long head, long body4, long tail,
                                                       uint32_t v) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g < head) x[g] = v;
    if (g < tail) x[head + 4 * body4 + g] = v;
    uint4* b = reinterpret_cast<uint4*>(x + head);
    const uint4 v4 = make_uint4(v, v, v, v);
    for (long i = g; i < body4; i += (long)gridDim.x * 256) b[i] = v4;
}

extern "C" int nm_fill_u32(void* stream, void* x, int64_t count, uint32_t pattern) {
    NM_REQUIRE(x && count >= 0 && (reinterpret_cast<uintptr_t>(x) & 3) == 0, "nm_fill_u32: bad arguments");
    if (count == 0) return NM_OK;
    long head = (long)(((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / 4);
    if (head > count) head = count;
    const long body4 = (count - head) / 4, tail = count - head - 4 * body4;
    long blocks = (body4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, nm_stream(stream),
                       reinterpret_cast<uint32_t*>(x), head, body4, tail, pattern);
    NM_LAUNCH_CHECK("nm_fill_u32");
}

extern "C" int nm_copy_d2d(void* stream, void* dst, const void* src, int64_t bytes) {
    NM_REQUIRE(dst && src && bytes >= 0, "nm_copy_d2d: bad arguments");
    if (bytes == 0 || dst == src) return NM_OK;
    if (hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, nm_stream(stream)) != hipSuccess)
        NM_FAIL(NM_ERR_HIP, "nm_copy_d2d: hipMemcpyAsync failed: %s", hipGetErrorString(hipGetLastError()));
    return NM_OK;
}

extern "C" int nm_optim_regularize_norms(void* stream, const float* theta, float* grad,
                                         const int64_t* chunk_start, const int32_t* chunk_len,
                                         const int32_t* chunk_seg, const int32_t* seg_first,
                                         const int32_t* seg_count, const int32_t* seg_flags, int64_t nchunk,
                                         int64_t nseg, float l1_weight, float l2_weight, float* l1l2_out,
                                         void* workspace, int64_t workspace_bytes) {
    NM_REQUIRE(theta && grad && chunk_start && chunk_len && chunk_seg && seg_first && seg_count &&
                   seg_flags && l1l2_out && workspace,
               "nm_optim_regularize_norms: null pointer");
    NM_REQUIRE(nchunk > 0 && nseg > 0 && workspace_bytes >= nm_optim_workspace_bytes(nchunk, nseg),
               "nm_optim_regularize_norms: bad sizes");
    NM_REQUIRE(nseg <= 8192, "nm_optim_regularize_norms: too many variables");
    int rc = nm_optim_partials(stream, theta, grad, chunk_start, chunk_len, chunk_seg, seg_first, seg_count, seg_flags,
                               nchunk, nseg, l1_weight, l2_weight, 0, nchunk, workspace, workspace_bytes);
    if (rc != NM_OK) return rc;
    return nm_optim_segments(stream, chunk_start, chunk_len, chunk_seg, seg_first, seg_count, seg_flags, nchunk, nseg,
                             l1l2_out, workspace, workspace_bytes);
}

extern "C" int nm_optim_clip_adam(void* stream, float* theta, const float* grad, float* m, float* v,
                                  const int64_t* chunk_start, const int32_t* chunk_len,
                                  const int32_t* chunk_seg, const int32_t* seg_first, const int32_t* seg_count,
                                  const int32_t* seg_flags, int64_t nchunk, int64_t nseg, float clip_norm,
                                  float lr_t, float beta1, float beta2, float epsilon, void* workspace,
                                  int64_t workspace_bytes) {
    NM_REQUIRE(theta && grad && m && v && workspace, "nm_optim_clip_adam: null pointer");
    NM_REQUIRE(nchunk > 0 && nseg > 0 && workspace_bytes >= nm_optim_workspace_bytes(nchunk, nseg),
               "nm_optim_clip_adam: bad sizes");
    return nm_optim_apply(stream, 0, theta, grad, m, v, chunk_start, chunk_len, chunk_seg, seg_first, seg_count,
                          seg_flags, nchunk, nseg, clip_norm, lr_t, beta1, beta2, epsilon, 0, nchunk, nullptr,
                          workspace, workspace_bytes);
}

extern "C" int nm_optim_clip_adadelta(void* stream, float* theta, const float* grad, float* accum, float* accum_update,
                                      const int64_t* chunk_start, const int32_t* chunk_len,
                                      const int32_t* chunk_seg, const int32_t* seg_first, const int32_t* seg_count,
                                      const int32_t* seg_flags, int64_t nchunk, int64_t nseg, float clip_norm,
                                      float lr, float rho, float epsilon, void* workspace,
                                      int64_t workspace_bytes) {
    NM_REQUIRE(theta && grad && accum && accum_update && workspace, "nm_optim_clip_adadelta: null pointer");
    NM_REQUIRE(nchunk > 0 && nseg > 0 && workspace_bytes >= nm_optim_workspace_bytes(nchunk, nseg),
               "nm_optim_clip_adadelta: bad sizes");
    return nm_optim_apply(stream, 1, theta, grad, accum, accum_update, chunk_start, chunk_len, chunk_seg, seg_first,
                          seg_count, seg_flags, nchunk, nseg, clip_norm, lr, rho, epsilon, 0.0f, 0, nchunk, nullptr,
                          workspace, workspace_bytes);
}
