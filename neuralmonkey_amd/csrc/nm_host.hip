// libnmhip: host-side utilities of the C ABI (no device code).
//
//   nm_create / nm_destroy / nm_ctx_bind   library contexts: switches, event pool of the live timer
//   nm_prof_enable / nm_prof_attn_step     the live attention-step timer of a context
//   nm_crc32c   CRC-32C (Castagnoli) of a host buffer -- the checksum TensorFlow's tensor-bundle
//               checkpoints carry per tensor and per index block (checkpoint import / export,
//               neuralmonkey/tf_manager.py:274-288 -> tf.train.Saver).  Slicing-by-8, ~1 GB/s.
#include "nm_common.h"

#include <mutex>
#include <stdlib.h>

#define NM_CTX_MAGIC 0x4e4d4358u      // "NMCX"

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static void switches_from_env(NmSwitches* sw) {
    sw->attn_maxrows = env_int("NM_ATTN_MAXROWS", 12);
    if (sw->attn_maxrows < 1 || sw->attn_maxrows > 16) sw->attn_maxrows = 12;
    sw->attn_nomerge = getenv("NM_ATTN_NOMERGE") != nullptr;
    sw->attn_nofast = getenv("NM_ATTN_NOFAST") != nullptr;
    sw->attn_whole = getenv("NM_ATTN_WHOLE") ? (env_int("NM_ATTN_WHOLE", 0) != 0 ? 1 : 0) : -1;
    sw->aeb_wide_off = getenv("NM_AEB_WIDE") && env_int("NM_AEB_WIDE", 1) == 0;
    sw->gemm_no16 = getenv("NM_GEMM_NO16") != nullptr;
    sw->gemm_swz = env_int("NM_GEMM_SWZ", 1);
    sw->gemm_nostore = getenv("NM_GEMM_NOSTORE") != nullptr;
    sw->gemm_sk = env_int("NM_GEMM_SK", 0);
    sw->gemm_cfg = env_int("NM_GEMM_CFG", 1);
    sw->gemm_chains = env_int("NM_GEMM_CHAINS", 1);
    sw->gemm_cfg64 = env_int("NM_GEMM_CFG64", 0);
    sw->gemm_bg_wgs = env_int("NM_GEMM_BG_WGS", 1);
    sw->gemm_bg_cfg = env_int("NM_GEMM_BG_CFG", 1);
    sw->background = 0;
    sw->step_prio = env_int("NM_STEP_PRIO", 1);
    sw->stats_cfg = env_int("NM_STATS_CFG", 3);
    sw->stats_ablate = getenv("NM_STATS_ABLATE") != nullptr;
    sw->beam_ns = env_int("NM_BEAM_NS", 0);
    sw->sdp_mfma = !(getenv("NM_SDP_MFMA") && env_int("NM_SDP_MFMA", 1) == 0);
    sw->medium_m = env_int("NM_STEP_MEDIUM", 1);
    sw->sdp_decode = !(getenv("NM_SDP_DECODE") && env_int("NM_SDP_DECODE", 1) == 0);
}

static NmCtx* ctx_new(int device) {
    NmCtx* c = new NmCtx();
    c->magic = NM_CTX_MAGIC;
    c->device = device;
    switches_from_env(&c->sw);
    c->prof_on = false;
    c->prof_used = 0;
    return c;
}

static thread_local NmCtx* t_bound = nullptr;
static NmCtx* g_default = nullptr;
static std::once_flag g_default_once;

NmCtx* nm_cur() {
    if (t_bound) return t_bound;
    std::call_once(g_default_once, [] {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;      // no GPU: device 0 is as good as any
        g_default = ctx_new(dev);
    });
    return g_default;
}

std::pair<hipEvent_t, hipEvent_t>* nm_prof_next_pair(NmCtx* c) {
    if (c->prof_used == c->prof_pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
        c->prof_pool.emplace_back(a, b);
    }
    return &c->prof_pool[c->prof_used++];
}

static NmCtx* ctx_of(void* handle) {           // null handle = the calling thread's context
    if (!handle) return nm_cur();
    NmCtx* c = static_cast<NmCtx*>(handle);
    return c->magic == NM_CTX_MAGIC ? c : nullptr;
}

// A context for `device` (>= 0; -1 = the current HIP device).  Reads the NM_* switches from the environment now.
extern "C" int nm_create(int device, void** out_ctx) {
    NM_REQUIRE(out_ctx, "nm_create: null output pointer");
    if (device < 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        device = dev;
    }
    *out_ctx = ctx_new(device);
    return NM_OK;
}

// Releases the context's events.  A context that is still bound to the calling thread is unbound first; the
// process default context cannot be destroyed.
extern "C" int nm_destroy(void* ctx) {
    NM_REQUIRE(ctx, "nm_destroy: null context");
    NmCtx* c = static_cast<NmCtx*>(ctx);
    NM_REQUIRE(c->magic == NM_CTX_MAGIC, "nm_destroy: not a context (or destroyed twice)");
    NM_REQUIRE(c != g_default, "nm_destroy: the default context belongs to the library");
    if (t_bound == c) t_bound = nullptr;
    for (auto& pr : c->prof_pool) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    c->magic = 0;
    delete c;
    return NM_OK;
}

// Binds `ctx` to the calling thread (null: back to the default context): every entry point called from this
// thread afterwards takes its switches and its timer from it.
extern "C" int nm_ctx_bind(void* ctx) {
    if (!ctx) {
        t_bound = nullptr;
        return NM_OK;
    }
    NmCtx* c = static_cast<NmCtx*>(ctx);
    NM_REQUIRE(c->magic == NM_CTX_MAGIC, "nm_ctx_bind: not a context");
    t_bound = c;
    return NM_OK;
}

extern "C" void* nm_ctx_current(void) { return nm_cur(); }

// Background mode of a context: the launches that follow are meant to run BESIDE a latency-bound loop of another
// stream (the encoder of the next batch under the decoding loop of the running one).  GEMMs chosen automatically
// and the recurrent time-loop kernels then cap their residency at NM_GEMM_BG_WGS workgroups per CU (nm_gemm_f32,
// algo 4) and do not raise their wave priority.  Captured launches keep the mode they were captured in.
extern "C" int nm_ctx_set_background(void* ctx, int on) {
    NmCtx* c = ctx_of(ctx);
    NM_REQUIRE(c, "nm_ctx_set_background: not a context");
    c->sw.background = on ? 1 : 0;
    return NM_OK;
}

extern "C" int nm_ctx_device(void* ctx) {
    NmCtx* c = ctx_of(ctx);
    return c ? c->device : -1;
}

// Value of a switch as the context read it when it was created (tests, tools): the NM_* name without the prefix,
// lower case, e.g. "attn_whole".
extern "C" int nm_ctx_switch(void* ctx, const char* name, int* value) {
    NmCtx* c = ctx_of(ctx);
    NM_REQUIRE(c && name && value, "nm_ctx_switch: bad arguments");
    const NmSwitches& s = c->sw;
    struct { const char* n; int v; } tab[] = {
        {"attn_maxrows", s.attn_maxrows}, {"attn_nomerge", s.attn_nomerge}, {"attn_nofast", s.attn_nofast},
        {"attn_whole", s.attn_whole}, {"aeb_wide_off", s.aeb_wide_off}, {"gemm_no16", s.gemm_no16},
        {"gemm_swz", s.gemm_swz}, {"gemm_nostore", s.gemm_nostore}, {"gemm_sk", s.gemm_sk},
        {"gemm_cfg", s.gemm_cfg}, {"gemm_chains", s.gemm_chains}, {"gemm_cfg64", s.gemm_cfg64}, {"gemm_bg_wgs", s.gemm_bg_wgs}, {"step_prio", s.step_prio},
        {"stats_cfg", s.stats_cfg}, {"stats_ablate", s.stats_ablate},
        {"beam_ns", s.beam_ns}, {"sdp_mfma", s.sdp_mfma}, {"medium_m", s.medium_m},
        {"sdp_decode", s.sdp_decode}};
    for (auto& t : tab)
        if (strcmp(t.n, name) == 0) {
            *value = t.v;
            return NM_OK;
        }
    NM_FAIL(NM_ERR_ARG, "nm_ctx_switch: unknown switch '%s'", name);
}

// ---- live timing of the attention step (everything nm_attn_fwd launches) with HIP events on the launch stream
// (bench.py's `roofline.achieved`); off by default, zero cost when off.  Per context.
extern "C" int nm_prof_enable(void* ctx, int on) {
    NmCtx* c = ctx_of(ctx);
    NM_REQUIRE(c, "nm_prof_enable: not a context");
    c->prof_on = on != 0;
    if (c->prof_on) c->prof_used = 0;
    return NM_OK;
}

// Sum / count of the recorded attention steps; resets the recorder.
extern "C" int nm_prof_attn_step(void* ctx, double* total_ms, int64_t* count) {
    NmCtx* c = ctx_of(ctx);
    NM_REQUIRE(c && total_ms && count, "nm_prof_attn_step: bad arguments");
    double tot = 0.0;
    for (size_t i = 0; i < c->prof_used; ++i) {
        float ms = 0.0f;
        if (hipEventSynchronize(c->prof_pool[i].second) != hipSuccess ||
            hipEventElapsedTime(&ms, c->prof_pool[i].first, c->prof_pool[i].second) != hipSuccess)
            NM_FAIL(NM_ERR_HIP, "nm_prof_attn_step: event query failed");
        tot += ms;
    }
    *total_ms = tot;
    *count = (int64_t)c->prof_used;
    c->prof_used = 0;
    return NM_OK;
}

static uint32_t g_crc_table[8][256];
static bool g_crc_ready = false;

static void crc32c_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
        g_crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t)
            g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xff];
    g_crc_ready = true;
}

// crc = nm_crc32c(crc_of_previous_bytes, data, n); start with 0
extern "C" uint32_t nm_crc32c(uint32_t crc, const void* data, int64_t n) {
    if (!g_crc_ready) crc32c_init();
    const uint8_t* p = static_cast<const uint8_t*>(data);
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n > 0 && (reinterpret_cast<uintptr_t>(p) & 7)) {
        c = g_crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
        --n;
    }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;
        c = g_crc_table[7][w & 0xff] ^ g_crc_table[6][(w >> 8) & 0xff] ^ g_crc_table[5][(w >> 16) & 0xff] ^
            g_crc_table[4][(w >> 24) & 0xff] ^ g_crc_table[3][(w >> 32) & 0xff] ^ g_crc_table[2][(w >> 40) & 0xff] ^
            g_crc_table[1][(w >> 48) & 0xff] ^ g_crc_table[0][(w >> 56) & 0xff];
        p += 8;
        n -= 8;
    }
    while (n-- > 0) c = g_crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
