// The time loops of a GRU layer as ONE launch each: weight-stationary workgroup clusters with point-to-point
// hand-offs instead of two dependent launches per step (nm_gru_gemm).
//
// Arithmetic: TF GRUCell under dynamic_rnn (nn/ortho_gru_cell.py:44-53, encoders/recurrent.py:86-102) and its
// back-propagation -- the products and epilogues of the per-step launches (nm_gemm.hip: skinny16_tile +
// gru_epi_apply, modes 1-4): K split over H/64 waves, a 16x16x4 fp32 MFMA chain per wave, partials added wave
// 0..NW-1.  Eight waves of ~110 registers (not sixteen of ~100) so that a workgroup leaves room on its CU: the loops run
// BESIDE the weight-gradient GEMMs of other streams (one residency-capped 8-wave workgroup per CU, nm_gemm_f32 algo 4),
// and a 16-wave workgroup that needs a whole CU's registers waits until those drain -- measured: loops ten times and
// the GEMMs two times slower than alone (profiles/r05_cluster_loops.md).
//
// Decomposition.  A step is a chain of two products with an elementwise stage after each (forward: gates, then
// candidate + blend; backward: d(r*h), then dh).  Rows are independent, columns are not: the second product needs
// ALL columns of the first stage's result for a row.  So the chip is cut into clusters = (direction, block of
// 16*RT rows) and every workgroup of a cluster owns 16 hidden units: its slices of the recurrent kernels (K x 48
// floats = 96 KB at H = 512) live in REGISTERS for the whole loop (48 per lane), the state of its (row, unit)
// elements lives in the registers of the threads that blend them.  What crosses workgroups per stage is the
// stage's output for the cluster's rows: 16*RT x 16 values per workgroup, published as 8-byte {value, tag} granules
// with ONE store each and read -- by every workgroup of the cluster, each WAVE only the K-slice it multiplies --
// with agent-scope (sc1: past the reader's L1) loads until the tags carry the stage's epoch: the data is the flag,
// there is no fence, no barrier and no cache invalidate on the path (cdna_hip_programming.md section 6,
// Guideline 16, form R2).
//
// Placement.  The hand-off is correct under ANY placement when the granules are stored write-through (sc1), and
// that is the fallback.  A write-through store drops the line from the writer's L2, though, so every hop then pays
// two round trips to the fabric (~1 us each).  When every workgroup of a cluster runs on ONE XCD, plain stores are
// enough -- the XCD's L2 is the coherence point of its 32 CUs, only the readers' L1 has to be bypassed -- and a hop
// is two L2 round trips.  HIP promises nothing about placement, so the kernels do not assume it: every workgroup
// reads its XCC_ID, takes a ticket on that XCD's counter, waits (once per launch) until all workgroups of the grid
// have done so, and only if every XCD got the tickets its clusters need do roles follow the tickets (cluster =
// XCD * clusters-per-XCD + ticket / (H/16)) and stores stay plain; otherwise roles follow blockIdx and stores are
// write-through.  Workgroups without a role exit at once (the grid is always one workgroup per CU).
//
// Tags count stages within the launch, the header and the granule buffers are zeroed by a memset node in front of
// every launch, spins are bounded by the wall clock: a workgroup that waits 0.2 s raises the error word, everyone
// stops waiting, and the results are garbage (nm_gru_seq_failed reports it).
//
// Why overwriting a granule needs no acknowledgement: a buffer that stage s published is next written by stage
// s + 2 of the same producer, which runs after the producer consumed stage s + 1 of EVERY cluster member, each of
// which published that only after all its waves had consumed stage s (their partial sums meet in LDS behind a
// workgroup barrier before the epilogue publishes).  The one buffer that would be rewritten a single stage later --
// the update-gate half of the backward loop's second operand -- alternates between two copies.
#include <stdlib.h>

#include "nm_gru.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

#define NM_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// workspace header (unsigned words): [0] error, [1] workgroups that took a ticket, [8 + x] tickets of XCD x
#define CLU_HDR_BYTES 256

struct GruClu {
    GruEpi e;                    // pointers of step 0
    int steps, ndir, nrb;        // row blocks of 16*RT rows per direction
    long h_step, ru_step, rh_step, c_step;     // added per step to h_out, ru, rh, c_save (forward) / ru, c (backward)
    const float* wg; long ldg, sg;             // state half of the gates kernel     [ndir][H][2H]
    const float* wc; long ldc, sc;             // state half of the candidate kernel [ndir][H][H]
    unsigned* hdr;
    u64* xa;                     // granules of the first stage's output  [cluster][16 RT][H]
    u64* xb;                     // granules of the second stage's output [cluster][16 RT][H] (backward: [2][..][2H])
    int force_global;            // NM_CLUSTER_PLACEMENT=blockidx: roles by blockIdx + write-through stores even when the tickets would do
    unsigned* sticky;            // the caller's error word: set (never cleared) when this launch gave up; may be null
    int force_fail;              // test hook (nm_gru_seq_force_give_up): this launch raises its error word at once
    long* dbg;                   // timing probe (NM_CLU_DEBUG builds only)
};

// ---- roles -------------------------------------------------------------------------------------------------------
struct CluRole {
    int cl, jb;                  // cluster, block of 16 hidden units
    bool active, local;          // local: the whole cluster sits on this XCD, plain stores publish
};

// workgroups XCD x needs when roles follow the tickets: it hosts clusters x * cpx .. , at most cpx of them
__device__ __forceinline__ int clu_need(int x, int ncl, int cpx, int nj) { return nj * max(0, min(cpx, ncl - x * cpx)); }

__device__ __forceinline__ bool clu_gave_up(gu32* err, long& t0, unsigned& spins) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63) != 0) return false;
    // ~ every 50 us: has anybody raised the error word, or have we waited 0.2 s (100 MHz clock)?
    if (__hip_atomic_load(err, NM_RLX_AGENT) != 0) return true;
    const long now = (long)wall_clock64();
    if (t0 == 0) t0 = now;
    else if (now - t0 > 20000000L) {
        __hip_atomic_store(err, 1u, NM_RLX_AGENT);
        return true;
    }
    return false;
}

__device__ __forceinline__ CluRole clu_roles(unsigned* hdr_, int ncl, int nj, int* sh, int force_global) {
    gu32* hdr = (gu32*)hdr_;
    const int cpx = (ncl + 7) / 8;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const unsigned ticket = __hip_atomic_fetch_add(hdr + 8 + xcc, 1u, NM_RLX_AGENT);
        // the arrival is counted only after the ticket has been taken: a RELEASE increment, and whoever sees all
        // arrivals re-reads the counter with ACQUIRE before it looks at the tickets (once per launch: the two fences
        // cost nothing next to the loop) -- every workgroup then decides local / global from the same ticket counts
        __hip_atomic_fetch_add(hdr + 1, 1u + (ticket >> 31), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long t0 = 0;
        unsigned spins = 0;
        bool lost = false;
        while (__hip_atomic_load(hdr + 1, NM_RLX_AGENT) < gridDim.x)
            if (clu_gave_up(hdr, t0, spins)) { lost = true; break; }
        (void)__hip_atomic_load(hdr + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        bool local = !lost && !force_global;
        for (int x = 0; x < 8; ++x) local &= (int)__hip_atomic_load(hdr + 8 + x, NM_RLX_AGENT) >= clu_need(x, ncl, cpx, nj);
        int cl, jb;
        bool active;
        if (local) {
            const int lc = (int)ticket / nj;
            cl = (int)xcc * cpx + lc;
            jb = (int)ticket % nj;
            active = lc < cpx && cl < ncl;
        } else {
            cl = (int)blockIdx.x % ncl;
            jb = (int)blockIdx.x / ncl;
            active = (int)blockIdx.x < ncl * nj;
        }
        sh[0] = cl; sh[1] = jb; sh[2] = active ? 1 : 0; sh[3] = local ? 1 : 0;
    }
    __syncthreads();
    CluRole r;
    r.cl = sh[0]; r.jb = sh[1]; r.active = sh[2] != 0; r.local = sh[3] != 0;
    __syncthreads();
    return r;
}

// ---- hand-off primitives ---------------------------------------------------------------------------------------
// Granule layout of one cluster's stage output (16*RT rows x K values): what ONE consuming wave reads with ONE load
// instruction is 1 KB of consecutive bytes -- [wave slice w][row tile rt][chunk c][half hh][k-quad kq][row i16]
// units of 16 bytes = two granules {value, tag} (k = 16 (w NCH + c) + 4 kq + 2 hh + slot).  Lane (i16, kq)
// of the MFMA fragment is lane 16 kq + i16, the position of its unit in the 1 KB: the sweep needs no shuffles.
template <int RT, int NCH>
__device__ __forceinline__ long clu_unit(int w, int c, int hh, int rt, int i16, int kq) {
    return (((((long)w * RT + rt) * NCH + c) * 2 + hh) * 4 + kq) * 16 + i16;
}

// value of (row tile rt, row r, k) of the cluster's stage output, tagged
template <int RT, int NCH>
__device__ __forceinline__ void clu_publish(u64* X, bool local, int rt, int r, int k, unsigned tag, float v) {
    const int n = k & 15, kc = k >> 4;
    u64* g = X + clu_unit<RT, NCH>(kc / NCH, kc % NCH, (n & 3) >> 1, rt, r, n >> 2) * 2 + (n & 1);
    const u64 x = ((u64)tag << 32) | (u64)__float_as_uint(v);
    if (local) __hip_atomic_store((gu64*)g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // stays in this XCD's L2
    else __hip_atomic_store((gu64*)g, x, NM_RLX_AGENT);                                            // write-through
}

// 16-byte agent-scope (sc1: past L1, served by L2) loads of 4 or 8 consecutive KB; a load holds two granules, each
// written by ONE 8-byte store.  Inline assembly: hipcc has no 16-byte atomic load, and it hoists buffer-load builtins
// out of a spin loop (a "memory" clobber does not stop it) -- loads and their wait are one volatile statement.
template <int N>
__device__ __forceinline__ void clu_load_kb(const char* base, u32x4 (&x)[N]) {
    static_assert(N == 4 || N == 8, "pieces of 4 or 8 KB per wave");
    if constexpr (N == 4) {
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n"
                     "global_load_dwordx4 %1, %4, off offset:1024 sc1\n"
                     "global_load_dwordx4 %2, %4, off offset:2048 sc1\n"
                     "global_load_dwordx4 %3, %4, off offset:3072 sc1\n"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(base) : "memory");
    } else {
        const char* hi = base + 4096;
        asm volatile("global_load_dwordx4 %0, %8, off sc1\n"
                     "global_load_dwordx4 %1, %8, off offset:1024 sc1\n"
                     "global_load_dwordx4 %2, %8, off offset:2048 sc1\n"
                     "global_load_dwordx4 %3, %8, off offset:3072 sc1\n"
                     "global_load_dwordx4 %4, %9, off sc1\n"
                     "global_load_dwordx4 %5, %9, off offset:1024 sc1\n"
                     "global_load_dwordx4 %6, %9, off offset:2048 sc1\n"
                     "global_load_dwordx4 %7, %9, off offset:3072 sc1\n"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
                     : "v"(base), "v"(hi) : "memory");
    }
}

// A wave waits for its producers: ONE granule per producer (lanes 0..NCH-1, one 8-byte load each -- a thousand lanes
// per CU polling would keep the memory pipes busy with polls), the one its LAST epilogue thread writes (row 15 of the
// last row tile, k 15 of the chunk).
struct CluWait {
    long t0;
    unsigned spins;
};

template <int RT, int NCH>
__device__ __forceinline__ void clu_wait(const u64* __restrict__ X, int wave, int lane, unsigned tag, gu32* err, CluWait& w) {
    w.t0 = 0;
    w.spins = 0;
    const gu64* poll = (const gu64*)(X + clu_unit<RT, NCH>(wave, min(lane, NCH - 1), 1, RT - 1, 15, 3) * 2 + 1);
    for (;;) {
        bool ok = true;
        if (lane < NCH) ok = (unsigned)(__hip_atomic_load(poll, NM_RLX_AGENT) >> 32) == tag;
        if (__all(ok)) return;
        if (clu_gave_up(err, w.t0, w.spins)) return;
    }
}

// ... and gathers its K-slice of one row tile of the cluster's operand in pieces of 8 KB = 4 chunks of 16 k (32
// registers in flight), each piece read again until its tags are right -- after the wait that is nearly always the
// first pass.
template <int RT, int NCH>
__device__ __forceinline__ void clu_gather(const u64* __restrict__ X, int wave, int lane, int rt, int piece, unsigned tag,
                                           gu32* err, CluWait& w, float (&a)[4][4]) {
    static_assert(NCH % 4 == 0, "whole pieces");
    const char* base = (const char*)X + (clu_unit<RT, NCH>(wave, 4 * piece, 0, rt, 0, 0) + lane) * 16;
    for (;;) {
        bool ok = true;
        u32x4 x[8];
        clu_load_kb<8>(base, x);
#pragma unroll
        for (int q = 0; q < 8; ++q) {                             // unit block (chunk q / 2 of the piece, hh = q % 2)
            a[q / 2][2 * (q & 1)] = __uint_as_float(x[q][0]);
            a[q / 2][2 * (q & 1) + 1] = __uint_as_float(x[q][2]);
            ok &= x[q][1] == tag && x[q][3] == tag;
        }
        if (__all(ok)) return;
        if (clu_gave_up(err, w.t0, w.spins)) return;
    }
}

// plain (not handed-off) operand rows of one row tile: the forward loop's initial state
__device__ __forceinline__ void clu_load_plain(const float* __restrict__ A, long ld, int R, int row0, int k_wave,
                                               int lane, float (&a)[4][4]) {
    const int i16 = lane & 15, kq = lane >> 4;
    const float* p = A + (long)min(row0 + i16, R - 1) * ld + k_wave + 4 * kq;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(p + 16 * c);
        a[c][0] = v.x; a[c][1] = v.y; a[c][2] = v.z; a[c][3] = v.w;
    }
}

// acc += A-piece . B-piece for one 16x16 output tile: the chain of nm_gemm.hip's skinny16_tile (chunk by chunk, the
// four k-phases of a chunk in order); ``b`` points at the piece's four chunks of this wave's kernel slice
__device__ __forceinline__ void clu_mma(f32x4& acc, const float (&a)[4][4], const float (*b)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[c][j], acc, 0, 0, 0);
}

// partial tile of this wave -> LDS [wave][tile][reg][lane]
__device__ __forceinline__ void clu_put(float* red, int ntiles, int wave, int tile, int lane, const f32x4& acc) {
    float* p = red + ((long)(wave * ntiles + tile) * 4) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i * 64] = acc[i];
}

// sum over the waves, in order (the order of skinny16_tile's epilogue)
__device__ __forceinline__ float clu_get(const float* red, int ntiles, int nw, int tile, int reg, int ln) {
    float s = 0.0f;
    for (int w = 0; w < nw; ++w) s += red[((long)(w * ntiles + tile) * 4 + reg) * 64 + ln];
    return s;
}

#ifdef NM_CLU_DEBUG
#define CLU_STAMP(i) do { if (q.dbg && role.cl == 3 && role.jb == 5 && lane == 0 && (wave == 0 || wave == NW - 1) && t >= 20 && t < 24) \
        q.dbg[((t - 20) * 2 + (wave ? 1 : 0)) * 8 + (i)] = (long)wall_clock64(); } while (0)
#else
#define CLU_STAMP(i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// forward: per step  r|u = sigmoid(xp + h.Wg_h), rh = r*h  ->  c = tanh(xp + rh.Wc_h), h' = u*h + (1-u)*c
// ---------------------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(512, 3) void gru_cluster_fwd_kernel(GruClu q) {
    constexpr int NCH = 4;                               // 64 k-values per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = (int)(blockDim.x >> 6);
    const int H = q.e.H, R = (int)q.e.R;
    if (q.force_fail && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store((gu32*)q.hdr, 1u, NM_RLX_AGENT);
    const CluRole role = clu_roles(q.hdr, q.ndir * q.nrb, H / 16, reinterpret_cast<int*>(lds), q.force_global);
    if (!role.active) {
        // (a workgroup without a role still reports a raised error word: with force_fail it may be the only one to see it early)
        if (tid == 0 && q.sticky && __hip_atomic_load((gu32*)q.hdr, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    const int jb = role.jb;
    const int d = role.cl / q.nrb, rb = role.cl % q.nrb;
    const int row0 = rb * 16 * RT;
    const int n16 = lane & 15, kq = lane >> 4;
    const int k_wave = wave * 16 * NCH;
    float* red1 = lds;                                   // [NW][2 RT][4][64]
    float* red2 = lds + (long)NW * 2 * RT * 256;         // [NW][RT][4][64]
    gu32* err = (gu32*)q.hdr;

    // this wave's slices of the recurrent kernels, for the whole loop
    float wr[NCH][4], wu[NCH][4], wk[NCH][4];
    {
        const float* Wg = q.wg + (long)d * q.sg + 16 * jb + n16;
        const float* Wc = q.wc + (long)d * q.sc + 16 * jb + n16;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long k = k_wave + 16 * c + 4 * kq + j;
                wr[c][j] = Wg[k * q.ldg];
                wu[c][j] = Wg[k * q.ldg + H];
                wk[c][j] = Wc[k * q.ldc];
            }
    }

    // epilogue threads: ONE element (row, col) each, laid out as skinny16_tile maps a 16x16 tile (threads 0..255 the
    // first row tile, 256..511 the second: a workgroup has at least 256 RT threads)
    const bool epi = tid < 256 * RT;
    const int ert = tid >> 8, reg = (tid >> 6) & 3, ln = tid & 63;
    const int rloc = 4 * (ln >> 4) + reg, col = 16 * jb + (ln & 15);
    const int row = row0 + 16 * ert + rloc;
    const bool mine = epi && row < R;
    const long ro = (long)d * R + min(row, R - 1);
    const int len = (mine && q.e.lengths) ? q.e.lengths[row] : 0x7fffffff;
    float hreg = mine ? q.e.h_in[ro * H + col] : 0.0f;
    const bool rev = ((q.e.rev_mask >> d) & 1) && q.e.lengths;
    u64* XA = q.xa + (long)role.cl * 16 * RT * H;        // this cluster's granules of r*h ...
    u64* XB = q.xb + (long)role.cl * 16 * RT * H;        // ... and of h'

    for (int t = 0; t < q.steps; ++t) {
        CLU_STAMP(0);
        // what the epilogues need from plain memory, requested before any waiting
        const bool live = mine && t < len;
        const int pos = rev ? len - 1 - t : t;
        float xr = 0.0f, xu = 0.0f, xc = 0.0f, ureg = 0.0f;
        if (live) {
            const float* x = q.e.xp + d * q.e.x_dir + (long)row * q.e.x_row + (long)pos * q.e.x_time + col;
            xr = x[0];
            xu = x[H];
            xc = x[2 * H];
        }
        float a[4][4];
        CluWait cw;
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        // ---- stage A: gates
        if (t > 0) clu_wait<RT, NCH>(XB, wave, lane, (unsigned)(2 * t), err, cw);
        CLU_STAMP(1);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (t == 0) clu_load_plain(q.e.h_in + (long)d * R * H, H, R, row0 + 16 * rt, k_wave, lane, a);
            else clu_gather<RT, NCH>(XB, wave, lane, rt, 0, (unsigned)(2 * t), err, cw, a);
            f32x4 ar = zero, au = zero;
            clu_mma(ar, a, wr);
            clu_mma(au, a, wu);
            clu_put(red1, 2 * RT, wave, 2 * rt, lane, ar);
            clu_put(red1, 2 * RT, wave, 2 * rt + 1, lane, au);
        }
        CLU_STAMP(2);
        __syncthreads();
        CLU_STAMP(3);
        if (epi) {
            const float sr = clu_get(red1, 2 * RT, NW, 2 * ert, reg, ln);
            const float su = clu_get(red1, 2 * RT, NW, 2 * ert + 1, reg, ln);
            const float r = live ? nm_sigmoid(xr + sr) : 0.0f;
            const float u = live ? nm_sigmoid(xu + su) : 0.0f;
            const float rh = live ? r * hreg : 0.0f;
            ureg = u;
            clu_publish<RT, NCH>(XA, role.local, ert, rloc, col, (unsigned)(2 * t + 1), rh);    // (rows past R: zeros)
            if (mine) {
                float* ru = q.e.ru + (long)t * q.ru_step + ro * 2 * H;
                ru[col] = r;
                ru[H + col] = u;
                if (q.e.rh) q.e.rh[(long)t * q.rh_step + ro * H + col] = rh;
            }
        }
        // ---- stage B: candidate + blend
        CLU_STAMP(4);
        clu_wait<RT, NCH>(XA, wave, lane, (unsigned)(2 * t + 1), err, cw);
        CLU_STAMP(5);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            clu_gather<RT, NCH>(XA, wave, lane, rt, 0, (unsigned)(2 * t + 1), err, cw, a);
            f32x4 ac = zero;
            clu_mma(ac, a, wk);
            clu_put(red2, RT, wave, rt, lane, ac);
        }
        __syncthreads();
        CLU_STAMP(6);
        if (epi) {
            const float sc = clu_get(red2, RT, NW, ert, reg, ln);
            float hn = hreg, c = 0.0f;
            if (live) {
                c = nm_tanh(xc + sc);
                hn = ureg * hreg + (1.0f - ureg) * c;
            }
            hreg = hn;
            if (t + 1 < q.steps) clu_publish<RT, NCH>(XB, role.local, ert, rloc, col, (unsigned)(2 * t + 2), hn);
            if (mine) {
                q.e.h_out[(long)t * q.h_step + ro * H + col] = hn;
                if (q.e.c_save) q.e.c_save[(long)t * q.c_step + ro * H + col] = c;
                if (live && q.e.out)
                    q.e.out[d * q.e.o_dir + (long)row * q.e.o_row + (long)pos * q.e.o_time + col] = hn;
            }
        }
        CLU_STAMP(7);
    }
    if (tid == 0 && q.sticky && __hip_atomic_load(err, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------
// backward (nm_gru_gemm modes 4, 3 per step, last step first).  State per element: dh.  Step t:
//   blend bwd   dhv = dh (+ sB of the step after: d(gates) . Wg_h^T) + dout;  dc_pre = dhv (1-u)(1-c^2),
//               du_pre = dhv (h_prev - c) u (1-u),  dh = dhv u                      -> publishes dc_pre and du_pre
//   stage A     sA = dc_pre . Wc_h^T = d(r*h)          (K = H)
//   gates bwd   dr_pre = sA h_prev r (1-r),  dh += sA r                               -> publishes dr_pre
//   stage B     sB = [dr_pre | du_pre] . Wg_h^T        (K = 2H: waves 0..NW/2-1 the reset half, the rest the update half)
// and after step 0: dh_0 = dh + sB.
// ---------------------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(512, 3) void gru_cluster_bwd_kernel(GruClu q) {
    constexpr int NCA = 4, NCB = 8;                      // 64 k-values per wave of K = H, 128 of K = 2H
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = (int)(blockDim.x >> 6);
    const int H = q.e.H, R = (int)q.e.R;
    if (q.force_fail && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store((gu32*)q.hdr, 1u, NM_RLX_AGENT);
    const CluRole role = clu_roles(q.hdr, q.ndir * q.nrb, H / 16, reinterpret_cast<int*>(lds), q.force_global);
    if (!role.active) {
        // (a workgroup without a role still reports a raised error word: with force_fail it may be the only one to see it early)
        if (tid == 0 && q.sticky && __hip_atomic_load((gu32*)q.hdr, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    const int jb = role.jb;
    const int d = role.cl / q.nrb, rb = role.cl % q.nrb;
    const int row0 = rb * 16 * RT;
    const int n16 = lane & 15, kq = lane >> 4;
    float* redA = lds;                                   // [NW][RT][4][64]
    float* redB = lds + (long)NW * RT * 256;             // [NW][RT][4][64]
    gu32* err = (gu32*)q.hdr;

    // rows 16 jb + n16 of the recurrent kernels ([K_in][N_out]: their transposes' columns), this wave's K-slices
    float wc[NCA][4], wg[NCB][4];
    {
        const float* Wc = q.wc + (long)d * q.sc + (long)(16 * jb + n16) * q.ldc + wave * 16 * NCA + 4 * kq;
        const float* Wg = q.wg + (long)d * q.sg + (long)(16 * jb + n16) * q.ldg + wave * 16 * NCB + 4 * kq;
#pragma unroll
        for (int c = 0; c < NCA; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(Wc + 16 * c);
            wc[c][0] = v.x; wc[c][1] = v.y; wc[c][2] = v.z; wc[c][3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(Wg + 16 * c);
            wg[c][0] = v.x; wg[c][1] = v.y; wg[c][2] = v.z; wg[c][3] = v.w;
        }
    }

    const bool epi = tid < 256 * RT;
    const int ert = tid >> 8, reg = (tid >> 6) & 3, ln = tid & 63;
    const int rloc = 4 * (ln >> 4) + reg, col = 16 * jb + (ln & 15);
    const int row = row0 + 16 * ert + rloc;
    const bool mine = epi && row < R;
    const long ro = (long)d * R + min(row, R - 1);
    const int len = (mine && q.e.lengths) ? q.e.lengths[row] : 0x7fffffff;
    float dh = mine ? q.e.dh[ro * H + col] : 0.0f;
    const bool rev = ((q.e.rev_mask >> d) & 1) && q.e.lengths;
    u64* XA = q.xa + (long)role.cl * 16 * RT * H;                       // dc_pre
    u64* XB = q.xb + (long)role.cl * 16 * RT * 2 * H;                   // [dr_pre | du_pre], two copies
    const long xb_copy = (long)q.ndir * q.nrb * 16 * RT * 2 * H;
    const unsigned upper = wave >= NW / 2 ? 1u : 0u;                    // this wave multiplies the update half
    float sB = 0.0f;

    for (int i = 0; i < q.steps; ++i) {
        const int t = q.steps - 1 - i;
        const bool live = mine && t < len;
        const int pos = rev ? len - 1 - t : t;
        const int ppos = rev ? pos + 1 : pos - 1;
        float r = 0.0f, u = 0.0f, c = 0.0f, hp = 0.0f, dout = 0.0f;
        if (live) {
            const float* ru = q.e.ru + (long)t * q.ru_step + ro * 2 * H;
            r = ru[col];
            u = ru[H + col];
            c = q.e.c[(long)t * q.c_step + ro * H + col];
            hp = gru_epi_hprev(q.e, ro, d, row, t, ppos, col);
            if (q.e.dout) dout = q.e.dout[d * q.e.do_dir + (long)row * q.e.do_row + (long)pos * q.e.do_time + col];
        }
        u64* XBi = XB + (i & 1) * xb_copy;
        // ---- blend backward of step t (sB: the product of the step after, reduced at the end of the last pass)
        if (epi) {
            const float s = (i == 0) ? dh : sB + dh;
            float dcp = 0.0f, dup = 0.0f;
            if (live) {
                const float dhv = s + dout;
                dcp = dhv * (1.0f - u) * (1.0f - c * c);
                dup = dhv * (hp - c) * u * (1.0f - u);
                dh = dhv * u;
            } else {
                dh = s;
            }
            clu_publish<RT, NCA>(XA, role.local, ert, rloc, col, (unsigned)(2 * i + 1), dcp);
            clu_publish<RT, NCB>(XBi, role.local, ert, rloc, H + col, (unsigned)(2 * i + 1), dup);
            if (live) {
                float* dx = q.e.dxp + d * q.e.dx_dir + (long)row * q.e.dx_row + (long)pos * q.e.dx_time;
                dx[H + col] = dup;
                dx[2 * H + col] = dcp;
            }
        }
        // ---- stage A: d(r*h) = dc_pre . Wc_h^T
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        {
            float a[4][4];
            CluWait cw;
            clu_wait<RT, NCA>(XA, wave, lane, (unsigned)(2 * i + 1), err, cw);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                clu_gather<RT, NCA>(XA, wave, lane, rt, 0, (unsigned)(2 * i + 1), err, cw, a);
                f32x4 acc = zero;
                clu_mma(acc, a, wc);
                clu_put(redA, RT, wave, rt, lane, acc);
            }
        }
        __syncthreads();
        if (epi) {
            const float sA = clu_get(redA, RT, NW, ert, reg, ln);
            float drp = 0.0f;
            if (live) {
                drp = sA * hp * r * (1.0f - r);
                dh = dh + sA * r;
                q.e.dxp[d * q.e.dx_dir + (long)row * q.e.dx_row + (long)pos * q.e.dx_time + col] = drp;
            }
            clu_publish<RT, NCB>(XBi, role.local, ert, rloc, col, (unsigned)(2 * i + 2), drp);
        }
        // ---- stage B: [dr_pre | du_pre] . Wg_h^T (the update half was published a stage earlier)
        {
            float a[4][4];
            CluWait cw;
            clu_wait<RT, NCB>(XBi, wave, lane, (unsigned)(2 * i + 2) - upper, err, cw);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 acc = zero;
#pragma unroll
                for (int piece = 0; piece < NCB / 4; ++piece) {
                    clu_gather<RT, NCB>(XBi, wave, lane, rt, piece, (unsigned)(2 * i + 2) - upper, err, cw, a);
                    clu_mma(acc, a, wg + 4 * piece);
                }
                clu_put(redB, RT, wave, rt, lane, acc);
            }
        }
        __syncthreads();
        if (epi) sB = clu_get(redB, RT, NW, ert, reg, ln);
    }
    if (mine) q.e.dh[ro * H + col] = sB + dh;
    if (tid == 0 && q.sticky && __hip_atomic_load(err, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------
// NematusGRUCell time loops (nn/ortho_gru_cell.py:73-105), round 6.  The reset gate is applied AFTER the state
// projection -- g = sigmoid(x_g + h.U_g), sc = h.U_c, c = tanh(x_c + sc * r), h' = u h + (1 - u) c -- so both recurrent
// products read h only: a step is ONE product (16 units' r, u and sc columns per workgroup) with one elementwise stage
// and ONE hand-off (h'), where the TF GRUCell above needs two.  Steps alternate between the two granule buffers (what
// step t published is overwritten by step t + 2, behind every consumer).  Saved for the backward pass: r | u, c and sc.
//   backward, step t (last first):  dhv = dh + dout;  dc' = dhv (1-u)(1-c^2);  du' = dhv (h_prev - c) u (1-u);
//   dsc = dc' r;  dr' = dc' sc r (1-r);  dh = dhv u + [dr' | du' | dsc] . [U_g ; U_c]^T     (ONE product, K = 3H)
// dxp (sequence-addressed, 4H wide per direction) receives [dr' | du' | dc' | dsc]: the input-kernel gradients come
// from its first 3H columns, the state-kernel gradients from columns [0, 2H) and [3H, 4H) against the shifted states.
// ---------------------------------------------------------------------------------------------------------------
struct NemClu {
    GruClu q;                    // e.rh / rh_step: sc of every step; xa / xb: the two buffers (forward: width H,
                                 // backward: width 3H each)
    const float* bgs;            // optional state biases [ndir][2H] / [ndir][H]
    const float* bcs;
};

template <int RT>
__global__ __launch_bounds__(512, 3) void nematus_cluster_fwd_kernel(NemClu a) {
    constexpr int NCH = 4;
    const GruClu& q = a.q;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = (int)(blockDim.x >> 6);
    const int H = q.e.H, R = (int)q.e.R;
    if (q.force_fail && blockIdx.x == 0 && tid == 0) __hip_atomic_store((gu32*)q.hdr, 1u, NM_RLX_AGENT);
    const CluRole role = clu_roles(q.hdr, q.ndir * q.nrb, H / 16, reinterpret_cast<int*>(lds), q.force_global);
    if (!role.active) {
        if (tid == 0 && q.sticky && __hip_atomic_load((gu32*)q.hdr, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    const int jb = role.jb;
    const int d = role.cl / q.nrb, rb = role.cl % q.nrb;
    const int row0 = rb * 16 * RT;
    const int n16 = lane & 15, kq = lane >> 4;
    const int k_wave = wave * 16 * NCH;
    float* red = lds;                                    // [NW][3 RT][4][64]
    gu32* err = (gu32*)q.hdr;
    float wr[NCH][4], wu[NCH][4], wk[NCH][4];
    {
        const float* Wg = q.wg + (long)d * q.sg + 16 * jb + n16;
        const float* Wc = q.wc + (long)d * q.sc + 16 * jb + n16;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long k = k_wave + 16 * c + 4 * kq + j;
                wr[c][j] = Wg[k * q.ldg];
                wu[c][j] = Wg[k * q.ldg + H];
                wk[c][j] = Wc[k * q.ldc];
            }
    }
    const bool epi = tid < 256 * RT;
    const int ert = tid >> 8, reg = (tid >> 6) & 3, ln = tid & 63;
    const int rloc = 4 * (ln >> 4) + reg, col = 16 * jb + (ln & 15);
    const int row = row0 + 16 * ert + rloc;
    const bool mine = epi && row < R;
    const long ro = (long)d * R + min(row, R - 1);
    const int len = (mine && q.e.lengths) ? q.e.lengths[row] : 0x7fffffff;
    float hreg = mine ? q.e.h_in[ro * H + col] : 0.0f;
    const bool rev = ((q.e.rev_mask >> d) & 1) && q.e.lengths;
    const float b_r = a.bgs ? a.bgs[(long)d * 2 * H + col] : 0.0f, b_u = a.bgs ? a.bgs[(long)d * 2 * H + H + col] : 0.0f;
    const float b_c = a.bcs ? a.bcs[(long)d * H + col] : 0.0f;
    u64* X0 = q.xa + (long)role.cl * 16 * RT * H;
    u64* X1 = q.xb + (long)role.cl * 16 * RT * H;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};

    for (int t = 0; t < q.steps; ++t) {
        const bool live = mine && t < len;
        const int pos = rev ? len - 1 - t : t;
        float xr = 0.0f, xu = 0.0f, xc = 0.0f;
        if (live) {
            const float* x = q.e.xp + d * q.e.x_dir + (long)row * q.e.x_row + (long)pos * q.e.x_time + col;
            xr = x[0]; xu = x[H]; xc = x[2 * H];
        }
        u64* Xin = (t & 1) ? X0 : X1;                     // what step t - 1 published
        u64* Xout = (t & 1) ? X1 : X0;
        float av[4][4];
        CluWait cw;
        if (t > 0) clu_wait<RT, NCH>(Xin, wave, lane, (unsigned)t, err, cw);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (t == 0) clu_load_plain(q.e.h_in + (long)d * R * H, H, R, row0 + 16 * rt, k_wave, lane, av);
            else clu_gather<RT, NCH>(Xin, wave, lane, rt, 0, (unsigned)t, err, cw, av);
            f32x4 ar = zero, au = zero, ak = zero;
            clu_mma(ar, av, wr);
            clu_mma(au, av, wu);
            clu_mma(ak, av, wk);
            clu_put(red, 3 * RT, wave, 3 * rt, lane, ar);
            clu_put(red, 3 * RT, wave, 3 * rt + 1, lane, au);
            clu_put(red, 3 * RT, wave, 3 * rt + 2, lane, ak);
        }
        __syncthreads();
        if (epi) {
            const float sr = clu_get(red, 3 * RT, NW, 3 * ert, reg, ln);
            const float su = clu_get(red, 3 * RT, NW, 3 * ert + 1, reg, ln);
            const float sc = clu_get(red, 3 * RT, NW, 3 * ert + 2, reg, ln) + b_c;
            float hn = hreg, r = 0.0f, u = 0.0f, c = 0.0f;
            if (live) {
                r = nm_sigmoid(xr + sr + b_r);
                u = nm_sigmoid(xu + su + b_u);
                c = nm_tanh(xc + sc * r);
                hn = u * hreg + (1.0f - u) * c;
            }
            hreg = hn;
            if (t + 1 < q.steps) clu_publish<RT, NCH>(Xout, role.local, ert, rloc, col, (unsigned)(t + 1), hn);
            if (mine) {
                float* ru = q.e.ru + (long)t * q.ru_step + ro * 2 * H;
                ru[col] = r;
                ru[H + col] = u;
                q.e.c_save[(long)t * q.c_step + ro * H + col] = c;
                q.e.rh[(long)t * q.rh_step + ro * H + col] = live ? sc : 0.0f;
                q.e.h_out[(long)t * q.h_step + ro * H + col] = hn;
                if (live && q.e.out)
                    q.e.out[d * q.e.o_dir + (long)row * q.e.o_row + (long)pos * q.e.o_time + col] = hn;
            }
        }
        __syncthreads();                                  // the reduction buffer is rewritten by the next step
    }
    if (tid == 0 && q.sticky && __hip_atomic_load(err, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
}

template <int RT>
__global__ __launch_bounds__(512, 3) void nematus_cluster_bwd_kernel(NemClu a) {
    constexpr int NCB = 12;                              // 192 k-values per wave of K = 3H
    const GruClu& q = a.q;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = (int)(blockDim.x >> 6);
    const int H = q.e.H, R = (int)q.e.R;
    if (q.force_fail && blockIdx.x == 0 && tid == 0) __hip_atomic_store((gu32*)q.hdr, 1u, NM_RLX_AGENT);
    const CluRole role = clu_roles(q.hdr, q.ndir * q.nrb, H / 16, reinterpret_cast<int*>(lds), q.force_global);
    if (!role.active) {
        if (tid == 0 && q.sticky && __hip_atomic_load((gu32*)q.hdr, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    const int jb = role.jb;
    const int d = role.cl / q.nrb, rb = role.cl % q.nrb;
    const int row0 = rb * 16 * RT;
    const int n16 = lane & 15, kq = lane >> 4;
    float* red = lds;                                    // [NW][RT][4][64]
    gu32* err = (gu32*)q.hdr;
    // rows 16 jb + n16 of [U_g | U_c] ([K_in = H][2H] and [H][H]): this wave's 192 values of the 3H-long row
    float w[NCB][4];
    {
        const float* Wg = q.wg + (long)d * q.sg + (long)(16 * jb + n16) * q.ldg;
        const float* Wc = q.wc + (long)d * q.sc + (long)(16 * jb + n16) * q.ldc;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int k0 = 16 * (wave * NCB + c) + 4 * kq;
            const float4 v = k0 < 2 * H ? *reinterpret_cast<const float4*>(Wg + k0)
                                        : *reinterpret_cast<const float4*>(Wc + (k0 - 2 * H));
            w[c][0] = v.x; w[c][1] = v.y; w[c][2] = v.z; w[c][3] = v.w;
        }
    }
    const bool epi = tid < 256 * RT;
    const int ert = tid >> 8, reg = (tid >> 6) & 3, ln = tid & 63;
    const int rloc = 4 * (ln >> 4) + reg, col = 16 * jb + (ln & 15);
    const int row = row0 + 16 * ert + rloc;
    const bool mine = epi && row < R;
    const long ro = (long)d * R + min(row, R - 1);
    const int len = (mine && q.e.lengths) ? q.e.lengths[row] : 0x7fffffff;
    float dh = mine ? q.e.dh[ro * H + col] : 0.0f;
    const bool rev = ((q.e.rev_mask >> d) & 1) && q.e.lengths;
    u64* X0 = q.xa + (long)role.cl * 16 * RT * 3 * H;
    u64* X1 = q.xb + (long)role.cl * 16 * RT * 3 * H;
    float sB = 0.0f;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};

    for (int i = 0; i < q.steps; ++i) {
        const int t = q.steps - 1 - i;
        const bool live = mine && t < len;
        const int pos = rev ? len - 1 - t : t;
        const int ppos = rev ? pos + 1 : pos - 1;
        u64* X = (i & 1) ? X1 : X0;
        if (epi) {
            const float s = (i == 0) ? dh : sB + dh;
            float drp = 0.0f, dup = 0.0f, dsc = 0.0f, dcp = 0.0f;
            if (live) {
                const float* ru = q.e.ru + (long)t * q.ru_step + ro * 2 * H;
                const float r = ru[col], u = ru[H + col];
                const float c = q.e.c[(long)t * q.c_step + ro * H + col];
                const float sc = q.e.rh[(long)t * q.rh_step + ro * H + col];
                const float hp = gru_epi_hprev(q.e, ro, d, row, t, ppos, col);
                const float dout = q.e.dout ? q.e.dout[d * q.e.do_dir + (long)row * q.e.do_row + (long)pos * q.e.do_time + col] : 0.0f;
                const float dhv = s + dout;
                dcp = dhv * (1.0f - u) * (1.0f - c * c);
                dup = dhv * (hp - c) * u * (1.0f - u);
                dsc = dcp * r;
                drp = dcp * sc * r * (1.0f - r);
                dh = dhv * u;
                float* dx = q.e.dxp + d * q.e.dx_dir + (long)row * q.e.dx_row + (long)pos * q.e.dx_time;
                dx[col] = drp;
                dx[H + col] = dup;
                dx[2 * H + col] = dcp;
                dx[3 * H + col] = dsc;
            } else {
                dh = s;
            }
            clu_publish<RT, NCB>(X, role.local, ert, rloc, col, (unsigned)(i + 1), drp);
            clu_publish<RT, NCB>(X, role.local, ert, rloc, H + col, (unsigned)(i + 1), dup);
            clu_publish<RT, NCB>(X, role.local, ert, rloc, 2 * H + col, (unsigned)(i + 1), dsc);
        }
        {
            float av[4][4];
            CluWait cw;
            clu_wait<RT, NCB>(X, wave, lane, (unsigned)(i + 1), err, cw);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 acc = zero;
#pragma unroll
                for (int piece = 0; piece < NCB / 4; ++piece) {
                    clu_gather<RT, NCB>(X, wave, lane, rt, piece, (unsigned)(i + 1), err, cw, av);
                    clu_mma(acc, av, w + 4 * piece);
                }
                clu_put(red, RT, wave, rt, lane, acc);
            }
        }
        __syncthreads();
        if (epi) sB = clu_get(red, RT, NW, ert, reg, ln);
        __syncthreads();
    }
    if (mine) q.e.dh[ro * H + col] = sB + dh;
    if (tid == 0 && q.sticky && __hip_atomic_load(err, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------
// LSTMCell time loops (tf.nn.rnn_cell.LSTMCell as encoders/recurrent.py:21 and decoders/decoder.py:29 build it: gate
// order i, j, f, o, forget bias 1.0, state (c, h)), round 6.  z = xp + h.W_h; c' = sigmoid(f + fb) c + sigmoid(i) tanh(j);
// h' = sigmoid(o) tanh(c').  One product per step (the four gate columns of 16 units per workgroup), one element-wise
// stage, ONE hand-off (h'); c never leaves its thread.  Saved per step: the activated gates [i | j | f | o] and c'.
//   backward, step t (last first): dhv = dh + dout; tc = tanh(c'); do' = dhv tc o(1-o); dct = dc + dhv o (1 - tc^2);
//   di' = dct j i(1-i); dj' = dct i (1-j^2); df' = dct c_prev f(1-f); dc = dct f;
//   dh = [di' | dj' | df' | do'] . W_h^T   (ONE product, K = 4H);  dxp (4H wide) receives the four pre-activation gradients.
// ---------------------------------------------------------------------------------------------------------------
struct LstmClu {
    GruClu q;                    // e.ru / ru_step: the gates of every step (4H wide); e.c_save / e.c, c_step: c' of every step
    float forget_bias;
};

template <int RT>
__global__ __launch_bounds__(512, 2) void lstm_cluster_fwd_kernel(LstmClu a) {
    constexpr int NCH = 4;
    const GruClu& q = a.q;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = (int)(blockDim.x >> 6);
    const int H = q.e.H, R = (int)q.e.R;
    if (q.force_fail && blockIdx.x == 0 && tid == 0) __hip_atomic_store((gu32*)q.hdr, 1u, NM_RLX_AGENT);
    const CluRole role = clu_roles(q.hdr, q.ndir * q.nrb, H / 16, reinterpret_cast<int*>(lds), q.force_global);
    if (!role.active) {
        if (tid == 0 && q.sticky && __hip_atomic_load((gu32*)q.hdr, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    const int jb = role.jb;
    const int d = role.cl / q.nrb, rb = role.cl % q.nrb;
    const int row0 = rb * 16 * RT;
    const int n16 = lane & 15, kq = lane >> 4;
    const int k_wave = wave * 16 * NCH;
    float* red = lds;                                    // [NW][4 RT][4][64]
    gu32* err = (gu32*)q.hdr;
    float wg[4][NCH][4];                                 // this wave's K-slice of the i, j, f, o columns of its 16 units
    {
        const float* W = q.wg + (long)d * q.sg + 16 * jb + n16;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    wg[g][c][j] = W[(long)(k_wave + 16 * c + 4 * kq + j) * q.ldg + g * H];
    }
    const bool epi = tid < 256 * RT;
    const int ert = tid >> 8, reg = (tid >> 6) & 3, ln = tid & 63;
    const int rloc = 4 * (ln >> 4) + reg, col = 16 * jb + (ln & 15);
    const int row = row0 + 16 * ert + rloc;
    const bool mine = epi && row < R;
    const long ro = (long)d * R + min(row, R - 1);
    const int len = (mine && q.e.lengths) ? q.e.lengths[row] : 0x7fffffff;
    float hreg = mine ? q.e.h_in[ro * H + col] : 0.0f;
    float creg = 0.0f;                                   // (zero initial cell state: what the encoders start from)
    const bool rev = ((q.e.rev_mask >> d) & 1) && q.e.lengths;
    u64* X0 = q.xa + (long)role.cl * 16 * RT * H;
    u64* X1 = q.xb + (long)role.cl * 16 * RT * H;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};

    for (int t = 0; t < q.steps; ++t) {
        const bool live = mine && t < len;
        const int pos = rev ? len - 1 - t : t;
        float x4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (live) {
            const float* x = q.e.xp + d * q.e.x_dir + (long)row * q.e.x_row + (long)pos * q.e.x_time + col;
#pragma unroll
            for (int g = 0; g < 4; ++g) x4[g] = x[g * H];
        }
        u64* Xin = (t & 1) ? X0 : X1;
        u64* Xout = (t & 1) ? X1 : X0;
        float av[4][4];
        CluWait cw;
        if (t > 0) clu_wait<RT, NCH>(Xin, wave, lane, (unsigned)t, err, cw);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (t == 0) clu_load_plain(q.e.h_in + (long)d * R * H, H, R, row0 + 16 * rt, k_wave, lane, av);
            else clu_gather<RT, NCH>(Xin, wave, lane, rt, 0, (unsigned)t, err, cw, av);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 acc = zero;
                clu_mma(acc, av, wg[g]);
                clu_put(red, 4 * RT, wave, 4 * rt + g, lane, acc);
            }
        }
        __syncthreads();
        if (epi) {
            float hn = hreg, gi = 0.0f, gj = 0.0f, gf = 0.0f, go = 0.0f;
            if (live) {
                gi = nm_sigmoid(x4[0] + clu_get(red, 4 * RT, NW, 4 * ert, reg, ln));
                gj = nm_tanh(x4[1] + clu_get(red, 4 * RT, NW, 4 * ert + 1, reg, ln));
                gf = nm_sigmoid(x4[2] + clu_get(red, 4 * RT, NW, 4 * ert + 2, reg, ln) + a.forget_bias);
                go = nm_sigmoid(x4[3] + clu_get(red, 4 * RT, NW, 4 * ert + 3, reg, ln));
                creg = gf * creg + gi * gj;
                hn = go * nm_tanh(creg);
            }
            hreg = hn;
            if (t + 1 < q.steps) clu_publish<RT, NCH>(Xout, role.local, ert, rloc, col, (unsigned)(t + 1), hn);
            if (mine) {
                float* gs = q.e.ru + (long)t * q.ru_step + ro * 4 * H;
                gs[col] = gi; gs[H + col] = gj; gs[2 * H + col] = gf; gs[3 * H + col] = go;
                q.e.c_save[(long)t * q.c_step + ro * H + col] = creg;
                q.e.h_out[(long)t * q.h_step + ro * H + col] = hn;
                if (live && q.e.out)
                    q.e.out[d * q.e.o_dir + (long)row * q.e.o_row + (long)pos * q.e.o_time + col] = hn;
            }
        }
        __syncthreads();
    }
    if (tid == 0 && q.sticky && __hip_atomic_load(err, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
}

template <int RT>
__global__ __launch_bounds__(512, 2) void lstm_cluster_bwd_kernel(LstmClu a) {
    constexpr int NCB = 16;                              // 256 k-values per wave of K = 4H
    const GruClu& q = a.q;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = (int)(blockDim.x >> 6);
    const int H = q.e.H, R = (int)q.e.R;
    if (q.force_fail && blockIdx.x == 0 && tid == 0) __hip_atomic_store((gu32*)q.hdr, 1u, NM_RLX_AGENT);
    const CluRole role = clu_roles(q.hdr, q.ndir * q.nrb, H / 16, reinterpret_cast<int*>(lds), q.force_global);
    if (!role.active) {
        if (tid == 0 && q.sticky && __hip_atomic_load((gu32*)q.hdr, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    const int jb = role.jb;
    const int d = role.cl / q.nrb, rb = role.cl % q.nrb;
    const int row0 = rb * 16 * RT;
    const int n16 = lane & 15, kq = lane >> 4;
    float* red = lds;                                    // [NW][RT][4][64]
    gu32* err = (gu32*)q.hdr;
    float w[NCB][4];                                     // row 16 jb + n16 of W_h ([H][4H]): this wave's 256 of its 4H values
    {
        const float* W = q.wg + (long)d * q.sg + (long)(16 * jb + n16) * q.ldg + wave * 16 * NCB + 4 * kq;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(W + 16 * c);
            w[c][0] = v.x; w[c][1] = v.y; w[c][2] = v.z; w[c][3] = v.w;
        }
    }
    const bool epi = tid < 256 * RT;
    const int ert = tid >> 8, reg = (tid >> 6) & 3, ln = tid & 63;
    const int rloc = 4 * (ln >> 4) + reg, col = 16 * jb + (ln & 15);
    const int row = row0 + 16 * ert + rloc;
    const bool mine = epi && row < R;
    const long ro = (long)d * R + min(row, R - 1);
    const int len = (mine && q.e.lengths) ? q.e.lengths[row] : 0x7fffffff;
    float dh = mine ? q.e.dh[ro * H + col] : 0.0f;
    float dc = 0.0f;
    const bool rev = ((q.e.rev_mask >> d) & 1) && q.e.lengths;
    u64* X0 = q.xa + (long)role.cl * 16 * RT * 4 * H;
    u64* X1 = q.xb + (long)role.cl * 16 * RT * 4 * H;
    float sB = 0.0f;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};

    for (int i = 0; i < q.steps; ++i) {
        const int t = q.steps - 1 - i;
        const bool live = mine && t < len;
        const int pos = rev ? len - 1 - t : t;
        u64* X = (i & 1) ? X1 : X0;
        if (epi) {
            const float s = (i == 0) ? dh : sB + dh;
            float dz[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (live) {
                const float* gs = q.e.ru + (long)t * q.ru_step + ro * 4 * H;
                const float gi = gs[col], gj = gs[H + col], gf = gs[2 * H + col], go = gs[3 * H + col];
                const float cn = q.e.c[(long)t * q.c_step + ro * H + col];
                const float cp = t > 0 ? q.e.c[(long)(t - 1) * q.c_step + ro * H + col] : 0.0f;
                const float dout = q.e.dout ? q.e.dout[d * q.e.do_dir + (long)row * q.e.do_row + (long)pos * q.e.do_time + col] : 0.0f;
                const float dhv = s + dout;
                const float tc = nm_tanh(cn);
                const float dct = dc + dhv * go * (1.0f - tc * tc);
                dz[0] = dct * gj * gi * (1.0f - gi);
                dz[1] = dct * gi * (1.0f - gj * gj);
                dz[2] = dct * cp * gf * (1.0f - gf);
                dz[3] = dhv * tc * go * (1.0f - go);
                dc = dct * gf;
                dh = 0.0f;                               // all of dh' flows through the gates: the product below
                float* dx = q.e.dxp + d * q.e.dx_dir + (long)row * q.e.dx_row + (long)pos * q.e.dx_time;
#pragma unroll
                for (int g = 0; g < 4; ++g) dx[g * H + col] = dz[g];
            } else {
                dh = s;                                  // beyond the sentence: both states are carried
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
                clu_publish<RT, NCB>(X, role.local, ert, rloc, g * H + col, (unsigned)(i + 1), dz[g]);
        }
        {
            float av[4][4];
            CluWait cw;
            clu_wait<RT, NCB>(X, wave, lane, (unsigned)(i + 1), err, cw);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 acc = zero;
#pragma unroll
                for (int piece = 0; piece < NCB / 4; ++piece) {
                    clu_gather<RT, NCB>(X, wave, lane, rt, piece, (unsigned)(i + 1), err, cw, av);
                    clu_mma(acc, av, w + 4 * piece);
                }
                clu_put(red, RT, wave, rt, lane, acc);
            }
        }
        __syncthreads();
        if (epi) sB = clu_get(red, RT, NW, ert, reg, ln);
        __syncthreads();
    }
    if (mine) q.e.dh[ro * H + col] = sB + dh;
    if (tid == 0 && q.sticky && __hip_atomic_load(err, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------
// One DECODING step's recurrent part as one launch (round 6): Decoder.next_state up to the attention query
// (decoders/decoder.py:279-325, plain GRUCell + input tables, what nm_decoder_step_fused launched as three dependent
// step groups: 7.2 + 7.2 + 9.4 us of kernels and two graph edges per greedy step):
//   stage 1   r | u = sigmoid(h . Wg_h + table[id, :2H] + bg),  rh = r * h                       -> publishes rh
//   stage 2   c = tanh(rh . Wc_h + table[id, 2H:3H]),  h' = u h + (1 - u) c                     -> publishes h'
//   stage 3   y = h' . Wq + bq  (attention query),  pre = h' . Wo_h + table[id, 3H:] + bo        -> plain stores
// Same clusters, roles, granules and hand-offs as the time loops above (16 rows per cluster, H / 16 workgroups of 16
// hidden units each, H / 64 waves that split K); stage 3's A + O output columns are dealt to the cluster's workgroups
// tile by tile.  The weights are NOT stationary here -- a launch is one step, the vocabulary projection and the argmax
// sit between two of them -- so every wave fetches its slices ([N, K] transposed weights: 16 bytes per lane) while
// the roles are agreed.  Nothing is zeroed between launches: the header counts launches (``epoch``), tags are
// 4 epoch + stage, and the workgroup that finishes last puts the role counters back and advances the epoch.
// ---------------------------------------------------------------------------------------------------------------
#define DEC_T3_MAX 4
struct DecClu {
    int R, H, A, O, ncl;                       // rows, state, query and output widths; clusters of 16 rows
    const float* h_in; long ld_h;
    const float* table; long ld_table; const int* ids;
    const float* bg; const float* bq; const float* bo;
    const float* wg_t; long ld_wg;             // [2H][K = H] state half of the gates kernel, transposed
    const float* wc_t; long ld_wc;             // [H][H]
    const float* wq_t; long ld_wq;             // [A][H]
    const float* wo_t; long ld_wo;             // [O][H]
    float* h_out; long ld_ho; float* h_out2; long ld_ho2;
    float* y; long ld_y; float* pre; long ld_pre;
    unsigned* hdr; u64* xa; u64* xb;
    int force_global; unsigned* sticky; int force_fail;
};

// header words: [0] error, [1] arrivals, [2] workgroups done, [3] epoch, [8 + x] tickets of XCD x
__device__ __forceinline__ void dec_load_w(const float* Wt, long ld, int n_first, int nmax, int k_wave, int lane,
                                           float (&w)[4][4]) {
    const int n16 = lane & 15, kq = lane >> 4;
    const float* p = Wt + (long)min(n_first + n16, nmax - 1) * ld + k_wave + 4 * kq;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(p + 16 * c);
        w[c][0] = v.x; w[c][1] = v.y; w[c][2] = v.z; w[c][3] = v.w;
    }
}

__global__ __launch_bounds__(512, 2) void dec_step_cluster_kernel(DecClu q) {
    constexpr int NCH = 4, RT = 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = (int)(blockDim.x >> 6);
    const int H = q.H, R = q.R;
    const int nj = H / 16;
    const int nt3 = (q.A + q.O) / 16, t3n = (nt3 + nj - 1) / nj;          // stage-3 tiles, per workgroup
    gu32* err = (gu32*)q.hdr;
    if (q.force_fail && blockIdx.x == 0 && tid == 0) __hip_atomic_store(err, 1u, NM_RLX_AGENT);
    // (the launch's epoch travels through a word BEHIND the reduction buffers: no static __shared__ object next to the
    // dynamic region, and nothing a fast wave writes later can land on it)
    unsigned* epoch_s = reinterpret_cast<unsigned*>(lds + (long)NW * (3 + t3n) * 256);
    if (tid == 0) *epoch_s = __hip_atomic_load((gu32*)q.hdr + 3, NM_RLX_AGENT);
    const CluRole role = clu_roles(q.hdr, q.ncl, nj, reinterpret_cast<int*>(lds), q.force_global);
    const unsigned epoch = *epoch_s;
    auto finish = [&]() {
        // the last workgroup of the launch puts the role counters back and opens the next epoch (every workgroup has
        // read the counters by then: it counts itself done only behind its own role agreement)
        if (tid == 0) {
            if (q.sticky && __hip_atomic_load(err, NM_RLX_AGENT) != 0) __hip_atomic_store((gu32*)q.sticky, 1u, NM_RLX_AGENT);
            const unsigned d = __hip_atomic_fetch_add((gu32*)q.hdr + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (d == gridDim.x - 1) {
                for (int x = 0; x < 8; ++x) __hip_atomic_store((gu32*)q.hdr + 8 + x, 0u, NM_RLX_AGENT);
                __hip_atomic_store((gu32*)q.hdr + 1, 0u, NM_RLX_AGENT);
                __hip_atomic_store((gu32*)q.hdr + 2, 0u, NM_RLX_AGENT);
                __hip_atomic_store(err, 0u, NM_RLX_AGENT);
                __hip_atomic_store((gu32*)q.hdr + 3, epoch + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (!role.active) { finish(); return; }
    __builtin_amdgcn_s_setprio(3);
    const int jb = role.jb, row0 = role.cl * 16;
    const int k_wave = wave * 16 * NCH;
    const unsigned tag1 = 4u * epoch + 1u, tag2 = 4u * epoch + 2u;
    float* red1 = lds;                                   // [NW][2][4][64]
    float* red2 = red1 + (long)NW * 2 * 256;             // [NW][1][4][64]
    float* red3 = red2 + (long)NW * 256;                 // [NW][t3n][4][64]

    // this wave's K-slices of the weights it multiplies (requested now, used stage by stage)
    float wr[NCH][4], wu[NCH][4], wk[NCH][4], w3[DEC_T3_MAX][NCH][4];
    dec_load_w(q.wg_t, q.ld_wg, 16 * jb, 2 * H, k_wave, lane, wr);
    dec_load_w(q.wg_t, q.ld_wg, H + 16 * jb, 2 * H, k_wave, lane, wu);
    dec_load_w(q.wc_t, q.ld_wc, 16 * jb, H, k_wave, lane, wk);
#pragma unroll
    for (int i = 0; i < DEC_T3_MAX; ++i) {
        const int t3 = jb * t3n + i;                      // tile of [y | pre]
        if (i < t3n && t3 < nt3) {
            if (16 * t3 < q.A) dec_load_w(q.wq_t, q.ld_wq, 16 * t3, q.A, k_wave, lane, w3[i]);
            else dec_load_w(q.wo_t, q.ld_wo, 16 * t3 - q.A, q.O, k_wave, lane, w3[i]);
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) w3[i][c][j] = 0.0f;
        }
    }

    // epilogue threads: one element (row, col) of a 16x16 tile each, laid out as skinny16_tile maps it
    const bool epi = tid < 256;
    const int reg = (tid >> 6) & 3, ln = tid & 63;
    const int rloc = 4 * (ln >> 4) + reg, c16 = ln & 15;
    const int row = row0 + rloc, col = 16 * jb + c16;
    const bool mine = epi && row < R;
    const int rr = min(row, R - 1);
    const float* trow = q.table + (long)(mine ? q.ids[rr] : 0) * q.ld_table;
    float hreg = 0.0f, xr = 0.0f, xu = 0.0f, xc = 0.0f;
    if (mine) {
        hreg = q.h_in[(long)row * q.ld_h + col];
        xr = trow[col] + q.bg[col];
        xu = trow[H + col] + q.bg[H + col];
        xc = trow[2 * H + col];
    }
    u64* XA = q.xa + (long)role.cl * 16 * H;             // this cluster's granules of r*h ...
    u64* XB = q.xb + (long)role.cl * 16 * H;             // ... and of h'
    float a[4][4];
    CluWait cw;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    float ureg = 0.0f;
    // ---- stage 1: gates
    {
        clu_load_plain(q.h_in, q.ld_h, R, row0, k_wave, lane, a);
        f32x4 ar = zero, au = zero;
        clu_mma(ar, a, wr);
        clu_mma(au, a, wu);
        clu_put(red1, 2, wave, 0, lane, ar);
        clu_put(red1, 2, wave, 1, lane, au);
    }
    __syncthreads();
    if (epi) {
        const float sr = clu_get(red1, 2, NW, 0, reg, ln), su = clu_get(red1, 2, NW, 1, reg, ln);
        const float r = mine ? nm_sigmoid(xr + sr) : 0.0f;
        ureg = mine ? nm_sigmoid(xu + su) : 0.0f;
        clu_publish<RT, NCH>(XA, role.local, 0, rloc, col, tag1, mine ? r * hreg : 0.0f);
    }
    // ---- stage 2: candidate + blend
    clu_wait<RT, NCH>(XA, wave, lane, tag1, err, cw);
    {
        clu_gather<RT, NCH>(XA, wave, lane, 0, 0, tag1, err, cw, a);
        f32x4 ac = zero;
        clu_mma(ac, a, wk);
        clu_put(red2, 1, wave, 0, lane, ac);
    }
    __syncthreads();
    if (epi) {
        const float sc = clu_get(red2, 1, NW, 0, reg, ln);
        float hn = 0.0f;
        if (mine) {
            const float c = nm_tanh(xc + sc);
            hn = ureg * hreg + (1.0f - ureg) * c;
            q.h_out[(long)row * q.ld_ho + col] = hn;
            if (q.h_out2) q.h_out2[(long)row * q.ld_ho2 + col] = hn;
        }
        clu_publish<RT, NCH>(XB, role.local, 0, rloc, col, tag2, hn);
    }
    // ---- stage 3: attention query and the state part of the output projection
    clu_wait<RT, NCH>(XB, wave, lane, tag2, err, cw);
    {
        clu_gather<RT, NCH>(XB, wave, lane, 0, 0, tag2, err, cw, a);
#pragma unroll
        for (int i = 0; i < DEC_T3_MAX; ++i) {
            if (i < t3n) {
                f32x4 acc = zero;
                clu_mma(acc, a, w3[i]);
                clu_put(red3, t3n, wave, i, lane, acc);
            }
        }
    }
    __syncthreads();
    if (mine) {
        for (int i = 0; i < t3n; ++i) {
            const int t3 = jb * t3n + i;
            if (t3 >= nt3) break;
            const float s3 = clu_get(red3, t3n, NW, i, reg, ln);
            const int c3 = 16 * t3 + c16;
            if (c3 < q.A) q.y[(long)row * q.ld_y + c3] = s3 + (q.bq ? q.bq[c3] : 0.0f);
            else {
                const int co = c3 - q.A;
                q.pre[(long)row * q.ld_pre + co] = s3 + (q.bo ? q.bo[co] : 0.0f) + trow[3 * H + co];
            }
        }
    }
    finish();
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct CluShape {
    int RT, NW, nrb, grid;
    long granules;               // of one stage output of width H, all clusters
};

// Which (R, H, ndir) the cluster kernels take: H a multiple of 128 between 256 and 512 (H/64 waves of 64 k-values, an
// even number of them: the two halves of the backward loop's 2H-wide operand fall on whole waves; one epilogue thread
// per element: 256 RT <= 64 NW), clusters of 16 or 32 rows whose H/16 workgroups all fit ONE XCD's CUs, one workgroup
// per CU.
static bool clu_shape(long R, long H, int ndir, CluShape* s) {
    if (H < 256 || H > 512 || H % 128 != 0 || R < 1 || ndir < 1 || ndir > 2) return false;
    int ncu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) return false;
    const int nj = (int)(H / 16), nw = (int)(H / 64);
    for (int rt = 1; rt <= 2 && 256 * rt <= 64 * nw; ++rt) {
        const int nrb = (int)((R + 16 * rt - 1) / (16 * rt));
        const int ncl = ndir * nrb, cpx = (ncl + 7) / 8;
        if (cpx * nj <= ncu / 8) {
            s->RT = rt; s->NW = nw; s->nrb = nrb; s->grid = ncu;
            s->granules = (long)ncl * 16 * rt * H;
            return true;
        }
    }
    return false;
}

extern "C" int nm_gru_seq_supported(int64_t R, int64_t H, int32_t ndir) {
    CluShape s;
    return clu_shape(R, H, ndir, &s) ? 1 : 0;
}

// header + granules: forward two stage outputs of width H, backward one of width H and two copies of width 2H
extern "C" int64_t nm_gru_seq_workspace_bytes(int64_t R, int64_t H, int32_t ndir) {
    CluShape s;
    if (!clu_shape(R, H, ndir, &s)) return CLU_HDR_BYTES;
    return CLU_HDR_BYTES + s.granules * 8 * 5;
}

template <typename Kern>
static bool clu_prepare(Kern kern, size_t lds) {
    static std::atomic<unsigned> devs{0};
    const unsigned ok_bit = 1u << (nm_cur()->device & 15), bad_bit = ok_bit << 16;
    unsigned seen = devs.load(std::memory_order_relaxed);
    if (!(seen & (ok_bit | bad_bit))) {
        bool ok = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess;
        int per_cu = 0;
        ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, 64 * 1024) == hipSuccess && per_cu >= 1;
        if (!ok) (void)hipGetLastError();
        seen = devs.fetch_or(ok ? ok_bit : bad_bit, std::memory_order_relaxed) | (ok ? ok_bit : bad_bit);
    }
    return (seen & ok_bit) && lds <= 64 * 1024;
}

static std::atomic<int> clu_force_fail{0};

// Test hook: the next ``launches`` cluster loops (nm_gru_seq_fwd / nm_gru_seq_bwd, any stream of this process) raise
// their error word at once -- what a launch does after 0.2 s without progress when something else holds compute
// units.  Their results are garbage and the caller's sticky word is set: the recovery paths can be tested on a
// healthy device.  Returns the number of forced launches that were still pending.
extern "C" int nm_gru_seq_force_give_up(int32_t launches) {
    return clu_force_fail.exchange(launches < 0 ? 0 : launches, std::memory_order_relaxed);
}

// Test utility: ``blocks`` workgroups that each pin ``lds_bytes`` of LDS and 256 threads on a CU for ``microseconds``
// (sleeping, not spinning) -- what a long-running kernel of another stream or process does to a cluster loop: the
// CUs it sits on cannot take the loop's workgroups, the rest of the loop waits for them and gives up after 0.2 s.
__global__ __launch_bounds__(256) void clu_hog_kernel(long ticks, int* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lds[threadIdx.x] = 0.0f;
    const long t0 = (long)wall_clock64();
    while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (sink && lds[threadIdx.x] != 0.0f) *sink = 1;
}

extern "C" int nm_gru_seq_test_hog(void* stream, int32_t blocks, int64_t lds_bytes, int64_t microseconds) {
    NM_REQUIRE(blocks > 0 && blocks <= 4096 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024 && microseconds >= 0 &&
                   microseconds <= 5000000, "nm_gru_seq_test_hog: bad arguments");
    if (hipFuncSetAttribute((const void*)clu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) !=
        hipSuccess)
        NM_FAIL(NM_ERR_HIP, "nm_gru_seq_test_hog: cannot reserve %ld bytes of LDS", (long)lds_bytes);
    hipLaunchKernelGGL(clu_hog_kernel, dim3((unsigned)blocks), dim3(256), (size_t)lds_bytes, nm_stream(stream),
                       (long)(microseconds * 100), (int*)nullptr);           // wall_clock64 ticks at 100 MHz
    NM_LAUNCH_CHECK("nm_gru_seq_test_hog");
}

// Test utility: which XCD every workgroup of a launch of ``blocks`` x ``threads`` landed on (HW_REG_XCC_ID), written to
// xcc_out[blockIdx.x].  Two kernels rest their SPEED (never their results) on the dispatcher dealing consecutive
// workgroups to the XCDs round-robin: attn_whole_wide (blocks i and i + 8 share an L2) and gemm_tiled's tile order.
__global__ void xcc_probe_kernel(int* __restrict__ out) {
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = (int)(xcc & 7u);
    }
}

extern "C" int nm_test_xcc_ids(void* stream, int32_t* xcc_out, int32_t blocks, int32_t threads) {
    NM_REQUIRE(xcc_out && blocks > 0 && blocks <= 65536 && threads >= 64 && threads <= 1024 && threads % 64 == 0,
               "nm_test_xcc_ids: bad arguments");
    hipLaunchKernelGGL(xcc_probe_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), 0, nm_stream(stream), xcc_out);
    NM_LAUNCH_CHECK("nm_test_xcc_ids");
}

static void clu_fill(GruClu& q, const nm_gru_epilogue* e) {
    GruEpi& d = q.e;
    d.mode = 0; d.lengths = e->lengths; d.t = 0; d.rev_mask = e->rev_mask; d.H = (int)e->H; d.R = e->R;
    d.xp = e->xp; d.x_dir = e->x_dir; d.x_row = e->x_row; d.x_time = e->x_time;
    d.h_in = e->h_in; d.h_out = e->h_out; d.ru = e->ru; d.rh = e->rh; d.c_save = e->c_save;
    d.out = e->out; d.o_dir = e->o_dir; d.o_row = e->o_row; d.o_time = e->o_time;
    d.dh = e->dh; d.dout = e->dout; d.do_dir = e->do_dir; d.do_row = e->do_row; d.do_time = e->do_time;
    d.c = e->c; d.h0 = e->h0; d.hseq = e->hseq; d.hs_dir = e->hs_dir; d.hs_row = e->hs_row; d.hs_time = e->hs_time;
    d.dxp = e->dxp; d.dx_dir = e->dx_dir; d.dx_row = e->dx_row; d.dx_time = e->dx_time;
    d.dgpre = e->dgpre; d.dcpre = e->dcpre;
    q.ndir = e->ndir;
    {   // test hook: the placement-independent path (what runs when an XCD does not get its tickets)
        const char* place = getenv("NM_CLUSTER_PLACEMENT");
        q.force_global = (place && strcmp(place, "blockidx") == 0) ? 1 : 0;
    }
    q.dbg = nullptr;
    {   // test hook: the next nm_gru_seq_force_give_up(n) launches behave like launches whose hand-offs timed out
        int left = clu_force_fail.load(std::memory_order_relaxed);
        q.force_fail = 0;
        while (left > 0 && !clu_force_fail.compare_exchange_weak(left, left - 1, std::memory_order_relaxed)) { }
        if (left > 0) q.force_fail = 1;
    }
#ifdef NM_CLU_DEBUG          // timing probe of one workgroup (tools/clu_stamps.py): a device buffer of 64 longs
    if (getenv("NM_CLU_DEBUG_PTR")) q.dbg = reinterpret_cast<long*>(strtoull(getenv("NM_CLU_DEBUG_PTR"), nullptr, 0));
#endif
}

extern "C" int nm_gru_seq_fwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t h_step,
                              int64_t ru_step, int64_t rh_step, int64_t c_step, const float* wgh, int64_t ld_g,
                              int64_t stride_g, const float* wch, int64_t ld_c, int64_t stride_c,
                              void* workspace, int64_t workspace_bytes, uint32_t* sticky_error) {
    NM_REQUIRE(e && wgh && wch && workspace, "nm_gru_seq_fwd: null pointer / workspace");
    NM_REQUIRE(steps >= 0 && e->R > 0 && e->H > 0 && e->ndir >= 1 && e->ndir <= 2,
               "nm_gru_seq_fwd: bad shape R=%ld H=%ld", (long)e->R, (long)e->H);
    NM_REQUIRE(e->xp && e->h_in && e->h_out && e->ru, "nm_gru_seq_fwd: missing operand");
    NM_REQUIRE(nm_aligned16(e->h_in) && nm_aligned16(workspace), "nm_gru_seq_fwd: operands must be 16-byte aligned");
    CluShape s;
    NM_REQUIRE(clu_shape(e->R, e->H, e->ndir, &s), "nm_gru_seq_fwd: shape R=%ld H=%ld ndir=%d not supported "
               "(nm_gru_seq_supported)", (long)e->R, (long)e->H, (int)e->ndir);
    NM_REQUIRE(workspace_bytes >= nm_gru_seq_workspace_bytes(e->R, e->H, e->ndir), "nm_gru_seq_fwd: workspace too small");
    if (steps == 0) return NM_OK;
    GruClu q;
    clu_fill(q, e);
    q.steps = steps; q.nrb = s.nrb;
    q.h_step = h_step; q.ru_step = ru_step; q.rh_step = rh_step; q.c_step = c_step;
    q.wg = wgh; q.ldg = ld_g; q.sg = stride_g; q.wc = wch; q.ldc = ld_c; q.sc = stride_c;
    q.hdr = reinterpret_cast<unsigned*>(workspace);
    q.sticky = sticky_error;
    q.xa = reinterpret_cast<u64*>(reinterpret_cast<char*>(workspace) + CLU_HDR_BYTES);
    q.xb = q.xa + s.granules;
    hipStream_t st = nm_stream(stream);
    if (hipMemsetAsync(workspace, 0, CLU_HDR_BYTES + (size_t)s.granules * 8 * 2, st) != hipSuccess)
        NM_FAIL(NM_ERR_HIP, "nm_gru_seq_fwd: memset failed");
    const size_t lds = (size_t)s.NW * s.RT * 3 * 1024;
    bool ok;
    if (s.RT == 1) {
        ok = clu_prepare(gru_cluster_fwd_kernel<1>, lds);
        if (ok) hipLaunchKernelGGL((gru_cluster_fwd_kernel<1>), dim3(s.grid), dim3(s.NW * 64), lds, st, q);
    } else {
        ok = clu_prepare(gru_cluster_fwd_kernel<2>, lds);
        if (ok) hipLaunchKernelGGL((gru_cluster_fwd_kernel<2>), dim3(s.grid), dim3(s.NW * 64), lds, st, q);
    }
    if (!ok) NM_FAIL(NM_ERR_HIP, "nm_gru_seq_fwd: the kernel cannot be made resident on this device");
    NM_LAUNCH_CHECK("nm_gru_seq_fwd");
}

// The whole BPTT loop: e->dh holds dL/dh after the last step on entry and dL/dh_0 on exit; step t reads ru + t*ru_step,
// c + t*c_step, h_prev through hseq / h0, dout at the step's position, and writes the three pre-activation gradients
// of the step's position into dxp.
extern "C" int nm_gru_seq_bwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t ru_step, int64_t c_step,
                              const float* wgh, int64_t ld_g, int64_t stride_g, const float* wch, int64_t ld_c,
                              int64_t stride_c, void* workspace, int64_t workspace_bytes, uint32_t* sticky_error) {
    NM_REQUIRE(e && wgh && wch && workspace, "nm_gru_seq_bwd: null pointer / workspace");
    NM_REQUIRE(steps >= 0 && e->R > 0 && e->H > 0 && e->ndir >= 1 && e->ndir <= 2,
               "nm_gru_seq_bwd: bad shape R=%ld H=%ld", (long)e->R, (long)e->H);
    NM_REQUIRE(e->dh && e->ru && e->c && e->hseq && e->dxp, "nm_gru_seq_bwd: missing operand");
    NM_REQUIRE(nm_aligned16(wgh) && nm_aligned16(wch) && ld_g % 4 == 0 && ld_c % 4 == 0 && stride_g % 4 == 0 &&
                   stride_c % 4 == 0 && nm_aligned16(workspace),
               "nm_gru_seq_bwd: kernels must be 16-byte aligned with leading dimensions %% 4 == 0");
    CluShape s;
    NM_REQUIRE(clu_shape(e->R, e->H, e->ndir, &s), "nm_gru_seq_bwd: shape R=%ld H=%ld ndir=%d not supported "
               "(nm_gru_seq_supported)", (long)e->R, (long)e->H, (int)e->ndir);
    NM_REQUIRE(workspace_bytes >= nm_gru_seq_workspace_bytes(e->R, e->H, e->ndir), "nm_gru_seq_bwd: workspace too small");
    if (steps == 0) return NM_OK;
    GruClu q;
    clu_fill(q, e);
    q.steps = steps; q.nrb = s.nrb;
    q.h_step = 0; q.ru_step = ru_step; q.rh_step = 0; q.c_step = c_step;
    q.wg = wgh; q.ldg = ld_g; q.sg = stride_g; q.wc = wch; q.ldc = ld_c; q.sc = stride_c;
    q.hdr = reinterpret_cast<unsigned*>(workspace);
    q.sticky = sticky_error;
    q.xa = reinterpret_cast<u64*>(reinterpret_cast<char*>(workspace) + CLU_HDR_BYTES);
    q.xb = q.xa + s.granules;
    hipStream_t st = nm_stream(stream);
    if (hipMemsetAsync(workspace, 0, CLU_HDR_BYTES + (size_t)s.granules * 8 * 5, st) != hipSuccess)
        NM_FAIL(NM_ERR_HIP, "nm_gru_seq_bwd: memset failed");
    const size_t lds = (size_t)s.NW * s.RT * 2 * 1024;
    bool ok;
    if (s.RT == 1) {
        ok = clu_prepare(gru_cluster_bwd_kernel<1>, lds);
        if (ok) hipLaunchKernelGGL((gru_cluster_bwd_kernel<1>), dim3(s.grid), dim3(s.NW * 64), lds, st, q);
    } else {
        ok = clu_prepare(gru_cluster_bwd_kernel<2>, lds);
        if (ok) hipLaunchKernelGGL((gru_cluster_bwd_kernel<2>), dim3(s.grid), dim3(s.NW * 64), lds, st, q);
    }
    if (!ok) NM_FAIL(NM_ERR_HIP, "nm_gru_seq_bwd: the kernel cannot be made resident on this device");
    NM_LAUNCH_CHECK("nm_gru_seq_bwd");
}

// 1 when a cluster loop that used ``workspace`` gave up waiting (its results are garbage); reads 4 bytes back, so
// the caller synchronises first
extern "C" int nm_gru_seq_failed(const void* workspace) {
    unsigned flag = 0;
    if (!workspace || hipMemcpy(&flag, workspace, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return flag != 0 ? 1 : 0;
}

static bool dec_clu_shape(long R, long H, long A, long O, int* ncl_out, int* grid_out) {
    if (H < 256 || H > 512 || H % 128 != 0 || R < 1 || A % 16 || O % 16 || A < 16 || O < 16) return false;
    int ncu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) return false;
    const int nj = (int)(H / 16), ncl = (int)((R + 15) / 16), cpx = (ncl + 7) / 8;
    if (cpx * nj > ncu / 8) return false;                                   // every cluster on ONE XCD
    const long nt3 = (A + O) / 16;
    if ((nt3 + nj - 1) / nj > DEC_T3_MAX) return false;
    *ncl_out = ncl;
    *grid_out = ncu;
    return true;
}

extern "C" int nm_dec_step_cluster_supported(int64_t R, int64_t H, int64_t A, int64_t O) {
    int ncl, grid;
    return dec_clu_shape(R, H, A, O, &ncl, &grid) ? 1 : 0;
}

// header + the granules of r*h and h' (8 bytes per value).  The workspace must be ZERO when it is first used (tags
// and counters start there) and is never cleared again.
extern "C" int64_t nm_dec_step_cluster_workspace_bytes(int64_t R, int64_t H) {
    const int64_t ncl = (R + 15) / 16;
    return CLU_HDR_BYTES + ncl * 16 * H * 8 * 2;
}

// internal: groups 1-3 of nm_decoder_step_fused (input tables) as one launch; false when the shape is not taken
bool nm_dec_step_cluster_try(hipStream_t st, int64_t R, int64_t H, int64_t A, int64_t O, const float* h_in, int64_t ld_h,
                             const float* table, int64_t ld_table, const int32_t* ids, const float* bg, const float* bq,
                             const float* bo, const float* wg_t, int64_t ld_wg, const float* wc_t, int64_t ld_wc,
                             const float* wq_t, int64_t ld_wq, const float* wo_t, int64_t ld_wo, float* h_out,
                             int64_t ld_ho, float* h_out2, int64_t ld_ho2, float* y, int64_t ld_y, float* pre,
                             int64_t ld_pre, void* workspace, int64_t workspace_bytes, uint32_t* sticky) {
    int ncl = 0, grid = 0;
    if (!workspace || !dec_clu_shape(R, H, A, O, &ncl, &grid)) return false;
    if (workspace_bytes < nm_dec_step_cluster_workspace_bytes(R, H)) return false;
    if (!nm_aligned16(h_in) || ld_h % 4 || !nm_aligned16(wg_t) || ld_wg % 4 || !nm_aligned16(wc_t) || ld_wc % 4 ||
        !nm_aligned16(wq_t) || ld_wq % 4 || !nm_aligned16(wo_t) || ld_wo % 4 || !nm_aligned16(workspace) || !bg)
        return false;
    const int nw = (int)(H / 64), nj = (int)(H / 16);
    const int t3n = (int)(((A + O) / 16 + nj - 1) / nj);
    const size_t lds = (size_t)nw * (2 + 1 + t3n) * 1024 + 64;
    if (!clu_prepare(dec_step_cluster_kernel, lds)) return false;
    DecClu q;
    q.R = (int)R; q.H = (int)H; q.A = (int)A; q.O = (int)O; q.ncl = ncl;
    q.h_in = h_in; q.ld_h = ld_h; q.table = table; q.ld_table = ld_table; q.ids = ids;
    q.bg = bg; q.bq = bq; q.bo = bo;
    q.wg_t = wg_t; q.ld_wg = ld_wg; q.wc_t = wc_t; q.ld_wc = ld_wc; q.wq_t = wq_t; q.ld_wq = ld_wq; q.wo_t = wo_t; q.ld_wo = ld_wo;
    q.h_out = h_out; q.ld_ho = ld_ho; q.h_out2 = h_out2; q.ld_ho2 = ld_ho2; q.y = y; q.ld_y = ld_y; q.pre = pre; q.ld_pre = ld_pre;
    q.hdr = reinterpret_cast<unsigned*>(workspace);
    q.xa = reinterpret_cast<u64*>(reinterpret_cast<char*>(workspace) + CLU_HDR_BYTES);
    q.xb = q.xa + (long)ncl * 16 * H;
    q.sticky = sticky;
    {
        const char* place = getenv("NM_CLUSTER_PLACEMENT");
        q.force_global = (place && strcmp(place, "blockidx") == 0) ? 1 : 0;
        int left = clu_force_fail.load(std::memory_order_relaxed);
        q.force_fail = 0;
        while (left > 0 && !clu_force_fail.compare_exchange_weak(left, left - 1, std::memory_order_relaxed)) { }
        if (left > 0) q.force_fail = 1;
    }
    hipLaunchKernelGGL(dec_step_cluster_kernel, dim3(grid), dim3(nw * 64), lds, st, q);
    return hipGetLastError() == hipSuccess;
}

// ---- NematusGRU loops: host side ------------------------------------------------------------------------------------
// workspace: header + two buffers of width 3H (the backward loop's; the forward loop uses the first H columns' worth)
extern "C" int64_t nm_nematus_seq_workspace_bytes(int64_t R, int64_t H, int32_t ndir) {
    CluShape s;
    if (!clu_shape(R, H, ndir, &s)) return CLU_HDR_BYTES;
    return CLU_HDR_BYTES + s.granules * 8 * 6;
}

static int nem_launch(bool backward, void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t h_step, int64_t ru_step,
                      int64_t sc_step, int64_t c_step, const float* ug, int64_t ld_g, int64_t stride_g, const float* uc,
                      int64_t ld_c, int64_t stride_c, const float* bgs, const float* bcs, void* workspace,
                      int64_t workspace_bytes, uint32_t* sticky_error, const char* who) {
    NM_REQUIRE(e && ug && uc && workspace, "%s: null pointer / workspace", who);
    NM_REQUIRE(steps >= 0 && e->R > 0 && e->H > 0 && e->ndir >= 1 && e->ndir <= 2, "%s: bad shape R=%ld H=%ld", who,
               (long)e->R, (long)e->H);
    NM_REQUIRE(nm_aligned16(workspace) && nm_aligned16(ug) && nm_aligned16(uc) && ld_g % 4 == 0 && ld_c % 4 == 0 &&
                   stride_g % 4 == 0 && stride_c % 4 == 0, "%s: kernels / workspace must be 16-byte aligned", who);
    if (backward) NM_REQUIRE(e->dh && e->ru && e->c && e->rh && e->hseq && e->dxp, "%s: missing operand", who);
    else {
        NM_REQUIRE(e->xp && e->h_in && e->h_out && e->ru && e->c_save && e->rh && nm_aligned16(e->h_in),
                   "%s: missing / unaligned operand", who);
        // (a step here is ONE stage: a workgroup may finish step 0 -- and write h_out -- while another still loads h_in)
        NM_REQUIRE(e->h_in != e->h_out, "%s: h_in and h_out must be different buffers", who);
    }
    CluShape s;
    NM_REQUIRE(clu_shape(e->R, e->H, e->ndir, &s), "%s: shape R=%ld H=%ld ndir=%d not supported (nm_gru_seq_supported)",
               who, (long)e->R, (long)e->H, (int)e->ndir);
    NM_REQUIRE(workspace_bytes >= nm_nematus_seq_workspace_bytes(e->R, e->H, e->ndir), "%s: workspace too small", who);
    if (steps == 0) return NM_OK;
    NemClu a;
    GruClu& q = a.q;
    clu_fill(q, e);
    q.e.rh = e->rh;
    q.steps = steps; q.nrb = s.nrb;
    q.h_step = h_step; q.ru_step = ru_step; q.rh_step = sc_step; q.c_step = c_step;
    q.wg = ug; q.ldg = ld_g; q.sg = stride_g; q.wc = uc; q.ldc = ld_c; q.sc = stride_c;
    q.hdr = reinterpret_cast<unsigned*>(workspace);
    q.sticky = sticky_error;
    q.xa = reinterpret_cast<u64*>(reinterpret_cast<char*>(workspace) + CLU_HDR_BYTES);
    q.xb = q.xa + s.granules * 3;
    a.bgs = bgs; a.bcs = bcs;
    hipStream_t st = nm_stream(stream);
    if (hipMemsetAsync(workspace, 0, CLU_HDR_BYTES + (size_t)s.granules * 8 * 6, st) != hipSuccess)
        NM_FAIL(NM_ERR_HIP, "%s: memset failed", who);
    const size_t lds = (size_t)s.NW * s.RT * (backward ? 1 : 3) * 1024;
    bool ok;
    if (backward) {
        if (s.RT == 1) { ok = clu_prepare(nematus_cluster_bwd_kernel<1>, lds); if (ok) hipLaunchKernelGGL((nematus_cluster_bwd_kernel<1>), dim3(s.grid), dim3(s.NW * 64), lds, st, a); }
        else { ok = clu_prepare(nematus_cluster_bwd_kernel<2>, lds); if (ok) hipLaunchKernelGGL((nematus_cluster_bwd_kernel<2>), dim3(s.grid), dim3(s.NW * 64), lds, st, a); }
    } else {
        if (s.RT == 1) { ok = clu_prepare(nematus_cluster_fwd_kernel<1>, lds); if (ok) hipLaunchKernelGGL((nematus_cluster_fwd_kernel<1>), dim3(s.grid), dim3(s.NW * 64), lds, st, a); }
        else { ok = clu_prepare(nematus_cluster_fwd_kernel<2>, lds); if (ok) hipLaunchKernelGGL((nematus_cluster_fwd_kernel<2>), dim3(s.grid), dim3(s.NW * 64), lds, st, a); }
    }
    if (!ok) NM_FAIL(NM_ERR_HIP, "%s: the kernel cannot be made resident on this device", who);
    NM_LAUNCH_CHECK(who);
}

// e->rh: sc of every step ([steps] x sc_step); ug [ndir][H][2H], uc [ndir][H][H]: the STATE projections; bgs / bcs: their
// optional biases.  Everything else as nm_gru_seq_fwd.
extern "C" int nm_nematus_seq_fwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t h_step, int64_t ru_step,
                                  int64_t sc_step, int64_t c_step, const float* ug, int64_t ld_g, int64_t stride_g,
                                  const float* uc, int64_t ld_c, int64_t stride_c, const float* bgs, const float* bcs,
                                  void* workspace, int64_t workspace_bytes, uint32_t* sticky_error) {
    return nem_launch(false, stream, e, steps, h_step, ru_step, sc_step, c_step, ug, ld_g, stride_g, uc, ld_c, stride_c,
                      bgs, bcs, workspace, workspace_bytes, sticky_error, "nm_nematus_seq_fwd");
}

// The whole BPTT loop: e->dh in / out as nm_gru_seq_bwd; e->rh = the saved sc; dxp 4H wide per direction:
// [dr' | du' | dc' | dsc] at the step's sequence position.
extern "C" int nm_nematus_seq_bwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t ru_step, int64_t sc_step,
                                  int64_t c_step, const float* ug, int64_t ld_g, int64_t stride_g, const float* uc,
                                  int64_t ld_c, int64_t stride_c, void* workspace, int64_t workspace_bytes,
                                  uint32_t* sticky_error) {
    return nem_launch(true, stream, e, steps, 0, ru_step, sc_step, c_step, ug, ld_g, stride_g, uc, ld_c, stride_c,
                      nullptr, nullptr, workspace, workspace_bytes, sticky_error, "nm_nematus_seq_bwd");
}

// ---- LSTM loops: host side ------------------------------------------------------------------------------------------
// workspace: header + two buffers of width 4H (the backward loop's; the forward loop's two of width H fit inside)
extern "C" int64_t nm_lstm_seq_workspace_bytes(int64_t R, int64_t H, int32_t ndir) {
    CluShape s;
    if (!clu_shape(R, H, ndir, &s)) return CLU_HDR_BYTES;
    return CLU_HDR_BYTES + s.granules * 8 * 8;
}

static int lstm_launch(bool backward, void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t h_step, int64_t g_step,
                       int64_t c_step, const float* wh, int64_t ld_w, int64_t stride_w, float forget_bias, void* workspace,
                       int64_t workspace_bytes, uint32_t* sticky_error, const char* who) {
    NM_REQUIRE(e && wh && workspace, "%s: null pointer / workspace", who);
    NM_REQUIRE(steps >= 0 && e->R > 0 && e->H > 0 && e->ndir >= 1 && e->ndir <= 2, "%s: bad shape R=%ld H=%ld", who,
               (long)e->R, (long)e->H);
    NM_REQUIRE(nm_aligned16(workspace) && nm_aligned16(wh) && ld_w % 4 == 0 && stride_w % 4 == 0,
               "%s: kernel / workspace must be 16-byte aligned", who);
    if (backward) NM_REQUIRE(e->dh && e->ru && e->c && e->dxp, "%s: missing operand", who);
    else {
        NM_REQUIRE(e->xp && e->h_in && e->h_out && e->ru && e->c_save && nm_aligned16(e->h_in),
                   "%s: missing / unaligned operand", who);
        NM_REQUIRE(e->h_in != e->h_out, "%s: h_in and h_out must be different buffers", who);
    }
    CluShape s;
    NM_REQUIRE(clu_shape(e->R, e->H, e->ndir, &s), "%s: shape R=%ld H=%ld ndir=%d not supported (nm_gru_seq_supported)",
               who, (long)e->R, (long)e->H, (int)e->ndir);
    NM_REQUIRE(workspace_bytes >= nm_lstm_seq_workspace_bytes(e->R, e->H, e->ndir), "%s: workspace too small", who);
    if (steps == 0) return NM_OK;
    LstmClu a;
    GruClu& q = a.q;
    clu_fill(q, e);
    q.steps = steps; q.nrb = s.nrb;
    q.h_step = h_step; q.ru_step = g_step; q.rh_step = 0; q.c_step = c_step;
    q.wg = wh; q.ldg = ld_w; q.sg = stride_w; q.wc = nullptr; q.ldc = 0; q.sc = 0;
    q.hdr = reinterpret_cast<unsigned*>(workspace);
    q.sticky = sticky_error;
    q.xa = reinterpret_cast<u64*>(reinterpret_cast<char*>(workspace) + CLU_HDR_BYTES);
    q.xb = q.xa + s.granules * 4;
    a.forget_bias = forget_bias;
    hipStream_t st = nm_stream(stream);
    if (hipMemsetAsync(workspace, 0, CLU_HDR_BYTES + (size_t)s.granules * 8 * 8, st) != hipSuccess)
        NM_FAIL(NM_ERR_HIP, "%s: memset failed", who);
    const size_t lds = (size_t)s.NW * s.RT * (backward ? 1 : 4) * 1024;
    bool ok;
    if (backward) {
        if (s.RT == 1) { ok = clu_prepare(lstm_cluster_bwd_kernel<1>, lds); if (ok) hipLaunchKernelGGL((lstm_cluster_bwd_kernel<1>), dim3(s.grid), dim3(s.NW * 64), lds, st, a); }
        else { ok = clu_prepare(lstm_cluster_bwd_kernel<2>, lds); if (ok) hipLaunchKernelGGL((lstm_cluster_bwd_kernel<2>), dim3(s.grid), dim3(s.NW * 64), lds, st, a); }
    } else {
        if (s.RT == 1) { ok = clu_prepare(lstm_cluster_fwd_kernel<1>, lds); if (ok) hipLaunchKernelGGL((lstm_cluster_fwd_kernel<1>), dim3(s.grid), dim3(s.NW * 64), lds, st, a); }
        else { ok = clu_prepare(lstm_cluster_fwd_kernel<2>, lds); if (ok) hipLaunchKernelGGL((lstm_cluster_fwd_kernel<2>), dim3(s.grid), dim3(s.NW * 64), lds, st, a); }
    }
    if (!ok) NM_FAIL(NM_ERR_HIP, "%s: the kernel cannot be made resident on this device", who);
    NM_LAUNCH_CHECK(who);
}

// e->ru: the activated gates [i | j | f | o] of every step ([steps] x g_step, 4H wide); e->c_save: c' of every step; xp 4H
// wide per direction; wh [ndir][H][4H]: the state half of the LSTM kernel; zero initial cell state.
extern "C" int nm_lstm_seq_fwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t h_step, int64_t g_step,
                               int64_t c_step, const float* wh, int64_t ld_w, int64_t stride_w, float forget_bias,
                               void* workspace, int64_t workspace_bytes, uint32_t* sticky_error) {
    return lstm_launch(false, stream, e, steps, h_step, g_step, c_step, wh, ld_w, stride_w, forget_bias, workspace,
                       workspace_bytes, sticky_error, "nm_lstm_seq_fwd");
}

// BPTT: e->dh in / out as nm_gru_seq_bwd; e->ru / e->c the saved gates and cell states; dxp 4H wide per direction.
extern "C" int nm_lstm_seq_bwd(void* stream, const nm_gru_epilogue* e, int32_t steps, int64_t g_step, int64_t c_step,
                               const float* wh, int64_t ld_w, int64_t stride_w, void* workspace, int64_t workspace_bytes,
                               uint32_t* sticky_error) {
    return lstm_launch(true, stream, e, steps, 0, g_step, c_step, wh, ld_w, stride_w, 0.0f, workspace, workspace_bytes,
                       sticky_error, "nm_lstm_seq_bwd");
}
