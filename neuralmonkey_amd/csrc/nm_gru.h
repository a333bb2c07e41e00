// Operands of one GRU time step as the fused epilogues see them (nm_gemm.hip: the per-step launches;
// nm_gru_cluster.hip: the whole time loop in one launch).  TF GRUCell (nn/ortho_gru_cell.py:44-53) under
// dynamic_rnn's length masking and reverse_sequence (encoders/recurrent.py:86-102).
#pragma once
#include "nm_common.h"

struct GruEpi {
    int mode;
    const int* lengths;
    int t, rev_mask, H;
    long R;
    // forward
    const float* xp; long x_dir, x_row, x_time;
    const float* h_in; float* h_out; float* ru; float* rh; float* c_save;
    float* out; long o_dir, o_row, o_time;
    // backward
    float* dh; const float* dout; long do_dir, do_row, do_time;
    const float* c; const float* h0; const float* hseq; long hs_dir, hs_row, hs_time;
    float* dxp; long dx_dir, dx_row, dx_time;
    float* dgpre; float* dcpre;
};

__device__ __forceinline__ bool gru_epi_pos(const GruEpi& e, int r, int d, int t, int& pos, int& ppos) {
    pos = t;
    const bool rev = (e.rev_mask >> d) & 1;
    if (e.lengths) {
        const int len = e.lengths[r];
        if (t >= len) return false;
        if (rev) pos = len - 1 - t;
    }
    ppos = rev ? pos + 1 : pos - 1;
    return true;
}

__device__ __forceinline__ float gru_epi_hprev(const GruEpi& e, long ro, int d, int r, int t, int ppos,
                                               int col) {
    if (t == 0) return e.h0 ? e.h0[ro * e.H + col] : 0.0f;
    return e.hseq[d * e.hs_dir + (long)r * e.hs_row + (long)ppos * e.hs_time + col];
}

struct nm_gru_epilogue {          // mirrors include/nmhip.h
    int32_t mode, t, rev_mask, ndir;
    int64_t R, H;
    const int32_t* lengths;
    const float* xp; int64_t x_dir, x_row, x_time;
    const float* h_in; float* h_out; float* ru; float* rh; float* c_save;
    float* out; int64_t o_dir, o_row, o_time;
    float* dh; const float* dout; int64_t do_dir, do_row, do_time;
    const float* c; const float* h0; const float* hseq; int64_t hs_dir, hs_row, hs_time;
    float* dxp; int64_t dx_dir, dx_row, dx_time;
    float* dgpre; float* dcpre;
};
