// Bidirectional single-layer LSTM over ragged sequences: the video / text encoders of
// /root/reference/video_nmn/module_net.py:39-47, 151-163 (nn.LSTM, gates i,f,g,o, zero state).
//
// Two stages:
//   1. input projection for all time steps and both directions as one fp32 MFMA GEMM each
//      (csrc/gemm.hip): xproj[row, dir*4Hh + g] = x[row] . W_ih[dir][g] + b_ih[dir][g] + b_hh[dir][g]
//      -- the one genuinely dense contraction of the path (537 of ~860 MFLOP per question).
//   2. a persistent recurrent kernel: one workgroup owns 16 sequences of one direction for all
//      time steps.  h lives in LDS (double buffered, one barrier per step), c in registers.
//      gates^T tile = h[16 seq, Hh] x W_hh^T via v_mfma_f32_16x16x4_f32; wave w owns hidden units
//      [16*NCT*w, 16*NCT*(w+1)) for all four gates, so the cell update is lane-local in the MFMA
//      C/D layout (col = unit, row = sequence).  W_hh is streamed from L2 every step (1 MB per
//      direction at Hh=256 does not fit LDS in fp32) with float4 loads along k; the k index inside
//      a 16-wide block is permuted identically for h and W_hh (lane group g owns k = 16b+4g..+3).
//      The xproj reads of a step are issued before its MFMA chain and consumed after it, which
//      hides their HBM latency behind the matrix work.
#include <cstdlib>

#include "common.h"

namespace stair {


struct LstmRecParams {
    const float *xproj;      // [rows, 8Hh]
    const float *w_pack;     // [2][Hh/32][4*Hh/16][2][64][4] W_hh in MFMA-fragment order (see whh_pack_kernel)
    const int32_t *seq_off;  // [n+1]
    const int32_t *seq_len;  // optional [n]: sequence s holds seq_len[s] rows from seq_off[s] (padded storage); null: off[s+1] - off[s]
    float *out;              // [rows, ldo]
    int64_t ldo;
    float *h_n;              // [n, 2Hh]
    float *cbuf;             // [rows, 2Hh] cell states, training mode only (else null)
    int n, Hh;
};

__global__ void bias_sum_kernel(const float *a0, const float *b0, const float *a1, const float *b1, float *out, int n4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = a0[i] + b0[i];
    else if (i < 2 * n4) out[i] = a1[i - n4] + b1[i - n4];
}

// Re-lay W_hh [4Hh, Hh] (both directions) into the order the recurrent kernel's B fragments are read:
//   pack[dir][kb][tile][half][lane][j] = W_hh[dir][tile*16 + (lane&15)][32*kb + 8*(lane>>4) + 4*half + j]
// so that one wave-wide 16-byte load is 1 KiB contiguous.  In W_hh's own layout the 16 lanes that form
// an MFMA column group sit in 16 different rows (1 KiB apart): every load instruction then touches 64
// scattered 16-B pieces and the texture path, not L2, bounds the recurrence (measured: MFMA pipe 40 % busy).
__global__ void whh_pack_kernel(const float *w0, const float *w1, float *pack, int Hh) {
    const int64_t per_dir = 4 * (int64_t)Hh * Hh;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 per thread
    if (i >= 2 * per_dir / 4) return;
    const int64_t e = i * 4;
    const int dir = (int)(e / per_dir);
    int64_t r = e - dir * per_dir;
    const int lane = (int)((r >> 2) & 63);
    r >>= 8;
    const int half = (int)(r & 1);
    r >>= 1;
    const int ntile = 4 * Hh / 16;
    const int tile = (int)(r % ntile), kb = (int)(r / ntile);
    const float *w = dir == 0 ? w0 : w1;
    const float4 v = *reinterpret_cast<const float4 *>(w + (int64_t)(tile * 16 + (lane & 15)) * Hh + 32 * kb + 8 * (lane >> 4) + 4 * half);
    *reinterpret_cast<float4 *>(pack + e) = v;
}

using v4f = __attribute__((ext_vector_type(4))) float;
using gv4p = const __attribute__((address_space(1))) v4f *;
using gfp = const __attribute__((address_space(1))) float *;

// One step of the recurrence per loop trip.  The W_hh fragments of a 32-wide k block are split in
// two groups (gates i,f | g,o per owned tile column); while the MFMAs of one group run, the loads
// of the other group (or of the next k block) are in flight, so the L2 latency of the weight
// stream is hidden behind matrix work instead of being paid 2*Hh/32 times per step.  Every lane
// reads two adjacent float4 of a W_hh row per k block, so the four lane groups of a row cover one
// full 128-B line (k permutation inside the block: lane group g owns k = 32b+8g..32b+8g+7).
template <int NCT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void lstm_rec_kernel(LstmRecParams p) {
    extern __shared__ __attribute__((aligned(16))) float hbuf[];  // [2][16][Hh+4]
    constexpr int NT = NCT * 4;      // MFMA tiles per wave: tile ti -> column block ti/4, gate ti%4
    constexpr int GS = NT / 2;       // tiles per pipeline group
    const int Hh = p.Hh, ldh = Hh + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & 15, g4 = lane >> 4;
    const int dir = blockIdx.y;
    const int s0 = blockIdx.x * 16;
    const int ntiles = Hh >> 4, nkb = Hh >> 5;
    const int ntile4 = 4 * ntiles;                               // fragment tiles per k block (4 gates)
    gfp whh = (gfp)p.w_pack + (int64_t)dir * 4 * Hh * Hh + lane * 4;
    gfp xproj = (gfp)p.xproj;
    __attribute__((address_space(1))) float *outp = (__attribute__((address_space(1))) float *)p.out;

    // my four sequences (accumulator rows 4*g4 + e)
    int off4[4], len4[4];
    int lmax = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int s = s0 + g4 * 4 + e;
        off4[e] = 0;
        len4[e] = 0;
        if (s < p.n) {
            off4[e] = p.seq_off[s];
            len4[e] = p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - off4[e];
        }
    }
    for (int s = s0; s < min(s0 + 16, p.n); ++s) lmax = max(lmax, p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - p.seq_off[s]);

    for (int i = tid; i < 2 * 16 * ldh; i += NWAVES * 64) hbuf[i] = 0.0f;
    float creg[NCT][4], hreg[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) creg[ct][e] = hreg[ct][e] = 0.0f;

    // tile columns owned by this wave; waves beyond the tile count idle but keep the barriers
    bool own[NCT];
    int unit[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int tile = wave * NCT + ct;
        own[ct] = tile < ntiles;
        unit[ct] = (own[ct] ? tile : 0) * 16 + cl;
    }
    // offsets (floats) of my B fragments inside a k block of the packed image: tile = gate*ntiles + column block
    int woff[NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        const int colblk = own[ti >> 2] ? wave * NCT + (ti >> 2) : 0;
        woff[ti] = ((ti & 3) * ntiles + colblk) * 512;
    }

    const int64_t ldx = 8 * (int64_t)Hh;
    const int xcol = dir * 4 * Hh;

    v4f bq[2][GS][2];
#define STAIR_LOADB(grp, kb)                                                      \
    _Pragma("unroll") for (int t_ = 0; t_ < GS; ++t_) {                            \
        gfp src_ = whh + (int64_t)(kb) * ntile4 * 512 + woff[(grp) * GS + t_];     \
        bq[grp][t_][0] = *(gv4p)(src_);                                            \
        bq[grp][t_][1] = *(gv4p)(src_ + 256);                                      \
    }
#define STAIR_MFMA(grp)                                                                            \
    _Pragma("unroll") for (int jj_ = 0; jj_ < 8; ++jj_)                                             \
        _Pragma("unroll") for (int t_ = 0; t_ < GS; ++t_)                                           \
            acc[(grp) * GS + t_] = __builtin_amdgcn_mfma_f32_16x16x4f32(                            \
                acur[jj_ >> 2][jj_ & 3], bq[grp][t_][jj_ >> 2][jj_ & 3], acc[(grp) * GS + t_], 0, 0, 0);

    // xproj values of the CURRENT step live in xp; they are re-loaded for the next step inside the
    // cell update, right after their last use, so their HBM latency overlaps the update + barrier.
    float xp[NCT][4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool act = 0 < len4[e];
        const int64_t row = off4[e] + (act ? (dir == 0 ? 0 : len4[e] - 1) : 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int gate = 0; gate < 4; ++gate)
                xp[ct][gate][e] = (act && own[ct]) ? xproj[row * ldx + xcol + gate * Hh + unit[ct]] : 0.0f;
    }
    STAIR_LOADB(0, 0);
    __syncthreads();

    for (int tau = 0; tau < lmax; ++tau) {
        const int cur = tau & 1;
        const float *hc = hbuf + cur * 16 * ldh;
        float *hn = hbuf + (cur ^ 1) * 16 * ldh;

        v4f acc[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) acc[ti] = v4f{0.f, 0.f, 0.f, 0.f};

        v4f acur[2], anxt[2];
        acur[0] = *reinterpret_cast<const v4f *>(hc + cl * ldh + 8 * g4);
        acur[1] = *reinterpret_cast<const v4f *>(hc + cl * ldh + 8 * g4 + 4);
        for (int kb = 0; kb < nkb; ++kb) {
            STAIR_LOADB(1, kb);
            __builtin_amdgcn_sched_barrier(0);
            STAIR_MFMA(0);
            const int kn = kb + 1 < nkb ? kb + 1 : 0;       // wraps to the next step's first block
            STAIR_LOADB(0, kn);
            anxt[0] = *reinterpret_cast<const v4f *>(hc + cl * ldh + kn * 32 + 8 * g4);
            anxt[1] = *reinterpret_cast<const v4f *>(hc + cl * ldh + kn * 32 + 8 * g4 + 4);
            __builtin_amdgcn_sched_barrier(0);
            STAIR_MFMA(1);
            acur[0] = anxt[0];
            acur[1] = anxt[1];
        }

#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool active = tau < len4[e];
            const int t = dir == 0 ? tau : len4[e] - 1 - tau;
            const int64_t row = off4[e] + (active ? t : 0);
            const bool act_n = tau + 1 < len4[e];
            const int64_t row_n = off4[e] + (act_n ? (dir == 0 ? tau + 1 : len4[e] - 2 - tau) : 0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (!own[ct]) continue;
                const float gi = acc[ct * 4 + 0][e] + xp[ct][0][e];
                const float gf = acc[ct * 4 + 1][e] + xp[ct][1][e];
                const float gg = acc[ct * 4 + 2][e] + xp[ct][2][e];
                const float go = acc[ct * 4 + 3][e] + xp[ct][3][e];
                const float si = sigmoid_fast(gi), sf = sigmoid_fast(gf), tg = tanh_fast(gg), so = sigmoid_fast(go);
                const float cn = sf * creg[ct][e] + si * tg;
                const float hv = so * tanh_fast(cn);
                if (active) {
                    creg[ct][e] = cn;
                    hreg[ct][e] = hv;
                    outp[row * p.ldo + dir * Hh + unit[ct]] = hv;
                    if (p.cbuf) {   // training: activated gates replace this row's xproj (already consumed), c saved
                        float *gsave = const_cast<float *>(p.xproj) + row * ldx + xcol + unit[ct];
                        gsave[0] = si; gsave[Hh] = sf; gsave[2 * Hh] = tg; gsave[3 * Hh] = so;
                        p.cbuf[row * 2 * Hh + dir * Hh + unit[ct]] = cn;
                    }
                }
                hn[(g4 * 4 + e) * ldh + unit[ct]] = hreg[ct][e];
#pragma unroll
                for (int gate = 0; gate < 4; ++gate)
                    xp[ct][gate][e] = act_n ? xproj[row_n * ldx + xcol + gate * Hh + unit[ct]] : 0.0f;
            }
        }
        __syncthreads();
    }
#undef STAIR_LOADB
#undef STAIR_MFMA

#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        if (!own[ct]) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = s0 + g4 * 4 + e;
            if (s < p.n) p.h_n[(int64_t)s * 2 * Hh + dir * Hh + unit[ct]] = hreg[ct][e];
        }
    }
}


// =============================================================================================
// split-precision recurrence (matmul mode bf16x3): same ownership and cell update as lstm_rec_kernel, but the
// recurrent product h . W_hh^T runs on v_mfma_f32_16x16x32_bf16 with h and W_hh split into bf16 hi + lo
// (hi*hi + hi*lo + lo*hi, fp32 accumulate; see csrc/gemm_bf16x3.hip).  W_hh is pre-split into a fragment-ordered
// bf16 image (same bytes as the fp32 image); the matrix work per 32-wide k block drops from 8*NT fp32 MFMAs
// (256 cycles each tile) to 3*NT bf16 MFMAs (48 cycles), so the kernel is purely a W_hh stream and the fragments of
// a WHOLE k block are prefetched while the previous block is multiplied.
// =============================================================================================
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using gbf8p = const __attribute__((address_space(1))) bf16x8 *;

// pack[dir][kb][tile][part][lane][j] (bf16): W_hh[dir][tile*16 + (lane&15)][32kb + 8(lane>>4) + j] split into part 0 = hi, 1 = lo
__global__ void whh_pack_bf16_kernel(const float *w0, const float *w1, __bf16 *pack, int Hh) {
    const int64_t per_dir = 8 * (int64_t)Hh * Hh;                 // bf16 elements per direction (hi + lo)
    const int64_t e8 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 8-element vector per thread
    if (e8 * 8 >= 2 * per_dir) return;
    const int dir = (int)(e8 * 8 / per_dir);
    int64_t r = e8 - dir * (per_dir / 8);
    const int lane = (int)(r & 63);
    r >>= 6;
    const int part = (int)(r & 1);
    r >>= 1;
    const int ntile = 4 * Hh / 16;
    const int tile = (int)(r % ntile), kb = (int)(r / ntile);
    const float *w = (dir == 0 ? w0 : w1) + (int64_t)(tile * 16 + (lane & 15)) * Hh + 32 * kb + 8 * (lane >> 4);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 hi = (__bf16)w[j];
        o[j] = part == 0 ? hi : (__bf16)(w[j] - (float)hi);
    }
    *reinterpret_cast<bf16x8 *>(pack + e8 * 8) = o;
}

template <int NCT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void lstm_rec_x3_kernel(LstmRecParams p) {
    extern __shared__ __attribute__((aligned(16))) float hbuf[];  // [2][16][Hh+4]
    constexpr int NT = NCT * 4;
    const int Hh = p.Hh, ldh = Hh + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & 15, g4 = lane >> 4;
    const int dir = blockIdx.y;
    const int s0 = blockIdx.x * 16;
    const int ntiles = Hh >> 4, nkb = Hh >> 5;          // nkb is even (launcher guarantees Hh % 64 == 0)
    const int ntile4 = 4 * ntiles;
    const __attribute__((address_space(1))) __bf16 *wp =
        (const __attribute__((address_space(1))) __bf16 *)p.w_pack + (int64_t)dir * 8 * Hh * Hh + lane * 8;
    gfp xproj = (gfp)p.xproj;
    __attribute__((address_space(1))) float *outp = (__attribute__((address_space(1))) float *)p.out;

    int off4[4], len4[4];
    int lmax = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int s = s0 + g4 * 4 + e;
        off4[e] = 0; len4[e] = 0;
        if (s < p.n) { off4[e] = p.seq_off[s]; len4[e] = p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - off4[e]; }
    }
    for (int s = s0; s < min(s0 + 16, p.n); ++s) lmax = max(lmax, p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - p.seq_off[s]);
    for (int i = tid; i < 2 * 16 * ldh; i += NWAVES * 64) hbuf[i] = 0.0f;
    float creg[NCT][4];          // h of a finished (inactive) sequence is carried in LDS, not in registers
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) creg[ct][e] = 0.0f;
    bool own[NCT];
    int unit[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int tile = wave * NCT + ct;
        own[ct] = tile < ntiles;
        unit[ct] = (own[ct] ? tile : 0) * 16 + cl;
    }
    // element offset of fragment ti (hi part) inside a k block: ((gate * ntiles) + column block) * 1024, recomputed
    // where used (two adds) instead of living in NT registers
    int cb[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) cb[ct] = (own[ct] ? wave * NCT + ct : 0) * 1024;
    const int64_t ldx = 8 * (int64_t)Hh;
    const int xcol = dir * 4 * Hh;

    bf16x8 bA[NT][2], bB[NT][2];
#define X3_LOADB(dst, kb)                                                              \
    _Pragma("unroll") for (int t_ = 0; t_ < NT; ++t_) {                                 \
        const int o_ = ((kb) * ntile4 + (t_ & 3) * ntiles) * 1024 + cb[t_ >> 2];        \
        dst[t_][0] = *(gbf8p)(wp + o_);                                                 \
        dst[t_][1] = *(gbf8p)(wp + o_ + 512);                                           \
    }
#define X3_MFMA(src, kb)                                                                                        \
    {                                                                                                           \
        const v4f a0_ = *reinterpret_cast<const v4f *>(hc + cl * ldh + (kb) * 32 + 8 * g4);                     \
        const v4f a1_ = *reinterpret_cast<const v4f *>(hc + cl * ldh + (kb) * 32 + 8 * g4 + 4);                 \
        bf16x8 ah_, al_;                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                      \
            ah_[j_] = (__bf16)a0_[j_]; al_[j_] = (__bf16)(a0_[j_] - (float)ah_[j_]);                            \
            ah_[4 + j_] = (__bf16)a1_[j_]; al_[4 + j_] = (__bf16)(a1_[j_] - (float)ah_[4 + j_]);                \
        }                                                                                                       \
        _Pragma("unroll") for (int t_ = 0; t_ < NT; ++t_) acc[t_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al_, src[t_][0], acc[t_], 0, 0, 0);   \
        _Pragma("unroll") for (int t_ = 0; t_ < NT; ++t_) acc[t_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_, src[t_][1], acc[t_], 0, 0, 0);   \
        _Pragma("unroll") for (int t_ = 0; t_ < NT; ++t_) acc[t_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_, src[t_][0], acc[t_], 0, 0, 0);   \
    }

    float xp[NCT][4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool act = 0 < len4[e];
        const int64_t row = off4[e] + (act ? (dir == 0 ? 0 : len4[e] - 1) : 0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int gate = 0; gate < 4; ++gate)
                xp[ct][gate][e] = (act && own[ct]) ? xproj[row * ldx + xcol + gate * Hh + unit[ct]] : 0.0f;
    }
    X3_LOADB(bA, 0);
    __syncthreads();

    for (int tau = 0; tau < lmax; ++tau) {
        const int cur = tau & 1;
        const float *hc = hbuf + cur * 16 * ldh;
        float *hn = hbuf + (cur ^ 1) * 16 * ldh;
        v4f acc[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) acc[ti] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < nkb; kb += 2) {
            X3_LOADB(bB, kb + 1);
            __builtin_amdgcn_sched_barrier(0);
            X3_MFMA(bA, kb);
            const int kn = kb + 2 < nkb ? kb + 2 : 0;            // wraps to the next step's first block
            X3_LOADB(bA, kn);
            __builtin_amdgcn_sched_barrier(0);
            X3_MFMA(bB, kb + 1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool active = tau < len4[e];
            const int t = dir == 0 ? tau : len4[e] - 1 - tau;
            const int64_t row = off4[e] + (active ? t : 0);
            const bool act_n = tau + 1 < len4[e];
            const int64_t row_n = off4[e] + (act_n ? (dir == 0 ? tau + 1 : len4[e] - 2 - tau) : 0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (!own[ct]) continue;
                const float gi = acc[ct * 4 + 0][e] + xp[ct][0][e];
                const float gf = acc[ct * 4 + 1][e] + xp[ct][1][e];
                const float gg = acc[ct * 4 + 2][e] + xp[ct][2][e];
                const float go = acc[ct * 4 + 3][e] + xp[ct][3][e];
                const float si = sigmoid_fast(gi), sf = sigmoid_fast(gf), tg = tanh_fast(gg), so = sigmoid_fast(go);
                const float cn = sf * creg[ct][e] + si * tg;
                const float hv = so * tanh_fast(cn);
                float hkeep = hc[(g4 * 4 + e) * ldh + unit[ct]];
                if (active) {
                    creg[ct][e] = cn;
                    hkeep = hv;
                    outp[row * p.ldo + dir * Hh + unit[ct]] = hv;
                    if (p.cbuf) {
                        float *gsave = const_cast<float *>(p.xproj) + row * ldx + xcol + unit[ct];
                        gsave[0] = si; gsave[Hh] = sf; gsave[2 * Hh] = tg; gsave[3 * Hh] = so;
                        p.cbuf[row * 2 * Hh + dir * Hh + unit[ct]] = cn;
                    }
                }
                hn[(g4 * 4 + e) * ldh + unit[ct]] = hkeep;
#pragma unroll
                for (int gate = 0; gate < 4; ++gate)
                    xp[ct][gate][e] = act_n ? xproj[row_n * ldx + xcol + gate * Hh + unit[ct]] : 0.0f;
            }
        }
        __syncthreads();
    }
#undef X3_LOADB
#undef X3_MFMA
    const float *hl = hbuf + (lmax & 1) * 16 * ldh;       // buffer written by the last step (zeros if lmax == 0)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        if (!own[ct]) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = s0 + g4 * 4 + e;
            if (s < p.n) p.h_n[(int64_t)s * 2 * Hh + dir * Hh + unit[ct]] = hl[(g4 * 4 + e) * ldh + unit[ct]];
        }
    }
}

// =============================================================================================
// backward through time
// =============================================================================================
// pack^T image for the recurrent product of the backward pass, dh_prev = dgates . W_hh:
//   packT[dir][kb][colblk][half][lane][j] = W_hh[dir][32*kb + 8*(lane>>4) + 4*half + j][colblk*16 + (lane&15)]
// (contraction over the 4Hh gate rows, output column = hidden unit).
__global__ void whh_packT_kernel(const float *w0, const float *w1, float *pack, int Hh) {
    const int64_t per_dir = 4 * (int64_t)Hh * Hh;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float per thread (strided source)
    if (e >= 2 * per_dir) return;
    const int dir = (int)(e / per_dir);
    int64_t r = e - dir * per_dir;
    const int j = (int)(r & 3), lane = (int)((r >> 2) & 63);
    r >>= 8;
    const int half = (int)(r & 1);
    r >>= 1;
    const int ncol = Hh / 16;
    const int colblk = (int)(r % ncol), kb = (int)(r / ncol);
    const float *w = dir == 0 ? w0 : w1;
    pack[e] = w[(int64_t)(32 * kb + 8 * (lane >> 4) + 4 * half + j) * Hh + colblk * 16 + (lane & 15)];
}

// hprev[row][dir*Hh + u] = h of the previous step of that direction (0 at the sequence start)
// (padded storage: the rows past a sequence's length get h_prev = 0 and their gate-gradient rows are cleared, so that the
// weight-gradient GEMMs, which run over ALL rows, add nothing for them)
// grid (n, ceil(max_len / 8)): a block copies 8 rows of one sequence as float4s, every load issued before the first store
// (one block per sequence with a scalar loop took 46 us for 128 sequences: a latency chain, not bandwidth)
__global__ __launch_bounds__(256) void lstm_hprev_kernel(const float *out, int64_t ldo, const int32_t *seq_off, const int32_t *seq_len, int n, int Hh,
                                                         float *hprev, float *G) {
    const int s = blockIdx.x;
    const int beg = seq_off[s], span = seq_off[s + 1] - beg, len = seq_len ? seq_len[s] : span;
    const int t0 = blockIdx.y * 8;
    if (t0 >= span) return;
    const int q4 = 2 * Hh / 4;                                 // float4s per row
    const int rows = min(8, span - t0);
    using f4 = __attribute__((ext_vector_type(4))) float;
    for (int i0 = threadIdx.x; i0 < rows * q4; i0 += 4 * 256) {
        f4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            v[k] = f4{0.f, 0.f, 0.f, 0.f};
            if (i < rows * q4) {
                const int t = t0 + i / q4, c = 4 * (i - (i / q4) * q4);
                const int tp = c < Hh ? t - 1 : t + 1;
                if (t < len && tp >= 0 && tp < len) v[k] = *reinterpret_cast<const f4 *>(out + (int64_t)(beg + tp) * ldo + c);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            if (i < rows * q4) {
                const int t = t0 + i / q4, c = 4 * (i - (i / q4) * q4);
                *reinterpret_cast<f4 *>(hprev + (int64_t)(beg + t) * 2 * Hh + c) = v[k];
            }
        }
    }
    if (seq_len && len < span) {                               // gate-gradient rows past the sequence's length: cleared
        const int z0 = max(t0, len), z1 = min(t0 + 8, span);
        for (int i = threadIdx.x; i < (z1 - z0) * 2 * Hh; i += 256)
            *reinterpret_cast<f4 *>(G + (int64_t)(beg + z0) * 8 * Hh + 4 * (int64_t)i) = f4{0.f, 0.f, 0.f, 0.f};
    }
}

// out rows past a sequence's length (padded storage) are defined: zero
__global__ void lstm_zero_tail_kernel(float *out, int64_t ldo, const int32_t *seq_off, const int32_t *seq_len, int Hh) {
    const int s = blockIdx.x;
    const int beg = seq_off[s], span = seq_off[s + 1] - beg, len = seq_len[s];
    for (int i = threadIdx.x; i < (span - len) * 2 * Hh; i += blockDim.x) {
        const int t = len + i / (2 * Hh), c = i % (2 * Hh);
        out[(int64_t)(beg + t) * ldo + c] = 0.0f;
    }
}

struct LstmBwdParams {
    float *G;                // [rows, 8Hh]: activated gates in, gate pre-activation gradients out (in place)
    const float *cbuf;       // [rows, 2Hh]
    const float *d_out;      // [rows, ldd] gradient w.r.t. the layer output
    int64_t ldd;
    const float *d_hn;       // [n, 2Hh] or null
    const float *w_packT;
    const int32_t *seq_off;
    const int32_t *seq_len;  // optional, as in LstmRecParams
    int n, Hh;
};

// Reverse-time recurrence.  Same ownership as the forward kernel (workgroup = 16 sequences x one
// direction, lane = (sequence 4*g4+e, hidden unit) in the MFMA C/D layout), so dh and dc stay in
// registers; only the gate gradients cross waves, through an LDS image [16][4Hh] that is the A
// operand of dh_prev = dgates . W_hh (v_mfma_f32_16x16x4_f32, contraction over the 4Hh gate rows).
template <int NCT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void lstm_bwd_kernel(LstmBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float dgs[];   // [16][4Hh+4]
    const int Hh = p.Hh, ldg = 4 * Hh + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & 15, g4 = lane >> 4;
    const int dir = blockIdx.y;
    const int s0 = blockIdx.x * 16;
    const int ntiles = Hh >> 4, nkb = (4 * Hh) >> 5;
    const int64_t ldx = 8 * (int64_t)Hh;
    const int xcol = dir * 4 * Hh;

    int off4[4], len4[4];
    int lmax = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int s = s0 + g4 * 4 + e;
        off4[e] = 0; len4[e] = 0;
        if (s < p.n) { off4[e] = p.seq_off[s]; len4[e] = p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - off4[e]; }
    }
    for (int s = s0; s < min(s0 + 16, p.n); ++s) lmax = max(lmax, p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - p.seq_off[s]);

    bool own[NCT];
    int unit[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int tile = wave * NCT + ct;
        own[ct] = tile < ntiles;
        unit[ct] = (own[ct] ? tile : 0) * 16 + cl;
    }
    gfp wp = (gfp)p.w_packT + (int64_t)dir * 4 * Hh * Hh + lane * 4;

    float dh[NCT][4], dc[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = s0 + g4 * 4 + e;
            dc[ct][e] = 0.0f;
            dh[ct][e] = (p.d_hn && s < p.n && own[ct]) ? p.d_hn[(int64_t)s * 2 * Hh + dir * Hh + unit[ct]] : 0.0f;
        }
    for (int i = tid; i < 16 * ldg; i += NWAVES * 64) dgs[i] = 0.0f;
    __syncthreads();

    for (int tau = lmax - 1; tau >= 0; --tau) {
        // ---- cell backward (lane-local) ----
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool active = tau < len4[e];
            const int t = dir == 0 ? tau : len4[e] - 1 - tau;
            const int64_t row = off4[e] + (active ? t : 0);
            const int64_t rowp = row + (dir == 0 ? -1 : 1);      // row of step tau-1 (valid iff tau > 0)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (!own[ct]) continue;
                float di = 0.f, df = 0.f, dg = 0.f, dob = 0.f;
                if (active) {
                    float *g = p.G + row * ldx + xcol + unit[ct];
                    const float gi = g[0], gf = g[Hh], gg = g[2 * Hh], go = g[3 * Hh];
                    const float c = p.cbuf[row * 2 * Hh + dir * Hh + unit[ct]];
                    const float cp = tau > 0 ? p.cbuf[rowp * 2 * Hh + dir * Hh + unit[ct]] : 0.0f;
                    const float dht = dh[ct][e] + p.d_out[row * p.ldd + dir * Hh + unit[ct]];
                    const float tc = tanh_fast(c);
                    dob = dht * tc * go * (1.0f - go);
                    const float dct = dc[ct][e] + dht * go * (1.0f - tc * tc);
                    di = dct * gg * gi * (1.0f - gi);
                    df = dct * cp * gf * (1.0f - gf);
                    dg = dct * gi * (1.0f - gg * gg);
                    dc[ct][e] = dct * gf;
                    g[0] = di; g[Hh] = df; g[2 * Hh] = dg; g[3 * Hh] = dob;
                }
                float *l = dgs + (g4 * 4 + e) * ldg + unit[ct];
                l[0] = di; l[Hh] = df; l[2 * Hh] = dg; l[3 * Hh] = dob;
            }
        }
        __syncthreads();
        // ---- dh_prev = dgates . W_hh ----
        v4f acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < nkb; ++kb) {
            const v4f a0 = *reinterpret_cast<const v4f *>(dgs + cl * ldg + kb * 32 + 8 * g4);
            const v4f a1 = *reinterpret_cast<const v4f *>(dgs + cl * ldg + kb * 32 + 8 * g4 + 4);
            v4f b[NCT][2];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int colblk = own[ct] ? wave * NCT + ct : 0;
                gfp src = wp + ((int64_t)kb * ntiles + colblk) * 512;
                b[ct][0] = *(gv4p)(src);
                b[ct][1] = *(gv4p)(src + 256);
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(jj < 4 ? a0[jj & 3] : a1[jj & 3], b[ct][jj >> 2][jj & 3],
                                                                   acc[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (tau < len4[e]) dh[ct][e] = acc[ct][e];
        __syncthreads();
    }
}

// ---- split-precision reverse recurrence (matmul mode bf16x3) -------------------------------------------------
// packT[dir][kb][colblk][part][lane][j] (bf16) = W_hh[dir][32kb + 8(lane>>4) + j][colblk*16 + (lane&15)], part 0 hi / 1 lo
__global__ void whh_packT_bf16_kernel(const float *w0, const float *w1, __bf16 *pack, int Hh) {
    const int64_t per_dir = 8 * (int64_t)Hh * Hh;
    const int64_t e8 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e8 * 8 >= 2 * per_dir) return;
    const int dir = (int)(e8 * 8 / per_dir);
    int64_t r = e8 - dir * (per_dir / 8);
    const int lane = (int)(r & 63);
    r >>= 6;
    const int part = (int)(r & 1);
    r >>= 1;
    const int ncol = Hh / 16;
    const int colblk = (int)(r % ncol), kb = (int)(r / ncol);
    const float *w = (dir == 0 ? w0 : w1) + (int64_t)(32 * kb + 8 * (lane >> 4)) * Hh + colblk * 16 + (lane & 15);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = w[(int64_t)j * Hh];
        const __bf16 hi = (__bf16)x;
        o[j] = part == 0 ? hi : (__bf16)(x - (float)hi);
    }
    *reinterpret_cast<bf16x8 *>(pack + e8 * 8) = o;
}

// Same ownership as lstm_bwd_kernel.  The gate gradients are written to LDS ALREADY split (bf16 hi and lo images
// [16][4Hh+8]), so the A fragments of dh_prev = dgates . W_hh are plain ds_read_b128 and the MFMA loop carries no
// conversions; W_hh^T fragments are prefetched four k blocks (of 32 gate rows) at a time.
template <int NCT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void lstm_bwd_x3_kernel(LstmBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) __bf16 dgs16[];   // [2 parts][16][4Hh+8]
    const int Hh = p.Hh, ldg = 4 * Hh + 8;
    __bf16 *dg_hi = dgs16, *dg_lo = dgs16 + 16 * ldg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & 15, g4 = lane >> 4;
    const int dir = blockIdx.y;
    const int s0 = blockIdx.x * 16;
    const int ntiles = Hh >> 4, nkb = (4 * Hh) >> 5;      // nkb is a multiple of 8 (Hh % 64 == 0)
    const int64_t ldx = 8 * (int64_t)Hh;
    const int xcol = dir * 4 * Hh;

    int off4[4], len4[4];
    int lmax = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int s = s0 + g4 * 4 + e;
        off4[e] = 0; len4[e] = 0;
        if (s < p.n) { off4[e] = p.seq_off[s]; len4[e] = p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - off4[e]; }
    }
    for (int s = s0; s < min(s0 + 16, p.n); ++s) lmax = max(lmax, p.seq_len ? p.seq_len[s] : p.seq_off[s + 1] - p.seq_off[s]);
    bool own[NCT];
    int unit[NCT], cb[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int tile = wave * NCT + ct;
        own[ct] = tile < ntiles;
        unit[ct] = (own[ct] ? tile : 0) * 16 + cl;
        cb[ct] = (own[ct] ? tile : 0) * 1024;
    }
    const __attribute__((address_space(1))) __bf16 *wp =
        (const __attribute__((address_space(1))) __bf16 *)p.w_packT + (int64_t)dir * 8 * Hh * Hh + lane * 8;

    float dh[NCT][4], dc[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = s0 + g4 * 4 + e;
            dc[ct][e] = 0.0f;
            dh[ct][e] = (p.d_hn && s < p.n && own[ct]) ? p.d_hn[(int64_t)s * 2 * Hh + dir * Hh + unit[ct]] : 0.0f;
        }
    for (int i = tid; i < 2 * 16 * ldg; i += NWAVES * 64) dgs16[i] = (__bf16)0.0f;
    __syncthreads();

    bf16x8 bA[4][NCT][2], bB[4][NCT][2];
#define B3_LOAD(dst, kb0)                                                                         \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                               \
        _Pragma("unroll") for (int ct_ = 0; ct_ < NCT; ++ct_) {                                    \
            const int o_ = (((kb0) + q_) * ntiles) * 1024 + cb[ct_];                               \
            dst[q_][ct_][0] = *(gbf8p)(wp + o_);                                                   \
            dst[q_][ct_][1] = *(gbf8p)(wp + o_ + 512);                                             \
        }
#define B3_MFMA(src, kb0)                                                                                          \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                              \
        const bf16x8 ah_ = *reinterpret_cast<const bf16x8 *>(dg_hi + cl * ldg + ((kb0) + q_) * 32 + 8 * g4);        \
        const bf16x8 al_ = *reinterpret_cast<const bf16x8 *>(dg_lo + cl * ldg + ((kb0) + q_) * 32 + 8 * g4);        \
        _Pragma("unroll") for (int ct_ = 0; ct_ < NCT; ++ct_) {                                                     \
            acc[ct_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al_, src[q_][ct_][0], acc[ct_], 0, 0, 0);            \
            acc[ct_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_, src[q_][ct_][1], acc[ct_], 0, 0, 0);            \
            acc[ct_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah_, src[q_][ct_][0], acc[ct_], 0, 0, 0);            \
        }                                                                                                           \
    }
    B3_LOAD(bA, 0);

    for (int tau = lmax - 1; tau >= 0; --tau) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool active = tau < len4[e];
            const int t = dir == 0 ? tau : len4[e] - 1 - tau;
            const int64_t row = off4[e] + (active ? t : 0);
            const int64_t rowp = row + (dir == 0 ? -1 : 1);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (!own[ct]) continue;
                float di = 0.f, df = 0.f, dg = 0.f, dob = 0.f;
                if (active) {
                    float *g = p.G + row * ldx + xcol + unit[ct];
                    const float gi = g[0], gf = g[Hh], gg = g[2 * Hh], go = g[3 * Hh];
                    const float c = p.cbuf[row * 2 * Hh + dir * Hh + unit[ct]];
                    const float cp = tau > 0 ? p.cbuf[rowp * 2 * Hh + dir * Hh + unit[ct]] : 0.0f;
                    const float dht = dh[ct][e] + p.d_out[row * p.ldd + dir * Hh + unit[ct]];
                    const float tc = tanh_fast(c);
                    dob = dht * tc * go * (1.0f - go);
                    const float dct = dc[ct][e] + dht * go * (1.0f - tc * tc);
                    di = dct * gg * gi * (1.0f - gi);
                    df = dct * cp * gf * (1.0f - gf);
                    dg = dct * gi * (1.0f - gg * gg);
                    dc[ct][e] = dct * gf;
                    g[0] = di; g[Hh] = df; g[2 * Hh] = dg; g[3 * Hh] = dob;
                }
                const int o = (g4 * 4 + e) * ldg + unit[ct];
                const float vals[4] = {di, df, dg, dob};
#pragma unroll
                for (int gate = 0; gate < 4; ++gate) {
                    const __bf16 hi = (__bf16)vals[gate];
                    dg_hi[o + gate * Hh] = hi;
                    dg_lo[o + gate * Hh] = (__bf16)(vals[gate] - (float)hi);
                }
            }
        }
        __syncthreads();
        v4f acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < nkb; kb += 8) {
            B3_LOAD(bB, kb + 4);
            __builtin_amdgcn_sched_barrier(0);
            B3_MFMA(bA, kb);
            const int kn = kb + 8 < nkb ? kb + 8 : 0;        // wraps to the next step
            B3_LOAD(bA, kn);
            __builtin_amdgcn_sched_barrier(0);
            B3_MFMA(bB, kb + 4);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (tau < len4[e]) dh[ct][e] = acc[ct][e];
        __syncthreads();
    }
#undef B3_LOAD
#undef B3_MFMA
}

bool lstm_bwd_takes_coop(const stair_lstm_bwd_args &a) {
    const char *cmax = getenv("STAIR_LSTM_COOP_BWD_MAX_N");
    return a.coop_ws && lstm_coop_usable(a.Hh) && a.n <= (cmax ? atoi(cmax) : 1024);
}

// reverse-time recurrence only: gates (activated, saved by the forward pass) -> gate pre-activation gradients, in place
int launch_lstm_bwd_recur(const stair_lstm_bwd_args &a, hipStream_t s) {
    STAIR_CHECK(a.n >= 0 && a.rows >= 0 && a.I > 0 && a.Hh > 0, "bad shape");
    STAIR_CHECK(a.Hh % 32 == 0 && a.Hh <= 256, "LSTM hidden size must be a multiple of 32, at most 256");
    STAIR_CHECK(a.gates && a.cbuf && a.out && a.d_out && a.whh_pack_ws && a.hprev_ws, "null buffer");
    if (a.n == 0 || a.rows == 0) return 0;
    const int Hh = a.Hh;
    // Cooperative BPTT (csrc/lstm_coop.hip) while every 32-sequence tile gets a group of its own: 338 us at n = 8 and 819 us at
    // n = 1024 against 727 / 920 us of the one-workgroup kernel; beyond that both are bound by the saved-state traffic
    // (2.8 GB per launch at n = 2048) and the one-workgroup kernel's 1.1 ms beats two tiles per group (1.6 ms).
    int rc_coop = -1;
    if (lstm_bwd_takes_coop(a)) rc_coop = launch_lstm_bwd_coop(a, s);      // -1: not co-resident on this device
    if (rc_coop > 0) return rc_coop;
    if (rc_coop < 0) {
        const bool split = matmul_mode() != STAIR_MATMUL_F32 && Hh % 64 == 0;
        if (split) {
            const int64_t n8 = 2 * 8 * (int64_t)Hh * Hh / 8;
            hipLaunchKernelGGL(whh_packT_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, a.w_hh[0], a.w_hh[1],
                               reinterpret_cast<__bf16 *>(a.whh_pack_ws), Hh);
            STAIR_LAUNCH_CHECK();
        } else {
            const int64_t ne = 2 * 4 * (int64_t)Hh * Hh;
            hipLaunchKernelGGL(whh_packT_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, a.w_hh[0], a.w_hh[1],
                               a.whh_pack_ws, Hh);
            STAIR_LAUNCH_CHECK();
        }
        LstmBwdParams p;
        p.G = a.gates; p.cbuf = a.cbuf; p.d_out = a.d_out; p.ldd = a.ldd; p.d_hn = a.d_hn; p.w_packT = a.whh_pack_ws;
        p.seq_off = a.seq_off; p.seq_len = a.seq_len; p.n = a.n; p.Hh = Hh;
        const dim3 grid((a.n + 15) / 16, 2);
        const size_t shmem = split ? 2 * 16 * (4 * Hh + 8) * sizeof(__bf16) : 16 * (4 * Hh + 4) * sizeof(float);
        const int tiles = Hh / 16;
        if (shmem > 48 * 1024) {   // 16 x (4*256+4) floats = 65.8 KB: above the default dynamic-LDS limit
            STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_bwd_kernel<2, 8>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_bwd_x3_kernel<2, 8>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        }
        STAIR_ACCT_MFMA(split ? "lstm_bwd_x3" : "lstm_bwd_f32", 0, 2ll * 2 * a.rows * 4 * Hh * Hh);
        if (split) {
            if (tiles > 8) hipLaunchKernelGGL((lstm_bwd_x3_kernel<2, 8>), grid, dim3(512), shmem, s, p);
            else if (tiles > 4) hipLaunchKernelGGL((lstm_bwd_x3_kernel<1, 8>), grid, dim3(512), shmem, s, p);
            else hipLaunchKernelGGL((lstm_bwd_x3_kernel<1, 4>), grid, dim3(256), shmem, s, p);
        } else if (tiles > 8) hipLaunchKernelGGL((lstm_bwd_kernel<2, 8>), grid, dim3(512), shmem, s, p);
        else if (tiles > 4) hipLaunchKernelGGL((lstm_bwd_kernel<1, 8>), grid, dim3(512), shmem, s, p);
        else if (tiles > 2) hipLaunchKernelGGL((lstm_bwd_kernel<1, 4>), grid, dim3(256), shmem, s, p);
        else hipLaunchKernelGGL((lstm_bwd_kernel<1, 2>), grid, dim3(128), shmem, s, p);
        STAIR_LAUNCH_CHECK();
    }
    return 0;
}

// h(t-1) rows + the four weight-gradient products of a layer whose gate gradients are in a.gates
// One weight-gradient product of a layer: through the slab-reduced kernel (csrc/gemm_tn_x3tr.hip: no atomics; the sums are added
// by the caller's tn_x3tr_flush) when the shape is its and the scratch suffices, the < 32 rows past the last whole stage through the
// atomic kernel (one add per element: still deterministic); otherwise the atomic kernel.
static int lstm_weight_product(stair_gemm_tn_args g, float *&scr, int64_t &left, hipStream_t s) {
    if (g.b_is_bf16 && scr) {               // dW_ih on stored-bf16 rows: the transposed-read kernel with its 8 slabs stored, then added in order
        const int64_t need = align_up((int64_t)8 * g.N * g.K, 64);
        if (need <= left) {
            const int rc = launch_gemm_tn_tr_slabs(g, scr, s);
            if (rc >= 0) {
                if (rc == 0) { scr += need; left -= need; }
                return rc;
            }
        }
    }
    stair_gemm_tn_args h = g;
    h.M = g.M & ~31;
    if (scr && h.M >= 2048 && tn_x3tr_takes(h)) {
        const int64_t need = align_up(tn_x3tr_scratch_floats(h.M, h.N, h.K), 64);
        if (need <= left) {
            if (int rc = launch_gemm_tn_x3tr(h, scr, s)) return rc;
            scr += need; left -= need;
            if (g.M == h.M) return 0;
            g.A += (int64_t)h.M * g.lda; g.B += (int64_t)h.M * g.ldb; g.M -= h.M;
        }
    }
    return launch_gemm_tn(g, s);
}

int launch_lstm_bwd_weights(const stair_lstm_bwd_args &a, hipStream_t s) {
    if (a.n == 0 || a.rows == 0) return 0;
    const int Hh = a.Hh;
    hipLaunchKernelGGL(lstm_hprev_kernel, dim3(a.n, (a.max_len + 7) / 8), dim3(256), 0, s, a.out, a.ldo, a.seq_off, a.seq_len, a.n, Hh, a.hprev_ws, a.gates);
    STAIR_LAUNCH_CHECK();
    float *scr = a.tn_ws;
    int64_t left = a.tn_ws ? a.tn_ws_floats : 0;
    // weight gradients: dW_ih = dG^T X, dW_hh = dG^T Hprev, db_ih = db_hh = colsum(dG)
    for (int dir = 0; dir < 2; ++dir) {
        stair_gemm_tn_args g = {};
        g.A = a.gates + dir * 4 * Hh; g.lda = 8 * (int64_t)Hh;
        g.B = a.x_bf16 ? static_cast<const float *>(a.x_bf16) : a.x; g.b_is_bf16 = a.x_bf16 ? 1 : 0;
        g.ldb = a.ldx; g.b_gstride = a.ldx; g.rows_per_group = 1;
        g.C = a.dw_ih[dir]; g.ldc = a.I; g.M = a.rows; g.N = 4 * Hh; g.K = a.I;
        if (int rc = lstm_weight_product(g, scr, left, s)) return rc;
        g.b_is_bf16 = 0;
        g.B = a.hprev_ws + dir * Hh; g.ldb = 2 * (int64_t)Hh; g.b_gstride = 2 * (int64_t)Hh;
        g.C = a.dw_hh[dir]; g.ldc = Hh; g.K = Hh;
        g.colsum = a.db_ih[dir]; g.colsum2 = a.db_hh[dir];     // db_ih = db_hh = colsum(dG), with the smaller of the two products
        if (int rc = lstm_weight_product(g, scr, left, s)) return rc;
    }
    return 0;
}

int launch_lstm_bwd(const stair_lstm_bwd_args &a, hipStream_t s) {
    if (int rc = launch_lstm_bwd_recur(a, s)) return rc;
    return launch_lstm_bwd_weights(a, s);
}

int launch_lstm_zero_tail(const stair_lstm_args &a, hipStream_t s);

// input projection (+ bias sums, tail zeroing) of a layer: everything of launch_lstm before the recurrence
int launch_lstm_project(const stair_lstm_args &a, hipStream_t s) {
    STAIR_CHECK(a.n >= 0 && a.rows >= 0 && a.I > 0 && a.Hh > 0, "bad shape");
    STAIR_CHECK(a.Hh % 32 == 0 && a.Hh <= 256, "LSTM hidden size must be a multiple of 32, at most 256");
    STAIR_CHECK(a.I % 4 == 0 && a.ldx % 4 == 0, "LSTM input size / ldx must be multiples of 4");
    STAIR_CHECK(a.x || a.x_bf16, "no input rows");
    if (a.n == 0 || a.rows == 0) return 0;
    const int Hh = a.Hh;
    hipLaunchKernelGGL(bias_sum_kernel, dim3((8 * Hh + 255) / 256), dim3(256), 0, s, a.b_ih[0], a.b_hh[0], a.b_ih[1],
                       a.b_hh[1], a.bias_ws, 4 * Hh);
    STAIR_LAUNCH_CHECK();
    if (a.x_bf16) {
        // stored bf16 features: ONE plane GEMM for both directions (N = 8 Hh: an A panel leaves HBM once for all gate
        // columns), W_ih of both directions split into tiled hi/lo planes first (csrc/gemm_planes.hip)
        STAIR_CHECK(matmul_mode() == STAIR_MATMUL_BF16X3, "bf16 input rows need the bf16x3 matmul mode (STAIR_MATMUL)");
        STAIR_CHECK(a.I % 32 == 0 && a.ldx % 8 == 0, "bf16 input rows need I % 32 == 0 and ldx % 8 == 0");
        STAIR_CHECK(a.wih_planes_ws != nullptr, "wih_planes_ws missing");
        char *hi = static_cast<char *>(a.wih_planes_ws), *lo = hi + (size_t)8 * Hh * a.I * 2;
        // W_ih in MFMA fragment order, read global -> VGPR by the waves that multiply it (LDS-DMA stages the rows alone): STAIR_PLANES_WR=1
        static const bool wr = [] { const char *e = getenv("STAIR_PLANES_WR"); return e && e[0] == '1'; }();
        const bool use_wr = wr && a.I % 64 == 0 && Hh % 8 == 0;
        stair_gemm_planes_args g = {};
        if (use_wr) {
            const float *src[2] = {a.w_ih[0], a.w_ih[1]};
            void *dst[2] = {hi, hi + (size_t)4 * Hh * a.I * 2 * 2};
            if (int rc = launch_pack_wfrag_many(src, dst, 2, 4 * Hh, a.I, s)) return rc;
        } else {
            for (int dir = 0; dir < 2; ++dir)
                if (int rc = launch_split_planes_tiled(a.w_ih[dir], hi, lo, 4 * Hh, a.I, s, dir * 4 * Hh, 8 * Hh)) return rc;
        }
        g.A_hi = a.x_bf16; g.A_lo = nullptr; g.lda = a.ldx;
        g.W_hi = hi; g.W_lo = lo; g.ldw = 0; g.w_tiled = use_wr ? 2 : 1;
        g.bias = a.bias_ws; g.C = a.xproj_ws; g.ldc = 8 * (int64_t)Hh;
        g.M = a.rows; g.N = 8 * Hh; g.K = a.I; g.act = 0;
        if (int rc = launch_gemm_planes(g, s)) return rc;
    } else if (a.x_planes_ws && a.wih_planes_ws && matmul_mode() == STAIR_MATMUL_BF16X3 && 8 * Hh >= 256) {
        // (at EVERY row count: a row's result must not depend on how many other rows share the launch -- the kernel's tile is 256 rows,
        // a short batch fills part of one)
        // fp32 input rows (the text encoder: E = 300): split ONCE into zero-padded hi / lo planes, then the same LDS-DMA plane GEMM
        // for both directions with three products per operand pair (the register-staged kernel reads and splits every A row once
        // per column tile and direction)
        const int Kp = (a.I + 31) / 32 * 32;
        char *xh = static_cast<char *>(a.x_planes_ws), *xl = xh + (size_t)a.rows * Kp * 2;
        char *wh = static_cast<char *>(a.wih_planes_ws), *wl = wh + (size_t)8 * Hh * Kp * 2;
        if (int rc = launch_split_planes_pad(a.x, a.ldx, xh, xl, a.rows, a.I, Kp, false, s)) return rc;
        for (int dir = 0; dir < 2; ++dir)
            if (int rc = launch_split_planes_pad(a.w_ih[dir], a.I, wh, wl, 4 * Hh, a.I, Kp, true, s, dir * 4 * Hh, 8 * Hh)) return rc;
        stair_gemm_planes_args g = {};
        g.A_hi = xh; g.A_lo = xl; g.lda = Kp;
        g.W_hi = wh; g.W_lo = wl; g.ldw = 0; g.w_tiled = 1;
        g.bias = a.bias_ws; g.C = a.xproj_ws; g.ldc = 8 * (int64_t)Hh;
        g.M = a.rows; g.N = 8 * Hh; g.K = Kp; g.act = 0;
        if (int rc = launch_gemm_planes(g, s)) return rc;
    } else
    for (int dir = 0; dir < 2; ++dir) {
        stair_gemm_args g = {};
        g.A = a.x; g.lda = a.ldx; g.a_gstride = a.ldx;
        g.W = a.w_ih[dir]; g.ldw = a.I; g.bias = a.bias_ws + dir * 4 * Hh;
        g.C = a.xproj_ws + dir * 4 * Hh; g.ldc = 8 * (int64_t)Hh; g.c_gstride = 8 * (int64_t)Hh;
        g.groups = a.rows; g.rows_per_group = 1; g.N = 4 * Hh; g.K = a.I; g.act = 0;
        if (int rc = launch_gemm(g, s)) return rc;
    }
    if (a.out) return launch_lstm_zero_tail(a, s);       // (no output rows yet: projections enqueued ahead of their plan, stair_encoders_project)
    return 0;
}

// padded storage: rows past each sequence's length read as zero downstream
int launch_lstm_zero_tail(const stair_lstm_args &a, hipStream_t s) {
    if (a.n == 0 || a.rows == 0 || !a.seq_len) return 0;
    STAIR_CHECK(a.out, "null output rows");
    hipLaunchKernelGGL(lstm_zero_tail_kernel, dim3(a.n), dim3(256), 0, s, a.out, a.ldo, a.seq_off, a.seq_len, a.Hh);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// the recurrence over projected inputs (a.xproj_ws)
int launch_lstm_recur(const stair_lstm_args &a, hipStream_t s) {
    if (a.n == 0 || a.rows == 0) return 0;
    STAIR_CHECK(a.whh_pack_ws != nullptr, "whh_pack_ws missing");
    const int Hh = a.Hh;
    if (a.coop_ws && lstm_coop_usable(Hh)) {          // hidden units split over co-resident workgroups
        const int rc = launch_lstm_rec_coop(a, s);
        if (rc >= 0) return rc;                       // -1: the grid does not fit this device at once -> one-workgroup kernel below
    }
    const bool split = matmul_mode() != STAIR_MATMUL_F32 && Hh % 64 == 0;     // the split kernel walks k blocks in pairs
    if (split) {
        const int64_t n8 = 2 * 8 * (int64_t)Hh * Hh / 8;
        hipLaunchKernelGGL(whh_pack_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, a.w_hh[0], a.w_hh[1],
                           reinterpret_cast<__bf16 *>(a.whh_pack_ws), Hh);
        STAIR_LAUNCH_CHECK();
    } else {
        const int64_t n4 = 2 * 4 * (int64_t)Hh * Hh / 4;
        hipLaunchKernelGGL(whh_pack_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a.w_hh[0], a.w_hh[1],
                           a.whh_pack_ws, Hh);
        STAIR_LAUNCH_CHECK();
    }
    LstmRecParams p;
    p.xproj = a.xproj_ws; p.w_pack = a.whh_pack_ws;
    p.seq_off = a.seq_off; p.seq_len = a.seq_len; p.out = a.out; p.ldo = a.ldo; p.h_n = a.h_n; p.cbuf = a.cbuf; p.n = a.n; p.Hh = Hh;
    const dim3 grid((a.n + 15) / 16, 2);
    const size_t shmem = 2 * 16 * (Hh + 4) * sizeof(float);
    const int tiles = Hh / 16;
    STAIR_ACCT_MFMA(split ? "lstm_rec_x3" : "lstm_rec_f32", 0, 2ll * 2 * a.rows * 4 * Hh * Hh);
    if (split) {
        if (tiles > 8) hipLaunchKernelGGL((lstm_rec_x3_kernel<2, 8>), grid, dim3(512), shmem, s, p);
        else if (tiles > 4) hipLaunchKernelGGL((lstm_rec_x3_kernel<1, 8>), grid, dim3(512), shmem, s, p);
        else hipLaunchKernelGGL((lstm_rec_x3_kernel<1, 4>), grid, dim3(256), shmem, s, p);
    } else if (tiles > 8) hipLaunchKernelGGL((lstm_rec_kernel<2, 8>), grid, dim3(512), shmem, s, p);
    else if (tiles > 4) hipLaunchKernelGGL((lstm_rec_kernel<1, 8>), grid, dim3(512), shmem, s, p);
    else if (tiles > 2) hipLaunchKernelGGL((lstm_rec_kernel<1, 4>), grid, dim3(256), shmem, s, p);
    else hipLaunchKernelGGL((lstm_rec_kernel<1, 2>), grid, dim3(128), shmem, s, p);
    STAIR_LAUNCH_CHECK();
    return 0;
}

int launch_lstm(const stair_lstm_args &a, hipStream_t s) {
    if (int rc = launch_lstm_project(a, s)) return rc;
    return launch_lstm_recur(a, s);
}

}  // namespace stair

extern "C" int stair_lstm_bidir_fwd(const stair_lstm_args *args, stair_stream stream) {
    if (!args) {
        stair::set_error("stair_lstm_bidir_fwd: null args");
        return 1;
    }
    return stair::launch_lstm(*args, static_cast<hipStream_t>(stream));
}

extern "C" int stair_lstm_bidir_bwd(const stair_lstm_bwd_args *args, stair_stream stream) {
    if (!args) {
        stair::set_error("stair_lstm_bidir_bwd: null args");
        return 1;
    }
    // with tn_ws the slab-reduced weight gradients are queued by the launcher: added here, before returning (inside
    // stair_plan_backward the plan does that once for both encoders)
    stair::tn_x3tr_discard();
    int rc = stair::launch_lstm_bwd(*args, static_cast<hipStream_t>(stream));
    if (rc == 0) rc = stair::tn_x3tr_flush(static_cast<hipStream_t>(stream));
    else stair::tn_x3tr_discard();
    return rc;
}
