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
#include "common.h"

namespace stair {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct LstmRecParams {
    const float *xproj;      // [rows, 8Hh]
    const float *w_hh[2];    // [4Hh, Hh]
    const int32_t *seq_off;  // [n+1]
    float *out;              // [rows, ldo]
    int64_t ldo;
    float *h_n;              // [n, 2Hh]
    int n, Hh;
};

__global__ void bias_sum_kernel(const float *a0, const float *b0, const float *a1, const float *b1, float *out, int n4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = a0[i] + b0[i];
    else if (i < 2 * n4) out[i] = a1[i - n4] + b1[i - n4];
}

template <int NCT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void lstm_rec_kernel(LstmRecParams p) {
    extern __shared__ __attribute__((aligned(16))) float hbuf[];  // [2][16][Hh+4]
    const int Hh = p.Hh, ldh = Hh + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & 15, g4 = lane >> 4;
    const int dir = blockIdx.y;
    const int s0 = blockIdx.x * 16;
    const int ntiles = Hh >> 4;
    const float *__restrict__ whh = p.w_hh[dir];

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
            len4[e] = p.seq_off[s + 1] - off4[e];
        }
    }
    for (int s = s0; s < min(s0 + 16, p.n); ++s) lmax = max(lmax, p.seq_off[s + 1] - p.seq_off[s]);

    for (int i = tid; i < 2 * 16 * ldh; i += NWAVES * 64) hbuf[i] = 0.0f;
    float creg[NCT][4], hreg[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) creg[ct][e] = hreg[ct][e] = 0.0f;
    __syncthreads();

    // tiles owned by this wave; waves beyond the tile count idle but keep the barriers
    bool own[NCT];
    int unit[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int tile = wave * NCT + ct;
        own[ct] = tile < ntiles;
        unit[ct] = (own[ct] ? tile : 0) * 16 + cl;
    }

    const int64_t ldx = 8 * (int64_t)Hh;
    for (int tau = 0; tau < lmax; ++tau) {
        const int cur = tau & 1;
        const float *hc = hbuf + cur * 16 * ldh;
        float *hn = hbuf + (cur ^ 1) * 16 * ldh;
        bool active[4];
        int64_t row[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            active[e] = tau < len4[e];
            const int t = dir == 0 ? tau : len4[e] - 1 - tau;
            row[e] = off4[e] + (active[e] ? t : 0);
        }
        // issue this step's xproj reads now; they are consumed after the MFMA chain
        float xp[NCT][4][4];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int gate = 0; gate < 4; ++gate)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    xp[ct][gate][e] = (active[e] && own[ct])
                                          ? p.xproj[row[e] * ldx + dir * 4 * Hh + gate * Hh + unit[ct]]
                                          : 0.0f;

        f32x4 acc[NCT][4];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int gate = 0; gate < 4; ++gate) acc[ct][gate] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kb = 0; kb < ntiles; ++kb) {
            const float4 a4 = *reinterpret_cast<const float4 *>(hc + cl * ldh + kb * 16 + 4 * g4);
            const float *ap = reinterpret_cast<const float *>(&a4);
            float4 b4[NCT][4];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int gate = 0; gate < 4; ++gate)
                    b4[ct][gate] = *reinterpret_cast<const float4 *>(
                        whh + (int64_t)(gate * Hh + unit[ct]) * Hh + kb * 16 + 4 * g4);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int gate = 0; gate < 4; ++gate)
                        acc[ct][gate] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            ap[jj], reinterpret_cast<const float *>(&b4[ct][gate])[jj], acc[ct][gate], 0, 0, 0);
        }

#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            if (!own[ct]) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gi = acc[ct][0][e] + xp[ct][0][e];
                const float gf = acc[ct][1][e] + xp[ct][1][e];
                const float gg = acc[ct][2][e] + xp[ct][2][e];
                const float go = acc[ct][3][e] + xp[ct][3][e];
                const float cn = sigmoid_fast(gf) * creg[ct][e] + sigmoid_fast(gi) * tanh_fast(gg);
                const float hv = sigmoid_fast(go) * tanh_fast(cn);
                if (active[e]) {
                    creg[ct][e] = cn;
                    hreg[ct][e] = hv;
                    p.out[row[e] * p.ldo + dir * Hh + unit[ct]] = hv;
                }
                hn[(g4 * 4 + e) * ldh + unit[ct]] = hreg[ct][e];
            }
        }
        __syncthreads();
    }

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

int launch_lstm(const stair_lstm_args &a, hipStream_t s) {
    STAIR_CHECK(a.n >= 0 && a.rows >= 0 && a.I > 0 && a.Hh > 0, "bad shape");
    STAIR_CHECK(a.Hh % 16 == 0 && (a.Hh <= 128 || a.Hh == 256), "LSTM hidden size must be 16..128 (multiple of 16) or 256");
    STAIR_CHECK(a.I % 4 == 0 && a.ldx % 4 == 0, "LSTM input size / ldx must be multiples of 4");
    if (a.n == 0 || a.rows == 0) return 0;
    const int Hh = a.Hh;
    hipLaunchKernelGGL(bias_sum_kernel, dim3((8 * Hh + 255) / 256), dim3(256), 0, s, a.b_ih[0], a.b_hh[0], a.b_ih[1],
                       a.b_hh[1], a.bias_ws, 4 * Hh);
    STAIR_LAUNCH_CHECK();
    for (int dir = 0; dir < 2; ++dir) {
        stair_gemm_args g = {};
        g.A = a.x; g.lda = a.ldx; g.a_gstride = a.ldx;
        g.W = a.w_ih[dir]; g.ldw = a.I; g.bias = a.bias_ws + dir * 4 * Hh;
        g.C = a.xproj_ws + dir * 4 * Hh; g.ldc = 8 * (int64_t)Hh; g.c_gstride = 8 * (int64_t)Hh;
        g.groups = a.rows; g.rows_per_group = 1; g.N = 4 * Hh; g.K = a.I; g.act = 0;
        if (int rc = launch_gemm(g, s)) return rc;
    }
    LstmRecParams p;
    p.xproj = a.xproj_ws; p.w_hh[0] = a.w_hh[0]; p.w_hh[1] = a.w_hh[1];
    p.seq_off = a.seq_off; p.out = a.out; p.ldo = a.ldo; p.h_n = a.h_n; p.n = a.n; p.Hh = Hh;
    const dim3 grid((a.n + 15) / 16, 2);
    const size_t shmem = 2 * 16 * (Hh + 4) * sizeof(float);
    const int tiles = Hh / 16;
    if (tiles == 16) hipLaunchKernelGGL((lstm_rec_kernel<2, 8>), grid, dim3(512), shmem, s, p);
    else if (tiles > 4) hipLaunchKernelGGL((lstm_rec_kernel<1, 8>), grid, dim3(512), shmem, s, p);
    else if (tiles > 2) hipLaunchKernelGGL((lstm_rec_kernel<1, 4>), grid, dim3(256), shmem, s, p);
    else hipLaunchKernelGGL((lstm_rec_kernel<1, 2>), grid, dim3(128), shmem, s, p);
    STAIR_LAUNCH_CHECK();
    return 0;
}

}  // namespace stair

extern "C" int stair_lstm_bidir_fwd(const stair_lstm_args *args, stair_stream stream) {
    if (!args) {
        stair::set_error("stair_lstm_bidir_fwd: null args");
        return 1;
    }
    return stair::launch_lstm(*args, static_cast<hipStream_t>(stream));
}
