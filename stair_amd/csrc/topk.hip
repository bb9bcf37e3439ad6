// Cosine top-k of query rows against a small table of candidate representations:
// /root/reference/evaluate.py:65-117 (get_filter_text_results) ranks the 214 phrase representations of
// filter_answers.json against every Filter output with nn.CosineSimilarity + argsort and keeps the first 10.
//
// One wave per query.  The query (H <= 1024 floats) sits in registers, H/64 per lane; the candidate table (C x H, 438 KB
// at 214 x 512) is read coalesced from L2 by every wave; one butterfly reduction per candidate leaves the similarity
// in all lanes, and lane c % 64 keeps it.  Selection is k rounds of a wave-wide arg-max (ties: lower index), so
// the kernel is bound by C wave reductions per query, not by memory: n*C*H*4 bytes of L2 reads, n*k*8 bytes written.
#include "common.h"
#include "ops.h"

namespace stair {

namespace {
constexpr int TK_MAXJ = 4;       // H <= 1024: float4 groups per lane
constexpr int TK_SLOTS = 16;     // C <= 1024: candidates per lane
}  // namespace

__global__ __launch_bounds__(256) void key_invnorm_kernel(const float *keys, float *inv, int C, int H) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    float s = 0.0f;
    for (int h = lane; h < H; h += 64) { const float v = keys[(int64_t)c * H + h]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) inv[c] = 1.0f / fmaxf(sqrtf(s), 1e-8f);        // nn.CosineSimilarity eps, per operand
}

__global__ __launch_bounds__(256) void cosine_topk_kernel(const float *queries, int64_t ldq, const int32_t *q_idx,
                                                          const float *keys, const float *kinv, int n, int C, int H,
                                                          int k, int32_t *out_idx, float *out_sim) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= n) return;
    const float *qp = queries + (int64_t)(q_idx ? q_idx[q] : q) * ldq;
    const int nj = H >> 8;                     // full float4 groups per lane (H % 256 == 0 fast path)
    const bool vec = (H & 255) == 0 && nj <= TK_MAXJ;
    float4 qv[TK_MAXJ];
    float qn = 0.0f;
    if (vec) {
#pragma unroll
        for (int j = 0; j < TK_MAXJ; ++j)
            if (j < nj) {
                qv[j] = *reinterpret_cast<const float4 *>(qp + j * 256 + lane * 4);
                qn += qv[j].x * qv[j].x + qv[j].y * qv[j].y + qv[j].z * qv[j].z + qv[j].w * qv[j].w;
            }
    } else {
        for (int h = lane; h < H; h += 64) qn += qp[h] * qp[h];
    }
    const float qinv = 1.0f / fmaxf(sqrtf(wave_sum(qn)), 1e-8f);

    float sims[TK_SLOTS];
#pragma unroll
    for (int t = 0; t < TK_SLOTS; ++t) sims[t] = -INFINITY;
    for (int c = 0; c < C; ++c) {
        const float *kp = keys + (int64_t)c * H;
        float d = 0.0f;
        if (vec) {
#pragma unroll
            for (int j = 0; j < TK_MAXJ; ++j)
                if (j < nj) {
                    const float4 kv = *reinterpret_cast<const float4 *>(kp + j * 256 + lane * 4);
                    d += qv[j].x * kv.x + qv[j].y * kv.y + qv[j].z * kv.z + qv[j].w * kv.w;
                }
        } else {
            for (int h = lane; h < H; h += 64) d += qp[h] * kp[h];
        }
        d = wave_sum(d) * qinv * kinv[c];
        const int slot = c >> 6;
#pragma unroll
        for (int t = 0; t < TK_SLOTS; ++t)
            if (t == slot && (c & 63) == lane) sims[t] = d;
    }
    for (int r = 0; r < k; ++r) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < TK_SLOTS; ++t) {
            const int c = t * 64 + lane;
            if (c < C && (sims[t] > best || (sims[t] == best && c < bi))) { best = sims[t]; bi = c; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { out_idx[(int64_t)q * k + r] = bi; out_sim[(int64_t)q * k + r] = best; }
#pragma unroll
        for (int t = 0; t < TK_SLOTS; ++t)
            if (t * 64 + lane == bi) sims[t] = -INFINITY;
    }
}

}  // namespace stair

extern "C" int stair_cosine_topk(const float *queries, int64_t ldq, const int32_t *q_idx, const float *keys,
                                 float *key_invnorm_ws, int32_t n, int32_t C, int32_t H, int32_t k, int32_t *out_idx,
                                 float *out_sim, stair_stream stream) {
    using namespace stair;
    STAIR_CHECK(queries && keys && key_invnorm_ws && out_idx && out_sim, "null argument");
    STAIR_CHECK(n >= 0 && C > 0 && H > 0, "bad sizes");
    STAIR_CHECK(C <= 64 * TK_SLOTS, "at most 1024 candidates");
    STAIR_CHECK(k > 0 && k <= C, "k must be in 1..C");
    STAIR_CHECK(ldq >= H && (ldq % 4 == 0) && (reinterpret_cast<uintptr_t>(queries) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(keys) % 16 == 0),
                "queries/keys must be 16-byte aligned with ldq a multiple of 4");
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(key_invnorm_kernel, dim3((C + 3) / 4), dim3(256), 0, s, keys, key_invnorm_ws, C, H);
    STAIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(cosine_topk_kernel, dim3((n + 3) / 4), dim3(256), 0, s, queries, ldq, q_idx, keys, key_invnorm_ws, n,
                       C, H, k, out_idx, out_sim);
    STAIR_LAUNCH_CHECK();
    return 0;
}
