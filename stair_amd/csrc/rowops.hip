// HBM-bound row kernels of the NMN path: cosine attention, temporal relate nets, LayerNorm,
// reductions over frames, concatenations, softmaxes.  One wave (64 lanes) per row wherever a row
// reduction is needed, float4 loads along H, __shfl_xor wave reductions, no LDS unless stated.
#include <algorithm>
#include "ops.h"

namespace stair {

namespace {
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;

__device__ __forceinline__ int idx_or_id(const int32_t *idx, int i) { return idx ? idx[i] : i; }
__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
}  // namespace

// ---------------------------------------------------------------------------------------------
__global__ void span_mean_kernel(const float *tok, int64_t ld, const int32_t *start, const int32_t *count,
                                 float *vec, const int32_t *out_idx, int n, int H) {
    const int i = blockIdx.x;
    const int s = start[i], c = count[i];
    float *o = vec + (int64_t)idx_or_id(out_idx, i) * H;
    const float inv = 1.0f / (float)c;
    for (int col = threadIdx.x; col < H; col += blockDim.x) {
        float acc = 0.0f;
        for (int r = 0; r < c; ++r) acc += tok[(int64_t)(s + r) * ld + col];
        o[col] = acc * inv;   // torch.mean = sum / count; sum*inv differs by <= 1 ulp
    }
}
int launch_span_mean(const float *tok, int64_t ld, const int32_t *start, const int32_t *count, float *vec,
                     const int32_t *out_idx, int n, int H, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(span_mean_kernel, dim3(n), dim3(128), 0, s, tok, ld, start, count, vec, out_idx, n, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ void pack_kernel(int mode, const float *A, const int32_t *ia, const float *B, const int32_t *ib,
                            float *out, int n, int H) {
    const int i = blockIdx.x;
    const float *a = A + (int64_t)idx_or_id(ia, i) * H;
    const float *b = B + (int64_t)idx_or_id(ib, i) * H;
    const int segs = mode == PACK_CAT2 ? 2 : 3;
    float *o = out + (int64_t)i * segs * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float x = a[c], y = b[c];
        if (mode == PACK_CAT2) {
            o[c] = x; o[H + c] = y;
        } else if (mode == PACK_EXISTS) {
            o[c] = x; o[H + c] = y; o[2 * H + c] = x * y;
        } else {
            o[c] = fabsf(x - y); o[H + c] = x; o[2 * H + c] = y;
        }
    }
}
int launch_pack(int mode, const float *A, const int32_t *ia, const float *B, const int32_t *ib, float *out, int n,
                int H, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(pack_kernel, dim3(n), dim3(128), 0, s, mode, A, ia, B, ib, out, n, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// (cos + 1) * 0.49 of LocalizeModule / ExistsFrameModule, modules.py:170-177, 203-216.  One wave
// per (pair, frame) row; the keyword row is re-read from L2 by the T waves of a pair.
__global__ void cosine_attn_kernel(const float *F, int64_t f_gstride, const int32_t *f_idx, const float *Kmat,
                                   const int32_t *k_idx, float *att, const int32_t *out_idx, int npairs, int T,
                                   int H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= (int64_t)npairs * T) return;
    const int p = (int)(row / T), t = (int)(row - (int64_t)p * T);
    const float4 *f = reinterpret_cast<const float4 *>(F + (int64_t)idx_or_id(f_idx, p) * f_gstride + (int64_t)t * H);
    const float4 *k = reinterpret_cast<const float4 *>(Kmat + (int64_t)idx_or_id(k_idx, p) * H);
    float d = 0.f, nf = 0.f, nk = 0.f;
    for (int c = lane; c < H / 4; c += 64) {
        const float4 a = f[c], b = k[c];
        d += dot4(a, b); nf += dot4(a, a); nk += dot4(b, b);
    }
    d = wave_sum(d); nf = wave_sum(nf); nk = wave_sum(nk);
    if (lane == 0) {
        const float eps = 1e-8f;
        const float c = d / (fmaxf(sqrtf(nf), eps) * fmaxf(sqrtf(nk), eps));
        att[(int64_t)idx_or_id(out_idx, p) * T + t] = (c + 1.0f) * 0.49f;
    }
}
// Grouped form for Superlative (modules.py:220-248): ALL Ka pairs of an instance score the same private [T,H] tile, so the
// scores of an instance are one [Ka x H] . [H x T] product.  One block per instance walks H in 64-column chunks: the chunk's
// K [Ka][64] and F [T][64] slices in LDS (rows 65 floats apart: conflict-free for both operand orders), the products on
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate), the row norms summed from the same loaded values.
// (cosine_attn_kernel above re-reads the instance's tile once per pair: 2 GB through L2 for 254 instances with Ka = T = 64,
// 317 us of the training step's forward pass.)
using f32x16_ro = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void cosine_attn_grouped_kernel(const float *F, int64_t f_gstride, const float *Kmat,
                                                                  const int32_t *pair_start, const int32_t *pair_cnt, float *att,
                                                                  int n, int T, int H, int ka_max) {
    extern __shared__ float sm[];          // kt [KP][65] | ft [TP][65] | nk [KP] | nf [TP]
    const int i = blockIdx.x;
    const int p0 = pair_start[i], Ka = pair_cnt[i];
    const int TP = (T + 31) / 32 * 32, KP = (ka_max + 31) / 32 * 32;
    float *kt = sm, *ft = kt + KP * 65, *nks = ft + TP * 65, *nfs = nks + KP;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int r = lane & 31, kk = lane >> 5;
    const float *f = F + (int64_t)i * f_gstride;
    const int tilesT = TP / 32, tilesK = (Ka + 31) / 32, tiles = tilesT * tilesK;     // <= 16 (T, Ka <= 128)
    f32x16_ro acc[4];                       // tiles part, part + 4, ...: up to 4 per wave
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    float sqf[4][8], sqk[4][8];             // row 32 m + part + 4 j of F / K: squares of this lane's column, summed over the chunks
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) sqf[m][j] = sqk[m][j] = 0.f;
    for (int c0 = 0; c0 < H; c0 += 64) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {               // loads first (8 independent ones per thread), then the LDS stores
            if (32 * m >= TP) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int t = 32 * m + part + 4 * j; v[j] = t < T ? f[(int64_t)t * H + c0 + lane] : 0.f; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { ft[(32 * m + part + 4 * j) * 65 + lane] = v[j]; sqf[m][j] += v[j] * v[j]; }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (32 * m >= KP) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int a = 32 * m + part + 4 * j; v[j] = a < Ka ? Kmat[(int64_t)(p0 + a) * H + c0 + lane] : 0.f; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { kt[(32 * m + part + 4 * j) * 65 + lane] = v[j]; sqk[m][j] += v[j] * v[j]; }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tile = part + 4 * q;
            if (tile < tiles) {
                const int at = tile / tilesT, tt = tile - at * tilesT;
                const float *ka = kt + (32 * at + r) * 65 + kk, *fb = ft + (32 * tt + r) * 65 + kk;
                for (int c = 0; c < 64; c += 2) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[c], fb[c], acc[q], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // row norms: row 32 m + part + 4 j belongs to this wave alone
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = wave_sum(sqf[m][j]), b = wave_sum(sqk[m][j]);
            if (lane == 0) {
                if (32 * m < TP) nfs[32 * m + part + 4 * j] = a;
                if (32 * m < KP) nks[32 * m + part + 4 * j] = b;
            }
        }
    __syncthreads();
    const float eps = 1e-8f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tile = part + 4 * q;
        if (tile >= tiles) continue;
        const int at = tile / tilesT, tt = tile - at * tilesT;
        const int t = 32 * tt + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int a = 32 * at + 8 * (e >> 2) + 4 * kk + (e & 3);
            if (a < Ka && t < T) {
                const float c = acc[q][e] / (fmaxf(sqrtf(nfs[t]), eps) * fmaxf(sqrtf(nks[a]), eps));
                att[(int64_t)(p0 + a) * T + t] = (c + 1.0f) * 0.49f;
            }
        }
    }
}
int launch_cosine_attn_grouped(const float *F, int64_t f_gstride, const float *Kmat, const int32_t *pair_start, const int32_t *pair_cnt,
                               float *att, int n, int npairs, int T, int H, int ka_max, hipStream_t s) {
    if (n == 0) return 0;
    STAIR_ACCT("cosine_attn_grouped_kernel", ((int64_t)n * T * H + (int64_t)npairs * H + (int64_t)npairs * T) * 4);
    STAIR_CHECK(H % 64 == 0 && T <= 128 && ka_max <= 128, "grouped cosine: H % 64 == 0, T and pairs per instance <= 128");
    const int TP = (T + 31) / 32 * 32, KP = (ka_max + 31) / 32 * 32;
    const size_t shmem = ((size_t)(KP + TP) * 65 + KP + TP) * sizeof(float);
    if (shmem > 48 * 1024)
        STAIR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cosine_attn_grouped_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(cosine_attn_grouped_kernel, dim3(n), dim3(256), shmem, s, F, f_gstride, Kmat, pair_start, pair_cnt, att, n, T, H, ka_max);
    STAIR_LAUNCH_CHECK();
    return 0;
}
int launch_cosine_attn(const float *F, int64_t f_gstride, const int32_t *f_idx, const float *Kmat,
                       const int32_t *k_idx, float *att, const int32_t *out_idx, int npairs, int T, int H,
                       hipStream_t s) {
    if (npairs == 0) return 0;
    STAIR_ACCT("cosine_attn_kernel", ((int64_t)npairs * T * H + (int64_t)npairs * H + (int64_t)npairs * T) * 4);
    STAIR_CHECK(H % 4 == 0 && f_gstride % 4 == 0, "H and f_gstride must be multiples of 4");
    const int64_t rows = (int64_t)npairs * T;
    hipLaunchKernelGGL(cosine_attn_kernel, dim3((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock),
                       0, s, F, f_gstride, f_idx, Kmat, k_idx, att, out_idx, npairs, T, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// TemporalModule relate nets, modules.py:255-277 (definition) and :317-323 (use).
struct RelateW { const float *w[6]; };
// `len` (optional): frames of instance i's clip when the batch mixes clip lengths; rows keep the stride T, the nets see
// a sequence of L = len[i] frames (Conv1d 'same' pads with zeros beyond L, as it does for a clip of that length) and
// frames >= L of the output are zero.
__global__ void temporal_relate_kernel(const float *att, const int32_t *att_idx, const int32_t *att_k, float *out,
                                       const int32_t *out_idx, int n, int T, int mode, int conv, int ksize,
                                       RelateW W, const int32_t *len) {
    extern __shared__ float sm[];   // two ping-pong rows [T]
    float *x = sm, *y = sm + T;
    const int i = blockIdx.x;
    const int K = att_k[i];
    const int L = len ? len[i] : T;
    const float *a = att + (int64_t)att_idx[i] * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc += a[(int64_t)k * T + t];
        x[t] = t < L ? acc / (float)K : 0.0f;          // torch.mean(attention_scores, dim=0)
        y[t] = 0.0f;
    }
    __syncthreads();
    if (mode != 0) {
        for (int layer = 0; layer < 3; ++layer) {
            const float *w = W.w[2 * layer], *b = W.w[2 * layer + 1];
            if (conv) {
                // Conv1d(1,1,k,padding='same'): left pad (k-1)/2, the odd element goes right
                const int k = layer < 2 ? ksize : 2 * ksize + 1;
                const int left = (k - 1) / 2;
                for (int t = threadIdx.x; t < L; t += blockDim.x) {
                    float acc = b[0];
                    for (int j = 0; j < k; ++j) {
                        const int u = t + j - left;
                        if (u >= 0 && u < L) acc += w[j] * x[u];
                    }
                    y[t] = layer < 2 ? fmaxf(acc, 0.0f) : sigmoid_acc(acc);
                }
            } else {
                for (int t = threadIdx.x; t < T; t += blockDim.x) {
                    float acc = b[t];
                    for (int u = 0; u < T; ++u) acc += w[(int64_t)t * T + u] * x[u];
                    y[t] = layer < 2 ? fmaxf(acc, 0.0f) : sigmoid_acc(acc);
                }
            }
            __syncthreads();
            float *tmp = x; x = y; y = tmp;
        }
    }
    float *o = out + (int64_t)idx_or_id(out_idx, i) * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x) o[t] = x[t];
}
int launch_temporal_relate(const float *att, const int32_t *att_idx, const int32_t *att_k, float *out,
                           const int32_t *out_idx, int n, int T, int mode, int conv, int ksize,
                           const float *const w[6], hipStream_t s, const int32_t *len) {
    if (n == 0) return 0;
    STAIR_CHECK(mode >= 0 && mode <= 3, "mode must be 0..3");
    RelateW W;
    for (int i = 0; i < 6; ++i) W.w[i] = (mode != 0 && w) ? w[i] : nullptr;
    if (mode != 0) for (int i = 0; i < 6; ++i) STAIR_CHECK(W.w[i] != nullptr, "relate weights missing");
    STAIR_CHECK(!len || conv || mode == 0, "Linear(T,T) relate nets take one clip length (modules.py:266-277)");
    hipLaunchKernelGGL(temporal_relate_kernel, dim3(n), dim3(64), 2 * T * sizeof(float), s, att, att_idx, att_k, out,
                       out_idx, n, T, mode, conv, ksize, W, len);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ void layernorm_kernel(const float *Y, float *X, int64_t gstride, const int32_t *gidx, int n, int T, int H,
                                 const float *gamma, const float *beta, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= (int64_t)n * T) return;
    const int g = (int)(row / T), t = (int)(row - (int64_t)g * T);
    const float4 *y = reinterpret_cast<const float4 *>(Y + row * H);
    float4 *x = reinterpret_cast<float4 *>(X + (int64_t)idx_or_id(gidx, g) * gstride + (int64_t)t * H);
    const int n4 = H / 4;
    float sum = 0.f;
    for (int c = lane; c < n4; c += 64) { const float4 v = y[c]; sum += v.x + v.y + v.z + v.w; }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
    for (int c = lane; c < n4; c += 64) {
        const float4 v = y[c];
        const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
        sq += a * a + b * b + cc * cc + d * d;
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
    const float4 *g4 = reinterpret_cast<const float4 *>(gamma), *b4 = reinterpret_cast<const float4 *>(beta);
    for (int c = lane; c < n4; c += 64) {
        float4 v = y[c];
        const float4 gg = g4[c], bb = b4[c];
        v.x = (v.x - mean) * rstd * gg.x + bb.x; v.y = (v.y - mean) * rstd * gg.y + bb.y;
        v.z = (v.z - mean) * rstd * gg.z + bb.z; v.w = (v.w - mean) * rstd * gg.w + bb.w;
        x[c] = v;
    }
}
int launch_layernorm(const float *Y, float *X, int64_t gstride, const int32_t *gidx, int n, int T, int H, const float *gamma,
                     const float *beta, float eps, hipStream_t s) {
    if (n == 0) return 0;
    STAIR_ACCT("layernorm_kernel", 2ll * n * T * H * 4);
    const int64_t rows = (int64_t)n * T;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0,
                       s, Y, X, gstride, gidx, n, T, H, gamma, beta, eps);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ void sum_rows_kernel(const float *X, float *out, int n, int T, int H, const int32_t *len) {
    const int g = blockIdx.x;
    const float *x = X + (int64_t)g * T * H;
    const int L = len ? len[g] : T;                 // frames of this instance's clip (rows L..T-1 are padding)
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float acc = 0.f;
        for (int t = 0; t < L; ++t) acc += x[(int64_t)t * H + c];
        out[(int64_t)g * H + c] = acc;
    }
}
// float4 form: thread = (16-byte column piece, row phase); the phases' partial sums meet in LDS
__global__ void sum_rows_v4_kernel(const float *X, float *out, int T, int H4, const int32_t *len) {
    extern __shared__ __attribute__((aligned(16))) float part[];          // [phases][H]
    using v4 = __attribute__((ext_vector_type(4))) float;
    const int g = blockIdx.x;
    const int L = len ? len[g] : T;
    const int phases = blockDim.x / H4, c = threadIdx.x % H4, ph = threadIdx.x / H4;
    const v4 *x = reinterpret_cast<const v4 *>(X) + (int64_t)g * T * H4 + c;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    if (ph < phases)
        for (int t = ph; t < L; t += phases) acc += x[(int64_t)t * H4];
    if (ph < phases) reinterpret_cast<v4 *>(part)[ph * H4 + c] = acc;
    __syncthreads();
    if (ph == 0) {
        for (int q = 1; q < phases; ++q) acc += reinterpret_cast<v4 *>(part)[q * H4 + c];
        reinterpret_cast<v4 *>(out)[(int64_t)g * H4 + c] = acc;
    }
}
int launch_sum_rows(const float *X, float *out, int n, int T, int H, hipStream_t s, const int32_t *len) {
    if (n == 0) return 0;
    if (H % 4 == 0 && H / 4 <= 256 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        STAIR_ACCT("sum_rows_kernel", ((int64_t)n * T * H + (int64_t)n * H) * 4);
        const int H4 = H / 4, phases = std::max(1, 512 / H4), threads = phases * H4;
        hipLaunchKernelGGL(sum_rows_v4_kernel, dim3(n), dim3(threads), (size_t)phases * H * sizeof(float), s, X, out, T, H4, len);
        STAIR_LAUNCH_CHECK();
        return 0;
    }
    STAIR_ACCT("sum_rows_kernel", ((int64_t)n * T * H + (int64_t)n * H) * 4);
    hipLaunchKernelGGL(sum_rows_kernel, dim3(n), dim3(kBlock), 0, s, X, out, n, T, H, len);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ void rowdot_sigmoid_kernel(const float *X, int n, int T, int H, const float *w, const float *b,
                                      const float *extra, float *out, const int32_t *out_idx, int64_t out_gstride) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= (int64_t)n * T) return;
    const int g = (int)(row / T), t = (int)(row - (int64_t)g * T);
    const float4 *x = reinterpret_cast<const float4 *>(X + row * H);
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    float d = 0.f;
    for (int c = lane; c < H / 4; c += 64) d += dot4(x[c], w4[c]);
    d = wave_sum(d);
    if (lane == 0) out[(int64_t)idx_or_id(out_idx, g) * out_gstride + t] = sigmoid_acc(d + b[0] + (extra ? extra[g] : 0.f));
}
int launch_rowdot_sigmoid(const float *X, int n, int T, int H, const float *w, const float *b, const float *extra,
                          float *out, const int32_t *out_idx, int64_t out_gstride, hipStream_t s) {
    if (n == 0) return 0;
    STAIR_ACCT("rowdot_sigmoid_kernel", ((int64_t)n * T * H + (int64_t)n * T) * 4);
    const int64_t rows = (int64_t)n * T;
    hipLaunchKernelGGL(rowdot_sigmoid_kernel, dim3((unsigned)((rows + kWavesPerBlock - 1) / kWavesPerBlock)),
                       dim3(kBlock), 0, s, X, n, T, H, w, b, extra, out, out_idx, out_gstride);
    STAIR_LAUNCH_CHECK();
    return 0;
}

__global__ void vecdot_kernel(const float *V, const int32_t *idx, const float *w, float *out, int n, int H) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float4 *x = reinterpret_cast<const float4 *>(V + (int64_t)idx_or_id(idx, i) * H);
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    float d = 0.f;
    for (int c = lane; c < H / 4; c += 64) d += dot4(x[c], w4[c]);
    d = wave_sum(d);
    if (lane == 0) out[i] = d;
}
int launch_vecdot(const float *V, const int32_t *idx, const float *w, float *out, int n, int H, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(vecdot_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, V, idx, w,
                       out, n, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ void relate_softmax_kernel(float *att, const int32_t *in_idx, const int32_t *out_idx, const float *beta,
                                      float sign, int n, int T, const int32_t *len) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *x = att + (int64_t)in_idx[i] * T;
    float *o = att + (int64_t)out_idx[i] * T;
    const int L = len ? len[i] : T;                 // softmax over the clip's own frames (beta[:T] of modules.py:431 with T = L)
    float m = -INFINITY;
    for (int t = lane; t < L; t += 64) m = fmaxf(m, x[t] + sign * beta[t]);
    m = wave_max(m);
    float sum = 0.f;
    for (int t = lane; t < L; t += 64) sum += expf(x[t] + sign * beta[t] - m);
    sum = wave_sum(sum);
    for (int t = lane; t < T; t += 64) o[t] = t < L ? expf(x[t] + sign * beta[t] - m) / sum : 0.f;
}
int launch_relate_softmax(float *att, const int32_t *in_idx, const int32_t *out_idx, const float *beta, float sign,
                          int n, int T, hipStream_t s, const int32_t *len) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(relate_softmax_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, att,
                       in_idx, out_idx, beta, sign, n, T, len);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ void eltwise_kernel(int mode, float *base, const int32_t *ia, const int32_t *ib, const int32_t *io, int n,
                               int len) {
    const int i = blockIdx.x;
    const float *a = base + (int64_t)ia[i] * len, *b = base + (int64_t)ib[i] * len;
    float *o = base + (int64_t)io[i] * len;
    for (int c = threadIdx.x; c < len; c += blockDim.x) o[c] = mode == 0 ? fminf(a[c], b[c]) : fabsf(a[c] - b[c]);
}
int launch_eltwise(int mode, float *base, const int32_t *ia, const int32_t *ib, const int32_t *io, int n, int len,
                   hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(eltwise_kernel, dim3(n), dim3(128), 0, s, mode, base, ia, ib, io, n, len);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ void attnvideo_kernel(float *map, const int32_t *in_idx, const float *att, const int32_t *att_idx,
                                 const int32_t *out_idx, int n, int T, int H) {
    const int64_t tile = (int64_t)T * H / 4;
    const int64_t total = (int64_t)n * tile;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / tile);
        const int64_t r = e - (int64_t)i * tile;
        const int t = (int)(r / (H / 4));
        const float a = att[(int64_t)att_idx[i] * T + t];
        float4 v = reinterpret_cast<const float4 *>(map + (int64_t)in_idx[i] * T * H)[r];
        v.x *= a; v.y *= a; v.z *= a; v.w *= a;
        reinterpret_cast<float4 *>(map + (int64_t)out_idx[i] * T * H)[r] = v;
    }
}
int launch_attnvideo(float *map, const int32_t *in_idx, const float *att, const int32_t *att_idx,
                     const int32_t *out_idx, int n, int T, int H, hipStream_t s) {
    if (n == 0) return 0;
    STAIR_ACCT("attnvideo_kernel", (2ll * n * T * H + (int64_t)n * T) * 4);
    const int64_t total = (int64_t)n * T * H / 4;
    const int blocks = (int)std::min<int64_t>((total + kBlock - 1) / kBlock, 256 * 8);
    hipLaunchKernelGGL(attnvideo_kernel, dim3(blocks), dim3(kBlock), 0, s, map, in_idx, att, att_idx, out_idx, n, T, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// ChooseModule, modules.py:40-56.  The reference branches on the host (`if cos1 > cos2`), which is a
// device sync per call; here the select stays on the device.  Cosines are formed the way ATen does
// (normalise each operand by max(||.||, eps), then dot) so that a near-tie resolves identically.
__global__ void choose_kernel(float *vec, const int32_t *k1, const int32_t *k2, const int32_t *q, const int32_t *out,
                              int n, int H) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *a = vec + (int64_t)k1[i] * H, *b = vec + (int64_t)k2[i] * H, *c = vec + (int64_t)q[i] * H;
    float na = 0.f, nb = 0.f, nc = 0.f;
    for (int e = lane; e < H; e += 64) { na += a[e] * a[e]; nb += b[e] * b[e]; nc += c[e] * c[e]; }
    na = fmaxf(sqrtf(wave_sum(na)), 1e-8f); nb = fmaxf(sqrtf(wave_sum(nb)), 1e-8f); nc = fmaxf(sqrtf(wave_sum(nc)), 1e-8f);
    float da = 0.f, db = 0.f;
    for (int e = lane; e < H; e += 64) {
        const float cn = c[e] / nc;
        da += (a[e] / na) * cn; db += (b[e] / nb) * cn;
    }
    da = wave_sum(da); db = wave_sum(db);
    const float *src = da > db ? a : b;
    float *o = vec + (int64_t)out[i] * H;
    for (int e = lane; e < H; e += 64) o[e] = src[e];
}
int launch_choose(float *vec, const int32_t *k1, const int32_t *k2, const int32_t *q, const int32_t *out, int n,
                  int H, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(choose_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, vec, k1, k2,
                       q, out, n, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// SuperlativeModule pooling, modules.py:244-247.
__global__ void superlative_pool_kernel(const float *S, const float *rowbase, const int32_t *row_id,
                                        const int32_t *row_start, const int32_t *row_cnt, int is_min, float *out,
                                        int n, int T, int H, const int32_t *len) {
    extern __shared__ float wsm[];   // [Ka]
    const int i = blockIdx.x;
    const int r0 = row_start[i], Ka = row_cnt[i];
    const int L = len ? len[i] : T;                 // frames of the clip: the score rows are summed over them only
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = wave; a < Ka; a += kWavesPerBlock) {
        const float *s = S + (int64_t)(r0 + a) * T;
        float acc = 0.f;
        for (int t = lane; t < L; t += 64) acc += s[t];
        acc = wave_sum(acc);
        if (lane == 0) wsm[a] = acc;
    }
    __syncthreads();
    if (wave == 0) {
        float m = -INFINITY;
        for (int a = lane; a < Ka; a += 64) m = fmaxf(m, wsm[a]);
        m = wave_max(m);
        float sum = 0.f;
        for (int a = lane; a < Ka; a += 64) sum += expf(wsm[a] - m);
        sum = wave_sum(sum);
        for (int a = lane; a < Ka; a += 64) {
            const float w = expf(wsm[a] - m) / sum;
            wsm[a] = is_min ? 1.0f - w : w;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float acc = 0.f;
        for (int a = 0; a < Ka; ++a) acc += wsm[a] * rowbase[(int64_t)row_id[r0 + a] * H + c];
        out[(int64_t)i * H + c] = acc;
    }
}
int launch_superlative_pool(const float *S, const float *rowbase, const int32_t *row_id, const int32_t *row_start,
                            const int32_t *row_cnt, int is_min, float *out, int n, int T, int H, hipStream_t s, const int32_t *len) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(superlative_pool_kernel, dim3(n), dim3(kBlock), (size_t)std::max(T, 2) * sizeof(float), s, S,
                       rowbase, row_id, row_start, row_cnt, is_min, out, n, T, H, len);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ void argmax_kernel(const float *logits, int32_t *out, int n, int A) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *x = logits + (int64_t)i * A;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < A; c += 64)
        if (x[c] > best) { best = x[c]; bi = c; }     // first maximum, like torch.argmax
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) out[i] = bi;
}
int launch_argmax(const float *logits, int32_t *out, int n, int A, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(argmax_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, logits, out,
                       n, A);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// L2Normalize, module_net.py:211-216: x / max(||x||_2, 1e-12) (F.normalize, dim 0 of a [H] vector)
__global__ void l2norm_kernel(const float *x, float *out, int n, int H) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n) return;
    const float *a = x + (int64_t)i * H;
    float ss = 0.f;
    for (int e = lane; e < H; e += 64) ss += a[e] * a[e];
    const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    for (int e = lane; e < H; e += 64) out[(int64_t)i * H + e] = a[e] / nrm;
}
int launch_l2norm(const float *x, float *out, int n, int H, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(l2norm_kernel, dim3((n + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, s, x, out, n, H);
    STAIR_LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// nn.Dropout in training mode (the `D` positions of modules.py: after the ReLU of every hidden Linear, after the dense
// layers of FilterFrame / Temporal, after HasItem's Sigmoid; p = config['dropout'], args.py:31).  torch draws its mask
// from a Philox stream that cannot be reproduced here; the mask below is a counter-based hash of (seed, site,
// element), i.e. a pure function: the backward pass needs no stored mask (a dropped element is an exact zero of the
// saved activation) and a step can be replayed.
// ---------------------------------------------------------------------------------------------
// (drop_hash4 / drop_keep: csrc/common.h -- the fused tile operator draws the same bits)
__global__ void dropout_rows_kernel(float *X, int64_t gstride, const int32_t *gidx, int groups, int64_t rowlen, uint32_t thresh,
                                    float inv_keep, uint64_t seed, uint32_t site) {
    const int64_t total = (int64_t)groups * rowlen;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(e / rowlen);
        const int64_t r = e - (int64_t)g * rowlen;
        float *x = X + (int64_t)idx_or_id(gidx, g) * gstride + r;
        *x = drop_keep(drop_hash4(seed, site, (uint64_t)e >> 2), (int)(e & 3), thresh) ? *x * inv_keep : 0.0f;
    }
}
int launch_dropout_rows(float *X, int64_t gstride, const int32_t *gidx, int groups, int64_t rowlen, float p, uint64_t seed,
                        uint32_t site, hipStream_t s) {
    if (groups == 0 || rowlen == 0 || p <= 0.0f) return 0;
    STAIR_CHECK(p < 1.0f, "dropout probability must be below 1");
    const int64_t total = (int64_t)groups * rowlen;
    const uint32_t thresh = (uint32_t)(p * 65536.0f);            // drop when the element's 16 hash bits are below p * 2^16
    hipLaunchKernelGGL(dropout_rows_kernel, dim3((unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, 8192)), dim3(kBlock), 0,
                       s, X, gstride, gidx, groups, rowlen, thresh, 1.0f / (1.0f - p), seed, site);
    STAIR_LAUNCH_CHECK();
    return 0;
}

// ---- zero fill ------------------------------------------------------------------------------------------------------------
// Every "start from zero" of the library (status / queue words, hand-off flags of the cooperative recurrences, gradient arenas)
// is THIS kernel, never hipMemsetAsync: a memset recorded into a torch-captured hipGraph was observed to fill its target with
// stale kernel-argument bytes instead of zeros from the second replay on (profiles/r04_queue_probe.json);
// a kernel node replays like every other launch of the pass.  16-byte stores over the aligned body, words at both ends.
__global__ void zero_fill_kernel(uint32_t *p, int64_t head, int64_t body16, int64_t tail) {
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
    uint4 *b = reinterpret_cast<uint4 *>(p + head);
    for (int64_t i = i0; i < body16; i += step) b[i] = uint4{0u, 0u, 0u, 0u};
    if (i0 < head) p[i0] = 0u;
    if (i0 < tail) p[head + 4 * body16 + i0] = 0u;
}

int launch_zero(void *ptr, int64_t bytes, hipStream_t s) {
    if (bytes <= 0) return 0;
    STAIR_CHECK(ptr && bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(ptr) & 3) == 0, "zero fill: 4-byte aligned words");
    const int64_t n = bytes / 4;
    const int64_t head = std::min<int64_t>(n, ((16 - (int64_t)(reinterpret_cast<uintptr_t>(ptr) & 15)) & 15) / 4);
    const int64_t body16 = (n - head) / 4, tail = n - head - 4 * body16;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((body16 + 255) / 256, 4096));
    STAIR_ACCT("zero_fill", bytes);
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint32_t *>(ptr), head, body16, tail);
    STAIR_LAUNCH_CHECK();
    return 0;
}

}  // namespace stair

extern "C" int stair_l2normalize_fwd(const float *x, float *out, int32_t n, int32_t H, stair_stream stream) {
    return stair::launch_l2norm(x, out, n, H, static_cast<hipStream_t>(stream));
}

extern "C" int stair_cosine_attn_fwd(const float *F, int64_t f_gstride, const int32_t *f_idx, const float *Kmat,
                                     const int32_t *k_idx, float *att, const int32_t *out_idx, int32_t npairs,
                                     int32_t T, int32_t H, stair_stream stream) {
    return stair::launch_cosine_attn(F, f_gstride, f_idx, Kmat, k_idx, att, out_idx, npairs, T, H,
                                     static_cast<hipStream_t>(stream));
}

extern "C" int stair_temporal_relate_fwd(const float *att, const int32_t *att_idx, const int32_t *att_k, float *out,
                                         const int32_t *out_idx, int32_t n, int32_t T, int32_t mode, int32_t conv,
                                         int32_t ksize, const float *const w[6], stair_stream stream) {
    return stair::launch_temporal_relate(att, att_idx, att_k, out, out_idx, n, T, mode, conv, ksize, w,
                                         static_cast<hipStream_t>(stream));
}

extern "C" int stair_dropout_fwd(float *x, int64_t gstride, const int32_t *gidx, int32_t groups, int64_t rowlen, float p,
                                 uint64_t seed, uint32_t site, stair_stream stream) {
    STAIR_CHECK(x != nullptr && groups >= 0 && rowlen >= 0 && p >= 0.0f, "bad argument");
    return stair::launch_dropout_rows(x, gstride, gidx, groups, rowlen, p, seed, site, static_cast<hipStream_t>(stream));
}
