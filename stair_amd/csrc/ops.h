// Internal launchers of the HBM-bound row kernels (csrc/rowops.hip); see each kernel for the
// reference lines it replaces.  A null index array means identity (i -> i).
#pragma once
#include "common.h"

namespace stair {

// vec[out_idx[i]][:] = mean(tok[start[i] .. start[i]+count[i]-1][:])      module_net.py:128-129
int launch_span_mean(const float *tok, int64_t ld, const int32_t *start, const int32_t *count, float *vec,
                     const int32_t *out_idx, int n, int H, hipStream_t s);

enum PackMode { PACK_CAT2 = 0, PACK_EXISTS = 1, PACK_XOR = 2 };
// out[i] = CAT2: [a,b]   EXISTS: [a, b, a*b]   XOR: [|a-b|, a, b]   with a = A[ia[i]], b = B[ib[i]]
int launch_pack(int mode, const float *A, const int32_t *ia, const float *B, const int32_t *ib, float *out, int n,
                int H, hipStream_t s);

int launch_cosine_attn(const float *F, int64_t f_gstride, const int32_t *f_idx, const float *Kmat,
                       const int32_t *k_idx, float *att, const int32_t *out_idx, int npairs, int T, int H,
                       hipStream_t s);

int launch_temporal_relate(const float *att, const int32_t *att_idx, const int32_t *att_k, float *out,
                           const int32_t *out_idx, int n, int T, int mode, int conv, int ksize,
                           const float *const w[6], hipStream_t s);

// in-place LayerNorm over H of every row of tiles X + gidx[g]*gstride, [T,H]      modules.py:283,327
int launch_layernorm(float *X, int64_t gstride, const int32_t *gidx, int n, int T, int H, const float *gamma,
                     const float *beta, float eps, hipStream_t s);

// out[g][:] = sum_t X[g][t][:]                                                 modules.py:374,376
int launch_sum_rows(const float *X, float *out, int n, int T, int H, hipStream_t s);

// out[out_idx[g]*out_gstride + t] = sigmoid(X[g][t] . w + b[0] + (extra ? extra[g] : 0))
int launch_rowdot_sigmoid(const float *X, int n, int T, int H, const float *w, const float *b, const float *extra,
                          float *out, const int32_t *out_idx, int64_t out_gstride, hipStream_t s);
// out[i] = V[idx[i]] . w
int launch_vecdot(const float *V, const int32_t *idx, const float *w, float *out, int n, int H, hipStream_t s);

// att[out[i]] = softmax_T(att[in[i]] + sign * beta[:T])                         modules.py:417-435
int launch_relate_softmax(float *att, const int32_t *in_idx, const int32_t *out_idx, const float *beta, float sign,
                          int n, int T, hipStream_t s);

// mode 0: min(a,b) (AndModule :7-12); mode 1: |a-b| (XorFrameModule :75-80); rows of `len` floats
int launch_eltwise(int mode, float *base, const int32_t *ia, const int32_t *ib, const int32_t *io, int n, int len,
                   hipStream_t s);

// map[out][t][:] = att[a][t] * map[in][t][:]                                    modules.py:330-340
int launch_attnvideo(float *map, const int32_t *in_idx, const float *att, const int32_t *att_idx,
                     const int32_t *out_idx, int n, int T, int H, hipStream_t s);

// vec[out] = cos(k1,q) > cos(k2,q) ? vec[k1] : vec[k2]                          modules.py:40-56
int launch_choose(float *vec, const int32_t *k1, const int32_t *k2, const int32_t *q, const int32_t *out, int n,
                  int H, hipStream_t s);

// Superlative pooling: w = softmax_a(sum_t S[a][t]) (1-w for min); out[i] = sum_a w_a * rows[row_id[a]]
int launch_superlative_pool(const float *S, const float *rowbase, const int32_t *row_id, const int32_t *row_start,
                            const int32_t *row_cnt, int is_min, float *out, int n, int T, int H, hipStream_t s);

int launch_l2norm(const float *x, float *out, int n, int H, hipStream_t s);
int launch_argmax(const float *logits, int32_t *out, int n, int A, hipStream_t s);

}  // namespace stair
