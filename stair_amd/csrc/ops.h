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

int launch_cosine_attn_grouped(const float *F, int64_t f_gstride, const float *Kmat, const int32_t *pair_start, const int32_t *pair_cnt,
                               float *att, int n, int npairs, int T, int H, int ka_max, hipStream_t s);
int launch_cosine_attn(const float *F, int64_t f_gstride, const int32_t *f_idx, const float *Kmat,
                       const int32_t *k_idx, float *att, const int32_t *out_idx, int npairs, int T, int H,
                       hipStream_t s);

int launch_temporal_relate(const float *att, const int32_t *att_idx, const int32_t *att_k, float *out,
                           const int32_t *out_idx, int n, int T, int mode, int conv, int ksize,
                           const float *const w[6], hipStream_t s, const int32_t *len = nullptr);

// LayerNorm over H: rows of Y [n*T, H] (contiguous) -> tiles X + gidx[g]*gstride, [T,H]      modules.py:283,327
int launch_layernorm(const float *Y, float *X, int64_t gstride, const int32_t *gidx, int n, int T, int H, const float *gamma,
                     const float *beta, float eps, hipStream_t s);

// out[g][:] = sum_t X[g][t][:]                                                 modules.py:374,376
// `len` (optional, here and below): frames per instance when a batch mixes clip lengths; rows keep the stride T
int launch_sum_rows(const float *X, float *out, int n, int T, int H, hipStream_t s, const int32_t *len = nullptr);

// out[out_idx[g]*out_gstride + t] = sigmoid(X[g][t] . w + b[0] + (extra ? extra[g] : 0))
int launch_rowdot_sigmoid(const float *X, int n, int T, int H, const float *w, const float *b, const float *extra,
                          float *out, const int32_t *out_idx, int64_t out_gstride, hipStream_t s);
// out[i] = V[idx[i]] . w
int launch_vecdot(const float *V, const int32_t *idx, const float *w, float *out, int n, int H, hipStream_t s);

// att[out[i]] = softmax_T(att[in[i]] + sign * beta[:T])                         modules.py:417-435
int launch_relate_softmax(float *att, const int32_t *in_idx, const int32_t *out_idx, const float *beta, float sign,
                          int n, int T, hipStream_t s, const int32_t *len = nullptr);

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
                            const int32_t *row_cnt, int is_min, float *out, int n, int T, int H, hipStream_t s,
                            const int32_t *len = nullptr);

int launch_l2norm(const float *x, float *out, int n, int H, hipStream_t s);

// ---- backward kernels (csrc/rowops_bwd.hip) --------------------------------------------------
int launch_mask_relu(float *dst, const float *G, int64_t g_gs, const int32_t *g_idx, const float *Y, int64_t y_gs,
                     const int32_t *y_idx, int groups, int rowlen, hipStream_t s, float scale = 1.0f);
int launch_bcast_mask_relu(float *dst, const float *dsum, const float *Y, int groups, int T, int H, hipStream_t s,
                           float scale = 1.0f, const int32_t *len = nullptr);
int launch_span_mean_bwd_rows(float *dtok, int64_t ld, int rows, const int32_t *row_ptr, const int32_t *row_span, const int32_t *count,
                              const float *dvec, const int32_t *out_idx, int H, hipStream_t s);
int launch_rowsum_small(const float *X, float *out, int n, int T, hipStream_t s);
int launch_grad_fanin(const int32_t *tab, int n, float *gvec, float *gmap, float *gatt, int H, int T, hipStream_t s);
int launch_scatter_add_rows(float *dst, const int32_t *dst_idx, const float *src, int n, int len, float scale, hipStream_t s);
int launch_pack_bwd(int mode, const float *A, const int32_t *ia, const float *B, const int32_t *ib, const float *g,
                    float *dA, float *dB, int n, int H, hipStream_t s);
int launch_span_mean_bwd(float *dtok, int64_t ld, const int32_t *start, const int32_t *count, const float *dvec,
                         const int32_t *out_idx, int n, int H, hipStream_t s);
int launch_cosine_attn_bwd(const float *F, int64_t f_gs, const int32_t *f_idx, const float *Kmat, const int32_t *k_idx,
                           const float *datt, const int32_t *out_idx, float *dF, float *dK, int npairs, int T, int H,
                           hipStream_t s, const int32_t *gf_idx = nullptr, const int32_t *gk_idx = nullptr);
int launch_cosine_attn_bwd_grouped(const float *F, const float *Kmat, const float *score, const int32_t *score_idx,
                                   const float *dscore, const int32_t *dscore_idx, const int32_t *pair_start,
                                   const int32_t *pair_cnt, float *dF, float *dK, float *nf_ws, float *nk_ws, int n, int npairs,
                                   int T, int H, int ka_max, hipStream_t s);
int launch_temporal_relate_bwd(const float *att, const int32_t *att_idx, const int32_t *att_k, const float *drel,
                               const int32_t *rel_idx, float *datt, int n, int T, int mode, int conv, int ksize,
                               const float *const w[6], float *const dw[6], hipStream_t s, const int32_t *len = nullptr,
                               const int32_t *gatt_idx = nullptr);
int launch_layernorm_bwd(const float *dOut, int64_t g_gs, const int32_t *g_idx, const float *Y, int n, int T, int H,
                         const float *gamma, float eps, float *dZ, float *stats, float *dgamma, float *dbeta, hipStream_t s,
                         float scale = 1.0f);
int launch_rowscale_bwd(const float *G, const float *X, int64_t x_gs, const int32_t *x_idx, const float *rs, int64_t rs_gs,
                        const int32_t *rs_idx, float *dX, float *drs, int n, int T, int H, hipStream_t s, const int32_t *gx_idx = nullptr);
int launch_rowdot_sigmoid_bwd(const float *dOut, int64_t o_gs, const int32_t *o_idx, const float *A, int64_t a_gs,
                              const int32_t *a_idx, const float *w, float *dXdst, int add_mode, float *dpre_out,
                              float *dextra, int n, int T, int H, hipStream_t s, float keep = 1.0f);
// training-mode nn.Dropout(p) applied in place to `groups` rows of `rowlen` floats (row g at X + idx[g] * gstride):
// x <- keep(seed, site, element) ? x / (1 - p) : 0 with a counter-based hash, so the mask is a pure function of the seed
int launch_dropout_rows(float *X, int64_t gstride, const int32_t *gidx, int groups, int64_t rowlen, float p, uint64_t seed,
                        uint32_t site, hipStream_t s);
int launch_weighted_colsum(const float *X, int64_t ld, const int32_t *x_idx, const float *scale, float *out, int rows, int H,
                           hipStream_t s);
int launch_axpy_rows(float *dV, const int32_t *idx, const float *scale, const float *w, int n, int H, hipStream_t s);
int launch_sum_all(const float *x, float *out, int n, hipStream_t s);
int launch_relate_softmax_bwd(const float *att, float *datt, const int32_t *in_idx, const int32_t *out_idx, float *dbeta,
                              float sign, int n, int T, hipStream_t s, const int32_t *len = nullptr, const int32_t *gin_idx = nullptr);
int launch_eltwise_bwd(int mode, const float *base, float *dbase, const int32_t *ia, const int32_t *ib, const int32_t *io,
                       int n, int len, hipStream_t s, const int32_t *gia = nullptr, const int32_t *gib = nullptr);
int launch_attnvideo_bwd(const float *map, float *dmap, const float *att, float *datt, const int32_t *in_idx,
                         const int32_t *att_idx, const int32_t *out_idx, int n, int T, int H, hipStream_t s,
                         const int32_t *gin_idx = nullptr, const int32_t *gatt_idx = nullptr);
int launch_choose_bwd(const float *vec, float *dvec, const int32_t *k1, const int32_t *k2, const int32_t *q,
                      const int32_t *out, int n, int H, hipStream_t s, const int32_t *gk1 = nullptr, const int32_t *gk2 = nullptr);
int launch_superlative_pool_bwd(const float *S, const float *rowbase, float *drowbase, const int32_t *row_id,
                                const int32_t *row_start, const int32_t *row_cnt, int is_min, const float *dpre, float *dS,
                                int n, int T, int H, hipStream_t s, const int32_t *len = nullptr, const int32_t *g_row_id = nullptr);
int launch_scale_rows(float *G, const float *rs, int64_t rows, int H, hipStream_t s);
int launch_ce_loss(const float *logits, const int32_t *answers, float scale, float *loss, float *dlogits, int n, int A,
                   hipStream_t s);
int launch_adam(float *p, const float *g, float *m, float *v, const int32_t *seg_of_block, const int32_t *touched,
                const float *step_of_seg, float lr, float b1, float b2, float eps, float wd, int64_t n, const uint32_t *guard, hipStream_t s);
int launch_argmax(const float *logits, int32_t *out, int n, int A, hipStream_t s);

}  // namespace stair
