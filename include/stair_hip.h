/*
 * stair_hip.h -- C ABI of libstair_hip.so, the MI355X (gfx950) executor for STAIR's
 * neural-module-network hot path.
 *
 * The reference (yellow-binary-tree/STAIR) is pure Python and has no FFI of its own; the boundary
 * this library replaces is the Python-level one of
 *     /root/reference/video_nmn/module_net.py:65-145   VideoNMN.forward (stack interpreter)
 *     /root/reference/video_nmn/module_net.py:151-163  encode_question / encode_video (bi-LSTM)
 *     /root/reference/video_nmn/modules.py:7-465       the 18 registered module operators
 * Each entry point below names the reference lines it stands in for.  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer named *_dev / documented "device" is a HIP device pointer owned by the CALLER
 *     (allocated by torch in the Python host); the library never allocates, frees or retains
 *     caller memory beyond the duration documented per call.  Weights are BORROWED until the ctx
 *     is destroyed or the weight is re-set.
 *   - `stream` is a hipStream_t passed as void* (the caller's current torch-ROCm stream).  All
 *     work is enqueued on it; nothing synchronises the device.
 *   - return value 0 = success; non-zero = error, message via stair_last_error() (thread local).
 *   - one thread at a time per ctx; several contexts per process are supported (one per GPU and thread, or side by side on one
 *     GPU): policy settings can be overridden per context (stair_ctx_set_option), call-scoped state is thread-local.
 *   - all floating-point data is IEEE fp32, row-major.
 */
#ifndef STAIR_HIP_H
#define STAIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STAIR_ABI_VERSION 6

typedef struct stair_ctx stair_ctx;
typedef struct stair_plan stair_plan;
typedef void *stair_stream; /* hipStream_t */

/* Model configuration: the keys of the reference's config dict that change shapes
 * (/root/reference/train_module.py:304-310). */
typedef struct stair_config {
    int32_t hidden_size;         /* H, multiple of 64, <= 512 (Hh = H/2 tiles in blocks of 32) */
    int32_t video_size;          /* V, multiple of 4 */
    int32_t text_size;           /* E (300), multiple of 4 */
    int32_t answer_vocab_length; /* A */
    int32_t max_video_length;    /* selects Conv1d (>32) or Linear(T,T) Temporal nets, modules.py:255 */
    int32_t object_types;        /* O, FilterFrame pretrain head width */
    int32_t have_pretrain_head;  /* 0/1 */
} stair_config;

/* program token codes (stair_amd/spec.py mirrors these): module tokens 0..17 in the order of
 * NAME_TO_MODULE (modules.py:446-465), keyword tokens 100.., everything else = span mean. */
enum stair_token {
    STAIR_OP_AND = 0, STAIR_OP_ATTNVIDEO, STAIR_OP_CHOOSE, STAIR_OP_COMPARE, STAIR_OP_EQUALS,
    STAIR_OP_EXISTS, STAIR_OP_EXISTSFRAME, STAIR_OP_FILTER, STAIR_OP_FILTERFRAME, STAIR_OP_HASITEM,
    STAIR_OP_LOCALIZE, STAIR_OP_RELATE, STAIR_OP_SUPERLATIVE, STAIR_OP_TEMPORAL, STAIR_OP_TOACTION,
    STAIR_OP_XOR, STAIR_OP_XORFRAME, STAIR_OP_ARRAY2,
    STAIR_OP_COUNT = 18,
    STAIR_KW_FORWARD = 100, STAIR_KW_BACKWARD, STAIR_KW_WHILE, STAIR_KW_BETWEEN, STAIR_KW_BEFORE,
    STAIR_KW_AFTER, STAIR_KW_MAX, STAIR_KW_MIN, STAIR_KW_START, STAIR_KW_END, STAIR_KW_VIDEO,
    STAIR_KW_ACTIONS, STAIR_KW_OBJECTS, STAIR_KW_RELATIONS,
    STAIR_TOK_SPAN = 200
};

/* value kinds a program node can produce (returned by stair_plan_node) */
enum stair_value_kind {
    STAIR_VAL_STR = 0,   /* keyword string, no storage */
    STAIR_VAL_VEC = 1,   /* [H]      in the vec arena,  slot = row                        */
    STAIR_VAL_MAP = 2,   /* [T,H]    in the map arena,  slot = tile                       */
    STAIR_VAL_ATT = 3,   /* [K,T]    in the att arena,  slot = first row, aux = K         */
    STAIR_VAL_FRAME = 4, /* [T]      in the att arena,  slot = row                        */
    STAIR_VAL_PAIR = 5   /* [2,H]    two vec rows (Array2), slot = first row, aux = second row */
};

int stair_abi_version(void);
const char *stair_last_error(void);

/* Measurement aid: while enabled, every launch of an HBM-bound row kernel adds the bytes it must move (inputs once +
 * outputs once) to a per-kernel table, and every GEMM / recurrence launcher adds the kernel variant it selected with its
 * algorithmic flops; stair_acct_dump writes "kernel launches bytes flops" lines.  tools/row_kernels.py pairs the
 * table with rocprofv3 kernel durations (GB/s per kernel against the 8 TB/s HBM peak).  Off by default, process-wide. */
void stair_acct_enable(int32_t on);
int stair_acct_dump(char *buf, int32_t cap);

/* ---- context: configuration + borrowed weight pointers ------------------------------------ */
int stair_ctx_create(const stair_config *cfg, stair_ctx **out);
void stair_ctx_destroy(stair_ctx *ctx);

/* Weight table: ids 0..count-1 enumerate the reference's state_dict keys in state_dict order,
 * aliases excluded (module_net.py:27-53; Superlative.localize_module.* IS Localize.*). */
int stair_weight_count(const stair_ctx *ctx);
const char *stair_weight_name(const stair_ctx *ctx, int id);
int64_t stair_weight_numel(const stair_ctx *ctx, int id);
int stair_ctx_set_weight(stair_ctx *ctx, int id, const float *dev_ptr, int64_t numel);
/* Gradient buffer of weight `id` (same shape), needed only for stair_plan_backward; gradients are ACCUMULATED
 * into it, the caller zeroes it (optimizer.zero_grad(), train_module.py:411). */
int stair_ctx_set_grad(stair_ctx *ctx, int id, float *dev_ptr, int64_t numel);

/* Arithmetic of the large contractions (process-wide; default from the environment variable STAIR_MATMUL =
 * "f32" | "bf16x3" | "bf16", else bf16x3):
 *   STAIR_MATMUL_F32     v_mfma_f32_32x32x2_f32, exact fp32 products, 157 TFLOP/s peak
 *   STAIR_MATMUL_BF16X3  each fp32 operand split into bf16 hi + lo on the fly, hi*hi + hi*lo + lo*hi on
 *                        v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~4e-6 relative error per dot product,
 *                        3/16 of the fp32 MFMA's matrix-pipe time.
 *   STAIR_MATMUL_BF16    operands rounded to bf16 on the fly, ONE product per pair (the "bf16" of BASELINE.json
 *                        configs[1]): ~2e-3 relative error per dot product, logits move by ~1e-3 -- outside the 1e-4 parity
 *                        budget, so this mode is never the default; it aims at unchanged top-1 answers and
 *                        bench.py reports the measured agreement (99.2 % of 2048 questions after 12 optimizer steps).  GEMMs only; the LSTM recurrences keep the three-product form.
 * Inputs, outputs and accumulation are fp32 in every mode. */
#define STAIR_MATMUL_F32 0
#define STAIR_MATMUL_BF16X3 1
#define STAIR_MATMUL_BF16 2
int stair_set_matmul_mode(int32_t mode);
int stair_get_matmul_mode(void);
/* Per-context options (ABI 5).  stair_set_matmul_mode / stair_set_tile_mlp / stair_set_tile_queue / stair_set_tn_slab_min_rows and the
 * environment set the process-wide DEFAULTS; a context may override each, and its values are in force on the calling thread for the
 * duration of stair_plan_run[_flags] / stair_plan_backward on that context, so several contexts of one process (one per GPU and
 * thread, or two configurations side by side) do not see each other's settings.  value < 0: inherit the process default again.
 * The building-block entry points (stair_gemm_f32, stair_tile_mlp_fwd, ...) take no context and follow the process-wide settings. */
enum stair_option { STAIR_OPT_MATMUL_MODE = 0, STAIR_OPT_TILE_MLP = 1, STAIR_OPT_TILE_QUEUE = 2, STAIR_OPT_VEC_GROUP = 3,
                    STAIR_OPT_TN_SLAB_MIN_ROWS = 4, STAIR_OPT_COUNT = 5 };
int stair_ctx_set_option(stair_ctx *ctx, int32_t option, int32_t value);
int stair_ctx_get_option(const stair_ctx *ctx, int32_t option, int32_t *value);
int stair_set_split_min_rows(int32_t rows); /* GEMMs with fewer rows use the exact kernel (default 1 = none) */

/* ---- building blocks (exported for unit tests and reuse; the plan runner calls the same code) */

/* C[g][r][n] = act( sum_k rs[g][r] * A[g][r][k] * W[n][k] + bias[n] ),  g < groups, r < rows_per_group.
 * Replaces every nn.Linear on the path (modules.py: Filter :347-360, Localize :187-195, ...).
 * Group g of A lives at A + (a_gidx ? a_gidx[g] : g) * a_gstride (same for C and row_scale), which
 * is how ragged program nodes are packed into one launch.  K % 4 == 0, lda/ldw % 4 == 0, pointers
 * 16-byte aligned.  act: 0 none, 1 ReLU, 2 sigmoid. */
typedef struct stair_gemm_args {
    const float *A; int64_t lda; int64_t a_gstride; const int32_t *a_gidx;
    const float *W; int64_t ldw; const float *bias;
    float *C; int64_t ldc; int64_t c_gstride; const int32_t *c_gidx;
    const float *row_scale; int64_t rs_gstride; const int32_t *rs_gidx;
    int32_t groups, rows_per_group, N, K, act;
    int32_t accumulate; /* 0: C = ...; 1: C += ... with fp32 atomics (act must be 0) -- the dX products of the
                           backward pass, where several program nodes may read the same slot */
    float *splitk_ws; int64_t splitk_ws_floats; /* optional scratch: launches of <= 64 output tiles split their K loop over
                           128-wide pieces when ceil(K / 128) * M * N floats are available here (partials reduced in a fixed order:
                           deterministic, and the same for every batch that takes this path); NULL = never split a forward product */
} stair_gemm_args;
int stair_gemm_f32(const stair_gemm_args *args, stair_stream stream);

/* C[n][k] += sum_m A[m][n] * (rs[m] * B[m][k]): the weight gradient dW = dZ^T X of every nn.Linear on the
 * path (autograd of modules.py's Linear layers; train_module.py:408 backward()).  A [M,N] plain rows;
 * B [M,K] gathered in groups of rows_per_group rows like stair_gemm_args.A.  C [N,K] is ACCUMULATED
 * with fp32 atomics (M is split over workgroups), so zero it first.  N, K multiples of 4. */
typedef struct stair_gemm_tn_args {
    const float *A; int64_t lda;
    const float *B; int64_t ldb; int64_t b_gstride; const int32_t *b_gidx;
    const float *row_scale; int64_t rs_gstride; const int32_t *rs_gidx;
    float *C; int64_t ldc;
    int32_t M, rows_per_group, N, K;
    float *colsum, *colsum2; /* optional [N]: += sum_m A[m][n] (the bias gradient db = colsum(dZ) of the same Linear;
                                colsum2 receives the same sums: nn.LSTM's b_ih and b_hh), accumulated with atomics */
    int32_t b_is_bf16; /* 1: B points to bf16 rows (exact values: stored clip features), ldb / b_gstride in ELEMENTS; needs a plain
                          row matrix (rows_per_group = 1, no index, no row scale), K % 8 == 0 and a split matmul mode */
} stair_gemm_tn_args;
int stair_gemm_tn_f32(const stair_gemm_tn_args *args, stair_stream stream);
/* The same contraction as one long reduction with a deterministic result (csrc/gemm_tn_x3tr.hip): M is cut into <= 32 slabs, each
 * slab's partial [N, K] product (and partial column sums) is stored to `scratch`, and a second launch adds the slabs to C / colsum
 * in slab order -- no atomics, bit-identical from run to run.  This is how stair_plan_backward forms the weight gradient of every
 * module-level nn.Linear (one call per WEIGHT over all instances that used it; autograd of /root/reference/video_nmn/modules.py's
 * Linear layers, train_module.py:408).  Split matmul mode only; fp32 operands; N % 256 == 0, K % 128 == 0, ldc == K; M and
 * rows_per_group multiples of 32; colsum2 unused.  scratch: stair_gemm_tn_slabs_scratch(M, N, K) floats. */
/* stair_plan_backward sends a per-bucket weight-gradient product of at least this many rows through the slab kernel (default 4096;
 * the per-weight products and the encoders' always take it).  Tests lower it to exercise that path on small batches. */
int stair_set_tn_slab_min_rows(int32_t rows);
int64_t stair_gemm_tn_slabs_scratch(int64_t M, int64_t N, int64_t K);
int stair_gemm_tn_slabs(const stair_gemm_tn_args *args, float *scratch, int64_t scratch_floats, stair_stream stream);

/* ---- pre-split operands: bf16 planes staged by LDS-DMA (csrc/gemm_planes.hip) -----------------------------------
 * x [n] fp32 -> hi[i] = bf16(x[i]) (round to nearest even), lo[i] = bf16(x[i] - hi[i]); x = hi + lo + O(2^-17 |x|).
 * lo may be NULL (plain rounding to bf16: how a loader produces the bf16 clip-feature format of BASELINE.json
 * configs[1] from the fp32 .npy rows of /root/reference/video_nmn/dataset.py:134-143).  n % 8 == 0, 16-byte aligned. */
int stair_split_planes(const float *x, void *hi, void *lo, int64_t n, stair_stream stream);
/* The same split of a [rows, cols] row-major matrix (an nn.Linear / nn.LSTM weight) written in the TILED layout
 * [cols/32][rows][32]: the 32 k-values a GEMM stage needs from 16 consecutive rows are then 1 KB of contiguous memory,
 * one whole LDS-DMA instruction.  cols % 32 == 0; both planes required. */
int stair_split_planes_tiled(const float *x, void *hi, void *lo, int32_t rows, int32_t cols, stair_stream stream);

/* C[m][n] = act( sum_k A[m][k] * W[n][k] + bias[n] ) with A = A_hi (+ A_lo) and W = W_hi + W_lo given as bf16 planes
 * (row-major, lda / ldw in ELEMENTS, multiples of 8), fp32 accumulation and output.  A_lo == NULL means A is exact in
 * bf16 (stored clip features): two MFMA products per operand pair instead of three.  K % 32 == 0, M, N >= 1.  Same contraction as
 * stair_gemm_f32 for the nn.Linear / nn.LSTM input projections of module_net.py:39-47; used by the plan runner when
 * the clip features arrive in bf16 (STAIR_RUN_VIDEO_BF16). */
typedef struct stair_gemm_planes_args {
    const void *A_hi, *A_lo; int64_t lda;
    const void *W_hi, *W_lo; int64_t ldw;
    const float *bias;
    float *C; int64_t ldc;
    int32_t M, N, K, act;
    int32_t w_tiled; /* 0: W planes row-major [N, ldw]; 1: tiled [K/32][N][32] (stair_split_planes_tiled; ldw ignored);
                        2: W_hi = ONE image in MFMA fragment order, [N/32][K/16][hi, lo][64 lanes][8] (stair_pack_wfrag), W_lo unused
                           but non-NULL: W never enters LDS, each wave loads its own fragments (A_lo NULL, act 0, N % 32, K % 64) */
} stair_gemm_planes_args;
int stair_gemm_planes(const stair_gemm_planes_args *args, stair_stream stream);
/* Measurement aid (ABI 5): the bf16 MFMA rate this device SUSTAINS on random operands -- a full grid (one 512-thread workgroup per CU)
 * of v_mfma_f32_32x32x16_bf16 issued back to back from registers, no memory traffic, `repeats` launches of `iters` x 8 MFMAs per wave
 * after one warm-up launch, timed with events on `stream` (the call waits for them).  Under such a load the chip holds its clock
 * well below 2.4 GHz, so the result lies well under the 2.5 PFLOP/s dense peak: it is the ceiling of the EXECUTED flop rate of any
 * bf16 MFMA kernel on this device (bench.py reports it beside `roofline`, whose `peak` stays the datasheet's). */
int stair_mfma_probe(int32_t iters, int32_t repeats, double *tflops, stair_stream stream);

/* Bidirectional single-layer LSTM over n ragged sequences (nn.LSTM as used at
 * module_net.py:39-47,151-163).  x [rows, I] with sequence s at rows seq_off[s]..seq_off[s+1]-1
 * (seq_off on device, int32 [n+1]; max_len = longest sequence).  w_* are the eight tensors of the
 * reference layer, forward then reverse.  Device scratch: xproj_ws [rows, 8*Hh] floats, bias_ws
 * [8*Hh], whh_pack_ws [8*Hh*Hh] (W_hh re-laid in MFMA fragment order, rebuilt every call).
 * out [rows, 2*Hh]; h_n [n, 2*Hh] = [h_fwd(last) ; h_bwd(first)]. */
typedef struct stair_lstm_args {
    const float *x; int64_t ldx; int32_t rows, n, max_len, I, Hh;
    const int32_t *seq_off;
    const float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];
    float *xproj_ws, *bias_ws, *whh_pack_ws;
    float *out; int64_t ldo; float *h_n;
    float *cbuf; /* NULL for inference.  Training: [rows, 2*Hh] cell states are saved here and xproj_ws is left
                    holding the ACTIVATED gates (i, f, g, o) of every step, both consumed by stair_lstm_bidir_bwd */
    const void *x_bf16; /* optional: the same input rows stored as bf16 [rows, ldx] (the clip-feature format of BASELINE.json
                    configs[1]).  When set, x may be NULL and the input projection runs as a plane GEMM (stair_gemm_planes: A
                    exact in bf16, W_ih split once into hi/lo planes, two MFMA products per pair); needs I % 32 == 0 and a
                    split matmul mode. */
    void *wih_planes_ws; /* scratch for the W_ih planes when x_bf16 (or x_planes_ws) is set: 2 planes x [8*Hh, I] bf16 = 32*Hh*I bytes */
    void *coop_ws; int64_t coop_ws_bytes; /* optional scratch (>= stair_lstm_coop_ws_bytes(n), 256-byte aligned) for the
                    cooperative recurrence (csrc/lstm_coop.hip: Hh = 256, split matmul modes): hidden units split over groups
                    of 4 co-resident workgroups that exchange h every step, W_hh resident in registers.  NULL = the
                    one-workgroup-per-16-sequences kernel.  The launch occupies up to one workgroup on every CU and its
                    workgroups wait for each other: do not run two of them concurrently on different streams. */
    const int32_t *seq_len; /* optional device [n]: PADDED storage -- sequence s occupies rows seq_off[s] .. seq_off[s+1]-1 but
                    only its first seq_len[s] rows are data (clips of different frame counts stored at one stride,
                    /root/reference/video_nmn/dataset.py:137-143); out rows past the length are written as zero.  NULL: every
                    row of the span is data.  The padding rows of x must hold FINITE values (zeros): the weight-gradient
                    products run over all rows with zeroed gate gradients, and 0 * NaN is NaN. */
    uint32_t *status; /* optional device word, caller-owned and STICKY: the cooperative kernels set it to 1 when a hand-off
                    between workgroups timed out (the grid was not co-resident after all -- another queue on the device, a
                    partitioned GPU); the wave that gave up writes NaN from then on.  The library never clears it.  Before
                    a cooperative launch the launcher checks hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs >= grid for
                    the kernel on the current device and otherwise runs the one-workgroup kernel; this word covers what that
                    check cannot see.  stair_plan_run keeps such a word in the plan's workspace (stair_plan_info.status_off);
                    stair_adam_step refuses to update when its `guard` points at a set word. */
    void *x_planes_ws; /* optional scratch (ABI 5), fp32 input rows only: 2 planes x [rows, Ip] bf16 with Ip = I rounded up to a multiple
                    of 32 (4 * rows * Ip bytes, 16-byte aligned).  With it AND wih_planes_ws (then 32 * Hh * Ip bytes) the input
                    projection splits x and W_ih once into zero-padded hi / lo planes and runs as ONE plane GEMM for both
                    directions (three MFMA products per pair; Hh >= 32, split matmul mode; at every row count, so that a row's
                    result does not depend on the batch around it) -- the text encoder's E = 300. */
} stair_lstm_args;
int64_t stair_lstm_coop_ws_bytes(int32_t n);
/* Upper bound on the workgroups a cooperative recurrence may use on the current process's devices (default: every CU;
 * env STAIR_LSTM_COOP_MAX_BLOCKS).  A launch needs 32 workgroups per pair of sequence-tile groups; geometries that do not fit
 * under the cap -- or under hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs -- run on the one-workgroup kernels instead.
 * max_blocks < 0 restores the default.  For a GPU shared with another queue, or tests of the fallback. */
int stair_lstm_coop_limit(int32_t max_blocks);
int stair_lstm_bidir_fwd(const stair_lstm_args *args, stair_stream stream);

/* Backward through time of the same layer (autograd of nn.LSTM in train_module.py:408).  gates = the
 * xproj_ws of a training-mode forward (overwritten in place with the gate pre-activation gradients),
 * cbuf/out from that forward, d_out [rows, ldd] and d_hn [n, 2*Hh] (may be NULL) the incoming gradients.
 * dw_ih/dw_hh/db_ih/db_hh are ACCUMULATED (fp32 atomics).  No input gradient is produced: the inputs
 * of both encoders are data (video features, GloVe vectors).  Scratch: whh_pack_ws [8*Hh*Hh],
 * hprev_ws [rows, 2*Hh]. */
typedef struct stair_lstm_bwd_args {
    const float *x; int64_t ldx; int32_t rows, n, max_len, I, Hh;
    const int32_t *seq_off;
    const float *w_hh[2];
    float *gates; const float *cbuf; const float *out; int64_t ldo;
    const float *d_out; int64_t ldd; const float *d_hn;
    float *whh_pack_ws, *hprev_ws;
    float *dw_ih[2], *dw_hh[2], *db_ih[2], *db_hh[2];
    const void *x_bf16; /* optional, as in stair_lstm_args: dW_ih = dG^T X then reads X as exact bf16 (two products per pair) */
    const int32_t *seq_len; /* optional, as in stair_lstm_args (the gate-gradient rows past a sequence's length are cleared) */
    void *coop_ws; int64_t coop_ws_bytes; /* optional: >= stair_lstm_coop_bwd_ws_bytes(n), 256-byte aligned -> cooperative BPTT (Hh = 256, split mode) */
    uint32_t *status; /* optional sticky timeout word, as in stair_lstm_args */
    float *tn_ws; int64_t tn_ws_floats; /* optional scratch for the slab-reduced weight-gradient products (stair_gemm_tn_slabs): when it is
                                           large enough, dW_hh (and dW_ih on fp32 rows) are formed without atomics; their sums reach the
                                           dw_* / db_* buffers at the END of the stair_plan_backward that passed the scratch, or before
                                           stair_lstm_bidir_bwd returns in a direct call */
} stair_lstm_bwd_args;
int64_t stair_lstm_coop_bwd_ws_bytes(int32_t n);
int stair_lstm_bidir_bwd(const stair_lstm_bwd_args *args, stair_stream stream);

/* ---- fused per-clip tile operators (stair_tile_mlp) ------------------------------------------
 * One workgroup carries one [T, H] tile (the T <= 64 frames of one module instance, H = 512) through up to three Linear
 * layers and the module's tile-local tail; the intermediates stay in LDS (csrc/tile_mlp.hip).  Replaces the GEMM -> HBM ->
 * GEMM -> HBM -> row-kernel sequences of /root/reference/video_nmn/modules.py: Localize :199-217 (2 layers + cosine tail),
 * Filter :363-378 (2 layers + sum over frames), FilterFrame :399-414 (3 layers, sigmoid attention between 2 and 3), HasItem
 * :123-138 (1 layer + row-dot sigmoid), Temporal :310-327 (row-scaled input, 1 layer + LayerNorm).
 * Weights are passed as bf16 hi / lo planes in MFMA fragment order, written by stair_pack_wfrag (2 * N * K * 2 bytes each);
 * products are the split-bf16 three-product form of STAIR_MATMUL_BF16X3 with fp32 accumulation (the only mode supported).
 *   tile of instance i   X + (x_idx ? x_idx[i] : i) * x_gstride, T rows of H floats; rows are multiplied by
 *                        row_scale[(rs_idx ? rs_idx[i] : i) * T + t] when row_scale is set
 *   layer l              act[l] (0 none, 1 ReLU) of tile . W[l]^T + bias[l]; save[l] (optional, [cnt, T, H]) receives that
 *                        activation -- what a backward pass needs, written once
 *   mid_rowdot           (3 layers) after layer 2: a_t = sigmoid(vw . row_t + extra[i] + vb[0]), rs_out[i*T + t] = a_t
 *                        (optional), and layer 3 runs on a_t * row_t
 *   tail                 on the last layer's activation F [T, H]:
 *     STORE           out + o * out_gstride <- F                              (o = out_idx ? out_idx[i] : i)
 *     SUM_ROWS        out[o * out_gstride + n] = sum over t < len[i] (or T) of F[t][n]
 *     COSINE          for pair j in pair_first[i] .. +pair_cnt[i]: att[att_idx[j] * T + t] = (cos(F[t], kb[j]) + 1) * 0.49
 *     ROWDOT_SIGMOID  out[o * out_gstride + t] = sigmoid(vw . F[t] + vb[0] + (extra ? extra[i] : 0))
 *     LAYERNORM       out + o * out_gstride <- LayerNorm_H(F) * gamma + beta, eps = ln_eps, biased variance
 *     ACCUMULATE      out + o * out_gstride += F (atomic)
 * Rows t >= T of a tile do not exist (nothing is read or written there). */
enum stair_tile_tail { STAIR_TILE_NONE = 0, STAIR_TILE_STORE = 1, STAIR_TILE_SUM_ROWS = 2, STAIR_TILE_COSINE = 3,
                       STAIR_TILE_ROWDOT_SIGMOID = 4, STAIR_TILE_LAYERNORM = 5, STAIR_TILE_ACCUMULATE = 6,
                       STAIR_TILE_STORE_ROWS = 7, STAIR_TILE_ROWSCALE_ADJ = 8 };
typedef struct stair_tile_mlp_args {
    const float *X; int64_t x_gstride; const int32_t *x_idx;
    const float *row_scale; const int32_t *rs_idx;
    const void *W[3]; const float *bias[3]; int32_t act[3]; int32_t n_layers;
    float *save[3];
    int32_t mid_rowdot; const float *vw, *vb, *extra; float *rs_out;
    int32_t tail;
    float *out; int64_t out_gstride; const int32_t *out_idx;
    const float *gamma, *beta; float ln_eps;
    const float *kb; const int32_t *pair_first, *pair_cnt, *att_idx; float *att;
    const int32_t *len;
    int32_t cnt, T, H;
    /* backward chains (autograd of the same modules: dX = (dZ2 W2 * relu'(Z1)) W1 on the tile, stair_plan_backward):
     *   act[l] == 3      the layer's result is multiplied by act_scale where act_mask[l][i][t][n] > 0, else 0 (relu' of a saved
     *                    activation, [cnt, T, H]); W[l] then holds the planes of W^T (stair_pack_wfrag(transpose = 1))
     *   in_mask          the input tile is multiplied the same way by relu'(in_mask + (in_mask_idx ? in_mask_idx[i] : i) * in_mask_gstride)
     *                    and in_scale;  x_broadcast: the tile's rows t < len[i] are all the ONE row X + x * x_gstride (x_gstride = H)
     *   save_in          [cnt, T, H]: the transformed input tile (the dZ a weight-gradient product needs)
     *   tail ACCUMULATE  out + o * out_gstride += F with fp32 atomics (several instances may share a gradient tile) */
    const float *act_mask[3]; float act_scale;
    const float *in_mask; int64_t in_mask_gstride; const int32_t *in_mask_idx; float in_scale; int32_t x_broadcast;
    float *save_in;
    /* vector-level modules (modules.py:15-37 Compare / Equals, :59-72 Xor, :102-120 ToAction, :141-159 Exists): the "tile" is 64
     * INSTANCES, one [H] row each, and the first layer runs on a concatenation that is never materialised for the GEMM:
     *   vec_pack         0 off; 1: [a, b]; 2: [|a - b|, a, b]; 3: [a, b, a * b] with a = pk_a + pk_a_idx[i] * H, b = pk_b + pk_b_idx[i] * H
     *                    (i = 0 .. vec_cnt - 1; cnt must be (vec_cnt + 63) / 64).  W[0] holds 2 or 3 consecutive [H, H] plane
     *                    images, one per H-wide column block of the [H, 2H | 3H] weight (stair_pack_wfrag_ld); the layer
     *                    accumulates over them.  X, x_gstride, x_idx, T are ignored.
     *   cat_save         optional [vec_cnt, 2H | 3H]: the concatenated rows (what the weight-gradient product needs)
     *   save[l]          [vec_cnt, H]
     *   tail STORE_ROWS  row i of the last layer -> out + out_row_idx[i] * out_gstride */
    int32_t vec_pack; const float *pk_a, *pk_b; const int32_t *pk_a_idx, *pk_b_idx; int32_t vec_cnt;
    float *cat_save; const int32_t *out_row_idx;
    /* Temporal's backward as ONE chain per tile (autograd of modules.py:310-327, y = LayerNorm(ReLU(Lin(r_t feat_t)))); ABI 5:
     *   ln_bwd != 0      the input tile is the adjoint of LayerNorm (weight gamma, ln_eps) applied to the incoming gradient rows
     *                    X (x_idx, x_gstride) at the saved pre-LayerNorm rows in_mask (in_mask_idx, in_mask_gstride), times
     *                    relu'(in_mask) and in_scale: dZ of the dense layer (kept in save_in for its weight-gradient product);
     *                    dgamma[H] / dbeta[H] receive the LayerNorm parameter gradients (added; the chains of one launch share them).
     *                    One layer: W[0] = the planes of W^T, act 0, no bias.  Comes with
     *   tail ROWSCALE_ADJ  G = the layer's rows = the gradient of the SCALED input: out tile (out_idx) += r_t G_t (fp32 atomics),
     *                    adj_drs[s][t] += G_t . feat_t with feat = adj_feat + (adj_feat_idx ? adj_feat_idx[i] : i) * adj_feat_gstride,
     *                    r = adj_rs + s * T, s = adj_rs_idx ? adj_rs_idx[i] : i */
    int32_t ln_bwd; float *dgamma, *dbeta;
    const float *adj_feat; int64_t adj_feat_gstride; const int32_t *adj_feat_idx;
    const float *adj_rs; const int32_t *adj_rs_idx; float *adj_drs;
    /* acc_exclusive != 0: the caller promises that no two instances of THIS launch add into the same output tile (tails ACCUMULATE,
     * ROWSCALE_ADJ): the tile is read, added to and written back with plain 16-byte accesses instead of float atomics
     * (stair_plan_backward: the deterministic fan-in gives every same-level reader of a slot a target of its own) */
    int32_t acc_exclusive;
    /* relu' of a saved activation as ONE bit per element (4 KB per tile instead of a 128 KB read in the backward chain):
     *   save_bits[l]     [cnt][512] 64-bit words, written by a forward launch: word (i, 64 w + c), byte j, bit e = (layer l's
     *                    activation of tile i, row w + 8 j, column 8 c + e) > 0
     *   act_bits[l]      what a chain uses instead of act_mask[l] (act[l] == 3, act_mask[l] == NULL);  in_bits instead of in_mask
     *                    (both indexed by the instance number i, never through in_mask_idx) */
    unsigned long long *save_bits[3]; const unsigned long long *act_bits[3]; const unsigned long long *in_bits;
    /* nn.Dropout behind layer l's activation (the `D` positions of modules.py), forward launches, ABI 6: drop_site[l] = 1 + the site of
     * stair_dropout_fwd (0: none); element e = (instance * T + row) * H + column of the launch draws the same bit as stair_dropout_fwd on
     * the [cnt, T, H] rows would, so the fused and the launch-per-layer forms of a plan drop the same elements.  One probability and seed
     * per LAUNCH (the first bucket's are taken).  The saved activations / bits are those AFTER the dropout: a backward chain needs only the
     * factor 1 / (1 - p) beside its relu' masks (act_scale, in_scale). */
    uint32_t drop_site[3]; float drop_p; uint64_t drop_seed;
} stair_tile_mlp_args;
int stair_tile_mlp_fwd(const stair_tile_mlp_args *args, stair_stream stream);
/* ---- grouped vector-level products (csrc/vec_group.hip) ------------------------------------------------------------------
 * Every Linear layer that acts on ONE [H] row per instance -- Compare / Equals / Xor / ToAction / Exists (modules.py:15-37, 59-72,
 * 102-120, 141-159), Filter's dense layer (:376-378), Localize's keyword projection (:199-203), the decoder (module_net.py:49-53)
 * -- as a list of problems carried by ONE launch: work items are (problem, 32-row tile, 32-column block), the reduction dimension
 * is never split across workgroups (no scratch, no reduction launch, deterministic, a row's result is independent of the other rows).
 * H = 512 (the reference's hidden size), split matmul mode.
 *   kind FWD   out[io[i]] (+)= act(in(i) W^T + bias), W [N, nseg * 512 (ldw)] row-major fp32
 *   kind ADJ   the backward of such a layer THROUGH its concatenation: in(i) (a gradient row, optionally masked) times
 *              W = the transposed weight image [nseg * 512, 512 (ldw)] gives the nseg H-wide blocks of d(concatenation), and the
 *              epilogue adds the adjoint of the concatenation into the operands' gradient rows ga[fia[i]], gb[fib[i]]
 *              (N = nseg * 512; adj = the forward layer's input form; fa / fb = its operand rows, read by EXISTS and XOR)
 *   input form in(i), from the operand rows a_i = a + (ia ? ia[i] : i) * lda, b_i likewise:
 *     IN_A [a_i] (kred <= 512 columns);  IN_CAT2 [a_i, b_i];  IN_XOR [|a_i - b_i|, a_i, b_i];  IN_EXISTS [a_i, b_i, a_i * b_i];
 *     IN_MASK [a_i * in_scale where b_i > 0, else 0] (relu' of a saved output applied to an incoming gradient)
 *   in_save    optional [rows, ld_save]: the formed input rows (the operand a weight-gradient product needs)
 *   act        0 none, 1 ReLU, 2: times escale where emask[i * ldm + n] > 0, else 0;  accumulate != 0: float atomics instead of stores */
enum { STAIR_VEC_FWD = 0, STAIR_VEC_ADJ = 1 };
enum { STAIR_VEC_IN_A = 0, STAIR_VEC_IN_CAT2 = 1, STAIR_VEC_IN_XOR = 2, STAIR_VEC_IN_EXISTS = 3, STAIR_VEC_IN_MASK = 4 };
typedef struct stair_vec_problem {
    int32_t kind, rows;
    const float *a, *b; const int32_t *ia, *ib; int64_t lda, ldb;
    int32_t pack; float in_scale; int32_t kred;
    const float *W; int64_t ldw; const float *bias; int32_t N, act;
    const void *wplanes; /* optional: the weight as bf16 hi / lo fragment-order planes (stair_pack_wfrag), one [512 x 512] image per
                            (block b of 512 output columns, input segment s) at wplanes + (b * nseg + s) * 512 * 512 * 4 bytes;
                            NULL: the fp32 rows of W are read and split (plain and two-segment forward problems only;
                            needed when N % 32 or kred != 512) */
    const float *emask; int64_t ldm; float escale;
    float *out; const int32_t *io; int64_t ldo; int32_t accumulate;
    float *in_save; int64_t ld_save;
    int32_t adj; const float *fa, *fb; const int32_t *fia, *fib; int64_t ldfa, ldfb; float *ga, *gb;
    const int32_t *gia, *gib; /* optional: the gradient rows ga + gia[i] * ldfa, gb + gib[i] * ldfb when they are not the rows fia / fib
                                 (a plan sends all but one same-level reader of a shared operand to private staging rows, so that every
                                 address receives at most one atomic add per launch: stair_plan_backward's deterministic fan-in) */
    uint32_t drop_site; float drop_p; uint64_t drop_seed; /* ABI 6, forward problems with act 1: nn.Dropout behind the ReLU -- drop_site = 1 + the
                                 site of stair_dropout_fwd (0: none); element e = row * N + column draws the bits stair_dropout_fwd draws
                                 for the [rows, N] matrix */
} stair_vec_problem;
int stair_vec_group(const stair_vec_problem *problems, int32_t count, stair_stream stream);

/* stair_plan_run uses the fused operators where they apply (hidden_size 512, T <= 64, split matmul mode; with dropout too since ABI 6);
 * on = 0 keeps the GEMM / row-kernel sequences everywhere, on < 0 restores the default (env STAIR_TILE_MLP, default on). */
int stair_set_tile_mlp(int32_t on);
/* Tiles of a fused launch are dealt out through a self-resetting atomic work queue in the plan workspace (on != 0, default; two
 * words that every launch leaves at zero, so eager runs and hipGraph replays need no reset between launches) or by a static round
 * robin (on = 0); on < 0 restores the default (env STAIR_TILE_QUEUE).  Results are bit-identical either way. */
int stair_set_tile_queue(int32_t on);
/* W [N, K] fp32 row-major -> planes: [N/32][K/16][hi, lo][64 lanes][8 bf16] (2 * N * K * 2 bytes, 16-byte aligned);
 * N % 32 == 0, K % 16 == 0.  transpose != 0: W is stored [K, N] and the planes are those of W^T (backward chains). */
int stair_pack_wfrag(const float *W, void *planes, int32_t N, int32_t K, int32_t transpose, stair_stream stream);
/* The same for a [N, K] column block of a wider row-major matrix (row stride ld floats): the H-wide blocks of a vector-level
 * module's [H, 2H | 3H] first-layer weight. */
int stair_pack_wfrag_ld(const float *W, int64_t ld, void *planes, int32_t N, int32_t K, stair_stream stream);
/* Measurement aid: while on, every tile-operator launch is bracketed by HIP events on its stream; stair_tile_timing_read waits for
 * them and returns the summed device time (ms) and the number of launches since stair_tile_timing(1) (bench.py's
 * roofline_tile_operator: the kernel's live time without a profiler).  Not thread-safe; off by default. */
int stair_tile_timing(int32_t on);
int stair_tile_timing_read(double *ms, int32_t *launches);

/* att[p][t] = (cos(F[f_idx[p]][t][:], Kmat[k_idx[p]][:]) + 1) * 0.49 -- nn.CosineSimilarity(dim=-1,
 * eps=1e-8) of LocalizeModule / ExistsFrameModule (modules.py:162-217) without materialising the
 * [K,T,H] expands.  F tile p at F + f_idx[p]*f_gstride, [T,H]; output row att + out_idx[p]*T. */
int stair_cosine_attn_fwd(const float *F, int64_t f_gstride, const int32_t *f_idx, const float *Kmat,
                          const int32_t *k_idx, float *att, const int32_t *out_idx, int32_t npairs,
                          int32_t T, int32_t H, stair_stream stream);

/* r = relate_mode(mean_k att[att_idx[i] .. +K[i]-1]) for TemporalModule (modules.py:255-277,317-323).
 * mode 0 while (identity), 1 before, 2 after, 3 between.  conv != 0: three Conv1d(1,1,k,'same')
 * (k, k, 2k+1) with ReLU, ReLU, Sigmoid; else three Linear(T,T).  w[6] = weight/bias of layers 0,2,4. */
int stair_temporal_relate_fwd(const float *att, const int32_t *att_idx, const int32_t *att_k,
                              float *out, const int32_t *out_idx, int32_t n, int32_t T, int32_t mode,
                              int32_t conv, int32_t ksize, const float *const w[6], stair_stream stream);

/* Adjoint of stair_cosine_attn_fwd (autograd of nn.CosineSimilarity in modules.py:170-177,203-216): given d_att rows
 * (gradient w.r.t. the (cos+1)*0.49 outputs, row out_idx[p]) ADD the gradients into dF (tile f_idx[p], same layout as F)
 * and dK (row k_idx[p]); both must be initialised by the caller (several pairs may share a tile or a keyword row:
 * fp32 atomics). */
int stair_cosine_attn_bwd(const float *F, int64_t f_gstride, const int32_t *f_idx, const float *Kmat, const int32_t *k_idx,
                          const float *d_att, const int32_t *out_idx, float *dF, float *dK, int32_t npairs, int32_t T,
                          int32_t H, stair_stream stream);
/* Adjoint of stair_temporal_relate_fwd (TemporalModule's relate nets, modules.py:255-277,317-323): d_out row out_idx[i]
 * is the gradient w.r.t. the relate output of instance i; the gradient w.r.t. each of its att_k[i] attention rows is ADDED
 * into d_att (rows att_idx[i] .. +att_k[i]-1, the mean over K splits evenly), the gradients of the three layers' weights
 * and biases are ADDED into dw[6] (same order as w[6]); mode 0 (`while`) has no parameters. */
int stair_temporal_relate_bwd(const float *att, const int32_t *att_idx, const int32_t *att_k, const float *d_out,
                              const int32_t *out_idx, float *d_att, int32_t n, int32_t T, int32_t mode, int32_t conv,
                              int32_t ksize, const float *const w[6], float *const dw[6], stair_stream stream);

/* out[i] = x[i] / max(||x[i]||_2, 1e-12): L2Normalize, the contrastive / pretrain head of Filter,
 * Superlative and ToAction (module_net.py:21, 211-216).  x, out [n,H]. */
int stair_l2normalize_fwd(const float *x, float *out, int32_t n, int32_t H, stair_stream stream);

/* ---- data-parallel exchange: the one collective of a training step ---------------------
 * RCCL, resolved at run time (the copy torch already loaded is reused).  One communicator per process / GPU.
 * stair_comm_unique_id: rank 0 fills 128 bytes (ncclUniqueId) and hands them to the other ranks by any side channel
 * (Python: torch.distributed.broadcast_object_list over gloo, a file, ...); every rank then calls stair_comm_create.
 * stair_allreduce_grads: in-place SUM over ranks of the flat fp32 bucket [gradients | touched mask as floats]
 * (stair_amd/train.py lays it out) on `stream`; replaces the accumulation of /root/reference/train_module.py:386-412
 * across a sharded window.  Asynchronous like every other entry point. */
typedef struct stair_comm stair_comm;
int stair_comm_unique_id(void *id128);
int stair_comm_create(const void *id128, int32_t rank, int32_t world, stair_comm **out);
void stair_comm_destroy(stair_comm *comm);
/* rank and size as the RCCL communicator itself reports them (ncclCommUserRank / ncclCommCount); either may be NULL */
int stair_comm_info(const stair_comm *comm, int32_t *rank, int32_t *nranks);
int stair_allreduce_grads(stair_comm *comm, float *bucket, int64_t n, stair_stream stream);

/* ---- program plans: the batched stack interpreter (module_net.py:94-138) -------------------- */

/* Rank C candidate representations by cosine similarity to each query row and keep the k best
 * (/root/reference/evaluate.py:95-98: nn.CosineSimilarity()(result, filter_ans_reps), argsort descending, [:10]).
 * queries: row q_idx[i] (or i when q_idx is NULL) of a matrix with leading dimension ldq, H floats each; keys [C,H]
 * contiguous; key_invnorm_ws [C] scratch.  out_idx/out_sim [n,k], best first; equal similarities keep the lower index
 * (torch's argsort leaves that order unspecified).  C <= 1024, 1 <= k <= C. */
int stair_cosine_topk(const float *queries, int64_t ldq, const int32_t *q_idx, const float *keys,
                      float *key_invnorm_ws, int32_t n, int32_t C, int32_t H, int32_t k, int32_t *out_idx,
                      float *out_sim, stair_stream stream);

/* Compile n questions' prefix programs into a launch plan: nodes are levelled as
 * utils/program_parser.py:307-321 (stat_module_levels) and all nodes of one (level, module,
 * keyword-variant) go into one packed launch.  Host arrays:
 *   prog_off [n+1]      token range of question q in `tokens`
 *   tokens   [ntok]     enum stair_token codes
 *   span_lo/hi [ntok]   question-token span of token i (prog_str_to_question_tokens), used for STAIR_TOK_SPAN
 *   q_off    [n+1]      row range of question q in the packed question embedding matrix
 *   T                   frames per video in this batch (<= max_video_length; == for Linear Temporal)
 * Errors (non-zero) mirror the reference's failures: invalid program (assert len(stack)==1,
 * module_net.py:135), operand of the wrong kind, span outside the question. */
#define STAIR_PLAN_TRAIN 1 /* flags: keep every intermediate and lay out gradient arenas for stair_plan_backward */
#define STAIR_PLAN_NO_CSE 2 /* flags: compute every node of every question, as module_net.py:100-106 does.  By default a node whose
                               operands are the encoded clip, keyword strings, identical question spans or other such nodes is
                               computed ONCE per batch and every other occurrence -- in another question about the same clip, or
                               again in the same program -- aliases its slot (same value; gradients of all users add up in it).
                               Filter's tensor keyword does not enter the comparison: its attention is identically 1
                               (modules.py:354,373).  Env STAIR_PLAN_CSE=0 has the same effect. */
#define STAIR_PLAN_EXT_PROJECTION 4 /* flags (ABI 6): the regions of the encoders' input projections (x W_ih^T + biases of every clip frame and
                               token row: the gates of a training plan; bias sums; W_ih / token-row planes) are NOT laid out in the plan's
                               workspace but live in a caller-owned buffer of stair_projection_floats floats, handed over with
                               stair_plan_set_projection before the first pass.  Their layout depends on the batch SHAPE alone, so
                               stair_encoders_project can fill the buffer before the plan exists (see there). */
int stair_plan_build(stair_ctx *ctx, int32_t n, const int32_t *prog_off, const int32_t *tokens,
                     const int32_t *span_lo, const int32_t *span_hi, const int32_t *q_off, int32_t T,
                     int32_t flags, stair_plan **out);
/* Same, for batches in which several questions ask about the same video (AGQA averages tens of questions per
 * video, and module_net.py:74 re-encodes the video for each of them): `video` then holds n_videos distinct clips
 * [n_videos, T, V] and video_of_question[q] names the clip of question q.  The video bi-LSTM (the largest single
 * cost of the path) runs once per clip and every question's `video` token aliases the shared encoded map, so the
 * results are those of stair_plan_build on the expanded batch.  video_of_question == NULL requires n_videos == n
 * (identity).  Works for STAIR_PLAN_TRAIN too: gradients of all consumers accumulate into the clip's map. */
int stair_plan_build_shared(stair_ctx *ctx, int32_t n, const int32_t *prog_off, const int32_t *tokens,
                            const int32_t *span_lo, const int32_t *span_hi, const int32_t *q_off,
                            int32_t n_videos, const int32_t *video_of_question, int32_t T,
                            int32_t flags, stair_plan **out);
/* Same, for batches whose clips differ in FRAME COUNT (real I3D .npy clips keep their own length and are only truncated
 * above max_video_length, dataset.py:137-143): `video` is [n_videos, T, V] with clip v's video_len[v] <= T frames first and
 * padding behind; every [T,H] map and [T] attention row keeps the stride T.  Each question is computed as the reference
 * computes a clip of ITS length: the LSTM runs video_len steps, Filter sums, Relate's softmax, Superlative's pooling (and
 * its action count when the actions are a map), the Conv1d Temporal nets ('same' zero padding at the clip's own end) and
 * the attention criterion see exactly video_len frames; padded frames of map outputs hold finite don't-care values and
 * padded attention frames of Temporal / Relate outputs are zero.  Needs the Conv1d Temporal configuration
 * (max_video_length > 32).  video_len == NULL: every clip has T frames (== stair_plan_build_shared). */
int stair_plan_build_ragged(stair_ctx *ctx, int32_t n, const int32_t *prog_off, const int32_t *tokens,
                            const int32_t *span_lo, const int32_t *span_hi, const int32_t *q_off,
                            int32_t n_videos, const int32_t *video_of_question, const int32_t *video_len, int32_t T,
                            int32_t flags, stair_plan **out);
void stair_plan_destroy(stair_plan *plan);

/* Workspace the caller must provide to stair_plan_run (bytes, device) and its arena layout
 * (offsets in floats from the workspace base) so a host can read any intermediate result. */
typedef struct stair_plan_info {
    int64_t workspace_bytes;
    int64_t vec_off, map_off, att_off, tok_off, qfeat_off, logits_off; /* float offsets */
    int64_t gvec_off, gmap_off, gatt_off; /* gradient arenas of a STAIR_PLAN_TRAIN plan (same slot numbering), else -1 */
    int64_t status_off; /* float offset of the plan's status word (one uint32): cleared by stair_plan_run, set by a cooperative
                           recurrence whose hand-off timed out (forward or backward); see stair_plan_status */
    int32_t n_vec, n_map, n_att, n_tok_rows;
    int32_t n_nodes, n_launches, n_levels, n_questions, T;
    int32_t n_aliased; /* program nodes that alias another node's value (common subexpressions, see STAIR_PLAN_NO_CSE) */
    int32_t n_vec_stage, n_map_stage, n_att_stage; /* STAIR_PLAN_TRAIN: staging rows / tiles / rows behind the gradient arenas.  A value that
                           several nodes of ONE level read gets one staging slot per extra reader; stair_plan_backward adds them to the
                           value's gradient slot in a fixed order before it walks the level that produced the value, so that no address
                           receives two atomic adds from one launch (run-to-run reproducible sums) */
} stair_plan_info;
int stair_plan_get_info(const stair_plan *plan, stair_plan_info *info);

/* Reads the plan's status word back (copy on `stream` + stream synchronisation).  Returns 0 when the passes run on this
 * workspace since the last stair_plan_run completed normally; non-zero (message via stair_last_error) when a cooperative
 * recurrence timed out -- the logits / gradients of that run contain NaN and must be discarded.  A training loop that does
 * not want the synchronisation passes the word's device address (workspace + status_off floats) to stair_adam_step as
 * `guard` and checks later. */
int stair_plan_status(const stair_plan *plan, const void *workspace, stair_stream stream);

/* Diagnostics (tools/queue_probe.py; DESIGN.md section 2, "the replay abort"): enqueues on `stream` -- which may be capturing -- the
 * reset and `launches` work-queue kernels laid out as stair_plan_run lays out its fused tile launches, with a kernel that RECORDS the
 * tickets it draws instead of using them (no tile work, safe whatever the words hold).  words: >= 64 uint32, queue words from
 * word 16 on; seen: [launches][grid][2] uint32 = (first ticket, tiles taken) per workgroup; reset_mode 0: hipMemsetAsync of the
 * 256 bytes (round 3), 1: a zeroing kernel (now), 2: none; per_launch_heads != 0: launch l draws from words 16 + 2 l (round 3),
 * else all launches share words 16, 17; self_reset != 0: the last workgroup to leave zeroes the pair (now). */
int stair_debug_memset(void *ptr, int64_t bytes, int32_t mode, stair_stream stream); /* mode 0: hipMemsetAsync(ptr, 0, bytes); 1: the library's zero-fill kernel */
int stair_debug_queue_probe(void *words, uint32_t *seen, int32_t launches, int32_t grid, int32_t total, int32_t reset_mode,
                            int32_t per_launch_heads, int32_t self_reset, stair_stream stream);

/* kind/slot/aux of program token `tok` (global index into `tokens`); level as stat_module_levels;
 * rel_slot = att-arena row of Temporal's related_attn (-1 otherwise). */
int stair_plan_node(const stair_plan *plan, int32_t tok, int32_t *kind, int32_t *slot, int32_t *aux,
                    int32_t *level, int32_t *rel_slot);

/* Introspection for tests: where a STAIR_PLAN_TRAIN plan keeps the activations its backward pass reads (the ReLU masks of
 * the pass).  which = 0 / 1: first / second saved activation of the node's tile MLP ([T, H] floats: Filter, FilterFrame,
 * Localize, Superlative; first only: HasItem, Temporal) or the hidden row [H] of Exists / ToAction (which = 0);
 * tok = -1 - q: the decoder's hidden row [2H] of question q.  *offset = float offset into the workspace, -1 = none. */
int stair_plan_saved_offset(const stair_plan *plan, int32_t tok, int32_t which, int64_t *offset);

/* The same for every token at once: host arrays of stair_plan_info.n_nodes int32 (any may be NULL).  What a loss driver
 * needs to address the supervised nodes of a whole batch without one call per node. */
int stair_plan_nodes(const stair_plan *plan, int32_t *kind, int32_t *slot, int32_t *aux, int32_t *level,
                     int32_t *rel_slot, int32_t count);

/* Run the whole path for the batch: encode_video, encode_question, every program level, decoder,
 * argmax.  video [n,T,V], question [q_off[n],E] device fp32; logits [n,A]; argmax [n] int32
 * (either may be NULL).  Everything is enqueued on `stream`. */
int stair_plan_run(stair_ctx *ctx, stair_plan *plan, const float *video, const float *question,
                   void *workspace, int64_t workspace_bytes, float *logits, int32_t *argmax,
                   stair_stream stream);

/* Training-mode dropout (nn.Dropout(config['dropout']) at the `D` positions of video_nmn/modules.py, active in the
 * reference whenever model.train() is: after the ReLU of every hidden Linear of Filter / FilterFrame / Localize /
 * Exists / HasItem / ToAction / the decoder, after the dense layers of FilterFrame and Temporal, after HasItem's
 * sigmoid).  Call on a STAIR_PLAN_TRAIN plan before stair_plan_run; p = 0 switches it off again.  The masks are a
 * counter-based hash of (seed, position, element) -- torch's Philox stream cannot be reproduced -- so a step is
 * replayable and stair_plan_backward needs no stored masks.  Inference plans never drop.  A plan with
 * shared subexpressions (n_aliased > 0) is refused: the reference draws an independent mask per question and occurrence
 * (module_net.py:100-106 under model.train()), so build dropout plans with STAIR_PLAN_NO_CSE. */
int stair_plan_set_dropout(stair_plan *plan, float p, uint64_t seed);
/* Data-parallel training: `event` (a hipEvent_t the caller owns; NULL switches it off) is recorded on stair_plan_backward's stream at
 * the point where every gradient EXCEPT the two encoders' is final -- decoder, all program levels and their weight-gradient products
 * -- i.e. before BPTT and the encoders' dW_ih / dW_hh.  A trainer lets a side stream wait for it and reduces that part of its
 * gradient bucket beside the rest of the pass (stair_amd/train.py; the reference is single-process, train_module.py:408). */
int stair_plan_set_backward_event(stair_plan *plan, void *event);
/* The mask generator on its own (building block / test hook): in place on `groups` rows of `rowlen` floats, row g at
 * x + (gidx ? gidx[g] : g) * gstride; element e of the launch is kept iff its 16 bits of hash64(seed, site, e / 4) (bits 16 (e % 4) .. +15) are >= p * 2^16 and then
 * divided by (1 - p). */
int stair_dropout_fwd(float *x, int64_t gstride, const int32_t *gidx, int32_t groups, int64_t rowlen, float p,
                      uint64_t seed, uint32_t site, stair_stream stream);

/* The same pass split for stream capture (BASELINE.json configs[3], "hipGraph-captured module chains"):
 * stair_plan_upload copies the plan's index image (slot columns, sequence offsets) into the workspace ONCE, outside
 * the capture; stair_plan_run_flags(..., STAIR_RUN_INDEX_RESIDENT, ...) then enqueues kernels only -- no host-to-device
 * copy, no allocation, no synchronisation -- so it can be recorded between hipStreamBeginCapture / EndCapture (or
 * torch.cuda.graph) and replayed while video / question / workspace / logits keep their addresses.  The caller must not
 * let anything else write the workspace's index region in between (give a captured plan its own workspace), and must
 * have run the plan once un-captured first: kernels with more than 64 KB of LDS set their function attribute on first
 * use, which is not a stream operation. */
#define STAIR_RUN_INDEX_RESIDENT 1
#define STAIR_RUN_VIDEO_BF16 2 /* `video` points to bf16 [n_videos, T, V] (V % 32 == 0): the stored clip-feature format of
                                  BASELINE.json configs[1]; results equal the fp32 path fed the same (rounded) values up to
                                  the split-product error.  Also a flag of stair_plan_backward. */
#define STAIR_RUN_PROJECTED 4 /* (ABI 6) a STAIR_PLAN_EXT_PROJECTION plan whose buffer already holds the projections of THESE inputs
                                 (stair_encoders_project on the same stream, same weights): the pass starts at the recurrences.  Without the
                                 flag the pass computes the projections into the buffer itself (a captured plan's replay does). */
int stair_plan_upload(stair_plan *plan, void *workspace, int64_t workspace_bytes, stair_stream stream);
/* The encoders' input projections ahead of the plan.  module_net.py:74-75 encodes the clip and the question before the interpreter
 * looks at the program; here the encoders' FIRST half (the projection GEMMs: 0.60 of the 0.86 GFLOP of a question, ~2.2 ms of device
 * time at 2048 questions) depends on nothing but the batch's inputs, while packing the programs and building the plan is 3-4 ms of
 * host work: a caller that enqueues the projections first lets the device work through them meanwhile (a loop that reads the loss
 * after every step otherwise idles the device for that long at every step entry).
 *   stair_projection_floats   size of the buffer for n_videos clips of T frames and question_rows token rows (floats; -1 on bad input)
 *   stair_encoders_project    enqueue both projections into buf (256-byte aligned, device); video is fp32 [n_videos, T, V] or, with
 *                             video_is_bf16, bf16 (V % 32 == 0); reads the bound weights (stair_ctx_set_weight) at execution time
 *   stair_plan_set_projection attach the buffer to a plan built with STAIR_PLAN_EXT_PROJECTION for the same batch shape; it must stay
 *                             alive and unmodified until the plan's last pass has executed (stair_plan_backward reads the gates there)
 * then stair_plan_run_flags(..., STAIR_RUN_PROJECTED | ...).  Results are bit-identical to the plan computing the projections itself. */
int64_t stair_projection_floats(const stair_ctx *ctx, int32_t n_videos, int32_t T, int64_t question_rows);
int stair_encoders_project(stair_ctx *ctx, const void *video, int32_t video_is_bf16, int32_t n_videos, int32_t T,
                           const float *question, int64_t question_rows, float *buf, int64_t buf_floats, stair_stream stream);
int stair_plan_set_projection(const stair_ctx *ctx, stair_plan *plan, float *buf, int64_t buf_floats);
int stair_plan_run_flags(stair_ctx *ctx, stair_plan *plan, const float *video, const float *question,
                         void *workspace, int64_t workspace_bytes, float *logits, int32_t *argmax, int32_t flags,
                         stair_stream stream);

/* Reverse pass of a STAIR_PLAN_TRAIN plan after stair_plan_run on the same workspace: decoder cross
 * entropy against answers[n] (train_module.py:193-194,376-380), d(loss_scale * sum_i CE_i) propagated
 * through decoder, every program level in reverse, and both encoders (BPTT); parameter gradients are
 * accumulated into the buffers given to stair_ctx_set_grad.  loss_out [n] (device, may be NULL)
 * receives the unscaled per-question CE.  answers[i] < 0 leaves question i without a decoder loss (its CE is
 * reported as 0): the train_decoder_after_iters gate of train_module.py:376.  What torch autograd does for :408. */
int stair_plan_backward(stair_ctx *ctx, stair_plan *plan, const float *video, const float *question,
                        void *workspace, int64_t workspace_bytes, const int32_t *answers, float loss_scale,
                        float *loss_out, int32_t flags, stair_stream stream);
#define STAIR_BWD_KEEP_ARENAS 1 /* flags: the gradient arenas were cleared by stair_plan_zero_grads and already hold
                                   the gradients injected by the stair_loss_* functions below.  STAIR_RUN_VIDEO_BF16 (2) must be
                                   repeated here when the forward pass ran on bf16 clip features. */
int stair_plan_zero_grads(stair_plan *plan, void *workspace, stair_stream stream);

/* ---- per-module intermediate-supervision losses (train_module.py:33-194, CriterionByModule) -------------------
 * Each call evaluates `n` loss items, writes the unscaled loss of item i to loss[i] and ADDS scale * dloss/dresult
 * into the gradient arena at the result's slot; head parameters' gradients are added to dW/db.  All arrays device.
 * The gradient pointers (d_att / d_vec / dW / db) may be NULL: the call then only evaluates the losses, which is how the
 * validation loop (train_module.py:219-270) scores an inference plan. */

/* Reproducible criteria (ABI 5; the supervised step, BASELINE configs[4], bit-identical from run to run like the decoder-only step):
 *   stair_loss_groups        the NEXT stair_loss_* call on this thread takes its n items in groups: items order[grp_off[g]] ..
 *                            order[grp_off[g+1] - 1] (indices into that call's item arrays) add into the same gradient slot -- the
 *                            caller has sorted them (supervised nodes that alias one slot) -- and ONE workgroup evaluates them
 *                            one after the other, adding with plain read - add - write instead of float atomics.  Device arrays
 *                            order [n], grp_off [n_groups + 1]; consumed by that call; n_groups = 0 cancels.
 *   stair_grad_shadows_begin opens the fixed-point accumulation scope of stair_plan_backward EARLY on this thread: kernels launched
 *                            from here on that add into a bound gradient tensor of `ctx` from several workgroups (the head weights
 *                            of stair_loss_head / stair_loss_filterframe) add into its 64-bit fixed-point shadow, where the order
 *                            of the adds cannot matter; the next stair_plan_backward(ctx, ...) on this thread adds the shadows to
 *                            the gradients and closes the scope.  A no-op under STAIR_DETERMINISTIC=0. */
int stair_loss_groups(const int32_t *order, const int32_t *grp_off, int32_t n_groups);
int stair_grad_shadows_begin(stair_ctx *ctx, stair_stream stream);

/* attention_score_criterion (:83-90) on att rows slot[i] .. slot[i]+K[i]-1 against the soft interval masks of
 * span_to_attention (:67-81); intervals[2*(iv_off[i]+r)] = (start, end) of row r in frames (double, like the
 * reference).  Localize (:173-182, K rows, mean over K*T), Temporal / ExistsFrame (:157-164, :184-191, K = 1). */
int stair_loss_attention(const float *att, float *d_att, const int32_t *slot, const int32_t *K,
                         const int32_t *iv_off, const double *intervals, int32_t n, int32_t T, float scale,
                         float *loss, stair_stream stream);
/* The same with per-item clip lengths (plans built by stair_plan_build_ragged): len[i] frames of the rows of item i count,
 * the interval masks are drawn for len[i] frames.  len == NULL: T. */
int stair_loss_attention_len(const float *att, float *d_att, const int32_t *slot, const int32_t *K,
                             const int32_t *iv_off, const double *intervals, const int32_t *len, int32_t n, int32_t T,
                             float scale, float *loss, stair_stream stream);
/* Linear pretrain head W [nout,H] on vec[slot[i]] + loss: nout = 2 CrossEntropy vs bool label (Exists, Xor :92-99),
 * nout = 1 squared error vs 0/1 (Equals :101-107). */
int stair_loss_head(int32_t nout, const float *vec, float *d_vec, const int32_t *slot, const int32_t *label,
                    const float *W, const float *b, float *dW, float *db, int32_t n, int32_t H, float scale,
                    float *loss, stair_stream stream);
/* Contrastive CE of Filter / ToAction / Superlative (:113-125) with the per-window class pooling of :388-406:
 * pred = L2Normalize(vec[slot[i]]); logits over class reps G[win_start[i] .. +win_cnt[i]) ; positive row pos[i]
 * (absolute row of G).  max_classes = max(win_cnt). */
int stair_loss_contrastive(const float *vec, float *d_vec, const int32_t *slot, const int32_t *pos,
                           const int32_t *win_start, const int32_t *win_cnt, const float *G, int32_t n, int32_t H,
                           int32_t max_classes, float scale, float *loss, stair_stream stream);
/* The same criterion against a TABLE of all class representations reps [n_cls, H]: item i's pool is the set of classes c with
 * presence[win_row[i]][c] > 0 (presence [n_windows, n_cls] floats) and its positive is class pos_class[i].  Under data
 * parallelism each rank marks the classes of its own questions and the table is summed over ranks ON THE DEVICE (one small
 * all-reduce on the stream) -- the pools of train_module.py:388-406 without any rank learning the others' class lists on the
 * host.  The value equals stair_loss_contrastive on the pooled rows up to the order of the softmax sum. */
int stair_loss_contrastive_table(const float *vec, float *d_vec, const int32_t *slot, const int32_t *pos_class,
                                 const int32_t *win_row, const float *presence, const float *reps, int32_t n,
                                 int32_t n_cls, int32_t H, float scale, float *loss, stair_stream stream);
/* Decoder cross entropy without gradients -- the validation loop's loss (train_module.py:193-194, 246-248):
 * loss[i] = logsumexp(logits[i]) - logits[i][answers[i]]; answers[i] < 0: 0, answers[i] >= A: NaN. */
int stair_loss_decoder_ce(const float *logits, const int32_t *answers, float *loss, int32_t n, int32_t A, stair_stream stream);
/* 'cont-valid' score of Filter / ToAction / Superlative in validation (train_module.py:127-132): cosine between
 * vec[slot[i]] and the mean of reps[seg_off[i] .. seg_off[i+1]) (the question's own gold class representations, [rows,H]);
 * 0 for an empty list. */
int stair_score_cosine_to_mean(const float *vec, const int32_t *slot, const float *reps, const int32_t *seg_off, float *out,
                               int32_t n, int32_t H, stair_stream stream);

/* FilterFrame (:141-155): pretrain head W [O,H], b [O] on the T frames of map tile slot[i] (rows (slot*T + t) of the map
 * arena), softmax over the O object classes, BCELoss against gold [n,T,O] = the row-normalised interval masks the host
 * builds from the gold {entity: (start, end)} dict (0 rows where no entity is present), mean over T*O.  Gradients go
 * to d_map (the map gradient arena), dW, db.  The reference leaves this loss out of training by default (args.py:62)
 * but scores it in validation. */
int stair_loss_filterframe(const float *map, float *d_map, const int32_t *slot, const float *gold, const float *W,
                           const float *b, float *dW, float *db, int32_t n, int32_t T, int32_t H, int32_t O,
                           float scale, float *loss, stair_stream stream);
/* The same for clips of different lengths in one batch: len[i] = frames of item i's clip (the tile keeps stride T; gold
 * [n][T][O] is built for len[i] frames, rows past it are ignored); the mean runs over len[i] * O. */
int stair_loss_filterframe_len(const float *map, float *d_map, const int32_t *slot, const float *gold, const float *W,
                               const float *b, float *dW, float *db, const int32_t *len, int32_t n, int32_t T, int32_t H,
                               int32_t O, float scale, float *loss, stair_stream stream);

/* Test hook: every region of the workspace layout as (name, begin, end) float offsets; returns the region count.
 * Regions must be pairwise disjoint (tests/test_abi.py checks it for inference and training plans). */
int stair_plan_regions(stair_plan *plan, const stair_ctx *ctx, const char **names, int64_t *beg, int64_t *end, int32_t cap);

/* touched[id] = 1 iff weight `id` receives a gradient from this plan (host array, count = stair_weight_count).
 * Parameters of modules that no program used keep grad == None in the reference and are skipped by Adam. */
int stair_plan_touched(const stair_ctx *ctx, const stair_plan *plan, int32_t *touched, int32_t count);

/* torch.optim.Adam (train_module.py:326) over a flat fp32 parameter buffer of n floats.  Parameter tensors
 * ("segments") start on multiples of 256 floats; seg_of_block[b] = segment of elements 256b..256b+255;
 * segments with touched[seg] == 0 are skipped (no moment decay, no step), step_of_seg[seg] is the 1-based
 * Adam step of the segment (already incremented by the caller).  All arrays on the device.
 * guard (optional device word): when *guard != 0 at execution time the kernel changes NOTHING -- parameters and both
 * moments keep their values.  Pass the status word of the plan that produced `grads` (stair_plan_info.status_off): a
 * recurrence that timed out then cannot feed NaN gradients into the optimizer state. */
int stair_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                    const int32_t *seg_of_block, const int32_t *touched, const float *step_of_seg, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int64_t n, const uint32_t *guard,
                    stair_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* STAIR_HIP_H */
