/* C-ABI of libdvae_hip.so -- the MI355X (gfx950) implementation of the D-VAE
 * train-step hot path (encode -> teacher-forced loss -> backward).
 *
 * The reference (muhanzhang/D-VAE) is pure Python/PyTorch and has NO FFI or
 * plugin interface for this path: its boundary is the
 * Python class API models.DVAE / models.DVAE_BN.  This C-ABI therefore sits
 * BELOW that boundary; each entry point cites the reference code it replaces.
 * The Python host mirror (d-vae_amd/models.py) keeps the reference's class API
 * and calls these through ctypes.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  All `float*` / `int32_t*`
 *     arguments are DEVICE pointers unless a comment says "host".
 *   - every function returns 0 on success or a negative code and records a
 *     message retrievable with dvae_last_error() (thread-local).
 *   - functions taking `void* stream` (a hipStream_t) only enqueue work on that
 *     stream; they never allocate, never synchronise the device.
 *   - all arithmetic is fp32 (MFMA v_mfma_f32_32x32x2_f32 for contractions) unless the
 *     model descriptor asks for bf16 matrix-unit inputs (mfma_bf16, reported separately).
 *   - feature dimensions are zero-padded to multiples of 32 ("Hp" = padded hs).
 */
#ifndef DVAE_HIP_H
#define DVAE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* dvae_last_error(void);
int dvae_version(void);

/* ---------------------------------------------------------------------------
 * Building blocks (exported so that tests can check each kernel on its own)
 * ------------------------------------------------------------------------- */

/* C[MxN] (+)= act(A'[MxK] B'[KxN] + bias).  Replaces the torch.nn.Linear /
 * addmm call sites of the path: models.py:629 (gate, mapper), :670 (GRUCell
 * hidden product), :745 (add_edge), :823 (add_vertex), :730 (fc1, fc2), :810
 * (fc3), and their autograd backward products.
 * a_kmajor / b_kmajor = 0: operand stored [rows][K] (K contiguous);
 *                     = 1: operand stored [K][rows].
 * a_rows/b_rows (nullable) gather rows of the stored matrix; c_rows (nullable)
 * scatters output rows (indices must be unique when accumulate = 1).
 * act: 0 none, 1 relu, 2 tanh. */
typedef struct dvae_gemm_desc {
    const float* A;
    const float* B;
    float* C;
    const int32_t* a_rows;
    const int32_t* b_rows;
    const int32_t* c_rows;
    const float* bias;
    int32_t lda, ldb, ldc;
    int32_t M, N, K;
    int32_t a_kmajor, b_kmajor;
    int32_t act;
    int32_t accumulate;
    const float* relu_mask; /* nullable, [M][ldc] laid out like C: C = mask > 0 ? result : 0 (the backward of a
                               ReLU whose output is `mask`, models.py:566 add_vertex; not with accumulate) */
    int32_t b_rows_bound;   /* with b_kmajor and b_rows: the number of rows of the stored B matrix (every index in
                               b_rows is below it).  Required there: K-major operands are addressed with 32-bit byte
                               offsets, and a gather cannot be rebased chunk by chunk, so b_rows_bound * ldb * 4 must
                               stay below 4 GiB or the call is refused.  Ignored otherwise. */
    int32_t a_bf16, b_bf16; /* dvae_gemm_bf16 only: the operand is STORED as bf16 (A / B then point at uint16 bf16 words,
                               lda / ldb count bf16 elements).  Two K-contiguous operands, either or both stored so
                               (16-byte aligned rows, K % 8 == 0; gemm_bf16_nn.hip), or the weight-gradient form -- two
                               K-major operands, accumulate, no bias / activation -- with a_bf16 and optionally b_bf16
                               (8-byte aligned rows; gemm_bf16.hip's 256 x 256 tile, whatever the shape).  Half the
                               operand bytes and no conversion in the K loop.  dvae_gemm_f32 refuses them. */
    int32_t c_bf16;         /* dvae_gemm_bf16 only, products of two K-contiguous operands only: C is STORED as bf16 (C then
                               points at uint16 bf16 words, ldc counts bf16 elements; 8-byte aligned rows, N % 4 == 0;
                               bias and c_rows allowed; no activation, accumulate or relu_mask).  For outputs that only
                               feed bf16-tolerant consumers (the edge predictor's tables in bf16 mode). */
} dvae_gemm_desc;
int dvae_gemm_f32(const dvae_gemm_desc* d, void* stream);
/* The same product on the kernel for FEW ROWS (csrc/gemm_skinny.hip): 16-row x (16 nt)-column tiles whose eight waves
 * are K slices, operands straight from L2 into 16x16x4 MFMA fragments.  dvae_gemm_f32 (and, in bf16 mode, the train
 * step: such products stay exact fp32) picks it by itself up to 1.6e8 multiply-adds (DVAE_GEMM_SKINNY_MMAC); this entry point forces
 * it at any M with nt = 1, 2 or 4 blocks per tile (0: the library's choice; a K-major B takes 1 or 4) - tests and
 * tools/gemm_skinny_bench.py.  Serves A K-contiguous (a_rows allowed), no b_rows, K % 16 == 0, N % 4 == 0, 16-byte
 * aligned operands; anything else is refused.  The products it replaces: every nn.Linear forward / input gradient of
 * a batch-32 step, e.g. models.py:563-567 (add_vertex), :568-572 (add_edge), :557-558 (fc1, fc2), :562 (fc3). */
int dvae_gemm_skinny_f32(const dvae_gemm_desc* d, int32_t nt, void* stream);
/* Diagnostics: the size (millions of multiply-adds) up to which products go to that kernel by themselves; 0 = never,
 * negative restores the default (160, or DVAE_GEMM_SKINNY_MMAC).  Tests of the bf16 mode set 0 so that small test
 * batches reach the bf16 kernels.  Process-wide, not thread-safe; no counterpart in the reference. */
int dvae_debug_set_gemm_skinny_mmac(int32_t mmac);
/* Diagnostics: lowers the span (bytes from its base pointer) a K-major operand may cover, normally
 * 4 GiB - a contraction over a longer K-major operand runs as accumulating K chunks, a longer gathered one is
 * refused - so that tests reach both paths at small sizes.  0 restores the default.  Process-wide, not
 * thread-safe; no counterpart in the reference. */
int dvae_debug_set_kmajor_span_limit(int64_t bytes);
/* The same product with the operands rounded to bf16 (round to nearest even) on
 * their way into the matrix unit (v_mfma_f32_32x32x16_bf16, fp32 accumulate);
 * operands and result stay fp32 in memory.  K-contiguous operands need only
 * K % 4 == 0.  This is the reduced-precision mode of dvae_model_desc.mfma_bf16. */
int dvae_gemm_bf16(const dvae_gemm_desc* d, void* stream);
/* dst[i] = bf16(src[i]) (round to nearest even), n % 4 == 0: makes the bf16 operand copies that a_bf16 / b_bf16 name.
 * No counterpart in the reference (which has no reduced-precision mode). */
int dvae_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);

/* out[r][0:width] (+)= sum_{k in [off[r], off[r+1])} src[idx[k]][0:width]
 * The gated-sum over predecessors of models.py:661-669 (padding rows of the
 * reference contribute exactly 0 and are simply absent from the CSR lists). */
int dvae_gather_sum_f32(float* out, int32_t ld_out, const float* src, int32_t ld_src, const int32_t* off,
                        const int32_t* idx, int32_t rows, int32_t width, int32_t accumulate, void* stream);

/* dst[idx[k]][0:width] += src[r][0:width] for k in [off[r], off[r+1]); the idx
 * values must be globally unique (no atomics are used). */
int dvae_scatter_rows_add_f32(float* dst, int32_t ld_dst, const float* src, int32_t ld_src, const int32_t* off,
                              const int32_t* idx, int32_t rows, int32_t width, void* stream);

/* torch.nn.GRUCell pointwise part (models.py:670) on a padded layout.
 * gh [rows][3Hp]: in = h_prev W_hh^T (no bias); out = (r, z, n) activations.
 * gi_tab [T][3Hp] = W_ih[:, t] + b_ih (+ b_hh for the r and z blocks),
 * bhn [Hp] = b_hh of the n block, types[rows] selects the gi_tab row,
 * hprev nullable (zeros).  ghn_out [rows][Hp] keeps W_hn h + b_hn for backward. */
int dvae_gru_pointwise_fwd(float* gh, const float* gi_tab, const int32_t* types, const float* bhn,
                           const float* hprev, int32_t ld_hprev, float* hout, int32_t ld_hout, float* ghn_out,
                           int32_t rows, int32_t Hp, void* stream);
/* backward of the above: dgh [rows][3Hp] = (d pre_r, d pre_z, d ghn),
 * dgin [rows][Hp] = d pre_n, dhprev [rows][Hp] = dh * z (direct path). */
int dvae_gru_pointwise_bwd(const float* dh, int32_t ld_dh, const float* gates, const float* ghn,
                           const float* hprev, int32_t ld_hprev, float* dgh, float* dgin, float* dhprev,
                           int32_t ld_dhprev, int32_t rows, int32_t Hp, void* stream);

/* _gated (models.py:628-629) pointwise part: gm [rows][2Hp] in = (Wg h | Wm h),
 * out = (g = sigmoid(. + gv_g), m = . + gv_m); msg = g * m.  gv [levels][2Hp]
 * holds the vertex-id column of the gate/mapper weights (+ gate bias); the row
 * used is (row_base + row) / rows_per_level. */
int dvae_gated_pointwise_fwd(float* gm, const float* gv, float* msg, int32_t ld_msg, int32_t rows, int32_t Hp,
                             int32_t row_base, int32_t rows_per_level, void* stream);
int dvae_gated_pointwise_bwd(const float* dmsg, int32_t ld_dmsg, const float* gm, float* dgm, int32_t rows,
                             int32_t Hp, void* stream);

/* The same forward pointwise part for the level stored at rows [cur_row0,
 * cur_row0 + rows) of msg, fused with the gather-sum of models.py:661-669 for the
 * NEXT level of the pass: next_agg[r] = sum_{k in [next_off[r], next_off[r+1])}
 * msg[idx[k]] (next_agg NULL: pointwise part only).  gm / gv_row point at the
 * current level's rows / vertex-id row.  A listed source slot that lies in the
 * current level must be cur_row0 + r (the same graph's vertex), which holds for
 * the packer's CSR lists; it is taken from registers. */
int dvae_gated_fwd_gather(float* gm, const float* gv_row, float* msg, int32_t ld_msg, int32_t cur_row0,
                          float* next_agg, int32_t ld_agg, const int32_t* next_off, const int32_t* idx, int32_t rows,
                          int32_t Hp, void* stream);
/* Backward twin: d msg[r] = sum of dagg[idx[k]] over list 1 (+ dagg2[idx2[k]] over
 * the optional list 2), fed straight into the pointwise backward -> dgm. */
int dvae_gated_bwd_gather(const float* dagg, int32_t ld, const int32_t* off, const int32_t* idx, const float* dagg2,
                          const int32_t* off2, const int32_t* idx2, const float* gm, float* dgm, int32_t rows,
                          int32_t Hp, void* stream);

/* One level of the state chain in ONE launch (fp32 MFMA; csrc/chain.hip): the hidden product of
 * torch.nn.GRUCell (models.py:670) and its pointwise part.  hprev [rows][Hp] (ld_hprev) is both
 * the product's operand and h_prev; whh [3Hp][Hp] is the padded W_hh (r, z, n blocks); gi_tab /
 * types / bhn as for dvae_gru_pointwise_fwd.  Writes gates [rows][3Hp] = (r, z, n), ghn_out
 * [rows][Hp] = W_hn h + b_hn and hout.  The pre-activations never reach memory.  Equals
 * dvae_gemm_f32 + dvae_gru_pointwise_fwd up to the summation order over K (two halves). */
int dvae_gru_level_fwd(const float* hprev, int32_t ld_hprev, const float* whh, const float* gi_tab,
                       const int32_t* types, const float* bhn, float* gates, float* ghn_out, float* hout,
                       int32_t ld_hout, int32_t rows, int32_t Hp, void* stream);
/* _gated (models.py:628-629) of a level in ONE launch: the product h [Wg | Wm]^T (wgm [2Hp][Hp]),
 * its pointwise part (gm [rows][2Hp] = (g, m), msg[cur_row0 + r] = g * m) and the gather-sum of
 * models.py:661-669 for the next level (next_agg nullable), with the arguments and the contract
 * of dvae_gated_fwd_gather. */
int dvae_gated_level_fwd(const float* h, int32_t ld_h, const float* wgm, const float* gv_row, float* gm, float* msg,
                         int32_t ld_msg, int32_t cur_row0, float* next_agg, int32_t ld_agg, const int32_t* next_off,
                         const int32_t* idx, int32_t rows, int32_t Hp, void* stream);

/* Backward twins (csrc/chain.hip; deterministic, no atomics).  w [K][Hp] is an nn.Linear weight
 * [out = K][in = Hp]; dy [rows][K].
 *   dvae_level_dgrad_acc: dx[rows][Hp] += dy w           (the input gradient of the Linear)
 *   dvae_gru_level_bwd:   dh += dy w, then the pointwise backward of torch.nn.GRUCell from dh in
 *                         the same launch, with the arguments and outputs of dvae_gru_pointwise_bwd
 *                         (dgh, dgin, dhprev). */
int dvae_level_dgrad_acc(const float* dy, int32_t ld_dy, int32_t K, const float* w, float* dx, int32_t ld_dx,
                         int32_t rows, int32_t Hp, void* stream);
int dvae_gru_level_bwd(const float* dy, int32_t ld_dy, int32_t K, const float* w, float* dh, int32_t ld_dh,
                       const float* gates, const float* ghn, const float* hprev, int32_t ld_hprev, float* dgh,
                       float* dgin, float* dhprev, int32_t ld_dhprev, int32_t rows, int32_t Hp, void* stream);

/* The four level operations above as ONE launch over up to two independent levels of the same
 * shape (the two directions of the bidirectional encoder, models.py:724-727).  Fields a level
 * operation does not use are ignored:
 *   GRU_FWD    a = hprev (lda), w = whh, gi_tab, types, bhn -> gates, ghn, hout (ld_hout)
 *   GATED_FWD  a = h (lda), w = wgm, gv_row -> gm, msg (ld_msg, cur_row0), next_agg (ld_agg, next_off, idx)
 *   DGRAD_ACC  a = dy (lda), K, w -> dx (ld_dx); optionally (dgm_next != NULL) followed, in the same launch,
 *              by dvae_gated_bwd_gather for the level processed NEXT, whose consumers are dx rows of this
 *              launch (slot cur_row0 + r: taken from registers) and finished ones: gather_src (all levels'
 *              dx, pitch ld_dx), next_off, idx (+ optional gather_src2, off2, idx2), gm -> dgm_next
 *   GRU_BWD    a = dy (lda), K, w, gates, ghn, hprev (ld_hprev) -> dx = dh (ld_dx), dgh, dgin, dhprev (ld_dhprev)
 * With two levels the products run as ONE grid whose workgroups contract their tile for both levels,
 * alternating K steps (two accumulator sets): per-workgroup latencies are paid once. */
enum { DVAE_LEVEL_GRU_FWD = 0, DVAE_LEVEL_GATED_FWD = 1, DVAE_LEVEL_DGRAD_ACC = 2, DVAE_LEVEL_GRU_BWD = 3 };
typedef struct dvae_level_desc {
    const float* a;
    const float* w;
    const float* gi_tab;
    const int32_t* types;
    const float* bhn;
    float* gates;
    float* ghn;
    float* hout;
    const float* gv_row;
    float* gm;
    float* msg;
    float* next_agg;
    const int32_t* next_off;
    const int32_t* idx;
    float* dx;
    const float* hprev;
    float* dgh;
    float* dgin;
    float* dhprev;
    const float* gather_src;
    const float* gather_src2;
    const int32_t* off2;
    const int32_t* idx2;
    float* dgm_next;
    int32_t lda, K, ld_hout, ld_msg, cur_row0, ld_agg, ld_dx, ld_hprev, ld_dhprev;
    /* bf16 matrix-unit inputs (optional; NULL = the exact-fp32 kernels): a K-CONTIGUOUS bf16 copy of the weight, row
     * pitch ldw_bf16 elements -- the forward operations' [NGATE * Hp][Hp] as stored, the backward operations' weight
     * TRANSPOSED, [Hp][K] -- for the bf16 level kernels (`a` rounded to bf16 on its way to the matrix unit, fp32
     * accumulation and tails).  Needs Hp % 64 == 0 and K % 64 == 0; one level per launch. */
    int32_t ldw_bf16;
    const uint16_t* w_bf16;
} dvae_level_desc;
int dvae_chain_level_batch(int32_t op, const dvae_level_desc* levels, int32_t count, int32_t rows, int32_t Hp,
                           void* stream);

/* out[seg][0:width] += sum over the seg_rows rows of segment seg of x */
int dvae_segment_colsum_f32(float* out, int32_t ld_out, const float* x, int32_t ld_x, int32_t rows,
                            int32_t seg_rows, int32_t width, void* stream);

/* The input-side and bias gradients of a GRUCell whose input is a one-hot vertex type (models.py:653,
 * :670): with x = onehot(type), d W_ih[:, t] and d b_ih are per-type column sums of the gate gradients.
 * dvae_type_csr sorts the rows na + nb (two type arrays laid end to end; nb may be 0) by type:
 * off[T + 1], idx[na + nb].  dvae_type_colsum reads dgh [rows][3Hp] and dgin [rows][Hp] once and ADDS
 *   gi[t][0:2Hp]   += sum over rows of type t of dgh[:, 0:2Hp]
 *   bhn[0:Hp]      += sum over all rows of dgh[:, 2Hp:3Hp]
 *   gi[t][2Hp:3Hp] += sum over rows of type t of dgin
 * (gi [T][3Hp] as for dvae_gru_pointwise_fwd; fp32 atomics, so the last bits vary run to run). */
int dvae_type_csr(const int32_t* types_a, int32_t na, const int32_t* types_b, int32_t nb, int32_t T, int32_t* off,
                  int32_t* idx, void* stream);
int dvae_type_colsum(const float* dgh, const float* dgin, const int32_t* off, const int32_t* idx, int32_t T, float* gi,
                     float* bhn, int32_t rows, int32_t Hp, void* stream);

/* Edge predictor tail + BCE (models.py:745, :855): per pair p
 *   s = b2 + sum_j w2[j] relu(EA[a_p][j] + EB[h_p][j] (+ EC[c_p][j]) + b1[j])
 *   prob = sigmoid(s);  loss_p = -(t log prob + (1-t) log(1-prob)), logs clamped at -100
 * Writes dl[p] = d loss_p / d s (PyTorch BCE-on-probabilities backward) and adds
 * sum_p loss_p into *loss_sum.  ec/pair_c nullable (DVAE).
 * For the backward passes it also leaves (both nullable):
 *   mask         the ReLU mask of every pair, dvae_pair_mask_words(n_pairs, HEp) uint32 words
 *                (1 bit per hidden unit; layout private to the two kernels)
 *   dw2_partial  [n_blocks][HEp]: per-workgroup partial sums of dl[p] relu(pre_p[j]); their
 *                column sum times the upstream gradient is the gradient of w2.
 * The launch uses exactly n_blocks workgroups. */
int64_t dvae_pair_mask_words(int32_t n_pairs, int32_t HEp);
int dvae_pair_bce_fwd(const float* ea, const float* eb, const float* ec, const int32_t* pair_a,
                      const int32_t* pair_h, const int32_t* pair_c, const uint8_t* pair_t, const float* b1,
                      const float* w2, const float* b2, float* dl, float* loss_sum, uint32_t* mask,
                      float* dw2_partial, int32_t n_blocks, int32_t n_pairs, int32_t HEp, void* stream);
/* Grouped backward pass onto one of the edge-predictor tables.  The pairs that
 * share one row of the table form a group; groups are cut into chunks (one
 * wavefront each): chunk q covers grp_pairs[chunk_off[q] .. chunk_off[q+1])
 * (grp_pairs NULL: the positions are the pair ids), chunk_row[q] = the row, with
 * bit 31 set when the row has more than one chunk (those chunks atomicAdd into
 * the pre-zeroed table).
 *   dtab[row][j] = scale * w2[j] * sum_p dl[p] [pre_p[j] > 0]   (mask from the forward) */
int dvae_pair_bce_bwd_grouped(const int32_t* chunk_off, const int32_t* chunk_row, const int32_t* grp_pairs,
                              const float* dl, const uint32_t* mask, const float* w2, const float* scale,
                              float* dtab, int32_t n_chunks, int32_t HEp, void* stream);

/* Vertex-type CE (models.py:823-826): per row of logits [R][ld] with T valid
 * columns and a CSR list of targets, adds -sum log_softmax[target] into
 * *loss_sum and writes dlogits[row][0:Tp] = sum_items (softmax - onehot). */
int dvae_vertex_ce(const float* logits, int32_t ld, const int32_t* ce_off, const int32_t* ce_tgt, float* dlogits,
                   float* loss_sum, int32_t R, int32_t T, int32_t Tp, void* stream);

/* ---------------------------------------------------------------------------
 * Model description, weights
 * ------------------------------------------------------------------------- */

/* Constructor arguments of models.DVAE / DVAE_BN (models.py:537, :980). */
typedef struct dvae_model_desc {
    int32_t max_n, nvt, hs, nz;
    int32_t bidir, vid, is_bn;
    int32_t start_type, end_type;
    int32_t mfma_bf16; /* 0: every contraction in exact fp32 (the parity mode); 1: contractions take
                          bf16-rounded operands with fp32 accumulation (dvae_gemm_bf16); weights,
                          activations, gradients and all pointwise arithmetic stay fp32 */
} dvae_model_desc;

/* Parameters are passed as an array of DVAE_NUM_PARAMS device pointers in
 * state_dict order: grue_forward.{weight_ih,weight_hh,bias_ih,
 * bias_hh}, grue_backward.*, fc1.{weight,bias}, fc2.*, grud.*, fc3.*,
 * add_vertex.0.*, add_vertex.2.*, add_edge.0.*, add_edge.2.*,
 * gate_forward.0.{weight,bias}, gate_backward.0.*, mapper_forward.0.weight,
 * mapper_backward.0.weight, hv_unify.0.{weight,bias}, hg_unify.0.{weight,bias}.
 * Absent parameters (hv/hg_unify when unidirectional) are NULL. */
#define DVAE_NUM_PARAMS 36

/* size (floats) of the packed, zero-padded weight buffer / gradient buffer */
int64_t dvae_packed_floats(const dvae_model_desc* m);
/* native parameters -> packed buffer (call after every optimizer step) */
int dvae_pack_weights(const dvae_model_desc* m, const float* const* params, float* packed, void* stream);
/* the same into a buffer that already holds a pack of this model (its zero padding is intact): only the
 * live entries are rewritten, in one launch */
int dvae_repack_weights(const dvae_model_desc* m, const float* const* params, float* packed, void* stream);
/* packed gradient buffer -> native gradient tensors; want[i] != 0 selects which
 * of grads[0..DVAE_NUM_PARAMS) are written: fully overwritten, or, with accumulate != 0, added to
 * in place (PyTorch's .grad accumulation without a temporary). */
int dvae_unpack_grads(const dvae_model_desc* m, const float* packed_w, const float* packed_g, float* const* grads,
                      const uint8_t* want, int32_t accumulate, void* stream);

/* One Adam step over a flat fp32 buffer of n floats (n % 4 == 0; parameters, gradients and the
 * two moments laid out alike): torch.optim.Adam's update as the reference configures it
 * (train.py:174; no weight decay, no amsgrad).  `step` counts from 1 (bias correction).  zero_grad != 0: the
 * gradient buffer is left at zero (the optimizer.zero_grad() of the next step, train.py:225, in the same pass). */
int dvae_adam_step_f32(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                       float eps, int32_t step, int32_t zero_grad, void* stream);
/* The same update with the learning rate and the step count kept in DEVICE memory (lr_dev: one float; step_dev: one
 * int32 = steps taken so far, incremented by the call), so that the enqueued work carries no per-step host scalar
 * and a captured train step can be replayed as it is; torch.optim.Adam's `capturable=True` is the counterpart. */
int dvae_adam_step_dev_f32(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1,
                           float beta2, float eps, int32_t* step_dev, int32_t zero_grad, void* stream);

/* ---------------------------------------------------------------------------
 * DAG batching (host).  Replaces models.py:631-632 (_collate_fn), the per-vertex
 * gather / zero-pad of models.py:637-668 and util.py:557-573 (batch sharding is
 * done by the caller choosing which graphs go into a batch).
 * ------------------------------------------------------------------------- */
typedef struct dvae_batch_counts {
    int32_t B, n, T, is_bn;
    int32_t nB;          /* n * B vertex slots, slot(v, b) = v * B + b          */
    int32_t E;           /* edges in the batch                                   */
    int32_t PT, PL;      /* partial-state rows (first T: one per type), list len  */
    int32_t NA, NB;      /* rows of the two edge-predictor tables                */
    int32_t NP;          /* vertex pairs = B * n (n-1) / 2                       */
    int32_t CA, CH, CC;  /* backward chunks grouped by EA row / EB row / graph    */
    int32_t R, RL, NCE;  /* vertex-type rows, their source-list length, CE items  */
    int32_t HGF, HGB;    /* graph-state source-list lengths                      */
    int32_t level_major; /* 1: every graph has n vertices, so the rows of the EA table and the
                            vertex-type rows of level v are [v B, (v+1) B): the loss forward then
                            runs those products level by level beside the state chain            */
    int64_t total_ints;  /* size of the exported index blob, in int32            */
} dvae_batch_counts;

/* HOST pointers: types [B][max_n] (entries past a graph's size ignored),
 * pred_mask [B][max_n] (bit u of pred_mask[b][v] set iff edge u -> v; vertex ids
 * are a topological order as in models.py:677), n_vertices [B].  Malformed
 * input (type out of range, edge u >= v, too many vertices) is rejected. */
int dvae_batch_create(int32_t B, int32_t max_n, int32_t nvt, int32_t start_type, int32_t is_bn,
                      const int32_t* types, const uint64_t* pred_mask, const int32_t* n_vertices, void** handle,
                      dvae_batch_counts* counts);
/* writes the index blob (counts.total_ints int32) into HOST memory `out` */
int dvae_batch_export(void* handle, int32_t* out, int64_t out_ints);
/* Shape-stable export for callers that replay ONE recorded launch sequence (a captured HIP graph of the train step,
 * train.py:224-253) over many batches.  The counts that depend on the batch's edges -- E, PT, PL, NB, CH -- are
 * raised to `caps` with inert padding entries (empty lists, rows no pair refers to, empty backward chunks; see
 * csrc/packer.cpp), so every batch of the same (B, max_n, nvt, kind, caps) yields the same `counts_out`, blob
 * layout, workspace sizes, grids and kernel arguments, and the same results as the plain export.  Requires every
 * graph to have max_n vertices; a batch above a cap is refused (negative return): export it plainly instead.
 * No counterpart in the reference (its batching is a Python list, models.py:631-632). */
typedef struct dvae_batch_caps {
    int32_t E, PT, PL, NB, CH;
} dvae_batch_caps;
int dvae_batch_export_padded(void* handle, const dvae_batch_caps* caps, int32_t* out, int64_t out_ints,
                             dvae_batch_counts* counts_out);
void dvae_batch_destroy(void* handle);

/* ---------------------------------------------------------------------------
 * The hot path.  `blob` is the exported index blob copied to the device, `pw`
 * the packed weights, `ws` a workspace of dvae_*_ws_floats() floats that must
 * stay untouched between a forward and its backward, `pg` a packed gradient
 * buffer (dvae_packed_floats) that the backward overwrites.
 * ------------------------------------------------------------------------- */
int64_t dvae_encode_ws_floats(const dvae_model_desc* m, const dvae_batch_counts* c);
int64_t dvae_loss_ws_floats(const dvae_model_desc* m, const dvae_batch_counts* c);

/* Concurrency.  The four entry points below run part of their work on an internal side stream (the second
 * encoder direction, the predictors' products and weight gradients), forked from and joined back into the
 * caller's stream with events, so that on return everything is ordered on the caller's stream.  Side stream
 * and events are kept per (device, caller stream): calls on different streams -- from one host thread or
 * several -- are independent of each other; calls that share one stream are ordered by it as usual.  A
 * workspace belongs to one call at a time.  dvae_debug_side_stream reports the side stream the library pairs
 * with a caller stream on the current device (diagnostics / tests; no counterpart in the reference). */
int dvae_debug_side_stream(void* caller_stream, void** side_stream);
/* Kernel launches the library has enqueued in this process so far (all threads, all devices; its memsets and copies
 * are not counted).  Diagnostics: bench.py reports the difference over a step as config.launches_per_step; no
 * counterpart in the reference. */
int64_t dvae_debug_launch_count(void);

/* DVAE.encode, models.py:720-731 (DVAE_BN.encode :1068-1080): mu, logvar [B][nz] */
int dvae_encode_forward(const dvae_model_desc* m, const dvae_batch_counts* c, const int32_t* blob, const float* pw,
                        float* ws, float* mu, float* logvar, void* stream);
/* autograd of the above */
int dvae_encode_backward(const dvae_model_desc* m, const dvae_batch_counts* c, const int32_t* blob, const float* pw,
                         float* ws, const float* dmu, const float* dlogvar, float* pg, void* stream);

/* DVAE.loss, models.py:806-860 (reparameterize :733-740 with the normal draw
 * `eps` [B][nz] injected): out3 = {res + beta*kld, res, kld}. */
int dvae_loss_forward(const dvae_model_desc* m, const dvae_batch_counts* c, const int32_t* blob, const float* pw,
                      float* ws, const float* mu, const float* logvar, const float* eps, float eps_scale,
                      int32_t training, float beta, float* out3, void* stream);
/* autograd of the above; g3 = upstream gradients of the three outputs */
int dvae_loss_backward(const dvae_model_desc* m, const dvae_batch_counts* c, const int32_t* blob, const float* pw,
                       float* ws, const float* mu, const float* logvar, const float* eps, float eps_scale,
                       int32_t training, float beta, const float* g3, float* dmu, float* dlogvar, float* pg,
                       void* stream);

/* ---------------------------------------------------------------------------
 * Generation.  DVAE.decode, models.py:747-804 (DVAE_BN inherits it, with the
 * helpers of models.py:1006-1085): B latent points z [B][nz] -> B DAGs, grown
 * vertex by vertex entirely on the device.  Outputs (device): types [B][max_n],
 * pred_mask [B][max_n] (bit u of pred_mask[b][v] <=> edge u -> v), n_vertices [B].
 * stochastic != 0: the vertex type is an inverse-CDF sample of the softmax with
 * u_type[b][idx] (the rule of np.random.choice, models.py:766) and edge vi -> idx
 * is taken iff u_edge[b][idx][vi] < probability (models.py:783-784); both arrays
 * hold uniforms in [0, 1), shapes [B][max_n] and [B][max_n][max_n].
 * stochastic == 0: argmax type, edge iff probability > 0.5 (u_* may be NULL).
 * ------------------------------------------------------------------------- */
int64_t dvae_decode_ws_floats(const dvae_model_desc* m, int32_t B);
int dvae_decode(const dvae_model_desc* m, int32_t B, const float* pw, float* ws, const float* z, int32_t stochastic,
                const float* u_type, const float* u_edge, int32_t* types, uint64_t* pred_mask, int32_t* n_vertices,
                void* stream);

/* dvae_decode for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops that turn latent points into graphs, harnesses that forbid hipMalloc).  It has NO
 * counterpart in the reference.  Same arguments, outputs and results as dvae_decode, minus `ws`: the
 * workspace is a fixed array of DVAE_DECODE_STATIC_FLOATS floats (32 MiB) inside the library's code
 * object, one copy per GPU, held until the process ends.  The call allocates nothing and rejects a
 * (model, B) whose dvae_decode_ws_floats exceeds that size (negative return, text in
 * dvae_last_error()): e.g. B <= 263 at max_n = 8, hs = 501; B <= 20 at max_n = 64, hs = 1024.
 * All calls on one GPU share that GPU's array, so calls must not overlap on one GPU.  Calls on ONE
 * stream are ordered by the stream.  A call that arrives on another stream while the previous call's
 * work has not finished is REFUSED (negative return, "still running" in dvae_last_error()) instead
 * of racing: each call leaves an event behind its work and the next one queries it.  (Set-up on the
 * first call on a device: hipGetSymbolAddress and one hipEventCreate; no call allocates device
 * memory.  Under stream capture no event is left or queried: the order of captured calls is the
 * capturing caller's graph.) */
#define DVAE_DECODE_STATIC_FLOATS (8 * 1024 * 1024)
int64_t dvae_decode_static_max_floats(void);
int dvae_decode_static(const dvae_model_desc* m, int32_t B, const float* pw, const float* z, int32_t stochastic,
                       const float* u_type, const float* u_edge, int32_t* types, uint64_t* pred_mask,
                       int32_t* n_vertices, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVAE_HIP_H */
