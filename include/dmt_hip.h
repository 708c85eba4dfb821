/*
 * dmt_hip.h -- C ABI of libdmt_hip.so: the MI355X (gfx950) kernels behind the DMT train-step hot path.
 *
 * The reference (guyulongcs/CIKM2020_DMT) has no FFI: the hot path is TensorFlow-1.12 graph ops called
 * from plain Python.  Each entry point below therefore replaces the TF op group that one
 * reference Python function emits; the comment on every function cites that function
 * (paths relative to /root/reference/DMT_code/).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain C, POD arguments only; every pointer is a DEVICE pointer unless named h_*;
 *   - the caller owns every buffer (incl. workspaces); the library allocates nothing, keeps no global
 *     mutable state except the thread-local error string (and the off-by-default diagnostic launch-route
 *     counters of dmt_route_trace), never synchronises, and launches only on
 *     the `stream` argument (a hipStream_t passed as void*);
 *   - return 0 on success, <0 on error (message: dmt_last_error()); no C++ exception crosses the ABI;
 *   - dtype codes: DMT_F32 = 0, DMT_BF16 = 1.  Accumulation is always fp32.  Parameters that are
 *     vectors (bias, LayerNorm gamma/beta) and all embedding tables are fp32 masters in both modes.
 */
#ifndef DMT_HIP_H
#define DMT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMT_F32 0
#define DMT_BF16 1
#define DMT_FP8_E4M3 2     /* OCP e4m3fn: only as dmt_attn_desc.mma_dtype (long-sequence attention forward) */

#define DMT_OK 0
#define DMT_ERR_ARG (-1)
#define DMT_ERR_LAUNCH (-2)
#define DMT_ERR_UNSUPPORTED (-3)

#define DMT_MAX_FEATURES 32
#define DMT_MAX_SEQS 4
#define DMT_MAX_TABLES 32

/* ABI revision: bumped whenever an exported signature or a descriptor layout changes incompatibly (a binding compares it with the
 * revision it was written against before its first call -- cikm2020_dmt_amd/_lib.py does).
 *   1  rounds 1-2.   2  round 3: dmt_set/get_deterministic removed; dmt_colsum / dmt_colsum_drop (ordered), dmt_softmax_fwd / _bwd
 *   (causal) gained an int before `stream`; dmt_wgrad_desc grew (det_ws).   3  round 4: dmt_mhsa_block_fwd re-implemented (a new weight-image layout: images of revision 2 are not
 *   readable -- rebuild with dmt_mhsa_image_build; s_out may be NULL; B * T * 1920 < 2^31).   4  round 4: dmt_mmoe_desc grew (ws, ws_bytes,
 *   gate_dx); dmt_mmoe_experts_ws_bytes added.   5  round 5: PACKED ROWS (below): dmt_gather_desc / dmt_embgrad_desc grew (seq_row_off,
 *   seq_row_len), dmt_attn_desc (row_off, ex_list, n_list), dmt_mhsa_desc (packed-row fields), dmt_q1mem_desc (row_off); dmt_colsum_rows_packed added.
 *   6  round 6: dmt_mhsa_block_bwd + dmt_mhsa_bwd_image_* added; dmt_sort_pairs / dmt_segment_heads are the library's own kernels (vals_in may be NULL; 16-byte aligned workspace; the
 *   workspace sizes changed), dmt_embgrad_keys takes vals == NULL.
 *
 * PACKED ROWS.  A behaviour sequence of a batch may be stored WITHOUT its padding: example b's rows t = 0 .. len[b] - 1 are rows
 * row_off[b] + t of an [R, d] matrix, R = sum_b len[b] (row_off: int32 [B], any order of the examples -- the engine groups examples of
 * one length class).  Every row-wise kernel (dmt_chain2, dmt_proj, dmt_gemm, dmt_ln_*, dmt_wgrad320) just sees R rows; the kernels that
 * know about examples take row_off (NULL = the dense [B, T, d] layout, row b * T + t).  Rows t >= len[b] do not exist: in the dense
 * layout they hold finite values nothing reads (the reference masks them as keys, and their gradients are zero), so the
 * two layouts agree on every row that exists.  Dropout counters keep the DENSE element index, so a packed and a dense run draw the same mask. */
#define DMT_ABI_VERSION 6
const char* dmt_last_error(void);
int dmt_version(void);
/* gfx arch string the device code was built for ("gfx950"). */
const char* dmt_build_arch(void);
/* sizeof() of the ABI structs: 0 gather_feature, 1 gather_desc, 2 embgrad_desc, 3 gemm_desc, 4 attn_desc,
 * 5 attn_bwd_desc, 6 table_map, 7 cast_job, 8 chain_desc, 9 wgrad_desc, 10 mhsa_desc, 11 mmoe_desc, 12 heads_desc, 13 q1mem_desc,
 * 14 mhsa_bwd_desc (lets a binding verify its struct layout). */
int dmt_struct_size(int which);
/* Launch-route trace (diagnostic, off by default).  dmt_route_trace(1) clears the counters and starts counting every successful launch
 * under its route label (the name of the kernel variant an entry point dispatched to, e.g. "dmt_attn_fwd(mfma, coalesced)",
 * "dmt_proj", "dmt_chain2", "dmt_wgrad320", "dmt_q1mem_fwd"); dmt_route_trace(0) stops.  dmt_route_count(label) -> launches counted
 * under exactly that label; dmt_route_dump writes "label=count" lines.  Used by the parity tests to state which kernels they covered. */
int dmt_route_trace(int32_t on);
int64_t dmt_route_count(const char* label);
int dmt_route_dump(char* buf, int32_t cap);

/* ------------------------------------------------------------------------------------------------
 * Embedding gather + concat + mean-pool (forward).
 * Replaces: mmoe_transformer_unbias.generate_data (model/net/mmoe_transformer_unbias.py:130-186),
 *           base.embedding / base.embedding_combiner (model/net/base.py:81-134), the input prep of
 *           TransformerModel.encode/decode (model/net/TransformerModel.py:96-100,146-147: *sqrt(d_model)
 *           + learned positions) and the tf.concat that assembles the MMoE input
 *           (mmoe_transformer_unbias.py:226-233).
 * One feature = one (table, id column) pair of dmt.conf's `emb` list.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const float* table;   /* fp32 [rows, dim] embedding variable (NO zero row prepended)              */
  int32_t rows, dim;
  const int32_t* idx;   /* [B, T] left-aligned, padded with 0                                      */
  const float* wts;     /* [B, T] sp_weights or NULL (== 1.0)                                      */
  const int32_t* lens;  /* [B] number of valid entries                                             */
  int32_t T;            /* padded length == row stride of idx / wts                                */
  int32_t pooled_off;   /* column of the mean-pooled embedding in `pooled` (<0: not pooled)        */
  int32_t seq_id;       /* sequence output this feature feeds (0..n_seq-1), DMT_SEQ_TARGET, or <0  */
  int32_t seq_off;      /* column inside the d_model-wide sequence / target row                    */
  int32_t group;        /* features with equal group are processed by the same workgroup           */
  float* inv_wsum;      /* [B] out: 1/sum(w) per example (0 when empty), needed by backward; or NULL */
  /* Row-cache form (row-sharded tables, BASELINE configs[3]): `table` is then the step's cache of the rows this batch reads
   * ([n_distinct, row_stride] fp32, filled by the all-to-all exchange), idx holds cache slots for the pooled path and idx_seq
   * (slot + 1, 0 = the zero row of [0;E]) for the sequence path.  Plain tables: row_stride = 0 (= dim), idx_seq = NULL.       */
  int32_t row_stride;
  const int32_t* idx_seq;
} dmt_gather_feature;

#define DMT_SEQ_TARGET 100

typedef struct {
  int32_t B;
  int32_t n_features;
  dmt_gather_feature feat[DMT_MAX_FEATURES];
  int32_t n_seq;
  void* seq_out[DMT_MAX_SEQS];      /* [B, seq_T[s], d_model]  = seq_scale * [0;E][idx] + pos[s][t]  */
  int32_t seq_T[DMT_MAX_SEQS];
  const float* pos[DMT_MAX_SEQS];   /* fp32 [>=seq_T, d_model] learned positions or NULL             */
  void* tar_out;                    /* [B, d_model] = seq_scale * [0;E][item idx]  (or NULL)         */
  int32_t d_model;
  float seq_scale;                  /* sqrt(d_model) (TransformerModel.py:96) or 1.0 for raw output  */
  void* pooled;                     /* [B, ld_pooled]: dense | pooled embeddings (MMoE input z)      */
  int64_t ld_pooled;
  const float* dense;               /* fp32 [B, n_dense] 'features' or NULL                          */
  int32_t n_dense;                  /* copied to pooled[:, 0:n_dense]                                */
  int32_t out_dtype;                /* dtype of seq_out / tar_out / pooled                           */
  /* Block-input dropout fused into the sequence outputs (TransformerModel.py:101  tf.layers.dropout(enc, rate, training)):
   * seq_out[s] element i (flat index into [B, seq_T[s], d_model]) is kept iff the counter hash of
   * (i ^ seq_drop_seed[s]) passes seq_drop_keep, and scaled by 1/keep; seq_drop_keep <= 0 or >= 1: off.  */
  uint32_t seq_drop_seed[DMT_MAX_SEQS];
  float seq_drop_keep;
  /* revision 5, packed rows: seq_out[s] is [R_s, d_model]; example b's rows t < seq_row_len[s][b] go to rows seq_row_off[s][b] + t, rows
   * past seq_row_len are not written.  NULL (both): the dense layout.  The dropout index stays the dense one.                          */
  const int32_t* seq_row_off[DMT_MAX_SEQS];
  const int32_t* seq_row_len[DMT_MAX_SEQS];
} dmt_gather_desc;

int dmt_gather_fwd(const dmt_gather_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embedding gradient: sparse (row, grad) pairs.
 * Replaces: the IndexedSlices gradients TF autodiff produces for tf.nn.embedding_lookup /
 *           embedding_lookup_sparse (call sites base.py:116, mmoe_transformer_unbias.py:155,158).
 * Entry e of the flattened list  [feature f][kind: 0 pooled, 1 seq][b][t]  contributes
 *      kind 0: (w[b,t] * inv_wsum[b]) * dpooled[b, pooled_off : +dim]  to table row  idx[b,t]
 *      kind 1: seq_scale * dseq[seq_id][b, t, seq_off : +dim]          to table row  idx[b,t]-1 (idx>0)
 * Rows are addressed by a GLOBAL row id = row_base[table] + row (tables concatenated).
 * Pipeline: dmt_embgrad_keys -> dmt_sort_pairs -> dmt_segment_heads -> dmt_embgrad_reduce.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t B;
  int32_t n_features;
  dmt_gather_feature feat[DMT_MAX_FEATURES];   /* same list as forward (table ptr unused)            */
  int32_t row_base[DMT_MAX_FEATURES];          /* global row id of row 0 of this feature's table     */
  int32_t entry_base[DMT_MAX_FEATURES + 1];    /* first entry of feature f; kinds laid out pooled then seq */
  int32_t total_rows;                          /* invalid key == total_rows                          */
  const void* dseq[DMT_MAX_SEQS];              /* grad wrt seq_out[s] [B, seq_T[s], d_model]         */
  int32_t seq_T[DMT_MAX_SEQS];
  const void* dtar;                            /* grad wrt tar_out [B, d_model]                      */
  const void* dpooled;                         /* grad wrt pooled [B, ld_pooled]                     */
  int64_t ld_pooled;
  int32_t d_model;
  float seq_scale;
  int32_t grad_dtype;
  /* the same dropout mask as dmt_gather_desc, applied to dseq[s] while it is read (gradient of the fused dropout) */
  uint32_t seq_drop_seed[DMT_MAX_SEQS];
  float seq_drop_keep;
  /* revision 5, packed rows: dseq[s] is [R_s, d_model]; entry (b, t) reads row seq_row_off[s][b] + t when t < seq_row_len[s][b] and
   * contributes nothing otherwise.  NULL: dense.                                                                                       */
  const int32_t* seq_row_off[DMT_MAX_SEQS];
  const int32_t* seq_row_len[DMT_MAX_SEQS];
} dmt_embgrad_desc;

/* keys[e] (uint32 global row or total_rows if the entry carries no gradient), vals[e] = e (vals may be NULL: dmt_sort_pairs numbers
 * the entries itself when its vals_in is NULL).                                                                                     */
int dmt_embgrad_keys(const dmt_embgrad_desc* d, uint32_t* keys, uint32_t* vals, void* stream);

/* out[i, :] = in[perm[i], :] for fp32 rows of `dim` floats (dim % 4 == 0), written as fp32 or rounded to bf16: groups a
 * rank's gradient rows by owner rank and puts them into the wire format of the data-parallel exchange in one pass.     */
int dmt_rows_permute(const float* in_rows, const int64_t* perm, int64_t n, int32_t dim, int32_t out_dtype, void* out_rows,
                     void* stream);

/* Stable LSD radix sort of (key, value) pairs on bits [0, end_bit): the library's own kernels (csrc/dmt_sort.hip: 8-bit digits, per pass a
 * tile histogram, a one-workgroup scan and a ranked scatter through LDS), no vendor sort.  Equal keys keep their input order -- the order
 * in which the reference's IndexedSlices of one variable are summed is the entry order (run_dnn.py:203-207 densifies them per variable).
 * vals_in == NULL: values are the entry numbers 0 .. n-1.  Inputs are left untouched; outputs must not alias them.
 * ws_bytes: in/out workspace size; call with ws == NULL to query (16-byte aligned workspace; n < 2^31 - 4096).                          */
int dmt_sort_pairs(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                   int64_t n, int32_t end_bit, void* ws, uint64_t* ws_bytes, void* stream);

/* From sorted keys: seg_id[e] = rank of key[e] among distinct keys (inclusive scan of head flags - 1);
 * uniq_keys[seg] = key; n_uniq[0] = number of distinct keys < invalid_key.  sorted_keys, seg_id and ws 16-byte aligned.
 * Three launches of the library's own (count run heads per 4096-entry tile, scan the tile counts, write).                */
int dmt_segment_heads(const uint32_t* sorted_keys, int64_t n, uint32_t invalid_key, int32_t* seg_id,
                      uint32_t* uniq_keys, int32_t* n_uniq, void* ws, uint64_t* ws_bytes, void* stream);

/* grad_rows[seg, 0:dim_of(table)] += contribution of every entry (row stride = max_dim, fp32).
 * grad_rows must be zeroed by the caller.                                                            */
/* Row-sharded tables: the row-cache slot of every entry after dmt_sort_pairs + dmt_segment_heads (keys_sorted, vals_sorted, seg):
 * slots[entry] = slot_of_row[seg] (or seg when slot_of_row is NULL) for a pooled entry, + 1 for a sequence entry (0 = zero row / padding).
 * The slices slots[entry_base[f] ...] are the idx / idx_seq columns of dmt_gather_feature in its row-cache form.                  */
int dmt_entry_slots(const dmt_embgrad_desc* d, const uint32_t* keys_sorted, const uint32_t* vals_sorted, const int32_t* seg,
                    const int32_t* slot_of_row, int64_t n, int32_t* slots, void* stream);
/* Ordered (bit-reproducible) reductions are chosen PER CALL; the library keeps no mode.  By default partial sums that meet in one
 * output element are combined with fp32 atomics -- their order, hence the last bits, depends on scheduling.  The segmented reductions
 * (dmt_embgrad_reduce, dmt_rows_reduce*) take the ordered form when the caller passes a workspace (det_ws != NULL, of
 * dmt_reduce_det_ws_bytes(n, max_dim) bytes, 16-byte aligned): runs that cross a 64-entry chunk go through the workspace and the
 * pieces are added in chunk order.  dmt_colsum / dmt_colsum_drop take `ordered` (one pass per column, in row order).  A caller that
 * wants a bit-reproducible step also keeps to split_k = 1 in dmt_gemm and away from dmt_wgrad320 / dmt_heads_bwd (fp32 atomics by
 * construction) -- cikm2020_dmt_amd/ops.py does so under ops.set_deterministic (tests/test_gpu_deterministic.py).  Results equal the
 * default forms' to fp32 rounding.  The reference's counterpart is TF's own non-deterministic unsorted_segment_sum / atomics on GPU;
 * on CPU (the reference run: run_dnn.py:45-80) the order is fixed. */
uint64_t dmt_reduce_det_ws_bytes(int64_t n, int32_t max_dim);

int dmt_embgrad_reduce(const dmt_embgrad_desc* d, const uint32_t* sorted_keys, const uint32_t* sorted_vals,
                       const int32_t* seg_id, int64_t n, float* grad_rows, int32_t max_dim, void* det_ws, uint64_t det_ws_bytes,
                       void* stream);

/* rows[0 : min(n_rows[0] + extra, max_rows), 0 : row_elems] = 0 with the row count read ON THE DEVICE (the number of distinct rows
 * of a step is only known there): clears what the reduce kernels will accumulate into instead of the whole capacity. */
int dmt_zero_rows(float* rows, const int32_t* n_rows, int64_t extra, int64_t max_rows, int32_t row_elems, void* stream);

/* Same reduction for already-materialised rows (data-parallel merge of per-rank sparse gradients):
 * out_rows[seg_id[e]] += in_rows[sorted_vals[e]].                                                    */
int dmt_rows_reduce(const uint32_t* sorted_keys, const uint32_t* sorted_vals, const int32_t* seg_id, int64_t n,
                    uint32_t invalid_key, const float* in_rows, float* out_rows, int32_t max_dim, void* det_ws, uint64_t det_ws_bytes,
                    void* stream);
/* The same with bf16 in_rows (the transport format of the data-parallel exchange in bf16 mode; the sum stays fp32). */
int dmt_rows_reduce_bf16(const uint32_t* sorted_keys, const uint32_t* sorted_vals, const int32_t* seg_id, int64_t n,
                         uint32_t invalid_key, const void* in_rows_bf16, float* out_rows, int32_t max_dim, void* det_ws,
                         uint64_t det_ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[m,n] = epi( sum_k A(m,k) * B(k,n) ),  A(m,k) = A[m*a_rs + k*a_cs],
 * B(k,n) = B[k*b_rs + n*b_cs].  bf16 operands use v_mfma_f32_32x32x16_bf16, fp32 operands
 * v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
 *   epi(x) = ((x + bias[n]) relu-if(n < act_ncols)) * (gate[m,n] > 0 if gate) + resid[m,n]
 * Replaces: tf.layers.dense / tf.matmul(+b) call sites -- TransformerModel_util.py:188-190 (QKV),
 *           :224-227 (FFN), base.py:39-68 dense_layer (experts, gates, towers),
 *           mmoe_transformer_unbias.py:266-287 (bias MLP) -- and their autodiff gradients
 *           (dX = dY W^T, dW = X^T dY, db = colsum dY via `a_ones_row`).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t in_dtype;            /* dtype of A, B, gate                                               */
  int32_t out_dtype;           /* dtype of C, resid                                                 */
  int32_t M, N, K;
  const void* A; int64_t a_rs, a_cs;
  const void* B; int64_t b_rs, b_cs;
  void* C; int64_t ldc;
  const float* bias;           /* [N] or NULL                                                       */
  int32_t act_ncols;           /* relu applied to columns n < act_ncols (0: none)                   */
  const void* gate; int64_t ldg;     /* relu-gradient mask source or NULL                           */
  const void* resid; int64_t ldr;    /* residual or NULL                                            */
  int32_t a_ones_row;          /* 1: logical row M-1 of A is all ones; its output row goes to c_last */
  float* c_last;               /* fp32 [N] (bias gradient) when a_ones_row                          */
  int32_t split_k;             /* >1: K split over workgroups, fp32 atomicAdd into C (C must be fp32, zeroed) */
  int32_t accumulate;          /* 1: C (fp32) and c_last are accumulated into (atomicAdd) even without split_k  */
  int32_t batch;               /* >=1 */
  int64_t a_bs, b_bs, c_bs, bias_bs, gate_bs, resid_bs, clast_bs;   /* per-batch element strides    */
} dmt_gemm_desc;

int dmt_gemm(const dmt_gemm_desc* d, void* stream);
/* revision 5: n GEMMs of the weight-gradient class -- C[M, N] (fp32) (+)= A^T B with bf16 operands whose ROW index is the reduction
 * dimension (a_cs != 1, b_rs != 1 in the descriptor's terms: dW = X^T dY), K % 64 == 0, K / split_k <= 4096, no epilogue -- in ONE launch
 * (the decoders' B-row weight gradients: six 15 us launches per sequence, none of which fills the chip).  All or nothing:
 * DMT_ERR_UNSUPPORTED when a descriptor is of another class (nothing has been launched then).  Jobs that write the same elements of C must
 * all have `accumulate` set.                                                                                                            */
int dmt_gemm_dw_batched(const dmt_gemm_desc* descs, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Short-sequence multi-head attention core (one wavefront per (example, head)).
 *   out[b,q,:] = concat_h softmax(mask(Q_h K_h^T / sqrt(dh))) V_h + resid[b,q,:]
 * with the reference's masking: invalid keys <- -2^32+1 before softmax; rows of invalid queries
 * <- -2^32+1 AFTER softmax (TransformerModel_util.py:11-56, 80-108).  No output projection (F8).
 * Replaces: scaled_dot_product_attention + head split/concat + residual of multihead_attention
 *           (TransformerModel_util.py:160-205).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t dtype;               /* Q, K, V, resid, out (and dout, dq, dk, dv) */
  int32_t B, H, dh, Tq, Tk;
  const void* Q; int64_t q_bs, q_rs;       /* Q[b*q_bs + t*q_rs + h*dh + j] */
  const void* K; int64_t k_bs, k_rs;
  const void* V; int64_t v_bs, v_rs;
  const int32_t* q_lens;       /* [B] or NULL (all queries valid) */
  const int32_t* k_lens;       /* [B] */
  const void* resid; int64_t r_bs, r_rs;   /* or NULL */
  void* out; int64_t o_bs, o_rs;
  uint32_t drop_seed;          /* dropout on the attention weights (TransformerModel_util.py:51), applied after the query mask: */
  float drop_keep;             /* keep probability; >= 1 (or 0) disables.  mask(i) = dmt_dropout_keep(drop_seed, i, keep),      */
                               /* i = ((b*H + h)*Tq + q)*Tk + k                                                                  */
  int32_t mma_dtype;           /* 0: the MFMAs run in `dtype`.  DMT_FP8_E4M3: long-sequence forward (64 < T <= 256, Tq > 1) converts Q, K, V and   */
                               /* the weights to OCP e4m3 for its two matrix products (BASELINE configs[4]); ignored elsewhere                 */
  /* revision 5, packed rows (dmt_attn_bwd, bf16, self-attention form Tq == Tk <= 64 only): row_off != NULL: example b's operand rows are
   * rows row_off[b] + t of Q / K / V / dout / dQ / dK / dV (the *_bs strides are ignored), and it has k_lens[b] rows (queries and keys).
   * ex_list != NULL: the launch covers the n_list examples ex_list[0 .. n_list - 1] instead of 0 .. B - 1 (the engine launches one
   * length class at a time, so that short examples take the one-tile kernel: max_len > 0 promises that no covered example is longer).
   * Tq / Tk stay the DENSE lengths (dropout index).                                                                                    */
  const int32_t* row_off;
  const int32_t* ex_list;
  int32_t n_list;
  int32_t max_len;
} dmt_attn_desc;

int dmt_attn_fwd(const dmt_attn_desc* d, void* stream);

/* Backward: given dout (grad wrt out, same layout as out) produce dQ, dK, dV (layouts as Q, K, V;
 * every element of the (b, t<T, h, j) range is written).  The residual gradient is dout itself.      */
typedef struct {
  dmt_attn_desc f;
  const void* dout; int64_t do_bs, do_rs;
  void* dQ; int64_t dq_bs, dq_rs;
  void* dK; int64_t dk_bs, dk_rs;
  void* dV; int64_t dv_bs, dv_rs;
} dmt_attn_bwd_desc;

int dmt_attn_bwd(const dmt_attn_bwd_desc* d, void* stream);

/* Long sequences (64 < max(Tq, Tk) <= 256; BASELINE "long-seq variant": clk / ord histories of 200): same operation, same
 * descriptors, flash-style workgroup kernels (dmt_attn_long.hip) -- no [B,H,T,T] tensor reaches HBM.  dmt_attn_fwd / dmt_attn_bwd
 * route here by themselves when a sequence is longer than 64; the entry points are exported for direct use and tests.
 * Requirements: bf16, dh in {16, 32, 64, 80}, every operand row 16-byte aligned (base % 16 == 0, strides % 8 == 0).
 * Replaces: scaled_dot_product_attention at T = 200 (TransformerModel_util.py:11-56) incl. its three [N*h, T, T] intermediates. */
int dmt_attn_long_supported(int32_t dtype, int32_t dh, int32_t Tq, int32_t Tk);
int dmt_attn_long_fwd(const dmt_attn_desc* d, void* stream);
int dmt_attn_long_bwd(const dmt_attn_bwd_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the last dim, biased variance, eps inside the sqrt (TransformerModel_util.py:58-78).
 *   y = gamma * (x - mean) / sqrt(var + eps) + beta ; stats[r] = {mean, rstd}
 * ------------------------------------------------------------------------------------------------ */
int dmt_ln_fwd(int32_t dtype, int64_t rows, int32_t d, const void* x, int64_t ldx, const float* gamma,
               const float* beta, float eps, void* y, int64_t ldy, float* stats, void* stream);
/* dx written; dgamma/dbeta (fp32 [d]) are ACCUMULATED with atomics-free two-stage reduction through
 * `partials` (fp32 [n_part, 2*d], n_part = dmt_ln_bwd_partials(rows)).                               */
int32_t dmt_ln_bwd_partials(int64_t rows);
int dmt_ln_bwd(int32_t dtype, int64_t rows, int32_t d, const void* x, int64_t ldx, const float* gamma,
               const float* stats, const void* dy, int64_t lddy, void* dx, int64_t lddx, float* dgamma,
               float* dbeta, float* partials, void* stream);
/* dgamma == dbeta == NULL: only dx and `partials` are written; the caller finishes a list of such gradients in ONE launch
 * (a train step has twelve): job i adds the column sums of partials_i [n_part_i, 2 d_i] to dgamma_i / dbeta_i, in a fixed order; jobs
 * that name the same dgamma (shared LayerNorm parameters) are summed by the same wavefront, one after the other.                   */
#define DMT_LN_FINISH_MAX 16
typedef struct { const float* partials; float* dgamma; float* dbeta; int32_t n_part; int32_t d; } dmt_ln_finish_job;
int dmt_ln_bwd_finish_batched(const dmt_ln_finish_job* jobs, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MMoE mixture: gate softmax + weighted sum of expert outputs and its gradient
 * (mmoe_transformer_unbias.py:80-105: dense_layer(..., tf.nn.softmax), stack/tile/reduce_sum).
 *   gates[t][b,:] = softmax(glogit[b, t*E : (t+1)*E]);  mix[t][b,:] = sum_e gates[t][b,e] * expert[b, e*U : (e+1)*U]
 * ------------------------------------------------------------------------------------------------ */
int dmt_mmoe_mix_fwd(int32_t dtype, int32_t B, int32_t E, int32_t U, int32_t n_tasks, const void* expert,
                     int64_t ld_expert, const void* glogit, int64_t ld_glogit, float* gates /* [n_tasks,B,E] */,
                     void* mix /* [n_tasks, B, U] */, void* stream);
int dmt_mmoe_mix_bwd(int32_t dtype, int32_t B, int32_t E, int32_t U, int32_t n_tasks, const void* expert,
                     int64_t ld_expert, const float* gates, const void* dmix, void* dexpert, int64_t ld_dexpert,
                     void* dglogit, int64_t ld_dglogit, int32_t relu_mask /* 1: expert is a relu output, return the
                     gradient wrt its pre-activation */, void* stream);

/* y[b,t,:] = scale * x[b,t,:] + pos[t,:]: the input prep of TransformerModel.encode / decode
 * (model/net/TransformerModel.py:96-100,146-147) when the sequence embedding was gathered unscaled. pos may be NULL. */
int dmt_scale_add_pos(int32_t dtype, int64_t B, int32_t T, int32_t d, const void* x, float scale, const float* pos, void* y,
                      void* stream);

/* tf.layers.dropout (TransformerModel.py:101,151; mmoe_transformer_unbias.py:274-278): y[i] = x[i] * keep(i) / keep_prob with the
 * counter-based mask  keep(i) = (mix32(i ^ seed) >> 8) < keep_prob * 2^24,  mix32 = murmur3 finaliser.  The same call is its own
 * gradient (dx = dropout(dy)).  Deterministic in (seed, i): the oracle reproduces the mask (oracle/dmt_oracle.py:dropout_mask). */
int dmt_dropout(int32_t dtype, int64_t n, const void* x, void* y, uint32_t seed, float keep_prob, void* stream);

/* Gradient of relu given its OUTPUT y: dz = dy * (y > 0)  (tf.nn.relu at base.py:64, TransformerModel_util.py:224). */
int dmt_relu_bwd(int32_t dtype, int64_t rows, int64_t cols, const void* dy, int64_t lddy, const void* y, int64_t ldy,
                 void* dz, int64_t lddz, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Loss of the unbias model (model/inference_mlp.py:162-223) and its gradient in one pass.
 *   method: 0 two_head_add, 1 two_head_multiply, 2 plain logit_loss (sigmoid_cross_entropy_with_logits on the two
 *   logits, model/inference_mlp.py:228-258; ybias ignored);  ctr_rel: 1 adds the relevance-only cross entropies.
 *   loss[0] = scalar loss (batch mean, class weights w_ctr/w_ecvr[5], loss weights lw[2]);
 *   p_ctr, p_cvr [B] as run_dnn.py:90-101; d_click/d_order/d_bias [B] = dloss/dlogit * grad_scale.
 *   logits are fp32 [B] (the towers' final 1-wide layer is produced in fp32).
 * ------------------------------------------------------------------------------------------------ */
int dmt_loss_unbias(int32_t B, const float* click, const float* order, const float* ybias, const float* mask5,
                    const float* w_ctr, const float* w_ecvr, float lw_clk, float lw_ord, int32_t method,
                    int32_t ctr_rel, float grad_scale, float* loss, float* p_ctr, float* p_cvr, float* d_click,
                    float* d_order, float* d_bias, void* stream);

/* ------------------------------------------------------------------------------------------------
 * tf.train.AdamOptimizer arithmetic (model/inference_mlp.py:264-273 -> TF ApplyAdam), see
 * oracle/dmt_oracle.py:TFAdam.  `state` (device, fp32[4]) = {beta1_power, beta2_power, lr_t, step};
 * dmt_adam_begin_step computes lr_t for this step and appends it to lr_hist[step] (for lazy rows),
 * dmt_adam_end_step advances the powers.  Hyper-parameters are passed by value.
 * ------------------------------------------------------------------------------------------------ */
int dmt_adam_begin_step(float* state, float* lr_hist, int32_t lr_hist_cap, float lr, float beta1, float beta2,
                        void* stream);
int dmt_adam_end_step(float* state, float beta1, float beta2, void* stream);
/* Dense: p, m, v, g fp32 [n];  optional bf16 shadow written alongside (lp may be NULL).               */
int dmt_adam_dense(int64_t n, float* p, float* m, float* v, const float* g, float grad_scale, const float* state,
                   float beta1, float beta2, float eps, void* lp_bf16, void* stream);
/* Sparse rows with exact lazy catch-up: for every u < n_uniq[0]: row = uniq_keys[u] (global id);
 * replay the zero-gradient updates of steps last_step[row]+1 .. step-1 (bit-identical to the dense
 * sweep), then apply step `step` with gradient grad_rows[u].  Tables are described by
 * (row_base, dim, element offset into the p/m/v arenas).                                             */
typedef struct {
  int32_t n_tables;
  int32_t row_base[DMT_MAX_TABLES + 1];   /* ascending; row_base[n_tables] = total_rows              */
  int32_t dim[DMT_MAX_TABLES];
  int64_t elem_off[DMT_MAX_TABLES];       /* offset of the table inside p / m / v                    */
  /* Row-sharded tables (BASELINE configs[3]): shard_w > 1 -> this rank's p / m / v / last_step hold only the rows with
   * global id % shard_w == shard_r, densely (local row (row - row_base[t]) / shard_w; every row_base is a multiple of shard_w;
   * last_step index row / shard_w).  Keys this rank does not own are skipped.  0 / 1: replicated layout.                      */
  int32_t shard_w, shard_r;
} dmt_table_map;
int dmt_adam_sparse_rows(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step,
                         const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq, const float* grad_rows,
                         int32_t max_dim, float grad_scale, const float* state, const float* lr_hist, float beta1,
                         float beta2, float eps, void* stream);
/* The same with bf16 gradient rows (the data-parallel wire format of the reduced rows).  Both variants skip keys
 * >= the total row count: padding slots of a rank-major gathered list.                                          */
int dmt_adam_sparse_rows_bf16(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step,
                              const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq,
                              const void* grad_rows_bf16, int32_t max_dim, float grad_scale, const float* state,
                              const float* lr_hist, float beta1, float beta2, float eps, void* stream);
/* Replay the pending zero-gradient steps of the listed rows up to the last COMPLETED step (state.step): must run
 * before the forward pass gathers those rows, so the values read are the ones the dense sweep would have produced. */
int dmt_adam_catchup_rows(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step,
                          const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq, const float* state,
                          const float* lr_hist, float beta1, float beta2, float eps, void* stream);
/* The same through an explicit step `to_step` (>= 0), skipping rows whose stamp[row] == stamp_skip (stamp may be NULL).  Used for the
 * EARLY catch-up of the NEXT batch while an optimizer step is in flight: a zero-gradient update of step t needs only (p, m, v) and
 * lr_t, which dmt_adam_begin_step has already written to lr_hist[t]; rows the step in flight updates itself (stamped by
 * dmt_rows_stamp with that step's number) are left to it.  The dense-sweep result is unchanged (the same updates, applied earlier). */
int dmt_adam_catchup_rows_to(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step,
                             const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq, const float* state,
                             const float* lr_hist, float beta1, float beta2, float eps, int32_t to_step, const int32_t* stamp,
                             int32_t stamp_skip, void* stream);
/* stamp[row] = value for the listed global rows (indexing as last_step: row / shard_w); keys >= the total row count are skipped. */
int dmt_rows_stamp(const dmt_table_map* tm, const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq, int32_t* stamp,
                   int32_t value, void* stream);
/* Bring every row of every table up to date (before checkpoint / evaluation of untouched rows).      */
int dmt_adam_flush_rows(const dmt_table_map* tm, float* p, float* m, float* v, int32_t* last_step,
                        const float* state, const float* lr_hist, float beta1, float beta2, float eps, void* stream);
/* After dmt_adam_flush_rows: restart the device-side step counter that indexes lr_hist (state[3] = 0, last_step[:] = 0), so a run
 * longer than the history's capacity -- or one resumed from model.ckpt-<large N> -- keeps going.  Values are untouched. */
int dmt_adam_rebase(float* state, int32_t* last_step, int64_t rows, void* stream);
/* ------------------------------------------------------------------------------------------------
 * The other optimizers get_optimizer returns (model/inference_mlp.py:264-280: sgd, adadelta, adagrad, ftrl, rmsprop), each with the
 * constructor defaults of TF 1.12 (the reference passes the learning rate only); see oracle/dmt_oracle.py:TFOptimizer.
 *   kind      s0                     s1                   h0           h1            h2
 *   SGD       -                      -                    -            -             -
 *   ADAGRAD   accumulator (init 0.1) -                    -            -             -
 *   ADADELTA  accum                  accum_update         rho 0.95     epsilon 1e-8  -
 *   RMSPROP   rms (init 1.0)         momentum (== 0.0)    decay 0.9    momentum 0.0  epsilon 1e-10
 *   FTRL      accum (init 0.1)       linear               l1 0.0       l2 0.0        -          (learning_rate_power -0.5)
 * s0 / s1 always point at storage (the Adam slot arenas are reused); lr is this step's learning rate, `step` the 1-based local step
 * number that indexes last_step (0 = never updated).  Sparse rows: the zero-gradient steps last_step[row]+1 .. step-1 are replayed on
 * the slots (they never move var, so nothing has to run in front of the gather), then step `step` is applied.  dmt_opt_flush_rows
 * brings every row's slots to `step`; for FTRL it recomputes var from (accum, linear) and `lr` -- what the reference's dense ApplyFtrl
 * does to every element at every step: zero for rows never updated (linear == 0), a rescaled value when the schedule has changed the
 * learning rate, the value already held otherwise: call it after the first step and after every step whose lr differs from the last.
 * ------------------------------------------------------------------------------------------------ */
enum { DMT_OPT_SGD = 1, DMT_OPT_ADAGRAD = 2, DMT_OPT_ADADELTA = 3, DMT_OPT_RMSPROP = 4, DMT_OPT_FTRL = 5 };
int dmt_opt_dense(int32_t kind, int64_t n, float* p, float* s0, float* s1, const float* g, float grad_scale, float lr, float h0,
                  float h1, float h2, void* lp_bf16, void* stream);
int dmt_opt_sparse_rows(int32_t kind, const dmt_table_map* tm, float* p, float* s0, float* s1, int32_t* last_step,
                        const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq, const void* grad_rows,
                        int32_t grad_is_bf16, int32_t max_dim, float grad_scale, int32_t step, float lr, float h0, float h1,
                        float h2, void* stream);
int dmt_opt_flush_rows(int32_t kind, const dmt_table_map* tm, float* p, float* s0, float* s1, int32_t* last_step, int32_t step,
                       float lr, float h0, float h1, float h2, void* stream);
/* Row-sharded tables, owner side of the forward exchange: out[u, :] = p[row keys[u]] for u < n (fp32, row stride max_dim, columns
 * past the table's dim zeroed; keys this rank does not own give zero rows).  The all-to-all that answers the index exchange of
 * BASELINE configs[3] sends these rows back to the ranks that asked for them (replaces the /cpu:0 embedding_lookup of base.py:81-91
 * for a table no single device holds).                                                                                          */
int dmt_rows_gather(const dmt_table_map* tm, const float* p, const uint32_t* keys, int64_t n, float* out, int32_t max_dim, void* stream);

/* fp32 -> bf16 shadow copies of 2-D weights, plain and transposed: dst[n*ld_t + k] = src[k*ld + n].    */
int dmt_cast_bf16(int64_t n, const float* src, void* dst, void* stream);
int dmt_cast_transpose_bf16(int32_t rows, int32_t cols, const float* src, int64_t ld_src, void* dst_plain,
                            int64_t ld_plain, void* dst_t, int64_t ld_t, void* stream);

/* The same for every 2-D weight of the model in one launch (after each optimizer step): jobs_dev is a device array;
 * job j owns the 32x32 tiles [tile_begin, tile_begin + tiles_x*tiles_y), total_tiles = sum over jobs.          */
typedef struct dmt_cast_job {
  const float* src;
  void* dst_plain;   /* may be NULL */
  void* dst_t;       /* may be NULL */
  int64_t ld_src, ld_plain, ld_t;
  int32_t rows, cols, tile_begin, tiles_x;
} dmt_cast_job;
int dmt_cast_transpose_bf16_batched(int32_t n_jobs, const dmt_cast_job* jobs_dev, int32_t total_tiles, void* stream);

/* Row softmax of the UNFUSED attention used for sequences longer than 64 keys (dmt_attn_fwd/bwd take T <= 64): the scores
 * S = Q K^T and the products around it run as batched dmt_gemm launches; these do what lies between them
 * (model/net/TransformerModel_util.py:11-56 scaled_dot_product_attention, :80-108 mask).  S [B, H, Tq, ld] holds the raw
 * scores on entry and the dropped weights (A operand of P.V) on return; P receives the weights before dropout (saved for
 * the backward).  Same masking rules, padding value and dropout counter as dmt_attn_fwd.  causal != 0: future blinding
 * (mask(type="future"), :34-36, 99-105): key k of query q also gets the padding value when k > q.                      */
int dmt_softmax_fwd(int32_t dtype, int32_t B, int32_t H, int32_t Tq, int32_t Tk, void* S, int64_t ld, const int32_t* q_lens,
                    const int32_t* k_lens, float scale, uint32_t drop_seed, float drop_keep, void* P, int32_t causal, void* stream);
/* dP_dS: gradient w.r.t. the dropped weights on entry, dS (gradient of the scaled scores' pre-image Q K^T) on return;
 * Pd receives the dropped weights again (the operand of dV = Pd^T dO).                                                */
int dmt_softmax_bwd(int32_t dtype, int32_t B, int32_t H, int32_t Tq, int32_t Tk, const void* P, void* dP_dS, void* Pd,
                    int64_t ld, const int32_t* q_lens, const int32_t* k_lens, float scale, uint32_t drop_seed,
                    float drop_keep, int32_t causal, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused position-wise feed-forward + LayerNorm (bf16): two chained MFMA GEMMs whose d_ff-wide intermediate stays on the
 * compute unit (registers), input rows resident in registers, weights streamed through LDS from a prebuilt image.
 * Replaces: ff(inputs, [d_ff, d_model]) -- dense(relu) -> dense -> += inputs -> ln()
 *           (model/net/TransformerModel_util.py:212-235 with ln() :58-78) as ONE launch, and its input gradient.
 *   mode DMT_CHAIN_FFN_LN : s = relu(in A1^T + bias1) A2^T + bias2 + in;  y = gamma (s - mean) / sqrt(var + eps) + beta
 *        A1[j, k] = W1[k, j]  ("dense/kernel"   [d_model, d_ff]),  A2[n, j] = W2[j, n]  ("dense_1/kernel" [d_ff, d_model])
 *        outputs: y_out, s_out (pre-LN sum, or NULL), stats [M, 2] = (mean, 1/sqrt(var + eps)) (or NULL),
 *                 mid_out = h [M, nmid] (or NULL: inference), mask = relu gate bits (or NULL)
 *   mode DMT_CHAIN_FFN_BWD: dh = (in A1^T) * gate;  dx = dh A2^T + in        (in = ds, the gradient w.r.t. s)
 *        A1[j, n] = W2[j, n],  A2[k, j] = W1[k, j];  outputs: s_out = dx, mid_out = dh (for the weight gradients), mask = input
 * Images are built by dmt_chain_image_build from the fp32 masters (after every optimizer step); geometry (kin, nmid, nout) must
 * be one dmt_chain_supported() accepts.  mask: uint16 [4 * ceil(M / 128)][nmid / 32][64] (one bit per element of mid, in the
 * kernel's own lane order: only dmt_chain2 reads it back).
 * ------------------------------------------------------------------------------------------------ */
#define DMT_CHAIN_FFN_LN 0
#define DMT_CHAIN_FFN_BWD 1

typedef struct {
  int32_t mode;
  int32_t kin, nmid, nout;
  int64_t M;
  const void* in;        /* bf16 [M, kin], row stride ld_in (multiple of 8), 16-byte aligned */
  int64_t ld_in;
  const void* image;     /* dmt_chain_image_build output for this mode                       */
  const float* bias2;    /* fp32 [nout]   (FFN_LN)                                           */
  const float* gamma;    /* fp32 [nout]   (FFN_LN)                                           */
  const float* beta;     /* fp32 [nout]   (FFN_LN)                                           */
  float eps;             /* 1e-8 in the reference (TransformerModel_util.py:58)              */
  void* s_out;           /* bf16 [M, nout] pre-LN sum (FFN_LN, optional) / dx (FFN_BWD)      */
  void* y_out;           /* bf16 [M, nout] (FFN_LN)                                          */
  int64_t ld_out;
  float* stats;          /* fp32 [M, 2] or NULL                                              */
  void* mid_out;         /* bf16 [M, nmid] or NULL                                           */
  int64_t ld_mid;
  void* mask;            /* see above                                                        */
} dmt_chain_desc;

int dmt_chain_supported(int32_t kin, int32_t nmid, int32_t nout);
int dmt_chain_image_bytes(int32_t kin, int32_t nmid, int32_t nout, int64_t* bytes);
/* A1[j, k] = a1[j * a1_rs + k * a1_cs] (j < nmid, k < kin), A2[n, j] = a2[n * a2_rs + j * a2_cs] (n < nout, j < nmid), bias1 [nmid] or NULL */
int dmt_chain_image_build(int32_t kin, int32_t nmid, int32_t nout, const float* a1, int64_t a1_rs, int64_t a1_cs, const float* a2,
                          int64_t a2_rs, int64_t a2_cs, const float* bias1, void* image, void* stream);
int dmt_chain2(const dmt_chain_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense projection of long-row activations with the weights streamed as an LDS image (bf16 in / out, fp32 accumulation):
 *     out[m, :] = in[m, :] W + b        in [M, kin], W [kin, n], b [n] or NULL (then 0)
 * The input rows of a wavefront stay in registers, W passes through LDS once per 128 rows (GEMM 1 of dmt_chain2 alone).
 * Replaces: tf.layers.dense producing the packed Q | K | V of the self-attention block,
 *           model/net/TransformerModel_util.py:188-190 (kin = d_model = 320, n = 3 d_model = 960).
 * dmt_proj_image_build takes W[k, j] = w[k * w_rs + j * w_cs] (fp32) and must be re-run after every change of W or b.
 * ------------------------------------------------------------------------------------------------ */
int dmt_proj_supported(int32_t kin, int32_t n);
int dmt_proj_image_bytes(int32_t kin, int32_t n, int64_t* bytes);
int dmt_proj_image_build(int32_t kin, int32_t n, const float* w, int64_t w_rs, int64_t w_cs, const float* bias, void* image, void* stream);
int dmt_proj(int32_t kin, int32_t n, int64_t M, const void* in, int64_t ld_in, const void* image, void* out, int64_t ld_out, void* stream);
/* All weight images of a model rebuilt in ONE launch (after every optimizer step): a table of jobs in DEVICE memory, each
 * dmt_image_job_bytes() long, filled on the host by dmt_chain_image_job / dmt_proj_image_job (same arguments as the builders above,
 * job_out = host pointer to one slot) and uploaded once; the pointers in a job must stay valid. */
int32_t dmt_image_job_bytes(void);
int dmt_chain_image_job(int32_t kin, int32_t nmid, int32_t nout, const float* a1, int64_t a1_rs, int64_t a1_cs, const float* a2,
                        int64_t a2_rs, int64_t a2_cs, const float* bias1, void* image, void* job_out);
int dmt_proj_image_job(int32_t kin, int32_t n, const float* w, int64_t w_rs, int64_t w_cs, const float* bias, void* image, void* job_out);
int dmt_image_build_batched(int32_t n_jobs, const void* jobs_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weight gradient of a dense layer whose input OR output is 320 (= d_model of the E64 configuration) wide, as one wide-block
 * reduction over the batch x sequence rows (bf16 operands, fp32 ACCUMULATED into C by atomics -- C is the gradient arena):
 *     transposed == 0:  C[i * ldc + j] += sum_m A[m, i] B[m, j]      (dW = X^T dY with X 320 wide:  A = X, B = dY)
 *     transposed == 1:  C[j * ldc + i] += sum_m A[m, i] B[m, j]      (dW = X^T dY with dY 320 wide: A = dY, B = X)
 *     bias_of == 1: bias[j] += sum_m B[m, j];   bias_of == 2: bias[i] += sum_m A[m, i]   (db = column sums of dY)
 * Ordered form (bit-reproducible): with det_ws != NULL (dmt_wgrad320_det_ws_bytes(M, N) bytes, 16-byte aligned) every row split
 * writes its partial [320, N] block (and its bias partials) to the workspace instead of adding it to C with atomics, and a second
 * launch adds the splits in split order onto C / bias.  Launches that target the same C must be ordered by the caller (one stream).
 * Replaces: the kernel / bias gradients of tf.layers.dense at model/net/TransformerModel_util.py:188-190 (Q, K, V) and
 *           :224-228 (position-wise feed-forward), i.e. what dmt_gemm computes with a_ones_row for any shape.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  const void* A; int64_t ld_a; int32_t a_cols;   /* bf16 [M, 320]                                   */
  const void* B; int64_t ld_b;                   /* bf16 [M, N]                                     */
  int64_t M;
  int32_t N;                                     /* multiple of 8                                   */
  float* C; int64_t ldc;
  int32_t transposed;
  float* bias;
  int32_t bias_of;
  void* det_ws; uint64_t det_ws_bytes;           /* NULL / 0: fp32 atomics (default)                  */
} dmt_wgrad_desc;
int dmt_wgrad320(const dmt_wgrad_desc* d, void* stream);
uint64_t dmt_wgrad320_det_ws_bytes(int64_t M, int32_t N);

/* ------------------------------------------------------------------------------------------------
 * Self-attention block of the sequence encoder in ONE launch (bf16; built for d_model 320 = 4 heads x 80, T <= 64):
 *     y = ln(x + concat_h softmax(mask(Q_h K_h^T / sqrt(d_h))) V_h),   (Q | K | V) = x Wqkv + bias
 * QKV projection, key mask, softmax, query mask, attention dropout, P V, head concat, residual and LayerNorm without any
 * intermediate travelling back from memory.
 * Replaces: multihead_attention(queries, keys, values, ...) with queries == keys == values (the encoder's self-attention,
 *           model/net/TransformerModel.py:103-115 -> TransformerModel_util.py:160-209, 11-56, 80-108, 58-78).
 * image: dmt_mhsa_image_build(Wqkv fp32 [320, 960] packed dense | dense_1 | dense_2 kernels); dmt_mhsa_image_bytes() bytes; rebuilt after
 *        every optimizer step (the layout is private to the library: 64 stages of 32 head-major columns x 160 k).
 * Side outputs for the backward pass: qkv [B*T, 960] (or NULL: inference), s_out = pre-LN sum (or NULL: inference; the sum then passes
 * through y_out), stats [B*T, 2] = (mean, rstd) (or NULL).  Limits: B * T * 1920 < 2^31 (32-bit byte offsets).
 * One workgroup of 8 wavefronts per CU, persistent over tiles of 256 rows.
 * drop_keep in (0, 1): attention-weight dropout with the library's counter mask, element index ((b*H + h)*T + q)*T + k.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t d_model, num_heads;
  int32_t B, T;
  const void* x;          /* bf16 [B, T, d_model] contiguous            */
  const int32_t* lens;    /* [B] valid length (key AND query mask)      */
  const void* image;
  const float* bias;      /* fp32 [3 * d_model]                          */
  const float* gamma;     /* fp32 [d_model]                              */
  const float* beta;
  float eps;
  void* qkv;              /* bf16 [B*T, 3 * d_model] or NULL             */
  void* s_out;            /* bf16 [B*T, d_model] or NULL                 */
  void* y_out;            /* bf16 [B*T, d_model]                         */
  float* stats;           /* fp32 [B*T, 2] or NULL                       */
  uint32_t drop_seed;
  float drop_keep;
  /* revision 5, packed rows: blocks != NULL: x / qkv / s_out / y_out / stats are [R, .] (R = n_rows) and the workgroups walk n_tiles row
   * tiles of 8 blocks of 32 local rows each, described by blocks[tile][8][2] (int32 x 4 per entry: example, its length, its first packed
   * row, log2 of the tile's padded example length Tp in {16, 32, 64}; example = -1: nothing there).  A block holds 32 / Tp examples
   * (entry [.][.][1] is the second one of a Tp = 16 block), an example of Tp = 64 spans two blocks.  Built by the caller from the
   * lengths (cikm2020_dmt_amd/engine.py SeqPack); B and T stay the dense dimensions (dropout index).                                   */
  const int32_t* blocks;
  int32_t n_tiles;
  int64_t n_rows;
} dmt_mhsa_desc;
int dmt_mhsa_image_bytes(int64_t* bytes);
int dmt_mhsa_image_build(const float* wqkv, int64_t ldw, void* image, void* stream);
int dmt_mhsa_block_fwd(const dmt_mhsa_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * revision 6.  Backward of the self-attention block in ONE launch (bf16; d_model 320 = 4 heads x 80, T <= 64), from the gradient ds of
 * the pre-LayerNorm sum (what dmt_ln_bwd hands back for s_out) and the forward's saved qkv:
 *     dqkv = gradient of concat_h softmax(mask(Q_h K_h^T / sqrt(d_h))) V_h   w.r.t. (Q | K | V)      [rows, 960]
 *     dx   = dqkv Wqkv^T + ds                                                                         [rows, 320]
 * Replaces: the gradient TF autodiff builds for multihead_attention(queries == keys == values) up to its three dense layers' inputs
 *           (TransformerModel_util.py:160-209, 11-56, 80-108): dmt_attn_bwd + the [rows, 960] x [960, 320] dmt_gemm with its residual.
 *           dWqkv = x^T dqkv stays dmt_wgrad320 (it reads the dqkv written here); dbias = column sums of dqkv.
 * image: dmt_mhsa_bwd_image_build(Wqkv fp32 [320, 960], ldw) -- dmt_mhsa_bwd_image_bytes() bytes, rebuilt after every optimizer step (private
 *        layout: 60 stages of 32 dx columns x 160 dqkv columns).
 * Masks, dropout counter (element index ((b*H + h)*T + q)*T + k) and packed rows (blocks / n_tiles / n_rows: the forward's table) as in
 * dmt_mhsa_block_fwd.  Rows t >= T of a padded tile do not exist; padded queries of the dense layout (lens[b] <= t < T) get dq = 0 and
 * reach dv with the reference's constant -2^32 + 1 weights, as in dmt_attn_bwd.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t d_model, num_heads;
  int32_t B, T;
  const void* ds;         /* bf16 [rows, d_model]                        */
  const void* qkv;        /* bf16 [rows, 3 * d_model] (forward side output) */
  const int32_t* lens;    /* [B]                                         */
  const void* image;
  void* dqkv;             /* bf16 [rows, 3 * d_model] out                */
  void* dx;               /* bf16 [rows, d_model] out                    */
  uint32_t drop_seed;
  float drop_keep;
  const int32_t* blocks;  /* packed rows (NULL: dense), as dmt_mhsa_desc */
  int32_t n_tiles;
  int64_t n_rows;
} dmt_mhsa_bwd_desc;

int dmt_mhsa_bwd_image_bytes(int64_t* bytes);
int dmt_mhsa_bwd_image_build(const float* wqkv, int64_t ldw, void* image, void* stream);
int dmt_mhsa_block_bwd(const dmt_mhsa_bwd_desc* d, void* stream);

/* Column sums: out[c] += sum_r x[r*ldx + c]  (fp32 out, atomics across row blocks; out zeroed by caller).
 * Used for the learned-position gradient (TransformerModel_util.py:302-306 lookup by range(T)).       */
int dmt_colsum(int32_t dtype, int64_t rows, int64_t cols, const void* x, int64_t ldx, float scale, float* out,
               int32_t ordered, void* stream);
/* out[c] += scale * sum_r mask(r*cols + c) * x[r, c] / keep  with the dmt_dropout counter mask (flat index r*cols + c, x must be
 * dense: ldx == cols): the gradient of the learned positions behind the dropout fused into dmt_gather_fwd. */
int dmt_colsum_drop(int32_t dtype, int64_t rows, int64_t cols, const void* x, float scale, float* out, uint32_t seed,
                    float keep_prob, int32_t ordered, void* stream);

/* The same gradient for a PACKED sequence (revision 5): out[t, c] += scale * sum_{b: lens[b] > t} mask((b*T + t)*d + c) * x[row_off[b] + t, c] / keep,
 * out fp32 [T, d]; bf16 rows, d % 8 == 0; keep_prob outside (0, 1): no mask.  ordered: one pass in example order (bit-reproducible). */
int dmt_colsum_rows_packed(int32_t dtype, int32_t B, int32_t T, int32_t d, const void* x, const int32_t* row_off, const int32_t* lens,
                           float scale, float* out, uint32_t seed, float keep_prob, int32_t ordered, void* stream);

/* Streaming 200-threshold confusion histogram behind tf.metrics.auc (run_dnn.py:228-241):
 * hist[(label?1:0) * (n_thr+1) + #thresholds below pred] += 1 (int64).                                */
int dmt_auc_hist(int32_t B, const float* pred, const float* label, int32_t n_thr, long long* hist, void* stream);
/* Streaming confusion counts behind tf.metrics.precision / tf.metrics.recall (run_dnn.py:221-227, 230-238; predictions are
 * tf.greater(p, 0.5) there, labels are cast to bool):  counts[0..3] += (tp, fp, fn, tn) with prediction = pred > threshold. */
int dmt_confusion_counts(int32_t B, const float* pred, const float* label, float threshold, long long* counts, void* stream);
/* out += sum over the DISTINCT ids of one feature of ||E[id]||^2 / 2: the per-feature term of l2_norm
 * (model/net/mmoe_transformer_unbias.py:42-60: tf.nn.l2_loss(tf.gather(E, tf.unique(ids)))).  seen: rows / 32 + 1 zeroed words. */
int dmt_l2_unique_rows(int32_t B, int32_t T, const int32_t* idx, const int32_t* lens, const float* table, int32_t rows, int32_t dim,
                       uint32_t* seen, float* out, void* stream);
/* The same, and mult[id] += 1 for every distinct id (mult: int32 [rows] of THIS table, zeroed by the caller before the first entry):
 * after all embedding_list entries, mult[row] = the number of entries whose batch holds the row = the weight of E[row] in the
 * gradient of l2_norm.  dmt_l2_rows_add then adds  coef[0] * mult[row] * E[row]  to the reduced embedding-gradient row of every
 * listed global row (uniq_keys / n_uniq / grad_rows as dmt_embgrad_reduce leaves them; mult indexed by GLOBAL row; coef on the device:
 * dLoss/dl2 * l2_emb_lambda / batch_size).  Replicated tables only.                                                               */
int dmt_l2_unique_rows_count(int32_t B, int32_t T, const int32_t* idx, const int32_t* lens, const float* table, int32_t rows, int32_t dim,
                             uint32_t* seen, float* out, int32_t* mult, void* stream);
int dmt_l2_rows_add(const dmt_table_map* tm, const float* p, const uint32_t* uniq_keys, const int32_t* n_uniq, int32_t max_uniq,
                    const int32_t* mult, const float* coef, float* grad_rows, int32_t max_dim, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused expert-MLP + gate kernels of the MMoE bottom (bf16; expert widths 512 -> 256 -> 128).
 *   g1 [B, E*512 + T*E]: relu'd layer-0 outputs of the E experts side by side | the T gates' logits (the output of the one
 *   concatenated layer-0 GEMM).
 *   forward : h1_e = relu(x_e W1_e + b1_e), h2_e = relu(h1_e W2_e + b2_e), gates[t] = softmax_e(logits[t]),
 *             mix[t] = sum_e gates[t][:, e] * h2_e                                       (one launch)
 *   backward: d mix -> dg1 (gradient of ALL columns of g1: expert inputs and gate logits), plus dh1 / dh2, the operands of the
 *             weight-gradient GEMMs dW1_e = x_e^T dh1_e, dW2_e = h1_e^T dh2_e            (one launch)
 * Weights are read from their bf16 shadows: transposed [E][N][ld] (k-contiguous) in the forward pass, plain [E][K][N] in the
 * backward pass; per-expert strides in elements.
 * Replaces: expert_gate's dense_layer stack for layers >= 1, the gate softmax and the weighted sum
 *           (mmoe_transformer_unbias.py:63-105; base.dense_layer base.py:58-70).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t B, E, T;             /* rows, experts, tasks (T <= 4, T*E <= 16) */
  int32_t u0, u1, u2;          /* 512, 256, 128 */
  const void* g1; int64_t ldg;
  const void* w1t; int64_t w1t_expert_stride, w1t_ld;     /* [E][u1][ld >= u0] bf16 */
  const void* w2t; int64_t w2t_expert_stride, w2t_ld;     /* [E][u2][ld >= u1] bf16 */
  const void* w1; int64_t w1_expert_stride;               /* [E][u0][u1] bf16 (backward) */
  const void* w2; int64_t w2_expert_stride;               /* [E][u1][u2] bf16 (backward) */
  const float* b1; int64_t b1_expert_stride;              /* [E][u1] fp32 */
  const float* b2; int64_t b2_expert_stride;              /* [E][u2] fp32 */
  void* h1; void* h2;          /* [B, E*u1], [B, E*u2] bf16: written by forward, read by backward */
  float* gates;                /* [T, B, E] fp32: written by forward, read by backward */
  void* mix;                   /* [T, B, u2] bf16 (forward) */
  const void* dmix;            /* [T, B, u2] bf16 (backward) */
  void* dh1; void* dh2;        /* [B, E*u1], [B, E*u2] bf16 (backward) */
  void* dg1; int64_t lddg;     /* [B, >= E*u0 + T*E] bf16 (backward) */
  /* revision 4.  ws: NULL, or a scratch buffer of >= dmt_mmoe_experts_ws_bytes(B) bytes (no initialisation needed; holds the experts'
   * d gate partials between the two launches of the backward): the E experts of a 32-row tile then run as E workgroups side by side
   * and a second small launch sums the mixtures / finishes the gate-logit gradient in expert order -- bit-identical to the
   * one-workgroup-per-tile form that ws == NULL selects.  One call at a time per buffer.
   * gate_dx != 0 (backward, ws form only): the expert-input columns of dg1 are multiplied by (g1 > 0), i.e. the relu gradient of the
   * layer that produced g1 is applied here instead of by a pass of its own. */
  void* ws; int64_t ws_bytes;
  int32_t gate_dx;
} dmt_mmoe_desc;

int64_t dmt_mmoe_experts_ws_bytes(int32_t B);

int dmt_mmoe_experts_supported(int32_t u0, int32_t u1, int32_t u2, int32_t E, int32_t T);
int dmt_mmoe_experts_fwd(const dmt_mmoe_desc* d, void* stream);
int dmt_mmoe_experts_bwd(const dmt_mmoe_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The heads in two launches (bf16): T task towers  mix[t] [128] -> relu(fc 32) -> 1 logit  (build_tower,
 * mmoe_transformer_unbias.py:107-126) and the position-bias tower  20 -> relu(32) -> dropout -> relu(16) -> dropout -> 1
 * (embedding_mlp_bias, :259-289).
 *   forward : logits [T+1][B] fp32 (task logits, then y_bias) + saved activations h_fc, h0, h1 (h0 / h1 after dropout).
 *   backward: dlogits [T+1][B] fp32 -> dmix, dzb, the pre-activation gradients dz_fc / dz0 / dz1 (operands of the weight-gradient
 *             GEMMs of the hidden layers), and the weight / bias gradients of the three 1-wide output layers ACCUMULATED into
 *             g_out_w / g_out_b / g_bias_w2 / g_bias_b2 (fp32 atomics: not for a bit-reproducible step).
 * Weights: bf16 plain shadows [K][N] row-major, fp32 biases.  Dropout: counter mask, index row * width + unit.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t B, T;                /* rows, task towers (<= 2) */
  int32_t u_in, u_fc;          /* 128, 32 */
  int32_t b_in, b_h0, b_h1;    /* 20, 32, 16 */
  const void* mix;             /* [T][B][u_in] bf16 */
  const void* zb; int64_t ld_zb;   /* [B][>= b_in] bf16 */
  const void* fc_w[2]; const float* fc_b[2];
  const void* out_w[2]; const float* out_b[2];
  const void* bias_w[3]; const float* bias_b[3];
  uint32_t drop_seed[2]; float drop_keep[2];     /* after bias-tower layers 0 and 1; keep >= 1 (or 0): off */
  float* logits;               /* [T+1][B] */
  void* h_fc; void* h0; void* h1;      /* [T][B][u_fc], [B][b_h0], [B][b_h1] bf16 */
  const float* dlogits;        /* [T+1][B] (backward) */
  void* dmix;                  /* [T][B][u_in] bf16 */
  void* dzb; int64_t ld_dzb;   /* [B][>= b_in] bf16 */
  void* dz_fc; void* dz0; void* dz1;   /* [T][B][u_fc], [B][b_h0], [B][b_h1] bf16 */
  float* g_out_w[2]; float* g_out_b[2]; float* g_bias_w2; float* g_bias_b2;
} dmt_heads_desc;

int dmt_heads_supported(int32_t u_in, int32_t u_fc, int32_t b_in, int32_t b_h0, int32_t b_h1, int32_t T);
int dmt_heads_fwd(const dmt_heads_desc* d, void* stream);
int dmt_heads_bwd(const dmt_heads_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Target-as-query cross attention over the RAW memory rows (Tq = 1; the decoder of every behaviour sequence).
 * With one query per example the K / V projections of the memory re-associate (dmt_q1mem.hip):
 *   q'_h = Q_h Wk[:, hc]^T [d]        score_k = q'_h . mem_k / sqrt(dh)        (the bias term Q_h . bk_h shifts all scores alike)
 *   ctx_h = sum_k Pd_k mem_k [d], S_h = sum_k Pd_k        out_h = ctx_h Wv[:, hc] + S_h bv_h
 * so the [B*T, d] x [d, 2d] projection, its input gradient and its weight gradient disappear; what is left are B-row GEMMs
 * (q', the V projection of (ctx | S), their gradients) around these two kernels.
 *   forward : ctx [B][H][ctx_hs]: cols 0..d-1 = ctx_h, col d = S_h, cols d+1.. = 0   (bf16)
 *   backward: dctx [B][H][d] bf16 + dout [B][H*dh] (dS_h = dout_h . bv_h) -> dqp [B][H][d] bf16, dmem [B][T][d] bf16
 * Masks, softmax, dropout counter as dmt_attn_fwd with Tq = 1 and no query lengths; d = 320, H <= 4, T <= 256, bf16.
 * Replaces: multihead_attention(target, memory, memory) of TransformerModel.decode (TransformerModel.py:146-160,
 *           TransformerModel_util.py:160-209) together with the K / V halves of its tf.layers.dense projections.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t B, T, H, d, dh;
  const void* mem; int64_t m_bs, m_rs;       /* [B][T][d] bf16 */
  const int32_t* k_lens;                     /* [B] or NULL */
  const void* qp;                            /* [B][H][d] bf16 */
  void* ctx; int64_t ctx_hs;                 /* [B][H][ctx_hs >= d + 8] bf16 */
  uint32_t drop_seed; float drop_keep;
  const void* dctx;                          /* [B][H][d] bf16 (backward) */
  const void* dout; int64_t do_bs;           /* [B][H*dh] bf16 */
  const float* bv;                           /* [H*dh] fp32 */
  void* dqp;                                 /* [B][H][d] bf16 */
  void* dmem; int64_t dm_bs, dm_rs;          /* [B][T][d] bf16 */
  /* revision 5, packed rows: row_off != NULL: example b's memory rows are rows row_off[b] + k of mem / dmem (m_bs / dm_bs ignored), k <
   * k_lens[b] (required then); T stays the dense length (dropout index, LDS sizing).                                                   */
  const int32_t* row_off;
} dmt_q1mem_desc;

int dmt_q1mem_supported(int32_t dtype, int32_t d, int32_t H, int32_t T);
int dmt_q1mem_fwd(const dmt_q1mem_desc* d, void* stream);
int dmt_q1mem_bwd(const dmt_q1mem_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMT_HIP_H */
