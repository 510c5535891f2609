/*
 * tmf.h - C ABI of libtmf.so, the MI355X (gfx950) engine behind
 * teamoflow.mf.MatrixFactorization.fit / predict / recall_at_k / retrieve_user_recs.
 *
 * The reference (GitHubOfAndrew/TeAMOFlow v0.0.2) has no FFI of its own: its hot path is a
 * sequence of TensorFlow eager ops issued from src/teamoflow/mf/matrix_factorization.py.  Each
 * entry point below replaces one group of those call sites (cited per function, paths relative to
 * the reference's src/teamoflow/mf/).  The Python host (teamoflow_amd/_lib.py) binds them with
 * ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a BORROWED DEVICE pointer (owned by the caller, e.g. a torch tensor);
 *     nothing is allocated, freed or synchronised inside the library;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the call returns
 *     immediately (graph-capture safe);
 *   - return value 0 = success, negative = TMF_E_* below; tmf_last_error() gives a thread-local
 *     message for the last failing call on this thread;
 *   - factor tables are fp32 row-major [rows, ld] with ld = tmf_padded_ld(n_components); the
 *     columns [n_components, ld) are zero and stay zero;
 *   - ids are int32, offsets into interaction / sample lists are int64.
 */
#ifndef TMF_H
#define TMF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMF_VERSION 203 /* 0.2.3: the balanced row-stationary item pass tmf_wsum_rows5; the 72-byte tmf_slice_lists carries `flags`: n_items counts only with TMF_SLICE_N_ITEMS_STATED */

#define TMF_OK 0
#define TMF_E_INVALID (-1)   /* bad argument (null pointer, unsupported rank, size mismatch) */
#define TMF_E_LAUNCH (-2)    /* hipLaunch / runtime error, text in tmf_last_error() */
#define TMF_E_UNSUPPORTED (-3)

/* epilogue selector of the row passes */
#define TMF_EPI_ADAM 0 /* out[row] = fresh-Adam(old[row], g[row])  (matrix_factorization.py:176) */
#define TMF_EPI_GRAD 1 /* out[row] = g[row]  (multi-GPU: reduce-scatter first, then tmf_adam_fresh_rows) */

int tmf_version(void);
const char* tmf_last_error(void);

/* Row length (in floats) the kernels use for an n_components-wide factor table: the next
 * size that is 4 floats x a power-of-two number of lanes (<= 64), or a multiple of 256 above that.
 * Returns 0 when n_components is outside [1, 1024]. */
int tmf_padded_ld(int n_components);

/* Same for bf16-stored tables (8 elements per 16-byte lane): the next 8 x power-of-two (<= 512), else 1024. */
int tmf_padded_ld_bf16(int n_components);

/* Scalars of one Keras Adam step at iteration 1 with zero moments, in fp32 - the reference builds
 * a new optimizer every epoch (matrix_factorization.py:176), so this is the whole optimizer state.
 *   alpha = lr*sqrt(1-b2)/(1-b1), one_minus_b1, one_minus_b2, eps = 1e-7 */
typedef struct tmf_adam {
    float alpha, one_minus_b1, one_minus_b2, eps;
} tmf_adam;
tmf_adam tmf_adam_fresh(float lr);

/* A pass over the rows of one side (users or items), cut into segments of at most `chunk`
 * list entries so that heavy rows are shared by several waves:
 *   segment s covers entries [rowptr[seg_row[s]] + seg_chunk[s]*chunk, ... + chunk) of its row;
 *   seg_slab[s] = -1 when the row has a single segment (the epilogue runs in the pass itself),
 *   otherwise the slot of `slab` [n_slab, ld] that receives this segment's partial gradient; rows with
 *   several segments are finished by tmf_combine_rows in slot order (deterministic, atomics-free). */
typedef struct tmf_segments {
    const int64_t* rowptr;    /* [rows + 1] */
    const int32_t* seg_row;   /* [nseg] */
    const int32_t* seg_chunk; /* [nseg] */
    const int32_t* seg_slab;  /* [nseg] */
    int64_t nseg;
    int32_t chunk;
    int32_t row_mod; /* 0, or n_rows when the lists are split by blocks of the OTHER side: list row =
                        block * n_rows + table row (every segment then owns a slab slot) */
} tmf_segments;

/* Index preparation, once per fit() (there is no counterpart in the reference: it gathers from the dense
 * [m, n] score matrix with SparseTensor.indices, loss_graphs.py:47-50; these views are what lets the passes stream).
 *   tmf_csr_build: COO (indices [nnz, 2] int64 pairs in any order, values [nnz]) -> CSR by user, interactions of a user in
 *     ascending item order (the row-major order tf.sparse.SparseTensor is specified in; duplicates keep their input
 *     order): rowptr_u [n_users + 1], col_u / val_u / user_of [nnz].
 *   tmf_csc_perm:  stable order of the CSR entries by item: perm [nnz] (CSR positions), rowptr_i [n_items + 1].
 *   tmf_stable_order_i32: the building block - stable radix sort of int32 keys in [0, n_rows) with their positions,
 *     plus rowptr[r] = first sorted position with key >= r.  sorted_keys may be NULL.
 * workspace: caller-provided device scratch of at least the matching *_workspace_bytes(). */
size_t tmf_csr_build_workspace_bytes(int64_t nnz);
int tmf_csr_build(const int64_t* indices, const float* values, int64_t nnz, int32_t n_users, int32_t n_items,
                  int64_t* rowptr_u, int32_t* col_u, float* val_u, int32_t* user_of, void* workspace,
                  size_t workspace_bytes, void* stream);
size_t tmf_stable_order_workspace_bytes(int64_t n);
int tmf_csc_perm(const int32_t* col_u, int64_t nnz, int32_t n_items, int64_t* rowptr_i, int64_t* perm,
                 void* workspace, size_t workspace_bytes, void* stream);
int tmf_stable_order_i32(const int32_t* keys, int64_t n, int64_t n_rows, int64_t* perm, int32_t* sorted_keys,
                         int64_t* rowptr, void* workspace, size_t workspace_bytes, void* stream);

/* Index structures of the sliced WMRB pass (built once per fit from the static negative table, utils.py:20):
 *   tmf_sort_samples: R_sorted[u, :] = R[u, :] in ascending item order (the order of a user's negatives is immaterial
 *     to the loss, loss_graphs.py:80-86).
 *   tmf_slice_offsets: off[row][b] = first position of the row's ascending ids with id >= b * ceil(n_items / n_slices),
 *     b = 0 .. n_slices (the last one = row length).  Rows have a fixed length `stride` (rowptr NULL: the negatives) or
 *     are CSR rows (the interactions; offsets relative to rowptr[row]).
 *   tmf_wmrb_entry_lists: the item-side lists of the WMRB gradient, split by user block (list row = user block * n_items
 *     + item, blocks of ceil(n_users / user_chunks) users): every list holds the item's positives (ascending CSR
 *     position) followed by the (user, sample) pairs whose negative is the item.  Entry ids: i < nnz = interaction i of
 *     the CSR, nnz + u * S + s = negative s of user u (R_sorted order).  Outputs, E = nnz + n_users * S (< 2^31):
 *       ent_row [E] user of every list entry; ent_id [E] entry id of every list entry (= its index into the weight
 *       buffer [delta | D] of tmf_wsum_pass; stored values <= 0 sit in a dummy list row behind all others);
 *       rowptr_e [user_chunks * n_items + 2]. */
size_t tmf_sort_samples_workspace_bytes(int32_t n_users, int32_t n_samples);
int tmf_sort_samples(const int32_t* R, int32_t n_users, int32_t n_samples, int32_t n_items, int32_t* R_sorted,
                     void* workspace, size_t workspace_bytes, void* stream);
int tmf_slice_offsets(const int32_t* ids, const int64_t* rowptr, int64_t stride, int32_t n_rows, int32_t n_items,
                      int32_t n_slices, int32_t* off, void* stream);
size_t tmf_wmrb_entry_lists_workspace_bytes(int64_t nnz, int32_t n_users, int32_t n_samples);
int tmf_wmrb_entry_lists(const int32_t* user_of, const int32_t* col_u, const float* val_u, int64_t nnz,
                         const int32_t* R_sorted, int32_t n_users, int32_t n_samples, int32_t n_items, int32_t user_chunks,
                         int32_t* ent_row, int32_t* ent_id, int64_t* rowptr_e, void* workspace, size_t workspace_bytes,
                         void* stream);

/* K1+K2 / K3: one side of an MSE epoch (loss_graphs.py:47-52 forward; tape.gradient
 * matrix_factorization.py:170-171; Adam :176) evaluated sparsely:
 *   for every entry k of row i:  p = <X_old[i], Y_old[other[k]]>, e = val[k] - p,
 *   loss += e*e, g[i] += (-2e) * Y_old[other[k]]; then the epilogue `epi` writes X_out[i].
 * Call once with (X=U, Y=V, CSR by user) and once with (X=V, Y=U, CSC by item); both read the
 * PRE-update tables.  loss_part (optional, [nseg]) receives the per-segment sum of e*e. */
int tmf_mse_pass_f32(const tmf_segments* seg, const int32_t* other, const float* val,
                     const float* X_old, const float* Y_old, float* X_out, float* slab,
                     float* loss_part, int n_components, int epi, tmf_adam adam, void* stream);

/* Weighted row-gather-sum pass (item side of WMRB, matrix_factorization.py:170-171 through
 * loss_graphs.py:80-88):  g[i] = sum over entries e of row i of  wbuf[ent_w[e]] * T[ent_row[e]]
 * (entries with weight exactly 0 are skipped), then the epilogue writes X_out[i]. */
int tmf_wsum_pass_f32(const tmf_segments* seg, const int32_t* ent_row, const int32_t* ent_w,
                      const float* wbuf, const float* T, const float* X_old, float* X_out,
                      float* slab, int n_components, int epi, tmf_adam adam, void* stream);

/* Row-stationary form of tmf_wsum_pass + tmf_combine_rows over user-blocked lists (speed only): the lists are those of
 * tmf_wmrb_entry_lists with `n_blocks` user blocks - list row = block * n_rows + i, rowptr [n_blocks * n_rows + 1] - but a
 * lane group owns a few consecutive output rows i, keeps their sums in registers and walks the blocks itself: no slab, no
 * combine.  g[i] = sum over blocks (ascending) and list entries e of wbuf[ent_w[e]] * T[ent_row[e]], then the epilogue.
 * The rows are launched `rows_per_launch` at a time (a multiple of tmf_wsum_rows4_rows_per_group(); all workgroups of a
 * launch resident together); workspace (optional): tmf_wsum_rows4_workspace_bytes() of device memory for the per-block
 * rendezvous counters, zeroed by the call.  Rows of at least 16 lanes. */
int tmf_wsum_rows4_rows_per_group(int n_components, int bf16);
size_t tmf_wsum_rows4_workspace_bytes(int32_t n_rows, int32_t n_blocks, int32_t rows_per_launch);
int tmf_wsum_rows4_f32(const int64_t* rowptr, int32_t n_rows, int32_t n_blocks, const int32_t* ent_row, const int32_t* ent_w,
                       const float* wbuf, const float* T, const float* X_old, float* X_out, int n_components, int epi,
                       tmf_adam adam, int32_t rows_per_launch, void* workspace, size_t workspace_bytes, void* stream);

/* BALANCED row-stationary form (round 5; speed only): what the lane groups own are VIRTUAL rows - (output row, part p of P), listed
 * in (row, part) order - so that a row with far more list entries than the average (a popular item: one entry per user against
 * 1,400 on average at config 5) is cut into P parts of about the average size and every lane group walks about the same number of
 * entries.  In block t part p of row i takes the entries [b + p L / P, b + (p + 1) L / P) of that block's list [b, b + L).
 *   vr_item / vr_part / vr_nparts [n_vrows + 1]: output row, part and number of parts of every virtual row; the last element is
 *   the sentinel (n_rows, 0, 1);  vr_slot [n_vrows]: -1 = the row is whole (P = 1) and finished by the epilogue, else the slab slot
 *   its partial sum goes to - the caller then finishes the cut rows with tmf_combine_rows: slots of a row consecutive, in part
 *   order.  A fixed order of additions: results are bit-reproducible; against tmf_wsum_rows4 / tmf_wsum_pass equal to rounding.
 * rows_per_launch counts virtual rows (a multiple of tmf_wsum_rows4_rows_per_group()). */
size_t tmf_wsum_rows5_workspace_bytes(int32_t n_vrows, int32_t n_blocks, int32_t rows_per_launch);
int tmf_wsum_rows5_f32(const int64_t* rowptr, int32_t n_rows, int32_t n_blocks, const int32_t* ent_row, const int32_t* ent_w,
                       const float* wbuf, const float* T, const float* X_old, float* X_out, float* slab, const int32_t* vr_item,
                       const int32_t* vr_part, const int32_t* vr_nparts, const int32_t* vr_slot, int32_t n_vrows, int n_components,
                       int epi, tmf_adam adam, int32_t rows_per_launch, void* workspace, size_t workspace_bytes, void* stream);

/* Finishes the rows that tmf_*_pass cut into several segments: g[row] = sum of its slab slots
 * [slab_beg[i], slab_beg[i+1]) in order, then the epilogue. */
int tmf_combine_rows_f32(const int32_t* long_rows, const int64_t* slab_beg, int64_t n_long,
                         const float* slab, const float* X_old, float* X_out, int n_components,
                         int epi, tmf_adam adam, void* stream);

/* K4+K5: user side of a WMRB epoch for users [0, n_users) in ONE kernel (matrix_factorization.py:153-154 sampled
 * and serial scores, loss_graphs.py:74-88, gradient w.r.t. U, Adam :176) - the form for catalogs whose V table the L2s
 * hold and sample counts whose scores fit LDS (tmf_wmrb_user_pass_fits); everything else takes the sliced pass below.
 *   R [n_users, S] int32 static negative table (utils.py:20), c = n_items / n_samples (ctor ints);
 *   writes delta [nnz] (d loss / d p_k, 0 for non-positive entries), D [n_users, S],
 *   loss_part [n_users] (sum of log(1+M_k) over the user's positives), pos_part [n_users] (#positives),
 *   and U_out via the epilogue. */
int tmf_wmrb_user_pass_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                           const int32_t* R, int32_t n_users, int32_t S, float c,
                           const float* U_old, const float* V_old, float* U_out, float* delta,
                           float* D, float* loss_part, float* pos_part,
                           int n_components, int epi, tmf_adam adam, void* stream);
/* 1 when the scores and D of one user (n_samples of them) fit the 160 KB of LDS next to the kernel's other buffers. */
int tmf_wmrb_user_pass_fits(int32_t n_samples, int n_components);

/* Sliced form of the same user pass (speed only - same contract): the catalog is walked in n_slices slices of
 * ~4 MB of V rows with a slice-major grid, so the row gathers of the resident workgroups hit the XCD L2s; it also has
 * no limit on n_samples, and its hinge step costs O((S + P_u) log P_u) per user instead of O(P_u S).  The lists: */
typedef struct tmf_slice_lists {
    const int32_t* R_sorted;  /* [n_users, n_samples] negatives, ascending item id per user (tmf_sort_samples) */
    const int32_t* slice_off; /* [n_users, n_slices + 1] (tmf_slice_offsets on R_sorted) */
    const int64_t* rowptr;    /* [n_users + 1] CSR of the interactions, ascending item id inside a user (tmf_csr_build) */
    const int32_t* col;       /* [nnz] */
    const int32_t* pos_off;   /* [n_users, n_slices + 1] (tmf_slice_offsets on col with rowptr) */
    int32_t n_users, n_samples, n_slices;
    /* Window of the catalog a launch covers (item-row-sharded V, one window of rows resident at a time): slices
     * [slice_begin, slice_begin + slice_count) and the V pointer of the call addresses item row `item_base` (the first row
     * of the window).  All three 0 = the whole catalog. */
    int32_t slice_begin, slice_count, item_base;
    /* Bit set (this field was `xcd_major` = 0 | 1 up to version 202, so older callers keep their meaning):
     *   TMF_SLICE_XCD_MAJOR (1)      block order of the slice kernels (speed only): clear = slice-major (every resident workgroup
     *                                walks the same slice, each XCD L2 holds a copy of it), set = XCD-major (the XCD of block b - b mod 8
     *                                under the observed round-robin placement - walks the slices slice_begin + 8 i + (b mod 8): eight
     *                                different slices resident, one per L2);
     *   TMF_SLICE_N_ITEMS_STATED (2) `n_items` below is meaningful.  The struct is 72 bytes: 5 pointers + 8 int32.  Up to version 201
     *                                it ended after this field (68 bytes + 4 of tail padding), so a caller built against that
     *                                header passes indeterminate bytes where n_items now sits: they are ignored unless this bit says
     *                                the caller filled them in. */
    int32_t flags;
    /* Items in the catalog the ids of R_sorted / col index (the rows of V); counts only with TMF_SLICE_N_ITEMS_STATED.  Speed
     * only: when stated and n_items * row bytes < 2^32, tmf_wmrb_scores3 addresses V with 32-bit offsets (one instruction per row
     * address).  TMF_CHECK_IDS=1 in the environment makes the call verify (synchronously) that every id is below it. */
    int32_t n_items;
} tmf_slice_lists;
#define TMF_SLICE_XCD_MAJOR 1
#define TMF_SLICE_N_ITEMS_STATED 2
/* Kernels, called in this order on one stream (tables float (_f32) or bf16 (_bf16) rows as void*; sp / p / D / delta /
 * part / w_ent are fp32):
 *   tmf_wmrb_scores3_*  sp[u, s] = <U[u], V[R_sorted[u, s]]>, p[k] = <U[u_k], V[col[k]]>         (slice-major grid)
 *   tmf_wmrb_hinge2     delta [nnz], D [n_users, n_samples] (R_sorted order), loss_part [n_users]; reads no table
 *   tmf_wmrb_gradu3_*   part[slice][u] = sum_{s in slice} D[u, s] V[R_sorted[u, s]] + sum_{k in slice} delta_k V[col[k]]
 *                       per_slice_launches = 0: one launch, part is [n_slices * n_users, ld], finish gets n_slices;
 *                       per_slice_launches = 1: one launch per slice adding into ONE [n_users, ld] layer (memory-light;
 *                       finish is then called with n_slices = 1); the first slice launched overwrites the layer;
 *                       per_slice_launches = 3: one launch per ROUND of eight slices into EIGHT layers (part is
 *                       [8 * n_users, ld]; the slice slice_begin + 8 i + x goes to layer x; finish gets min(8, slices));
 *                       per_slice_launches = 2: the same as 1, but the first slice launched adds to the layer as well (the
 *                       later windows of a windowed pass)
 *   tmf_wmrb_finish_*   U_out[u] = epilogue(sum_slice part[slice][u]); with TMF_EPI_GRAD, U_out == part is allowed (the
 *                       layers are summed in place into layer 0) */
int tmf_wmrb_scores3_f32(const tmf_slice_lists* lists, const void* U, const void* V, float* sp, float* p,
                         int n_components, void* stream);
int tmf_wmrb_scores3_bf16(const tmf_slice_lists* lists, const void* U, const void* V, float* sp, float* p,
                          int n_components, void* stream);
/* Row-stationary form of scores3 (speed only - the same sp / p up to the order of the fp32 sum inside a dot product; exact on
 * dyadic data): for catalogs far beyond the L2s, where a (user, slice) visit holds a handful of rows and scores3 is bound by what
 * every visit reads besides them (matrix_factorization.py:153-154 and utils.py:94-105 are what both compute).  A workgroup owns
 * tmf_wmrb_scores5_users_per_workgroup() consecutive users, keeps their rows in LDS and walks ONE flat stream of its (user, item)
 * pairs - negatives and interactions alike, ordered by item (slice) so that all workgroups of a launch gather from the same
 * cache-sized window of V at any time:
 *   ids [E8]     (local user << 24) | item            (E8 = every workgroup's entries padded to a multiple of 8)
 *   outs [E8]    >= 0: index into sp ([n_users, n_samples] flat); < 0: ~index into p ([nnz]); INT32_MIN: padding entry
 *   wg_ptr [n_wg + 1]  first entry of every workgroup's stream (multiples of 8), n_wg = ceil(n_users / users_per_workgroup)
 * wgs_per_launch <= 0: one workgroup per CU (all workgroups of a launch resident, walking the catalog at the same pace).
 * Pacing (optional, speed only): wstart [n_wg, n_windows + 1] = the first step (8 entries) of every catalog window in every
 * workgroup's stream (wstart[., n_windows] = its steps) and a workspace of tmf_wmrb_scores5_workspace_bytes() (zeroed by the
 * call): a workgroup then starts window w only when the workgroups sharing its XCD have completed window w - lag - 1 (bounded
 * waits), so that the rows being gathered span lag + 1 windows.  NULL = every workgroup runs freely.
 * Needs rows of 32 lanes (fp32 65..128, bf16 129..256 components), n_items < 2^24 and a V table below 4 GB:
 * tmf_wmrb_scores5_supported(). */
int tmf_wmrb_scores5_users_per_workgroup(void);
int tmf_wmrb_scores5_supported(int n_components, int bf16, int64_t n_items);
size_t tmf_wmrb_scores5_workspace_bytes(int64_t n_wg, int32_t n_windows, int wgs_per_launch);
int tmf_wmrb_scores5_f32(const int32_t* ids, const int32_t* outs, const int64_t* wg_ptr, int64_t n_wg, int64_t n_users,
                         int64_t n_items, const void* U, const void* V, float* sp, float* p, int n_components,
                         int wgs_per_launch, const int32_t* wstart, int32_t n_windows, int lag, void* workspace,
                         size_t workspace_bytes, void* stream);
int tmf_wmrb_scores5_bf16(const int32_t* ids, const int32_t* outs, const int64_t* wg_ptr, int64_t n_wg, int64_t n_users,
                          int64_t n_items, const void* U, const void* V, float* sp, float* p, int n_components,
                          int wgs_per_launch, const int32_t* wstart, int32_t n_windows, int lag, void* workspace,
                          size_t workspace_bytes, void* stream);
/* Flat streams on the slice-major grid (round 5; speed only - the same sp / p as tmf_wmrb_scores3 up to the order of the fp32 sum
 * inside a dot product, exact on dyadic data): one workgroup per CHUNK = (item slice, group of tmf_wmrb_scores6_users_per_group()
 * consecutive users); the group's rows are held in LDS and the chunk's entries are one contiguous piece of
 *   ids [E8]   (user - first user of the group) << 24 | item      outs [E8]   as for tmf_wmrb_scores5
 *   chunk_ptr [n_slices * n_groups + 1]   first entry of chunk slice * n_groups + group (every chunk padded to a multiple of 8)
 * - interactions first, then negatives, each by user and item - so a (user, slice) visit costs no offsets, no row of U and no
 * dependent round trip of its own.  The slices only order the stream and the grid (the kernel never sees their bounds): any
 * partition of the item ids into n_slices ascending ranges will do; ~4 MB of V rows each keeps a slice in the XCD L2s.
 * Same limits as tmf_wmrb_scores5: rows of 32 lanes, n_items < 2^24, V below 4 GB (tmf_wmrb_scores6_supported). */
int tmf_wmrb_scores6_users_per_group(void);
int tmf_wmrb_scores6_supported(int n_components, int bf16, int64_t n_items);
int tmf_wmrb_scores6_f32(const int32_t* ids, const int32_t* outs, const int64_t* chunk_ptr, int64_t n_groups, int32_t n_slices,
                         int64_t n_users, int64_t n_items, const void* U, const void* V, float* sp, float* p, int n_components,
                         void* stream);
int tmf_wmrb_scores6_bf16(const int32_t* ids, const int32_t* outs, const int64_t* chunk_ptr, int64_t n_groups, int32_t n_slices,
                          int64_t n_users, int64_t n_items, const void* U, const void* V, float* sp, float* p, int n_components,
                          void* stream);
int tmf_wmrb_hinge2(const int64_t* rowptr, const float* val, const float* p, const float* sp, int32_t n_users,
                    int32_t n_samples, float c, float* delta, float* D, float* loss_part, void* stream);
/* The same with the order in which the waves take the users (a permutation of 0 .. n_users - 1, or NULL = 0, 1, 2 ...): a user
 * costs one pass per 255 interactions, so the few with thousands should start first, not last.  Speed only - every user is
 * computed by itself and the outputs do not depend on the order. */
int tmf_wmrb_hinge2_ordered(const int64_t* rowptr, const float* val, const float* p, const float* sp, int32_t n_users,
                            int32_t n_samples, float c, float* delta, float* D, float* loss_part, const int32_t* user_order,
                            void* stream);
int tmf_wmrb_gradu3_f32(const tmf_slice_lists* lists, const float* D, const float* delta, const void* V, float* part,
                        int per_slice_launches, int n_components, void* stream);
int tmf_wmrb_gradu3_bf16(const tmf_slice_lists* lists, const float* D, const float* delta, const void* V, float* part,
                         int per_slice_launches, int n_components, void* stream);
/* Row-stationary form of gradu3 + finish in one kernel (speed only - same result up to the order of the fp32 sum over a
 * user's slices, which is ascending here too): every lane group owns a few users, keeps their gradient rows in registers
 * and walks all slices itself; U_out[u] = epilogue(sum_slice ...).  The users are launched in blocks of `users_per_launch`
 * (all workgroups of a block resident together, so they walk the slices in loose lockstep and the slice stays in the L2s).
 * For catalogs far beyond the L2s, where a (user, slice) range holds a handful of rows. */
int tmf_wmrb_gradu4_supported(int n_components, int bf16);   /* rows of at least 8 lanes (r > 28 fp32, r > 56 bf16) */
size_t tmf_wmrb_gradu4_workspace_bytes(int32_t n_users, int32_t n_slices, int32_t users_per_launch);
/* workspace (optional, device memory): one int per (launch, slice), zeroed by the call; with it the workgroups of a launch
 * rendezvous once per slice (bounded wait, speed only) so that they stay within two slices of each other. */
int tmf_wmrb_gradu4_f32(const tmf_slice_lists* lists, const float* D, const float* delta, const void* V, const void* U_old,
                        void* U_out, int n_components, int epi, tmf_adam adam, int32_t users_per_launch, void* workspace,
                        size_t workspace_bytes, void* stream);
int tmf_wmrb_gradu4_bf16(const tmf_slice_lists* lists, const float* D, const float* delta, const void* V, const void* U_old,
                         void* U_out, int n_components, int epi, tmf_adam adam, int32_t users_per_launch, void* workspace,
                         size_t workspace_bytes, void* stream);
int tmf_wmrb_finish_f32(const float* part, int32_t n_slices, int32_t n_users, const void* U_old, void* U_out,
                        int n_components, int epi, tmf_adam adam, void* stream);
int tmf_wmrb_finish_bf16(const float* part, int32_t n_slices, int32_t n_users, const void* U_old, void* U_out,
                         int n_components, int epi, tmf_adam adam, void* stream);

/* K6 standalone: W[rows] = fresh-Adam(W[rows], G[rows]) in place over n_rows x ld floats. */
int tmf_adam_fresh_rows_f32(float* W, const float* G, int64_t n_rows, int n_components,
                            tmf_adam adam, void* stream);

/* OPT-IN EXTENSION, not the reference's optimiser (which is rebuilt every epoch, matrix_factorization.py:176): Keras
 * Adam with persistent moments.  tmf_adam_step gives the scalars of iteration `step` (1-based; step 1 == tmf_adam_fresh),
 * tmf_adam_state_rows_f32 applies one step in place to a whole [n_rows, ld] table from its raw gradient G (the
 * TMF_EPI_GRAD output of the passes) and the moment tables M, V (zero before the first step). */
tmf_adam tmf_adam_step(float lr, int step);
int tmf_adam_state_rows_f32(float* W, const float* G, float* M, float* V, int64_t n_rows, int n_components,
                            tmf_adam adam, void* stream);

/* Deterministic sum of `n` floats into out[0] (fp64 accumulate, fixed order) - the reduce_mean
 * numerator of matrix_factorization.py:179. */
int tmf_sum_f32(const float* x, int64_t n, double* out, void* stream);

/* utils.py:94-105 gather_matrix_indices: out[i, c] = X[i, idx[i, c]];  X [rows, cols] fp32,
 * idx [rows, k] int64. */
int tmf_gather_rows_cols_f32(const float* X, const int64_t* idx, float* out, int64_t rows,
                             int64_t cols, int64_t k, void* stream);

/* K7: C[m, n] = A[m, :r] . B[n, :r]^T in exact fp32 on the f32 MFMA (matrix_factorization.py:149,195).
 * lda / ldb / ldc in floats. */
int tmf_predict_gemm_f32(const float* A, const float* B, float* C, int64_t m, int64_t n, int r,
                         int64_t lda, int64_t ldb, int64_t ldc, void* stream);

/* K8: row-wise top-k of X [rows, cols] ordered (value desc, index asc) - tf.math.top_k's contract
 * (matrix_factorization.py:245,429-438), any k in [1, cols] including the full ranking retrieve_user_recs(k=None) and
 * dcg / ndcg ask for (:336,:367,:424-438).  clamp_negatives != 0 first maps x <= 0 to 0.0 (matrix_factorization.py:237).
 * out_idx [rows, k] int32, out_val optional [rows, k].  k <= 64, or rows of at most 16384 columns, need no workspace;
 * larger k over wider rows is a stable segmented radix sort through `workspace` (tmf_topk_workspace_bytes; rows * cols
 * < 2^32 per call). */
size_t tmf_topk_workspace_bytes(int64_t rows, int64_t cols, int k);
int tmf_topk_stable_f32(const float* X, int64_t rows, int64_t cols, int64_t ldx, int k,
                        int clamp_negatives, int32_t* out_idx, float* out_val, void* workspace,
                        size_t workspace_bytes, void* stream);

/* bf16-storage / fp32-arithmetic variants (BASELINE config 5: "bf16 factors / fp32 accum"; an extension - the
 * reference is fp32 throughout).  Tables are bf16 row-major [rows, tmf_padded_ld_bf16(r)]; every product,
 * sum, loss and the optimiser step are computed in fp32 and the new row is rounded to bf16 once
 * (round-to-nearest-even).  slab / loss / delta / D / TMF_EPI_GRAD outputs stay fp32 (the raw gradient is what
 * RCCL reduce-scatters).  Arguments otherwise exactly as the _f32 functions. */
int tmf_mse_pass_bf16(const tmf_segments* seg, const int32_t* other, const float* val,
                      const void* X_old, const void* Y_old, void* X_out, float* slab,
                      float* loss_part, int n_components, int epi, tmf_adam adam, void* stream);
int tmf_wsum_pass_bf16(const tmf_segments* seg, const int32_t* ent_row, const int32_t* ent_w,
                       const float* wbuf, const void* T, const void* X_old, void* X_out,
                       float* slab, int n_components, int epi, tmf_adam adam, void* stream);
int tmf_wsum_rows4_bf16(const int64_t* rowptr, int32_t n_rows, int32_t n_blocks, const int32_t* ent_row, const int32_t* ent_w,
                        const float* wbuf, const void* T, const void* X_old, void* X_out, int n_components, int epi,
                        tmf_adam adam, int32_t rows_per_launch, void* workspace, size_t workspace_bytes, void* stream);
int tmf_wsum_rows5_bf16(const int64_t* rowptr, int32_t n_rows, int32_t n_blocks, const int32_t* ent_row, const int32_t* ent_w,
                        const float* wbuf, const void* T, const void* X_old, void* X_out, float* slab, const int32_t* vr_item,
                        const int32_t* vr_part, const int32_t* vr_nparts, const int32_t* vr_slot, int32_t n_vrows, int n_components,
                        int epi, tmf_adam adam, int32_t rows_per_launch, void* workspace, size_t workspace_bytes, void* stream);
int tmf_combine_rows_bf16(const int32_t* long_rows, const int64_t* slab_beg, int64_t n_long,
                          const float* slab, const void* X_old, void* X_out, int n_components,
                          int epi, tmf_adam adam, void* stream);
int tmf_wmrb_user_pass_bf16(const int64_t* rowptr, const int32_t* col, const float* val,
                            const int32_t* R, int32_t n_users, int32_t S, float c,
                            const void* U_old, const void* V_old, void* U_out, float* delta,
                            float* D, float* loss_part, float* pos_part,
                            int n_components, int epi, tmf_adam adam, void* stream);
int tmf_adam_fresh_rows_bf16(void* W, const float* G, int64_t n_rows, int n_components,
                             tmf_adam adam, void* stream);

/* K7+K8 fused: out_idx[u, :k] = top-k (value desc, index asc) of A[u, :r] . B[:, :r]^T over all n items,
 * without materialising the [m, n] scores (recall_at_k / retrieve_user_recs, matrix_factorization.py:236-248,
 * :424-438, at catalog sizes where the dense matrix does not fit).  Supports k <= 64 and r <= 256; returns
 * TMF_E_UNSUPPORTED otherwise (callers then score block-wise with tmf_predict_gemm_f32 + tmf_topk_stable_f32). */
int tmf_predict_topk_f32(const float* A, const float* B, int64_t m, int64_t n, int r, int64_t lda,
                         int64_t ldb, int k, int clamp_negatives, int32_t* out_idx, float* out_val,
                         void* stream);

/* The same for bf16-stored tables (A [m, lda], B [n, ldb] bf16, ld %% 8 == 0) on the bf16 MFMA: products of
 * bf16 values are exact and accumulate in fp32 ("bf16 factors / fp32 accum").  k <= 32, r <= 256. */
int tmf_predict_topk_bf16(const void* A, const void* B, int64_t m, int64_t n, int r, int64_t lda,
                          int64_t ldb, int k, int clamp_negatives, int32_t* out_idx, float* out_val,
                          void* stream);

/* tmf_predict_topk_f32 for fp32 tables on the bf16 matrix cores, fp32-accurate: every factor is split exactly into three
 * bf16 planes and a score is the sum of the six plane products that weigh >= 2^-24 of the leading one, small ones first,
 * accumulated in fp32 (csrc/tmf_predict_split.hip; errors against fp64 at or below those of the fp32 MFMA kernel).  Same
 * outputs, order and tie rule as tmf_predict_topk_f32; values may differ from it by fp32 rounding.  r <= 256 (since version 203; 128 before), k <= 40 (32 before the 4-wave
 * instances of round 5: ask tmf_predict_topk_split_supported); `workspace` holds the item table's planes (query the size; overwritten per call). */
int tmf_predict_topk_split_supported(int r, int k);
size_t tmf_predict_topk_split_workspace_bytes(int64_t n, int r);
int tmf_predict_topk_split_f32(const float* A, const float* B, int64_t m, int64_t n, int r, int64_t lda,
                               int64_t ldb, int k, int clamp_negatives, int32_t* out_idx, float* out_val,
                               void* workspace, size_t workspace_bytes, void* stream);
/* The same ranking with TWO fp16 planes per factor and three plane products (h2 v1 + h1 v2 + h1 v1, fp32 accumulate): every
 * user row is scaled by its own power of two and the item table by one, so that the planes sit in fp16's normal range; 22
 * bits of every factor take part (all 24 with the three bf16 planes above), and a factor below ~1e-6 of its table's largest
 * magnitude keeps fewer.  Values against fp64: at the fp32 MFMA kernel's error (csrc/tmf_predict_split.hip, tests).  Half the
 * matrix-core work of the three-plane form.  r <= 256, k <= 32 (tmf_predict_topk_half2_supported); its own workspace size. */
int tmf_predict_topk_half2_supported(int r, int k);
size_t tmf_predict_topk_half2_workspace_bytes(int64_t n, int r);
int tmf_predict_topk_half2_f32(const float* A, const float* B, int64_t m, int64_t n, int r, int64_t lda,
                               int64_t ldb, int k, int clamp_negatives, int32_t* out_idx, float* out_val,
                               void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TMF_H */
