/*
 * pygat_amd.h -- C ABI of the MI355X (gfx950) GAT attention-layer hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  The
 * reference (ArielleRosinski/pyGAT) is pure Python and owns no FFI; the entry
 * points below replace the ATen call sequences of its layer/op boundary:
 *
 *   GraphAttentionLayer.forward          layers.py:32-53   (dense N x N path)
 *   _prepare_attentional_mechanism_input layers.py:55-64   (s_i + t_j split)
 *   SpecialSpmmFunction.forward/backward layers.py:72-90   (COO spmm + its grads)
 *   SpGraphAttentionLayer.forward        layers.py:125-173 (edge-list path)
 *   GAT.forward head concat / head mean  models.py:29-35
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host"; the caller owns all
 *     memory including workspaces; nothing here allocates, frees or synchronises.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); kernels
 *     are only enqueued, never waited for.
 *   - return 0 on success, a negative PYGAT_E* code otherwise; the message of the
 *     last error on the calling thread is at pygat_last_error().  Nothing throws.
 *   - floats are fp32, indices int32.  Feature tables are row-major and PADDED:
 *     a level with H heads of F' outputs uses Fp = pygat_padded_width(F') columns
 *     per head and R = H*Fp floats per node row ("head-interleaved row").
 *   - graph = CSR pattern of the adjacency: row i lists the j with adj[i][j] != 0
 *     (layers.py:129: edge[0] = i = softmax row, edge[1] = j = gathered node).
 *   - no mutable library state besides the thread-local error string, and (since ABI 13) no environment variable is
 *     read while the library runs: two process-level defaults are read ONCE when it is loaded -- PYGAT_GEMM_F32=1
 *     (what PYGAT_GEMM_DEFAULT means, see the GEMM section) and PYGAT_NARROW=0 (pygat_dropout_narrow answers 0) --
 *     and every layout choice a caller can see (head windows of the backward, slot length, GEMM product mode, split-K)
 *     is an ARGUMENT.
 */
#ifndef PYGAT_AMD_H
#define PYGAT_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYGAT_ABI_VERSION 14

enum {
  PYGAT_OK = 0,
  PYGAT_EINVAL = -1,   /* bad argument (null pointer, size, alignment, unsupported width) */
  PYGAT_EHIP = -2,     /* a HIP launch failed; message holds hipGetErrorString */
  PYGAT_ENODEV = -3    /* no gfx950 device / code object not loadable */
};

/* flags for pygat_gat_forward / pygat_gat_backward */
enum {
  PYGAT_F_ELU = 1,      /* apply ELU to (attention + skip): concat=True, layers.py:50-51,168-170 */
  PYGAT_F_SKIP = 2,     /* add the skip projection rows `sk` (layers.py:47-48,165-166) */
  /* pygat_gat_forward only (ABI 14): run ONE phase of the pass over the graph's slot range -- the main launch (rows that lie
   * inside a slot are final after it) or the fix-up launch (finishes the rows a slot border cuts, from the main launch's
   * partial records).  A row-chunk pipeline puts chunk c's fix-up on a second stream beside chunk c + 1's main launch.
   * Needs the level in one head window: pygat_gat_forward_phases_ok(n, H, F') == 1. */
  PYGAT_F_MAIN_ONLY = 4,
  PYGAT_F_FIXUP_ONLY = 8
};

int pygat_abi_version(void);
const char* pygat_last_error(void);
/* Fp for a head width F' (power of two >= 4, <= 256); 0 if unsupported. */
int pygat_padded_width(int f_out);
/* number of HIP devices visible / name of the current one (host buffers). */
int pygat_device_count(void);
int pygat_device_name(char* host_buf, int len);
/* Registers per lane and scratch bytes per lane of a tuned kernel as the LOADED code object reports them (hipFuncGetAttributes):
 * "k2_headline" (fused forward, 8 heads x 16, training), "k4_headline_da" (column pass with the a-gradient sums), "tn_x3w"
 * (streamed-K weight gradient), "x3gw" (general split-bf16 GEMM).  The attention kernels were tuned at four waves per SIMD
 * without scratch; `amdgpu_waves_per_eu` makes the compiler spill rather than fail, so tests/test_gpu_properties.py asks. */
int pygat_kernel_footprint(const char* kernel, int* num_regs, int* scratch_bytes);

/* How the fp32 GEMMs of a level (the projection `mm(h, W)`, layers.py:35,134, its weight and input gradients) form
 * their products.  Operands, accumulators and results are fp32 in both modes.
 *   PYGAT_GEMM_SPLIT_BF16: each operand is cut EXACTLY into three bf16 pieces (8 + 8 + 8 significant bits) and all
 *     nine piece products are summed into fp32 accumulators by v_mfma_f32_32x32x16_bf16 -- no operand bit is dropped,
 *     the fp32 additions happen in another order than below; 288 instead of 512 MFMA cycles per 32 x 32 x 16 block.
 *     Shapes the split kernels do not take use the fp32 kernels.  Caveats: the cut is exact for finite operands whose
 *     mid and low pieces stay normal bf16 numbers (|x| >= 2^-110 or x == 0; smaller magnitudes lose their low bits,
 *     i.e. are treated as fp32 denormal-range values); a NON-FINITE operand gives NaN where the fp32 kernels give
 *     +-inf (inf - inf inside the cut).
 *   PYGAT_GEMM_FP32_MFMA: v_mfma_f32_32x32x2_f32 throughout.
 *   PYGAT_GEMM_DEFAULT: the process default -- split-bf16, or fp32 MFMA with PYGAT_GEMM_F32=1 in the environment
 *     (read once when the library is loaded; pygat_default_gemm_mode() reports it).
 * The mode is an ARGUMENT (`gemm_mode`) of every entry point that runs a GEMM: the library holds no mutable state
 * besides the thread-local error string, so two threads / two models may use different modes at the same time. */
enum { PYGAT_GEMM_DEFAULT = -1, PYGAT_GEMM_SPLIT_BF16 = 0, PYGAT_GEMM_FP32_MFMA = 1 };
int pygat_default_gemm_mode(void);

/* ------------------------------------------------------------------ K0: graph
 * Replaces `adj.nonzero().t()` (layers.py:129), run ONCE per graph instead of
 * once per head per forward, and the `adj > 0` mask of layers.py:41.
 * mode 0: pattern = (adj != 0) (sparse layer), mode 1: pattern = (adj > 0) (dense layer).
 */
/* counts[i] = nnz of row i of the dense n x n matrix (ld = row stride in floats). */
int pygat_dense_row_counts(const float* adj, int n, int64_t ld, int mode,
                           int32_t* counts, void* stream);
/* exclusive scan: out[0] = 0, out[i+1] = sum(in[0..i]); in/out length n / n+1.
 * ws: >= pygat_scan_workspace_bytes(n) bytes. */
size_t pygat_scan_workspace_bytes(int64_t n);
int pygat_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, void* ws, void* stream);
/* col[rowptr[i] ...] = sorted column indices of row i's pattern. */
int pygat_dense_fill_cols(const float* adj, int n, int64_t ld, int mode,
                          const int32_t* rowptr, int32_t* col, void* stream);
/* For a structurally symmetric CSR with sorted rows: perm[k] = position of edge
 * (j,i) for the edge k = (i,j); flags[0] is set to 1 if some edge has no mirror
 * (pattern not symmetric), flags[1] to 1 if some row is empty. flags must be zeroed. */
int pygat_csr_symmetric_perm(int n, const int32_t* rowptr, const int32_t* col,
                             int32_t* perm, int32_t* flags, void* stream);

/* ------------------------------------------------------------ K1: projection
 * Replaces torch.mm(h, W) per head (layers.py:35,134), the skip torch.mm
 * (layers.py:48,166) and the two a-halves matmuls (layers.py:60-61):
 *   C[M x N] = op(A) * op(B) (+ C if accumulate), fp32 MFMA, row-major.
 * Output columns can be routed to up to PYGAT_MAX_SEGMENTS destination tables (seg_*): columns
 * [seg_col[s], seg_col[s+1]) of C go to seg_ptr[s] with row stride seg_ld[s] (the projection's Wh | Sk | s
 * tables; the per-rank column blocks of a head-parallel level's input gradient, pygat_amd/dist.py).
 */
#define PYGAT_MAX_SEGMENTS 4
typedef struct {
  int nseg;                                       /* 1..PYGAT_MAX_SEGMENTS */
  int32_t col_start[PYGAT_MAX_SEGMENTS + 1];      /* col_start[0] = 0, col_start[nseg] = N */
  float* ptr[PYGAT_MAX_SEGMENTS];
  int64_t ld[PYGAT_MAX_SEGMENTS];
} pygat_out_segments;

/* transA: A is stored [K x M]; transB: B is stored [N x K]. split_k >= 1: when >1 the
 * K range is cut in split_k slabs whose partial products go to `ws`
 * (>= pygat_gemm_workspace_bytes) and are then summed in slab order (deterministic). */
size_t pygat_gemm_workspace_bytes(int M, int N, int split_k);
int pygat_gemm_f32(int transA, int transB, int M, int N, int64_t K,
                   const float* A, int64_t lda, const float* B, int64_t ldb,
                   const pygat_out_segments* out, int accumulate,
                   int split_k, void* ws, int gemm_mode, void* stream);

/* Column-blocked matrices (ABI 14): the activation of a head-parallel hidden level as the exchange over xGMI delivers it
 * (pygat_amd/dist.py; replaces torch.cat([...], dim=1), models.py:32, WITHOUT the concatenating copy).  A logical
 * [rows x cols] row-major matrix is stored as cols / w BLOCKS of [rows x w], block b (= rank b's heads, all rows) at
 * base + b * stride floats, row stride ld inside a block:
 *      element (r, c)  at  base + (c / w) * stride + r * ld + (c % w)
 * w: a power of two >= 16 dividing cols (w = 0, or a NULL descriptor: an ordinary matrix); stride, ld, base: multiples
 * of 4 floats / 16-byte aligned.  `a_blk` describes the STORED matrix of operand A along its contiguous dimension (the
 * K columns of a plain A, the M columns of a transA one: in both cases X [nodes x features] blocked by features);
 * `c_blk` the output C along its N columns (then `out` has ONE segment: ptr[0] = base, ld[0] = row stride, and
 * split_k must be 1) -- an input gradient written straight into the layout the reduce-scatter sends.
 * Blocked operands run on the general kernels of either product mode (no streamed fast path takes them). */
typedef struct {
  int w;             /* columns per block; 0 = not blocked */
  int64_t stride;    /* floats between block b and block b + 1 */
} pygat_col_blocks;
int pygat_gemm_f32_blocked(int transA, int transB, int M, int N, int64_t K,
                           const float* A, int64_t lda, const pygat_col_blocks* a_blk, const float* B, int64_t ldb,
                           const pygat_out_segments* out, const pygat_col_blocks* c_blk, int accumulate,
                           int split_k, void* ws, int gemm_mode, void* stream);

/* Pack the per-head parameters of one level into the projection operand
 *   Wcat [Fin x ldw], columns: [0,R) W heads (padded to Fp), [R,2R) skip heads if
 *   w_skip != NULL, then H columns W_h a_src_h and H columns W_h a_dst_h (so the
 *   projection also yields s and t), zero padded up to ldw (multiple of 4);
 *   a_pad [H x 2 x Fp]: a_src then a_dst per head, zero padded.
 * W [H x Fin x F'], a [H x 2F'] (a[:F'] = a_src multiplies Wh_i, layers.py:60),
 * w_skip [H x Fin x F'] or NULL. */
int pygat_pack_params(int H, int Fin, int Fo, const float* W, const float* a,
                      const float* w_skip, float* Wcat, int64_t ldw, float* a_pad,
                      void* stream);
/* The same with the parameters where the reference keeps them: one W [Fin x F'], a (2F' values) and skip_projection
 * [Fin x F'] tensor PER HEAD module (layers.py:21-28,111-119; models.py:15-27).  W, a, w_skip (or NULL): HOST arrays of H
 * device pointers, H <= PYGAT_MAX_HEADS_TABLE (they travel as kernel arguments) -- no stacking copy before the level. */
#define PYGAT_MAX_HEADS_TABLE 16
int pygat_pack_params_heads(int H, int Fin, int Fo, const float* const* W, const float* const* a,
                            const float* const* w_skip, float* Wcat, int64_t ldw, float* a_pad, void* stream);
/* torch.stack of the per-head parameters in ONE launch: W_out [H x nW], a_out [H x nA], skip_out [H x nS] (or NULL) from host
 * arrays of H device pointers (the level flavours that take stacked parameters: dropout, GATv2). */
int pygat_stack_heads(int H, int64_t nW, int nA, int64_t nS, const float* const* W, const float* const* a,
                      const float* const* w_skip, float* W_out, float* a_out, float* skip_out, void* stream);
/* The same with output blocks of nW_out >= nW (nS_out >= nS) elements per head, the tail zero: [Fin, F'] weights with zero
 * rows appended, for a level whose Fin is not a multiple of 16 run on zero-padded input columns (PPI: 50 -> 64; the split-
 * bf16 kernels take K in steps of 16, the fp32 fallback ran that level's three products at a sixth of their speed). */
int pygat_stack_heads_padded(int H, int64_t nW, int64_t nW_out, int nA, int64_t nS, int64_t nS_out, const float* const* W,
                             const float* const* a, const float* const* w_skip, float* W_out, float* a_out, float* skip_out,
                             void* stream);
/* Projection of one level in one GEMM: [Wh | Sk | s] = X * Wcat[:, :R (+R) + H]; the H columns behind the
 * heads are W_h a_src_h (pygat_pack_params), so s_i = Wh_i . a_src (layers.py:60) comes out of the same pass.
 * a_pad (pygat_pack_params; may be NULL): lets the kernel form s from the Wh accumulators themselves -- the
 * reference's own order, Wh_i . a_src -- where a head is 8 or 16 columns wide and Fin is 64 or 128.
 * Sk may be NULL (no skip).  split_k / ws as pygat_gemm_f32. */
int pygat_project(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const float* Wcat, int64_t ldw, const float* a_pad,
                  float* Wh, float* Sk, float* s, int split_k, void* ws, int gemm_mode, void* stream);
/* The same with X column-blocked (x_blk as in pygat_gemm_f32_blocked; NULL / w = 0: pygat_project). */
int pygat_project_blocked(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const pygat_col_blocks* x_blk,
                          const float* Wcat, int64_t ldw, const float* a_pad, float* Wh, float* Sk, float* s, int split_k,
                          void* ws, int gemm_mode, void* stream);
/* s[n x H], t[n x H] (t may be NULL) from the masked Wh table: s_ih = Wh_ih . a_src_h, t_ih = Wh_ih . a_dst_h
 * (layers.py:60-61 after the Wh dropout of layers.py:37,136).  wh_mask [n x R] (pre-scaled) non-NULL: that dropout is
 * applied HERE, in place (Wh *= wh_mask), instead of by a launch of its own; NULL: Wh is taken as it is.
 * a_pad as written by pygat_pack_params. */
int pygat_attn_scores(int n, int H, int Fo, float* Wh, const float* wh_mask, const float* a_pad,
                      float* s, float* t, void* stream);
/* Inverse for gradients: dW[H x Fin x F'] (+)= columns of dWcat [Fin x ld]. */
int pygat_unpack_wgrad(int H, int Fin, int Fo, const float* dWcat, int64_t ld,
                       int col_offset, float* dW, void* stream);

/* ---------------------------------------------- K2: fused edge-softmax + aggregate
 * Replaces layers.py:141-170 (and 40-51): per row i and head h
 *   e_ij = LeakyReLU(s_i + t_j), m_i = max_j e_ij, p_ij = exp(e_ij - m_i),
 *   Z_i = sum_j p_ij, hattn_i = (sum_j p_ij Wh_j) / Z_i,
 *   out_i = [ELU](hattn_i [+ sk_i]).
 * Work is split by EDGES, not rows ("nnz split"): the CSR edge list is cut into
 * slots of `slot_edges` consecutive edges; a slot walks its edges with the
 * online-softmax recurrence and finishes every row that lies wholly inside it.
 * Rows cut by a slot border leave partial (m, Z, acc) records in `part`, merged in
 * slot order by a second small launch (deterministic, no atomics).
 */
typedef struct {
  int n;                     /* nodes */
  int64_t nnz;               /* edges */
  const int32_t* rowptr;     /* [n+1] */
  const int32_t* edge_rc;    /* [nnz][2]: (row i, column j) of every edge, CSR order (pygat_edge_pairs) */
  int slot_edges;            /* nominal edges per slot: multiple of 4, >= 4 */
  const int32_t* slot_begin; /* NULL: slot k = edges [k*slot_edges, (k+1)*slot_edges); else [n_slots+1]
                                row-snapped borders from pygat_slot_bounds (fewer rows are cut) */
  const int32_t* cut_rows;   /* NULL, or [n_cut][3] = (first slot k, row, number of pieces) of every row that a
                                slot border cuts, sorted by pieces descending: lets the fix-up launches go
                                straight to the cut rows instead of screening every slot */
  int n_cut;
  int n_cut_wide;            /* leading entries with more than 32 pieces (merged by a whole work-group) */
  /* pygat_gat_forward only: work on the slots [slot_first, slot_first + slot_count) -- a range of whole rows (its
     first slot starts a row, the slot after its last one too); cut_rows then lists the cut rows of that range alone.
     slot_count = 0: all slots.  Lets a caller pipeline row chunks of a level (pygat_amd/dist.py: chunk k's head
     outputs travel over xGMI while chunk k+1 is computed).  Every other entry point needs 0, 0. */
  int64_t slot_first;
  int64_t slot_count;
  /* NULL, or [n_slots][4] = (first edge, end edge, row of the first edge, flags) of every slot, flags bit 0: the slot's
     first row began in an earlier slot, bit 1: its last row continues in a later one (pygat_slot_meta).  One 16-byte load
     then replaces the slot's start-up chain slot_begin -> edge_rc -> rowptr and the rowptr load at its end: what a slot
     costs in DEPENDENT memory round trips is what bounds the attention kernels on small graphs and on narrow rows. */
  const int32_t* slot_meta;
  /* NULL, or [n_slots] a permutation of the slot ids (ABI 14): the MAIN launches of pygat_gat_forward and
     pygat_gat_backward_col hand slot slot_order[q] to grid position q (work-groups are dispatched in grid order, eight
     lane groups of consecutive q each), i.e. the ORDER in which slots are walked -- and with it which rows' gathers are in
     flight together -- without renumbering a node or moving a table row.  Results do not depend on it (partial records
     and fix-up launches go by the slot id).  Needs slot_count = 0. */
  const int32_t* slot_order;
  /* NULL, or [n] (ABI 14): this pattern is an INTERNAL renumbering of the caller's graph (pygat_amd/graph.py, degree order) and
     user_row[i] is the caller's row of internal node i.  Every node table the kernels exchange among themselves is in internal
     order; the caller-facing ones are addressed through this map: pygat_gat_forward writes row user_row[i] of `out`
     (pygat_gat_backward_prepare takes the same map as an argument for G and the saved output). */
  const int32_t* user_row;
} pygat_graph;

/* edge_rc[k] = (i, col[k]) for rowptr[i] <= k < rowptr[i+1] */
int pygat_edge_pairs(int n, const int32_t* rowptr, const int32_t* col, int32_t* edge_rc, void* stream);

/* slot_begin[k] = k*slot_edges, moved forward to the next row start when that is less than
 * slot_edges/2 edges away (so most slot borders fall between rows and only rows longer than
 * slot_edges/2 are ever cut); slot_begin[n_slots] = nnz, n_slots = ceil(nnz / slot_edges). */
int pygat_slot_bounds(int n, int64_t nnz, const int32_t* rowptr, const int32_t* edge_rc, int slot_edges,
                      int32_t* slot_begin, void* stream);
/* slot_meta[k] = (first edge, end edge, row of the first edge, flags) for the slots of `slot_begin` (NULL: uniform
 * slots of slot_edges) -- see pygat_graph.slot_meta.  slot_meta: [n_slots][4] int32, 16-byte aligned. */
int pygat_slot_meta(int n, int64_t nnz, const int32_t* rowptr, const int32_t* edge_rc, int slot_edges,
                    const int32_t* slot_begin, int32_t* slot_meta, void* stream);

/* bytes of `part` workspace needed by forward / column backward for this graph and row width */
size_t pygat_partials_bytes(int64_t nnz, int slot_edges, int H, int Fp);

/* Heads per backward kernel pass ("head window").  The attention entry points take the level's
 * full-width tables.  The forward walks rows of up to 1024 floats per pass; the two backward passes of
 * a LARGE graph (gathered table beyond the 256 MiB Infinity Cache) with rows wider than 512 floats walk
 * the heads in windows of at most 256 floats (one 16-byte chunk per lane of a wave64): window w covers
 * heads [w*hg, min(H, (w+1)*hg)), hg = pygat_head_group(n, H, F').  Callers only need it to find Gp
 * inside GR (see K3a below).  Returns H when the backward takes the whole row in one pass, 0 on bad
 * arguments.  It is a pure function of its arguments and only the DEFAULT: prepare / row / col take `head_group`
 * (heads per window; 0 = this default; any value whose windows are at most 1024 floats wide) as an argument, and
 * the caller passes the SAME value to all three and lays GR out accordingly.  (Until ABI 12 an environment
 * variable read on every call moved the threshold: process state deciding a layout the caller allocates.) */
int pygat_head_group(int n, int H, int Fo);

/* Wh [n x R], s,t [n x H], sk [n x R] or NULL.
 * out [n x H*F'] compact (may be NULL), hattn [n x R] padded (may be NULL; needed for the
 * head mean), m,Z [n x H] (may be NULL together in eval).
 * aneg [n x R], qneg [n x H] (NULL together; need m,Z): the training forward also leaves the share of each row
 * sum that went through the alpha branch of the LeakyReLU,
 *     aneg_i = sum_{j: s_i+t_j <= 0} alpha_ij mask_ij Wh_j,     qneg_i = sum_{j: s_i+t_j <= 0} alpha_ij,
 * from which pygat_gat_backward_prepare takes the row sums ds_i = sum_j dz_ij without touching an edge again
 * (sum_j de_ij = 0, so ds_i = -(1 - alpha)(Gp_i . aneg_i - D_i qneg_i)). */
int pygat_gat_forward_phases_ok(int n, int H, int Fo);   /* 1: PYGAT_F_MAIN_ONLY / PYGAT_F_FIXUP_ONLY may be used */
int pygat_gat_forward(const pygat_graph* g, int H, int Fo, float alpha, int flags,
                      const float* Wh, const float* s, const float* a_pad, const float* sk,
                      const float* att_mask,
                      float* out, float* hattn, float* m, float* Z, float* aneg, float* qneg,
                      void* part, void* stream);

/* models.py:34: out[n x F'] = mean over heads of (hattn [+ sk]) (padded inputs). */
int pygat_head_mean(int n, int H, int Fo, const float* hattn, const float* sk,
                    float* out, void* stream);

/* ------------------------------------------------ K3/K4: backward, no N x N, no atomics
 * Replaces SpecialSpmmFunction.backward (layers.py:81-90) and the autograd of
 * layers.py:141-170.
 *   K3a prepare (per row):  Gp_i = G_i * ELU'(.), D_i = Gp_i . hattn_i,
 *        GR_i = [ Gp_i (R floats) | (s_i, m_i, 1/Z_i, D_i) per head (4H floats) ]   -> GR [n x (R+4H)]
 *        With more than one head window (pygat_head_group(n,H,F') < H) the row is laid out window by
 *        window, each [ Gp of its heads | their 4-float records ], window w starting at float
 *        w*hg*(Fp+4): head h's Gp slice is at  (h/hg)*hg*(Fp+4) + (h%hg)*Fp.
 *        mean_mode 0: G is [n x H*F'] and y is the forward OUTPUT out [n x H*F'] (hattn is
 *                     recovered from it: out > 0 ? out : log1p(out), minus sk);
 *        mean_mode 1: G is [n x F'] (every head receives G/H, models.py:34), y is hattn [n x R].
 *   K3b row pass (nnz split over the forward pattern g): per edge (i,j)
 *        alpha_ij = exp(LeakyReLU(s_i + t_j) - m_i)/Z_i, dz_ij = alpha_ij (mask_ij Gp_i.Wh_j - D_i) LeakyReLU'(.)
 *        ds_i = sum_j dz_ij            (GR_i row-local, Wh_j gathered; the layers.py:85 entry, per edge)
 *   K4 column pass (nnz split over the TRANSPOSED pattern gT, any pattern, symmetric or not):
 *        alpha_ij, dz_ij recomputed (GR_i gathered as one contiguous row, Wh_j row-local)
 *        dWh_j = sum_i alpha_ij mask_ij Gp_i + ds_j a_src + dt_j a_dst,  dt_j = sum_i dz_ij
 * Nothing per-edge is stored between the passes.  att_mask [nnz x H] (forward edge order) or NULL;
 * perm_t[k] = forward edge of gT's edge k, needed only to index att_mask (may be NULL without mask).
 * pygat_gat_backward_col takes EITHER ds (from the row pass; dz_t = NULL) OR dz_t (ds = NULL; see
 * pygat_gat_backward_rowsum below).
 * part: >= pygat_partials_bytes for both passes.
 *
 * Head range.  The backward entry points (prepare, row, col, rowsum, a_grad, wgrad) end in (h_first, h_count):
 * gradients are produced for the heads [h_first, h_first + h_count) of the level only; 0, 0 = all heads.  The
 * level's tables keep their full width (H heads) and are addressed with the level's strides; only GR, which
 * nothing but the backward touches, is then COMPACT for the range ([n x h_count*(Fp+4)], same layout rules with
 * h_count in place of H), and dW / da rows outside the range are left untouched.  Use: a rank of a head-parallel
 * run that ran the forward for all heads (cheaper than receiving them over one xGMI link) back-propagates its own.
 */
/* pygat_gat_backward_prepare: aneg, qneg (from the training forward) and ds non-NULL together: K3a also writes
 * ds [n x H] (LeakyReLU slope `alpha`), after which pygat_gat_backward_col is called WITH ds and neither the row
 * pass nor the row-sum pass runs -- the default backward.  All three NULL: GR only. */
int pygat_gat_backward_prepare(int n, int H, int Fo, int flags, int mean_mode,
                               const float* G, const float* y, const float* sk,
                               const float* s, const float* m, const float* Z,
                               float* GR, const float* aneg, const float* qneg, float alpha, float* ds,
                               int h_first, int h_count, int head_group, const int32_t* user_row, void* stream);
/* user_row (ABI 14; NULL = identity): the tables are in the INTERNAL node order of a renumbered pattern
 * (pygat_graph.user_row): G -- and, in concat mode, y = the level's saved output -- are the caller's arrays and are read at
 * row user_row[i] for internal node i; every other table is internal. */
int pygat_gat_backward_row(const pygat_graph* g, int H, int Fo, float alpha,
                           const float* Wh, const float* a_pad, const float* GR,
                           const float* att_mask, float* ds, void* part, int h_first, int h_count, int head_group,
                           void* stream);
/* da_part (or NULL): the column pass also takes the attention-vector gradient along (autograd of layers.py:60-61 /
 * a.mm(edge_h) layers.py:144): where row j finishes -- dt_j just formed, ds_j loaded for the dWh term, Wh_j in L1 from the
 * row's edges -- every lane adds ds_j Wh_j and dt_j Wh_j into two float4s of its own in LDS, and each work-group leaves ONE
 * record [2 R] = (sum ds_j Wh_j | sum dt_j Wh_j) over the rows it finished, lane groups added in a fixed order.  Rows cut by a
 * slot border are finished by the fix-up launch and are NOT in the records: pygat_a_grad_fold adds them from the cut-row
 * list.  Needs ds (row sums known), a level of 8 heads x 16 in one window (the instantiation that keeps four waves per
 * SIMD with the sums; the others lose a wave to them and more time than the a-gradient pass costs) and a graph with a cut-row
 * list: pygat_gat_backward_col_da_bytes returns the size of da_part, 0 when the pass does not do it (then pygat_a_grad).
 * What it replaces: a launch that streams Wh, ds and dt again (0.6 GB, 0.12 ms at config 5). */
size_t pygat_gat_backward_col_da_bytes(const pygat_graph* gT, int H, int Fo, int head_group);
int pygat_gat_backward_col(const pygat_graph* gT, const int32_t* perm_t, int H, int Fo, float alpha,
                           const float* Wh, const float* a_pad, const float* GR,
                           const float* att_mask, const float* ds,
                           float* dWh, float* dt, float* dz_t, void* part, float* da_part, int h_first, int h_count,
                           int head_group, void* stream);
/* da [H x 2F'] from the records of a column pass that ran with da_part, plus the rows of gT's cut-row list (their ds, dt
 * and Wh rows are read here).  Same gT / head_group as that pass; ws >= pygat_agrad_workspace_bytes(H, F').  Fixed
 * summation order, no atomics: bitwise reproducible. */
int pygat_a_grad_fold(const pygat_graph* gT, int H, int Fo, const float* Wh, const float* ds, const float* dt,
                      const float* da_part, float* da, void* ws, int head_group, void* stream);
/* Row sums without the row pass.  The column pass computes every dz_ij anyway (for dt_j): called with ds = NULL
 * and dz_t [nnz x H] it writes them out per TRANSPOSED edge and leaves the ds_j a_src term out of dWh_j; then
 *   pygat_gat_backward_rowsum   ds_i = sum over the forward edges k of row i of dz_t[perm_f[k]]
 * replaces pygat_gat_backward_row: 4H-byte records per edge instead of a gathered Wh row (a quarter less HBM
 * traffic for the whole backward at 8 heads x 16), fixed summation order, still no atomics; and pygat_a_grad,
 * which streams Wh, ds, dt anyway, finishes dWh_i += ds_i a_src when given dWh.
 * g = forward pattern, perm_f[k] = position of forward edge k in gT.  part >= pygat_partials_bytes. */
int pygat_gat_backward_rowsum(const pygat_graph* g, const int32_t* perm_f, int H, int Fo, const float* dz_t,
                              float* ds, void* part, int h_first, int h_count, void* stream);
/* da[H x 2F'] : da_src = sum_i ds_i Wh_i, da_dst = sum_j dt_j Wh_j (per head).
 * ws >= pygat_agrad_workspace_bytes(H, Fo).  dWh (with a_pad) non-NULL: also dWh_i += ds_i a_src, see above. */
size_t pygat_agrad_workspace_bytes(int H, int Fo);
/* dwh_mask [n x R] (pre-scaled; needs dWh): dWh is also taken back through the Wh dropout of layers.py:37,136 in the same
 * pass (after the finishing term), instead of by a launch of its own. */
int pygat_a_grad(int n, int H, int Fo, const float* Wh, const float* ds, const float* dt,
                 float* da, void* ws, const float* a_pad, float* dWh, const float* dwh_mask, int h_first, int h_count,
                 void* stream);

/* Weight gradient of one level, the backward counterpart of pygat_project (autograd of layers.py:35,134):
 *   dW[h] = X^T dWh[:, head h]                       dW [H x Fin x F'], X [n x Fin], dWh [n x R]
 * With ds [n x H] (and a_pad) the dWh passed in still lacks its ds_i a_src term (pygat_gat_backward_col with
 * dz_t): X^T(dWh' + ds (x) a_src) = X^T dWh' + (X^T ds) (x) a_src, so ds rides along as H extra columns of the
 * same GEMM and the rank-1 terms are added while unpacking -- dWh is not read-modified-written for it (then
 * pygat_a_grad is called WITHOUT dWh).  ds = NULL: plain X^T dWh.
 * ws >= pygat_wgrad_workspace_bytes(Fin, H, F', split_k); split_k as in pygat_gemm_f32. */
size_t pygat_wgrad_workspace_bytes(int Fin, int H, int Fo, int split_k);
int pygat_wgrad(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const float* dWh, const float* ds,
                const float* a_pad, float* dW, int split_k, void* ws, int h_first, int h_count, int gemm_mode, void* stream);

/* The same with X column-blocked (pygat_col_blocks; NULL / w = 0: pygat_wgrad). */
int pygat_wgrad_blocked(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const pygat_col_blocks* x_blk,
                        const float* dWh, const float* ds, const float* a_pad, float* dW, int split_k, void* ws,
                        int h_first, int h_count, int gemm_mode, void* stream);

/* Self-loop-only nodes (ABI 14; csrc/k12_tail.hip).  A node whose only neighbour is itself has alpha_ii = 1: its forward is
 * h'_i = ELU(Wh_i (+ skip_i)), its backward dWh_i = Gp_i, ds_i = dt_i = 0 (layers.py:146-170 / 81-90 with one edge).  In a
 * degree-ordered pattern (pygat_graph.user_row != NULL) these nodes are the rows [row_first, row_first + n_rows) at the END of the
 * row range and their slots a suffix of the slot list: the caller runs pygat_gat_forward / pygat_gat_backward_col /
 * pygat_a_grad_fold on the slot PREFIX before them (pygat_graph.slot_first = 0, slot_count = the first tail slot; same struct for
 * all three) and these two streams on the tail.  Concat levels (out only), GR in ONE head window ([Gp R | records 4H] per row).
 * forward_tail also writes m = 0, Z = 1 (and qneg = 0) for the tail rows when given, so that pygat_gat_backward_prepare finds
 * finite records there. */
int pygat_gat_forward_tail(int row_first, int n_rows, int H, int Fo, int flags, const float* Wh, int64_t ldwh, const float* sk,
                           float* out, const int32_t* user_row, float* m, float* Z, float* qneg, void* stream);
/* (ldwh: row stride of Wh in floats, 0 = H*Fp; GATv2 passes its [Whi | Whj] table with 2 H Fp: h'_i = ELU(Whi_i (+ skip_i)),
 *  layers.py:296 with one edge) */
int pygat_gat_backward_col_tail(int row_first, int n_rows, int H, int Fo, const float* GR, float* dWh, float* dt, void* stream);
/* The tail's whole backward in one stream, for levels WITHOUT a skip projection (nothing else reads the tail's Gp): dWh_j =
 * G_u ELU'(out_u) with u = user_row[j], ds_j = dt_j = 0 (ds may be NULL); pygat_gat_backward_prepare is then called with
 * n = row_first (the rows before the tail) and the tail's rows of GR stay untouched.  Concat levels (y = the saved output). */
int pygat_gat_backward_tail(int row_first, int n_rows, int H, int Fo, int flags, const float* G, const float* y,
                            const int32_t* user_row, float* dWh, int64_t ld_dwh, int zero_cols, float* ds, float* dt, void* stream);
/* (ld_dwh: row stride of dWh, 0 = H*Fp; zero_cols: columns behind the H*Fp gradient columns of a row to clear -- GATv2: dWW_i =
 *  [Gp_i | 0] with ld_dwh = 2 H Fp, zero_cols = H Fp; ds, dt may be NULL) */

/* ------------------------------------------------ GATv2 (next row of the scope table)
 * The reference's SpGraphAttentionLayerV2 (layers.py:258-313): per head
 *   e_ij = a . LeakyReLU(Whi_i + Whj_j), alpha = row softmax, h'_i = sum_j alpha_ij Whi_j  (Whi is aggregated,
 *   layers.py:296).  WW [n x 2R] holds [Whi | Whj] rows (one projection GEMM with W[:Fin] and W[Fin:]),
 *   a2 [H x Fp] the zero-padded attention vectors.  Same outputs / flags / part as pygat_gat_forward.
 * (GraphAttentionLayerV2, layers.py:204-230, broadcasts one logit per ROW and is therefore a neighbour mean:
 *  it maps onto pygat_gat_forward with s = 0 and a = 0.)
 */
int pygat_gatv2_forward(const pygat_graph* g, int H, int Fo, float alpha, int flags,
                        const float* WW, const float* a2, const float* sk, const float* att_mask,
                        float* out, float* hattn, float* m, float* Z, void* part, void* stream);
/* GRW [n x (2R+4H)] = [Gp | (., m, 1/Z, D) per head | Whi]; G, y, sk, mean_mode as pygat_gat_backward_prepare. */
int pygat_gatv2_backward_prepare(int n, int H, int Fo, int flags, int mean_mode,
                                 const float* G, const float* y, const float* sk,
                                 const float* m, const float* Z, const float* WW, float* GRW, const int32_t* user_row,
                                 void* stream);   /* user_row: as in pygat_gat_backward_prepare */
/* Column pass over gT, then row pass over g:
 *   dWW [n x 2R] = [dWhi | dWhj], da [H x F'].  perm_t (transposed position -> forward edge) only indexes att_mask and may
 *   be NULL without one; perm_f (forward edge -> transposed position; the same array for a symmetric pattern) lets the row
 *   pass fetch the de_ij the column pass left per transposed edge.  ws >= pygat_gatv2_workspace_bytes. */
size_t pygat_gatv2_workspace_bytes(int64_t nnz, int slot_edges, int H, int Fo);
int pygat_gatv2_backward(const pygat_graph* g, const pygat_graph* gT, const int32_t* perm_t, const int32_t* perm_f,
                         int H, int Fo, float alpha, const float* WW, const float* a2, const float* GRW,
                         const float* att_mask, float* dWW, float* da, void* ws, void* stream);

/* ------------------------------------------------ K7: train-mode dropout around the projection
 * The reference drops out inside every head, each head with its own masks (models.py:32,34 call the heads
 * one after another): the input (layers.py:34,132), Wh (layers.py:37,136), the attention (layers.py:43,153).
 * Masks are pre-scaled keep masks (0 or 1/(1-p)).  Either the caller passes one explicitly (`mask`,
 * [H x n x Fin]; tests) or it is drawn in-kernel: Philox-4x32-10, key = *seed (a uint64 in DEVICE memory, so
 * a captured HIP graph replays with fresh masks when the caller refreshes it), `stream_id` separating the
 * different masks drawn from one seed.
 *   pygat_dropout_mask      out[e] = mask, e < count                          (Wh and attention masks)
 *   pygat_dropout_expand    A'[i, h*Fin + k] = x[i,k] * m_h[i,k]   -> A' [n x H*Fin] (ldo >= H*Fin)
 *   pygat_pack_blockdiag    B'[h*Fin + k, :] = [ W_h[k,:] in head h's padded columns | same for Wskip ]
 *                           so that A' B' = [Wh | Sk] for all heads in ONE GEMM (K = H*Fin)
 *   pygat_unpack_blockdiag  dW[h,k,f] = dB'[h*Fin + k, col_offset + h*Fp + f]  (diagonal blocks of A'^T dWh)
 *   pygat_dropout_head_sum  dx[i,k] (+)= sum_h m_h[i,k] dxe[i, h*Fin + k]      (dxe = dWh B'^T; same mask/seed/
 *                           stream_id as the expand call)
 */
int pygat_dropout_mask(int64_t count, float p, const void* seed, uint32_t stream_id, float* out, void* stream);
/* two of them (a level's Wh and attention masks) in one launch */
int pygat_dropout_mask2(float p, const void* seed, int64_t count1, uint32_t stream1, float* out1, int64_t count2,
                        uint32_t stream2, float* out2, void* stream);
int pygat_dropout_expand(int n, int Fin, int H, const float* x, int64_t ldx, const float* mask, float p,
                         const void* seed, uint32_t stream_id, float* out, int64_t ldo, void* stream);
int pygat_dropout_head_sum(int n, int Fin, int H, const float* dxe, int64_t lde, const float* mask, float p,
                           const void* seed, uint32_t stream_id, float* dx, int64_t ldx, int accumulate,
                           void* stream);
int pygat_pack_blockdiag(int H, int Fin, int Fo, const float* W, const float* w_skip, float* Bp, int64_t ldb,
                         void* stream);
int pygat_unpack_blockdiag(int H, int Fin, int Fo, const float* dBp, int64_t ldb, int col_offset, float* dW,
                           void* stream);

/* Per-head input dropout WITHOUT the wide operand (default since ABI 9 where supported: H <= 8 and
 * (skip ? 2 : 1) * Fp <= 256, pygat_headmask_supported).  The decisions of all heads for x[i,k] are one byte,
 *   pygat_dropout_bits        bits[i,k] bit h = head h keeps x[i,k]   (word h & 3 of Philox counter (k, i, stream_id, h >> 2))
 * and the projection and its weight gradient run for ALL heads in one launch with X read once -- the A tile is staged
 * in LDS with its mask bytes and the heads are an inner loop over the MFMA fragments (a_h = bit_h ? x : 0):
 *   pygat_project_dropout     [Wh | Sk] = 1/(1-p) (X .* m_h) [W_h | Wskip_h]           Wcat from pygat_pack_params;
 *                             split_k K slabs over Fin for graphs with few 128-row tiles
 *   pygat_wgrad_dropout       dWc [Fin x R (+R)] = 1/(1-p) (X .* m_h)^T [dWh_h | Gp_h]  (then pygat_unpack_wgrad);
 *                             split_k K slabs over the nodes: a slab is ONE fp32 accumulator chain per output, so give
 *                             slabs of a few hundred rows (pygat_amd/dropout.py: >= 64 rows until the chip is full) when
 *                             the result should match a blocked CPU product to its last bits
 *   pygat_dropout_head_sum_bits  dx[i,k] (+)= 1/(1-p) sum_h bit_h[i,k] dxe[i, h*Fin + k]
 * Memory: N*Fin bytes instead of N*H*Fin*4 (Citeseer: 12 MB instead of 394 MB); no products with zero blocks.
 * X must be dense (ldx == Fin).  Explicit masks (tests) are packed into the same bytes by the caller. */
int pygat_headmask_supported(int H, int Fo, int skip);
int pygat_dropout_bits(int n, int Fin, int H, float p, const void* seed, int stream_id, unsigned char* bits, void* stream);
size_t pygat_project_dropout_workspace_bytes(int n, int H, int Fo, int skip, int split_k);
int pygat_project_dropout(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const unsigned char* bits, float p,
                          const float* Wcat, int64_t ldw, float* Wh, float* Sk, int split_k, void* ws, void* stream);
size_t pygat_wgrad_dropout_workspace_bytes(int Fin, int H, int Fo, int skip, int split_k);
int pygat_wgrad_dropout(int n, int Fin, int H, int Fo, const float* X, int64_t ldx, const unsigned char* bits, float p,
                        const float* dWh, const float* Gp, int64_t ldgp, float* dWc, int split_k, void* ws, void* stream);
int pygat_dropout_head_sum_bits(int n, int Fin, int H, const float* dxe, int64_t lde, const unsigned char* bits, float p,
                                float* dx, int64_t ldx, int accumulate, void* stream);
/* NARROW levels (Fin <= 128, H <= 8, (skip ? 2 : 1) H F'p <= 128 -- the second level of the citation models: 64 inputs,
 * 1 x 7 or 8 x 3 outputs; pygat_dropout_narrow says whether a shape qualifies): pygat_project_dropout and
 * pygat_wgrad_dropout then run on the vector ALUs with the weight table in registers (csrc/k10_narrow.hip) -- split_k of
 * the projection is ignored, split_k of the weight gradient is the number of row slabs (one wave each; a few rows per slab
 * on a small graph, 8-32 on a large one) and its workspace is always used -- and the gradient into the level's input is one
 * launch instead of a GEMM per head and a fold:
 *   pygat_dx_dropout   dx[i,k] (+)= 1/(1-p) sum_h bit_h(i,k) ( dWh_h[i,:] . W_h[k,:] + Gp_h[i,:] . Wskip_h[k,:] )
 * dWh [n x R], Gp rows of stride ldgp or NULL, Wcat from pygat_pack_params.  PYGAT_NARROW=0 (read when the library loads)
 * switches the narrow kernels off. */
int pygat_dropout_narrow(int Fin, int H, int Fo, int skip);
int pygat_dx_dropout(int n, int Fin, int H, int Fo, const float* dWh, const float* Gp, int64_t ldgp, const unsigned char* bits,
                     float p, const float* Wcat, int64_t ldw, float* dx, int64_t lddx, int accumulate, void* stream);

/* ------------------------------------------------ the citation scripts' training loss (next row 8(f)-2: fused epoch)
 * train.py:151-152,159: loss = nll_loss(log_softmax(elu(out), dim=1)[idx], labels[idx]), forward and backward as ONE
 * launch each (ATen: 17).  The index set is given as per-row weights: weight[r] = (multiplicity of r in idx) / len(idx),
 * label[r] the class of row r (any value where weight[r] == 0).
 *   forward:  loss[0] = sum_r weight[r] * -log_softmax(elu(out[r, :]))[label[r]]     (deterministic two-stage sum inside
 *             the launch; ws >= pygat_nll_workspace_bytes(n), ZEROED once by the caller, left zeroed by every launch)
 *   backward: dout[r, c] = gscale[0] * d loss / d out[r, c]   (every row written; gscale = the upstream gradient, on the device) */
size_t pygat_nll_workspace_bytes(int n);
int pygat_elu_logsoftmax_nll(int n, int C, const float* out, int64_t ldo, const int32_t* label, const float* weight,
                             void* ws, float* loss, void* stream);
int pygat_elu_logsoftmax_nll_backward(int n, int C, const float* out, int64_t ldo, const int32_t* label, const float* weight,
                                      const float* gscale, float* dout, int64_t ldd, void* stream);

/* ------------------------------------------------ sparse input features (level 1 of the citation configurations)
 * The reference densifies its bag-of-words features (utils.py:38-41,60) and multiplies the zeros (Cora: 98.7 % of X).
 * With the non-zero pattern of X extracted once -- CSR (rowptr, col, val) for the projection, its transpose (colptr, row,
 * val, cut into segments) for the weight gradient -- these two replace pygat_project / pygat_project_dropout and pygat_wgrad /
 * pygat_wgrad_dropout for a first level (no gradient into X is formed):
 *   pygat_project_sparse   [Wh | Sk | s] = scale * (X .* m_h) Wcat      Wcat from pygat_pack_params; Sk, s may be NULL
 *   pygat_wgrad_sparse     dW_h = scale * (X .* m_h)^T dWh_h, dWskip_h = ... Gp_h   written as [H x Fin x F'] directly
 * p = 0: no dropout (m = 1, scale = 1).  p > 0: per-head input dropout with the decisions of pygat_dropout_bits -- drawn
 * from (seed, stream_id) per non-zero, or read from explicit `bits` [n x Fin] (H <= 8) -- and scale = 1 / (1 - p); s must
 * be NULL then (scores come from the masked Wh, pygat_attn_scores).  At most 512 output columns (2 R + H). */
int pygat_project_sparse(int n, int Fin, int H, int Fo, const int32_t* rowptr, const int32_t* col, const float* val,
                         const float* Wcat, int64_t ldw, float p, const void* seed, int stream_id, const unsigned char* bits,
                         float* Wh, float* Sk, float* s, void* stream);
/* The transpose is cut into nseg SEGMENTS of at most 128 entries, none crossing a feature column (a frequent word's column is
 * thousands of entries long): segment s covers entries [seg_begin[s], seg_end[s]) of (row, val) -- which are in column
 * order -- and belongs to column seg_col[s]; column k owns the segments [colseg[k], colseg[k+1]) (at least one each, so
 * nseg >= Fin).  One wave per segment, then one per column adds its segments in order.  ws >=
 * pygat_wgrad_sparse_workspace_bytes(nseg, H, F', skip). */
size_t pygat_wgrad_sparse_workspace_bytes(int nseg, int H, int Fo, int skip);
int pygat_wgrad_sparse(int n, int Fin, int H, int Fo, int nseg, const int32_t* colseg, const int32_t* seg_col,
                       const int32_t* seg_begin, const int32_t* seg_end, const int32_t* row, const float* val, float p,
                       const void* seed, int stream_id, const unsigned char* bits, const float* dWh, const float* Gp, int64_t ldg,
                       void* ws, float* dW, float* dWskip, void* stream);

/* train_ppi.py:114,157: nn.BCEWithLogitsLoss(reduction='mean') over `total` logits / targets (fp32, contiguous), one launch
 * each way: loss[0] = mean( max(x,0) - x y + log1p(exp(-|x|)) );  dlogits = gscale[0] / total * (sigmoid(x) - y).
 * ws >= pygat_bce_workspace_bytes(total), zero before the first call (the kernel leaves its counter at zero). */
size_t pygat_bce_workspace_bytes(int64_t total);
int pygat_bce_with_logits(int64_t total, const float* logits, const float* targets, void* ws, float* loss, void* stream);
int pygat_bce_with_logits_backward(int64_t total, const float* logits, const float* targets, const float* gscale, float* dlogits,
                                   void* stream);

/* ---------------------------------------------- K11: optimiser step (csrc/k11_adam.hip)
 * torch.optim.Adam's update (train.py:64-66,122: lr 0.005, weight_decay 5e-4; train_ppi.py:58-60) for up to
 * PYGAT_ADAM_MAX_TENSORS parameter tensors in ONE launch:
 *   g += wd p;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * (hyper-parameters as doubles, like the Python floats torch holds: 1 - beta and the bias corrections are formed in double).
 * params / grads / exp_avg / exp_avg_sq: HOST arrays of ntensors device pointers (fp32, contiguous), numel their sizes.
 * state: PYGAT_ADAM_STATE_BYTES of DEVICE memory, 8-byte aligned ({int32 t, uint32 scratch, double beta1^t, double beta2^t}),
 * zero before the first step; the kernel advances t itself, so a
 * captured HIP graph replays successive steps.  More tensors: several calls must not share one state (each would advance
 * it) -- give every group of 48 its own. */
#define PYGAT_ADAM_MAX_TENSORS 48
#define PYGAT_ADAM_STATE_BYTES 24
int pygat_adam_step(int ntensors, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, double lr, double beta1, double beta2, double eps,
                    double weight_decay, void* state, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYGAT_AMD_H */
