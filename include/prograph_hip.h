/*
 * prograph_hip.h — C ABI of the MI355X (gfx950) graph-construction hot path.
 *
 * The reference (acmater/prograph) has no FFI: its plug-in contract is the Python
 * distance-function protocol `distance(X (N,D), Y (M,D), similarity=False) -> (M,N)`
 * (prograph/distance/hamming.py:8-39, README.md:48) plus the stock torch ops that
 * `Prograph.build_graph` (prograph/prograph.py:656-765) and `Prograph.indexing`
 * (prograph/prograph.py:254-343) run on `cuda:0`.  Each entry point below names the
 * reference call site(s) it replaces; INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors are the
 *     storage container); the library never allocates or frees device memory and never
 *     synchronises.  The all-pairs calls (pg_eps_slots[_sym], pg_eps_fill_rows,
 *     pg_knn_hamming[_round]) take a caller-owned `workspace` of pg_workspace_bytes(nrows)
 *     bytes: launch-private device state (the pass counters of the engine's persistent waves,
 *     the data probe's counts and decision words, and - kNN calls of more than 65 536 rows,
 *     5.2 MB - the partial neighbour lists of rows swept in column pieces; one word per row for
 *     the rows a kNN launch finishes separately),
 *     initialised by the call on `stream`; ONE workspace per launch in flight - a workspace
 *     may be reused once the launch that got it has completed, or by later launches on the
 *     same stream;
 *   - every launch goes to the caller-supplied `stream` (a hipStream_t passed as void*,
 *     NULL = the null stream) on the CURRENT HIP device of the calling thread; calls are
 *     re-entrant per stream and per device;
 *   - return value: 0 = ok, >0 = a hipError_t from the launch, <0 = PG_E_* below;
 *     `pg_last_error()` returns a thread-local human readable message;
 *   - no exceptions cross the boundary.  Process-wide state, all of it host side: the
 *     thread-local error string, the per-device cache of the compute-unit count and of the
 *     engine instances' occupancy, and the run-time binding of RCCL (pg_comm_*).
 *
 * Token storage: bit-sliced records in chunk-major order ("planes").  A sequence of L tokens
 * of `bits` bits each is G = ceil(L/32) groups of `bits` bit-plane dwords: dword p*G+g (plane
 * major) has bit j = bit p of token 32g+j (positions past L are 0).  The W = G*bits dwords of a record
 * are split into Q = ceil(W/4) 16-byte chunks (tail dwords zero); chunk q of sequence n lives
 * at byte offset (q*Npad + n)*16, Npad = pg_npad(N) (a multiple of 256, sequences past N are
 * zero).  A 64-lane wavefront that owns 64 consecutive sequences therefore reads one chunk per
 * lane as a single fully coalesced 1 KiB `global_load_dwordx4`, and the Hamming distance of
 * two records is B+1 VALU instructions per 32 tokens (xor, v_bitop3 x (B-1), v_bcnt).
 * Behind the Q chunk arrays the buffer carries the SIGNATURE SECTION (32 * Npad bytes): per sequence
 * the 54-bit filter signature (its plane-0 bits, 64 positions XOR-folded to 54) as FP4 (E2M1) elements -
 * 1.0 per set bit, then ten 1.0 bias slots (0 for padding sequences) - per 32 sequences one 1 KiB block in
 * v_mfma_f32_32x32x64_f8f6f4 fragment order: the column operand of the matrix-core
 * filter stage of the all-pairs engine (prograph_amd/csrc/pg_mm.h) - and behind it the FOLD SECTION
 * (32 * Npad bytes): two 16-byte arrays of Npad entries with the sequence's plane folds (plane p's G
 * dwords XOR-ed into one), planes 0..3 and 4..7 (unused planes zero) - the operands of the engine's
 * folded-exact bound on dense data.  pg_pack_planes writes all three parts.
 * Buffer size: pg_planes_bytes(N, L, bits) = (pg_nchunks(L, bits) * 16 + 64) * pg_npad(N) bytes.
 */
#ifndef PROGRAPH_HIP_H
#define PROGRAPH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_ABI_VERSION 3   /* 2: plane buffers carry the signature and fold sections; 3: caller-owned workspace */

/* library error codes (negative return values) */
#define PG_E_BADARG   (-1)   /* NULL pointer, negative size, k out of range ...        */
#define PG_E_TOOLONG  (-2)   /* L > PG_MAX_L (PG_MAX_L_5BIT with 5 bit planes)          */
#define PG_E_TOOMANY  (-3)   /* N > PG_MAX_N_KNN (24-bit column index in packed keys)   */
#define PG_E_NODEV    (-4)   /* no HIP device / wrong architecture                      */
#define PG_E_COMM     (-5)   /* RCCL missing or a collective failed (pg_last_error)     */

#define PG_MAX_L      128          /* tokens per sequence, 8 bit planes                  */
#define PG_MAX_L_5BIT 255          /* tokens per sequence, 5 bit planes (distance fits uint8) */
#define PG_MAX_N_KNN  16777216     /* 2^24                                               */
#define PG_MAX_K      63           /* k+1 sorted keys live in the 64 lanes of one VGPR   */
#define PG_LEV_MAX_BAND 8          /* banded Levenshtein keeps 2*8+1 diagonals in registers */

/* bit planes per token: fixed when a matrix is packed, passed to every call that reads it */
#define PG_BITS_5 5                /* every token <= 31 (20 amino acids + pad): 6 VALU ops / 32 tokens */
#define PG_BITS_8 8                /* any byte token 0..255:                    9 VALU ops / 32 tokens */

/* comparator codes for pg_eps_*: the reference's `comp` argument (operator.le default,
 * prograph/prograph.py:665) restricted to the five orderings                      */
#define PG_CMP_LE 0
#define PG_CMP_LT 1
#define PG_CMP_EQ 2
#define PG_CMP_GE 3
#define PG_CMP_GT 4

int         pg_version(void);
const char *pg_last_error(void);
int         pg_device_info(int *cu_count, int *wave_size, char *arch, int arch_len);

/* Npad for N sequences (multiple of 256); groups of 32 positions; 16-byte chunks per record. */
int64_t     pg_npad(int64_t n);
int         pg_ngroups(int l);
int         pg_nchunks(int l, int bits);
/* bytes of a plane buffer for n sequences of l tokens: chunk arrays + signature section */
int64_t     pg_planes_bytes(int64_t n, int l, int bits);
/* bytes of the launch-private `workspace` of an all-pairs call over nrows rows (see Conventions) */
int64_t     pg_workspace_bytes(int64_t nrows);

/*
 * pg_pack_planes — row-major tokens -> plane layout.
 * Replaces the H->D staging `torch.as_tensor(self(representation), dtype=float16,
 * device="cuda:0")[idxs,:]` (prograph/prograph.py:726) and the zero right-padding of
 * `clean_input` (prograph/distance/utils.py:32-38).
 *   src        (n, l) row-major, leading dimension `ld` ELEMENTS, element size
 *              `elem_bytes` in {1,2,4,8} (uint8 / int16 / int32 / int64 tokens)
 *   rows       optional int64[n] gather list (the reference's `idxs`), NULL = identity
 *   bits       PG_BITS_5 or PG_BITS_8
 *   planes     out, pg_planes_bytes(n,l,bits) bytes, fully overwritten (padding zeroed)
 *   flags      out, uint32[1]: set to 1 when some token is outside 0..2^bits-1 (such tokens
 *              are truncated; the caller must not use the result)
 */
int pg_pack_planes(const void *src, int elem_bytes, int64_t n, int l, int64_t ld,
                   const int64_t *rows, int bits, void *planes, int64_t npad, uint32_t *flags,
                   void *stream);

/*
 * pg_pack_bytes — tokenise AND pack on the device.
 * Replaces `Prograph.tokenize` (prograph/prograph.py:454-474: one np.where pass per alphabet letter over the
 * fixed-width byte view of the sequence strings, table :127) together with the staging above: `src` is that byte
 * view, (n, width) uint8 row-major (numpy 'S<width>' storage: short sequences are NUL padded), `lut256` the
 * 256-entry letter table on the device (letter j of the alphabet -> j+1, everything else -> 0).
 *   tokens_out optional uint8 (n, width) row-major: the token matrix itself, for hosts that expose it
 *   flags      as in pg_pack_planes (set when a table entry does not fit `bits` planes)
 */
int pg_pack_bytes(const uint8_t *src, int64_t n, int width, int64_t ld, const int64_t *rows, const uint8_t *lut256,
                  int bits, void *planes, int64_t npad, uint8_t *tokens_out, uint32_t *flags, void *stream);

/*
 * pg_hamming_dense — all-pairs Hamming distance matrix.
 * Replaces `torch.sum(X != Y[:,None,:], axis=2)` (prograph/distance/hamming.py:34;
 * K2+K3 of SURVEY.md §2.2).  out[m*ldo + n] = #{j : Y[m,j] != X[n,j]}, (M,N) like the
 * reference.  out_elem_bytes in {1,2,4,8}: uint8 / fp16 / int32 / int64 (= the reference's dtype).
 * fp16 holds the integer itself (exact up to 2048 positions): a block in that form is the operand of
 * pg_f16_knn / pg_f16_eps_* below - graphs of sequences longer than one record of the fused engines.
 * accumulate != 0 adds to `out` instead of overwriting it: sequences longer than one record
 * (255 / 128 tokens) are handled as a sum over column segments packed separately.
 */
int pg_hamming_dense(const void *x_planes, int64_t n, int64_t x_npad,
                     const void *y_planes, int64_t m, int64_t y_npad,
                     int l, int bits, void *out, int out_elem_bytes, int64_t ldo,
                     int accumulate, void *stream);

/*
 * pg_eps_slots — the N^2 pass of the epsilon-neighbourhood graph.
 * Replaces the hot loop `distance(X,batch)` -> `comp(d,eps) & (d>0)` -> `torch.where`
 * -> gather of `build_graph` (prograph/prograph.py:731-739; K2..K7) for rows
 * [row0, row0+nrows) of `row_planes` against all `ncols` sequences of `col_planes`.
 * For every row the matching column indices are written into the row's slot (capacity `cap`)
 * in ascending order of their 32-column tile - within one tile in any order: a slot is an
 * intermediate, pg_eps_compact puts every entry in its place (an entry is at most 31 positions
 * from it) - and the exact number of matches into counts[] even when it exceeds `cap`
 * (pg_eps_compact recomputes such rows).
 *   cmp, eps   PG_CMP_* and the threshold; pairs with d == 0 are always excluded
 *   slot_idx   int32 [nrows*cap], slot_w uint8 [nrows*cap], counts uint32 [nrows]
 */
int pg_eps_slots(const void *row_planes, int64_t row_npad, int64_t row0, int64_t nrows,
                 const void *col_planes, int64_t col_npad, int64_t ncols,
                 int l, int bits, int cmp, double eps, int cap,
                 int32_t *slot_idx, uint8_t *slot_w, uint32_t *counts, void *workspace, void *stream);

/*
 * Square self-graph variant of pg_eps_slots / pg_eps_compact: rows [0,n) against the same n sequences.
 * Hamming and the five comparators are symmetric, so every unordered pair is evaluated ONCE (the
 * reference evaluates both orders, prograph/prograph.py:731-739): a match (i, j), i < j, is written to
 * the front of row i's slot in column order and to the back of row j's slot in arrival order
 * (counts_lo[j] is an atomic counter, zeroed by the call).  The total per row is
 * counts_up[i] + counts_lo[i]: the caller adds them, scans (pg_exclusive_scan) and calls
 * pg_eps_compact_sym, which emits each CSR row in ascending column order (rows that overflow `cap`,
 * or hold more than 512 entries from below, are recomputed exactly as in pg_eps_compact).  n < 2^27.
 * The result is identical to pg_eps_slots + pg_eps_compact on the same input.
 */
int pg_eps_slots_sym(const void *planes, int64_t npad, int64_t n, int l, int bits, int cmp, double eps, int cap,
                     int32_t *slot_idx, uint8_t *slot_w, uint32_t *counts_up, uint32_t *counts_lo, void *workspace,
                     void *stream);
int pg_eps_compact_sym(const void *planes, int64_t npad, int64_t n, int l, int bits, int cmp, double eps, int cap,
                       const int32_t *slot_idx, const uint8_t *slot_w, const uint32_t *counts_up,
                       const uint32_t *counts_lo, const int64_t *indptr, int32_t *indices, uint8_t *weights,
                       int leave_overflow, void *stream);

/*
 * pg_exclusive_scan — indptr[0..n] = exclusive prefix sum of counts[0..n) (int64).
 * Replaces the per-row split of `prod_neighbours` (prograph/prograph.py:646-654).
 * `scratch` must hold pg_scan_scratch_bytes(n) bytes.
 */
int64_t pg_scan_scratch_bytes(int64_t n);
int pg_exclusive_scan(const uint32_t *counts, int64_t n, int64_t *indptr, void *scratch,
                      void *stream);

/*
 * pg_eps_compact — slots -> CSR.  indices/weights must hold indptr[nrows] entries.
 * Output: indices int32 ascending per row (the order `torch.where` yields,
 * prograph/prograph.py:736), weights uint8 = the Hamming distance.
 * Rows whose count exceeded `cap`: leave_overflow = 0 recomputes each of them here (one wavefront per
 * row over all columns: fine for a handful of rows); leave_overflow = 1 skips them and the caller runs
 * pg_eps_fill_rows over the list of such rows (the engine again, exact and at engine speed however
 * many rows overflow - dense graphs).
 */
int pg_eps_compact(const void *row_planes, int64_t row_npad, int64_t row0, int64_t nrows,
                   const void *col_planes, int64_t col_npad, int64_t ncols,
                   int l, int bits, int cmp, double eps, int cap,
                   const int32_t *slot_idx, const uint8_t *slot_w, const uint32_t *counts,
                   const int64_t *indptr, int32_t *indices, uint8_t *weights, int leave_overflow, void *stream);

/*
 * pg_eps_fill_rows — the epsilon pass for a LIST of rows, written straight into the CSR.
 * row_list: int64[n_list] row numbers relative to row0 (ascending or not); for each the matches among
 * all ncols columns go, in ascending column order, to indices/weights at indptr[row] (indptr as for
 * pg_eps_compact: relative to row0; the counts that produced it are exact, so the segments fit).
 * scratch_counts: uint32[n_list] (receives the per-row counts again).
 */
int pg_eps_fill_rows(const void *row_planes, int64_t row_npad, int64_t row0, const int64_t *row_list, int64_t n_list,
                     const void *col_planes, int64_t col_npad, int64_t ncols, int l, int bits, int cmp, double eps,
                     const int64_t *indptr, int32_t *indices, uint8_t *weights, uint32_t *scratch_counts, void *workspace,
                     void *stream);

/*
 * pg_knn_hamming — k nearest neighbours under the canonical (distance, index) order.
 * Replaces `torch.sort(distance(X,batch),dim=1)` + `[:,1:k+1]` (prograph/prograph.py:
 * 758-762; K8,K9): for each row the k+1 smallest (d, column) pairs are kept, rank 0 is
 * dropped (not "self": prograph/prograph.py:761-763), ranks 1..k are written.
 * Ranks that do not exist (ncols < k+1) get idx = -1, dist = 255.
 *   idx_out int32 [nrows*k], dist_out uint8 [nrows*k];  1 <= k <= PG_MAX_K
 */
int pg_knn_hamming(const void *row_planes, int64_t row_npad, int64_t row0, int64_t nrows,
                   const void *col_planes, int64_t col_npad, int64_t ncols,
                   int l, int bits, int k, int32_t *idx_out, uint8_t *dist_out,
                   void *workspace, void *stream);

/*
 * pg_knn_hamming_round — kNN beyond 63 neighbours, 63/64 ranks per all-pairs round.
 * Round 1 (first_round = 1): like pg_knn_hamming (rank 0 dropped, ranks 1..k, k <= 63) and in
 * addition last_keys[row] = packed key (distance << 24 | column) of the last rank written.
 * Later rounds (first_round = 0): only pairs whose key is greater than floor_keys[row] (= the
 * previous round's last_keys) are candidates; the k <= 64 smallest of them are written, i.e. the
 * next k ranks of the same canonical order.  idx_out / dist_out hold nrows*k entries per round.
 */
int pg_knn_hamming_round(const void *row_planes, int64_t row_npad, int64_t row0, int64_t nrows,
                         const void *col_planes, int64_t col_npad, int64_t ncols, int l, int bits,
                         int k, int first_round, const uint32_t *floor_keys, uint32_t *last_keys,
                         int32_t *idx_out, uint8_t *dist_out, void *workspace, void *stream);

/*
 * pg_index_flags — the fused 1xN pass of `Prograph.indexing` (prograph/prograph.py:
 * 298-325): distance of every sequence to reference row `ref`, a 256-bin histogram of
 * those distances (for the `d in np.unique(d_data)` assertion, :305), and
 * flags[n] = dist_ok(n) && pos_ok(n) where
 *   dist_ok = want_dist == NULL || bit d of the 256-bit set want_dist[8] is set
 *   pos_ok  = pos_mode == 0, or: (pos_mode 1 "or": some byte selected by pos_mask
 *             differs from the reference row; 2 "and": all selected bytes differ) and no
 *             byte selected by not_mask differs          (:316-325)
 * pos_mask / not_mask: uint32[pg_ngroups(l)] device arrays, bit j of word g selects position 32g+j.
 *   dist_out uint8[n] (may be NULL), hist uint64[256] (may be NULL; must be zeroed by
 *   the caller), flags uint8[n] (may be NULL)
 */
int pg_index_flags(const void *planes, int64_t n, int64_t npad, int l, int bits,
                   int64_t ref, const uint32_t *want_dist, int pos_mode,
                   const uint32_t *pos_mask, const uint32_t *not_mask,
                   uint8_t *dist_out, uint64_t *hist, uint8_t *flags, void *stream);

/*
 * Banded Levenshtein kNN — BUILD DEFINED, no reference counterpart (BASELINE.json configs[4],
 * SURVEY.md §8 row a9; parity unpinned).  d(a,b) = min(edit_distance(a,b), band+1) over the
 * non-zero prefixes of zero-right-padded uint8 token rows (tokens 1..31); neighbours ordered by
 * (d, column), rank 0 dropped, ranks 1..k written (missing ranks: idx -1, dist 255).
 *   pg_lev_profile     tokens (n,l) row-major uint8 -> bag profiles (3*npad*16 bytes) + lens[n];
 *                      flags[0] = 1 if a token > 31 or an interior zero was seen
 *   pg_lev_candidates  all-pairs necessary-condition filter max(SAD, 2|dlen|) <= 2*band into
 *                      per-row candidate slots (ascending columns, exact counts[] even past cap;
 *                      the caller re-runs with a larger cap when max(counts) > cap)
 *   pg_lev_candidates_sym  the same for all rows at once with every unordered pair filtered once
 *                      (slots as in pg_eps_slots_sym: counts_up / counts_lo, the row itself in
 *                      neither); the caller re-runs with a larger cap when max(up + lo) > cap
 *   pg_lev_knn         exact banded edit distance per candidate (bit-parallel diagonal band) +
 *                      canonical kNN selection; `planes128` = the same tokens packed with
 *                      pg_pack_planes(bits = 5) at width l = 128 (chunk p = bit plane p);
 *                      counts_lo = NULL for pg_lev_candidates slots, else the symmetric pair.
 *                      slot_aux (int32 [n*cap], optional, filled by pg_lev_candidates_sym) holds for
 *                      every front entry the position of its mirror entry: with it and slot_w
 *                      (scratch, uint8 [n*cap]) every candidate PAIR is evaluated once and the
 *                      distance stored with both entries before the selection runs
 */
int pg_lev_profile(const uint8_t *tokens, int64_t n, int l, int64_t ld, void *profiles,
                   int64_t npad, int32_t *lens, uint32_t *flags, void *stream);
int pg_lev_candidates(const void *profiles, int64_t npad, int64_t n, int64_t row0, int64_t nrows,
                      int band, int cap, int32_t *slot_idx, uint8_t *slot_w, uint32_t *counts,
                      void *stream);
int pg_lev_candidates_sym(const void *profiles, int64_t npad, int64_t n, int band, int cap,
                          int32_t *slot_idx, uint8_t *slot_w, int32_t *slot_aux, uint32_t *counts_up,
                          uint32_t *counts_lo, void *stream);
int pg_lev_knn(const uint8_t *tokens, int64_t n, int l, int64_t ld, const void *planes128,
               int64_t npad, const int32_t *lens, int64_t row0, int64_t nrows, int band, int k,
               int cap, const int32_t *slot_idx, uint8_t *slot_w, const int32_t *slot_aux,
               const uint32_t *counts, const uint32_t *counts_lo, int32_t *idx_out, uint8_t *dist_out,
               void *stream);

/*
 * pg_csr_row_stats — per-row reductions over a CSR graph for the analytics that consume the
 * `Neighbours` column (prograph/prograph.py:797-946: degree, laplacian, dirichlet, local_variance):
 *   deg[r] = sum_j w_rj,  sum_f[r] = sum_j f[col_j],  sum_wf[r] = sum_j w_rj * f[col_j],
 *   self_w[r] = weight of the entry whose column is the row's own node row0 + r (kNN lists of
 *   duplicated sequences contain it; the reference's Laplacian overwrites that diagonal term,
 *   prograph.py:894-896),  col_sum[c] += w_rc (in-degree, `mode="indegree"`; caller zeroes it).
 * weights: uint8 distances OR float32 (exactly one non-NULL; both NULL = boolean weights 1);
 * f = per-node values (double[ncols]); any output may be NULL.
 */
int pg_csr_row_stats(const int64_t *indptr, const int32_t *indices, const uint8_t *weights_u8,
                     const float *weights_f32, int64_t nrows, int64_t row0, const double *f, double *deg,
                     double *sum_f, double *sum_wf, double *self_w, double *col_sum, void *stream);

/*
 * pg_compact_flags — ascending indices of the non-zero flags (np.where(...)[0]).
 *   out_idx int64[n] (worst case), out_count int64[1]; scratch: pg_scan_scratch_bytes(n)
 */
int pg_compact_flags(const uint8_t *flags, int64_t n, int64_t *out_idx, int64_t *out_count,
                     void *scratch, void *stream);

/*
 * Minkowski (p = 2) graphs of fp16 embeddings: `build_graph(representation="Embedded",
 * distance=minkowski)`, prograph/distance/minkowski.py:8-41 through prograph/prograph.py:726-764.  Every
 * elementwise step rounds to fp16 exactly as the reference's fp16 tensor expression does (difference,
 * square, sum, root, and 1/(1+d) for similarities); see prograph_amd/csrc/pg_mink.hip for the tolerance.
 *   pg_pack_f16          (n, d) fp16 row-major (leading dimension ld elements, optional row gather list)
 *                        -> chunk-major: chunk q (8 halfs) of vector n at byte (q*npad + n)*16;
 *                        buffer pg_f16_nchunks(d) * npad * 16 bytes
 *   pg_minkowski_dense   out[m*ldo + n] = fp16 distance (similarity != 0: 1/(1+d)) of Y[m] and X[n]
 *   pg_f16_knn           ranks first..first+k-1 of every row of such a block in (value, column) order
 *                        (descending != 0: largest first, the similarity sort of :758); first + k <= 64
 *   pg_f16_eps_count/_fill  comp(d, eps) & (d > 0)  [similarities: comp(eps, s) & (s < 1)], eps_f16 = the
 *                        threshold rounded to fp16 as torch does when it compares an fp16 tensor with a
 *                        Python number; count -> pg_exclusive_scan -> fill (columns ascending, fp16 weights)
 */
int pg_f16_nchunks(int d);
int pg_pack_f16(const void *src_f16, int64_t n, int d, int64_t ld, const int64_t *rows, void *packed, int64_t npad,
                void *stream);
int pg_minkowski_dense(const void *x_packed, int64_t n, int64_t x_npad, const void *y_packed, int64_t m,
                       int64_t y_npad, int d, int similarity, void *out_f16, int64_t ldo, void *stream);
int pg_f16_knn(const void *dist_f16, int64_t m, int64_t n, int64_t ld, int k, int first, int descending,
               int32_t *idx_out, void *w_out_f16, void *stream);
int pg_f16_eps_count(const void *dist_f16, int64_t m, int64_t n, int64_t ld, int cmp, float eps_f16, int similarity,
                     uint32_t *counts, void *stream);
int pg_f16_eps_fill(const void *dist_f16, int64_t m, int64_t n, int64_t ld, int cmp, float eps_f16, int similarity,
                    const int64_t *indptr, int32_t *indices, void *weights_f16, void *stream);

/*
 * Multi-GPU: the path's ONE collective (e).  The N^2 pair space shards row-block
 * wise, one process per GPU; every rank needs the whole token matrix, so the ranks all-gather their
 * row shards once (RCCL over xGMI: 64 MB at N = 1M, L = 64) and never talk again.  The reference has
 * no counterpart (single hard-coded cuda:0, prograph/prograph.py:726).
 *   pg_comm_unique_id   rank 0 creates the 128-byte id (ncclGetUniqueId); the host carries it to the
 *                       other ranks by its own means (MPI, a file, torch.distributed's store, ...)
 *   pg_comm_init        every rank, on its GPU (the current HIP device): ncclCommInitRank
 *   pg_allgather_tokens shard (rows_per_rank, l) uint8, contiguous, the same rows_per_rank on every
 *                       rank (pad the last block with zero rows); full (nranks*rows_per_rank, l);
 *                       enqueued on `stream`, no host synchronisation
 * RCCL is resolved at run time (the copy the host process already loaded, else the ROCm installation's):
 * a host without librccl.so still loads this library and gets PG_E_COMM from these calls only.
 *   pg_comm_available   1 when this process can bind RCCL (a local check, no communication): hosts let every rank
 *                       agree on it BEFORE anybody enters the collective pg_comm_init
 */
#define PG_COMM_ID_BYTES 128
int pg_comm_available(void);
int pg_comm_unique_id(void *id128);
int pg_comm_init(void **comm, int nranks, int rank, const void *id128);
int pg_comm_destroy(void *comm);
int pg_allgather_tokens(void *comm, const void *shard, int64_t rows_per_rank, int l, void *full, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PROGRAPH_HIP_H */
