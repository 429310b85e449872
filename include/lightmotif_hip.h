/*
 * lightmotif_hip.h -- C ABI of the MI355X (gfx950) scoring back-end.
 *
 * This is the drop-in boundary for ONE hot path of althonos/lightmotif:
 *
 *     pssm.score(&striped)  ->  StripedScores  ->  argmax / max / threshold
 *
 * The reference has no FFI; its boundary is the trait surface
 * `Score` / `Maximum` / `Threshold` (+ `Stripe`, `Encode`) of
 * lightmotif/src/pli/mod.rs:34-222, implemented per back-end marker and fanned
 * out by `enum Dispatch` (lightmotif/src/pli/dispatch.rs:33-41).  A new
 * `Dispatch::Hip` arm (dispatch.rs:90-106, 160-177) -- or an out-of-crate type
 * implementing the public traits -- forwards raw pointers and dimensions to
 * the functions below (INTEGRATION.md has the Rust shim).  Every entry point
 * names the reference function it stands in for.
 *
 * Conventions
 *  - plain C types only; no C++ exceptions cross the boundary;
 *  - every function returns an `lm_hip_status`; `lm_hip_last_error()` gives the
 *    thread-local message of the last failure.  The reference's trait methods
 *    return ()/Option and *panic* on misuse (avx2.rs:832-837, 354-358); the shim
 *    turns a non-zero status into `panic!`.  Degenerate inputs are NOT errors:
 *    L < M or an empty row range gives zero rows (pli/mod.rs:85-88), empty
 *    scores give found = 0 (pli/mod.rs:136-138);
 *  - a NULL handle or a NULL required pointer is LM_HIP_ERR_BAD_ARGS (checked before any device
 *    work: the context stays usable); `*_destroy(NULL)` and lm_hip_free(NULL) are no-ops.  Outputs
 *    that may be NULL when the caller does not want them: `best` / `value` next to `found`
 *    (argmax, max, the batch argmax), `values` next to `coords` (threshold batch), `out_rows` /
 *    `max_index`, `bad_index` (tests/test_abi_exports.py, tests/test_gpu_abi_misuse.py);
 *  - memory layout is the reference's: striped sequence
 *    (rows+wrap) x stride bytes, PSSM M x stride f32 (stride 8 for DNA, 24 for
 *    protein), scores rows x stride f32, all row-major, `stride` in ELEMENTS
 *    (DenseMatrix::stride, dense.rs:126-128);
 *  - `*_dptr` functions take DEVICE pointers and enqueue on the context's
 *    stream; results that come back to the host (argmax, threshold, counts)
 *    synchronise that stream before returning.  Host-pointer functions are
 *    fully synchronous (the GPU analogue of the `_mm_sfence` at avx2.rs:198);
 *  - all entry points are thread-safe; a context serialises its own calls.
 */
#ifndef LIGHTMOTIF_HIP_H
#define LIGHTMOTIF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped when an entry point is removed or changes its meaning.  Rounds 2-6 only ADDED entry points (the one removal, of
 * lm_hip_ctx_set_xcd_remap in round 5, was taken back: it is a no-op now), so a binding built against an earlier header
 * still loads and links. */
#define LM_HIP_ABI_VERSION 1

typedef enum lm_hip_status {
    LM_HIP_OK = 0,
    LM_HIP_ERR_BAD_ARGS = 1,    /* null pointer, stride < cols, rows out of range ... */
    LM_HIP_ERR_WRAP = 2,        /* "not enough wrapping rows for motif of length M" (avx2.rs:832-837) */
    LM_HIP_ERR_HIP = 3,         /* a HIP runtime call failed */
    LM_HIP_ERR_OOM = 4,         /* host or device allocation failed */
    LM_HIP_ERR_NO_DEVICE = 5,   /* no gfx950 device: maps to UnsupportedBackend (err.rs:34) */
    LM_HIP_ERR_INVALID_SYMBOL = 6, /* Encode: InvalidSymbol (err.rs:10) */
    LM_HIP_ERR_CAPACITY = 7,    /* caller-provided buffer too small (output list, adopted matrix) */
    LM_HIP_ERR_COMM = 8         /* RCCL missing, or a collective / communicator call failed */
} lm_hip_status;

/* (row, col) of a cell of a striped matrix -- dense.rs:28-39 MatrixCoordinates. */
typedef struct lm_hip_coords {
    size_t row;
    size_t col;
} lm_hip_coords;

/* One above-threshold position of the fused scan (scan.rs:52-75 `Hit`). */
typedef struct lm_hip_hit {
    size_t position; /* col * rows + row, scores.rs:155-157 */
    float score;
} lm_hip_hit;

typedef struct lm_hip_ctx lm_hip_ctx;       /* device + stream + scratch */
typedef struct lm_hip_pssm lm_hip_pssm;     /* ScoringMatrix data resident on the device */
typedef struct lm_hip_seq lm_hip_seq;       /* StripedSequence resident on the device */
typedef struct lm_hip_scores lm_hip_scores; /* StripedScores<f32> resident on the device */
typedef struct lm_hip_comm lm_hip_comm;     /* RCCL communicator of a row-sharded job */

/* ---- library ------------------------------------------------------------ */

int lm_hip_abi_version(void);
const char *lm_hip_last_error(void);
/* Pipeline::avx2()/neon() -> Result<_, UnsupportedBackend> (pli/mod.rs:401-407):
 * number of usable devices; 0 devices is reported through *count, not an error. */
int lm_hip_device_count(int *count);
/* HIP ordinal of the index-th usable (gfx950) device, 0 <= index < lm_hip_device_count: the
 * value lm_hip_ctx_create and $LM_HIP_DEVICE take.  On a node with other GPUs in between,
 * iterate over this instead of range(count).  LM_HIP_ERR_NO_DEVICE past the end. */
int lm_hip_device_ordinal(int index, int *ordinal);
/* Frees buffers this library returned to the caller (threshold / hit lists). */
void lm_hip_free(void *p);
/* The pool behind large result arrays (threshold / hit lists): page-locked 2 MB-aligned blocks re-used between calls.
 * Bytes it holds idle, bytes handed out and not yet freed, and the idle budget ($LM_HIP_RESULT_POOL_MB, default 256;
 * blocks in use may reach four times that, results beyond come from plain malloc).  For hosts that watch their
 * locked memory; a pointer may be NULL (not all three). */
int lm_hip_result_pool_info(size_t *pinned_idle, size_t *pinned_in_use, size_t *budget);

/* Diagnostic: the shader clock (MHz) the device sustains over the next `window_us` microseconds, measured by one
 * mostly-sleeping wavefront on a stream of its own (s_memtime ticks per s_memrealtime tick) -- call it from
 * a second host thread while the kernels of interest run to learn the clock THEY get (the part clocks to its power
 * budget; a roofline priced at the 2.4 GHz of the data sheet is not what an LDS- or VALU-bound kernel can reach).
 * The probe's stream and record are made once per device and kept.  (Two marks enqueued IN the measured stream were tried
 * and dropped: the counter is per XCD and not comparable between the places two one-workgroup kernels land.)
 * No reference counterpart; bench.py reports it next to every LDS fraction. */
int lm_hip_device_clock_mhz(int device, unsigned window_us, double *mhz);

/* DenseMatrix::stride (dense.rs:126-128) for x86-64 hosts: elements per row. */
size_t lm_hip_stride(size_t cols, size_t elem_size);

/* ---- context ------------------------------------------------------------ */

/* `Pipeline::dispatch()` is re-created per call in the reference
 * (pwm/mod.rs:646, scores.rs:182); the GPU state it would need lives here. */
/* `device` is a HIP ordinal (what hipSetDevice / torch.device("cuda", i) / LOCAL_RANK use);
 * LM_HIP_ERR_NO_DEVICE if it is not a gfx950 device (lm_hip_device_ordinal lists those). */
int lm_hip_ctx_create(int device, lm_hip_ctx **out);
/* Borrow an existing hipStream_t (e.g. PyTorch's current stream) instead of
 * creating one.  The stream must outlive the context. */
int lm_hip_ctx_create_on_stream(int device, void *hip_stream, lm_hip_ctx **out);
int lm_hip_ctx_destroy(lm_hip_ctx *ctx);
int lm_hip_ctx_sync(lm_hip_ctx *ctx);
/* hipStream_t the context enqueues on (for hipEvent timing by the caller). */
int lm_hip_ctx_stream(lm_hip_ctx *ctx, void **hip_stream);
/* Tuning knob: output rows each wavefront half sweeps in the C=32 score
 * kernels (0 = library default). */
int lm_hip_ctx_set_rows_per_stream(lm_hip_ctx *ctx, size_t rows);
/* Tuning knob of the fused score+threshold scans (lm_hip_score_threshold_f32_dptr,
 * lm_hip_scan_f32, lm_hip_scan_threshold_batch): 1 (default) = candidates are found
 * with a packed 16-bit over-estimating discretisation of the PSSM (the GPU form of the
 * reference Scanner's DiscreteMatrix prefilter, scan.rs:169-198) and re-scored exactly;
 * 0 = every position is scored in f32.  The hit lists are identical either way. */
int lm_hip_ctx_set_prefilter(lm_hip_ctx *ctx, int enabled);
/* lm_hip_score_rows_into / lm_hip_score_into on handles: 1 (default) = for matrices of at least
 * 8 Mi cells the store kernel also tracks the maximum, and a following lm_hip_argmax on the
 * same lm_hip_scores (the reference's score_into + argmax flow, lightmotif-bench dna.rs:
 * 104-107) is a 16-byte read instead of a second pass over the matrix (1.65 -> 1.03 ms per
 * Gbp for the pair; the two tiny reduction launches add ~2 % to a score_into that is never
 * followed by argmax); 0 = plain store.  The cached result is invalidated by the library's
 * own writes to the handle -- not by external writes through the device pointer of
 * lm_hip_scores_info. */
int lm_hip_ctx_set_track_argmax(lm_hip_ctx *ctx, int enabled);
/* Selects an alternative path inside the library for this context.  Every option leaves the RESULTS unchanged (the
 * test-suite runs both sides of each against the oracle); they exist for A/B measurements and to exercise paths that
 * are otherwise taken only for some shapes.  Names: "track_argmax", "prefilter" (= the setters above),
 * "pair_prefilter", "pair_prefilter_protein", "speculate_order", "suffix_argmax", "suffix_occurrences", "multi_motif",
 * "skip_unreachable", "quad_loads", "xlong_store", "host_fold", "chunked_fused", "chunk_rows", "tiled", "sort_hits",
 * "short_order", "time_scan", "drop_last", "list_scan_max" (0 = Scanner::max always walks windows of materialised u8 scores),
 * "block_prefilter" (0 = the protein one-symbol scans load a byte per lane and row
 * instead of 4-row blocks), "poll_done" (1 = a single fused threshold call polls the word the ranking kernel raises in pinned
 * memory instead of waiting for its stream: 3-4 us less per call, unless the caller synchronises the device right behind it); "xcd_remap" is accepted and ignored (the remap it selected was removed in round 5).  Unknown names:
 * LM_HIP_ERR_BAD_ARGS.  The shipped library reads none of them from the environment. */
int lm_hip_ctx_set_option(lm_hip_ctx *ctx, const char *name, double value);
/* Round-4 ABI: selected an XCD-aware workgroup remap of the store kernel, which measured slower and was removed in round 5.
 * Kept as a no-op returning LM_HIP_OK so that embedders built against the round-4 header still load and link. */
int lm_hip_ctx_set_xcd_remap(lm_hip_ctx *ctx, int enabled);
/* Name of the kernel the last score call on this context launched
 * (for profiling tools); valid until the next call.  "score_c32<M,MODE>" names
 * the kernel family and the length it ran at (MODE 0 store, 1 fused argmax,
 * 2 fused threshold); the store kernel's tracking forms (lm_hip_ctx_set_track_argmax:
 * template modes 3 and 5 in a rocprofv3 trace) report as "score_c32<M,0>". */
const char *lm_hip_ctx_last_kernel(lm_hip_ctx *ctx);
/* Diagnostic: what the last fused threshold scan on this context (single or batched) produced -- hits, and CANDIDATES: the
 * pieces of up to 32 rows the discrete prefilter flagged for exact re-scoring.  Candidates per hit is how much a
 * sequence costs beyond the scan itself (low-complexity tracts and N runs raise it: profiles/r05_realistic_inputs.json).
 * Either output may be NULL. */
int lm_hip_ctx_last_scan_counts(lm_hip_ctx *ctx, unsigned long long *hits, unsigned long long *candidates);
/* Diagnostic: with the context option "time_scan" = 1, the duration (ms, HIP events on the context's stream) of the scan
 * kernel(s) of the last fused threshold / fused argmax call -- the part of the call that the LDS roofline bounds, without
 * the re-scoring, the ordering of the hit list and the read-back behind it (bench.py: roofline.kernel_frac of the fused
 * blocks).  -1 when the option is off or the call took a route without a scan kernel. */
int lm_hip_ctx_last_scan_kernel_ms(lm_hip_ctx *ctx, float *ms);
/* Diagnostic ("time_scan" = 1): the last fused THRESHOLD call (single or batched) by phase, in ms -- [0] the scan kernels,
 * [1] the exact re-scoring of their candidates, [2] the kernels that order the hit list (HIP events on the context's
 * stream), [3] the rest of the call on the host clock: enqueueing, the synchronisation's wake-up, the copy of the results
 * out of the pinned block.  -1 where nothing was recorded. */
int lm_hip_ctx_last_phases_ms(lm_hip_ctx *ctx, float phases[4]);
/* Diagnostic: what the scan kernel of the last SINGLE-job fused call on this context looked up per position -- the motif
 * rows it scanned (a pair scan of M = 20, 24, ... 36 may look M - 1 rows up and credit the last one with its best weight)
 * and the bytes of LDS table that costs (what a roofline figure must be priced against).  Zeros: no such scan ran (a batch,
 * an argmax settled from the last rows, a chunked route).  Either pointer may be NULL. */
int lm_hip_ctx_last_scan_info(lm_hip_ctx *ctx, size_t *motif_rows_scanned, size_t *lds_bytes_per_position);

/* ---- PSSM ---------------------------------------------------------------- */

/* ScoringMatrix<A>.matrix() (pwm/mod.rs:561-564): `pssm` = pssm[0].as_ptr(),
 * m = rows(), stride = stride() (8 for Dna, 24 for Protein), k = A::K::USIZE.
 * Columns >= k of a row are never read (A4: they may be uninitialised). */
int lm_hip_pssm_create(lm_hip_ctx *ctx, const float *pssm, size_t m, size_t stride,
                       size_t k, lm_hip_pssm **out);
int lm_hip_pssm_destroy(lm_hip_pssm *pssm);
size_t lm_hip_pssm_len(const lm_hip_pssm *pssm);

/* ScoringMatrix::<Dna>::reverse_complement (pwm/mod.rs:566-577): a new resident matrix with
 * rows reversed and the columns of complementary nucleotides swapped (A<->T, C<->G, N kept;
 * abc.rs Dna::complement) -- the second strand of `lightmotif-cli --reverse` (main.rs) without
 * a host round trip.  DNA matrices (k == 5) only: LM_HIP_ERR_BAD_ARGS otherwise. */
int lm_hip_pssm_reverse_complement(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, lm_hip_pssm **out);

/* ---- Score (device pointers) ---------------------------------------------- */

/*
 * Score::score_rows_into (pli/mod.rs:72-106) / Avx2::score_f32_rows_into
 * (avx2.rs:889-904).
 *   d_seq            seq.matrix()[0].as_ptr(), on the device
 *   seq_rows_total   seq.matrix().rows()   (includes the wrap rows)
 *   seq_stride       seq.matrix().stride()
 *   cols             C (32 through pssm.score() on x86-64, dispatch.rs:45)
 *   wrap, length     seq.wrap(), seq.len()
 *   [row_begin,row_end)  the `rows` range
 *   d_out            scores.matrix_mut()[0].as_mut_ptr(), on the device, with
 *                    room for (row_end-row_begin) rows of out_stride floats
 * On return *out_rows / *max_index hold the arguments the reference passes to
 * scores.resize (pli/mod.rs:86,91): (0,0) in the degenerate case, else
 * (row_end-row_begin, L+1-M).  Asynchronous on the context's stream.
 * Errors: LM_HIP_ERR_WRAP if wrap < M-1; BAD_ARGS if row_end > rows.
 */
int lm_hip_score_f32_dptr(lm_hip_ctx *ctx, const lm_hip_pssm *pssm,
                          const uint8_t *d_seq, size_t seq_rows_total, size_t seq_stride,
                          size_t cols, size_t wrap, size_t length,
                          size_t row_begin, size_t row_end,
                          float *d_out, size_t out_stride,
                          size_t *out_rows, size_t *max_index);

/* ---- Maximum / Threshold (device pointers) -------------------------------- */

/* Maximum::argmax, Generic semantics (pli/mod.rs:135-155): the maximal cell
 * that is LAST in (row, col) order; NaN never wins; if scores[0][0] is NaN the
 * answer is (0,0).  *found = 0 when rows == 0.  `value` (optional) receives
 * Maximum::max (pli/mod.rs:158-160).  Synchronises. */
int lm_hip_argmax_f32_dptr(lm_hip_ctx *ctx, const float *d_scores, size_t rows,
                           size_t stride, size_t cols, int *found,
                           lm_hip_coords *best, float *value);
/* Maximum::max (pli/mod.rs:158-160) on its own: the value AT that argmax -- NaN when
 * scores[0][0] is NaN, -inf for an all -inf matrix; *found = 0 (None) when rows == 0.
 * (NOT Avx2::max_f32, which seeds with 0.0: avx2.rs:438-441.) */
int lm_hip_max_f32_dptr(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride,
                        size_t cols, int *found, float *value);

/* Same for one SHARD of a row-partitioned score matrix (multi-GPU):
 * with first_cell_rule = 0 the "scores[0][0] is NaN -> (0,0)" rule is not applied,
 * because the shard's first cell is not the matrix's first cell; the caller
 * applies it once, on the shard that holds row 0.  first_cell_rule = 1 is
 * lm_hip_argmax_f32_dptr. */
int lm_hip_argmax_shard_f32_dptr(lm_hip_ctx *ctx, const float *d_scores, size_t rows,
                                 size_t stride, size_t cols, int first_cell_rule, int *found,
                                 lm_hip_coords *best, float *value);

/* Threshold::threshold (pli/mod.rs:210-221): every (row, col) with x >= t, in
 * row-major order (the reference's push order).  *coords is malloc'ed by the
 * library (NULL when *n == 0); release with lm_hip_free.  Synchronises. */
int lm_hip_threshold_f32_dptr(lm_hip_ctx *ctx, const float *d_scores, size_t rows,
                              size_t stride, size_t cols, float t,
                              lm_hip_coords **coords, size_t *n);

/* ---- Score / Maximum / Threshold on u8 (DiscreteMatrix scores) ------------- */

/* Score<u8, A, C>::score_rows_into with the weights of a DiscreteMatrix
 * (pwm/mod.rs:754-791; what Scanner::next scores first, scan.rs:174-178).
 *   weights          dm.matrix()[0].as_ptr() ON THE HOST: m rows of weights_stride bytes,
 *                    k = A::K used per row (DenseMatrix<u8, K>; 400 bytes at most, copied per call)
 *   d_seq .. row_end as lm_hip_score_f32_dptr
 *   d_out            u8 scores on the device, (row_end-row_begin) rows of out_stride bytes
 *   saturate         1: the saturating byte adds of the SIMD back-ends (avx2.rs:336
 *                    `_mm256_adds_epu8`; what `dispatch` runs on x86-64) -> min(sum, 255);
 *                    0: Generic's `+=` on u8 (pli/mod.rs:98-102, wrapping in release
 *                    builds) -> sum mod 256.  They agree whenever no sum exceeds 255.
 * Same pre-checks, degenerate cases and *out_rows / *max_index as the f32 form.
 * Asynchronous on the context's stream. */
int lm_hip_score_u8_dptr(lm_hip_ctx *ctx, const uint8_t *weights, size_t m, size_t weights_stride,
                         size_t k, const uint8_t *d_seq, size_t seq_rows_total, size_t seq_stride,
                         size_t cols, size_t wrap, size_t length, size_t row_begin, size_t row_end,
                         uint8_t *d_out, size_t out_stride, int saturate,
                         size_t *out_rows, size_t *max_index);

/* The same on a resident sequence handle, scores returned to the HOST: `out` receives
 * (row_end-row_begin) rows of out_stride bytes (`cols` written per row), i.e. the
 * StripedScores<u8, C> matrix a Rust caller owns (scores.matrix_mut()[0].as_mut_ptr()).
 * Synchronises. */
int lm_hip_score_u8(lm_hip_ctx *ctx, const uint8_t *weights, size_t m, size_t weights_stride,
                    size_t k, const lm_hip_seq *seq, size_t row_begin, size_t row_end, int saturate,
                    uint8_t *out, size_t out_stride, size_t *out_rows, size_t *max_index);

/* Maximum<u8, C>::argmax / max, Generic semantics (pli/mod.rs:135-160; scan.rs:181 takes
 * `max`): the maximal cell that is LAST in (row, col) order.  Synchronises. */
int lm_hip_argmax_u8_dptr(lm_hip_ctx *ctx, const uint8_t *d_scores, size_t rows, size_t stride,
                          size_t cols, int *found, lm_hip_coords *best, uint8_t *value);

/* Threshold<u8, C>::threshold (pli/mod.rs:210-221; scan.rs:184): every (row, col) with
 * x >= t in row-major order; *coords malloc'ed, release with lm_hip_free.  Synchronises. */
int lm_hip_threshold_u8_dptr(lm_hip_ctx *ctx, const uint8_t *d_scores, size_t rows, size_t stride,
                             size_t cols, uint8_t t, lm_hip_coords **coords, size_t *n);

/* ---- fused score + reduce (no score matrix is written) -------------------- */

/* Equivalent to score_rows_into followed by argmax on the result
 * (lightmotif-bench/dna.rs:104-107 times exactly that pair).  `best` is in the
 * coordinates of the scored block (row relative to row_begin).  *found = 0 in
 * the degenerate case. */
int lm_hip_score_argmax_f32_dptr(lm_hip_ctx *ctx, const lm_hip_pssm *pssm,
                                 const uint8_t *d_seq, size_t seq_rows_total,
                                 size_t seq_stride, size_t cols, size_t wrap, size_t length,
                                 size_t row_begin, size_t row_end,
                                 int *found, lm_hip_coords *best, float *value);

/* Shard form of the fused argmax (see lm_hip_argmax_shard_f32_dptr). */
int lm_hip_score_argmax_shard_f32_dptr(lm_hip_ctx *ctx, const lm_hip_pssm *pssm,
                                       const uint8_t *d_seq, size_t seq_rows_total,
                                       size_t seq_stride, size_t cols, size_t wrap, size_t length,
                                       size_t row_begin, size_t row_end, int first_cell_rule,
                                       int *found, lm_hip_coords *best, float *value);

/* Equivalent to score_rows_into followed by threshold(t): (row, col) list in
 * row-major order, rows relative to row_begin. */
int lm_hip_score_threshold_f32_dptr(lm_hip_ctx *ctx, const lm_hip_pssm *pssm,
                                    const uint8_t *d_seq, size_t seq_rows_total,
                                    size_t seq_stride, size_t cols, size_t wrap, size_t length,
                                    size_t row_begin, size_t row_end, float t,
                                    lm_hip_coords **coords, float **values, size_t *n);

/* ---- many motifs x one resident sequence ---------------------------------- */

/* lightmotif-cli sends every motif against every sequence (main.rs:554-561), one
 * Scanner per (motif, sequence) job.  These run `n` motifs over one resident
 * sequence as back-to-back fused kernels with a single synchronisation; no score
 * matrix is written.  The sequence must have wrap >= max(M) - 1 (the CLI does
 * configure_wrap(max_m), main.rs:540-546).  Per motif the results equal
 * score_into + argmax / threshold on the full row range.
 *   found/best/value : arrays of n
 *   counts           : array of n; *coords / *values hold sum(counts) entries,
 *                      motif after motif, each motif's hits in row-major order;
 *                      release both with lm_hip_free (NULL when there are no hits). */
int lm_hip_scan_argmax_batch(lm_hip_ctx *ctx, const lm_hip_pssm *const *pssms, size_t n,
                             const lm_hip_seq *seq, int *found, lm_hip_coords *best, float *value);
int lm_hip_scan_threshold_batch(lm_hip_ctx *ctx, const lm_hip_pssm *const *pssms,
                                const float *thresholds, size_t n, const lm_hip_seq *seq,
                                size_t *counts, lm_hip_coords **coords, float **values);

/* Scanner (scan.rs:96-250), collected: every position with score >= threshold and
 * position + M <= L (scan.rs:185-190), as (position, f32 score) sorted by position
 * (the reference yields them in block-LIFO order and callers sort, scan.rs:291).
 * Like the reference Scanner (scan.rs:169-198) it finds candidates with a discretised
 * over-estimate of the PSSM (a packed 16-bit prefilter here, u8 DiscreteMatrix there) and
 * re-scores them exactly in f32, so the hits are those of the exact scores;
 * lm_hip_ctx_set_prefilter(ctx, 0) scores every position in f32 instead (same result).
 * Errors with LM_HIP_ERR_WRAP if wrap < M-1 (scan.rs:127-131 panics).
 * *hits is malloc'ed (NULL when *n == 0); release with lm_hip_free. */
int lm_hip_scan_f32(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq,
                    float threshold, lm_hip_hit **hits, size_t *n);

/* Scanner::max EXACTLY as the reference computes it (scan.rs:200-249) -- which is not "the best
 * hit": the u8 scores of the DiscreteMatrix steer the walk.  Cells are visited in row-major
 * order (blocks ascending, Threshold order inside a block); a cell whose u8 score reaches the
 * current level is re-scored in f32 and replaces the best hit when its score is greater, or
 * equal at a greater position, and the level becomes ITS u8 score; while no hit is held the
 * first candidate is taken as it is (no `score >= threshold` test) and the level stays the
 * scaled threshold; positions are not tested against position + M <= L, and a candidate whose
 * window leaves the striped matrix makes the reference panic -- LM_HIP_ERR_BAD_ARGS here.
 *   dweights  the DiscreteMatrix's u8 weights (HOST, M x dweights_stride; pwm/mod.rs:754-791)
 *   saturate  1: the u8 adds of the SIMD back-ends (avx2.rs:336), 0: Generic's wrapping `+=`
 *   level     dm.scale(threshold), or dm.scale(best pending score) when `have`
 *   have / position / score   the best hit among those already collected and not yet yielded
 *             (scan.rs:207-210; 0 on a fresh scanner)
 *   first_row the row the scanner has reached (scan.rs `self.row`; 0 on a fresh scanner)
 * Runs on the device window by window (u8 + f32 scores of a window, an ordered parallel search
 * per update of the walk's state): ~5 ms per Gbp.  Synchronises. */
int lm_hip_scan_max_f32(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq,
                        const uint8_t *dweights, size_t dweights_stride, int saturate, unsigned level,
                        int have, size_t position, float score, size_t first_row, int *found,
                        lm_hip_hit *best);

/* ---- Encode / Stripe (device pointers) ------------------------------------ */

/* Encode::encode_into (pli/mod.rs:56-66): ASCII -> symbol index.  alphabet is
 * 'D' (abc.rs:91-135) or 'P' (abc.rs:193-256).  lossy != 0 maps unknown bytes
 * to the default symbol (seq.rs:122-129) and never fails; otherwise the first
 * invalid byte's index is stored in *bad_index and LM_HIP_ERR_INVALID_SYMBOL
 * returned.  Synchronises. */
int lm_hip_encode_dptr(lm_hip_ctx *ctx, char alphabet, const uint8_t *d_ascii, size_t len,
                       int lossy, uint8_t *d_dst, size_t *bad_index);

/* Stripe::stripe_into (pli/mod.rs:178-200) + configure_wrap(wrap)
 * (seq.rs:369-381): writes ceil(len/cols)+wrap rows of `stride` bytes.
 * Bytes past `cols` in each row (struct padding of the reference's Row, unspecified there)
 * are set to the default symbol, like every element of a fresh row (dense.rs:144-147
 * `resize_with(rows, Default::default)`), so every byte of the matrix is a valid symbol. */
int lm_hip_stripe_dptr(lm_hip_ctx *ctx, const uint8_t *d_encoded, size_t len, size_t cols,
                       uint8_t default_symbol, size_t wrap, uint8_t *d_data, size_t stride);

/* StripedSequence::configure_wrap (seq.rs:369-381) on a resident matrix that
 * has room for rows + new_wrap rows. */
int lm_hip_configure_wrap_dptr(lm_hip_ctx *ctx, uint8_t *d_data, size_t rows, size_t stride,
                               size_t cols, size_t new_wrap, uint8_t default_symbol);

/* ---- resident handles ------------------------------------------------------ */

/* Upload an existing StripedSequence (seq.rs:288-294): data = matrix()[0].as_ptr(),
 * rows_total = matrix().rows() (incl. wrap -- any wrap, e.g. configure_wrap(max_m) of the
 * CLI), k = alphabet size (default symbol = k-1).  The `cols` live bytes of every row must
 * be symbols < k (LM_HIP_ERR_INVALID_SYMBOL otherwise; the reference's symbols are enums,
 * abc.rs:113-135).  The raw *_dptr entry points do NOT check this: bytes >= k there read
 * past the PSSM tables (garbage scores), which is the caller's precondition to keep. */
int lm_hip_seq_upload(lm_hip_ctx *ctx, const uint8_t *data, size_t rows_total, size_t stride,
                      size_t cols, size_t wrap, size_t length, size_t k, lm_hip_seq **out);
/* The same for a matrix that already lives on the device and stays the CALLER's (a buffer of the
 * host application, a torch tensor, one row shard of a multi-GPU job): nothing is copied or
 * checked, lm_hip_seq_destroy does not free it.  rows_total rows (incl. `wrap` wrap rows) are
 * valid; the buffer has room for capacity_rows >= rows_total rows, and
 * lm_hip_seq_configure_wrap grows the wrap inside that room only (LM_HIP_ERR_CAPACITY beyond). */
int lm_hip_seq_adopt_dptr(lm_hip_ctx *ctx, uint8_t *d_data, size_t rows_total, size_t capacity_rows,
                          size_t stride, size_t cols, size_t wrap, size_t length, size_t k,
                          lm_hip_seq **out);
/* EncodedSequence::to_striped (seq.rs:168-175) on the device: uploads `len`
 * symbol bytes (each < k, else LM_HIP_ERR_INVALID_SYMBOL) and stripes them there. */
int lm_hip_seq_from_encoded(lm_hip_ctx *ctx, const uint8_t *encoded, size_t len, size_t cols,
                            size_t k, lm_hip_seq **out);
/* Same, from ASCII text: encode (strict or lossy) + stripe on the device. */
int lm_hip_seq_from_ascii(lm_hip_ctx *ctx, char alphabet, const uint8_t *ascii, size_t len,
                          size_t cols, int lossy, lm_hip_seq **out, size_t *bad_index);
/* DNA packed 4 bases per byte -- base i in bits 2*(i%4).. of byte i/4, values A0 C1 T2 G3 (the
 * Nucleotide discriminants, abc.rs:115-135): a quarter of the bytes over PCIe, unpacked straight
 * into the striped matrix (pli/mod.rs:178-200 layout).  N positions: n_runs = n_run_count pairs
 * {start, size} (the .2bit container's nBlockStarts / nBlockSizes; a genome's N are few long runs)
 * and / or n_mask, one bit per base (bit i%8 of byte i/8 set: position i is N); both may be NULL /
 * 0.  What a host that keeps genomes in 2-bit form ("ship raw/2-bit sequence") hands
 * over instead of an EncodedSequence. */
int lm_hip_seq_from_2bit(lm_hip_ctx *ctx, const uint8_t *packed, const uint8_t *n_mask,
                         const uint64_t *n_runs, size_t n_run_count, size_t len, size_t cols,
                         lm_hip_seq **out);
/* StripedSequence::configure_wrap (seq.rs:369-381); `m` = wrap rows wanted. */
int lm_hip_seq_configure_wrap(lm_hip_ctx *ctx, lm_hip_seq *seq, size_t m);
/* len(), wrap(), matrix().rows() - wrap(), stride, cols, device pointer. */
int lm_hip_seq_info(const lm_hip_seq *seq, size_t *length, size_t *wrap, size_t *rows,
                    size_t *stride, size_t *cols, const uint8_t **d_data);
/* Copies (rows+wrap) x stride bytes back. */
int lm_hip_seq_download(lm_hip_ctx *ctx, const lm_hip_seq *seq, uint8_t *dst);
int lm_hip_seq_destroy(lm_hip_seq *seq);

/* StripedScores::empty() (scores.rs:118-121). */
int lm_hip_scores_create(lm_hip_ctx *ctx, size_t cols, lm_hip_scores **out);
int lm_hip_scores_info(const lm_hip_scores *scores, size_t *rows, size_t *stride, size_t *cols,
                       size_t *max_index, const float **d_data);
/* Copies rows x stride floats back. */
int lm_hip_scores_download(lm_hip_ctx *ctx, const lm_hip_scores *scores, float *dst);
/* Rows [row_begin, row_end) only, (row_end - row_begin) x stride floats: what
 * `scores[i]` (Index<usize>, scores.rs:246-254) and windowed reads need without moving
 * a multi-gigabyte matrix. */
int lm_hip_scores_download_rows(lm_hip_ctx *ctx, const lm_hip_scores *scores, size_t row_begin,
                                size_t row_end, float *dst);
int lm_hip_scores_destroy(lm_hip_scores *scores);

/* Score::score_rows_into on handles; resizes `scores` like scores.resize. */
int lm_hip_score_rows_into(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq,
                           size_t row_begin, size_t row_end, lm_hip_scores *scores);
/* Score::score_into (pli/mod.rs:109-117): rows = matrix.rows() - wrap. */
int lm_hip_score_into(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq,
                      lm_hip_scores *scores);
/* Maximum::argmax / max, Threshold::threshold on a resident StripedScores. */
int lm_hip_argmax(lm_hip_ctx *ctx, const lm_hip_scores *scores, int *found,
                  lm_hip_coords *best, float *value);
int lm_hip_max(lm_hip_ctx *ctx, const lm_hip_scores *scores, int *found, float *value);
int lm_hip_threshold(lm_hip_ctx *ctx, const lm_hip_scores *scores, float t,
                     lm_hip_coords **coords, size_t *n);

/* ---- row-sharded jobs across the GPUs of a node --------------- */

/*
 * The path shards by ROW RANGES: Score::score_rows_into takes `rows: Range<usize>` for exactly
 * this (pli/mod.rs:72-78) and output row r needs input rows r .. r+M-1 only (pli/mod.rs:99-101).
 * One process per GPU holds rows [a_g, b_g) of the striped matrix plus an M-1-row halo and
 * scores them with the entry points above -- no collective.  What has to come out is what
 * StripedScores::{argmax, max, threshold} return for the WHOLE matrix (scores.rs:181-213);
 * the functions below produce it over RCCL (bound directly by this library: librccl.so.1 is
 * opened on first use; LM_HIP_ERR_COMM when it is absent).  Collectives: every rank of the
 * communicator must make the same call, with arguments that are valid on every rank or on none
 * (argument errors are reported before the collective is entered; the one per-shard condition,
 * a shard shorter than the halo, is exchanged WITH the collective and reported on every rank).
 * Waits are bounded: a collective that does not complete within LM_HIP_COMM_TIMEOUT_MS
 * (environment, read by lm_hip_comm_create; default 120000, 0 = wait for ever) -- a peer died or
 * never made the call -- aborts the communicator (ncclCommAbort) and returns LM_HIP_ERR_COMM,
 * as does every later call on it; destroy it and make a new one.  Synchronous collectives
 * (context stream) and pipelined ones (the communicator's side stream) are ordered against each
 * other on the device, so they may be mixed freely from one host thread.
 */

/* A StripedScores that is one row shard: enabled = 0 when it does NOT hold the matrix's first
 * row, so Maximum::argmax's "scores[0][0] is NaN -> (0,0)" rule (pli/mod.rs:142-146) is skipped
 * by lm_hip_argmax and by the maximum the store kernel tracks in lm_hip_score_into.  Default 1. */
int lm_hip_scores_set_first_cell_rule(lm_hip_scores *scores, int enabled);

/* The merge rule itself, on host arrays (for hosts with their own transport -- MPI, a Rust
 * channel): n per-shard results in ascending row order, rows already GLOBAL, each computed
 * with the first-cell rule on the shard holding row 0 only.  Result = Maximum::argmax of the
 * whole matrix (pli/mod.rs:135-155): greatest value; ties -> the LAST cell in (row, col)
 * order; NaN never wins except as that first-cell (0,0).  No device needed. */
int lm_hip_combine_argmax(const int *found, const lm_hip_coords *best, const float *value, size_t n,
                          int *found_out, lm_hip_coords *best_out, float *value_out);

#define LM_HIP_COMM_ID_BYTES 128
/* ncclGetUniqueId: call on ONE rank, hand the 128 bytes to the others (MPI_Bcast, a file, a
 * torch store ...), then every rank calls lm_hip_comm_create (ncclCommInitRank; blocks until
 * all nranks ranks arrive).  One GPU per rank. */
int lm_hip_comm_unique_id(uint8_t *id);
int lm_hip_comm_create(lm_hip_ctx *ctx, const uint8_t *id, int nranks, int rank, lm_hip_comm **out);
int lm_hip_comm_destroy(lm_hip_comm *comm);
int lm_hip_comm_info(const lm_hip_comm *comm, int *rank, int *nranks);

/* Set-up: fills rows [rows, rows + halo_rows) of this rank's shard with the first halo_rows rows
 * of rank+1's shard; the last rank receives rank 0's rows turned into the reference's wrap rows
 * (seq.rs:373-378: wrap[i][j] = data[i][j+1], last column = default symbol).  One all_gather
 * of halo_rows x stride bytes per rank.  Every shard needs >= halo_rows rows.  Synchronises. */
int lm_hip_exchange_halo_dptr(lm_hip_ctx *ctx, lm_hip_comm *comm, uint8_t *d_shard, size_t rows,
                              size_t stride, size_t cols, size_t halo_rows, uint8_t default_symbol);

/* StripedScores::argmax / max (scores.rs:181-192) of the whole matrix from this rank's shard
 * result (row relative to the shard; computed with first_cell_rule = (row_offset == 0)) --
 * one all_gather of a 32-byte record per rank, then lm_hip_combine_argmax.  Every rank gets
 * the same answer, rows GLOBAL. */
int lm_hip_merge_argmax(lm_hip_ctx *ctx, lm_hip_comm *comm, int found_local,
                        const lm_hip_coords *best_local, float value_local, size_t row_offset,
                        int *found, lm_hip_coords *best, float *value);
/* The same starting from the resident shard: uses the maximum the store kernel tracked when
 * lm_hip_score_into wrote `scores` (with lm_hip_scores_set_first_cell_rule(scores,
 * row_offset == 0)), else one pass over the shard; the record never leaves the device before
 * the all_gather. */
int lm_hip_argmax_sharded(lm_hip_ctx *ctx, lm_hip_comm *comm, const lm_hip_scores *scores,
                          size_t row_offset, int *found, lm_hip_coords *best, float *value);
/* The same in two halves, for a host that scores shard after shard (Scanner-style block loops,
 * scan.rs:174-178; the CLI's job loop, main.rs:502-561): _begin enqueues the shard record, its
 * all_gather and the read-back (the last two on a stream of the communicator's own) and returns at
 * once with a ticket; `scores` may be overwritten by the next lm_hip_score_into right away.  _end
 * waits for that merge only and applies lm_hip_combine_argmax.  At most two merges in flight per
 * communicator; every rank must issue the same sequence of collectives; one host thread per
 * communicator. */
int lm_hip_argmax_sharded_begin(lm_hip_ctx *ctx, lm_hip_comm *comm, const lm_hip_scores *scores,
                                size_t row_offset, int *ticket);
int lm_hip_argmax_sharded_end(lm_hip_ctx *ctx, lm_hip_comm *comm, int ticket, int *found,
                              lm_hip_coords *best, float *value);
/* Maximum::max (pli/mod.rs:158-160) when no NaN can occur (NaN contributions are dropped). */
int lm_hip_merge_max(lm_hip_ctx *ctx, lm_hip_comm *comm, int found_local, float value_local,
                     int *found, float *value);
/* StripedScores::threshold (scores.rs:207-213) of the whole matrix: this rank's row-major list
 * (rows relative to the shard) -> the concatenation over ranks in rank order with GLOBAL rows,
 * which is the reference's row-major push order (pli/mod.rs:212-218).  Counts travel by
 * all_gather, lists by one exact-length broadcast per rank.  *all is malloc'ed on every rank
 * (NULL when *n_all == 0); release with lm_hip_free. */
int lm_hip_merge_threshold(lm_hip_ctx *ctx, lm_hip_comm *comm, const lm_hip_coords *coords, size_t n,
                           size_t row_offset, lm_hip_coords **all, size_t *n_all);

/* ---- host-pointer convenience forms (synchronous, PCIe both ways) ---------- */

/* Exactly what a Rust shim can obtain from &StripedSequence / &DenseMatrix / &mut StripedScores: pageable host
 * matrices in, pageable host matrices out (csrc/hostptr.hip).  No context argument: every calling host thread is given
 * a lane of its own (context + stream, persistent staging, a cache of device PSSM tables keyed on the weights), so
 * threads overlap.  Lanes sit on ONE device unless told otherwise: $LM_HIP_DEVICE (read once per process; must name a
 * usable ordinal), else the HIP device current in the thread that made the first host-pointer call -- so a job of one
 * process per GPU never opens contexts on its neighbours' devices.  lm_hip_host_spread_lanes(1) deals new lanes
 * round-robin over every usable device instead: the reference's parallel axis is the CLI's worker threads over
 * (motif, sequence) jobs (lightmotif-cli main.rs:240-378), which then spread over the GPUs and PCIe links of a node with
 * one call added to the host; lm_hip_host_bind_thread places one thread.  Large calls (>= 96 MB of scores) run as a tile pipeline over a per-device ring of
 * pinned buffers (allocated on, and served by helper threads bound to, the GPU's NUMA node) and take turns on it.
 * 1 B per position travels up and 4 B down per lm_hip_score_f32 call. */
int lm_hip_score_f32(const uint8_t *seq, size_t seq_rows_total, size_t seq_stride, size_t cols,
                     size_t wrap, size_t length,
                     const float *pssm, size_t m, size_t pssm_stride, size_t k,
                     size_t row_begin, size_t row_end,
                     float *out, size_t out_stride, size_t *out_rows, size_t *max_index);
/* Score<u8, A, C>::score_rows_into with a DiscreteMatrix's weights on host matrices (pwm/mod.rs:754-791; the AVX2 impl is
 * pli/mod.rs:437-476 over avx2.rs:294-347) -- what Scanner::next calls per block (scan.rs:174-178).  `saturate` != 0: the
 * SIMD back-ends' saturating adds (avx2.rs:336), 0: Generic's wrapping `+=`.  Same conventions as lm_hip_score_f32. */
int lm_hip_score_u8_host(const uint8_t *seq, size_t seq_rows_total, size_t seq_stride, size_t cols,
                         size_t wrap, size_t length,
                         const uint8_t *weights, size_t m, size_t weights_stride, size_t k,
                         size_t row_begin, size_t row_end, int saturate,
                         uint8_t *out, size_t out_stride, size_t *out_rows, size_t *max_index);
int lm_hip_argmax_f32(const float *scores, size_t rows, size_t stride, size_t cols,
                      int *found, lm_hip_coords *best, float *value);
/* Maximum::max on a host matrix (pli/mod.rs:158-160; SURVEY 8b export list). */
int lm_hip_max_f32(const float *scores, size_t rows, size_t stride, size_t cols,
                   int *found, float *value);
int lm_hip_threshold_f32(const float *scores, size_t rows, size_t stride, size_t cols, float t,
                         lm_hip_coords **coords, size_t *n);
/* Scanner (scan.rs:96-250) on a HOST StripedSequence: what `Dispatch::Hip` specialises `Scanner` onto instead of driving
 * its 256-row block loop through the Score<u8> arm (a block is ~19 us of launch + link latency for 8 192 cells).  The striped
 * matrix goes up once (1 B per position), then lm_hip_scan_f32 / lm_hip_scan_max_f32 run on it; same results, same
 * conventions, same errors as those.  A host that scans one sequence with MANY motifs should upload it once
 * (lm_hip_seq_upload) and use the resident forms. */
int lm_hip_scan_f32_host(const uint8_t *seq, size_t seq_rows_total, size_t seq_stride, size_t cols,
                         size_t wrap, size_t length,
                         const float *pssm, size_t m, size_t pssm_stride, size_t k,
                         float threshold, lm_hip_hit **hits, size_t *n);
int lm_hip_scan_max_f32_host(const uint8_t *seq, size_t seq_rows_total, size_t seq_stride, size_t cols,
                             size_t wrap, size_t length,
                             const float *pssm, size_t m, size_t pssm_stride, size_t k,
                             const uint8_t *dweights, size_t dweights_stride, int saturate, unsigned level,
                             int have, size_t position, float score, size_t first_row,
                             int *found, lm_hip_hit *best);

/* ---- the `Dispatch::Hip` policy --------------------------------------------------------------------
 *
 * lightmotif/src/pli/dispatch.rs has seven `match self.backend` sites, and `Scanner` hard-codes `Pipeline<A, Dispatch>`
 * (scan.rs:166).  A `Hip` variant must say what it does at EACH of them: a call on host matrices pays a launch, a
 * synchronisation and the link both ways, so below some size -- and for operations that only move bytes -- the right arm is
 * the CPU tier the variant carries (`Hip { cpu: Avx2 | Sse2 | Neon | Generic }` = what Pipeline::dispatch() would have
 * picked without a GPU, pli/mod.rs:269-308), never `_ => Generic`.  lm_hip_host_crossover returns, per site, the number of
 * cells (rows x columns of the call) from which the host-pointer entry point is expected to beat ONE thread of the AVX2
 * tier: SIZE_MAX = the site stays on the CPU tier at every size (Encode, Stripe: one pass over bytes that a core streams
 * faster than the link carries them; Maximum / Threshold on u8: Scanner-internal, specialised away), 0 = always the GPU.
 * The constants are measured (tools/crossover.py on MI355X + EPYC 9575F, profiles/r05_crossover.json) and a function of
 * the motif length where the CPU's cost is (Score: CPU time ~ M x cells, GPU time ~ latency + 5 B x cells / link). */
typedef enum lm_hip_host_op {
    LM_HIP_OP_ENCODE = 0,        /* Encode::encode_into           dispatch.rs:58-77   -> CPU tier            */
    LM_HIP_OP_SCORE_F32 = 1,     /* Score<f32>::score_rows_into   dispatch.rs:79-108  -> lm_hip_score_f32     */
    LM_HIP_OP_SCORE_U8 = 2,      /* Score<u8>::score_rows_into    dispatch.rs:110-137 -> lm_hip_score_u8_host */
    LM_HIP_OP_STRIPE = 3,        /* Stripe::stripe_into           dispatch.rs:139-153 -> CPU tier            */
    LM_HIP_OP_MAXIMUM_F32 = 4,   /* Maximum<f32>::argmax / max    dispatch.rs:155-179 -> lm_hip_argmax_f32 / lm_hip_max_f32 */
    LM_HIP_OP_MAXIMUM_U8 = 5,    /* Maximum<u8>::argmax / max     dispatch.rs:181-205 -> CPU tier            */
    LM_HIP_OP_THRESHOLD_F32 = 6, /* Threshold<f32>::threshold     dispatch.rs:205     -> lm_hip_threshold_f32 */
    LM_HIP_OP_THRESHOLD_U8 = 7,  /* Threshold<u8>::threshold      dispatch.rs:207     -> CPU tier            */
    LM_HIP_OP_SCAN = 8           /* Scanner::next / max           scan.rs:166-249     -> lm_hip_scan_f32_host / lm_hip_scan_max_f32_host */
} lm_hip_host_op;
/* m = motif length (0 where it plays no part), k = alphabet size.  LM_HIP_ERR_BAD_ARGS for an unknown op. */
int lm_hip_host_crossover(int op, size_t m, size_t k, size_t *cells);
/* The reference picks its back-end per HOST, at run time (pli/mod.rs:269-308); so does the policy.  The model behind
 * lm_hip_host_crossover is  GPU call ~ t0 + g x cells,  CPU tier ~ (c + c_row x M) x cells,  crossover = t0 / (c + c_row M - g)
 * (never while the tier costs less than 1.25 x the link per cell).  The compiled constants are one box's (EPYC 9575F, PCIe
 * Gen5); these calls replace them with this process's own:
 *   lm_hip_host_calibrate   measures the GPU side (t0, g) of every site that can leave the tier by timing the host-pointer
 *                           entry points themselves on synthetic matrices (8 192 and 4 Mi cells), within ~budget_ms in total
 *                           (a few tens of milliseconds suffice).  Once per process unless `force`; thread-safe.
 *   lm_hip_host_set_cpu_cost  the CPU tier's side, which only the caller can time (its tier is Rust code the library never
 *                           sees): ns per cell, and ns per cell and motif row (0 where M plays no part).
 *   lm_hip_host_set_crossover pins the answer for a site whatever M (SIZE_MAX = never the GPU, 0 = always); (size_t)-2
 *                           (LM_HIP_CROSSOVER_MODEL) unpins it.
 *   lm_hip_host_cost_model  reads the constants in force and whether the GPU side was measured here.  Pointers may be NULL. */
#define LM_HIP_CROSSOVER_MODEL ((size_t)-2)
int lm_hip_host_calibrate(double budget_ms, int force);
int lm_hip_host_set_cpu_cost(int op, double ns_per_cell, double ns_per_cell_and_motif_row);
int lm_hip_host_set_crossover(int op, size_t cells);
int lm_hip_host_cost_model(int op, double *t0_us, double *gpu_ns_per_cell, double *cpu_ns_per_cell, double *cpu_ns_per_cell_and_motif_row,
                           int *gpu_side_measured);

/* Hands back what the host-pointer functions keep between calls: the pinned rings of the tile pipelines (128 MB of
 * page-locked memory per device used), their device tiles, and the device staging AND reduction scratch of this thread's
 * lane and of lanes whose threads have exited; lanes of other live threads hand theirs back at the end of their next
 * call.  Contexts and cached PSSM tables stay; the next call sets up again what it needs.  Safe at any time (waits for
 * a large call in flight); for hosts that score a genome and then sit idle. */
int lm_hip_host_trim(void);
/* != 0: NEW host-pointer lanes are dealt round-robin over every usable device (one process driving a whole node); 0 (the
 * default): every lane on the process's one host-pointer device (see above).  Lanes that exist stay where they are. */
int lm_hip_host_spread_lanes(int enabled);
/* != 0: a lane keeps the device copy of the score matrix its last lm_hip_score_f32 produced, and lm_hip_argmax_f32 /
 * lm_hip_max_f32 / lm_hip_threshold_f32 called by the same thread on that very matrix (same pointer, shape and stride) reduce
 * the copy instead of uploading the matrix again -- the reference's own benchmark shape, `score_into` followed by `argmax`
 * (lightmotif-bench/dna.rs:81-116: 150 -> ~110 us per iteration at 464 kbp).  The caller promises not to write to the matrix
 * between the two calls; the library checks a 64-bit digest of 67 sampled cells and falls back to the upload when it differs,
 * which catches a recycled buffer, not a deliberate single-cell edit.  0 (the default): every call uploads what it is given.
 * Process-wide; always LM_HIP_OK. */
int lm_hip_host_reuse_scores(int enabled);
/* How many calls of the calling thread's lane took the kept copy so far (a diagnostic for tests and tools). */
int lm_hip_host_reuse_count(size_t *count);
/* Puts the calling thread's host-pointer lane on `device` (a HIP ordinal from lm_hip_device_ordinal) from its next call
 * on; -1 = back to the automatic placement ($LM_HIP_DEVICE, else the process's device or the round-robin).  For hosts that place their worker
 * threads themselves (one thread per GPU, threads pinned next to their GPU). */
int lm_hip_host_bind_thread(int device);
/* Where the calling thread's lane is (creates it if need be): its device ordinal, the NUMA node the platform reports for
 * that GPU (-1: none reported) and how many CPUs of that node the pipeline's helper threads are confined to (0: unbound).
 * Any pointer may be NULL (not all three). */
int lm_hip_host_lane_info(int *device, int *numa_node, int *helper_cpus);

#ifdef __cplusplus
}
#endif
#endif /* LIGHTMOTIF_HIP_H */
