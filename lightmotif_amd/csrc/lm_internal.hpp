// lm_internal.hpp -- shared declarations of the gfx950 back-end (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "lightmotif_hip.h"

namespace lm {

// ---- error plumbing ---------------------------------------------------------

int fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define LM_HIP_TRY(expr)                                                               \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return ::lm::fail(_e == hipErrorOutOfMemory ? LM_HIP_ERR_OOM : LM_HIP_ERR_HIP, \
                              "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                              __FILE__, __LINE__);                                     \
    } while (0)

#define LM_TRY(expr)                 \
    do {                             \
        int _s = (expr);             \
        if (_s != LM_HIP_OK)         \
            return _s;               \
    } while (0)

// Makes `dev` current for the scope of a call, restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        if (prev != dev)
            ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

// ---- device scratch ---------------------------------------------------------

struct Scratch {
    void *ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t n);  // grows (never shrinks); contents are not preserved
    void release();
    void forget() { ptr = nullptr; bytes = 0; }  // leaks the block on purpose: work that may still write it is queued (comm.hip)
};

// Record written by the reductions (device) and read back through pinned memory.
struct ArgmaxRecord {
    float value;
    int found;
    long long index;  // flat index row * cols + col, -1 = none
};

constexpr size_t kPinnedBytes = 4u << 20;  // pinned staging buffer per context

// Host arrays handed to the caller (released with lm_hip_free = free).  Large ones are
// 2 MB-aligned and marked for transparent huge pages: a fresh 50 MB block otherwise takes
// ~13 000 first-touch page faults while the read-back lands in it (25 ms on the JASPAR batch,
// more than half of the scans it follows).
void *result_alloc(size_t bytes);
void result_free(void *p);  // what lm_hip_free does: back to the pool, or free()

}  // namespace lm

// ---- opaque handle definitions ------------------------------------------------

struct lm_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipStream_t aux_stream = nullptr;  // second lane for independent jobs of a batch
    hipEvent_t fork_event = nullptr, join_event = nullptr;
    // host -> device ingest (handles.hip: ingest_tiled): tiles are uploaded on `copy_stream` two ahead of the
    // stripe kernels that consume them on `stream`
    hipStream_t copy_stream = nullptr;
    hipEvent_t tile_copied[2] = {nullptr, nullptr}, tile_consumed[2] = {nullptr, nullptr};
    std::mutex mu;
    lm::Scratch scratch;        // block partials, chunk counts, hit lists
    lm::Scratch scratch2;
    lm::Scratch scan_buf;       // Scanner::max walk (scanmax.hip): one window of u8 scores and the walk's state
    lm::Scratch chunk_scores;   // fused reductions of sliced (M > 36) motifs: one chunk of f32 scores (score_launch.hpp)
    size_t chunk_rows = 1u << 22; // rows of that chunk (512 MB at C = 32; option "chunk_rows")
    bool chunked_fused = true;  // A/B knob: 0 = such motifs go cell by cell (option "chunked_fused")
    void *pinned = nullptr;     // kPinnedBytes of host-pinned memory for read-backs
    unsigned fold_generation = 0; // of the last single-job fused argmax whose kernel wrote its result into `pinned`
    unsigned *d_ticket = nullptr; // "last workgroup folds the records" counter of the single-launch argmax forms (zero between launches)
    unsigned *d_short = nullptr;  // short hit lists (hits.hip, ShortOrder): bucket counts | cursors | zero tiles | offsets; counts and cursors are zero between calls
    unsigned short_generation = 0;  // of the last short ordering whose end the host polled (hits.hip: done_flag)
    bool poll_done = false;       // option "poll_done" = 1: the short ordering's end is polled from pinned memory, not waited for (hits.hip)
    bool short_dirty = false;     // ... unless a call failed between the count and the clean-up: the next one clears them first
    size_t rows_per_stream = 0; // 0 = default
    bool use_prefilter = true;  // fused threshold: packed 16-bit discrete prefilter (A/B knob)
    bool pair_prefilter = true;  // DNA prefilter scans look up two symbols at a time
    bool block_prefilter = true;           // protein one-symbol scans fetch symbols in 4-row blocks (score_prefilter_blk.hpp); 0 = byte loads
    bool pair_prefilter_protein = false;  // the 441-row protein pair scan: correct, measured 4 % slower
    bool quad_loads = true;      // store kernel: quad-gathered dword symbol loads (M % 4 == 0; +1 %)
    bool track_argmax = true;    // score_into on handles also tracks the best cell (cached argmax)
    bool xlong_store = true;     // motifs of 65 ... kMaxStoreM rows are stored in one pass (option "xlong_store" = 0: slices of <= 64)
    bool host_fold = true;       // ... small matrices: per-wavefront records folded by the host (option "host_fold" = 0: on the device)
    bool speculate_order = true; // fused threshold: order the hit list before the host knows its length
    bool sort_hits = true;       // ... long lists by radix sort instead of the bucket passes (hits.hip; option "sort_hits")
    bool time_scan = false;      // diagnostic (option "time_scan"): events around the scan kernel(s) of a fused call -> last_scan_kernel_ms
    hipEvent_t scan_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // before the scans | behind them | behind the re-scoring | behind the ordering
    bool scan_timed = false;
    int scan_marks = 0;          // events 2 ... recorded in this call
    float last_scan_kernel_ms = -1.0f;
    // (time_scan) the last fused threshold call by phase: scan kernels, re-scoring, ordering kernels (events on the stream),
    // and the host's share behind the synchronisation (copy out of the pinned block, allocation) -- lm_hip_ctx_last_phases_ms
    float last_phase_ms[4] = {-1.0f, -1.0f, -1.0f, -1.0f};
    // what the scan of the last single-job fused call looked up per position (lm_hip_ctx_last_scan_info): motif rows and
    // bytes of LDS table; 0 = no scan kernel of the prefilter / exact families ran (a batch, the suffix route, chunks)
    unsigned last_scan_rows = 0, last_scan_lds_bytes = 0;
    bool list_scan_max = true;   // Scanner::max over a candidate list (scanmax.hip; option "list_scan_max" = 0: always the window walk)
    bool drop_last = true;       // single pair scans of M = 20, 24, ... 36 over M - 1 rows (option "drop_last"; lm_hip_pssm::d_image2_drop)
    bool short_order = true;     // ... short lists of one job counted by the re-scoring kernel, two launches behind it (hits.hip; option "short_order")
    bool suffix_argmax = true;   // fused argmax of short motifs: try the last rows first (score_argmax.hip)
    bool multi_motif = true;     // many-motif threshold batches: several motifs of one length per pass
    bool skip_unreachable = true; // fused threshold: no scan when the threshold exceeds the best k-mer's score
    bool tiled = true;           // column counts off the C = 32 / 16 kernels: LDS-tiled kernel (0: one thread per cell)
    double suffix_occurrences = 0; // fused argmax of short motifs: best k-mers the suffix should hold (0 = ln lambda, score_argmax.hip)
    int num_cus = 256;
    unsigned long long last_hit_count = 0;  // sizes the next fused-threshold hit list
    unsigned long long last_cand_count = 0; // ... and its candidate list
    const char *last_kernel = "";
    // lm_hip_score_u8*: device tables of the DiscreteMatrix used last (a scanner scores the same
    // matrix block after block); `u8_key` = {m, k, pairs} + the weights they were built from
    lm::Scratch u8_tables;
    std::vector<uint8_t> u8_key;
};

namespace lm {
// option "time_scan": the scan kernels of a fused call between two events on their stream, read after the call's
// synchronisation (lm_hip_ctx_last_scan_kernel_ms).  Off: nothing is recorded.
inline int scan_timer_begin(lm_hip_ctx *ctx, hipStream_t st)
{
    ctx->last_scan_kernel_ms = -1.0f;
    ctx->scan_timed = false;
    ctx->scan_marks = 0;
    for (float &x : ctx->last_phase_ms)
        x = -1.0f;
    if (!ctx->time_scan)
        return 0;
    for (hipEvent_t &e : ctx->scan_ev)
        if (!e && hipEventCreate(&e) != hipSuccess)
            return 0;  // (a diagnostic: the call goes on untimed)
    ctx->scan_timed = hipEventRecord(ctx->scan_ev[0], st) == hipSuccess;
    return 0;
}
inline void scan_timer_end(lm_hip_ctx *ctx, hipStream_t st)
{
    if (ctx->scan_timed)
        ctx->scan_timed = hipEventRecord(ctx->scan_ev[1], st) == hipSuccess;
}
// a further mark behind what has been enqueued so far: which = 2 (re-scoring done), 3 (ordering kernels done)
inline void scan_timer_mark(lm_hip_ctx *ctx, hipStream_t st, int which)
{
    if (ctx->scan_timed && which == ctx->scan_marks + 2 && hipEventRecord(ctx->scan_ev[which], st) == hipSuccess)
        ctx->scan_marks = which - 1;
}
inline void scan_timer_read(lm_hip_ctx *ctx)  // the stream has been synchronised
{
    float ms = -1.0f;
    if (ctx->scan_timed && hipEventElapsedTime(&ms, ctx->scan_ev[0], ctx->scan_ev[1]) == hipSuccess) {
        ctx->last_scan_kernel_ms = ms;
        ctx->last_phase_ms[0] = ms;
        for (int i = 1; i <= ctx->scan_marks; ++i)
            if (hipEventElapsedTime(&ms, ctx->scan_ev[i], ctx->scan_ev[i + 1]) == hipSuccess)
                ctx->last_phase_ms[i] = ms;
    }
    ctx->scan_timed = false;
}
}  // namespace lm

struct lm_hip_pssm {
    int device = 0;
    size_t m = 0, k = 0;
    std::vector<float> host;  // m x k, dense (row-major, stride k)
    // Transposed, padded copy for the C=32 kernels: table[s * ts + j] = pssm[j][s],
    // ts = floats per symbol row (multiple of 4, ts/4 odd), zero padded.
    float *d_table = nullptr;
    size_t ts = 0;
    // M % 4 != 0 (and M + lead <= 32): the store kernels' table with `lead` leading all-zero rows, so that
    // the padded length is a multiple of 4 and the dword symbol loads apply (0.0 + 0.0 + P[0] ... is the
    // same f32 sequence as 0.0 + P[0] ...: bit-identical)
    float *d_table_pad = nullptr;
    size_t lead = 0;
    // M > kMaxFastM (C = 32): slices of <= kMaxLongM rows, each with its own transposed table; the
    // first is scored with the store kernel, the others continue from the stored partial sums.
    // M <= kMaxLongM is ONE slice: the single-pass long kernels (score_plan.hip: exact_motif)
    struct Part {
        size_t off = 0, m = 0, ts = 0, lead = 0;  // `m` includes `lead` leading zero rows (table only)
        float *d_table = nullptr;
    };
    std::vector<Part> parts;
    // Row-major dense copy for the generic kernel: d_dense[j * k + s].
    float *d_dense = nullptr;
    // Discrete prefilter of the fused threshold scan (score_prefilter.hpp): LDS image
    // (u16 layout EVEN | u16 layout ODD) and the affine map
    // discrete = (score - pre_offset) / pre_factor, pre_emax = f32 rounding-error bound.
    unsigned *d_image = nullptr;
    unsigned *d_image2 = nullptr;  // DNA / protein: pair-symbol table of score_prefilter2.hpp (25 / 441 rows)
    // DNA, M = 20, 24, ... 36: the pair table of the first M - 1 rows.  The padded length of such a motif wastes a 16-byte
    // read per table row (M' = M + 3); a prefilter only has to over-estimate, so a single scan looks the first M - 1 rows up
    // and credits the last row with its best weight, `drop_dmax` (score_threshold.hip: drop_last_form)
    unsigned *d_image2_drop = nullptr;
    // DNA: the pair table in the layout the multi-motif passes of a batch read (8-byte LDS slots, score_prefilter2.hpp:
    // kDnaMulti); nullptr = those passes are not available for this matrix (the jobs run one motif per pass)
    unsigned *d_image2_multi = nullptr;
    unsigned drop_dmax = 0;
    bool has_prefilter = false;
    double pre_offset = 0, pre_factor = 0, pre_emax = 0;
};

struct lm_hip_seq {
    int device = 0;
    uint8_t *d_data = nullptr;
    size_t capacity_rows = 0;  // allocated rows
    size_t rows = 0;           // non-wrap rows = ceil(length / cols)
    size_t wrap = 0;
    size_t stride = 0, cols = 0, length = 0, k = 0;
    bool owns = true;          // false: d_data is the caller's (lm_hip_seq_adopt_dptr)
};

struct lm_hip_scores {
    int device = 0;
    float *d_data = nullptr;
    size_t capacity_rows = 0;
    size_t rows = 0, stride = 0, cols = 0, max_index = 0;
    // argmax of the matrix as tracked by the store kernel that last wrote it (device record;
    // valid until the next library write into this handle)
    lm::ArgmaxRecord *d_best = nullptr;
    bool best_valid = false;
    // pinned, device-visible copy of *d_best written by the same kernel (small score_into: MODE_STORE_TRACK):
    // lm_hip_argmax then needs a stream synchronisation and no copy command
    lm::ArgmaxRecord *h_best = nullptr;   // 64 pinned bytes: the record, then the generation word of the launch that wrote it
    bool best_on_host = false;
    unsigned best_generation = 0;         // of the last tracked launch into this handle
    // small inputs: the store kernel's per-wavefront records (16 bytes each, + the first-cell slot) in pinned memory,
    // folded by the host in lm_hip_argmax (handles.hip: host_fold); `folded` caches the fold of the current generation
    void *h_records = nullptr;
    size_t h_records_cap = 0;             // records the block has room for
    unsigned n_records = 0;               // wavefront records of the last launch (the first-cell slot follows them)
    bool records_on_host = false, folded = false;
    lm::ArgmaxRecord folded_record{};
    // 0: this StripedScores is a row shard that does NOT hold the matrix's first cell, so the
    // "scores[0][0] is NaN -> (0,0)" rule of Maximum::argmax is skipped (lm_hip_scores_set_first_cell_rule)
    bool first_cell_rule = true;
};

namespace lm {

// A dense result (a threshold every cell passes: 16 B per cell on the device while it is ordered and copied) must not
// leave the context holding tens of gigabytes for good: scratch above this is handed back when such a call ends.
constexpr size_t kScratchKeepBytes = (size_t)2 << 30;
struct ScratchTrim {
    lm_hip_ctx *ctx;
    explicit ScratchTrim(lm_hip_ctx *c) : ctx(c) {}
    ~ScratchTrim()
    {
        for (Scratch *s : {&ctx->scratch, &ctx->scratch2})
            if (s->bytes > kScratchKeepBytes) {
                (void)hipStreamSynchronize(ctx->stream);
                s->release();
            }
    }
};


// ArgmaxRecord -> the (found, coordinates, value) outputs of the ABI
inline void record_to_coords(const ArgmaxRecord &rec, size_t cols, int *found, lm_hip_coords *best,
                             float *value)
{
    if (found)
        *found = rec.found;
    if (rec.found) {
        if (best) {
            best->row = (size_t)(rec.index / (long long)cols);
            best->col = (size_t)(rec.index % (long long)cols);
        }
        if (value)
            *value = rec.value;
    }
}


// ---- kernel launchers (score_*.hip, reduce.hip, layout.hip) ----------------------

enum class ScoreMode { Store, Argmax, Threshold };

struct ScoreArgs {
    const lm_hip_pssm *pssm;
    const uint8_t *d_seq;     // row 0 of the striped matrix
    size_t seq_stride, cols;
    size_t row_begin, row_end;
    // Store
    float *d_out; size_t out_stride;
    // Store, optional (small inputs off the C = 32 kernels: score_tiled): per-wavefront (value, cell) records into this
    // pinned array of `track_cap` slots; *track_nrec receives the number of wavefront records written (0: none)
    uint4 *track_records = nullptr;
    unsigned track_generation = 0;
    size_t track_cap = 0;
    unsigned *track_nrec = nullptr;
};

// Geometry checks shared by every score entry point (avx2.rs:832-837: the wrap check; row range inside the matrix)
int check_score_args(const lm_hip_pssm *pssm, size_t seq_rows_total, size_t seq_stride, size_t cols, size_t wrap,
                     size_t row_begin, size_t row_end);

// Materialising score kernels.
int launch_score_store(lm_hip_ctx *ctx, const ScoreArgs &a);

// Score<u8, ..> with a DiscreteMatrix (score_store.hip; reductions in discrete.hip)
struct DiscreteArgs {
    const uint8_t *weights;   // HOST: M x wstride u8 (DenseMatrix<u8, K>, pwm/mod.rs:757)
    size_t m, wstride, k;
    const uint8_t *d_seq;     // row 0 of the striped matrix (device)
    size_t seq_stride, cols, row_begin, row_end;
    uint8_t *d_out;           // row 0 of the u8 scores = sequence row row_begin (device)
    size_t out_stride;
    bool saturate;            // true: avx2.rs:336 saturating adds; false: Generic's wrapping `+=`
};
int launch_score_u8(lm_hip_ctx *ctx, const DiscreteArgs &a);
// Folds the per-wavefront records a small tracking kernel left in pinned memory (FusedOut::host_records; n records +
// the first-cell slot, both halves of each carrying `generation`) with the Generic argmax rule.  Polls for their
// arrival; synchronises ctx->stream if they take too long.
int fold_host_records(lm_hip_ctx *ctx, const void *records, unsigned n, unsigned generation, bool first_cell_rule,
                      ArgmaxRecord *out);
// Scanner::max as the reference walks it (scan.rs:200-249), scanmax.hip; synchronises
int launch_scan_max(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq, const uint8_t *weights, size_t wstride,
                    bool saturate, unsigned level, bool have, unsigned long long position, float score, size_t first_row,
                    int *found, unsigned long long *best_position, float *best_score);
int launch_argmax_u8(lm_hip_ctx *ctx, const uint8_t *d_scores, size_t rows, size_t stride, size_t cols,
                     ArgmaxRecord *out);
int launch_threshold_u8(lm_hip_ctx *ctx, const uint8_t *d_scores, size_t rows, size_t stride, size_t cols,
                        unsigned t, lm_hip_coords **coords, size_t *n);
// ... also leaving the argmax of the written rows in *d_result (device); *tracked = false
// when the shape falls back to a plain store
int launch_score_store_argmax(lm_hip_ctx *ctx, const ScoreArgs &a, ArgmaxRecord *d_result, bool *tracked,
                              int first_cell_rule = 1);
// small inputs: store + (value, cell) tracking + the fold of the workgroup records in ONE launch
// (MODE_STORE_TRACK); the record also lands in *h_result (pinned, optional)
// the store of rows [a.row_begin, a.row_end) into scores->d_data (= a.d_out) that also tracks the best cell, as
// lm_hip_score_rows_into runs it: lm_hip_argmax on `scores` is then a read of the record(s) (handles.hip; the host-pointer
// lanes use it for lm_hip_host_reuse_scores)
int score_store_tracked(lm_hip_ctx *ctx, const ScoreArgs &a, lm_hip_scores *scores);
int launch_score_store_track(lm_hip_ctx *ctx, const ScoreArgs &a, ArgmaxRecord *d_result, ArgmaxRecord *h_result,
                             unsigned generation, bool *tracked, int first_cell_rule = 1, lm_hip_scores *host_fold = nullptr);
int finalize_argmax_materialised(lm_hip_ctx *ctx, const ArgmaxRecord *d_blocks, unsigned nblocks,
                                 const float *d_scores, int first_cell_rule, ArgmaxRecord *d_out);
// Fused score+argmax: leaves one ArgmaxRecord at ctx->scratch (device) -> out.
int launch_score_argmax(lm_hip_ctx *ctx, const ScoreArgs &a, int first_cell_rule, ArgmaxRecord *out);
// Result of a fused score+threshold batch: malloc'ed host arrays in key order (the
// caller takes them over or calls release()); job j owns [job_start[j], job_start[j+1]).
struct HitOutput {
    size_t total = 0;
    std::vector<size_t> job_start;
    lm_hip_coords *coords = nullptr;  // HitKeys::RowMajor: (row, col) + values
    float *values = nullptr;
    lm_hip_hit *hits = nullptr;       // HitKeys::Position: (sequence position, score)
    void release();
};

// Order of the hits of one job: the reference's row-major push order
// (pli/mod.rs:212-218), or ascending sequence position col * rows + row
// (scores.rs:155-157) for Scanner-style output.
enum class HitKeys { RowMajor, Position };

// Fused score+threshold of n independent jobs (row indices relative to each job's
// row_begin).  HitKeys::Position requires row_begin == 0 on every job.
int launch_score_threshold_batch(lm_hip_ctx *ctx, const ScoreArgs *jobs, const float *ts, size_t n,
                                 HitKeys keys, HitOutput *out);

// Batched fused score+argmax: n independent jobs, one stream synchronisation.
int launch_score_argmax_batch(lm_hip_ctx *ctx, const ScoreArgs *jobs, size_t n,
                              int first_cell_rule, ArgmaxRecord *out);

// hits.hip: device-side ordering of the fused kernels' hit list
struct HitRecord;
// Short lists of ONE job (a scan at p = 1e-5 leaves ~10^4 hits per Gbp): the re-scoring kernel counts the records per
// bucket as it stores them, so the ordering behind it is two launches instead of five (hits.hip).  `on` = the geometry
// below was handed to launch_rescore and order_hits must be called with the same object.
struct ShortOrder {
    bool on = false;
    int shift = 0;
    unsigned long long nb = 0;
    unsigned *counts = nullptr;
    // the list's two counters {hits, candidates} live in the context as well, zero between calls: the scans need no
    // head block copied in first (the single job reaches the re-scoring kernel as an argument)
    unsigned long long *counters = nullptr, *counters_copy = nullptr;
};
int short_order_begin(lm_hip_ctx *ctx, unsigned long long expected, size_t njobs, unsigned long long max_low, ShortOrder *so);
// count == ~0: speculative (the host has not read the counters yet); see hits.hip
int order_hits(lm_hip_ctx *ctx, const HitRecord *d_hits, const unsigned long long *d_counters,
               unsigned long long count, unsigned long long cap, unsigned long long cand_cap,
               unsigned long long expected, size_t njobs, unsigned long long max_low, int emit, size_t cols,
               HitOutput *out, int *status, unsigned long long counts_out[2], const ShortOrder *so = nullptr);

// reduce.hip: exclusive scan of n u32 counts (async on ctx->stream); the offset of
// element i is tiles[i / kScanTile] + offsets[i], *total the grand total.
constexpr int kScanTile = 1024;
int launch_scan_u32(lm_hip_ctx *ctx, const unsigned *counts, unsigned long long n,
                    unsigned long long *offsets, unsigned long long *tiles,
                    unsigned long long *total);

int launch_argmax(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride,
                  size_t cols, int first_cell_rule, ArgmaxRecord *out);
// Block records of one contiguous (stride == cols, 16-byte aligned) piece of a larger matrix: `grid`
// records at `d_blocks`, indices offset by `index_base`.  No synchronisation.
int launch_argmax_blocks_flat(lm_hip_ctx *ctx, hipStream_t stream, const float *d_scores, unsigned long long ncells,
                              long long index_base, unsigned grid, ArgmaxRecord *d_blocks);
int launch_argmax_device(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride,
                         size_t cols, int first_cell_rule, ArgmaxRecord *d_out);
int launch_threshold(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride,
                     size_t cols, float t, lm_hip_coords **coords, size_t *n);

int launch_encode(lm_hip_ctx *ctx, char alphabet, const uint8_t *d_ascii, size_t len, int lossy,
                  uint8_t *d_dst, size_t *bad_index);
// largest symbol byte among the `cols` live bytes of `rows` rows (synchronises)
int launch_max_symbol(lm_hip_ctx *ctx, const uint8_t *d_data, size_t rows, size_t stride, size_t cols,
                      unsigned *max_symbol);
int launch_stripe(lm_hip_ctx *ctx, const uint8_t *d_encoded, size_t len, size_t cols,
                  uint8_t default_symbol, size_t wrap, uint8_t *d_data, size_t stride);
// One tile of the striped matrix (layout.hip): output rows [rbase, rbase + nrows) of `rows`, from `cols` pieces of
// `pitch` bytes (piece c = the symbols of column c for those rows), converted on the way.  Async on ctx->stream.
struct StripeTile {
    enum Transform { None, Check, Ascii, TwoBit };
    const uint8_t *d_src = nullptr;   // pieces; TwoBit: the WHOLE packed sequence (4 bases per byte) + 16 spare bytes
    size_t pitch = 0;
    size_t len = 0, rows = 0;         // of the whole sequence
    size_t rbase = 0, nrows = 0;
    size_t cols = 0, stride = 0;
    uint8_t def = 0;
    uint8_t *d_data = nullptr;        // row 0 of the striped matrix
    Transform transform = None;
    size_t k = 0;                     // Check: alphabet size
    bool protein = false, lossy = false;          // Ascii
    const uint8_t *d_mask = nullptr;              // TwoBit: N mask (1 bit per base) or null, + 16 spare bytes
    unsigned long long *d_first_bad = nullptr;    // Check / Ascii: atomicMin of the first offending position
};
int launch_stripe_tile(lm_hip_ctx *ctx, const StripeTile &t);
// d_runs: nruns x {start, size} on the device; cells of those positions become `def` (async on ctx->stream)
int launch_n_runs(lm_hip_ctx *ctx, const unsigned long long *d_runs, size_t nruns, unsigned long long longest, size_t rows,
                  size_t stride, uint8_t def, uint8_t *d_data);
int launch_wrap(lm_hip_ctx *ctx, uint8_t *d_data, size_t rows, size_t stride, size_t cols,
                size_t new_wrap, uint8_t default_symbol);

}  // namespace lm
