// score_api.hip -- Score / Maximum / Threshold of the C ABI on DEVICE pointers, the fused forms and the batched scans.
//
// Pre-checks and result shapes follow the reference wrappers:
//   Avx2::score_f32_rows_into_permute (lightmotif/src/pli/platform/avx2.rs:817-851):
//     wrap check (:832-837), degenerate check (:839-842), scores.resize (:844)
//   Score::score_into (pli/mod.rs:109-117), StripedScores::{argmax,threshold} (scores.rs:181-213),
//   Scanner (scan.rs:150-250), the CLI's (motif, sequence) fan-out (lightmotif-cli main.rs:502-561).
#include <algorithm>
#include <cstring>

#include "lm_internal.hpp"

using namespace lm;

extern "C" {

// ---- Score (device pointers) -----------------------------------------------------------------

int lm_hip_score_f32_dptr(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const uint8_t *d_seq,
                          size_t seq_rows_total, size_t seq_stride, size_t cols, size_t wrap,
                          size_t length, size_t row_begin, size_t row_end, float *d_out,
                          size_t out_stride, size_t *out_rows, size_t *max_index)
{
    if (!ctx)
        return fail(LM_HIP_ERR_BAD_ARGS, "null context");
    LM_TRY(check_score_args(pssm, seq_rows_total, seq_stride, cols, wrap, row_begin, row_end));
    // pli/mod.rs:85-88
    if (length < pssm->m || row_begin >= row_end) {
        if (out_rows) *out_rows = 0;
        if (max_index) *max_index = 0;
        return LM_HIP_OK;
    }
    if (!d_seq || !d_out || out_stride < cols)
        return fail(LM_HIP_ERR_BAD_ARGS, "score: null buffer or out stride %zu < columns %zu",
                    out_stride, cols);
    if (out_rows) *out_rows = row_end - row_begin;       // pli/mod.rs:91
    if (max_index) *max_index = length + 1 - pssm->m;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    ScoreArgs a{pssm, d_seq, seq_stride, cols, row_begin, row_end, d_out, out_stride};
    return launch_score_store(ctx, a);
}

int lm_hip_score_u8_dptr(lm_hip_ctx *ctx, const uint8_t *weights, size_t m, size_t weights_stride,
                         size_t k, const uint8_t *d_seq, size_t seq_rows_total, size_t seq_stride,
                         size_t cols, size_t wrap, size_t length, size_t row_begin, size_t row_end,
                         uint8_t *d_out, size_t out_stride, int saturate, size_t *out_rows,
                         size_t *max_index)
{
    if (!ctx)
        return fail(LM_HIP_ERR_BAD_ARGS, "null context");
    if (!weights || m == 0 || k == 0 || weights_stride < k)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_u8: bad discrete matrix (%zu x %zu, stride %zu)", m, k,
                    weights_stride);
    lm_hip_pssm shape;  // the geometry checks only look at the motif length
    shape.m = m;
    shape.k = k;
    LM_TRY(check_score_args(&shape, seq_rows_total, seq_stride, cols, wrap, row_begin, row_end));
    if (length < m || row_begin >= row_end) {  // pli/mod.rs:85-88
        if (out_rows) *out_rows = 0;
        if (max_index) *max_index = 0;
        return LM_HIP_OK;
    }
    if (!d_seq || !d_out || out_stride < cols)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_u8: null buffer or out stride %zu < columns %zu", out_stride,
                    cols);
    if (out_rows) *out_rows = row_end - row_begin;  // pli/mod.rs:91
    if (max_index) *max_index = length + 1 - m;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    DiscreteArgs a{weights, m, weights_stride, k, d_seq, seq_stride, cols, row_begin, row_end, d_out,
                   out_stride, saturate != 0};
    return launch_score_u8(ctx, a);
}

int lm_hip_score_u8(lm_hip_ctx *ctx, const uint8_t *weights, size_t m, size_t weights_stride, size_t k,
                    const lm_hip_seq *seq, size_t row_begin, size_t row_end, int saturate, uint8_t *out,
                    size_t out_stride, size_t *out_rows, size_t *max_index)
{
    if (!ctx || !seq)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_u8: null argument");
    if (!weights || m == 0 || k == 0 || weights_stride < k)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_u8: bad discrete matrix (%zu x %zu, stride %zu)", m, k,
                    weights_stride);
    lm_hip_pssm shape;
    shape.m = m;
    shape.k = k;
    LM_TRY(check_score_args(&shape, seq->rows + seq->wrap, seq->stride, seq->cols, seq->wrap, row_begin,
                            row_end));
    if (seq->length < m || row_begin >= row_end) {  // pli/mod.rs:85-88
        if (out_rows) *out_rows = 0;
        if (max_index) *max_index = 0;
        return LM_HIP_OK;
    }
    if (!out || out_stride < seq->cols)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_u8: null buffer or out stride %zu < columns %zu", out_stride,
                    seq->cols);
    if (out_rows) *out_rows = row_end - row_begin;
    if (max_index) *max_index = seq->length + 1 - m;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    const size_t nrows = row_end - row_begin, cols = seq->cols;
    if (nrows * cols + 4096 <= kPinnedBytes / 2 && nrows * cols <= (128u << 10)) {
        // a Scanner block (scan.rs:174-178: 256 rows): the kernel writes the u8 scores straight into pinned host
        // memory -- no copy command, one synchronisation (tools/kbench/hostpipe_bench.hip: 16 us against 28 us)
        uint8_t *z_out = static_cast<uint8_t *>(ctx->pinned) + 4096;
        DiscreteArgs a{weights, m, weights_stride, k, seq->d_data, seq->stride, cols, row_begin, row_end, z_out, cols,
                       saturate != 0};
        LM_TRY(launch_score_u8(ctx, a));
        LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (out_stride == cols)
            memcpy(out, z_out, nrows * cols);
        else
            for (size_t r = 0; r < nrows; ++r)
                memcpy(out + r * out_stride, z_out + r * cols, cols);
        return LM_HIP_OK;
    }
    LM_TRY(ctx->scratch.reserve(nrows * cols));
    uint8_t *d_out = static_cast<uint8_t *>(ctx->scratch.ptr);
    DiscreteArgs a{weights, m, weights_stride, k, seq->d_data, seq->stride, cols, row_begin, row_end, d_out,
                   cols, saturate != 0};
    LM_TRY(launch_score_u8(ctx, a));
    LM_HIP_TRY(out_stride == cols
                   ? hipMemcpyAsync(out, d_out, nrows * cols, hipMemcpyDeviceToHost, ctx->stream)
                   : hipMemcpy2DAsync(out, out_stride, d_out, cols, cols, nrows, hipMemcpyDeviceToHost,
                                      ctx->stream));
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return LM_HIP_OK;
}

int lm_hip_argmax_u8_dptr(lm_hip_ctx *ctx, const uint8_t *d_scores, size_t rows, size_t stride, size_t cols,
                          int *found, lm_hip_coords *best, uint8_t *value)
{
    if (!ctx || !found)
        return fail(LM_HIP_ERR_BAD_ARGS, "argmax_u8: null argument");
    *found = 0;
    if (rows == 0)  // pli/mod.rs:136-138
        return LM_HIP_OK;
    if (!d_scores || cols == 0 || stride < cols)
        return fail(LM_HIP_ERR_BAD_ARGS, "argmax_u8: bad matrix (stride %zu, columns %zu)", stride, cols);
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    ArgmaxRecord rec{};
    LM_TRY(launch_argmax_u8(ctx, d_scores, rows, stride, cols, &rec));
    *found = rec.found;
    if (rec.found) {
        if (best) {
            best->row = (size_t)(rec.index / (long long)cols);
            best->col = (size_t)(rec.index % (long long)cols);
        }
        if (value)
            *value = (uint8_t)rec.value;
    }
    return LM_HIP_OK;
}

int lm_hip_threshold_u8_dptr(lm_hip_ctx *ctx, const uint8_t *d_scores, size_t rows, size_t stride,
                             size_t cols, uint8_t t, lm_hip_coords **coords, size_t *n)
{
    if (!ctx || !coords || !n)
        return fail(LM_HIP_ERR_BAD_ARGS, "threshold_u8: null argument");
    *coords = nullptr;
    *n = 0;
    if (rows == 0)
        return LM_HIP_OK;
    if (!d_scores || cols == 0 || stride < cols)
        return fail(LM_HIP_ERR_BAD_ARGS, "threshold_u8: bad matrix (stride %zu, columns %zu)", stride, cols);
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    ScratchTrim trim(ctx);
    return launch_threshold_u8(ctx, d_scores, rows, stride, cols, t, coords, n);
}

int lm_hip_argmax_f32_dptr(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride,
                           size_t cols, int *found, lm_hip_coords *best, float *value)
{
    return lm_hip_argmax_shard_f32_dptr(ctx, d_scores, rows, stride, cols, 1, found, best, value);
}

int lm_hip_argmax_shard_f32_dptr(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride,
                                 size_t cols, int first_cell_rule, int *found, lm_hip_coords *best,
                                 float *value)
{
    if (!ctx || !found)
        return fail(LM_HIP_ERR_BAD_ARGS, "argmax: null argument");
    *found = 0;
    if (rows == 0)  // pli/mod.rs:136-138
        return LM_HIP_OK;
    if (!d_scores || cols == 0 || stride < cols)
        return fail(LM_HIP_ERR_BAD_ARGS, "argmax: bad matrix (stride %zu, columns %zu)", stride, cols);
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    ArgmaxRecord rec{};
    LM_TRY(launch_argmax(ctx, d_scores, rows, stride, cols, first_cell_rule, &rec));
    record_to_coords(rec, cols, found, best, value);
    return LM_HIP_OK;
}

int lm_hip_threshold_f32_dptr(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride,
                              size_t cols, float t, lm_hip_coords **coords, size_t *n)
{
    if (!ctx || !coords || !n)
        return fail(LM_HIP_ERR_BAD_ARGS, "threshold: null argument");
    *coords = nullptr;
    *n = 0;
    if (rows == 0)
        return LM_HIP_OK;
    if (!d_scores || cols == 0 || stride < cols)
        return fail(LM_HIP_ERR_BAD_ARGS, "threshold: bad matrix (stride %zu, columns %zu)", stride, cols);
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    ScratchTrim trim(ctx);
    return launch_threshold(ctx, d_scores, rows, stride, cols, t, coords, n);
}

// ---- fused ---------------------------------------------------------------------------------------

int lm_hip_score_argmax_f32_dptr(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const uint8_t *d_seq,
                                 size_t seq_rows_total, size_t seq_stride, size_t cols, size_t wrap,
                                 size_t length, size_t row_begin, size_t row_end, int *found,
                                 lm_hip_coords *best, float *value)
{
    return lm_hip_score_argmax_shard_f32_dptr(ctx, pssm, d_seq, seq_rows_total, seq_stride, cols,
                                              wrap, length, row_begin, row_end, 1, found, best,
                                              value);
}

int lm_hip_score_argmax_shard_f32_dptr(lm_hip_ctx *ctx, const lm_hip_pssm *pssm,
                                       const uint8_t *d_seq, size_t seq_rows_total,
                                       size_t seq_stride, size_t cols, size_t wrap, size_t length,
                                       size_t row_begin, size_t row_end, int first_cell_rule,
                                       int *found, lm_hip_coords *best, float *value)
{
    if (!ctx || !found)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_argmax: null argument");
    *found = 0;
    LM_TRY(check_score_args(pssm, seq_rows_total, seq_stride, cols, wrap, row_begin, row_end));
    if (length < pssm->m || row_begin >= row_end)
        return LM_HIP_OK;  // empty scores -> None
    if (!d_seq)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_argmax: null sequence");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    ScoreArgs a{pssm, d_seq, seq_stride, cols, row_begin, row_end, nullptr, 0};
    ArgmaxRecord rec{};
    LM_TRY(launch_score_argmax(ctx, a, first_cell_rule, &rec));
    record_to_coords(rec, cols, found, best, value);
    return LM_HIP_OK;
}

int lm_hip_score_threshold_f32_dptr(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const uint8_t *d_seq,
                                    size_t seq_rows_total, size_t seq_stride, size_t cols,
                                    size_t wrap, size_t length, size_t row_begin, size_t row_end,
                                    float t, lm_hip_coords **coords, float **values, size_t *n)
{
    if (!ctx || !coords || !n)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_threshold: null argument");
    *coords = nullptr;
    if (values)
        *values = nullptr;
    *n = 0;
    LM_TRY(check_score_args(pssm, seq_rows_total, seq_stride, cols, wrap, row_begin, row_end));
    if (length < pssm->m || row_begin >= row_end)
        return LM_HIP_OK;
    if (!d_seq)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_threshold: null sequence");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    ScoreArgs a{pssm, d_seq, seq_stride, cols, row_begin, row_end, nullptr, 0};
    HitOutput ho;
    ScratchTrim trim(ctx);
    LM_TRY(launch_score_threshold_batch(ctx, &a, &t, 1, HitKeys::RowMajor, &ho));
    *coords = ho.coords;
    if (values)
        *values = ho.values;
    else
        result_free(ho.values);
    *n = ho.total;
    return LM_HIP_OK;
}

// ---- many motifs x one resident sequence -----------------------------------------------------------------

static int batch_jobs(const lm_hip_pssm *const *pssms, size_t n, const lm_hip_seq *seq,
                      std::vector<ScoreArgs> *jobs, std::vector<char> *degenerate)
{
    if (!pssms || !seq)
        return fail(LM_HIP_ERR_BAD_ARGS, "scan batch: null argument");
    jobs->clear();
    degenerate->assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
        LM_TRY(check_score_args(pssms[i], seq->rows + seq->wrap, seq->stride, seq->cols, seq->wrap,
                                0, seq->rows));
        if (seq->length < pssms[i]->m || seq->rows == 0)
            (*degenerate)[i] = 1;  // pli/mod.rs:85-88: empty scores
        jobs->push_back(ScoreArgs{pssms[i], seq->d_data, seq->stride, seq->cols, 0, seq->rows,
                                  nullptr, 0});
    }
    return LM_HIP_OK;
}

int lm_hip_scan_argmax_batch(lm_hip_ctx *ctx, const lm_hip_pssm *const *pssms, size_t n,
                             const lm_hip_seq *seq, int *found, lm_hip_coords *best, float *value)
{
    if (!ctx || (n && !found))
        return fail(LM_HIP_ERR_BAD_ARGS, "scan_argmax_batch: null argument");
    std::vector<ScoreArgs> jobs;
    std::vector<char> degenerate;
    LM_TRY(batch_jobs(pssms, n, seq, &jobs, &degenerate));
    std::vector<ScoreArgs> live;
    std::vector<size_t> live_idx;
    for (size_t i = 0; i < n; ++i) {
        found[i] = 0;
        if (!degenerate[i]) {
            live.push_back(jobs[i]);
            live_idx.push_back(i);
        }
    }
    if (live.empty())
        return LM_HIP_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    std::vector<ArgmaxRecord> recs(live.size());
    LM_TRY(launch_score_argmax_batch(ctx, live.data(), live.size(), 1, recs.data()));
    for (size_t k = 0; k < live.size(); ++k) {
        const size_t i = live_idx[k];
        record_to_coords(recs[k], seq->cols, &found[i], best ? &best[i] : nullptr,
                         value ? &value[i] : nullptr);
    }
    return LM_HIP_OK;
}

int lm_hip_scan_threshold_batch(lm_hip_ctx *ctx, const lm_hip_pssm *const *pssms,
                                const float *thresholds, size_t n, const lm_hip_seq *seq,
                                size_t *counts, lm_hip_coords **coords, float **values)
{
    if (!ctx || !coords || (n && (!counts || !thresholds)))
        return fail(LM_HIP_ERR_BAD_ARGS, "scan_threshold_batch: null argument");
    *coords = nullptr;
    if (values)
        *values = nullptr;
    std::vector<ScoreArgs> jobs;
    std::vector<char> degenerate;
    LM_TRY(batch_jobs(pssms, n, seq, &jobs, &degenerate));
    std::vector<ScoreArgs> live;
    std::vector<float> live_t;
    std::vector<size_t> live_idx;
    for (size_t i = 0; i < n; ++i) {
        counts[i] = 0;
        if (!degenerate[i]) {
            live.push_back(jobs[i]);
            live_t.push_back(thresholds[i]);
            live_idx.push_back(i);
        }
    }
    if (live.empty())
        return LM_HIP_OK;
    HitOutput ho;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard guard(ctx->device);
        ScratchTrim trim(ctx);
        LM_TRY(launch_score_threshold_batch(ctx, live.data(), live_t.data(), live.size(),
                                            HitKeys::RowMajor, &ho));
    }
    // live jobs are in caller order, so the concatenated list already is the output
    for (size_t k = 0; k < live.size(); ++k)
        counts[live_idx[k]] = ho.job_start[k + 1] - ho.job_start[k];
    *coords = ho.coords;
    if (values)
        *values = ho.values;
    else
        result_free(ho.values);
    return LM_HIP_OK;
}

int lm_hip_scan_f32(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq,
                    float threshold, lm_hip_hit **hits, size_t *n)
{
    if (!ctx || !pssm || !seq || !hits || !n)
        return fail(LM_HIP_ERR_BAD_ARGS, "scan: null argument");
    *hits = nullptr;
    *n = 0;
    LM_TRY(check_score_args(pssm, seq->rows + seq->wrap, seq->stride, seq->cols, seq->wrap, 0,
                            seq->rows));
    if (seq->length < pssm->m || seq->rows == 0)
        return LM_HIP_OK;
    HitOutput ho;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard guard(ctx->device);
        ScoreArgs a{pssm, seq->d_data, seq->stride, seq->cols, 0, seq->rows, nullptr, 0};
        ScratchTrim trim(ctx);
        // keys are sequence positions col * rows + row (scan.rs:185, scores.rs:155-157),
        // so the list comes back in ascending position
        LM_TRY(launch_score_threshold_batch(ctx, &a, &threshold, 1, HitKeys::Position, &ho));
    }
    // scan.rs:186: only positions where the whole motif fits; the others are cells of
    // the padded tail, i.e. the largest positions = the end of the list
    size_t keep = ho.total;
    while (keep && ho.hits[keep - 1].position + pssm->m > seq->length)
        --keep;
    if (keep == 0) {
        ho.release();
        return LM_HIP_OK;
    }
    *hits = ho.hits;
    *n = keep;
    return LM_HIP_OK;
}

// Scanner::max with the reference's own walk (scan.rs:200-249): see scanmax.hip.
int lm_hip_scan_max_f32(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq, const uint8_t *dweights,
                        size_t dweights_stride, int saturate, unsigned level, int have, size_t position, float score,
                        size_t first_row, int *found, lm_hip_hit *best)
{
    if (!ctx || !pssm || !seq || !dweights || !found || !best)
        return fail(LM_HIP_ERR_BAD_ARGS, "scan_max: null argument");
    if (dweights_stride < pssm->k)
        return fail(LM_HIP_ERR_BAD_ARGS, "scan_max: discrete weights stride %zu < alphabet size %zu", dweights_stride, pssm->k);
    LM_TRY(check_score_args(pssm, seq->rows + seq->wrap, seq->stride, seq->cols, seq->wrap, 0, seq->rows));
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    unsigned long long pos = position;
    float sc = score;
    LM_TRY(launch_scan_max(ctx, pssm, seq, dweights, dweights_stride, saturate != 0, level, have != 0, pos, sc, first_row, found,
                           &pos, &sc));
    best->position = (size_t)pos;
    best->score = sc;
    return LM_HIP_OK;
}

}  // extern "C"
