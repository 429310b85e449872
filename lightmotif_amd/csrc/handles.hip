// handles.hip -- resident StripedSequence / StripedScores handles of the C ABI: Encode / Stripe / configure_wrap on the
// device (pli/mod.rs:47-66, 178-200; seq.rs:362-381), host -> device ingest tile by tile, score_into / argmax / max /
// threshold on handles (pli/mod.rs:109-117, scores.rs:181-213).
#include <algorithm>
#include <cstring>
#include <new>

#include "lm_internal.hpp"

using namespace lm;

extern "C" {

// ---- Encode / Stripe (device pointers) ---------------------------------------------------------------

int lm_hip_encode_dptr(lm_hip_ctx *ctx, char alphabet, const uint8_t *d_ascii, size_t len, int lossy,
                       uint8_t *d_dst, size_t *bad_index)
{
    if (!ctx || (len && (!d_ascii || !d_dst)))
        return fail(LM_HIP_ERR_BAD_ARGS, "encode: null argument");
    if (alphabet != 'D' && alphabet != 'P')
        return fail(LM_HIP_ERR_BAD_ARGS, "encode: alphabet must be 'D' or 'P'");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    return launch_encode(ctx, alphabet, d_ascii, len, lossy, d_dst, bad_index);
}

int lm_hip_stripe_dptr(lm_hip_ctx *ctx, const uint8_t *d_encoded, size_t len, size_t cols,
                       uint8_t default_symbol, size_t wrap, uint8_t *d_data, size_t stride)
{
    if (!ctx || cols == 0 || stride < cols || (len && (!d_encoded || !d_data)))
        return fail(LM_HIP_ERR_BAD_ARGS, "stripe: bad argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    return launch_stripe(ctx, d_encoded, len, cols, default_symbol, wrap, d_data, stride);
}

int lm_hip_configure_wrap_dptr(lm_hip_ctx *ctx, uint8_t *d_data, size_t rows, size_t stride,
                               size_t cols, size_t new_wrap, uint8_t default_symbol)
{
    if (!ctx || cols == 0 || stride < cols || (new_wrap && !d_data))
        return fail(LM_HIP_ERR_BAD_ARGS, "configure_wrap: bad argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    return launch_wrap(ctx, d_data, rows, stride, cols, new_wrap, default_symbol);
}

// ---- resident handles ---------------------------------------------------------------------------------

// Symbols are enums in the reference (abc.rs:113-135, 231-256); through the C ABI they are
// bytes, and a byte >= k would index past the M x K tables of the kernels.
static int check_symbols(lm_hip_ctx *ctx, const uint8_t *d_data, size_t rows, size_t stride, size_t cols,
                         size_t k, const char *what)
{
    unsigned mx = 0;
    LM_TRY(launch_max_symbol(ctx, d_data, rows, stride, cols, &mx));
    if (mx >= k)
        return fail(LM_HIP_ERR_INVALID_SYMBOL, "%s: symbol byte %u is not below the alphabet size %zu", what, mx, k);
    return LM_HIP_OK;
}

static int seq_alloc(lm_hip_ctx *ctx, size_t rows, size_t stride, size_t cols, size_t length,
                     size_t k, lm_hip_seq **out, size_t min_capacity_rows = 0)
{
    lm_hip_seq *s = new (std::nothrow) lm_hip_seq();
    if (!s)
        return fail(LM_HIP_ERR_OOM, "out of host memory");
    s->device = ctx->device;
    s->rows = rows;
    s->stride = stride;
    s->cols = cols;
    s->length = length;
    s->k = k;
    s->capacity_rows = std::max(rows + 32, min_capacity_rows);  // seq.rs:285 DEFAULT_EXTRA_ROWS
    hipError_t e = hipMalloc(&s->d_data, s->capacity_rows * stride);
    if (e != hipSuccess) {
        delete s;
        return fail(LM_HIP_ERR_OOM, "hipMalloc(sequence) failed: %s", hipGetErrorString(e));
    }
    *out = s;
    return LM_HIP_OK;
}

int lm_hip_seq_upload(lm_hip_ctx *ctx, const uint8_t *data, size_t rows_total, size_t stride,
                      size_t cols, size_t wrap, size_t length, size_t k, lm_hip_seq **out)
{
    if (!ctx || !out || (rows_total && !data))
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_upload: null argument");
    *out = nullptr;
    if (cols == 0 || stride < cols || wrap > rows_total || k == 0 || k > 256)
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_upload: bad geometry");
    if ((rows_total - wrap) * cols < length)  // seq.rs:303-304 InvalidData
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_upload: matrix stores fewer than %zu symbols", length);
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    lm_hip_seq *s = nullptr;
    // any wrap the caller's matrix already has (configure_wrap(max_m) of the CLI, M = 40 / 64 ...)
    LM_TRY(seq_alloc(ctx, rows_total - wrap, stride, cols, length, k, &s, rows_total));
    s->wrap = wrap;
    if (rows_total) {
        hipError_t e = hipMemcpyAsync(s->d_data, data, rows_total * stride, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            lm_hip_seq_destroy(s);
            return fail(LM_HIP_ERR_HIP, "sequence upload failed: %s", hipGetErrorString(e));
        }
        const int st = check_symbols(ctx, s->d_data, rows_total, stride, cols, k, "seq_upload");
        if (st != LM_HIP_OK) {
            lm_hip_seq_destroy(s);
            return st;
        }
    }
    *out = s;
    return LM_HIP_OK;
}

int lm_hip_seq_adopt_dptr(lm_hip_ctx *ctx, uint8_t *d_data, size_t rows_total, size_t capacity_rows,
                          size_t stride, size_t cols, size_t wrap, size_t length, size_t k, lm_hip_seq **out)
{
    if (!ctx || !out || (rows_total && !d_data))
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_adopt: null argument");
    *out = nullptr;
    if (cols == 0 || stride < cols || wrap > rows_total || capacity_rows < rows_total || k == 0 || k > 256)
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_adopt: bad geometry");
    lm_hip_seq *s = new (std::nothrow) lm_hip_seq();
    if (!s)
        return fail(LM_HIP_ERR_OOM, "out of host memory");
    s->device = ctx->device;
    s->d_data = d_data;
    s->owns = false;
    s->capacity_rows = capacity_rows;
    s->rows = rows_total - wrap;
    s->wrap = wrap;
    s->stride = stride;
    s->cols = cols;
    s->length = length;
    s->k = k;
    *out = s;
    return LM_HIP_OK;
}

// ---- host sequence -> resident StripedSequence (Encode + Stripe, pli/mod.rs:56-66, 178-200) ------------
//
// The caller's buffer (ASCII text or symbol bytes, pageable) goes to the device TILE by tile: tile t = rows
// [t * TR, (t + 1) * TR) of the striped matrix needs, for every column c, the TR bytes at position c * R + t * TR
// -- one strided 2-D copy (pitch R) into one of two staging tiles on the copy stream, while the stripe kernel of
// the previous tile (conversion fused: layout.hip) runs on the context's stream.  Scratch = two tiles (64 MB)
// whatever the genome's size, nothing is staged twice, and the H2D transfer -- the floor of this step: a
// pageable 1 GB buffer moves at ~55 GB/s on this host, 18 ms -- hides the 0.8 ms/Gbp of kernels behind it.
// (Round 2 copied the whole text into a 2 x len scratch with one hipMemcpyAsync, then encoded, then striped.)
static int ingest_streams(lm_hip_ctx *ctx)
{
    if (ctx->copy_stream)
        return LM_HIP_OK;
    // made into locals, committed to the context only when all of them exist: a failure part-way leaks nothing and a
    // later call starts over
    hipStream_t st = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; ++i)
        e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        for (hipEvent_t x : ev)
            if (x)
                (void)hipEventDestroy(x);
        if (st)
            (void)hipStreamDestroy(st);
        return fail(LM_HIP_ERR_HIP, "ingest streams: %s", hipGetErrorString(e));
    }
    for (int b = 0; b < 2; ++b) {
        ctx->tile_copied[b] = ev[2 * b];
        ctx->tile_consumed[b] = ev[2 * b + 1];
    }
    ctx->copy_stream = st;
    return LM_HIP_OK;
}

constexpr size_t kIngestTileBytes = 32u << 20;

static int ingest_tiled(lm_hip_ctx *ctx, const uint8_t *host, size_t len, size_t cols, size_t k,
                        StripeTile::Transform transform, bool protein, bool lossy, lm_hip_seq **out, size_t *bad_index,
                        const char *what)
{
    const size_t rows = (len + cols - 1) / cols;  // pli/mod.rs:182
    const size_t stride = lm_hip_stride(cols, 1);
    lm_hip_seq *s = nullptr;
    LM_TRY(seq_alloc(ctx, rows, stride, cols, len, k, &s));
    auto give_up = [&](int st) {
        if (ctx->copy_stream)
            (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(ctx->stream);
        lm_hip_seq_destroy(s);
        return st;
    };
    if (rows == 0) {
        *out = s;
        return LM_HIP_OK;
    }
    {
        const int sst = ingest_streams(ctx);
        if (sst != LM_HIP_OK)
            return give_up(sst);
    }
    // rows per tile: ~32 MB of input, a multiple of the stripe kernels' workgroup tile
    size_t tr = std::max<size_t>(kIngestTileBytes / cols / 1024 * 1024, 1024);
    tr = std::min(tr, (rows + 15) / 16 * 16);
    const size_t tile_bytes = tr * cols;
    int st = ctx->scratch2.reserve(2 * tile_bytes + 64);
    if (st != LM_HIP_OK)
        return give_up(st);
    uint8_t *stage = static_cast<uint8_t *>(ctx->scratch2.ptr);
    unsigned long long *d_bad = reinterpret_cast<unsigned long long *>(stage + 2 * tile_bytes);  // 8-byte aligned: tile_bytes % 16 == 0
    hipError_t e = hipMemsetAsync(d_bad, 0xff, 8, ctx->stream);
    // the staging tiles may still be read by kernels enqueued earlier on the context's stream
    if (e == hipSuccess)
        e = hipEventRecord(ctx->tile_consumed[0], ctx->stream);
    if (e == hipSuccess)
        e = hipEventRecord(ctx->tile_consumed[1], ctx->stream);
    for (size_t t = 0, rbase = 0; rbase < rows && e == hipSuccess; ++t, rbase += tr) {
        const int b = (int)(t & 1);
        const size_t w = std::min(tr, rows - rbase);
        uint8_t *dst = stage + (size_t)b * tile_bytes;
        e = hipStreamWaitEvent(ctx->copy_stream, ctx->tile_consumed[b], 0);
        // columns whose w bytes all exist: c * rows + rbase + w <= len; then at most one partial column
        const size_t nfull = len >= rbase + w ? std::min(cols, (len - rbase - w) / rows + 1) : 0;
        if (e == hipSuccess && nfull)
            e = hipMemcpy2DAsync(dst, tr, host + rbase, rows, w, nfull, hipMemcpyHostToDevice, ctx->copy_stream);
        if (e == hipSuccess && nfull < cols && nfull * rows + rbase < len)
            e = hipMemcpyAsync(dst + nfull * tr, host + nfull * rows + rbase, len - (nfull * rows + rbase),
                               hipMemcpyHostToDevice, ctx->copy_stream);
        if (e == hipSuccess)
            e = hipEventRecord(ctx->tile_copied[b], ctx->copy_stream);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(ctx->stream, ctx->tile_copied[b], 0);
        if (e != hipSuccess)
            break;
        StripeTile tile;
        tile.d_src = dst;
        tile.pitch = tr;
        tile.len = len;
        tile.rows = rows;
        tile.rbase = rbase;
        tile.nrows = w;
        tile.cols = cols;
        tile.stride = stride;
        tile.def = (uint8_t)(k - 1);
        tile.d_data = s->d_data;
        tile.transform = transform;
        tile.k = k;
        tile.protein = protein;
        tile.lossy = lossy;
        tile.d_first_bad = d_bad;
        st = launch_stripe_tile(ctx, tile);
        if (st != LM_HIP_OK)
            return give_up(st);
        e = hipEventRecord(ctx->tile_consumed[b], ctx->stream);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(ctx->pinned, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess)
        return give_up(fail(LM_HIP_ERR_HIP, "%s: upload failed: %s", what, hipGetErrorString(e)));
    const unsigned long long bad = *static_cast<unsigned long long *>(ctx->pinned);
    if (bad != ~0ull) {
        lm_hip_seq_destroy(s);
        if (bad_index)
            *bad_index = (size_t)bad;
        if (transform == StripeTile::Check)
            return fail(LM_HIP_ERR_INVALID_SYMBOL, "%s: symbol byte %u at position %llu is not below the alphabet size %zu",
                        what, (unsigned)host[bad], bad, k);
        return fail(LM_HIP_ERR_INVALID_SYMBOL, "invalid symbol at position %llu", bad);
    }
    *out = s;
    return LM_HIP_OK;
}

int lm_hip_seq_from_encoded(lm_hip_ctx *ctx, const uint8_t *encoded, size_t len, size_t cols,
                            size_t k, lm_hip_seq **out)
{
    if (!ctx || !out || (len && !encoded) || cols == 0 || k == 0 || k > 256)
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_from_encoded: bad argument");
    *out = nullptr;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    // bytes from the caller: validated against the alphabet size on the way (abc.rs:113-135: symbols are enums)
    return ingest_tiled(ctx, encoded, len, cols, k, StripeTile::Check, false, false, out, nullptr, "seq_from_encoded");
}

int lm_hip_seq_from_ascii(lm_hip_ctx *ctx, char alphabet, const uint8_t *ascii, size_t len,
                          size_t cols, int lossy, lm_hip_seq **out, size_t *bad_index)
{
    if (!ctx || !out || (len && !ascii) || cols == 0)
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_from_ascii: bad argument");
    if (alphabet != 'D' && alphabet != 'P')
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_from_ascii: alphabet must be 'D' or 'P'");
    *out = nullptr;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    return ingest_tiled(ctx, ascii, len, cols, alphabet == 'P' ? 21 : 5, StripeTile::Ascii, alphabet == 'P', lossy != 0, out,
                        bad_index, "seq_from_ascii");
}

// DNA packed 4 bases per byte (base i in bits 2 * (i % 4) .. of byte i / 4, values A0 C1 T2 G3 = the reference's
// Nucleotide discriminants, abc.rs:115-135): a quarter of the bytes over PCIe.  N positions come as a list of runs
// {start, size} (what a .2bit file stores: a genome's N are few long runs) and / or as a bit mask (bit i % 8 of byte
// i / 8; half as many bytes again as the bases).  The packed text is uploaded whole (len / 4 bytes of scratch) and
// unpacked straight into the striped matrix by the stripe kernel -- no intermediate symbol array.
int lm_hip_seq_from_2bit(lm_hip_ctx *ctx, const uint8_t *packed, const uint8_t *n_mask, const uint64_t *n_runs,
                         size_t n_run_count, size_t len, size_t cols, lm_hip_seq **out)
{
    if (!ctx || !out || (len && !packed) || cols == 0 || (n_run_count && !n_runs))
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_from_2bit: bad argument");
    *out = nullptr;
    unsigned long long longest = 0;
    for (size_t r = 0; r < n_run_count; ++r) {
        const uint64_t start = n_runs[2 * r], size = n_runs[2 * r + 1];
        if (start > len || size > len - start)
            return fail(LM_HIP_ERR_BAD_ARGS, "seq_from_2bit: N run %zu (%llu + %llu) leaves the sequence of %zu bases", r,
                        (unsigned long long)start, (unsigned long long)size, len);
        longest = std::max<unsigned long long>(longest, size);
    }
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    const size_t rows = (len + cols - 1) / cols, stride = lm_hip_stride(cols, 1);
    const size_t pbytes = (len + 3) / 4, mbytes = n_mask ? (len + 7) / 8 : 0;
    const size_t moff = (pbytes + 16 + 15) / 16 * 16, roff = (moff + mbytes + 16 + 15) / 16 * 16;
    LM_TRY(ctx->scratch2.reserve(roff + n_run_count * 16 + 16));
    uint8_t *d_packed = static_cast<uint8_t *>(ctx->scratch2.ptr);
    uint8_t *d_mask = n_mask ? d_packed + moff : nullptr;
    unsigned long long *d_runs = reinterpret_cast<unsigned long long *>(d_packed + roff);
    lm_hip_seq *s = nullptr;
    LM_TRY(seq_alloc(ctx, rows, stride, cols, len, 5, &s));
    hipError_t e = hipSuccess;
    if (pbytes)
        e = hipMemcpyAsync(d_packed, packed, pbytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && mbytes)
        e = hipMemcpyAsync(d_mask, n_mask, mbytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n_run_count)
        e = hipMemcpyAsync(d_runs, n_runs, n_run_count * 16, hipMemcpyHostToDevice, ctx->stream);
    int st = LM_HIP_OK;
    if (e == hipSuccess && rows) {
        StripeTile tile;
        tile.d_src = d_packed;
        tile.len = len;
        tile.rows = tile.nrows = tile.pitch = rows;
        tile.cols = cols;
        tile.stride = stride;
        tile.def = 4;
        tile.d_data = s->d_data;
        tile.transform = StripeTile::TwoBit;
        tile.k = 5;
        tile.d_mask = d_mask;
        st = launch_stripe_tile(ctx, tile);
        if (st == LM_HIP_OK && n_run_count && longest)
            st = launch_n_runs(ctx, d_runs, n_run_count, longest, rows, stride, 4, s->d_data);
    }
    if (e == hipSuccess && st == LM_HIP_OK)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || st != LM_HIP_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        lm_hip_seq_destroy(s);
        return st != LM_HIP_OK ? st : fail(LM_HIP_ERR_HIP, "seq_from_2bit: upload failed: %s", hipGetErrorString(e));
    }
    *out = s;
    return LM_HIP_OK;
}

int lm_hip_seq_configure_wrap(lm_hip_ctx *ctx, lm_hip_seq *seq, size_t m)
{
    if (!ctx || !seq)
        return fail(LM_HIP_ERR_BAD_ARGS, "configure_wrap: null argument");
    if (m <= seq->wrap)  // seq.rs:370
        return LM_HIP_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    if (seq->rows + m > seq->capacity_rows && !seq->owns)
        return fail(LM_HIP_ERR_CAPACITY, "configure_wrap: the adopted matrix has room for %zu rows, %zu needed",
                    seq->capacity_rows, seq->rows + m);
    if (seq->rows + m > seq->capacity_rows) {
        const size_t cap = seq->rows + m + 32;
        uint8_t *nd = nullptr;
        LM_HIP_TRY(hipMalloc(&nd, cap * seq->stride));
        hipError_t e = hipMemcpyAsync(nd, seq->d_data, seq->rows * seq->stride, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(nd);
            return fail(LM_HIP_ERR_HIP, "sequence regrow failed: %s", hipGetErrorString(e));
        }
        (void)hipFree(seq->d_data);
        seq->d_data = nd;
        seq->capacity_rows = cap;
    }
    LM_TRY(launch_wrap(ctx, seq->d_data, seq->rows, seq->stride, seq->cols, m, (uint8_t)(seq->k - 1)));
    seq->wrap = m;
    return LM_HIP_OK;
}

int lm_hip_seq_info(const lm_hip_seq *seq, size_t *length, size_t *wrap, size_t *rows,
                    size_t *stride, size_t *cols, const uint8_t **d_data)
{
    if (!seq)
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_info: null sequence");
    if (length) *length = seq->length;
    if (wrap) *wrap = seq->wrap;
    if (rows) *rows = seq->rows;
    if (stride) *stride = seq->stride;
    if (cols) *cols = seq->cols;
    if (d_data) *d_data = seq->d_data;
    return LM_HIP_OK;
}

int lm_hip_seq_download(lm_hip_ctx *ctx, const lm_hip_seq *seq, uint8_t *dst)
{
    if (!ctx || !seq || !dst)
        return fail(LM_HIP_ERR_BAD_ARGS, "seq_download: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    const size_t bytes = (seq->rows + seq->wrap) * seq->stride;
    if (bytes) {
        LM_HIP_TRY(hipMemcpyAsync(dst, seq->d_data, bytes, hipMemcpyDeviceToHost, ctx->stream));
        LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return LM_HIP_OK;
}

int lm_hip_seq_destroy(lm_hip_seq *seq)
{
    if (!seq)
        return LM_HIP_OK;
    DeviceGuard guard(seq->device);
    if (seq->d_data && seq->owns)
        (void)hipFree(seq->d_data);
    delete seq;
    return LM_HIP_OK;
}

int lm_hip_scores_create(lm_hip_ctx *ctx, size_t cols, lm_hip_scores **out)
{
    if (!ctx || !out || cols == 0)
        return fail(LM_HIP_ERR_BAD_ARGS, "scores_create: bad argument");
    lm_hip_scores *s = new (std::nothrow) lm_hip_scores();
    if (!s)
        return fail(LM_HIP_ERR_OOM, "out of host memory");
    s->device = ctx->device;
    s->cols = cols;
    s->stride = lm_hip_stride(cols, sizeof(float));
    {
        DeviceGuard guard(ctx->device);
        if (hipMalloc(&s->d_best, sizeof(ArgmaxRecord)) != hipSuccess)
            s->d_best = nullptr;  // no cached argmax then; everything else works
        if (hipHostMalloc(reinterpret_cast<void **>(&s->h_best), 64, hipHostMallocDefault) != hipSuccess)
            s->h_best = nullptr;
        else
            memset(s->h_best, 0, 64);
    }
    *out = s;
    return LM_HIP_OK;
}

int lm_hip_scores_info(const lm_hip_scores *s, size_t *rows, size_t *stride, size_t *cols,
                       size_t *max_index, const float **d_data)
{
    if (!s)
        return fail(LM_HIP_ERR_BAD_ARGS, "scores_info: null scores");
    if (rows) *rows = s->rows;
    if (stride) *stride = s->stride;
    if (cols) *cols = s->cols;
    if (max_index) *max_index = s->max_index;
    if (d_data) *d_data = s->d_data;
    return LM_HIP_OK;
}

int lm_hip_scores_download(lm_hip_ctx *ctx, const lm_hip_scores *s, float *dst)
{
    if (!ctx || !s || (s->rows && !dst))
        return fail(LM_HIP_ERR_BAD_ARGS, "scores_download: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    if (s->rows) {
        LM_HIP_TRY(hipMemcpyAsync(dst, s->d_data, s->rows * s->stride * sizeof(float),
                                  hipMemcpyDeviceToHost, ctx->stream));
        LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return LM_HIP_OK;
}

int lm_hip_scores_download_rows(lm_hip_ctx *ctx, const lm_hip_scores *s, size_t row_begin,
                                size_t row_end, float *dst)
{
    if (!ctx || !s)
        return fail(LM_HIP_ERR_BAD_ARGS, "scores_download_rows: null argument");
    if (row_begin > row_end || row_end > s->rows)
        return fail(LM_HIP_ERR_BAD_ARGS, "scores_download_rows: rows %zu..%zu of %zu", row_begin, row_end,
                    s->rows);
    if (row_begin == row_end)
        return LM_HIP_OK;
    if (!dst)
        return fail(LM_HIP_ERR_BAD_ARGS, "scores_download_rows: null destination");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    LM_HIP_TRY(hipMemcpyAsync(dst, s->d_data + row_begin * s->stride,
                              (row_end - row_begin) * s->stride * sizeof(float), hipMemcpyDeviceToHost,
                              ctx->stream));
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return LM_HIP_OK;
}

int lm_hip_scores_destroy(lm_hip_scores *s)
{
    if (!s)
        return LM_HIP_OK;
    DeviceGuard guard(s->device);
    if (s->d_data)
        (void)hipFree(s->d_data);
    if (s->d_best)
        (void)hipFree(s->d_best);
    if (s->h_best)
        (void)hipHostFree(s->h_best);
    if (s->h_records)
        (void)hipHostFree(s->h_records);
    delete s;
    return LM_HIP_OK;
}

// scores.resize(rows, max_index) (scores.rs:148-152): grows the allocation when needed.
static int scores_resize(lm_hip_ctx *ctx, lm_hip_scores *s, size_t rows, size_t max_index)
{
    if (rows > s->capacity_rows) {
        LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (s->d_data)
            LM_HIP_TRY(hipFree(s->d_data));
        s->d_data = nullptr;
        s->capacity_rows = 0;
        LM_HIP_TRY(hipMalloc(&s->d_data, rows * s->stride * sizeof(float)));
        s->capacity_rows = rows;
        // alignment padding past `cols` is never written by the kernels and reads as
        // zero in the reference (DenseMatrix rows are default-initialised, dense.rs:144-147)
        if (s->stride != s->cols)
            LM_HIP_TRY(hipMemsetAsync(s->d_data, 0, rows * s->stride * sizeof(float), ctx->stream));
    }
    s->rows = rows;
    s->max_index = max_index;
    return LM_HIP_OK;
}

int lm_hip_score_rows_into(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq,
                           size_t row_begin, size_t row_end, lm_hip_scores *scores)
{
    if (!ctx || !pssm || !seq || !scores)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_rows_into: null argument");
    if (scores->cols != seq->cols)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_rows_into: scores have %zu columns, sequence %zu",
                    scores->cols, seq->cols);
    LM_TRY(check_score_args(pssm, seq->rows + seq->wrap, seq->stride, seq->cols, seq->wrap,
                            row_begin, row_end));
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    scores->best_valid = false;
    if (seq->length < pssm->m || row_begin >= row_end)  // pli/mod.rs:85-88
        return scores_resize(ctx, scores, 0, 0);
    LM_TRY(scores_resize(ctx, scores, row_end - row_begin, seq->length + 1 - pssm->m));
    ScoreArgs a{pssm, seq->d_data, seq->stride, seq->cols, row_begin, row_end, scores->d_data,
                scores->stride};
    // the reference's flow is score_into + argmax (lightmotif-bench dna.rs:104-107): the
    // store kernel tracks the best cell on the way, so that lm_hip_argmax on this handle is
    // a 16-byte read instead of a second pass over 4 B per position
    // (small inputs are launch-latency bound: the extra reduction launch costs more than the
    //  second pass it saves)
    return score_store_tracked(ctx, a, scores);
}

}  // extern "C"

int lm::score_store_tracked(lm_hip_ctx *ctx, const ScoreArgs &a, lm_hip_scores *scores)
{
    const size_t row_begin = a.row_begin, row_end = a.row_end;
    scores->best_valid = false;
    scores->best_on_host = false;
    scores->records_on_host = false;
    scores->folded = false;
    if (!scores->d_best || !ctx->track_argmax)
        return launch_score_store(ctx, a);
    bool tracked = false;
    if ((row_end - row_begin) * a.cols < (8u << 20)) {
        // small inputs are launch-latency bound: ONE launch stores, tracks the best cell and folds the
        // workgroup records (MODE_STORE_TRACK), and leaves the record in pinned memory as well
        const unsigned gen = ++scores->best_generation ? scores->best_generation : ++scores->best_generation;  // never 0
        LM_TRY(launch_score_store_track(ctx, a, scores->d_best, scores->h_best, gen, &tracked,
                                        scores->first_cell_rule ? 1 : 0, ctx->host_fold ? scores : nullptr));
        scores->best_valid = tracked;   // false when the records went to the host (scores->records_on_host)
        scores->best_on_host = tracked && scores->h_best != nullptr;
        return LM_HIP_OK;
    }
    LM_TRY(launch_score_store_argmax(ctx, a, scores->d_best, &tracked, scores->first_cell_rule ? 1 : 0));
    scores->best_valid = tracked;
    return LM_HIP_OK;
}

extern "C" {

int lm_hip_score_into(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq,
                      lm_hip_scores *scores)
{
    if (!seq)
        return fail(LM_HIP_ERR_BAD_ARGS, "score_into: null sequence");
    return lm_hip_score_rows_into(ctx, pssm, seq, 0, seq->rows, scores);  // pli/mod.rs:115-116
}

static int host_fold(lm_hip_ctx *ctx, lm_hip_scores *s)
{
    if (s->folded)
        return LM_HIP_OK;
    LM_TRY(fold_host_records(ctx, s->h_records, s->n_records, s->best_generation, s->first_cell_rule, &s->folded_record));
    s->folded = true;
    return LM_HIP_OK;
}

int lm_hip_argmax(lm_hip_ctx *ctx, const lm_hip_scores *s, int *found, lm_hip_coords *best,
                  float *value)
{
    if (!s)
        return fail(LM_HIP_ERR_BAD_ARGS, "argmax: null scores");
    if (ctx && found && s->records_on_host && s->rows) {  // small matrix: the store kernel's records, folded here
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard guard(ctx->device);
        LM_TRY(host_fold(ctx, const_cast<lm_hip_scores *>(s)));
        record_to_coords(s->folded_record, s->cols, found, best, value);
        return LM_HIP_OK;
    }
    if (ctx && found && s->best_valid && s->rows) {  // tracked by the kernel that wrote the scores
        std::lock_guard<std::mutex> lock(ctx->mu);
        DeviceGuard guard(ctx->device);
        if (s->best_on_host) {
            // the kernel that wrote the scores leaves the record in pinned memory and raises the generation word
            // behind it: poll that (a PCIe write after the fold) rather than wait for the completion signal of the
            // kernel -- ~10 us of every 22 us iteration of the reference's bench loop (dna.rs:104-107).  A kernel
            // that never gets there (a fault) is caught by the bounded spin: the stream is synchronised instead.
            const volatile unsigned *gen = reinterpret_cast<const volatile unsigned *>(s->h_best + 1);
            bool seen = false;
            for (unsigned spin = 0; spin < (1u << 20); ++spin) {  // tens of milliseconds at most
                if (__atomic_load_n(gen, __ATOMIC_ACQUIRE) == s->best_generation) {
                    seen = true;
                    break;
                }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
                __builtin_ia32_pause();
#endif
            }
            if (!seen)
                LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
            record_to_coords(*s->h_best, s->cols, found, best, value);
            return LM_HIP_OK;
        }
        LM_HIP_TRY(hipMemcpyAsync(ctx->pinned, s->d_best, sizeof(ArgmaxRecord), hipMemcpyDeviceToHost,
                                  ctx->stream));
        LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        record_to_coords(*static_cast<const ArgmaxRecord *>(ctx->pinned), s->cols, found, best, value);
        return LM_HIP_OK;
    }
    return lm_hip_argmax_shard_f32_dptr(ctx, s->d_data, s->rows, s->stride, s->cols, s->first_cell_rule ? 1 : 0,
                                        found, best, value);
}

// Maximum::max (pli/mod.rs:158-160): `self.argmax(scores).map(|c| scores.matrix()[c])` -- the value AT
// the Generic argmax (so a NaN first cell gives NaN, an all -inf matrix -inf), None when empty.
int lm_hip_max(lm_hip_ctx *ctx, const lm_hip_scores *s, int *found, float *value)
{
    lm_hip_coords c{0, 0};
    return lm_hip_argmax(ctx, s, found, &c, value);
}

int lm_hip_max_f32_dptr(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride, size_t cols,
                        int *found, float *value)
{
    lm_hip_coords c{0, 0};
    return lm_hip_argmax_f32_dptr(ctx, d_scores, rows, stride, cols, found, &c, value);
}

int lm_hip_scores_set_first_cell_rule(lm_hip_scores *s, int enabled)
{
    if (!s)
        return fail(LM_HIP_ERR_BAD_ARGS, "scores_set_first_cell_rule: null scores");
    if (s->first_cell_rule != (enabled != 0)) {
        s->best_valid = false;  // the device record was reduced under the other rule
        s->best_on_host = false;
        s->folded = false;      // (host-side records are folded again, under the new rule)
    }
    s->first_cell_rule = enabled != 0;
    return LM_HIP_OK;
}

int lm_hip_threshold(lm_hip_ctx *ctx, const lm_hip_scores *s, float t, lm_hip_coords **coords,
                     size_t *n)
{
    if (!s)
        return fail(LM_HIP_ERR_BAD_ARGS, "threshold: null scores");
    return lm_hip_threshold_f32_dptr(ctx, s->d_data, s->rows, s->stride, s->cols, t, coords, n);
}

}  // extern "C"
