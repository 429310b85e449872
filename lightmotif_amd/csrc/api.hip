// api.hip -- the extern "C" surface declared in include/lightmotif_hip.h.
//
// Pre-checks and result shapes follow the reference wrappers:
//   Avx2::score_f32_rows_into_permute (lightmotif/src/pli/platform/avx2.rs:817-851):
//     wrap check (:832-837), degenerate check (:839-842), scores.resize (:844)
//   Score::score_into (pli/mod.rs:109-117), StripedScores::{argmax,threshold}
//     (scores.rs:181-213).
#include <algorithm>
#include <sys/mman.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "score_prefilter2.hpp"

namespace lm {

static thread_local char g_err[512] = "";

int fail(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return status;
}

int Scratch::reserve(size_t n)
{
    if (n <= bytes)
        return LM_HIP_OK;
    if (ptr) {
        LM_HIP_TRY(hipFree(ptr));
        ptr = nullptr;
        bytes = 0;
    }
    const size_t want = (n + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
    LM_HIP_TRY(hipMalloc(&ptr, want));
    bytes = want;
    return LM_HIP_OK;
}

void Scratch::release()
{
    if (ptr)
        (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

int check_score_args(const lm_hip_pssm *pssm, size_t seq_rows_total, size_t seq_stride, size_t cols, size_t wrap,
                     size_t row_begin, size_t row_end)
{
    if (!pssm)
        return fail(LM_HIP_ERR_BAD_ARGS, "score: null pssm");
    if (cols == 0 || seq_stride < cols)
        return fail(LM_HIP_ERR_BAD_ARGS, "score: stride %zu < columns %zu", seq_stride, cols);
    if (wrap > seq_rows_total)
        return fail(LM_HIP_ERR_BAD_ARGS, "score: wrap %zu > matrix rows %zu", wrap, seq_rows_total);
    // avx2.rs:832-837
    if (pssm->m > 0 && wrap < pssm->m - 1)
        return fail(LM_HIP_ERR_WRAP, "not enough wrapping rows for motif of length %zu", pssm->m);
    if (row_begin < row_end && row_end > seq_rows_total - wrap)
        return fail(LM_HIP_ERR_BAD_ARGS, "score: row range %zu..%zu exceeds the %zu sequence rows",
                    row_begin, row_end, seq_rows_total - wrap);
    return LM_HIP_OK;
}

// Builds the LDS image of score_c32_prefilter<M>: [u16 layout EVEN | u16 layout ODD]
// and the affine map discrete ~ (score - offset) / factor.  Follows the
// idea of DiscreteMatrix (pwm/mod.rs:665-696: per-row offsets, one global factor,
// weights rounded UP) on 16 bits.  Returns false when no sound prefilter exists.
static bool build_prefilter(lm_hip_pssm &p, std::vector<unsigned> *image, std::vector<unsigned> *image2)
{
    const int m = (int)p.m, k = (int)p.k;
    if (m < 1)
        return false;
    const int mp = prefilter_mp(m), shift = mp - m;
    std::vector<double> off(m), top(m);
    double offset = 0, range = 0, abs_sum = 0;
    for (int j = 0; j < m; ++j) {
        double lo = INFINITY, hi = -INFINITY, amax = 0;
        for (int s = 0; s < k; ++s) {
            const float x = p.host[(size_t)j * k + s];
            if (x != x || x == INFINITY)
                return false;             // NaN / +inf: score semantics the bound cannot cover
            if (x == -INFINITY)
                continue;                 // stands for the row minimum (over-estimate)
            lo = std::min(lo, (double)x);
            hi = std::max(hi, (double)x);
            amax = std::max(amax, std::fabs((double)x));
        }
        if (lo == INFINITY)
            return false;                 // a row of -inf only: every score is -inf
        off[j] = lo;
        top[j] = hi;
        offset += lo;
        range += hi - lo;
        abs_sum += amax;
    }
    if (!(range > 0))
        return false;
    const double factor = range / 65000.0;
    // discrete weights d'[0..mp): leading zero row when m is odd
    std::vector<unsigned> d((size_t)mp * k, 0);
    for (int j = 0; j < m; ++j)
        for (int s = 0; s < k; ++s) {
            const float x = p.host[(size_t)j * k + s];
            const double v = (x == -INFINITY) ? 0.0 : ((double)x - off[j]) / factor;
            unsigned q = (unsigned)std::ceil(v);
            if ((double)q < v + 1e-9)     // guard the ceil against representation error
                q += 1;
            d[(size_t)(j + shift) * k + s] = q;
        }
    image->assign((size_t)prefilter_image_dw(m, k), 0u);
    prefilter_pack_image(d.data(), m, k, image->data());
    // pair-symbol table of score_c32_prefilter2<M> (DNA only): row (a, b) holds
    // E[e] = d[e-1][a] + d[e][b] over the motif padded to an ODD length M' by a leading
    // zero row; dword m = (lo E[2m+1], hi E[2m]).  Same weights, same sums, same bound.
    image2->clear();
    if (k == 5 || k == 21) {  // DNA: 25 pair rows; protein: 441
        image2->assign((size_t)prefilter2_image_dw(m, k), 0u);
        prefilter2_pack_image(d.data() + (size_t)shift * k, m, image2->data(), k);
    }
    p.pre_offset = offset;
    p.pre_factor = factor;
    // |f32 sum - real sum| <= (M-1) * 2^-24 * sum |terms|  (each add rounds to nearest)
    p.pre_emax = (double)m * std::ldexp(1.0, -24) * abs_sum * 1.5;
    return true;
}

// ---- result blocks ------------------------------------------------------------------------------
// Host arrays handed to the caller (threshold / hit lists) and released with lm_hip_free.  A dense
// hit list is tens of megabytes per call: a fresh malloc'ed block costs ~5 000 page faults while the
// read-back lands in it and the copy engine has to stage pageable memory through bounce buffers --
// at a p = 1e-3 hit rate (1e6 hits per Gbp, 20 MB) that was more than half of the call
// (profiles/r01_timeline_fused_p1e-3.txt: 1.06 ms of device work in a 2.14 ms call).  Large blocks
// therefore come from a small process-wide pool: 2 MB-aligned, page-locked (hipHostRegister, so the
// read-back is one DMA at link rate) and REUSED when the caller frees them.  LM_HIP_RESULT_POOL_MB
// bounds what the pool keeps when idle (default 256; 0 or negative = no pooling, plain malloc); blocks the
// caller still holds count too: once pinned memory -- idle and handed out -- reaches four times that budget,
// further results are plain (pageable, unpooled) allocations, so a caller that keeps many results alive
// cannot lock an unbounded amount of host memory.
namespace {
struct ResultBlock {
    void *ptr;
    size_t cap;
    bool pinned, in_use;
    unsigned long long stamp;
};
std::mutex g_pool_mu;
std::vector<ResultBlock> g_pool;
unsigned long long g_pool_stamp = 0;
constexpr size_t kPoolMin = 1u << 20;  // smaller results: malloc

size_t pool_budget()
{
    static const size_t b = [] {
        const char *e = getenv("LM_HIP_RESULT_POOL_MB");
        const long long mb = e ? atoll(e) : 256;
        return (size_t)(mb < 0 ? 0 : mb > (1ll << 20) ? (1ll << 20) : mb) << 20;
    }();
    return b;
}

void pool_trim_locked()
{
    for (;;) {
        size_t idle = 0;
        int oldest = -1;
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (!g_pool[i].in_use) {
                idle += g_pool[i].cap;
                if (oldest < 0 || g_pool[i].stamp < g_pool[(size_t)oldest].stamp)
                    oldest = (int)i;
            }
        if (oldest < 0 || idle <= pool_budget())
            return;
        if (g_pool[(size_t)oldest].pinned)
            (void)hipHostUnregister(g_pool[(size_t)oldest].ptr);
        free(g_pool[(size_t)oldest].ptr);
        g_pool.erase(g_pool.begin() + oldest);
    }
}
}  // namespace

void *result_alloc(size_t bytes)
{
    constexpr size_t kHuge = 2u << 20;
    if (bytes < kPoolMin || pool_budget() == 0)
        return malloc(bytes);
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        int pick = -1;
        for (size_t i = 0; i < g_pool.size(); ++i)  // best fit, at most 4x too large
            if (!g_pool[i].in_use && g_pool[i].cap >= bytes && g_pool[i].cap / 4 <= bytes &&
                (pick < 0 || g_pool[i].cap < g_pool[(size_t)pick].cap))
                pick = (int)i;
        if (pick >= 0) {
            g_pool[(size_t)pick].in_use = true;
            return g_pool[(size_t)pick].ptr;
        }
    }
    const size_t cap = (bytes + bytes / 4 + kHuge - 1) / kHuge * kHuge;  // some headroom: counts vary call to call
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        size_t pinned_total = 0;
        for (const ResultBlock &b : g_pool)
            pinned_total += b.cap;
        if (pinned_total + cap > 4 * pool_budget())
            return malloc(bytes);  // the cap on page-locked memory is reached: a plain block, freed by free()
    }
    void *p = nullptr;
    if (posix_memalign(&p, kHuge, cap) != 0)
        return malloc(bytes);
    (void)madvise(p, cap, MADV_HUGEPAGE);  // advisory: plain pages if unavailable
    const bool pinned = hipHostRegister(p, cap, hipHostRegisterPortable) == hipSuccess;
    if (!pinned)
        (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(g_pool_mu);
    g_pool.push_back(ResultBlock{p, cap, pinned, true, 0});
    return p;
}

void result_free(void *p)
{
    if (!p)
        return;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        for (auto &b : g_pool)
            if (b.ptr == p) {
                b.in_use = false;
                b.stamp = ++g_pool_stamp;
                pool_trim_locked();
                return;
            }
    }
    free(p);
}

}  // namespace lm

using namespace lm;

extern "C" {

// ---- library ----------------------------------------------------------------------------

int lm_hip_abi_version(void) { return LM_HIP_ABI_VERSION; }

const char *lm_hip_last_error(void) { return g_err; }

int lm_hip_device_count(int *count)
{
    if (!count)
        return fail(LM_HIP_ERR_BAD_ARGS, "device_count: null output");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        n = 0;
    int usable = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0)
            ++usable;
    }
    *count = usable;
    return LM_HIP_OK;
}

int lm_hip_device_ordinal(int index, int *ordinal)
{
    if (!ordinal || index < 0)
        return fail(LM_HIP_ERR_BAD_ARGS, "device_ordinal: bad argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        n = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0 &&
            index-- == 0) {
            *ordinal = d;
            return LM_HIP_OK;
        }
    }
    return fail(LM_HIP_ERR_NO_DEVICE, "fewer usable (gfx950) devices than index + 1");
}

void lm_hip_free(void *p) { result_free(p); }

size_t lm_hip_stride(size_t cols, size_t elem_size)
{
    // dense.rs:43-48 (Row is repr(align(32)) on x86-64) + dense.rs:126-128
    if (elem_size == 0)
        return 0;
    const size_t bytes = (cols * elem_size + 31) / 32 * 32;
    return bytes / elem_size;
}

// ---- context ----------------------------------------------------------------------------

// Development / test switches, read once per context: each selects an alternative path that gives the SAME results
// (the GPU suite runs several of them to cover those paths; tools/README.md lists them).  None is part of the ABI.
static void read_dev_switches(lm_hip_ctx *ctx)
{
    if (const char *e = getenv("LM_HIP_TRACK_ARGMAX"))  // A/B switch: 0 = plain store in score_into
        ctx->track_argmax = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_XLONG"))  // A/B switch: 0 = motifs beyond 64 rows stored in slices (round 2 / 3 form)
        ctx->xlong_store = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_HOST_FOLD"))  // A/B switch: 0 = small score_into folds its records on the device
        ctx->host_fold = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_SPECULATE_ORDER"))  // A/B switch: 0 = read the counts first
        ctx->speculate_order = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_SUFFIX_ARGMAX"))  // A/B switch: 0 = always scan the whole range
        ctx->suffix_argmax = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_MULTI_MOTIF"))  // A/B switch: 0 = one motif per workgroup pass
        ctx->multi_motif = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_QUAD_LOADS"))  // A/B switch of the store kernel's symbol loads
        ctx->quad_loads = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_SKIP_UNREACHABLE"))  // A/B switch: 0 = scan even when no cell can reach the threshold
        ctx->skip_unreachable = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_PAIR_PREFILTER"))  // A/B switch: 0 = one symbol per lookup
        ctx->pair_prefilter = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_PAIR_PREFILTER_PROTEIN"))  // A/B switch: 1 = 441-row pair scan for K = 21
        ctx->pair_prefilter_protein = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_CHUNKED_FUSED"))  // A/B switch: 0 = fused scans of M > 36 go cell by cell
        ctx->chunked_fused = atoi(e) != 0;
    if (const char *e = getenv("LM_HIP_CHUNK_ROWS"))  // rows per chunk of those scans
        if (atoll(e) >= 64)
            ctx->chunk_rows = (size_t)atoll(e);
}

static int ctx_create(int device, void *stream, bool borrow, lm_hip_ctx **out)
{
    if (!out)
        return fail(LM_HIP_ERR_BAD_ARGS, "ctx_create: null output");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(LM_HIP_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= n)
        return fail(LM_HIP_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, n);
    hipDeviceProp_t prop;
    LM_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LM_HIP_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                    device, prop.gcnArchName);
    DeviceGuard guard(device);
    if (!guard.ok)
        return fail(LM_HIP_ERR_HIP, "hipSetDevice(%d) failed", device);
    lm_hip_ctx *ctx = new (std::nothrow) lm_hip_ctx();
    if (!ctx)
        return fail(LM_HIP_ERR_OOM, "out of host memory");
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    read_dev_switches(ctx);
    if (borrow) {
        ctx->stream = static_cast<hipStream_t>(stream);
        ctx->owns_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return fail(LM_HIP_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        ctx->owns_stream = true;
    }
    hipError_t e = hipHostMalloc(&ctx->pinned, kPinnedBytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        if (ctx->owns_stream)
            (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return fail(LM_HIP_ERR_OOM, "hipHostMalloc failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return LM_HIP_OK;
}

int lm_hip_ctx_create(int device, lm_hip_ctx **out) { return ctx_create(device, nullptr, false, out); }

int lm_hip_ctx_create_on_stream(int device, void *hip_stream, lm_hip_ctx **out)
{
    return ctx_create(device, hip_stream, true, out);
}

int lm_hip_ctx_destroy(lm_hip_ctx *ctx)
{
    if (!ctx)
        return LM_HIP_OK;
    DeviceGuard guard(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->scratch.release();
    ctx->scratch2.release();
    ctx->chunk_scores.release();
    ctx->scan_buf.release();
    ctx->u8_tables.release();
    if (ctx->pinned)
        (void)hipHostFree(ctx->pinned);
    if (ctx->d_ticket)
        (void)hipFree(ctx->d_ticket);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
        for (int b = 0; b < 2; ++b) {
            (void)hipEventDestroy(ctx->tile_copied[b]);
            (void)hipEventDestroy(ctx->tile_consumed[b]);
        }
    }
    if (ctx->aux_stream) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
        (void)hipEventDestroy(ctx->fork_event);
        (void)hipEventDestroy(ctx->join_event);
    }
    if (ctx->owns_stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return LM_HIP_OK;
}

int lm_hip_ctx_sync(lm_hip_ctx *ctx)
{
    if (!ctx)
        return fail(LM_HIP_ERR_BAD_ARGS, "null context");
    DeviceGuard guard(ctx->device);
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return LM_HIP_OK;
}

int lm_hip_ctx_stream(lm_hip_ctx *ctx, void **hip_stream)
{
    if (!ctx || !hip_stream)
        return fail(LM_HIP_ERR_BAD_ARGS, "null argument");
    *hip_stream = ctx->stream;
    return LM_HIP_OK;
}

int lm_hip_ctx_set_rows_per_stream(lm_hip_ctx *ctx, size_t rows)
{
    if (!ctx)
        return fail(LM_HIP_ERR_BAD_ARGS, "null context");
    ctx->rows_per_stream = rows;
    return LM_HIP_OK;
}

int lm_hip_ctx_set_xcd_remap(lm_hip_ctx *ctx, int enabled)
{
    if (!ctx)
        return fail(LM_HIP_ERR_BAD_ARGS, "null context");
    ctx->xcd_remap = enabled != 0;
    return LM_HIP_OK;
}

int lm_hip_ctx_set_prefilter(lm_hip_ctx *ctx, int enabled)
{
    if (!ctx)
        return fail(LM_HIP_ERR_BAD_ARGS, "null context");
    ctx->use_prefilter = enabled != 0;
    return LM_HIP_OK;
}

int lm_hip_ctx_set_track_argmax(lm_hip_ctx *ctx, int enabled)
{
    if (!ctx)
        return fail(LM_HIP_ERR_BAD_ARGS, "null context");
    ctx->track_argmax = enabled != 0;
    return LM_HIP_OK;
}

const char *lm_hip_ctx_last_kernel(lm_hip_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

// ---- PSSM ---------------------------------------------------------------------------------

int lm_hip_pssm_create(lm_hip_ctx *ctx, const float *pssm, size_t m, size_t stride, size_t k,
                       lm_hip_pssm **out)
{
    if (!ctx || !out || (!pssm && m))
        return fail(LM_HIP_ERR_BAD_ARGS, "pssm_create: null argument");
    *out = nullptr;
    if (k == 0 || k > 256 || stride < k)
        return fail(LM_HIP_ERR_BAD_ARGS, "pssm_create: bad alphabet size %zu / stride %zu", k, stride);
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    lm_hip_pssm *p = new (std::nothrow) lm_hip_pssm();
    if (!p)
        return fail(LM_HIP_ERR_OOM, "out of host memory");
    p->device = ctx->device;
    p->m = m;
    p->k = k;
    p->host.resize(m * k);
    for (size_t j = 0; j < m; ++j)
        for (size_t s = 0; s < k; ++s)
            p->host[j * k + s] = pssm[j * stride + s];
    auto cleanup = [&](int st) {
        lm_hip_pssm_destroy(p);
        return st;
    };
    if (m) {
        hipError_t e = hipMalloc(&p->d_dense, m * k * sizeof(float));
        if (e != hipSuccess)
            return cleanup(fail(LM_HIP_ERR_OOM, "hipMalloc(pssm) failed: %s", hipGetErrorString(e)));
        e = hipMemcpyAsync(p->d_dense, p->host.data(), m * k * sizeof(float), hipMemcpyHostToDevice,
                           ctx->stream);
        if (e != hipSuccess)
            return cleanup(fail(LM_HIP_ERR_HIP, "pssm upload failed: %s", hipGetErrorString(e)));
        if (m <= (size_t)kMaxFastM) {
            // transposed, padded table of score_c32<M>: table[s * ts + j] = pssm[j][s]
            // (K > 16: rows of 2 * odd dwords for the 8-byte reads of the WIDE kernels, see table_stride)
            p->ts = (size_t)table_stride((int)m, lds_wide((int)k));
            std::vector<float> table(k * p->ts, 0.0f);
            for (size_t s = 0; s < k; ++s)
                for (size_t j = 0; j < m; ++j)
                    table[s * p->ts + j] = p->host[j * k + s];
            e = hipMalloc(&p->d_table, table.size() * sizeof(float));
            if (e != hipSuccess)
                return cleanup(fail(LM_HIP_ERR_OOM, "hipMalloc(table) failed: %s", hipGetErrorString(e)));
            e = hipMemcpyAsync(p->d_table, table.data(), table.size() * sizeof(float),
                               hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(ctx->stream);  // `table` dies with this scope
            if (e != hipSuccess)
                return cleanup(fail(LM_HIP_ERR_HIP, "table upload failed: %s", hipGetErrorString(e)));
            // lengths that are no multiple of 4: a second table with leading all-zero rows (see
            // lm_hip_pssm::d_table_pad); 33..35 stay as they are (36 rows cost more than the byte loads)
            if (m % 4 != 0 && (m + 3) / 4 * 4 <= 32) {
                const size_t mp = (m + 3) / 4 * 4, lead = mp - m, tsp = (size_t)table_stride((int)mp, lds_wide((int)k));
                std::vector<float> padded(k * tsp, 0.0f);
                for (size_t s = 0; s < k; ++s)
                    for (size_t j = 0; j < m; ++j)
                        padded[s * tsp + lead + j] = p->host[j * k + s];
                e = hipMalloc(&p->d_table_pad, padded.size() * sizeof(float));
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_OOM, "hipMalloc(table) failed: %s", hipGetErrorString(e)));
                e = hipMemcpyAsync(p->d_table_pad, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice,
                                   ctx->stream);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(ctx->stream);  // `padded` dies with this scope
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_HIP, "table upload failed: %s", hipGetErrorString(e)));
                p->lead = lead;
            }
            // discrete prefilter image (score_prefilter.hpp); absent when the matrix has
            // NaN / +inf entries or no spread -- the exact f32 fused kernel is used then
            std::vector<unsigned> image;
            std::vector<unsigned> image2;
            if (build_prefilter(*p, &image, &image2)) {
                e = hipMalloc(&p->d_image, image.size() * sizeof(unsigned));
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_OOM, "hipMalloc(prefilter) failed: %s", hipGetErrorString(e)));
                e = hipMemcpyAsync(p->d_image, image.data(), image.size() * sizeof(unsigned),
                                   hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_HIP, "prefilter upload failed: %s", hipGetErrorString(e)));
                if (!image2.empty()) {  // DNA: pair-symbol table (score_prefilter2.hpp)
                    e = hipMalloc(&p->d_image2, image2.size() * sizeof(unsigned));
                    if (e == hipSuccess)
                        e = hipMemcpyAsync(p->d_image2, image2.data(), image2.size() * sizeof(unsigned),
                                           hipMemcpyHostToDevice, ctx->stream);
                    if (e == hipSuccess)
                        e = hipStreamSynchronize(ctx->stream);
                    if (e != hipSuccess)
                        return cleanup(fail(LM_HIP_ERR_HIP, "pair prefilter upload failed: %s",
                                            hipGetErrorString(e)));
                }
                p->has_prefilter = true;
            }
        }
        if (m > (size_t)kMaxFastM && k <= 64) {
            // long motifs: slices of <= kMaxLongM rows (multiples of 4 rows: dword symbol loads).  Up to
            // kMaxLongM that is ONE slice -- a single pass of the long kernel family (score_long_inst.hip);
            // beyond, the first slice is stored and the others continue in place (MODE_CONTINUE)
            // (65 ... kMaxStoreM rows: ONE slice as well, padded to a multiple of 8 -- the store-only kernels of
            //  score_xlong_inst.hip; LM_HIP_XLONG=0 keeps the slices for A/B runs)
            const bool xlong = m > (size_t)kMaxLongM && m <= (size_t)kMaxStoreM && ctx->xlong_store;
            const size_t nparts = xlong ? 1 : (m + kMaxLongM - 1) / kMaxLongM;
            const size_t len = xlong ? m : std::min<size_t>(((m + nparts - 1) / nparts + 3) / 4 * 4, (size_t)kMaxLongM);
            for (size_t off = 0; off < m; off += len) {
                lm_hip_pssm::Part part;
                part.off = off;
                const size_t real = std::min(len, m - off);
                const size_t unit = xlong ? 8 : 4;
                part.lead = (unit - real % unit) % unit;  // the last slice: leading zero rows up to a multiple of 4 (8)
                part.m = real + part.lead;
                part.ts = (size_t)table_stride((int)part.m, lds_wide((int)k));
                std::vector<float> table(k * part.ts, 0.0f);
                for (size_t s = 0; s < k; ++s)
                    for (size_t j = 0; j < real; ++j)
                        table[s * part.ts + part.lead + j] = p->host[(off + j) * k + s];
                e = hipMalloc(&part.d_table, table.size() * sizeof(float));
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_OOM, "hipMalloc(table) failed: %s", hipGetErrorString(e)));
                p->parts.push_back(part);  // owned from here on (freed by lm_hip_pssm_destroy)
                e = hipMemcpyAsync(part.d_table, table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice,
                                   ctx->stream);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(ctx->stream);  // `table` dies with this scope
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_HIP, "table upload failed: %s", hipGetErrorString(e)));
            }
        }
        if (m > (size_t)kMaxFastM && m <= (size_t)kMaxPairM && k == 5) {
            // 36 < M <= 128, DNA: the pair-symbol prefilter table, so that the fused threshold / argmax scans of these
            // lengths flag candidates like the shorter ones do (the one-symbol u16 scan ends at kMaxFastM)
            std::vector<unsigned> image, image2;
            if (build_prefilter(*p, &image, &image2) && !image2.empty()) {
                e = hipMalloc(&p->d_image2, image2.size() * sizeof(unsigned));
                if (e == hipSuccess)
                    e = hipMemcpyAsync(p->d_image2, image2.data(), image2.size() * sizeof(unsigned), hipMemcpyHostToDevice,
                                       ctx->stream);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_HIP, "pair prefilter upload failed: %s", hipGetErrorString(e)));
                p->has_prefilter = true;
            }
        }
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess)
            return cleanup(fail(LM_HIP_ERR_HIP, "pssm upload failed: %s", hipGetErrorString(e)));
    }
    *out = p;
    return LM_HIP_OK;
}

int lm_hip_pssm_reverse_complement(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, lm_hip_pssm **out)
{
    if (!ctx || !pssm || !out)
        return fail(LM_HIP_ERR_BAD_ARGS, "pssm_reverse_complement: null argument");
    *out = nullptr;
    if (pssm->k != 5)
        return fail(LM_HIP_ERR_BAD_ARGS, "pssm_reverse_complement: only DNA matrices (K = 5) have a complement");
    static const int comp[5] = {2, 3, 0, 1, 4};  // A C T G N -> T G A C N
    const size_t m = pssm->m, k = pssm->k;
    std::vector<float> rc(m * k);
    for (size_t i = 0; i < m; ++i)  // pwm/mod.rs:570-574
        for (size_t s = 0; s < k; ++s)
            rc[i * k + s] = pssm->host[(m - 1 - i) * k + comp[s]];
    return lm_hip_pssm_create(ctx, rc.data(), m, k, k, out);
}

int lm_hip_pssm_destroy(lm_hip_pssm *p)
{
    if (!p)
        return LM_HIP_OK;
    DeviceGuard guard(p->device);
    if (p->d_dense)
        (void)hipFree(p->d_dense);
    if (p->d_table)
        (void)hipFree(p->d_table);
    if (p->d_table_pad)
        (void)hipFree(p->d_table_pad);
    for (auto &part : p->parts)
        if (part.d_table)
            (void)hipFree(part.d_table);
    if (p->d_image)
        (void)hipFree(p->d_image);
    if (p->d_image2)
        (void)hipFree(p->d_image2);
    delete p;
    return LM_HIP_OK;
}

size_t lm_hip_pssm_len(const lm_hip_pssm *p) { return p ? p->m : 0; }

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
    return launch_threshold_u8(ctx, d_scores, rows, stride, cols, t, coords, n);
}

static void record_to_coords(const ArgmaxRecord &rec, size_t cols, int *found, lm_hip_coords *best,
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
    hipStream_t st = nullptr;
    LM_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
        LM_HIP_TRY(hipEventCreateWithFlags(&ctx->tile_copied[b], hipEventDisableTiming));
        LM_HIP_TRY(hipEventCreateWithFlags(&ctx->tile_consumed[b], hipEventDisableTiming));
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
    scores->best_on_host = false;
    scores->records_on_host = false;
    scores->folded = false;
    if (!scores->d_best || !ctx->track_argmax)
        return launch_score_store(ctx, a);
    bool tracked = false;
    if ((row_end - row_begin) * seq->cols < (8u << 20)) {
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
