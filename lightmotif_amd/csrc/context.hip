// context.hip -- the library's plumbing behind include/lightmotif_hip.h: error text, device scratch, the page-locked pool of
// result blocks, library / device queries and the context (stream, options).  The other parts of the C ABI:
//   pssm.hip       lm_hip_pssm_*: device tables of a scoring matrix (transposed f32 table, prefilter images)
//   score_api.hip  Score / Maximum / Threshold on device pointers, the fused and batched scans
//   handles.hip    resident StripedSequence / StripedScores handles, host -> device ingest
//   hostptr.hip    the host-pointer entry points a reference-side shim binds
//   comm.hip       row-sharded jobs over RCCL
#include <algorithm>
#include <sys/mman.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <new>
#include <string>

#include "score_prefilter2.hpp"

#include "lm_internal.hpp"

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
size_t g_pool_reserved = 0;  // bytes of blocks being page-locked right now (counted against the cap)
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
        if (pinned_total + g_pool_reserved + cap > 4 * pool_budget())
            return malloc(bytes);  // the cap on page-locked memory is reached: a plain block, freed by free()
        g_pool_reserved += cap;    // check and reservation under ONE lock: concurrent callers cannot exceed the cap together
    }
    void *p = nullptr;
    if (posix_memalign(&p, kHuge, cap) != 0) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        g_pool_reserved -= cap;
        return malloc(bytes);
    }
    (void)madvise(p, cap, MADV_HUGEPAGE);  // advisory: plain pages if unavailable
    const bool pinned = hipHostRegister(p, cap, hipHostRegisterPortable) == hipSuccess;
    if (!pinned)
        (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(g_pool_mu);
    g_pool_reserved -= cap;
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

int lm_hip_result_pool_info(size_t *pinned_idle, size_t *pinned_in_use, size_t *budget)
{
    if (!pinned_idle && !pinned_in_use && !budget)
        return fail(LM_HIP_ERR_BAD_ARGS, "result_pool_info: no output asked for");
    size_t idle = 0, used = 0;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        for (const ResultBlock &b : g_pool)
            (b.in_use ? used : idle) += b.cap;
    }
    if (pinned_idle) *pinned_idle = idle;
    if (pinned_in_use) *pinned_in_use = used;
    if (budget) *budget = pool_budget();
    return LM_HIP_OK;
}

size_t lm_hip_stride(size_t cols, size_t elem_size)
{
    // dense.rs:43-48 (Row is repr(align(32)) on x86-64) + dense.rs:126-128
    if (elem_size == 0)
        return 0;
    const size_t bytes = (cols * elem_size + 31) / 32 * 32;
    return bytes / elem_size;
}

// ---- shader clock ------------------------------------------------------------------------

// One wavefront that mostly sleeps: the shader clock counter (s_memtime ticks at the clock the SIMDs run at,
// MI355X_MICROARCH.md "s_memtime tick = shader cycle") against the constant-rate counter (s_memrealtime) over a window.
// Launched on a stream of its own beside the kernels under measurement it reports the clock THEY run at: the part
// clocks to its power budget, so the 2.4 GHz of the data sheet is not what an LDS- or VALU-bound kernel sustains.
namespace lm {
__global__ void clock_probe_kernel(unsigned long long window_ticks, unsigned long long *out)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < window_ticks) {
        __builtin_amdgcn_s_sleep(127);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}
}  // namespace lm

int lm_hip_device_clock_mhz(int device, unsigned window_us, double *mhz)
{
    if (!mhz)
        return fail(LM_HIP_ERR_BAD_ARGS, "device_clock_mhz: null output");
    *mhz = 0.0;
    if (window_us == 0 || window_us > 2000000u)
        return fail(LM_HIP_ERR_BAD_ARGS, "device_clock_mhz: window of %u us (1 ... 2 000 000)", window_us);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n)
        return fail(LM_HIP_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, n);
    DeviceGuard guard(device);
    int wall_khz = 0;
    LM_HIP_TRY(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, device));
    if (wall_khz <= 0)
        return fail(LM_HIP_ERR_HIP, "device_clock_mhz: the device reports no constant-rate counter");
    // One probe stream and one pinned record per device, made once and kept: creating and freeing them per window
    // (hipHostFree and hipStreamDestroy synchronise the device) stalled every other thread's synchronising calls --
    // fused scans took 1.4 ms instead of 0.29 beside the probe (profiles/r05_clock_probe.json).  Default priority: on a
    // high-priority queue the probe's long-lived wavefront held back the dispatch of the kernels under measurement.
    struct Probe {
        hipStream_t stream = nullptr;
        unsigned long long *rec = nullptr;
    };
    static std::mutex probe_mu;
    static Probe probes[64];
    std::lock_guard<std::mutex> probe_lock(probe_mu);  // one window at a time per process
    if (device >= 64)
        return fail(LM_HIP_ERR_BAD_ARGS, "device_clock_mhz: device ordinal %d", device);
    Probe &pr = probes[device];
    hipError_t e = hipSuccess;
    if (!pr.stream)
        e = hipStreamCreateWithFlags(&pr.stream, hipStreamNonBlocking);
    if (e == hipSuccess && !pr.rec)
        e = hipHostMalloc(reinterpret_cast<void **>(&pr.rec), 2 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) {
        pr.rec[0] = pr.rec[1] = 0;
        hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, pr.stream,
                           (unsigned long long)window_us * (unsigned long long)wall_khz / 1000ull, pr.rec);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(pr.stream);
    const unsigned long long ticks = pr.rec ? pr.rec[0] : 0, ref = pr.rec ? pr.rec[1] : 0;
    if (e != hipSuccess)
        return fail(LM_HIP_ERR_HIP, "device_clock_mhz: %s", hipGetErrorString(e));
    if (ref == 0)
        return fail(LM_HIP_ERR_HIP, "device_clock_mhz: the constant-rate counter did not advance");
    *mhz = (double)ticks / (double)ref * (double)wall_khz / 1000.0;
    return LM_HIP_OK;
}

// ---- context ----------------------------------------------------------------------------

// Options of a context: each selects an alternative path that gives the SAME results (the GPU suite sets several of them
// to cover those paths).  Set through lm_hip_ctx_set_option; a -DLM_HIP_DEV_SWITCHES build (tools/build_variant.py)
// also reads LM_HIP_<NAME IN CAPITALS> from the environment when a context is created, for A/B runs of whole commands.
static const char *const kOptionNames[] = {"track_argmax", "xlong_store", "host_fold", "speculate_order", "suffix_argmax",
                                           "multi_motif", "quad_loads", "skip_unreachable", "pair_prefilter",
                                           "pair_prefilter_protein", "chunked_fused", "chunk_rows", "tiled",
                                           "suffix_occurrences", "prefilter", "sort_hits", "short_order", "time_scan", "drop_last", "block_prefilter", "list_scan_max", "poll_done"};

static int set_option(lm_hip_ctx *ctx, const char *name, double value)
{
    const bool on = value != 0;
    const std::string n(name);
    if (n == "track_argmax") ctx->track_argmax = on;                 // 0 = plain store in score_into
    else if (n == "xlong_store") ctx->xlong_store = on;              // 0 = motifs of 65 ... 88 rows stored in slices
    else if (n == "host_fold") ctx->host_fold = on;                  // 0 = small score_into folds its records on the device
    else if (n == "speculate_order") ctx->speculate_order = on;      // 0 = fused threshold reads the counts first
    else if (n == "sort_hits") ctx->sort_hits = on;                  // 0 = long hit lists through the bucket passes too
    else if (n == "time_scan") ctx->time_scan = on;                  // 1 = events around the scan kernels of fused calls (lm_hip_ctx_last_scan_kernel_ms)
    else if (n == "poll_done") ctx->poll_done = on;                  // 1 = single fused threshold calls poll the ranking kernel's word in pinned memory instead of waiting for their stream
    else if (n == "list_scan_max") ctx->list_scan_max = on;          // 0 = Scanner::max always walks windows of materialised u8 scores
    else if (n == "drop_last") ctx->drop_last = on;                  // 0 = single pair scans of M = 4 k look all M rows up
    else if (n == "short_order") ctx->short_order = on;              // 0 = short hit lists of one job through the five-launch form too
    else if (n == "suffix_argmax") ctx->suffix_argmax = on;          // 0 = fused argmax always scans the whole range
    else if (n == "multi_motif") ctx->multi_motif = on;              // 0 = one motif per workgroup pass in batches
    else if (n == "quad_loads") ctx->quad_loads = on;                // 0 = byte symbol loads in the store kernel
    else if (n == "skip_unreachable") ctx->skip_unreachable = on;    // 0 = scan even when no cell can reach the threshold
    else if (n == "pair_prefilter") ctx->pair_prefilter = on;        // 0 = one symbol per prefilter lookup
    else if (n == "block_prefilter") ctx->block_prefilter = on;      // 0 = protein one-symbol scans load a byte per lane and row
    else if (n == "pair_prefilter_protein") ctx->pair_prefilter_protein = on;  // 1 = 441-row pair scan for K = 21
    else if (n == "chunked_fused") ctx->chunked_fused = on;          // 0 = fused scans of M > 36 go cell by cell
    else if (n == "tiled") ctx->tiled = on;                          // 0 = column counts off 32 / 16 go cell by cell
    else if (n == "prefilter") ctx->use_prefilter = on;
    else if (n == "xcd_remap") (void)on;                             // round-4 option, removed with the remap: accepted, ignored
    else if (n == "chunk_rows") {
        if (!(value >= 64) || !(value <= 2147483648.0))  // (also rejects NaN and +inf: the cast below must be defined)
            return fail(LM_HIP_ERR_BAD_ARGS, "ctx_set_option: chunk_rows must be 64 ... 2^31");
        ctx->chunk_rows = (size_t)value;
    } else if (n == "suffix_occurrences") {
        ctx->suffix_occurrences = value > 0 ? value : 0;
    } else {
        return fail(LM_HIP_ERR_BAD_ARGS, "ctx_set_option: unknown option '%s'", name);
    }
    return LM_HIP_OK;
}

static void read_dev_switches(lm_hip_ctx *ctx)
{
#ifdef LM_HIP_DEV_SWITCHES
    for (const char *name : kOptionNames) {
        std::string env = "LM_HIP_";
        for (const char *c = name; *c; ++c)
            env += (char)toupper(*c);
        if (const char *e = getenv(env.c_str()))
            (void)set_option(ctx, name, atof(e));
    }
#else
    (void)ctx;
    (void)kOptionNames;
#endif
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
    if (ctx->d_short)
        (void)hipFree(ctx->d_short);
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
    for (hipEvent_t e : ctx->scan_ev)
        if (e)
            (void)hipEventDestroy(e);
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

int lm_hip_ctx_set_option(lm_hip_ctx *ctx, const char *name, double value)
{
    if (!ctx || !name)
        return fail(LM_HIP_ERR_BAD_ARGS, "ctx_set_option: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    return set_option(ctx, name, value);
}

int lm_hip_ctx_set_xcd_remap(lm_hip_ctx *ctx, int enabled)  // round-4 ABI, a no-op since round 5 (see the header)
{
    (void)enabled;
    return ctx ? LM_HIP_OK : fail(LM_HIP_ERR_BAD_ARGS, "null context");
}

const char *lm_hip_ctx_last_kernel(lm_hip_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

int lm_hip_ctx_last_scan_counts(lm_hip_ctx *ctx, unsigned long long *hits, unsigned long long *candidates)
{
    if (!ctx)
        return fail(LM_HIP_ERR_BAD_ARGS, "ctx_last_scan_counts: null context");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (hits)
        *hits = ctx->last_hit_count;
    if (candidates)
        *candidates = ctx->last_cand_count;
    return LM_HIP_OK;
}

int lm_hip_ctx_last_phases_ms(lm_hip_ctx *ctx, float phases[4])
{
    if (!ctx || !phases)
        return fail(LM_HIP_ERR_BAD_ARGS, "ctx_last_phases_ms: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    for (int i = 0; i < 4; ++i)
        phases[i] = ctx->last_phase_ms[i];
    return LM_HIP_OK;
}

int lm_hip_ctx_last_scan_info(lm_hip_ctx *ctx, size_t *motif_rows_scanned, size_t *lds_bytes_per_position)
{
    if (!ctx || (!motif_rows_scanned && !lds_bytes_per_position))
        return fail(LM_HIP_ERR_BAD_ARGS, "ctx_last_scan_info: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (motif_rows_scanned)
        *motif_rows_scanned = ctx->last_scan_rows;
    if (lds_bytes_per_position)
        *lds_bytes_per_position = ctx->last_scan_lds_bytes;
    return LM_HIP_OK;
}

int lm_hip_ctx_last_scan_kernel_ms(lm_hip_ctx *ctx, float *ms)
{
    if (!ctx || !ms)
        return fail(LM_HIP_ERR_BAD_ARGS, "ctx_last_scan_kernel_ms: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    *ms = ctx->last_scan_kernel_ms;
    return LM_HIP_OK;
}

}  // extern "C"
