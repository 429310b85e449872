// comm.hip -- the multi-GPU seam of the scoring path behind the C ABI.
//
// The path shards by row ranges (Score::score_rows_into takes a Range for exactly that,
// lightmotif/src/pli/mod.rs:72-78): one process per GPU scores rows [a_g, b_g) of the striped
// matrix + an M-1-row halo with NO collective.  What must come out is what
// StripedScores::{argmax, max, threshold} return for the WHOLE matrix (scores.rs:181-213):
//
//   halo       rank g needs the first M-1 rows of rank g+1; the last rank turns rank 0's
//              rows into the reference's wrap rows (seq.rs:373-378)       -> ncclAllGather
//   argmax     one 32-byte (found, value, row, col) record per rank        -> ncclAllGather
//              + the Generic rule (pli/mod.rs:135-155) over the records
//   max        value of that                                              -> same
//   threshold  per-rank row-major lists concatenated in rank order = the reference's
//              row-major push order (pli/mod.rs:212-218)   -> counts by ncclAllGather, lists by
//              one exact-length ncclBroadcast per rank inside a group
//
// RCCL is bound directly (no torch in between): librccl.so.1 is opened on first use, so the
// library itself loads on hosts without RCCL and a process that already carries an RCCL
// (PyTorch) shares that copy instead of loading a second one.  Payloads are bytes to
// kilobytes except dense threshold lists; xGMI latency, not bandwidth, is what matters.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <chrono>
#include <thread>

#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A ROCm install without the RCCL development headers: the handful of types the dlopen'ed entry points
// take, as nccl.h has declared them since NCCL 2.x (the ABI RCCL keeps).
extern "C" {
typedef struct ncclComm *ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct {
    char internal[NCCL_UNIQUE_ID_BYTES];
} ncclUniqueId;
typedef enum {
    ncclSuccess = 0,
    ncclUnhandledCudaError = 1,
    ncclSystemError = 2,
    ncclInternalError = 3,
    ncclInvalidArgument = 4,
    ncclInvalidUsage = 5,
    ncclRemoteError = 6,
    ncclInProgress = 7
} ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
}
#endif

#include "lm_internal.hpp"

namespace lm {

// What one rank contributes to a merged argmax: 32 bytes, rows in GLOBAL coordinates.
struct MergeRecord {
    long long row, col;
    float value;
    int found;
    long long pad;
};
static_assert(sizeof(MergeRecord) == 32, "MergeRecord is the 32-byte all_gather payload");

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "";
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("LM_HIP_RCCL_LIBRARY"), "librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n)
                continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle)
                break;
            snprintf(r.why, sizeof r.why, "%s", dlerror());
        }
        if (!r.handle)
            return;
        bool all = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p) {
                all = false;
                snprintf(r.why, sizeof r.why, "librccl lacks %s", name);
            }
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));  // required: every wait is bounded by it
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = all;
    });
    return r;
}

int need_rccl()
{
    Rccl &r = rccl();
    if (!r.ok)
        return fail(LM_HIP_ERR_COMM, "RCCL is not available: %s", r.why);
    return LM_HIP_OK;
}

#define LM_NCCL_TRY(expr)                                                                    \
    do {                                                                                     \
        ncclResult_t _r = (expr);                                                            \
        if (_r != ncclSuccess)                                                               \
            return ::lm::fail(LM_HIP_ERR_COMM, "%s failed: %s", #expr, rccl().GetErrorString(_r)); \
    } while (0)

// ArgmaxRecord of a shard (flat index row * cols + col, rows relative to the shard) -> MergeRecord
__global__ void globalize_record(const ArgmaxRecord *__restrict__ local, const unsigned long long cols,
                                 const unsigned long long row_offset, MergeRecord *__restrict__ out)
{
    MergeRecord r;
    r.found = local->found;
    r.value = local->value;
    r.row = local->found ? local->index / (long long)cols + (long long)row_offset : 0;
    r.col = local->found ? local->index % (long long)cols : 0;
    r.pad = 0;
    *out = r;
}

// Last rank: the successor's head rows (rank 0's) become wrap rows, seq.rs:373-378:
// wrap[i][j] = data[i][j+1] for j < C-1, wrap[i][C-1] = default; other ranks copy.
__global__ void halo_fill(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, const unsigned long long n,
                          const unsigned stride, const unsigned cols, const uint8_t def, const int as_wrap)
{
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < n;
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned j = (unsigned)(b % stride);
        uint8_t v = def;
        if (j < cols)
            v = as_wrap ? (j + 1 < cols ? src[b + 1] : def) : src[b];
        else if (!as_wrap)
            v = src[b];
        dst[b] = v;
    }
}

}  // namespace

}  // namespace lm

struct lm_hip_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, nranks = 1, device = 0;
    bool broken = false;  // a collective timed out or failed: the communicator was aborted, every later call fails
    bool undrained = false;  // ... and a stream behind it did not drain within the bound: its buffers are leaked, never freed under it
    long long timeout_ms = 120000;  // LM_HIP_COMM_TIMEOUT_MS when the communicator was made; 0 = wait for ever
    lm::Scratch buf;  // device staging of the collectives
    // pipelined argmax merges (lm_hip_argmax_sharded_begin / _end): two slots, each with its own device
    // records and its own pinned read-back area; the all_gather and the read-back run on `side`
    struct Slot {
        hipEvent_t ready = nullptr, done = nullptr;
        bool pending = false;
    } slot[2];
    hipStream_t side = nullptr;
    hipEvent_t side_tail = nullptr;     // the `done` event of the collective enqueued last on `side`
    lm::Scratch abuf;                  // per slot: ArgmaxRecord | MergeRecord mine | MergeRecord all[nranks]
    lm::MergeRecord *h_slots = nullptr;  // pinned, 2 x nranks
    int next = 0;
    size_t slot_bytes() const { return 32 + sizeof(lm::MergeRecord) * ((size_t)nranks + 1); }
};

using namespace lm;

namespace {

int comm_usable(const lm_hip_comm *comm)
{
    if (comm->broken || !comm->nccl)
        return fail(LM_HIP_ERR_COMM, "the communicator was aborted after a failed or timed-out collective; "
                                     "destroy it and make a new one on every rank");
    return LM_HIP_OK;
}

// A collective that a peer never enters would block its stream for ever.  Every wait on one is a poll
// with a deadline; past it the communicator is aborted (ncclCommAbort ends the kernel that spins on the
// missing peer) and the call -- and every later one on this communicator -- returns LM_HIP_ERR_COMM.
// Work queued BEHIND the stuck collective (read-backs into pinned areas and into result blocks the caller is about to
// free) would still land after the call has returned: ncclCommAbort has released the kernel, so the streams drain in
// bounded time, and they are drained here before anything is handed back or freed.
// (The drain is itself a bounded poll: if ncclCommAbort fails or does not release the kernel, a blocking synchronise would
// hang the very call whose job is to return LM_HIP_ERR_COMM.  A stream that has not drained by then marks the communicator
// `undrained`: its device buffers, pinned areas and streams are leaked at destruction rather than freed under queued work.)
bool drain_bounded(hipStream_t st, long long bound_ms)
{
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady)
            return true;  // drained, or in an error state nothing will write through
        if (std::chrono::duration_cast<std::chrono::milliseconds>(clock::now() - t0).count() > bound_ms)
            return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

void abort_comm(lm_hip_comm *comm, hipStream_t waited_on = nullptr)
{
    comm->broken = true;
    bool released = true;
    if (comm->nccl) {
        released = rccl().CommAbort(comm->nccl) == ncclSuccess;
        comm->nccl = nullptr;
    }
    const long long bound_ms = released ? 5000 : 500;
    if (waited_on && !drain_bounded(waited_on, bound_ms))
        comm->undrained = true;
    if (comm->side && comm->side != waited_on && !drain_bounded(comm->side, bound_ms))
        comm->undrained = true;
    for (auto &sl : comm->slot)
        sl.pending = false;
}

template <class Query>
int wait_bounded(lm_hip_comm *comm, Query query, const char *what, hipStream_t stream = nullptr)
{
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = query();
        if (e == hipSuccess)
            return LM_HIP_OK;
        if (e != hipErrorNotReady) {
            abort_comm(comm, stream);
            return fail(LM_HIP_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        }
        if (spins >= 4096) {  // ~a few hundred microseconds of tight polling cover every healthy merge
            const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(clock::now() - t0).count();
            if (comm->timeout_ms > 0 && ms > comm->timeout_ms) {
                abort_comm(comm, stream);
                return fail(LM_HIP_ERR_COMM, "%s: no completion after %lld ms (rank %d of %d): a peer rank did not "
                                             "enter the collective; communicator aborted (LM_HIP_COMM_TIMEOUT_MS)",
                            what, ms, comm->rank, comm->nranks);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(ms > 50 ? 200 : 20));
        }
    }
}

int wait_stream(lm_hip_comm *comm, hipStream_t st, const char *what)
{
    return wait_bounded(comm, [st] { return hipStreamQuery(st); }, what, st);
}

// Collectives of one communicator must not run concurrently: the synchronous entry points use the
// context's stream, the pipelined merges the communicator's side stream.  A synchronous collective
// therefore waits (on the device) for the side stream's last one; the side stream already waits for
// the context stream through each slot's `ready` event.
int order_after_side(lm_hip_ctx *ctx, lm_hip_comm *comm)
{
    if (comm->side_tail)
        LM_HIP_TRY(hipStreamWaitEvent(ctx->stream, comm->side_tail, 0));
    return LM_HIP_OK;
}

}  // namespace

extern "C" {

// ---- the merge rules, on the host (any transport: RCCL below, MPI, a Rust channel ...) -----

int lm_hip_combine_argmax(const int *found, const lm_hip_coords *best, const float *value, size_t n,
                          int *found_out, lm_hip_coords *best_out, float *value_out)
{
    if ((n && (!found || !best || !value)) || !found_out)
        return fail(LM_HIP_ERR_BAD_ARGS, "combine_argmax: null argument");
    // pli/mod.rs:135-155 over shards in ascending row order: the maximal score; ties go to the
    // LAST cell in (row, col) order; NaN never wins -- except through the first-cell rule, which
    // only the shard holding row 0 applies: it then reports (0, 0) with a NaN value, and that wins
    bool have = false;
    lm_hip_coords b{0, 0};
    float v = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        if (!found[i])
            continue;
        if (value[i] != value[i]) {
            if (best[i].row == 0 && best[i].col == 0) {
                have = true;
                b = best[i];
                v = value[i];
                break;
            }
            continue;
        }
        if (!have || value[i] > v ||
            (value[i] == v && (best[i].row > b.row || (best[i].row == b.row && best[i].col > b.col)))) {
            have = true;
            b = best[i];
            v = value[i];
        }
    }
    *found_out = have ? 1 : 0;
    if (have) {
        if (best_out)
            *best_out = b;
        if (value_out)
            *value_out = v;
    }
    return LM_HIP_OK;
}

// ---- communicator ---------------------------------------------------------------------------

int lm_hip_comm_unique_id(uint8_t *id)
{
    if (!id)
        return fail(LM_HIP_ERR_BAD_ARGS, "comm_unique_id: null output");
    LM_TRY(need_rccl());
    ncclUniqueId u;
    LM_NCCL_TRY(rccl().GetUniqueId(&u));
    static_assert(sizeof u.internal == LM_HIP_COMM_ID_BYTES, "unique id size");
    memcpy(id, u.internal, LM_HIP_COMM_ID_BYTES);
    return LM_HIP_OK;
}

int lm_hip_comm_create(lm_hip_ctx *ctx, const uint8_t *id, int nranks, int rank, lm_hip_comm **out)
{
    if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(LM_HIP_ERR_BAD_ARGS, "comm_create: bad argument");
    *out = nullptr;
    LM_TRY(need_rccl());
    // (the context is NOT locked: ncclCommInitRank blocks until every rank has arrived, and a host that gives up
    //  on it must still be able to use the context from another thread)
    DeviceGuard guard(ctx->device);
    lm_hip_comm *c = new (std::nothrow) lm_hip_comm();
    if (!c)
        return fail(LM_HIP_ERR_OOM, "out of host memory");
    c->rank = rank;
    c->nranks = nranks;
    c->device = ctx->device;
    if (const char *t = getenv("LM_HIP_COMM_TIMEOUT_MS")) {
        const long long v = atoll(t);
        c->timeout_ms = v < 0 ? 0 : v;
    }
    ncclUniqueId u;
    memcpy(u.internal, id, LM_HIP_COMM_ID_BYTES);
    ncclResult_t r = rccl().CommInitRank(&c->nccl, nranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(LM_HIP_ERR_COMM, "ncclCommInitRank failed: %s", rccl().GetErrorString(r));
    }
    *out = c;
    return LM_HIP_OK;
}

int lm_hip_comm_destroy(lm_hip_comm *comm)
{
    if (!comm)
        return LM_HIP_OK;
    DeviceGuard guard(comm->device);
    // first the collectives, then what they read and write: a merge still queued on the side stream is waited for with
    // the usual bound (a peer that never came aborts the communicator, which drains the stream); only then do the
    // communicator, its buffers and events go
    if (comm->side && !comm->broken)
        (void)wait_stream(comm, comm->side, "comm_destroy (pending merges)");
    if (comm->nccl && rccl().ok)
        (void)rccl().CommDestroy(comm->nccl);  // (an aborted communicator is gone already: never destroyed twice)
    comm->nccl = nullptr;
    if (comm->undrained) {
        // work may still be queued on streams the abort could not release: nothing it reads or writes is freed
        comm->buf.forget();
        comm->abuf.forget();
        delete comm;
        return LM_HIP_OK;
    }
    if (comm->side)
        (void)hipStreamDestroy(comm->side);
    for (auto &sl : comm->slot) {
        if (sl.ready) (void)hipEventDestroy(sl.ready);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    comm->buf.release();
    comm->abuf.release();
    if (comm->h_slots)
        (void)hipHostFree(comm->h_slots);
    delete comm;
    return LM_HIP_OK;
}

int lm_hip_comm_info(const lm_hip_comm *comm, int *rank, int *nranks)
{
    if (!comm)
        return fail(LM_HIP_ERR_BAD_ARGS, "comm_info: null communicator");
    if (rank)
        *rank = comm->rank;
    if (nranks)
        *nranks = comm->nranks;
    return LM_HIP_OK;
}

// ---- halo ---------------------------------------------------------------------------------------

int lm_hip_exchange_halo_dptr(lm_hip_ctx *ctx, lm_hip_comm *comm, uint8_t *d_shard, size_t rows,
                              size_t stride, size_t cols, size_t halo_rows, uint8_t default_symbol)
{
    // (argument errors of this block must be the same on every rank -- they are properties of the job,
    //  not of a shard; the one per-shard condition, a shard shorter than the halo, travels WITH the
    //  collective below so that every rank learns of it and none is left waiting)
    if (!ctx || !comm || cols == 0 || stride < cols || (halo_rows && !d_shard))
        return fail(LM_HIP_ERR_BAD_ARGS, "exchange_halo: bad argument");
    if (halo_rows == 0)
        return LM_HIP_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    LM_TRY(comm_usable(comm));
    const size_t nbytes = halo_rows * stride;
    const size_t rec = nbytes + 16;  // head rows + a status word (16 bytes keep the records aligned)
    const int nr = comm->nranks;
    if ((size_t)nr * 16 > kPinnedBytes)
        return fail(LM_HIP_ERR_BAD_ARGS, "exchange_halo: too many ranks");
    LM_TRY(comm->buf.reserve(rec * ((size_t)nr + 1)));
    uint8_t *mine = static_cast<uint8_t *>(comm->buf.ptr);
    uint8_t *all = mine + rec;
    const bool short_shard = rows < halo_rows;
    LM_TRY(order_after_side(ctx, comm));
    if (short_shard)
        LM_HIP_TRY(hipMemsetAsync(mine, default_symbol, nbytes, ctx->stream));
    else
        LM_HIP_TRY(hipMemcpyAsync(mine, d_shard, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    LM_HIP_TRY(hipMemsetAsync(mine + nbytes, short_shard ? 1 : 0, 16, ctx->stream));
    // every rank's head rows to every rank (a few KB in total): no point-to-point pairing to get wrong
    LM_NCCL_TRY(rccl().AllGather(mine, all, rec, ncclUint8, comm->nccl, ctx->stream));
    uint8_t *h_status = static_cast<uint8_t *>(ctx->pinned);
    LM_HIP_TRY(hipMemcpy2DAsync(h_status, 16, all + nbytes, rec, 16, (size_t)nr, hipMemcpyDeviceToHost, ctx->stream));
    LM_TRY(wait_stream(comm, ctx->stream, "exchange_halo (ncclAllGather)"));
    for (int r = 0; r < nr; ++r)
        if (h_status[(size_t)r * 16])
            return fail(LM_HIP_ERR_BAD_ARGS, "exchange_halo: the shard of rank %d has fewer than %zu rows and cannot "
                                             "give its predecessor a halo (reported on every rank)", r, halo_rows);
    const uint8_t *succ = all + (size_t)((comm->rank + 1) % nr) * rec;
    const int as_wrap = comm->rank == nr - 1;
    hipLaunchKernelGGL(halo_fill, dim3((unsigned)std::min<size_t>((nbytes + 255) / 256, 1024)), dim3(256), 0,
                       ctx->stream, d_shard + rows * stride, succ, (unsigned long long)nbytes, (unsigned)stride,
                       (unsigned)cols, default_symbol, as_wrap);
    LM_HIP_TRY(hipGetLastError());
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return LM_HIP_OK;
}

// ---- argmax / max ---------------------------------------------------------------------------

static int combine_records(const MergeRecord *h, int n, int *found, lm_hip_coords *best, float *value)
{
    std::vector<int> f(n);
    std::vector<lm_hip_coords> b(n);
    std::vector<float> v(n);
    for (int i = 0; i < n; ++i) {
        f[i] = h[i].found;
        b[i].row = (size_t)h[i].row;
        b[i].col = (size_t)h[i].col;
        v[i] = h[i].value;
    }
    return lm_hip_combine_argmax(f.data(), b.data(), v.data(), (size_t)n, found, best, value);
}

// all_gather of this rank's record (already on the device at `d_mine`) + the combine rule
static int gather_and_combine(lm_hip_ctx *ctx, lm_hip_comm *comm, const MergeRecord *d_mine, MergeRecord *d_all,
                              int *found, lm_hip_coords *best, float *value)
{
    const int n = comm->nranks;
    if (sizeof(MergeRecord) * (size_t)n > kPinnedBytes / 2)
        return fail(LM_HIP_ERR_BAD_ARGS, "merge: too many ranks");
    LM_TRY(order_after_side(ctx, comm));
    LM_NCCL_TRY(rccl().AllGather(d_mine, d_all, sizeof(MergeRecord), ncclUint8, comm->nccl, ctx->stream));
    MergeRecord *h = static_cast<MergeRecord *>(ctx->pinned);
    LM_HIP_TRY(hipMemcpyAsync(h, d_all, sizeof(MergeRecord) * n, hipMemcpyDeviceToHost, ctx->stream));
    LM_TRY(wait_stream(comm, ctx->stream, "argmax merge (ncclAllGather)"));
    return combine_records(h, n, found, best, value);
}

int lm_hip_merge_argmax(lm_hip_ctx *ctx, lm_hip_comm *comm, int found_local, const lm_hip_coords *best_local,
                        float value_local, size_t row_offset, int *found, lm_hip_coords *best, float *value)
{
    if (!ctx || !comm || !found || (found_local && !best_local))
        return fail(LM_HIP_ERR_BAD_ARGS, "merge_argmax: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    LM_TRY(comm_usable(comm));
    LM_TRY(comm->buf.reserve(sizeof(MergeRecord) * (comm->nranks + 1)));
    MergeRecord *d = static_cast<MergeRecord *>(comm->buf.ptr);
    MergeRecord mine{};
    mine.found = found_local ? 1 : 0;
    if (found_local) {
        mine.row = (long long)(best_local->row + row_offset);
        mine.col = (long long)best_local->col;
        mine.value = value_local;
    }
    // staged through the pinned block (the tail: gather_and_combine reads the head)
    MergeRecord *stage = static_cast<MergeRecord *>(ctx->pinned) + comm->nranks;
    *stage = mine;
    LM_HIP_TRY(hipMemcpyAsync(d, stage, sizeof(MergeRecord), hipMemcpyHostToDevice, ctx->stream));
    return gather_and_combine(ctx, comm, d, d + 1, found, best, value);
}

// the shard's ArgmaxRecord on the device (tracked by the store kernel, or reduced now) -> MergeRecord at d_mine
static int local_record(lm_hip_ctx *ctx, const lm_hip_scores *s, size_t row_offset, ArgmaxRecord *d_local,
                        MergeRecord *d_mine)
{
    const int rule = row_offset == 0 ? 1 : 0;  // only the shard holding row 0 holds scores[0][0]
    if (s->rows == 0) {
        LM_HIP_TRY(hipMemsetAsync(d_local, 0, sizeof(ArgmaxRecord), ctx->stream));  // found = 0
    } else if (s->best_valid && (int)s->first_cell_rule == rule) {
        // tracked by the store kernel that wrote the shard (score_into on a handle): read in place
        d_local = s->d_best;
    } else {
        LM_TRY(launch_argmax_device(ctx, s->d_data, s->rows, s->stride, s->cols, rule, d_local));
    }
    hipLaunchKernelGGL(globalize_record, dim3(1), dim3(1), 0, ctx->stream, (const ArgmaxRecord *)d_local,
                       (unsigned long long)s->cols, (unsigned long long)row_offset, d_mine);
    LM_HIP_TRY(hipGetLastError());
    return LM_HIP_OK;
}

int lm_hip_argmax_sharded(lm_hip_ctx *ctx, lm_hip_comm *comm, const lm_hip_scores *s, size_t row_offset,
                          int *found, lm_hip_coords *best, float *value)
{
    if (!ctx || !comm || !s || !found)
        return fail(LM_HIP_ERR_BAD_ARGS, "argmax_sharded: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    LM_TRY(comm_usable(comm));
    LM_TRY(comm->buf.reserve(sizeof(MergeRecord) * (comm->nranks + 1) + sizeof(ArgmaxRecord)));
    MergeRecord *d = static_cast<MergeRecord *>(comm->buf.ptr);
    ArgmaxRecord *d_local = reinterpret_cast<ArgmaxRecord *>(d + comm->nranks + 1);
    LM_TRY(local_record(ctx, s, row_offset, d_local, d));
    return gather_and_combine(ctx, comm, d, d + 1, found, best, value);
}

// ---- pipelined form: the merge of step i overlaps the scoring of step i+1 --------------------

static int async_init(lm_hip_comm *comm)
{
    if (comm->side)
        return LM_HIP_OK;
    LM_TRY(comm->abuf.reserve(2 * comm->slot_bytes()));
    LM_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&comm->h_slots), 2 * sizeof(MergeRecord) * comm->nranks,
                             hipHostMallocDefault));
    for (auto &sl : comm->slot) {
        LM_HIP_TRY(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
        LM_HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    LM_HIP_TRY(hipStreamCreateWithFlags(&comm->side, hipStreamNonBlocking));
    return LM_HIP_OK;
}

int lm_hip_argmax_sharded_begin(lm_hip_ctx *ctx, lm_hip_comm *comm, const lm_hip_scores *s, size_t row_offset,
                                int *ticket)
{
    if (!ctx || !comm || !s || !ticket)
        return fail(LM_HIP_ERR_BAD_ARGS, "argmax_sharded_begin: null argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    LM_TRY(comm_usable(comm));
    LM_TRY(async_init(comm));
    const int k = comm->next;
    lm_hip_comm::Slot &sl = comm->slot[k];
    // (a property of the calling sequence, which every rank of an SPMD host runs alike: all ranks fail here, or none)
    if (sl.pending)
        return fail(LM_HIP_ERR_BAD_ARGS, "argmax_sharded_begin: two merges are in flight, collect one with _end first");
    char *base = static_cast<char *>(comm->abuf.ptr) + (size_t)k * comm->slot_bytes();
    ArgmaxRecord *d_local = reinterpret_cast<ArgmaxRecord *>(base);
    MergeRecord *d_mine = reinterpret_cast<MergeRecord *>(base + 32);
    MergeRecord *d_all = d_mine + 1;
    LM_TRY(local_record(ctx, s, row_offset, d_local, d_mine));
    // from here on the scores (and the tracked record) may be overwritten: the side stream only
    // needs the 32-byte record.  `ready` also orders this collective after every synchronous one
    // enqueued on the context's stream before it.
    LM_HIP_TRY(hipEventRecord(sl.ready, ctx->stream));
    LM_HIP_TRY(hipStreamWaitEvent(comm->side, sl.ready, 0));
    LM_NCCL_TRY(rccl().AllGather(d_mine, d_all, sizeof(MergeRecord), ncclUint8, comm->nccl, comm->side));
    LM_HIP_TRY(hipMemcpyAsync(comm->h_slots + (size_t)k * comm->nranks, d_all, sizeof(MergeRecord) * comm->nranks,
                              hipMemcpyDeviceToHost, comm->side));
    LM_HIP_TRY(hipEventRecord(sl.done, comm->side));
    comm->side_tail = sl.done;
    sl.pending = true;
    comm->next ^= 1;
    *ticket = k;
    return LM_HIP_OK;
}

int lm_hip_argmax_sharded_end(lm_hip_ctx *ctx, lm_hip_comm *comm, int ticket, int *found, lm_hip_coords *best,
                              float *value)
{
    if (!ctx || !comm || !found || ticket < 0 || ticket > 1)
        return fail(LM_HIP_ERR_BAD_ARGS, "argmax_sharded_end: bad argument");
    hipEvent_t done;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        LM_TRY(comm_usable(comm));
        if (!comm->slot[ticket].pending)
            return fail(LM_HIP_ERR_BAD_ARGS, "argmax_sharded_end: no merge in flight under this ticket");
        done = comm->slot[ticket].done;
    }
    {
        DeviceGuard guard(ctx->device);  // (the wait itself leaves the context free for other threads)
        LM_TRY(wait_bounded(comm, [done] { return hipEventQuery(done); }, "pipelined argmax merge (ncclAllGather)"));
    }
    std::lock_guard<std::mutex> lock(ctx->mu);
    comm->slot[ticket].pending = false;
    return combine_records(comm->h_slots + (size_t)ticket * comm->nranks, comm->nranks, found, best, value);
}

int lm_hip_merge_max(lm_hip_ctx *ctx, lm_hip_comm *comm, int found_local, float value_local, int *found,
                     float *value)
{
    // Maximum::max = the value at the merged argmax (pli/mod.rs:158-160); without coordinates a
    // NaN from the first-cell rule cannot be told apart, so NaN contributions are dropped here --
    // use lm_hip_merge_argmax when the matrix may hold NaN
    lm_hip_coords c{1, 0};
    return lm_hip_merge_argmax(ctx, comm, found_local && value_local == value_local, &c, value_local, 0, found,
                               nullptr, value);
}

// ---- threshold ----------------------------------------------------------------------------------

int lm_hip_merge_threshold(lm_hip_ctx *ctx, lm_hip_comm *comm, const lm_hip_coords *coords, size_t n,
                           size_t row_offset, lm_hip_coords **all, size_t *n_all)
{
    if (!ctx || !comm || !all || !n_all || (n && !coords))
        return fail(LM_HIP_ERR_BAD_ARGS, "merge_threshold: null argument");
    *all = nullptr;
    *n_all = 0;
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    LM_TRY(comm_usable(comm));
    LM_TRY(order_after_side(ctx, comm));
    const int nr = comm->nranks;
    // (1) hit counts
    LM_TRY(comm->buf.reserve(sizeof(unsigned long long) * (nr + 1)));
    unsigned long long *d_cnt = static_cast<unsigned long long *>(comm->buf.ptr);
    unsigned long long *h = static_cast<unsigned long long *>(ctx->pinned);
    h[nr] = (unsigned long long)n;
    LM_HIP_TRY(hipMemcpyAsync(d_cnt + nr, h + nr, 8, hipMemcpyHostToDevice, ctx->stream));
    LM_NCCL_TRY(rccl().AllGather(d_cnt + nr, d_cnt, 8, ncclUint8, comm->nccl, ctx->stream));
    LM_HIP_TRY(hipMemcpyAsync(h, d_cnt, 8 * nr, hipMemcpyDeviceToHost, ctx->stream));
    LM_TRY(wait_stream(comm, ctx->stream, "threshold merge, hit counts (ncclAllGather)"));
    std::vector<unsigned long long> cnt(h, h + nr), off(nr + 1, 0);
    for (int r = 0; r < nr; ++r)
        off[r + 1] = off[r] + cnt[r];
    const unsigned long long total = off[nr];
    if (total == 0)
        return LM_HIP_OK;
    // (2) the lists: rows to global coordinates, then one exact-length broadcast per rank straight
    // into its slot of the concatenation (rank order = ascending rows = row-major order)
    lm_hip_coords *res = static_cast<lm_hip_coords *>(result_alloc(total * sizeof(lm_hip_coords)));
    if (!res)
        return fail(LM_HIP_ERR_OOM, "out of host memory (%llu merged hits)", total);
    lm_hip_coords *mine = res + off[comm->rank];
    for (size_t i = 0; i < n; ++i) {
        mine[i].row = coords[i].row + row_offset;
        mine[i].col = coords[i].col;
    }
    int st = comm->buf.reserve(total * sizeof(lm_hip_coords));
    if (st != LM_HIP_OK) {
        result_free(res);
        return st;
    }
    lm_hip_coords *d_all = static_cast<lm_hip_coords *>(comm->buf.ptr);
    hipError_t e = hipSuccess;
    if (n)
        e = hipMemcpyAsync(d_all + off[comm->rank], mine, n * sizeof(lm_hip_coords), hipMemcpyHostToDevice,
                           ctx->stream);
    ncclResult_t nr_ = ncclSuccess;
    if (e == hipSuccess && nr > 1) {
        nr_ = rccl().GroupStart();
        for (int r = 0; r < nr && nr_ == ncclSuccess; ++r)
            if (cnt[r])
                nr_ = rccl().Broadcast(d_all + off[r], d_all + off[r], cnt[r] * sizeof(lm_hip_coords), ncclUint8, r,
                                       comm->nccl, ctx->stream);
        const ncclResult_t ge = rccl().GroupEnd();
        if (nr_ == ncclSuccess)
            nr_ = ge;
    }
    if (e == hipSuccess && nr_ == ncclSuccess && nr > 1) {
        // everything but this rank's own slice comes back from the device
        if (off[comm->rank])
            e = hipMemcpyAsync(res, d_all, off[comm->rank] * sizeof(lm_hip_coords), hipMemcpyDeviceToHost, ctx->stream);
        const unsigned long long after = off[comm->rank + 1];
        if (e == hipSuccess && total > after)
            e = hipMemcpyAsync(res + after, d_all + after, (total - after) * sizeof(lm_hip_coords),
                               hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess && nr_ == ncclSuccess) {
        const int st2 = wait_stream(comm, ctx->stream, "threshold merge, hit lists (ncclBroadcast group)");
        if (st2 != LM_HIP_OK) {
            result_free(res);
            return st2;
        }
    }
    if (e != hipSuccess || nr_ != ncclSuccess) {
        result_free(res);
        if (nr_ != ncclSuccess)
            return fail(LM_HIP_ERR_COMM, "threshold merge failed: %s", rccl().GetErrorString(nr_));
        return fail(LM_HIP_ERR_HIP, "threshold merge failed: %s", hipGetErrorString(e));
    }
    *all = res;
    *n_all = (size_t)total;
    return LM_HIP_OK;
}

}  // extern "C"
