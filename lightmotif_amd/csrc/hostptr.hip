// hostptr.hip -- the host-pointer entry points: what the reference-side shim binds.
//
//   lm_hip_score_f32      Score::score_rows_into on host matrices   (pli/mod.rs:72-106, avx2.rs:889-904)
//   lm_hip_score_u8_host  the same with a DiscreteMatrix, u8 scores   (avx2.rs:294-347, 921-931; scan.rs:174-178)
//   lm_hip_argmax_f32     StripedScores::argmax                     (scores.rs:181-186, pli/mod.rs:135-155)
//   lm_hip_max_f32        StripedScores::max                        (scores.rs:188-192, pli/mod.rs:158-160)
//   lm_hip_threshold_f32  StripedScores::threshold                  (scores.rs:207-213, pli/mod.rs:210-221)
//
// The caller's matrices are pageable host memory (a Rust Vec); every call moves 1 B per position up and 4 B per
// position down over PCIe, so the link is the floor.  What this file is about is paying nothing else:
//
//  * LANES.  Every host thread that calls in gets a lane of its own -- a context (its own stream), device staging
//    buffers that persist between calls, and a small cache of device-side PSSM tables keyed on the weights -- so the
//    CLI's worker threads (main.rs:270) and GIL-released Python threads (lightmotif-py lib.rs:865) overlap instead of
//    queueing on one context, and a loop over one motif (lightmotif-bench dna.rs:104-107) builds its tables once.
//    A lane whose thread has exited is handed to the next new thread.  Lanes are dealt ROUND-ROBIN over the usable
//    devices of the process (the reference's own parallel axis is the CLI's worker threads over (motif, sequence) jobs,
//    lightmotif-cli main.rs:240-378, 502-524: eight threads on an 8-GPU node then use eight GPUs and eight PCIe links
//    with no change to the caller); $LM_HIP_DEVICE (read once) pins every lane to one ordinal, lm_hip_host_bind_thread
//    one thread.
//  * SMALL calls (the reference's own bench: 464 165 positions; a Scanner block: 256 rows) are latency-bound:
//    one pageable copy up, one kernel, one pageable copy down, one synchronisation; no allocation, no table build.
//    The tiniest (<= 128 KB each way) skip the copy commands altogether: the kernel reads the symbols from and writes
//    the scores to pinned host memory (tools/kbench/hostpipe_bench.hip: 15.9 us against 27.8 us per iteration).
//  * LARGE calls (>= 96 MB of scores) are link-bound and run as a three-stage pipeline over row tiles: an uploader thread copies tile
//    t + 1 .. t + 3 (pageable H2D, 56 GB/s) while the store kernel of tile t runs and tile t - 1 travels back.  The
//    way back is the long one (4 B per position): the runtime's pageable D2H reaches 48 GB/s, a copy into pinned
//    memory 56.6 GB/s -- so tiles land in a ring of four pinned 32 MB buffers and four copier threads move them into
//    the caller's matrix while the next tiles are in flight (measured: hostpipe_bench, profiles/r04_hostpipe_bench.txt).
//    The ring, its streams and the device tile buffers exist once per DEVICE: large calls of several threads on one
//    device take turns on it (the link is the bound; side by side they only get in each other's way), calls on
//    different devices run side by side.  The ring is allocated, and the uploader / copier threads run, on the CPUs of
//    the GPU's NUMA node (/sys/bus/pci/devices/<bdf>/numa_node), inside the affinity mask the process already has.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cctype>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

#include <pthread.h>
#include <sched.h>

#include "lm_internal.hpp"

namespace lm {
namespace {

constexpr size_t kZeroCopyBytes = 128u << 10;   // tiniest calls: symbols / scores straight from / to pinned memory
constexpr size_t kKeepBytes = 256u << 20;       // staging buffers above this are released when the call ends
constexpr size_t kPipeMinOutBytes = 96u << 20;  // score matrices from here on take the tile pipeline (below, its ramp --
                                                // helper threads, one tile each of upload / read-back / copy-out latency --
                                                // costs more than the 15 % the pinned read-back gains: r04_host_pointer.json)
constexpr size_t kPipeMinTileBytes = 4u << 20;  // tiles: a twelfth of the score matrix, between this and a ring slot
constexpr size_t kTileBytes = 32u << 20;        // one tile of scores (and one pinned ring slot)
constexpr int kInSlots = 3, kOutSlots = 4, kCopiers = 4;
constexpr int kMaxOutSlots = 8;

// Shape of the pipeline.  Shipped: the constants above; a -DLM_HIP_DEV_SWITCHES build (tools/build_variant.py) reads
// LM_HIP_PIPE_{TILE_MB,OUT_SLOTS,COPIERS} once, for sweeps (tools/hostpipe_sweep.sh).
struct PipeShape {
    size_t tile_bytes = kTileBytes;
    int out_slots = kOutSlots, copiers = kCopiers;
};
const PipeShape &pipe_shape()
{
    static const PipeShape shape = [] {
        PipeShape s;
#ifdef LM_HIP_DEV_SWITCHES
        if (const char *e = getenv("LM_HIP_PIPE_TILE_MB"))
            s.tile_bytes = (size_t)std::max(1, std::min(atoi(e), 256)) << 20;
        if (const char *e = getenv("LM_HIP_PIPE_OUT_SLOTS"))
            s.out_slots = std::max(2, std::min(atoi(e), kMaxOutSlots));
        if (const char *e = getenv("LM_HIP_PIPE_COPIERS"))
            s.copiers = std::max(1, std::min(atoi(e), 32));
#endif
        return s;
    }();
    return shape;
}
constexpr size_t kPssmCache = 16;

struct CachedPssm {
    uint64_t hash = 0, stamp = 0;
    lm_hip_pssm *p = nullptr;
};

struct HostLane {
    lm_hip_ctx *ctx = nullptr;
    int device = -1;
    Scratch d_in, d_out;
    uint8_t *zc = nullptr;  // 2 x kZeroCopyBytes, pinned and device-visible
    std::vector<CachedPssm> pssms;
    uint64_t stamp = 0;
    uint64_t trim_epoch = 0;  // g_trim_epoch as of this lane's last trim
    // the f32 score matrix the last lm_hip_score_f32 of this lane left on the device, dense (lm_hip_host_reuse_scores):
    // where it is, whose it was on the host, and a digest of a sample of what the caller got
    const float *kept_dev = nullptr;
    const float *kept_host = nullptr;
    size_t kept_rows = 0, kept_cols = 0, kept_stride = 0;
    uint64_t kept_digest = 0;
    size_t reuses = 0;
    const void *piece_dev = nullptr;  // where the last score call left its whole (dense) result on the device, if it did
    // ... and, for matrices small enough for the tracking store kernel, the best cell that kernel found on the way
    // (an lm_hip_scores record whose d_data is BORROWED from the staging above, never owned): lm_hip_argmax_f32 on the
    // kept matrix is then the fold of a few records, like lm_hip_argmax on a handle
    lm_hip_scores *track = nullptr;
    bool tracked = false;
};

// lm_hip_host_reuse_scores: off unless the embedder asks for it
std::atomic<bool> g_reuse_scores{false};

// lm_hip_host_trim bumps it: a lane that belongs to a live thread hands its staging back at the end of its next call
std::atomic<uint64_t> g_trim_epoch{0};

// process lifetime (never destroyed: host threads may still be inside the library when the process exits)
std::mutex &lanes_mu()
{
    static std::mutex *mu = new std::mutex();
    return *mu;
}
std::vector<HostLane *> &idle_lanes()
{
    static std::vector<HostLane *> *v = new std::vector<HostLane *>();
    return *v;
}

struct LaneRef {
    HostLane *lane = nullptr;
    int want_device = -1;  // lm_hip_host_bind_thread
    ~LaneRef()
    {
        if (lane) {  // the thread is gone; its lane (stream idle: every call synchronises) serves the next one
            std::lock_guard<std::mutex> lock(lanes_mu());
            idle_lanes().push_back(lane);
        }
    }
};
thread_local LaneRef t_lane;

// $LM_HIP_DEVICE, read ONCE per process: every lane on that ordinal (-1: not set, -2: set to something that is no ordinal)
int env_device()
{
    static const int dev = [] {
        const char *e = getenv("LM_HIP_DEVICE");
        if (!e || !*e)
            return -1;
        char *end = nullptr;
        const long v = strtol(e, &end, 10);
        return (end && *end == 0 && v >= 0 && v < 4096) ? (int)v : -2;
    }();
    return dev;
}

// HIP ordinals of the usable (gfx950) devices, in ordinal order
const std::vector<int> &usable_devices()
{
    static const std::vector<int> *list = [] {
        auto *v = new std::vector<int>();
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess)
            n = 0;
        for (int d = 0; d < n; ++d) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0)
                v->push_back(d);
        }
        return v;
    }();
    return *list;
}

bool is_usable(int dev)
{
    const std::vector<int> &devs = usable_devices();
    return std::find(devs.begin(), devs.end(), dev) != devs.end();
}

// lm_hip_host_spread_lanes: new lanes are dealt over every usable device in turn (a single process driving a whole node:
// the reference CLI's worker threads, main.rs:240-378).  Off by default: a job of one process per GPU with every GPU visible
// must not open contexts, pinned rings and PSSM caches on its neighbours' devices just because it scores from two threads.
std::atomic<int> g_spread{0};

// Device of a NEW lane: the thread's binding, else $LM_HIP_DEVICE, else -- spread on -- the next usable device in turn, else
// the process's host-pointer device: the HIP device current in the thread that made the FIRST unbound call (what a rank set
// with hipSetDevice / torch.cuda.set_device before it started scoring; ordinal 0 if it never did).
int pick_device(int *out)
{
    static std::atomic<unsigned> next{0};
    static std::atomic<int> home{-1};
    if (t_lane.want_device >= 0) {
        *out = t_lane.want_device;
    } else if (env_device() != -1) {
        if (env_device() < 0 || !is_usable(env_device()))
            return fail(LM_HIP_ERR_NO_DEVICE, "LM_HIP_DEVICE does not name a usable gfx950 device (lm_hip_device_ordinal lists them)");
        *out = env_device();
    } else {
        const std::vector<int> &devs = usable_devices();
        if (devs.empty())
            return fail(LM_HIP_ERR_NO_DEVICE, "no gfx950 device available");
        if (g_spread.load(std::memory_order_relaxed)) {
            *out = devs[next.fetch_add(1, std::memory_order_relaxed) % devs.size()];
        } else {
            int h = home.load(std::memory_order_acquire);
            if (h < 0) {
                int cur = 0;
                if (hipGetDevice(&cur) != hipSuccess || !is_usable(cur))
                    cur = devs[0];
                int expected = -1;
                home.compare_exchange_strong(expected, cur, std::memory_order_acq_rel);
                h = home.load(std::memory_order_acquire);
            }
            *out = h;
        }
    }
    return LM_HIP_OK;
}

int acquire_lane(HostLane **out)
{
    const int demanded = t_lane.want_device >= 0 ? t_lane.want_device : env_device();
    if (t_lane.lane && demanded >= 0 && t_lane.lane->device != demanded) {  // re-bound: this lane serves someone else
        std::lock_guard<std::mutex> lock(lanes_mu());
        idle_lanes().push_back(t_lane.lane);
        t_lane.lane = nullptr;
    }
    if (!t_lane.lane) {
        {
            std::lock_guard<std::mutex> lock(lanes_mu());
            std::vector<HostLane *> &idle = idle_lanes();
            for (size_t i = idle.size(); i-- > 0;)
                if (demanded < 0 || idle[i]->device == demanded) {
                    t_lane.lane = idle[i];
                    t_lane.lane->kept_dev = nullptr;  // (what another thread's call left behind is not this thread's)
                    t_lane.lane->reuses = 0;
                    idle.erase(idle.begin() + (long)i);
                    break;
                }
        }
        if (!t_lane.lane) {
            int dev = 0;
            LM_TRY(pick_device(&dev));
            lm_hip_ctx *ctx = nullptr;
            LM_TRY(lm_hip_ctx_create(dev, &ctx));
            HostLane *lane = new (std::nothrow) HostLane();
            if (!lane) {
                lm_hip_ctx_destroy(ctx);
                return fail(LM_HIP_ERR_OOM, "out of host memory");
            }
            lane->ctx = ctx;
            lane->device = dev;
            lane->trim_epoch = g_trim_epoch.load(std::memory_order_acquire);
            DeviceGuard guard(ctx->device);
            if (hipHostMalloc(reinterpret_cast<void **>(&lane->zc), 2 * kZeroCopyBytes, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                lane->zc = nullptr;  // tiny calls take the copy path then
            }
            t_lane.lane = lane;
        }
    }
    *out = t_lane.lane;
    return LM_HIP_OK;
}

// ---- NUMA placement of the pipeline's helper threads and its pinned ring ------------------------------------------------

struct NodeCpus {
    int node = -1;
    bool valid = false;  // `set` = the node's CPUs inside the process's own affinity mask, non-empty
    cpu_set_t set;
};

// "0-63,128-191" -> cpu set
bool parse_cpulist(const char *text, cpu_set_t *set)
{
    CPU_ZERO(set);
    bool any = false;
    const char *p = text;
    while (*p) {
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p)
            break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            if (end == p + 1)
                break;
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0) {
                CPU_SET((int)c, set);
                any = true;
            }
        if (*p == ',')
            ++p;
        else
            break;
    }
    return any;
}

NodeCpus probe_node(int device)
{
    NodeCpus out;
    CPU_ZERO(&out.set);
    char bdf[64] = "";
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) {
        (void)hipGetLastError();
        return out;
    }
    for (char *c = bdf; *c; ++c)
        *c = (char)tolower((unsigned char)*c);
    char path[160], text[4096] = "";
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE *f = fopen(path, "r");
    if (!f)
        return out;
    if (fscanf(f, "%d", &out.node) != 1)
        out.node = -1;
    fclose(f);
    if (out.node < 0)
        return out;  // the platform reports no affinity (single node, or a VM): nothing to bind to
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", out.node);
    f = fopen(path, "r");
    if (!f)
        return out;
    const bool got = fgets(text, sizeof text, f) != nullptr;
    fclose(f);
    cpu_set_t node_set, mine;
    if (!got || !parse_cpulist(text, &node_set) || sched_getaffinity(0, sizeof mine, &mine) != 0)
        return out;
    CPU_AND(&out.set, &node_set, &mine);  // never outside what the process was given (taskset, cgroup cpusets)
    out.valid = CPU_COUNT(&out.set) > 0;
    return out;
}

const NodeCpus &device_node(int device)
{
    static std::mutex mu;
    static std::vector<NodeCpus *> *cache = new std::vector<NodeCpus *>();
    std::lock_guard<std::mutex> lock(mu);
    if ((size_t)device >= cache->size())
        cache->resize((size_t)device + 1, nullptr);
    if (!(*cache)[(size_t)device])
        (*cache)[(size_t)device] = new NodeCpus(probe_node(device));
    return *(*cache)[(size_t)device];
}

// helper threads only (never the caller's own thread)
void bind_this_thread_to_node(int device)
{
    const NodeCpus &n = device_node(device);
    if (n.valid)
        (void)pthread_setaffinity_np(pthread_self(), sizeof n.set, &n.set);
}

uint64_t hash_weights(const float *w, size_t m, size_t stride, size_t k)
{
    uint64_t h = 0xcbf29ce484222325ull ^ (m * 0x9E3779B97F4A7C15ull) ^ (k << 48);
    for (size_t j = 0; j < m; ++j)
        for (size_t s = 0; s < k; ++s) {
            uint32_t bits;
            memcpy(&bits, &w[j * stride + s], 4);
            h = (h ^ bits) * 0x100000001b3ull;
        }
    return h;
}

// The lane's device tables for this matrix: built on first sight, found again by hash + full comparison.
int lane_pssm(HostLane *lane, const float *w, size_t m, size_t stride, size_t k, lm_hip_pssm **out)
{
    if (!w && m)
        return fail(LM_HIP_ERR_BAD_ARGS, "score: null pssm");
    if (k == 0 || k > 256 || stride < k)
        return fail(LM_HIP_ERR_BAD_ARGS, "score: bad alphabet size %zu / pssm stride %zu", k, stride);
    const uint64_t h = hash_weights(w, m, stride, k);
    for (CachedPssm &e : lane->pssms) {
        if (e.hash != h || e.p->m != m || e.p->k != k)
            continue;
        bool same = true;
        for (size_t j = 0; j < m && same; ++j)
            same = memcmp(&e.p->host[j * k], &w[j * stride], k * sizeof(float)) == 0;
        if (same) {
            e.stamp = ++lane->stamp;
            *out = e.p;
            return LM_HIP_OK;
        }
    }
    if (lane->pssms.size() >= kPssmCache) {  // the stream is idle between calls: safe to destroy
        size_t oldest = 0;
        for (size_t i = 1; i < lane->pssms.size(); ++i)
            if (lane->pssms[i].stamp < lane->pssms[oldest].stamp)
                oldest = i;
        lm_hip_pssm_destroy(lane->pssms[oldest].p);
        lane->pssms.erase(lane->pssms.begin() + (long)oldest);
    }
    lm_hip_pssm *p = nullptr;
    LM_TRY(lm_hip_pssm_create(lane->ctx, w, m, stride, k, &p));
    lane->pssms.push_back(CachedPssm{h, ++lane->stamp, p});
    *out = p;
    return LM_HIP_OK;
}

void trim(Scratch &s)
{
    if (s.bytes > kKeepBytes)
        s.release();
}

// End of a call on `lane` (its stream is idle): staging above the keep threshold goes back; after an lm_hip_host_trim on
// another thread, everything does.
void lane_trim(HostLane *lane)
{
    const uint64_t epoch = g_trim_epoch.load(std::memory_order_acquire);
    if (lane->trim_epoch != epoch) {
        lane->trim_epoch = epoch;
        lane->d_in.release();
        lane->d_out.release();
        lane->ctx->scratch.release();
        lane->ctx->scratch2.release();
        lane->ctx->chunk_scores.release();
        lane->ctx->scan_buf.release();
        return;
    }
    trim(lane->d_in);
    trim(lane->d_out);
}

void copy_rows(char *dst, size_t dst_pitch, const char *src, size_t src_pitch, size_t width, size_t nrows)
{
    if (dst_pitch == width && src_pitch == width) {
        memcpy(dst, src, width * nrows);
        return;
    }
    for (size_t r = 0; r < nrows; ++r)
        memcpy(dst + r * dst_pitch, src + r * src_pitch, width);
}

// ---- the tile pipeline of large calls ---------------------------------------------------------------------------------

struct BigPipe {
    std::mutex mu;  // one large call at a time
    int device = -1;
    hipStream_t s_up = nullptr, s_dn = nullptr;
    hipEvent_t kdone[kInSlots] = {}, landed[kMaxOutSlots] = {};
    Scratch d_in, d_out;
    char *pinned = nullptr;  // out_slots x tile_bytes
};

// one per device (process lifetime)
std::mutex &pipes_mu()
{
    static std::mutex *mu = new std::mutex();
    return *mu;
}
std::vector<BigPipe *> &pipes()
{
    static std::vector<BigPipe *> *v = new std::vector<BigPipe *>();
    return *v;
}
BigPipe &big_pipe(int device)
{
    std::lock_guard<std::mutex> lock(pipes_mu());
    std::vector<BigPipe *> &v = pipes();
    if ((size_t)device >= v.size())
        v.resize((size_t)device + 1, nullptr);
    if (!v[(size_t)device])
        v[(size_t)device] = new BigPipe();
    return *v[(size_t)device];
}

int pipe_prepare(BigPipe &bp, int device)
{
    if (bp.device == device)
        return LM_HIP_OK;
    if (bp.device >= 0)  // (cannot happen: one pipe per device) -- the caller falls back to pieces on its own lane
        return LM_HIP_ERR_CAPACITY;
    hipStream_t up = nullptr, dn = nullptr;
    hipEvent_t ev[kInSlots + kMaxOutSlots] = {};
    char *pin = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&dn, hipStreamNonBlocking);
    for (int i = 0; i < kInSlots + kMaxOutSlots && e == hipSuccess; ++i)
        e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    if (e == hipSuccess) {
        // the ring's pages come from the node of the thread that allocates them (first touch under the default policy):
        // allocate from a thread that sits on the GPU's node, so the DMA writes and the copiers' reads stay on one socket
        const size_t bytes = (size_t)pipe_shape().out_slots * pipe_shape().tile_bytes;
        auto alloc = [&] {
            if (hipSetDevice(device) != hipSuccess) {
                e = hipGetLastError();
                return;
            }
            bind_this_thread_to_node(device);
            e = hipHostMalloc(reinterpret_cast<void **>(&pin), bytes, hipHostMallocDefault);
        };
        if (device_node(device).valid) {
            try {
                std::thread(alloc).join();
            } catch (const std::exception &) {
                alloc();  // no thread to be had: from here, wherever this thread runs
            }
        } else {
            e = hipHostMalloc(reinterpret_cast<void **>(&pin), bytes, hipHostMallocDefault);
        }
    }
    if (e != hipSuccess) {  // nothing half-made is kept
        for (hipEvent_t x : ev)
            if (x)
                (void)hipEventDestroy(x);
        if (up)
            (void)hipStreamDestroy(up);
        if (dn)
            (void)hipStreamDestroy(dn);
        return fail(e == hipErrorOutOfMemory ? LM_HIP_ERR_OOM : LM_HIP_ERR_HIP, "host pipeline set-up failed: %s",
                    hipGetErrorString(e));
    }
    bp.s_up = up;
    bp.s_dn = dn;
    for (int i = 0; i < kInSlots; ++i)
        bp.kdone[i] = ev[i];
    for (int i = 0; i < kMaxOutSlots; ++i)
        bp.landed[i] = ev[kInSlots + i];
    bp.pinned = pin;
    bp.device = device;
    return LM_HIP_OK;
}

// One pipelined job: `upload` (uploader thread; blocks until tile t is in input slot `slot`), `compute` (calling
// thread: enqueues tile t's kernels on the lane's stream), `download` (calling thread: enqueues the copy of output
// slot `slot` into the pinned slot on bp.s_dn) and `copy_out` (copier j of n: its share of tile t, pinned -> caller).
struct TileJob {
    size_t ntiles = 0;
    std::function<hipError_t(size_t t, int slot)> upload;
    std::function<int(size_t t, int in_slot, int out_slot)> compute;
    std::function<hipError_t(size_t t, int slot)> download;
    std::function<void(size_t t, int slot, int j, int n)> copy_out;
};

int run_pipeline(lm_hip_ctx *ctx, BigPipe &bp, const TileJob &job)
{
    struct Shared {
        std::mutex mu;
        std::condition_variable cv;
        size_t uploaded = 0, launched = 0, issued = 0, copied = 0;
        unsigned done[kMaxOutSlots] = {};
        int status = LM_HIP_OK;
        char err[256] = "";
    } sh;
    auto raise = [&](int status, const char *what, hipError_t e) {
        std::lock_guard<std::mutex> lock(sh.mu);
        if (sh.status == LM_HIP_OK) {
            sh.status = status;
            snprintf(sh.err, sizeof sh.err, "%s failed: %s", what, hipGetErrorString(e));
        }
        sh.cv.notify_all();
    };
    const int device = ctx->device, out_slots = pipe_shape().out_slots, ncopiers = pipe_shape().copiers;
    const size_t n = job.ntiles;
#ifdef LM_HIP_DEV_SWITCHES  // where the wall time of a large call goes (LM_HIP_PIPE_TRACE=1)
    std::atomic<long long> ns_up_wait{0}, ns_up_copy{0}, ns_main_wait{0}, ns_cp_wait{0}, ns_cp_copy{0};
    auto tick = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point t0) {
        return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    };
    const auto t_start = tick();
#define LM_PIPE_T0 const auto _t0 = tick()
#define LM_PIPE_ADD(counter) counter += since(_t0)
#else
#define LM_PIPE_T0 (void)0
#define LM_PIPE_ADD(counter) (void)0
#endif
    std::thread uploader;
    std::vector<std::thread> copiers;
    auto abandon = [&](const char *why) {  // a helper thread could not be started: stop the ones that were, report
        {
            std::lock_guard<std::mutex> lock(sh.mu);
            if (sh.status == LM_HIP_OK) {
                sh.status = LM_HIP_ERR_OOM;
                snprintf(sh.err, sizeof sh.err, "host pipeline: %s", why);
            }
            sh.cv.notify_all();
        }
        if (uploader.joinable())
            uploader.join();
        for (std::thread &c : copiers)
            if (c.joinable())
                c.join();
        return fail(LM_HIP_ERR_OOM, "%s", sh.err);
    };
    auto upload_loop = [&] {
        if (hipSetDevice(device) != hipSuccess)
            return raise(LM_HIP_ERR_HIP, "hipSetDevice", hipGetLastError());
        bind_this_thread_to_node(device);
        for (size_t t = 0; t < n; ++t) {
            hipError_t e = hipSuccess;
            {   // input slot t % kInSlots was read by tile t - kInSlots: its kernel must have been launched ...
                LM_PIPE_T0;
                std::unique_lock<std::mutex> lock(sh.mu);
                sh.cv.wait(lock, [&] { return sh.launched + kInSlots > t || sh.status != LM_HIP_OK; });
                if (sh.status != LM_HIP_OK)
                    return;
                lock.unlock();
                e = t >= (size_t)kInSlots ? hipEventSynchronize(bp.kdone[t % kInSlots]) : hipSuccess;  // ... and have finished
                LM_PIPE_ADD(ns_up_wait);
            }
            if (e == hipSuccess) {
                LM_PIPE_T0;
                e = job.upload(t, (int)(t % kInSlots));
                LM_PIPE_ADD(ns_up_copy);
            }
            if (e != hipSuccess)
                return raise(e == hipErrorOutOfMemory ? LM_HIP_ERR_OOM : LM_HIP_ERR_HIP, "tile upload", e);
            std::lock_guard<std::mutex> lock(sh.mu);
            sh.uploaded = t + 1;
            sh.cv.notify_all();
        }
    };
    auto copy_loop = [&](int j) {
        if (hipSetDevice(device) != hipSuccess)
            return raise(LM_HIP_ERR_HIP, "hipSetDevice", hipGetLastError());
        bind_this_thread_to_node(device);
        for (size_t t = 0; t < n; ++t) {
            const int slot = (int)(t % (size_t)out_slots);
            {
                LM_PIPE_T0;
                std::unique_lock<std::mutex> lock(sh.mu);
                sh.cv.wait(lock, [&] { return sh.issued > t || sh.status != LM_HIP_OK; });
                if (sh.status != LM_HIP_OK)
                    return;
                lock.unlock();
                const hipError_t e = hipEventSynchronize(bp.landed[slot]);
                if (e != hipSuccess)
                    return raise(LM_HIP_ERR_HIP, "tile read-back", e);
                LM_PIPE_ADD(ns_cp_wait);
            }
            {
                LM_PIPE_T0;
                job.copy_out(t, slot, j, ncopiers);
                LM_PIPE_ADD(ns_cp_copy);
            }
            std::lock_guard<std::mutex> lock(sh.mu);
            if (++sh.done[slot] == (unsigned)ncopiers) {
                sh.done[slot] = 0;
                sh.copied = t + 1;
                sh.cv.notify_all();
            }
        }
    };
    try {
        uploader = std::thread(upload_loop);
        for (int j = 0; j < ncopiers; ++j)
            copiers.emplace_back(copy_loop, j);
    } catch (const std::exception &ex) {  // std::system_error: no more threads
        return abandon(ex.what());
    }
    for (size_t t = 0; t < n; ++t) {
        {   // tile t is on the device, and output slot t % out_slots (device + pinned) has been emptied into the caller's matrix
            LM_PIPE_T0;
            std::unique_lock<std::mutex> lock(sh.mu);
            sh.cv.wait(lock, [&] { return (sh.uploaded > t && sh.copied + (size_t)out_slots > t) || sh.status != LM_HIP_OK; });
            if (sh.status != LM_HIP_OK)
                break;
            LM_PIPE_ADD(ns_main_wait);
        }
        const int is = (int)(t % kInSlots), os = (int)(t % (size_t)out_slots);
        const int st = job.compute(t, is, os);
        if (st != LM_HIP_OK) {
            std::lock_guard<std::mutex> lock(sh.mu);
            if (sh.status == LM_HIP_OK) {
                sh.status = st;
                snprintf(sh.err, sizeof sh.err, "%s", lm_hip_last_error());
            }
            sh.cv.notify_all();
            break;
        }
        hipError_t e = hipEventRecord(bp.kdone[is], ctx->stream);
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lock(sh.mu);
            sh.launched = t + 1;
            sh.cv.notify_all();
        }
        if (e == hipSuccess)
            e = hipStreamWaitEvent(bp.s_dn, bp.kdone[is], 0);
        if (e == hipSuccess)
            e = job.download(t, os);
        if (e == hipSuccess)
            e = hipEventRecord(bp.landed[os], bp.s_dn);
        if (e != hipSuccess) {
            raise(LM_HIP_ERR_HIP, "tile launch", e);
            break;
        }
        std::lock_guard<std::mutex> lock(sh.mu);
        sh.issued = t + 1;
        sh.cv.notify_all();
    }
    uploader.join();
    for (std::thread &c : copiers)
        c.join();
    // whatever happened, nothing may still be in flight into the ring or out of the tile buffers when this returns
    (void)hipStreamSynchronize(bp.s_up);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(bp.s_dn);
#ifdef LM_HIP_DEV_SWITCHES
    if (getenv("LM_HIP_PIPE_TRACE"))
        fprintf(stderr, "[lm_hip pipe] %zu tiles in %.2f ms: uploader waits %.2f copies %.2f | caller waits %.2f | copiers (each of %d) wait %.2f copy %.2f\n",
                n, since(t_start) * 1e-6, ns_up_wait * 1e-6, ns_up_copy * 1e-6, ns_main_wait * 1e-6, ncopiers,
                ns_cp_wait * 1e-6 / ncopiers, ns_cp_copy * 1e-6 / ncopiers);
#endif
#undef LM_PIPE_T0
#undef LM_PIPE_ADD
    if (sh.status != LM_HIP_OK)
        return fail(sh.status, "%s", sh.err);
    return LM_HIP_OK;
}

// One Score::score_rows_into call on host matrices, f32 (a PSSM) or u8 (a DiscreteMatrix): `launch` enqueues the store
// kernels of `rows` rows whose first sequence row is at `d_seq`, dense output rows (stride = cols) at `d_out`.
struct ScoreCall {
    const uint8_t *seq;  // row `row_begin` of the caller's striped matrix
    size_t seq_stride, cols, nrows, halo;
    char *out;           // the caller's score matrix, row 0 of the range
    size_t out_stride;   // in elements
    size_t elem;         // bytes per score: 4 (f32) or 1 (u8)
    std::function<int(lm_hip_ctx *ctx, const uint8_t *d_seq, size_t rows, void *d_out)> launch;
};

// Rows [r0, r1) of the call on the lane's own buffers: copy up, score, copy down (the copies of pageable memory
// return when they are done), or for the tiniest pieces no copy command at all.
int score_piece(HostLane *lane, const ScoreCall &c, size_t r0, size_t r1)
{
    lm_hip_ctx *ctx = lane->ctx;
    const size_t w = r1 - r0, in_bytes = (w + c.halo) * c.seq_stride, row_bytes = c.cols * c.elem, out_bytes = w * row_bytes;
    const uint8_t *src = c.seq + r0 * c.seq_stride;
    char *dst = c.out + r0 * c.out_stride * c.elem;
    if (lane->zc && in_bytes <= kZeroCopyBytes && out_bytes <= kZeroCopyBytes) {
        uint8_t *zin = lane->zc;
        char *zout = reinterpret_cast<char *>(lane->zc + kZeroCopyBytes);
        memcpy(zin, src, in_bytes);
        LM_TRY(c.launch(ctx, zin, w, zout));
        LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        copy_rows(dst, c.out_stride * c.elem, zout, row_bytes, row_bytes, w);
        lane->piece_dev = (r0 == 0 && r1 == c.nrows) ? zout : nullptr;
        return LM_HIP_OK;
    }
    LM_TRY(lane->d_in.reserve(in_bytes + 64));
    LM_TRY(lane->d_out.reserve(out_bytes));
    uint8_t *d_in = static_cast<uint8_t *>(lane->d_in.ptr);
    char *d_out = static_cast<char *>(lane->d_out.ptr);
    LM_HIP_TRY(hipMemcpyAsync(d_in, src, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    LM_TRY(c.launch(ctx, d_in, w, d_out));
    // only the `cols` scored cells of each row: the caller's alignment padding is left as it was (pli/mod.rs:103)
    LM_HIP_TRY(c.out_stride == c.cols
                   ? hipMemcpyAsync(dst, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream)
                   : hipMemcpy2DAsync(dst, c.out_stride * c.elem, d_out, row_bytes, row_bytes, w, hipMemcpyDeviceToHost,
                                      ctx->stream));
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    lane->piece_dev = (r0 == 0 && r1 == c.nrows) ? d_out : nullptr;
    return LM_HIP_OK;
}

int score_pipelined(HostLane *lane, BigPipe &bp, const ScoreCall &c)
{
    lm_hip_ctx *ctx = lane->ctx;
    LM_TRY(pipe_prepare(bp, ctx->device));
    // rows per tile: a ring slot (32 MB) of scores for genome-sized calls, smaller tiles for smaller calls -- the
    // pipeline's ramp is one tile each of upload, read-back and copy-out latency, a tile costs ~20 us of its own --
    // and at most two slots' worth of symbols
    const size_t tile_bytes = pipe_shape().tile_bytes, row_bytes = c.cols * c.elem;
    const size_t target = std::min(tile_bytes, std::max(kPipeMinTileBytes, c.nrows * row_bytes / 12));
    size_t tr = std::min(target / row_bytes, (2 * tile_bytes) / c.seq_stride);
    tr = std::max<size_t>(tr / 256 * 256, 256);
    const size_t in_tile = ((tr + c.halo) * c.seq_stride + 255) / 256 * 256, out_tile = (tr * row_bytes + 255) / 256 * 256;
    if (tr * row_bytes > tile_bytes)  // (more than 32 K columns: not a shape this path is for)
        return LM_HIP_ERR_CAPACITY;
    LM_TRY(bp.d_in.reserve(kInSlots * in_tile + 64));
    LM_TRY(bp.d_out.reserve((size_t)pipe_shape().out_slots * out_tile));
    uint8_t *d_in = static_cast<uint8_t *>(bp.d_in.ptr);
    char *d_out = static_cast<char *>(bp.d_out.ptr);
    TileJob job;
    job.ntiles = (c.nrows + tr - 1) / tr;
    auto width = [&](size_t t) { return std::min(tr, c.nrows - t * tr); };
    job.upload = [&](size_t t, int slot) {
        hipError_t e = hipMemcpyAsync(d_in + (size_t)slot * in_tile, c.seq + t * tr * c.seq_stride,
                                      (width(t) + c.halo) * c.seq_stride, hipMemcpyHostToDevice, bp.s_up);
        return e == hipSuccess ? hipStreamSynchronize(bp.s_up) : e;
    };
    job.compute = [&](size_t t, int is, int os) {
        return c.launch(ctx, d_in + (size_t)is * in_tile, width(t), d_out + (size_t)os * out_tile);
    };
    job.download = [&](size_t t, int slot) {
        return hipMemcpyAsync(bp.pinned + (size_t)slot * tile_bytes, d_out + (size_t)slot * out_tile, width(t) * row_bytes,
                              hipMemcpyDeviceToHost, bp.s_dn);
    };
    job.copy_out = [&](size_t t, int slot, int j, int n) {
        const size_t w = width(t), a = w * (size_t)j / (size_t)n, b = w * (size_t)(j + 1) / (size_t)n;
        copy_rows(c.out + (t * tr + a) * c.out_stride * c.elem, c.out_stride * c.elem,
                  bp.pinned + (size_t)slot * tile_bytes + a * row_bytes, row_bytes, row_bytes, b - a);
    };
    const int st = run_pipeline(ctx, bp, job);
    trim(bp.d_in);
    return st;
}

// The whole call: the tile pipeline for large score matrices, piece by piece on the lane otherwise.
int score_call(HostLane *lane, const ScoreCall &c)
{
    lm_hip_ctx *ctx = lane->ctx;
    int st = LM_HIP_ERR_CAPACITY;
    lane->piece_dev = nullptr;  // the lane's buffers are about to be overwritten: whatever was kept is gone
    lane->kept_dev = nullptr;
    if (lane->track)
        lane->track->d_data = nullptr;
    if (c.nrows * c.cols * c.elem >= kPipeMinOutBytes) {
        // link-bound: large calls of several threads take turns on the ring (run side by side through the runtime's
        // pageable copies they were 2.2 x slower than one after the other -- profiles/r04_host_pointer.json)
        BigPipe &bp = big_pipe(ctx->device);
        std::lock_guard<std::mutex> pipe(bp.mu);
        st = score_pipelined(lane, bp, c);
        if (st != LM_HIP_ERR_CAPACITY)
            lane_trim(lane);  // (a trim asked for by another thread while this call ran: honoured at its end, like the small calls')
    }
    if (st == LM_HIP_ERR_CAPACITY) {  // small (or a shape the tiles do not fit): piece by piece on this lane
        const size_t piece = std::max<size_t>((64u << 20) / (c.cols * c.elem), 1);
        st = LM_HIP_OK;
        for (size_t r0 = 0; r0 < c.nrows && st == LM_HIP_OK; r0 += piece)
            st = score_piece(lane, c, r0, std::min(c.nrows, r0 + piece));
        if (st != LM_HIP_OK)
            (void)hipStreamSynchronize(ctx->stream);
        lane_trim(lane);
    }
    return st;
}

// 64 bits over the shape and 67 cells spread over the matrix (first, last, 65 in between), by bit pattern.
uint64_t sample_digest(const float *scores, size_t rows, size_t stride, size_t cols)
{
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (rows * 0x100000001b3ull) ^ (cols << 32);
    const size_t cells = rows * cols;
    constexpr size_t kSamples = 67;
    for (size_t i = 0; i < kSamples && cells; ++i) {
        const size_t cell = cells <= kSamples ? i % cells : (size_t)((unsigned __int128)i * (cells - 1) / (kSamples - 1));
        uint32_t bits;
        memcpy(&bits, scores + (cell / cols) * stride + cell % cols, 4);
        h = (h ^ bits) * 0xff51afd7ed558ccdull;
        h ^= h >> 29;
    }
    return h;
}

// After a successful lm_hip_score_f32 whose scores went through the lane's buffers in one piece: remember them.
void keep_scores(HostLane *lane, const float *dev, const float *host, size_t rows, size_t stride, size_t cols)
{
    lane->kept_dev = nullptr;
    if (!g_reuse_scores.load(std::memory_order_relaxed) || !dev)
        return;
    const bool zc = lane->zc && dev == reinterpret_cast<const float *>(lane->zc + kZeroCopyBytes);
    if (!zc && dev != lane->d_out.ptr)  // (the call's own trim handed the buffer back)
        return;
    lane->kept_dev = dev;
    lane->kept_host = host;
    lane->kept_rows = rows;
    lane->kept_cols = cols;
    lane->kept_stride = stride;
    lane->kept_digest = sample_digest(host, rows, stride, cols);
}

// The device copy of `scores` if this lane's previous score call produced exactly this matrix (same pointer and shape,
// same sampled contents) and the buffer is still there; nullptr otherwise.
const float *kept_scores(HostLane *lane, const float *scores, size_t rows, size_t stride, size_t cols)
{
    if (!lane->kept_dev || !g_reuse_scores.load(std::memory_order_relaxed))
        return nullptr;
    const bool zc = lane->zc && lane->kept_dev == reinterpret_cast<const float *>(lane->zc + kZeroCopyBytes);
    if (lane->kept_host != scores || lane->kept_rows != rows || lane->kept_cols != cols || lane->kept_stride != stride)
        return nullptr;  // another matrix: uploaded to the lane's INPUT staging, what is kept stays
    if ((!zc && lane->kept_dev != lane->d_out.ptr) ||  // (trimmed or re-allocated since)
        lane->kept_digest != sample_digest(scores, rows, stride, cols)) {
        lane->kept_dev = nullptr;
        return nullptr;
    }
    ++lane->reuses;
    return lane->kept_dev;
}

// The lane's scores record for `cols` columns (dense rows); false when it cannot be had (the plain store runs then).
bool lane_track(HostLane *lane, size_t cols)
{
    if (!lane->track) {
        lm_hip_scores *t = nullptr;
        if (lm_hip_scores_create(lane->ctx, cols, &t) != LM_HIP_OK)
            return false;
        lane->track = t;
    }
    lane->track->cols = lane->track->stride = cols;
    lane->track->d_data = nullptr;
    lane->track->capacity_rows = 0;  // (nothing of its own to free or to grow)
    lane->track->rows = 0;
    return lane->track->d_best != nullptr;
}

// The caller's score matrix on the device, dense (stride == cols): `*d` points into the lane's staging buffer.
int stage_scores(HostLane *lane, const float *scores, size_t rows, size_t stride, size_t cols, const float **d)
{
    lm_hip_ctx *ctx = lane->ctx;
    LM_TRY(lane->d_in.reserve(rows * cols * sizeof(float)));
    float *dev = static_cast<float *>(lane->d_in.ptr);
    LM_HIP_TRY(stride == cols ? hipMemcpyAsync(dev, scores, rows * cols * sizeof(float), hipMemcpyHostToDevice, ctx->stream)
                              : hipMemcpy2DAsync(dev, cols * sizeof(float), scores, stride * sizeof(float),
                                                 cols * sizeof(float), rows, hipMemcpyHostToDevice, ctx->stream));
    *d = dev;
    return LM_HIP_OK;
}

}  // namespace
}  // namespace lm

using namespace lm;

// No C++ exception may cross the C ABI (std::bad_alloc from a table cache or a job closure, std::system_error from a thread).
template <class Body>
static int guarded(const char *what, Body body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(LM_HIP_ERR_OOM, "%s: out of host memory", what);
    } catch (const std::exception &ex) {
        return fail(LM_HIP_ERR_HIP, "%s: %s", what, ex.what());
    }
}

extern "C" {

int lm_hip_score_f32(const uint8_t *seq, size_t seq_rows_total, size_t seq_stride, size_t cols,
                     size_t wrap, size_t length, const float *pssm, size_t m, size_t pssm_stride,
                     size_t k, size_t row_begin, size_t row_end, float *out, size_t out_stride,
                     size_t *out_rows, size_t *max_index)
{
    return guarded("score", [&]() -> int {
        HostLane *lane = nullptr;
        LM_TRY(acquire_lane(&lane));
        lm_hip_ctx *ctx = lane->ctx;  // the lane is this thread's alone: nothing to lock
        DeviceGuard guard(ctx->device);
        lm_hip_pssm *p = nullptr;
        LM_TRY(lane_pssm(lane, pssm, m, pssm_stride, k, &p));
        LM_TRY(check_score_args(p, seq_rows_total, seq_stride, cols, wrap, row_begin, row_end));
        if (length < m || row_begin >= row_end) {  // pli/mod.rs:85-88
            if (out_rows) *out_rows = 0;
            if (max_index) *max_index = 0;
            return LM_HIP_OK;
        }
        if (!seq || !out || out_stride < cols)
            return fail(LM_HIP_ERR_BAD_ARGS, "score: null buffer or out stride %zu < columns %zu", out_stride, cols);
        // only the rows the range needs travel: [row_begin, row_end + m - 1)
        ScoreCall c{seq + row_begin * seq_stride, seq_stride, cols, row_end - row_begin, m ? m - 1 : 0,
                    reinterpret_cast<char *>(out), out_stride, sizeof(float), nullptr};
        // (lm_hip_host_reuse_scores: a matrix the tracking store kernel covers -- one piece, below 8 M cells -- is scored
        //  through the lane's own scores record, so that the argmax that follows reads the kernel's records)
        const size_t whole = c.nrows;
        lane->tracked = false;
        const bool track = g_reuse_scores.load(std::memory_order_relaxed) && whole * cols < (8u << 20) && lane_track(lane, cols);
        c.launch = [p, seq_stride, cols, lane, track, whole](lm_hip_ctx *cx, const uint8_t *d_seq, size_t rows, void *d_out) {
            ScoreArgs a{p, d_seq, seq_stride, cols, 0, rows, static_cast<float *>(d_out), cols};
            if (!track || rows != whole)
                return launch_score_store(cx, a);
            lm_hip_scores *t = lane->track;
            t->d_data = static_cast<float *>(d_out);
            t->rows = rows;
            const int st = score_store_tracked(cx, a, t);
            lane->tracked = st == LM_HIP_OK;
            return st;
        };
        LM_TRY(score_call(lane, c));
        keep_scores(lane, static_cast<const float *>(lane->piece_dev), out, c.nrows, out_stride, cols);
        if (!lane->kept_dev || (lane->track && lane->track->d_data != lane->kept_dev))
            lane->tracked = false;
        if (out_rows) *out_rows = c.nrows;       // pli/mod.rs:91
        if (max_index) *max_index = length + 1 - m;
        return LM_HIP_OK;
    });
}

// Score<u8, A, C>::score_rows_into with a DiscreteMatrix on host matrices (pli/mod.rs:437-476 is the AVX2 impl; what
// Scanner::next calls per 256-row block, scan.rs:174-178).  `saturate`: avx2.rs:336 (adds_epu8) or Generic's wrapping `+=`.
int lm_hip_score_u8_host(const uint8_t *seq, size_t seq_rows_total, size_t seq_stride, size_t cols, size_t wrap,
                         size_t length, const uint8_t *weights, size_t m, size_t weights_stride, size_t k,
                         size_t row_begin, size_t row_end, int saturate, uint8_t *out, size_t out_stride,
                         size_t *out_rows, size_t *max_index)
{
    return guarded("score_u8", [&]() -> int {
        if (!weights || m == 0 || k == 0 || k > 256 || weights_stride < k)
            return fail(LM_HIP_ERR_BAD_ARGS, "score_u8: bad discrete matrix (%zu x %zu, stride %zu)", m, k, weights_stride);
        lm_hip_pssm shape;  // the geometry checks only look at the motif length
        shape.m = m;
        shape.k = k;
        LM_TRY(check_score_args(&shape, seq_rows_total, seq_stride, cols, wrap, row_begin, row_end));
        if (length < m || row_begin >= row_end) {  // pli/mod.rs:85-88
            if (out_rows) *out_rows = 0;
            if (max_index) *max_index = 0;
            return LM_HIP_OK;
        }
        if (!seq || !out || out_stride < cols)
            return fail(LM_HIP_ERR_BAD_ARGS, "score_u8: null buffer or out stride %zu < columns %zu", out_stride, cols);
        HostLane *lane = nullptr;
        LM_TRY(acquire_lane(&lane));
        DeviceGuard guard(lane->ctx->device);
        ScoreCall c{seq + row_begin * seq_stride, seq_stride, cols, row_end - row_begin, m - 1, reinterpret_cast<char *>(out),
                    out_stride, 1, nullptr};
        c.launch = [=](lm_hip_ctx *cx, const uint8_t *d_seq, size_t rows, void *d_out) {
            // (the device tables of the matrix are kept by the lane's context until the weights change: launch_score_u8)
            DiscreteArgs a{weights, m, weights_stride, k, d_seq, seq_stride, cols, 0, rows, static_cast<uint8_t *>(d_out), cols,
                           saturate != 0};
            return launch_score_u8(cx, a);
        };
        LM_TRY(score_call(lane, c));
        if (out_rows) *out_rows = c.nrows;
        if (max_index) *max_index = length + 1 - m;
        return LM_HIP_OK;
    });
}

// What the host-pointer path keeps between calls -- the pinned ring (128 MB), its device tiles, the staging buffers of
// lanes whose threads have exited and of the calling thread's own lane -- handed back; the next call sets them up again.
// Contexts, streams and cached PSSM tables stay.  Waits for a large call in flight on another thread.
int lm_hip_host_trim(void)
{
    return guarded("host_trim", [&]() -> int {
        std::vector<BigPipe *> all;
        {
            std::lock_guard<std::mutex> lock(pipes_mu());
            all = pipes();
        }
        for (BigPipe *pb : all) {
            if (!pb)
                continue;
            BigPipe &bp = *pb;
            std::lock_guard<std::mutex> pipe(bp.mu);
            if (bp.device >= 0) {
                DeviceGuard guard(bp.device);
                (void)hipStreamSynchronize(bp.s_up);
                (void)hipStreamSynchronize(bp.s_dn);
                bp.d_in.release();
                bp.d_out.release();
                for (hipEvent_t e : bp.kdone)
                    (void)hipEventDestroy(e);
                for (hipEvent_t e : bp.landed)
                    (void)hipEventDestroy(e);
                (void)hipStreamDestroy(bp.s_up);
                (void)hipStreamDestroy(bp.s_dn);
                (void)hipHostFree(bp.pinned);
                bp.s_up = bp.s_dn = nullptr;
                bp.pinned = nullptr;
                bp.device = -1;
            }
        }
        std::vector<HostLane *> lanes;
        {
            std::lock_guard<std::mutex> lock(lanes_mu());
            lanes = idle_lanes();  // idle: no thread owns them; they stay on the list
            if (t_lane.lane)
                lanes.push_back(t_lane.lane);
            for (HostLane *lane : lanes) {
                DeviceGuard guard(lane->ctx->device);
                (void)hipStreamSynchronize(lane->ctx->stream);
                lane->d_in.release();
                lane->d_out.release();
                // the lane context's own scratch: reduction partials / hit lists, the chunk of a sliced u8 store
                lane->ctx->scratch.release();
                lane->ctx->scratch2.release();
                lane->ctx->chunk_scores.release();
                lane->ctx->scan_buf.release();
            }
        }
        // lanes of threads that are still alive cannot be touched from here (their streams may be busy): they trim
        // themselves at the end of their next call
        g_trim_epoch.fetch_add(1, std::memory_order_release);
        return LM_HIP_OK;
    });
}

int lm_hip_host_bind_thread(int device)
{
    return guarded("host_bind_thread", [&]() -> int {
        if (device >= 0) {
            const std::vector<int> &devs = usable_devices();
            if (std::find(devs.begin(), devs.end(), device) == devs.end())
                return fail(LM_HIP_ERR_NO_DEVICE, "host_bind_thread: %d is not a usable (gfx950) device ordinal", device);
        }
        t_lane.want_device = device < 0 ? -1 : device;  // takes effect at the thread's next host-pointer call
        return LM_HIP_OK;
    });
}

int lm_hip_host_spread_lanes(int enabled)
{
    return guarded("host_spread_lanes", [&]() -> int {
        g_spread.store(enabled != 0, std::memory_order_relaxed);  // lanes that exist stay where they are
        return LM_HIP_OK;
    });
}

int lm_hip_host_reuse_scores(int enabled)
{
    return guarded("host_reuse_scores", [&]() -> int {
        g_reuse_scores.store(enabled != 0, std::memory_order_relaxed);
        return LM_HIP_OK;
    });
}

int lm_hip_host_reuse_count(size_t *count)
{
    return guarded("host_reuse_count", [&]() -> int {
        if (!count)
            return fail(LM_HIP_ERR_BAD_ARGS, "host_reuse_count: null argument");
        HostLane *lane = nullptr;
        LM_TRY(acquire_lane(&lane));
        *count = lane->reuses;
        return LM_HIP_OK;
    });
}

int lm_hip_host_lane_info(int *device, int *numa_node, int *helper_cpus)
{
    return guarded("host_lane_info", [&]() -> int {
        if (!device && !numa_node && !helper_cpus)
            return fail(LM_HIP_ERR_BAD_ARGS, "host_lane_info: no output asked for");
        HostLane *lane = nullptr;
        LM_TRY(acquire_lane(&lane));
        const NodeCpus &n = device_node(lane->device);
        if (device) *device = lane->device;
        if (numa_node) *numa_node = n.node;
        if (helper_cpus) *helper_cpus = n.valid ? CPU_COUNT(&n.set) : 0;
        return LM_HIP_OK;
    });
}

int lm_hip_argmax_f32(const float *scores, size_t rows, size_t stride, size_t cols, int *found,
                      lm_hip_coords *best, float *value)
{
    return guarded("argmax", [&]() -> int {
        if (!found)
            return fail(LM_HIP_ERR_BAD_ARGS, "argmax: null argument");
        *found = 0;
        if (rows == 0)  // pli/mod.rs:136-138
            return LM_HIP_OK;
        if (!scores || cols == 0 || stride < cols)
            return fail(LM_HIP_ERR_BAD_ARGS, "argmax: bad matrix");
        HostLane *lane = nullptr;
        LM_TRY(acquire_lane(&lane));
        lm_hip_ctx *ctx = lane->ctx;  // the lane is this thread's alone: nothing to lock
        DeviceGuard guard(ctx->device);
        ScratchTrim scratch_trim(ctx);
        const float *d = kept_scores(lane, scores, rows, stride, cols);
        if (d && lane->tracked && lane->track && lane->track->d_data == d && lane->track->rows == rows) {
            // the store kernel that wrote the kept matrix tracked its best cell: fold / read its records (handles.hip)
            lm_hip_coords cell{0, 0};
            float v = 0.0f;
            const int st = lm_hip_argmax(ctx, lane->track, found, &cell, &v);
            lane_trim(lane);
            if (st != LM_HIP_OK)
                return st;
            if (*found) {
                if (best) *best = cell;
                if (value) *value = v;
            }
            return LM_HIP_OK;
        }
        ArgmaxRecord rec{};
        int st = d ? (int)LM_HIP_OK : stage_scores(lane, scores, rows, stride, cols, &d);
        if (st == LM_HIP_OK)
            st = launch_argmax(ctx, d, rows, cols, cols, 1, &rec);
        if (st != LM_HIP_OK)
            (void)hipStreamSynchronize(ctx->stream);
        lane_trim(lane);
        if (st != LM_HIP_OK)
            return st;
        *found = rec.found;
        if (rec.found) {
            if (best) {
                best->row = (size_t)(rec.index / (long long)cols);
                best->col = (size_t)(rec.index % (long long)cols);
            }
            if (value)
                *value = rec.value;
        }
        return LM_HIP_OK;
    });
}

int lm_hip_max_f32(const float *scores, size_t rows, size_t stride, size_t cols, int *found, float *value)
{
    lm_hip_coords c{0, 0};
    return lm_hip_argmax_f32(scores, rows, stride, cols, found, &c, value);
}

int lm_hip_threshold_f32(const float *scores, size_t rows, size_t stride, size_t cols, float t,
                         lm_hip_coords **coords, size_t *n)
{
    return guarded("threshold", [&]() -> int {
        if (!coords || !n)
            return fail(LM_HIP_ERR_BAD_ARGS, "threshold: null argument");
        *coords = nullptr;
        *n = 0;
        if (rows == 0)
            return LM_HIP_OK;
        if (!scores || cols == 0 || stride < cols)
            return fail(LM_HIP_ERR_BAD_ARGS, "threshold: bad matrix");
        HostLane *lane = nullptr;
        LM_TRY(acquire_lane(&lane));
        lm_hip_ctx *ctx = lane->ctx;  // the lane is this thread's alone: nothing to lock
        DeviceGuard guard(ctx->device);
        // a dense list grows ctx->scratch / scratch2 to 16 B per hit: handed back when the call ends (score_api.hip does the same)
        ScratchTrim scratch_trim(ctx);
        const float *d = kept_scores(lane, scores, rows, stride, cols);
        int st = d ? (int)LM_HIP_OK : stage_scores(lane, scores, rows, stride, cols, &d);
        if (st == LM_HIP_OK)
            st = launch_threshold(ctx, d, rows, cols, cols, t, coords, n);
        if (st != LM_HIP_OK)
            (void)hipStreamSynchronize(ctx->stream);
        lane_trim(lane);
        return st;
    });
}

// ---- Scanner on host matrices (scan.rs:166-249) ------------------------------------------------------------------------

// The caller's StripedSequence on the lane's staging buffer, as a (borrowed) resident sequence: 1 B per position up.
static int stage_sequence(HostLane *lane, const uint8_t *seq, size_t seq_rows_total, size_t seq_stride, size_t cols, size_t wrap,
                          size_t length, size_t k, lm_hip_seq *out)
{
    if (!seq || cols == 0 || seq_stride < cols || wrap > seq_rows_total)
        return fail(LM_HIP_ERR_BAD_ARGS, "scan: bad sequence matrix (%zu rows, stride %zu, %zu columns, wrap %zu)",
                    seq_rows_total, seq_stride, cols, wrap);
    lm_hip_ctx *ctx = lane->ctx;
    const size_t bytes = seq_rows_total * seq_stride;
    LM_TRY(lane->d_in.reserve(bytes + 64));
    LM_HIP_TRY(hipMemcpyAsync(lane->d_in.ptr, seq, bytes, hipMemcpyHostToDevice, ctx->stream));
    out->device = ctx->device;
    out->d_data = static_cast<uint8_t *>(lane->d_in.ptr);
    out->capacity_rows = seq_rows_total;
    out->rows = seq_rows_total - wrap;
    out->wrap = wrap;
    out->stride = seq_stride;
    out->cols = cols;
    out->length = length;
    out->k = k;
    out->owns = false;
    return LM_HIP_OK;
}

int lm_hip_scan_f32_host(const uint8_t *seq, size_t seq_rows_total, size_t seq_stride, size_t cols, size_t wrap, size_t length,
                         const float *pssm, size_t m, size_t pssm_stride, size_t k, float threshold, lm_hip_hit **hits,
                         size_t *n)
{
    return guarded("scan", [&]() -> int {
        if (!hits || !n)
            return fail(LM_HIP_ERR_BAD_ARGS, "scan: null argument");
        *hits = nullptr;
        *n = 0;
        HostLane *lane = nullptr;
        LM_TRY(acquire_lane(&lane));
        lm_hip_ctx *ctx = lane->ctx;
        DeviceGuard guard(ctx->device);
        lm_hip_pssm *p = nullptr;
        LM_TRY(lane_pssm(lane, pssm, m, pssm_stride, k, &p));
        lm_hip_seq view;
        int st = stage_sequence(lane, seq, seq_rows_total, seq_stride, cols, wrap, length, k, &view);
        if (st == LM_HIP_OK)
            st = lm_hip_scan_f32(ctx, p, &view, threshold, hits, n);  // synchronises
        if (st != LM_HIP_OK)
            (void)hipStreamSynchronize(ctx->stream);
        lane_trim(lane);
        return st;
    });
}

int lm_hip_scan_max_f32_host(const uint8_t *seq, size_t seq_rows_total, size_t seq_stride, size_t cols, size_t wrap,
                             size_t length, const float *pssm, size_t m, size_t pssm_stride, size_t k,
                             const uint8_t *dweights, size_t dweights_stride, int saturate, unsigned level, int have,
                             size_t position, float score, size_t first_row, int *found, lm_hip_hit *best)
{
    return guarded("scan_max", [&]() -> int {
        if (!found || !best || !dweights)
            return fail(LM_HIP_ERR_BAD_ARGS, "scan_max: null argument");
        *found = 0;
        HostLane *lane = nullptr;
        LM_TRY(acquire_lane(&lane));
        lm_hip_ctx *ctx = lane->ctx;
        DeviceGuard guard(ctx->device);
        lm_hip_pssm *p = nullptr;
        LM_TRY(lane_pssm(lane, pssm, m, pssm_stride, k, &p));
        lm_hip_seq view;
        int st = stage_sequence(lane, seq, seq_rows_total, seq_stride, cols, wrap, length, k, &view);
        if (st == LM_HIP_OK)
            st = lm_hip_scan_max_f32(ctx, p, &view, dweights, dweights_stride, saturate, level, have, position, score, first_row,
                                     found, best);
        if (st != LM_HIP_OK)
            (void)hipStreamSynchronize(ctx->stream);
        lane_trim(lane);
        return st;
    });
}

// ---- the Dispatch::Hip policy: where a call on host matrices should go ------------------------------

// Cost model behind lm_hip_host_crossover, constants measured by tools/crossover.py (profiles/r05_crossover.json: MI355X on
// PCIe Gen5, 2 x EPYC 9575F; one CPU thread, as the reference gives a call):
//   GPU call  ~ t0 + g * cells     t0 = launch + synchronisation + copy set-up of the copy path (calls above 128 KB;
//                                  the zero-copy path below is cheaper but so is the CPU there), g = link bytes per cell
//   CPU tier  ~ c * cells          Score: c = c_row * M (AVX2 f32 permute / u8 shuffle kernels), reductions: one pass
// crossover = t0 / (c - g), never when the CPU's per-cell cost does not exceed the link's by a margin.
namespace {
struct HostCost {
    double t0_us, gpu_ns, cpu_ns, cpu_ns_per_row;
};
//                                      t0 us  GPU ns/cell  CPU ns/cell  CPU ns/(cell x motif row)
constexpr HostCost kCostScoreF32{47.0, 0.093, 0.0, 0.0316};      // 1 B up + 4 B down; AVX2 f32 permute kernel (M = 4 ... 30 measured)
constexpr HostCost kCostScoreU8{60.0, 0.036, 0.0, 0.0071};       // 1 B up + 1 B down; AVX2 u8 shuffle kernel
constexpr HostCost kCostMaximumF32{37.0, 0.070, 0.247, 0.0};     // 4 B up; the Generic scan (the rule the variant keeps, pli/mod.rs:135-160)
constexpr HostCost kCostThresholdF32{63.0, 0.073, 0.252, 0.0};   // 4 B up; the default body (pli/mod.rs:210-221)
// Scanner: t0 and g measured (tests/cpp/test_dispatch --bench -> profiles/r05_scan_crossover.json: 29 us at 10 kbp, 41 at
// 100 kbp, 78 at 1 Mbp with the short form of the hit-list ordering; 46 / 49 / 85 before it); the CPU side is the reference's block loop on its AVX2 tier, PRICED from the u8 shuffle kernel + one pass of the
// vectorised u8 reductions -- the C++ port that stands in for it in the tests has scalar reductions (0.54 ns per cell: it
// crosses over at ~85 k cells); a Rust build should re-measure with its own tier (lm_hip_host_set_cpu_cost).
constexpr HostCost kCostScan{38.0, 0.044, 0.03, 0.0071};
constexpr double kCrossoverMargin = 1.25;  // the CPU must cost this much more per cell than the link before a call leaves it
constexpr int kOps = 9;
constexpr size_t kModel = (size_t)-2;      // lm_hip_host_set_crossover: "no pin, ask the model"

// The model in force: the compiled constants (one EPYC 9575F + PCIe Gen5 box) until lm_hip_host_calibrate measures the GPU side
// on THIS host and link, and lm_hip_host_set_cpu_cost the CPU tier's side; per-op pins on top (lm_hip_host_set_crossover).
struct CostTable {
    std::mutex mu;
    HostCost cost[kOps] = {{}, kCostScoreF32, kCostScoreU8, {}, kCostMaximumF32, {}, kCostThresholdF32, {}, kCostScan};
    bool gpu_measured[kOps] = {};
    bool cpu_set[kOps] = {};
    size_t pin[kOps] = {kModel, kModel, kModel, kModel, kModel, kModel, kModel, kModel, kModel};
    bool calibrated = false;
};
CostTable &cost_table()
{
    static CostTable *t = new CostTable();
    return *t;
}
bool leaves_cpu_tier(int op)
{
    return op == LM_HIP_OP_SCORE_F32 || op == LM_HIP_OP_SCORE_U8 || op == LM_HIP_OP_MAXIMUM_F32 || op == LM_HIP_OP_THRESHOLD_F32 ||
           op == LM_HIP_OP_SCAN;
}
size_t crossover_cells(const HostCost &k, size_t m)
{
    const double c = k.cpu_ns + k.cpu_ns_per_row * (double)m;
    if (c <= k.gpu_ns * kCrossoverMargin)
        return SIZE_MAX;
    const double cells = k.t0_us * 1e3 / (c - k.gpu_ns);
    return cells >= 9e18 ? SIZE_MAX : (size_t)cells + 1;
}

// median wall time (us) of `reps` calls of f
double median_us(const std::function<void()> &f, int reps)
{
    std::vector<double> t;
    for (int i = 0; i < reps; ++i) {
        const auto a = std::chrono::steady_clock::now();
        f();
        t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
}  // namespace

int lm_hip_host_crossover(int op, size_t m, size_t k, size_t *cells)
{
    (void)k;
    if (!cells)
        return fail(LM_HIP_ERR_BAD_ARGS, "host_crossover: null output");
    if (op < 0 || op >= kOps)
        return fail(LM_HIP_ERR_BAD_ARGS, "host_crossover: unknown operation %d", op);
    CostTable &t = cost_table();
    std::lock_guard<std::mutex> lock(t.mu);
    if (t.pin[op] != kModel) {
        *cells = t.pin[op];
        return LM_HIP_OK;
    }
    // Encode / Stripe: one pass over bytes a core streams faster than the link carries them (1 B up + 1 B down); resident
    // sequences are made by lm_hip_seq_from_ascii / _encoded.  Maximum / Threshold on u8: Scanner-internal (scan.rs:181-184),
    // specialised away by LM_HIP_OP_SCAN.  The others: the model (Score: 1 B up + 4 / 1 B down per cell against ~0.031 / 0.007
    // ns per cell and motif row of the AVX2 kernels; reductions: 4 B up against one pass; Scanner: 1 B up + the scan against
    // the reference's block loop).
    *cells = leaves_cpu_tier(op) ? crossover_cells(t.cost[op], m) : SIZE_MAX;
    return LM_HIP_OK;
}

int lm_hip_host_cost_model(int op, double *t0_us, double *gpu_ns_per_cell, double *cpu_ns_per_cell, double *cpu_ns_per_cell_row,
                           int *gpu_side_measured)
{
    if (op < 0 || op >= kOps)
        return fail(LM_HIP_ERR_BAD_ARGS, "host_cost_model: unknown operation %d", op);
    if (!t0_us && !gpu_ns_per_cell && !cpu_ns_per_cell && !cpu_ns_per_cell_row && !gpu_side_measured)
        return fail(LM_HIP_ERR_BAD_ARGS, "host_cost_model: no output asked for");
    CostTable &t = cost_table();
    std::lock_guard<std::mutex> lock(t.mu);
    if (t0_us) *t0_us = t.cost[op].t0_us;
    if (gpu_ns_per_cell) *gpu_ns_per_cell = t.cost[op].gpu_ns;
    if (cpu_ns_per_cell) *cpu_ns_per_cell = t.cost[op].cpu_ns;
    if (cpu_ns_per_cell_row) *cpu_ns_per_cell_row = t.cost[op].cpu_ns_per_row;
    if (gpu_side_measured) *gpu_side_measured = t.gpu_measured[op];
    return LM_HIP_OK;
}

int lm_hip_host_set_cpu_cost(int op, double ns_per_cell, double ns_per_cell_row)
{
    if (op < 0 || op >= kOps || !leaves_cpu_tier(op))
        return fail(LM_HIP_ERR_BAD_ARGS, "host_set_cpu_cost: operation %d has no cost model (it stays on the CPU tier)", op);
    if (!(ns_per_cell >= 0) || !(ns_per_cell_row >= 0) || ns_per_cell > 1e6 || ns_per_cell_row > 1e6)
        return fail(LM_HIP_ERR_BAD_ARGS, "host_set_cpu_cost: costs must be finite and >= 0");
    CostTable &t = cost_table();
    std::lock_guard<std::mutex> lock(t.mu);
    t.cost[op].cpu_ns = ns_per_cell;
    t.cost[op].cpu_ns_per_row = ns_per_cell_row;
    t.cpu_set[op] = true;
    return LM_HIP_OK;
}

int lm_hip_host_set_crossover(int op, size_t cells)
{
    if (op < 0 || op >= kOps)
        return fail(LM_HIP_ERR_BAD_ARGS, "host_set_crossover: unknown operation %d", op);
    CostTable &t = cost_table();
    std::lock_guard<std::mutex> lock(t.mu);
    t.pin[op] = cells;  // (size_t)-2 = back to the model
    return LM_HIP_OK;
}

int lm_hip_host_calibrate(double budget_ms, int force)
{
    return guarded("host_calibrate", [&]() -> int {
        if (!(budget_ms > 0))
            return fail(LM_HIP_ERR_BAD_ARGS, "host_calibrate: the budget must be positive");
        CostTable &t = cost_table();
        {
            std::lock_guard<std::mutex> lock(t.mu);
            if (t.calibrated && !force)
                return LM_HIP_OK;
        }
        // Synthetic host matrices in the reference's layouts (dense.rs:43-48): a striped DNA sequence of `big` rows (+ wrap),
        // a length-16 PSSM, a DiscreteMatrix, a score matrix.  Two sizes per site: a small call is ~t0, the difference to a
        // large one the per-cell cost.  Every site gets a fifth of the budget.
        const size_t m = 16, wrap = m - 1, small_rows = 256, big_rows = (size_t)1 << 17;  // 8 192 and 4 Mi cells
        std::vector<uint8_t> seq((big_rows + wrap) * 32);
        uint64_t x = 0x5EED0006u;
        for (uint8_t &b : seq) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            b = (uint8_t)((x >> 33) & 3u);
        }
        std::vector<float> pssm(m * 8, 0.0f), scores(big_rows * 32);
        std::vector<uint8_t> dw(m * 32, 0), u8out(big_rows * 32);
        for (size_t j = 0; j < m; ++j)
            for (size_t sy = 0; sy < 4; ++sy) {
                pssm[j * 8 + sy] = (float)((int)((j * 7 + sy * 3) % 9) - 4) * 0.37f;
                dw[j * 32 + sy] = (uint8_t)((j * 5 + sy * 3) % 12);
            }
        for (size_t j = 0; j < m; ++j)
            pssm[j * 8 + 4] = -INFINITY;
        const auto deadline_per_site = budget_ms * 1e3 / 5.0;  // us
        struct Probe {
            int op;
            std::function<int(size_t rows)> call;
        };
        size_t orow = 0, mi = 0;
        int found = 0;
        lm_hip_coords best;
        float val = 0;
        const float thr = 1e30f;  // nothing passes: the threshold call's list stays empty (its cost is the pass)
        std::vector<Probe> probes;
        probes.push_back({LM_HIP_OP_SCORE_F32, [&](size_t rows) {
                              return lm_hip_score_f32(seq.data(), rows + wrap, 32, 32, wrap, rows * 32, pssm.data(), m, 8, 5, 0, rows, scores.data(), 32,
                                                      &orow, &mi);
                          }});
        probes.push_back({LM_HIP_OP_SCORE_U8, [&](size_t rows) {
                              return lm_hip_score_u8_host(seq.data(), rows + wrap, 32, 32, wrap, rows * 32, dw.data(), m, 32, 5, 0, rows, 1,
                                                          u8out.data(), 32, &orow, &mi);
                          }});
        probes.push_back({LM_HIP_OP_MAXIMUM_F32, [&](size_t rows) { return lm_hip_argmax_f32(scores.data(), rows, 32, 32, &found, &best, &val); }});
        probes.push_back({LM_HIP_OP_THRESHOLD_F32, [&](size_t rows) {
                              lm_hip_coords *c = nullptr;
                              size_t n = 0;
                              const int st = lm_hip_threshold_f32(scores.data(), rows, 32, 32, thr, &c, &n);
                              lm_hip_free(c);
                              return st;
                          }});
        probes.push_back({LM_HIP_OP_SCAN, [&](size_t rows) {
                              lm_hip_hit *h = nullptr;
                              size_t n = 0;
                              const int st = lm_hip_scan_f32_host(seq.data(), rows + wrap, 32, 32, wrap, rows * 32, pssm.data(), m, 8, 5, thr, &h, &n);
                              lm_hip_free(h);
                              return st;
                          }});
        HostCost measured[kOps] = {};
        bool ok[kOps] = {};
        for (const Probe &p : probes) {
            int st = p.call(small_rows);  // first call: lane, PSSM cache, staging
            if (st == LM_HIP_OK)
                st = p.call(big_rows);
            if (st != LM_HIP_OK)
                return st;
            const auto t_begin = std::chrono::steady_clock::now();
            auto spent = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(); };
            const double t_small = median_us([&] { (void)p.call(small_rows); }, 15);
            const double one_big = median_us([&] { (void)p.call(big_rows); }, 1);
            int reps = (int)std::max(1.0, std::min(9.0, (deadline_per_site - spent()) / std::max(one_big, 1.0)));
            const double t_big = std::min(one_big, median_us([&] { (void)p.call(big_rows); }, reps));
            const double cells_small = (double)small_rows * 32, cells_big = (double)big_rows * 32;
            double g = (t_big - t_small) * 1e3 / (cells_big - cells_small);  // ns per cell
            if (!(g > 0))
                g = 0.001;
            measured[p.op].gpu_ns = g;
            measured[p.op].t0_us = std::max(t_small - g * cells_small * 1e-3, 1.0);
            ok[p.op] = true;
        }
        std::lock_guard<std::mutex> lock(t.mu);
        for (int op = 0; op < kOps; ++op)
            if (ok[op]) {
                t.cost[op].t0_us = measured[op].t0_us;
                t.cost[op].gpu_ns = measured[op].gpu_ns;
                t.gpu_measured[op] = true;
            }
        t.calibrated = true;
        return LM_HIP_OK;
    });
}

}  // extern "C"
