// hits.hip -- puts the hit list of the fused threshold kernels in key order ON THE DEVICE.
//
// The fused kernels append (key, score) records with atomics, i.e. in no particular
// order, while the reference's Threshold pushes cells in row-major order
// (pli/mod.rs:212-218) and its Scanner yields positions in sequence order.  Keys are
// unique (every cell is reported once) and live in a bounded universe --
// key = (job << 40) | low, low < cells of the job -- so ordering is a bucket sort:
//
//   1. hits_bucket_count    bucket = job * nb + (low >> shift); histogram with atomics
//   2. exclusive scan of the histogram (launch_scan_u32, reduce.hip)
//   3. hits_bucket_scatter  records grouped by bucket (any order inside a bucket)
//   4. hits_rank_emit       rank of a record inside its bucket = number of smaller
//                           keys there; the final slot is bucket start + rank, written
//                           directly in the caller's output format
//
// Short lists of ONE job skip steps 1 and 2 as launches: the re-scoring kernel counts while it stores, every workgroup
// of the scatter scans the histogram itself (ShortOrder, below).
//
// Long lists (the JASPAR batch leaves 2.6 M hits, a non-i.i.d. genome 4.6 M) take a different road.  The histogram and the
// scatter are one device-scope atomic per record on buckets whose geometry assumes an even hit density: fine on uniform input
// (0.15 + 0.19 + 0.07 ms for 2.6 M records), but where hits cluster -- low-complexity tracts, short motifs -- or the expected
// count is off (a context's first call: 2.2 + 2.4 + 0.9 ms) the atomics pile up on few addresses.  From kSortFrom records on
// the keys are radix-sorted instead (rocPRIM's device radix sort, LDS-privatised digit histograms: 0.32 ms for 2.6 M (key, value)
// pairs on 52 bits, tools/kbench/radix_sort_bench.hip; insensitive to the distribution): hits_split (records -> key / value
// arrays, unused slots = an all-ones sentinel that sorts last), the sort, hits_sorted_emit (caller's output format),
// hits_sorted_starts (lower bound of every job's first key).  JASPAR batch: 17.6 vs 17.7 ms uniform, 21.5 vs 23.2 ms on the
// non-i.i.d. sequence (tools/sort_ab.py).
//
// `shift` is chosen from the hit density so that a bucket holds ~1 record on
// average; a bucket never holds more records than it has cells (2^shift), which bounds
// step 4 at 8 x (cells of the batch) comparisons whatever the distribution of hits.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "score_kernels.hpp"

#include <rocprim/device/device_radix_sort.hpp>

namespace lm {

namespace {

constexpr unsigned long long kLowMask = (1ull << 40) - 1;

__device__ __forceinline__ unsigned long long bucket_of(unsigned long long key, int shift,
                                                        unsigned long long nb)
{
    return (key >> 40) * nb + ((key & kLowMask) >> shift);
}

__device__ __forceinline__ unsigned long long bucket_start(
    unsigned long long b, unsigned long long nbuckets, unsigned long long count,
    const unsigned long long *__restrict__ offsets, const unsigned long long *__restrict__ tiles)
{
    return b < nbuckets ? tiles[b / kScanTile] + offsets[b] : count;
}

// Every kernel reads the record count from the device counter the scans bumped (clamped to
// the list capacity), so the ordering can be enqueued BEFORE the host knows the count.
__device__ __forceinline__ unsigned long long live_count(const unsigned long long *count_ptr,
                                                         const unsigned long long cap)
{
    const unsigned long long n = *count_ptr;
    return n < cap ? n : cap;
}

__global__ __launch_bounds__(kBlock) void hits_bucket_count(const HitRecord *__restrict__ hits,
                                                            const unsigned long long *__restrict__ count_ptr,
                                                            const unsigned long long cap,
                                                            const int shift,
                                                            const unsigned long long nb,
                                                            unsigned *__restrict__ counts)
{
    const unsigned long long count = live_count(count_ptr, cap);
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * kBlock)
        atomicAdd(&counts[bucket_of(hits[i].key, shift, nb)], 1u);
}

__global__ __launch_bounds__(kBlock) void hits_bucket_scatter(
    const HitRecord *__restrict__ hits, const unsigned long long *__restrict__ count_ptr,
    const unsigned long long cap, const int shift,
    const unsigned long long nb, const unsigned long long *__restrict__ offsets,
    const unsigned long long *__restrict__ tiles, unsigned *__restrict__ counts,
    HitRecord *__restrict__ grouped)
{
    const unsigned long long count = live_count(count_ptr, cap);
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * kBlock) {
        const HitRecord r = hits[i];
        const unsigned long long b = bucket_of(r.key, shift, nb);
        const unsigned left = atomicSub(&counts[b], 1u);  // counts down to 0
        grouped[tiles[b / kScanTile] + offsets[b] + left - 1] = r;
    }
}

// EMIT 0: lm_hip_coords {low / cols, low % cols} + score (Threshold, pli/mod.rs:215)
// EMIT 1: lm_hip_hit {low, score} (Scanner: low is the sequence position)
template <int EMIT>
__global__ __launch_bounds__(kBlock) void hits_rank_emit(
    const HitRecord *__restrict__ grouped, const unsigned long long *__restrict__ count_ptr,
    const unsigned long long cap, const int shift,
    const unsigned long long nb, const unsigned long long nbuckets,
    const unsigned long long *__restrict__ offsets, const unsigned long long *__restrict__ tiles,
    const unsigned long long cols, lm_hip_coords *__restrict__ coords, float *__restrict__ values,
    lm_hip_hit *__restrict__ out_hits, const unsigned long long max_bucket, unsigned *__restrict__ abort_flag,
    void *__restrict__ pre_out, float *__restrict__ pre_values, const unsigned long long pre,
    const unsigned long long njobs, unsigned long long *__restrict__ starts, unsigned long long *__restrict__ header,
    unsigned *__restrict__ clean_counts, unsigned *__restrict__ clean_cursors, unsigned long long *__restrict__ clean_counters,
    unsigned *__restrict__ done_ticket, unsigned *__restrict__ done_flag, const unsigned generation)
{
    const unsigned long long count = live_count(count_ptr, cap);
    if (clean_counts) {  // ShortOrder: nothing reads the histogram, the cursors or the list's own counters any more (`count_ptr`
                         // is their copy); the next call finds them zero
        for (unsigned long long b = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; b < nbuckets;
             b += (unsigned long long)gridDim.x * kBlock) {
            clean_counts[b] = 0u;
            clean_cursors[b] = 0u;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            clean_counters[0] = 0ull;
            clean_counters[1] = 0ull;
        }
    }
    if (starts && blockIdx.x == 0) {  // few jobs: the job offsets and the raw counters ride along (no launch of their own)
        if (threadIdx.x == 0 && header) {
            header[0] = count_ptr[0];
            header[1] = count_ptr[1];
        }
        for (unsigned long long j = threadIdx.x; j <= njobs; j += kBlock)
            starts[j] = bucket_start(j * nb, nbuckets, count, offsets, tiles);
    }
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * kBlock) {
        const HitRecord r = grouped[i];
        const unsigned long long b = bucket_of(r.key, shift, nb);
        const unsigned long long lo = bucket_start(b, nbuckets, count, offsets, tiles);
        const unsigned long long hi = bucket_start(b + 1, nbuckets, count, offsets, tiles);
        if (hi - lo > max_bucket) {  // the bucket geometry was guessed far too coarse: give up, the
            *abort_flag = 1u;        // host re-runs the ordering with the true count
            continue;
        }
        unsigned long long rank = 0;
        for (unsigned long long k = lo; k < hi; ++k) {
            const unsigned long long other = grouped[k].key;
            rank += (other < r.key) || (other == r.key && k < i);
        }
        const unsigned long long pos = lo + rank;
        const unsigned long long low = r.key & kLowMask;
        if (EMIT == 0) {
            lm_hip_coords c;
            c.row = low / cols;
            c.col = low - c.row * cols;
            coords[pos] = c;
            values[pos] = r.value;
            if (pos < pre) {  // the head of the list is mirrored into the staging block
                static_cast<lm_hip_coords *>(pre_out)[pos] = c;
                pre_values[pos] = r.value;
            }
        } else {
            lm_hip_hit h;
            h.position = low;
            h.score = r.value;
            out_hits[pos] = h;
            if (pos < pre)
                static_cast<lm_hip_hit *>(pre_out)[pos] = h;
        }
    }
    // ShortOrder: the host polls a word in the pinned staging block instead of waiting for the kernel's completion signal
    // (~10 us of a 90-250 us call).  Every workgroup makes its writes -- to the pinned block as well -- visible system-wide,
    // then takes a ticket; the last one resets the ticket for the next call and releases `generation` into the word.
    if (done_flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(done_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                __hip_atomic_store(done_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(done_flag, generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ void hits_job_starts(const unsigned long long njobs, const unsigned long long nb,
                                const unsigned long long nbuckets,
                                const unsigned long long *__restrict__ count_ptr, const unsigned long long cap,
                                const unsigned long long *__restrict__ offsets,
                                const unsigned long long *__restrict__ tiles,
                                unsigned long long *__restrict__ starts,
                                unsigned long long *__restrict__ header)
{
    const unsigned long long count = live_count(count_ptr, cap);
    const unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 && header) {  // raw counters {hits, candidates} for the host
        header[0] = count_ptr[0];
        header[1] = count_ptr[1];
    }
    if (j <= njobs)
        starts[j] = bucket_start(j * nb, nbuckets, count, offsets, tiles);
}

// ---- short lists of one job (ShortOrder) ---------------------------------------------------------------------
//
// Each launch of the tail costs 2-5 us whatever it does (5 as the profiler shows them), and a 1 Gbp scan at p = 1e-5 spent
// 28 of its 280 us in the five above (fill, count, scan, scatter, rank: profiles/r05_timeline_fused.txt), a copy-in of
// zeroed counters + the job table before its scan; a 5 Mbp scan spends more there than in its scan.  For ONE job with a short
// list expected:
//   * the re-scoring kernel bumps the bucket count of every record it stores (rescore_candidates);
//   * hits_short_scatter: every workgroup scans the whole histogram in LDS (<= kShortBuckets counts), workgroup 0
//     publishes the offsets, records go to offset + cursor++;
//   * hits_rank_emit as above, which also clears counts and cursors: the buffers are zero between calls (ctx->d_short,
//     ctx->short_dirty covers calls that failed in between);
//   * the list's counters {hits, candidates} live there as well: the scatter kernel leaves a copy for the ranking kernel,
//     which clears the originals -- the next scan starts without a head block copied in, its job an argument of the
//     re-scoring kernel.
constexpr unsigned long long kShortRecords = 40960;  // expected records (the previous call's count + 25 %) up to which a list is "short"
constexpr int kShortBuckets = 10496;                 // >= kShortRecords / 4 + 1 (bucket_geometry), a multiple of kBlock; 41 KB of LDS
constexpr size_t kShortTiles = kShortBuckets / kScanTile + 1;
// ctx->d_short: counts | cursors | tiles (zero for ever: one "tile" per kScanTile buckets, see bucket_start) | offsets |
// the list's counters (zero between calls) | their copy for the ranking kernel, which clears the originals
constexpr size_t kShortOffCursors = kShortBuckets * 4, kShortOffTiles = 2 * kShortOffCursors,
                 kShortOffOffsets = kShortOffTiles + 128,
                 kShortOffCounters = (kShortOffOffsets + (kShortBuckets + 1) * 8 + 255) / 256 * 256,  // {hits, candidates}: a line of their own
                 kShortOffCopy = kShortOffCounters + 256, kShortOffTicket = kShortOffCopy + 128,  // (the ticket of hits_rank_emit's last workgroup)
                 kShortBytes = kShortOffCopy + 256;
static_assert(kShortTiles * 8 <= 128 && kShortBuckets % kBlock == 0 && kShortRecords / 4 + 1 <= kShortBuckets, "layout of the short form");

__global__ __launch_bounds__(kBlock) void hits_short_scatter(
    const HitRecord *__restrict__ hits, const unsigned long long *__restrict__ count_ptr, const unsigned long long cap,
    const int shift, const unsigned nb, const unsigned *__restrict__ counts, unsigned *__restrict__ cursors,
    unsigned long long *__restrict__ offsets, HitRecord *__restrict__ grouped, unsigned long long *__restrict__ counters_copy)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // what the ranking kernel reads while it clears the counters themselves
        counters_copy[0] = count_ptr[0];
        counters_copy[1] = count_ptr[1];
    }
    constexpr int PER = kShortBuckets / kBlock;  // buckets per thread, contiguous
    __shared__ unsigned pre[kShortBuckets];
    __shared__ unsigned wave_sum[kBlock / 64];
    unsigned v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const unsigned b = threadIdx.x * PER + k;
        v[k] = b < nb ? counts[b] : 0u;
        sum += v[k];
    }
    unsigned inc = sum;  // inclusive scan over the wavefront
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off);
        if (lane >= off)
            inc += o;
    }
    if (lane == 63)
        wave_sum[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned run = inc - sum;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w)
        run += wave_sum[w];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        pre[threadIdx.x * PER + k] = run;
        run += v[k];
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (unsigned b = threadIdx.x; b < nb; b += kBlock)
            offsets[b] = pre[b];
    const unsigned long long count = live_count(count_ptr, cap);
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < count;
         i += (unsigned long long)gridDim.x * kBlock) {
        const HitRecord r = hits[i];
        const unsigned b = (unsigned)((r.key & kLowMask) >> shift);
        grouped[pre[b] + atomicAdd(&cursors[b], 1u)] = r;
    }
}

// bucket geometry of order_hits for `sized_for` expected records (one place: the re-scoring kernel counts with it)
void bucket_geometry(unsigned long long sized_for, size_t njobs, unsigned long long max_low, int *shift_out,
                     unsigned long long *nb_out)
{
    // ~1 record per bucket on average; at most 2^26 buckets.  (Ranking is quadratic in the bucket
    // size and the density is far from uniform across jobs: in the JASPAR batch a length-4 motif
    // has 150 x the average hit density; at 8 records per average bucket its buckets held 2 800
    // records and hits_rank_emit took 7 ms of the batch's 47.)
    int shift = 5;
    const long double universe = (long double)max_low * (long double)njobs;
    while (shift < 40 && ((long double)(1ull << shift) * (long double)sized_for < 1.0L * universe))
        ++shift;
    while (shift < 40 && (((max_low - 1) >> shift) + 1) * njobs > (1ull << 26))
        ++shift;
    // short lists (one motif at p = 1e-5 leaves ~10^4 hits per Gbp): four records per bucket keep the histogram small
    // (<= kShortBuckets: the short form above; three tiles of the single-launch scan otherwise), and ranking 4 x 4 keys
    // costs nothing
    if (sized_for <= kShortRecords && njobs == 1 && shift + 2 < 40)
        shift += 2;
    *shift_out = shift;
    *nb_out = ((max_low - 1) >> shift) + 1;
}

constexpr unsigned long long kSortFrom = 1ull << 17;  // expected records from which the list is radix-sorted

__global__ __launch_bounds__(kBlock) void hits_split(const HitRecord *__restrict__ hits,
                                                     const unsigned long long *__restrict__ count_ptr,
                                                     const unsigned long long cap, const unsigned long long n_sort,
                                                     unsigned long long *__restrict__ keys, float *__restrict__ values,
                                                     unsigned *__restrict__ abort_flag)
{
    const unsigned long long count = live_count(count_ptr, cap);
    if (count > n_sort) {  // more records than the arrays were sized for (speculative form): the host runs the exact form
        if (blockIdx.x == 0 && threadIdx.x == 0 && abort_flag)
            *abort_flag = 1u;
        return;
    }
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < n_sort;
         i += (unsigned long long)gridDim.x * kBlock) {
        if (i < count) {
            const HitRecord r = hits[i];
            keys[i] = r.key;
            values[i] = r.value;
        } else {
            keys[i] = ~0ull;
        }
    }
}

template <int EMIT>
__global__ __launch_bounds__(kBlock) void hits_sorted_emit(
    const unsigned long long *__restrict__ keys, const float *__restrict__ vals,
    const unsigned long long *__restrict__ count_ptr, const unsigned long long cap, const unsigned long long n_sort,
    const unsigned long long cols, lm_hip_coords *__restrict__ coords, float *__restrict__ values,
    lm_hip_hit *__restrict__ out_hits, void *__restrict__ pre_out, float *__restrict__ pre_values,
    const unsigned long long pre, unsigned long long *__restrict__ header)
{
    unsigned long long count = live_count(count_ptr, cap);
    if (blockIdx.x == 0 && threadIdx.x == 0 && header) {  // raw counters {hits, candidates} for the host
        header[0] = count_ptr[0];
        header[1] = count_ptr[1];
    }
    if (count > n_sort)
        return;  // (hits_split raised the abort flag)
    for (unsigned long long pos = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; pos < count;
         pos += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long low = keys[pos] & kLowMask;
        const float v = vals[pos];
        if (EMIT == 0) {
            lm_hip_coords c;
            c.row = low / cols;
            c.col = low - c.row * cols;
            coords[pos] = c;
            values[pos] = v;
            if (pos < pre) {
                static_cast<lm_hip_coords *>(pre_out)[pos] = c;
                pre_values[pos] = v;
            }
        } else {
            lm_hip_hit h;
            h.position = low;
            h.score = v;
            out_hits[pos] = h;
            if (pos < pre)
                static_cast<lm_hip_hit *>(pre_out)[pos] = h;
        }
    }
}

// starts[j] = number of keys below (j << 40): lower bound in the sorted keys (sentinels sort behind every real key)
__global__ void hits_sorted_starts(const unsigned long long *__restrict__ keys,
                                   const unsigned long long *__restrict__ count_ptr, const unsigned long long cap,
                                   const unsigned long long n_sort, const unsigned long long njobs,
                                   unsigned long long *__restrict__ starts)
{
    const unsigned long long count = std::min(live_count(count_ptr, cap), n_sort);
    const unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > njobs)
        return;
    const unsigned long long want = j << 40;
    unsigned long long lo = 0, hi = count;
    while (lo < hi) {
        const unsigned long long mid = (lo + hi) / 2;
        if (keys[mid] < want)
            lo = mid + 1;
        else
            hi = mid;
    }
    starts[j] = lo;
}

size_t align16(size_t x) { return (x + 15) / 16 * 16; }

}  // namespace

int short_order_begin(lm_hip_ctx *ctx, unsigned long long expected, size_t njobs, unsigned long long max_low, ShortOrder *so)
{
    so->on = false;
    const unsigned long long sized_for = std::max<unsigned long long>(expected, 4096);
    if (njobs != 1 || sized_for > kShortRecords || max_low == 0 || max_low > kLowMask + 1)
        return LM_HIP_OK;
    bucket_geometry(sized_for, njobs, max_low, &so->shift, &so->nb);
    if (so->nb > (unsigned long long)kShortBuckets)
        return LM_HIP_OK;
    if (!ctx->d_short) {
        LM_HIP_TRY(hipMalloc(&ctx->d_short, kShortBytes));
        ctx->short_dirty = false;
        LM_HIP_TRY(hipMemsetAsync(ctx->d_short, 0, kShortBytes, ctx->stream));
    } else if (ctx->short_dirty) {
        LM_HIP_TRY(hipMemsetAsync(ctx->d_short, 0, kShortBytes, ctx->stream));
    }
    ctx->short_dirty = true;  // until order_hits has seen the clean-up through
    so->counts = ctx->d_short;
    so->counters = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ctx->d_short) + kShortOffCounters);
    so->counters_copy = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ctx->d_short) + kShortOffCopy);
    so->on = true;
    return LM_HIP_OK;
}

void HitOutput::release()
{
    result_free(coords);
    result_free(values);
    result_free(hits);
    coords = nullptr;
    values = nullptr;
    hits = nullptr;
    total = 0;
}

// `d_hits`: the records (in ctx->scratch) of `njobs` jobs whose keys' low parts are all below
// `max_low`; `d_counters` = the two device counters {hits, candidates} the scans bumped.
//
// count == ~0 ("speculative"): the host does not know the count yet.  The ordering is
// enqueued right behind the scans with a bucket geometry sized for `expected` records and
// every array sized for `cap`; counters, job offsets and the first kPrefix records come
// back in one pinned copy behind ONE synchronisation -- the usual p = 1e-5 scan then costs
// one host round trip instead of two.  *status: 0 = done, 1 = a list overflowed (counts in
// counts_out, nothing produced), 2 = counts fine but the geometry guess was too coarse
// (the caller runs the exact form).  With a known `count` the geometry is exact.
// On success `out` owns malloc'ed host arrays in key order and job_start[j] ..
// job_start[j + 1] delimit job j.  Uses ctx->scratch2; synchronises.
constexpr unsigned long long kPrefix = 12800;  // x 20 B = 256 KB

int order_hits(lm_hip_ctx *ctx, const HitRecord *d_hits, const unsigned long long *d_counters,
               unsigned long long count, unsigned long long cap, unsigned long long cand_cap,
               unsigned long long expected, size_t njobs, unsigned long long max_low, int emit, size_t cols,
               HitOutput *out, int *status, unsigned long long counts_out[2], const ShortOrder *so)
{
    const bool speculative = count == ~0ull;
    *status = 0;
    out->job_start.assign(njobs + 1, 0);
    out->total = 0;
    if (!speculative && count == 0)
        return LM_HIP_OK;
    if (max_low == 0)
        max_low = 1;
    if (max_low > kLowMask + 1)
        return fail(LM_HIP_ERR_CAPACITY, "fused threshold: %llu cells per job exceed the 2^40 key space",
                    max_low);
    const unsigned long long sized_for = speculative ? std::max<unsigned long long>(expected, 4096) : count;
    const unsigned long long room = speculative ? cap : count;  // records the arrays must hold
    int shift = 0;
    unsigned long long nb = 0;
    bucket_geometry(sized_for, njobs, max_low, &shift, &nb);
    const bool short_form = so && so->on;
    if (short_form && (!speculative || so->shift != shift || so->nb != nb || njobs != 1 || d_counters != so->counters))
        return fail(LM_HIP_ERR_BAD_ARGS, "fused threshold: the short ordering was begun with another geometry");
    const unsigned long long nbuckets = nb * njobs;
    const unsigned long long ntiles = (nbuckets + kScanTile - 1) / kScanTile;

    const size_t rec_bytes = emit == 0 ? sizeof(lm_hip_coords) : sizeof(lm_hip_hit);
    // long lists: radix sort (see the head of this file).  Speculative form: the arrays hold the expected count with a
    // quarter of headroom; a longer list raises the abort flag and the exact form sorts the true count.
    const bool sorted = ctx->sort_hits && sized_for >= kSortFrom && njobs < (1ull << 22);
    const unsigned long long n_sort = !sorted ? 0 : speculative ? std::min(room, sized_for + sized_for / 4 + 4096) : count;
    int end_bit = 41;
    while ((1ull << (end_bit - 41)) < njobs)
        ++end_bit;  // keys are below njobs << 40; one spare bit keeps the all-ones sentinel strictly behind them
    size_t sort_temp = 0;
    if (sorted) {
        unsigned long long *kp = nullptr;
        float *vp = nullptr;
        LM_HIP_TRY(rocprim::radix_sort_pairs(nullptr, sort_temp, kp, kp, vp, vp, (size_t)n_sort, 0u, (unsigned)end_bit, ctx->stream));
    }
    const size_t off_keys_in = 0, off_keys_out = off_keys_in + align16(n_sort * 8);
    const size_t off_vals_in = off_keys_out + align16(n_sort * 8), off_vals_out = off_vals_in + align16(n_sort * 4);
    const size_t off_sort_temp = off_vals_out + align16(n_sort * 4);
    const size_t off_grouped = 0;
    const size_t off_counts = off_grouped + align16(room * sizeof(HitRecord));
    const size_t off_offsets = off_counts + (nbuckets * 4 + 255) / 256 * 256;
    const size_t off_tiles = off_offsets + align16(nbuckets * 8);
    const size_t off_total = sorted ? (off_sort_temp + sort_temp + 255) / 256 * 256 : off_tiles + align16(ntiles * 8);
    // head of the list mirrored into the staging block: the previous call's length with headroom,
    // bounded so that the staging copy stays a small pinned transfer
    const unsigned long long pre =
        speculative ? std::min(room, std::min<unsigned long long>(std::max<unsigned long long>(sized_for, 2048), kPrefix)) : 0;
    // exact form: starts | records | values contiguous in scratch2 (one read-back copy)
    const size_t off_starts = off_total + 16;
    const size_t off_out = off_starts + align16((njobs + 1) * 8);
    const size_t off_values = off_out + align16(room * rec_bytes);
    const size_t bytes = off_values + align16(room * sizeof(float));
    // speculative form: the staging block -- counters | abort flag | starts | head of the list --
    // lives in the context's pinned host buffer and the kernels write it THERE (posted writes
    // over PCIe, visible after the stream synchronisation): no copy command, no memset
    const size_t p_abort = 16, p_starts = 32;
    const size_t p_out = p_starts + align16((njobs + 1) * 8);
    const size_t p_values = p_out + align16(pre * rec_bytes);
    const size_t p_bytes = p_values + align16(pre * sizeof(float));
    char *pin = static_cast<char *>(ctx->pinned);
    static_assert(sizeof(size_t) == 8, "job offsets are read back as 64-bit values");
    const size_t starts_bytes = (njobs + 1) * 8;
    if (speculative && p_bytes > kPinnedBytes / 2) {
        *status = 2;  // too many jobs for the staging block: the caller runs the exact form
        LM_HIP_TRY(hipMemcpyAsync(pin, d_counters, 16, hipMemcpyDeviceToHost, ctx->stream));
        LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        counts_out[0] = reinterpret_cast<unsigned long long *>(pin)[0];
        counts_out[1] = reinterpret_cast<unsigned long long *>(pin)[1];
        if (counts_out[0] > cap || counts_out[1] > cand_cap)
            *status = 1;
        return LM_HIP_OK;
    }
    LM_TRY(ctx->scratch2.reserve(bytes));
    char *base = static_cast<char *>(ctx->scratch2.ptr);
    HitRecord *grouped = reinterpret_cast<HitRecord *>(base + off_grouped);
    unsigned *counts = reinterpret_cast<unsigned *>(base + off_counts);
    unsigned long long *offsets = reinterpret_cast<unsigned long long *>(base + off_offsets);
    unsigned long long *tiles = reinterpret_cast<unsigned long long *>(base + off_tiles);
    unsigned long long *total = reinterpret_cast<unsigned long long *>(base + off_total);
    if (short_form) {  // histogram, cursors, offsets and the (zero) tiles of the short form live in the context
        char *sb = reinterpret_cast<char *>(ctx->d_short);
        counts = reinterpret_cast<unsigned *>(sb);
        tiles = reinterpret_cast<unsigned long long *>(sb + kShortOffTiles);
        offsets = reinterpret_cast<unsigned long long *>(sb + kShortOffOffsets);
    }
    unsigned long long *header = reinterpret_cast<unsigned long long *>(pin);
    unsigned *abort_flag = reinterpret_cast<unsigned *>(pin + p_abort);
    void *pre_out = pin + p_out;
    float *pre_values = reinterpret_cast<float *>(pin + p_values);
    unsigned long long *starts =
        reinterpret_cast<unsigned long long *>(speculative ? pin + p_starts : base + off_starts);
    void *d_out = base + off_out;
    float *d_values = reinterpret_cast<float *>(base + off_values);
    if (speculative)
        memset(pin, 0, 32);  // counters and abort flag (the previous call's results were consumed)
    // a guess that is off by orders of magnitude (first call of a context: 4 096 expected, millions
    // found) leaves hundreds of records per bucket and a quadratic ranking pass (23 ms on the JASPAR
    // batch): give up early, the exact form costs a fraction of a millisecond
    const unsigned long long max_bucket = speculative ? 256 : ~0ull;

    hipStream_t st = ctx->stream;
    // short form, option "poll_done" = 1: the end of the call is polled (hits_rank_emit's last workgroup; the word sits in the
    // zeroed head of the staging block, the ticket next to the context's counters) instead of waited for on the stream.
    // 2.6-4.4 us less per call when nothing synchronises behind it -- and ~13 us MORE when the caller follows the call with a
    // device synchronisation of its own, which then finds the ranking kernel still retiring and pays a full wake-up
    // (bench.py's protocol does: 0.252 -> 0.265 ms).  Hence off unless asked for (profiles/r06_poll_done_ab.txt).
    unsigned *done_ticket = nullptr, *done_flag = nullptr;
    unsigned generation = 0;
    if (short_form && speculative && ctx->poll_done) {
        done_ticket = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(ctx->d_short) + kShortOffTicket);
        done_flag = reinterpret_cast<unsigned *>(pin + 24);
        generation = ++ctx->short_generation ? ctx->short_generation : ++ctx->short_generation;  // never 0
    }
    const unsigned grid = (unsigned)std::max<unsigned long long>(
        std::min<unsigned long long>((sized_for + kBlock - 1) / kBlock, (unsigned long long)ctx->num_cus * 32), 1);
    if (sorted) {
        unsigned long long *keys_in = reinterpret_cast<unsigned long long *>(base + off_keys_in);
        unsigned long long *keys_out = reinterpret_cast<unsigned long long *>(base + off_keys_out);
        float *vals_in = reinterpret_cast<float *>(base + off_vals_in), *vals_out = reinterpret_cast<float *>(base + off_vals_out);
        hipLaunchKernelGGL(hits_split, dim3(grid), dim3(kBlock), 0, st, d_hits, d_counters, cap, n_sort, keys_in, vals_in,
                           speculative ? abort_flag : static_cast<unsigned *>(nullptr));
        LM_HIP_TRY(hipGetLastError());
        size_t tb = sort_temp;
        LM_HIP_TRY(rocprim::radix_sort_pairs(base + off_sort_temp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n_sort, 0u,
                                             (unsigned)end_bit, st));
        if (emit == 0)
            hipLaunchKernelGGL(hits_sorted_emit<0>, dim3(grid), dim3(kBlock), 0, st, keys_out, vals_out, d_counters, cap, n_sort,
                               (unsigned long long)cols, static_cast<lm_hip_coords *>(d_out), d_values,
                               static_cast<lm_hip_hit *>(nullptr), pre_out, pre_values, pre,
                               speculative ? header : static_cast<unsigned long long *>(nullptr));
        else
            hipLaunchKernelGGL(hits_sorted_emit<1>, dim3(grid), dim3(kBlock), 0, st, keys_out, vals_out, d_counters, cap, n_sort,
                               (unsigned long long)cols, static_cast<lm_hip_coords *>(nullptr), static_cast<float *>(nullptr),
                               static_cast<lm_hip_hit *>(d_out), pre_out, pre_values, pre,
                               speculative ? header : static_cast<unsigned long long *>(nullptr));
        hipLaunchKernelGGL(hits_sorted_starts, dim3((unsigned)((njobs + 1 + 255) / 256)), dim3(256), 0, st, keys_out, d_counters, cap,
                           n_sort, (unsigned long long)njobs, starts);
        LM_HIP_TRY(hipGetLastError());
    } else {
    const bool inline_starts = njobs <= 1024;  // (one workgroup of hits_rank_emit writes them)
    unsigned *cursors = short_form ? counts + kShortBuckets : nullptr;
    const unsigned long long *rank_counters = short_form ? so->counters_copy : d_counters;
    if (short_form) {  // the re-scoring kernel has counted
        hipLaunchKernelGGL(hits_short_scatter, dim3(grid), dim3(kBlock), 0, st, d_hits, d_counters, cap, shift, (unsigned)nb,
                           counts, cursors, offsets, grouped, so->counters_copy);
    } else {
        LM_HIP_TRY(hipMemsetAsync(counts, 0, (nbuckets * 4 + 255) / 256 * 256, st));  // whole 256-B units: one fill kernel
        hipLaunchKernelGGL(hits_bucket_count, dim3(grid), dim3(kBlock), 0, st, d_hits, d_counters, cap, shift, nb,
                           counts);
        LM_TRY(launch_scan_u32(ctx, counts, nbuckets, offsets, tiles, total));
        hipLaunchKernelGGL(hits_bucket_scatter, dim3(grid), dim3(kBlock), 0, st, d_hits, d_counters, cap, shift, nb,
                           offsets, tiles, counts, grouped);
    }
    if (emit == 0)
        hipLaunchKernelGGL(hits_rank_emit<0>, dim3(grid), dim3(kBlock), 0, st, grouped, rank_counters, cap, shift,
                           nb, nbuckets, offsets, tiles, (unsigned long long)cols,
                           static_cast<lm_hip_coords *>(d_out), d_values,
                           static_cast<lm_hip_hit *>(nullptr), max_bucket, abort_flag, pre_out, pre_values, pre,
                           (unsigned long long)njobs, inline_starts ? starts : nullptr, inline_starts && speculative ? header : nullptr,
                           short_form ? counts : nullptr, cursors, short_form ? so->counters : nullptr, done_ticket, done_flag,
                           generation);
    else
        hipLaunchKernelGGL(hits_rank_emit<1>, dim3(grid), dim3(kBlock), 0, st, grouped, rank_counters, cap, shift,
                           nb, nbuckets, offsets, tiles, (unsigned long long)cols,
                           static_cast<lm_hip_coords *>(nullptr), static_cast<float *>(nullptr),
                           static_cast<lm_hip_hit *>(d_out), max_bucket, abort_flag, pre_out, pre_values, pre,
                           (unsigned long long)njobs, inline_starts ? starts : nullptr, inline_starts && speculative ? header : nullptr,
                           short_form ? counts : nullptr, cursors, short_form ? so->counters : nullptr, done_ticket, done_flag,
                           generation);
    if (!inline_starts)
        hipLaunchKernelGGL(hits_job_starts, dim3((unsigned)((njobs + 1 + 255) / 256)), dim3(256), 0, st,
                           (unsigned long long)njobs, nb, nbuckets, d_counters, cap, offsets, tiles, starts,
                           speculative ? header : static_cast<unsigned long long *>(nullptr));
    LM_HIP_TRY(hipGetLastError());
    }

    scan_timer_mark(ctx, st, 3);  // (time_scan) behind the ordering kernels
    if (speculative) {
        bool seen = false;
        if (done_flag) {  // the ranking kernel's last workgroup raises the word behind everything it and the others wrote
            const volatile unsigned *flag = done_flag;
            for (unsigned spin = 0; spin < (1u << 23); ++spin) {  // bounded: a kernel that never gets there is caught below
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == generation) {
                    seen = true;
                    break;
                }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
                __builtin_ia32_pause();
#endif
            }
        }
        if (!seen || ctx->scan_timed)  // (option "time_scan": the events behind the kernels are read next -- they must have completed)
            LM_HIP_TRY(hipStreamSynchronize(st));
        if (short_form)
            ctx->short_dirty = false;  // both kernels ran: counts and cursors are zero again
        counts_out[0] = reinterpret_cast<unsigned long long *>(pin)[0];
        counts_out[1] = reinterpret_cast<unsigned long long *>(pin)[1];
        if (counts_out[0] > cap || counts_out[1] > cand_cap) {
            *status = 1;
            return LM_HIP_OK;
        }
        if (*reinterpret_cast<unsigned *>(pin + p_abort)) {
            *status = 2;
            return LM_HIP_OK;
        }
        count = counts_out[0];
        if (count == 0)
            return LM_HIP_OK;
        void *host = result_alloc(count * rec_bytes);
        float *host_values = emit == 0 ? static_cast<float *>(result_alloc(count * sizeof(float))) : nullptr;
        if (!host || (emit == 0 && !host_values)) {
            result_free(host);
            result_free(host_values);
            return fail(LM_HIP_ERR_OOM, "fused threshold: cannot allocate %llu hits on the host", count);
        }
        memcpy(out->job_start.data(), pin + p_starts, starts_bytes);
        const unsigned long long have = std::min(count, pre);
        memcpy(host, pin + p_out, have * rec_bytes);
        if (host_values)
            memcpy(host_values, pin + p_values, have * sizeof(float));
        if (count > have) {  // long list: the rest comes straight from the device
            hipError_t e = hipMemcpyAsync(static_cast<char *>(host) + have * rec_bytes,
                                          static_cast<char *>(d_out) + have * rec_bytes,
                                          (count - have) * rec_bytes, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && host_values)
                e = hipMemcpyAsync(host_values + have, d_values + have, (count - have) * sizeof(float),
                                   hipMemcpyDeviceToHost, st);
            if (e == hipSuccess)
                e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                result_free(host);
                result_free(host_values);
                return fail(LM_HIP_ERR_HIP, "fused threshold: read-back failed: %s", hipGetErrorString(e));
            }
        }
        out->total = (size_t)count;
        if (emit == 0) {
            out->coords = static_cast<lm_hip_coords *>(host);
            out->values = host_values;
        } else {
            out->hits = static_cast<lm_hip_hit *>(host);
        }
        return LM_HIP_OK;
    }

    void *host = result_alloc(count * rec_bytes);
    float *host_values = emit == 0 ? static_cast<float *>(result_alloc(count * sizeof(float))) : nullptr;
    if (!host || (emit == 0 && !host_values)) {
        result_free(host);
        result_free(host_values);
        return fail(LM_HIP_ERR_OOM, "fused threshold: cannot allocate %llu hits on the host", count);
    }
    // Read-back.  starts | records | values are contiguous in scratch2: small results come back
    // as ONE copy into the pinned buffer (a copy into pageable memory costs ~15 us each),
    // large ones go straight into the arrays handed to the caller.
    const size_t block_bytes = off_values + count * sizeof(float) - off_starts;
    hipError_t e;
    if (block_bytes <= (256u << 10)) {
        e = hipMemcpyAsync(pin, base + off_starts, block_bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
        if (e == hipSuccess) {
            memcpy(out->job_start.data(), pin, starts_bytes);
            memcpy(host, pin + (off_out - off_starts), count * rec_bytes);
            if (host_values)
                memcpy(host_values, pin + (off_values - off_starts), count * sizeof(float));
        }
    } else {
        e = hipMemcpyAsync(out->job_start.data(), starts, starts_bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(host, d_out, count * rec_bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && host_values)
            e = hipMemcpyAsync(host_values, d_values, count * sizeof(float), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
        result_free(host);
        result_free(host_values);
        return fail(LM_HIP_ERR_HIP, "fused threshold: ordering the hit list failed: %s",
                    hipGetErrorString(e));
    }
    out->total = (size_t)count;
    if (emit == 0) {
        out->coords = static_cast<lm_hip_coords *>(host);
        out->values = host_values;
    } else {
        out->hits = static_cast<lm_hip_hit *>(host);
    }
    return LM_HIP_OK;
}

}  // namespace lm
