// scanmax.hip -- `Scanner::max` exactly as the reference computes it (lightmotif/src/scan.rs:200-249), on the device.
//
// The reference walks the sequence block by block: the u8 scores of a DiscreteMatrix pick candidate cells (u8 score
// >= the current level, visited in row-major order inside a block, blocks ascending -- i.e. in row-major order of
// the whole matrix), each candidate is re-scored in f32 (`score_position`) and replaces the best hit when its score
// is greater, or equal at a greater position; the level then becomes THAT cell's u8 score.  While no hit is held
// the first candidate is taken as it is and the level stays the scaled threshold.  No `position + M <= L` test is
// made; a candidate whose window leaves the striped matrix makes the reference panic (seq.rs:433-442 indexes
// column C).  The answer depends on the order of the walk, so it cannot be a plain reduction.
//
// Device form: the state (have, best score, best position, level) lives on the device; rows are scored window by
// window (the DiscreteMatrix's u8 scores into one reusable buffer; windows start small and double, the walk makes
// most of its updates early), and inside a window the walk is a chain of "find the FIRST cell at or after the cursor
// that the reference would act on" -- a parallel search with an ordered minimum -- followed by a one-thread state
// update.  The f32 score of a candidate is computed where it is needed (`score_cell`: the M sequential adds of
// `score_position`, pwm/mod.rs:651-662 -- the same f32 sequence as the store kernel's, so the same bits): candidates
// are the cells whose u8 score reaches the current level, a vanishing fraction once the first windows are walked,
// and the first version of this file spent a third of its time materialising 4 B per cell for them.  The expected
// number of updates over n cells of continuous scores is ~ln n, less than one per doubling window, so beyond the
// first windows the host does not wait per window either: it enqueues a batch of windows, each with a fixed number of
// find / update rounds, and a window that would have needed more rounds raises a `stall` mark that turns the rest of
// the batch into no-ops; the host then finishes that window round by round and carries on behind it.
#include <algorithm>
#include <utility>
#include <vector>
#include <cstdio>
#include <cstdlib>

#include "score_launch.hpp"

namespace lm {

struct ScanMaxState {
    unsigned long long index;   // position col * rows + row of the best hit
    unsigned long long cursor;  // first cell (row-major, within the window) not yet walked
    unsigned long long found;   // search result: flat cell of the window, ~0 = none
    float score;
    unsigned level;
    int have;
    int err;                    // 1: a candidate's window leaves the striped matrix (the reference panics)
    int more;                   // the last update consumed a cell: search again
    int stall;                  // window id + 1 of a pipelined window that ran out of rounds (the kernels behind it do nothing)
    unsigned pad[2];
};

namespace {

// Where the f32 score of a cell comes from: the striped sequence and the dense M x K weights.
struct ScanMaxSource {
    const uint8_t *seq;          // striped matrix, rows + wrap rows
    unsigned long long stride;
    const float *dense;          // dense[j * k + s] = pssm[j][s]
    unsigned k;
};

// `score_position` (pwm/mod.rs:651-662): 0.0 + P[0][s0] + P[1][s1] + ... in motif order, f32, no contraction --
// the add sequence of score_rows_into (pli/mod.rs:98-102), hence the value the store kernel would have written.
__device__ __forceinline__ float score_cell(const ScanMaxSource &src, const unsigned long long row, const unsigned col,
                                            const unsigned m)
{
    const uint8_t *p = src.seq + row * src.stride + col;
    float acc = 0.0f;
    for (unsigned j = 0; j < m; ++j)
        acc = acc + src.dense[j * src.k + p[(unsigned long long)j * src.stride]];
    return acc;
}

// The state update behind one search (ONE wavefront, a launch of its own: letting the search's last workgroup do it
// was tried -- a ticket per workgroup means an agent-scope release per workgroup, an L2 write-back on this
// multi-XCD part, and the search got 2.7 x slower): the found cell becomes the best hit (scan.rs:229-242), or the
// window is walked.  `last_round` != 0 (pipelined windows): this was the window's last search; if its walk is not
// finished, mark the window (`stall = window_id + 1`) -- every kernel enqueued behind it then leaves the state alone.
//
// Behind the found cell the wavefront WALKS ON, 64 cells at a time, for as long as the cells keep updating the state:
// on continuous scores the next update is ~n cells away and the first chunk ends the walk (a few microseconds), but
// "equal score at a greater position replaces the best" (scan.rs:237) fires once per ROW across a run of equal
// best-scoring cells -- homopolymers, low-complexity repeats, constant sequences -- and a search + update launch pair
// per row made such inputs O(rows) launches (tens of seconds per 100 Mbp where the reference's single pass takes two).
// Each lane scores its cell once; the updates inside a chunk are then resolved in cell order with ballots (the level
// only rises, so the chunk's candidates at its first level contain all later ones).  Bounded per launch.
constexpr unsigned long long kWalkCells = 1ull << 15;

__global__ __launch_bounds__(64) void scanmax_apply(const uint8_t *__restrict__ d, const ScanMaxSource src,
                                                    const unsigned long long ncells, const unsigned cols,
                                                    const unsigned long long row0, const unsigned long long rows,
                                                    const unsigned m, const int last_round, const int window_id,
                                                    ScanMaxState *__restrict__ st, ScanMaxState *__restrict__ host_copy)
{
    ScanMaxState t = *st;  // uniform over the wavefront
    if (t.err | t.stall)
        return;
    const unsigned lane = threadIdx.x;
    const unsigned long long total = rows * cols;
    if (t.found != ~0ull) {
        const unsigned long long f = t.found;
        const unsigned long long r = row0 + f / cols, col = f % cols;
        const unsigned long long index = col * rows + r;
        if (index + m > total) {
            t.err = 1;
            t.index = index;
            t.cursor = ncells;
            t.more = 0;
        } else {
            if (t.have)
                t.level = d[f];  // scan.rs:238 best_discrete = dscore (the first hit keeps the scaled threshold, :241)
            t.have = 1;
            t.score = score_cell(src, r, (unsigned)col, m);  // = score_position (pwm/mod.rs:651-662)
            t.index = index;
            t.cursor = f + 1;
            bool hot = true;
            for (unsigned long long walked = 0; hot && !t.err && t.cursor < ncells && walked < kWalkCells; walked += 64) {
                hot = false;
                const unsigned long long g = t.cursor + lane;
                const bool valid = g < ncells;
                const unsigned dv = valid ? d[g] : 0u;
                const unsigned long long gr = row0 + g / cols, gc = g % cols;
                const unsigned long long gi = gc * rows + gr;  // scan.rs:231
                const bool cand = valid && dv >= t.level;
                const bool leaves = cand && gi + m > total;  // score_position would index column C: the reference panics
                const float x = (cand && !leaves) ? score_cell(src, gr, (unsigned)gc, m) : 0.0f;
                unsigned long long pending = __ballot(cand);
                while (pending) {
                    const bool act = ((pending >> lane) & 1ull) && dv >= t.level &&
                                     (leaves || x > t.score || (x == t.score && gi > t.index));
                    const unsigned long long am = __ballot(act);
                    if (!am)
                        break;
                    const int fl = __ffsll((long long)am) - 1;
                    const unsigned long long i_s = __shfl(gi, fl);
                    if (__shfl((int)leaves, fl)) {
                        t.err = 1;
                        t.index = i_s;
                        break;
                    }
                    t.level = __shfl(dv, fl);
                    t.score = __shfl(x, fl);
                    t.index = i_s;
                    hot = true;
                    pending &= ~((2ull << fl) - 1ull);  // the cells behind it, against the new state
                }
                t.cursor = t.cursor + 64 < ncells ? t.cursor + 64 : ncells;
            }
            if (t.err)
                t.cursor = ncells;
            t.more = !t.err && t.cursor < ncells;
        }
    } else {
        t.more = 0;
        t.cursor = ncells;
    }
    t.found = ~0ull;
    if (last_round && t.more)
        t.stall = window_id + 1;
    if (lane == 0) {
        *st = t;
        *host_copy = t;
    }
}

// One search of the walk over the window in `d`: the FIRST cell at or after the cursor the reference's loop would ACT
// on -- u8 score >= level and (no hit held, or a greater f32 score, or an equal one at a greater position,
// scan.rs:229-242), or a window that leaves the matrix -- found by all workgroups with an ordered minimum.
// `new_window`: the cursor starts at the window's first cell.  Every workgroup owns ONE contiguous span of 4 096-cell
// chunks, in ascending order; a lane turns its 16 cells into a bit mask of `u8 >= level` (all but never empty), and
// only the set bits are looked at.  `found` is polled from L2 every fourth chunk: every wavefront asking for that one
// address per chunk was 1 M same-address requests per Gbp; a late answer only delays the early exit.
__global__ __launch_bounds__(kBlock) void scanmax_find(const uint8_t *__restrict__ d, const ScanMaxSource src,
                                                       const unsigned long long ncells, const unsigned cols,
                                                       const unsigned long long row0, const unsigned long long rows,
                                                       const unsigned m, const int new_window, ScanMaxState *__restrict__ st)
{
    if (st->err | st->stall)
        return;
    const unsigned long long cursor = new_window ? 0ull : st->cursor;
    if (cursor >= ncells)
        return;
    const unsigned level = st->level;
    const int have = st->have;
    const float best = st->score;
    const unsigned long long best_index = st->index;
    const unsigned long long total = rows * cols;
    constexpr unsigned long long CH = (unsigned long long)kBlock * 16;  // cells per workgroup chunk
    const unsigned long long first_chunk = cursor / CH, nchunks = (ncells + CH - 1) / CH;
    const unsigned long long per = (nchunks - first_chunk + gridDim.x - 1) / gridDim.x;
    unsigned long long c = first_chunk + (unsigned long long)blockIdx.x * per;
    const unsigned long long c_end = c + per < nchunks ? c + per : nchunks;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // a lane's cells ascend from chunk to chunk, so its first find is its smallest; the wavefront leaves the span as
    // soon as any lane has one (a ballot per chunk -- the 64-bit minimum over the lanes is taken once, afterwards)
    unsigned long long mine = ~0ull;
    // the lane's 16 cells of chunk `cc` as four dwords (zero past the window); the next chunk's are requested before
    // this chunk's are looked at
    auto load_cells = [&](const unsigned long long cc, unsigned (&w)[4]) {
        const unsigned long long f0 = cc * CH + 16ull * threadIdx.x;
        w[0] = w[1] = w[2] = w[3] = 0u;
        if (f0 + 16 <= ncells && (reinterpret_cast<uintptr_t>(d + f0) & 15) == 0) {
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(d + f0));
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            for (int i = 0; i < 16; ++i)
                if (f0 + i < ncells)
                    w[i / 4] |= (unsigned)d[f0 + i] << (8 * (i % 4));
        }
    };
    unsigned w[4], wn[4] = {0u, 0u, 0u, 0u};
    if (c < c_end)
        load_cells(c, w);
    for (unsigned it = 0; c < c_end; ++c, ++it) {
        const unsigned long long c0 = c * CH;
        // chunks are taken in ascending order: nothing at or beyond an already found cell can be the first
        if ((it & 3u) == 0 && c0 >= __hip_atomic_load(&st->found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            break;
        if (c + 1 < c_end)
            load_cells(c + 1, wn);
        const unsigned long long f0 = c0 + 16ull * threadIdx.x;
        if (f0 < ncells) {
            unsigned mask = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                mask |= (unsigned)(((w[i / 4] >> (8 * (i % 4))) & 0xffu) >= level) << i;
            if (f0 + 16 > ncells)
                mask &= (1u << (unsigned)(ncells - f0)) - 1u;   // cells past the window
            if (f0 < cursor)
                mask = cursor - f0 >= 16 ? 0u : mask & ~((1u << (unsigned)(cursor - f0)) - 1u);  // cells already walked
            while (mask && mine == ~0ull) {
                const int i = __ffs(mask) - 1;
                mask &= mask - 1;
                const unsigned long long f = f0 + i;
                const unsigned long long r = row0 + f / cols, col = f % cols;
                const unsigned long long index = col * rows + r;  // scan.rs:231
                bool act = index + m > total;                    // score_position would index column C
                if (!act) {
                    const float x = score_cell(src, r, (unsigned)col, m);
                    act = !have || x > best || (x == best && index > best_index);
                }
                if (act)
                    mine = f;
            }
        }
        if (__ballot(mine != ~0ull))  // later chunks of the span cannot hold the first cell
            break;
        w[0] = wn[0]; w[1] = wn[1]; w[2] = wn[2]; w[3] = wn[3];
    }
    if (__ballot(mine != ~0ull)) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(mine, off);
            mine = o < mine ? o : mine;
        }
        if ((threadIdx.x & 63) == 0)
            atomicMin(&st->found, mine);
    }
}

// the host takes a stalled window over (its u8 scores are in the buffer again): cursor and level stay
__global__ void scanmax_resume(ScanMaxState *__restrict__ st)
{
    st->stall = 0;
    st->found = ~0ull;
}

}  // namespace

// ---- the walk over a candidate LIST (round 6) -------------------------------------------------------------------------
//
// The level only rises, so every cell the walk ever acts on has a u8 score >= the level it STARTS with.  At the thresholds a
// Scanner is used with (the CLI's p = 1e-5: ten thousand such cells per Gbp) materialising 1 B per cell and searching it
// window by window is the wrong shape: the u8 sums of a DiscreteMatrix are exactly what the pair scan adds up
// (score_c32_prefilter2 on the pair table of the u8 weights: min(sum, 255) >= level <=> sum >= level for the saturating
// tiers, avx2.rs:336), so ONE flag scan at td = level finds the row ranges that hold them, `scanmax_gate` turns those into
// records (row-major cell, u8 score, f32 score = score_position's add sequence), the hit-list ordering of the fused threshold
// puts them in the walk's order -- the key is cell * 256 + u8, so the u8 rides through the ordering untouched -- and the HOST
// walks the few thousand records with the reference's rule (scan.rs:229-242).  1.3 -> 0.3 ms per Gbp at p = 1e-5.  Lists
// that would be long (a low threshold: every cell a candidate), wrapping sums (Generic), other geometries and alphabets take
// the window walk below, which the tests hold equal to this one.

__global__ __launch_bounds__(kBlock) void scanmax_gate(const FusedOut fo, const uint8_t *__restrict__ seq, const float *__restrict__ dense,
                                                       const uint8_t *__restrict__ dw, const unsigned m, const unsigned k,
                                                       const unsigned level, const unsigned long long first_row)
{
    unsigned long long n = *fo.cand_count;
    if (n > fo.cand_capacity)
        n = fo.cand_capacity;  // overflow: the host sees the count and takes the window walk
    const unsigned lane = threadIdx.x & 31;
    const unsigned long long stride = (unsigned long long)gridDim.x * (kBlock / 32);
    for (unsigned long long c0 = (unsigned long long)blockIdx.x * (kBlock / 32); c0 < n; c0 += stride) {  // block-uniform trip count
        const unsigned long long c = c0 + (threadIdx.x >> 5);
        bool hit = false;
        HitRecord r{};
        if (c < n) {
            const Candidate cd = fo.cands[c];
            const unsigned long long r0 = cd.key & ((1ull << 40) - 1);  // relative to first_row
            if (lane < cd.nrows) {
                const unsigned long long row = first_row + r0 + lane;
                const uint8_t *p = seq + row * 32 + cd.col;
                unsigned sum = 0;
                float sc = 0.0f;
                for (unsigned j = 0; j < m; ++j) {  // the u8 weights and the f32 weights of the same symbols, in motif order
                    const unsigned sy = p[(unsigned long long)j * 32];
                    sum += dw[j * k + sy];
                    sc = sc + dense[j * k + sy];  // score_position (pwm/mod.rs:651-662)
                }
                if (sum >= level) {
                    hit = true;
                    r.key = ((row * 32ull + cd.col) << 8) | (sum < 255u ? sum : 255u);
                    r.value = sc;
                    r.pad = 0;
                }
            }
        }
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            const int wl = threadIdx.x & 63;
            const int leader = __ffsll((long long)mask) - 1;
            unsigned long long base = 0;
            if (wl == leader)
                base = atomicAdd(fo.hit_count, (unsigned long long)__popcll(mask));
            base = __shfl(base, leader);
            const unsigned long long slot = base + __popcll(mask & ((1ull << wl) - 1ull));
            if (hit && slot < fo.hit_capacity)
                fo.hits[slot] = r;
        }
    }
}

// The walk's state between two row ranges (and what the window walk takes over when a range's list turns out long)
struct WalkState {
    unsigned level;
    bool have;
    unsigned long long position;
    float score;
};

// Walks rows [first_row, row_end) from `st`.  *route = 0: walked, `st` is the state behind row_end; 1: not this route (no
// state change): the caller takes the window walk from first_row.
static int scan_max_by_list(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq, const uint8_t *weights, size_t wstride,
                            size_t first_row, size_t row_end, WalkState *st, int *route)
{
    *route = 1;
    const size_t rows = seq->rows, cols = seq->cols, m = pssm->m, k = pssm->k;
    const unsigned level = st->level;
    const unsigned long long cells = (unsigned long long)(row_end - first_row) * cols;
    // the list's keys are cell * 256 + u8 on 40 bits; a level of 0 makes every cell a candidate
    if (!ctx->list_scan_max || cols != 32 || seq->stride != 32 || k != 5 || m < 2 || m > (size_t)kMaxFastM || level < 1 || level > 255 ||
        (unsigned long long)rows * cols >= (1ull << 32) || reinterpret_cast<uintptr_t>(seq->d_data) % 4 != 0)
        return LM_HIP_OK;
    PrefilterLauncher scan = score_c32_prefilter2_lookup((int)m, 5);
    ScoreArgs sa{nullptr, seq->d_data, seq->stride, cols, first_row, row_end, nullptr, cols};
    const C32Plan plan = plan_c32(ctx, MotifShape{m, k, true}, sa, false, 2, 1);
    if (!scan || !plan.ok)
        return LM_HIP_OK;
    // the pair table and the dense copy of the u8 weights (host-packed, one small copy; the matrix of a Scanner changes per call)
    const size_t image_bytes = (size_t)prefilter2_image_dw((int)m) * 4, dense_bytes = (m * k + 15) / 16 * 16;
    std::vector<char> stage(image_bytes + dense_bytes, 0);
    {
        std::vector<unsigned> d(m * k, 0u);
        for (size_t j = 0; j < m; ++j)
            for (size_t sy = 0; sy < k; ++sy) {
                d[j * k + sy] = weights[j * wstride + sy];
                stage[image_bytes + j * k + sy] = (char)weights[j * wstride + sy];
            }
        prefilter2_pack_image(d.data(), (int)m, reinterpret_cast<unsigned *>(stage.data()));
    }
    // records: room for 2^-12 of the range's cells, at least 16 k, at most 256 k -- a longer list means the level is still
    // low for this range, and the window walk is the better shape for it
    const unsigned long long cap = std::min<unsigned long long>(std::max<unsigned long long>(cells >> 12, 1ull << 14), 1ull << 18);
    const unsigned long long ccap = 2 * cap;
    const size_t off_tables = 256, off_hits = off_tables + (image_bytes + dense_bytes + 255) / 256 * 256;
    const size_t off_cands = off_hits + cap * sizeof(HitRecord);
    LM_TRY(ctx->scratch.reserve(off_cands + ccap * sizeof(Candidate)));
    char *base = static_cast<char *>(ctx->scratch.ptr);
    FusedOut fo{};
    fo.hit_count = reinterpret_cast<unsigned long long *>(base);
    fo.cand_count = fo.hit_count + 1;
    fo.hits = reinterpret_cast<HitRecord *>(base + off_hits);
    fo.hit_capacity = cap;
    fo.cands = reinterpret_cast<Candidate *>(base + off_cands);
    fo.cand_capacity = ccap;
    LM_HIP_TRY(hipMemsetAsync(base, 0, 16, ctx->stream));
    LM_HIP_TRY(hipMemcpyAsync(base + off_tables, stage.data(), stage.size(), hipMemcpyHostToDevice, ctx->stream));  // (pageable: copied out before the call returns)
    const unsigned *d_image = reinterpret_cast<const unsigned *>(base + off_tables);
    const uint8_t *d_dw = reinterpret_cast<const uint8_t *>(base + off_tables + image_bytes);
    LM_HIP_TRY(scan(plan.grid, plan.lds, ctx->stream, seq->d_data, d_image, 5, first_row, row_end, plan.T, plan.nstreams, level, fo));
    hipLaunchKernelGGL(scanmax_gate, dim3((unsigned)ctx->num_cus * 8), dim3(kBlock), 0, ctx->stream, fo, seq->d_data, pssm->d_dense, d_dw,
                       (unsigned)m, (unsigned)k, level, (unsigned long long)first_row);
    LM_HIP_TRY(hipGetLastError());
    // the counters and the head of the (unordered) list in one read-back: a list of a few thousand records is put in order by
    // the host while a longer one goes through the device's ordering passes (hits.hip) -- one synchronisation instead of two
    // and no ordering launches for the ranges behind the first, which hold hundreds of records
    constexpr unsigned long long kHostSort = 4096;
    char *pin = static_cast<char *>(ctx->pinned);
    unsigned long long *h_counts = reinterpret_cast<unsigned long long *>(pin);
    HitRecord *h_recs = reinterpret_cast<HitRecord *>(pin + 256);
    static_assert(256 + kHostSort * sizeof(HitRecord) <= kPinnedBytes / 2, "the head of the list must fit the pinned block");
    LM_HIP_TRY(hipMemcpyAsync(h_counts, base, 16, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP_TRY(hipMemcpyAsync(h_recs, fo.hits, std::min(cap, kHostSort) * sizeof(HitRecord), hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const unsigned long long count = h_counts[0], ncand = h_counts[1];
    if (count > cap || ncand > ccap)
        return LM_HIP_OK;  // a long list after all: the window walk
    *route = 0;
    if (count == 0)
        return LM_HIP_OK;
    HitOutput out;
    std::vector<std::pair<unsigned long long, float>> sorted;
    if (count <= kHostSort) {
        sorted.reserve(count);
        for (unsigned long long i = 0; i < count; ++i)
            sorted.emplace_back(h_recs[i].key, h_recs[i].value);
        std::sort(sorted.begin(), sorted.end(), [](const auto &a, const auto &b) { return a.first < b.first; });  // (cells are unique)
    } else {
        int status = 0;
        unsigned long long counts[2];
        LM_TRY(order_hits(ctx, fo.hits, fo.hit_count, count, cap, ccap, count, 1, ((unsigned long long)rows * cols) << 8, 1, cols, &out,
                          &status, counts));
    }
    // the walk (scan.rs:229-242), over records in row-major cell order
    const unsigned long long total = (unsigned long long)rows * cols;
    int err = LM_HIP_OK;
    const size_t nrec = sorted.empty() ? out.total : sorted.size();
    for (size_t i = 0; i < nrec; ++i) {
        const unsigned long long key = sorted.empty() ? out.hits[i].position : sorted[i].first, flat = key >> 8;
        const unsigned u8 = (unsigned)(key & 255u);
        if (u8 < st->level)
            continue;
        const unsigned long long r = flat / cols, c = flat % cols, index = c * rows + r;  // scan.rs:231
        const float x = sorted.empty() ? out.hits[i].score : sorted[i].second;
        if (index + m > total) {  // score_position would index column C: the reference panics (seq.rs:433-442)
            err = fail(LM_HIP_ERR_BAD_ARGS,
                       "Scanner::max: the window of candidate position %llu (+ %zu rows) leaves the striped matrix; the "
                       "reference panics here (seq.rs:433-442)", index, m);
            break;
        }
        if (!st->have) {  // the first candidate is taken as it is; the level stays the scaled threshold (scan.rs:241)
            st->have = true;
            st->score = x;
            st->position = index;
        } else if (x > st->score || (x == st->score && index > st->position)) {
            st->score = x;
            st->position = index;
            st->level = u8;  // scan.rs:238
        }
    }
    out.release();
    if (getenv("LM_HIP_TRACE"))
        fprintf(stderr, "[lm_hip] Scanner::max by list: rows %zu ... %zu at level %u: %llu candidate pieces, %llu records -> level %u\n",
                first_row, row_end, level, ncand, count, st->level);
    return err;
}

// `weights`: the DiscreteMatrix's u8 weights on the HOST (M x wstride), `level` / `have` / `position` / `score`: the
// walk's state on entry (a fresh scanner: level = dm.scale(threshold), no hit), `first_row`: where the walk starts.
int launch_scan_max(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, const lm_hip_seq *seq, const uint8_t *weights, size_t wstride,
                    bool saturate, unsigned level, bool have, unsigned long long position, float score, size_t first_row,
                    int *found, unsigned long long *best_position, float *best_score)
{
    const size_t rows = seq->rows, cols = seq->cols, m = pssm->m;
    *found = have ? 1 : 0;
    *best_position = position;
    *best_score = score;
    if (first_row >= rows || seq->length < m)
        return LM_HIP_OK;
    if (saturate && (unsigned long long)(rows - first_row) * cols >= (1ull << 20)) {
        // Two row ranges (more on very long sequences: each 128 times the one before), each scanned at the level the walk has
        // reached: the level rises fast at first -- the first 1/128 of a 1 Gbp sequence lifts it from the scaled threshold
        // (2e-4 of the cells at p = 1e-5: 211 k records per Gbp) to the top of the u8 range -- so the range behind it
        // yields hundreds of records instead of hundreds of thousands.
        WalkState ws{level, have, position, score};
        size_t r = first_row;
        size_t step = std::max<size_t>((rows - first_row) / 128, (size_t)1 << 15);
        bool walked_all = true;
        while (r < rows) {
            const size_t r_end = rows - r <= step + step / 2 ? rows : r + step;
            int route = 1;
            LM_TRY(scan_max_by_list(ctx, pssm, seq, weights, wstride, r, r_end, &ws, &route));
            if (route != 0) {
                walked_all = false;
                break;
            }
            r = r_end;
            step *= 128;
        }
        // the state so far; the window walk below goes on from row r where a list turned out long (or never applied)
        level = ws.level;
        have = ws.have;
        position = ws.position;
        score = ws.score;
        first_row = r;
        *found = have ? 1 : 0;
        *best_position = position;
        *best_score = score;
        if (walked_all) {
            ctx->last_kernel = "score_c32_prefilter2+scanmax_gate";
            return LM_HIP_OK;
        }
    }
    // window buffer: the u8 scores, dense rows (stride = cols); up to 256 M cells per window (64 M: 8 % slower per Gbp), matrices of very many columns
    // keep at least 64 rows.  (The state block holds 64-bit words the search updates atomically: 256-byte aligned.)
    const size_t max_rows = std::max<size_t>(((size_t)256 << 20) / cols, 64);
    const size_t wrows_cap = std::min(max_rows, rows - first_row);
    const size_t d_bytes = (wrows_cap * cols + 255) / 256 * 256;
    LM_TRY(ctx->scan_buf.reserve(d_bytes + 256));
    uint8_t *d_d = static_cast<uint8_t *>(ctx->scan_buf.ptr);
    ScanMaxState *d_st = reinterpret_cast<ScanMaxState *>(d_d + d_bytes);
    ScanMaxState *h_st = static_cast<ScanMaxState *>(ctx->pinned);
    ScanMaxState init{};
    init.index = position;
    init.cursor = 0;
    init.found = ~0ull;
    init.score = score;
    init.level = level;
    init.have = have ? 1 : 0;
    h_st[1] = init;  // staged through the pinned block ([0] receives the kernel's copies)
    h_st[0] = init;
    LM_HIP_TRY(hipMemcpyAsync(d_st, &h_st[1], sizeof(ScanMaxState), hipMemcpyHostToDevice, ctx->stream));
    const ScanMaxSource src{seq->d_data, (unsigned long long)seq->stride, pssm->d_dense, (unsigned)pssm->k};
    const unsigned grid = (unsigned)ctx->num_cus * 8;
    struct Window {
        size_t r, rb;
    };
    auto score_window = [&](const Window &wd) {
        DiscreteArgs da{weights, m, wstride, pssm->k, seq->d_data, seq->stride, cols, wd.r, wd.rb, d_d, cols, saturate};
        return launch_score_u8(ctx, da);
    };
    // `rounds` searches (each followed by its update) over the window in the buffer; the last one closes a pipelined window
    auto walk_rounds = [&](const Window &wd, int rounds, bool fresh, bool pipelined, int id) {
        const unsigned long long ncells = (unsigned long long)(wd.rb - wd.r) * cols;
        for (int rep = 0; rep < rounds; ++rep) {
            hipLaunchKernelGGL(scanmax_find, dim3(grid), dim3(kBlock), 0, ctx->stream, d_d, src, ncells, (unsigned)cols,
                               (unsigned long long)wd.r, (unsigned long long)rows, (unsigned)m, fresh && rep == 0 ? 1 : 0, d_st);
            hipLaunchKernelGGL(scanmax_apply, dim3(1), dim3(64), 0, ctx->stream, d_d, src, ncells, (unsigned)cols,
                               (unsigned long long)wd.r, (unsigned long long)rows, (unsigned)m,
                               pipelined && rep == rounds - 1 ? 1 : 0, id, d_st, h_st);
        }
    };
    auto panicked = [&]() {
        return fail(LM_HIP_ERR_BAD_ARGS,
                    "Scanner::max: the window of candidate position %llu (+ %zu rows) leaves the striped matrix; the "
                    "reference panics here (seq.rs:433-442)", h_st[0].index, m);
    };
    // the window in the buffer, round by round under the host's eyes until its walk is complete
    auto finish_window = [&](const Window &wd, bool fresh) -> int {
        for (;; fresh = false) {
            walk_rounds(wd, 3, fresh, false, 0);  // a few rounds per synchronisation
            LM_HIP_TRY(hipGetLastError());
            LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (h_st[0].err)
                return panicked();
            if (!h_st[0].more)
                return LM_HIP_OK;
        }
    };
    constexpr size_t kWatchedRows = 1u << 16;  // the walk updates often at first: windows below this size are watched
    constexpr int kBatch = 8, kRounds = 3;     // pipelined windows per synchronisation, searches per pipelined window
    size_t w = std::min<size_t>(4096, wrows_cap);
    size_t r = first_row;
    bool resumed = false;
    while (r < rows) {
        if (w < kWatchedRows) {
            const Window wd{r, std::min(rows, r + w)};
            LM_TRY(score_window(wd));
            LM_TRY(finish_window(wd, true));
            r = wd.rb;
            w = std::min(2 * w, wrows_cap);
            continue;
        }
        Window batch[kBatch];
        size_t wsize[kBatch];
        int nb = 0;
        for (size_t q = r; q < rows && nb < kBatch; ++nb) {
            batch[nb] = Window{q, std::min(rows, q + w)};
            wsize[nb] = w;
            LM_TRY(score_window(batch[nb]));
            walk_rounds(batch[nb], kRounds, true, true, nb);
            q = batch[nb].rb;
            w = std::min(2 * w, wrows_cap);
        }
        LM_HIP_TRY(hipGetLastError());
        LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (h_st[0].err)
            return panicked();
        if (h_st[0].stall) {
            // window `id` ran out of rounds; what was enqueued behind it did nothing.  Its u8 scores again, then on
            // from its cursor; the windows behind it are enqueued anew
            const int id = h_st[0].stall - 1;
            resumed = true;
            LM_TRY(score_window(batch[id]));
            hipLaunchKernelGGL(scanmax_resume, dim3(1), dim3(1), 0, ctx->stream, d_st);
            LM_TRY(finish_window(batch[id], false));
            r = batch[id].rb;
            w = std::min(2 * wsize[id], wrows_cap);
        } else {
            r = batch[nb - 1].rb;
        }
    }
    ctx->last_kernel = resumed ? "scanmax_find (a batched window resumed)" : "scanmax_find";
    *found = h_st[0].have;
    *best_position = h_st[0].index;
    *best_score = h_st[0].score;
    return LM_HIP_OK;
}

}  // namespace lm
