// score_threshold.hip -- fused score + Threshold (pli/mod.rs:210-221 over scores that are never written): discrete
// prefilter scans -> candidates -> exact re-scoring -> ordered hit lists; batches of independent jobs.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "score_launch.hpp"

namespace lm {

// Appends every cell >= t of a contiguous chunk (stride == cols) to the hit list (rows relative to the job): float4
// reads when the rows are whole 16-byte pieces (cols % 4 == 0), one cell per thread otherwise.
__global__ __launch_bounds__(kBlock) void chunk_emit_hits(const float *__restrict__ s, const unsigned long long ncells,
                                                          const unsigned long long row_base, const unsigned cols,
                                                          const FusedOut fo)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(s);
    const unsigned long long n4 = ncells / 4;
    const float t = fo.threshold;
    if (cols % 4 != 0) {
        for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < ncells;
             i += (unsigned long long)gridDim.x * kBlock) {
            const float x = s[i];
            if (x >= t)
                record_hit(fo, row_base + i / cols, (unsigned)(i % cols), cols, x);
        }
        return;
    }
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < n4;
         i += (unsigned long long)gridDim.x * kBlock) {
        const f32x4 x = __builtin_nontemporal_load(&s4[i]);
        if (!(x.x >= t || x.y >= t || x.z >= t || x.w >= t))
            continue;
        const unsigned per_row = cols / 4;
        const unsigned long long row = row_base + i / per_row;
        const unsigned col = (unsigned)(i % per_row) * 4;
        if (x.x >= t) record_hit(fo, row, col, cols, x.x);
        if (x.y >= t) record_hit(fo, row, col + 1, cols, x.y);
        if (x.z >= t) record_hit(fo, row, col + 2, cols, x.z);
        if (x.w >= t) record_hit(fo, row, col + 3, cols, x.w);
    }
}

// ---- fused threshold --------------------------------------------------------------------

// Hits are staged in LDS and flushed with ONE global atomicAdd per ~1000 records: a
// million per-hit atomics on the single list counter would serialise in L2 (measured:
// 6 ms per 1e6 hits).
constexpr int kRescoreBlock = 1024;  // big workgroups: one global atomic per workgroup and flush, and
constexpr int kHitStage = 4096;      // 2 048 small workgroups' atomics on one counter cost 12 us per launch
constexpr int kRescoreBlocksPerCu = 2;
constexpr int kRescoreCheck = 2;     // rounds between two flush decisions

// LDS_TAB: the launch has few jobs (one PSSM, or both strands) whose dense tables fit
// kRescoreTabFloats: they are staged in LDS once per workgroup, and the M weight lookups of a
// row become LDS gathers instead of M dependent global gathers through the vector cache
// (20 per lane and piece at M = 20: the texture addresser, not arithmetic, bounded the kernel
// at dense hit rates -- 0.34 ms for 1.7 M pieces at p = 1e-3).  Many-motif batches keep the
// tables in global memory (2 346 JASPAR tables = 1 MB).
constexpr int kRescoreTabFloats = 2048;
constexpr int kRescoreTabJobs = 8;

// `bucket_counts` (ShortOrder; one job): every record stored also bumps the count of its bucket, key >> bucket_shift --
// the histogram pass of the ordering, without a launch of its own.
template <bool LDS_TAB, int BLOCK = kRescoreBlock, int STAGE = kHitStage>
__global__ __launch_bounds__(BLOCK) void rescore_candidates(const RescoreJob *__restrict__ jobs,
                                                             const FusedOut fo, const unsigned njobs,
                                                             unsigned *__restrict__ bucket_counts, const int bucket_shift,
                                                             const RescoreJob job0)
{
    const bool by_value = bucket_counts != nullptr;  // ShortOrder: the one job is `job0`, `jobs` is not read
    __shared__ HitRecord stage[STAGE];
    __shared__ unsigned nstage;
    __shared__ unsigned long long gbase;
    __shared__ float tab[LDS_TAB ? kRescoreTabFloats : 1];
    __shared__ unsigned tab_off[LDS_TAB ? kRescoreTabJobs : 1];
    if (LDS_TAB) {
        unsigned off = 0;
        for (unsigned j = 0; j < njobs; ++j) {  // block-uniform
            const unsigned nf = by_value ? job0.m * job0.k : jobs[j].m * jobs[j].k;
            const float *dense = by_value ? job0.dense : jobs[j].dense;
            for (unsigned i = threadIdx.x; i < nf; i += BLOCK)
                tab[off + i] = dense[i];
            if (threadIdx.x == 0)
                tab_off[j] = off;
            off += nf;
        }
    }
    if (threadIdx.x == 0)
        nstage = 0;
    __syncthreads();
    unsigned long long n = *fo.cand_count;
    if (n > fo.cand_capacity)
        n = fo.cand_capacity;  // overflow: the launcher re-runs the batch with more room
    const unsigned lane = threadIdx.x & 31;
    const unsigned long long stride = (unsigned long long)gridDim.x * (BLOCK / 32);
    auto flush = [&]() {  // block-uniform
        __syncthreads();
        const unsigned cnt = nstage;
        if (threadIdx.x == 0 && cnt)
            gbase = atomicAdd(fo.hit_count, (unsigned long long)cnt);
        __syncthreads();
        for (unsigned i = threadIdx.x; i < cnt; i += BLOCK)
            if (gbase + i < fo.hit_capacity) {
                fo.hits[gbase + i] = stage[i];
                if (bucket_counts)
                    atomicAdd(&bucket_counts[(stage[i].key & ((1ull << 40) - 1)) >> bucket_shift], 1u);
            }
        __syncthreads();
        if (threadIdx.x == 0)
            nstage = 0;
        __syncthreads();
    };
    // block-uniform trip count: one candidate piece per half-wave per round
    constexpr unsigned kWin = (32 + kMaxPairM - 1 + 31) / 32 * 32;  // a piece of <= 32 rows of a motif of <= kMaxPairM rows
    __shared__ uint8_t window[BLOCK / 32][kWin];  // symbols of rows r0 .. r0 + nrows + M - 2
    unsigned round = 0;
    for (unsigned long long c0 = (unsigned long long)blockIdx.x * (BLOCK / 32); c0 < n; c0 += stride) {
        const unsigned long long c = c0 + (threadIdx.x >> 5);
        if (c < n) {
            const Candidate cd = fo.cands[c];
            const RescoreJob jb = by_value ? job0 : jobs[cd.key >> 40];
            const unsigned long long r0 = cd.key & ((1ull << 40) - 1);
            // every symbol of the piece's column window is loaded once (<= 3 loads per
            // lane, all in flight together) and shared through LDS
            uint8_t *win = window[threadIdx.x >> 5];
            const unsigned nsym = cd.nrows + jb.m - 1;  // <= 32 + kMaxPairM - 1 (longer motifs never come here)
            const uint8_t *p = jb.seq + r0 * 32 + cd.col;
            uint8_t s0 = 0, s1 = 0, s2 = 0;
            if (lane < nsym)
                s0 = p[(unsigned long long)lane * 32];
            if (lane + 32 < nsym)
                s1 = p[(unsigned long long)(lane + 32) * 32];
            if (lane + 64 < nsym)
                s2 = p[(unsigned long long)(lane + 64) * 32];
            win[lane] = s0;
            win[lane + 32] = s1;
            win[lane + 64] = s2;
            for (unsigned o = lane + 96; o < nsym; o += 32)  // motifs beyond 65 rows
                win[o] = p[(unsigned long long)o * 32];
            // (a half-wave runs in lockstep inside its wavefront: no barrier needed
            // between the LDS writes above and the reads below)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < cd.nrows) {
                const float *dense = LDS_TAB ? tab + tab_off[cd.key >> 40] : jb.dense;
                float sc = 0.0f;
                unsigned j = 0;
                for (; j + 8 <= jb.m; j += 8) {
                    float w[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        w[q] = dense[(j + q) * jb.k + win[lane + j + q]];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        sc = sc + w[q];
                }
                for (; j < jb.m; ++j)
                    sc = sc + dense[j * jb.k + win[lane + j]];
                if (sc >= jb.threshold) {
                    const unsigned long long row = r0 + lane;
                    HitRecord r;
                    r.key = (cd.key & ~((1ull << 40) - 1)) |
                            (jb.key_rows ? cd.col * jb.key_rows + row : row * 32ull + cd.col);
                    r.value = sc;
                    r.pad = 0;
                    stage[atomicAdd(&nstage, 1u)] = r;  // <= BLOCK records per round
                }
            }
        }
        // Every kRescoreCheck rounds the workgroup agrees on whether to flush: a round stages at
        // most kRescoreBlock records, so the stage must have room for kRescoreCheck more rounds.
        // Between checks the wavefronts run free and overlap their load chains.  The count is
        // read between two barriers: a wavefront that raced ahead into the next round must not
        // be able to change what a slower one reads (the decision has to be uniform).
        if (++round % kRescoreCheck == 0) {
            __syncthreads();
            const unsigned cnt = nstage;
            __syncthreads();
            if (cnt > STAGE - kRescoreCheck * BLOCK)
                flush();
        }
    }
    flush();
}

int launch_rescore(lm_hip_ctx *ctx, hipStream_t st, const RescoreJob *d_jobs, const FusedOut &fo,
                          const RescoreJob *host_jobs, size_t n, const ShortOrder *so)
{
    unsigned *counts = so && so->on ? so->counts : nullptr;
    const int shift = so && so->on ? so->shift : 0;
    size_t floats = 0;
    for (size_t i = 0; i < n && i <= (size_t)kRescoreTabJobs; ++i)
        floats += (size_t)host_jobs[i].m * host_jobs[i].k;
    const dim3 grid((unsigned)ctx->num_cus * kRescoreBlocksPerCu), block(kRescoreBlock);
    // (workgroups of 256 threads with a 1 024-record stage for the short lists of single jobs -- more, smaller workgroups on
    //  the call's critical path -- were measured in round 6: 38 instead of 19 us for 24 k pieces at 1 Gbp, 10.6 instead of 8.0
    //  at 200 Mres: four times the global atomics on the list's counter, as the constants above already say)
    if (n <= (size_t)kRescoreTabJobs && floats <= (size_t)kRescoreTabFloats)
        hipLaunchKernelGGL(rescore_candidates<true>, grid, block, 0, st, d_jobs, fo, (unsigned)n, counts, shift, host_jobs[0]);
    else
        hipLaunchKernelGGL(rescore_candidates<false>, grid, block, 0, st, d_jobs, fo, (unsigned)n, counts, shift, host_jobs[0]);
    LM_HIP_TRY(hipGetLastError());
    return LM_HIP_OK;
}

// Fused score+threshold of `n` jobs.  The C = 32 kernels flag candidate row ranges
// (discrete prefilter or exact f32 compare), `rescore_candidates` turns them into
// (key, score) hit records, key = (job << 40) | row-major cell index or sequence
// position; hits.hip then orders the list on the device (the reference's row-major
// push order, pli/mod.rs:212-218, or ascending position) and the result is copied
// straight into the arrays handed to the caller.  One synchronisation when the ordering can be
// enqueued behind the scans (hits.hip: speculative form), two when the count is read first.
// Equal-length DNA motifs of a batch share passes over the sequence (score_c32_prefilter2_multi).
int launch_score_threshold_batch(lm_hip_ctx *ctx, const ScoreArgs *jobs, const float *ts, size_t n,
                                 HitKeys keys, HitOutput *out)
{
    out->job_start.assign(n + 1, 0);
    out->total = 0;
    if (n == 0)
        return LM_HIP_OK;
    if (n > (1u << 23))
        return fail(LM_HIP_ERR_CAPACITY, "fused threshold: at most 2^23 jobs per batch");
    unsigned long long max_low = 0, total_cells = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned long long rows = jobs[i].row_end - jobs[i].row_begin;
        if (keys == HitKeys::Position && jobs[i].row_begin != 0)
            return fail(LM_HIP_ERR_BAD_ARGS, "fused threshold: position keys need row_begin == 0");
        if (jobs[i].cols != jobs[0].cols)
            return fail(LM_HIP_ERR_BAD_ARGS, "fused threshold: the jobs of a batch must share `cols`");
        max_low = std::max(max_low, rows * jobs[i].cols);
        total_cells += rows * jobs[i].cols;
    }
    // Hit-list capacity: room for a 1.2e-4 hit rate over the whole batch (the CLI's
    // default p-value is 1e-5, main.rs:487), at least what the previous call on this
    // context needed, never more than every cell; twice that many candidate pieces.
    // An overflow of either list re-runs the batch with the exact counts.
    unsigned long long cap = std::max<unsigned long long>(total_cells / 8192, 1 << 16);
    cap = std::max(cap, ctx->last_hit_count + ctx->last_hit_count / 2);
    cap = std::min(cap, total_cells + 64);
    unsigned long long ccap = std::max(2 * cap, ctx->last_cand_count + ctx->last_cand_count / 2);
    std::vector<RescoreJob> rjobs(n);
    const auto t_begin = std::chrono::steady_clock::now();
    // which kernel scores each job: the discrete prefilter (score_prefilter.hpp) when a
    // sound one exists and the threshold maps into its 16-bit range, the exact f32
    // kernel otherwise, the generic kernel for shapes the C = 32 kernels do not cover
    std::vector<unsigned> tds(n, 0);
    const std::vector<JobGroup> groups = group_jobs(ctx, jobs, n, [&](size_t i) {
        const ScoreArgs &a = jobs[i];
        // A threshold above the best k-mer's score selects nothing, whatever the sequence: such a job is not
        // scanned at all.  (At the CLI's p = 1e-5 that is every motif too short to reach the p-value -- 1 042 of
        // the 2 346 JASPAR matrices, all of length <= 8 -- which the reference scans like any other.)
        if (ctx->skip_unreachable && a.pssm->has_prefilter && a.pssm->m >= 1 && ts[i] > best_kmer_score(a.pssm))
            return (int)KIND_SKIP;
        if (a.pssm->has_prefilter && ctx->use_prefilter && std::isfinite(ts[i])) {
            const double scaled = std::floor(((double)ts[i] - a.pssm->pre_offset) / a.pssm->pre_factor) -
                                  std::ceil(a.pssm->pre_emax / a.pssm->pre_factor) - 1.0;
            if (scaled >= 1.0) {
                tds[i] = scaled > 65535.0 ? 65535u : (unsigned)scaled;
                if (ctx->pair_prefilter && plan_c32(ctx, a, false, 2).ok)
                    return (int)KIND_PREFILTER2;  // DNA: two symbols per lookup
                if (plan_c32(ctx, a, false, 1).ok)
                    return (int)KIND_PREFILTER;
            }
        }
        return plan_c32(ctx, a, false).ok ? (int)KIND_EXACT : chunked_ok(ctx, a) ? (int)KIND_CHUNKED : (int)KIND_GENERIC;
    });
    const unsigned long long key_rows =
        keys == HitKeys::Position ? (unsigned long long)(jobs[0].row_end - jobs[0].row_begin) : 0;
    // job table in launch order: the jobs of a group are contiguous from group_pos[g] on.  Groups
    // of the pair scan with several jobs run `per_pass[g]` motifs per pass
    // (score_c32_prefilter2_multi) and are padded to a multiple of that with entries that flag nothing.
    std::vector<BatchParams> bparams;
    std::vector<size_t> group_pos(groups.size());
    std::vector<int> per_pass(groups.size(), 1);
    bparams.reserve(n + 4 * groups.size());
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const JobGroup &g = groups[gi];
        group_pos[gi] = bparams.size();
        // several motifs per pass (score_c32_prefilter2_multi): every matrix of the group needs its table in that kernel's layout
        bool multi = g.kind == KIND_PREFILTER2 && ctx->multi_motif && n > 1 && g.idx.size() >= 2 && jobs[g.idx[0]].pssm->k == 5 &&
                     score_c32_prefilter2_multi_lookup((int)jobs[g.idx[0]].pssm->m);
        for (size_t i : g.idx)
            multi = multi && jobs[i].pssm->d_image2_multi != nullptr;
        for (size_t i : g.idx) {
            const ScoreArgs &a = jobs[i];
            if (keys == HitKeys::Position && a.row_end - a.row_begin != key_rows)
                return fail(LM_HIP_ERR_BAD_ARGS, "fused threshold: position keys need equal row ranges");
            bparams.push_back(BatchParams{multi                      ? (const void *)a.pssm->d_image2_multi
                                          : g.kind == KIND_PREFILTER2 ? (const void *)a.pssm->d_image2
                                          : g.kind == KIND_PREFILTER ? (const void *)a.pssm->d_image
                                          : g.kind == KIND_EXACT     ? (const void *)exact_motif(a.pssm, a.d_seq).table
                                                                     : (const void *)a.pssm->d_table,
                                          nullptr, ts[i], tds[i], (unsigned long long)i << 40});
            rjobs[i] = RescoreJob{a.d_seq + a.row_begin * a.seq_stride, a.pssm->d_dense,
                                  (unsigned)a.pssm->m, (unsigned)a.pssm->k, ts[i], 0, key_rows};
        }
        const int m = (int)jobs[g.idx[0]].pssm->m;
        if (multi) {
            per_pass[gi] = prefilter2_multi(m);
            while ((bparams.size() - group_pos[gi]) % per_pass[gi]) {
                BatchParams pad = bparams.back();
                pad.td = 0xffffffffu;  // no sum reaches it: the padding job flags nothing
                bparams.push_back(pad);
            }
        }
    }
    const size_t nbp = bparams.size();
    // One pair scan of a motif of M = 20, 24, ... 36 rows: its first M - 1 rows are looked up and the last row is credited with
    // its best weight (lm_hip_pssm::d_image2_drop) -- table rows of one 16-byte read less, up to four times the candidate
    // pieces, identical hits (every candidate is re-scored over all M rows).  Not when the last row carries a large part of
    // the threshold: the scan would flag too much.
    C32Plan drop_plan;
    if (n == 1 && groups.size() == 1 && groups[0].kind == KIND_PREFILTER2 && ctx->drop_last && jobs[0].pssm->d_image2_drop &&
        (unsigned long long)jobs[0].pssm->drop_dmax * 4 <= tds[0] &&
        score_c32_prefilter2_lookup((int)jobs[0].pssm->m - 1, (int)jobs[0].pssm->k))
        drop_plan = plan_c32(ctx, MotifShape{jobs[0].pssm->m - 1, jobs[0].pssm->k, true}, jobs[0], false, 2, 1);
    const bool drop_last_form = drop_plan.ok;
    ctx->last_scan_rows = ctx->last_scan_lds_bytes = 0;
    if (n == 1 && groups.size() == 1) {
        const size_t scanned = jobs[0].pssm->m - (drop_last_form ? 1 : 0);
        const size_t em = groups[0].kind == KIND_EXACT ? exact_motif(jobs[0].pssm, jobs[0].d_seq).m : scanned;
        ctx->last_scan_lds_bytes = scan_lds_bytes(groups[0].kind, em, jobs[0].pssm->k);
        ctx->last_scan_rows = ctx->last_scan_lds_bytes ? (unsigned)scanned : 0u;
    }
    for (int attempt = 0; attempt < 3; ++attempt) {
        // layout: [hit count u64][candidate count u64][jobs][batch][HitRecord x cap][Candidate x ccap];
        // the head -- zeroed counters and the two job tables -- is assembled in the upper half of the
        // pinned buffer and reaches the device as ONE copy
        // (the counters get 256 bytes of their own: the scans' atomics on them would otherwise
        // fight with every read of a job table entry in the same cache line -- measured +30 % on the
        // re-scoring kernel at 10^6 hits)
        const size_t off_jobs = 256;
        const size_t off_batch = off_jobs + (n * sizeof(RescoreJob) + 15) / 16 * 16;
        const size_t off_hits = off_batch + (nbp * sizeof(BatchParams) + 15) / 16 * 16;
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
        fo.key_rows = key_rows;
        RescoreJob *d_jobs = reinterpret_cast<RescoreJob *>(base + off_jobs);
        BatchParams *d_bparams = reinterpret_cast<BatchParams *>(base + off_batch);
        // one job whose hits all pass through the re-scoring kernel, list expected short: the kernel counts them per
        // bucket as well, and the counters live in the context -- nothing to copy in before the scan (hits.hip, ShortOrder)
        ShortOrder so;
        const unsigned long long expected = ctx->last_hit_count + ctx->last_hit_count / 4;
        if (attempt == 0 && ctx->speculate_order && ctx->short_order && n == 1 && groups.size() == 1 &&
            (groups[0].kind == KIND_PREFILTER || groups[0].kind == KIND_PREFILTER2 || groups[0].kind == KIND_EXACT))
            LM_TRY(short_order_begin(ctx, expected, n, max_low, &so));
        if (so.on) {
            fo.hit_count = so.counters;
            fo.cand_count = so.counters + 1;
        } else if (off_hits <= kPinnedBytes / 2) {
            char *head = static_cast<char *>(ctx->pinned) + kPinnedBytes / 2;
            memset(head, 0, off_jobs);
            memcpy(head + off_jobs, rjobs.data(), n * sizeof(RescoreJob));
            if (n > 1)
                memcpy(head + off_batch, bparams.data(), nbp * sizeof(BatchParams));
            LM_HIP_TRY(hipMemcpyAsync(base, head, n > 1 ? off_hits : off_batch, hipMemcpyHostToDevice, ctx->stream));
        } else {
            LM_HIP_TRY(hipMemsetAsync(base, 0, 16, ctx->stream));
            LM_HIP_TRY(hipMemcpyAsync(d_jobs, rjobs.data(), n * sizeof(RescoreJob), hipMemcpyHostToDevice,
                                      ctx->stream));
            LM_HIP_TRY(hipMemcpyAsync(d_bparams, bparams.data(), nbp * sizeof(BatchParams), hipMemcpyHostToDevice,
                                      ctx->stream));
        }
        const bool two_streams = groups.size() > 1;
        scan_timer_begin(ctx, ctx->stream);
        if (two_streams)
            LM_TRY(batch_fork(ctx));
        bool any_candidates = false;
        size_t launch = 0;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const JobGroup &g = groups[gi];
            const size_t bp_pos = group_pos[gi];
            const size_t i = g.idx[0];
            const ScoreArgs &a = jobs[i];
            // (launch order and stream balance do not matter: most expensive length class first on the less loaded stream
            //  measured 14.59 against 14.61 ms of scans on the JASPAR batch, round 6 -- the scans are LDS-bound, not gap-bound)
            hipStream_t st = (two_streams && (launch++ & 1)) ? ctx->aux_stream : ctx->stream;
            fo.threshold = ts[i];
            fo.job_key = (unsigned long long)i << 40;
            fo.batch = (n > 1 && !kind_solo(g.kind)) ? d_bparams + bp_pos : nullptr;
            if (per_pass[gi] > 1) {  // several motifs of this length per pass over the sequence
                dim3 grid = g.plan.grid;
                grid.y = (unsigned)((g.idx.size() + per_pass[gi] - 1) / per_pass[gi]);
                ctx->last_kernel = "score_c32_prefilter2_multi";
                LM_HIP_TRY(score_c32_prefilter2_multi_lookup((int)a.pssm->m)(grid, st, a.d_seq, a.row_begin, a.row_end,
                                                                            g.plan.T, g.plan.nstreams, fo));
                any_candidates = true;
            } else if (g.kind == KIND_PREFILTER || g.kind == KIND_PREFILTER2) {
                const bool pairs = g.kind == KIND_PREFILTER2;
                ctx->last_kernel = pairs ? "score_c32_prefilter2" : block_scan(ctx, a) ? "score_c32_prefilter_blk" : "score_c32_prefilter";
                if (pairs && drop_last_form) {
                    PrefilterLauncher fn = score_c32_prefilter2_lookup((int)a.pssm->m - 1, (int)a.pssm->k);
                    LM_HIP_TRY(fn(drop_plan.grid, drop_plan.lds, st, a.d_seq, a.pssm->d_image2_drop, (int)a.pssm->k, a.row_begin,
                                  a.row_end, drop_plan.T, drop_plan.nstreams, tds[i] - a.pssm->drop_dmax, fo));
                } else {
                    PrefilterLauncher fn = pairs ? score_c32_prefilter2_lookup((int)a.pssm->m, (int)a.pssm->k)
                                                 : score_c32_prefilter_lookup((int)a.pssm->m, lds_wide((int)a.pssm->k), block_scan(ctx, a));
                    LM_HIP_TRY(fn(g.plan.grid, g.plan.lds, st, a.d_seq, pairs ? a.pssm->d_image2 : a.pssm->d_image,
                                  (int)a.pssm->k, a.row_begin, a.row_end, g.plan.T, g.plan.nstreams, tds[i], fo));
                }
                any_candidates = true;
            } else if (g.kind == KIND_EXACT) {
                const ExactMotif em = exact_motif(a.pssm, a.d_seq);
                FusedOut efo = fo;
                efo.lead_rows = em.lead;
                ScoreC32Launcher fn = score_c32_lookup((int)em.m, MODE_THRESHOLD, lds_wide((int)a.pssm->k));
                ctx->last_kernel = score_c32_name((int)em.m, MODE_THRESHOLD);
                LM_HIP_TRY(fn(g.plan.grid, g.plan.lds, st, a.d_seq, em.table, (int)a.pssm->k,
                              a.row_begin, a.row_end, g.plan.T, g.plan.nstreams, nullptr, efo));
                any_candidates = true;
            } else if (g.kind == KIND_CHUNKED) {  // appends hits directly, chunk by chunk, on ctx->stream
                const FusedOut cfo = fo;
                LM_TRY(for_each_scored_chunk(ctx, a, [&](const float *buf, unsigned long long c0, unsigned long long rows) {
                    const unsigned long long ncells = rows * a.cols;
                    const unsigned grid = (unsigned)std::max<unsigned long long>(
                        std::min<unsigned long long>((ncells / 4 + kBlock - 1) / kBlock, (unsigned long long)ctx->num_cus * 16), 1);
                    hipLaunchKernelGGL(chunk_emit_hits, dim3(grid), dim3(kBlock), 0, ctx->stream, buf, ncells, c0,
                                       (unsigned)a.cols, cfo);
                    LM_HIP_TRY(hipGetLastError());
                    return (int)LM_HIP_OK;
                }));
            } else {
                ctx->last_kernel = "score_generic<2>";  // appends hits directly
                const unsigned long long ncells =
                    (unsigned long long)(a.row_end - a.row_begin) * a.cols;
                LM_TRY(launch_generic<MODE_THRESHOLD>(ctx, a, fo, generic_grid(ctx, ncells), st));
            }
        }
        if (two_streams)
            LM_TRY(batch_join(ctx));
        scan_timer_end(ctx, ctx->stream);
        if (any_candidates || so.on) {
            LM_TRY(launch_rescore(ctx, ctx->stream, d_jobs, fo, rjobs.data(), n, &so));
            LM_HIP_TRY(hipGetLastError());
        }
        scan_timer_mark(ctx, ctx->stream, 2);
        const int emit = keys == HitKeys::Position ? 1 : 0;
        unsigned long long count = 0, ncand = 0;
        const auto t_scan = std::chrono::steady_clock::now();
        bool ordered = false;
        if (attempt == 0 && ctx->speculate_order) {
            // First try: enqueue the ordering right behind the scans, sized from the previous
            // call's count, and learn the counts from the same single synchronisation.
            int status = 0;
            unsigned long long counts[2] = {0, 0};
            LM_TRY(order_hits(ctx, fo.hits, fo.hit_count, ~0ull, cap, ccap, expected, n, max_low, emit, jobs[0].cols, out,
                              &status, counts, &so));
            count = counts[0];
            ncand = counts[1];
            ordered = status == 0;
            if (status == 1) {  // a list overflowed: grow and run the scans again
                ctx->last_cand_count = ncand;
                if (ncand > ccap) {
                    ccap = ncand + ncand / 8 + 64;
                    continue;
                }
                ctx->last_hit_count = count;
                cap = count + count / 8 + 64;
                continue;
            }
        } else {
            LM_HIP_TRY(hipMemcpyAsync(ctx->pinned, base, 16, hipMemcpyDeviceToHost, ctx->stream));
            LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
            count = static_cast<unsigned long long *>(ctx->pinned)[0];
            ncand = static_cast<unsigned long long *>(ctx->pinned)[1];
            if (ncand > ccap) {
                // the hit count of a truncated candidate list means nothing yet
                ctx->last_cand_count = ncand;
                ccap = ncand + ncand / 8 + 64;
                continue;
            }
            if (count > cap) {
                ctx->last_hit_count = count;
                cap = count + count / 8 + 64;
                continue;
            }
        }
        ctx->last_cand_count = ncand;
        ctx->last_hit_count = count;
        if (!ordered) {  // exact form: the count is known
            int status = 0;
            unsigned long long counts[2];
            // (after a short form that gave up, the list's counters have been cleared: their copy still holds them)
            LM_TRY(order_hits(ctx, fo.hits, so.on ? so.counters_copy : fo.hit_count, count, cap, ccap, count, n, max_low, emit, jobs[0].cols,
                              out, &status, counts));
        }
        scan_timer_read(ctx);
        if (ctx->last_phase_ms[0] >= 0)  // (time_scan) the host's share: everything of the call the events do not cover
            ctx->last_phase_ms[3] = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count() -
                                    (ctx->last_phase_ms[0] + std::max(ctx->last_phase_ms[1], 0.0f) + std::max(ctx->last_phase_ms[2], 0.0f));
        if (getenv("LM_HIP_TRACE")) {
            const auto t_end = std::chrono::steady_clock::now();
            fprintf(stderr, "[lm_hip] fused threshold: %zu jobs, %llu candidates, %llu hits; %s; scans enqueued in "
                            "%.3f ms, wait + ordering + read-back %.3f ms\n", n, ncand, count,
                    ordered ? "ordered behind the scans (one synchronisation)" : "ordered after reading the count",
                    std::chrono::duration<double, std::milli>(t_scan - t_begin).count(),
                    std::chrono::duration<double, std::milli>(t_end - t_scan).count());
        }
        return LM_HIP_OK;
    }
    return fail(LM_HIP_ERR_HIP, "fused threshold: hit list kept overflowing");
}

}  // namespace lm
