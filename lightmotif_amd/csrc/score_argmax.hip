// score_argmax.hip -- fused score + Maximum (pli/mod.rs:135-160 over scores that are never written): the exact kernels,
// the route through the discrete prefilter, and the suffix route of short motifs; batches of independent jobs.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "score_launch.hpp"

namespace lm {

// ---- fused argmax ---------------------------------------------------------------------

// Final reduction of per-block records; also applies the reference's "scores[0]
// is NaN -> (0,0)" rule (pli/mod.rs:142,146: nothing ever compares >= NaN).  In
// the fused path scores[0][0] does not exist in memory, so it is recomputed here.
__global__ __launch_bounds__(kBlock) void argmax_finalize(
    const ArgmaxRecord *__restrict__ blocks, const unsigned nblocks,
    const float *__restrict__ scores00,  // non-null: materialised scores
    const uint8_t *__restrict__ seq00, const unsigned long long seq_stride,
    const float *__restrict__ pssm, const int M, const int K, const int first_cell_rule,
    ArgmaxRecord *__restrict__ out)
{
    __shared__ float sm_v[kBlock / 64];
    __shared__ long long sm_i[kBlock / 64];
    float v = -INFINITY;
    long long i = -1;
    for (unsigned b = threadIdx.x; b < nblocks; b += kBlock)
        if (blocks[b].found)
            best_merge(v, i, blocks[b].value, blocks[b].index);
    best_block_reduce(v, i, sm_v, sm_i);
    if (threadIdx.x == 0 && first_cell_rule) {
        float first;
        if (scores00) {
            first = scores00[0];
        } else {
            first = 0.0f;
            for (int j = 0; j < M; ++j)
                first = first + pssm[j * K + seq00[j * seq_stride]];
        }
        if (first != first) {  // NaN
            v = first;
            i = 0;
        }
    }
    if (threadIdx.x == 0) {
        out->value = v;
        out->index = i;
        out->found = i >= 0;
    }
}

// One entry per job of a batch: where its block records are and how to recompute its
// scores[0][0] (first-cell rule).
struct FinalizeJob {
    const ArgmaxRecord *blocks;
    unsigned nblocks;
    int M, K;
    const uint8_t *seq00;
    unsigned long long seq_stride;
    const float *pssm;
};

// grid = number of jobs: block j reduces the block records of job j.
__global__ __launch_bounds__(kBlock) void argmax_finalize_batch(const FinalizeJob *__restrict__ jobs,
                                                                const int first_cell_rule,
                                                                ArgmaxRecord *__restrict__ out)
{
    __shared__ float sm_v[kBlock / 64];
    __shared__ long long sm_i[kBlock / 64];
    const FinalizeJob job = jobs[blockIdx.x];
    float v = -INFINITY;
    long long i = -1;
    for (unsigned b = threadIdx.x; b < job.nblocks; b += kBlock)
        if (job.blocks[b].found)
            best_merge(v, i, job.blocks[b].value, job.blocks[b].index);
    best_block_reduce(v, i, sm_v, sm_i);
    if (threadIdx.x == 0) {
        if (first_cell_rule) {
            float first = 0.0f;
            for (int j = 0; j < job.M; ++j)
                first = first + job.pssm[j * job.K + job.seq00[j * job.seq_stride]];
            if (first != first) {  // NaN (pli/mod.rs:142-146)
                v = first;
                i = 0;
            }
        }
        out[blockIdx.x].value = v;
        out[blockIdx.x].index = i;
        out[blockIdx.x].found = i >= 0;
    }
}

// Fused score+argmax of `n` independent jobs (one motif each): the n scoring kernels
// are enqueued back to back, each leaving per-workgroup records in its own region,
// then ONE finalize launch reduces every job and ONE synchronisation returns.
static int launch_score_argmax_exact(lm_hip_ctx *ctx, const ScoreArgs *jobs, size_t n,
                                     int first_cell_rule, ArgmaxRecord *out)
{
    if (n == 0)
        return LM_HIP_OK;
    // One small job off the C = 32 kernels (C = 1 is the Generic geometry of the reference's own bench, dna.rs:112-116):
    // ONE launch.  The tiled store kernel scores into the chunk buffer and leaves a (value, cell) record per workgroup
    // in the pinned block, folded here -- what score_into + argmax on handles do (27 us per iteration of configs[0]);
    // the chunked route below takes a store, a reduction and a finalize launch (41 us).
    // (C = 16 has an unrolled store kernel of its own, which leaves no records)
    if (n == 1 && ctx->host_fold && jobs[0].cols != 32 && jobs[0].cols != 16 && chunked_ok(ctx, jobs[0]) && !plan_c32(ctx, jobs[0], false).ok) {
        const ScoreArgs &a = jobs[0];
        const unsigned long long rows = a.row_end - a.row_begin;
        const unsigned nrec = tiled_records(ctx, a);
        if (nrec && rows <= chunk_rows_for(ctx, a) && rows * a.cols < (1ull << 32) && 4096 + ((size_t)nrec + 1) * 16 <= kPinnedBytes) {
            LM_TRY(ctx->chunk_scores.reserve(rows * a.cols * sizeof(float)));
            uint4 *records = reinterpret_cast<uint4 *>(static_cast<char *>(ctx->pinned) + 4096);
            for (size_t r = 0; r <= nrec; ++r)  // shared staging: stale bytes must not look like this launch's generation
                reinterpret_cast<volatile unsigned long long *>(records)[2 * r] = 0ull;
            const unsigned gen = ++ctx->fold_generation ? ctx->fold_generation : ++ctx->fold_generation;
            unsigned written = 0;
            ScoreArgs t = a;
            t.d_out = static_cast<float *>(ctx->chunk_scores.ptr);
            t.out_stride = a.cols;
            t.track_records = records;
            t.track_generation = gen;
            t.track_cap = (size_t)nrec + 1;
            t.track_nrec = &written;
            LM_TRY(launch_score_store(ctx, t));
            if (written) {
                ctx->last_kernel = "score_tiled+host_fold";
                return fold_host_records(ctx, records, written, gen, first_cell_rule != 0, out);
            }
            LM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // (no records after all: the general route, buffer quiescent)
        }
    }
    const std::vector<JobGroup> groups = group_jobs(ctx, jobs, n, [&](size_t i) {
        return plan_c32(ctx, jobs[i], false).ok ? KIND_EXACT : chunked_ok(ctx, jobs[i]) ? KIND_CHUNKED : KIND_GENERIC;
    });
    ctx->last_scan_rows = ctx->last_scan_lds_bytes = 0;
    if (n == 1 && groups.size() == 1 && groups[0].kind == KIND_EXACT) {
        ctx->last_scan_rows = (unsigned)jobs[0].pssm->m;
        ctx->last_scan_lds_bytes = scan_lds_bytes(KIND_EXACT, exact_motif(jobs[0].pssm, jobs[0].d_seq).m, jobs[0].pssm->k);
    }
    std::vector<unsigned> grids(n);
    size_t total_blocks = 0;
    for (const JobGroup &g : groups)
        for (size_t i : g.idx) {
            const unsigned long long ncells =
                (unsigned long long)(jobs[i].row_end - jobs[i].row_begin) * jobs[i].cols;
            grids[i] = g.kind == KIND_GENERIC   ? generic_grid(ctx, ncells).x
                       : g.kind == KIND_CHUNKED ? (unsigned)(chunk_count(ctx, jobs[i]) * chunk_argmax_grid(ctx))
                                                : g.plan.grid.x;
            total_blocks += grids[i];
        }
    const size_t off_blocks = sizeof(ArgmaxRecord) * n;
    const size_t off_jobs = off_blocks + sizeof(ArgmaxRecord) * total_blocks;
    LM_TRY(ctx->scratch.reserve(off_jobs + sizeof(FinalizeJob) * n));
    char *base = static_cast<char *>(ctx->scratch.ptr);
    ArgmaxRecord *results = reinterpret_cast<ArgmaxRecord *>(base);
    ArgmaxRecord *blocks = reinterpret_cast<ArgmaxRecord *>(base + off_blocks);
    FinalizeJob *d_jobs = reinterpret_cast<FinalizeJob *>(base + off_jobs);
    // Small batches skip both copies: the job table is written into the context's
    // pinned (device-visible) buffer, the finalize kernel reads it from there and
    // writes the results next to it, and the host reads them after the one
    // synchronisation.  A single short scan is launch-latency bound, so the two
    // staged copies were a third of its wall time.
    const size_t pin_jobs_off = (sizeof(ArgmaxRecord) * n + 63) / 64 * 64;
    const bool zero_copy = pin_jobs_off + sizeof(FinalizeJob) * n <= kPinnedBytes;
    std::vector<FinalizeJob> fj_heap(zero_copy ? 0 : n);
    FinalizeJob *fj = zero_copy
                          ? reinterpret_cast<FinalizeJob *>(static_cast<char *>(ctx->pinned) + pin_jobs_off)
                          : fj_heap.data();
    if (zero_copy) {
        d_jobs = fj;
        results = static_cast<ArgmaxRecord *>(ctx->pinned);
    }
    // block records: job i owns grids[i] records starting at block_pos[i]
    std::vector<size_t> block_pos(n);
    {
        size_t pos = 0;
        for (size_t i = 0; i < n; ++i) {
            block_pos[i] = pos;
            pos += grids[i];
        }
    }
    std::vector<BatchParams> bparams(n);
    BatchParams *d_bparams = nullptr;
    if (n > 1) {
        for (size_t i = 0; i < n; ++i)
            bparams[i] = BatchParams{exact_motif(jobs[i].pssm, jobs[i].d_seq).table, blocks + block_pos[i], 0.0f, 0u, 0ull};
        LM_TRY(ctx->scratch2.reserve(sizeof(BatchParams) * n));
        d_bparams = static_cast<BatchParams *>(ctx->scratch2.ptr);
    }
    std::vector<BatchParams> ordered;  // in launch order: the jobs of a group are contiguous
    ordered.reserve(n);
    for (const JobGroup &g : groups)
        for (size_t i : g.idx)
            ordered.push_back(bparams[i]);
    if (n > 1)
        LM_HIP_TRY(hipMemcpyAsync(d_bparams, ordered.data(), sizeof(BatchParams) * n,
                                  hipMemcpyHostToDevice, ctx->stream));
    const bool two_streams = groups.size() > 1;
    if (two_streams)
        LM_TRY(batch_fork(ctx));
    // one job through the exact kernel: its last workgroup folds the records and writes the result straight
    // into the pinned block -- one launch instead of two (a small scan is launch-latency bound)
    const bool fold_in_kernel = n == 1 && groups.size() == 1 && groups[0].kind == KIND_EXACT && zero_copy;
    uint4 *host_records = nullptr;  // fold_in_kernel, small job: the wavefronts' records in the pinned block
    unsigned host_nrec = 0;
    if (fold_in_kernel)
        LM_TRY(ensure_ticket(ctx));
    size_t launch = 0, bp_pos = 0;
    for (const JobGroup &g : groups) {
        const ScoreArgs &a = jobs[g.idx[0]];
        hipStream_t st = (two_streams && (launch++ & 1)) ? ctx->aux_stream : ctx->stream;
        FusedOut fo{};
        fo.block_best = blocks + block_pos[g.idx[0]];
        if (fold_in_kernel) {
            fo.ticket = ctx->d_ticket;
            fo.final_out = reinterpret_cast<ArgmaxRecord *>(ctx->d_ticket + 4);  // device copy nobody reads: 16 spare bytes
            fo.final_host = results;                                             // pinned: record, then the generation word
            fo.generation = ++ctx->fold_generation ? ctx->fold_generation : ++ctx->fold_generation;
            *reinterpret_cast<volatile unsigned *>(results + 1) = 0u;  // (the block is shared staging: no stale match)
            fo.first_cell_rule = first_cell_rule;
            // few wavefronts and 32-bit cell indices: no fold on the device, the wavefronts' records go straight
            // into the pinned block (from byte 4096 on) and are folded below
            const size_t nrec = (size_t)g.plan.grid.x * (kBlock / 64);
            const unsigned long long ncells = (unsigned long long)(a.row_end - a.row_begin) * a.cols;
            if (ctx->host_fold && ncells < (1ull << 32) && 4096 + (nrec + 1) * 16 <= kPinnedBytes) {
                host_records = reinterpret_cast<uint4 *>(static_cast<char *>(ctx->pinned) + 4096);
                host_nrec = (unsigned)nrec;
                // the block is shared staging: stale bytes must not look like this launch's generation
                for (size_t r = 0; r <= nrec; ++r)
                    reinterpret_cast<volatile unsigned long long *>(host_records)[2 * r] = 0ull;
                fo.host_records = host_records;
            }
        }
        if (g.kind == KIND_EXACT) {
            fo.batch = n > 1 ? d_bparams + bp_pos : nullptr;
            const ExactMotif em = exact_motif(a.pssm, a.d_seq);  // (a group shares length, hence padding)
            fo.lead_rows = em.lead;
            ScoreC32Launcher fn = score_c32_lookup((int)em.m, MODE_ARGMAX, lds_wide((int)a.pssm->k));
            ctx->last_kernel = score_c32_name((int)em.m, MODE_ARGMAX);
            LM_HIP_TRY(fn(g.plan.grid, g.plan.lds, st, a.d_seq, em.table, (int)a.pssm->k,
                          a.row_begin, a.row_end, g.plan.T, g.plan.nstreams, nullptr, fo));
        } else if (g.kind == KIND_CHUNKED) {
            // (always on ctx->stream: the chunk buffer is shared)
            const unsigned cg = chunk_argmax_grid(ctx);
            ArgmaxRecord *recs = fo.block_best;
            const unsigned long long per = chunk_rows_for(ctx, a);
            LM_TRY(for_each_scored_chunk(ctx, a, [&](const float *buf, unsigned long long c0, unsigned long long rows) {
                return launch_argmax_blocks_flat(ctx, ctx->stream, buf, rows * a.cols, (long long)(c0 * a.cols), cg,
                                                 recs + (c0 / per) * cg);
            }));
        } else {
            ctx->last_kernel = "score_generic<1>";
            LM_TRY(launch_generic<MODE_ARGMAX>(ctx, a, fo, dim3(grids[g.idx[0]]), st));
        }
        bp_pos += g.idx.size();
    }
    for (size_t i = 0; i < n; ++i) {
        const ScoreArgs &a = jobs[i];
        fj[i] = FinalizeJob{blocks + block_pos[i], grids[i], (int)a.pssm->m, (int)a.pssm->k,
                            a.d_seq + a.row_begin * a.seq_stride,
                            (unsigned long long)a.seq_stride, a.pssm->d_dense};
    }
    if (two_streams)
        LM_TRY(batch_join(ctx));
    if (!zero_copy)
        LM_HIP_TRY(hipMemcpyAsync(d_jobs, fj, sizeof(FinalizeJob) * n, hipMemcpyHostToDevice,
                                  ctx->stream));
    if (!fold_in_kernel) {
        hipLaunchKernelGGL(argmax_finalize_batch, dim3((unsigned)n), dim3(kBlock), 0, ctx->stream,
                           d_jobs, first_cell_rule, results);
        LM_HIP_TRY(hipGetLastError());
    }
    if (!zero_copy)
        LM_HIP_TRY(hipMemcpyAsync(out, results, sizeof(ArgmaxRecord) * n, hipMemcpyDeviceToHost,
                                  ctx->stream));
    if (fold_in_kernel && host_records)
        return fold_host_records(ctx, host_records, host_nrec, ctx->fold_generation, first_cell_rule != 0, out);
    if (fold_in_kernel) {
        // the kernel raises the generation word behind the pinned record once it is written: poll it (a PCIe write
        // after the fold) instead of waiting for the kernel's completion signal; bounded, then synchronise
        const volatile unsigned *gen = reinterpret_cast<const volatile unsigned *>(results + 1);
        bool seen = false;
        for (unsigned spin = 0; spin < (1u << 20) && !seen; ++spin) {
            seen = __atomic_load_n(gen, __ATOMIC_ACQUIRE) == ctx->fold_generation;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
            __builtin_ia32_pause();
#endif
        }
        if (!seen)
            LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
        memcpy(out, results, sizeof(ArgmaxRecord));
        return LM_HIP_OK;
    }
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));  // also keeps `fj` alive long enough
    if (zero_copy)
        memcpy(out, results, sizeof(ArgmaxRecord) * n);
    return LM_HIP_OK;
}


// ---- fused argmax through the prefilter ---------------------------------------------------------
//
// For large inputs the maximum is found like the Scanner finds hits: (1) the exact scores of
// an evenly spread SAMPLE of the job's rows give a lower bound L of the maximum (it IS a score of
// the matrix); (2) the packed 16-bit prefilter scan flags every row range that may hold a
// score >= L; (3) `rescore_candidates` turns them into exact hits -- all cells with score
// >= L, so the maximum and every tie of it are among them; (4) the hit list is reduced with
// the Generic rule: greatest score, ties -> greatest row-major index (pli/mod.rs:144-151).
// The scan costs half the LDS traffic and adds of the exact kernel (0.42 vs 0.66 ms per Gbp
// at M = 20); sample, re-scoring and reduction add a few tens of microseconds.  PSSMs with
// a prefilter have no NaN / +inf weights, so no score is NaN and the first-cell rule is
// moot.  Jobs whose sample is all -inf, or whose lists overflow (many cells tie with the
// bound), are left to the exact kernel.

struct SampleJob {
    const uint8_t *seq;  // row `row_begin` of the striped matrix (C = 32, stride 32)
    const float *dense;  // M x K weights
    unsigned m, k;
    unsigned long long nchunks, stride;  // chunk c starts at row c * stride
    double pre_offset, pre_factor, pre_emax;
    unsigned td_drop;  // drop-last form of the scan (lm_hip_pssm::d_image2_drop): what the unscanned last row may add; else 0
};

__device__ __forceinline__ unsigned ordered_bits(float v)
{
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone map f32 -> u32 (no NaN here)
}
__device__ __forceinline__ float from_ordered_bits(unsigned k)
{
    return __builtin_bit_cast(float, (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
constexpr unsigned kOrderedNegInf = 0x007fffffu;  // ordered_bits(-inf)

// grid (blocks, jobs): exact scores of `nchunks` chunks of kSampleRows x 32 cells spread evenly
// over the job's rows.  Chunks, not single cells: a lone cell costs M cache sectors for M
// bytes (3.9 M scattered cells = 1.3 ms), a chunk reads its rows once.
constexpr unsigned kSampleRows = kBlock / 32;  // one row per half-wave: a chunk is one pass of the block
constexpr unsigned kMaxSampleM = kMaxLongM;    // the candidate route needs a prefilter kernel: the DNA pair scan goes up to kMaxLongM
// exact score of the cell at `p` for a motif of at most MAXM rows: all symbol loads in flight at
// once, then all weight loads, then the reference's add order -- one HBM latency + one L2 latency
// per chunk instead of M / 4 of each (slots past the motif re-read its last row; the wrap rows
// keep the address valid)
template <unsigned MAXM>
__device__ __forceinline__ float sample_cell(const SampleJob &jb, const uint8_t *__restrict__ p)
{
    unsigned sy[MAXM];
    float w[MAXM];
#pragma unroll
    for (unsigned j = 0; j < MAXM; ++j)
        sy[j] = p[(j < jb.m ? j : jb.m - 1) * 32];
#pragma unroll
    for (unsigned j = 0; j < MAXM; ++j)
        w[j] = jb.dense[(j < jb.m ? j : jb.m - 1) * jb.k + sy[j]];
    float sc = 0.0f;
#pragma unroll
    for (unsigned j = 0; j < MAXM; ++j)
        sc = j < jb.m ? sc + w[j] : sc;
    return sc;
}

// motifs beyond kMaxSampleM rows (the pair scan goes up to kMaxPairM): row by row, the same add order
__device__ __forceinline__ float sample_cell_loop(const SampleJob &jb, const uint8_t *__restrict__ p)
{
    float sc = 0.0f;
    for (unsigned j = 0; j < jb.m; ++j)
        sc = sc + jb.dense[j * jb.k + p[j * 32]];
    return sc;
}

__global__ __launch_bounds__(kBlock) void argmax_sample(const SampleJob *__restrict__ jobs,
                                                        unsigned *__restrict__ partial)
{
    const SampleJob jb = jobs[blockIdx.y];
    unsigned best = kOrderedNegInf;
    const unsigned col = threadIdx.x & 31, sub = threadIdx.x >> 5;
    for (unsigned long long c = blockIdx.x; c < jb.nchunks; c += gridDim.x) {
        const unsigned long long r0 = c * jb.stride;  // chunk rows r0 .. r0 + kSampleRows - 1
        const uint8_t *p = jb.seq + (r0 + sub) * 32 + col;
        // (three unroll depths: most motifs of a batch are short, and every slot costs two loads)
        const float sc = jb.m <= 12   ? sample_cell<12>(jb, p)
                         : jb.m <= 24 ? sample_cell<24>(jb, p)
                         : jb.m <= 36 ? sample_cell<36>(jb, p)
                         : jb.m <= kMaxSampleM ? sample_cell<kMaxSampleM>(jb, p)
                                               : sample_cell_loop(jb, p);
        const unsigned key = ordered_bits(sc);
        best = key > best ? key : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    // one record per workgroup, folded by argmax_prepare: thousands of atomics on one address serialise at ~12 ns each
    // (round 5 tried a ticket per workgroup so that the last one folds, to save that launch: 19 + 5 -> 62 us)
    __shared__ unsigned wave_best[kBlock / 64];
    if ((threadIdx.x & 63) == 0)
        wave_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w)
            best = wave_best[w] > best ? wave_best[w] : best;
        partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = best;
    }
}

// one wavefront per job: folds the sample's per-workgroup records into the lower bound, then
// lower bound -> f32 threshold of the re-scoring and discrete threshold of the scan (same
// formula as launch_score_threshold_batch); td = 0xffffffff = "skip"
__global__ __launch_bounds__(64) void argmax_prepare(const SampleJob *__restrict__ jobs, const unsigned n,
                                                     const unsigned *__restrict__ partial, const unsigned nper,
                                                     RescoreJob *__restrict__ rjobs,
                                                     BatchParams *__restrict__ bparams)
{
    const unsigned j = blockIdx.x;
    unsigned bound = kOrderedNegInf;
    for (unsigned i = threadIdx.x; i < nper; i += 64) {
        const unsigned v = partial[(size_t)j * nper + i];
        bound = v > bound ? v : bound;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = __shfl_xor(bound, off);
        bound = o > bound ? o : bound;
    }
    if (threadIdx.x != 0)
        return;
    unsigned td = 0xffffffffu;
    float t = INFINITY;
    if (bound != kOrderedNegInf) {
        t = from_ordered_bits(bound);
        const double scaled = floor(((double)t - jobs[j].pre_offset) / jobs[j].pre_factor) -
                              ceil(jobs[j].pre_emax / jobs[j].pre_factor) - 1.0;
        if (scaled >= 1.0)
            td = scaled > 65535.0 ? 65535u : (unsigned)scaled;
        else
            t = INFINITY;  // the bound is too low for the 16-bit range: leave the job to the exact kernel
        if (td != 0xffffffffu && jobs[j].td_drop) {
            if (td > 4u * jobs[j].td_drop) {
                td -= jobs[j].td_drop;
            } else {  // the unscanned row carries too much of the bound: nothing is flagged, the exact kernel takes over
                td = 0xffffffffu;
                t = INFINITY;
            }
        }
    }
    rjobs[j].threshold = t;
    bparams[j].td = td;
}

// reduction of the hit list: per job the greatest score, then the greatest key among its ties
// atomicMax(&target[job], value) for every live lane, with ONE atomic per distinct job of the
// wavefront: the list is clustered by job in runs shorter than a wavefront (a re-scoring
// workgroup stages the hits of several rounds = several jobs before it flushes), and a
// wavefront-wide atomic with 64 different addresses costs 64 operations -- 10^6 records took
// 2.4 ms that way.  `value` 0 = nothing to contribute (0 is below every real value here).
template <typename T>
__device__ __forceinline__ void wave_atomic_max_by_job(const unsigned long long job, const T value,
                                                       const bool live, T *__restrict__ target)
{
    const int lane = threadIdx.x & 63;
    unsigned long long active = __ballot(live);
    while (active) {  // wave-uniform
        const int leader = __ffsll((long long)active) - 1;
        const unsigned long long j = __shfl(job, leader);
        const bool mine = live && job == j;
        T x = mine ? value : (T)0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const T o = __shfl_xor(x, off);
            x = o > x ? o : x;
        }
        if (lane == leader && x != (T)0)
            atomicMax(&target[j], x);
        active &= ~__ballot(mine);
    }
}

__global__ __launch_bounds__(kBlock) void hits_best_value(const HitRecord *__restrict__ hits,
                                                          const unsigned long long *__restrict__ count,
                                                          const unsigned long long capacity,
                                                          unsigned *__restrict__ best_value)
{
    unsigned long long n = *count;
    if (n > capacity)
        n = capacity;
    const unsigned long long span = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long i0 = (unsigned long long)blockIdx.x * kBlock; i0 < n; i0 += span) {
        const unsigned long long i = i0 + threadIdx.x;
        const bool live = i < n;
        HitRecord h{};
        if (live)
            h = hits[i];
        wave_atomic_max_by_job(h.key >> 40, live ? ordered_bits(h.value) : 0u, live, best_value);
    }
}

__global__ __launch_bounds__(kBlock) void hits_best_key(const HitRecord *__restrict__ hits,
                                                        const unsigned long long *__restrict__ count,
                                                        const unsigned long long capacity,
                                                        const unsigned *__restrict__ best_value,
                                                        unsigned long long *__restrict__ best_key)
{
    unsigned long long n = *count;
    if (n > capacity)
        n = capacity;
    const unsigned long long span = (unsigned long long)gridDim.x * kBlock;
    for (unsigned long long i0 = (unsigned long long)blockIdx.x * kBlock; i0 < n; i0 += span) {
        const unsigned long long i = i0 + threadIdx.x;
        const bool live = i < n;
        HitRecord h{};
        if (live)
            h = hits[i];
        const unsigned long long job = h.key >> 40;
        // 0 = not a tie of the job's best score
        const unsigned long long k = (live && ordered_bits(h.value) == best_value[job])
                                         ? (h.key & ((1ull << 40) - 1)) + 1 : 0ull;
        wave_atomic_max_by_job(job, k, live, best_key);
    }
}

__global__ void argmax_collect(const unsigned n, const unsigned *__restrict__ best_value,
                               const unsigned long long *__restrict__ best_key,
                               ArgmaxRecord *__restrict__ out,
                               const unsigned long long *__restrict__ counters,
                               unsigned long long *__restrict__ counters_out)
{
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 && counters_out) {  // {hits, candidates}: the host checks them for overflow
        counters_out[0] = counters[0];
        counters_out[1] = counters[1];
    }
    if (j >= n)
        return;
    ArgmaxRecord r;
    r.found = best_key[j] != 0;
    r.value = from_ordered_bits(best_value[j]);
    r.index = (long long)best_key[j] - 1;
    out[j] = r;
}

// ONE job: the three launches above in one workgroup -- best value, greatest key among its ties, the record and the
// counters for the host.  The list holds ~10^3 records (at most its capacity, 8 192 + 65 536) and every launch of the
// tail costs ~5 us whatever it does (profiles/r05_timeline_fused.txt), 15 of the call's 280.
constexpr int kBestSingleBlock = 1024;
__global__ __launch_bounds__(kBestSingleBlock) void hits_best_single(const HitRecord *__restrict__ hits,
                                                                     const unsigned long long *__restrict__ counters,
                                                                     const unsigned long long capacity,
                                                                     ArgmaxRecord *__restrict__ out,
                                                                     unsigned long long *__restrict__ counters_out)
{
    __shared__ unsigned wave_value[kBestSingleBlock / 64];
    __shared__ unsigned long long wave_key[kBestSingleBlock / 64];
    unsigned long long n = counters[0];
    if (n > capacity)
        n = capacity;
    unsigned bv = kOrderedNegInf;
    for (unsigned long long i = threadIdx.x; i < n; i += kBestSingleBlock) {
        const unsigned v = ordered_bits(hits[i].value);
        bv = v > bv ? v : bv;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = __shfl_xor(bv, off);
        bv = o > bv ? o : bv;
    }
    if ((threadIdx.x & 63) == 0)
        wave_value[threadIdx.x >> 6] = bv;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kBestSingleBlock / 64; ++w)
        bv = wave_value[w] > bv ? wave_value[w] : bv;
    unsigned long long bk = 0;  // (low + 1) of the greatest tied key; 0 = no record reaches the value
    for (unsigned long long i = threadIdx.x; i < n; i += kBestSingleBlock) {
        const HitRecord h = hits[i];
        const unsigned long long k = ordered_bits(h.value) == bv ? (h.key & ((1ull << 40) - 1)) + 1 : 0ull;
        bk = k > bk ? k : bk;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(bk, off);
        bk = o > bk ? o : bk;
    }
    if ((threadIdx.x & 63) == 0)
        wave_key[threadIdx.x >> 6] = bk;
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    for (int w = 0; w < kBestSingleBlock / 64; ++w)
        bk = wave_key[w] > bk ? wave_key[w] : bk;
    if (counters_out) {
        counters_out[0] = counters[0];
        counters_out[1] = counters[1];
    }
    ArgmaxRecord r;
    r.found = bk != 0;
    r.value = from_ordered_bits(bv);
    r.index = (long long)bk - 1;
    out[0] = r;
}

// The route has ~60 us of fixed cost per call (sample, five small launches): it pays from
// ~100 M cells per call on (measured crossover at M = 20), whether in one job or in a batch.
constexpr unsigned long long kPrefilterArgmaxMinCells = 8ull << 20;     // per job
constexpr unsigned long long kPrefilterArgmaxMinTotal = 100ull * 1000 * 1000;  // per call

// Tries the candidate route for the jobs that qualify; done[i] = 1 and out[i] filled for
// the ones it settled.  Anything else (small jobs, odd shapes, no prefilter, all -inf
// samples, list overflow) is left for the exact kernel.
static int argmax_by_prefilter(lm_hip_ctx *ctx, const ScoreArgs *jobs, size_t n, ArgmaxRecord *out,
                               char *done)
{
    std::vector<size_t> pick;
    for (size_t i = 0; i < n; ++i) {
        const ScoreArgs &a = jobs[i];
        const unsigned long long cells = (unsigned long long)(a.row_end - a.row_begin) * a.cols;
        // short motifs have few distinct scores: the best k-mer alone occurs cells / (K-1)^M
        // times and every occurrence is a hit -- worth it while that stays within the list's
        // 8 192 records per job (100 Mbp: M >= 7; the 801 JASPAR motifs of length 7 and 8 take
        // 23 / 13 us each on this route against 60 / 50 us on the exact kernel)
        const double kmers = std::pow((double)(a.pssm->k - 1), (double)a.pssm->m);
        if (!done[i] && a.pssm->has_prefilter && a.pssm->m >= 2 && cells >= kPrefilterArgmaxMinCells &&
            cells < (1ull << 40) && kmers >= (double)cells / 8192.0 &&
            (plan_c32(ctx, a, false, 1).ok || plan_c32(ctx, a, false, 2).ok))
            pick.push_back(i);
    }
    const size_t nq = pick.size();
    unsigned long long picked_cells = 0;
    for (size_t i : pick)
        picked_cells += (unsigned long long)(jobs[i].row_end - jobs[i].row_begin) * jobs[i].cols;
    if (nq == 0 || nq > (1u << 20) || picked_cells < kPrefilterArgmaxMinTotal)
        return LM_HIP_OK;
    std::vector<ScoreArgs> qjobs(nq);
    std::vector<SampleJob> sjobs(nq);
    std::vector<RescoreJob> rjobs(nq);
    unsigned long long max_chunks = 0;
    for (size_t q = 0; q < nq; ++q) {
        const ScoreArgs &a = jobs[pick[q]];
        qjobs[q] = a;
        // 1/1024 of the rows, in chunks of kSampleRows rows spread evenly (rows >= 2^20 here)
        const unsigned long long rows = a.row_end - a.row_begin;
        unsigned long long nchunks = std::max<unsigned long long>(rows / 1024 / kSampleRows, 32);
        // a lone job: at most one chunk per workgroup of the sample's grid (1 Gbp: 2 048 chunks instead of 3 815 -- one
        // chain of loads per workgroup instead of two, 19 -> 14 us; the bound of a smaller sample is lower, ~1 900 cells
        // tie or beat it instead of ~1 000, which the re-scoring does not notice).  More workgroups instead: 24 -> 33 us.
        if (nq == 1)
            nchunks = std::min<unsigned long long>(nchunks, (unsigned long long)ctx->num_cus * 8);
        max_chunks = std::max(max_chunks, nchunks);
        sjobs[q] = SampleJob{a.d_seq + a.row_begin * a.seq_stride, a.pssm->d_dense, (unsigned)a.pssm->m,
                             (unsigned)a.pssm->k, nchunks, (rows - kSampleRows) / (nchunks - 1),
                             a.pssm->pre_offset, a.pssm->pre_factor, a.pssm->pre_emax, 0u};
        rjobs[q] = RescoreJob{sjobs[q].seq, a.pssm->d_dense, (unsigned)a.pssm->m, (unsigned)a.pssm->k,
                              INFINITY, 0, 0};
    }
    const std::vector<JobGroup> groups = group_jobs(ctx, qjobs.data(), nq, [&](size_t q) {
        return (ctx->pair_prefilter && plan_c32(ctx, qjobs[q], false, 2).ok) ? (int)KIND_PREFILTER2
                                                                             : (int)KIND_PREFILTER;
    });
    std::vector<char> pairs_of(nq, 0);
    for (const JobGroup &g : groups)
        for (size_t q : g.idx)
            pairs_of[q] = g.kind == KIND_PREFILTER2;
    // a lone pair scan of M = 20, 24, ... 36 rows: the drop-last form (score_threshold.hip, lm_hip_pssm::d_image2_drop)
    C32Plan drop_plan;
    if (nq == 1 && groups.size() == 1 && groups[0].kind == KIND_PREFILTER2 && ctx->drop_last && qjobs[0].pssm->d_image2_drop &&
        score_c32_prefilter2_lookup((int)qjobs[0].pssm->m - 1, (int)qjobs[0].pssm->k))
        drop_plan = plan_c32(ctx, MotifShape{qjobs[0].pssm->m - 1, qjobs[0].pssm->k, true}, qjobs[0], false, 2, 1);
    const bool drop_last_form = drop_plan.ok;
    if (drop_last_form)
        sjobs[0].td_drop = qjobs[0].pssm->drop_dmax;
    ctx->last_scan_rows = ctx->last_scan_lds_bytes = 0;
    if (nq == 1 && groups.size() == 1) {
        const size_t scanned = qjobs[0].pssm->m - (drop_last_form ? 1 : 0);
        ctx->last_scan_lds_bytes = scan_lds_bytes(groups[0].kind, scanned, qjobs[0].pssm->k);
        ctx->last_scan_rows = (unsigned)scanned;
    }
    std::vector<BatchParams> bparams;  // launch order; rjobs / sjobs are permuted the same way
    // positions in launch order; groups of the pair scan with several jobs run `per_pass[g]`
    // motifs per pass and are padded to a multiple of that (a padding position samples nothing,
    // so argmax_prepare leaves its td at "skip" and it flags nothing)
    std::vector<size_t> order, group_pos(groups.size());
    std::vector<char> is_pad, in_multi;  // in_multi: the position's group runs several motifs per pass (tables in that kernel's layout)
    std::vector<int> per_pass(groups.size(), 1);
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const JobGroup &g = groups[gi];
        group_pos[gi] = order.size();
        for (size_t q : g.idx) {
            order.push_back(q);
            is_pad.push_back(0);
        }
        const int m = (int)qjobs[g.idx[0]].pssm->m;
        bool multi = g.kind == KIND_PREFILTER2 && ctx->multi_motif && g.idx.size() >= 2 && qjobs[g.idx[0]].pssm->k == 5 &&
                     score_c32_prefilter2_multi_lookup(m);
        for (size_t q : g.idx)  // (every matrix of the group needs its table in that kernel's layout)
            multi = multi && qjobs[q].pssm->d_image2_multi != nullptr;
        if (multi) {
            per_pass[gi] = prefilter2_multi(m);
            while ((order.size() - group_pos[gi]) % per_pass[gi]) {
                order.push_back(g.idx.back());
                is_pad.push_back(1);
            }
        }
        in_multi.resize(order.size(), multi ? 1 : 0);
    }
    const size_t npos = order.size();
    std::vector<SampleJob> sj(npos);
    std::vector<RescoreJob> rj(npos);
    for (size_t pos = 0; pos < npos; ++pos) {
        const size_t q = order[pos];
        sj[pos] = sjobs[q];
        if (is_pad[pos])
            sj[pos].nchunks = 0;
        rj[pos] = rjobs[q];
        bparams.push_back(BatchParams{drop_last_form ? qjobs[q].pssm->d_image2_drop
                                      : in_multi[pos] ? qjobs[q].pssm->d_image2_multi
                                      : pairs_of[q]  ? qjobs[q].pssm->d_image2
                                                     : qjobs[q].pssm->d_image,
                                      nullptr, 0.0f, 0xffffffffu, (unsigned long long)pos << 40});
    }
    // ~1024 cells tie with or beat the bound of a 1/1024 sample; leave room for 8x that
    // (bounded at 32 M / 128 M records = 2.5 GB for very large batches: lists that overflow send
    // the batch to the exact kernel, they never change a result)
    const unsigned long long cap = std::min<unsigned long long>(npos * 8192 + (1 << 16), 32ull << 20),
                             ccap = 4 * cap;
    // layout: the head -- counters | sample bounds | best values | best keys | the three job tables --
    // is assembled in the upper half of the pinned buffer and reaches the device as ONE copy
    // (three memsets and three staged copies from pageable memory cost more than the sample pass)
    const unsigned sgrid = (unsigned)std::min<unsigned long long>(
        max_chunks, std::max<unsigned long long>((unsigned long long)ctx->num_cus * 8 / npos, 16));
    const size_t off_bval = 256;  // the counters keep their cache lines to themselves
    const size_t off_bkey = (off_bval + npos * 4 + 15) / 16 * 16;
    const size_t off_rj = off_bkey + npos * 8;
    const size_t off_bp = off_rj + (npos * sizeof(RescoreJob) + 15) / 16 * 16;
    const size_t off_sj = off_bp + (npos * sizeof(BatchParams) + 15) / 16 * 16;
    const size_t off_res = off_sj + (npos * sizeof(SampleJob) + 15) / 16 * 16;
    const size_t off_hits = off_res + npos * sizeof(ArgmaxRecord);
    const size_t off_cands = off_hits + cap * sizeof(HitRecord);
    const size_t off_partial = off_cands + ccap * sizeof(Candidate);  // the sample's per-workgroup maxima
    LM_TRY(ctx->scratch.reserve(off_partial + npos * sgrid * sizeof(unsigned)));
    char *base = static_cast<char *>(ctx->scratch.ptr);
    FusedOut fo{};
    fo.hit_count = reinterpret_cast<unsigned long long *>(base);
    fo.cand_count = fo.hit_count + 1;
    unsigned *d_partial = reinterpret_cast<unsigned *>(base + off_partial);
    unsigned *d_bval = reinterpret_cast<unsigned *>(base + off_bval);
    unsigned long long *d_bkey = reinterpret_cast<unsigned long long *>(base + off_bkey);
    fo.hits = reinterpret_cast<HitRecord *>(base + off_hits);
    fo.hit_capacity = cap;
    fo.cands = reinterpret_cast<Candidate *>(base + off_cands);
    fo.cand_capacity = ccap;
    RescoreJob *d_rj = reinterpret_cast<RescoreJob *>(base + off_rj);
    BatchParams *d_bp = reinterpret_cast<BatchParams *>(base + off_bp);
    SampleJob *d_sj = reinterpret_cast<SampleJob *>(base + off_sj);
    ArgmaxRecord *d_res = reinterpret_cast<ArgmaxRecord *>(base + off_res);
    hipStream_t st = ctx->stream;
    if (off_res <= kPinnedBytes / 2) {
        char *head = static_cast<char *>(ctx->pinned) + kPinnedBytes / 2;
        memset(head, 0, off_rj);
        for (size_t q = 0; q < npos; ++q)  // best values start at -inf
            reinterpret_cast<unsigned *>(head + off_bval)[q] = kOrderedNegInf;
        memcpy(head + off_rj, rj.data(), npos * sizeof(RescoreJob));
        memcpy(head + off_bp, bparams.data(), npos * sizeof(BatchParams));
        memcpy(head + off_sj, sj.data(), npos * sizeof(SampleJob));
        LM_HIP_TRY(hipMemcpyAsync(base, head, off_res, hipMemcpyHostToDevice, st));
    } else {
        LM_HIP_TRY(hipMemsetAsync(base, 0, 16, st));
        LM_HIP_TRY(hipMemsetAsync(d_bkey, 0, npos * 8, st));
        LM_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_bval), (int)kOrderedNegInf, npos, st));
        LM_HIP_TRY(hipMemcpyAsync(d_rj, rj.data(), npos * sizeof(RescoreJob), hipMemcpyHostToDevice, st));
        LM_HIP_TRY(hipMemcpyAsync(d_bp, bparams.data(), npos * sizeof(BatchParams), hipMemcpyHostToDevice, st));
        LM_HIP_TRY(hipMemcpyAsync(d_sj, sj.data(), npos * sizeof(SampleJob), hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(argmax_sample, dim3(sgrid, (unsigned)npos), dim3(kBlock), 0, st, d_sj, d_partial);
    hipLaunchKernelGGL(argmax_prepare, dim3((unsigned)npos), dim3(64), 0, st, d_sj, (unsigned)npos, d_partial, sgrid,
                       d_rj, d_bp);
    LM_HIP_TRY(hipGetLastError());
    const bool two_streams = groups.size() > 1;
    scan_timer_begin(ctx, st);
    if (two_streams)
        LM_TRY(batch_fork(ctx));
    size_t launch = 0;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const JobGroup &g = groups[gi];
        const ScoreArgs &a = qjobs[g.idx[0]];
        hipStream_t ls = (two_streams && (launch++ & 1)) ? ctx->aux_stream : st;
        fo.batch = d_bp + group_pos[gi];
        if (per_pass[gi] > 1) {  // several motifs of this length per pass over the sequence
            dim3 grid = g.plan.grid;
            grid.y = (unsigned)((g.idx.size() + per_pass[gi] - 1) / per_pass[gi]);
            ctx->last_kernel = "score_c32_prefilter2_multi";
            LM_HIP_TRY(score_c32_prefilter2_multi_lookup((int)a.pssm->m)(grid, ls, a.d_seq, a.row_begin, a.row_end,
                                                                        g.plan.T, g.plan.nstreams, fo));
            continue;
        }
        const bool pairs = g.kind == KIND_PREFILTER2;
        ctx->last_kernel = pairs ? "score_c32_prefilter2" : block_scan(ctx, a) ? "score_c32_prefilter_blk" : "score_c32_prefilter";
        if (drop_last_form) {  // (table and threshold come from the job's BatchParams)
            PrefilterLauncher fn = score_c32_prefilter2_lookup((int)a.pssm->m - 1, (int)a.pssm->k);
            LM_HIP_TRY(fn(drop_plan.grid, drop_plan.lds, ls, a.d_seq, a.pssm->d_image2_drop, (int)a.pssm->k, a.row_begin, a.row_end,
                          drop_plan.T, drop_plan.nstreams, 0xffffffffu, fo));
            continue;
        }
        PrefilterLauncher fn = pairs ? score_c32_prefilter2_lookup((int)a.pssm->m, (int)a.pssm->k)
                                     : score_c32_prefilter_lookup((int)a.pssm->m, lds_wide((int)a.pssm->k), block_scan(ctx, a));
        LM_HIP_TRY(fn(g.plan.grid, g.plan.lds, ls, a.d_seq, pairs ? a.pssm->d_image2 : a.pssm->d_image,
                      (int)a.pssm->k, a.row_begin, a.row_end, g.plan.T, g.plan.nstreams, 0xffffffffu, fo));
    }
    if (two_streams)
        LM_TRY(batch_join(ctx));
    scan_timer_end(ctx, st);
    fo.batch = nullptr;
    LM_TRY(launch_rescore(ctx, st, d_rj, fo, rj.data(), npos));
    // results and the two list counters are written straight into the pinned buffer's lower half
    const bool pin = 16 + npos * sizeof(ArgmaxRecord) <= kPinnedBytes / 2;
    ArgmaxRecord *res = pin ? reinterpret_cast<ArgmaxRecord *>(static_cast<char *>(ctx->pinned) + 16) : d_res;
    if (npos == 1) {
        hipLaunchKernelGGL(hits_best_single, dim3(1), dim3(kBestSingleBlock), 0, st, fo.hits, fo.hit_count, cap, res,
                           pin ? static_cast<unsigned long long *>(ctx->pinned) : static_cast<unsigned long long *>(nullptr));
    } else {
        const unsigned hgrid = (unsigned)ctx->num_cus * 2;
        hipLaunchKernelGGL(hits_best_value, dim3(hgrid), dim3(kBlock), 0, st, fo.hits, fo.hit_count, cap, d_bval);
        hipLaunchKernelGGL(hits_best_key, dim3(hgrid), dim3(kBlock), 0, st, fo.hits, fo.hit_count, cap, d_bval,
                           d_bkey);
        hipLaunchKernelGGL(argmax_collect, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, st, (unsigned)npos,
                           d_bval, d_bkey, res, fo.hit_count,
                           pin ? static_cast<unsigned long long *>(ctx->pinned) : static_cast<unsigned long long *>(nullptr));
    }
    LM_HIP_TRY(hipGetLastError());
    std::vector<ArgmaxRecord> host_res(pin ? 0 : npos);
    if (!pin) {
        LM_HIP_TRY(hipMemcpyAsync(ctx->pinned, base, 16, hipMemcpyDeviceToHost, st));
        LM_HIP_TRY(hipMemcpyAsync(host_res.data(), d_res, npos * sizeof(ArgmaxRecord), hipMemcpyDeviceToHost, st));
    }
    LM_HIP_TRY(hipStreamSynchronize(st));
    scan_timer_read(ctx);
    const unsigned long long nhits = static_cast<unsigned long long *>(ctx->pinned)[0];
    const unsigned long long ncand = static_cast<unsigned long long *>(ctx->pinned)[1];
    if (getenv("LM_HIP_TRACE"))
        fprintf(stderr, "[lm_hip] candidate-route argmax: %zu jobs, %llu candidates (room %llu), %llu hits (room %llu)\n",
                npos, ncand, ccap, nhits, cap);
    if (nhits > cap || ncand > ccap)
        return LM_HIP_OK;  // truncated lists prove nothing: the exact kernel takes over
    const ArgmaxRecord *r = pin ? res : host_res.data();
    for (size_t pos = 0; pos < npos; ++pos)
        if (r[pos].found) {
            const size_t i = pick[order[pos]];
            out[i] = r[pos];
            done[i] = 1;
        }
    return LM_HIP_OK;
}

// Fused score+argmax of `n` independent jobs: the candidate route where it applies, the
// exact kernel for the rest.
// ---- fused argmax of short motifs: the last rows suffice -----------------------------------------
//
// A short motif's best k-mer occurs all over a long sequence, and the Generic argmax is the LAST
// maximal cell in (row, col) order (pli/mod.rs:144-151).  B = the sequential f32 sum of the row
// maxima of the PSSM is an upper bound of every score (rounding is monotone, so the sum of
// termwise larger weights in the same order is not smaller) and IS the score of the cells that
// hold a best k-mer.  So: score only the last rows of the range -- enough cells that a best
// k-mer is expected ~24 times, (K-1)^M * 24 -- and if their maximum equals B bit for bit, their
// argmax is the answer: no later cell exists, and no earlier cell can beat B.  Otherwise (the
// best k-mer is absent from the suffix) the job takes the usual routes.  On the JASPAR batch
// the motifs up to length 9 (61 % of them) are settled from ~5 % of the rows or less.
// How many best k-mers the suffix should hold on average.  With lambda = cells / (K-1)^M expected in the whole
// range, scanning a fraction f costs f + exp(-lambda * f) of a full scan (the miss falls back to the usual
// routes), minimal at f = ln(lambda) / lambda: the suffix holds ln(lambda) occurrences, between 1.5 (a miss
// every fifth motif, still a net gain) and 8.  Round 1 used a flat 24: never a miss, four times the rows
// (JASPAR batch 14.0-14.7 -> 11.0 ms; the context option "suffix_occurrences" pins the value for A/B runs).
static double suffix_occurrences(const lm_hip_ctx *ctx, double lambda)
{
    if (ctx->suffix_occurrences > 0)
        return ctx->suffix_occurrences;
    return std::min(8.0, std::max(1.5, std::log(std::max(lambda, 1.0))));
}
constexpr unsigned long long kSuffixMinRows = 1ull << 15;  // keeps the streams long enough

static int argmax_by_suffix(lm_hip_ctx *ctx, const ScoreArgs *jobs, size_t n, ArgmaxRecord *out,
                            char *done)
{
    // Two ways to look at a suffix.  Where a best k-mer is dense in it (short motifs) the exact
    // argmax kernel scores it.  Where it is sparse, the suffix may be long (up to half the
    // range) and the fused THRESHOLD at t = B scans it instead: the packed pair scan flags the
    // cells that can reach B, the exact re-scoring keeps those that do, and the last hit in
    // row-major order is the argmax.
    std::vector<ScoreArgs> subs, tsubs;
    std::vector<size_t> idx, tidx;
    std::vector<float> bound, tbound;
    for (size_t i = 0; i < n; ++i) {
        const ScoreArgs &a = jobs[i];
        const lm_hip_pssm *p = a.pssm;
        if (done[i] || !p->has_prefilter || p->m < 1)  // has_prefilter: no NaN / +inf weights
            continue;
        const unsigned long long rows = a.row_end - a.row_begin;
        const double kmers = std::pow((double)(p->k - 1), (double)p->m);
        if (!(kmers < 1e15))
            continue;
        const double lambda = (double)rows * (double)a.cols / kmers;
        if (lambda < 2.0)
            continue;  // a best k-mer is not expected in the range at all
        const double need_cells = suffix_occurrences(ctx, lambda) * kmers;
        const unsigned long long need_rows =
            std::max<unsigned long long>((unsigned long long)(need_cells / (double)a.cols) + 1, kSuffixMinRows);
        const bool dense = (double)need_rows * (double)a.cols / kmers > 256.0;  // expected hits at t = B
        if (need_rows > (dense ? rows / 4 : rows / 2))
            continue;
        const float b = best_kmer_score(p);
        if (!std::isfinite(b))
            continue;
        ScoreArgs sub = a;
        sub.row_begin = a.row_end - need_rows;
        (dense ? subs : tsubs).push_back(sub);
        (dense ? idx : tidx).push_back(i);
        (dense ? bound : tbound).push_back(b);
    }
    if (!subs.empty()) {
        std::vector<ArgmaxRecord> recs(subs.size());
        LM_TRY(launch_score_argmax_exact(ctx, subs.data(), subs.size(), 0, recs.data()));
        for (size_t q = 0; q < subs.size(); ++q) {
            const ArgmaxRecord &r = recs[q];
            if (!r.found || !(r.value == bound[q]))
                continue;  // no best k-mer among the last rows
            const size_t i = idx[q];
            out[i] = r;
            out[i].index += (long long)((subs[q].row_begin - jobs[i].row_begin) * jobs[i].cols);
            done[i] = 1;
        }
    }
    if (!tsubs.empty()) {
        // (the list-size memory of the threshold calls belongs to the caller's threshold scans)
        const unsigned long long keep_hits = ctx->last_hit_count, keep_cands = ctx->last_cand_count;
        HitOutput ho;
        const int st = launch_score_threshold_batch(ctx, tsubs.data(), tbound.data(), tsubs.size(), HitKeys::RowMajor, &ho);
        ctx->last_hit_count = keep_hits;
        ctx->last_cand_count = keep_cands;
        if (st != LM_HIP_OK) {
            ho.release();
            return st;
        }
        for (size_t q = 0; q < tsubs.size(); ++q) {
            if (ho.job_start[q + 1] == ho.job_start[q])
                continue;  // no best k-mer among the last rows
            const size_t last = ho.job_start[q + 1] - 1, i = tidx[q];
            out[i].value = ho.values[last];
            out[i].found = 1;
            out[i].index = (long long)((ho.coords[last].row + (tsubs[q].row_begin - jobs[i].row_begin)) * jobs[i].cols +
                                       ho.coords[last].col);
            done[i] = 1;
        }
        ho.release();
    }
    return LM_HIP_OK;
}

int launch_score_argmax_batch(lm_hip_ctx *ctx, const ScoreArgs *jobs, size_t n,
                              int first_cell_rule, ArgmaxRecord *out)
{
    if (n == 0)
        return LM_HIP_OK;
    std::vector<char> done(n, 0);
    if (ctx->use_prefilter && ctx->suffix_argmax)
        LM_TRY(argmax_by_suffix(ctx, jobs, n, out, done.data()));
    if (ctx->use_prefilter)
        LM_TRY(argmax_by_prefilter(ctx, jobs, n, out, done.data()));
    std::vector<ScoreArgs> rest;
    std::vector<size_t> rest_idx;
    for (size_t i = 0; i < n; ++i)
        if (!done[i]) {
            rest.push_back(jobs[i]);
            rest_idx.push_back(i);
        }
    if (rest.empty())
        return LM_HIP_OK;
    if (rest.size() == n)
        return launch_score_argmax_exact(ctx, jobs, n, first_cell_rule, out);
    std::vector<ArgmaxRecord> recs(rest.size());
    LM_TRY(launch_score_argmax_exact(ctx, rest.data(), rest.size(), first_cell_rule, recs.data()));
    for (size_t k = 0; k < rest.size(); ++k)
        out[rest_idx[k]] = recs[k];
    return LM_HIP_OK;
}

int launch_score_argmax(lm_hip_ctx *ctx, const ScoreArgs &a, int first_cell_rule,
                        ArgmaxRecord *out)
{
    return launch_score_argmax_batch(ctx, &a, 1, first_cell_rule, out);
}

}  // namespace lm
