// score_launch.hpp -- what the launch code of the scoring kernels shares between its translation units
// (score_plan.hip: registry + stream geometry; score_store.hip: Score into a matrix; score_argmax.hip and
// score_threshold.hip: the fused Maximum / Threshold routes).  Kernel bodies: score_kernels.hpp.
#pragma once

#include <algorithm>
#include <map>
#include <tuple>
#include <vector>

#include "score_prefilter2.hpp"
#include "score_u8.hpp"

namespace lm {

// ---- registry look-ups defined with the registry (score_plan.hip; the others are declared next to their kernels) ----

ScoreU8Launcher score_c32_lookup_u8(int M, bool pairs, bool wide);
PrefilterMultiLauncher score_c32_prefilter2_multi_lookup(int M);

// ---- stream geometry (score_plan.hip) -------------------------------------------------------------------

struct C32Plan {
    bool ok = false;
    unsigned long long T = 0, nstreams = 0;
    dim3 grid;
    size_t lds = 0;
};

// Rows per stream T = q*M + 1.  Short streams win: the rows being written by all
// resident wavefronts then form a compact window that moves through memory in
// order, which HBM (and the TLB) reward more than the M-1 fill steps per stream
// cost -- the kernel is HBM-bound, not LDS-bound (profiles/r01_kbench2_nt.txt:
// T=61 0.947 ms, T=501 0.995 ms, T=4001 1.12 ms at M=20 on 1 Gbp).
// (An XCD-aware block remap -- each XCD one contiguous eighth of the rows -- won 3 % on fresh,
// separately hipMalloc'ed buffers and lost 3 % inside one large arena or under PyTorch's
// allocator: eight distant windows instead of one compact one.  Removed in round 5; numbers in
// profiles/r01_kbench5_ab.txt, r01_kbench6_place.txt.)
// `prefilter`: 0 = exact kernels, 1 = one-symbol prefilter (streams of q*MP + 1 rows),
// 2 = pair-symbol prefilter (streams of q*RING + 2 rows)
// what the planner needs to know about the matrix (a.pssm may be absent: u8 scores)
struct MotifShape {
    size_t m, k;
    bool pair_table;
};

C32Plan plan_c32(const lm_hip_ctx *ctx, const MotifShape &ms, const ScoreArgs &a, bool store, int prefilter = 0,
                 size_t batch = 1, unsigned long long default_rows = 0, bool allow16 = false);

// How the exact C = 32 kernels see a motif: up to kMaxFastM rows as they are (byte symbol loads, any length);
// kMaxFastM < M <= kMaxLongM as ONE slice padded with leading zero rows to a multiple of 4 (the long family:
// dword symbol loads, so the matrix must be 4-byte aligned); longer motifs have no single-pass kernel.
struct ExactMotif {
    size_t m = 0;            // rows the kernel is instantiated for
    const float *table = nullptr;
    unsigned lead = 0;       // leading all-zero rows among them
};
ExactMotif exact_motif(const lm_hip_pssm *p, const uint8_t *d_seq);
C32Plan plan_c32(const lm_hip_ctx *ctx, const ScoreArgs &a, bool store, int prefilter = 0, size_t batch = 1);

// the one-symbol prefilter scan of this job runs on 4-row symbol blocks (score_prefilter_blk.hpp): protein, dword-aligned matrix
static inline bool block_scan(const lm_hip_ctx *ctx, const ScoreArgs &a)
{
    return ctx->block_prefilter && a.pssm->k == (size_t)kBlkKA && reinterpret_cast<uintptr_t>(a.d_seq) % 4 == 0;
}

dim3 generic_grid(const lm_hip_ctx *ctx, unsigned long long ncells);
size_t generic_lds(const lm_hip_pssm *p, int *use_lds);

template <int MODE>
int launch_generic(lm_hip_ctx *ctx, const ScoreArgs &a, const FusedOut &fo, dim3 grid,
                          hipStream_t stream = nullptr)
{
    if (!stream)
        stream = ctx->stream;
    int use_lds = 0;
    const size_t lds = generic_lds(a.pssm, &use_lds);
    hipLaunchKernelGGL((score_generic<MODE>), grid, dim3(kBlock), lds, stream, a.d_seq,
                       (unsigned long long)a.seq_stride, (int)a.cols, a.pssm->d_dense,
                       (int)a.pssm->m, (int)a.pssm->k, use_lds,
                       (unsigned long long)a.row_begin, (unsigned long long)a.row_end, a.d_out,
                       (unsigned long long)a.out_stride, fo);
    LM_HIP_TRY(hipGetLastError());
    return LM_HIP_OK;
}

// ---- batches (score_plan.hip) -----------------------------------------------------------------------------

// Enqueues the fused argmax of one job: block records -> `blocks`, result -> `d_result`.
// Independent jobs of a batch alternate between the context's stream and an auxiliary
// one, so the tail of one motif's kernel (the last, partially filled round of
// workgroups) overlaps the head of the next.  fork: aux waits for everything already
// enqueued on the main stream; join: the main stream waits for aux.
int batch_fork(lm_hip_ctx *ctx);
int batch_join(lm_hip_ctx *ctx);

// Jobs of a batch that can share ONE launch (grid.y = jobs): same kernel, motif length,
// alphabet and sequence rows.  Many short per-motif launches lose ~15 % to their ramps
// and to the short streams a small grid needs; a launch per motif LENGTH keeps streams
// long and the chip full (2 346 JASPAR motifs -> ~50 launches).
enum : int { KIND_GENERIC = 0, KIND_EXACT = 1, KIND_PREFILTER = 2, KIND_PREFILTER2 = 3, KIND_CHUNKED = 4, KIND_SKIP = 5 };
static inline bool kind_solo(int kind) { return kind == KIND_GENERIC || kind == KIND_CHUNKED; }
// LDS table bytes a scan of `kind` reads per position for a motif of `m` rows of which `scanned` are looked up (the pair scans'
// drop-last form: m - 1): pair scans one row of ((scanned | 3) + 1) u16 entries per TWO positions, one-symbol scans a row of
// prefilter_mp u16 entries per position, exact kernels m floats
static inline unsigned scan_lds_bytes(int kind, size_t scanned, size_t k)
{
    return kind == KIND_PREFILTER2  ? (unsigned)((scanned | 3) + 1)
           : kind == KIND_PREFILTER ? 2u * (unsigned)prefilter_mp((int)scanned, lds_wide((int)k))
           : kind == KIND_EXACT     ? 4u * (unsigned)scanned
                                    : 0u;
}

// B = the score of a best k-mer: the row maxima added in motif order.  It bounds every score of the matrix from
// above -- f32 rounding is monotone, so termwise larger weights added in the same order cannot give a smaller
// sum -- provided the weights hold no NaN / +inf (lm_hip_pssm::has_prefilter).
float best_kmer_score(const lm_hip_pssm *p);

struct JobGroup {
    int kind = KIND_GENERIC;
    std::vector<size_t> idx;  // job indices, ascending
    C32Plan plan;
};

template <typename KindOf>
std::vector<JobGroup> group_jobs(const lm_hip_ctx *ctx, const ScoreArgs *jobs, size_t n,
                                        KindOf kind_of)
{
    typedef std::tuple<int, size_t, size_t, const uint8_t *, size_t, size_t> Key;
    std::map<Key, size_t> where;
    std::vector<JobGroup> groups;
    for (size_t i = 0; i < n; ++i) {
        const ScoreArgs &a = jobs[i];
        const int kind = kind_of(i);
        if (kind == KIND_SKIP)
            continue;  // provably nothing to report: no launch
        if (kind_solo(kind)) {  // one launch (or chain of launches) each
            groups.push_back(JobGroup{kind, {i}, C32Plan{}});
            continue;
        }
        const Key key(kind, a.pssm->m, a.pssm->k, a.d_seq, a.row_begin, a.row_end);
        auto it = where.find(key);
        if (it == where.end() || groups[it->second].idx.size() >= 32768) {  // grid.y <= 65535
            where[key] = groups.size();
            groups.push_back(JobGroup{kind, {i}, C32Plan{}});
        } else {
            groups[it->second].idx.push_back(i);
        }
    }
    for (JobGroup &g : groups)
        if (!kind_solo(g.kind)) {
            g.plan = plan_c32(ctx, jobs[g.idx[0]], false,
                              g.kind == KIND_PREFILTER2 ? 2 : (g.kind == KIND_PREFILTER ? 1 : 0), g.idx.size());
            g.plan.grid.y = (unsigned)g.idx.size();
        }
    return groups;
}

// ---- fused reductions of sliced motifs (M > kMaxFastM at C = 32) -----------------------------------
//
// No fused kernel holds more than kMaxFastM accumulators, and one thread per cell costs 23-50 ms per
// Gbp.  Such a job is scored in CHUNKS of ctx->chunk_rows rows through the sliced store path
// (launch_score_store: first slice stored, further slices continued in place) into one reusable
// buffer, and each chunk is reduced right behind its last slice: block maxima with
// global indices (argmax) or direct appends to the hit list (threshold).  Same values as the
// materialised matrix, hence the same results; the buffer never exceeds 512 MB however long the
// sequence.
// The same detour pays for every column count other than 32 (C = 16: the unrolled four-stream store kernel,
// 690 Gpos/s; the others: score_tiled into DENSE rows -- at C = 1 that is 4 instead of 32 bytes written per
// position) once the input is large enough for two more launches not to matter: their fused forms
// otherwise run one thread per cell (30-70 Gpos/s).
bool chunked_ok(const lm_hip_ctx *ctx, const ScoreArgs &a);
unsigned tiled_records(const lm_hip_ctx *ctx, const ScoreArgs &a);  // score_store.hip: records of the tiled store kernel, 0 = not its shape
unsigned long long chunk_rows_for(const lm_hip_ctx *ctx, const ScoreArgs &a);  // ctx->chunk_rows is quoted for C = 32
unsigned long long chunk_count(const lm_hip_ctx *ctx, const ScoreArgs &a);
unsigned chunk_argmax_grid(const lm_hip_ctx *ctx);  // workgroups of the per-chunk argmax (a full chunk is 2^25 cells)

// f(buffer, first row of the chunk relative to a.row_begin, rows of the chunk) after each chunk's scores
// are enqueued on ctx->stream
template <typename PerChunk>
int for_each_scored_chunk(lm_hip_ctx *ctx, const ScoreArgs &a, PerChunk f)
{
    const unsigned long long n = a.row_end - a.row_begin;
    const unsigned long long chunk = std::min<unsigned long long>(n, chunk_rows_for(ctx, a));
    LM_TRY(ctx->chunk_scores.reserve(chunk * a.cols * sizeof(float)));
    float *buf = static_cast<float *>(ctx->chunk_scores.ptr);
    for (unsigned long long c0 = 0; c0 < n; c0 += chunk) {
        const unsigned long long c1 = std::min(n, c0 + chunk);
        ScoreArgs b = a;
        b.row_begin = a.row_begin + c0;
        b.row_end = a.row_begin + c1;
        b.d_out = buf;
        b.out_stride = a.cols;  // contiguous rows: the flat reductions apply
        LM_TRY(launch_score_store(ctx, b));
        LM_TRY(f(buf, c0, c1 - c0));
    }
    ctx->last_kernel = a.cols == 32 ? "score_c32_sliced+reduce" : "score_store+reduce";
    return LM_HIP_OK;
}

// ---- candidates -> exact hits (score_threshold.hip) ------------------------------------------------------------

// Exact re-scoring of the candidate row ranges the fused threshold kernels flagged
// (the GPU form of scan.rs:187-190: `score_position` on the prefilter's candidates).
// One half-wave per candidate piece (<= 32 rows of one column): lane L re-computes
// output row r0 + L with the reference's add order -- M sequential f32 adds from
// +0.0, pli/mod.rs:98-102 -- and appends it to the hit list when score >= t.  The
// symbol loads of neighbouring lanes are 32 bytes apart and the windows of
// neighbouring rows overlap, so a piece touches ~(nrows + M) cache sectors once.
struct RescoreJob {
    const uint8_t *seq;    // row `row_begin` of the striped matrix (C = 32, stride 32)
    const float *dense;    // M x K weights, row-major
    unsigned m, k;
    float threshold;
    unsigned pad;
    unsigned long long key_rows;
};

// `so` on: one job, handed to the kernel by value (d_jobs is not read), records counted per bucket (hits.hip)
int launch_rescore(lm_hip_ctx *ctx, hipStream_t st, const RescoreJob *d_jobs, const FusedOut &fo, const RescoreJob *host_jobs,
                   size_t n, const ShortOrder *so = nullptr);

// ---- reductions of block records (score_argmax.hip, score_store.hip) ------------------------------------------

int ensure_ticket(lm_hip_ctx *ctx);  // the "last workgroup folds" counter of the single-launch argmax forms

__global__ void argmax_finalize(const ArgmaxRecord *__restrict__ blocks, const unsigned nblocks, const float *__restrict__ scores00,
                                const uint8_t *__restrict__ seq00, const unsigned long long seq_stride,
                                const float *__restrict__ pssm, const int M, const int K, const int first_cell_rule,
                                ArgmaxRecord *__restrict__ out);
__global__ void argmax_fold(const ArgmaxRecord *__restrict__ blocks, const unsigned nblocks, ArgmaxRecord *__restrict__ out);

}  // namespace lm
