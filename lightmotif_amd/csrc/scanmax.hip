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
// window (u8 scores + f32 scores into two reusable buffers; windows start small and double, the walk makes most of
// its updates early), and inside a window the walk is a chain of "find the FIRST cell at or after the cursor that
// the reference would act on" -- a parallel search with an ordered minimum -- followed by a one-thread state
// update.  The expected number of updates over n cells of continuous scores is ~ln n, less than one per doubling
// window, so the whole walk costs about two passes over the two score matrices: ~5 ms per Gbp instead of
// downloading 5 GB of scores to walk them on the host.
#include <algorithm>

#include "score_kernels.hpp"

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
    int pad;
};

namespace {

// The first cell at or after the cursor the reference's loop would ACT on: u8 score >= level and -- no hit held, or
// a greater f32 score, or an equal one at a greater position (scan.rs:229-242) -- or a window that leaves the matrix.
__global__ __launch_bounds__(kBlock) void scanmax_find(const uint8_t *__restrict__ d, const float *__restrict__ s,
                                                       const unsigned long long ncells, const unsigned cols,
                                                       const unsigned long long row0, const unsigned long long rows,
                                                       const unsigned m, ScanMaxState *__restrict__ st)
{
    const unsigned long long cursor = st->cursor;
    if (cursor >= ncells)
        return;
    const unsigned level = st->level;
    const int have = st->have;
    const float best = st->score;
    const unsigned long long best_index = st->index;
    const unsigned long long total = rows * cols;
    constexpr unsigned long long CH = (unsigned long long)kBlock * 16;  // cells per workgroup chunk
    const unsigned long long first_chunk = cursor / CH;
    for (unsigned long long c = first_chunk + blockIdx.x;; c += gridDim.x) {
        const unsigned long long c0 = c * CH;
        // chunks are taken in ascending order: nothing at or beyond an already found cell can be the first
        if (c0 >= ncells || c0 >= __hip_atomic_load(&st->found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            return;
        const unsigned long long f0 = c0 + 16ull * threadIdx.x;
        unsigned long long mine = ~0ull;
        if (f0 < ncells) {
            unsigned char b[16];
            if (f0 + 16 <= ncells && (reinterpret_cast<uintptr_t>(d + f0) & 15) == 0) {
                *reinterpret_cast<uint4 *>(b) = *reinterpret_cast<const uint4 *>(d + f0);
            } else {
                for (int i = 0; i < 16; ++i)
                    b[i] = f0 + i < ncells ? d[f0 + i] : 0;
            }
            for (int i = 0; i < 16 && mine == ~0ull; ++i) {
                const unsigned long long f = f0 + i;
                if (f < cursor || f >= ncells || b[i] < level)
                    continue;
                const unsigned long long r = row0 + f / cols, col = f % cols;
                const unsigned long long index = col * rows + r;  // scan.rs:231
                bool act = index + m > total;                    // score_position would index column C
                if (!act) {
                    const float x = s[f];
                    act = !have || x > best || (x == best && index > best_index);
                }
                if (act)
                    mine = f;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(mine, off);
            mine = o < mine ? o : mine;
        }
        if ((threadIdx.x & 63) == 0 && mine != ~0ull)
            atomicMin(&st->found, mine);
    }
}

__global__ void scanmax_apply(const uint8_t *__restrict__ d, const float *__restrict__ s, const unsigned long long ncells,
                              const unsigned cols, const unsigned long long row0, const unsigned long long rows,
                              const unsigned m, ScanMaxState *__restrict__ st, ScanMaxState *__restrict__ host_copy)
{
    ScanMaxState t = *st;
    if (t.found != ~0ull && !t.err) {
        const unsigned long long f = t.found;
        const unsigned long long r = row0 + f / cols, col = f % cols;
        const unsigned long long index = col * rows + r;
        if (index + m > rows * cols) {
            t.err = 1;
            t.index = index;
            t.cursor = ncells;
            t.more = 0;
        } else {
            if (t.have)
                t.level = d[f];  // scan.rs:238 best_discrete = dscore (the first hit keeps the scaled threshold, :241)
            t.have = 1;
            t.score = s[f];      // = score_position (pwm/mod.rs:651-662): the same M sequential f32 adds
            t.index = index;
            t.cursor = f + 1;
            t.more = 1;
        }
    } else {
        t.more = 0;
        t.cursor = ncells;
    }
    t.found = ~0ull;
    *st = t;
    *host_copy = t;
}

__global__ void scanmax_window(ScanMaxState *__restrict__ st)
{
    st->cursor = 0;
    st->found = ~0ull;
    st->more = 0;
}

}  // namespace

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
    // window buffers: u8 scores | f32 scores, dense rows (stride = cols)
    // (~64 M cells per window = 320 MB of u8 + f32 scores; matrices of very many columns keep at least 64 rows)
    const size_t max_rows = std::max<size_t>(((size_t)64 << 20) / cols, 64);
    const size_t wrows_cap = std::min(max_rows, rows - first_row);
    const size_t d_bytes = (wrows_cap * cols + 255) / 256 * 256;
    // (the state block holds 64-bit words the search updates atomically: both score buffers are rounded to 256 B)
    const size_t s_bytes = (wrows_cap * cols * sizeof(float) + 255) / 256 * 256;
    LM_TRY(ctx->scan_buf.reserve(d_bytes + s_bytes + 256));
    uint8_t *d_d = static_cast<uint8_t *>(ctx->scan_buf.ptr);
    float *d_s = reinterpret_cast<float *>(d_d + d_bytes);
    ScanMaxState *d_st = reinterpret_cast<ScanMaxState *>(reinterpret_cast<char *>(d_s) + s_bytes);
    ScanMaxState *h_st = static_cast<ScanMaxState *>(ctx->pinned);
    ScanMaxState init{};
    init.index = position;
    init.cursor = 0;
    init.found = ~0ull;
    init.score = score;
    init.level = level;
    init.have = have ? 1 : 0;
    h_st[1] = init;  // staged through the pinned block ([0] receives the kernel's copies)
    LM_HIP_TRY(hipMemcpyAsync(d_st, &h_st[1], sizeof(ScanMaxState), hipMemcpyHostToDevice, ctx->stream));
    const unsigned grid = (unsigned)ctx->num_cus * 8;
    size_t w = std::min<size_t>(4096, wrows_cap);
    for (size_t r = first_row; r < rows;) {
        const size_t rb = std::min(rows, r + w);
        const unsigned long long ncells = (unsigned long long)(rb - r) * cols;
        DiscreteArgs da{weights, m, wstride, pssm->k, seq->d_data, seq->stride, cols, r, rb, d_d, cols, saturate};
        LM_TRY(launch_score_u8(ctx, da));
        ScoreArgs sa{pssm, seq->d_data, seq->stride, cols, r, rb, d_s, cols};
        LM_TRY(launch_score_store(ctx, sa));
        hipLaunchKernelGGL(scanmax_window, dim3(1), dim3(1), 0, ctx->stream, d_st);
        for (;;) {
            for (int rep = 0; rep < 3; ++rep) {  // a few rounds per synchronisation: most windows hold 0-1 updates
                hipLaunchKernelGGL(scanmax_find, dim3(grid), dim3(kBlock), 0, ctx->stream, d_d, d_s, ncells, (unsigned)cols,
                                   (unsigned long long)r, (unsigned long long)rows, (unsigned)m, d_st);
                hipLaunchKernelGGL(scanmax_apply, dim3(1), dim3(1), 0, ctx->stream, d_d, d_s, ncells, (unsigned)cols,
                                   (unsigned long long)r, (unsigned long long)rows, (unsigned)m, d_st, h_st);
            }
            LM_HIP_TRY(hipGetLastError());
            LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (h_st[0].err)
                return fail(LM_HIP_ERR_BAD_ARGS,
                            "Scanner::max: the window of candidate position %llu (+ %zu rows) leaves the striped matrix; the "
                            "reference panics here (seq.rs:433-442)", h_st[0].index, m);
            if (!h_st[0].more)
                break;
        }
        r = rb;
        w = std::min(2 * w, wrows_cap);
    }
    ctx->last_kernel = "scanmax_find";
    *found = h_st[0].have;
    *best_position = h_st[0].index;
    *best_score = h_st[0].score;
    return LM_HIP_OK;
}

}  // namespace lm
