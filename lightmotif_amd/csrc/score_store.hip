// score_store.hip -- Score::score_rows_into (pli/mod.rs:72-106): the materialising launches -- f32 store kernels (with
// and without the tracked maximum), u8 scores of a DiscreteMatrix, and the folds of their argmax records.
#include <algorithm>
#include <cstring>

#include "score_launch.hpp"

namespace lm {

// ---- Store --------------------------------------------------------------------------

// Plain store kernel only: (padded) length 12 at C = 32 runs fastest with ONE group per stream, T = 13
// (0.916 vs 0.954 ms per Gbp at T = 37, reproduced on three runs and for M = 10, 11 padded to 12;
// profiles/r02_tsweep_short_streams.txt).  The same choice is wrong for every other length (M' = 8: 0.998,
// 16: 1.02, 20: 1.06), so it is a table entry, not a rule.  0 = the planner's default.
static unsigned long long store_rows_hint(size_t m_kernel, size_t cols)
{
    return (cols == 32 && m_kernel == 12) ? 12 : 0;
}

// Geometry of the LDS-tiled store kernel (any column count): workgroups of `*tr` rows whose tile -- the dense table +
// tr + M - 1 rows of `cols` symbols -- fits 40 KB.  Returns the grid size, 0 when the shape goes cell by cell.
static unsigned tiled_plan(const lm_hip_ctx *ctx, const ScoreArgs &a, unsigned long long *tr_out, size_t *lds_out)
{
    if (!ctx->tiled)
        return 0;
    const size_t tab_bytes = (a.pssm->m * a.pssm->k * 4 + 15) / 16 * 16;
    const size_t budget = 40 * 1024;
    const unsigned long long n = a.row_end - a.row_begin;
    if (!(a.pssm->m >= 1 && tab_bytes + (a.pssm->m + 8) * a.cols <= budget && a.cols <= 4096))
        return 0;
    unsigned long long tr = (budget - tab_bytes) / a.cols - (a.pssm->m - 1);
    tr = std::min<unsigned long long>(tr / kTiledStrip * kTiledStrip, 2048);
    // enough workgroups to fill the chip
    while (tr > kTiledStrip * 4 && (n + tr - 1) / tr < (unsigned long long)ctx->num_cus * 4)
        tr = (tr / 2 + kTiledStrip - 1) / kTiledStrip * kTiledStrip;
    if (tr < (unsigned long long)kTiledStrip)
        return 0;
    *tr_out = tr;
    *lds_out = tab_bytes + (tr + a.pssm->m - 1) * a.cols + 16;
    return (unsigned)((n + tr - 1) / tr);
}

// workgroup records the tiled kernel would leave for `a` (0: the shape does not go through it)
unsigned tiled_records(const lm_hip_ctx *ctx, const ScoreArgs &a)
{
    unsigned long long tr = 0;
    size_t lds = 0;
    return tiled_plan(ctx, a, &tr, &lds);
}

int launch_score_store(lm_hip_ctx *ctx, const ScoreArgs &a)
{
    FusedOut fo{};
    const bool wide = lds_wide((int)a.pssm->k);
    const bool dwords = ctx->quad_loads && reinterpret_cast<uintptr_t>(a.d_seq) % 4 == 0;
    if (dwords && a.pssm->d_table_pad) {
        // M % 4 != 0: the table padded with leading zero rows to M' = 4 * ceil(M / 4) -- the same f32
        // sums (0.0 + 0.0 + P[0] ... ), with the dword symbol loads and 4-row blocks of the M' kernel
        const size_t mp = a.pssm->m + a.pssm->lead;
        const MotifShape ms{mp, a.pssm->k, false};
        const C32Plan pp = plan_c32(ctx, ms, a, true, 0, 1, store_rows_hint(mp, a.cols), true);
        ScoreC32Launcher pfn = a.cols == 16 ? score_c32_lookup_c16((int)mp, wide) : score_c32_lookup_ql((int)mp, wide);
        if (pp.ok && pfn) {
            fo.lead_rows = (unsigned)a.pssm->lead;
            ctx->last_kernel = score_c32_name((int)mp, MODE_STORE);
            LM_HIP_TRY(pfn(pp.grid, pp.lds, ctx->stream, a.d_seq, a.pssm->d_table_pad,
                                                    (int)a.pssm->k, a.row_begin, a.row_end, pp.T, pp.nstreams, a.d_out,
                                                    fo));
            return LM_HIP_OK;
        }
    }
    const bool c16 = a.cols == 16 && dwords && score_c32_lookup_c16((int)a.pssm->m, wide);
    const C32Plan p = a.pssm->m <= (size_t)kMaxFastM ? plan_c32(ctx, MotifShape{a.pssm->m, a.pssm->k, false}, a, true, 0, 1,
                                                                  store_rows_hint(a.pssm->m, a.cols), c16)
                                                      : C32Plan{};  // longer: the slices below
    if (p.ok) {
        ScoreC32Launcher fn = score_c32_lookup((int)a.pssm->m, MODE_STORE, wide);
        if (dwords && score_c32_lookup_ql((int)a.pssm->m, wide))
            fn = score_c32_lookup_ql((int)a.pssm->m, wide);  // dword symbol loads (M % 4 == 0)
        if (c16)
            fn = score_c32_lookup_c16((int)a.pssm->m, wide);  // four streams of 16 columns per wavefront
        ctx->last_kernel = score_c32_name((int)a.pssm->m, MODE_STORE);
        LM_HIP_TRY(fn(p.grid, p.lds, ctx->stream, a.d_seq, a.pssm->d_table, (int)a.pssm->k,
                      a.row_begin, a.row_end, p.T, p.nstreams, a.d_out, fo));
        return LM_HIP_OK;
    }
    // motifs longer than kMaxFastM at C = 32: slices of <= kMaxFastM rows.  The first slice is an ordinary
    // store pass; every further slice continues IN PLACE from the partial sums (MODE_CONTINUE: same add
    // order, bit-identical), over whole streams only -- a cell must be read and rewritten exactly once,
    // so no shifted or repeated stream -- and the few rows left over go cell by cell.
    if (!a.pssm->parts.empty() && a.cols == 32 && a.seq_stride == 32 && a.out_stride == 32 &&
        reinterpret_cast<uintptr_t>(a.d_seq) % 4 == 0 && a.row_end - a.row_begin > a.pssm->parts[0].m) {
        const unsigned long long n = a.row_end - a.row_begin;
        bool ok = true;
        for (size_t i = 0; i < a.pssm->parts.size() && ok; ++i) {
            const lm_hip_pssm::Part &part = a.pssm->parts[i];
            const MotifShape ms{part.m, a.pssm->k, false};
            ScoreArgs sa = a;
            sa.d_seq = a.d_seq + part.off * a.seq_stride;  // slice row j reads sequence row r + off + j
            const C32Plan p = plan_c32(ctx, ms, sa, true);
            if (!p.ok) {
                ok = false;
                break;
            }
            FusedOut pfo = fo;
            pfo.lead_rows = (unsigned)part.lead;
            if (i == 0) {
                ScoreC32Launcher fn = score_c32_lookup_ql((int)part.m, wide);
                if (!fn)
                    fn = score_c32_lookup((int)part.m, MODE_STORE, wide);
                LM_HIP_TRY(fn(p.grid, p.lds, ctx->stream, sa.d_seq, part.d_table, (int)a.pssm->k, a.row_begin, a.row_end,
                              p.T, p.nstreams, a.d_out, pfo));
                continue;
            }
            const unsigned long long nfull = n / p.T;
            if (nfull) {
                const dim3 grid((unsigned)((nfull + kStreamsPerBlock - 1) / kStreamsPerBlock));
                LM_HIP_TRY(score_c32_lookup_continue((int)part.m, wide)(grid, p.lds, ctx->stream, sa.d_seq, part.d_table,
                                                                  (int)a.pssm->k, a.row_begin, a.row_begin + nfull * p.T,
                                                                  p.T, nfull, a.d_out, pfo));
            }
            if (nfull * p.T < n) {
                const unsigned long long r0 = a.row_begin + nfull * p.T;
                const unsigned long long cells = (a.row_end - r0) * a.cols;
                hipLaunchKernelGGL(score_continue_cells<0>, dim3((unsigned)((cells + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                                   ctx->stream, sa.d_seq, (unsigned long long)a.seq_stride, (int)a.cols,
                                   a.pssm->d_dense + part.off * a.pssm->k, (int)(part.m - part.lead), (int)a.pssm->k, r0,
                                   (unsigned long long)a.row_end, a.d_out + (r0 - a.row_begin) * a.out_stride,
                                   (unsigned long long)a.out_stride);
                LM_HIP_TRY(hipGetLastError());
            }
        }
        if (ok) {
            ctx->last_kernel = a.pssm->parts.size() == 1 ? score_c32_name((int)a.pssm->parts[0].m, MODE_STORE)
                                                         : "score_c32_sliced";
            return LM_HIP_OK;
        }
    }
    // any other geometry: the tiled kernel when its LDS tile fits (the dense table + TR + M - 1 rows of
    // `cols` symbols), else one thread per cell (context option "tiled" = 0: always, for A/B runs)
    unsigned long long tr = 0;
    size_t lds = 0;
    const unsigned grid_x = tiled_plan(ctx, a, &tr, &lds);
    if (grid_x) {
        ctx->last_kernel = "score_tiled";
        const dim3 grid(grid_x);
        const size_t nrec = grid.x;  // one record per workgroup
        const unsigned long long n = a.row_end - a.row_begin;
        const bool track = a.track_records && nrec + 1 <= a.track_cap && n * a.cols < (1ull << 32);
        if (a.track_nrec)
            *a.track_nrec = track ? (unsigned)nrec : 0u;
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, grid, dim3(kBlock), lds, ctx->stream, a.d_seq, (unsigned long long)a.seq_stride,
                               (int)a.cols, a.pssm->d_dense, (int)a.pssm->m, (int)a.pssm->k,
                               (unsigned long long)a.row_begin, (unsigned long long)a.row_end, (int)tr, a.d_out,
                               (unsigned long long)a.out_stride, track ? a.track_records : (uint4 *)nullptr,
                               a.track_generation);
        };
        if (a.pssm->k == 5)
            launch(score_tiled<kTiledStrip, 5>);
        else if (a.pssm->k == 21)
            launch(score_tiled<kTiledStrip, 21>);
        else
            launch(score_tiled<kTiledStrip, 0>);
        LM_HIP_TRY(hipGetLastError());
        return LM_HIP_OK;
    }
    ctx->last_kernel = "score_generic<0>";
    const unsigned long long ncells = (unsigned long long)(a.row_end - a.row_begin) * a.cols;
    return launch_generic<MODE_STORE>(ctx, a, fo, generic_grid(ctx, ncells));
}

// ---- Store, u8 scores of a DiscreteMatrix --------------------------------------------------

// Any geometry: one thread per cell, weights read from a dense M x K byte table (cached).
__global__ __launch_bounds__(kBlock) void score_generic_u8(
    const uint8_t *__restrict__ seq, const unsigned long long seq_stride, const unsigned cols,
    const uint8_t *__restrict__ dense, const unsigned m, const unsigned k,
    const unsigned long long row_begin, const unsigned long long row_end, uint8_t *__restrict__ out,
    const unsigned long long out_stride, const unsigned wrap_mask)
{
    const unsigned long long ncells = (row_end - row_begin) * cols;
    for (unsigned long long cell = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; cell < ncells;
         cell += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long r = cell / cols;
        const unsigned c = (unsigned)(cell - r * cols);
        const uint8_t *sp = seq + (row_begin + r) * seq_stride + c;
        unsigned sum = 0;  // exact: m * 255 fits easily
        for (unsigned j = 0; j < m; ++j)
            sum += dense[j * k + sp[j * seq_stride]];
        out[r * out_stride + c] = (uint8_t)(wrap_mask ? (sum & wrap_mask) : (sum < 255u ? sum : 255u));
    }
}

// out[i] = out[i] + add[i] per byte, saturating at 255 or wrapping mod 256, four cells per lane-word.
__global__ __launch_bounds__(kBlock) void u8_combine(unsigned *__restrict__ out, const unsigned *__restrict__ add,
                                                     const unsigned long long nwords, const int saturate)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < nwords;
         i += (unsigned long long)gridDim.x * kBlock) {
        const unsigned a = out[i], b = add[i];
        const unsigned sum = ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);  // bytewise, no carries across
        const unsigned carry = ((a & b) | ((a | b) & ~sum)) & 0x80808080u;                      // bytes that overflowed
        out[i] = saturate ? (sum | ((carry >> 7) * 0xffu)) : sum;
    }
}

int launch_score_u8(lm_hip_ctx *ctx, const DiscreteArgs &a)
{
    // Motifs beyond kMaxFastM at C = 32: slices of <= kMaxFastM rows through the fast kernels, the first into the
    // score matrix, the others into a temporary, added bytewise.  Exact for both flavours: u8 weights are
    // non-negative, so saturating partial sums combine to min(255, total); wrapping sums are sums mod 256.
    if (a.m > (size_t)kMaxFastM && a.cols == 32 && a.seq_stride == 32 && a.out_stride == 32 &&
        reinterpret_cast<uintptr_t>(a.d_out) % 4 == 0 && a.row_end > a.row_begin + (size_t)kMaxFastM) {
        const size_t nslices = (a.m + kMaxFastM - 1) / kMaxFastM, len = (a.m + nslices - 1) / nslices;
        const unsigned long long n = a.row_end - a.row_begin;
        LM_TRY(ctx->chunk_scores.reserve(n * 32));
        uint8_t *tmp = static_cast<uint8_t *>(ctx->chunk_scores.ptr);
        for (size_t off = 0; off < a.m; off += len) {
            DiscreteArgs part = a;
            part.weights = a.weights + off * a.wstride;
            part.m = std::min(len, a.m - off);
            part.d_seq = a.d_seq + off * a.seq_stride;  // slice row j reads sequence row r + off + j
            part.d_out = off == 0 ? a.d_out : tmp;
            LM_TRY(launch_score_u8(ctx, part));
            if (off) {
                const unsigned long long nwords = n * 32 / 4;
                const unsigned grid = (unsigned)std::min<unsigned long long>((nwords + kBlock - 1) / kBlock,
                                                                             (unsigned long long)ctx->num_cus * 16);
                hipLaunchKernelGGL(u8_combine, dim3(grid), dim3(kBlock), 0, ctx->stream,
                                   reinterpret_cast<unsigned *>(a.d_out), reinterpret_cast<const unsigned *>(tmp), nwords,
                                   a.saturate ? 1 : 0);
                LM_HIP_TRY(hipGetLastError());
            }
        }
        ctx->last_kernel = "score_c32_u8_sliced";
        return LM_HIP_OK;
    }
    const int m = (int)a.m, k = (int)a.k;
    const unsigned wrap_mask = a.saturate ? 0u : 0xffu;
    // plan with the f32 planner: same stream geometry as the packed prefilter scans.  DNA takes
    // the pair-symbol scan (two rows per lookup) when the matrix allows dword symbol loads.
    const MotifShape ms{a.m, a.k, true};
    ScoreArgs sa{nullptr, a.d_seq, a.seq_stride, a.cols, a.row_begin, a.row_end, nullptr, a.out_stride};
    // (both fast kernels write dwords: the score matrix must be 4-byte aligned)
    const bool out_aligned = reinterpret_cast<uintptr_t>(a.d_out) % 4 == 0;
    // streams of ~128 rows: 1 B + 1 B per cell leaves the kernel between the f32 store kernel
    // (HBM-bound, short streams) and the scans (issue-bound, long streams); measured at 1 Gbp
    // x M = 20: T = 64 0.449 ms, 128 0.426, 256 0.434, 1024 0.456, 4096 0.486
    // (the u8 pair kernel exists for DNA only)
    C32Plan p = (out_aligned && ctx->pair_prefilter && a.k == 5) ? plan_c32(ctx, ms, sa, true, 2, 1, 128) : C32Plan();
    const bool pairs = p.ok;
    if (!pairs && out_aligned)
        p = plan_c32(ctx, ms, sa, true, 1, 1, 128);
    ScoreU8Launcher fn = p.ok ? score_c32_lookup_u8(m, pairs, lds_wide(k)) : nullptr;
    // device copies (scratch2): [packed image | dense table]
    const size_t image_bytes = !fn ? 0 : pairs ? (size_t)prefilter2_image_dw(m) * 4 : (size_t)prefilter_image_dw(m, k) * 4;
    const size_t dense_bytes = ((size_t)m * k + 15) / 16 * 16;
    // The tables live in a buffer of their own and are rebuilt only when the matrix changes.
    // (The call returns without synchronising, so they are staged in pageable memory: the
    // runtime copies that out before hipMemcpyAsync returns, whereas the shared pinned buffer
    // could be overwritten by the next call while the copy is still queued.)
    std::vector<uint8_t> key{(uint8_t)(m & 0xff), (uint8_t)((m >> 8) & 0xff), (uint8_t)((m >> 16) & 0xff),
                             (uint8_t)k, (uint8_t)(fn ? (pairs ? 2 : 1) : 0)};
    for (int j = 0; j < m; ++j)
        key.insert(key.end(), a.weights + (size_t)j * a.wstride, a.weights + (size_t)j * a.wstride + k);
    if (key != ctx->u8_key || ctx->u8_tables.bytes < image_bytes + dense_bytes) {
        LM_TRY(ctx->u8_tables.reserve(std::max<size_t>(image_bytes + dense_bytes, 64 * 1024)));
        std::vector<char> stage_buf(image_bytes + dense_bytes, 0);
        char *stage = stage_buf.data();
        if (fn) {
            const int mp = prefilter_mp(m, lds_wide(k)), shift = pairs ? 0 : mp - m;
            std::vector<unsigned> d((size_t)(m + shift) * k, 0u);
            for (int j = 0; j < m; ++j)
                for (int s = 0; s < k; ++s)
                    d[(size_t)(j + shift) * k + s] = a.weights[(size_t)j * a.wstride + s];
            if (pairs)
                prefilter2_pack_image(d.data(), m, reinterpret_cast<unsigned *>(stage));
            else
                prefilter_pack_image(d.data(), m, k, reinterpret_cast<unsigned *>(stage));
        }
        for (int j = 0; j < m; ++j)
            memcpy(stage + image_bytes + (size_t)j * k, a.weights + (size_t)j * a.wstride, (size_t)k);
        ctx->u8_key.clear();  // stays empty if the copy fails
        LM_HIP_TRY(hipMemcpyAsync(ctx->u8_tables.ptr, stage, image_bytes + dense_bytes, hipMemcpyHostToDevice,
                                  ctx->stream));
        ctx->u8_key = std::move(key);
    }
    char *dev = static_cast<char *>(ctx->u8_tables.ptr);
    if (fn) {
        ctx->last_kernel = pairs ? "score_c32_u8_pairs" : "score_c32_u8";
        LM_HIP_TRY(fn(p.grid, p.lds, ctx->stream, a.d_seq, reinterpret_cast<const unsigned *>(dev), k, a.row_begin,
                      a.row_end, p.T, p.nstreams, a.d_out, wrap_mask));
        return LM_HIP_OK;
    }
    ctx->last_kernel = "score_generic_u8";
    const unsigned long long ncells = (unsigned long long)(a.row_end - a.row_begin) * a.cols;
    hipLaunchKernelGGL(score_generic_u8, generic_grid(ctx, ncells), dim3(kBlock), 0, ctx->stream, a.d_seq,
                       (unsigned long long)a.seq_stride, (unsigned)a.cols,
                       reinterpret_cast<const uint8_t *>(dev + image_bytes), (unsigned)m, (unsigned)k,
                       (unsigned long long)a.row_begin, (unsigned long long)a.row_end, a.d_out,
                       (unsigned long long)a.out_stride, wrap_mask);
    LM_HIP_TRY(hipGetLastError());
    return LM_HIP_OK;
}

__global__ void argmax_fold(const ArgmaxRecord *__restrict__ blocks, const unsigned nblocks,
                            ArgmaxRecord *__restrict__ out);  // defined with the other reductions below

// Final step of the store+argmax flow, ONE workgroup: (1) reduce the (max value, workgroup)
// records with the Generic rule (greater value; ties -> later workgroup = later rows) and
// apply the first-cell NaN rule on the stored matrix; (2) find the LAST cell of the winning
// workgroup's rows whose stored score equals that value -- the Generic argmax (pli/mod.rs:
// 144-151).  The winning workgroup wrote <= 8 streams x T rows; they are re-read 16 bytes
// per lane.
__global__ __launch_bounds__(kBlock) void argmax_finalize_locate(
    const ArgmaxRecord *__restrict__ recs, const unsigned nrecs, const float *__restrict__ scores,
    const unsigned long long rows, const unsigned long long T, const unsigned long long nstreams,
    const int first_cell_rule, ArgmaxRecord *__restrict__ out)
{
    __shared__ float sm_v[kBlock / 64];
    __shared__ long long sm_i[kBlock / 64];
    __shared__ float win_v;
    __shared__ long long win_i;
    float v = -INFINITY;
    long long wg = -1;
    for (unsigned b = threadIdx.x; b < nrecs; b += kBlock)
        if (recs[b].found)
            best_merge(v, wg, recs[b].value, recs[b].index);
    best_block_reduce(v, wg, sm_v, sm_i);
    if (threadIdx.x == 0) {
        const float first = first_cell_rule ? scores[0] : 0.0f;  // row shards that do not hold row 0 skip the rule
        if (first != first) {  // scores[0][0] is NaN: nothing ever compares >= it (pli/mod.rs:142-146)
            ArgmaxRecord o;
            o.value = first;
            o.index = 0;
            o.found = 1;
            *out = o;
            wg = -2;
        } else if (wg < 0) {
            ArgmaxRecord o;
            o.value = v;
            o.index = -1;
            o.found = 0;
            *out = o;
        }
        win_v = v;
        win_i = wg;
    }
    __syncthreads();
    v = win_v;
    wg = win_i;
    if (wg < 0)
        return;
    // stream s covers rows [min(s*T, rows - T), +T): the workgroup's streams form one run
    unsigned long long s0 = (unsigned long long)wg * kStreamsPerBlock, s1 = s0 + kStreamsPerBlock;
    if (s1 > nstreams)
        s1 = nstreams;
    const unsigned long long last = rows - T;
    const unsigned long long lo = s0 * T < last ? s0 * T : last;
    const unsigned long long hi = ((s1 - 1) * T < last ? (s1 - 1) * T : last) + T;
    long long best = -1;
    const float4 *s4 = reinterpret_cast<const float4 *>(scores);
    for (unsigned long long i = lo * 8 + threadIdx.x; i < hi * 8; i += kBlock) {  // 8 float4 per row
        const float4 x = s4[i];
        if (x.x == v) best = (long long)(4 * i);
        if (x.y == v) best = (long long)(4 * i + 1);
        if (x.z == v) best = (long long)(4 * i + 2);
        if (x.w == v) best = (long long)(4 * i + 3);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        sm_i[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w)
            best = sm_i[w] > best ? sm_i[w] : best;
        ArgmaxRecord o;
        o.index = best;
        o.found = best >= 0;
        o.value = best >= 0 ? scores[best] : v;  // the cell's own bits (-0.0 == +0.0)
        *out = o;
    }
}

// Store + running best: the scores are written exactly like launch_score_store does; the
// kernel's per-workgroup (max value, workgroup) records are reduced (Generic rule, first-cell
// NaN rule on the stored matrix) and the cell is found in the winning
// workgroup's rows; the result lands in `d_result`, all on the same stream.  Returns
// *tracked = false (after a plain store) for shapes the C = 32 kernels do not cover.
int launch_score_store_argmax(lm_hip_ctx *ctx, const ScoreArgs &a, ArgmaxRecord *d_result, bool *tracked,
                              int first_cell_rule)
{
    *tracked = false;
    // (lengths that are no multiple of 4 run the padded table, see launch_score_store)
    const bool longm = a.pssm->m > (size_t)kMaxFastM;
    const ExactMotif em = longm ? exact_motif(a.pssm, a.d_seq) : ExactMotif{};
    const bool pad = longm ? em.m != 0 : (a.pssm->d_table_pad != nullptr && ctx->quad_loads);
    const size_t mk = longm ? em.m : a.pssm->m + (pad ? a.pssm->lead : 0);
    const unsigned lead = longm ? em.lead : (unsigned)a.pssm->lead;
    const float *pad_table = longm ? em.table : a.pssm->d_table_pad;
    const MotifShape ms{mk, a.pssm->k, false};
    const C32Plan p = mk >= 1 ? plan_c32(ctx, ms, a, true) : C32Plan{};
    ScoreC32Launcher fn = p.ok ? score_c32_lookup_store_argmax((int)mk, lds_wide((int)a.pssm->k)) : nullptr;
    if (!fn || reinterpret_cast<uintptr_t>(a.d_seq) % 4 != 0)
        return launch_score_store(ctx, a);
    if (a.out_stride != 32)
        return launch_score_store(ctx, a);
    const unsigned nrec = p.grid.x * (kBlock / 64);  // one record per wavefront
    LM_TRY(ctx->scratch.reserve(sizeof(ArgmaxRecord) * ((size_t)nrec + 256 + 1)));
    FusedOut fo{};
    fo.block_best = static_cast<ArgmaxRecord *>(ctx->scratch.ptr);
    ArgmaxRecord *folded = fo.block_best + nrec;
    fo.lead_rows = pad ? lead : 0u;
    ctx->last_kernel = score_c32_name((int)mk, MODE_STORE);
    LM_HIP_TRY(fn(p.grid, p.lds, ctx->stream, a.d_seq, pad ? pad_table : a.pssm->d_table, (int)a.pssm->k,
                  a.row_begin, a.row_end, p.T, p.nstreams, a.d_out, fo));
    const ArgmaxRecord *recs = fo.block_best;
    unsigned n = nrec;
    if (n > 4096) {  // ~256 K wavefront records per Gbp: fold them to 256 first
        hipLaunchKernelGGL(argmax_fold, dim3(256), dim3(kBlock), 0, ctx->stream, recs, n, folded);
        recs = folded;
        n = 256;
    }
    hipLaunchKernelGGL(argmax_finalize_locate, dim3(1), dim3(kBlock), 0, ctx->stream, recs, n, a.d_out,
                       (unsigned long long)(a.row_end - a.row_begin), p.T, p.nstreams, first_cell_rule, d_result);
    LM_HIP_TRY(hipGetLastError());
    *tracked = true;
    return LM_HIP_OK;
}

// Folds the per-wavefront records a small tracking kernel left in pinned memory (score_kernels.hpp:
// FusedOut::host_records) with the Generic rule -- greater value, ties to the later cell, NaN never, scores[0][0]
// NaN -> (0, 0) (pli/mod.rs:135-155) -- as soon as they have all arrived: each 8-byte half of a record carries the
// launch's generation, so a record is complete when both halves show it.  Arrival is polled (PCIe posted writes, a
// microsecond behind the wavefronts); a bounded spin, then the stream is synchronised and the records re-read.
int fold_host_records(lm_hip_ctx *ctx, const void *records, unsigned n, unsigned gen, bool first_cell_rule,
                      ArgmaxRecord *out)
{
    const volatile unsigned long long *rec = static_cast<const volatile unsigned long long *>(records);
    float v = -INFINITY;
    long long cell = -1;
    bool synced = false;
    for (unsigned i = 0; i <= n; ++i) {  // record n = the first-cell slot
        unsigned long long lo = 0, hi = 0;
        for (unsigned spin = 0;; ++spin) {
            lo = __atomic_load_n(&rec[2 * i], __ATOMIC_ACQUIRE);
            hi = __atomic_load_n(&rec[2 * i + 1], __ATOMIC_ACQUIRE);
            if ((unsigned)lo == gen && (unsigned)hi == gen)
                break;
            if (spin >= (1u << 20)) {
                if (synced)
                    return fail(LM_HIP_ERR_HIP, "argmax: the kernel's record %u of %u never arrived", i, n);
                LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
                synced = true;
                spin = 0;
            }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
            __builtin_ia32_pause();
#endif
        }
        const unsigned vbits = (unsigned)(lo >> 32), c = (unsigned)(hi >> 32);
        float x;
        memcpy(&x, &vbits, 4);
        if (i == n) {
            if (first_cell_rule && x != x) {  // scores[0][0] is NaN: nothing ever compares >= it
                v = x;
                cell = 0;
            }
        } else if (c != 0xffffffffu && (cell < 0 || x > v || (x == v && (long long)c > cell))) {
            v = x;
            cell = (long long)c;
        }
    }
    out->value = v;
    out->index = cell;
    out->found = cell >= 0;
    return LM_HIP_OK;
}

int ensure_ticket(lm_hip_ctx *ctx)
{
    if (ctx->d_ticket)
        return LM_HIP_OK;
    LM_HIP_TRY(hipMalloc(&ctx->d_ticket, 64));
    LM_HIP_TRY(hipMemsetAsync(ctx->d_ticket, 0, 64, ctx->stream));
    return LM_HIP_OK;
}

// Small inputs (the reference's own bench is 464 165 bp, lightmotif-bench dna.rs:81-109): `score_into` + `argmax`
// are launch-latency bound, so the store kernel tracks (value, cell) per lane and its last workgroup folds the
// workgroup records -- one launch, no copy command (the record is also written to *h_result, pinned).
// `host_fold` (the scores handle the rows go into): when its pinned record block can be had, the kernel folds
// nothing -- every wavefront writes one record there and lm_hip_argmax folds them on the host (*tracked stays false:
// there is no device record; host_fold->records_on_host says where the result is).
int launch_score_store_track(lm_hip_ctx *ctx, const ScoreArgs &a, ArgmaxRecord *d_result, ArgmaxRecord *h_result,
                             unsigned generation, bool *tracked, int first_cell_rule, lm_hip_scores *host_fold)
{
    *tracked = false;
    const bool longm = a.pssm->m > (size_t)kMaxFastM;
    const ExactMotif em = longm ? exact_motif(a.pssm, a.d_seq) : ExactMotif{};
    const bool pad = longm ? em.m != 0 : (a.pssm->d_table_pad != nullptr && ctx->quad_loads);
    const size_t mk = longm ? em.m : a.pssm->m + (pad ? a.pssm->lead : 0);
    const unsigned lead = longm ? em.lead : (unsigned)a.pssm->lead;
    const float *table = pad ? (longm ? em.table : a.pssm->d_table_pad) : a.pssm->d_table;
    const C32Plan p = (mk >= 1 && mk % 4 == 0 && a.cols == 32 && a.out_stride == 32 && ctx->quad_loads &&
                       reinterpret_cast<uintptr_t>(a.d_seq) % 4 == 0)
                          ? plan_c32(ctx, MotifShape{mk, a.pssm->k, false}, a, true, 0, 1, store_rows_hint(mk, a.cols))
                          : C32Plan{};
    ScoreC32Launcher fn = p.ok ? score_c32_lookup_store_track((int)mk, lds_wide((int)a.pssm->k)) : nullptr;
    auto ensure_records = [&](size_t need) {  // (re)allocate the pinned record block of the handle
        if (!host_fold || need <= host_fold->h_records_cap)
            return;
        if (host_fold->h_records) {
            (void)hipStreamSynchronize(ctx->stream);  // an earlier launch may still be writing its records there
            (void)hipHostFree(host_fold->h_records);
        }
        host_fold->h_records = nullptr;
        host_fold->h_records_cap = 0;
        const size_t cap = std::max<size_t>(2 * need, 1024);
        void *blk = nullptr;
        if (hipHostMalloc(&blk, cap * 16, hipHostMallocDefault) == hipSuccess) {
            memset(blk, 0, cap * 16);
            host_fold->h_records = blk;
            host_fold->h_records_cap = cap;
        } else {
            (void)hipGetLastError();
        }
    };
    if (!fn || !table) {
        // off the C = 32 kernels (C = 1: the Generic bench geometry of dna.rs:113-116, C = 16 shapes, odd strides): the
        // tiled store kernel leaves the same per-wavefront records when the handle can take them
        {   // the record block is sized from the grid the tiled kernel will actually run (one record per workgroup)
            unsigned long long tr = 0;
            size_t lds = 0;
            const unsigned g = tiled_plan(ctx, a, &tr, &lds);
            if (g && g < 16384)
                ensure_records((size_t)g + 1);
        }
        ScoreArgs t = a;
        unsigned nrec_t = 0;
        if (host_fold && host_fold->h_records) {
            t.track_records = static_cast<uint4 *>(host_fold->h_records);
            t.track_generation = generation;
            t.track_cap = host_fold->h_records_cap;
            t.track_nrec = &nrec_t;
        }
        LM_TRY(launch_score_store(ctx, t));
        if (nrec_t) {
            host_fold->n_records = nrec_t;
            host_fold->records_on_host = true;
            host_fold->folded = false;
        }
        return LM_HIP_OK;
    }
    const size_t nrec = (size_t)p.grid.x * (kBlock / 64);
    ensure_records(nrec + 1);
    const bool on_host = host_fold && host_fold->h_records;
    if (!on_host) {
        LM_TRY(ensure_ticket(ctx));
        LM_TRY(ctx->scratch.reserve(sizeof(ArgmaxRecord) * (size_t)p.grid.x));
    }
    FusedOut fo{};
    fo.block_best = on_host ? nullptr : static_cast<ArgmaxRecord *>(ctx->scratch.ptr);
    fo.host_records = on_host ? static_cast<uint4 *>(host_fold->h_records) : nullptr;
    fo.lead_rows = pad ? lead : 0u;
    fo.ticket = ctx->d_ticket;
    fo.final_out = d_result;
    fo.final_host = h_result;
    fo.generation = generation;
    fo.first_cell_rule = first_cell_rule;
    ctx->last_kernel = score_c32_name((int)mk, MODE_STORE);
    LM_HIP_TRY(fn(p.grid, p.lds, ctx->stream, a.d_seq, table, (int)a.pssm->k, a.row_begin, a.row_end, p.T, p.nstreams,
                  a.d_out, fo));
    if (on_host) {
        host_fold->n_records = (unsigned)nrec;
        host_fold->records_on_host = true;
        host_fold->folded = false;
    } else {
        *tracked = true;
    }
    return LM_HIP_OK;
}

// Many block records (the store kernel runs ~64 K small workgroups per Gbp): a first level
// of 256 workgroups folds them to 256 records in place of one workgroup reading a megabyte.
__global__ __launch_bounds__(kBlock) void argmax_fold(const ArgmaxRecord *__restrict__ blocks,
                                                      const unsigned nblocks,
                                                      ArgmaxRecord *__restrict__ out)
{
    __shared__ float sm_v[kBlock / 64];
    __shared__ long long sm_i[kBlock / 64];
    float v = -INFINITY;
    long long i = -1;
    for (unsigned b = blockIdx.x * kBlock + threadIdx.x; b < nblocks; b += gridDim.x * kBlock)
        if (blocks[b].found)
            best_merge(v, i, blocks[b].value, blocks[b].index);
    best_block_reduce(v, i, sm_v, sm_i);
    if (threadIdx.x == 0) {
        out[blockIdx.x].value = v;
        out[blockIdx.x].index = i;
        out[blockIdx.x].found = i >= 0;
    }
}

// `d_blocks` must have room for 256 more records behind the first `nblocks` when nblocks > 4096.
int finalize_argmax_materialised(lm_hip_ctx *ctx, const ArgmaxRecord *d_blocks, unsigned nblocks,
                                 const float *d_scores, int first_cell_rule, ArgmaxRecord *d_out)
{
    if (nblocks > 4096) {
        ArgmaxRecord *folded = const_cast<ArgmaxRecord *>(d_blocks) + nblocks;
        hipLaunchKernelGGL(argmax_fold, dim3(256), dim3(kBlock), 0, ctx->stream, d_blocks, nblocks, folded);
        d_blocks = folded;
        nblocks = 256;
    }
    hipLaunchKernelGGL(argmax_finalize, dim3(1), dim3(kBlock), 0, ctx->stream, d_blocks, nblocks,
                       d_scores, (const uint8_t *)nullptr, 0ull, (const float *)nullptr, 0, 0,
                       first_cell_rule, d_out);
    LM_HIP_TRY(hipGetLastError());
    return LM_HIP_OK;
}

}  // namespace lm
