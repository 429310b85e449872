// score.hip -- launch logic of the scoring kernels (kernel bodies: score_kernels.hpp).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "score_prefilter2.hpp"
#include "score_u8.hpp"

namespace lm {

// ---- registry of the unrolled C=32 kernels ---------------------------------------

void register_score_c32_0(const KernelRegistry &r);
void register_score_c32_1(const KernelRegistry &r);
void register_score_c32_2(const KernelRegistry &r);
void register_score_c32_3(const KernelRegistry &r);
void register_score_c32_4(const KernelRegistry &r);
void register_score_c32_5(const KernelRegistry &r);
void register_score_c32_6(const KernelRegistry &r);
void register_score_c32_7(const KernelRegistry &r);
void register_score_c32_8(const KernelRegistry &r);
void register_score_c32_long_40(const KernelRegistry &r);
void register_score_c32_long_44(const KernelRegistry &r);
void register_score_c32_long_48(const KernelRegistry &r);
void register_score_c32_long_52(const KernelRegistry &r);
void register_score_c32_long_56(const KernelRegistry &r);
void register_score_c32_long_60(const KernelRegistry &r);
void register_score_c32_long_64(const KernelRegistry &r);
void register_score_pair_65(const KernelRegistry &r);
void register_score_pair_81(const KernelRegistry &r);
void register_score_c32_xlong_72(const KernelRegistry &r);
void register_score_c32_xlong_80(const KernelRegistry &r);
void register_score_c32_xlong_88(const KernelRegistry &r);

void register_score_pair_97(const KernelRegistry &r);
void register_score_pair_113(const KernelRegistry &r);

static ScoreC32Launcher g_c32[kMaxStoreM + 1][kRegistrySlots];  // rows kMaxFastM + 1 ..: the long family (M % 4 == 0); beyond kMaxLongM: store only (M % 8 == 0)
static ScoreC32Launcher g_c32w[kMaxStoreM + 1][kRegistrySlots];  // wide alphabets (lds_wide(K))
static PrefilterLauncher g_prew[kMaxFastM + 1];
static ScoreU8Launcher g_u8w[kMaxFastM + 1];
static PrefilterLauncher g_pre[kMaxFastM + 1];
static PrefilterLauncher g_pre2[kMaxPairM + 1];  // DNA pair scan: every length up to kMaxPairM
static PrefilterLauncher g_pre2_protein[kMaxFastM + 1];
static ScoreU8Launcher g_u8[kMaxFastM + 1];
static ScoreU8Launcher g_u8_pairs[kMaxFastM + 1];
static PrefilterMultiLauncher g_pre2_multi[kMaxFastM + 1];
static char g_c32_names[kMaxStoreM + 1][3][32];
static std::once_flag g_c32_once;

static void init_registry()
{
    const KernelRegistry r{g_c32, g_pre, g_pre2, g_pre2_protein, g_u8, g_u8_pairs, g_pre2_multi, g_c32w, g_prew, g_u8w};
    register_score_c32_0(r);
    register_score_c32_1(r);
    register_score_c32_2(r);
    register_score_c32_3(r);
    register_score_c32_4(r);
    register_score_c32_5(r);
    register_score_c32_6(r);
    register_score_c32_7(r);
    register_score_c32_8(r);
    register_score_c32_long_40(r);
    register_score_c32_long_44(r);
    register_score_c32_long_48(r);
    register_score_c32_long_52(r);
    register_score_c32_long_56(r);
    register_score_c32_long_60(r);
    register_score_c32_long_64(r);
    register_score_c32_xlong_72(r);
    register_score_c32_xlong_80(r);
    register_score_c32_xlong_88(r);
    register_score_pair_65(r);
    register_score_pair_81(r);
    register_score_pair_97(r);
    register_score_pair_113(r);
    for (int m = 0; m <= kMaxStoreM; ++m)
        for (int mode = 0; mode < 3; ++mode)
            snprintf(g_c32_names[m][mode], sizeof g_c32_names[m][mode], "score_c32<%d,%d>", m, mode);
}

ScoreC32Launcher score_c32_lookup(int M, int mode, bool xcd_remap, bool wide)
{
    std::call_once(g_c32_once, init_registry);
    if (M < 1 || M > kMaxStoreM || mode < 0 || mode > 2)
        return nullptr;
    if (wide)
        return g_c32w[M][mode];  // (no XCD-remap variant: an A/B knob of the DNA store kernel)
    if (mode == MODE_STORE && xcd_remap)
        return g_c32[M][3];
    return g_c32[M][mode];
}

PrefilterLauncher score_c32_prefilter_lookup(int M, bool wide)
{
    std::call_once(g_c32_once, init_registry);
    return (M >= 1 && M <= kMaxFastM) ? (wide ? g_prew[M] : g_pre[M]) : nullptr;
}

PrefilterLauncher score_c32_prefilter2_lookup(int M, int K)
{
    std::call_once(g_c32_once, init_registry);
    if (M < 1 || M > (K == 5 ? kMaxPairM : kMaxFastM))
        return nullptr;
    return K == 5 ? g_pre2[M] : K == 21 ? g_pre2_protein[M] : nullptr;
}

PrefilterMultiLauncher score_c32_prefilter2_multi_lookup(int M)
{
    std::call_once(g_c32_once, init_registry);
    return (M >= 1 && M <= kMaxFastM) ? g_pre2_multi[M] : nullptr;
}

static ScoreC32Launcher c32_slot(int M, int slot, bool wide)
{
    std::call_once(g_c32_once, init_registry);
    return (M >= 1 && M <= kMaxStoreM) ? (wide ? g_c32w : g_c32)[M][slot] : nullptr;
}

ScoreC32Launcher score_c32_lookup_store_argmax(int M, bool wide) { return c32_slot(M, 8, wide); }
ScoreC32Launcher score_c32_lookup_continue(int M, bool wide) { return c32_slot(M, 9, wide); }
ScoreC32Launcher score_c32_lookup_c16(int M, bool wide) { return c32_slot(M <= kMaxFastM ? M : 0, 10, wide); }
ScoreC32Launcher score_c32_lookup_store_track(int M, bool wide) { return c32_slot(M, 11, wide); }
ScoreC32Launcher score_c32_lookup_ql(int M, bool wide) { return c32_slot(M, 7, wide); }

ScoreU8Launcher score_c32_lookup_u8(int M, bool pairs, bool wide)
{
    std::call_once(g_c32_once, init_registry);
    if (M < 1 || M > kMaxFastM)
        return nullptr;
    return pairs ? g_u8_pairs[M] : wide ? g_u8w[M] : g_u8[M];
}

const char *score_c32_name(int M, int mode)
{
    std::call_once(g_c32_once, init_registry);
    return (M >= 0 && M <= kMaxStoreM && mode >= 0 && mode < 3) ? g_c32_names[M][mode] : "score_c32";
}

// ---- stream geometry ---------------------------------------------------------------

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
// The XCD-aware block remap (lm_hip_ctx_set_xcd_remap) is OFF by default: it wins
// 3 % when the buffers are fresh, separately hipMalloc'ed regions (kbench5_ab.txt:
// 0.908 vs 0.941 ms) but loses 3 % inside one large arena or under PyTorch's
// allocator (kbench6_place.txt, `bench.py --ab`: 1.00 vs 0.97 ms on the same box) --
// eight distant windows instead of one compact one; the compact window is the
// robust choice.
// `prefilter`: 0 = exact kernels, 1 = one-symbol prefilter (streams of q*MP + 1 rows),
// 2 = pair-symbol prefilter (streams of q*RING + 2 rows)
// what the planner needs to know about the matrix (a.pssm may be absent: u8 scores)
struct MotifShape {
    size_t m, k;
    bool pair_table;
};

static C32Plan plan_c32(const lm_hip_ctx *ctx, const MotifShape &ms, const ScoreArgs &a, bool store,
                        int prefilter = 0, size_t batch = 1, unsigned long long default_rows = 0,
                        bool allow16 = false);

// How the exact C = 32 kernels see a motif: up to kMaxFastM rows as they are (byte symbol loads, any length);
// kMaxFastM < M <= kMaxLongM as ONE slice padded with leading zero rows to a multiple of 4 (the long family:
// dword symbol loads, so the matrix must be 4-byte aligned); longer motifs have no single-pass kernel.
struct ExactMotif {
    size_t m = 0;            // rows the kernel is instantiated for
    const float *table = nullptr;
    unsigned lead = 0;       // leading all-zero rows among them
};
static ExactMotif exact_motif(const lm_hip_pssm *p, const uint8_t *d_seq)
{
    if (p->m <= (size_t)kMaxFastM)
        return ExactMotif{p->m, p->d_table, 0u};
    if (p->parts.size() == 1 && p->parts[0].m <= (size_t)kMaxLongM && reinterpret_cast<uintptr_t>(d_seq) % 4 == 0)
        return ExactMotif{p->parts[0].m, p->parts[0].d_table, (unsigned)p->parts[0].lead};
    return ExactMotif{};
}

static C32Plan plan_c32(const lm_hip_ctx *ctx, const ScoreArgs &a, bool store, int prefilter = 0,
                        size_t batch = 1)
{
    const size_t m = prefilter == 0 ? exact_motif(a.pssm, a.d_seq).m : a.pssm->m;
    const MotifShape ms{m, a.pssm->k, a.pssm->d_image2 != nullptr};
    return plan_c32(ctx, ms, a, store, prefilter, batch);
}

// `allow16`: the caller also has a kernel for C = 16 (plain store, score rows of 16 floats; a
// wavefront then carries four streams)
static C32Plan plan_c32(const lm_hip_ctx *ctx, const MotifShape &ms, const ScoreArgs &a, bool store,
                        int prefilter, size_t batch, unsigned long long default_rows, bool allow16)
{
    C32Plan p;
    const size_t K = ms.k;
    // rows per unrolled group: the motif length, padded for the prefilter kernels
    const size_t M = prefilter == 2   ? (size_t)prefilter2_ring((int)ms.m)
                     : prefilter == 1 ? (size_t)prefilter_mp((int)ms.m)
                                      : ms.m;
    const size_t extra = prefilter == 2 ? 2 : 1;  // rows of a stream beyond q groups
    const unsigned long long n = a.row_end - a.row_begin;
    const bool c16 = allow16 && store && prefilter == 0 && a.cols == 16;
    if ((a.cols != 32 && !c16) || a.seq_stride != 32 || (store && a.out_stride != a.cols))
        return p;
    if (ms.m < 1 || ms.m > (size_t)(prefilter == 0 ? (store ? kMaxStoreM : kMaxLongM) : (prefilter == 2 && K == 5) ? kMaxPairM : kMaxFastM) || n < M + extra)
        return p;
    if (prefilter == 0 && ms.m > (size_t)kMaxFastM && (ms.m % 4 != 0 || reinterpret_cast<uintptr_t>(a.d_seq) % 4 != 0))
        return p;  // the long family: padded lengths, dword symbol loads
    // (the pair-symbol kernel fetches symbols with dword loads: 4-byte aligned matrix)
    if (prefilter == 2 && ((K != 5 && !(K == 21 && ctx->pair_prefilter_protein)) || !ms.pair_table || ms.m < 2 ||
                           reinterpret_cast<uintptr_t>(a.d_seq) % 4 != 0))
        return p;
    const size_t lds = prefilter == 2   ? (size_t)prefilter2_image_dw((int)ms.m, (int)K) * 4
                       : prefilter == 1 ? (size_t)prefilter_image_dw((int)ms.m, (int)K) * 4
                                        : std::max<size_t>(K * table_stride((int)M, lds_wide((int)K)) * sizeof(float), 64);
    if (lds > 60 * 1024)
        return p;
    // The fused kernels write nothing, so they are LDS/VALU-bound and prefer long
    // streams (fewer fill steps): T=1001 0.69 ms vs T=61 0.79 ms on the same input.
    // Store kernel: three groups per stream (T = 3M + 1) up to M = 26, where the write pattern
    // binds and short streams keep the window of rows in flight compact (M = 12: T = 37 0.872 ms,
    // T = 73 0.894; M = 20: T = 61 0.940, T = 121 0.970; M = 24: T = 73 1.001, T = 145 1.024);
    // longer motifs are bound by the LDS gather and want the fill / drain groups amortised over
    // six groups (M = 28: T = 57 1.29 ms, T = 169 1.08; M = 33: T = 67 1.30, T = 199 1.22;
    // profiles/r02_edge_ab.txt).  Very short motifs keep ~24 rows per stream.
    unsigned long long target = ctx->rows_per_stream ? ctx->rows_per_stream
                                : default_rows         ? default_rows
                                : !store               ? 1024
                                : prefilter != 0       ? 64
                                : ms.m > 26            ? 6 * M
                                                       : std::max<size_t>(3 * M, 24);
    // keep at least ~4 streams per SIMD lane-half in flight on small inputs
    // Enough workgroups for several rounds of the chip's resident capacity (6 x 256 CUs
    // of 8-stream workgroups), so the last partial round costs little; the fused
    // kernels run back to back per motif, where that tail is paid every launch.
    // (a multi-job launch brings `batch` times as many workgroups, so each job needs fewer)
    const unsigned long long want_streams =
        std::max<unsigned long long>((unsigned long long)ctx->num_cus * (store ? 64 : 128) / batch, 64);
    if (n / target < want_streams)
        target = std::max<unsigned long long>(n / want_streams, 1);
    target = std::min<unsigned long long>(target, 1ull << 30);  // step indices are 32-bit
    unsigned long long q = std::max<unsigned long long>((target + M / 2) / M, 1);
    if (q * M + extra > n)
        q = (n - extra) / M;
    if (q < 1)
        return p;
    p.T = q * M + extra;
    p.nstreams = (n + p.T - 1) / p.T;
    const unsigned long long per_block = c16 ? 2 * kStreamsPerBlock : kStreamsPerBlock;
    p.grid = dim3((unsigned)((p.nstreams + per_block - 1) / per_block));
    p.lds = lds;
    p.ok = true;
    return p;
}

static dim3 generic_grid(const lm_hip_ctx *ctx, unsigned long long ncells)
{
    unsigned long long blocks = (ncells + kBlock - 1) / kBlock;
    const unsigned long long cap = (unsigned long long)ctx->num_cus * 32;
    return dim3((unsigned)std::max<unsigned long long>(std::min(blocks, cap), 1));
}

static size_t generic_lds(const lm_hip_pssm *p, int *use_lds)
{
    const size_t bytes = p->m * p->k * sizeof(float);
    *use_lds = bytes <= 48 * 1024 && bytes > 0;
    return std::max<size_t>(*use_lds ? bytes : 0, 64);
}

template <int MODE>
static int launch_generic(lm_hip_ctx *ctx, const ScoreArgs &a, const FusedOut &fo, dim3 grid,
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

int launch_score_store(lm_hip_ctx *ctx, const ScoreArgs &a)
{
    FusedOut fo{};
    const bool wide = lds_wide((int)a.pssm->k);
    const bool dwords = ctx->quad_loads && !ctx->xcd_remap && reinterpret_cast<uintptr_t>(a.d_seq) % 4 == 0;
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
        ScoreC32Launcher fn = score_c32_lookup((int)a.pssm->m, MODE_STORE, ctx->xcd_remap, wide);
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
                    fn = score_c32_lookup((int)part.m, MODE_STORE, false, wide);
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
            const int mp = prefilter_mp(m), shift = pairs ? 0 : mp - m;
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

static int ensure_ticket(lm_hip_ctx *ctx)
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
    const C32Plan p = (mk >= 1 && mk % 4 == 0 && a.cols == 32 && a.out_stride == 32 && ctx->quad_loads && !ctx->xcd_remap &&
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

// Enqueues the fused argmax of one job: block records -> `blocks`, result -> `d_result`.
// Independent jobs of a batch alternate between the context's stream and an auxiliary
// one, so the tail of one motif's kernel (the last, partially filled round of
// workgroups) overlaps the head of the next.  fork: aux waits for everything already
// enqueued on the main stream; join: the main stream waits for aux.
static int batch_fork(lm_hip_ctx *ctx)
{
    if (!ctx->aux_stream) {
        LM_HIP_TRY(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
        LM_HIP_TRY(hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
        LM_HIP_TRY(hipEventCreateWithFlags(&ctx->join_event, hipEventDisableTiming));
    }
    LM_HIP_TRY(hipEventRecord(ctx->fork_event, ctx->stream));
    LM_HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->fork_event, 0));
    return LM_HIP_OK;
}

static int batch_join(lm_hip_ctx *ctx)
{
    LM_HIP_TRY(hipEventRecord(ctx->join_event, ctx->aux_stream));
    LM_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->join_event, 0));
    return LM_HIP_OK;
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

// Jobs of a batch that can share ONE launch (grid.y = jobs): same kernel, motif length,
// alphabet and sequence rows.  Many short per-motif launches lose ~15 % to their ramps
// and to the short streams a small grid needs; a launch per motif LENGTH keeps streams
// long and the chip full (2 346 JASPAR motifs -> ~50 launches).
enum : int { KIND_GENERIC = 0, KIND_EXACT = 1, KIND_PREFILTER = 2, KIND_PREFILTER2 = 3, KIND_CHUNKED = 4, KIND_SKIP = 5 };
static inline bool kind_solo(int kind) { return kind == KIND_GENERIC || kind == KIND_CHUNKED; }
// B = the score of a best k-mer: the row maxima added in motif order.  It bounds every score of the matrix from
// above -- f32 rounding is monotone, so termwise larger weights added in the same order cannot give a smaller
// sum -- provided the weights hold no NaN / +inf (lm_hip_pssm::has_prefilter).
static float best_kmer_score(const lm_hip_pssm *p)
{
    float b = 0.0f;
    for (size_t j = 0; j < p->m; ++j) {
        float best = p->host[j * p->k];
        for (size_t s = 1; s < p->k; ++s)
            best = p->host[j * p->k + s] > best ? p->host[j * p->k + s] : best;
        b = b + best;
    }
    return b;
}

struct JobGroup {
    int kind = KIND_GENERIC;
    std::vector<size_t> idx;  // job indices, ascending
    C32Plan plan;
};

template <typename KindOf>
static std::vector<JobGroup> group_jobs(const lm_hip_ctx *ctx, const ScoreArgs *jobs, size_t n,
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
static bool chunked_ok(const lm_hip_ctx *ctx, const ScoreArgs &a)
{
    if (!ctx->chunked_fused)
        return false;
    const unsigned long long n = a.row_end - a.row_begin;
    if (a.cols == 32)
        return !a.pssm->parts.empty() && a.seq_stride == 32 && reinterpret_cast<uintptr_t>(a.d_seq) % 4 == 0 &&
               n > (unsigned long long)a.pssm->parts[0].m;
    return a.cols >= 1 && a.cols <= 4096 && n * a.cols >= (1ull << 16) && n > a.pssm->m;
}

// rows per chunk: ctx->chunk_rows is quoted for C = 32; other column counts keep the chunk's cell count
static unsigned long long chunk_rows_for(const lm_hip_ctx *ctx, const ScoreArgs &a)
{
    return std::max<unsigned long long>((unsigned long long)ctx->chunk_rows * 32 / a.cols, 1);
}

static unsigned long long chunk_count(const lm_hip_ctx *ctx, const ScoreArgs &a)
{
    const unsigned long long n = a.row_end - a.row_begin, per = chunk_rows_for(ctx, a);
    return (n + per - 1) / per;
}

// Workgroups of the per-chunk argmax (a full chunk is 2^25 cells)
static unsigned chunk_argmax_grid(const lm_hip_ctx *ctx)
{
    return (unsigned)ctx->num_cus * 8;
}

// f(buffer, first row of the chunk relative to a.row_begin, rows of the chunk) after each chunk's scores
// are enqueued on ctx->stream
template <typename PerChunk>
static int for_each_scored_chunk(lm_hip_ctx *ctx, const ScoreArgs &a, PerChunk f)
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

// Fused score+argmax of `n` independent jobs (one motif each): the n scoring kernels
// are enqueued back to back, each leaving per-workgroup records in its own region,
// then ONE finalize launch reduces every job and ONE synchronisation returns.
static int launch_score_argmax_exact(lm_hip_ctx *ctx, const ScoreArgs *jobs, size_t n,
                                     int first_cell_rule, ArgmaxRecord *out)
{
    if (n == 0)
        return LM_HIP_OK;
    const std::vector<JobGroup> groups = group_jobs(ctx, jobs, n, [&](size_t i) {
        return plan_c32(ctx, jobs[i], false).ok ? KIND_EXACT : chunked_ok(ctx, jobs[i]) ? KIND_CHUNKED : KIND_GENERIC;
    });
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
            ScoreC32Launcher fn = score_c32_lookup((int)em.m, MODE_ARGMAX, false, lds_wide((int)a.pssm->k));
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


// ---- fused threshold --------------------------------------------------------------------

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

template <bool LDS_TAB>
__global__ __launch_bounds__(kRescoreBlock) void rescore_candidates(const RescoreJob *__restrict__ jobs,
                                                             const FusedOut fo, const unsigned njobs)
{
    __shared__ HitRecord stage[kHitStage];
    __shared__ unsigned nstage;
    __shared__ unsigned long long gbase;
    __shared__ float tab[LDS_TAB ? kRescoreTabFloats : 1];
    __shared__ unsigned tab_off[LDS_TAB ? kRescoreTabJobs : 1];
    if (LDS_TAB) {
        unsigned off = 0;
        for (unsigned j = 0; j < njobs; ++j) {  // block-uniform
            const unsigned nf = jobs[j].m * jobs[j].k;
            for (unsigned i = threadIdx.x; i < nf; i += kRescoreBlock)
                tab[off + i] = jobs[j].dense[i];
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
    const unsigned long long stride = (unsigned long long)gridDim.x * (kRescoreBlock / 32);
    auto flush = [&]() {  // block-uniform
        __syncthreads();
        const unsigned cnt = nstage;
        if (threadIdx.x == 0 && cnt)
            gbase = atomicAdd(fo.hit_count, (unsigned long long)cnt);
        __syncthreads();
        for (unsigned i = threadIdx.x; i < cnt; i += kRescoreBlock)
            if (gbase + i < fo.hit_capacity)
                fo.hits[gbase + i] = stage[i];
        __syncthreads();
        if (threadIdx.x == 0)
            nstage = 0;
        __syncthreads();
    };
    // block-uniform trip count: one candidate piece per half-wave per round
    constexpr unsigned kWin = (32 + kMaxPairM - 1 + 31) / 32 * 32;  // a piece of <= 32 rows of a motif of <= kMaxPairM rows
    __shared__ uint8_t window[kRescoreBlock / 32][kWin];  // symbols of rows r0 .. r0 + nrows + M - 2
    unsigned round = 0;
    for (unsigned long long c0 = (unsigned long long)blockIdx.x * (kRescoreBlock / 32); c0 < n; c0 += stride) {
        const unsigned long long c = c0 + (threadIdx.x >> 5);
        if (c < n) {
            const Candidate cd = fo.cands[c];
            const RescoreJob jb = jobs[cd.key >> 40];
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
                    stage[atomicAdd(&nstage, 1u)] = r;  // <= kRescoreBlock records per round
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
            if (cnt > kHitStage - kRescoreCheck * kRescoreBlock)
                flush();
        }
    }
    flush();
}

static int launch_rescore(lm_hip_ctx *ctx, hipStream_t st, const RescoreJob *d_jobs, const FusedOut &fo,
                          const RescoreJob *host_jobs, size_t n)
{
    size_t floats = 0;
    for (size_t i = 0; i < n && i <= (size_t)kRescoreTabJobs; ++i)
        floats += (size_t)host_jobs[i].m * host_jobs[i].k;
    const dim3 grid((unsigned)ctx->num_cus * kRescoreBlocksPerCu), block(kRescoreBlock);
    if (n <= (size_t)kRescoreTabJobs && floats <= (size_t)kRescoreTabFloats)
        hipLaunchKernelGGL(rescore_candidates<true>, grid, block, 0, st, d_jobs, fo, (unsigned)n);
    else
        hipLaunchKernelGGL(rescore_candidates<false>, grid, block, 0, st, d_jobs, fo, (unsigned)n);
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
        for (size_t i : g.idx) {
            const ScoreArgs &a = jobs[i];
            if (keys == HitKeys::Position && a.row_end - a.row_begin != key_rows)
                return fail(LM_HIP_ERR_BAD_ARGS, "fused threshold: position keys need equal row ranges");
            bparams.push_back(BatchParams{g.kind == KIND_PREFILTER2  ? (const void *)a.pssm->d_image2
                                          : g.kind == KIND_PREFILTER ? (const void *)a.pssm->d_image
                                          : g.kind == KIND_EXACT     ? (const void *)exact_motif(a.pssm, a.d_seq).table
                                                                     : (const void *)a.pssm->d_table,
                                          nullptr, ts[i], tds[i], (unsigned long long)i << 40});
            rjobs[i] = RescoreJob{a.d_seq + a.row_begin * a.seq_stride, a.pssm->d_dense,
                                  (unsigned)a.pssm->m, (unsigned)a.pssm->k, ts[i], 0, key_rows};
        }
        const int m = (int)jobs[g.idx[0]].pssm->m;
        if (g.kind == KIND_PREFILTER2 && ctx->multi_motif && n > 1 && g.idx.size() >= 2 &&
            jobs[g.idx[0]].pssm->k == 5 && score_c32_prefilter2_multi_lookup(m)) {
            per_pass[gi] = prefilter2_multi(m);
            while ((bparams.size() - group_pos[gi]) % per_pass[gi]) {
                BatchParams pad = bparams.back();
                pad.td = 0xffffffffu;  // no sum reaches it: the padding job flags nothing
                bparams.push_back(pad);
            }
        }
    }
    const size_t nbp = bparams.size();
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
        if (off_hits <= kPinnedBytes / 2) {
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
        if (two_streams)
            LM_TRY(batch_fork(ctx));
        bool any_candidates = false;
        size_t launch = 0;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const JobGroup &g = groups[gi];
            const size_t bp_pos = group_pos[gi];
            const size_t i = g.idx[0];
            const ScoreArgs &a = jobs[i];
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
                PrefilterLauncher fn = pairs ? score_c32_prefilter2_lookup((int)a.pssm->m, (int)a.pssm->k)
                                             : score_c32_prefilter_lookup((int)a.pssm->m, lds_wide((int)a.pssm->k));
                ctx->last_kernel = pairs ? "score_c32_prefilter2" : "score_c32_prefilter";
                LM_HIP_TRY(fn(g.plan.grid, g.plan.lds, st, a.d_seq, pairs ? a.pssm->d_image2 : a.pssm->d_image,
                              (int)a.pssm->k, a.row_begin, a.row_end, g.plan.T, g.plan.nstreams, tds[i], fo));
                any_candidates = true;
            } else if (g.kind == KIND_EXACT) {
                const ExactMotif em = exact_motif(a.pssm, a.d_seq);
                FusedOut efo = fo;
                efo.lead_rows = em.lead;
                ScoreC32Launcher fn = score_c32_lookup((int)em.m, MODE_THRESHOLD, false, lds_wide((int)a.pssm->k));
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
        if (any_candidates) {
            LM_TRY(launch_rescore(ctx, ctx->stream, d_jobs, fo, rjobs.data(), n));
            LM_HIP_TRY(hipGetLastError());
        }
        const int emit = keys == HitKeys::Position ? 1 : 0;
        unsigned long long count = 0, ncand = 0;
        const auto t_scan = std::chrono::steady_clock::now();
        bool ordered = false;
        if (attempt == 0 && ctx->speculate_order) {
            // First try: enqueue the ordering right behind the scans, sized from the previous
            // call's count, and learn the counts from the same single synchronisation.
            int status = 0;
            unsigned long long counts[2] = {0, 0};
            LM_TRY(order_hits(ctx, fo.hits, fo.hit_count, ~0ull, cap, ccap, ctx->last_hit_count + ctx->last_hit_count / 4,
                              n, max_low, emit, jobs[0].cols, out, &status, counts));
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
            LM_TRY(order_hits(ctx, fo.hits, fo.hit_count, count, cap, ccap, count, n, max_low, emit, jobs[0].cols,
                              out, &status, counts));
        }
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
    // one record per workgroup, folded by argmax_prepare: 2 048 atomics on one address would
    // serialise at ~6 ns each (12 us, more than the sampling itself)
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
        const unsigned long long nchunks = std::max<unsigned long long>(rows / 1024 / kSampleRows, 32);
        max_chunks = std::max(max_chunks, nchunks);
        sjobs[q] = SampleJob{a.d_seq + a.row_begin * a.seq_stride, a.pssm->d_dense, (unsigned)a.pssm->m,
                             (unsigned)a.pssm->k, nchunks, (rows - kSampleRows) / (nchunks - 1),
                             a.pssm->pre_offset, a.pssm->pre_factor, a.pssm->pre_emax};
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
    std::vector<BatchParams> bparams;  // launch order; rjobs / sjobs are permuted the same way
    // positions in launch order; groups of the pair scan with several jobs run `per_pass[g]`
    // motifs per pass and are padded to a multiple of that (a padding position samples nothing,
    // so argmax_prepare leaves its td at "skip" and it flags nothing)
    std::vector<size_t> order, group_pos(groups.size());
    std::vector<char> is_pad;
    std::vector<int> per_pass(groups.size(), 1);
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const JobGroup &g = groups[gi];
        group_pos[gi] = order.size();
        for (size_t q : g.idx) {
            order.push_back(q);
            is_pad.push_back(0);
        }
        const int m = (int)qjobs[g.idx[0]].pssm->m;
        if (g.kind == KIND_PREFILTER2 && ctx->multi_motif && g.idx.size() >= 2 &&
            qjobs[g.idx[0]].pssm->k == 5 && score_c32_prefilter2_multi_lookup(m)) {
            per_pass[gi] = prefilter2_multi(m);
            while ((order.size() - group_pos[gi]) % per_pass[gi]) {
                order.push_back(g.idx.back());
                is_pad.push_back(1);
            }
        }
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
        bparams.push_back(BatchParams{pairs_of[q] ? qjobs[q].pssm->d_image2 : qjobs[q].pssm->d_image, nullptr,
                                      0.0f, 0xffffffffu, (unsigned long long)pos << 40});
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
        PrefilterLauncher fn = pairs ? score_c32_prefilter2_lookup((int)a.pssm->m, (int)a.pssm->k)
                                     : score_c32_prefilter_lookup((int)a.pssm->m, lds_wide((int)a.pssm->k));
        ctx->last_kernel = pairs ? "score_c32_prefilter2" : "score_c32_prefilter";
        LM_HIP_TRY(fn(g.plan.grid, g.plan.lds, ls, a.d_seq, pairs ? a.pssm->d_image2 : a.pssm->d_image,
                      (int)a.pssm->k, a.row_begin, a.row_end, g.plan.T, g.plan.nstreams, 0xffffffffu, fo));
    }
    if (two_streams)
        LM_TRY(batch_join(ctx));
    fo.batch = nullptr;
    LM_TRY(launch_rescore(ctx, st, d_rj, fo, rj.data(), npos));
    const unsigned hgrid = (unsigned)ctx->num_cus * 2;
    hipLaunchKernelGGL(hits_best_value, dim3(hgrid), dim3(kBlock), 0, st, fo.hits, fo.hit_count, cap, d_bval);
    hipLaunchKernelGGL(hits_best_key, dim3(hgrid), dim3(kBlock), 0, st, fo.hits, fo.hit_count, cap, d_bval,
                       d_bkey);
    // results and the two list counters are written straight into the pinned buffer's lower half
    const bool pin = 16 + npos * sizeof(ArgmaxRecord) <= kPinnedBytes / 2;
    ArgmaxRecord *res = pin ? reinterpret_cast<ArgmaxRecord *>(static_cast<char *>(ctx->pinned) + 16) : d_res;
    hipLaunchKernelGGL(argmax_collect, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, st, (unsigned)npos,
                       d_bval, d_bkey, res, fo.hit_count,
                       pin ? static_cast<unsigned long long *>(ctx->pinned) : static_cast<unsigned long long *>(nullptr));
    LM_HIP_TRY(hipGetLastError());
    std::vector<ArgmaxRecord> host_res(pin ? 0 : npos);
    if (!pin) {
        LM_HIP_TRY(hipMemcpyAsync(ctx->pinned, base, 16, hipMemcpyDeviceToHost, st));
        LM_HIP_TRY(hipMemcpyAsync(host_res.data(), d_res, npos * sizeof(ArgmaxRecord), hipMemcpyDeviceToHost, st));
    }
    LM_HIP_TRY(hipStreamSynchronize(st));
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
