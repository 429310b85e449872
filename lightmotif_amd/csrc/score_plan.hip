// score_plan.hip -- registry of the unrolled kernels and the stream geometry every launch shares (score_launch.hpp).
#include <cstdio>
#include <cstring>
#include <mutex>

#include "score_launch.hpp"

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
static PrefilterLauncher g_preblk[kMaxFastM + 1];
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
    const KernelRegistry r{g_c32, g_pre, g_pre2, g_pre2_protein, g_u8, g_u8_pairs, g_pre2_multi, g_c32w, g_prew, g_u8w, g_preblk};
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

ScoreC32Launcher score_c32_lookup(int M, int mode, bool wide)
{
    std::call_once(g_c32_once, init_registry);
    if (M < 1 || M > kMaxStoreM || mode < 0 || mode > 2)
        return nullptr;
    return (wide ? g_c32w : g_c32)[M][mode];
}

PrefilterLauncher score_c32_prefilter_lookup(int M, bool wide, bool blocks)
{
    std::call_once(g_c32_once, init_registry);
    return (M >= 1 && M <= kMaxFastM) ? (wide ? (blocks ? g_preblk[M] : g_prew[M]) : g_pre[M]) : nullptr;
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

ExactMotif exact_motif(const lm_hip_pssm *p, const uint8_t *d_seq)
{
    if (p->m <= (size_t)kMaxFastM)
        return ExactMotif{p->m, p->d_table, 0u};
    if (p->parts.size() == 1 && p->parts[0].m <= (size_t)kMaxLongM && reinterpret_cast<uintptr_t>(d_seq) % 4 == 0)
        return ExactMotif{p->parts[0].m, p->parts[0].d_table, (unsigned)p->parts[0].lead};
    return ExactMotif{};
}

C32Plan plan_c32(const lm_hip_ctx *ctx, const ScoreArgs &a, bool store, int prefilter, size_t batch)
{
    const size_t m = prefilter == 0 ? exact_motif(a.pssm, a.d_seq).m : a.pssm->m;
    const MotifShape ms{m, a.pssm->k, a.pssm->d_image2 != nullptr};
    return plan_c32(ctx, ms, a, store, prefilter, batch);
}

// `allow16`: the caller also has a kernel for C = 16 (plain store, score rows of 16 floats; a
// wavefront then carries four streams)
C32Plan plan_c32(const lm_hip_ctx *ctx, const MotifShape &ms, const ScoreArgs &a, bool store,
                        int prefilter, size_t batch, unsigned long long default_rows, bool allow16)
{
    C32Plan p;
    const size_t K = ms.k;
    // rows per unrolled group: the motif length, padded for the prefilter kernels
    const size_t M = prefilter == 2   ? (size_t)prefilter2_ring((int)ms.m)
                     : prefilter == 1 ? (size_t)prefilter_mp((int)ms.m, lds_wide((int)K))
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

dim3 generic_grid(const lm_hip_ctx *ctx, unsigned long long ncells)
{
    unsigned long long blocks = (ncells + kBlock - 1) / kBlock;
    const unsigned long long cap = (unsigned long long)ctx->num_cus * 32;
    return dim3((unsigned)std::max<unsigned long long>(std::min(blocks, cap), 1));
}

size_t generic_lds(const lm_hip_pssm *p, int *use_lds)
{
    const size_t bytes = p->m * p->k * sizeof(float);
    *use_lds = bytes <= 48 * 1024 && bytes > 0;
    return std::max<size_t>(*use_lds ? bytes : 0, 64);
}

// ---- batches ---------------------------------------------------------------------------

int batch_fork(lm_hip_ctx *ctx)
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

int batch_join(lm_hip_ctx *ctx)
{
    LM_HIP_TRY(hipEventRecord(ctx->join_event, ctx->aux_stream));
    LM_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->join_event, 0));
    return LM_HIP_OK;
}

float best_kmer_score(const lm_hip_pssm *p)
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

bool chunked_ok(const lm_hip_ctx *ctx, const ScoreArgs &a)
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
unsigned long long chunk_rows_for(const lm_hip_ctx *ctx, const ScoreArgs &a)
{
    return std::max<unsigned long long>((unsigned long long)ctx->chunk_rows * 32 / a.cols, 1);
}

unsigned long long chunk_count(const lm_hip_ctx *ctx, const ScoreArgs &a)
{
    const unsigned long long n = a.row_end - a.row_begin, per = chunk_rows_for(ctx, a);
    return (n + per - 1) / per;
}

// Workgroups of the per-chunk argmax (a full chunk is 2^25 cells)
unsigned chunk_argmax_grid(const lm_hip_ctx *ctx)
{
    return (unsigned)ctx->num_cus * 8;
}

}  // namespace lm
