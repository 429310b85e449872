// test_dispatch.cpp -- the `Dispatch::Hip { cpu }` arm table through its C++ twin
// `lightmotif::HipDispatch<Cpu>`: every one of the seven `match self.backend` sites of lightmotif/src/pli/dispatch.rs:58-207
// plus the Scanner (scan.rs:166-249).  Asserted here:
//   * small inputs take the CPU tier, large ones the GPU, and both give IDENTICAL bits (scores), cells (argmax, threshold)
//     and hits (Scanner) -- the route may never show in the result;
//   * Encode / Stripe / Maximum<u8> / Threshold<u8> stay on the tier at every size;
//   * the shipped crossovers (lm_hip_host_crossover) are what the policy uses unless forced.
// The CPU tier is test infrastructure (cpu_tier.hpp over the oracle's C libraries), standing in for the reference's
// own Avx2 / Generic back-ends.
#include <cstdio>
#include <random>

#include "cpu_tier.hpp"

using namespace lightmotif;
using lightmotif_test::PortTier;

static int failures = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                          \
        }                                                                        \
    } while (0)

static ScoringMatrix<Dna> mx000001()
{
    std::vector<EncodedSequence<Dna>> sites;
    sites.push_back(EncodedSequence<Dna>::encode("GTTGACCTTATCAAC"));
    sites.push_back(EncodedSequence<Dna>::encode("GTTGATCCAGTCAAC"));
    return CountMatrix<Dna>::from_sequences(sites).to_freq(0.1f).to_scoring();
}

static std::string random_dna(size_t n, unsigned seed, double p_n = 0.0)
{
    std::mt19937 rng(seed);
    std::string s(n, 'A');
    for (size_t i = 0; i < n; ++i) {
        const unsigned r = rng();
        s[i] = (p_n > 0 && (r >> 8) % 10000 < p_n * 10000) ? 'N' : "ACGT"[r & 3];
    }
    return s;
}

template <class T>
static bool same_bits(const host::StripedScores<T> &a, const host::StripedScores<T> &b)
{
    if (a.matrix().rows() != b.matrix().rows() || a.max_index() != b.max_index())
        return false;
    for (size_t r = 0; r < a.matrix().rows(); ++r)
        if (std::memcmp(a.matrix()[r], b.matrix()[r], a.matrix().columns() * sizeof(T)) != 0)
            return false;
    return true;
}

static bool same_cells(const std::vector<MatrixCoordinates> &a, const std::vector<MatrixCoordinates> &b)
{
    if (a.size() != b.size())
        return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (a[i].row != b[i].row || a[i].col != b[i].col)
            return false;
    return true;
}

static bool same_hits(const std::vector<Hip::Hit> &a, const std::vector<Hip::Hit> &b)
{
    if (a.size() != b.size())
        return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (a[i].position != b[i].position || std::memcmp(&a[i].score, &b[i].score, 4) != 0)
            return false;
    return true;
}

// the shipped policy: sizes either side of every crossover take different routes and agree
static void test_policy_routes_by_size()
{
    HipDispatch<PortTier> pli;
    const auto pssm = mx000001();
    size_t x_score = 0, x_max = 0, x_thr = 0, x_u8 = 0, x_scan = 0, x_enc = 0, x_stripe = 0, x_mu8 = 0, x_tu8 = 0;
    CHECK(lm_hip_host_crossover(LM_HIP_OP_SCORE_F32, pssm.len(), 5, &x_score) == LM_HIP_OK);
    CHECK(lm_hip_host_crossover(LM_HIP_OP_MAXIMUM_F32, 0, 0, &x_max) == LM_HIP_OK);
    CHECK(lm_hip_host_crossover(LM_HIP_OP_THRESHOLD_F32, 0, 0, &x_thr) == LM_HIP_OK);
    CHECK(lm_hip_host_crossover(LM_HIP_OP_SCORE_U8, pssm.len(), 5, &x_u8) == LM_HIP_OK);
    CHECK(lm_hip_host_crossover(LM_HIP_OP_SCAN, pssm.len(), 5, &x_scan) == LM_HIP_OK);
    CHECK(lm_hip_host_crossover(LM_HIP_OP_ENCODE, 0, 5, &x_enc) == LM_HIP_OK && x_enc == SIZE_MAX);
    CHECK(lm_hip_host_crossover(LM_HIP_OP_STRIPE, 0, 5, &x_stripe) == LM_HIP_OK && x_stripe == SIZE_MAX);
    CHECK(lm_hip_host_crossover(LM_HIP_OP_MAXIMUM_U8, 0, 0, &x_mu8) == LM_HIP_OK && x_mu8 == SIZE_MAX);
    CHECK(lm_hip_host_crossover(LM_HIP_OP_THRESHOLD_U8, 0, 0, &x_tu8) == LM_HIP_OK && x_tu8 == SIZE_MAX);
    CHECK(lm_hip_host_crossover(99, 0, 0, &x_tu8) == LM_HIP_ERR_BAD_ARGS);
    CHECK(x_score > 1000 && x_score < (size_t)1 << 30);     // a real crossover: neither "always" nor "never"
    std::printf("crossovers (cells): score_f32 %zu (M = %zu), maximum_f32 %zu, threshold_f32 %zu, score_u8 %zu, scan %zu\n", x_score,
                pssm.len(), x_max, x_thr, x_u8, x_scan);

    // Encode + Stripe: the tier, whatever the size
    const std::string text = random_dna(3'000'000, 7);
    std::vector<uint8_t> enc;
    pli.encode_into<Dna>(text, enc);
    CHECK(pli.last_route == Route::Cpu && pli.cpu.n.encode == 1);
    const EncodedSequence<Dna> encoded(enc);
    host::StripedSequence<Dna> big = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode("A"), 32);
    pli.stripe_into(encoded, big);
    CHECK(pli.last_route == Route::Cpu && pli.cpu.n.stripe == 1 && big.len() == text.size());
    big.configure(pssm);

    // the reference's published small benchmark: score + argmax of a 10 kb sequence (README.md:111-118) -> the tier
    host::StripedSequence<Dna> small = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode(text.substr(0, 10'000)), 32);
    small.configure(pssm);
    const size_t gpu0 = pli.calls_gpu;
    const auto s_small = pli.score(pssm, small);
    CHECK(pli.last_route == Route::Cpu);
    const auto a_small = pli.argmax(s_small);
    CHECK(pli.last_route == Route::Cpu && pli.calls_gpu == gpu0);
    // ... and the same call forced to the GPU: identical bits, identical cell
    HipDispatch<PortTier> forced;
    for (int op = 0; op < 9; ++op)
        forced.policy.force((lm_hip_host_op)op, 0);
    const auto s_small_gpu = forced.score(pssm, small);
    CHECK(forced.last_route == Route::Gpu && same_bits(s_small, s_small_gpu));
    const auto a_small_gpu = forced.argmax(s_small_gpu);
    CHECK(forced.last_route == Route::Gpu && a_small && a_small_gpu && a_small->row == a_small_gpu->row && a_small->col == a_small_gpu->col);

    // 3 Mbp: Score<f32> on the GPU by the shipped policy, identical to the tier forced the other way
    const auto s_big = pli.score(pssm, big);
    CHECK(pli.last_route == Route::Gpu);
    HipDispatch<PortTier> never;
    for (int op = 0; op < 9; ++op)
        never.policy.force((lm_hip_host_op)op, SIZE_MAX);
    const auto s_big_cpu = never.score(pssm, big);
    CHECK(never.last_route == Route::Cpu && same_bits(s_big, s_big_cpu));
    // Maximum / Threshold on the 3 M-cell matrix: whichever route the policy takes, the forced GPU and forced CPU answers agree
    const auto am = pli.argmax(s_big), am_g = forced.argmax(s_big), am_c = never.argmax(s_big);
    CHECK(am && am_g && am_c && am->row == am_g->row && am->col == am_g->col && am->row == am_c->row && am->col == am_c->col);
    const auto mx_g = forced.max(s_big), mx_c = never.max(s_big);
    CHECK(mx_g && mx_c && std::memcmp(&*mx_g, &*mx_c, 4) == 0);
    const float t = *mx_c - 6.0f;
    const auto th = pli.threshold(s_big, t);
    CHECK(same_cells(th, forced.threshold(s_big, t)) && same_cells(th, never.threshold(s_big, t)) && !th.empty());
    // Score<u8>: saturating like its tier (avx2.rs:336), both routes
    const auto dm = pssm.to_discrete();
    host::StripedScores<uint8_t> d_g(32), d_c(32);
    forced.score_rows_into(dm, big, 0, big.matrix().rows() - big.wrap(), d_g);
    never.score_rows_into(dm, big, 0, big.matrix().rows() - big.wrap(), d_c);
    CHECK(forced.last_route == Route::Gpu && never.last_route == Route::Cpu && same_bits(d_g, d_c));
    // Maximum<u8> / Threshold<u8>: the tier even when everything else is forced to the GPU
    const size_t g1 = forced.calls_gpu;
    const auto du = forced.argmax(d_g);
    const auto dt = forced.threshold(d_g, forced.max(d_g).value_or(0));
    CHECK(forced.calls_gpu == g1 && du.has_value() && !dt.empty() && forced.cpu.n.maximum_u8 == 2 && forced.cpu.n.threshold_u8 == 1);
}

// Scanner: the specialised GPU form against the reference's block loop over the tier, yield order included
static void test_scanner_specialisation()
{
    const auto pssm = mx000001();
    HipDispatch<PortTier> gpu, cpu;
    gpu.policy.force(LM_HIP_OP_SCAN, 0);
    cpu.policy.force(LM_HIP_OP_SCAN, SIZE_MAX);
    for (size_t len : {(size_t)63, (size_t)10'000, (size_t)777'777}) {
        for (double pn : {0.0, 0.02}) {
            host::StripedSequence<Dna> seq = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode_lossy(random_dna(len, (unsigned)len, pn)), 32);
            seq.configure(pssm);
            for (float t : {-5.0f, 3.0f, 12.0f}) {
                for (size_t bs : {(size_t)256, (size_t)100}) {
                    const auto a = gpu.scan(pssm, seq, t, bs), b = cpu.scan(pssm, seq, t, bs);
                    CHECK(gpu.last_route == Route::Gpu && cpu.last_route == Route::Cpu);
                    CHECK(same_hits(a, b));
                }
                if (len >= 10'000) {  // (the walk's panics on tiny matrices are the reference's own: not the point here)
                    const auto ma = gpu.scan_max(pssm, seq, t), mb = cpu.scan_max(pssm, seq, t);
                    CHECK(ma.has_value() == mb.has_value());
                    if (ma && mb)
                        CHECK(ma->position == mb->position && std::memcmp(&ma->score, &mb->score, 4) == 0);
                }
            }
        }
    }
    // the shipped policy sends a bacterial genome's worth to the GPU and a 10 kb sequence to the tier
    HipDispatch<PortTier> pli;
    host::StripedSequence<Dna> small = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode(random_dna(10'000, 3)), 32);
    small.configure(pssm);
    (void)pli.scan(pssm, small, 5.0f);
    const Route r_small = pli.last_route;
    host::StripedSequence<Dna> large = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode(random_dna(4'641'652, 4)), 32);
    large.configure(pssm);
    (void)pli.scan(pssm, large, 5.0f);
    CHECK(pli.last_route == Route::Gpu);
    std::printf("scan: 10 kb -> %s, 4.6 Mbp -> gpu\n", r_small == Route::Cpu ? "cpu tier" : "gpu");
}

// columns other than 32 (the U1 / U16 geometries of tests/dna.rs): both routes agree there too
static void test_other_geometries()
{
    const auto pssm = mx000001();
    HipDispatch<PortTier> gpu, cpu;
    for (int op = 0; op < 9; ++op) {
        gpu.policy.force((lm_hip_host_op)op, 0);
        cpu.policy.force((lm_hip_host_op)op, SIZE_MAX);
    }
    for (size_t cols : {(size_t)1, (size_t)16}) {
        host::StripedSequence<Dna> seq = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode(random_dna(20'011, 11)), cols);
        seq.configure(pssm);
        const auto a = gpu.score(pssm, seq), b = cpu.score(pssm, seq);
        CHECK(same_bits(a, b));
        const auto x = gpu.argmax(a), y = cpu.argmax(b);
        CHECK(x && y && x->row == y->row && x->col == y->col);
        CHECK(same_cells(gpu.threshold(a, 0.0f), cpu.threshold(b, 0.0f)));
    }
}

// `test_dispatch --bench`: the Scanner site's two routes by sequence length (one CPU thread, as the reference gives a
// Scanner), as JSON lines -- the numbers behind kCostScan (csrc/hostptr.hip) and INTEGRATION.md 3.
#include <chrono>
template <class F>
static double median_us(F &&f, int reps)
{
    std::vector<double> t;
    for (int i = 0; i < reps + 2; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        f();
        const auto t1 = std::chrono::steady_clock::now();
        if (i >= 2)
            t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

static int bench_scan()
{
    HipDispatch<PortTier> gpu, cpu;
    gpu.policy.force(LM_HIP_OP_SCAN, 0);
    cpu.policy.force(LM_HIP_OP_SCAN, SIZE_MAX);
    const auto pssm = mx000001();
    std::printf("{\"motif\": \"MX000001 (M = 15)\", \"threshold\": 10.0, \"scan\": [");
    bool first = true;
    for (size_t len : {(size_t)10'000, (size_t)50'000, (size_t)100'000, (size_t)464'165, (size_t)1'000'000, (size_t)4'641'652, (size_t)20'000'000}) {
        host::StripedSequence<Dna> seq = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode(random_dna(len, (unsigned)len)), 32);
        seq.configure(pssm);
        const int reps = len > 2'000'000 ? 5 : 25;
        size_t na = 0, nb = 0;
        const double g = median_us([&] { na = gpu.scan(pssm, seq, 10.0f).size(); }, reps);
        const double c = median_us([&] { nb = cpu.scan(pssm, seq, 10.0f).size(); }, reps);
        const double gm = median_us([&] { (void)gpu.scan_max(pssm, seq, 10.0f); }, reps);
        const double cm = median_us([&] { (void)cpu.scan_max(pssm, seq, 10.0f); }, reps);
        std::printf("%s{\"length\": %zu, \"hits\": %zu, \"same_hit_count\": %s, \"scan_host_pointer_us\": %.2f, \"scan_cpu_tier_us\": %.2f, "
                    "\"scan_max_host_pointer_us\": %.2f, \"scan_max_cpu_tier_us\": %.2f}",
                    first ? "" : ", ", len, na, na == nb ? "true" : "false", g, c, gm, cm);
        first = false;
    }
    std::printf("]}\n");
    return 0;
}

// the reference's bench loop (dna.rs:81-116): score_into, then argmax on the matrix just written -- with Hip::reuse_scores
// the reduction takes the copy the score call left on the device; same cells, same bits, and only for THAT matrix
static void test_reuse_scores()
{
    HipDispatch<PortTier> forced, never;
    for (int op = 0; op < 9; ++op) {
        forced.policy.force((lm_hip_host_op)op, 0);
        never.policy.force((lm_hip_host_op)op, SIZE_MAX);
    }
    const auto pssm = mx000001();
    host::StripedSequence<Dna> seq = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode(random_dna(464'165, 7)), 32);
    seq.configure(pssm);
    auto count = [] {
        size_t n = 0;
        CHECK(lm_hip_host_reuse_count(&n) == LM_HIP_OK);
        return n;
    };
    Hip::reuse_scores(true);
    const size_t c0 = count();
    const auto s = forced.score(pssm, seq);
    const auto a = forced.argmax(s);
    CHECK(count() == c0 + 1);
    const auto want = never.argmax(never.score(pssm, seq));
    CHECK(a && want && a->row == want->row && a->col == want->col);
    const auto mx = forced.max(s);
    CHECK(count() == c0 + 2 && mx);
    CHECK(same_cells(forced.threshold(s, *mx - 4.0f), never.threshold(s, *mx - 4.0f)) && count() == c0 + 3);
    const auto other = never.score(pssm, seq);   // same contents, another matrix: uploaded
    const auto b = forced.argmax(other);
    CHECK(count() == c0 + 3 && b && b->row == want->row && b->col == want->col);
    Hip::reuse_scores(false);
    const auto s2 = forced.score(pssm, seq);
    (void)forced.argmax(s2);
    CHECK(count() == c0 + 3);
}

// the policy's constants re-measured on THIS host (the reference picks its back-end per host at run time, pli/mod.rs:269-308):
// the GPU side by the library, the tier's side by the twin; the routes still agree bit for bit either side of the new numbers
static void test_calibrated_policy()
{
    double t0 = 0, g = 0, c = 0, cr = 0;
    int measured = 1;
    CHECK(lm_hip_host_cost_model(LM_HIP_OP_SCORE_F32, &t0, &g, &c, &cr, &measured) == LM_HIP_OK && !measured && t0 == 47.0);
    HipDispatch<PortTier> pli;
    const auto cal = pli.calibrate(60.0);
    CHECK(lm_hip_host_cost_model(LM_HIP_OP_SCORE_F32, &t0, &g, &c, &cr, &measured) == LM_HIP_OK && measured);
    CHECK(t0 > 2.0 && t0 < 2000.0 && g > 0.005 && g < 5.0 && cr > 0.001 && cr < 5.0);   // a launch + link, a SIMD kernel: sane magnitudes
    std::printf("calibrated on this host: score_f32 t0 %.1f us, link %.4f ns/cell, tier %.4f + %.4f M ns/cell -> crossovers (cells): "
                "score_f32(M = 16) %zu, score_u8(M = 16) %zu, maximum_f32 %zu, threshold_f32 %zu, scan(M = 16) %zu\n",
                t0, g, c, cr, cal.score_f32_m16, cal.score_u8_m16, cal.maximum_f32, cal.threshold_f32, cal.scan_m16);
    CHECK(cal.score_f32_m16 > 500 && cal.score_f32_m16 < ((size_t)1 << 28));
    // either side of the calibrated crossover: different routes, identical bits
    const auto pssm = mx000001();
    size_t x = 0;
    CHECK(lm_hip_host_crossover(LM_HIP_OP_SCORE_F32, pssm.len(), 5, &x) == LM_HIP_OK && x > 64 && x < ((size_t)1 << 28));
    const std::string text = random_dna(2 * x + 4096, 99);
    host::StripedSequence<Dna> below = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode(text.substr(0, x / 2)), 32);
    host::StripedSequence<Dna> above = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode(text), 32);
    below.configure(pssm);
    above.configure(pssm);
    HipDispatch<PortTier> all_cpu, all_gpu;
    for (int op = 0; op < 9; ++op) {
        all_cpu.policy.force((lm_hip_host_op)op, SIZE_MAX);
        all_gpu.policy.force((lm_hip_host_op)op, 0);
    }
    const auto s_below = pli.score(pssm, below);
    CHECK(pli.last_route == Route::Cpu && same_bits(s_below, all_gpu.score(pssm, below)));
    const auto s_above = pli.score(pssm, above);
    CHECK(pli.last_route == Route::Gpu && same_bits(s_above, all_cpu.score(pssm, above)));
    // pins: a site forced through the C ABI answers the pin whatever M, and goes back to the model
    CHECK(lm_hip_host_set_crossover(LM_HIP_OP_SCORE_F32, 12345) == LM_HIP_OK);
    CHECK(lm_hip_host_crossover(LM_HIP_OP_SCORE_F32, 30, 5, &x) == LM_HIP_OK && x == 12345);
    CHECK(lm_hip_host_set_crossover(LM_HIP_OP_SCORE_F32, LM_HIP_CROSSOVER_MODEL) == LM_HIP_OK);
    CHECK(lm_hip_host_crossover(LM_HIP_OP_SCORE_F32, 30, 5, &x) == LM_HIP_OK && x != 12345);
    CHECK(lm_hip_host_set_cpu_cost(LM_HIP_OP_ENCODE, 1.0, 0.0) == LM_HIP_ERR_BAD_ARGS);   // a site without a model
    CHECK(lm_hip_host_calibrate(-1.0, 0) == LM_HIP_ERR_BAD_ARGS);
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "--bench")
        return Hip::available() ? bench_scan() : 2;
    bool threw = false;
    if (!Hip::available()) {
        try {
            HipDispatch<PortTier> none;
        } catch (const UnsupportedBackend &) {
            threw = true;   // without a device the variant cannot be made: there is no CPU-only `Hip`
        }
        std::fprintf(stderr, "UnsupportedBackend: no gfx950 device%s\n", threw ? "" : " (but the variant was constructed!)");
        return threw ? 2 : 1;
    }
    test_policy_routes_by_size();
    test_scanner_specialisation();
    test_other_geometries();
    test_reuse_scores();
    test_calibrated_policy();   // last: it replaces the process's cost model
    if (failures) {
        std::fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    std::puts("test_dispatch: all checks passed");
    return 0;
}
