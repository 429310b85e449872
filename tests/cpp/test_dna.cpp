// C++ counterpart of the reference's integration tests for the scoring path, written
// against the host mirror (lightmotif_amd/host/lightmotif_hip.hpp) so it reads like
//   lightmotif/tests/dna.rs     (score_rows, score, argmax, threshold)
//   lightmotif/tests/stripe.rs  (stripe property)
//   lightmotif/tests/encode.rs  (encode / unknown symbol)
//   lightmotif/src/pli/mod.rs:603-623 (empty row range)
// The expected values are the literals those tests hold (tests/golden/reference_vectors.json).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "lightmotif_hip.hpp"

using namespace lightmotif;

static int failures = 0;
#define CHECK(cond)                                                             \
    do {                                                                        \
        if (!(cond)) {                                                          \
            std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                         \
        }                                                                       \
    } while (0)

static const char *SEQUENCE = "ATGTCCCAACAACGATACCCCGAGCCCATCGCCGTCATCGGCTCGGCATGCAGATTCCCAGGCG";
static const std::vector<std::string> PATTERNS = {"GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"};

// scores computed with Bio.motifs (tests/dna.rs:22-38)
static const float EXPECTED[50] = {
    -23.07094f,  -18.678621f, -15.219191f, -17.745737f, -18.678621f, -23.07094f,  -17.745737f,
    -19.611507f, -27.463257f, -29.989803f, -14.286304f, -26.53037f,  -15.219191f, -10.826873f,
    -10.826873f, -22.138054f, -38.774437f, -30.922688f, -5.50167f,   -24.003826f, -18.678621f,
    -15.219191f, -35.315006f, -17.745737f, -10.826873f, -30.922688f, -23.07094f,  -6.4345555f,
    -31.855574f, -23.07094f,  -15.219191f, -31.855574f, -8.961102f,  -26.53037f,  -27.463257f,
    -14.286304f, -15.219191f, -26.53037f,  -23.07094f,  -18.678621f, -14.286304f, -18.678621f,
    -26.53037f,  -16.152077f, -17.745737f, -18.678621f, -17.745737f, -14.286304f, -30.922688f,
    -18.678621f};

static ScoringMatrix<Dna> golden_pssm()
{
    std::vector<EncodedSequence<Dna>> sites;
    for (const auto &p : PATTERNS)
        sites.push_back(EncodedSequence<Dna>::encode(p));
    const auto cm = CountMatrix<Dna>::from_sequences(sites);
    const auto pbm = cm.to_freq(0.1f);
    const auto pwm = pbm.to_weight();
    return pwm.to_scoring();
}

// tests/dna.rs:40-63
static void test_score_rows(const Pipeline<Dna> &pli, size_t columns)
{
    const auto encoded = EncodedSequence<Dna>::encode(SEQUENCE);
    auto striped = pli.stripe(encoded, columns);
    const auto pssm = golden_pssm();
    striped.configure(pssm);
    auto scores = pli.empty_scores(columns);

    const size_t rows = (64 + columns - 1) / columns;
    pli.score_rows_into(pssm, striped, 0, std::min<size_t>(2, rows), scores);
    CHECK(scores.matrix().rows() == std::min<size_t>(2, rows));
    CHECK(scores.matrix()(0, 0) == EXPECTED[0]);
    if (rows > 1) {
        CHECK(scores.matrix()(1, 0) == EXPECTED[1]);
        pli.score_rows_into(pssm, striped, 1, 2, scores);
        CHECK(scores.matrix().rows() == 1);
        CHECK(scores.matrix()(0, 0) == EXPECTED[1]);
    }
}

// tests/dna.rs:65-91
static void test_score(const Pipeline<Dna> &pli, size_t columns)
{
    auto striped = pli.stripe(EncodedSequence<Dna>::encode(SEQUENCE), columns);
    const auto pssm = golden_pssm();
    striped.configure(pssm);
    const auto result = pli.score(pssm, striped);
    const auto scores = result.unstripe();
    CHECK(scores.size() == 50);
    for (size_t i = 0; i < scores.size() && i < 50; ++i)
        CHECK(std::fabs(scores[i] - EXPECTED[i]) < 1e-5f);
}

// tests/dna.rs:123-139
static void test_argmax(const Pipeline<Dna> &pli, size_t columns)
{
    auto striped = pli.stripe(EncodedSequence<Dna>::encode(SEQUENCE), columns);
    const auto pssm = golden_pssm();
    striped.configure(pssm);
    const auto result = pli.score(pssm, striped);
    const auto mc = pli.argmax(result);
    CHECK(mc.has_value() && result.offset(*mc) == 18);
    CHECK(result.argmax() == std::optional<size_t>(18));            // README.md:85-86
    CHECK(std::fabs(*result.max() - (-5.50167f)) < 1e-5f);
}

// tests/dna.rs:141-173
static void test_threshold(const Pipeline<Dna> &pli, size_t columns)
{
    auto striped = pli.stripe(EncodedSequence<Dna>::encode(SEQUENCE), columns);
    const auto pssm = golden_pssm();
    striped.configure(pssm);
    const auto result = pli.score(pssm, striped);

    std::vector<size_t> indices;
    for (const auto &c : pli.threshold(result, -10.0f))
        indices.push_back(result.offset(c));
    std::sort(indices.begin(), indices.end());
    CHECK((indices == std::vector<size_t>{18, 27, 32}));

    indices.clear();
    for (const auto &c : pli.threshold(result, -15.0f))
        indices.push_back(result.offset(c));
    std::sort(indices.begin(), indices.end());
    CHECK((indices == std::vector<size_t>{10, 13, 14, 18, 24, 27, 32, 35, 40, 47}));
    CHECK(result.threshold(10.0f).empty());                          // README.md:89-90
    CHECK(std::fabs(result.at(18) - (-5.50167f)) < 1e-5f);           // scores[i], scores.rs:246-254
    CHECK(result.at(0) == -23.07094f);
}

// tests/dna.rs:93-120 `test_score_discrete`: unscale(u8 score) >= the f32 score, position by position
static void test_score_discrete(const Pipeline<Dna> &pli)
{
    auto striped = pli.stripe(EncodedSequence<Dna>::encode(SEQUENCE));
    const auto pssm = golden_pssm();
    const auto dm = pssm.to_discrete();
    striped.configure(pssm);
    const auto result = pli.score(dm, striped);
    CHECK(result.len() == sizeof(EXPECTED) / sizeof(EXPECTED[0]));
    for (size_t i = 0; i < result.len(); ++i)
        CHECK(dm.unscale(result.at(i)) >= EXPECTED[i]);
    // no sum reaches 255 here, so Generic's wrapping adds give the same matrix
    const auto wrapped = pli.score(dm, striped, false);
    for (size_t i = 0; i < result.len(); ++i)
        CHECK(wrapped.at(i) == result.at(i));
    // scan.rs:169-172: the scaled threshold under-estimates, so every hit of the Scanner test
    // (positions 18, 27, 32 at t = -10) passes the u8 test
    const uint8_t t = dm.scale(-10.0f);
    CHECK(result.at(18) >= t && result.at(27) >= t && result.at(32) >= t);
    CHECK(dm.scale(-1e30f) == 0 && dm.scale(1e30f) == 255);
}

// pwm/mod.rs:566-577: the reverse complement of the matrix scores the reverse complement of the
// sequence like the matrix scores the sequence (position i <-> L - M - i); the library's resident
// form (lm_hip_pssm_reverse_complement) gives the same matrix
static void test_reverse_complement(const Pipeline<Dna> &pli, lm_hip_ctx *ctx)
{
    const auto pssm = golden_pssm();
    const auto rc = pssm.reverse_complement();
    std::string rev(SEQUENCE);
    std::reverse(rev.begin(), rev.end());
    for (char &c : rev)
        c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
    auto fwd = pli.stripe(EncodedSequence<Dna>::encode(SEQUENCE));
    auto bwd = pli.stripe(EncodedSequence<Dna>::encode(rev));
    fwd.configure(pssm);
    bwd.configure(rc);
    const auto a = pli.score(pssm, fwd), b = pli.score(rc, bwd);
    const size_t n = a.len();
    CHECK(n == 50 && b.len() == n);
    for (size_t i = 0; i < n; ++i)
        CHECK(std::fabs(a.at(i) - b.at(n - 1 - i)) < 1e-4f);  // same terms, added in the opposite order
    lm_hip_pssm *dev_rc = nullptr;
    CHECK(lm_hip_pssm_reverse_complement(ctx, pssm.device(ctx), &dev_rc) == LM_HIP_OK);
    if (dev_rc) {
        int found = 0, found2 = 0;
        lm_hip_coords best{}, best2{};
        float v = 0, v2 = 0;
        lm_hip_seq *h = bwd.handle();
        size_t len = 0, wrap = 0, rows = 0, stride = 0, cols = 0;
        const uint8_t *dptr = nullptr;
        CHECK(lm_hip_seq_info(h, &len, &wrap, &rows, &stride, &cols, &dptr) == LM_HIP_OK);
        CHECK(lm_hip_score_argmax_f32_dptr(ctx, dev_rc, dptr, rows + wrap, stride, cols, wrap, len, 0, rows, &found,
                                           &best, &v) == LM_HIP_OK);
        CHECK(lm_hip_score_argmax_f32_dptr(ctx, rc.device(ctx), dptr, rows + wrap, stride, cols, wrap, len, 0, rows,
                                           &found2, &best2, &v2) == LM_HIP_OK);
        CHECK(found && found2 && best.row == best2.row && best.col == best2.col && v == v2);
        lm_hip_pssm_destroy(dev_rc);
    }
}

// scan.rs:279-353 / lightmotif-py test_scanner.py:64-80
static void test_scanner(const Pipeline<Dna> &pli)
{
    auto striped = pli.stripe(EncodedSequence<Dna>::encode(SEQUENCE));
    const auto pssm = golden_pssm();
    striped.configure(pssm);
    CHECK(pli.scan(pssm, striped, 0.0f).empty());
    const auto hits = pli.scan(pssm, striped, -10.0f);
    CHECK(hits.size() == 3);
    if (hits.size() == 3) {
        CHECK(hits[0].position == 18 && std::fabs(hits[0].score - (-5.50167f)) < 1e-5f);
        CHECK(hits[1].position == 27 && std::fabs(hits[1].score - (-6.4345555f)) < 1e-5f);
        CHECK(hits[2].position == 32 && std::fabs(hits[2].score - (-8.961102f)) < 1e-5f);
    }
    // the reference's yield order (scan.rs:184-198): 64 bp at C = 32 are 2 rows, the hits sit in
    // cells (0, 9), (1, 13), (0, 16); one block of 256 rows pushes 18, 32, 27 and pops 27, 32, 18;
    // blocks of one row yield 32, 18 (row 0) and then 27
    const auto one_block = Pipeline<Dna>::scan_order(hits, striped.rows());
    CHECK(one_block.size() == 3 && one_block[0].position == 27 && one_block[1].position == 32 &&
          one_block[2].position == 18);
    const auto row_blocks = Pipeline<Dna>::scan_order(hits, striped.rows(), 1);
    CHECK(row_blocks.size() == 3 && row_blocks[0].position == 32 && row_blocks[1].position == 18 &&
          row_blocks[2].position == 27);
    const auto best = Pipeline<Dna>::scan_max_valid(hits);
    CHECK(best && best->position == 18 && std::fabs(best->score - (-5.50167f)) < 1e-5f);
    CHECK(!Pipeline<Dna>::scan_max_valid({}));
    // Scanner::max as the reference walks it, on the device (scan.rs:317-333: best = (18, -5.50167) at t = -10)
    const auto walked = pli.scan_max(pssm, striped, -10.0f);
    CHECK(walked && walked->position == 18 && std::fabs(walked->score - (-5.50167f)) < 1e-5f);
}

// Many motifs over one resident sequence (lightmotif-cli main.rs:554-561 fans the motifs
// out with rayon; here one batched call): each motif must give what the single-motif
// trait calls give (tests/dna.rs:104-137 literals for the golden motif).
static void test_batch(const Pipeline<Dna> &pli)
{
    auto striped = pli.stripe_text(SEQUENCE);                         // device-side encode + stripe
    const auto reference = pli.stripe(EncodedSequence<Dna>::encode(SEQUENCE));
    const auto a = striped.matrix(), b = reference.matrix();
    CHECK(a.rows() == b.rows());
    for (size_t r = 0; r < a.rows() && r < b.rows(); ++r)
        for (size_t c = 0; c < 32; ++c)
            CHECK(a(r, c) == b(r, c));
    const auto pssm = golden_pssm();
    // a second, shorter motif: the first 8 columns of the golden one
    DenseMatrix<float> head(8, Dna::K);
    for (size_t i = 0; i < 8; ++i)
        for (size_t j = 0; j < Dna::K; ++j)
            head(i, j) = pssm.matrix()(i, j);
    const ScoringMatrix<Dna> shorter(pssm.background, head);
    striped.configure(pssm);
    const std::vector<const ScoringMatrix<Dna> *> motifs = {&pssm, &shorter};

    // fused single-motif forms == score() followed by argmax() / threshold(), knobs change nothing
    for (bool on : {true, false}) {
        pli.set_prefilter(on);
        pli.set_track_argmax(on);
        const auto scores = pli.score(pssm, striped);
        const auto fused = pli.score_argmax(pssm, striped);
        CHECK(fused && pli.argmax(scores) && fused->cell == *pli.argmax(scores) && fused->score == *scores.max());
        const auto cells = pli.score_threshold(pssm, striped, -15.0f);
        CHECK(cells.coords == pli.threshold(scores, -15.0f) && cells.coords.size() == 10);
    }
    pli.set_prefilter(true);
    pli.set_track_argmax(true);

    const auto best = pli.scan_argmax_batch(motifs, striped);
    CHECK(best.size() == 2 && best[0] && best[1]);
    for (size_t i = 0; i < 2 && i < best.size(); ++i) {
        const auto scores = pli.score(*motifs[i], striped);
        const auto want = pli.argmax(scores);
        CHECK(want && best[i] && *want == best[i]->cell);
        CHECK(best[i] && best[i]->score == *scores.max());
    }
    if (best.size() == 2 && best[0]) {
        const size_t rows = striped.matrix().rows() - striped.wrap();
        CHECK(best[0]->cell.col * rows + best[0]->cell.row == 18);     // tests/dna.rs:104-111
    }

    const auto cells = pli.scan_threshold_batch(motifs, {-10.0f, -4.0f}, striped);
    CHECK(cells.size() == 2);
    for (size_t i = 0; i < 2 && i < cells.size(); ++i) {
        const auto scores = pli.score(*motifs[i], striped);
        const auto want = pli.threshold(scores, i == 0 ? -10.0f : -4.0f);
        CHECK(want == cells[i].coords);
        const auto m = scores.matrix();
        for (size_t k = 0; k < cells[i].coords.size(); ++k)
            CHECK(cells[i].scores[k] == m(cells[i].coords[k].row, cells[i].coords[k].col));
    }
    if (cells.size() == 2) {
        std::vector<size_t> idx;
        const size_t rows = pli.score(pssm, striped).rows();
        for (const auto &c : cells[0].coords)
            idx.push_back(c.col * rows + c.row);
        std::sort(idx.begin(), idx.end());
        CHECK((idx == std::vector<size_t>{18, 27, 32}));              // tests/dna.rs:124-128
    }
    bool threw = false;
    try {
        pli.stripe_text("ATGCZ");                                     // tests/encode.rs: unknown symbol
    } catch (const InvalidSymbol &e) {
        threw = e.symbol == 'Z';
    }
    CHECK(threw);
}

// tests/stripe.rs:17-45
static void test_stripe(const Pipeline<Dna> &pli, const std::string &sequence, size_t columns)
{
    const auto encoded = EncodedSequence<Dna>::encode(sequence);
    const auto striped = pli.stripe(encoded, columns);
    const auto matrix = striped.matrix();
    if (matrix.rows() > 0) CHECK(matrix(0, 0) == 0 /* A */);
    if (matrix.rows() > 1) CHECK(matrix(1, 0) == 2 /* T */);
    if (matrix.rows() > 2) CHECK(matrix(2, 0) == 3 /* G */);
    if (matrix.rows() > 3) CHECK(matrix(3, 0) == 2 /* T */);
    for (size_t i = 0; i < encoded.len(); ++i)
        CHECK(matrix(i % matrix.rows(), i / matrix.rows()) == encoded.data[i]);
    for (size_t i = sequence.size(); i < matrix.rows() * matrix.columns(); ++i)
        CHECK(matrix(i % matrix.rows(), i / matrix.rows()) == 4 /* Nucleotide::default() */);
}

// seq.rs:509-540
static void test_stripe_literals(const Pipeline<Dna> &pli)
{
    auto striped = pli.stripe(EncodedSequence<Dna>::encode("ATGCA"), 4);
    auto m = striped.matrix();
    const uint8_t A = 0, C = 1, T = 2, G = 3, N = 4;
    CHECK(m.rows() == 2);
    CHECK(m(0, 0) == A && m(0, 1) == G && m(0, 2) == A && m(0, 3) == N);
    CHECK(m(1, 0) == T && m(1, 1) == C && m(1, 2) == N && m(1, 3) == N);
    striped.configure_wrap(2);
    m = striped.matrix();
    CHECK(m.rows() == 4);
    CHECK(m(2, 0) == G && m(2, 1) == A && m(2, 2) == N && m(2, 3) == N);
    CHECK(m(3, 0) == C && m(3, 1) == N && m(3, 2) == N && m(3, 3) == N);
}

// tests/encode.rs:10-26
static void test_encode(const Pipeline<Dna> &pli)
{
    const auto encoded = pli.encode(SEQUENCE);
    CHECK(encoded.len() == 64 && encoded.data[0] == 0 && encoded.data[1] == 2 && encoded.data[2] == 3);
    bool threw = false;
    try {
        pli.encode("ATGTCCCAACAACGATACCNN..................NNNNNNNNATGCAGATTCCCAGGCG");
    } catch (const InvalidSymbol &e) {
        threw = e.symbol == '.';
    }
    CHECK(threw);
}

// pli/mod.rs:603-623 + avx2.rs:832-837
static void test_edge_cases(const Pipeline<Dna> &pli)
{
    auto striped = pli.stripe(EncodedSequence<Dna>::encode("ATGCA"), 4);
    std::vector<EncodedSequence<Dna>> sites = {EncodedSequence<Dna>::encode("ATTA"),
                                               EncodedSequence<Dna>::encode("ATTC")};
    const auto pssm = CountMatrix<Dna>::from_sequences(sites).to_freq(0.1f).to_weight().to_scoring();
    striped.configure(pssm);
    auto scores = pli.empty_scores(4);
    pli.score_rows_into(pssm, striped, 1, 1, scores);   // must not fail
    CHECK(scores.is_empty() && scores.max_index() == 0);
    CHECK(!pli.argmax(scores).has_value() && pli.threshold(scores, 0.0f).empty());

    auto bare = pli.stripe(EncodedSequence<Dna>::encode(SEQUENCE));
    bool threw = false;
    try {
        pli.score(golden_pssm(), bare);                 // no wrap rows configured
    } catch (const std::runtime_error &e) {
        threw = std::string(e.what()).find("not enough wrapping rows for motif of length 15") != std::string::npos;
    }
    CHECK(threw);
}

// The row-sharded path through the C++ mirror: a world of one rank (two ranks cannot share the
// box's single GPU under RCCL) and the merge rule on explicit per-shard records.
static void test_sharded(const Pipeline<Dna> &pli)
{
    // (1) combine rule: ties go to the later (row, col); NaN only as shard 0's first cell
    {
        std::vector<ShardBest> s = {{true, {5, 3}, 2.0f}, {false, {}, 0.0f}, {true, {9, 0}, 2.0f}, {true, {12, 31}, 1.5f}};
        const ShardBest b = combine_argmax(s);
        CHECK(b.found && b.cell.row == 9 && b.cell.col == 0 && b.score == 2.0f);
        s[0] = {true, {0, 0}, std::nanf("")};
        const ShardBest n = combine_argmax(s);
        CHECK(n.found && n.cell.row == 0 && n.cell.col == 0 && n.score != n.score);
        s[0] = {true, {4, 4}, std::nanf("")};          // a NaN that is not the first cell never wins
        CHECK(combine_argmax(s).cell.row == 9);
        CHECK(!combine_argmax({{false, {}, 0.0f}}).found);
    }
    // (2) one rank: score_into + argmax / threshold through the communicator == the plain calls
    const auto pssm = golden_pssm();
    auto striped = pli.stripe(EncodedSequence<Dna>::encode(SEQUENCE));
    striped.configure(pssm);
    auto scores = pli.score(pssm, striped);
    ShardComm<Dna> comm(pli, ShardComm<Dna>::unique_id(), 1, 0);
    scores.set_first_cell_rule(true);
    const ShardBest best = comm.argmax(scores, 0);
    CHECK(best.found && scores.offset(best.cell) == 18);             // tests/dna.rs:138
    const auto hits = comm.threshold(scores, -10.0f, 0);
    std::vector<size_t> pos;
    for (const auto &h : hits)
        pos.push_back(scores.offset(h));
    std::sort(pos.begin(), pos.end());
    CHECK((pos == std::vector<size_t>{18, 27, 32}));                 // tests/dna.rs:158-165
    const auto shifted = comm.threshold(scores, -10.0f, 1000);       // rows become global
    CHECK(shifted.size() == 3 && shifted[0].row == hits[0].row + 1000);
    // (3) the two-halves form: the merge is collected after the shard was scored over
    const int t0 = comm.argmax_begin(scores, 0);
    pli.score_into(pssm, striped, scores);
    const int t1 = comm.argmax_begin(scores, 0);
    const ShardBest b0 = comm.argmax_end(t0), b1 = comm.argmax_end(t1);
    CHECK(b0.found && scores.offset(b0.cell) == 18 && b0.score == best.score);
    CHECK(b1.found && scores.offset(b1.cell) == 18 && b1.score == best.score);
}

// The residency flow a Rust caller takes to profit from the GPU: the genome goes up ONCE --
// as text, as symbol bytes or as 2-bit -- and stays; matrices, scores and reductions work on the resident handles.
static void test_resident_flow(const Pipeline<Dna> &pli)
{
    std::string genome;
    for (int i = 0; i < 4000; ++i)
        genome += SEQUENCE;
    genome.replace(1000, 37, std::string(37, 'N'));                     // an N run
    genome[200001] = 'N';
    const auto enc = EncodedSequence<Dna>::encode(genome);
    auto from_text = pli.stripe_text(genome);                             // lm_hip_seq_from_ascii: tiled, encode fused
    auto from_bytes = pli.stripe(enc);                                    // lm_hip_seq_from_encoded: tiled, validated
    auto from_2bit = pli.stripe_2bit(Pipeline<Dna>::pack_2bit(enc));      // lm_hip_seq_from_2bit
    const auto a = from_text.matrix(), b = from_bytes.matrix(), c = from_2bit.matrix();
    CHECK(a.rows() == b.rows() && a.rows() == c.rows());
    bool same = true;
    for (size_t r = 0; r < a.rows() && same; ++r)
        for (size_t j = 0; j < 32; ++j)
            same = same && a(r, j) == b(r, j) && a(r, j) == c(r, j);
    CHECK(same);
    const auto pssm = golden_pssm();
    from_2bit.configure(pssm);
    from_text.configure(pssm);
    StripedScores s2 = pli.empty_scores(32), st = pli.empty_scores(32);
    for (int rep = 0; rep < 3; ++rep) {                                   // dna.rs:104-107: one StripedScores re-used
        pli.score_into(pssm, from_2bit, s2);
        pli.score_into(pssm, from_text, st);
        const auto b2 = s2.argmax(), bt = st.argmax();
        CHECK(b2 && bt && *b2 == *bt);
        const auto m2 = s2.max(), mt = st.max();
        CHECK(m2 && mt && *m2 == *mt && *m2 == s2.at(*b2) && *m2 >= -5.50167f - 1e-5f);   // at least the golden best site
    }
    CHECK(s2.threshold(-5.6f) == st.threshold(-5.6f) && !s2.threshold(-5.6f).empty());
}

// lightmotif/tests/dna.rs as it would run with `Dispatch::Hip` selected: the sequence, the PSSM and the scores are the
// reference's own HOST structs (striped and wrapped by the Generic loops), and every call goes through the host-pointer
// entry points the Rust shim of INTEGRATION.md binds (`Hip` = the C++ twin of its `impl Hip`).
static void test_dispatch_hip_on_host_matrices(size_t columns)
{
    CHECK(Hip::available());
    const auto encoded = EncodedSequence<Dna>::encode(SEQUENCE);
    auto striped = host::StripedSequence<Dna>::stripe(encoded, columns);
    const auto pssm = golden_pssm();
    striped.configure(pssm);
    CHECK(striped.wrap() == pssm.len() - 1);

    // tests/dna.rs:40-63 test_score_rows
    host::StripedScores<float> part(columns);
    const size_t rows = (64 + columns - 1) / columns;
    Hip::score_rows_into(pssm, striped, 0, std::min<size_t>(2, rows), part);
    CHECK(part.matrix().rows() == std::min<size_t>(2, rows));
    CHECK(part.matrix()(0, 0) == EXPECTED[0]);
    if (rows > 1) {
        CHECK(part.matrix()(1, 0) == EXPECTED[1]);
        Hip::score_rows_into(pssm, striped, 1, 2, part);
        CHECK(part.matrix().rows() == 1 && part.matrix()(0, 0) == EXPECTED[1]);
    }
    Hip::score_rows_into(pssm, striped, 1, 1, part);         // pli/mod.rs:603-623: an empty range resizes to nothing
    CHECK(part.matrix().rows() == 0 && part.max_index() == 0);

    // tests/dna.rs:65-91 test_score
    const auto result = Hip::score(pssm, striped);
    const auto scores = result.unstripe();
    CHECK(scores.size() == 50);
    for (size_t i = 0; i < scores.size() && i < 50; ++i)
        CHECK(std::fabs(scores[i] - EXPECTED[i]) < 1e-5f);

    // tests/dna.rs:123-139 test_argmax, scores.rs:181-192
    const auto mc = Hip::argmax(result);
    CHECK(mc.has_value() && result.offset(*mc) == 18);
    const auto top = Hip::max(result);
    CHECK(top.has_value() && std::fabs(*top - (-5.50167f)) < 1e-5f);

    // tests/dna.rs:141-173 test_threshold
    std::vector<size_t> indices;
    for (const auto &c : Hip::threshold(result, -10.0f))
        indices.push_back(result.offset(c));
    if (columns == 32)
        CHECK((indices == std::vector<size_t>{18, 32, 27}));         // the reference's row-major push order, unsorted
    std::sort(indices.begin(), indices.end());
    CHECK((indices == std::vector<size_t>{18, 27, 32}));
    indices.clear();
    for (const auto &c : Hip::threshold(result, -15.0f))
        indices.push_back(result.offset(c));
    std::sort(indices.begin(), indices.end());
    CHECK((indices == std::vector<size_t>{10, 13, 14, 18, 24, 27, 32, 35, 40, 47}));
    CHECK(Hip::threshold(result, 10.0f).empty());

    // tests/dna.rs:93-120 test_score_discrete: unscale(u8 score) >= the f32 score, position by position
    const auto dm = pssm.to_discrete();
    host::StripedScores<uint8_t> dscores(columns);
    Hip::score_rows_into(dm, striped, 0, striped.matrix().rows() - striped.wrap(), dscores);
    const auto d = dscores.unstripe();
    CHECK(d.size() == 50);
    for (size_t i = 0; i < d.size() && i < 50; ++i)
        CHECK(dm.unscale(d[i]) >= scores[i]);

    // a sequence shorter than the motif: Ok, nothing scored (pli/mod.rs:85-88); too few wrap rows: the reference panics
    auto tiny = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode("ACGT"), columns);
    tiny.configure(pssm);
    const auto none = Hip::score(pssm, tiny);
    CHECK(none.matrix().rows() == 0 && !Hip::argmax(none).has_value());
    auto bare = host::StripedSequence<Dna>::stripe(encoded, columns);   // no configure(): wrap = 0
    bool threw = false;
    try {
        (void)Hip::score(pssm, bare);
    } catch (const std::runtime_error &e) {
        threw = std::string(e.what()).find("wrapping rows") != std::string::npos;   // avx2.rs:832-837
    }
    CHECK(threw);
}

// lightmotif/tests/argmax.rs:23-52 with `Dispatch::Hip`: a tenth of an E.-coli-sized sequence (random stand-in, the PRODORIC
// site planted once so that the maximum is unique, which is all the reference's test exercises), scores from the back-end,
// the expected cell from `unstripe().max_by(..)` on the host.
static void test_argmax_property_dispatch_hip(size_t columns)
{
    const size_t n = 464165;
    std::string seq(n, 'A');
    unsigned long long z = 0x5EED0003ull;
    for (size_t i = 0; i < n; ++i) {
        z = z * 6364136223846793005ull + 1442695040888963407ull;
        seq[i] = "ACGT"[(z >> 33) & 3];
    }
    seq.replace(391677, 15, "GTTGACCTTATCAAC");
    auto striped = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode(seq), columns);
    const auto pssm = golden_pssm();
    striped.configure(pssm);
    const auto scores = Hip::score(pssm, striped);
    const auto flat = scores.unstripe();
    CHECK(flat.size() == n - 15 + 1);
    size_t best = 0;
    for (size_t i = 1; i < flat.size(); ++i)
        if (!(flat[i] < flat[best]))   // max_by keeps the last of equal elements
            best = i;
    const auto m = Hip::argmax(scores);
    CHECK(m.has_value() && scores.offset(*m) == best && best == 391677);
    CHECK(m.has_value() && scores.matrix()(m->row, m->col) == flat[best]);
    const auto top = Hip::max(scores);
    CHECK(top.has_value() && *top == flat[best]);
}

int main()
{
    for (size_t columns : {32u, 1u, 16u}) {
        test_dispatch_hip_on_host_matrices(columns);
        test_argmax_property_dispatch_hip(columns);
    }
    Pipeline<Dna> pli = Pipeline<Dna>::hip();
    for (size_t columns : {32u, 1u, 16u}) {   // mod generic: U32, U1 (+ sse2's U16), tests/dna.rs:201-233
        test_score_rows(pli, columns);
        test_score(pli, columns);
        test_argmax(pli, columns);
        test_threshold(pli, columns);
    }
    std::string s1;
    for (int i = 0; i < 16; ++i) s1 += SEQUENCE;
    s1 += "TTATTAT";                              // tests/stripe.rs S1
    for (size_t columns : {32u, 16u}) {
        test_stripe(pli, s1, columns);
        test_stripe(pli, SEQUENCE, columns);
    }
    test_stripe_literals(pli);
    test_score_discrete(pli);
    test_reverse_complement(pli, pli.context());
    test_scanner(pli);
    test_batch(pli);
    test_encode(pli);
    test_edge_cases(pli);
    test_sharded(pli);
    test_resident_flow(pli);
    if (failures) {
        std::fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    std::puts("test_dna: all checks passed");
    return 0;
}
