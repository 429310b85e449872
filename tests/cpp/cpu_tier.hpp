// cpu_tier.hpp -- TEST INFRASTRUCTURE (like everything under oracle/): the CPU tier a `Dispatch::Hip { cpu }` variant
// carries, for the C++ twin `lightmotif::HipDispatch<Cpu>`.  In the reference the tier is its own `Avx2` / `Generic`
// back-end; this image has no rustc, so the stand-in is built on the oracle's C libraries (oracle/lm_avx2.c: the AVX2 score
// kernels, oracle/lm_oracle.c: the Generic bodies) plus the Scanner loop of scan.rs:169-249 restated over the tier.  It counts
// its calls so that tests can assert where `HipDispatch` sent a call.  Never part of the product.
#pragma once

#include <cstdlib>
#include <cstring>

#include "lightmotif_hip.hpp"

extern "C" {
size_t lmo_encode(char alphabet, const uint8_t *ascii, size_t len, int lossy, uint8_t *dst);
size_t lmo_stripe(const uint8_t *seq, size_t len, size_t cols, uint8_t default_symbol, uint8_t *data, size_t stride);
void lmo_score_rows_f32(const uint8_t *seq, size_t seq_stride, size_t cols, size_t length, const float *pssm, size_t m,
                        size_t pssm_stride, size_t row_begin, size_t row_end, float *out, size_t out_stride, size_t *out_rows,
                        size_t *max_index);
void lmo_score_rows_u8(const uint8_t *seq, size_t seq_stride, size_t cols, size_t length, const uint8_t *pssm, size_t m,
                       size_t pssm_stride, size_t row_begin, size_t row_end, uint8_t *out, size_t out_stride, size_t *out_rows,
                       size_t *max_index);
int lmo_argmax_f32(const float *scores, size_t rows, size_t stride, size_t cols, size_t *row, size_t *col);
size_t lmo_threshold_f32(const float *scores, size_t rows, size_t stride, size_t cols, float t, size_t *rc, size_t cap);
float lmo_score_position(const uint8_t *seq, size_t seq_stride, size_t rows, const float *pssm, size_t m, size_t pssm_stride,
                         size_t pos);
int lma_score_rows_f32(const uint8_t *seq, size_t seq_stride, size_t wrap, size_t length, const float *pssm, size_t m,
                       size_t pssm_stride, size_t k, size_t row_begin, size_t row_end, float *out, size_t out_stride);
int lma_score_rows_u8(const uint8_t *seq, size_t seq_stride, size_t wrap, size_t length, const uint8_t *weights, size_t m,
                      size_t wstride, size_t row_begin, size_t row_end, uint8_t *out, size_t out_stride);
}

namespace lightmotif_test {

using namespace lightmotif;

// 32-byte aligned copy of a matrix (the AVX2 kernels use aligned loads and stores like the reference, dense.rs:43;
// the mirror's DenseMatrix sits in a std::vector)
template <class T>
struct Aligned {
    T *p = nullptr;
    size_t n = 0;
    explicit Aligned(size_t count) : n(count)
    {
        if (posix_memalign(reinterpret_cast<void **>(&p), 32, std::max<size_t>(count * sizeof(T), 32)) != 0)
            throw std::bad_alloc();
    }
    Aligned(const T *src, size_t count) : Aligned(count)
    {
        if (count)
            std::memcpy(p, src, count * sizeof(T));
    }
    ~Aligned() { free(p); }
    Aligned(const Aligned &) = delete;
};

// `Hip { cpu: Avx2 }`: Score on the AVX2 kernels (C = 32; other column counts: the Generic loops, as `Dispatch` does for
// U1 / U16), every reduction on the Generic default bodies.
struct PortTier {
    static constexpr bool saturating_u8 = true;  // avx2.rs:336
    struct Counts {
        size_t encode = 0, stripe = 0, score_f32 = 0, score_u8 = 0, maximum_f32 = 0, threshold_f32 = 0, maximum_u8 = 0,
               threshold_u8 = 0, scan = 0;
    };
    mutable Counts n;
    using Hit = Hip::Hit;

    template <class A>
    void encode_into(const std::string &text, std::vector<uint8_t> &dst) const  // pli/mod.rs:56-66
    {
        ++n.encode;
        dst.resize(text.size());
        const size_t bad = lmo_encode(A::K == 21 ? 'P' : 'D', reinterpret_cast<const uint8_t *>(text.data()), text.size(), 0, dst.data());
        if (bad != 0)  // index + 1 of the first invalid byte
            throw InvalidSymbol(text[bad - 1]);
    }
    template <class A>
    void stripe_into(const EncodedSequence<A> &seq, host::StripedSequence<A> &striped) const  // pli/mod.rs:178-200
    {
        ++n.stripe;
        striped = host::StripedSequence<A>::stripe(seq, striped.columns());
    }
    template <class A>
    void score_rows_into(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq, size_t rb, size_t re,
                         host::StripedScores<float> &scores) const
    {
        ++n.score_f32;
        const DenseMatrix<float> &w = pssm.matrix();
        const DenseMatrix<uint8_t> &m = seq.matrix();
        if (w.rows() > 0 && seq.wrap() + 1 < w.rows())
            throw std::runtime_error("not enough wrapping rows for motif of length " + std::to_string(w.rows()));  // avx2.rs:832-837
        if (seq.len() < w.rows() || rb >= re) {
            scores.resize(0, 0);
            return;
        }
        scores.resize(re - rb, seq.len() + 1 - w.rows());
        size_t out_rows = 0, mi = 0;
        if (seq.columns() == 32) {
            Aligned<uint8_t> s(m.ptr(), m.rows() * m.stride());
            Aligned<float> p(w.ptr(), w.rows() * w.stride());
            Aligned<float> o((re - rb) * scores.data.stride());
            lma_score_rows_f32(s.p, m.stride(), seq.wrap(), seq.len(), p.p, w.rows(), w.stride(), A::K, rb, re, o.p,
                               scores.data.stride());
            std::memcpy(scores.data.ptr(), o.p, o.n * sizeof(float));
        } else {
            lmo_score_rows_f32(m.ptr(), m.stride(), seq.columns(), seq.len(), w.ptr(), w.rows(), w.stride(), rb, re,
                               scores.data.ptr(), scores.data.stride(), &out_rows, &mi);
        }
    }
    template <class A>
    void score_rows_into(const DiscreteMatrix<A> &dm, const host::StripedSequence<A> &seq, size_t rb, size_t re,
                         host::StripedScores<uint8_t> &scores) const
    {
        ++n.score_u8;
        const DenseMatrix<uint8_t> &w = dm.matrix();
        const DenseMatrix<uint8_t> &m = seq.matrix();
        if (seq.len() < w.rows() || rb >= re) {
            scores.resize(0, 0);
            return;
        }
        scores.resize(re - rb, seq.len() + 1 - w.rows());
        if (seq.columns() == 32) {
            Aligned<uint8_t> s(m.ptr(), m.rows() * m.stride());
            Aligned<uint8_t> p(w.ptr(), w.rows() * w.stride());
            Aligned<uint8_t> o((re - rb) * scores.data.stride());
            lma_score_rows_u8(s.p, m.stride(), seq.wrap(), seq.len(), p.p, w.rows(), w.stride(), rb, re, o.p, scores.data.stride());
            std::memcpy(scores.data.ptr(), o.p, o.n);
        } else {  // (saturating, to stay this tier's rule: Generic would wrap)
            for (size_t r = rb; r < re; ++r)
                for (size_t c = 0; c < seq.columns(); ++c) {
                    unsigned sum = 0;
                    for (size_t j = 0; j < w.rows(); ++j)
                        sum = std::min(255u, sum + w(j, m(r + j, c)));
                    scores.data(r - rb, c) = (uint8_t)sum;
                }
        }
    }
    // Maximum<f32>: the Generic rule (pli/mod.rs:135-160) -- what the GPU computes, so the variant has one answer at any size
    std::optional<MatrixCoordinates> argmax(const host::StripedScores<float> &scores) const
    {
        ++n.maximum_f32;
        const DenseMatrix<float> &d = scores.matrix();
        size_t r = 0, c = 0;
        if (!lmo_argmax_f32(d.ptr(), d.rows(), d.stride(), d.columns(), &r, &c))
            return std::nullopt;
        return MatrixCoordinates{r, c};
    }
    std::optional<float> max(const host::StripedScores<float> &scores) const
    {
        const auto mc = argmax(scores);
        return mc ? std::optional<float>(scores.matrix()(mc->row, mc->col)) : std::nullopt;
    }
    std::vector<MatrixCoordinates> threshold(const host::StripedScores<float> &scores, float t) const  // pli/mod.rs:210-221
    {
        ++n.threshold_f32;
        const DenseMatrix<float> &d = scores.matrix();
        std::vector<size_t> rc(2 * d.rows() * d.columns() + 2);
        const size_t k = lmo_threshold_f32(d.ptr(), d.rows(), d.stride(), d.columns(), t, rc.data(), d.rows() * d.columns());
        std::vector<MatrixCoordinates> out(k);
        for (size_t i = 0; i < k; ++i)
            out[i] = MatrixCoordinates{rc[2 * i], rc[2 * i + 1]};
        return out;
    }
    // Maximum<u8> / Threshold<u8>: the Generic default bodies (pli/mod.rs:135-160, 210-221)
    std::optional<MatrixCoordinates> argmax(const host::StripedScores<uint8_t> &scores) const
    {
        ++n.maximum_u8;
        const DenseMatrix<uint8_t> &d = scores.matrix();
        if (d.rows() == 0)
            return std::nullopt;
        size_t br = 0, bc = 0;
        uint8_t best = d(0, 0);
        for (size_t r = 0; r < d.rows(); ++r)
            for (size_t c = 0; c < d.columns(); ++c)
                if (d(r, c) >= best) {
                    best = d(r, c);
                    br = r;
                    bc = c;
                }
        return MatrixCoordinates{br, bc};
    }
    std::optional<uint8_t> max(const host::StripedScores<uint8_t> &scores) const
    {
        const auto mc = argmax(scores);
        return mc ? std::optional<uint8_t>(scores.matrix()(mc->row, mc->col)) : std::nullopt;
    }
    std::vector<MatrixCoordinates> threshold(const host::StripedScores<uint8_t> &scores, uint8_t t) const
    {
        ++n.threshold_u8;
        const DenseMatrix<uint8_t> &d = scores.matrix();
        std::vector<MatrixCoordinates> out;
        for (size_t r = 0; r < d.rows(); ++r)
            for (size_t c = 0; c < d.columns(); ++c)
                if (d(r, c) >= t)
                    out.push_back(MatrixCoordinates{r, c});
        return out;
    }
    // One Scanner block on copies aligned ONCE per scan (the port wants 32-byte aligned rows; a Rust `StripedSequence` is
    // aligned already -- copying the sequence per block made the loop quadratic and the crossover measurement meaningless)
    template <class A>
    void score_u8_block(const DiscreteMatrix<A> &dm, const host::StripedSequence<A> &seq, const Aligned<uint8_t> *s,
                        const Aligned<uint8_t> *p, Aligned<uint8_t> *o, size_t rb, size_t re, host::StripedScores<uint8_t> &scores) const
    {
        if (!s || seq.columns() != 32) {
            score_rows_into(dm, seq, rb, re, scores);
            return;
        }
        ++n.score_u8;
        const DenseMatrix<uint8_t> &w = dm.matrix();
        const DenseMatrix<uint8_t> &m = seq.matrix();
        if (seq.len() < w.rows() || rb >= re) {
            scores.resize(0, 0);
            return;
        }
        scores.resize(re - rb, seq.len() + 1 - w.rows());
        lma_score_rows_u8(s->p, m.stride(), seq.wrap(), seq.len(), p->p, w.rows(), w.stride(), rb, re, o->p, scores.data.stride());
        std::memcpy(scores.data.ptr(), o->p, (re - rb) * scores.data.stride());   // (the port streams to 32-byte aligned rows)
    }
    // Scanner::next until exhaustion, in yield order (scan.rs:169-198)
    template <class A>
    std::vector<Hit> scan(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq, float threshold, size_t block_size) const
    {
        ++n.scan;
        std::vector<Hit> out;
        if (seq.wrap() + 1 < pssm.len())
            throw std::runtime_error("not enough wrapping rows for motif of length " + std::to_string(pssm.len()));  // scan.rs:127-131
        const DiscreteMatrix<A> dm = pssm.to_discrete();
        const uint8_t t = dm.scale(threshold);
        const DenseMatrix<uint8_t> &m = seq.matrix();
        const DenseMatrix<float> &w = pssm.matrix();
        const size_t total = m.rows(), seq_rows = total - seq.wrap();
        host::StripedScores<uint8_t> ds(seq.columns());
        const bool port = seq.columns() == 32 && seq.len() >= pssm.len();
        const Aligned<uint8_t> sa(port ? m.ptr() : nullptr, port ? m.rows() * m.stride() : 0);
        const Aligned<uint8_t> pa(port ? dm.matrix().ptr() : nullptr, port ? dm.matrix().rows() * dm.matrix().stride() : 0);
        Aligned<uint8_t> oa(port ? block_size * ds.data.stride() + 64 : 0);
        for (size_t row = 0; row < total; row += block_size) {
            const size_t end = std::min(row + block_size, seq_rows);
            score_u8_block(dm, seq, port ? &sa : nullptr, port ? &pa : nullptr, &oa, row, end, ds);
            std::vector<Hit> hits;
            if (max(ds).value_or(0) >= t)
                for (const auto &c : threshold_u8(ds, t)) {
                    const size_t index = c.col * seq_rows + row + c.row;
                    if (index + pssm.len() <= seq.len()) {
                        const float score = lmo_score_position(m.ptr(), m.stride(), seq_rows, w.ptr(), w.rows(), w.stride(), index);
                        if (score >= threshold)
                            hits.push_back(Hit{index, score});
                    }
                }
            while (!hits.empty()) {  // self.hits.pop()
                out.push_back(hits.back());
                hits.pop_back();
            }
        }
        return out;
    }
    // Scanner::max on a fresh scanner (scan.rs:200-249)
    template <class A>
    std::optional<Hit> scan_max(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq, float threshold, size_t block_size) const
    {
        ++n.scan;
        const DiscreteMatrix<A> dm = pssm.to_discrete();
        std::optional<Hit> best;
        uint8_t best_discrete = dm.scale(threshold);
        const DenseMatrix<uint8_t> &m = seq.matrix();
        const DenseMatrix<float> &w = pssm.matrix();
        const size_t total = m.rows(), seq_rows = total - seq.wrap();
        host::StripedScores<uint8_t> ds(seq.columns());
        const bool port = seq.columns() == 32 && seq.len() >= pssm.len();
        const Aligned<uint8_t> sa(port ? m.ptr() : nullptr, port ? m.rows() * m.stride() : 0);
        const Aligned<uint8_t> pa(port ? dm.matrix().ptr() : nullptr, port ? dm.matrix().rows() * dm.matrix().stride() : 0);
        Aligned<uint8_t> oa(port ? block_size * ds.data.stride() + 64 : 0);
        for (size_t row = 0; row < total; row += block_size) {
            const size_t end = std::min(row + block_size, seq_rows);
            score_u8_block(dm, seq, port ? &sa : nullptr, port ? &pa : nullptr, &oa, row, end, ds);
            if (max(ds).value_or(0) >= best_discrete)
                for (const auto &c : threshold_u8(ds, best_discrete)) {
                    const uint8_t dscore = ds.matrix()(c.row, c.col);
                    if (dscore >= best_discrete) {
                        const size_t index = c.col * seq_rows + row + c.row;
                        if (index + pssm.len() > seq_rows * seq.columns() + seq.wrap())  // the reference indexes past the matrix: panic
                            throw std::runtime_error("Scanner::max: window leaves the matrix");
                        const float score = lmo_score_position(m.ptr(), m.stride(), seq_rows, w.ptr(), w.rows(), w.stride(), index);
                        if (best) {
                            if ((score > best->score) | (score == best->score && index > best->position)) {
                                best = Hit{index, score};
                                best_discrete = dscore;
                            }
                        } else {
                            best = Hit{index, score};
                        }
                    }
                }
        }
        return best;
    }

private:
    std::vector<MatrixCoordinates> threshold_u8(const host::StripedScores<uint8_t> &s, uint8_t t) const { return threshold(s, t); }
};

}  // namespace lightmotif_test
