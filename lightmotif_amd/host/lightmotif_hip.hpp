// lightmotif_hip.hpp -- header-only C++ host mirror of lightmotif's scoring interface
// for the MI355X back-end, written over the C ABI of include/lightmotif_hip.h.
//
// The reference's host language is Rust; this image has no Rust toolchain, so the
// host side above the C ABI is C++ (the reference is compiled code).  Names, argument
// meaning and error behaviour follow the reference so tests/cpp/test_dna.cpp reads
// like lightmotif/tests/dna.rs:
//
//   Pipeline<A>::hip()                      pli/mod.rs:401-407 (Result -> throws UnsupportedBackend)
//   pli.encode / pli.stripe                 pli/mod.rs:34-67, 164-201
//   pli.score_rows_into / score_into / score pli/mod.rs:69-130
//   pli.argmax / max / threshold            pli/mod.rs:132-161, 203-222
//   StripedSequence::configure{,_wrap}      seq.rs:362-381
//   StripedScores::{offset,len,at,unstripe,argmax,max,threshold}  scores.rs:148-213, 246-288
//   CountMatrix -> FrequencyMatrix -> WeightMatrix -> ScoringMatrix  pwm/mod.rs:209-258, 376-431, 505-526
//
// Sequence and score data are device-resident; every scoring operation runs in the
// HIP kernels.  Misuse that panics in the reference throws std::runtime_error here.
#pragma once

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "lightmotif_hip.h"

namespace lightmotif {

struct UnsupportedBackend : std::runtime_error {  // err.rs:34
    using std::runtime_error::runtime_error;
};
struct InvalidSymbol : std::invalid_argument {  // err.rs:10
    char symbol;
    explicit InvalidSymbol(char c) : std::invalid_argument(std::string("invalid symbol: ") + c), symbol(c) {}
};
struct InvalidData : std::invalid_argument {  // err.rs:22
    InvalidData() : std::invalid_argument("invalid data") {}
};

inline void check(int status)
{
    if (status == LM_HIP_OK)
        return;
    const std::string msg = lm_hip_last_error();
    if (status == LM_HIP_ERR_NO_DEVICE)
        throw UnsupportedBackend(msg);
    throw std::runtime_error(msg);  // the reference panics
}

// ---- alphabets (abc.rs:91-135, 193-256) ---------------------------------------------------

struct Dna {
    static constexpr size_t K = 5;
    static constexpr char code = 'D';
    static const char *symbols() { return "ACTGN"; }
};
struct Protein {
    static constexpr size_t K = 21;
    static constexpr char code = 'P';
    static const char *symbols() { return "ACDEFGHIKLMNPQRSTVWYX"; }
};

struct MatrixCoordinates {  // dense.rs:28-39
    size_t row = 0, col = 0;
    bool operator==(const MatrixCoordinates &o) const { return row == o.row && col == o.col; }
};

// ---- host-side dense matrix with the reference's padded rows (dense.rs:43-55) -----------------

template <class T>
class DenseMatrix {
public:
    DenseMatrix(size_t rows, size_t cols)
        : rows_(rows), cols_(cols), stride_(lm_hip_stride(cols, sizeof(T))), data_(rows * stride_, T()) {}
    size_t rows() const { return rows_; }
    size_t columns() const { return cols_; }
    size_t stride() const { return stride_; }  // dense.rs:126-128
    T *operator[](size_t r) { return data_.data() + r * stride_; }
    const T *operator[](size_t r) const { return data_.data() + r * stride_; }
    T &operator()(size_t r, size_t c) { return data_[r * stride_ + c]; }
    const T &operator()(size_t r, size_t c) const { return data_[r * stride_ + c]; }
    T *ptr() { return data_.data(); }
    const T *ptr() const { return data_.data(); }

private:
    size_t rows_, cols_, stride_;
    std::vector<T> data_;
};

// ---- sequences --------------------------------------------------------------------------------

template <class A>
class EncodedSequence {  // seq.rs:83-98
public:
    explicit EncodedSequence(std::vector<uint8_t> d) : data(std::move(d)) {}
    // EncodedSequence::encode (seq.rs:110-113) -> Err(InvalidSymbol)
    static EncodedSequence encode(const std::string &text) { return from(text, false); }
    // seq.rs:122-129
    static EncodedSequence encode_lossy(const std::string &text) { return from(text, true); }
    size_t len() const { return data.size(); }
    std::vector<uint8_t> data;

private:
    static EncodedSequence from(const std::string &text, bool lossy)
    {
        int8_t lut[256];
        std::memset(lut, -1, sizeof lut);
        for (size_t i = 0; i < A::K; ++i)
            lut[(uint8_t)A::symbols()[i]] = (int8_t)i;
        std::vector<uint8_t> out(text.size());
        for (size_t i = 0; i < text.size(); ++i) {
            int8_t s = lut[(uint8_t)text[i]];
            if (s < 0) {
                if (!lossy)
                    throw InvalidSymbol(text[i]);
                s = (int8_t)(A::K - 1);
            }
            out[i] = (uint8_t)s;
        }
        return EncodedSequence(std::move(out));
    }
};

template <class A>
class Pipeline;
template <class A>
class ScoringMatrix;

struct CtxHandle {
    lm_hip_ctx *ctx = nullptr;
    explicit CtxHandle(int device) { check(lm_hip_ctx_create(device, &ctx)); }
    ~CtxHandle() { lm_hip_ctx_destroy(ctx); }
    CtxHandle(const CtxHandle &) = delete;
    CtxHandle &operator=(const CtxHandle &) = delete;
};

template <class A>
class StripedSequence {  // seq.rs:288-294, device-resident
public:
    StripedSequence(std::shared_ptr<CtxHandle> c, lm_hip_seq *h) : ctx_(std::move(c)), h_(h) {}
    ~StripedSequence() { lm_hip_seq_destroy(h_); }
    StripedSequence(StripedSequence &&o) noexcept : ctx_(std::move(o.ctx_)), h_(o.h_) { o.h_ = nullptr; }
    StripedSequence(const StripedSequence &) = delete;

    size_t len() const { return info().length; }
    size_t wrap() const { return info().wrap; }
    size_t rows() const { return info().rows; }  // sequence rows, without the wrap rows
    size_t columns() const { return info().cols; }
    // Reconfigure for a motif (seq.rs:362-366)
    void configure(const ScoringMatrix<A> &motif)
    {
        if (motif.len() > 0)
            configure_wrap(motif.len() - 1);
    }
    // seq.rs:369-381
    void configure_wrap(size_t m) { check(lm_hip_seq_configure_wrap(ctx_->ctx, h_, m)); }
    // Host copy of the (rows + wrap) x stride matrix.
    DenseMatrix<uint8_t> matrix() const
    {
        const Info i = info();
        DenseMatrix<uint8_t> m(i.rows + i.wrap, i.cols);
        check(lm_hip_seq_download(ctx_->ctx, h_, m.ptr()));
        return m;
    }
    lm_hip_seq *handle() const { return h_; }

private:
    struct Info { size_t length, wrap, rows, stride, cols; };
    Info info() const
    {
        Info i{};
        check(lm_hip_seq_info(h_, &i.length, &i.wrap, &i.rows, &i.stride, &i.cols, nullptr));
        return i;
    }
    std::shared_ptr<CtxHandle> ctx_;
    lm_hip_seq *h_;
};

// ---- matrices (pwm/mod.rs) ---------------------------------------------------------------------

template <class A>
inline std::vector<float> uniform_background()  // abc.rs:473-487
{
    std::vector<float> bg(A::K, 1.0f / (float)(A::K - 1));
    bg[A::K - 1] = 0.0f;
    return bg;
}

// pwm/mod.rs:754-791: u8 weights that over-estimate the f32 scores
template <class A>
class DiscreteMatrix {
public:
    DiscreteMatrix(DenseMatrix<uint8_t> d, float f, std::vector<float> offs, float off)
        : data(std::move(d)), factor(f), offsets(std::move(offs)), offset(off) {}
    size_t len() const { return data.rows(); }
    const DenseMatrix<uint8_t> &matrix() const { return data; }
    // pwm/mod.rs:777-779: rounds DOWN (an f32 threshold -> a u8 threshold); Rust's saturating `as u8`
    uint8_t scale(float score) const { return to_u8(std::floor((score - offset) / factor)); }
    // pwm/mod.rs:783-785
    float unscale(uint8_t score) const { return (float)score * factor + offset; }
    static uint8_t to_u8(float x) { return x != x ? 0 : x <= 0.0f ? 0 : x >= 255.0f ? 255 : (uint8_t)x; }
    DenseMatrix<uint8_t> data;
    float factor;
    std::vector<float> offsets;
    float offset;
};

template <class A>
class ScoringMatrix {  // pwm/mod.rs:561-564
public:
    ScoringMatrix(std::vector<float> bg, DenseMatrix<float> d) : background(std::move(bg)), data(std::move(d)) {}
    ~ScoringMatrix()
    {
        if (dev_)
            lm_hip_pssm_destroy(dev_);
    }
    ScoringMatrix(ScoringMatrix &&o) noexcept
        : background(std::move(o.background)), data(std::move(o.data)), dev_(o.dev_), dev_ctx_(o.dev_ctx_)
    {
        o.dev_ = nullptr;
    }
    ScoringMatrix(const ScoringMatrix &o) : background(o.background), data(o.data) {}
    size_t len() const { return data.rows(); }
    const DenseMatrix<float> &matrix() const { return data; }
    // pwm/mod.rs:604-615: sum over the rows of the best weight among the K-1 real symbols
    float max_score() const
    {
        float total = 0.0f;
        for (size_t i = 0; i < data.rows(); ++i) {
            float best = data(i, 0);
            for (size_t j = 1; j + 1 < A::K; ++j)
                best = data(i, j) > best ? data(i, j) : best;
            total += best;
        }
        return total;
    }
    // pwm/mod.rs:566-577 (ComplementableAlphabet = Dna): rows reversed, A<->T and C<->G swapped
    ScoringMatrix reverse_complement() const
    {
        static_assert(A::K == 5, "only the DNA alphabet has a complement");
        static const size_t comp[5] = {2, 3, 0, 1, 4};
        DenseMatrix<float> out(data.rows(), A::K);
        for (size_t i = 0; i < data.rows(); ++i)
            for (size_t s = 0; s < A::K; ++s)
                out(i, s) = data(data.rows() - 1 - i, comp[s]);
        return ScoringMatrix(background, std::move(out));
    }
    // pwm/mod.rs:665-696
    DiscreteMatrix<A> to_discrete() const
    {
        const float top = max_score();
        std::vector<float> offsets(data.rows());
        float offset = 0.0f;
        for (size_t i = 0; i < data.rows(); ++i) {
            float lo = std::numeric_limits<float>::infinity();
            for (size_t j = 0; j + 1 < A::K; ++j) {
                const float x = std::isinf(data(i, j)) ? -top : data(i, j);
                lo = x < lo ? x : lo;
            }
            offsets[i] = lo;
            offset += lo;
        }
        const float factor = (top - offset) / 255.0f;
        DenseMatrix<uint8_t> d(data.rows(), A::K);
        for (size_t i = 0; i < data.rows(); ++i)
            for (size_t j = 0; j < A::K; ++j)
                d(i, j) = DiscreteMatrix<A>::to_u8(std::ceil((data(i, j) - offsets[i]) / factor));
        return DiscreteMatrix<A>(std::move(d), factor, std::move(offsets), offset);
    }
    lm_hip_pssm *device(lm_hip_ctx *ctx) const
    {
        if (!dev_ || dev_ctx_ != ctx) {
            if (dev_)
                lm_hip_pssm_destroy(dev_);
            check(lm_hip_pssm_create(ctx, data.ptr(), data.rows(), data.stride(), A::K, &dev_));
            dev_ctx_ = ctx;
        }
        return dev_;
    }
    std::vector<float> background;
    DenseMatrix<float> data;

private:
    mutable lm_hip_pssm *dev_ = nullptr;
    mutable lm_hip_ctx *dev_ctx_ = nullptr;
};

template <class A>
class WeightMatrix {  // pwm/mod.rs:450-456
public:
    WeightMatrix(std::vector<float> bg, DenseMatrix<float> d) : background(std::move(bg)), data(std::move(d)) {}
    // pwm/mod.rs:505-526 (base 2)
    ScoringMatrix<A> to_scoring() const
    {
        DenseMatrix<float> out = data;
        for (size_t i = 0; i < out.rows(); ++i)
            for (size_t j = 0; j < A::K; ++j)
                out(i, j) = std::log2(out(i, j));
        return ScoringMatrix<A>(background, std::move(out));
    }
    std::vector<float> background;
    DenseMatrix<float> data;
};

template <class A>
class FrequencyMatrix {  // pwm/mod.rs:340-431
public:
    explicit FrequencyMatrix(DenseMatrix<float> d) : data(std::move(d)) {}
    // pwm/mod.rs:376-392; None background = uniform
    WeightMatrix<A> to_weight() const
    {
        const std::vector<float> bg = uniform_background<A>();
        DenseMatrix<float> w(data.rows(), A::K);
        for (size_t i = 0; i < data.rows(); ++i)
            for (size_t j = 0; j < A::K; ++j)
                w(i, j) = bg[j] == 0.0f ? 0.0f : data(i, j) / bg[j];
        return WeightMatrix<A>(bg, std::move(w));
    }
    // pwm/mod.rs:415-430
    ScoringMatrix<A> to_scoring() const
    {
        const std::vector<float> bg = uniform_background<A>();
        DenseMatrix<float> s(data.rows(), A::K);
        for (size_t i = 0; i < data.rows(); ++i)
            for (size_t j = 0; j < A::K; ++j)
                s(i, j) = bg[j] == 0.0f ? -std::numeric_limits<float>::infinity()
                                        : std::log2(data(i, j) / bg[j]);
        return ScoringMatrix<A>(bg, std::move(s));
    }
    DenseMatrix<float> data;
};

template <class A>
class CountMatrix {  // pwm/mod.rs:178-258
public:
    explicit CountMatrix(DenseMatrix<uint32_t> d) : data(std::move(d)) {}
    // pwm/mod.rs:209-237
    static CountMatrix from_sequences(const std::vector<EncodedSequence<A>> &seqs)
    {
        const size_t m = seqs.empty() ? 0 : seqs[0].len();
        DenseMatrix<uint32_t> d(m, A::K);
        for (const auto &s : seqs) {
            if (s.len() != m)
                throw InvalidData();
            for (size_t i = 0; i < m; ++i)
                d(i, s.data[i]) += 1;
        }
        return CountMatrix(std::move(d));
    }
    // pwm/mod.rs:240-258 with a scalar pseudocount (abc.rs:558-573)
    FrequencyMatrix<A> to_freq(float pseudo) const
    {
        DenseMatrix<float> p(data.rows(), A::K);
        for (size_t i = 0; i < data.rows(); ++i) {
            float sum = 0.0f;
            for (size_t j = 0; j < A::K; ++j) {
                p(i, j) = (float)data(i, j) + (j != A::K - 1 ? pseudo : 0.0f);
                sum = sum + p(i, j);
            }
            for (size_t j = 0; j < A::K; ++j)
                p(i, j) = p(i, j) / sum;
        }
        return FrequencyMatrix<A>(std::move(p));
    }
    DenseMatrix<uint32_t> data;
};

// ---- scores ---------------------------------------------------------------------------------------

class StripedScores {  // scores.rs:102-107, device-resident
public:
    StripedScores(std::shared_ptr<CtxHandle> c, lm_hip_scores *h) : ctx_(std::move(c)), h_(h) {}
    ~StripedScores() { lm_hip_scores_destroy(h_); }
    StripedScores(StripedScores &&o) noexcept : ctx_(std::move(o.ctx_)), h_(o.h_) { o.h_ = nullptr; }
    StripedScores(const StripedScores &) = delete;

    size_t max_index() const { return info().max_index; }        // scores.rs:124-127
    bool is_empty() const { return info().rows == 0; }           // scores.rs:130-133
    size_t rows() const { return info().rows; }
    DenseMatrix<float> matrix() const                            // host copy of rows x stride
    {
        const Info i = info();
        DenseMatrix<float> m(i.rows, i.cols);
        check(lm_hip_scores_download(ctx_->ctx, h_, m.ptr()));
        return m;
    }
    float at(size_t index) const                                 // Index<usize>, scores.rs:246-254
    {
        const Info i = info();
        if (index >= i.rows * i.cols)
            throw std::out_of_range("score index out of range");
        std::vector<float> row(i.stride);
        check(lm_hip_scores_download_rows(ctx_->ctx, h_, index % i.rows, index % i.rows + 1, row.data()));
        return row[index / i.rows];
    }
    size_t offset(MatrixCoordinates mc) const { return mc.col * info().rows + mc.row; }  // scores.rs:155-157
    size_t len() const                                           // scores.rs:274-279
    {
        const Info i = info();
        return std::min(i.max_index, i.rows * i.cols);
    }
    std::vector<float> unstripe() const                          // scores.rs:167-170
    {
        const DenseMatrix<float> m = matrix();
        const size_t n = len(), r = m.rows();
        std::vector<float> out(n);
        for (size_t i = 0; i < n; ++i)
            out[i] = m(i % r, i / r);
        return out;
    }
    // scores.rs:181-213 (the wrappers map coordinates through offset())
    std::optional<size_t> argmax() const
    {
        int found = 0;
        lm_hip_coords best{};
        check(lm_hip_argmax(ctx_->ctx, h_, &found, &best, nullptr));
        if (!found)
            return std::nullopt;
        return offset({best.row, best.col});
    }
    std::optional<float> max() const
    {
        int found = 0;
        float v = 0;
        check(lm_hip_max(ctx_->ctx, h_, &found, &v));  // Maximum::max, pli/mod.rs:158-160
        return found ? std::optional<float>(v) : std::nullopt;
    }
    std::vector<size_t> threshold(float t) const
    {
        lm_hip_coords *c = nullptr;
        size_t n = 0;
        check(lm_hip_threshold(ctx_->ctx, h_, t, &c, &n));
        std::vector<size_t> out(n);
        const size_t r = info().rows;
        for (size_t i = 0; i < n; ++i)
            out[i] = c[i].col * r + c[i].row;
        lm_hip_free(c);
        return out;
    }
    lm_hip_scores *handle() const { return h_; }
    // This StripedScores is one row shard of a larger matrix: `holds_first_row == false` skips
    // Maximum::argmax's "scores[0][0] is NaN -> (0,0)" rule (pli/mod.rs:142-146) on it.
    void set_first_cell_rule(bool holds_first_row) const
    {
        check(lm_hip_scores_set_first_cell_rule(h_, holds_first_row ? 1 : 0));
    }

private:
    struct Info { size_t rows, stride, cols, max_index; };
    Info info() const
    {
        Info i{};
        check(lm_hip_scores_info(h_, &i.rows, &i.stride, &i.cols, &i.max_index, nullptr));
        return i;
    }
    std::shared_ptr<CtxHandle> ctx_;
    lm_hip_scores *h_;
};

// ---- the literal drop-in: host matrices in, host matrices out ------------------------------------------------
//
// The C++ twin of the Rust shim in INTEGRATION.md 2-4b.  `host::StripedSequence` / `host::StripedScores` are the
// reference's own structs (seq.rs:288-294, scores.rs:102-107) with their data in ordinary host memory, striped and wrapped
// by the Generic loops (pli/mod.rs:178-200, seq.rs:362-381) -- what a Rust caller holds when `Dispatch::Hip` is asked to
// score.  `Hip` binds exactly the host-pointer entry points that shim binds, with the same resize-then-call order
// (avx2.rs:839-844); `HipDispatch<Cpu>` is the `Dispatch::Hip { cpu }` variant: the arm table with its size policy.

namespace host {

template <class A>
class StripedSequence {
public:
    StripedSequence(DenseMatrix<uint8_t> d, size_t length, size_t wrap, size_t cols)
        : data(std::move(d)), length_(length), wrap_(wrap), cols_(cols) {}
    // Stripe::stripe_into, Generic (pli/mod.rs:178-200): position i -> [i % R][i / R], the tail = default symbol
    static StripedSequence stripe(const EncodedSequence<A> &seq, size_t cols = 32)
    {
        const size_t len = seq.len(), rows = (len + cols - 1) / cols;
        DenseMatrix<uint8_t> m(rows, cols);
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c)
                m(r, c) = (uint8_t)(A::K - 1);
        for (size_t i = 0; i < len; ++i)
            m(i % rows, i / rows) = seq.data[i];
        return StripedSequence(std::move(m), len, 0, cols);
    }
    // seq.rs:362-381: rows() + motif_len - 1 rows; wrap row i = row i shifted one column left, last column = default
    void configure_wrap(size_t motif_len)
    {
        if (motif_len <= wrap_)
            return;
        const size_t rows = data.rows() - wrap_;
        DenseMatrix<uint8_t> m(rows + motif_len, cols_);
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols_; ++c)
                m(r, c) = data(r, c);
        for (size_t i = 0; i < motif_len; ++i) {
            for (size_t c = 0; c + 1 < cols_; ++c)
                m(rows + i, c) = m(i, c + 1);
            m(rows + i, cols_ - 1) = (uint8_t)(A::K - 1);
        }
        data = std::move(m);
        wrap_ = motif_len;
    }
    void configure(const ScoringMatrix<A> &pssm)
    {
        if (pssm.len() > 0)
            configure_wrap(pssm.len() - 1);
    }
    size_t len() const { return length_; }
    size_t wrap() const { return wrap_; }
    size_t columns() const { return cols_; }
    const DenseMatrix<uint8_t> &matrix() const { return data; }
    DenseMatrix<uint8_t> data;

private:
    size_t length_, wrap_, cols_;
};

template <class T>
class StripedScores {
public:
    explicit StripedScores(size_t cols = 32) : data(0, cols), cols_(cols) {}
    void resize(size_t rows, size_t max_index)  // scores.rs:148-152
    {
        if (rows != data.rows())
            data = DenseMatrix<T>(rows, cols_);
        max_index_ = max_index;
    }
    size_t max_index() const { return max_index_; }
    size_t len() const { return std::min(max_index_, data.rows() * cols_); }            // scores.rs:274-279
    size_t offset(MatrixCoordinates mc) const { return mc.col * data.rows() + mc.row; }  // scores.rs:155-157
    std::vector<T> unstripe() const                                                      // scores.rs:167-170
    {
        std::vector<T> out(len());
        for (size_t i = 0; i < out.size(); ++i)
            out[i] = data(i % data.rows(), i / data.rows());
        return out;
    }
    const DenseMatrix<T> &matrix() const { return data; }
    DenseMatrix<T> data;

private:
    size_t cols_, max_index_ = 0;
};

}  // namespace host

// The back-end marker and its associated functions (`impl Hip { ... }` of INTEGRATION.md 2, cf. `Avx2` in
// platform/avx2.rs): the raw host-pointer bindings, no policy.
struct Hip {
    // Hip::available(): `Pipeline::hip()` mirrors `Pipeline::avx2()` (pli/mod.rs:401-407)
    static bool available()
    {
        int n = 0;
        return lm_hip_device_count(&n) == LM_HIP_OK && n > 0;
    }
    // `score_into` followed by `argmax` / `threshold` on the same StripedScores (the reference's bench loop, dna.rs:81-116):
    // the reduction takes the copy the score call left on the device instead of uploading the matrix again.  A shim can
    // switch it on for good: between the two calls the scores sit behind a `&StripedScores` nobody writes through.
    static void reuse_scores(bool on) { check(lm_hip_host_reuse_scores(on ? 1 : 0)); }
    // Score<f32, A, C>::score_rows_into for Dispatch::Hip (dispatch.rs:90-106; contract of avx2.rs:889-904)
    template <class A>
    static void score_rows_into(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq, size_t row_begin,
                                size_t row_end, host::StripedScores<float> &scores)
    {
        const DenseMatrix<float> &w = pssm.matrix();
        if (seq.len() < w.rows() || row_begin >= row_end) {  // avx2.rs:839-842
            scores.resize(0, 0);
            return;
        }
        scores.resize(row_end - row_begin, seq.len() + 1 - w.rows());
        size_t out_rows = 0, max_index = 0;
        const DenseMatrix<uint8_t> &m = seq.matrix();
        check(lm_hip_score_f32(m.ptr(), m.rows(), m.stride(), seq.columns(), seq.wrap(), seq.len(), w.ptr(), w.rows(),
                               w.stride(), A::K, row_begin, row_end, scores.data.ptr(), scores.data.stride(), &out_rows,
                               &max_index));
    }
    template <class A>
    static host::StripedScores<float> score(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq)
    {
        host::StripedScores<float> scores(seq.columns());
        score_rows_into(pssm, seq, 0, seq.matrix().rows() - seq.wrap(), scores);  // pli/mod.rs:115-116
        return scores;
    }
    // Score<u8, A, C> with a DiscreteMatrix (pli/mod.rs:437-476; avx2.rs:921-931)
    template <class A>
    static void score_rows_into(const DiscreteMatrix<A> &dm, const host::StripedSequence<A> &seq, size_t row_begin,
                                size_t row_end, host::StripedScores<uint8_t> &scores, bool saturate = true)
    {
        const DenseMatrix<uint8_t> &w = dm.matrix();
        if (seq.len() < w.rows() || row_begin >= row_end) {
            scores.resize(0, 0);
            return;
        }
        scores.resize(row_end - row_begin, seq.len() + 1 - w.rows());
        size_t out_rows = 0, max_index = 0;
        const DenseMatrix<uint8_t> &m = seq.matrix();
        check(lm_hip_score_u8_host(m.ptr(), m.rows(), m.stride(), seq.columns(), seq.wrap(), seq.len(), w.ptr(), w.rows(),
                                   w.stride(), A::K, row_begin, row_end, saturate ? 1 : 0, scores.data.ptr(),
                                   scores.data.stride(), &out_rows, &max_index));
    }
    // Maximum<f32, C> (dispatch.rs:160-177): the Generic rule
    static std::optional<MatrixCoordinates> argmax(const host::StripedScores<float> &scores)
    {
        const DenseMatrix<float> &d = scores.matrix();
        if (d.rows() == 0)
            return std::nullopt;
        int found = 0;
        lm_hip_coords best{};
        check(lm_hip_argmax_f32(d.ptr(), d.rows(), d.stride(), d.columns(), &found, &best, nullptr));
        return found ? std::optional<MatrixCoordinates>(MatrixCoordinates{best.row, best.col}) : std::nullopt;
    }
    static std::optional<float> max(const host::StripedScores<float> &scores)  // pli/mod.rs:158-160
    {
        const DenseMatrix<float> &d = scores.matrix();
        if (d.rows() == 0)
            return std::nullopt;
        int found = 0;
        float v = 0;
        check(lm_hip_max_f32(d.ptr(), d.rows(), d.stride(), d.columns(), &found, &v));
        return found ? std::optional<float>(v) : std::nullopt;
    }
    // Threshold<f32, C> (dispatch.rs:205; pli/mod.rs:210-221): row-major push order
    static std::vector<MatrixCoordinates> threshold(const host::StripedScores<float> &scores, float t)
    {
        const DenseMatrix<float> &d = scores.matrix();
        std::vector<MatrixCoordinates> out;
        if (d.rows() == 0)
            return out;
        lm_hip_coords *c = nullptr;
        size_t n = 0;
        check(lm_hip_threshold_f32(d.ptr(), d.rows(), d.stride(), d.columns(), t, &c, &n));
        out.reserve(n);
        for (size_t i = 0; i < n; ++i)
            out.push_back(MatrixCoordinates{c[i].row, c[i].col});
        lm_hip_free(c);
        return out;
    }
    // Scanner (scan.rs:96-250) on a host sequence: every hit (score >= t, position + M <= L), ascending position
    struct Hit {
        size_t position;
        float score;
    };
    template <class A>
    static std::vector<Hit> scan(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq, float threshold)
    {
        const DenseMatrix<float> &w = pssm.matrix();
        const DenseMatrix<uint8_t> &m = seq.matrix();
        lm_hip_hit *h = nullptr;
        size_t n = 0;
        check(lm_hip_scan_f32_host(m.ptr(), m.rows(), m.stride(), seq.columns(), seq.wrap(), seq.len(), w.ptr(), w.rows(),
                                   w.stride(), A::K, threshold, &h, &n));
        std::vector<Hit> out(n);
        for (size_t i = 0; i < n; ++i)
            out[i] = Hit{h[i].position, h[i].score};
        lm_hip_free(h);
        return out;
    }
    // Scanner::max as the reference walks it (scan.rs:200-249) on a fresh scanner
    template <class A>
    static std::optional<Hit> scan_max(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq, float threshold,
                                       bool saturate = true)
    {
        const DiscreteMatrix<A> dm = pssm.to_discrete();
        const DenseMatrix<float> &w = pssm.matrix();
        const DenseMatrix<uint8_t> &m = seq.matrix();
        int found = 0;
        lm_hip_hit best{0, 0.0f};
        check(lm_hip_scan_max_f32_host(m.ptr(), m.rows(), m.stride(), seq.columns(), seq.wrap(), seq.len(), w.ptr(), w.rows(),
                                       w.stride(), A::K, dm.data.ptr(), dm.data.stride(), saturate ? 1 : 0, dm.scale(threshold), 0,
                                       0, 0.0f, 0, &found, &best));
        return found ? std::optional<Hit>(Hit{best.position, best.score}) : std::nullopt;
    }
};

// ---- enum Dispatch { ..., Hip { cpu } } ---------------------------------------------------------------
//
// The complete arm table of lightmotif/src/pli/dispatch.rs:58-207 + Scanner (scan.rs:166-249) for a `Hip` variant
// (INTEGRATION.md 3).  The variant CARRIES the CPU tier `Pipeline::dispatch()` would have picked without a GPU
// (pli/mod.rs:269-308) and hands a call to it where the GPU cannot win: operations that only move bytes (Encode, Stripe),
// the Scanner-internal u8 reductions, and any call below the measured crossover (lm_hip_host_crossover).  In the Rust
// shim the tier is the reference's own Avx2 / Sse2 / Neon / Generic marker; in this C++ twin it is a template parameter
// with the trait methods' names -- this header ships NO CPU scoring code (there is no CPU fallback in the product: without a
// device `Hip::available()` is false and the variant is never constructed).  `Cpu` provides:
//
//   static constexpr bool saturating_u8;                       // Score<u8> saturates (avx2.rs:336) or wraps (Generic)
//   encode_into<A>(text, dst) / stripe_into<A>(encoded, striped)
//   score_rows_into(pssm | dm, seq, rb, re, scores)            // f32: every tier is bit-identical to Generic
//   argmax / max (f32): MUST be the Generic rule (pli/mod.rs:135-160) -- Avx2::argmax_f32 differs on ties and
//                      Avx2::max_f32 seeds with 0.0 (avx2.rs:351-441), and a variant has to give ONE answer at every size
//   threshold (f32), argmax / max / threshold (u8)
//   scan / scan_max: the reference's Scanner loop over the tier (scan.rs:169-249)
enum class Route { Cpu, Gpu };

struct HipPolicy {
    // cells (rows x columns of the call) from which the site goes to the GPU; SIZE_MAX = never, 0 = always
    size_t cells(lm_hip_host_op op, size_t m, size_t k) const
    {
        if (forced[op] != kUnset)
            return forced[op];
        size_t c = 0;
        check(lm_hip_host_crossover((int)op, m, k, &c));
        return c;
    }
    void force(lm_hip_host_op op, size_t cells) { forced[op] = cells; }   // tests / embedders with their own numbers
    void reset(lm_hip_host_op op) { forced[op] = kUnset; }
    static constexpr size_t kUnset = (size_t)-2;
    std::array<size_t, 9> forced{kUnset, kUnset, kUnset, kUnset, kUnset, kUnset, kUnset, kUnset, kUnset};
};

template <class Cpu>
class HipDispatch {
public:
    using Hit = Hip::Hit;
    explicit HipDispatch(Cpu tier = Cpu()) : cpu(std::move(tier))
    {
        if (!Hip::available())  // Pipeline::hip() -> Err(UnsupportedBackend) (pli/mod.rs:401-407): never a silent CPU-only variant
            throw UnsupportedBackend("no gfx950 device");
    }
    Cpu cpu;
    HipPolicy policy;
    mutable Route last_route = Route::Cpu;   // where the last call went
    mutable size_t calls_cpu = 0, calls_gpu = 0;

    // Encode<A> (dispatch.rs:58-77): the CPU tier, at every size -- a LUT pass over bytes a core streams faster than the
    // link carries them (1 B up + 1 B down).  (Host text that should END UP on the device: lm_hip_seq_from_ascii.)
    template <class A>
    void encode_into(const std::string &text, std::vector<uint8_t> &dst) const
    {
        went(Route::Cpu);
        cpu.template encode_into<A>(text, dst);
    }
    // Stripe<A, C> (dispatch.rs:139-153): the CPU tier, same argument (lm_hip_seq_from_encoded for resident sequences)
    template <class A>
    void stripe_into(const EncodedSequence<A> &seq, host::StripedSequence<A> &striped) const
    {
        went(Route::Cpu);
        cpu.template stripe_into<A>(seq, striped);
    }
    // Score<f32, A, C> (dispatch.rs:79-108)
    template <class A>
    void score_rows_into(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq, size_t row_begin, size_t row_end,
                         host::StripedScores<float> &scores) const
    {
        const size_t cells = row_end > row_begin ? (row_end - row_begin) * seq.columns() : 0;
        if (cells < policy.cells(LM_HIP_OP_SCORE_F32, pssm.len(), A::K)) {
            went(Route::Cpu);
            cpu.score_rows_into(pssm, seq, row_begin, row_end, scores);
        } else {
            went(Route::Gpu);
            Hip::score_rows_into(pssm, seq, row_begin, row_end, scores);
        }
    }
    template <class A>
    host::StripedScores<float> score(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq) const
    {
        host::StripedScores<float> scores(seq.columns());
        score_rows_into(pssm, seq, 0, seq.matrix().rows() - seq.wrap(), scores);  // pli/mod.rs:115-116
        return scores;
    }
    // Score<u8, Dna, C> (dispatch.rs:110-137): the tier's own overflow rule travels with the call
    template <class A>
    void score_rows_into(const DiscreteMatrix<A> &dm, const host::StripedSequence<A> &seq, size_t row_begin, size_t row_end,
                         host::StripedScores<uint8_t> &scores) const
    {
        const size_t cells = row_end > row_begin ? (row_end - row_begin) * seq.columns() : 0;
        if (cells < policy.cells(LM_HIP_OP_SCORE_U8, dm.len(), A::K)) {
            went(Route::Cpu);
            cpu.score_rows_into(dm, seq, row_begin, row_end, scores);
        } else {
            went(Route::Gpu);
            Hip::score_rows_into(dm, seq, row_begin, row_end, scores, Cpu::saturating_u8);
        }
    }
    // Maximum<f32, C> (dispatch.rs:155-179)
    std::optional<MatrixCoordinates> argmax(const host::StripedScores<float> &scores) const
    {
        if (cells_of(scores) < policy.cells(LM_HIP_OP_MAXIMUM_F32, 0, 0)) {
            went(Route::Cpu);
            return cpu.argmax(scores);
        }
        went(Route::Gpu);
        return Hip::argmax(scores);
    }
    std::optional<float> max(const host::StripedScores<float> &scores) const
    {
        if (cells_of(scores) < policy.cells(LM_HIP_OP_MAXIMUM_F32, 0, 0)) {
            went(Route::Cpu);
            return cpu.max(scores);
        }
        went(Route::Gpu);
        return Hip::max(scores);
    }
    // Threshold<f32, C> (dispatch.rs:205: today an empty impl = the default body, pli/mod.rs:210-221)
    std::vector<MatrixCoordinates> threshold(const host::StripedScores<float> &scores, float t) const
    {
        if (cells_of(scores) < policy.cells(LM_HIP_OP_THRESHOLD_F32, 0, 0)) {
            went(Route::Cpu);
            return cpu.threshold(scores, t);
        }
        went(Route::Gpu);
        return Hip::threshold(scores, t);
    }
    // Maximum<u8, C> / Threshold<u8, C> (dispatch.rs:181-207): the CPU tier.  Their only caller is the Scanner's block loop
    // (scan.rs:181-184), which this variant replaces as a whole below; a stray call on a host u8 matrix is one pass over
    // 1 B per cell.
    std::optional<MatrixCoordinates> argmax(const host::StripedScores<uint8_t> &scores) const
    {
        went(Route::Cpu);
        return cpu.argmax(scores);
    }
    std::optional<uint8_t> max(const host::StripedScores<uint8_t> &scores) const
    {
        went(Route::Cpu);
        return cpu.max(scores);
    }
    std::vector<MatrixCoordinates> threshold(const host::StripedScores<uint8_t> &scores, uint8_t t) const
    {
        went(Route::Cpu);
        return cpu.threshold(scores, t);
    }
    // Scanner (scan.rs:166 hard-codes Pipeline<A, Dispatch>): with `Hip` the iterator is SPECIALISED -- the whole scan is
    // one lm_hip_scan_f32_host (upload + prefilter scan + exact re-scoring), the hits are yielded in the reference's
    // own order (blocks ascending, last row-major hit of a block first: scan.rs:184-198) -- instead of driving 256-row
    // blocks through the Score<u8> arm.  Small sequences stay on the tier's Scanner.
    template <class A>
    std::vector<Hit> scan(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq, float threshold,
                          size_t block_size = 256) const
    {
        const size_t rows = seq.matrix().rows() - seq.wrap();
        if (rows * seq.columns() < policy.cells(LM_HIP_OP_SCAN, pssm.len(), A::K)) {
            went(Route::Cpu);
            return cpu.scan(pssm, seq, threshold, block_size);
        }
        went(Route::Gpu);
        std::vector<Hit> hits = Hip::scan(pssm, seq, threshold);
        if (rows == 0 || block_size == 0)
            return hits;
        std::sort(hits.begin(), hits.end(), [=](const Hit &a, const Hit &b) {
            const size_t ra = a.position % rows, rb = b.position % rows;
            if (ra / block_size != rb / block_size)
                return ra / block_size < rb / block_size;
            if (ra != rb)
                return ra > rb;
            return a.position / rows > b.position / rows;
        });
        return hits;
    }
    // Scanner::max (scan.rs:200-249), quirks included, on a fresh scanner
    template <class A>
    std::optional<Hit> scan_max(const ScoringMatrix<A> &pssm, const host::StripedSequence<A> &seq, float threshold,
                                size_t block_size = 256) const
    {
        const size_t rows = seq.matrix().rows() - seq.wrap();
        if (rows * seq.columns() < policy.cells(LM_HIP_OP_SCAN, pssm.len(), A::K)) {
            went(Route::Cpu);
            return cpu.scan_max(pssm, seq, threshold, block_size);
        }
        went(Route::Gpu);
        return Hip::scan_max(pssm, seq, threshold, Cpu::saturating_u8);
    }

    // The reference picks its back-end per HOST, at run time (pli/mod.rs:269-308); the size policy likewise.  The library
    // measures the GPU side of its cost model on this host and link (lm_hip_host_calibrate: call latency and per-cell link
    // cost of every site that can leave the tier); THIS tier is timed here -- the library never sees it -- on a synthetic
    // 1 Mi-position DNA sequence: Score<f32> and Score<u8> at two motif lengths (a cost per cell and motif row), argmax and
    // threshold on the f32 scores, the Scanner loop.  ~budget_ms in total, once per process for the GPU side.  Pins made
    // with policy.force() stay on top.
    struct Calibration {
        size_t score_f32_m16 = 0, score_u8_m16 = 0, maximum_f32 = 0, threshold_f32 = 0, scan_m16 = 0;  // crossovers, cells
    };
    Calibration calibrate(double budget_ms = 40.0)
    {
        check(lm_hip_host_calibrate(budget_ms * 0.5, 0));
        const size_t n = (size_t)1 << 20;
        std::string text(n, 'A');
        uint64_t x = 0x5EED0007u;
        for (size_t i = 0; i < n; ++i) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            text[i] = "ACGT"[(x >> 33) & 3u];
        }
        auto motif = [&](size_t m) {
            std::vector<EncodedSequence<Dna>> sites;
            for (int s = 0; s < 4; ++s)
                sites.push_back(EncodedSequence<Dna>::encode(text.substr(1000 * (size_t)(s + 1), m)));
            return CountMatrix<Dna>::from_sequences(sites).to_freq(0.1f).to_scoring();
        };
        auto timed_ns = [&](auto &&f, double share_ms) {  // best of a few runs inside its share of the budget, per call
            double best = 1e300;
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double, std::milli>(share_ms);
            int runs = 0;
            do {
                const auto a = std::chrono::steady_clock::now();
                f();
                best = std::min(best, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - a).count());
            } while (++runs < 5 && std::chrono::steady_clock::now() < t_end);
            return best;
        };
        const auto p8 = motif(8), p24 = motif(24);
        host::StripedSequence<Dna> seq = host::StripedSequence<Dna>::stripe(EncodedSequence<Dna>::encode(text), 32);
        seq.configure(p24);
        const size_t rows = seq.matrix().rows() - seq.wrap();
        const double cells = (double)rows * 32, share = budget_ms * 0.5 / 7.0;
        host::StripedScores<float> sc(32);
        const double f8 = timed_ns([&] { cpu.score_rows_into(p8, seq, 0, rows, sc); }, share) / cells;
        const double f24 = timed_ns([&] { cpu.score_rows_into(p24, seq, 0, rows, sc); }, share) / cells;
        const double f_row = std::max((f24 - f8) / 16.0, 0.0);
        check(lm_hip_host_set_cpu_cost(LM_HIP_OP_SCORE_F32, std::max(f8 - 8.0 * f_row, 0.0), f_row));
        const auto d8 = p8.to_discrete(), d24 = p24.to_discrete();
        host::StripedScores<uint8_t> su(32);
        const double u8 = timed_ns([&] { cpu.score_rows_into(d8, seq, 0, rows, su); }, share) / cells;
        const double u24 = timed_ns([&] { cpu.score_rows_into(d24, seq, 0, rows, su); }, share) / cells;
        const double u_row = std::max((u24 - u8) / 16.0, 0.0);
        check(lm_hip_host_set_cpu_cost(LM_HIP_OP_SCORE_U8, std::max(u8 - 8.0 * u_row, 0.0), u_row));
        check(lm_hip_host_set_cpu_cost(LM_HIP_OP_MAXIMUM_F32, timed_ns([&] { (void)cpu.argmax(sc); }, share) / cells, 0.0));
        check(lm_hip_host_set_cpu_cost(LM_HIP_OP_THRESHOLD_F32, timed_ns([&] { (void)cpu.threshold(sc, 1e30f); }, share) / cells, 0.0));
        // the Scanner's loop: the u8 score of every block + its reductions (a cost per cell and motif row like Score<u8>,
        // plus one pass per cell)
        const double s24 = timed_ns([&] { (void)cpu.scan(p24, seq, 1e30f, 256); }, share) / cells;
        check(lm_hip_host_set_cpu_cost(LM_HIP_OP_SCAN, std::max(s24 - 24.0 * u_row, 0.0), u_row));
        Calibration c;
        check(lm_hip_host_crossover(LM_HIP_OP_SCORE_F32, 16, 5, &c.score_f32_m16));
        check(lm_hip_host_crossover(LM_HIP_OP_SCORE_U8, 16, 5, &c.score_u8_m16));
        check(lm_hip_host_crossover(LM_HIP_OP_MAXIMUM_F32, 0, 0, &c.maximum_f32));
        check(lm_hip_host_crossover(LM_HIP_OP_THRESHOLD_F32, 0, 0, &c.threshold_f32));
        check(lm_hip_host_crossover(LM_HIP_OP_SCAN, 16, 5, &c.scan_m16));
        return c;
    }

private:
    template <class T>
    static size_t cells_of(const host::StripedScores<T> &s) { return s.matrix().rows() * s.matrix().columns(); }
    void went(Route r) const
    {
        last_route = r;
        ++(r == Route::Cpu ? calls_cpu : calls_gpu);
    }
};

// ---- Pipeline<A, Hip> --------------------------------------------------------------------------------

template <class A>
class Pipeline {
public:
    // Pipeline::avx2() -> Result<Self, UnsupportedBackend> (pli/mod.rs:401-407)
    static Pipeline hip(int device = 0) { return Pipeline(std::make_shared<CtxHandle>(device)); }
    // the library context behind this pipeline, for direct C-ABI calls
    lm_hip_ctx *context() const { return ctx_->ctx; }

    // the best cell and its score; the cells with score >= t in row-major order
    struct Best {
        MatrixCoordinates cell;
        float score;
    };
    struct Cells {
        std::vector<MatrixCoordinates> coords;
        std::vector<float> scores;
    };

    // Encode (pli/mod.rs:47-50)
    EncodedSequence<A> encode(const std::string &text) const { return EncodedSequence<A>::encode(text); }

    // Stripe::stripe (pli/mod.rs:166-175), on the device
    StripedSequence<A> stripe(const EncodedSequence<A> &seq, size_t columns = 32) const
    {
        lm_hip_seq *h = nullptr;
        check(lm_hip_seq_from_encoded(ctx_->ctx, seq.data.data(), seq.len(), columns, A::K, &h));
        return StripedSequence<A>(ctx_, h);
    }

    StripedScores empty_scores(size_t columns = 32) const  // StripedScores::empty() (scores.rs:118-121)
    {
        lm_hip_scores *h = nullptr;
        check(lm_hip_scores_create(ctx_->ctx, columns, &h));
        return StripedScores(ctx_, h);
    }

    // Score::score_rows_into (pli/mod.rs:72-106); rows = [begin, end)
    void score_rows_into(const ScoringMatrix<A> &pssm, const StripedSequence<A> &seq, size_t begin,
                         size_t end, StripedScores &scores) const
    {
        check(lm_hip_score_rows_into(ctx_->ctx, pssm.device(ctx_->ctx), seq.handle(), begin, end,
                                     scores.handle()));
    }
    // Score::score_into (pli/mod.rs:109-117)
    void score_into(const ScoringMatrix<A> &pssm, const StripedSequence<A> &seq, StripedScores &scores) const
    {
        check(lm_hip_score_into(ctx_->ctx, pssm.device(ctx_->ctx), seq.handle(), scores.handle()));
    }
    // Score::score (pli/mod.rs:120-129)
    StripedScores score(const ScoringMatrix<A> &pssm, const StripedSequence<A> &seq) const
    {
        StripedScores s = empty_scores(seq.columns());
        score_into(pssm, seq, s);
        return s;
    }
    // Score<u8, A, C>::score with a DiscreteMatrix (pli/mod.rs:72-129): the u8 StripedScores the
    // Scanner computes first (scan.rs:174-178), returned on the host.  `saturate`: the SIMD
    // back-ends' saturating adds (avx2.rs:336) or Generic's wrapping `+=`.
    struct DiscreteScores {
        DenseMatrix<uint8_t> data;
        size_t max_index;
        // scores.rs:246-254: the score of position i
        uint8_t at(size_t i) const { return data(i % data.rows(), i / data.rows()); }
        size_t len() const { return max_index; }
    };
    DiscreteScores score(const DiscreteMatrix<A> &dm, const StripedSequence<A> &seq, bool saturate = true) const
    {
        size_t rows = 0, max_index = 0;
        DenseMatrix<uint8_t> out(seq.rows(), seq.columns());
        check(lm_hip_score_u8(ctx_->ctx, dm.data.ptr(), dm.data.rows(), dm.data.stride(), A::K, seq.handle(), 0,
                              seq.rows(), saturate ? 1 : 0, out.ptr(), out.stride(), &rows, &max_index));
        if (rows == 0)
            return DiscreteScores{DenseMatrix<uint8_t>(0, seq.columns()), 0};
        return DiscreteScores{std::move(out), max_index};
    }
    // Maximum::argmax / max (pli/mod.rs:135-160)
    std::optional<MatrixCoordinates> argmax(const StripedScores &scores) const
    {
        int found = 0;
        lm_hip_coords best{};
        check(lm_hip_argmax(ctx_->ctx, scores.handle(), &found, &best, nullptr));
        if (!found)
            return std::nullopt;
        return MatrixCoordinates{best.row, best.col};
    }
    std::optional<float> max(const StripedScores &scores) const { return scores.max(); }
    // Threshold::threshold (pli/mod.rs:210-221)
    std::vector<MatrixCoordinates> threshold(const StripedScores &scores, float t) const
    {
        lm_hip_coords *c = nullptr;
        size_t n = 0;
        check(lm_hip_threshold(ctx_->ctx, scores.handle(), t, &c, &n));
        std::vector<MatrixCoordinates> out(n);
        for (size_t i = 0; i < n; ++i)
            out[i] = {c[i].row, c[i].col};
        lm_hip_free(c);
        return out;
    }

    // Scanner::new(pssm, seq).threshold(t).collect() sorted by position (scan.rs:96-250)
    struct Hit {
        size_t position;
        float score;
    };
    std::vector<Hit> scan(const ScoringMatrix<A> &pssm, const StripedSequence<A> &seq, float threshold) const
    {
        lm_hip_hit *h = nullptr;
        size_t n = 0;
        check(lm_hip_scan_f32(ctx_->ctx, pssm.device(ctx_->ctx), seq.handle(), threshold, &h, &n));
        std::vector<Hit> out(n);
        for (size_t i = 0; i < n; ++i)
            out[i] = Hit{h[i].position, h[i].score};
        lm_hip_free(h);
        return out;
    }

    // The order in which the reference's Scanner YIELDS those hits: blocks of `block_size` rows
    // ascending, inside a block the last cell in row-major order first (hits are pushed in
    // Threshold order and popped from the end of the vector, scan.rs:184-198).  `rows` =
    // seq.rows() (position = col * rows + row, scan.rs:185).
    static std::vector<Hit> scan_order(std::vector<Hit> hits, size_t rows, size_t block_size = 256)
    {
        if (rows == 0 || block_size == 0)
            return hits;
        std::sort(hits.begin(), hits.end(), [=](const Hit &a, const Hit &b) {
            const size_t ra = a.position % rows, rb = b.position % rows;
            if (ra / block_size != rb / block_size)
                return ra / block_size < rb / block_size;
            if (ra != rb)
                return ra > rb;
            return a.position / rows > b.position / rows;
        });
        return hits;
    }

    // Scanner::new(pssm, seq).threshold(t).max() EXACTLY as the reference computes it (scan.rs:200-249): the u8
    // DiscreteMatrix scores steer the walk (level = u8 score of the current best, first candidate taken below the
    // threshold, no position + M <= L test; see lm_hip_scan_max_f32).  The walk runs on the device.  `saturate`:
    // the u8 adds of the x86-64 `dispatch` pipeline (avx2.rs:336); false = Generic's wrapping adds.
    std::optional<Hit> scan_max(const ScoringMatrix<A> &pssm, const StripedSequence<A> &seq, float threshold,
                                bool saturate = true) const
    {
        const DiscreteMatrix<A> dm = pssm.to_discrete();
        int found = 0;
        lm_hip_hit best{0, 0.0f};
        check(lm_hip_scan_max_f32(ctx_->ctx, pssm.device(ctx_->ctx), seq.handle(), dm.data.ptr(), dm.data.stride(),
                                  saturate ? 1 : 0, dm.scale(threshold), 0, 0, 0.0f, 0, &found, &best));
        return found ? std::optional<Hit>(Hit{best.position, best.score}) : std::nullopt;
    }

    // NOT the reference's max(): the best of an already collected hit list -- greater score wins, equal scores go
    // to the greater position (the comparison of scan.rs:237 without the u8 steering)
    static std::optional<Hit> scan_max_valid(const std::vector<Hit> &hits)
    {
        std::optional<Hit> best;
        for (const Hit &h : hits)
            if (!best || h.score > best->score || (h.score == best->score && h.position > best->position))
                best = h;
        return best;
    }

    // score + argmax / score + threshold without materialising the scores (what the bench
    // harness dna.rs:104-107 and the Scanner need); same results as score() followed by
    // argmax() / threshold(), cells in row-major order
    std::optional<Best> score_argmax(const ScoringMatrix<A> &pssm, const StripedSequence<A> &seq) const
    {
        const SeqView v = view(seq);
        int found = 0;
        lm_hip_coords best{};
        float value = 0;
        check(lm_hip_score_argmax_f32_dptr(ctx_->ctx, pssm.device(ctx_->ctx), v.data, v.rows + v.wrap, v.stride,
                                           v.cols, v.wrap, v.length, 0, v.rows, &found, &best, &value));
        if (!found)
            return std::nullopt;
        return Best{{best.row, best.col}, value};
    }
    Cells score_threshold(const ScoringMatrix<A> &pssm, const StripedSequence<A> &seq, float t) const
    {
        const SeqView v = view(seq);
        lm_hip_coords *c = nullptr;
        float *vals = nullptr;
        size_t n = 0;
        check(lm_hip_score_threshold_f32_dptr(ctx_->ctx, pssm.device(ctx_->ctx), v.data, v.rows + v.wrap,
                                              v.stride, v.cols, v.wrap, v.length, 0, v.rows, t, &c, &vals, &n));
        Cells out;
        out.coords.resize(n);
        out.scores.assign(vals, vals + n);
        for (size_t k = 0; k < n; ++k)
            out.coords[k] = {c[k].row, c[k].col};
        lm_hip_free(c);
        lm_hip_free(vals);
        return out;
    }
    // tuning knobs of the context (results never depend on them)
    void set_prefilter(bool on) const { check(lm_hip_ctx_set_prefilter(ctx_->ctx, on ? 1 : 0)); }
    void set_track_argmax(bool on) const { check(lm_hip_ctx_set_track_argmax(ctx_->ctx, on ? 1 : 0)); }

    // Encode + Stripe of raw text in one go, on the device (pli/mod.rs:47-66 + 166-175):
    // the text is uploaded once, encoded and striped by the kernels.  Throws InvalidSymbol.
    StripedSequence<A> stripe_text(const std::string &text, size_t columns = 32, bool lossy = false) const
    {
        lm_hip_seq *h = nullptr;
        size_t bad = 0;
        const int st = lm_hip_seq_from_ascii(ctx_->ctx, A::code, reinterpret_cast<const uint8_t *>(text.data()),
                                             text.size(), columns, lossy ? 1 : 0, &h, &bad);
        if (st == LM_HIP_ERR_INVALID_SYMBOL)
            throw InvalidSymbol(text[bad]);
        check(st);
        return StripedSequence<A>(ctx_, h);
    }

    // A DNA sequence held 4 bases per byte (base i in bits 2 * (i % 4).. of byte i / 4, A0 C1 T2 G3) with its N runs
    // {start, size} like a .2bit file's N blocks: a quarter of the bytes over PCIe, unpacked straight into the
    // striped matrix (lm_hip_seq_from_2bit).  `pack_2bit` builds both from an EncodedSequence.
    struct Packed2bit {
        std::vector<uint8_t> packed;
        std::vector<uint64_t> n_runs;  // pairs
        size_t length = 0;
    };
    static Packed2bit pack_2bit(const EncodedSequence<A> &seq)
    {
        Packed2bit p;
        p.length = seq.len();
        p.packed.assign((p.length + 3) / 4, 0);
        for (size_t i = 0; i < p.length; ++i) {
            const uint8_t s = seq.data[i];
            if (s > 3) {
                if (!p.n_runs.empty() && p.n_runs[p.n_runs.size() - 2] + p.n_runs.back() == i)
                    ++p.n_runs.back();
                else {
                    p.n_runs.push_back(i);
                    p.n_runs.push_back(1);
                }
            } else {
                p.packed[i / 4] |= (uint8_t)(s << (2 * (i % 4)));
            }
        }
        return p;
    }
    StripedSequence<A> stripe_2bit(const Packed2bit &p, size_t columns = 32) const
    {
        lm_hip_seq *h = nullptr;
        check(lm_hip_seq_from_2bit(ctx_->ctx, p.packed.data(), nullptr, p.n_runs.data(), p.n_runs.size() / 2, p.length,
                                   columns, &h));
        return StripedSequence<A>(ctx_, h);
    }

    // Many motifs over one resident sequence (the CLI's fan-out, lightmotif-cli main.rs:554-561):
    // per motif the best cell and its score, or nullopt when the sequence is shorter than the motif.
    std::vector<std::optional<Best>> scan_argmax_batch(const std::vector<const ScoringMatrix<A> *> &pssms,
                                                       const StripedSequence<A> &seq) const
    {
        const size_t n = pssms.size();
        std::vector<const lm_hip_pssm *> handles(n);
        for (size_t i = 0; i < n; ++i)
            handles[i] = pssms[i]->device(ctx_->ctx);
        std::vector<int> found(n);
        std::vector<lm_hip_coords> best(n);
        std::vector<float> value(n);
        check(lm_hip_scan_argmax_batch(ctx_->ctx, handles.data(), n, seq.handle(), found.data(), best.data(),
                                       value.data()));
        std::vector<std::optional<Best>> out(n);
        for (size_t i = 0; i < n; ++i)
            if (found[i])
                out[i] = Best{{best[i].row, best[i].col}, value[i]};
        return out;
    }
    // per motif: the cells with score >= thresholds[i] in row-major order, with their scores
    std::vector<Cells> scan_threshold_batch(const std::vector<const ScoringMatrix<A> *> &pssms,
                                            const std::vector<float> &thresholds,
                                            const StripedSequence<A> &seq) const
    {
        const size_t n = pssms.size();
        if (thresholds.size() != n)
            throw std::invalid_argument("one threshold per motif");
        std::vector<const lm_hip_pssm *> handles(n);
        for (size_t i = 0; i < n; ++i)
            handles[i] = pssms[i]->device(ctx_->ctx);
        std::vector<size_t> counts(n);
        lm_hip_coords *c = nullptr;
        float *v = nullptr;
        check(lm_hip_scan_threshold_batch(ctx_->ctx, handles.data(), thresholds.data(), n, seq.handle(),
                                          counts.data(), &c, &v));
        std::vector<Cells> out(n);
        size_t pos = 0;
        for (size_t i = 0; i < n; ++i) {
            out[i].coords.resize(counts[i]);
            out[i].scores.assign(v + pos, v + pos + counts[i]);
            for (size_t k = 0; k < counts[i]; ++k)
                out[i].coords[k] = {c[pos + k].row, c[pos + k].col};
            pos += counts[i];
        }
        lm_hip_free(c);
        lm_hip_free(v);
        return out;
    }

    // ---- row-sharded jobs across the GPUs of a node -----------------------------------
    // Score::score_rows_into takes a row range so that a sequence can be cut (pli/mod.rs:72-78):
    // one process per GPU scores rows [a_g, b_g) + an M-1-row halo; `ShardComm` (below) carries the
    // halo hand-over and the argmax / threshold merge over RCCL, bound directly by the C library.
    lm_hip_ctx *raw() const { return ctx_->ctx; }
    // A StripedSequence over a device matrix that stays the caller's (one row shard, a buffer of
    // the host application): nothing is copied.
    StripedSequence<A> adopt(uint8_t *d_data, size_t rows, size_t wrap, size_t capacity_rows, size_t stride,
                             size_t columns, size_t length) const
    {
        lm_hip_seq *h = nullptr;
        check(lm_hip_seq_adopt_dptr(ctx_->ctx, d_data, rows + wrap, capacity_rows, stride, columns, wrap, length,
                                    A::K, &h));
        return StripedSequence<A>(ctx_, h);
    }

private:
    struct SeqView {
        const uint8_t *data = nullptr;
        size_t length = 0, wrap = 0, rows = 0, stride = 0, cols = 0;
    };
    static SeqView view(const StripedSequence<A> &seq)
    {
        SeqView v;
        check(lm_hip_seq_info(seq.handle(), &v.length, &v.wrap, &v.rows, &v.stride, &v.cols, &v.data));
        return v;
    }
    explicit Pipeline(std::shared_ptr<CtxHandle> c) : ctx_(std::move(c)) {}
    std::shared_ptr<CtxHandle> ctx_;
};

// The merge rule on host arrays (any transport): per-shard results in ascending row order, rows
// global, first-cell rule applied on the shard holding row 0 only -> Maximum::argmax of the whole
// matrix (pli/mod.rs:135-155).
struct ShardBest {
    bool found = false;
    MatrixCoordinates cell{};
    float score = 0.0f;
};
inline ShardBest combine_argmax(const std::vector<ShardBest> &shards)
{
    const size_t n = shards.size();
    std::vector<int> f(n);
    std::vector<lm_hip_coords> b(n);
    std::vector<float> v(n);
    for (size_t i = 0; i < n; ++i) {
        f[i] = shards[i].found ? 1 : 0;
        b[i] = {shards[i].cell.row, shards[i].cell.col};
        v[i] = shards[i].score;
    }
    ShardBest out;
    int fo = 0;
    lm_hip_coords bo{};
    check(lm_hip_combine_argmax(f.data(), b.data(), v.data(), n, &fo, &bo, &out.score));
    out.found = fo != 0;
    out.cell = {bo.row, bo.col};
    return out;
}

// RCCL communicator of a row-sharded job: rank g holds rows [row_offset, row_offset + rows).
// One rank draws `unique_id()` and hands the 128 bytes to the others (MPI_Bcast, a file ...).
template <class A>
class ShardComm {
public:
    static std::array<uint8_t, LM_HIP_COMM_ID_BYTES> unique_id()
    {
        std::array<uint8_t, LM_HIP_COMM_ID_BYTES> id{};
        check(lm_hip_comm_unique_id(id.data()));
        return id;
    }
    ShardComm(const Pipeline<A> &pli, const std::array<uint8_t, LM_HIP_COMM_ID_BYTES> &id, int nranks, int rank)
        : ctx_(pli.raw())
    {
        check(lm_hip_comm_create(ctx_, id.data(), nranks, rank, &h_));
    }
    ~ShardComm() { lm_hip_comm_destroy(h_); }
    ShardComm(const ShardComm &) = delete;
    // rows [rows, rows + halo) of this rank's shard <- the first rows of rank + 1's (the last rank
    // gets rank 0's turned into wrap rows, seq.rs:373-378)
    void exchange_halo(uint8_t *d_shard, size_t rows, size_t stride, size_t columns, size_t halo_rows) const
    {
        check(lm_hip_exchange_halo_dptr(ctx_, h_, d_shard, rows, stride, columns, halo_rows, (uint8_t)(A::K - 1)));
    }
    // StripedScores::argmax of the WHOLE matrix from this rank's resident shard
    ShardBest argmax(const StripedScores &shard, size_t row_offset) const
    {
        ShardBest out;
        int found = 0;
        lm_hip_coords best{};
        check(lm_hip_argmax_sharded(ctx_, h_, shard.handle(), row_offset, &found, &best, &out.score));
        out.found = found != 0;
        out.cell = {best.row, best.col};
        return out;
    }
    // The same in two halves: `argmax_begin` returns at once and the shard may be scored over right
    // away; `argmax_end(ticket)` waits for that merge only (at most two in flight)
    int argmax_begin(const StripedScores &shard, size_t row_offset) const
    {
        int ticket = -1;
        check(lm_hip_argmax_sharded_begin(ctx_, h_, shard.handle(), row_offset, &ticket));
        return ticket;
    }
    ShardBest argmax_end(int ticket) const
    {
        ShardBest out;
        int found = 0;
        lm_hip_coords best{};
        check(lm_hip_argmax_sharded_end(ctx_, h_, ticket, &found, &best, &out.score));
        out.found = found != 0;
        out.cell = {best.row, best.col};
        return out;
    }
    // StripedScores::threshold of the whole matrix: (row, col) in the reference's row-major order
    std::vector<MatrixCoordinates> threshold(const StripedScores &shard, float t, size_t row_offset) const
    {
        lm_hip_coords *mine = nullptr, *all = nullptr;
        size_t n = 0, n_all = 0;
        check(lm_hip_threshold(ctx_, shard.handle(), t, &mine, &n));
        const int st = lm_hip_merge_threshold(ctx_, h_, mine, n, row_offset, &all, &n_all);
        lm_hip_free(mine);
        check(st);
        std::vector<MatrixCoordinates> out(n_all);
        for (size_t i = 0; i < n_all; ++i)
            out[i] = {all[i].row, all[i].col};
        lm_hip_free(all);
        return out;
    }

private:
    lm_hip_ctx *ctx_;
    lm_hip_comm *h_ = nullptr;
};

}  // namespace lightmotif
