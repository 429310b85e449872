// pssm.hip -- lm_hip_pssm_*: the device-side tables of a ScoringMatrix (pwm/mod.rs:529-662).
//   d_dense      m x k row-major weights (generic / tiled kernels, re-scoring, Scanner::max)
//   d_table      transposed, padded table of score_c32<M> (+ d_table_pad with leading zero rows for M % 4 != 0)
//   parts        slices of motifs beyond kMaxFastM rows
//   d_image(2)   u16 images of the discrete prefilter scans (score_prefilter.hpp, score_prefilter2.hpp)
#include <algorithm>
#include <cmath>
#include <new>

#include "score_prefilter2.hpp"

namespace lm {

// Builds the LDS image of score_c32_prefilter<M>: [u16 layout EVEN | u16 layout ODD]
// and the affine map discrete ~ (score - offset) / factor.  Follows the
// idea of DiscreteMatrix (pwm/mod.rs:665-696: per-row offsets, one global factor,
// weights rounded UP) on 16 bits.  Returns false when no sound prefilter exists.
static bool build_prefilter(lm_hip_pssm &p, std::vector<unsigned> *image, std::vector<unsigned> *image2,
                            std::vector<unsigned> *image2_drop = nullptr, std::vector<unsigned> *image2_multi = nullptr)
{
    const int m = (int)p.m, k = (int)p.k;
    if (m < 1)
        return false;
    const int mp = prefilter_mp(m, lds_wide(k)), shift = mp - m;
    std::vector<double> off(m), top(m);
    double offset = 0, range = 0, abs_sum = 0;
    for (int j = 0; j < m; ++j) {
        double lo = INFINITY, hi = -INFINITY, amax = 0;
        for (int s = 0; s < k; ++s) {
            const float x = p.host[(size_t)j * k + s];
            if (x != x || x == INFINITY)
                return false;             // NaN / +inf: score semantics the bound cannot cover
            if (x == -INFINITY)
                continue;                 // stands for the row minimum (over-estimate)
            lo = std::min(lo, (double)x);
            hi = std::max(hi, (double)x);
            amax = std::max(amax, std::fabs((double)x));
        }
        if (lo == INFINITY)
            return false;                 // a row of -inf only: every score is -inf
        off[j] = lo;
        top[j] = hi;
        offset += lo;
        range += hi - lo;
        abs_sum += amax;
    }
    if (!(range > 0))
        return false;
    const double factor = range / (double)kPrefilterTop;
    // discrete weights d'[0..mp): leading zero rows pad the motif (even length; a multiple of 4 for wide alphabets)
    std::vector<unsigned> d((size_t)mp * k, 0);
    for (int j = 0; j < m; ++j)
        for (int s = 0; s < k; ++s) {
            const float x = p.host[(size_t)j * k + s];
            const double v = (x == -INFINITY) ? 0.0 : ((double)x - off[j]) / factor;
            unsigned q = (unsigned)std::ceil(v);
            if ((double)q < v + 1e-9)     // guard the ceil against representation error
                q += 1;
            d[(size_t)(j + shift) * k + s] = q;
        }
    image->assign((size_t)prefilter_image_dw(m, k), 0u);
    prefilter_pack_image(d.data(), m, k, image->data());
    // pair-symbol table of score_c32_prefilter2<M> (DNA only): row (a, b) holds
    // E[e] = d[e-1][a] + d[e][b] over the motif padded to an ODD length M' by a leading
    // zero row; dword m = (lo E[2m+1], hi E[2m]).  Same weights, same sums, same bound.
    image2->clear();
    if (k == 5 || k == 21) {  // DNA: 25 pair rows; protein: 441
        image2->assign((size_t)prefilter2_image_dw(m, k), 0u);
        prefilter2_pack_image(d.data() + (size_t)shift * k, m, image2->data(), k);
    }
    // the same table in the layout of the batch's multi-motif passes (lm_hip_pssm::d_image2_multi)
    if (image2_multi) {
        image2_multi->clear();
        if (k == 5 && m <= kMaxFastM) {
            image2_multi->assign((size_t)prefilter2_image_dw(m, kDnaMulti), 0u);
            prefilter2_pack_image(d.data() + (size_t)shift * k, m, image2_multi->data(), kDnaMulti);
        }
    }
    // the same table without the motif's last row, for lengths whose padding wastes a read (lm_hip_pssm::d_image2_drop);
    // what the last row can add at most goes into the bound
    if (image2_drop) {
        image2_drop->clear();
        p.drop_dmax = 0;
        if (k == 5 && m >= 20 && m % 4 == 0) {  // (M = 12, 16: the shorter ring of M - 1 rows costs more than the read it saves: 188 -> 219, 183 -> 192 us per Gbp)
            image2_drop->assign((size_t)prefilter2_image_dw(m - 1, k), 0u);
            prefilter2_pack_image(d.data() + (size_t)shift * k, m - 1, image2_drop->data(), k);
            for (int sy = 0; sy < k; ++sy)
                p.drop_dmax = std::max(p.drop_dmax, d[(size_t)(m - 1 + shift) * k + sy]);
        }
    }
    p.pre_offset = offset;
    p.pre_factor = factor;
    // |f32 sum - real sum| <= (M-1) * 2^-24 * sum |terms|  (each add rounds to nearest)
    p.pre_emax = (double)m * std::ldexp(1.0, -24) * abs_sum * 1.5;
    return true;
}

}  // namespace lm

using namespace lm;

extern "C" {

// ---- PSSM ---------------------------------------------------------------------------------

int lm_hip_pssm_create(lm_hip_ctx *ctx, const float *pssm, size_t m, size_t stride, size_t k,
                       lm_hip_pssm **out)
{
    if (!ctx || !out || (!pssm && m))
        return fail(LM_HIP_ERR_BAD_ARGS, "pssm_create: null argument");
    *out = nullptr;
    if (k == 0 || k > 256 || stride < k)
        return fail(LM_HIP_ERR_BAD_ARGS, "pssm_create: bad alphabet size %zu / stride %zu", k, stride);
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard guard(ctx->device);
    lm_hip_pssm *p = new (std::nothrow) lm_hip_pssm();
    if (!p)
        return fail(LM_HIP_ERR_OOM, "out of host memory");
    p->device = ctx->device;
    p->m = m;
    p->k = k;
    p->host.resize(m * k);
    for (size_t j = 0; j < m; ++j)
        for (size_t s = 0; s < k; ++s)
            p->host[j * k + s] = pssm[j * stride + s];
    auto cleanup = [&](int st) {
        lm_hip_pssm_destroy(p);
        return st;
    };
    if (m) {
        hipError_t e = hipMalloc(&p->d_dense, m * k * sizeof(float));
        if (e != hipSuccess)
            return cleanup(fail(LM_HIP_ERR_OOM, "hipMalloc(pssm) failed: %s", hipGetErrorString(e)));
        e = hipMemcpyAsync(p->d_dense, p->host.data(), m * k * sizeof(float), hipMemcpyHostToDevice,
                           ctx->stream);
        if (e != hipSuccess)
            return cleanup(fail(LM_HIP_ERR_HIP, "pssm upload failed: %s", hipGetErrorString(e)));
        if (m <= (size_t)kMaxFastM) {
            // transposed, padded table of score_c32<M>: table[s * ts + j] = pssm[j][s]
            // (K > 16: rows of 2 * odd dwords for the 8-byte reads of the WIDE kernels, see table_stride)
            p->ts = (size_t)table_stride((int)m, lds_wide((int)k));
            std::vector<float> table(k * p->ts, 0.0f);
            for (size_t s = 0; s < k; ++s)
                for (size_t j = 0; j < m; ++j)
                    table[s * p->ts + j] = p->host[j * k + s];
            e = hipMalloc(&p->d_table, table.size() * sizeof(float));
            if (e != hipSuccess)
                return cleanup(fail(LM_HIP_ERR_OOM, "hipMalloc(table) failed: %s", hipGetErrorString(e)));
            e = hipMemcpyAsync(p->d_table, table.data(), table.size() * sizeof(float),
                               hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(ctx->stream);  // `table` dies with this scope
            if (e != hipSuccess)
                return cleanup(fail(LM_HIP_ERR_HIP, "table upload failed: %s", hipGetErrorString(e)));
            // lengths that are no multiple of 4: a second table with leading all-zero rows (see
            // lm_hip_pssm::d_table_pad); 33..35 stay as they are (36 rows cost more than the byte loads)
            if (m % 4 != 0 && (m + 3) / 4 * 4 <= 32) {
                const size_t mp = (m + 3) / 4 * 4, lead = mp - m, tsp = (size_t)table_stride((int)mp, lds_wide((int)k));
                std::vector<float> padded(k * tsp, 0.0f);
                for (size_t s = 0; s < k; ++s)
                    for (size_t j = 0; j < m; ++j)
                        padded[s * tsp + lead + j] = p->host[j * k + s];
                e = hipMalloc(&p->d_table_pad, padded.size() * sizeof(float));
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_OOM, "hipMalloc(table) failed: %s", hipGetErrorString(e)));
                e = hipMemcpyAsync(p->d_table_pad, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice,
                                   ctx->stream);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(ctx->stream);  // `padded` dies with this scope
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_HIP, "table upload failed: %s", hipGetErrorString(e)));
                p->lead = lead;
            }
            // discrete prefilter image (score_prefilter.hpp); absent when the matrix has
            // NaN / +inf entries or no spread -- the exact f32 fused kernel is used then
            std::vector<unsigned> image;
            std::vector<unsigned> image2, image2_drop, image2_multi;
            if (build_prefilter(*p, &image, &image2, &image2_drop, &image2_multi)) {
                e = hipMalloc(&p->d_image, image.size() * sizeof(unsigned));
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_OOM, "hipMalloc(prefilter) failed: %s", hipGetErrorString(e)));
                e = hipMemcpyAsync(p->d_image, image.data(), image.size() * sizeof(unsigned),
                                   hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_HIP, "prefilter upload failed: %s", hipGetErrorString(e)));
                if (!image2.empty()) {  // DNA: pair-symbol table (score_prefilter2.hpp)
                    e = hipMalloc(&p->d_image2, image2.size() * sizeof(unsigned));
                    if (e == hipSuccess)
                        e = hipMemcpyAsync(p->d_image2, image2.data(), image2.size() * sizeof(unsigned),
                                           hipMemcpyHostToDevice, ctx->stream);
                    if (e == hipSuccess)
                        e = hipStreamSynchronize(ctx->stream);
                    if (e != hipSuccess)
                        return cleanup(fail(LM_HIP_ERR_HIP, "pair prefilter upload failed: %s",
                                            hipGetErrorString(e)));
                }
                if (!image2_multi.empty()) {
                    e = hipMalloc(&p->d_image2_multi, image2_multi.size() * sizeof(unsigned));
                    if (e == hipSuccess)
                        e = hipMemcpyAsync(p->d_image2_multi, image2_multi.data(), image2_multi.size() * sizeof(unsigned),
                                           hipMemcpyHostToDevice, ctx->stream);
                    if (e == hipSuccess)
                        e = hipStreamSynchronize(ctx->stream);
                    if (e != hipSuccess)
                        return cleanup(fail(LM_HIP_ERR_HIP, "pair prefilter upload failed: %s", hipGetErrorString(e)));
                }
                if (!image2_drop.empty()) {
                    e = hipMalloc(&p->d_image2_drop, image2_drop.size() * sizeof(unsigned));
                    if (e == hipSuccess)
                        e = hipMemcpyAsync(p->d_image2_drop, image2_drop.data(), image2_drop.size() * sizeof(unsigned),
                                           hipMemcpyHostToDevice, ctx->stream);
                    if (e == hipSuccess)
                        e = hipStreamSynchronize(ctx->stream);
                    if (e != hipSuccess)
                        return cleanup(fail(LM_HIP_ERR_HIP, "pair prefilter upload failed: %s", hipGetErrorString(e)));
                }
                p->has_prefilter = true;
            }
        }
        if (m > (size_t)kMaxFastM && k <= 64) {
            // long motifs: slices of <= kMaxLongM rows (multiples of 4 rows: dword symbol loads).  Up to
            // kMaxLongM that is ONE slice -- a single pass of the long kernel family (score_long_inst.hip);
            // beyond, the first slice is stored and the others continue in place (MODE_CONTINUE)
            // (65 ... kMaxStoreM rows: ONE slice as well, padded to a multiple of 8 -- the store-only kernels of
            //  score_xlong_inst.hip; the context option "xlong_store" = 0 keeps the slices for A/B runs)
            const bool xlong = m > (size_t)kMaxLongM && m <= (size_t)kMaxStoreM && ctx->xlong_store;
            const size_t nparts = xlong ? 1 : (m + kMaxLongM - 1) / kMaxLongM;
            const size_t len = xlong ? m : std::min<size_t>(((m + nparts - 1) / nparts + 3) / 4 * 4, (size_t)kMaxLongM);
            for (size_t off = 0; off < m; off += len) {
                lm_hip_pssm::Part part;
                part.off = off;
                const size_t real = std::min(len, m - off);
                const size_t unit = xlong ? 8 : 4;
                part.lead = (unit - real % unit) % unit;  // the last slice: leading zero rows up to a multiple of 4 (8)
                part.m = real + part.lead;
                part.ts = (size_t)table_stride((int)part.m, lds_wide((int)k));
                std::vector<float> table(k * part.ts, 0.0f);
                for (size_t s = 0; s < k; ++s)
                    for (size_t j = 0; j < real; ++j)
                        table[s * part.ts + part.lead + j] = p->host[(off + j) * k + s];
                e = hipMalloc(&part.d_table, table.size() * sizeof(float));
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_OOM, "hipMalloc(table) failed: %s", hipGetErrorString(e)));
                p->parts.push_back(part);  // owned from here on (freed by lm_hip_pssm_destroy)
                e = hipMemcpyAsync(part.d_table, table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice,
                                   ctx->stream);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(ctx->stream);  // `table` dies with this scope
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_HIP, "table upload failed: %s", hipGetErrorString(e)));
            }
        }
        if (m > (size_t)kMaxFastM && m <= (size_t)kMaxPairM && k == 5) {
            // 36 < M <= 128, DNA: the pair-symbol prefilter table, so that the fused threshold / argmax scans of these
            // lengths flag candidates like the shorter ones do (the one-symbol u16 scan ends at kMaxFastM)
            std::vector<unsigned> image, image2;
            if (build_prefilter(*p, &image, &image2) && !image2.empty()) {
                e = hipMalloc(&p->d_image2, image2.size() * sizeof(unsigned));
                if (e == hipSuccess)
                    e = hipMemcpyAsync(p->d_image2, image2.data(), image2.size() * sizeof(unsigned), hipMemcpyHostToDevice,
                                       ctx->stream);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess)
                    return cleanup(fail(LM_HIP_ERR_HIP, "pair prefilter upload failed: %s", hipGetErrorString(e)));
                p->has_prefilter = true;
            }
        }
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess)
            return cleanup(fail(LM_HIP_ERR_HIP, "pssm upload failed: %s", hipGetErrorString(e)));
    }
    *out = p;
    return LM_HIP_OK;
}

int lm_hip_pssm_reverse_complement(lm_hip_ctx *ctx, const lm_hip_pssm *pssm, lm_hip_pssm **out)
{
    if (!ctx || !pssm || !out)
        return fail(LM_HIP_ERR_BAD_ARGS, "pssm_reverse_complement: null argument");
    *out = nullptr;
    if (pssm->k != 5)
        return fail(LM_HIP_ERR_BAD_ARGS, "pssm_reverse_complement: only DNA matrices (K = 5) have a complement");
    static const int comp[5] = {2, 3, 0, 1, 4};  // A C T G N -> T G A C N
    const size_t m = pssm->m, k = pssm->k;
    std::vector<float> rc(m * k);
    for (size_t i = 0; i < m; ++i)  // pwm/mod.rs:570-574
        for (size_t s = 0; s < k; ++s)
            rc[i * k + s] = pssm->host[(m - 1 - i) * k + comp[s]];
    return lm_hip_pssm_create(ctx, rc.data(), m, k, k, out);
}

int lm_hip_pssm_destroy(lm_hip_pssm *p)
{
    if (!p)
        return LM_HIP_OK;
    DeviceGuard guard(p->device);
    if (p->d_dense)
        (void)hipFree(p->d_dense);
    if (p->d_table)
        (void)hipFree(p->d_table);
    if (p->d_table_pad)
        (void)hipFree(p->d_table_pad);
    for (auto &part : p->parts)
        if (part.d_table)
            (void)hipFree(part.d_table);
    if (p->d_image)
        (void)hipFree(p->d_image);
    if (p->d_image2)
        (void)hipFree(p->d_image2);
    if (p->d_image2_drop)
        (void)hipFree(p->d_image2_drop);
    if (p->d_image2_multi)
        (void)hipFree(p->d_image2_multi);
    delete p;
    return LM_HIP_OK;
}

size_t lm_hip_pssm_len(const lm_hip_pssm *p) { return p ? p->m : 0; }

}  // extern "C"
