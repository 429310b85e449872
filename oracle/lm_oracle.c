/*
 * lm_oracle.c -- CPU restatement of lightmotif's Generic pipeline (see
 * lm_oracle.h).  TEST INFRASTRUCTURE ONLY: never linked into the product.
 *
 * Build: make -C oracle   (gcc -O2 -fno-fast-math; the f32 sums must stay
 * M sequential IEEE adds, so no -ffast-math / -Ofast here).
 */
#include "lm_oracle.h"

#include <math.h>
#include <string.h>

/* dense.rs:43-48: Row is repr(align(32)) on x86-64, so its size is C*sizeof(T)
 * rounded up to 32 bytes; dense.rs:126-128: stride = size_of::<Row>() /
 * size_of::<T>().  (dense.rs:367-391 tests: <u8,32>->32, <u8,16>->32,
 * <u8,33>->64, <f32,5>->8, <f32,21>->24.) */
size_t lmo_stride(size_t cols, size_t elem_size)
{
    size_t bytes = cols * elem_size;
    bytes = (bytes + 31) / 32 * 32;
    return bytes / elem_size;
}

/* abc.rs:166-171 / abc.rs:296-325 */
static int dna_from_ascii(uint8_t c)
{
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'T': return 2;
    case 'G': return 3;
    case 'N': return 4;
    default: return -1;
    }
}

static int protein_from_ascii(uint8_t c)
{
    static const char order[] = "ACDEFGHIKLMNPQRSTVWYX"; /* abc.rs:193-256 */
    for (int i = 0; i < 21; i++)
        if ((uint8_t)order[i] == c)
            return i;
    return -1;
}

size_t lmo_encode(char alphabet, const uint8_t *ascii, size_t len, int lossy,
                  uint8_t *dst)
{
    const int def = (alphabet == 'P') ? 20 : 4;
    for (size_t i = 0; i < len; i++) { /* pli/mod.rs:62-64 */
        int s = (alphabet == 'P') ? protein_from_ascii(ascii[i])
                                  : dna_from_ascii(ascii[i]);
        if (s < 0) {
            if (!lossy)
                return i + 1; /* Err(InvalidSymbol) */
            s = def;          /* seq.rs:126 unwrap_or_default */
        }
        dst[i] = (uint8_t)s;
    }
    return 0;
}

size_t lmo_stripe(const uint8_t *seq, size_t len, size_t cols,
                  uint8_t default_symbol, uint8_t *data, size_t stride)
{
    const size_t rows = (len + cols - 1) / cols; /* pli/mod.rs:182 */
    if (rows == 0)
        return 0;
    /* dense.rs:144-147: `resize_with(rows, Default::default)` -- every element of a
     * fresh row is T::default(), which for Nucleotide / AminoAcid is the LAST symbol
     * (N = 4 / X = 20, abc.rs:113-135, 231-256), not zero.  The bytes between `cols`
     * and `stride` are struct padding of the reference's Row (unspecified there); this
     * restatement and the HIP back-end fill them with the default symbol too, so that
     * every byte of the matrix is a valid symbol index. */
    memset(data, default_symbol, rows * stride);
    for (size_t i = 0; i < len; i++) /* pli/mod.rs:191-193 */
        data[(i % rows) * stride + (i / rows)] = seq[i];
    for (size_t i = len; i < rows * cols; i++) /* pli/mod.rs:194-196 */
        data[(i % rows) * stride + (i / rows)] = default_symbol;
    return rows;
}

size_t lmo_configure_wrap(uint8_t *data, size_t rows, size_t stride,
                          size_t cols, size_t old_wrap, size_t new_wrap,
                          uint8_t default_symbol)
{
    if (new_wrap <= old_wrap) /* seq.rs:370 */
        return old_wrap;
    /* seq.rs:372: resize to rows + m (new rows = default symbol, dense.rs:144-147) */
    memset(data + (rows + old_wrap) * stride, default_symbol,
           (new_wrap - old_wrap) * stride);
    for (size_t i = 0; i < new_wrap; i++) { /* seq.rs:373-378 */
        for (size_t j = 0; j + 1 < cols; j++)
            data[(rows + i) * stride + j] = data[i * stride + j + 1];
        data[(rows + i) * stride + cols - 1] = default_symbol;
    }
    return new_wrap;
}

void lmo_score_rows_f32(const uint8_t *seq, size_t seq_stride, size_t cols,
                        size_t length, const float *pssm, size_t m,
                        size_t pssm_stride, size_t row_begin, size_t row_end,
                        float *out, size_t out_stride, size_t *out_rows,
                        size_t *max_index)
{
    if (length < m || row_begin >= row_end) { /* pli/mod.rs:85-88 */
        *out_rows = 0;
        *max_index = 0;
        return;
    }
    *out_rows = row_end - row_begin;                 /* pli/mod.rs:91 */
    *max_index = (length + 1 > m) ? length + 1 - m : 0;
    for (size_t r = row_begin; r < row_end; r++) {   /* pli/mod.rs:96 */
        for (size_t c = 0; c < cols; c++) {          /* :97 */
            float score = 0.0f;                      /* :98 T::default() */
            for (size_t j = 0; j < m; j++) {         /* :99 */
                const uint8_t sym = seq[(r + j) * seq_stride + c]; /* :100 */
                score = score + pssm[j * pssm_stride + sym];       /* :101 */
            }
            out[(r - row_begin) * out_stride + c] = score;         /* :103 */
        }
    }
}

void lmo_score_rows_u8(const uint8_t *seq, size_t seq_stride, size_t cols,
                       size_t length, const uint8_t *pssm, size_t m,
                       size_t pssm_stride, size_t row_begin, size_t row_end,
                       uint8_t *out, size_t out_stride, size_t *out_rows,
                       size_t *max_index)
{
    if (length < m || row_begin >= row_end) {
        *out_rows = 0;
        *max_index = 0;
        return;
    }
    *out_rows = row_end - row_begin;
    *max_index = (length + 1 > m) ? length + 1 - m : 0;
    for (size_t r = row_begin; r < row_end; r++) {
        for (size_t c = 0; c < cols; c++) {
            uint8_t score = 0;
            for (size_t j = 0; j < m; j++) {
                const uint8_t sym = seq[(r + j) * seq_stride + c];
                score = (uint8_t)(score + pssm[j * pssm_stride + sym]);
            }
            out[(r - row_begin) * out_stride + c] = score;
        }
    }
}

int lmo_argmax_f32(const float *scores, size_t rows, size_t stride,
                   size_t cols, size_t *row, size_t *col)
{
    if (rows == 0) /* pli/mod.rs:136-138 */
        return 0;
    size_t best_row = 0, best_col = 0; /* :140-141 */
    float best = scores[0];            /* :142 scores[0] == data[0][0] */
    for (size_t i = 0; i < rows; i++) {       /* :144 */
        for (size_t j = 0; j < cols; j++) {   /* :145 */
            const float x = scores[i * stride + j];
            if (x >= best) { /* :146 -- NaN never replaces; ties go later */
                best_row = i;
                best_col = j;
                best = x;
            }
        }
    }
    *row = best_row;
    *col = best_col;
    return 1;
}

int lmo_max_f32(const float *scores, size_t rows, size_t stride, size_t cols,
                float *value)
{
    size_t r, c;
    if (!lmo_argmax_f32(scores, rows, stride, cols, &r, &c))
        return 0;
    *value = scores[r * stride + c]; /* pli/mod.rs:159 */
    return 1;
}

size_t lmo_threshold_f32(const float *scores, size_t rows, size_t stride,
                         size_t cols, float t, size_t *rc, size_t cap)
{
    size_t n = 0;
    for (size_t i = 0; i < rows; i++) {     /* pli/mod.rs:212 */
        for (size_t j = 0; j < cols; j++) { /* :213 */
            if (scores[i * stride + j] >= t) { /* :215 */
                if (n < cap) {
                    rc[2 * n] = i;
                    rc[2 * n + 1] = j;
                }
                n++;
            }
        }
    }
    return n;
}

size_t lmo_offset(size_t rows, size_t row, size_t col)
{
    return col * rows + row; /* scores.rs:156 */
}

size_t lmo_unstripe_f32(const float *scores, size_t rows, size_t stride,
                        size_t cols, size_t max_index, float *dst)
{
    size_t end = rows * cols; /* scores.rs:274-276 */
    if (max_index < end)
        end = max_index;
    for (size_t i = 0; i < end; i++) { /* scores.rs:283-287 */
        const size_t col = i / rows, row = i % rows;
        dst[i] = scores[row * stride + col];
    }
    return end;
}

void lmo_pssm_from_sites(const uint8_t *sites, size_t nsites, size_t m,
                         size_t k, float pseudocount, const float *background,
                         float *pssm, size_t pssm_stride)
{
    const size_t def = k - 1; /* default symbol is the last one (N / X) */
    float bg[64];
    for (size_t j = 0; j < k; j++) /* abc.rs:473-487 Background::uniform */
        bg[j] = background ? background[j]
                           : (j != def ? 1.0f / (float)(k - 1) : 0.0f);
    memset(pssm, 0, m * pssm_stride * sizeof(float));
    for (size_t i = 0; i < m; i++) {
        float *dst = pssm + i * pssm_stride;
        /* pwm/mod.rs:228-230 counts, :249-251 add pseudocounts
         * (abc.rs:558-573: scalar pseudocount on all but the default) */
        for (size_t j = 0; j < k; j++) {
            unsigned cnt = 0;
            for (size_t s = 0; s < nsites; s++)
                cnt += (sites[s * m + i] == j);
            dst[j] = (float)cnt + (j != def ? pseudocount : 0.0f);
        }
        float sum = 0.0f; /* :252 iter().sum() */
        for (size_t j = 0; j < k; j++)
            sum = sum + dst[j];
        for (size_t j = 0; j < k; j++) /* :253-255 */
            dst[j] = dst[j] / sum;
        for (size_t j = 0; j < k; j++) { /* pwm/mod.rs:420-427 */
            if (bg[j] == 0.0f)
                dst[j] = -INFINITY;
            else
                dst[j] = log2f(dst[j] / bg[j]);
        }
    }
}

float lmo_score_position(const uint8_t *seq, size_t seq_stride, size_t rows,
                         const float *pssm, size_t m, size_t pssm_stride,
                         size_t pos)
{
    float score = 0.0f; /* pwm/mod.rs:656 */
    for (size_t j = 0; j < m; j++) {
        const size_t idx = pos + j; /* seq.rs:436-441 */
        const uint8_t sym = seq[(idx % rows) * seq_stride + idx / rows];
        score = score + pssm[j * pssm_stride + sym];
    }
    return score;
}
