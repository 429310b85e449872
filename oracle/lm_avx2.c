/*
 * lm_avx2.c -- C-intrinsics port of the reference's AVX2 back-end, used ONLY
 * as the CPU baseline timed by bench.py (`cpu_baseline.kind == "port"`) and
 * cross-checked against lm_oracle.c in tests/.  TEST/BENCH INFRASTRUCTURE:
 * the product never links it.
 *
 * Follows, instruction for instruction:
 *   score : lightmotif/src/pli/platform/avx2.rs:104-199 (score_f32_avx2_permute,
 *           K <= 8) and :204-290 (score_f32_avx2_gather, any K)
 *   argmax: lightmotif/src/pli/platform/avx2.rs:351-426 (argmax_f32_avx2)
 *   u8    : lightmotif/src/pli/platform/avx2.rs:292-347 (score_u8_avx2_shuffle: what
 *           `Dispatch::Avx2` runs for Score<u8, Dna> -- dispatch.rs:124 -- i.e. per Scanner block)
 * The reference core crate is single-threaded; the *_mt entry points split
 * the row range across pthreads using the score_rows_into row-range contract
 * (pli/mod.rs:72-78), which is how a caller would parallelise it.
 *
 * Build: gcc -O3 -mavx2 (see Makefile).  Requires 32-byte aligned rows, like
 * the reference (dense.rs:43; debug_assert at avx2.rs:158,165).
 */
#include <immintrin.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

/* avx2.rs:104-199 / 204-290.  seq -> first row of the striped matrix (stride
 * 32), pssm stride in floats, rows [a,b), out -> row 0 of the score matrix
 * (stride 32 floats). */
static void score_rows_avx2(const uint8_t *seq, size_t seq_stride,
                            const float *pssm, size_t m, size_t pssm_stride,
                            size_t k, size_t a, size_t b, float *out,
                            size_t out_stride)
{
    /* avx2.rs:125-145: byte -> epi32 broadcast masks */
    const __m256i m1 = _mm256_set_epi32(0xFFFFFF03, 0xFFFFFF02, 0xFFFFFF01, 0xFFFFFF00,
                                        0xFFFFFF03, 0xFFFFFF02, 0xFFFFFF01, 0xFFFFFF00);
    const __m256i m2 = _mm256_set_epi32(0xFFFFFF07, 0xFFFFFF06, 0xFFFFFF05, 0xFFFFFF04,
                                        0xFFFFFF07, 0xFFFFFF06, 0xFFFFFF05, 0xFFFFFF04);
    const __m256i m3 = _mm256_set_epi32(0xFFFFFF0B, 0xFFFFFF0A, 0xFFFFFF09, 0xFFFFFF08,
                                        0xFFFFFF0B, 0xFFFFFF0A, 0xFFFFFF09, 0xFFFFFF08);
    const __m256i m4 = _mm256_set_epi32(0xFFFFFF0F, 0xFFFFFF0E, 0xFFFFFF0D, 0xFFFFFF0C,
                                        0xFFFFFF0F, 0xFFFFFF0E, 0xFFFFFF0D, 0xFFFFFF0C);
    const uint8_t *seqptr = seq + a * seq_stride;
    float *rowptr = out;
    for (size_t r = a; r < b; r++) {                  /* avx2.rs:146 */
        __m256 s1 = _mm256_setzero_ps(), s2 = _mm256_setzero_ps();
        __m256 s3 = _mm256_setzero_ps(), s4 = _mm256_setzero_ps();
        const uint8_t *seqrow = seqptr;
        const float *psmrow = pssm;
        for (size_t j = 0; j < m; j++) {              /* avx2.rs:156 */
            const __m256i x = _mm256_load_si256((const __m256i *)seqrow);
            const __m256i x1 = _mm256_shuffle_epi8(x, m1);
            const __m256i x2 = _mm256_shuffle_epi8(x, m2);
            const __m256i x3 = _mm256_shuffle_epi8(x, m3);
            const __m256i x4 = _mm256_shuffle_epi8(x, m4);
            __m256 b1, b2, b3, b4;
            if (k <= 8) {                             /* avx2.rs:166-171 */
                const __m256 t = _mm256_load_ps(psmrow);
                b1 = _mm256_permutevar8x32_ps(t, x1);
                b2 = _mm256_permutevar8x32_ps(t, x2);
                b3 = _mm256_permutevar8x32_ps(t, x3);
                b4 = _mm256_permutevar8x32_ps(t, x4);
            } else {                                  /* avx2.rs:259-262 */
                b1 = _mm256_i32gather_ps(psmrow, x1, 4);
                b2 = _mm256_i32gather_ps(psmrow, x2, 4);
                b3 = _mm256_i32gather_ps(psmrow, x3, 4);
                b4 = _mm256_i32gather_ps(psmrow, x4, 4);
            }
            s1 = _mm256_add_ps(s1, b1);               /* avx2.rs:173-176 */
            s2 = _mm256_add_ps(s2, b2);
            s3 = _mm256_add_ps(s3, b3);
            s4 = _mm256_add_ps(s4, b4);
            seqrow += seq_stride;
            psmrow += pssm_stride;
        }
        /* avx2.rs:182-185: restore column order */
        const __m256 r1 = _mm256_permute2f128_ps(s1, s2, 0x20);
        const __m256 r2 = _mm256_permute2f128_ps(s3, s4, 0x20);
        const __m256 r3 = _mm256_permute2f128_ps(s1, s2, 0x31);
        const __m256 r4 = _mm256_permute2f128_ps(s3, s4, 0x31);
        _mm256_stream_ps(rowptr + 0x00, r1);          /* avx2.rs:187-190 */
        _mm256_stream_ps(rowptr + 0x08, r2);
        _mm256_stream_ps(rowptr + 0x10, r3);
        _mm256_stream_ps(rowptr + 0x18, r4);
        rowptr += out_stride;
        seqptr += seq_stride;
    }
    _mm_sfence();                                     /* avx2.rs:198 */
}

/* avx2.rs:889-904 + 817-851: pre-checks then the kernel.  Returns 0 ok,
 * 2 when wrap < m-1 (the reference panics, avx2.rs:832-837). */
int lma_score_rows_f32(const uint8_t *seq, size_t seq_stride, size_t wrap,
                       size_t length, const float *pssm, size_t m,
                       size_t pssm_stride, size_t k, size_t row_begin,
                       size_t row_end, float *out, size_t out_stride)
{
    if (wrap + 1 < m)
        return 2;
    if (length < m || row_begin >= row_end) /* avx2.rs:839-842 */
        return 0;
    score_rows_avx2(seq, seq_stride, pssm, m, pssm_stride, k, row_begin,
                    row_end, out, out_stride);
    return 0;
}

struct job {
    const uint8_t *seq; size_t seq_stride; const float *pssm; size_t m;
    size_t pssm_stride; size_t k; size_t a, b; float *out; size_t out_stride;
};

static void *job_main(void *p)
{
    struct job *j = (struct job *)p;
    score_rows_avx2(j->seq, j->seq_stride, j->pssm, j->m, j->pssm_stride,
                    j->k, j->a, j->b, j->out, j->out_stride);
    return NULL;
}

/* Row range split evenly over `threads` pthreads (pli/mod.rs:72-78 contract:
 * each thread scores its own contiguous row block into its slice of `out`). */
int lma_score_rows_f32_mt(const uint8_t *seq, size_t seq_stride, size_t wrap,
                          size_t length, const float *pssm, size_t m,
                          size_t pssm_stride, size_t k, size_t row_begin,
                          size_t row_end, float *out, size_t out_stride,
                          int threads)
{
    if (wrap + 1 < m)
        return 2;
    if (length < m || row_begin >= row_end)
        return 0;
    if (threads < 1)
        threads = 1;
    const size_t n = row_end - row_begin;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    struct job *jobs = (struct job *)malloc(sizeof(struct job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        const size_t a = row_begin + n * (size_t)t / (size_t)threads;
        const size_t b = row_begin + n * (size_t)(t + 1) / (size_t)threads;
        jobs[t] = (struct job){seq, seq_stride, pssm, m, pssm_stride, k, a, b,
                               out + (a - row_begin) * out_stride, out_stride};
        pthread_create(&tid[t], NULL, job_main, &jobs[t]);
    }
    for (int t = 0; t < threads; t++)
        pthread_join(tid[t], NULL);
    free(tid);
    free(jobs);
    return 0;
}

/* avx2.rs:351-426.  Returns 1 + fills row/col, 0 if empty, -1 if
 * rows*32 > u32::MAX positions (the reference panics, avx2.rs:354-358; it
 * tests max_index, the caller passes it). */
int lma_argmax_f32(const float *scores, size_t rows, size_t stride,
                   size_t max_index, size_t *row, size_t *col)
{
    if (max_index > 0xFFFFFFFFull)
        return -1;
    if (rows == 0)
        return 0;
    const float *dataptr = scores;
    __m256i p1 = _mm256_setzero_si256(), p2 = p1, p3 = p1, p4 = p1;
    __m256 s1 = _mm256_load_ps(dataptr + 0x00), s2 = _mm256_load_ps(dataptr + 0x08);
    __m256 s3 = _mm256_load_ps(dataptr + 0x10), s4 = _mm256_load_ps(dataptr + 0x18);
    for (size_t i = 0; i < rows; i++) {               /* avx2.rs:376 */
        const __m256i index = _mm256_set1_epi32((int)i);
        const __m256 r1 = _mm256_load_ps(dataptr + 0x00);
        const __m256 r2 = _mm256_load_ps(dataptr + 0x08);
        const __m256 r3 = _mm256_load_ps(dataptr + 0x10);
        const __m256 r4 = _mm256_load_ps(dataptr + 0x18);
        const __m256 c1 = _mm256_cmp_ps(s1, r1, _CMP_LE_OS);
        const __m256 c2 = _mm256_cmp_ps(s2, r2, _CMP_LE_OS);
        const __m256 c3 = _mm256_cmp_ps(s3, r3, _CMP_LE_OS);
        const __m256 c4 = _mm256_cmp_ps(s4, r4, _CMP_LE_OS);
        p1 = _mm256_blendv_epi8(p1, index, _mm256_castps_si256(c1));
        p2 = _mm256_blendv_epi8(p2, index, _mm256_castps_si256(c2));
        p3 = _mm256_blendv_epi8(p3, index, _mm256_castps_si256(c3));
        p4 = _mm256_blendv_epi8(p4, index, _mm256_castps_si256(c4));
        s1 = _mm256_blendv_ps(s1, r1, c1);
        s2 = _mm256_blendv_ps(s2, r2, c2);
        s3 = _mm256_blendv_ps(s3, r3, c3);
        s4 = _mm256_blendv_ps(s4, r4, c4);
        dataptr += stride;
    }
    uint32_t x[32];
    _mm256_storeu_si256((__m256i *)(x + 0x00), p1);
    _mm256_storeu_si256((__m256i *)(x + 0x08), p2);
    _mm256_storeu_si256((__m256i *)(x + 0x10), p3);
    _mm256_storeu_si256((__m256i *)(x + 0x18), p4);
    size_t best_row = 0, best_col = 0;                /* avx2.rs:411-412 */
    float best = scores[0];
    for (size_t c = 0; c < 32; c++) {                 /* avx2.rs:414-421 */
        const float s = scores[(size_t)x[c] * stride + c];
        if (s > best) {
            best = s;
            best_row = x[c];
            best_col = c;
        }
    }
    *row = best_row;
    *col = best_col;
    return 1;
}

/* avx2.rs:292-347 (score_u8_avx2_shuffle) behind the pre-checks of avx2.rs:921-931: per output row the M sequence
 * rows are loaded (32 symbol bytes), each looks its weights up with one in-register byte shuffle of the (broadcast)
 * 16-byte PSSM row, saturating byte adds, non-temporal store.  weights: M rows of `wstride` bytes (>= 16, 16-byte
 * aligned), out: rows of `out_stride` bytes (32-byte aligned).  Returns 0 ok, 2 when wrap < m-1. */
int lma_score_rows_u8(const uint8_t *seq, size_t seq_stride, size_t wrap, size_t length,
                      const uint8_t *weights, size_t m, size_t wstride, size_t row_begin,
                      size_t row_end, uint8_t *out, size_t out_stride)
{
    if (wrap + 1 < m)
        return 2;
    if (length < m || row_begin >= row_end)
        return 0;
    uint8_t *rowptr = out;
    const uint8_t *seqptr = seq + row_begin * seq_stride;
    for (size_t r = row_begin; r < row_end; r++) {
        __m256i s = _mm256_setzero_si256();
        const uint8_t *seqrow = seqptr, *psmrow = weights;
        for (size_t j = 0; j < m; j++) {
            const __m256i x = _mm256_load_si256((const __m256i *)seqrow);
            const __m256i t = _mm256_castps_si256(_mm256_broadcast_ps((const __m128 *)psmrow));
            const __m256i y = _mm256_shuffle_epi8(t, x);
            s = _mm256_adds_epu8(s, y);
            seqrow += seq_stride;
            psmrow += wstride;
        }
        _mm256_stream_si256((__m256i *)rowptr, s);
        rowptr += out_stride;
        seqptr += seq_stride;
    }
    _mm_sfence();
    return 0;
}
