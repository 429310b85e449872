// layout.hip -- Encode / Stripe / configure_wrap on the device (the steps right
// before the scoring hot path).
//
//   encode          lightmotif/src/pli/mod.rs:56-66 + Symbol::from_ascii (abc.rs:166-171,
//                   296-325); lossy form seq.rs:122-129
//   stripe          pli/mod.rs:178-200: position i -> data[i % rows][i / rows],
//                   cells past the end = default symbol, row padding zeroed
//                   (dense.rs:144-147)
//   configure_wrap  seq.rs:369-381
// All integer/byte work, HBM-bound: coalesced loads, LDS-tiled transpose.
#include <algorithm>

#include "score_kernels.hpp"

namespace lm {

// ---- encode -----------------------------------------------------------------------------

__global__ __launch_bounds__(kBlock) void encode_kernel(const uint8_t *__restrict__ ascii,
                                                        const unsigned long long len,
                                                        const int protein, const int lossy,
                                                        uint8_t *__restrict__ dst,
                                                        unsigned long long *__restrict__ first_bad)
{
    __shared__ uint8_t lut[256];
    {
        // abc.rs:106-108 "ACTGN", abc.rs:193-256 "ACDEFGHIKLMNPQRSTVWYX"
        const char *order = protein ? "ACDEFGHIKLMNPQRSTVWYX" : "ACTGN";
        const int k = protein ? 21 : 5;
        lut[threadIdx.x] = 0xff;
        __syncthreads();
        if (threadIdx.x < k)
            lut[(uint8_t)order[threadIdx.x]] = (uint8_t)threadIdx.x;
        __syncthreads();
    }
    const uint8_t def = protein ? 20 : 4;
    unsigned long long bad = ~0ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < len;
         i += (unsigned long long)gridDim.x * kBlock) {
        uint8_t s = lut[ascii[i]];
        if (s == 0xff) {
            if (lossy)
                s = def;                 // seq.rs:126 unwrap_or_default
            else if (i < bad)
                bad = i;                 // pli/mod.rs:63 `?` -> Err(InvalidSymbol)
        }
        dst[i] = s;
    }
    if (bad != ~0ull)
        atomicMin(first_bad, bad);
}

int launch_encode(lm_hip_ctx *ctx, char alphabet, const uint8_t *d_ascii, size_t len, int lossy,
                  uint8_t *d_dst, size_t *bad_index)
{
    LM_TRY(ctx->scratch.reserve(16));
    unsigned long long *d_bad = static_cast<unsigned long long *>(ctx->scratch.ptr);
    LM_HIP_TRY(hipMemsetAsync(d_bad, 0xff, 8, ctx->stream));
    if (len) {
        const unsigned grid = (unsigned)std::min<unsigned long long>(
            (len + kBlock - 1) / kBlock, (unsigned long long)ctx->num_cus * 32);
        hipLaunchKernelGGL(encode_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d_ascii,
                           (unsigned long long)len, alphabet == 'P' ? 1 : 0, lossy, d_dst, d_bad);
        LM_HIP_TRY(hipGetLastError());
    }
    LM_HIP_TRY(hipMemcpyAsync(ctx->pinned, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const unsigned long long bad = *static_cast<unsigned long long *>(ctx->pinned);
    if (bad != ~0ull) {
        if (bad_index)
            *bad_index = (size_t)bad;
        return fail(LM_HIP_ERR_INVALID_SYMBOL, "invalid symbol at position %llu", bad);
    }
    return LM_HIP_OK;
}

// ---- stripe -------------------------------------------------------------------------------

// One workgroup transposes a tile of TR striped rows: for each column c the TR
// bytes enc[c*rows + r0 ..] are contiguous in the input (coalesced reads), and
// each output row is `stride` contiguous bytes (coalesced writes).
constexpr int kTileRows = 256;

__global__ __launch_bounds__(kBlock) void stripe_kernel(const uint8_t *__restrict__ enc,
                                                        const unsigned long long len,
                                                        const unsigned long long rows,
                                                        const unsigned cols, const uint8_t def,
                                                        uint8_t *__restrict__ data,
                                                        const unsigned long long stride)
{
    extern __shared__ uint8_t tile[];  // [kTileRows][stride + 1]
    const unsigned long long r0 = (unsigned long long)blockIdx.x * kTileRows;
    const unsigned pitch = (unsigned)stride + 1;
    const unsigned long long r = r0 + threadIdx.x;
    for (unsigned c = 0; c < stride; ++c) {
        uint8_t v = 0;  // alignment padding past `cols` (dense.rs:144-147)
        if (c < cols && r < rows) {
            const unsigned long long i = (unsigned long long)c * rows + r;  // pli/mod.rs:192
            v = i < len ? enc[i] : def;                                     // pli/mod.rs:195
        }
        tile[threadIdx.x * pitch + c] = v;
    }
    __syncthreads();
    const unsigned long long nrows = rows - r0 < kTileRows ? rows - r0 : kTileRows;
    const unsigned long long nbytes = nrows * stride;
    uint8_t *dst = data + r0 * stride;
    for (unsigned long long b = threadIdx.x; b < nbytes; b += kBlock) {
        const unsigned rr = (unsigned)(b / stride), cc = (unsigned)(b - (unsigned long long)rr * stride);
        dst[b] = tile[rr * pitch + cc];
    }
}

// Wrap rows in closed form.  seq.rs:373-378 runs
//     for i in 0..m { data[rows+i][j] = data[i][j+1] (j < C-1); data[rows+i][C-1] = default }
// sequentially in place, so for i >= rows the source row is itself a wrap row
// written earlier in the loop; unrolling that recursion gives
//     wrap[i][j] = orig[i % rows][j + 1 + i / rows]   if that column exists, else default.
// With rows == 0 the loop reads the (zero-filled) row it is writing: zeros, then default.
__global__ __launch_bounds__(kBlock) void wrap_kernel(uint8_t *__restrict__ data,
                                                      const unsigned long long rows,
                                                      const unsigned long long stride,
                                                      const unsigned cols,
                                                      const unsigned long long m, const uint8_t def)
{
    const unsigned long long n = m * stride;
    for (unsigned long long b = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; b < n;
         b += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long i = b / stride;
        const unsigned j = (unsigned)(b - i * stride);
        uint8_t v = 0;
        if (j < cols) {
            if (rows == 0) {
                v = (j == cols - 1) ? def : 0;
            } else {
                const unsigned long long src_col = (unsigned long long)j + 1 + i / rows;
                v = src_col < cols ? data[(i % rows) * stride + src_col] : def;
            }
        }
        data[(rows + i) * stride + j] = v;
    }
}

int launch_wrap(lm_hip_ctx *ctx, uint8_t *d_data, size_t rows, size_t stride, size_t cols,
                size_t new_wrap, uint8_t default_symbol)
{
    if (new_wrap == 0)
        return LM_HIP_OK;
    const unsigned long long n = (unsigned long long)new_wrap * stride;
    const unsigned grid = (unsigned)std::min<unsigned long long>((n + kBlock - 1) / kBlock, 4096);
    hipLaunchKernelGGL(wrap_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d_data,
                       (unsigned long long)rows, (unsigned long long)stride, (unsigned)cols,
                       (unsigned long long)new_wrap, default_symbol);
    LM_HIP_TRY(hipGetLastError());
    return LM_HIP_OK;
}

int launch_stripe(lm_hip_ctx *ctx, const uint8_t *d_encoded, size_t len, size_t cols,
                  uint8_t default_symbol, size_t wrap, uint8_t *d_data, size_t stride)
{
    const unsigned long long rows = (len + cols - 1) / cols;  // pli/mod.rs:182
    if (rows) {
        const unsigned grid = (unsigned)((rows + kTileRows - 1) / kTileRows);
        const size_t lds = (size_t)kTileRows * (stride + 1);
        if (lds > 60 * 1024)
            return fail(LM_HIP_ERR_BAD_ARGS, "stripe: stride %zu too large", stride);
        hipLaunchKernelGGL(stripe_kernel, dim3(grid), dim3(kBlock), lds, ctx->stream, d_encoded,
                           (unsigned long long)len, rows, (unsigned)cols, default_symbol, d_data,
                           (unsigned long long)stride);
        LM_HIP_TRY(hipGetLastError());
    }
    return launch_wrap(ctx, d_data, rows, stride, cols, wrap, default_symbol);
}

}  // namespace lm
