// layout.hip -- Encode / Stripe / configure_wrap on the device (the steps right
// before the scoring hot path).
//
//   encode          lightmotif/src/pli/mod.rs:56-66 + Symbol::from_ascii (abc.rs:166-171,
//                   296-325); lossy form seq.rs:122-129
//   stripe          pli/mod.rs:178-200: position i -> data[i % rows][i / rows],
//                   cells past the end = default symbol; fresh rows are T::default() =
//                   the default symbol (dense.rs:144-147), and so is the alignment
//                   padding past `cols` here (unspecified struct padding in the reference)
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
        const unsigned k = protein ? 21u : 5u;
        lut[threadIdx.x] = 0xff;
        __syncthreads();
        if (threadIdx.x < k)
            lut[(uint8_t)order[threadIdx.x]] = (uint8_t)threadIdx.x;
        __syncthreads();
    }
    const uint8_t def = protein ? 20 : 4;
    unsigned long long bad = ~0ull;
    // 16 bytes per lane per iteration (coalesced 16-B loads/stores); the byte -> symbol
    // map is a 256-entry LDS table (all 64 lanes hit <= 21 distinct bytes: broadcasts).
    const bool aligned = ((reinterpret_cast<uintptr_t>(ascii) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const unsigned long long n16 = aligned ? len / 16 : 0;
    const uint4 *src16 = reinterpret_cast<const uint4 *>(ascii);
    uint4 *dst16 = reinterpret_cast<uint4 *>(dst);
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < n16;
         i += (unsigned long long)gridDim.x * kBlock) {
        const uint4 v = src16[i];
        const unsigned in[4] = {v.x, v.y, v.z, v.w};
        unsigned out[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned o = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned sym = lut[(in[w] >> (8 * q)) & 0xff];
                if (sym == 0xff) {
                    if (lossy) {
                        sym = def;           // seq.rs:126 unwrap_or_default
                    } else {
                        const unsigned long long at = i * 16 + w * 4 + q;
                        if (at < bad)
                            bad = at;        // pli/mod.rs:63 `?` -> Err(InvalidSymbol)
                    }
                }
                o |= sym << (8 * q);
            }
            out[w] = o;
        }
        dst16[i] = make_uint4(out[0], out[1], out[2], out[3]);
    }
    for (unsigned long long i = n16 * 16 + (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
         i < len; i += (unsigned long long)gridDim.x * kBlock) {
        uint8_t sym = lut[ascii[i]];
        if (sym == 0xff) {
            if (lossy)
                sym = def;
            else if (i < bad)
                bad = i;
        }
        dst[i] = sym;
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
            (len / 16 + kBlock) / kBlock, (unsigned long long)ctx->num_cus * 64);
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

// ---- symbol validation ----------------------------------------------------------------------
// The reference's symbols are enums, so a StripedSequence cannot hold a byte >= K.  Bytes
// entering through the C ABI can; they would index past the M x K tables of the scoring
// kernels.  One pass over the `cols` live bytes of every row (4 B / lane when the layout
// allows) finds the largest symbol.
__global__ __launch_bounds__(kBlock) void max_symbol_kernel(const uint8_t *__restrict__ data,
                                                            const unsigned long long rows,
                                                            const unsigned long long stride,
                                                            const unsigned long long cols,
                                                            unsigned *__restrict__ out)
{
    unsigned m = 0;
    const unsigned long long tid = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long step = (unsigned long long)gridDim.x * kBlock;
    if (stride == cols) {
        // one flat run of rows * cols bytes (an encoded sequence arrives as rows = 1, cols = len,
        // and len may exceed 2^32): bytes up to the first 4-byte boundary, whole dwords, byte tail
        const unsigned long long n = rows * cols;
        const unsigned long long head = min(n, (unsigned long long)((4 - reinterpret_cast<uintptr_t>(data) % 4) % 4));
        const unsigned long long n4 = (n - head) / 4;
        const unsigned *d4 = reinterpret_cast<const unsigned *>(data + head);
        for (unsigned long long i = tid; i < n4; i += step) {
            const unsigned w = d4[i];
            m = max(max(m, w & 0xff), max((w >> 8) & 0xff, max((w >> 16) & 0xff, w >> 24)));
        }
        if (tid < head)
            m = max(m, (unsigned)data[tid]);
        const unsigned long long tail0 = head + n4 * 4;
        if (tid < n - tail0)
            m = max(m, (unsigned)data[tail0 + tid]);
    } else {
        // padded rows: only the `cols` live bytes of each row are symbols
        const unsigned long long n = rows * cols;
        for (unsigned long long i = tid; i < n; i += step)
            m = max(m, (unsigned)data[(i / cols) * stride + i % cols]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m)
        atomicMax(out, m);
}

int launch_max_symbol(lm_hip_ctx *ctx, const uint8_t *d_data, size_t rows, size_t stride, size_t cols,
                      unsigned *max_symbol)
{
    *max_symbol = 0;
    if (rows == 0 || cols == 0)
        return LM_HIP_OK;
    LM_TRY(ctx->scratch.reserve(16));
    unsigned *d_max = static_cast<unsigned *>(ctx->scratch.ptr);
    LM_HIP_TRY(hipMemsetAsync(d_max, 0, 4, ctx->stream));
    const unsigned long long work = (unsigned long long)rows * cols / 4 + 1;
    const unsigned grid = (unsigned)std::min<unsigned long long>((work + kBlock - 1) / kBlock,
                                                                 (unsigned long long)ctx->num_cus * 32);
    hipLaunchKernelGGL(max_symbol_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, d_data,
                       (unsigned long long)rows, (unsigned long long)stride, (unsigned long long)cols, d_max);
    LM_HIP_TRY(hipGetLastError());
    LM_HIP_TRY(hipMemcpyAsync(ctx->pinned, d_max, 4, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *max_symbol = *static_cast<unsigned *>(ctx->pinned);
    return LM_HIP_OK;
}

// ---- stripe -------------------------------------------------------------------------------
//
// Every stripe kernel works on a TILE of the striped matrix: output rows [rbase, rbase + nrows) of the
// `rows` the whole sequence has, read from `cols` pieces of `pitch` bytes -- piece c holds the symbols of
// column c for exactly those rows, src[c * pitch + rr] = symbol at position c * rows + rbase + rr
// (pli/mod.rs:192).  The classic whole-sequence call is the tile rbase = 0, nrows = pitch = rows.  Tiles are
// what the host-side ingest uploads, two in flight (handles.hip: ingest_tiled), so that a genome needs two tiles
// of staging instead of a device copy of itself.
//
// The bytes of a piece are turned into symbols on the way (XF): as they are (device-side encode ran before, or
// the caller vouches), validated against the alphabet size (host bytes: a byte >= k would index past the
// scoring kernels' tables), ASCII through Symbol::from_ascii's table (abc.rs:166-171, 296-325; pli/mod.rs:56-66;
// lossy: seq.rs:122-129), or unpacked from 2 bits per base (4 bases per byte, base i in bits 2*(i%4).. of byte
// i/4, values A0 C1 T2 G3 = abc.rs:115-135; optional N mask, bit i%8 of byte i/8).  The first offending
// POSITION of the sequence (smallest index) is what a failed conversion reports (pli/mod.rs:63).
enum : int { XF_NONE = 0, XF_CHECK = 1, XF_ASCII = 2, XF_2BIT = 3 };

struct StripeXf {
    const uint8_t *mask;              // XF_2BIT: N mask or null
    unsigned long long *first_bad;    // XF_CHECK / XF_ASCII (strict): smallest offending position
    unsigned k;                       // alphabet size
    int protein, lossy;
};

__device__ __forceinline__ void xf_fill_lut(uint8_t *lut, const int protein)
{
    // abc.rs:106-108 "ACTGN", abc.rs:193-256 "ACDEFGHIKLMNPQRSTVWYX"
    const char *order = protein ? "ACDEFGHIKLMNPQRSTVWYX" : "ACTGN";
    const unsigned k = protein ? 21u : 5u;
    for (unsigned i = threadIdx.x; i < 256; i += blockDim.x)
        lut[i] = 0xff;
    __syncthreads();
    if (threadIdx.x < k)
        lut[(uint8_t)order[threadIdx.x]] = (uint8_t)threadIdx.x;
    __syncthreads();
}

template <int XF>
__device__ __forceinline__ unsigned xf_symbol(const unsigned b, const uint8_t *lut, const StripeXf &xf, const uint8_t def,
                                              const unsigned long long pos, unsigned long long &bad)
{
    if (XF == XF_NONE)
        return b;
    unsigned sym = XF == XF_CHECK ? (b < xf.k ? b : 0xffu) : lut[b];
    if (sym == 0xffu) {
        sym = def;  // seq.rs:126 unwrap_or_default
        if (!xf.lossy && pos < bad)
            bad = pos;  // pli/mod.rs:63 `?` -> Err(InvalidSymbol)
    }
    return sym;
}

// 2-bit source: symbol at position i
__device__ __forceinline__ unsigned xf_unpack(const uint8_t *packed, const uint8_t *mask, const unsigned long long i)
{
    if (mask && ((mask[i >> 3] >> (i & 7)) & 1))
        return 4;
    return (packed[i >> 2] >> (2 * (i & 3))) & 3u;
}

// One workgroup transposes kTileRows striped rows: for each column c the bytes of piece c are contiguous in the
// input (coalesced reads), and each output row is `stride` contiguous bytes (coalesced writes).  Any geometry.
constexpr int kTileRows = 256;

template <int XF>
__global__ __launch_bounds__(kBlock) void stripe_kernel(const uint8_t *__restrict__ src, const unsigned long long pitch,
                                                        const unsigned long long len, const unsigned long long rows,
                                                        const unsigned long long rbase, const unsigned long long nrows,
                                                        const unsigned cols, const uint8_t def, uint8_t *__restrict__ data,
                                                        const unsigned long long stride, const StripeXf xf)
{
    extern __shared__ uint8_t tile[];  // [kTileRows][stride + 1]
    __shared__ uint8_t lut[256];
    if (XF == XF_ASCII)
        xf_fill_lut(lut, xf.protein);
    unsigned long long bad = ~0ull;
    const unsigned long long r0 = (unsigned long long)blockIdx.x * kTileRows;
    const unsigned tpitch = (unsigned)stride + 1;
    const unsigned long long rr = r0 + threadIdx.x;
    for (unsigned c = 0; c < stride; ++c) {
        unsigned v = def;  // alignment padding past `cols`: default symbol (dense.rs:144-147)
        if (c < cols && rr < nrows) {
            const unsigned long long i = (unsigned long long)c * rows + rbase + rr;  // pli/mod.rs:192
            if (i < len)                                                             // pli/mod.rs:195
                v = XF == XF_2BIT ? xf_unpack(src, xf.mask, i) : xf_symbol<XF>(src[c * pitch + rr], lut, xf, def, i, bad);
        }
        tile[threadIdx.x * tpitch + c] = (uint8_t)v;
    }
    __syncthreads();
    const unsigned long long n = nrows - r0 < kTileRows ? nrows - r0 : kTileRows;
    const unsigned long long nbytes = n * stride;
    uint8_t *dst = data + (rbase + r0) * stride;
    for (unsigned long long b = threadIdx.x; b < nbytes; b += kBlock) {
        const unsigned tr = (unsigned)(b / stride), cc = (unsigned)(b - (unsigned long long)tr * stride);
        dst[b] = tile[tr * tpitch + cc];
    }
    if ((XF == XF_CHECK || XF == XF_ASCII) && bad != ~0ull)
        atomicMin(xf.first_bad, bad);
}

// C = 32, stride 32 (the layout every scoring kernel runs on).  Phase 1 keeps the
// input's orientation: a lane loads 16 consecutive rows of one column with ONE 16-byte
// load (a wavefront reads 1 KB of a column per instruction -- with dword loads the
// kernel was bound by the load-request rate: 2.7 vs 5.1 TB/s, tools/kbench/stripe_bench)
// and stores them to tile[c][piece], consecutive lanes -> consecutive LDS addresses.
// Phase 2 gives every lane ONE output row: 32 byte reads tile[c][row] (consecutive
// lanes -> consecutive bytes of one LDS row: 16 dwords per wavefront, no conflicts),
// packed into two 16-byte stores, so a wavefront writes 2 KB of contiguous output per
// pair of store instructions.  The 2-bit form reads 16 bases = 32 bits (+ the bit offset of the
// piece, + 16 mask bits) instead of 16 bytes.
constexpr int kStripeRows = 4 * kBlock;

template <int XF>
__global__ __launch_bounds__(kBlock) void stripe_kernel_c32(const uint8_t *__restrict__ src, const unsigned long long pitch,
                                                            const unsigned long long len, const unsigned long long rows,
                                                            const unsigned long long rbase, const unsigned long long nrows,
                                                            const uint8_t def, uint8_t *__restrict__ data, const StripeXf xf)
{
    __shared__ uint4 tile[32][kStripeRows / 16];
    __shared__ uint8_t lut[256];
    if (XF == XF_ASCII)
        xf_fill_lut(lut, xf.protein);
    unsigned long long bad = ~0ull;
    const unsigned long long r0 = (unsigned long long)blockIdx.x * kStripeRows;
    constexpr unsigned per_col = kStripeRows / 16;  // 16-row pieces per column
#pragma unroll 4
    for (unsigned p = threadIdx.x; p < 32 * per_col; p += kBlock) {
        const unsigned c = p / per_col, q = p % per_col;
        const unsigned long long rr = r0 + 16ull * q;                             // row within the tile
        const unsigned long long i = (unsigned long long)c * rows + rbase + rr;  // pli/mod.rs:192
        unsigned w[4] = {0, 0, 0, 0};
        if (XF == XF_2BIT) {
            if (rr + 15 < nrows && i + 15 < len) {
                unsigned long long bits;  // 16 bases from bit 2 * (i % 4) of byte i / 4 on (the buffer carries 16 spare bytes)
                __builtin_memcpy(&bits, src + (i >> 2), 8);
                unsigned b32 = (unsigned)(bits >> (2 * (i & 3)));
                unsigned m16 = 0;
                if (xf.mask) {
                    unsigned mm;
                    __builtin_memcpy(&mm, xf.mask + (i >> 3), 4);
                    m16 = (mm >> (i & 7)) & 0xffffu;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const unsigned b = ((m16 >> k) & 1) ? 4u : ((b32 >> (2 * k)) & 3u);
                    w[k / 4] |= b << (8 * (k % 4));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const unsigned b = (rr + k < nrows) ? (i + k < len ? xf_unpack(src, xf.mask, i + k) : def) : 0;
                    w[k / 4] |= b << (8 * (k % 4));
                }
            }
        } else if (rr + 15 < nrows && i + 15 < len) {
            uint4 v;
            __builtin_memcpy(&v, src + c * pitch + rr, 16);  // unaligned 16-byte load
            if (XF == XF_NONE) {
                w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
            } else {
                const unsigned in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    w[k / 4] |= xf_symbol<XF>((in[k / 4] >> (8 * (k % 4))) & 0xffu, lut, xf, def, i + k, bad) << (8 * (k % 4));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                unsigned b = 0;
                if (rr + k < nrows)
                    b = i + k < len ? xf_symbol<XF>(src[c * pitch + rr + k], lut, xf, def, i + k, bad) : def;  // :195
                w[k / 4] |= b << (8 * (k % 4));
            }
        }
        tile[c][q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    const uint8_t *tb = reinterpret_cast<const uint8_t *>(&tile[0][0]);
#pragma unroll 1
    for (int it = 0; it < kStripeRows / kBlock; ++it) {
        const unsigned lr = it * kBlock + threadIdx.x;
        if (r0 + lr < nrows) {
            unsigned w[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const unsigned b0 = tb[(4 * g + 0) * kStripeRows + lr], b1 = tb[(4 * g + 1) * kStripeRows + lr];
                const unsigned b2 = tb[(4 * g + 2) * kStripeRows + lr], b3 = tb[(4 * g + 3) * kStripeRows + lr];
                w[g] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            }
            uint4 *dst = reinterpret_cast<uint4 *>(data + (rbase + r0 + lr) * 32);
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
    if ((XF == XF_CHECK || XF == XF_ASCII) && bad != ~0ull)
        atomicMin(xf.first_bad, bad);
}

// Whole-sequence fast path for the other 4-byte-aligned geometries (C = 16 ...; stride % 4 == 0, 4-byte aligned output): a workgroup transposes 1024
// striped rows.  Phase 1: for every column c the 1024 bytes enc[c*rows + r0 ..] are
// contiguous -> one (unaligned) dword load per lane = 4 consecutive rows, scattered
// into an LDS tile [row][pitch].  Phase 2: the tile is read back as dwords (4 columns
// of one row) and written with fully coalesced dword stores.
constexpr int kFastTileRows = 4 * kBlock;

__global__ __launch_bounds__(kBlock) void stripe_kernel_fast(const uint8_t *__restrict__ enc,
                                                             const unsigned long long len,
                                                             const unsigned long long rows,
                                                             const unsigned cols, const uint8_t def,
                                                             uint8_t *__restrict__ data,
                                                             const unsigned stride)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];  // [kFastTileRows][stride + 4]
    const unsigned pitch = stride + 4;
    const unsigned long long r0 = (unsigned long long)blockIdx.x * kFastTileRows;
    const unsigned long long r = r0 + 4ull * threadIdx.x;
    for (unsigned c = 0; c < cols; ++c) {
        const unsigned long long i = (unsigned long long)c * rows + r;  // pli/mod.rs:192
        unsigned v;
        if (r + 3 < rows && i + 3 < len) {
            __builtin_memcpy(&v, enc + i, 4);  // unaligned dword load
        } else {
            v = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned b = (r + q < rows && i + q < len) ? enc[i + q] : def;  // :195
                v |= b << (8 * q);
            }
        }
        uint8_t *t = tile + (4u * threadIdx.x) * pitch + c;
        t[0] = (uint8_t)v;
        t[pitch] = (uint8_t)(v >> 8);
        t[2 * pitch] = (uint8_t)(v >> 16);
        t[3 * pitch] = (uint8_t)(v >> 24);
    }
    // alignment padding past `cols`: the default symbol, like every element of a fresh row
    // (dense.rs:144-147)
    for (unsigned c = cols; c < stride; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            tile[(4u * threadIdx.x + q) * pitch + c] = def;
    __syncthreads();
    const unsigned long long nrows = rows - r0 < kFastTileRows ? rows - r0 : kFastTileRows;
    const unsigned dw_per_row = stride / 4;
    const unsigned long long ndw = nrows * dw_per_row;
    unsigned *dst = reinterpret_cast<unsigned *>(data + r0 * stride);
    for (unsigned long long d = threadIdx.x; d < ndw; d += kBlock) {
        const unsigned rr = (unsigned)(d / dw_per_row), cc = (unsigned)(d - (unsigned long long)rr * dw_per_row);
        dst[d] = *reinterpret_cast<const unsigned *>(tile + rr * pitch + 4 * cc);
    }
}

// N runs of a 2-bit genome (the .2bit container's nBlockStarts / nBlockSizes): positions [start, start + size) of run
// blockIdx.y become the default symbol.  Consecutive positions are consecutive ROWS of one column (pli/mod.rs:192),
// i.e. byte writes `stride` apart -- paid only for the N positions themselves.
__global__ __launch_bounds__(kBlock) void n_runs_kernel(const unsigned long long *__restrict__ runs, const unsigned long long rows,
                                                        const unsigned long long stride, const uint8_t def,
                                                        uint8_t *__restrict__ data)
{
    const unsigned long long start = runs[2 * blockIdx.y], size = runs[2 * blockIdx.y + 1];
    for (unsigned long long o = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; o < size;
         o += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long i = start + o;
        data[(i % rows) * stride + i / rows] = def;
    }
}

int launch_n_runs(lm_hip_ctx *ctx, const unsigned long long *d_runs, size_t nruns, unsigned long long longest, size_t rows,
                  size_t stride, uint8_t def, uint8_t *d_data)
{
    const unsigned gx = (unsigned)std::min<unsigned long long>(std::max<unsigned long long>((longest + kBlock - 1) / kBlock, 1), 256);
    for (size_t b = 0; b < nruns; b += 65535) {
        const unsigned gy = (unsigned)std::min<size_t>(nruns - b, 65535);
        hipLaunchKernelGGL(n_runs_kernel, dim3(gx, gy), dim3(kBlock), 0, ctx->stream, d_runs + 2 * b,
                           (unsigned long long)rows, (unsigned long long)stride, def, d_data);
        LM_HIP_TRY(hipGetLastError());
    }
    return LM_HIP_OK;
}

// Wrap rows in closed form.  seq.rs:373-378 runs
//     for i in 0..m { data[rows+i][j] = data[i][j+1] (j < C-1); data[rows+i][C-1] = default }
// sequentially in place, so for i >= rows the source row is itself a wrap row
// written earlier in the loop; unrolling that recursion gives
//     wrap[i][j] = orig[i % rows][j + 1 + i / rows]   if that column exists, else default.
// With rows == 0 the loop reads the fresh row it is writing (T::default() = the default
// symbol, dense.rs:144-147): every cell is the default symbol.
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
        uint8_t v = def;  // padding past `cols` and the rows == 0 case: default symbol
        if (j < cols) {
            if (rows == 0) {
                v = def;
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

template <int XF>
static int launch_stripe_tile_xf(lm_hip_ctx *ctx, const StripeTile &t, const StripeXf &xf)
{
    if (t.nrows == 0)
        return LM_HIP_OK;
    uint8_t *d_data = t.d_data;
    if (t.cols == 32 && t.stride == 32 && reinterpret_cast<uintptr_t>(d_data) % 16 == 0) {
        const unsigned grid = (unsigned)((t.nrows + kStripeRows - 1) / kStripeRows);
        hipLaunchKernelGGL(stripe_kernel_c32<XF>, dim3(grid), dim3(kBlock), 0, ctx->stream, t.d_src,
                           (unsigned long long)t.pitch, (unsigned long long)t.len, (unsigned long long)t.rows,
                           (unsigned long long)t.rbase, (unsigned long long)t.nrows, t.def, d_data, xf);
    } else {
        const unsigned grid = (unsigned)((t.nrows + kTileRows - 1) / kTileRows);
        const size_t lds = (size_t)kTileRows * (t.stride + 1);
        if (lds > 60 * 1024)
            return fail(LM_HIP_ERR_BAD_ARGS, "stripe: stride %zu too large", t.stride);
        hipLaunchKernelGGL(stripe_kernel<XF>, dim3(grid), dim3(kBlock), lds, ctx->stream, t.d_src,
                           (unsigned long long)t.pitch, (unsigned long long)t.len, (unsigned long long)t.rows,
                           (unsigned long long)t.rbase, (unsigned long long)t.nrows, (unsigned)t.cols, t.def, d_data,
                           (unsigned long long)t.stride, xf);
    }
    LM_HIP_TRY(hipGetLastError());
    return LM_HIP_OK;
}

int launch_stripe_tile(lm_hip_ctx *ctx, const StripeTile &t)
{
    StripeXf xf{t.d_mask, t.d_first_bad, (unsigned)t.k, t.protein ? 1 : 0, t.lossy ? 1 : 0};
    switch (t.transform) {
    case StripeTile::None: return launch_stripe_tile_xf<XF_NONE>(ctx, t, xf);
    case StripeTile::Check: return launch_stripe_tile_xf<XF_CHECK>(ctx, t, xf);
    case StripeTile::Ascii: return launch_stripe_tile_xf<XF_ASCII>(ctx, t, xf);
    case StripeTile::TwoBit: return launch_stripe_tile_xf<XF_2BIT>(ctx, t, xf);
    }
    return fail(LM_HIP_ERR_BAD_ARGS, "stripe: unknown transform");
}

int launch_stripe(lm_hip_ctx *ctx, const uint8_t *d_encoded, size_t len, size_t cols,
                  uint8_t default_symbol, size_t wrap, uint8_t *d_data, size_t stride)
{
    const unsigned long long rows = (len + cols - 1) / cols;  // pli/mod.rs:182
    // the fast path for other 4-byte-aligned geometries (C = 16 ...)
    const size_t fast_lds = (size_t)kFastTileRows * (stride + 4);
    if (rows && !(cols == 32 && stride == 32 && reinterpret_cast<uintptr_t>(d_data) % 16 == 0) && stride % 4 == 0 &&
        fast_lds <= 60 * 1024 && reinterpret_cast<uintptr_t>(d_data) % 4 == 0) {
        const unsigned grid = (unsigned)((rows + kFastTileRows - 1) / kFastTileRows);
        hipLaunchKernelGGL(stripe_kernel_fast, dim3(grid), dim3(kBlock), fast_lds, ctx->stream,
                           d_encoded, (unsigned long long)len, rows, (unsigned)cols, default_symbol,
                           d_data, (unsigned)stride);
        LM_HIP_TRY(hipGetLastError());
    } else if (rows) {
        StripeTile t;
        t.d_src = d_encoded;
        t.pitch = rows;
        t.len = len;
        t.rows = rows;
        t.rbase = 0;
        t.nrows = rows;
        t.cols = cols;
        t.stride = stride;
        t.def = default_symbol;
        t.d_data = d_data;
        LM_TRY(launch_stripe_tile(ctx, t));
    }
    return launch_wrap(ctx, d_data, rows, stride, cols, wrap, default_symbol);
}

}  // namespace lm
