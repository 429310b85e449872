// reduce.hip -- argmax / max / threshold over a materialised StripedScores matrix.
//
// Semantics follow the reference's *Generic* bodies, which differ from its AVX2
// and SSE2 ones on ties:
//   argmax    lightmotif/src/pli/mod.rs:135-155  `x >= best` in a row-major scan
//             => the maximal cell that is LAST in (row, col) order; NaN never wins;
//             scores[0][0] == NaN => (0, 0)
//   max       pli/mod.rs:158-160                 value at argmax
//   threshold pli/mod.rs:210-221                 every cell with x >= t, pushed in
//             row-major order (NaN never selected)
// All of them scan the whole rows x cols matrix, padded tail included (scores.rs:181-213
// do not clip to max_index).
//
// These are pure streaming reads (4 B per cell): the roofline is HBM.
#include <algorithm>

#include "score_kernels.hpp"

namespace lm {

int finalize_argmax_materialised(lm_hip_ctx *ctx, const ArgmaxRecord *d_blocks, unsigned nblocks,
                                 const float *d_scores, int first_cell_rule, ArgmaxRecord *d_out);

// ---- argmax ---------------------------------------------------------------------------

// Contiguous case (stride == cols): the matrix is one flat array of `ncells`
// floats whose index IS the row-major rank.  Every workgroup owns ONE contiguous span of
// 16-byte pieces and keeps eight non-temporal loads per lane in flight: spans in place of a
// grid-stride walk, and many short-lived workgroups in place of few long ones, are worth
// 0.72 -> 0.58 ms per 4 GB here (tools/kbench/read_bench.hip: 5.5 -> 6.9 TB/s).
constexpr int kArgmaxLoads = 8;                                  // 16-byte loads in flight per lane
constexpr unsigned long long kArgmaxSpan = 256ull * kArgmaxLoads * 4;  // cells of one trip of a workgroup
constexpr unsigned kArgmaxMaxGrid = 65536;

__global__ __launch_bounds__(kBlock) void argmax_flat(const float *__restrict__ s,
                                                      const unsigned long long ncells,
                                                      const long long index_base,
                                                      ArgmaxRecord *__restrict__ blocks)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    static_assert(kBlock == 256, "argmax_flat: spans are laid out for 256 lanes");
    float v = -INFINITY;
    long long bi = -1;
    const unsigned long long n4 = ncells / 4;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 *s4 = reinterpret_cast<const f32x4 *>(s);
    // ascending index order inside the thread, so `>=` keeps the later cell
    auto take = [&](const f32x4 x, const long long base) {
        if (x.x >= v) { v = x.x; bi = base; }
        if (x.y >= v) { v = x.y; bi = base + 1; }
        if (x.z >= v) { v = x.z; bi = base + 2; }
        if (x.w >= v) { v = x.w; bi = base + 3; }
    };
    const unsigned long long per = (n4 + gridDim.x - 1) / gridDim.x;
    const unsigned long long a = (unsigned long long)blockIdx.x * per;
    const unsigned long long b = a + per < n4 ? a + per : n4;
    unsigned long long i = a + threadIdx.x;
    for (; i + (kArgmaxLoads - 1) * kBlock < b; i += kArgmaxLoads * kBlock) {
        f32x4 x[kArgmaxLoads];
#pragma unroll
        for (int u = 0; u < kArgmaxLoads; ++u)
            x[u] = __builtin_nontemporal_load(&s4[i + u * kBlock]);
#pragma unroll
        for (int u = 0; u < kArgmaxLoads; ++u)
            take(x[u], (long long)((i + u * kBlock) * 4));
    }
    for (; i < b; i += kBlock)
        take(__builtin_nontemporal_load(&s4[i]), (long long)(i * 4));
    // tail (ncells % 4) handled by the last thread of the grid
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) {
        for (unsigned long long i = n4 * 4; i < ncells; ++i) {
            const float x = s[i];
            float tv = x;
            long long ti = (long long)i;
            if (x == x)
                best_merge(v, bi, tv, ti);
        }
    }
    long long *sm_i = reinterpret_cast<long long *>(lds_raw);
    float *sm_v = reinterpret_cast<float *>(lds_raw + 32);
    best_block_reduce(v, bi, sm_v, sm_i);
    if (threadIdx.x == 0) {
        blocks[blockIdx.x].value = v;
        blocks[blockIdx.x].index = bi >= 0 ? bi + index_base : bi;
        blocks[blockIdx.x].found = bi >= 0;
    }
}

int launch_argmax_blocks_flat(lm_hip_ctx *ctx, hipStream_t stream, const float *d_scores, unsigned long long ncells,
                              long long index_base, unsigned grid, ArgmaxRecord *d_blocks)
{
    (void)ctx;
    hipLaunchKernelGGL(argmax_flat, dim3(grid), dim3(kBlock), 64, stream, d_scores, ncells, index_base, d_blocks);
    LM_HIP_TRY(hipGetLastError());
    return LM_HIP_OK;
}

// Padded rows (stride > cols): index = row * cols + col.
__global__ __launch_bounds__(kBlock) void argmax_strided(const float *__restrict__ s,
                                                         const unsigned long long rows,
                                                         const unsigned long long stride,
                                                         const unsigned cols,
                                                         ArgmaxRecord *__restrict__ blocks)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    float v = -INFINITY;
    long long bi = -1;
    const unsigned long long ncells = rows * cols;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < ncells;
         i += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long r = i / cols;
        const float x = s[r * stride + (i - r * cols)];
        if (x >= v) {
            v = x;
            bi = (long long)i;
        }
    }
    long long *sm_i = reinterpret_cast<long long *>(lds_raw);
    float *sm_v = reinterpret_cast<float *>(lds_raw + 32);
    best_block_reduce(v, bi, sm_v, sm_i);
    if (threadIdx.x == 0) {
        blocks[blockIdx.x].value = v;
        blocks[blockIdx.x].index = bi;
        blocks[blockIdx.x].found = bi >= 0;
    }
}

// Enqueues the argmax of a materialised matrix; the record lands at `d_out` (device memory, or
// pinned host memory the device can write).  No synchronisation.
int launch_argmax_device(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride, size_t cols,
                         int first_cell_rule, ArgmaxRecord *d_out)
{
    const unsigned long long ncells = (unsigned long long)rows * cols;
    const bool flat = stride == cols && (reinterpret_cast<uintptr_t>(d_scores) % 16 == 0);
    // flat: one trip of eight loads per lane and workgroup up to 65 536 workgroups, longer spans beyond;
    // strided: a grid-stride walk cell by cell
    const unsigned grid = (unsigned)std::max<unsigned long long>(
        flat ? std::min<unsigned long long>((ncells + kArgmaxSpan - 1) / kArgmaxSpan, kArgmaxMaxGrid)
             : std::min<unsigned long long>((ncells / 4 + kBlock - 1) / kBlock, (unsigned long long)ctx->num_cus * 16),
        1);
    // (+256: finalize_argmax_materialised folds more than 4096 records through 256 records behind them)
    LM_TRY(ctx->scratch.reserve(sizeof(ArgmaxRecord) * ((size_t)grid + 1 + 256)));
    ArgmaxRecord *recs = static_cast<ArgmaxRecord *>(ctx->scratch.ptr);
    if (flat)
        hipLaunchKernelGGL(argmax_flat, dim3(grid), dim3(kBlock), 64, ctx->stream, d_scores, ncells,
                           0ll, recs + 1);
    else
        hipLaunchKernelGGL(argmax_strided, dim3(grid), dim3(kBlock), 64, ctx->stream, d_scores,
                           (unsigned long long)rows, (unsigned long long)stride, (unsigned)cols,
                           recs + 1);
    LM_HIP_TRY(hipGetLastError());
    return finalize_argmax_materialised(ctx, recs + 1, grid, d_scores, first_cell_rule, d_out);
}

int launch_argmax(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride, size_t cols,
                  int first_cell_rule, ArgmaxRecord *out)
{
    // the finalize kernel writes the record straight into pinned host memory
    LM_TRY(launch_argmax_device(ctx, d_scores, rows, stride, cols, first_cell_rule,
                                static_cast<ArgmaxRecord *>(ctx->pinned)));
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = *static_cast<const ArgmaxRecord *>(ctx->pinned);
    return LM_HIP_OK;
}

// ---- threshold: count -> scan -> ordered fill -------------------------------------------

constexpr int kChunk = 4096;               // cells per workgroup
constexpr int kPerThread = kChunk / kBlock; // 16 consecutive cells per thread

__device__ __forceinline__ float cell_at(const float *__restrict__ s, unsigned long long e,
                                         unsigned long long stride, unsigned cols, bool flat)
{
    if (flat)
        return s[e];
    const unsigned long long r = e / cols;
    return s[r * stride + (e - r * cols)];
}

// Loads the thread's 16 consecutive cells (row-major rank e0..e0+15) as a hit mask.
__device__ __forceinline__ unsigned hit_mask(const float *__restrict__ s, unsigned long long e0,
                                             unsigned long long ncells, unsigned long long stride,
                                             unsigned cols, bool flat, float t)
{
    unsigned mask = 0;
    if (flat && e0 + kPerThread <= ncells) {
        const float4 *p = reinterpret_cast<const float4 *>(s + e0);
#pragma unroll
        for (int q = 0; q < kPerThread / 4; ++q) {
            const float4 x = p[q];
            mask |= (unsigned)(x.x >= t) << (4 * q + 0);
            mask |= (unsigned)(x.y >= t) << (4 * q + 1);
            mask |= (unsigned)(x.z >= t) << (4 * q + 2);
            mask |= (unsigned)(x.w >= t) << (4 * q + 3);
        }
    } else {
#pragma unroll 4
        for (int q = 0; q < kPerThread; ++q)
            if (e0 + q < ncells)
                mask |= (unsigned)(cell_at(s, e0 + q, stride, cols, flat) >= t) << q;
    }
    return mask;
}

__global__ __launch_bounds__(kBlock) void threshold_count(const float *__restrict__ s,
                                                          const unsigned long long ncells,
                                                          const unsigned long long stride,
                                                          const unsigned cols, const int flat,
                                                          const float t,
                                                          unsigned *__restrict__ counts)
{
    __shared__ unsigned sm[kBlock / 64];
    const unsigned long long c0 = (unsigned long long)blockIdx.x * kChunk;
    unsigned c = 0;
    if (flat && c0 + kChunk <= ncells) {
        // a count does not care which lane sees which cell: lane-contiguous 16-byte loads (1 KB per wavefront
        // and instruction, non-temporal) in place of each lane's own 64 bytes -- 0.61 -> 0.57 ms per 4 GB
        // (tools/kbench/read_bench.hip)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 *p = reinterpret_cast<const f32x4 *>(s + c0) + threadIdx.x;
        f32x4 x[kPerThread / 4];
#pragma unroll
        for (int q = 0; q < kPerThread / 4; ++q)
            x[q] = __builtin_nontemporal_load(p + q * kBlock);
#pragma unroll
        for (int q = 0; q < kPerThread / 4; ++q)
            c += (unsigned)(x[q].x >= t) + (unsigned)(x[q].y >= t) + (unsigned)(x[q].z >= t) + (unsigned)(x[q].w >= t);
    } else {
        c = __popc(hit_mask(s, c0 + threadIdx.x * kPerThread, ncells, stride, cols, flat, t));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0)
        sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// Exclusive scan of `counts` in tiles of 1024: level 1 writes per-chunk offsets
// relative to the tile and the tile totals; level 2 scans the totals.

__device__ __forceinline__ unsigned long long block_exclusive_scan_1024(unsigned long long x,
                                                                        unsigned long long *total)
{
    __shared__ unsigned long long wsum[kScanTile / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long y = __shfl_up(incl, off);
        if (lane >= off)
            incl += y;
    }
    if (lane == 63)
        wsum[wave] = incl;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (int w = 0; w < kScanTile / 64; ++w) {
        if (w < wave)
            base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + incl - x;
}

__global__ __launch_bounds__(kScanTile) void scan_level1(const unsigned *__restrict__ counts,
                                                         const unsigned long long n,
                                                         unsigned long long *__restrict__ offsets,
                                                         unsigned long long *__restrict__ tile_totals)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * kScanTile + threadIdx.x;
    const unsigned long long x = i < n ? counts[i] : 0;
    unsigned long long total;
    const unsigned long long ex = block_exclusive_scan_1024(x, &total);
    if (i < n)
        offsets[i] = ex;
    if (threadIdx.x == 0)
        tile_totals[blockIdx.x] = total;
}

// Single block: exclusive scan of the tile totals (in place) + grand total.
__global__ __launch_bounds__(kScanTile) void scan_level2(unsigned long long *__restrict__ tile_totals,
                                                         const unsigned long long ntiles,
                                                         unsigned long long *__restrict__ grand_total)
{
    unsigned long long carry = 0;
    for (unsigned long long base = 0; base < ntiles; base += kScanTile) {
        const unsigned long long i = base + threadIdx.x;
        const unsigned long long x = i < ntiles ? tile_totals[i] : 0;
        unsigned long long total;
        const unsigned long long ex = block_exclusive_scan_1024(x, &total);
        if (i < ntiles)
            tile_totals[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0)
        *grand_total = carry;
}

// Both levels in one workgroup, for short inputs (a p = 1e-5 hit list has ~1 500 buckets):
// tile after tile with a running carry; same outputs as scan_level1 + scan_level2.
constexpr unsigned long long kScanSmallTiles = 3;  // beyond that the sequential tiles cost more than a launch
__global__ __launch_bounds__(kScanTile) void scan_small(const unsigned *__restrict__ counts,
                                                        const unsigned long long n,
                                                        unsigned long long *__restrict__ offsets,
                                                        unsigned long long *__restrict__ tile_totals,
                                                        unsigned long long *__restrict__ grand_total)
{
    unsigned long long carry = 0;
    for (unsigned long long base = 0, t = 0; base < n; base += kScanTile, ++t) {
        const unsigned long long i = base + threadIdx.x;
        const unsigned long long x = i < n ? counts[i] : 0;
        unsigned long long total;
        const unsigned long long ex = block_exclusive_scan_1024(x, &total);
        if (i < n)
            offsets[i] = ex;
        if (threadIdx.x == 0)
            tile_totals[t] = carry;
        carry += total;
    }
    if (threadIdx.x == 0)
        *grand_total = carry;
}

// One workgroup looks at kBlock consecutive chunks: their counts are loaded with one
// coalesced read, and only chunks that contain hits (rare: a p = 1e-5 tail touches
// ~4 % of the 4096-cell chunks) are re-read and compacted, in chunk order.
__global__ __launch_bounds__(kBlock) void threshold_fill(
    const float *__restrict__ s, const unsigned long long ncells, const unsigned long long stride,
    const unsigned cols, const int flat, const float t, const unsigned *__restrict__ counts,
    const unsigned long long nchunks, const unsigned long long *__restrict__ offsets,
    const unsigned long long *__restrict__ tile_offsets, lm_hip_coords *__restrict__ out)
{
    __shared__ unsigned sm[kBlock / 64];
    __shared__ unsigned list[kBlock];
    __shared__ unsigned nlist;
    if (threadIdx.x == 0)
        nlist = 0;
    __syncthreads();
    const unsigned long long my_chunk = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (my_chunk < nchunks && counts[my_chunk] != 0)
        list[atomicAdd(&nlist, 1u)] = threadIdx.x;
    __syncthreads();
    const unsigned n = nlist;
    for (unsigned li = 0; li < n; ++li) {
        const unsigned long long chunk = (unsigned long long)blockIdx.x * kBlock + list[li];
        const unsigned long long e0 = chunk * kChunk + threadIdx.x * kPerThread;
        unsigned mask = hit_mask(s, e0, ncells, stride, cols, flat, t);
        const unsigned c = __popc(mask);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned y = __shfl_up(incl, off);
            if (lane >= off)
                incl += y;
        }
        __syncthreads();  // previous iteration's readers of sm[] are done
        if (lane == 63)
            sm[wave] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; ++w)
            base += sm[w];
        unsigned long long pos = tile_offsets[chunk / kScanTile] + offsets[chunk] + base + incl - c;
        while (mask) {
            const int q = __ffs(mask) - 1;
            mask &= mask - 1;
            const unsigned long long e = e0 + q;
            const unsigned long long r = e / cols;
            out[pos].row = r;
            out[pos].col = e - r * cols;
            ++pos;
        }
    }
}

int launch_scan_u32(lm_hip_ctx *ctx, const unsigned *counts, unsigned long long n,
                    unsigned long long *offsets, unsigned long long *tiles,
                    unsigned long long *total)
{
    const unsigned long long ntiles = (n + kScanTile - 1) / kScanTile;
    if (ntiles <= kScanSmallTiles) {
        hipLaunchKernelGGL(scan_small, dim3(1), dim3(kScanTile), 0, ctx->stream, counts, n, offsets, tiles, total);
        LM_HIP_TRY(hipGetLastError());
        return LM_HIP_OK;
    }
    hipLaunchKernelGGL(scan_level1, dim3((unsigned)ntiles), dim3(kScanTile), 0, ctx->stream, counts, n,
                       offsets, tiles);
    hipLaunchKernelGGL(scan_level2, dim3(1), dim3(kScanTile), 0, ctx->stream, tiles, ntiles, total);
    LM_HIP_TRY(hipGetLastError());
    return LM_HIP_OK;
}

int launch_threshold(lm_hip_ctx *ctx, const float *d_scores, size_t rows, size_t stride,
                     size_t cols, float t, lm_hip_coords **coords, size_t *n)
{
    *coords = nullptr;
    *n = 0;
    const unsigned long long ncells = (unsigned long long)rows * cols;
    if (ncells == 0)
        return LM_HIP_OK;
    const unsigned long long nchunks = (ncells + kChunk - 1) / kChunk;
    const unsigned long long ntiles = (nchunks + kScanTile - 1) / kScanTile;
    const int flat = stride == cols && (reinterpret_cast<uintptr_t>(d_scores) % 16 == 0);
    // scratch: counts u32[nchunks] | offsets u64[nchunks] | tile totals u64[ntiles] | total u64
    const size_t off_counts = 0;
    const size_t off_offsets = (nchunks * 4 + 15) / 16 * 16;
    const size_t off_tiles = off_offsets + nchunks * 8;
    const size_t off_total = off_tiles + ntiles * 8;
    LM_TRY(ctx->scratch.reserve(off_total + 16));
    char *base = static_cast<char *>(ctx->scratch.ptr);
    unsigned *counts = reinterpret_cast<unsigned *>(base + off_counts);
    unsigned long long *offsets = reinterpret_cast<unsigned long long *>(base + off_offsets);
    unsigned long long *tiles = reinterpret_cast<unsigned long long *>(base + off_tiles);
    unsigned long long *total = reinterpret_cast<unsigned long long *>(base + off_total);

    hipLaunchKernelGGL(threshold_count, dim3((unsigned)nchunks), dim3(kBlock), 0, ctx->stream,
                       d_scores, ncells, (unsigned long long)stride, (unsigned)cols, flat, t, counts);
    LM_TRY(launch_scan_u32(ctx, counts, nchunks, offsets, tiles, total));
    LM_HIP_TRY(hipMemcpyAsync(ctx->pinned, total, 8, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const unsigned long long count = *static_cast<unsigned long long *>(ctx->pinned);
    if (count == 0)
        return LM_HIP_OK;

    lm_hip_coords *host = static_cast<lm_hip_coords *>(result_alloc(count * sizeof(lm_hip_coords)));
    if (!host)
        return fail(LM_HIP_ERR_OOM, "threshold: cannot allocate %llu hits on the host", count);
    int st = ctx->scratch2.reserve(count * sizeof(lm_hip_coords));
    if (st != LM_HIP_OK) {
        result_free(host);
        return st;
    }
    lm_hip_coords *d_out = static_cast<lm_hip_coords *>(ctx->scratch2.ptr);
    hipLaunchKernelGGL(threshold_fill, dim3((unsigned)((nchunks + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, ctx->stream, d_scores, ncells, (unsigned long long)stride,
                       (unsigned)cols, flat, t, counts, nchunks, offsets, tiles, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(host, d_out, count * sizeof(lm_hip_coords), hipMemcpyDeviceToHost,
                           ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        result_free(host);
        return fail(LM_HIP_ERR_HIP, "threshold fill failed: %s", hipGetErrorString(e));
    }
    *coords = host;
    *n = (size_t)count;
    return LM_HIP_OK;
}

}  // namespace lm
