// discrete.hip -- Maximum<u8, C> / Threshold<u8, C> over a materialised u8 StripedScores
// matrix (the scores of a DiscreteMatrix, score_u8.hpp).
//
// Same Generic bodies as the f32 reductions of reduce.hip, instantiated for u8:
//   argmax    lightmotif/src/pli/mod.rs:135-155  `x >= best` in a row-major scan => the
//             maximal cell that is LAST in (row, col) order (u8 has no NaN)
//   max       pli/mod.rs:158-160                 value at argmax
//   threshold pli/mod.rs:210-221                 every cell with x >= t in row-major order
// The whole rows x cols matrix is scanned, padded tail included.  Streaming reads of one
// byte per cell: the roofline is HBM.
#include <algorithm>

#include "score_kernels.hpp"

namespace lm {

__global__ void argmax_fold(const ArgmaxRecord *__restrict__ blocks, const unsigned nblocks,
                            ArgmaxRecord *__restrict__ out);  // score_store.hip

namespace {

constexpr int kCellsPerThread = 16;                // one 16-byte load
constexpr int kChunkU8 = kBlock * kCellsPerThread;  // cells per workgroup

__device__ __forceinline__ unsigned cell_u8(const uint8_t *__restrict__ s, unsigned long long e,
                                            unsigned long long stride, unsigned cols, bool flat)
{
    if (flat)
        return s[e];
    const unsigned long long r = e / cols;
    return s[r * stride + (e - r * cols)];
}

// The thread's 16 consecutive cells (row-major ranks e0 .. e0 + 15), zero past the end.
__device__ __forceinline__ void load_cells(const uint8_t *__restrict__ s, unsigned long long e0,
                                           unsigned long long ncells, unsigned long long stride,
                                           unsigned cols, bool flat, unsigned (&w)[4], unsigned &valid)
{
    if (flat && e0 + kCellsPerThread <= ncells) {
        const uint4 v = *reinterpret_cast<const uint4 *>(s + e0);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        valid = kCellsPerThread;
        return;
    }
    w[0] = w[1] = w[2] = w[3] = 0;
    valid = 0;
    for (int q = 0; q < kCellsPerThread; ++q)
        if (e0 + q < ncells) {
            w[q / 4] |= cell_u8(s, e0 + q, stride, cols, flat) << (8 * (q % 4));
            valid = q + 1;
        }
}

__global__ __launch_bounds__(kBlock) void argmax_u8(const uint8_t *__restrict__ s,
                                                    const unsigned long long ncells,
                                                    const unsigned long long stride, const unsigned cols,
                                                    const int flat, ArgmaxRecord *__restrict__ blocks)
{
    __shared__ float sm_v[kBlock / 64];
    __shared__ long long sm_i[kBlock / 64];
    int best = -1;
    long long bi = -1;
    for (unsigned long long e0 = ((unsigned long long)blockIdx.x * kBlock + threadIdx.x) * kCellsPerThread;
         e0 < ncells; e0 += (unsigned long long)gridDim.x * kChunkU8) {
        unsigned w[4], valid;
        load_cells(s, e0, ncells, stride, cols, flat != 0, w, valid);
        // ascending index order inside the thread, so `>=` keeps the later cell
#pragma unroll
        for (int q = 0; q < kCellsPerThread; ++q) {
            const int x = (int)((w[q / 4] >> (8 * (q % 4))) & 0xffu);
            if (q < (int)valid && x >= best) {
                best = x;
                bi = (long long)(e0 + q);
            }
        }
    }
    float v = (float)best;
    best_block_reduce(v, bi, sm_v, sm_i);
    if (threadIdx.x == 0) {
        blocks[blockIdx.x].value = v;
        blocks[blockIdx.x].index = bi;
        blocks[blockIdx.x].found = bi >= 0;
    }
}

__device__ __forceinline__ unsigned hit_mask_u8(const unsigned (&w)[4], unsigned valid, unsigned t)
{
    unsigned mask = 0;
#pragma unroll
    for (int q = 0; q < kCellsPerThread; ++q)
        mask |= (unsigned)(q < (int)valid && ((w[q / 4] >> (8 * (q % 4))) & 0xffu) >= t) << q;
    return mask;
}

__global__ __launch_bounds__(kBlock) void threshold_count_u8(const uint8_t *__restrict__ s,
                                                             const unsigned long long ncells,
                                                             const unsigned long long stride,
                                                             const unsigned cols, const int flat,
                                                             const unsigned t, unsigned *__restrict__ counts)
{
    __shared__ unsigned sm[kBlock / 64];
    const unsigned long long e0 = (unsigned long long)blockIdx.x * kChunkU8 + threadIdx.x * kCellsPerThread;
    unsigned w[4], valid;
    load_cells(s, e0, ncells, stride, cols, flat != 0, w, valid);
    unsigned c = __popc(hit_mask_u8(w, valid, t));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0)
        sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int i = 0; i < kBlock / 64; ++i)
            tot += sm[i];
        counts[blockIdx.x] = tot;
    }
}

// One workgroup per chunk with hits: thread order IS row-major order, so an exclusive scan of
// the per-thread hit counts gives every hit its slot.
__global__ __launch_bounds__(kBlock) void threshold_fill_u8(
    const uint8_t *__restrict__ s, const unsigned long long ncells, const unsigned long long stride,
    const unsigned cols, const int flat, const unsigned t, const unsigned *__restrict__ counts,
    const unsigned long long *__restrict__ offsets, const unsigned long long *__restrict__ tile_offsets,
    lm_hip_coords *__restrict__ out)
{
    __shared__ unsigned wave_tot[kBlock / 64];
    if (counts[blockIdx.x] == 0)  // block-uniform
        return;
    const unsigned long long e0 = (unsigned long long)blockIdx.x * kChunkU8 + threadIdx.x * kCellsPerThread;
    unsigned w[4], valid;
    load_cells(s, e0, ncells, stride, cols, flat != 0, w, valid);
    const unsigned mask = hit_mask_u8(w, valid, t);
    const unsigned mine = __popc(mask);
    unsigned incl = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned y = __shfl_up(incl, off);
        if (lane >= off)
            incl += y;
    }
    if (lane == 63)
        wave_tot[wave] = incl;
    __syncthreads();
    unsigned long long slot = tile_offsets[blockIdx.x / kScanTile] + offsets[blockIdx.x] + incl - mine;
    for (int i = 0; i < wave; ++i)
        slot += wave_tot[i];
    for (unsigned m = mask; m; m &= m - 1) {
        const unsigned long long e = e0 + (unsigned)(__ffs((int)m) - 1);
        lm_hip_coords c;
        c.row = e / cols;
        c.col = e - c.row * cols;
        out[slot++] = c;
    }
}

}  // namespace

int launch_argmax_u8(lm_hip_ctx *ctx, const uint8_t *d_scores, size_t rows, size_t stride, size_t cols,
                     ArgmaxRecord *out)
{
    const unsigned long long ncells = (unsigned long long)rows * cols;
    const unsigned grid = (unsigned)std::max<unsigned long long>(
        std::min<unsigned long long>((ncells + kChunkU8 - 1) / kChunkU8, (unsigned long long)ctx->num_cus * 16), 1);
    LM_TRY(ctx->scratch.reserve(sizeof(ArgmaxRecord) * (size_t)grid));
    ArgmaxRecord *recs = static_cast<ArgmaxRecord *>(ctx->scratch.ptr);
    const int flat = stride == cols && (reinterpret_cast<uintptr_t>(d_scores) % 16 == 0);
    hipLaunchKernelGGL(argmax_u8, dim3(grid), dim3(kBlock), 0, ctx->stream, d_scores, ncells,
                       (unsigned long long)stride, (unsigned)cols, flat, recs);
    // one workgroup folds the records straight into pinned host memory
    hipLaunchKernelGGL(argmax_fold, dim3(1), dim3(kBlock), 0, ctx->stream, recs, grid,
                       static_cast<ArgmaxRecord *>(ctx->pinned));
    LM_HIP_TRY(hipGetLastError());
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = *static_cast<const ArgmaxRecord *>(ctx->pinned);
    return LM_HIP_OK;
}

int launch_threshold_u8(lm_hip_ctx *ctx, const uint8_t *d_scores, size_t rows, size_t stride, size_t cols,
                        unsigned t, lm_hip_coords **coords, size_t *n)
{
    *coords = nullptr;
    *n = 0;
    const unsigned long long ncells = (unsigned long long)rows * cols;
    if (ncells == 0)
        return LM_HIP_OK;
    const unsigned long long nchunks = (ncells + kChunkU8 - 1) / kChunkU8;
    const unsigned long long ntiles = (nchunks + kScanTile - 1) / kScanTile;
    const int flat = stride == cols && (reinterpret_cast<uintptr_t>(d_scores) % 16 == 0);
    // scratch: counts u32[nchunks] | offsets u64[nchunks] | tile totals u64[ntiles] | total u64
    const size_t off_offsets = (nchunks * 4 + 15) / 16 * 16;
    const size_t off_tiles = off_offsets + nchunks * 8;
    const size_t off_total = off_tiles + ntiles * 8;
    LM_TRY(ctx->scratch.reserve(off_total + 16));
    char *base = static_cast<char *>(ctx->scratch.ptr);
    unsigned *counts = reinterpret_cast<unsigned *>(base);
    unsigned long long *offsets = reinterpret_cast<unsigned long long *>(base + off_offsets);
    unsigned long long *tiles = reinterpret_cast<unsigned long long *>(base + off_tiles);
    unsigned long long *total = reinterpret_cast<unsigned long long *>(base + off_total);
    hipLaunchKernelGGL(threshold_count_u8, dim3((unsigned)nchunks), dim3(kBlock), 0, ctx->stream, d_scores,
                       ncells, (unsigned long long)stride, (unsigned)cols, flat, t, counts);
    LM_TRY(launch_scan_u32(ctx, counts, nchunks, offsets, tiles, total));
    LM_HIP_TRY(hipMemcpyAsync(ctx->pinned, total, 8, hipMemcpyDeviceToHost, ctx->stream));
    LM_HIP_TRY(hipStreamSynchronize(ctx->stream));
    const unsigned long long count = *static_cast<unsigned long long *>(ctx->pinned);
    if (count == 0)
        return LM_HIP_OK;
    lm_hip_coords *host = static_cast<lm_hip_coords *>(result_alloc(count * sizeof(lm_hip_coords)));
    if (!host)
        return fail(LM_HIP_ERR_OOM, "threshold_u8: cannot allocate %llu hits on the host", count);
    int st = ctx->scratch2.reserve(count * sizeof(lm_hip_coords));
    if (st != LM_HIP_OK) {
        result_free(host);
        return st;
    }
    lm_hip_coords *d_out = static_cast<lm_hip_coords *>(ctx->scratch2.ptr);
    hipLaunchKernelGGL(threshold_fill_u8, dim3((unsigned)nchunks), dim3(kBlock), 0, ctx->stream, d_scores,
                       ncells, (unsigned long long)stride, (unsigned)cols, flat, t, counts, offsets, tiles, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(host, d_out, count * sizeof(lm_hip_coords), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        result_free(host);
        return fail(LM_HIP_ERR_HIP, "threshold_u8 fill failed: %s", hipGetErrorString(e));
    }
    *coords = host;
    *n = (size_t)count;
    return LM_HIP_OK;
}

}  // namespace lm
