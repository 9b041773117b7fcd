#include "scan.hpp"

namespace cship
{

namespace
{
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE  = SCAN_BLOCK * SCAN_ITEMS;

//! up to two independent scans of equally long arrays in one set of launches (blockIdx.y picks the array)
struct ScanJobs
{
    const uint32_t* in[2];
    uint32_t* out[2];
    uint32_t* sums[2];
    uint32_t* total[2];
    uint32_t init[2];
};

__global__ __launch_bounds__(SCAN_BLOCK) void blockSumKernel(ScanJobs jobs, size_t n)
{
    __shared__ uint32_t ws[4];
    const uint32_t* __restrict__ in = jobs.in[blockIdx.y];
    size_t base = size_t(blockIdx.x) * SCAN_TILE + size_t(threadIdx.x) * SCAN_ITEMS;
    uint32_t s  = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s += in[base + k];
    uint32_t total;
    blockExclusiveScan256(s, ws, &total);
    if (threadIdx.x == 0) jobs.sums[blockIdx.y][blockIdx.x] = total;
}

//! one workgroup per array: sums[0..m) -> exclusive scan in place (+init); grand total to *total
__global__ __launch_bounds__(SCAN_BLOCK) void scanSumsKernel(ScanJobs jobs, unsigned m)
{
    __shared__ uint32_t ws[4];
    uint32_t* __restrict__ sums = jobs.sums[blockIdx.x];
    uint32_t carry              = jobs.init[blockIdx.x];
    for (unsigned base = 0; base < m; base += SCAN_BLOCK)
    {
        unsigned i = base + threadIdx.x;
        uint32_t v = i < m ? sums[i] : 0u;
        uint32_t total;
        uint32_t ex = blockExclusiveScan256(v, ws, &total);
        if (i < m) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0 && jobs.total[blockIdx.x]) *jobs.total[blockIdx.x] = carry;
}

/*! SUMS_SCANNED: sums[] holds the exclusive scan of the tile sums (scanSumsKernel ran); otherwise the raw tile sums, and
 *  every workgroup adds up those of the tiles before its own (few tiles: one launch less; the last workgroup also
 *  leaves the grand total) */
template<bool SUMS_SCANNED>
__global__ __launch_bounds__(SCAN_BLOCK) void blockScanKernel(ScanJobs jobs, size_t n, bool inclusive)
{
    __shared__ uint32_t ws[4];
    const uint32_t* in = jobs.in[blockIdx.y]; // in == out is allowed
    uint32_t* out      = jobs.out[blockIdx.y];
    size_t base = size_t(blockIdx.x) * SCAN_TILE + size_t(threadIdx.x) * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
    {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        s += v[k];
    }
    uint32_t before;
    if constexpr (SUMS_SCANNED) { before = jobs.sums[blockIdx.y][blockIdx.x]; }
    else
    {
        const uint32_t* __restrict__ sums = jobs.sums[blockIdx.y];
        uint32_t mine = 0;
        for (unsigned t = threadIdx.x; t < blockIdx.x; t += SCAN_BLOCK)
            mine += sums[t];
        uint32_t tilesBefore;
        blockExclusiveScan256(mine, ws, &tilesBefore);
        __syncthreads(); // (ws is used again below)
        before = jobs.init[blockIdx.y] + tilesBefore;
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && jobs.total[blockIdx.y])
            *jobs.total[blockIdx.y] = before + sums[blockIdx.x];
    }
    uint32_t run = before + blockExclusiveScan256(s, ws, nullptr);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
    {
        if (base + k < n) out[base + k] = inclusive ? run + v[k] : run;
        run += v[k];
    }
}
} // namespace

static int scanJobs(cstone_hip_ctx* ctx, ScanJobs jobs, int count, size_t n, bool inclusive)
{
    // (Round 4 tried ONE launch per scan -- tiles chained by decoupled look-back, status words told apart by a generation
    //  tag instead of being cleared: no gain at any size that occurs here (1.10 against 1.09 ms per multi-rank sync at
    //  1.25e7 particles; at 2.2e6 elements a thousand tiles that start together spin on each other's words: 0.1 ms
    //  against 0.03 ms), and the reference's Domain<GpuTag> on the shim lost particles with it although the scans pass
    //  their own fuzz test -- taken out again)
    if (n == 0)
    {
        // nothing to scan: the grand totals are the initial values (written by the sums kernel over zero blocks)
        hipLaunchKernelGGL(scanSumsKernel, count, SCAN_BLOCK, 0, ctx->stream, jobs, 0u);
        CS_HIP(ctx, hipGetLastError());
        return CSTONE_OK;
    }
    unsigned blocks = unsigned((n + SCAN_TILE - 1) / SCAN_TILE);
    // block sums live in private slices at the END of the arena so callers may hold arena slices of their own
    size_t bytes = alignUp(size_t(blocks) * sizeof(uint32_t));
    for (int j = 0; j < count; ++j)
    {
        jobs.sums[j] = (uint32_t*)arenaTake(ctx, bytes);
        if (!jobs.sums[j])
            return fail(ctx, CSTONE_E_INTERNAL, "scan: arena exhausted (caller must reserve %zu extra bytes)", bytes * count);
    }
    hipLaunchKernelGGL(blockSumKernel, dim3(blocks, count), SCAN_BLOCK, 0, ctx->stream, jobs, n);
    // up to 1024 tiles (2e6 elements) every workgroup sums the tile sums before its own itself (at most four loads per
    // lane): two launches instead of three for the scans over leaves and tiles of a sync
    if (blocks <= 4 * SCAN_BLOCK)
        hipLaunchKernelGGL(blockScanKernel<false>, dim3(blocks, count), SCAN_BLOCK, 0, ctx->stream, jobs, n, inclusive);
    else
    {
        hipLaunchKernelGGL(scanSumsKernel, count, SCAN_BLOCK, 0, ctx->stream, jobs, blocks);
        hipLaunchKernelGGL(blockScanKernel<true>, dim3(blocks, count), SCAN_BLOCK, 0, ctx->stream, jobs, n, inclusive);
    }
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int scanU32(cstone_hip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t init, bool inclusive,
            uint32_t* totalOut)
{
    ScanJobs jobs{};
    jobs.in[0] = in, jobs.out[0] = out, jobs.total[0] = totalOut, jobs.init[0] = init;
    return scanJobs(ctx, jobs, 1, n, inclusive);
}

int scanU32Pair(cstone_hip_ctx* ctx, const uint32_t* inA, uint32_t* outA, const uint32_t* inB, uint32_t* outB, size_t n)
{
    ScanJobs jobs{};
    jobs.in[0] = inA, jobs.out[0] = outA, jobs.in[1] = inB, jobs.out[1] = outB;
    return scanJobs(ctx, jobs, 2, n, false);
}

//! bytes scanU32 takes from the arena for n elements
size_t scanArenaBytes(size_t n) { return alignUp(((n + SCAN_TILE - 1) / SCAN_TILE) * sizeof(uint32_t)) + 256; }

} // namespace cship
