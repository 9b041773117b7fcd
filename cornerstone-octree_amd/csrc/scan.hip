#include <algorithm>
#include <cstdlib>

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
    uint32_t run = jobs.sums[blockIdx.y][blockIdx.x] + blockExclusiveScan256(s, ws, nullptr);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
    {
        if (base + k < n) out[base + k] = inclusive ? run + v[k] : run;
        run += v[k];
    }
}
/*! The same scan in ONE launch (round 4): tiles chained by decoupled look-back.  A sync of the multi-rank domain runs
 *  eight scans over node arrays of 10^5..10^6 elements: at three launches each they were a quarter of its launches, and
 *  launch gaps are what such a sync is made of.  A workgroup draws a ticket (tiles are then owned in start order:
 *  forward progress under any dispatch order), scans its tile, publishes the tile's sum and walks back over its
 *  predecessors' status words until one of them carries an inclusive prefix.  A status word is {generation : 30, state :
 *  2, value : 32} in one 64-bit granule, read and written with RELAXED agent-scope atomics: the word carries its payload,
 *  nothing else written by a tile is read by another, so no fence is needed -- and an acquire / release at agent scope
 *  costs an invalidation / write-back of the XCD's L2 per access on this machine (measured: the leaf table's scans went
 *  from 0.05 to 0.2 ms).  The generation tells this launch's words from those of the launch before, the ticket base
 *  this launch's tickets: nothing is cleared between scans. */
constexpr unsigned long long SCAN_AGGREGATE = 1, SCAN_INCLUSIVE = 2;

__global__ __launch_bounds__(SCAN_BLOCK) void chainScanKernel(ScanJobs jobs, size_t n, bool inclusive,
                                                              unsigned long long* __restrict__ status, size_t tilesCap,
                                                              uint32_t* __restrict__ tickets, uint32_t base0,
                                                              uint32_t base1, uint32_t generation, unsigned numTiles,
                                                              int* __restrict__ errors)
{
    __shared__ uint32_t ws[4];
    __shared__ uint32_t sTile, sPrefix;
    const int job = blockIdx.y;
    if (threadIdx.x == 0) sTile = atomicAdd(&tickets[job], 1u) - (job ? base1 : base0);
    __syncthreads();
    const unsigned tile = sTile;
    const uint32_t* in  = jobs.in[job]; // in == out is allowed: a tile is read before it is written, by its own workgroup
    uint32_t* out       = jobs.out[job];
    const size_t base   = size_t(tile) * SCAN_TILE + size_t(threadIdx.x) * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
    {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        s += v[k];
    }
    uint32_t total;
    const uint32_t ex = blockExclusiveScan256(s, ws, &total);
    if (threadIdx.x < 64)
    {
        // the look-back, by the first wave: lane l reads the status word of tile - 1 - l, so one round trip covers 64
        // predecessors (walked one by one, a launch whose tiles all start together needs ~sqrt(2 tiles) dependent trips)
        const unsigned lane          = threadIdx.x;
        unsigned long long* st       = status + size_t(job) * tilesCap;
        const unsigned long long tag = (unsigned long long)(generation) << 34;
        uint32_t prefix              = jobs.init[job];
        if (tile > 0)
        {
            if (lane == 0)
                __hip_atomic_store(&st[tile], tag | (SCAN_AGGREGATE << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prefix        = 0;
            unsigned spin = 0;
            for (long long t = (long long)tile - 1;;)
            {
                const long long at = t - (long long)lane;
                // (in front of tile 0: nothing to add, and the walk ends there)
                unsigned long long w = tag | (SCAN_INCLUSIVE << 32);
                if (at >= 0) w = __hip_atomic_load(&st[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned state = unsigned(w >> 32) & 3u;
                const bool valid     = (w >> 34) == generation && state != 0;
                const uint64_t mInc  = __ballot(valid && state == SCAN_INCLUSIVE);
                const uint64_t mBad  = __ballot(!valid);
                const unsigned last  = mInc ? unsigned(__builtin_ctzll(mInc)) : 63u; // nearest inclusive word, if any
                const uint64_t need  = last >= 63u ? ~0ull : ((2ull << last) - 1ull);
                if (mBad & need)
                {
                    if (++spin > (1u << 26)) // (cannot happen: every lower tile is owned by a started workgroup)
                    {
                        if (lane == 0) atomicOr(errors, 0x400);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                uint32_t add = lane <= last ? uint32_t(w) : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
                    add += __shfl_xor(add, o);
                prefix += add;
                if (mInc) break;
                t -= 64;
            }
        }
        if (lane == 0)
        {
            __hip_atomic_store(&st[tile], tag | (SCAN_INCLUSIVE << 32) | uint32_t(prefix + total), __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_AGENT);
            sPrefix = prefix;
            if (tile + 1 == numTiles && jobs.total[job]) *jobs.total[job] = prefix + total;
        }
    }
    __syncthreads();
    uint32_t run = sPrefix + ex;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
    {
        if (base + k < n) out[base + k] = inclusive ? run + v[k] : run;
        run += v[k];
    }
}
} // namespace

static int chainScan(cstone_hip_ctx* ctx, ScanJobs jobs, int count, size_t n, bool inclusive)
{
    const unsigned tiles = unsigned((n + SCAN_TILE - 1) / SCAN_TILE);
    if (tiles > ctx->scanTilesCap || !ctx->scanTickets)
    {
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->scanStatus) CS_HIP(ctx, hipFree(ctx->scanStatus));
        ctx->scanStatus         = nullptr;
        const size_t cap        = std::max<size_t>(size_t(tiles) * 2, 4096);
        CS_HIP(ctx, hipMalloc((void**)&ctx->scanStatus, 2 * cap * sizeof(unsigned long long)));
        CS_HIP(ctx, hipMemsetAsync(ctx->scanStatus, 0, 2 * cap * sizeof(unsigned long long), ctx->stream));
        ctx->scanTilesCap = cap;
        if (!ctx->scanTickets)
        {
            CS_HIP(ctx, hipMalloc((void**)&ctx->scanTickets, 2 * sizeof(uint32_t)));
            CS_HIP(ctx, hipMemsetAsync(ctx->scanTickets, 0, 2 * sizeof(uint32_t), ctx->stream));
            ctx->scanTicketBase[0] = ctx->scanTicketBase[1] = 0;
        }
    }
    // generations 1 .. 2^30 - 1 (0 = a cleared word); on wrap-around the words are cleared once
    if (++ctx->scanGeneration >= (1u << 30))
    {
        CS_HIP(ctx, hipMemsetAsync(ctx->scanStatus, 0, 2 * ctx->scanTilesCap * sizeof(unsigned long long), ctx->stream));
        ctx->scanGeneration = 1;
    }
    hipLaunchKernelGGL(chainScanKernel, dim3(tiles, count), SCAN_BLOCK, 0, ctx->stream, jobs, n, inclusive, ctx->scanStatus,
                       ctx->scanTilesCap, ctx->scanTickets, ctx->scanTicketBase[0], ctx->scanTicketBase[1],
                       ctx->scanGeneration, tiles, ctx->devScalars + 63);
    for (int j = 0; j < count; ++j)
        ctx->scanTicketBase[j] += tiles; // (wraps like the device counter)
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

static int scanJobs(cstone_hip_ctx* ctx, ScanJobs jobs, int count, size_t n, bool inclusive)
{
    static const bool threePass = std::getenv("CSTONE_SCAN_3PASS") != nullptr; // (tuning: the three-launch formulation)
    if (n > 0 && !threePass) return chainScan(ctx, jobs, count, n, inclusive);
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
    hipLaunchKernelGGL(scanSumsKernel, count, SCAN_BLOCK, 0, ctx->stream, jobs, blocks);
    hipLaunchKernelGGL(blockScanKernel, dim3(blocks, count), SCAN_BLOCK, 0, ctx->stream, jobs, n, inclusive);
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
