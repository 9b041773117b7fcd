#include "scan.hpp"

namespace cship
{

namespace
{
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE  = SCAN_BLOCK * SCAN_ITEMS;

__global__ __launch_bounds__(SCAN_BLOCK) void blockSumKernel(const uint32_t* __restrict__ in, size_t n,
                                                             uint32_t* __restrict__ sums)
{
    __shared__ uint32_t ws[4];
    size_t base = size_t(blockIdx.x) * SCAN_TILE + size_t(threadIdx.x) * SCAN_ITEMS;
    uint32_t s  = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s += in[base + k];
    uint32_t total;
    blockExclusiveScan256(s, ws, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

//! single workgroup: sums[0..m) -> exclusive scan in place (+init); grand total to *totalOut
__global__ __launch_bounds__(SCAN_BLOCK) void scanSumsKernel(uint32_t* __restrict__ sums, unsigned m, uint32_t init,
                                                             uint32_t* __restrict__ totalOut)
{
    __shared__ uint32_t ws[4];
    uint32_t carry = init;
    for (unsigned base = 0; base < m; base += SCAN_BLOCK)
    {
        unsigned i = base + threadIdx.x;
        uint32_t v = i < m ? sums[i] : 0u;
        uint32_t total;
        uint32_t ex = blockExclusiveScan256(v, ws, &total);
        if (i < m) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0 && totalOut) *totalOut = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void blockScanKernel(const uint32_t* __restrict__ in,
                                                              uint32_t* __restrict__ out, size_t n,
                                                              const uint32_t* __restrict__ offsets, bool inclusive)
{
    __shared__ uint32_t ws[4];
    size_t base = size_t(blockIdx.x) * SCAN_TILE + size_t(threadIdx.x) * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
    {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        s += v[k];
    }
    uint32_t run = offsets[blockIdx.x] + blockExclusiveScan256(s, ws, nullptr);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
    {
        if (base + k < n) out[base + k] = inclusive ? run + v[k] : run;
        run += v[k];
    }
}
} // namespace

int scanU32(cstone_hip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t init, bool inclusive,
            uint32_t* totalOut)
{
    if (n == 0)
    {
        // nothing to scan: the grand total is the initial value (written by the sums kernel over zero blocks)
        if (totalOut) hipLaunchKernelGGL(scanSumsKernel, 1, SCAN_BLOCK, 0, ctx->stream, (uint32_t*)nullptr, 0u, init, totalOut);
        CS_HIP(ctx, hipGetLastError());
        return CSTONE_OK;
    }
    unsigned blocks = unsigned((n + SCAN_TILE - 1) / SCAN_TILE);
    // block sums live in a private slice at the END of the arena so callers may hold arena slices of their own
    size_t bytes = alignUp(size_t(blocks) * sizeof(uint32_t));
    auto* sums   = (uint32_t*)arenaTake(ctx, bytes);
    if (!sums) return fail(ctx, CSTONE_E_INTERNAL, "scan: arena exhausted (caller must reserve %zu extra bytes)", bytes);
    hipLaunchKernelGGL(blockSumKernel, blocks, SCAN_BLOCK, 0, ctx->stream, in, n, sums);
    hipLaunchKernelGGL(scanSumsKernel, 1, SCAN_BLOCK, 0, ctx->stream, sums, blocks, init, totalOut);
    hipLaunchKernelGGL(blockScanKernel, blocks, SCAN_BLOCK, 0, ctx->stream, in, out, n, sums, inclusive);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

//! bytes scanU32 takes from the arena for n elements
size_t scanArenaBytes(size_t n) { return alignUp(((n + SCAN_TILE - 1) / SCAN_TILE) * sizeof(uint32_t)) + 256; }

} // namespace cship
