// Device-wide prefix sums over uint32 (node arrays: <= a few million elements, 8 bytes/element).
// Three launches: block sums -> scan of block sums (one workgroup) -> block scans with offsets.
// Replaces thrust::exclusive_scan / inclusive_scan (R/primitives/primitives_gpu.cu:395-437).
#pragma once

#include "ctx.hpp"

namespace cship
{

//! out[i] = init + sum(in[0..i)) (exclusive) or init + sum(in[0..i]) (inclusive); in == out allowed.
//! If totalOut != nullptr, the grand total (init + sum of all) is also stored there (device pointer).
int scanU32(cstone_hip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t init, bool inclusive,
            uint32_t* totalOut = nullptr);

//! two exclusive scans (init 0) of arrays of n elements in one set of launches; arena: 2 * scanArenaBytes(n)
int scanU32Pair(cstone_hip_ctx* ctx, const uint32_t* inA, uint32_t* outA, const uint32_t* inB, uint32_t* outB, size_t n);

//! bytes scanU32 takes from the arena for n elements (callers reserve this much on top of their own slices)
size_t scanArenaBytes(size_t n);

//! wave64 inclusive scan via DPP-free shuffles
__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v, unsigned lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
        uint32_t t = __shfl_up(v, o);
        if (lane >= unsigned(o)) v += t;
    }
    return v;
}

//! workgroup (256 threads) exclusive scan of one value per thread; returns exclusive prefix, total in *total
__device__ __forceinline__ uint32_t blockExclusiveScan256(uint32_t v, uint32_t* waveSums /*LDS[4]*/, uint32_t* total)
{
    unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t inc  = waveInclusiveScan(v, lane);
    __syncthreads(); // waveSums may still be read from a previous call
    if (lane == 63) waveSums[w] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (unsigned i = 0; i < 4; ++i)
    {
        uint32_t s = waveSums[i];
        if (i < w) off += s;
        tot += s;
    }
    if (total) *total = tot;
    return off + inc - v;
}

} // namespace cship
