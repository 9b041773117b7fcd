// Neighbor search on gfx950.  Replaces findNeighbors (R/findneighbors.hpp:96-188) on the tree view of
// R/tree/octree.hpp:297-317.  Results follow the reference's CPU path: neighbors j != i with
// |r_i - r_j|^2 < (2 h_i)^2, stored in the order of its depth-first traversal (children 0..7, last
// pushed popped first), true count returned even beyond ngmax.
//
// Layout: one WAVE per 64 consecutive target particles (neighbours in SFC order, so their search spheres
// cover almost the same tree nodes) and ONE traversal per wave.  Every stack entry carries the 64-bit mask of
// the lanes whose own depth-first walk would have reached that node (own overlap test passed on the node and on
// all its ancestors); a node is visited while any lane is still interested, leaf particles are fetched once
// per wave (one coalesced load, then v_readlane broadcasts) and tested by the interested lanes.  Each lane therefore sees
// exactly the leaves, in exactly the order, of the reference's per-particle walk (the relative order of two
// leaves is decided at their lowest common ancestor and does not depend on what else is visited), so the
// stored lists are identical element for element.  The traversal itself lives in include/cstone_hip_device.hpp, the
// header client kernels include (cstone_hip::traverseNeighbors): the entry points here are its first clients.  The traversal stack (160 entries >= 7*21+1, the deepest a
// depth-first octree walk can get) lives in LDS; overflow is reported through the sticky error word.
// Compiled with -ffp-contract=off: distances must round like the CPU path (no FMA).
#include <algorithm>
#include <type_traits>

#include "cstone_hip_device.hpp"
#include "ctx.hpp"
#include "device_keys.hpp"

namespace cship
{

namespace
{

constexpr int NB_BLOCK = 256;
constexpr int NB_WAVES = NB_BLOCK / 64;

__device__ __forceinline__ NodeIdx uniform(NodeIdx v) { return __builtin_amdgcn_readfirstlane(v); }

//! the traversal counters of the reference's NcStats (R/traversal/find_neighbors.cuh:345-369,494-502): distance tests of
//! THIS target, tests the wave issued (64 lanes per leaf particle), deepest stack use
struct WalkStats
{
    uint32_t myTests = 0, issued = 0;
    int maxTop       = 0;
    __device__ void leaf(uint32_t particles, bool mine)
    {
        issued += particles;
        if (mine) myTests += particles;
    }
    __device__ void depth(int top) { maxTop = max(maxTop, top); }
};

/*! The list-producing entry points are clients of the SAME traversal that client kernels get (include/
 *  cstone_hip_device.hpp, cstone_hip::traverseNeighbors): the functor stores the neighbour index. */
template<class T, bool GROUPS, bool STATS>
__global__ __launch_bounds__(NB_BLOCK) void findNeighborsKernel(
    const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, const T* __restrict__ h, uint32_t first,
    uint32_t last, const uint32_t* __restrict__ groupStart, const uint32_t* __restrict__ groupEnd, uint32_t numGroups,
    cstone_hip::DeviceBox<T> box, cstone_hip::OctreeNsView<T> tree, float ext, uint32_t ngmax,
    uint32_t* __restrict__ neighbors, uint32_t* __restrict__ counts, int* __restrict__ errors,
    unsigned long long* __restrict__ stats, bool interleaved)
{
    __shared__ cstone_hip::TraversalStack stacks[NB_WAVES];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;

    // targets of this wave: 64 consecutive particles, or (GROUPS) the particles of one target group, 64 at a time
    uint32_t chunk = first + (blockIdx.x * NB_WAVES + wave) * 64u, chunkEnd = last;
    if (GROUPS)
    {
        const uint32_t g = blockIdx.x * NB_WAVES + wave;
        if (g >= numGroups) return;
        chunk    = max(first, uint32_t(uniform(NodeIdx(groupStart[g]))));
        chunkEnd = min(last, uint32_t(uniform(NodeIdx(groupEnd[g]))));
    }
    if (chunk >= chunkEnd) return;
    do
    {
        const bool valid   = chunk + lane < chunkEnd;
        const uint32_t i   = valid ? chunk + lane : chunkEnd - 1;
        const uint32_t tid = i - first;
        // neighbour k of target t: neighbors[t * ngmax + k] (the CPU findNeighbors, R/findneighbors.hpp:96-188), or in
        // blocks of 64 targets neighbors[((t / 64) * ngmax + k) * 64 + t % 64] (the warp-interleaved lists of
        // traverseNeighbors, R/traversal/find_neighbors.cuh:116, with targetSize = 64: a wave's stores of one k are one
        // contiguous line)
        const uint32_t outStride = interleaved ? 64u : 1u;
        uint32_t* out =
            interleaved ? neighbors + size_t(tid >> 6) * ngmax * 64u + (tid & 63u) : neighbors + size_t(tid) * ngmax;
        uint32_t stored = 0;
        auto keep       = [&](uint32_t j, T, T, T, T)
        {
            if (stored < ngmax) out[size_t(stored) * outStride] = j;
            ++stored;
        };
        uint32_t nn;
        WalkStats ws;
        if constexpr (STATS)
            nn = cstone_hip::traverseNeighbors(valid, i, x, y, z, h, tree, box, ext, stacks[wave], errors, keep, ws);
        else nn = cstone_hip::traverseNeighbors(valid, i, x, y, z, h, tree, box, ext, stacks[wave], errors, keep);
        if (valid) counts[tid] = nn;
        if constexpr (STATS)
        {
            unsigned long long sum = valid ? ws.myTests : 0u;
            uint32_t mx            = valid ? ws.myTests : 0u;
            for (int o = 32; o > 0; o >>= 1)
            {
                sum += __shfl_down(sum, o);
                mx = max(mx, uint32_t(__shfl_down(int(mx), o)));
            }
            if (lane == 0)
            {
                atomicAdd(&stats[0], sum);
                atomicMax(&stats[1], (unsigned long long)mx);
                atomicMax(&stats[2], (unsigned long long)ws.maxTop);
                atomicAdd(&stats[3], (unsigned long long)ws.issued * 64ull);
            }
        }
        chunk += 64;
    } while (GROUPS && chunk < chunkEnd);
}

} // namespace

} // namespace cship

using namespace cship;

namespace
{

template<bool GROUPS, bool STATS = false>
int launchFindNeighbors(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y, const void* z, const void* h,
                        uint32_t first, uint32_t last, const uint32_t* groupStart, const uint32_t* groupEnd,
                        uint32_t numGroups, const cstone_box* box_host, const int32_t* child_offsets,
                        const int32_t* internal_to_leaf, const uint32_t* layout, const void* centers, const void* sizes,
                        float ext, uint32_t ngmax, uint32_t* neighbors, uint32_t* counts,
                        unsigned long long* statsDev = nullptr, bool interleaved = false)
{
    if (!ctx || !x || !y || !z || !h || !box_host || !child_offsets || !internal_to_leaf || !layout || !centers ||
        !sizes || !counts || (ngmax && !neighbors) || last < first || (GROUPS && numGroups && (!groupStart || !groupEnd)))
        return fail(ctx, CSTONE_E_ARG, "find_neighbors: bad argument");
    if (last == first || (GROUPS && numGroups == 0)) return CSTONE_OK;
    if (real_bits != 32 && real_bits != 64) return fail(ctx, CSTONE_E_ARG, "find_neighbors: real_bits %d unsupported", real_bits);
    int* errors = ctx->devScalars + 63;
    {
        StageTimer timer(ctx, CSTONE_STAGE_NEIGHBORS);
        unsigned grid = GROUPS ? gridFor(numGroups, NB_WAVES) : gridFor(size_t(last - first), NB_BLOCK);
        if (real_bits == 32)
        {
            cstone_hip::OctreeNsView<float> tree{child_offsets, internal_to_leaf, layout, (const float*)centers,
                                                 (const float*)sizes};
            hipLaunchKernelGGL((findNeighborsKernel<float, GROUPS, STATS>), grid, NB_BLOCK, 0, ctx->stream, (const float*)x,
                               (const float*)y, (const float*)z, (const float*)h, first, last, groupStart, groupEnd,
                               numGroups, cstone_hip::makeDeviceBox<float>(*box_host), tree, ext, ngmax, neighbors, counts,
                               errors, statsDev, interleaved);
        }
        else
        {
            cstone_hip::OctreeNsView<double> tree{child_offsets, internal_to_leaf, layout, (const double*)centers,
                                                  (const double*)sizes};
            hipLaunchKernelGGL((findNeighborsKernel<double, GROUPS, STATS>), grid, NB_BLOCK, 0, ctx->stream,
                               (const double*)x, (const double*)y, (const double*)z, (const double*)h, first, last,
                               groupStart, groupEnd, numGroups, cstone_hip::makeDeviceBox<double>(*box_host), tree, ext,
                               ngmax, neighbors, counts, errors, statsDev, interleaved);
        }
    }
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // namespace

extern "C" int cstone_hip_find_neighbors(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y,
                                         const void* z, const void* h, uint32_t first, uint32_t last,
                                         const cstone_box* box_host, const int32_t* child_offsets,
                                         const int32_t* internal_to_leaf, const uint32_t* layout, const void* centers,
                                         const void* sizes, float ext, uint32_t ngmax, uint32_t* neighbors,
                                         uint32_t* counts)
{
    return launchFindNeighbors<false>(ctx, real_bits, x, y, z, h, first, last, nullptr, nullptr, 0, box_host,
                                      child_offsets, internal_to_leaf, layout, centers, sizes, ext, ngmax, neighbors,
                                      counts);
}

extern "C" int cstone_hip_find_neighbors_interleaved(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y,
                                                     const void* z, const void* h, uint32_t first, uint32_t last,
                                                     const cstone_box* box_host, const int32_t* child_offsets,
                                                     const int32_t* internal_to_leaf, const uint32_t* layout,
                                                     const void* centers, const void* sizes, float ext, uint32_t ngmax,
                                                     uint32_t* neighbors, uint32_t* counts)
{
    return launchFindNeighbors<false>(ctx, real_bits, x, y, z, h, first, last, nullptr, nullptr, 0, box_host,
                                      child_offsets, internal_to_leaf, layout, centers, sizes, ext, ngmax, neighbors,
                                      counts, nullptr, true);
}

extern "C" int cstone_hip_find_neighbors_groups(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y,
                                                const void* z, const void* h, uint32_t first, uint32_t last,
                                                const uint32_t* group_start, const uint32_t* group_end,
                                                uint32_t num_groups, const cstone_box* box_host,
                                                const int32_t* child_offsets, const int32_t* internal_to_leaf,
                                                const uint32_t* layout, const void* centers, const void* sizes,
                                                float ext, uint32_t ngmax, uint32_t* neighbors, uint32_t* counts)
{
    return launchFindNeighbors<true>(ctx, real_bits, x, y, z, h, first, last, group_start, group_end, num_groups,
                                     box_host, child_offsets, internal_to_leaf, layout, centers, sizes, ext, ngmax,
                                     neighbors, counts);
}

/* findNeighbors with the traversal counters of the reference's NcStats (R/traversal/find_neighbors.cuh:345-369,494-502) */
extern "C" int cstone_hip_find_neighbors_stats(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y,
                                               const void* z, const void* h, uint32_t first, uint32_t last,
                                               const cstone_box* box_host, const int32_t* child_offsets,
                                               const int32_t* internal_to_leaf, const uint32_t* layout,
                                               const void* centers, const void* sizes, float ext, uint32_t ngmax,
                                               uint32_t* neighbors, uint32_t* counts, uint64_t* stats_host)
{
    if (!ctx || !stats_host) return fail(ctx, CSTONE_E_ARG, "find_neighbors_stats: bad argument");
    CS_TRY(arenaReserve(ctx, 256));
    auto* dev = static_cast<unsigned long long*>(arenaTake(ctx, 4 * sizeof(unsigned long long)));
    int rc    = CSTONE_OK;
    if (hipMemsetAsync(dev, 0, 4 * sizeof(unsigned long long), ctx->stream) != hipSuccess)
        rc = fail(ctx, CSTONE_E_HIP, "find_neighbors_stats: memset failed");
    if (rc == CSTONE_OK)
        rc = launchFindNeighbors<false, true>(ctx, real_bits, x, y, z, h, first, last, nullptr, nullptr, 0, box_host,
                                              child_offsets, internal_to_leaf, layout, centers, sizes, ext, ngmax,
                                              neighbors, counts, dev);
    if (rc == CSTONE_OK)
    {
        hipError_t e = hipMemcpyAsync(stats_host, dev, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, CSTONE_E_HIP, "find_neighbors_stats: %s", hipGetErrorString(e));
    }
    arenaReset(ctx);
    return rc;
}
