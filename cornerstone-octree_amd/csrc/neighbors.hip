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
// stored lists are identical element for element.  The traversal stack (160 entries >= 7*21+1, the deepest a
// depth-first octree walk can get) lives in LDS; overflow is reported through the sticky error word.
// Compiled with -ffp-contract=off: distances must round like the CPU path (no FMA).
#include <algorithm>
#include <type_traits>

#include "ctx.hpp"
#include "device_keys.hpp"

namespace cship
{

namespace
{

constexpr int NB_BLOCK = 256;
constexpr int NB_WAVES = NB_BLOCK / 64;
constexpr int NB_STACK = 160; // >= 7 * 21 + 1

template<class T, bool PBC>
__device__ __forceinline__ T foldAxis(T dx, T len, T inv, bool periodic)
{
    // dX -= pbc * l * rint(dX * il), R/sfc/box.hpp:195-206
    if (PBC && periodic) return dx - len * rint(dx * inv);
    return dx;
}

__device__ __forceinline__ NodeIdx uniform(NodeIdx v) { return __builtin_amdgcn_readfirstlane(v); }

//! value of lane k (wave-uniform k) as a wave-uniform scalar
__device__ __forceinline__ float readLane(float v, unsigned k)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), int(k)));
}
__device__ __forceinline__ double readLane(double v, unsigned k)
{
    long long b = __double_as_longlong(v);
    int lo      = __builtin_amdgcn_readlane(int(b), int(k));
    int hi      = __builtin_amdgcn_readlane(int(b >> 32), int(k));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

template<class T, bool GROUPS, bool STATS>
__global__ __launch_bounds__(NB_BLOCK) void findNeighborsKernel(
    const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, const T* __restrict__ h, uint32_t first,
    uint32_t last, const uint32_t* __restrict__ groupStart, const uint32_t* __restrict__ groupEnd, uint32_t numGroups,
    DBox<T> box, const NodeIdx* __restrict__ childOffsets, const NodeIdx* __restrict__ internalToLeaf,
    const uint32_t* __restrict__ layout, const T* __restrict__ centers, const T* __restrict__ sizes, float ext,
    uint32_t ngmax, uint32_t* __restrict__ neighbors, uint32_t* __restrict__ counts, int* __restrict__ errors,
    unsigned long long* __restrict__ stats, bool interleaved)
{
    __shared__ NodeIdx stackNode[NB_WAVES][NB_STACK];
    __shared__ uint64_t stackMask[NB_WAVES][NB_STACK];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    NodeIdx* sNode  = stackNode[wave];
    uint64_t* sMask = stackMask[wave];

    // targets of this wave: 64 consecutive particles, or (GROUPS) the particles of one target group, 64 at a time
    uint32_t chunk = first + (blockIdx.x * NB_WAVES + wave) * 64u, chunkEnd = last;
    if (GROUPS)
    {
        const uint32_t g = blockIdx.x * NB_WAVES + wave;
        if (g >= numGroups) return;
        chunk    = max(first, uniform(NodeIdx(groupStart[g])));
        chunkEnd = min(last, uint32_t(uniform(NodeIdx(groupEnd[g]))));
    }
    if (chunk >= chunkEnd) return;
    do
    {
    const bool valid   = chunk + lane < chunkEnd;
    const uint32_t i   = valid ? chunk + lane : chunkEnd - 1;
    const uint32_t tid = i - first;

    const T xi = x[i], yi = y[i], zi = z[i];
    const T hi = h[i];
    const T radSq  = T(4.0) * hi * hi;
    const T cellSq = radSq * ext * ext;
    const bool px = box.bc[0] == 1, py = box.bc[1] == 1, pz = box.bc[2] == 1;
    const T s = T(2) * hi;
    bool inside = (xi - s >= box.lo[0]) && (yi - s >= box.lo[1]) && (zi - s >= box.lo[2]) && (xi + s <= box.hi[0]) &&
                  (yi + s <= box.hi[1]) && (zi + s <= box.hi[2]);
    const bool usePbc = (px || py || pz) && !inside;

    // neighbour k of target t: neighbors[t * ngmax + k] (the CPU findNeighbors, R/findneighbors.hpp:96-188), or in blocks
    // of 64 targets neighbors[((t / 64) * ngmax + k) * 64 + t % 64] (the warp-interleaved lists of traverseNeighbors,
    // R/traversal/find_neighbors.cuh:116, with targetSize = 64: a wave's stores of one k are one contiguous line)
    const uint32_t outStride = interleaved ? 64u : 1u;
    uint32_t* out = interleaved ? neighbors + size_t(tid >> 6) * ngmax * 64u + (tid & 63u) : neighbors + size_t(tid) * ngmax;
    uint32_t nn   = 0;
    // STATS (NcStats, R/traversal/find_neighbors.cuh:345-369,494-502): distance tests of THIS target, tests the wave
    // issued (64 lanes per leaf particle), deepest stack use
    uint32_t myTests = 0, issued = 0;
    int maxTop       = 0;

    // n is wave-uniform: centers and sizes come through the scalar cache
    auto overlaps = [&](NodeIdx n) -> bool
    {
        T dx = centers[3 * n] - xi, dy = centers[3 * n + 1] - yi, dz = centers[3 * n + 2] - zi;
        if (usePbc)
        {
            dx = foldAxis<T, true>(dx, box.len[0], box.inv[0], px);
            dy = foldAxis<T, true>(dy, box.len[1], box.inv[1], py);
            dz = foldAxis<T, true>(dz, box.len[2], box.inv[2], pz);
        }
        dx = fabs(dx) - sizes[3 * n], dy = fabs(dy) - sizes[3 * n + 1], dz = fabs(dz) - sizes[3 * n + 2];
        dx += fabs(dx), dy += fabs(dy), dz += fabs(dz);
        dx *= T(0.5), dy *= T(0.5), dz *= T(0.5);
        return dx * dx + (dy * dy + dz * dz) < cellSq; // right fold, R/util/array.hpp:253-256
    };
    // all particles of leaf node n against the lanes that reached it.  The wave fetches up to 64 leaf particles with
    // one coalesced load per coordinate (lane l holds particle base + l) and hands them round by v_readlane: no memory
    // traffic inside the test loop (wave-uniform scalar loads of every particle saturated the scalar data cache).
    // Two copies of the loop: the periodic fold is only compiled into the one taken when some lane needs it.
    auto searchLeaf = [&](NodeIdx n, bool mine)
    {
        NodeIdx leaf      = uniform(internalToLeaf[n]);
        const uint32_t jb = uniform(layout[leaf]), je = uniform(layout[leaf + 1]);
        const bool fold   = __any(mine && usePbc);
        if (STATS)
        {
            issued += je - jb;
            if (mine) myTests += je - jb;
        }
        for (uint32_t base = jb; base < je; base += 64)
        {
            const uint32_t cnt = min(64u, je - base);
            T xl = T(0), yl = T(0), zl = T(0);
            if (lane < cnt) xl = x[base + lane], yl = y[base + lane], zl = z[base + lane];
            auto test = [&](uint32_t k, auto withFold)
            {
                const uint32_t j = base + k;
                T dx = readLane(xl, k) - xi, dy = readLane(yl, k) - yi, dz = readLane(zl, k) - zi;
                if (decltype(withFold)::value && usePbc)
                {
                    dx = foldAxis<T, true>(dx, box.len[0], box.inv[0], px);
                    dy = foldAxis<T, true>(dy, box.len[1], box.inv[1], py);
                    dz = foldAxis<T, true>(dz, box.len[2], box.inv[2], pz);
                }
                const bool hit = mine && j != i && dx * dx + dy * dy + dz * dz < radSq;
                if (hit)
                {
                    if (nn < ngmax) out[size_t(nn) * outStride] = j;
                    ++nn;
                }
            };
            if (fold)
            {
                for (uint32_t k = 0; k < cnt; ++k)
                    test(k, std::true_type{});
            }
            else
            {
                uint32_t k = 0;
                for (; k + 4 <= cnt; k += 4)
                {
                    test(k, std::false_type{});
                    test(k + 1, std::false_type{});
                    test(k + 2, std::false_type{});
                    test(k + 3, std::false_type{});
                }
                for (; k < cnt; ++k)
                    test(k, std::false_type{});
            }
        }
    };

    // depth-first walk of R/traversal/traversal.hpp:69-110, once per wave
    bool ov0      = valid && overlaps(0);
    uint64_t root = __ballot(ov0);
    if (root != 0)
    {
        if (uniform(childOffsets[0]) == 0) { searchLeaf(0, ov0); }
        else
        {
            int top = 1;
            if (lane == 0)
            {
                sNode[0] = 0;
                sMask[0] = root;
            }
            NodeIdx node  = 0;
            uint64_t mask = root;
            do
            {
                const NodeIdx c0 = uniform(childOffsets[node]);
                const bool here  = (mask >> lane) & 1ull;
#pragma unroll 1
                for (int oct = 0; oct < 8; ++oct)
                {
                    const NodeIdx child = c0 + oct;
                    const bool ov       = here && overlaps(child);
                    const uint64_t cm   = __ballot(ov);
                    if (cm == 0) continue;
                    if (uniform(childOffsets[child]) == 0) { searchLeaf(child, ov); }
                    else if (top < NB_STACK)
                    {
                        if (lane == 0)
                        {
                            sNode[top] = child;
                            sMask[top] = cm;
                        }
                        ++top;
                        if (STATS) maxTop = max(maxTop, top);
                    }
                    else if (lane == 0) { atomicOr(errors, 4); }
                }
                --top;
                node = uniform(sNode[top]);
                mask = sMask[top];
                mask = (uint64_t(__builtin_amdgcn_readfirstlane(uint32_t(mask >> 32))) << 32) |
                       __builtin_amdgcn_readfirstlane(uint32_t(mask));
            } while (node != 0);
        }
    }
    if (valid) counts[tid] = nn;
    if (STATS)
    {
        unsigned long long sum = valid ? myTests : 0u;
        uint32_t mx            = valid ? myTests : 0u;
        for (int o = 32; o > 0; o >>= 1)
        {
            sum += __shfl_down(sum, o);
            mx = max(mx, uint32_t(__shfl_down(int(mx), o)));
        }
        if (lane == 0)
        {
            atomicAdd(&stats[0], sum);
            atomicMax(&stats[1], (unsigned long long)mx);
            atomicMax(&stats[2], (unsigned long long)maxTop);
            atomicAdd(&stats[3], (unsigned long long)issued * 64ull);
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
            hipLaunchKernelGGL((findNeighborsKernel<float, GROUPS, STATS>), grid, NB_BLOCK, 0, ctx->stream, (const float*)x,
                               (const float*)y, (const float*)z, (const float*)h, first, last, groupStart, groupEnd,
                               numGroups, makeDBox<float>(*box_host), child_offsets, internal_to_leaf, layout,
                               (const float*)centers, (const float*)sizes, ext, ngmax, neighbors, counts, errors,
                               statsDev, interleaved);
        else
            hipLaunchKernelGGL((findNeighborsKernel<double, GROUPS, STATS>), grid, NB_BLOCK, 0, ctx->stream,
                               (const double*)x, (const double*)y, (const double*)z, (const double*)h, first, last,
                               groupStart, groupEnd, numGroups, makeDBox<double>(*box_host), child_offsets,
                               internal_to_leaf, layout, (const double*)centers, (const double*)sizes, ext, ngmax,
                               neighbors, counts, errors, statsDev, interleaved);
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
