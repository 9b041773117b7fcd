// Cornerstone leaf array + fully linked octree on gfx950.
// Replaces R/tree/csarray_gpu.cu (computeNodeCountsGpu :102-131, computeNodeOpsGpu :191-205,
// rebalanceTreeGpu :207-224), R/tree/update_gpu.cuh:59-82 and R/tree/octree_gpu.cu:56-208.
// All arrays are tiny next to the particle data (L leaves ~ N/bucket): these kernels are launch- and
// latency-bound, so the design goal is few launches, uniform work per lane and ONE scalar read-back
// per rebalance step (the reference does 3-4 symbol copies / thrust copies per step).
#include <algorithm>

#include "ctx.hpp"
#include "device_keys.hpp"
#include "scan.hpp"

namespace cship
{

template<class K>
int sortPairsArena(cstone_hip_ctx* ctx, K* keys, uint32_t* vals, size_t n, int keyBits); // sort.hip

namespace
{

template<class K>
__device__ __forceinline__ size_t lowerBound(const K* __restrict__ a, size_t n, K v)
{
    size_t lo = 0, len = n;
    while (len > 0)
    {
        size_t half = len >> 1;
        bool right  = a[lo + half] < v;
        lo          = right ? lo + half + 1 : lo;
        len         = right ? len - half - 1 : half;
    }
    return lo;
}

// ---- counts[i] = min(#keys in [tree[i], tree[i+1]), maxCount)              R/tree/csarray.hpp:94-103
//      Two steps: the position of every leaf boundary in the sorted keys (numNodes + 1 searches instead of two per
//      leaf), then the differences.  With a guess per boundary (the positions of the previous step: particles move
//      little between two syncs -- the reference's useCountsAsGuess, csarray.hpp:117-186) the search gallops away from
//      the guess and typically ends after two or three probes instead of log2(N) = 27.
template<class K>
__global__ __launch_bounds__(256) void boundaryPositionsKernel(const K* __restrict__ tree, NodeIdx numNodes,
                                                               const K* __restrict__ keys, size_t n,
                                                               const uint32_t* __restrict__ guess,
                                                               uint32_t* __restrict__ pos)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i > numNodes) return;
    const K key = tree[i];
    size_t lo = 0, hi = n; // the answer (first index with keys[idx] >= key) lies in [lo, hi]
    if (guess != nullptr)
    {
        size_t g = guess[i] < n ? guess[i] : n;
        if (g < n && keys[g] < key)
        {
            lo          = g + 1;
            size_t step = 1, p = lo;
            while (p < n && keys[p] < key)
            {
                lo = p + 1;
                step *= 2;
                p = lo + step - 1;
            }
            hi = p < n ? p : n;
        }
        else
        {
            hi          = g;
            size_t step = 1;
            while (hi >= step && !(keys[hi - step] < key))
            {
                hi -= step;
                step *= 2;
            }
            lo = hi >= step ? hi - step + 1 : 0;
        }
    }
    size_t len = hi - lo;
    while (len > 0)
    {
        size_t half = len >> 1;
        if (keys[lo + half] < key) { lo += half + 1, len -= half + 1; }
        else { len = half; }
    }
    pos[i] = uint32_t(lo);
}

__global__ __launch_bounds__(256) void countsFromPositionsKernel(const uint32_t* __restrict__ pos, NodeIdx numNodes,
                                                                 uint32_t maxCount, uint32_t* __restrict__ counts)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numNodes) return;
    uint32_t c = pos[i + 1] - pos[i];
    counts[i]  = c < maxCount ? c : maxCount;
}

template<class K>
int nodeCounts(cstone_hip_ctx* ctx, const K* tree, uint32_t* counts, NodeIdx numNodes, const K* keys, size_t n,
               uint32_t maxCount, const uint32_t* guess)
{
    if (numNodes == 0) return CSTONE_OK;
    CS_TRY(arenaReserve(ctx, alignUp(size_t(numNodes + 1) * sizeof(uint32_t)) + 1024));
    auto* pos = (uint32_t*)arenaTake(ctx, size_t(numNodes + 1) * sizeof(uint32_t));
    hipLaunchKernelGGL(boundaryPositionsKernel<K>, gridFor(numNodes + 1, 256), 256, 0, ctx->stream, tree, numNodes, keys,
                       n, guess, pos);
    hipLaunchKernelGGL(countsFromPositionsKernel, gridFor(numNodes, 256), 256, 0, ctx->stream, pos, numNodes, maxCount,
                       counts);
    arenaReset(ctx);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

// ---- rebalance decision                                                     R/tree/csarray.hpp:270-310
template<class K>
__device__ __forceinline__ int nodeOp(const K* __restrict__ tree, NodeIdx i, const uint32_t* __restrict__ counts,
                                      uint32_t bucket)
{
    K start        = tree[i];
    unsigned level = levelOfSpan<K>(tree[i + 1] - start);
    if (level > 0)
    {
        int sib = octDigit(start, level);
        if (sib > 0)
        {
            NodeIdx first = i - sib;
            // all eight siblings present <=> the 8 nodes starting at `first` tile exactly one parent node
            if (tree[first + 8] == tree[first] + nodeSpan<K>(level - 1))
            {
                uint64_t parent = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    parent += counts[first + k];
                if (parent <= uint64_t(bucket)) return 0;
            }
        }
    }
    constexpr unsigned top = maxLevel<K>();
    uint32_t c             = counts[i];
    if (c > bucket * 512u && level + 3 < top) return 4096;
    if (c > bucket * 64u && level + 2 < top) return 512;
    if (c > bucket * 8u && level + 1 < top) return 64;
    if (c > bucket && level < top) return 8;
    return 1;
}

//! ops[numNodes+1] (ops[numNodes] = 0); changed[0] |= 1 if any op != 1
template<class K>
__global__ __launch_bounds__(256) void nodeOpsKernel(const K* __restrict__ tree, NodeIdx numNodes,
                                                     const uint32_t* __restrict__ counts, uint32_t bucket,
                                                     uint32_t* __restrict__ ops, int* __restrict__ changed)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    int op    = 1;
    if (i < numNodes)
    {
        // nodeOp reads tree[i - sib + 8]: in range, because the parent cell of node i holds at least
        // 8 - sib nodes from i onwards (one or more per remaining sibling cell), so i - sib + 8 <= numNodes
        op     = nodeOp<K>(tree, i, counts, bucket);
        ops[i] = uint32_t(op);
    }
    else if (i == numNodes) { ops[i] = 0; }
    // (set once: updates of tens of thousands of waves to one address serialise in the L2, reads of it do not)
    if (__any(op != 1) && (threadIdx.x & 63) == 0 && *reinterpret_cast<volatile int*>(changed) == 0) atomicOr(changed, 1);
}

//! new leaf j descends from the old node src with ops[src] <= j < ops[src+1]        R/tree/csarray.hpp:360-385
template<class K>
__global__ __launch_bounds__(256) void rebalanceKernel(const K* __restrict__ tree, NodeIdx numNodes,
                                                       const uint32_t* __restrict__ ops, NodeIdx newNumNodes,
                                                       K* __restrict__ newTree)
{
    NodeIdx j = blockIdx.x * 256 + threadIdx.x;
    if (j > newNumNodes) return;
    if (j == newNumNodes)
    {
        newTree[j] = tree[numNodes];
        return;
    }
    // upper_bound(ops, j) - 1 over ops[0..numNodes]
    NodeIdx lo = 0, len = numNodes + 1;
    while (len > 0)
    {
        NodeIdx half = len >> 1;
        bool right   = ops[lo + half] <= uint32_t(j);
        lo           = right ? lo + half + 1 : lo;
        len          = right ? len - half - 1 : half;
    }
    NodeIdx src    = lo - 1;
    uint32_t first = ops[src];
    uint32_t cnt   = ops[src + 1] - first;
    K start        = tree[src];
    unsigned level = levelOfSpan<K>(tree[src + 1] - start);
    // cnt in {1,8,64,512,4096} -> 0..4 levels down
    unsigned down = (31u - unsigned(__clz(cnt))) / 3u;
    newTree[j]    = start + K(uint32_t(j) - first) * nodeSpan<K>(level + down);
}

template<class K>
__global__ void initRootKernel(K* tree, uint32_t* counts, uint32_t n)
{
    tree[0]   = 0;
    tree[1]   = endKey<K>();
    counts[0] = n;
}

// ------------------------------------------------------------------------------------------------
// linked octree, R/tree/octree.hpp:73-211
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int digitWeight(int d) { return d >= 4 ? 7 - d : -d; } // R/sfc/common.hpp:270-275

template<class K>
__global__ __launch_bounds__(256) void unsortedLayoutKernel(const K* __restrict__ leaves, NodeIdx numInternal,
                                                            NodeIdx numLeaves, K* __restrict__ prefixes,
                                                            uint32_t* __restrict__ order)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numLeaves) return;
    K key                    = leaves[i];
    K next                   = leaves[i + 1];
    unsigned level           = levelOfSpan<K>(next - key);
    prefixes[i + numInternal] = toPrefix<K>(key, 3 * level);
    order[i + numInternal]    = uint32_t(i + numInternal);

    unsigned shared = sharedPrefixBits<K>(key, next);
    if (shared % 3 == 0 && i < numLeaves - 1)
    {
        // every internal node is emitted by exactly one leaf: the last leaf below its first..seventh child
        NodeIdx w = 0;
        for (unsigned l = 1; l <= shared / 3 + 1; ++l)
            w += digitWeight(octDigit(key, l));
        NodeIdx slot   = (i + w) / 7;
        prefixes[slot] = toPrefix<K>(key, shared);
        order[slot]    = uint32_t(slot);
    }
}

// order and internalToLeaf may be the same buffer (lane i reads order[i] before writing internalToLeaf[i])
/*! everything between the sort of the node keys and the linking in one launch: the two index maps from the sort's
 *  order, the child offsets cleared for linkKernel (it writes those of the internal nodes only), and -- first wave of the
 *  first workgroup -- where each level starts among the sorted keys */
template<class K>
__global__ __launch_bounds__(256) void invertOrderKernel(const uint32_t* order, NodeIdx numNodes,
                                                         NodeIdx numInternal, NodeIdx* internalToLeaf,
                                                         NodeIdx* __restrict__ leafToInternal,
                                                         const K* __restrict__ prefixes, NodeIdx* __restrict__ levelRange,
                                                         NodeIdx* __restrict__ childOffsets)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 64)
    {
        unsigned l = threadIdx.x;
        if (l <= maxLevel<K>()) levelRange[l] = NodeIdx(lowerBound(prefixes, size_t(numNodes), toPrefix<K>(K(0), 3 * l)));
        if (l == maxLevel<K>() + 1) levelRange[l] = numNodes;
    }
    if (i <= numNodes) childOffsets[i] = 0;
    if (i >= numNodes) return;
    uint32_t o        = order[i];
    leafToInternal[o] = i;
    internalToLeaf[i] = NodeIdx(o) - numInternal;
}

template<class K>
__global__ __launch_bounds__(256) void linkKernel(const K* __restrict__ prefixes, NodeIdx numInternal,
                                                  const NodeIdx* __restrict__ leafToInternal,
                                                  const NodeIdx* __restrict__ levelRange,
                                                  NodeIdx* __restrict__ childOffsets, NodeIdx* __restrict__ parents)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numInternal) return;
    NodeIdx a      = leafToInternal[i];
    K prefix       = prefixes[a];
    unsigned nb    = prefixBits(prefix);
    unsigned level = nb / 3;
    K child        = toPrefix<K>(fromPrefix(prefix), nb + 3);
    NodeIdx s = levelRange[level + 1], e = levelRange[level + 2];
    NodeIdx c = s + NodeIdx(lowerBound(prefixes + s, size_t(e - s), child));
    if (c != e && prefixes[c] == child)
    {
        childOffsets[a]      = c;
        parents[(c - 1) / 8] = a;
    }
}

//! one level of the bottom-up saturating sum; reads the level bounds on the device (no host round trip)
__global__ __launch_bounds__(256) void upsweepLevelKernel(int level, const NodeIdx* __restrict__ levelRange,
                                                          const NodeIdx* __restrict__ childOffsets,
                                                          uint32_t* __restrict__ q)
{
    NodeIdx start = levelRange[level], end = levelRange[level + 1];
    for (NodeIdx i = start + blockIdx.x * 256 + threadIdx.x; i < end; i += gridDim.x * 256)
    {
        NodeIdx c = childOffsets[i];
        if (c)
        {
            uint64_t s = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                s += q[c + k];
            q[i] = uint32_t(s < 0xFFFFFFFFull ? s : 0xFFFFFFFFull);
        }
    }
}

/*! the bottom-up sum from level `from` down to the root, a launch per level.  (Round 4 tried the top five levels -- at
 *  most 8^4 nodes each -- in ONE workgroup with a barrier between two levels: 38 us against 5 x 4.6 us for the
 *  launches it replaced; back-to-back launches of one stream leave no gap on the device, and a level inside the
 *  workgroup costs two dependent memory round trips just the same.) */
static void launchUpsweep(cstone_hip_ctx* ctx, int from, const NodeIdx* levelRange, const NodeIdx* childOffsets,
                          uint32_t* counts)
{
    unsigned grid = unsigned(ctx->numCu) * 4;
    for (int level = from; level >= 0; --level)
        hipLaunchKernelGGL(upsweepLevelKernel, grid, 256, 0, ctx->stream, level, levelRange, childOffsets, counts);
}

// ---- geometric node centers, R/sfc/box.hpp:335-352 (compiled with -ffp-contract=off like the encode)
template<class K, class T>
__global__ __launch_bounds__(256) void nodeCentersKernel(const K* __restrict__ prefixes, NodeIdx numNodes,
                                                         DBox<T> box, bool hilbert,
                                                         const uint16_t* __restrict__ decTable, T* __restrict__ centers,
                                                         T* __restrict__ sizes)
{
    __shared__ uint16_t dec[24 * 8];
    if (threadIdx.x < 24 * 8) dec[threadIdx.x] = decTable[threadIdx.x];
    __syncthreads();
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numNodes) return;
    K prefix       = prefixes[i];
    K start        = fromPrefix(prefix);
    unsigned level = prefixBits(prefix) / 3;
    K morton       = hilbert ? mortonFromHilbert<K>(start, dec) : start;
    unsigned ix, iy, iz;
    mortonDecode<K>(morton, ix, iy, iz);
    unsigned edge = 1u << (maxLevel<K>() - level);
    unsigned m    = ~(edge - 1);
    int lo[3]     = {int(ix & m), int(iy & m), int(iz & m)};
    constexpr int g = 1 << maxLevel<K>();
    constexpr T uL  = T(1.) / g;
#pragma unroll
    for (int d = 0; d < 3; ++d)
    {
        T half             = T(0.5) * uL * box.len[d];
        int hi             = lo[d] + int(edge);
        centers[3 * i + d] = box.lo[d] + T(hi + lo[d]) * half;
        sizes[3 * i + d]   = T(hi - lo[d]) * half;
    }
}

template<class K>
int updateOctree(cstone_hip_ctx* ctx, const K* keys, size_t n, uint32_t bucket, K* tree, uint32_t* counts,
                 int* numLeavesHost, int capLeaves, uint32_t maxCount, int* convergedHost)
{
    NodeIdx numNodes = *numLeavesHost;
    if (numNodes < 1 || numNodes > capLeaves) return fail(ctx, CSTONE_E_ARG, "update_octree: bad leaf count");

    size_t opsBytes = alignUp(size_t(numNodes + 1) * sizeof(uint32_t));
    CS_TRY(arenaReserve(ctx, opsBytes + alignUp(size_t(capLeaves + 1) * sizeof(K)) + scanArenaBytes(numNodes + 1) +
                                 4096));
    auto* ops     = (uint32_t*)arenaTake(ctx, opsBytes);
    K* newTree    = (K*)arenaTake(ctx, size_t(capLeaves + 1) * sizeof(K));
    int* scalars  = ctx->devScalars; // [0] changed flag, [1] new node count
    int rc        = CSTONE_OK;
    NodeIdx newNumNodes = 0;
    {
        StageTimer timer(ctx, CSTONE_STAGE_REBALANCE);
        (void)hipMemsetAsync(scalars, 0, 2 * sizeof(int), ctx->stream);
        hipLaunchKernelGGL(nodeOpsKernel<K>, gridFor(numNodes + 1, 256), 256, 0, ctx->stream, tree, numNodes, counts,
                           bucket, ops, scalars);
        rc = scanU32(ctx, ops, ops, size_t(numNodes) + 1, 0u, false, (uint32_t*)scalars + 1);
        if (rc == CSTONE_OK)
        {
            rc = copyToPinned(ctx, ctx->hostScalars, scalars, 2 * sizeof(int));
            hipError_t e = rc == CSTONE_OK ? hipStreamSynchronize(ctx->stream) : hipSuccess;
            if (e != hipSuccess) rc = fail(ctx, CSTONE_E_HIP, "update_octree: %s", hipGetErrorString(e));
        }
        if (rc == CSTONE_OK)
        {
            *convergedHost = ctx->hostScalars[0] == 0;
            newNumNodes    = ctx->hostScalars[1];
            if (newNumNodes > capLeaves)
            {
                *numLeavesHost = newNumNodes;
                rc = fail(ctx, CSTONE_E_CAPACITY, "update_octree: %d leaves needed, capacity %d", newNumNodes, capLeaves);
            }
        }
        if (rc == CSTONE_OK && !*convergedHost)
        {
            hipLaunchKernelGGL(rebalanceKernel<K>, gridFor(newNumNodes + 1, 256), 256, 0, ctx->stream, tree, numNodes,
                               ops, newNumNodes, newTree);
            (void)hipMemcpyAsync(tree, newTree, size_t(newNumNodes + 1) * sizeof(K), hipMemcpyDeviceToDevice, ctx->stream);
        }
        // converged: every node op is "keep", the leaf array is what it was (only the counts are refreshed below)
    }
    arenaReset(ctx);
    CS_TRY(rc);
    {
        StageTimer timer(ctx, CSTONE_STAGE_NODE_COUNTS);
        CS_TRY(nodeCounts<K>(ctx, tree, counts, newNumNodes, keys, n, maxCount, nullptr));
    }
    *numLeavesHost = newNumNodes;
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

template<class K>
int buildOctree(cstone_hip_ctx* ctx, const K* leaves, NodeIdx numLeaves, K* prefixes, NodeIdx* childOffsets,
                NodeIdx* parents, NodeIdx* levelRange, NodeIdx* internalToLeaf, NodeIdx* leafToInternal,
                int deepestLevel = int(maxLevel<K>()) /* bound on the level of the leaves, if the caller has one */)
{
    if (numLeaves < 1) return fail(ctx, CSTONE_E_ARG, "build_octree: need at least one leaf");
    NodeIdx numInternal = (numLeaves - 1) / 7;
    NodeIdx numNodes    = numLeaves + numInternal;
    StageTimer timer(ctx, CSTONE_STAGE_LINK_OCTREE);
    // internalToLeaf doubles as the sort payload (uint32 view); leafToInternal is written by the inversion
    auto* order = reinterpret_cast<uint32_t*>(internalToLeaf);
    hipLaunchKernelGGL(unsortedLayoutKernel<K>, gridFor(numLeaves, 256), 256, 0, ctx->stream, leaves, numInternal,
                       numLeaves, prefixes, order);
    // a node key of level l has its sentinel bit at 3 l: no digit pass over the zeros above the deepest level
    CS_TRY(sortPairsArena<K>(ctx, prefixes, order, size_t(numNodes), 3 * std::min(deepestLevel, int(maxLevel<K>())) + 1));
    // order -> (leafToInternal, internalToLeaf); reading and writing internalToLeaf[i] in the same lane is safe
    hipLaunchKernelGGL(invertOrderKernel<K>, gridFor(size_t(numNodes) + 1, 256), 256, 0, ctx->stream, order, numNodes,
                       numInternal, internalToLeaf, leafToInternal, prefixes, levelRange, childOffsets);
    if (numInternal > 0)
        hipLaunchKernelGGL(linkKernel<K>, gridFor(numInternal, 256), 256, 0, ctx->stream, prefixes, numInternal,
                           leafToInternal, levelRange, childOffsets, parents);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // namespace

} // namespace cship

namespace cship
{
int buildLinkedOctree(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int numLeaves, void* prefixes,
                      int32_t* childOffsets, int32_t* parents, int32_t* levelRange, int32_t* internalToLeaf,
                      int32_t* leafToInternal, int deepestLevel)
{
    if (key_bits == 32)
        return buildOctree<uint32_t>(ctx, (const uint32_t*)leaves, numLeaves, (uint32_t*)prefixes, childOffsets, parents,
                                     levelRange, internalToLeaf, leafToInternal, deepestLevel);
    return buildOctree<uint64_t>(ctx, (const uint64_t*)leaves, numLeaves, (uint64_t*)prefixes, childOffsets, parents,
                                 levelRange, internalToLeaf, leafToInternal, deepestLevel);
}

//! upsweep with the level ranges known on the host as well: empty levels are not launched
int upsweepSumLevels(cstone_hip_ctx* ctx, int numLevelsPlus2, const int32_t* levelRangeHost, const int32_t* levelRange,
                     const int32_t* childOffsets, uint32_t* counts)
{
    for (int level = numLevelsPlus2 - 2; level >= 0; --level)
    {
        int size = levelRangeHost[level + 1] - levelRangeHost[level];
        if (size <= 0) continue;
        unsigned grid = std::min(unsigned(ctx->numCu) * 4, gridFor(size_t(size), 256));
        hipLaunchKernelGGL(upsweepLevelKernel, grid, 256, 0, ctx->stream, level, levelRange, childOffsets, counts);
    }
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}
} // namespace cship

using namespace cship;

#define CS_KEY_DISPATCH(key_bits, CALL32, CALL64)                                                                      \
    if ((key_bits) == 32) { return CALL32; }                                                                           \
    if ((key_bits) == 64) { return CALL64; }                                                                           \
    return fail(ctx, CSTONE_E_ARG, "%s: key_bits %d unsupported", __func__, key_bits)

extern "C"
{

int cstone_hip_compute_node_counts_guided(cstone_hip_ctx* ctx, int key_bits, const void* tree, uint32_t* counts,
                                          int num_nodes, const void* keys, size_t n, uint32_t max_count,
                                          const uint32_t* guess_positions)
{
    if (!ctx || !tree || !counts || num_nodes < 0 || (n && !keys))
        return fail(ctx, CSTONE_E_ARG, "compute_node_counts: bad argument");
    if (num_nodes == 0) return CSTONE_OK;
    if (n >= (size_t(1) << 32)) return fail(ctx, CSTONE_E_ARG, "compute_node_counts: more than 2^32 - 1 keys");
    StageTimer timer(ctx, CSTONE_STAGE_NODE_COUNTS);
    if (key_bits == 32)
        return nodeCounts<uint32_t>(ctx, (const uint32_t*)tree, counts, num_nodes, (const uint32_t*)keys, n, max_count,
                                    guess_positions);
    if (key_bits == 64)
        return nodeCounts<uint64_t>(ctx, (const uint64_t*)tree, counts, num_nodes, (const uint64_t*)keys, n, max_count,
                                    guess_positions);
    return fail(ctx, CSTONE_E_ARG, "compute_node_counts: key_bits %d unsupported", key_bits);
}

int cstone_hip_compute_node_counts(cstone_hip_ctx* ctx, int key_bits, const void* tree, uint32_t* counts,
                                   int num_nodes, const void* keys, size_t n, uint32_t max_count)
{
    return cstone_hip_compute_node_counts_guided(ctx, key_bits, tree, counts, num_nodes, keys, n, max_count, nullptr);
}

int cstone_hip_compute_node_ops(cstone_hip_ctx* ctx, int key_bits, const void* tree, int num_nodes,
                                const uint32_t* counts, uint32_t bucket_size, int32_t* node_ops,
                                int* new_num_nodes_host, int* converged_host)
{
    if (!ctx || !tree || !counts || !node_ops || num_nodes < 1 || !new_num_nodes_host || !converged_host)
        return fail(ctx, CSTONE_E_ARG, "compute_node_ops: bad argument");
    if (key_bits != 32 && key_bits != 64) return fail(ctx, CSTONE_E_ARG, "compute_node_ops: bad key_bits");
    StageTimer timer(ctx, CSTONE_STAGE_REBALANCE);
    int* scalars = ctx->devScalars;
    auto* ops    = (uint32_t*)node_ops;
    CS_TRY(arenaReserve(ctx, scanArenaBytes(num_nodes + 1)));
    CS_HIP(ctx, hipMemsetAsync(scalars, 0, 2 * sizeof(int), ctx->stream));
    if (key_bits == 32)
        hipLaunchKernelGGL(nodeOpsKernel<uint32_t>, gridFor(num_nodes + 1, 256), 256, 0, ctx->stream,
                           (const uint32_t*)tree, num_nodes, counts, bucket_size, ops, scalars);
    else
        hipLaunchKernelGGL(nodeOpsKernel<uint64_t>, gridFor(num_nodes + 1, 256), 256, 0, ctx->stream,
                           (const uint64_t*)tree, num_nodes, counts, bucket_size, ops, scalars);
    int rc = scanU32(ctx, ops, ops, size_t(num_nodes) + 1, 0u, false, (uint32_t*)scalars + 1);
    arenaReset(ctx);
    CS_TRY(rc);
    CS_TRY(copyToPinned(ctx, ctx->hostScalars, scalars, 2 * sizeof(int)));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *converged_host     = ctx->hostScalars[0] == 0;
    *new_num_nodes_host = ctx->hostScalars[1];
    return CSTONE_OK;
}

int cstone_hip_rebalance_tree(cstone_hip_ctx* ctx, int key_bits, const void* tree, int num_nodes, int new_num_nodes,
                              const int32_t* node_ops, void* new_tree)
{
    if (!ctx || !tree || !node_ops || !new_tree || num_nodes < 1 || new_num_nodes < 1)
        return fail(ctx, CSTONE_E_ARG, "rebalance_tree: bad argument");
    StageTimer timer(ctx, CSTONE_STAGE_REBALANCE);
    auto* ops = (const uint32_t*)node_ops;
    if (key_bits == 32)
        hipLaunchKernelGGL(rebalanceKernel<uint32_t>, gridFor(new_num_nodes + 1, 256), 256, 0, ctx->stream,
                           (const uint32_t*)tree, num_nodes, ops, new_num_nodes, (uint32_t*)new_tree);
    else if (key_bits == 64)
        hipLaunchKernelGGL(rebalanceKernel<uint64_t>, gridFor(new_num_nodes + 1, 256), 256, 0, ctx->stream,
                           (const uint64_t*)tree, num_nodes, ops, new_num_nodes, (uint64_t*)new_tree);
    else
        return fail(ctx, CSTONE_E_ARG, "rebalance_tree: key_bits %d unsupported", key_bits);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_update_octree(cstone_hip_ctx* ctx, int key_bits, const void* keys, size_t n, uint32_t bucket_size,
                             void* tree, uint32_t* counts, int* num_leaves_host, int cap_leaves, uint32_t max_count,
                             int* converged_host)
{
    if (!ctx || !tree || !counts || !num_leaves_host || !converged_host || (n && !keys))
        return fail(ctx, CSTONE_E_ARG, "update_octree: bad argument");
    CS_KEY_DISPATCH(key_bits,
                    updateOctree<uint32_t>(ctx, (const uint32_t*)keys, n, bucket_size, (uint32_t*)tree, counts,
                                           num_leaves_host, cap_leaves, max_count, converged_host),
                    updateOctree<uint64_t>(ctx, (const uint64_t*)keys, n, bucket_size, (uint64_t*)tree, counts,
                                           num_leaves_host, cap_leaves, max_count, converged_host));
}

int cstone_hip_compute_octree(cstone_hip_ctx* ctx, int key_bits, const void* keys, size_t n, uint32_t bucket_size,
                              void* tree, uint32_t* counts, int* num_leaves_host, int cap_leaves, uint32_t max_count,
                              int* iterations_host)
{
    if (!ctx || !tree || !counts || !num_leaves_host || cap_leaves < 1 || (n && !keys))
        return fail(ctx, CSTONE_E_ARG, "compute_octree: bad argument");
    if (key_bits == 32)
        hipLaunchKernelGGL(initRootKernel<uint32_t>, 1, 1, 0, ctx->stream, (uint32_t*)tree, counts, uint32_t(n));
    else if (key_bits == 64)
        hipLaunchKernelGGL(initRootKernel<uint64_t>, 1, 1, 0, ctx->stream, (uint64_t*)tree, counts, uint32_t(n));
    else
        return fail(ctx, CSTONE_E_ARG, "compute_octree: key_bits %d unsupported", key_bits);
    *num_leaves_host = 1;
    int converged = 0, iters = 0;
    while (!converged)
    {
        CS_TRY(cstone_hip_update_octree(ctx, key_bits, keys, n, bucket_size, tree, counts, num_leaves_host, cap_leaves,
                                        max_count, &converged));
        ++iters;
        if (iters > 64) return fail(ctx, CSTONE_E_INTERNAL, "compute_octree: no convergence after 64 updates");
    }
    if (iterations_host) *iterations_host = iters;
    return CSTONE_OK;
}

int cstone_hip_build_octree(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves, void* prefixes,
                            int32_t* child_offsets, int32_t* parents, int32_t* level_range, int32_t* internal_to_leaf,
                            int32_t* leaf_to_internal)
{
    if (!ctx || !leaves || !prefixes || !child_offsets || !parents || !level_range || !internal_to_leaf ||
        !leaf_to_internal)
        return fail(ctx, CSTONE_E_ARG, "build_octree: null array");
    CS_KEY_DISPATCH(key_bits,
                    buildOctree<uint32_t>(ctx, (const uint32_t*)leaves, num_leaves, (uint32_t*)prefixes, child_offsets,
                                          parents, level_range, internal_to_leaf, leaf_to_internal),
                    buildOctree<uint64_t>(ctx, (const uint64_t*)leaves, num_leaves, (uint64_t*)prefixes, child_offsets,
                                          parents, level_range, internal_to_leaf, leaf_to_internal));
}

int cstone_hip_upsweep_sum(cstone_hip_ctx* ctx, int num_levels_plus2, const int32_t* level_range,
                           const int32_t* child_offsets, uint32_t* counts)
{
    if (!ctx || !level_range || !child_offsets || !counts || num_levels_plus2 < 2)
        return fail(ctx, CSTONE_E_ARG, "upsweep_sum: bad argument");
    launchUpsweep(ctx, num_levels_plus2 - 2, level_range, child_offsets, counts);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_build_octree_bounded(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves, void* prefixes,
                                    int32_t* child_offsets, int32_t* parents, int32_t* level_range,
                                    int32_t* internal_to_leaf, int32_t* leaf_to_internal, int deepest_level)
{
    if (!ctx || !leaves || !prefixes || !child_offsets || !parents || !level_range || !internal_to_leaf ||
        !leaf_to_internal || deepest_level < 0)
        return fail(ctx, CSTONE_E_ARG, "build_octree_bounded: bad argument");
    CS_KEY_DISPATCH(key_bits,
                    buildOctree<uint32_t>(ctx, (const uint32_t*)leaves, num_leaves, (uint32_t*)prefixes, child_offsets,
                                          parents, level_range, internal_to_leaf, leaf_to_internal, deepest_level),
                    buildOctree<uint64_t>(ctx, (const uint64_t*)leaves, num_leaves, (uint64_t*)prefixes, child_offsets,
                                          parents, level_range, internal_to_leaf, leaf_to_internal, deepest_level));
}

int cstone_hip_upsweep_sum_bounded(cstone_hip_ctx* ctx, int num_levels_plus2, const int32_t* level_range,
                                   const int32_t* child_offsets, uint32_t* counts, int deepest_level)
{
    if (!ctx || !level_range || !child_offsets || !counts || num_levels_plus2 < 2 || deepest_level < 0)
        return fail(ctx, CSTONE_E_ARG, "upsweep_sum_bounded: bad argument");
    // (the nodes of the deepest level are leaves: the first level with anything to sum is the one above)
    launchUpsweep(ctx, std::min(num_levels_plus2 - 2, deepest_level - 1), level_range, child_offsets, counts);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_node_centers(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                            int num_nodes, const cstone_box* box_host, void* centers, void* sizes)
{
    if (!ctx || !prefixes || !box_host || !centers || !sizes || num_nodes < 0)
        return fail(ctx, CSTONE_E_ARG, "node_centers: bad argument");
    if (num_nodes == 0) return CSTONE_OK;
    auto* dec    = (const uint16_t*)ctx->hilbertTables + 48 * 8;
    bool hilbert = curve == CSTONE_HILBERT;
    unsigned grid = gridFor(num_nodes, 256);
#define CS_LAUNCH_CENTERS(K, T)                                                                                        \
    hipLaunchKernelGGL((nodeCentersKernel<K, T>), grid, 256, 0, ctx->stream, (const K*)prefixes, num_nodes,            \
                       makeDBox<T>(*box_host), hilbert, dec, (T*)centers, (T*)sizes)
    if (key_bits == 32 && real_bits == 32) CS_LAUNCH_CENTERS(uint32_t, float);
    else if (key_bits == 32 && real_bits == 64) CS_LAUNCH_CENTERS(uint32_t, double);
    else if (key_bits == 64 && real_bits == 32) CS_LAUNCH_CENTERS(uint64_t, float);
    else if (key_bits == 64 && real_bits == 64) CS_LAUNCH_CENTERS(uint64_t, double);
    else return fail(ctx, CSTONE_E_ARG, "node_centers: unsupported type combination");
#undef CS_LAUNCH_CENTERS
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // extern "C"
