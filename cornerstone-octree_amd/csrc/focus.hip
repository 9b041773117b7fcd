// Focus-tree (locally essential tree) kernels for gfx950.  Replace the GPU seam of R/focus/rebalance_gpu.h:40-81
// (kernels R/focus/rebalance_gpu.cu:43-240; semantics R/focus/rebalance.hpp:50-301), markMacsGpu
// (R/traversal/collisions_gpu.h:68-77; semantics R/traversal/macs.hpp:96-270), countSfcGapsGpu / fillSfcGapsGpu
// (R/tree/csarray_gpu.h:78-88; semantics R/sfc/common.hpp:370-438) and the node-sphere functions of
// R/focus/source_center_gpu.h:50-89 (semantics R/focus/source_center.hpp:44-156, R/traversal/macs.hpp:44-93).
//
// What differs from the reference's kernels:
//   * enforceKeys: the protection of ancestors and the split request are atomicMax updates, so that concurrent keys
//     give the result of the reference's SEQUENTIAL CPU loop (its kernel races a plain read-modify-write)
//   * rangeCount: 16 lanes add up the global leaves under one focus leaf (the reference: one thread, serial)
//   * markMacs: a wave walks the tree for one target at a time, 8 nodes popped and their 64 children tested per step
//     (LDS stack), like findHalos; the MAC arithmetic keeps the reference's operation order (-ffp-contract=off)
#include <algorithm>
#include <vector>

#include "ctx.hpp"
#include "device_keys.hpp"
#include "scan.hpp"

namespace cship
{

namespace
{

template<class K>
__device__ __forceinline__ int ctzKey(K x)
{
    if constexpr (sizeof(K) == 4) return __ffs(int(x)) - 1;
    else return __ffsll((long long)x) - 1;
}

//! level of the biggest node that can start at key x, R/sfc/common.hpp:340-346
template<class K>
__device__ __forceinline__ int lastNonZeroPlace(K x)
{
    return x ? int(maxLevel<K>()) - ctzKey(x) / 3 : int(maxLevel<K>());
}

// ---------------------------------------------------------------------------------------------------
// rebalance decision with counts and MAC flags, R/focus/rebalance.hpp:50-88
// ---------------------------------------------------------------------------------------------------
template<class K>
__global__ __launch_bounds__(256) void essentialOpsKernel(const K* __restrict__ prefixes,
                                                          const NodeIdx* __restrict__ childOffsets,
                                                          const NodeIdx* __restrict__ parents,
                                                          const uint32_t* __restrict__ counts,
                                                          const char* __restrict__ macs, K focusStart, K focusEnd,
                                                          uint32_t bucket, NodeIdx* __restrict__ nodeOps,
                                                          NodeIdx numNodes)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numNodes) return;
    const K prefix       = prefixes[i];
    const unsigned level = prefixBits(prefix) / 3;
    int op               = 1;
    bool merged          = false;
    if (i > 0)
    {
        const NodeIdx parent = parents[(i - 1) / 8];
        const K groupStart   = fromPrefix(prefixes[parent]);
        const K groupEnd     = groupStart + 8 * nodeSpan<K>(level);
        // a sibling group that touches the focus is never given up for a passed MAC alone
        const bool fringe = groupEnd > focusStart && focusEnd > groupStart;
        merged            = counts[parent] <= bucket || (macs[parent] == 0 && !fringe);
    }
    if (merged) { op = 0; }
    else
    {
        const K start      = fromPrefix(prefix);
        const bool inFocus = start >= focusStart && start < focusEnd;
        if (childOffsets[i] == 0 && level < maxLevel<K>() && counts[i] > bucket && (macs[i] || inFocus)) op = 8;
    }
    nodeOps[i] = op;
}

//! R/focus/rebalance.hpp:81-88 and rebalance_gpu.cu:88-101
template<class K>
__global__ __launch_bounds__(256) void macRefineOpsKernel(const K* __restrict__ prefixes, const char* __restrict__ macs,
                                                          const NodeIdx* __restrict__ leafToInternal, NodeIdx numLeaves,
                                                          NodeIdx focusFirst, NodeIdx focusLast,
                                                          NodeIdx* __restrict__ nodeOps)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numLeaves) return;
    int op = 1;
    if (i < focusFirst || i >= focusLast)
    {
        NodeIdx n = leafToInternal[i];
        if (prefixBits(prefixes[n]) / 3 < maxLevel<K>() && macs[n]) op = 8;
    }
    nodeOps[i] = op;
}

/*! R/focus/rebalance.hpp:113-184: a node whose op is 0 takes over the op of its closest ancestor with a non-zero op if it
 *  is that ancestor's left-most descendant.  In place like the reference; the outcome does not depend on the order in
 *  which the threads run: ops that are non-zero never change, and a zero op that has already been replaced equals the op
 *  of the ancestor the walk would have reached (same start key), see DESIGN.md */
template<class K>
__global__ __launch_bounds__(256) void protectAncestorsKernel(const K* __restrict__ prefixes,
                                                              const NodeIdx* __restrict__ parents, NodeIdx* nodeOps,
                                                              NodeIdx numNodes, int* __restrict__ numChanged)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    int op    = 1;
    if (i < numNodes)
    {
        volatile NodeIdx* ops = nodeOps;
        NodeIdx a             = i;
        int opA               = ops[a];
        while (opA == 0 && a != 0)
        {
            a   = parents[(a - 1) / 8];
            opA = ops[a];
        }
        op = (a == i || fromPrefix(prefixes[i]) == fromPrefix(prefixes[a])) ? opA : 0;
        if (i == 0) op = opA;
        ops[i] = op;
    }
    uint64_t changed = __ballot(op != 1);
    // (a flag, not a count; set once: tens of thousands of waves updating ONE address serialise in the L2 -- atomics and
    //  plain stores alike, the stores 2-3 times worse --, reading it does not)
    if ((threadIdx.x & 63u) == 0 && changed && *reinterpret_cast<volatile int*>(numChanged) == 0) atomicOr(numChanged, 1);
}

/*! enforceKeySingle, R/focus/rebalance.hpp:199-250, one lane per mandatory key.  status: 0 converged, 1 cancelMerge,
 *  2 rebalance, 3 failed (ResolutionStatus, :186-196) */
template<class K>
__global__ __launch_bounds__(64) void enforceKeysKernel(const K* __restrict__ forcedKeys, NodeIdx numKeys,
                                                        const K* __restrict__ prefixes,
                                                        const NodeIdx* __restrict__ childOffsets,
                                                        const NodeIdx* __restrict__ parents, NodeIdx* nodeOps,
                                                        int* __restrict__ status)
{
    NodeIdx q = blockIdx.x * 64 + threadIdx.x;
    if (q >= numKeys) return;
    const K key = forcedKeys[q];
    if (key == 0 || key == endKey<K>()) return;

    const int wantLevel = lastNonZeroPlace(key);
    const K want        = toPrefix(key, 3 * wantLevel);
    // smallest node that contains the wanted node, R/tree/octree.hpp:245-262
    NodeIdx node = 0;
    for (int l = 1; l <= wantLevel; ++l)
    {
        if (childOffsets[node] == 0 || prefixes[node] == want) break;
        node = childOffsets[node] + NodeIdx(octDigit(key, unsigned(l)));
    }
    const K have        = prefixes[node];
    const int haveLevel = int(prefixBits(have) / 3);

    int st              = 0;
    const bool trySplit = have != want && haveLevel < int(maxLevel<K>());
    const bool undo     = *(volatile NodeIdx*)(nodeOps + node) == 0 || trySplit;
    if (undo && node > 0)
    {
        st        = 1;
        NodeIdx p = node;
        do
        {
            p             = parents[(p - 1) / 8];
            NodeIdx first = childOffsets[p];
            for (int s = 0; s < 8; ++s)
                atomicMax(nodeOps + first + s, 1); // a 0 (merge) becomes 1 (keep)
        } while (p != 0);
    }
    if (trySplit)
    {
        int levelDiff = wantLevel - haveLevel;
        st            = levelDiff > 1 ? 3 : 2; // only one level is ever added, :234-243
        levelDiff     = min(levelDiff, 1);
        atomicMax(nodeOps + node, 1 << (3 * levelDiff));
    }
    if (st) atomicMax(status, st);
}

/*! rangeCount, R/focus/rebalance.hpp:279-301: countsFocus[leaf] = saturated sum of the global counts of the global leaves
 *  under focus leaf `leaf`; 16 lanes per listed leaf */
template<class K>
__global__ __launch_bounds__(256) void rangeCountKernel(const K* __restrict__ leaves, NodeIdx numLeavesPlus1,
                                                        const uint32_t* __restrict__ counts,
                                                        const K* __restrict__ leavesFocus,
                                                        const NodeIdx* __restrict__ focusIdx, NodeIdx numIdx,
                                                        uint32_t* __restrict__ countsFocus)
{
    const unsigned sub = threadIdx.x & 15u;
    NodeIdx q          = NodeIdx(blockIdx.x) * 16 + NodeIdx(threadIdx.x >> 4);
    if (q >= numIdx) return; // whole 16-lane groups leave together
    const NodeIdx leaf = focusIdx[q];
    const K startKey = leavesFocus[leaf], endK = leavesFocus[leaf + 1];
    // the reference searches the whole span it is handed (the leaf keys INCLUDING the terminal key):
    // first = upper_bound(startKey) - 1, last = lower_bound(endKey), R/tree/csarray.hpp:78-90
    NodeIdx lo = 0, len = numLeavesPlus1;
    while (len > 0)
    {
        NodeIdx half = len >> 1;
        if (!(startKey < leaves[lo + half])) { lo += half + 1, len -= half + 1; }
        else { len = half; }
    }
    const NodeIdx first = lo - 1;
    lo = 0, len = numLeavesPlus1;
    while (len > 0)
    {
        NodeIdx half = len >> 1;
        if (leaves[lo + half] < endK) { lo += half + 1, len -= half + 1; }
        else { len = half; }
    }
    const NodeIdx last = lo;
    uint64_t sum       = 0;
    for (NodeIdx i = first + NodeIdx(sub); i < last; i += 16)
        sum += counts[i];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
        sum += __shfl_xor(sum, o);
    if (sub == 0) countsFocus[leaf] = uint32_t(sum < 0xFFFFFFFFull ? sum : 0xFFFFFFFFull);
}

// ---------------------------------------------------------------------------------------------------
// spanSfcRange, R/sfc/common.hpp:370-438: the keys of the coarsest valid cornerstone sub-tree covering [a, b)
// ---------------------------------------------------------------------------------------------------
template<class K, bool STORE>
__device__ __forceinline__ int spanRange(K a, K b, K* out)
{
    int num             = 0;
    const int firstDiff = (clzKey(K(a ^ b)) + 3 - int(KeyInfo<K>::spare)) / 3;
    const int lastA = lastNonZeroPlace(a), lastB = lastNonZeroPlace(b);
    for (int pos = lastA; pos > firstDiff; --pos)
    {
        int digits = (8 - int(octDigit(a, unsigned(pos)))) % 8;
        num += digits;
        while (digits--)
        {
            if (STORE) *out++ = a;
            a += nodeSpan<K>(unsigned(pos));
        }
    }
    for (int pos = firstDiff; pos <= lastB; ++pos)
    {
        int digits = int(octDigit(b, unsigned(pos))) - int(octDigit(a, unsigned(pos)));
        num += digits;
        while (digits--)
        {
            if (STORE) *out++ = a;
            a += nodeSpan<K>(unsigned(pos));
        }
    }
    return num;
}

template<class K>
__global__ __launch_bounds__(256) void countGapsKernel(const K* __restrict__ tree, NodeIdx numNodes,
                                                       NodeIdx* __restrict__ ops)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i < numNodes) ops[i] = spanRange<K, false>(tree[i], tree[i + 1], nullptr);
}

template<class K>
__global__ __launch_bounds__(256) void fillGapsKernel(const K* __restrict__ tree, NodeIdx numNodes,
                                                      const NodeIdx* __restrict__ ops, K* __restrict__ newTree)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i < numNodes) spanRange<K, true>(tree[i], tree[i + 1], newTree + ops[i]);
    if (i == numNodes) newTree[ops[i]] = tree[numNodes];
}

// ---------------------------------------------------------------------------------------------------
// node geometry in floating point (compiled with -ffp-contract=off: two roundings like the CPU path)
// ---------------------------------------------------------------------------------------------------
template<class K, bool HILBERT>
__device__ __forceinline__ void cornerOf(K key, unsigned level, const uint16_t* dec, int (&c)[3])
{
    K morton = key;
    if (HILBERT)
    {
        morton         = 0;
        unsigned state = 0;
        for (unsigned l = 1; l <= level; ++l)
        {
            unsigned e = dec[state * 8 + octDigit(key, l)];
            morton |= K(e & 7u) << (3u * (maxLevel<K>() - l));
            state = e >> 3;
        }
    }
    unsigned ix, iy, iz;
    mortonDecode<K>(morton, ix, iy, iz);
    c[0] = int(ix), c[1] = int(iy), c[2] = int(iz);
}

//! centerAndSize, R/sfc/box.hpp:335-352
template<class K, class T>
__device__ __forceinline__ void centerAndSize(const int (&lo)[3], const int (&hi)[3], const DBox<T>& box, T (&center)[3],
                                              T (&size)[3])
{
    constexpr int g = 1 << maxLevel<K>();
    constexpr T uL  = T(1.) / g;
#pragma unroll
    for (int d = 0; d < 3; ++d)
    {
        T half    = T(0.5) * uL * box.len[d];
        center[d] = box.lo[d] + T(hi[d] + lo[d]) * half;
        size[d]   = T(hi[d] - lo[d]) * half;
    }
}

template<class T>
__device__ __forceinline__ T max3(const T (&v)[3])
{
    T m = v[0] > v[1] ? v[0] : v[1];
    return m > v[2] ? m : v[2];
}

/*! MODE 0: geoMacSpheres = computeMinMacR2, R/traversal/macs.hpp:44-58: (geometric centre, (2 max(size) invTheta)^2)
 *  MODE 1: setMac = computeVecMacR2, :69-85 + R/focus/source_center.hpp:118-131: sphere[3] <- (2 max(size) invTheta +
 *          |centre of mass - geometric centre|)^2, or 0 for an empty node (mass 0) */
template<class K, class T, bool HILBERT, int MODE>
__global__ __launch_bounds__(256) void macSpheresKernel(const K* __restrict__ prefixes, NodeIdx numNodes,
                                                        T* __restrict__ spheres, float invTheta, DBox<T> box,
                                                        const uint16_t* __restrict__ decTable)
{
    __shared__ uint16_t dec[24 * 8];
    if (threadIdx.x < 24 * 8) dec[threadIdx.x] = decTable[threadIdx.x];
    __syncthreads();
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numNodes) return;
    const K prefix       = prefixes[i];
    const unsigned level = prefixBits(prefix) / 3;
    int lo[3], hi[3];
    cornerOf<K, HILBERT>(fromPrefix(prefix), level, dec, lo);
    const int edge = 1 << (maxLevel<K>() - level);
#pragma unroll
    for (int d = 0; d < 3; ++d)
        hi[d] = lo[d] + edge;
    T c[3], s[3];
    centerAndSize<K, T>(lo, hi, box, c, s);
    const T l = T(2) * max3(s);
    if (MODE == 0)
    {
        const T mac    = l * invTheta;
        spheres[4 * i] = c[0], spheres[4 * i + 1] = c[1], spheres[4 * i + 2] = c[2];
        spheres[4 * i + 3] = mac * mac;
    }
    else
    {
        const T m   = spheres[4 * i + 3];
        const T dx  = spheres[4 * i] - c[0], dy = spheres[4 * i + 1] - c[1], dz = spheres[4 * i + 2] - c[2];
        const T dist = sqrt(dx * dx + (dy * dy + dz * dz)); // fold order of util::dot, R/util/array.hpp:252-256
        const T mac  = l * invTheta + dist;
        spheres[4 * i + 3] = (m != T(0)) ? mac * mac : T(0);
    }
}

//! R/focus/source_center_gpu.cu:204-212
//! FocusedOctree::addMacs (R/focus/octree_focus_mpi.hpp:601-610): a leaf whose node fails the MAC counts as a halo leaf
__global__ __launch_bounds__(256) void addMacsKernel(const char* __restrict__ macs, const NodeIdx* __restrict__ leafToInternal,
                                                     NodeIdx numLeaves, int32_t* __restrict__ haloFlags)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numLeaves) return;
    if (macs[leafToInternal[i]] && !haloFlags[i]) haloFlags[i] = 1;
}

template<class T>
__global__ __launch_bounds__(256) void moveCentersKernel(const T* __restrict__ src, NodeIdx numNodes,
                                                         T* __restrict__ dst)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numNodes) return;
    dst[4 * i] = src[3 * i], dst[4 * i + 1] = src[3 * i + 1], dst[4 * i + 2] = src[3 * i + 2];
    dst[4 * i + 3] = T(1.0);
}

/*! massCenter per leaf, R/focus/source_center.hpp:44-77: centre = sum(|m| r) / sum(|m|), accumulated in particle order
 *  in Tf like the reference's serial loop (a different association would not be bit-identical) */
template<class Tc, class Tm, class Tf>
__global__ __launch_bounds__(256) void leafCentersKernel(const Tc* __restrict__ x, const Tc* __restrict__ y,
                                                         const Tc* __restrict__ z, const Tm* __restrict__ m,
                                                         const NodeIdx* __restrict__ leafToInternal, NodeIdx numLeaves,
                                                         const uint32_t* __restrict__ layout, Tf* __restrict__ centers)
{
    NodeIdx leaf = blockIdx.x * 256 + threadIdx.x;
    if (leaf >= numLeaves) return;
    Tf cx = 0, cy = 0, cz = 0, cm = 0;
    for (uint32_t i = layout[leaf]; i < layout[leaf + 1]; ++i)
    {
        Tf w = fabs(Tf(m[i]));
        cx += w * Tf(x[i]);
        cy += w * Tf(y[i]);
        cz += w * Tf(z[i]);
        cm += w;
    }
    Tf inv    = (cm != Tf(0.0)) ? Tf(1.0) / cm : Tf(1.0);
    NodeIdx n = leafToInternal[leaf];
    centers[4 * n] = cx * inv, centers[4 * n + 1] = cy * inv, centers[4 * n + 2] = cz * inv, centers[4 * n + 3] = cm;
}

//! CombineSourceCenter, R/focus/source_center.hpp:79-95, one level of the upsweep
template<class T>
__global__ __launch_bounds__(256) void upsweepCentersKernel(NodeIdx firstCell, NodeIdx lastCell,
                                                            const NodeIdx* __restrict__ childOffsets, T* centers)
{
    NodeIdx cell = firstCell + blockIdx.x * 256 + threadIdx.x;
    if (cell >= lastCell) return;
    NodeIdx child = childOffsets[cell];
    if (!child) return;
    T cx = 0, cy = 0, cz = 0, cm = 0;
    for (int k = 0; k < 8; ++k)
    {
        const T* s = centers + 4 * size_t(child + k);
        T w        = fabs(s[3]);
        cx += w * s[0];
        cy += w * s[1];
        cz += w * s[2];
        cm += w;
    }
    T inv = (cm != T(0.0)) ? T(1.0) / cm : T(1.0);
    centers[4 * size_t(cell)] = cx * inv, centers[4 * size_t(cell) + 1] = cy * inv;
    centers[4 * size_t(cell) + 2] = cz * inv, centers[4 * size_t(cell) + 3] = cm;
}

// ---------------------------------------------------------------------------------------------------
// markMacs, R/traversal/macs.hpp:199-270
// ---------------------------------------------------------------------------------------------------
constexpr int MAC_WAVES     = 4;
constexpr int MAC_STACK_CAP = 1024;

template<class K, class T, bool HILBERT>
__global__ __launch_bounds__(MAC_WAVES * 64) void markMacsKernel(
    const K* __restrict__ prefixes, const NodeIdx* __restrict__ childOffsets, const T* __restrict__ centers,
    DBox<T> box, const K* __restrict__ focusNodes, NodeIdx numFocusNodes, bool limitSource, char* markings,
    const uint16_t* __restrict__ tables, int* __restrict__ errors)
{
    __shared__ uint16_t enc[24 * 8];
    __shared__ uint16_t dec[24 * 8];
    __shared__ NodeIdx stacks[MAC_WAVES][MAC_STACK_CAP];
    if (threadIdx.x < 24 * 8)
    {
        enc[threadIdx.x] = tables[threadIdx.x];
        dec[threadIdx.x] = tables[48 * 8 + threadIdx.x];
    }
    __syncthreads();

    constexpr int R     = 1 << maxLevel<K>();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    NodeIdx* stack      = stacks[wave];
    const K focusStart = focusNodes[0], focusEnd = focusNodes[numFocusNodes];

    // ---- per lane: one target cell, its centre and size, and the "nothing outside the focus can be near" rejection
    const NodeIdx target = NodeIdx(blockIdx.x * (MAC_WAVES * 64) + threadIdx.x);
    bool active          = target < numFocusNodes;
    T tc[3] = {0, 0, 0}, ts[3] = {0, 0, 0};
    unsigned maxSourceLevel = maxLevel<K>();
    if (active)
    {
        const K start = focusNodes[target], end = focusNodes[target + 1];
        const unsigned level = levelOfSpan<K>(end - start);
        int lo[3], hi[3];
        cornerOf<K, HILBERT>(start, level, dec, lo);
        const int edge = 1 << (maxLevel<K>() - level);
        int elo[3], ehi[3];
#pragma unroll
        for (int d = 0; d < 3; ++d)
        {
            hi[d]  = lo[d] + edge;
            elo[d] = lo[d] - 1, ehi[d] = hi[d] + 1;
        }
        bool inside; // containedIn(focusStart, focusEnd, targetExt), R/traversal/boxoverlap.hpp:95-115
        if (min(min(elo[0], elo[1]), elo[2]) < 0 || max(max(ehi[0], ehi[1]), ehi[2]) > R)
        {
            inside = focusStart == 0 && focusEnd == endKey<K>();
        }
        else
        {
            K m0 = mortonEncode<K>(unsigned(elo[0]), unsigned(elo[1]), unsigned(elo[2]));
            K m1 = mortonEncode<K>(unsigned(ehi[0] - 1), unsigned(ehi[1] - 1), unsigned(ehi[2] - 1));
            K kLo = HILBERT ? hilbertFromMorton<K>(m0, enc) : m0;
            K kHi = HILBERT ? hilbertFromMorton<K>(m1, enc) : m1;
            unsigned common = unsigned(sharedPrefixBits<K>(kLo, kHi)) / 3u;
            K nodeStart     = kLo & ~K(nodeSpan<K>(common) - 1);
            inside          = nodeStart >= focusStart && nodeStart + nodeSpan<K>(common) <= focusEnd;
        }
        active = !inside;
        centerAndSize<K, T>(lo, hi, box, tc, ts);
        if (limitSource) maxSourceLevel = unsigned(max(int(level) - 1, 0));
    }

    T pbcLen[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        pbcLen[d] = T(box.bc[d] == 1) * box.len[d];

    uint64_t todo = __ballot(active);
    while (todo)
    {
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        T c[3], s[3];
#pragma unroll
        for (int d = 0; d < 3; ++d)
        {
            c[d] = __shfl(tc[d], src);
            s[d] = __shfl(ts[d], src);
        }
        const unsigned maxSrc = unsigned(__shfl(int(maxSourceLevel), src));

        // continuation criterion of markMacPerBox, :148-168; marks the node when it fails the MAC
        auto violates = [&](NodeIdx n) -> bool
        {
            const K prefix       = prefixes[n];
            const unsigned level = prefixBits(prefix) / 3;
            const K start        = fromPrefix(prefix);
            const K end          = start + nodeSpan<K>(level);
            if (!(start < focusStart || end > focusEnd)) return false; // inside the focus: never a remote source
            const T* sc = centers + 4 * size_t(n);
            T dX[3];
#pragma unroll
            for (int d = 0; d < 3; ++d)
            {
                T dx = c[d] - sc[d];
                dx -= pbcLen[d] * rint(dx * box.inv[d]); // applyPbc, R/sfc/box.hpp:195-206
                dx = fabs(dx);
                dx -= s[d];
                dx += fabs(dx);
                dx *= T(0.5);
                dX[d] = dx;
            }
            const T r2 = dX[0] * dX[0] + (dX[1] * dX[1] + dX[2] * dX[2]);
            const bool v = r2 < fabs(sc[3]) && level <= maxSrc;
            if (v && !markings[n]) markings[n] = 1;
            return v;
        };

        int top = 0;
        {
            bool go = false;
            if (lane == 0) go = violates(0);
            go = __shfl(int(go), 0);
            if (!go || childOffsets[0] == 0) continue;
            if (lane == 0) stack[0] = 0;
            top = 1;
        }
        while (top > 0)
        {
            const int take  = min(top, 8);
            const int slot  = int(lane >> 3);
            const bool mine = slot < take;
            NodeIdx par     = mine ? stack[top - 1 - slot] : 0;
            top -= take;
            NodeIdx child = 0;
            bool push     = false;
            if (mine)
            {
                child = childOffsets[par] + NodeIdx(lane & 7u);
                push  = violates(child) && childOffsets[child] != 0;
            }
            const uint64_t pm = __ballot(push);
            const int numPush = __popcll(pm);
            if (top + numPush > MAC_STACK_CAP)
            {
                if (lane == 0) atomicOr(errors, 4);
                top = 0;
                break;
            }
            if (push) stack[top + __popcll(pm & ((1ull << lane) - 1ull))] = child;
            top += numPush;
        }
    }
}

template<class K, class T>
int markMacs(cstone_hip_ctx* ctx, int curve, const void* prefixes, const int32_t* childOffsets, const void* centers,
             const cstone_box& box, const void* focusNodes, int numFocusNodes, int limitSource, char* markings)
{
    auto* tables  = (const uint16_t*)ctx->hilbertTables;
    int* errors   = ctx->devScalars + 63;
    unsigned grid = gridFor(size_t(numFocusNodes), MAC_WAVES * 64);
    DBox<T> b     = makeDBox<T>(box);
    if (curve == CSTONE_HILBERT)
        hipLaunchKernelGGL((markMacsKernel<K, T, true>), grid, MAC_WAVES * 64, 0, ctx->stream, (const K*)prefixes,
                           childOffsets, (const T*)centers, b, (const K*)focusNodes, numFocusNodes, limitSource != 0,
                           markings, tables, errors);
    else
        hipLaunchKernelGGL((markMacsKernel<K, T, false>), grid, MAC_WAVES * 64, 0, ctx->stream, (const K*)prefixes,
                           childOffsets, (const T*)centers, b, (const K*)focusNodes, numFocusNodes, limitSource != 0,
                           markings, tables, errors);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

template<class K, class T, int MODE>
int macSpheres(cstone_hip_ctx* ctx, int curve, const void* prefixes, int numNodes, void* spheres, float invTheta,
               const cstone_box& box)
{
    auto* dec     = (const uint16_t*)ctx->hilbertTables + 48 * 8;
    unsigned grid = gridFor(size_t(numNodes), 256);
    DBox<T> b     = makeDBox<T>(box);
    if (curve == CSTONE_HILBERT)
        hipLaunchKernelGGL((macSpheresKernel<K, T, true, MODE>), grid, 256, 0, ctx->stream, (const K*)prefixes, numNodes,
                           (T*)spheres, invTheta, b, dec);
    else
        hipLaunchKernelGGL((macSpheresKernel<K, T, false, MODE>), grid, 256, 0, ctx->stream, (const K*)prefixes,
                           numNodes, (T*)spheres, invTheta, b, dec);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

// ---------------------------------------------------------------------------------------------------
// findPeersMac, R/traversal/peers.hpp:63-118: dual traversal (R/traversal/traversal.hpp:135-188) of the replicated
// global tree against itself, started from each of the nodes that span the rank's key range.  One wave per start
// node; up to 8 node pairs are popped per step and their 64 child pairs tested one per lane (LDS stack of pairs), like
// markMacs above.  The set of pairs visited is that of the reference's serial walk (a pair is expanded iff it was
// reached and fails the MAC), so the order in which they are visited does not matter: the result is a set of flags.
// ---------------------------------------------------------------------------------------------------
constexpr int PEER_STACK_CAP = 4096;

template<class K, class T, bool HILBERT>
__global__ __launch_bounds__(64) void findPeersKernel(const K* __restrict__ prefixes,
                                                      const NodeIdx* __restrict__ childOffsets,
                                                      const NodeIdx* __restrict__ levelRange,
                                                      const K* __restrict__ spanKeys, int numSpan,
                                                      const K* __restrict__ assignment, int numRanks, DBox<T> box,
                                                      float invThetaEff, int32_t* __restrict__ peerFlags,
                                                      const uint16_t* __restrict__ tables, int* __restrict__ errors)
{
    __shared__ uint16_t dec[24 * 8];
    __shared__ NodeIdx stackA[PEER_STACK_CAP], stackB[PEER_STACK_CAP];
    for (unsigned i = threadIdx.x; i < 24 * 8; i += 64)
        dec[i] = tables[48 * 8 + i];
    __syncthreads();
    const unsigned lane = threadIdx.x;
    const int span      = blockIdx.x;
    if (span >= numSpan) return;
    const K domainStart = spanKeys[0], domainEnd = spanKeys[numSpan];

    T pbcLen[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        pbcLen[d] = T(box.bc[d] == 1) * box.len[d];

    auto geometry = [&](NodeIdx n, T (&c)[3], T (&s)[3])
    {
        const K prefix       = prefixes[n];
        const unsigned level = prefixBits(prefix) / 3;
        int lo[3], hi[3];
        cornerOf<K, HILBERT>(fromPrefix(prefix), level, dec, lo);
        const int edge = 1 << (maxLevel<K>() - level);
#pragma unroll
        for (int d = 0; d < 3; ++d)
            hi[d] = lo[d] + edge;
        centerAndSize<K, T>(lo, hi, box, c, s);
    };
    // minDistance(X, bCenter, bSize, box), R/traversal/boxoverlap.hpp:208-218, squared norm
    auto dist2 = [&](const T (&X)[3], const T (&bc)[3], const T (&bs)[3])
    {
        T dX[3];
#pragma unroll
        for (int d = 0; d < 3; ++d)
        {
            T dx = bc[d] - X[d];
            dx -= pbcLen[d] * rint(dx * box.inv[d]);
            dx = fabs(dx);
            dx -= bs[d];
            dx += fabs(dx);
            dx *= T(0.5);
            dX[d] = dx;
        }
        return dX[0] * dX[0] + (dX[1] * dX[1] + dX[2] * dX[2]);
    };
    // crossFocusPairs, peers.hpp:72-86: true = the pair fails the MAC and is followed
    auto follows = [&](NodeIdx a, NodeIdx b) -> bool
    {
        const K pa = prefixes[a], pb = prefixes[b];
        const K aStart = fromPrefix(pa), aEnd = aStart + nodeSpan<K>(prefixBits(pa) / 3);
        const K bStart = fromPrefix(pb), bEnd = bStart + nodeSpan<K>(prefixBits(pb) / 3);
        const bool aFocusOverlap = domainStart < aEnd && aStart < domainEnd; // overlapTwoRanges
        const bool bInFocus      = bStart >= domainStart && bEnd <= domainEnd; // containedIn
        if (!aFocusOverlap || bInFocus) return false;
        T ac[3], as[3], bc[3], bs[3];
        geometry(a, ac, as);
        geometry(b, bc, bs);
        // minVecMacMutual, R/traversal/macs.hpp:171-194
        const T macA = max3(bs) * 2 * invThetaEff;
        const bool passA = dist2(bc, ac, as) > macA * macA;
        const T macB = max3(as) * 2 * invThetaEff;
        const bool passB = dist2(ac, bc, bs) > macB * macB;
        return !(passA && passB);
    };
    auto mark = [&](NodeIdx b)
    {
        const K key = fromPrefix(prefixes[b]);
        int lo = 0, len = numRanks + 1; // findRank: upper_bound(assignment, key) - 1
        while (len > 0)
        {
            int half   = len >> 1;
            bool right = assignment[lo + half] <= key;
            lo         = right ? lo + half + 1 : lo;
            len        = right ? len - half - 1 : half;
        }
        const int rank = lo - 1;
        if (rank >= 0 && rank < numRanks) peerFlags[rank] = 1;
    };

    // the start node: locateNode(spanKeys[span], spanKeys[span + 1]), R/tree/octree.hpp:216-241
    NodeIdx a0 = 0;
    {
        const K start = spanKeys[span], end = spanKeys[span + 1];
        const int bits       = clzKey(K(end - start - 1)) - int(KeyInfo<K>::spare);
        const unsigned level = unsigned(bits) / 3u;
        const K want         = toPrefix<K>(start, bits);
        NodeIdx lo = levelRange[level], len = levelRange[level + 1] - lo;
        while (len > 0)
        {
            NodeIdx half = len >> 1;
            bool right   = prefixes[lo + half] < want;
            lo           = right ? lo + half + 1 : lo;
            len          = right ? len - half - 1 : half;
        }
        a0 = lo;
        if (lo >= levelRange[maxLevel<K>() + 1] || prefixes[lo] != want)
        {
            if (lane == 0) atomicOr(errors, 8);
            return;
        }
    }
    if (childOffsets[a0] == 0 && childOffsets[0] == 0)
    {
        if (lane == 0 && follows(a0, 0)) mark(0);
        return;
    }
    int top = 1;
    if (lane == 0) stackA[0] = a0, stackB[0] = 0;
    while (top > 0)
    {
        const int take  = min(top, 8);
        const int slot  = int(lane >> 3);
        const bool mine = slot < take;
        NodeIdx ta = 0, tb = 0;
        if (mine) ta = stackA[top - 1 - slot], tb = stackB[top - 1 - slot];
        top -= take;
        bool push  = false;
        NodeIdx na = 0, nb = 0;
        if (mine)
        {
            const bool leafT = childOffsets[ta] == 0, leafS = childOffsets[tb] == 0;
            const unsigned levelT = prefixBits(prefixes[ta]) / 3, levelS = prefixBits(prefixes[tb]) / 3;
            const bool splitTarget = (levelT < levelS && !leafT) || leafS;
            bool have = false;
            if (splitTarget)
            {
                if (!leafT) na = childOffsets[ta] + NodeIdx(lane & 7u), nb = tb, have = true;
            }
            else if (!leafS) { na = ta, nb = childOffsets[tb] + NodeIdx(lane & 7u), have = true; }
            if (have && follows(na, nb))
            {
                if (childOffsets[na] == 0 && childOffsets[nb] == 0) mark(nb);
                else push = true;
            }
        }
        const uint64_t pm = __ballot(push);
        const int numPush = __popcll(pm);
        if (top + numPush > PEER_STACK_CAP)
        {
            if (lane == 0) atomicOr(errors, 8);
            return;
        }
        if (push)
        {
            const int at = top + __popcll(pm & ((1ull << lane) - 1ull));
            stackA[at] = na, stackB[at] = nb;
        }
        top += numPush;
        __builtin_amdgcn_wave_barrier();
    }
}

//! the keys of the coarsest nodes that cover [a, b) (spanSfcRange, R/sfc/common.hpp:376-438), b appended
template<class K>
std::vector<K> spanningKeys(uint64_t a, uint64_t b)
{
    std::vector<K> out;
    const uint64_t end = uint64_t(endKey<K>());
    while (a < b)
    {
        uint64_t size = end;
        while (size > 1 && (a % size != 0 || size > b - a))
            size /= 8;
        out.push_back(K(a));
        a += size;
    }
    out.push_back(K(b));
    return out;
}

template<class K, class T>
int findPeers(cstone_hip_ctx* ctx, int curve, const void* prefixes, const int32_t* childOffsets,
              const int32_t* levelRange, const uint64_t* assignment, int numRanks, int myRank, const cstone_box& box,
              float invThetaEff, int32_t* peerFlagsHost)
{
    std::vector<K> span = spanningKeys<K>(assignment[myRank], assignment[myRank + 1]);
    const int numSpan   = int(span.size()) - 1;
    std::fill(peerFlagsHost, peerFlagsHost + numRanks, 0);
    if (numSpan <= 0) return CSTONE_OK;
    std::vector<K> asg(numRanks + 1);
    for (int r = 0; r <= numRanks; ++r)
        asg[r] = K(assignment[r]);
    const size_t spanBytes = alignUp(span.size() * sizeof(K)), asgBytes = alignUp(asg.size() * sizeof(K));
    const size_t flagBytes = alignUp(size_t(numRanks) * 4);
    CS_TRY(arenaReserve(ctx, spanBytes + asgBytes + flagBytes + 1024));
    K* dSpan    = (K*)arenaTake(ctx, spanBytes);
    K* dAsg     = (K*)arenaTake(ctx, asgBytes);
    auto* dFlag = (int32_t*)arenaTake(ctx, flagBytes);
    auto body   = [&]() -> int
    {
        CS_HIP(ctx, hipMemcpyAsync(dSpan, span.data(), span.size() * sizeof(K), hipMemcpyHostToDevice, ctx->stream));
        CS_HIP(ctx, hipMemcpyAsync(dAsg, asg.data(), asg.size() * sizeof(K), hipMemcpyHostToDevice, ctx->stream));
        CS_HIP(ctx, hipMemsetAsync(dFlag, 0, size_t(numRanks) * 4, ctx->stream));
        auto* tables  = (const uint16_t*)ctx->hilbertTables;
        int* errors   = ctx->devScalars + 63;
        DBox<T> b     = makeDBox<T>(box);
        if (curve == CSTONE_HILBERT)
            hipLaunchKernelGGL((findPeersKernel<K, T, true>), unsigned(numSpan), 64, 0, ctx->stream, (const K*)prefixes,
                               childOffsets, levelRange, dSpan, numSpan, dAsg, numRanks, b, invThetaEff, dFlag, tables,
                               errors);
        else
            hipLaunchKernelGGL((findPeersKernel<K, T, false>), unsigned(numSpan), 64, 0, ctx->stream, (const K*)prefixes,
                               childOffsets, levelRange, dSpan, numSpan, dAsg, numRanks, b, invThetaEff, dFlag, tables,
                               errors);
        CS_HIP(ctx, hipGetLastError());
        // the traversal reports an overflow of its pair stack through the sticky error word: read it with the flags -- an
        // incomplete (and then not mutual) peer list must not reach the treelet and count exchanges that follow
        CS_TRY(copyToPinned(ctx, ctx->hostScalars + 63, errors, sizeof(int)));
        CS_TRY(copyToHost(ctx, peerFlagsHost, dFlag, size_t(numRanks) * 4)); // (synchronises the stream)
        if (ctx->hostScalars[63] != 0)
            return fail(ctx, CSTONE_E_INTERNAL, "find_peers_mac: device-side check failed, code 0x%x (peer list incomplete)",
                        unsigned(ctx->hostScalars[63]));
        return CSTONE_OK;
    };
    int rc = body();
    arenaReset(ctx);
    return rc;
}

//! leafOps[i] = opsAll[leafToInternal[i]] for the leaves in leaf order (leafOps[numLeaves] = 0); changed += leaves whose
//! op is not "keep"
__global__ __launch_bounds__(256) void leafOpsCountKernel(const NodeIdx* __restrict__ opsAll,
                                                          const NodeIdx* __restrict__ leafToInternal, NodeIdx numLeaves,
                                                          uint32_t* __restrict__ leafOps, int* __restrict__ changed)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    int op    = 1;
    if (i < numLeaves)
    {
        op         = opsAll[leafToInternal[i]];
        leafOps[i] = uint32_t(op);
    }
    else if (i == numLeaves) { leafOps[i] = 0; }
    const uint64_t b = __ballot(op != 1);
    if ((threadIdx.x & 63u) == 0 && b && *reinterpret_cast<volatile int*>(changed) == 0) atomicOr(changed, 1); // (a flag, see protectAncestorsKernel)
}

//! one int from the device scalars to the host (synchronises the stream)
int readScalar(cstone_hip_ctx* ctx, int slot, int* out)
{
    CS_TRY(copyToPinned(ctx, ctx->hostScalars + slot, ctx->devScalars + slot, sizeof(int)));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = ctx->hostScalars[slot];
    return CSTONE_OK;
}

bool badKeyBits(int kb) { return kb != 32 && kb != 64; }
bool badCurve(int c) { return c != CSTONE_MORTON && c != CSTONE_HILBERT; }

} // namespace

} // namespace cship

using namespace cship;

#define CSTONE_KEY_SWITCH(key_bits, CALL)                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        if ((key_bits) == 32) { using K = uint32_t; CALL; }                                                            \
        else { using K = uint64_t; CALL; }                                                                             \
    } while (0)

extern "C"
{

int cstone_hip_rebalance_decision_essential(cstone_hip_ctx* ctx, int key_bits, const void* prefixes,
                                            const int32_t* child_offsets, const int32_t* parents,
                                            const uint32_t* counts, const char* macs, uint64_t focus_start,
                                            uint64_t focus_end, uint32_t bucket_size, int32_t* node_ops, int num_nodes)
{
    if (!ctx || badKeyBits(key_bits) || num_nodes < 0 ||
        (num_nodes && (!prefixes || !child_offsets || !counts || !macs || !node_ops)) || (num_nodes > 1 && !parents))
        return fail(ctx, CSTONE_E_ARG, "rebalance_decision_essential: bad argument");
    if (num_nodes == 0) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_REBALANCE);
    CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(essentialOpsKernel<K>, gridFor(size_t(num_nodes), 256), 256, 0,
                                                   ctx->stream, (const K*)prefixes, child_offsets, parents, counts, macs,
                                                   K(focus_start), K(focus_end), bucket_size, node_ops, num_nodes));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_mac_refine_decision(cstone_hip_ctx* ctx, int key_bits, const void* prefixes, const char* macs,
                                   const int32_t* leaf_to_internal, int num_leaves, int focus_first, int focus_last,
                                   int32_t* node_ops)
{
    if (!ctx || badKeyBits(key_bits) || num_leaves < 0 ||
        (num_leaves && (!prefixes || !macs || !leaf_to_internal || !node_ops)))
        return fail(ctx, CSTONE_E_ARG, "mac_refine_decision: bad argument");
    if (num_leaves == 0) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_REBALANCE);
    CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(macRefineOpsKernel<K>, gridFor(size_t(num_leaves), 256), 256, 0,
                                                   ctx->stream, (const K*)prefixes, macs, leaf_to_internal, num_leaves,
                                                   focus_first, focus_last, node_ops));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_protect_ancestors(cstone_hip_ctx* ctx, int key_bits, const void* prefixes, const int32_t* parents,
                                 int32_t* node_ops, int num_nodes, int* converged_host)
{
    if (!ctx || badKeyBits(key_bits) || num_nodes < 0 || !converged_host ||
        (num_nodes && (!prefixes || !node_ops)) || (num_nodes > 1 && !parents))
        return fail(ctx, CSTONE_E_ARG, "protect_ancestors: bad argument");
    *converged_host = 1;
    if (num_nodes == 0) return CSTONE_OK;
    int* counter = ctx->devScalars + 10;
    CS_HIP(ctx, hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));
    {
        StageTimer timer(ctx, CSTONE_STAGE_REBALANCE);
        CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(protectAncestorsKernel<K>, gridFor(size_t(num_nodes), 256), 256,
                                                       0, ctx->stream, (const K*)prefixes, parents, node_ops, num_nodes,
                                                       counter));
    }
    CS_HIP(ctx, hipGetLastError());
    int changed = 0;
    CS_TRY(readScalar(ctx, 10, &changed));
    *converged_host = changed == 0;
    return CSTONE_OK;
}

int cstone_hip_enforce_keys(cstone_hip_ctx* ctx, int key_bits, const void* forced_keys, int num_forced_keys,
                            const void* prefixes, const int32_t* child_offsets, const int32_t* parents,
                            int32_t* node_ops, int* status_host)
{
    if (!ctx || badKeyBits(key_bits) || num_forced_keys < 0 || !status_host ||
        (num_forced_keys && (!forced_keys || !prefixes || !child_offsets || !node_ops)))
        return fail(ctx, CSTONE_E_ARG, "enforce_keys: bad argument");
    *status_host = 0;
    if (num_forced_keys == 0) return CSTONE_OK;
    int* status = ctx->devScalars + 11;
    CS_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int), ctx->stream));
    {
        StageTimer timer(ctx, CSTONE_STAGE_REBALANCE);
        CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(enforceKeysKernel<K>, gridFor(size_t(num_forced_keys), 64), 64, 0,
                                                       ctx->stream, (const K*)forced_keys, num_forced_keys,
                                                       (const K*)prefixes, child_offsets, parents, node_ops, status));
    }
    CS_HIP(ctx, hipGetLastError());
    return readScalar(ctx, 11, status_host);
}

int cstone_hip_focus_update_ops(cstone_hip_ctx* ctx, int key_bits, const void* prefixes, const int32_t* child_offsets,
                                const int32_t* parents, const uint32_t* counts, const char* macs, uint64_t focus_start,
                                uint64_t focus_end, uint32_t bucket_size, const void* forced_keys, int num_forced_keys,
                                const int32_t* leaf_to_internal, int num_leaves, int num_nodes, int32_t* node_ops_all,
                                int32_t* leaf_ops, int* result_host)
{
    if (!ctx || badKeyBits(key_bits) || num_leaves < 1 || num_nodes < num_leaves || num_forced_keys < 0 || !prefixes ||
        !child_offsets || !counts || !macs || !leaf_to_internal || !node_ops_all || !leaf_ops || !result_host ||
        (num_forced_keys && !forced_keys) || (num_nodes > 1 && !parents))
        return fail(ctx, CSTONE_E_ARG, "focus_update_ops: bad argument");
    // devScalars 10: nodes whose protected op is not "keep", 11: status of the enforced keys, 12: leaves that do not
    // keep, 13: new number of leaves
    int* sc = ctx->devScalars + 10;
    CS_HIP(ctx, hipMemsetAsync(sc, 0, 4 * sizeof(int), ctx->stream));
    int rc = CSTONE_OK;
    {
        StageTimer timer(ctx, CSTONE_STAGE_REBALANCE);
        CSTONE_KEY_SWITCH(key_bits,
                          hipLaunchKernelGGL(essentialOpsKernel<K>, gridFor(size_t(num_nodes), 256), 256, 0, ctx->stream,
                                             (const K*)prefixes, child_offsets, parents, counts, macs, K(focus_start),
                                             K(focus_end), bucket_size, node_ops_all, num_nodes));
        if (num_forced_keys)
            CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(enforceKeysKernel<K>, gridFor(size_t(num_forced_keys), 64), 64,
                                                           0, ctx->stream, (const K*)forced_keys, num_forced_keys,
                                                           (const K*)prefixes, child_offsets, parents, node_ops_all,
                                                           sc + 1));
        CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(protectAncestorsKernel<K>, gridFor(size_t(num_nodes), 256), 256, 0,
                                                       ctx->stream, (const K*)prefixes, parents, node_ops_all, num_nodes,
                                                       sc));
        hipLaunchKernelGGL(leafOpsCountKernel, gridFor(size_t(num_leaves) + 1, 256), 256, 0, ctx->stream, node_ops_all,
                           leaf_to_internal, num_leaves, reinterpret_cast<uint32_t*>(leaf_ops), sc + 2);
        rc = arenaReserve(ctx, scanArenaBytes(size_t(num_leaves) + 1));
        if (rc == CSTONE_OK)
        {
            rc = scanU32(ctx, reinterpret_cast<uint32_t*>(leaf_ops), reinterpret_cast<uint32_t*>(leaf_ops),
                         size_t(num_leaves) + 1, 0u, false, reinterpret_cast<uint32_t*>(sc + 3));
            arenaReset(ctx);
        }
    }
    CS_TRY(rc);
    CS_HIP(ctx, hipGetLastError());
    CS_TRY(copyToPinned(ctx, ctx->hostScalars + 10, sc, 4 * sizeof(int)));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int changedNodes = ctx->hostScalars[10], status = ctx->hostScalars[11], changedLeaves = ctx->hostScalars[12];
    int converged = changedNodes == 0;                 // protectAncestors (R/focus/rebalance.hpp:171-184)
    if (status == 1) converged = changedLeaves == 0;   // cancelMerge: every LEAF keeps (R/focus/octree_focus.hpp:114-117)
    if (status >= 2) converged = 0;                    // rebalance / failed
    result_host[0] = status;
    result_host[1] = converged;
    result_host[2] = changedLeaves == 0;
    result_host[3] = ctx->hostScalars[13];
    return CSTONE_OK;
}

int cstone_hip_range_count(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves,
                           const uint32_t* counts, const void* leaves_focus, const int32_t* leaves_focus_idx,
                           int num_idx, uint32_t* counts_focus)
{
    if (!ctx || badKeyBits(key_bits) || num_leaves < 0 || num_idx < 0 ||
        (num_idx && (!leaves || !counts || !leaves_focus || !leaves_focus_idx || !counts_focus)))
        return fail(ctx, CSTONE_E_ARG, "range_count: bad argument");
    if (num_idx == 0) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_NODE_COUNTS);
    CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(rangeCountKernel<K>, gridFor(size_t(num_idx), 16), 256, 0, ctx->stream,
                                                   (const K*)leaves, num_leaves + 1, counts, (const K*)leaves_focus,
                                                   leaves_focus_idx, num_idx, counts_focus));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_count_sfc_gaps(cstone_hip_ctx* ctx, int key_bits, const void* tree, int num_nodes, int32_t* node_ops)
{
    if (!ctx || badKeyBits(key_bits) || num_nodes < 0 || (num_nodes && (!tree || !node_ops)))
        return fail(ctx, CSTONE_E_ARG, "count_sfc_gaps: bad argument");
    if (num_nodes == 0) return CSTONE_OK;
    CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(countGapsKernel<K>, gridFor(size_t(num_nodes), 256), 256, 0,
                                                   ctx->stream, (const K*)tree, num_nodes, node_ops));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_fill_sfc_gaps(cstone_hip_ctx* ctx, int key_bits, const void* tree, int num_nodes,
                             const int32_t* node_ops, void* new_tree)
{
    if (!ctx || badKeyBits(key_bits) || num_nodes < 0 || !tree || !node_ops || !new_tree)
        return fail(ctx, CSTONE_E_ARG, "fill_sfc_gaps: bad argument");
    CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(fillGapsKernel<K>, gridFor(size_t(num_nodes) + 1, 256), 256, 0,
                                                   ctx->stream, (const K*)tree, num_nodes, node_ops, (K*)new_tree));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_mark_macs(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                         const int32_t* child_offsets, const void* centers, const cstone_box* box_host,
                         const void* focus_nodes, int num_focus_nodes, int limit_source, char* markings)
{
    if (!ctx || badKeyBits(key_bits) || badCurve(curve) || (real_bits != 32 && real_bits != 64) || !box_host ||
        num_focus_nodes < 0 || !focus_nodes || (num_focus_nodes && (!prefixes || !child_offsets || !centers || !markings)))
        return fail(ctx, CSTONE_E_ARG, "mark_macs: bad argument");
    if (num_focus_nodes == 0) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_HALOS);
    if (key_bits == 32)
        return real_bits == 32 ? markMacs<uint32_t, float>(ctx, curve, prefixes, child_offsets, centers, *box_host,
                                                           focus_nodes, num_focus_nodes, limit_source, markings)
                               : markMacs<uint32_t, double>(ctx, curve, prefixes, child_offsets, centers, *box_host,
                                                            focus_nodes, num_focus_nodes, limit_source, markings);
    return real_bits == 32 ? markMacs<uint64_t, float>(ctx, curve, prefixes, child_offsets, centers, *box_host,
                                                       focus_nodes, num_focus_nodes, limit_source, markings)
                           : markMacs<uint64_t, double>(ctx, curve, prefixes, child_offsets, centers, *box_host,
                                                        focus_nodes, num_focus_nodes, limit_source, markings);
}

int cstone_hip_find_peers_mac(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                              const int32_t* child_offsets, const int32_t* level_range, const uint64_t* assignment_host,
                              int num_ranks, int my_rank, const cstone_box* box_host, float inv_theta_eff,
                              int32_t* peer_flags_host)
{
    if (!ctx || badKeyBits(key_bits) || badCurve(curve) || (real_bits != 32 && real_bits != 64) || !box_host ||
        !prefixes || !child_offsets || !level_range || !assignment_host || !peer_flags_host || num_ranks < 1 ||
        my_rank < 0 || my_rank >= num_ranks)
        return fail(ctx, CSTONE_E_ARG, "find_peers_mac: bad argument");
    StageTimer timer(ctx, CSTONE_STAGE_HALOS);
    if (key_bits == 32)
        return real_bits == 32 ? findPeers<uint32_t, float>(ctx, curve, prefixes, child_offsets, level_range,
                                                            assignment_host, num_ranks, my_rank, *box_host,
                                                            inv_theta_eff, peer_flags_host)
                               : findPeers<uint32_t, double>(ctx, curve, prefixes, child_offsets, level_range,
                                                             assignment_host, num_ranks, my_rank, *box_host,
                                                             inv_theta_eff, peer_flags_host);
    return real_bits == 32 ? findPeers<uint64_t, float>(ctx, curve, prefixes, child_offsets, level_range,
                                                        assignment_host, num_ranks, my_rank, *box_host, inv_theta_eff,
                                                        peer_flags_host)
                           : findPeers<uint64_t, double>(ctx, curve, prefixes, child_offsets, level_range,
                                                         assignment_host, num_ranks, my_rank, *box_host, inv_theta_eff,
                                                         peer_flags_host);
}

static int macSpheresEntry(cstone_hip_ctx* ctx, const char* name, int mode, int curve, int key_bits, int real_bits,
                           const void* prefixes, int num_nodes, void* spheres, float inv_theta,
                           const cstone_box* box_host)
{
    if (!ctx || badKeyBits(key_bits) || badCurve(curve) || (real_bits != 32 && real_bits != 64) || !box_host ||
        num_nodes < 0 || (num_nodes && (!prefixes || !spheres)))
        return fail(ctx, CSTONE_E_ARG, "%s: bad argument", name);
    if (num_nodes == 0) return CSTONE_OK;
#define CSTONE_SPHERES(K, T)                                                                                           \
    (mode == 0 ? macSpheres<K, T, 0>(ctx, curve, prefixes, num_nodes, spheres, inv_theta, *box_host)                   \
               : macSpheres<K, T, 1>(ctx, curve, prefixes, num_nodes, spheres, inv_theta, *box_host))
    if (key_bits == 32) return real_bits == 32 ? CSTONE_SPHERES(uint32_t, float) : CSTONE_SPHERES(uint32_t, double);
    return real_bits == 32 ? CSTONE_SPHERES(uint64_t, float) : CSTONE_SPHERES(uint64_t, double);
#undef CSTONE_SPHERES
}

int cstone_hip_geo_mac_spheres(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                               int num_nodes, void* spheres, float inv_theta, const cstone_box* box_host)
{
    return macSpheresEntry(ctx, "geo_mac_spheres", 0, curve, key_bits, real_bits, prefixes, num_nodes, spheres, inv_theta,
                           box_host);
}

int cstone_hip_set_mac(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes, int num_nodes,
                       void* spheres, float inv_theta, const cstone_box* box_host)
{
    return macSpheresEntry(ctx, "set_mac", 1, curve, key_bits, real_bits, prefixes, num_nodes, spheres, inv_theta,
                           box_host);
}

int cstone_hip_add_macs(cstone_hip_ctx* ctx, const char* macs, const int32_t* leaf_to_internal, int num_leaves,
                        int32_t* halo_flags)
{
    if (!ctx || num_leaves < 0 || (num_leaves && (!macs || !leaf_to_internal || !halo_flags)))
        return fail(ctx, CSTONE_E_ARG, "add_macs: bad argument");
    if (num_leaves == 0) return CSTONE_OK;
    hipLaunchKernelGGL(addMacsKernel, gridFor(size_t(num_leaves), 256), 256, 0, ctx->stream, macs, leaf_to_internal,
                       num_leaves, halo_flags);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_move_centers(cstone_hip_ctx* ctx, int real_bits, const void* src, int num_nodes, void* dst)
{
    if (!ctx || (real_bits != 32 && real_bits != 64) || num_nodes < 0 || (num_nodes && (!src || !dst)))
        return fail(ctx, CSTONE_E_ARG, "move_centers: bad argument");
    if (num_nodes == 0) return CSTONE_OK;
    unsigned grid = gridFor(size_t(num_nodes), 256);
    if (real_bits == 32)
        hipLaunchKernelGGL(moveCentersKernel<float>, grid, 256, 0, ctx->stream, (const float*)src, num_nodes, (float*)dst);
    else
        hipLaunchKernelGGL(moveCentersKernel<double>, grid, 256, 0, ctx->stream, (const double*)src, num_nodes,
                           (double*)dst);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_leaf_source_centers(cstone_hip_ctx* ctx, int coord_bits, int mass_bits, int center_bits, const void* x,
                                   const void* y, const void* z, const void* m, const int32_t* leaf_to_internal,
                                   int num_leaves, const uint32_t* layout, void* centers)
{
    if (!ctx || num_leaves < 0 || (num_leaves && (!x || !y || !z || !m || !leaf_to_internal || !layout || !centers)))
        return fail(ctx, CSTONE_E_ARG, "leaf_source_centers: bad argument");
    if (num_leaves == 0) return CSTONE_OK;
    unsigned grid = gridFor(size_t(num_leaves), 256);
    // the reference's instantiations, R/focus/source_center_gpu.cu:74-76
    if (coord_bits == 64 && mass_bits == 64 && center_bits == 64)
        hipLaunchKernelGGL((leafCentersKernel<double, double, double>), grid, 256, 0, ctx->stream, (const double*)x,
                           (const double*)y, (const double*)z, (const double*)m, leaf_to_internal, num_leaves, layout,
                           (double*)centers);
    else if (coord_bits == 64 && mass_bits == 32 && center_bits == 64)
        hipLaunchKernelGGL((leafCentersKernel<double, float, double>), grid, 256, 0, ctx->stream, (const double*)x,
                           (const double*)y, (const double*)z, (const float*)m, leaf_to_internal, num_leaves, layout,
                           (double*)centers);
    else if (coord_bits == 32 && mass_bits == 32 && center_bits == 32)
        hipLaunchKernelGGL((leafCentersKernel<float, float, float>), grid, 256, 0, ctx->stream, (const float*)x,
                           (const float*)y, (const float*)z, (const float*)m, leaf_to_internal, num_leaves, layout,
                           (float*)centers);
    else
        return fail(ctx, CSTONE_E_ARG, "leaf_source_centers: unsupported type combination %d/%d/%d", coord_bits,
                    mass_bits, center_bits);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_upsweep_centers(cstone_hip_ctx* ctx, int real_bits, int num_levels, const int32_t* level_range_host,
                               const int32_t* child_offsets, void* centers)
{
    if (!ctx || (real_bits != 32 && real_bits != 64) || num_levels < 0 || !level_range_host || !child_offsets || !centers)
        return fail(ctx, CSTONE_E_ARG, "upsweep_centers: bad argument");
    for (int level = num_levels - 1; level >= 0; --level)
    {
        int first = level_range_host[level], last = level_range_host[level + 1];
        if (last <= first) continue;
        unsigned grid = gridFor(size_t(last - first), 256);
        if (real_bits == 32)
            hipLaunchKernelGGL(upsweepCentersKernel<float>, grid, 256, 0, ctx->stream, first, last, child_offsets,
                               (float*)centers);
        else
            hipLaunchKernelGGL(upsweepCentersKernel<double>, grid, 256, 0, ctx->stream, first, last, child_offsets,
                               (double*)centers);
    }
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // extern "C"
