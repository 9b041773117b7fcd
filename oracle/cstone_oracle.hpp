// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
//
// CPU restatement ("oracle") of the cornerstone-octree hot path: SFC key encode, stable
// sort-by-key, cornerstone leaf-array build, linked octree, halo discovery, neighbor search.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
// The shipped library (cornerstone-octree_amd/csrc) never includes, links or calls it.
//
// Parity status: PINNED. Every function below is checked bit-for-bit against the reference
// itself (oracle/_ref, compiled from /root/reference/include by oracle/Makefile) in
// tests/test_oracle_vs_ref.py and against the literal known-answer vectors transcribed from
// the reference's unit tests in tests/golden/reference_kats.json.
//
// Every function cites the reference file:line whose behaviour it restates.
// R = /root/reference/include/cstone
#pragma once

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

namespace orc
{

using NodeIdx  = int;      // R/tree/definitions.h:41  (TreeNodeIndex)
using LocalIdx = unsigned; // R/tree/definitions.h:43  (LocalIndex)

enum Curve : int
{
    kMorton  = 0,
    kHilbert = 1
};

// ---------------------------------------------------------------------------------------------
// key-space constants, R/tree/definitions.h:46-91
// ---------------------------------------------------------------------------------------------
template<class K>
struct KeyInfo;
template<>
struct KeyInfo<uint32_t>
{
    static constexpr unsigned levels = 10, spare = 2;
};
template<>
struct KeyInfo<uint64_t>
{
    static constexpr unsigned levels = 21, spare = 1;
};

template<class K>
constexpr unsigned maxLevel()
{
    return KeyInfo<K>::levels;
}

//! number of keys covered by a node at @p level, R/sfc/common.hpp:97-104
template<class K>
constexpr K nodeSpan(unsigned level)
{
    return K(1) << (3u * (maxLevel<K>() - level));
}

//! the key one past the end of the SFC, doubles as the "remove me" marker, R/tree/definitions.h:87-91
template<class K>
constexpr K endKey()
{
    return nodeSpan<K>(0);
}

template<class K>
inline int clz(K x)
{
    if (x == 0) return 8 * sizeof(K);
    if constexpr (sizeof(K) == 4) { return __builtin_clz(x); }
    else { return __builtin_clzll(x); }
}

template<class K>
inline int ctz(K x)
{
    if constexpr (sizeof(K) == 4) { return __builtin_ctz(x); }
    else { return __builtin_ctzll(x); }
}

//! number of leading key bits two keys share, R/sfc/common.hpp:131-135
template<class K>
inline int sharedPrefixBits(K a, K b)
{
    return clz<K>(a ^ b) - int(KeyInfo<K>::spare);
}

//! tree level of a node spanning @p span keys (span must be a power of 8), R/sfc/common.hpp:143-148
template<class K>
inline unsigned levelOfSpan(K span)
{
    return (clz<K>(span - 1) - KeyInfo<K>::spare) / 3;
}

//! ceil(log8(n)), R/sfc/common.hpp:108-115
template<class K>
inline unsigned log8ceil(K n)
{
    if (n == 0) return 0;
    unsigned lz = clz<K>(n - 1);
    return maxLevel<K>() - (lz - KeyInfo<K>::spare) / 3;
}

//! Warren-Salmon placeholder-bit prefix: a 1 followed by @p nbits leading key bits, R/sfc/common.hpp:163-171
template<class K>
inline K toPrefix(K key, int nbits)
{
    return (K(1) << nbits) | (key >> (3 * maxLevel<K>() - nbits));
}

//! number of key bits stored in a placeholder prefix, R/sfc/common.hpp:183-187
template<class K>
inline unsigned prefixBits(K prefix)
{
    return 8 * sizeof(K) - 1 - clz<K>(prefix);
}

//! strip the placeholder bit and left-align the key again, R/sfc/common.hpp:190-198
template<class K>
inline K fromPrefix(K prefix)
{
    unsigned nb = prefixBits(prefix);
    return (prefix ^ (K(1) << nb)) << (3 * maxLevel<K>() - nb);
}

//! octal digit at @p pos (1 = most significant), R/sfc/common.hpp:236-240
template<class K>
inline unsigned octDigit(K key, unsigned pos)
{
    return (key >> (3u * (maxLevel<K>() - pos))) & 7u;
}

//! R/sfc/common.hpp:270-275
inline int digitWeight(int d) { return d >= 4 ? 7 - d : -d; }

//! lowest key of the level-@p level node enclosing @p key, R/sfc/common.hpp:285-291
template<class K>
inline K nodeStartOf(K key, unsigned level)
{
    return key & ~K(nodeSpan<K>(level) - 1);
}

//! position of the last non-zero octal digit, R/sfc/common.hpp:331-338
template<class K>
inline int lastNonZeroDigit(K x)
{
    return x ? int(maxLevel<K>()) - ctz<K>(x) / 3 : int(maxLevel<K>());
}

//! smallest placeholder-prefix that starts at key a, R/sfc/common.hpp:347-354
template<class K>
inline K makePrefix(K a)
{
    if (a == 0) return 1;
    return toPrefix<K>(a, 3 * lastNonZeroDigit(a));
}

//! number (and optionally the list) of octree nodes needed to tile [a,b), R/sfc/common.hpp:386-430
template<class K>
inline int spanRange(K a, K b, K* out)
{
    int n        = 0;
    int diverge  = (clz<K>(a ^ b) + 3 - int(KeyInfo<K>::spare)) / 3;
    int aLast    = lastNonZeroDigit(a);
    int bLast    = lastNonZeroDigit(b);
    auto step    = [](int pos) { return K(1) << 3 * (maxLevel<K>() - pos); };
    for (int pos = aLast; pos > diverge; --pos)
    {
        int reps = (8 - int(octDigit(a, pos))) % 8;
        n += reps;
        for (; reps > 0; --reps)
        {
            if (out) *out++ = a;
            a += step(pos);
        }
    }
    for (int pos = diverge; pos <= bLast; ++pos)
    {
        int reps = int(octDigit(b, pos)) - int(octDigit(a, pos));
        n += reps;
        for (; reps > 0; --reps)
        {
            if (out) *out++ = a;
            a += step(pos);
        }
    }
    return n;
}

// ---------------------------------------------------------------------------------------------
// boxes
// ---------------------------------------------------------------------------------------------

//! floating-point global bounding box, R/sfc/box.hpp:112-191. boundary codes: 0 open, 1 periodic, 2 fixed
template<class T>
struct Box
{
    T lo[3], hi[3], len[3], inv[3];
    int bc[3];

    Box() = default;
    Box(T xmin, T xmax, T ymin, T ymax, T zmin, T zmax, int bx = 0, int by = 0, int bz = 0)
    {
        lo[0] = xmin, hi[0] = xmax, lo[1] = ymin, hi[1] = ymax, lo[2] = zmin, hi[2] = zmax;
        bc[0] = bx, bc[1] = by, bc[2] = bz;
        for (int d = 0; d < 3; ++d)
        {
            len[d] = hi[d] - lo[d];
            inv[d] = T(1.) / (hi[d] - lo[d]); // R/sfc/box.hpp:135
        }
    }
};

//! integer box [min,max) per axis in grid units; may reach into [-R, 2R) under PBC, R/sfc/box.hpp:272-321
struct IBox
{
    int lo[3], hi[3];
};

// ---------------------------------------------------------------------------------------------
// Morton, R/sfc/morton.hpp:52-128,165-184
// ---------------------------------------------------------------------------------------------
inline uint32_t spread3(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0xFF0000FFu;
    v = (v | (v << 8)) & 0x0F00F00Fu;
    v = (v | (v << 4)) & 0xC30C30C3u;
    v = (v | (v << 2)) & 0x49249249u;
    return v;
}

inline uint64_t spread3(uint64_t v)
{
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x001f00000000ffffull;
    v = (v | v << 16) & 0x001f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

inline uint32_t squeeze3(uint32_t v)
{
    v &= 0x09249249u;
    v = (v ^ (v >> 2)) & 0x030c30c3u;
    v = (v ^ (v >> 4)) & 0x0300f00fu;
    v = (v ^ (v >> 8)) & 0xff0000ffu;
    v = (v ^ (v >> 16)) & 0x000003ffu;
    return v;
}

inline uint32_t squeeze3(uint64_t v)
{
    v &= 0x1249249249249249ull;
    v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ull;
    v = (v ^ (v >> 4)) & 0x100f00f00f00f00full;
    v = (v ^ (v >> 8)) & 0x001f0000ff0000ffull;
    v = (v ^ (v >> 16)) & 0x001f00000000ffffull;
    v = (v ^ (v >> 32)) & 0x00000000001fffffull;
    return uint32_t(v);
}

//! x is the most significant bit of every triplet, R/sfc/morton.hpp:114-128
template<class K>
inline K mortonEncode(unsigned ix, unsigned iy, unsigned iz)
{
    return spread3(K(ix)) * 4 + spread3(K(iy)) * 2 + spread3(K(iz));
}

template<class K>
inline void mortonDecode(K key, unsigned& ix, unsigned& iy, unsigned& iz)
{
    ix = squeeze3(K(key >> 2));
    iy = squeeze3(K(key >> 1));
    iz = squeeze3(key);
}

// ---------------------------------------------------------------------------------------------
// Hilbert, R/sfc/hilbert.hpp:58-107 (encode), :146-188 (decode)
// ---------------------------------------------------------------------------------------------
template<class K>
inline K hilbertEncode(unsigned px, unsigned py, unsigned pz)
{
    static constexpr unsigned octantToDigit[8] = {0, 1, 3, 2, 7, 6, 4, 5}; // R/sfc/hilbert.hpp:49,67
    K key = 0;
    for (int level = int(maxLevel<K>()) - 1; level >= 0; --level)
    {
        unsigned xi = (px >> level) & 1u, yi = (py >> level) & 1u, zi = (pz >> level) & 1u;
        key = (key << 3) + octantToDigit[(xi << 2) | (yi << 1) | zi];

        // reflect the remaining low bits of each axis, then permute the axes
        px ^= -(xi & ((!yi) | zi));
        py ^= -((xi & (yi | zi)) | (yi & (!zi)));
        pz ^= -((xi & (!yi) & (!zi)) | (yi & (!zi)));
        if (zi)
        {
            unsigned t = px; // (x,y,z) <- (y,z,x)
            px = py, py = pz, pz = t;
        }
        else if (!yi) { std::swap(px, pz); }
    }
    return key;
}

template<class K>
inline void hilbertDecode(K key, unsigned& ox, unsigned& oy, unsigned& oz)
{
    unsigned px = 0, py = 0, pz = 0;
    for (unsigned level = 0; level < maxLevel<K>(); ++level)
    {
        unsigned digit = (key >> (3 * level)) & 7u;
        unsigned xi = digit >> 2, yi = (digit >> 1) & 1u, zi = digit & 1u;

        if (yi ^ zi)
        {
            unsigned t = px; // (x,y,z) <- (z,x,y)
            px = pz, pz = py, py = t;
        }
        else if ((!xi & !yi & !zi) || (xi & yi & zi)) { std::swap(px, pz); }

        unsigned mask = (1u << level) - 1;
        px ^= mask & (-(xi & (yi | zi)));
        py ^= mask & (-((xi & ((!yi) | (!zi))) | ((!xi) & yi & zi)));
        pz ^= mask & (-((xi & (!yi) & (!zi)) | (yi & zi)));

        px |= (xi << level);
        py |= ((xi ^ yi) << level);
        pz |= ((yi ^ zi) << level);
    }
    ox = px, oy = py, oz = pz;
}

template<class K>
inline K sfcEncode(Curve c, unsigned ix, unsigned iy, unsigned iz)
{
    return c == kMorton ? mortonEncode<K>(ix, iy, iz) : hilbertEncode<K>(ix, iy, iz);
}

template<class K>
inline void sfcDecode(Curve c, K key, unsigned& ix, unsigned& iy, unsigned& iz)
{
    if (c == kMorton) { mortonDecode<K>(key, ix, iy, iz); }
    else { hilbertDecode<K>(key, ix, iy, iz); }
}

//! integer box of the level-@p level node starting at @p key: R/sfc/morton.hpp:178-184, R/sfc/hilbert.hpp:275-290
template<class K>
inline IBox nodeIBox(Curve c, K key, unsigned level)
{
    unsigned edge = 1u << (maxLevel<K>() - level);
    unsigned ix, iy, iz;
    sfcDecode<K>(c, key, ix, iy, iz);
    unsigned m = ~(edge - 1);
    ix &= m, iy &= m, iz &= m; // (no-op for Morton keys that start a node)
    return IBox{{int(ix), int(iy), int(iz)}, {int(ix + edge), int(iy + edge), int(iz + edge)}};
}

//! coordinate -> key, R/sfc/sfc.hpp:158-194.  The order of FP operations is part of the contract:
//! floor(x*m) - lo*m evaluated in T, then truncated to int, clamped from above only.
template<class K, class T>
inline K keyOfPoint(Curve c, T x, T y, T z, const Box<T>& box)
{
    constexpr int top   = (1u << maxLevel<K>()) - 1;
    constexpr unsigned g = 1u << maxLevel<K>();
    T mx = g * box.inv[0], my = g * box.inv[1], mz = g * box.inv[2];
    // built with -ffp-contract=off: the subtraction must not fuse with the product
    int ix = std::floor(x * mx) - box.lo[0] * mx;
    int iy = std::floor(y * my) - box.lo[1] * my;
    int iz = std::floor(z * mz) - box.lo[2] * mz;
    ix = std::min(ix, top), iy = std::min(iy, top), iz = std::min(iz, top);
    return sfcEncode<K>(c, ix, iy, iz);
}

//! R/sfc/sfc.hpp:284-291: entries holding the remove marker are left untouched
template<class K, class T>
void computeKeys(Curve c, const T* x, const T* y, const T* z, K* keys, size_t n, const Box<T>& box)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
    {
        if (keys[i] != endKey<K>()) keys[i] = keyOfPoint<K, T>(c, x[i], y[i], z[i], box);
    }
}

//! x in [0,1) -> grid, truncating / ceiling, clamped, R/sfc/common.hpp:58-88
template<class K, class T>
inline unsigned toGrid(T x)
{
    constexpr unsigned nb = maxLevel<K>();
    unsigned r = x * T(1u << nb);
    return std::min(r, (1u << nb) - 1u);
}
template<class K, class T>
inline unsigned toGridCeil(T x)
{
    constexpr unsigned nb = maxLevel<K>();
    unsigned r = std::ceil(x * T(1u << nb));
    return std::min(r, (1u << nb) - 1u);
}

// ---------------------------------------------------------------------------------------------
// stable sort by key with a LocalIdx payload, R/primitives/gather.hpp:59-90
// implemented as an LSD byte-radix sort (stable by construction => same permutation as
// std::stable_sort on (key,value) tuples compared by key only)
// ---------------------------------------------------------------------------------------------
template<class K, class V>
void sortByKey(K* keys, V* vals, size_t n)
{
    if (n == 0) return;
    std::vector<K> k2(n);
    std::vector<V> v2(n);
    K* ks[2] = {keys, k2.data()};
    V* vs[2] = {vals, v2.data()};
    int cur  = 0;
    for (unsigned shift = 0; shift < 8 * sizeof(K); shift += 8)
    {
        size_t hist[257] = {0};
        const K* ki = ks[cur];
        const V* vi = vs[cur];
        for (size_t i = 0; i < n; ++i)
            hist[((ki[i] >> shift) & 0xff) + 1]++;
        if (hist[((ki[0] >> shift) & 0xff) + 1] == n) continue; // all in one bin: pass is the identity
        for (int b = 0; b < 256; ++b)
            hist[b + 1] += hist[b];
        K* ko = ks[cur ^ 1];
        V* vo = vs[cur ^ 1];
        for (size_t i = 0; i < n; ++i)
        {
            size_t p = hist[(ki[i] >> shift) & 0xff]++;
            ko[p] = ki[i], vo[p] = vi[i];
        }
        cur ^= 1;
    }
    if (cur == 1)
    {
        std::copy(k2.begin(), k2.end(), keys);
        std::copy(v2.begin(), v2.end(), vals);
    }
}

template<class E>
void gather(const LocalIdx* map, size_t n, const E* src, E* dst)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
        dst[i] = src[map[i]];
}

// ---------------------------------------------------------------------------------------------
// cornerstone leaf array, R/tree/csarray.hpp
// ---------------------------------------------------------------------------------------------

//! counts[i] = min(#keys in [tree[i], tree[i+1]), maxCount), R/tree/csarray.hpp:94-103,200-254.
//! (the "guess" variant of the reference only accelerates the search; results are identical)
template<class K>
void nodeCounts(const K* tree, unsigned* counts, NodeIdx numNodes, const K* keys, size_t n, unsigned maxCount)
{
#pragma omp parallel for schedule(static)
    for (NodeIdx i = 0; i < numNodes; ++i)
    {
        size_t a = std::lower_bound(keys, keys + n, tree[i]) - keys;
        size_t b = std::lower_bound(keys, keys + n, tree[i + 1]) - keys;
        counts[i] = unsigned(std::min(b - a, size_t(maxCount)));
    }
}

//! R/tree/csarray.hpp:270-284: index among 8 same-level siblings if all 8 are present, else -1
template<class K>
inline int siblingIndex(const K* tree, NodeIdx i, unsigned& level)
{
    K start = tree[i];
    level   = levelOfSpan<K>(tree[i + 1] - start);
    if (level == 0) return -1;
    int s = octDigit(start, level);
    bool all8 = tree[i - s + 8] == tree[i - s] + nodeSpan<K>(level - 1);
    return all8 ? s : -1;
}

//! R/tree/csarray.hpp:288-310: 0 merge, 1 keep, 8/64/512/4096 split by 1..4 levels
template<class K>
inline int nodeOp(const K* tree, NodeIdx i, const unsigned* counts, unsigned bucket)
{
    unsigned level;
    int sib = siblingIndex(tree, i, level);
    if (sib > 0)
    {
        const unsigned* g = counts + i - sib;
        size_t parent = 0;
        for (int k = 0; k < 8; ++k)
            parent += g[k];
        if (parent <= size_t(bucket)) return 0;
    }
    constexpr unsigned top = maxLevel<K>();
    unsigned c = counts[i];
    if (c > bucket * 512 && level + 3 < top) return 4096;
    if (c > bucket * 64 && level + 2 < top) return 512;
    if (c > bucket * 8 && level + 1 < top) return 64;
    if (c > bucket && level < top) return 8;
    return 1;
}

//! R/tree/csarray.hpp:329-349; ops has numNodes+1 entries (last one untouched here)
template<class K>
bool rebalanceDecision(const K* tree, const unsigned* counts, NodeIdx numNodes, unsigned bucket, NodeIdx* ops)
{
    bool converged = true;
    for (NodeIdx i = 0; i < numNodes; ++i)
    {
        ops[i] = nodeOp(tree, i, counts, bucket);
        if (ops[i] != 1) converged = false;
    }
    return converged;
}

//! R/tree/csarray.hpp:360-409: exclusive scan of the ops, then emit 0/1/8^k start keys per old node
template<class K>
void rebalanceTree(const std::vector<K>& tree, std::vector<K>& newTree, NodeIdx* ops)
{
    NodeIdx numNodes = NodeIdx(tree.size()) - 1;
    NodeIdx run = 0;
    for (NodeIdx i = 0; i <= numNodes; ++i)
    {
        NodeIdx t = (i < numNodes) ? ops[i] : 0;
        ops[i] = run;
        run += t;
    }
    newTree.resize(ops[numNodes] + 1);
    for (NodeIdx i = 0; i < numNodes; ++i)
    {
        NodeIdx cnt = ops[i + 1] - ops[i];
        if (cnt == 0) continue;
        K start = tree[i];
        unsigned level = levelOfSpan<K>(tree[i + 1] - start);
        unsigned down  = (cnt == 1) ? 0 : log8ceil<unsigned>(unsigned(cnt)) ;
        K step = nodeSpan<K>(level + down);
        for (NodeIdx s = 0; s < cnt; ++s)
            newTree[ops[i] + s] = start + K(s) * step;
    }
    newTree.back() = tree.back();
}

//! one rebalance step + recount, R/tree/csarray.hpp:430-448
template<class K>
bool updateOctree(const K* keys, size_t n, unsigned bucket, std::vector<K>& tree, std::vector<unsigned>& counts,
                  unsigned maxCount)
{
    std::vector<NodeIdx> ops(tree.size());
    bool converged = rebalanceDecision(tree.data(), counts.data(), NodeIdx(tree.size()) - 1, bucket, ops.data());
    std::vector<K> next;
    rebalanceTree(tree, next, ops.data());
    tree.swap(next);
    counts.resize(tree.size() - 1);
    nodeCounts(tree.data(), counts.data(), NodeIdx(tree.size()) - 1, keys, n, maxCount);
    return converged;
}

//! from the root until converged, R/tree/csarray.hpp:453-466
template<class K>
int computeOctree(const K* keys, size_t n, unsigned bucket, std::vector<K>& tree, std::vector<unsigned>& counts,
                  unsigned maxCount)
{
    tree   = {K(0), endKey<K>()};
    counts = {unsigned(n)}; // seeded with n regardless of maxCount, csarray.hpp:460
    int iters = 0;
    while (!updateOctree(keys, n, bucket, tree, counts, maxCount))
        ++iters;
    return iters + 1;
}

//! R/tree/csarray.hpp:508-531
template<class K>
std::vector<K> spanningTree(const K* spanKeys, size_t numKeys)
{
    std::vector<K> out;
    for (size_t i = 0; i + 1 < numKeys; ++i)
    {
        size_t at = out.size();
        int cnt   = spanRange<K>(spanKeys[i], spanKeys[i + 1], nullptr);
        out.resize(at + cnt);
        spanRange<K>(spanKeys[i], spanKeys[i + 1], out.data() + at);
    }
    out.push_back(endKey<K>());
    return out;
}

// ---------------------------------------------------------------------------------------------
// fully linked octree, R/tree/octree.hpp:73-211
// ---------------------------------------------------------------------------------------------
template<class K>
struct LinkedOctree
{
    NodeIdx numLeaves = 0, numInternal = 0, numNodes = 0;
    std::vector<K> prefixes;             // [numNodes]  placeholder-bit keys, level-major / SFC-minor
    std::vector<NodeIdx> childOffsets;   // [numNodes+1] 0 => leaf
    std::vector<NodeIdx> parents;        // [(numNodes-1)/8] one per sibling group
    std::vector<NodeIdx> levelRange;     // [maxLevel+2]
    std::vector<NodeIdx> internalToLeaf; // [numNodes]  (value - numInternal); negative for internal nodes
    std::vector<NodeIdx> leafToInternal; // [numNodes]
};

//! R/tree/octree.hpp:73-82
template<class K>
inline NodeIdx keyWeight(K key, unsigned level)
{
    NodeIdx w = 0;
    for (unsigned l = 1; l <= level + 1; ++l)
        w += digitWeight(octDigit(key, l));
    return w;
}

template<class K>
void buildLinkedOctree(const K* leaves, NodeIdx numLeaves, LinkedOctree<K>& o)
{
    o.numLeaves   = numLeaves;
    o.numInternal = (numLeaves - 1) / 7;
    o.numNodes    = o.numLeaves + o.numInternal;
    NodeIdx nI = o.numInternal, nN = o.numNodes;
    o.prefixes.assign(nN, 0);
    o.childOffsets.assign(nN + 1, 0);
    o.parents.assign(std::max(0, (nN - 1) / 8), 0);
    o.levelRange.assign(maxLevel<K>() + 2, 0);
    o.internalToLeaf.assign(nN, 0);
    o.leafToInternal.assign(nN, 0);

    // unsorted layout, R/tree/octree.hpp:96-118
    for (NodeIdx i = 0; i < numLeaves; ++i)
    {
        K key          = leaves[i];
        unsigned level = levelOfSpan<K>(leaves[i + 1] - key);
        o.prefixes[i + nI]       = toPrefix<K>(key, 3 * level);
        o.internalToLeaf[i + nI] = i + nI;
        unsigned shared = sharedPrefixBits<K>(key, leaves[i + 1]);
        if (shared % 3 == 0 && i < numLeaves - 1)
        {
            NodeIdx slot        = (i + keyWeight<K>(key, shared / 3)) / 7;
            o.prefixes[slot]       = toPrefix<K>(key, shared);
            o.internalToLeaf[slot] = slot;
        }
    }
    // sort nodes by prefix: keys are unique, so any correct sort gives the reference order (:200)
    sortByKey<K, NodeIdx>(o.prefixes.data(), o.internalToLeaf.data(), nN);
    for (NodeIdx i = 0; i < nN; ++i)
        o.leafToInternal[o.internalToLeaf[i]] = i;
    for (NodeIdx i = 0; i < nN; ++i)
        o.internalToLeaf[i] -= nI;
    // level ranges, R/tree/octree.hpp:170-178
    for (unsigned l = 0; l <= maxLevel<K>(); ++l)
    {
        K first = toPrefix<K>(K(0), 3 * l);
        o.levelRange[l] = NodeIdx(std::lower_bound(o.prefixes.begin(), o.prefixes.end(), first) - o.prefixes.begin());
    }
    o.levelRange[maxLevel<K>() + 1] = nN;
    // link, R/tree/octree.hpp:133-166
    for (NodeIdx i = 0; i < nI; ++i)
    {
        NodeIdx a   = o.leafToInternal[i];
        K prefix    = o.prefixes[a];
        unsigned nb = prefixBits(prefix);
        unsigned level = nb / 3;
        K child     = toPrefix<K>(fromPrefix(prefix), nb + 3);
        NodeIdx s = o.levelRange[level + 1], e = o.levelRange[level + 2];
        NodeIdx c = NodeIdx(std::lower_bound(o.prefixes.begin() + s, o.prefixes.begin() + e, child) - o.prefixes.begin());
        if (c != e && o.prefixes[c] == child)
        {
            o.childOffsets[a]       = c;
            o.parents[(c - 1) / 8] = a;
        }
    }
}

//! saturating bottom-up sum of leaf counts into all nodes, R/tree/octree.hpp:584-628
inline void upsweepCounts(const NodeIdx* levelRange, int numLevelsPlus2, const NodeIdx* childOffsets, unsigned* q)
{
    for (int l = numLevelsPlus2 - 2; l >= 0; --l)
        for (NodeIdx i = levelRange[l]; i < levelRange[l + 1]; ++i)
            if (NodeIdx c = childOffsets[i])
            {
                uint64_t s = 0;
                for (int k = 0; k < 8; ++k)
                    s += q[c + k];
                q[i] = unsigned(std::min<uint64_t>(0xFFFFFFFFull, s));
            }
}

// ---------------------------------------------------------------------------------------------
// integer box overlap + halo discovery, R/traversal/boxoverlap.hpp, collisions.hpp, traversal.hpp
// ---------------------------------------------------------------------------------------------
inline bool rangesOverlap(int a, int b, int c, int d) { return b > c && d > a; } // boxoverlap.hpp:42-47

//! periodic overlap of [a,b) and [c,d) on a ring of circumference R, R/traversal/boxoverlap.hpp:57-71
inline bool ringOverlap(int R, int a, int b, int c, int d)
{
    return rangesOverlap(a, b, c, d) || rangesOverlap(a + R, b + R, c, d) || rangesOverlap(a, b, c + R, d + R);
}

template<class K>
inline bool boxesOverlap(const IBox& a, const IBox& b)
{
    constexpr int R = 1 << maxLevel<K>();
    return ringOverlap(R, a.lo[0], a.hi[0], b.lo[0], b.hi[0]) && ringOverlap(R, a.lo[1], a.hi[1], b.lo[1], b.hi[1]) &&
           ringOverlap(R, a.lo[2], a.hi[2], b.lo[2], b.hi[2]);
}

//! is the integer box fully inside the key range [lo,hi)?  R/traversal/boxoverlap.hpp:95-115
template<class K>
inline bool boxInsideKeyRange(Curve c, K lo, K hi, const IBox& b)
{
    constexpr int R = 1 << maxLevel<K>();
    if (std::min({b.lo[0], b.lo[1], b.lo[2]}) < 0 || std::max({b.hi[0], b.hi[1], b.hi[2]}) > R)
    {
        return lo == 0 && hi == endKey<K>();
    }
    K kLo = sfcEncode<K>(c, b.lo[0], b.lo[1], b.lo[2]);
    K kHi = sfcEncode<K>(c, b.hi[0] - 1, b.hi[1] - 1, b.hi[2] - 1);
    unsigned level = sharedPrefixBits<K>(kLo, kHi) / 3; // smallest common node, R/sfc/common.hpp:301-308
    K start = nodeStartOf<K>(kLo, level);
    return start >= lo && start + nodeSpan<K>(level) <= hi;
}

//! dilate a node box by radius (in coordinate units), R/traversal/boxoverlap.hpp:146-182
template<class K, class T, class Tr>
inline IBox haloBox(const IBox& node, Tr radius, const Box<T>& box)
{
    constexpr int R = 1 << maxLevel<K>();
    IBox out;
    for (int d = 0; d < 3; ++d)
    {
        int delta = toGridCeil<K>(radius * box.inv[d]);
        bool pbc  = box.bc[d] == 1;
        int lo = node.lo[d] - delta, hi = node.hi[d] + delta;
        out.lo[d] = pbc ? lo : std::min(std::max(0, lo), R);
        out.hi[d] = pbc ? hi : std::min(std::max(0, hi), R);
    }
    return out;
}

//! depth-first traversal; @p descend decides, @p leafHit is called on reached leaves, R/traversal/traversal.hpp:69-110
template<class C, class A>
inline void walkTree(const NodeIdx* childOffsets, C&& descend, A&& leafHit)
{
    if (!descend(0)) return;
    if (childOffsets[0] == 0)
    {
        leafHit(0);
        return;
    }
    std::vector<NodeIdx> stack{0};
    while (!stack.empty())
    {
        NodeIdx node = stack.back();
        stack.pop_back();
        for (int oct = 0; oct < 8; ++oct)
        {
            NodeIdx child = childOffsets[node] + oct;
            if (!descend(child)) continue;
            if (childOffsets[child] == 0) { leafHit(child); }
            else { stack.push_back(child); }
        }
    }
}

//! R/traversal/collisions.hpp:79-105 (+ findCollisions :40-57)
template<class K, class T, class Tr>
void findHalos(Curve c, const K* prefixes, const NodeIdx* childOffsets, const NodeIdx* internalToLeaf, const K* leaves,
               const Tr* radii, const Box<T>& box, NodeIdx first, NodeIdx last, int* flags)
{
    K lo = leaves[first], hi = leaves[last];
#pragma omp parallel for schedule(dynamic, 64)
    for (NodeIdx i = first; i < last; ++i)
    {
        unsigned level = levelOfSpan<K>(leaves[i + 1] - leaves[i]);
        IBox target    = haloBox<K, T, Tr>(nodeIBox<K>(c, leaves[i], level), radii[i], box);
        if (boxInsideKeyRange<K>(c, lo, hi, target)) continue;

        auto descend = [&](NodeIdx n)
        {
            K start    = fromPrefix(prefixes[n]);
            unsigned l = prefixBits(prefixes[n]) / 3;
            bool inside = !(start < lo || start + nodeSpan<K>(l) > hi);
            return !inside && boxesOverlap<K>(nodeIBox<K>(c, start, l), target);
        };
        auto hit = [&](NodeIdx n) { flags[internalToLeaf[n]] = 1; };
        walkTree(childOffsets, descend, hit);
    }
}

//! halo search radius per local leaf, CPU branch of R/halos/halos.hpp:168-180:
//! radii[i] = float( max(h[layout[i]..layout[i+1])) * 2 * ext ), 0 for empty leaves / outside [first,last)
template<class Th>
void haloRadii(const Th* h, const LocalIdx* layout, NodeIdx first, NodeIdx last, NodeIdx numLeaves, float ext,
               float* radii)
{
    std::fill(radii, radii + numLeaves, 0.0f);
    for (NodeIdx i = first; i < last; ++i)
    {
        LocalIdx a = layout[i - first], b = layout[i - first + 1];
        if (b > a) { radii[i] = *std::max_element(h + a, h + b) * 2 * ext; }
    }
}

// ---------------------------------------------------------------------------------------------
// geometric node centers, R/sfc/box.hpp:335-352, R/focus/source_center.hpp:145-156
// ---------------------------------------------------------------------------------------------
template<class K, class T>
void nodeCenters(Curve c, const K* prefixes, NodeIdx numNodes, const Box<T>& box, T* centers /*[n][3]*/,
                 T* sizes /*[n][3]*/)
{
    constexpr int g = 1u << maxLevel<K>();
    constexpr T uL  = T(1.) / g;
    for (NodeIdx i = 0; i < numNodes; ++i)
    {
        K start    = fromPrefix(prefixes[i]);
        unsigned l = prefixBits(prefixes[i]) / 3;
        IBox b     = nodeIBox<K>(c, start, l);
        for (int d = 0; d < 3; ++d)
        {
            T half            = T(0.5) * uL * box.len[d];
            centers[3 * i + d] = box.lo[d] + (b.hi[d] + b.lo[d]) * half;
            sizes[3 * i + d]   = (b.hi[d] - b.lo[d]) * half;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// neighbor search, R/findneighbors.hpp:51-188
// ---------------------------------------------------------------------------------------------
template<class T>
struct NsTree
{
    const NodeIdx* childOffsets;
    const NodeIdx* internalToLeaf;
    const LocalIdx* layout;
    const T* centers; // [numNodes][3]
    const T* sizes;   // [numNodes][3]
    float ext = 1.0f;
};

template<class T, class Th>
unsigned findNeighborsOf(LocalIdx i, const T* x, const T* y, const T* z, const Th* h, const NsTree<T>& tree,
                         const Box<T>& box, unsigned ngmax, LocalIdx* out)
{
    T xi = x[i], yi = y[i], zi = z[i];
    Th hi       = h[i];
    auto radSq  = Th(4.0) * hi * hi;
    auto cellSq = radSq * tree.ext * tree.ext;
    bool anyPbc = box.bc[0] == 1 || box.bc[1] == 1 || box.bc[2] == 1;
    T P[3]      = {xi, yi, zi};
    bool inside = true; // R/traversal/boxoverlap.hpp:186-195 with size = 2h
    for (int d = 0; d < 3; ++d)
    {
        T s = T(2) * hi;
        inside = inside && (P[d] - s >= box.lo[d]) && (P[d] + s <= box.hi[d]);
    }
    bool usePbc = anyPbc && !inside;
    unsigned nn = 0;

    auto fold = [&](T dx, int d) { return (box.bc[d] == 1) ? T(dx - box.len[d] * std::rint(dx * box.inv[d])) : dx; };

    auto descend = [&](NodeIdx n)
    {
        T sq[3];
        for (int d = 0; d < 3; ++d)
        {
            T dx = tree.centers[3 * n + d] - P[d];
            if (usePbc) { dx = fold(dx, d); }
            dx = std::abs(dx) - tree.sizes[3 * n + d];
            dx += std::abs(dx);
            dx *= T(0.5);
            sq[d] = dx * dx;
        }
        return sq[0] + (sq[1] + sq[2]) < cellSq; // right fold, R/util/array.hpp:253-256
    };
    auto leafHit = [&](NodeIdx n)
    {
        NodeIdx leaf = tree.internalToLeaf[n];
        for (LocalIdx j = tree.layout[leaf]; j < tree.layout[leaf + 1]; ++j)
        {
            if (j == i) continue;
            T dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
            if (usePbc) { dx = fold(dx, 0), dy = fold(dy, 1), dz = fold(dz, 2); }
            if (dx * dx + dy * dy + dz * dz < radSq)
            {
                if (nn < ngmax) out[nn] = j;
                ++nn;
            }
        }
    };
    walkTree(tree.childOffsets, descend, leafHit);
    return nn;
}

template<class T, class Th>
void findNeighbors(const T* x, const T* y, const T* z, const Th* h, LocalIdx first, LocalIdx last, const Box<T>& box,
                   const NsTree<T>& tree, unsigned ngmax, LocalIdx* neighbors, unsigned* counts)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (LocalIdx i = first; i < last; ++i)
        counts[i - first] = findNeighborsOf(i, x, y, z, h, tree, box, ngmax, neighbors + size_t(i - first) * ngmax);
}

// ---------------------------------------------------------------------------------------------------------------------
// target particle groups (GPU-only code in the reference: restated for a wavefront of W lanes, W = 64 on AMD hardware,
// R/cuda/gpu_config.cuh:41-49)
// ---------------------------------------------------------------------------------------------------------------------

//! R/traversal/groups_gpu.cu:41-71
inline std::vector<LocalIdx> fixedGroups(LocalIdx first, LocalIdx last, unsigned groupSize)
{
    LocalIdx numGroups = (last - first + groupSize - 1) / groupSize;
    std::vector<LocalIdx> groups(numGroups + 1);
    for (LocalIdx g = 0; g < numGroups; ++g)
        groups[g] = first + g * groupSize;
    groups[numGroups] = last;
    return groups;
}

//! split bits of one run of N*W positions, R/traversal/groups_gpu.cuh:57-93: bit l of word k is set when the particle
//! behind position k*W + l is farther than sqrt(distCritSq) away; the last position of the run compares with itself
template<class T>
inline std::vector<uint64_t> findSplits(const std::vector<std::array<T, 3>>& pos, T distCritSq, int W = 64)
{
    const int N = int(pos.size()) / W;
    std::vector<uint64_t> splits(N, 0);
    for (int k = 0; k < N; ++k)
        for (int l = 0; l < W; ++l)
        {
            const auto& a = pos[k * W + l];
            // shuffle down by one inside word k; lane W-1 takes lane 0 of word k+1, in the last word it keeps its own
            const auto& b = (l + 1 < W) ? pos[k * W + l + 1] : (k + 1 < N ? pos[(k + 1) * W] : a);
            T dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
            T distSq = dx * dx + (dy * dy + dz * dz); // right fold, R/util/array.hpp:253-256
            if (distSq > distCritSq) splits[k] |= uint64_t(1) << l;
        }
    return splits;
}

//! lengths of the zero runs between set bits, R/traversal/groups_gpu.cuh:107-130; popcount(all words) + 1 entries
inline std::vector<LocalIdx> makeSplits(const std::vector<uint64_t>& split, int W = 64)
{
    std::vector<LocalIdx> lengths;
    int carry = 0;
    for (uint64_t mask : split)
    {
        int bitsRemaining = W;
        while (mask)
        {
            int length = ctz(mask) + 1;
            bitsRemaining -= length;
            lengths.push_back(LocalIdx(length + carry));
            carry = 0;
            mask  = length < 64 ? mask >> length : 0;
        }
        carry += bitsRemaining;
    }
    lengths.push_back(LocalIdx(carry));
    return lengths;
}

//! computeGroupSplits, R/traversal/groups_gpu.cu:74-151 with the kernel of R/traversal/groups_gpu.cuh:165-232
template<class K, class T>
std::vector<LocalIdx> groupSplits(LocalIdx first, LocalIdx last, const T* x, const T* y, const T* z, const K* leaves,
                                  NodeIdx numLeaves, const LocalIdx* layout, const Box<T>& box, unsigned groupSize,
                                  float tolFactor, int W = 64)
{
    const int nwt            = int(groupSize) / W;
    const LocalIdx numFixed  = (last - first + groupSize - 1) / groupSize;
    std::vector<LocalIdx> newGroupSizes;
    for (LocalIdx w = 0; w < numFixed; ++w)
    {
        std::vector<LocalIdx> body(groupSize);
        for (unsigned p = 0; p < groupSize; ++p)
            body[p] = std::min(first + w * groupSize + p, last - 1);
        // :194-201: the volume loop runs nwt times over leafIdx[0], i.e. over the leaves of the first W bodies
        T nodeVolume = 1;
        for (int l = 0; l < W; ++l)
        {
            NodeIdx leaf = NodeIdx(std::upper_bound(layout, layout + numLeaves, body[l]) - layout) - 1;
            unsigned level = levelOfSpan<K>(leaves[leaf + 1] - leaves[leaf]);
            // centerAndSize in the unit box (R/sfc/box.hpp:334-351): half edge = (2^(maxLevel-level)) * 0.5 / 2^maxLevel
            T halfUnit = T(0.5) * (T(1.) / T(1u << maxLevel<K>()));
            T s        = T(int(1u << (maxLevel<K>() - level))) * halfUnit;
            T vol      = 8 * s * s * s;
            nodeVolume = std::min(vol, nodeVolume);
        }
        T distCrit = std::cbrt(nodeVolume) * tolFactor;
        std::vector<std::array<T, 3>> pos(groupSize);
        for (unsigned p = 0; p < groupSize; ++p)
            pos[p] = {x[body[p]] * box.inv[0], y[body[p]] * box.inv[1], z[body[p]] * box.inv[2]};
        auto lengths = makeSplits(findSplits(pos, T(distCrit * distCrit), W), W);
        newGroupSizes.insert(newGroupSizes.end(), lengths.begin(), lengths.end());
    }
    // exclusive scan of the lengths from `first`; the last entry is overwritten with `last` (:117-121)
    std::vector<LocalIdx> groups(newGroupSizes.size() + 1);
    LocalIdx run = first;
    for (size_t g = 0; g < newGroupSizes.size(); ++g)
    {
        groups[g] = run;
        run += newGroupSizes[g];
    }
    groups.back() = last;
    (void)nwt;
    return groups;
}

// ---------------------------------------------------------------------------------------------------------------------
// focus tree (locally essential tree), R/focus/rebalance.hpp, R/traversal/macs.hpp, R/focus/source_center.hpp
// ---------------------------------------------------------------------------------------------------------------------

//! mergeCountAndMacOp for every node, R/focus/rebalance.hpp:50-79,152-171: 0 merge, 1 keep, 8 split
template<class K>
void essentialOps(const K* prefixes, const NodeIdx* childOffsets, const NodeIdx* parents, const unsigned* counts,
                  const char* macs, K focusStart, K focusEnd, unsigned bucket, NodeIdx* ops, NodeIdx numNodes)
{
    for (NodeIdx i = 0; i < numNodes; ++i)
    {
        unsigned level = prefixBits(prefixes[i]) / 3;
        ops[i]         = 1;
        if (i > 0)
        {
            NodeIdx parent = parents[(i - 1) / 8];
            K groupStart   = fromPrefix(prefixes[parent]);
            K groupEnd     = groupStart + 8 * nodeSpan<K>(level);
            bool fringe    = groupEnd > focusStart && focusEnd > groupStart; // overlapTwoRanges, boxoverlap.hpp:42-47
            if (counts[parent] <= bucket || (macs[parent] == 0 && !fringe))
            {
                ops[i] = 0;
                continue;
            }
        }
        K start      = fromPrefix(prefixes[i]);
        bool inFocus = start >= focusStart && start < focusEnd;
        if (childOffsets[i] == 0 && level < maxLevel<K>() && counts[i] > bucket && (macs[i] || inFocus)) ops[i] = 8;
    }
}

//! macRefineOp per leaf outside the focus, R/focus/rebalance.hpp:81-88, R/focus/rebalance_gpu.cu:88-101
template<class K>
void macRefineOps(const K* prefixes, const char* macs, const NodeIdx* leafToInternal, NodeIdx numLeaves,
                  NodeIdx focusFirst, NodeIdx focusLast, NodeIdx* ops)
{
    for (NodeIdx i = 0; i < numLeaves; ++i)
    {
        ops[i] = 1;
        if (i < focusFirst || i >= focusLast)
        {
            NodeIdx n = leafToInternal[i];
            if (prefixBits(prefixes[n]) / 3 < maxLevel<K>() && macs[n]) ops[i] = 8;
        }
    }
}

//! protectAncestors, R/focus/rebalance.hpp:113-184 (sequential here: the result does not depend on the order);
//! returns true when every op is 1
template<class K>
bool protectAncestors(const K* prefixes, const NodeIdx* parents, NodeIdx* ops, NodeIdx numNodes)
{
    int changes = 0;
    for (NodeIdx i = 0; i < numNodes; ++i)
    {
        NodeIdx a = i;
        while (ops[a] == 0 && a != 0)
            a = parents[(a - 1) / 8];
        int op = (a == i || fromPrefix(prefixes[i]) == fromPrefix(prefixes[a])) ? ops[a] : 0;
        if (op != 1) ++changes;
        ops[i] = op;
    }
    return changes == 0;
}

//! smallest node that contains the node with placeholder prefix @p want, R/tree/octree.hpp:245-262
template<class K>
inline NodeIdx containingNode(K want, const K* prefixes, const NodeIdx* childOffsets)
{
    int level = prefixBits(want) / 3;
    K key     = fromPrefix(want);
    NodeIdx n = 0;
    for (int l = 1; l <= level; ++l)
    {
        if (childOffsets[n] == 0 || prefixes[n] == want) break;
        n = childOffsets[n] + NodeIdx(octDigit(key, unsigned(l)));
    }
    return n;
}

//! enforceKeys, R/focus/rebalance.hpp:199-267; status 0 converged, 1 cancelMerge, 2 rebalance, 3 failed (:186-196)
template<class K>
int enforceKeys(const K* forcedKeys, NodeIdx numKeys, const K* prefixes, const NodeIdx* childOffsets,
                const NodeIdx* parents, NodeIdx* ops)
{
    int status = 0;
    for (NodeIdx q = 0; q < numKeys; ++q)
    {
        K key = forcedKeys[q];
        if (key == 0 || key == endKey<K>()) continue;
        K want        = makePrefix(key);
        NodeIdx node  = containingNode(want, prefixes, childOffsets);
        int haveLevel = prefixBits(prefixes[node]) / 3;
        bool trySplit = prefixes[node] != want && haveLevel < int(maxLevel<K>());
        int st        = 0;
        if ((ops[node] == 0 || trySplit) && node > 0)
        {
            st        = 1;
            NodeIdx p = node;
            do
            {
                p = parents[(p - 1) / 8];
                for (NodeIdx c = childOffsets[p]; c < childOffsets[p] + 8; ++c)
                    if (ops[c] == 0) ops[c] = 1;
            } while (p != 0);
        }
        if (trySplit)
        {
            int levelDiff = lastNonZeroDigit(key) - haveLevel;
            st            = levelDiff > 1 ? 3 : 2;
            levelDiff     = std::min(levelDiff, 1);
            ops[node]     = std::max(ops[node], 1 << (3 * levelDiff));
        }
        status = std::max(status, st);
    }
    return status;
}

//! rangeCount, R/focus/rebalance.hpp:279-301 (findNodeBelow / findNodeAbove over ALL numLeaves + 1 keys, csarray.hpp:78-90)
template<class K>
void rangeCount(const K* leaves, NodeIdx numLeaves, const unsigned* counts, const K* leavesFocus, const NodeIdx* focusIdx,
                NodeIdx numIdx, unsigned* countsFocus)
{
    for (NodeIdx q = 0; q < numIdx; ++q)
    {
        NodeIdx leaf  = focusIdx[q];
        NodeIdx first = NodeIdx(std::upper_bound(leaves, leaves + numLeaves + 1, leavesFocus[leaf]) - leaves) - 1;
        NodeIdx last  = NodeIdx(std::lower_bound(leaves, leaves + numLeaves + 1, leavesFocus[leaf + 1]) - leaves);
        uint64_t sum  = 0;
        for (NodeIdx i = first; i < last; ++i)
            sum += counts[i];
        countsFocus[leaf] = unsigned(std::min<uint64_t>(0xFFFFFFFFull, sum));
    }
}

//! centre and half-size of an integer box, R/sfc/box.hpp:335-352
template<class K, class T>
inline void centerAndSize(const IBox& b, const Box<T>& box, T (&center)[3], T (&size)[3])
{
    constexpr int g = 1u << maxLevel<K>();
    constexpr T uL  = T(1.) / g;
    for (int d = 0; d < 3; ++d)
    {
        T half    = T(0.5) * uL * box.len[d];
        center[d] = box.lo[d] + (b.hi[d] + b.lo[d]) * half;
        size[d]   = (b.hi[d] - b.lo[d]) * half;
    }
}

//! computeMinMacR2 (mode 0), R/traversal/macs.hpp:44-58, and setMac / computeVecMacR2 (mode 1), :69-85 with
//! R/focus/source_center.hpp:118-131; spheres = Vec4<T>[numNodes]
template<class K, class T>
void macSpheres(Curve c, int mode, const K* prefixes, NodeIdx numNodes, T* spheres, float invTheta, const Box<T>& box)
{
    for (NodeIdx i = 0; i < numNodes; ++i)
    {
        unsigned level = prefixBits(prefixes[i]) / 3;
        IBox b         = nodeIBox<K>(c, fromPrefix(prefixes[i]), level);
        T ctr[3], sz[3];
        centerAndSize<K, T>(b, box, ctr, sz);
        T l = T(2) * std::max(sz[0], std::max(sz[1], sz[2]));
        if (mode == 0)
        {
            T mac              = l * invTheta;
            spheres[4 * i]     = ctr[0], spheres[4 * i + 1] = ctr[1], spheres[4 * i + 2] = ctr[2];
            spheres[4 * i + 3] = mac * mac;
        }
        else
        {
            T m  = spheres[4 * i + 3];
            T dx = spheres[4 * i] - ctr[0], dy = spheres[4 * i + 1] - ctr[1], dz = spheres[4 * i + 2] - ctr[2];
            T s  = std::sqrt(dx * dx + (dy * dy + dz * dz)); // right fold of util::dot, R/util/array.hpp:252-256
            T mac = l * invTheta + s;
            spheres[4 * i + 3] = (m != T(0)) ? mac * mac : T(0);
        }
    }
}

//! markMacs, R/traversal/macs.hpp:199-270 with markMacPerBox :138-170 and evaluateMacPbc :121-134
template<class K, class T>
void markMacs(Curve c, const K* prefixes, const NodeIdx* childOffsets, const T* centers /*Vec4*/, const Box<T>& box,
              const K* focusNodes, NodeIdx numFocusNodes, bool limitSource, char* markings)
{
    K focusStart = focusNodes[0], focusEnd = focusNodes[numFocusNodes];
    for (NodeIdx i = 0; i < numFocusNodes; ++i)
    {
        unsigned level = levelOfSpan<K>(focusNodes[i + 1] - focusNodes[i]);
        IBox target    = nodeIBox<K>(c, focusNodes[i], level);
        IBox ext       = target;
        for (int d = 0; d < 3; ++d)
            ext.lo[d] -= 1, ext.hi[d] += 1;
        if (boxInsideKeyRange<K>(c, focusStart, focusEnd, ext)) continue;
        T tc[3], ts[3];
        centerAndSize<K, T>(target, box, tc, ts);
        unsigned maxSource = limitSource ? unsigned(std::max(int(level) - 1, 0)) : maxLevel<K>();

        auto violates = [&](NodeIdx n)
        {
            unsigned l = prefixBits(prefixes[n]) / 3;
            K start    = fromPrefix(prefixes[n]);
            K end      = start + nodeSpan<K>(l);
            if (!(start < focusStart || end > focusEnd)) return false;
            const T* sc = centers + 4 * size_t(n);
            T dX[3];
            for (int d = 0; d < 3; ++d)
            {
                T dx = tc[d] - sc[d];
                dx -= T(box.bc[d] == 1) * box.len[d] * std::rint(dx * box.inv[d]); // applyPbc, R/sfc/box.hpp:195-206
                dx = std::abs(dx);
                dx -= ts[d];
                dx += std::abs(dx);
                dx *= T(0.5);
                dX[d] = dx;
            }
            T r2   = dX[0] * dX[0] + (dX[1] * dX[1] + dX[2] * dX[2]);
            bool v = r2 < std::abs(sc[3]) && l <= maxSource;
            if (v && !markings[n]) markings[n] = 1;
            return v;
        };
        walkTree(childOffsets, violates, [](NodeIdx) {});
    }
}

//! massCenter per leaf, R/focus/source_center.hpp:44-77,96-118
template<class Tc, class Tm, class Tf>
void leafSourceCenters(const Tc* x, const Tc* y, const Tc* z, const Tm* m, const NodeIdx* leafToInternal,
                       NodeIdx numLeaves, const LocalIdx* layout, Tf* centers /*Vec4*/)
{
    for (NodeIdx leaf = 0; leaf < numLeaves; ++leaf)
    {
        Tf c[4] = {0, 0, 0, 0};
        for (LocalIdx i = layout[leaf]; i < layout[leaf + 1]; ++i)
        {
            Tf w = std::abs(Tf(m[i]));
            c[0] += w * Tf(x[i]), c[1] += w * Tf(y[i]), c[2] += w * Tf(z[i]), c[3] += w;
        }
        Tf inv    = (c[3] != Tf(0.0)) ? Tf(1.0) / c[3] : Tf(1.0);
        NodeIdx n = leafToInternal[leaf];
        centers[4 * n] = c[0] * inv, centers[4 * n + 1] = c[1] * inv, centers[4 * n + 2] = c[2] * inv, centers[4 * n + 3] = c[3];
    }
}

//! CombineSourceCenter bottom-up, R/focus/source_center.hpp:79-95 with the level loop of source_center_gpu.cu:95-113
template<class T>
void upsweepCenters(int numLevels, const NodeIdx* levelRange, const NodeIdx* childOffsets, T* centers /*Vec4*/)
{
    for (int level = numLevels - 1; level >= 0; --level)
        for (NodeIdx cell = levelRange[level]; cell < levelRange[level + 1]; ++cell)
        {
            NodeIdx child = childOffsets[cell];
            if (!child) continue;
            T c[4] = {0, 0, 0, 0};
            for (int k = 0; k < 8; ++k)
            {
                const T* s = centers + 4 * size_t(child + k);
                T w        = std::abs(s[3]);
                c[0] += w * s[0], c[1] += w * s[1], c[2] += w * s[2], c[3] += w;
            }
            T inv = (c[3] != T(0.0)) ? T(1.0) / c[3] : T(1.0);
            centers[4 * size_t(cell)] = c[0] * inv, centers[4 * size_t(cell) + 1] = c[1] * inv;
            centers[4 * size_t(cell) + 2] = c[2] * inv, centers[4 * size_t(cell) + 3] = c[3];
        }
}

//! segmentMax, R/primitives/primitives_gpu.cu:241-259 (every segment starts at 0: an empty one gives 0)
template<class Tin, class Tout, class I>
void segmentMax(const Tin* in, const I* seg, size_t numSegments, Tout* out)
{
    for (size_t s = 0; s < numSegments; ++s)
    {
        Tin m = 0;
        for (I i = seg[s]; i < seg[s + 1]; ++i)
            m = std::max(m, in[i]);
        out[s] = Tout(m);
    }
}

//! gatherRanges, R/halos/gather_halos_gpu.cu:26-40
template<class E, class I>
void gatherRanges(const I* scan, const I* offsets, int numRanges, const E* src, E* buffer, size_t bufferSize)
{
    for (size_t i = 0; i < bufferSize; ++i)
    {
        int r     = int(std::upper_bound(scan, scan + numRanges, I(i)) - scan) - 1;
        buffer[i] = src[offsets[r] + I(i) - scan[r]];
    }
}

//! binary radix tree over the keys of a leaf array (Karras 2012), R/tree/btree.hpp:104-264: child[2 i], child[2 i + 1]
//! and prefix[i] of internal node i; a leaf child is stored as index - 2^31 (:48-66)
template<class K>
void binaryTree(const K* codes, NodeIdx numCodes, NodeIdx* child, K* prefix)
{
    auto cpl = [&](NodeIdx a, NodeIdx b) { return sharedPrefixBits<K>(codes[a], codes[b]); };
    for (NodeIdx first = 0; first < numCodes - 1; ++first)
    {
        int d = 1, minPrefix = -1;
        if (first > 0)
        {
            d         = cpl(first, first + 1) > cpl(first, first - 1) ? 1 : -1;
            minPrefix = cpl(first, first - d);
        }
        NodeIdx range = 2, second = first + range * d;
        while (0 <= second && second < numCodes && cpl(first, second) > minPrefix)
        {
            range *= 2;
            second = first + range * d;
        }
        second = first;
        do
        {
            range        = (range + 1) / 2;
            NodeIdx cand = second + range * d;
            if (0 <= cand && cand < numCodes && cpl(first, cand) > minPrefix) second = cand;
        } while (range > 1);
        int nbits     = cpl(first, second);
        K low         = (K(1) << (3 * maxLevel<K>() - nbits)) - 1;
        prefix[first] = toPrefix<K>(codes[first] & ~low, nbits);
        NodeIdx lo = std::min(first, second), hi = std::max(first, second), split;
        if (codes[lo] == codes[hi]) { split = (lo + hi) >> 1; }
        else
        {
            int common   = cpl(lo, hi);
            split        = lo;
            NodeIdx step = hi - lo;
            do
            {
                step         = (step + 1) / 2;
                NodeIdx cand = split + step;
                if (cand < hi && cpl(lo, cand) > common) split = cand;
            } while (step > 1);
        }
        const NodeIdx leafOffset = std::numeric_limits<NodeIdx>::min();
        child[2 * first]     = lo == split ? split + leafOffset : split;
        child[2 * first + 1] = hi == split + 1 ? split + 1 + leafOffset : split + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// peers of a rank and node lookup (locally essential tree on several ranks), R/traversal/peers.hpp, R/tree/octree.hpp
// ---------------------------------------------------------------------------------------------------------------------

//! node with key range [start, end): locateNode, R/tree/octree.hpp:216-241; numNodes if there is none
template<class K>
inline NodeIdx locateNode(K start, K end, const K* prefixes, const NodeIdx* levelRange)
{
    NodeIdx numNodes = levelRange[maxLevel<K>() + 1];
    int bits         = clz<K>(K(end - start - 1)) - int(KeyInfo<K>::spare);
    K want           = toPrefix<K>(start, bits);
    unsigned level   = unsigned(bits) / 3;
    const K* it = std::lower_bound(prefixes + levelRange[level], prefixes + levelRange[level + 1], want);
    if (it != prefixes + numNodes && *it == want) return NodeIdx(it - prefixes);
    return numNodes;
}

/*! findPeersMac, R/traversal/peers.hpp:63-118: ranks that own a leaf of the (replicated) global tree which fails the
 *  mutual MAC (minVecMacMutual, R/traversal/macs.hpp:171-194) paired with a leaf of @p myRank's range; dual traversal
 *  R/traversal/traversal.hpp:135-188 from every node that spans the range.  peerFlags[numRanks] gets 0 / 1. */
template<class K, class T>
void findPeersMac(Curve c, const K* prefixes, const NodeIdx* childOffsets, const NodeIdx* levelRange, const K* assignment,
                  int numRanks, int myRank, const Box<T>& box, float invThetaEff, int* peerFlags)
{
    std::fill(peerFlags, peerFlags + numRanks, 0);
    const K domainStart = assignment[myRank], domainEnd = assignment[myRank + 1];
    auto isLeaf = [&](NodeIdx n) { return childOffsets[n] == 0; };
    auto level  = [&](NodeIdx n) { return prefixBits(prefixes[n]) / 3; };
    auto start  = [&](NodeIdx n) { return fromPrefix(prefixes[n]); };
    auto end    = [&](NodeIdx n) { return K(fromPrefix(prefixes[n]) + nodeSpan<K>(level(n))); };
    // minDistance(X, bCenter, bSize, box), R/traversal/boxoverlap.hpp:208-218, squared
    auto dist2 = [&](const T (&X)[3], const T (&bc)[3], const T (&bs)[3])
    {
        T dX[3];
        for (int d = 0; d < 3; ++d)
        {
            T dx = bc[d] - X[d];
            dx -= T(box.bc[d] == 1) * box.len[d] * std::rint(dx * box.inv[d]);
            dx = std::abs(dx);
            dx -= bs[d];
            dx += std::abs(dx);
            dx *= T(0.5);
            dX[d] = dx;
        }
        return dX[0] * dX[0] + (dX[1] * dX[1] + dX[2] * dX[2]);
    };
    auto follows = [&](NodeIdx a, NodeIdx b)
    {
        bool aFocusOverlap = domainStart < end(a) && start(a) < domainEnd;      // overlapTwoRanges
        bool bInFocus      = start(b) >= domainStart && end(b) <= domainEnd;    // containedIn
        if (!aFocusOverlap || bInFocus) return false;
        T ac[3], as[3], bc[3], bs[3];
        centerAndSize<K, T>(nodeIBox<K>(c, start(a), level(a)), box, ac, as);
        centerAndSize<K, T>(nodeIBox<K>(c, start(b), level(b)), box, bc, bs);
        T macA     = std::max(bs[0], std::max(bs[1], bs[2])) * 2 * invThetaEff;
        bool passA = dist2(bc, ac, as) > macA * macA;
        T macB     = std::max(as[0], std::max(as[1], as[2])) * 2 * invThetaEff;
        bool passB = dist2(ac, bc, bs) > macB * macB;
        return !(passA && passB);
    };
    auto mark = [&](NodeIdx b)
    {
        int rank = int(std::upper_bound(assignment, assignment + numRanks + 1, start(b)) - assignment) - 1;
        peerFlags[rank] = 1;
    };
    std::vector<K> span(size_t(spanRange<K>(domainStart, domainEnd, nullptr)) + 1);
    spanRange<K>(domainStart, domainEnd, span.data());
    span.back() = domainEnd;
    for (size_t i = 0; i + 1 < span.size(); ++i)
    {
        NodeIdx a0 = locateNode<K>(span[i], span[i + 1], prefixes, levelRange);
        if (isLeaf(a0) && isLeaf(0))
        {
            if (follows(a0, 0)) mark(0);
            continue;
        }
        std::vector<std::pair<NodeIdx, NodeIdx>> stack{{a0, 0}};
        auto interact = [&](NodeIdx a, NodeIdx b)
        {
            if (!follows(a, b)) return;
            if (isLeaf(a) && isLeaf(b)) mark(b);
            else stack.push_back({a, b});
        };
        while (!stack.empty())
        {
            auto [target, source] = stack.back();
            stack.pop_back();
            if ((level(target) < level(source) && !isLeaf(target)) || isLeaf(source))
            {
                if (!isLeaf(target))
                    for (int oct = 0; oct < 8; ++oct)
                        interact(childOffsets[target] + oct, source);
            }
            else if (!isLeaf(source))
            {
                for (int oct = 0; oct < 8; ++oct)
                    interact(target, childOffsets[source] + oct);
            }
        }
    }
}

} // namespace orc
