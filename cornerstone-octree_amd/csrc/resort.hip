// Incremental re-sort of Domain::sync: leaf table, mover binning and the leaf pass (see resort.hpp for the scheme).
// Replaces, for a sync whose particles mostly stayed in their leaves, the reference's full sort of all keys
// (sortByKeyGpu, R/primitives/primitives_gpu.cu:305-353).  gfx950: wave64, 160 KB LDS per CU.
#include <algorithm>
#include <cstdlib>

#include "device_keys.hpp"
#include "resort.hpp"
#include "scan.hpp"

namespace cship
{
namespace
{

/*! Leaf-start bitmask without atomics: bit p of mask is set when a non-empty leaf starts at position p.  The thread of
 *  the first non-empty leaf that starts inside a 64-position word builds that word from the leaves that follow (a leaf
 *  holds a few dozen particles: one or two starts per word), clears the words between the previous start and its own
 *  and stores the words' bit counts for the rank scan.  The thread of the last non-empty leaf also clears the tail. */
__global__ __launch_bounds__(256) void leafStartWordsKernel(const uint32_t* __restrict__ layout, int numLeaves,
                                                            uint32_t n, uint64_t* __restrict__ mask,
                                                            uint32_t* __restrict__ bits, uint32_t words)
{
    int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= numLeaves) return;
    const uint32_t a = layout[l];
    if (layout[l + 1] <= a) return; // empty
    const uint32_t w = a >> 6;
    // the previous non-empty leaf (empty ones in between start at a as well)
    int prev = l - 1;
    while (prev >= 0 && layout[prev] == a)
        --prev;
    const bool first = prev < 0;
    if (!first && (layout[prev] >> 6) == w) return; // not the first start inside this word
    uint64_t word = 0;
    uint32_t pos  = a;
    int m         = l;
    while (m < numLeaves && (pos >> 6) == w)
    {
        if (layout[m + 1] > pos) word |= 1ull << (pos & 63u);
        ++m;
        pos = m < numLeaves ? layout[m] : n;
        // (empty leaves repeat the position of their successor: the bit is set once, by the non-empty one)
    }
    mask[w] = word;
    bits[w] = uint32_t(__popcll(word));
    for (uint32_t v = first ? 0u : (layout[prev] >> 6) + 1; v < w; ++v)
        mask[v] = 0, bits[v] = 0;
    if (m >= numLeaves || pos >= n)
    {
        // nothing starts behind this word
        for (uint32_t v = w + 1; v < words; ++v)
            mask[v] = 0, bits[v] = 0;
    }
}

//! the j-th non-empty leaf: first key (0 for the first one: keys in front of it belong to it) and first position;
//! entries J and J + 1 close the table: keys from endKey on (the remove markers) form a leaf of their own without positions
template<class K>
__global__ __launch_bounds__(256) void fillCompactLeavesKernel(const K* __restrict__ tree,
                                                               const uint32_t* __restrict__ layout, int numLeaves,
                                                               const uint64_t* __restrict__ mask,
                                                               const uint32_t* __restrict__ rank,
                                                               const uint32_t* __restrict__ numCompact, uint32_t n,
                                                               K* __restrict__ leafLo, uint32_t* __restrict__ leafPos,
                                                               uint32_t* __restrict__ outCount,
                                                               uint32_t* __restrict__ incoming)
{
    int l = blockIdx.x * 256 + threadIdx.x;
    if (l < numLeaves + 3) outCount[l] = 0, incoming[l] = 0; // the departure / arrival counters of this sync
    if (l == 0)
    {
        uint32_t J     = *numCompact;
        leafLo[J]      = endKey<K>();
        leafLo[J + 1]  = ~K(0);
        leafPos[J]     = n;
        leafPos[J + 1] = n;
    }
    if (l >= numLeaves) return;
    uint32_t a = layout[l], b = layout[l + 1];
    if (b <= a) return;
    uint32_t j = rank[a >> 6] + uint32_t(__popcll(mask[a >> 6] & ((1ull << (a & 63u)) - 1)));
    leafLo[j]  = j == 0 ? K(0) : tree[l];
    leafPos[j] = a;
}

//! coarse[c] = last leaf j with leafLo[j] <= c << RESORT_COARSE_SHIFT, for the 2^RESORT_COARSE_BITS + 1 values of c that
//! keys have in their leading bits: the search for a mover's leaf then starts from a handful of candidates
template<class K>
__global__ __launch_bounds__(256) void coarseLeafTableKernel(const K* __restrict__ leafLo,
                                                             const uint32_t* __restrict__ numCompact,
                                                             uint32_t* __restrict__ coarse)
{
    constexpr int shift = 3 * int(maxLevel<K>()) - RESORT_COARSE_BITS;
    const uint32_t c    = blockIdx.x * 256 + threadIdx.x;
    if (c > (1u << RESORT_COARSE_BITS)) return;
    const K key      = K(c) << shift; // c = 2^bits: endKey, the markers' entry
    const uint32_t J = *numCompact;
    uint32_t lo = 0, hi = J + 1;
    while (hi - lo > 1)
    {
        uint32_t mid = (lo + hi) / 2;
        if (leafLo[mid] <= key) lo = mid;
        else hi = mid;
    }
    coarse[c] = lo;
}

//! leaf of a mover's new key, its slot among the movers arriving there
template<class K>
__global__ __launch_bounds__(256) void binMoversKernel(const K* __restrict__ moverKeys,
                                                       const uint32_t* __restrict__ moverCount, uint32_t moverCap,
                                                       const K* __restrict__ leafLo,
                                                       const uint32_t* __restrict__ numCompact,
                                                       const uint32_t* __restrict__ coarse,
                                                       uint32_t* __restrict__ incoming, uint32_t* __restrict__ dest,
                                                       uint32_t* __restrict__ slot)
{
    const uint32_t M = *moverCount;
    if (M > moverCap) return; // list incomplete: the caller falls back
    const uint32_t J = *numCompact;
    for (uint32_t m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256)
    {
        const K key = moverKeys[m];
        // last j in [0, J] with leafLo[j] <= key (leafLo[0] = 0)
        uint32_t lo = 0, hi = J + 1;
        if (coarse)
        {
            // the leaves of the key's coarse cell: from the last leaf at or before the cell's first key to the last one
            // at or before the next cell's
            constexpr int shift = 3 * int(maxLevel<K>()) - RESORT_COARSE_BITS;
            const uint32_t c    = uint32_t(key >> shift); // (a marker: 2^bits)
            lo                  = coarse[c];
            hi                  = c < (1u << RESORT_COARSE_BITS) ? coarse[c + 1] + 1 : J + 1;
        }
        while (hi - lo > 1)
        {
            uint32_t mid = (lo + hi) / 2;
            if (leafLo[mid] <= key) lo = mid;
            else hi = mid;
        }
        dest[m] = lo;
        slot[m] = atomicAdd(&incoming[lo], 1u);
    }
}

//! size of every leaf after the moves; [1] |= 1: a leaf too long for the leaf pass
__global__ __launch_bounds__(256) void newLeafSizesKernel(const uint32_t* __restrict__ leafPos,
                                                          const uint32_t* __restrict__ outCount,
                                                          const uint32_t* __restrict__ incoming,
                                                          const uint32_t* __restrict__ numCompact, uint32_t entries,
                                                          uint32_t* __restrict__ newCount, int* __restrict__ scalars)
{
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= entries) return;
    const uint32_t J = *numCompact;
    uint32_t c       = 0;
    if (j <= J)
    {
        uint32_t old = j < J ? leafPos[j + 1] - leafPos[j] : 0u;
        c            = old - outCount[j] + incoming[j];
        // (entry J, the remove markers, does not go through the leaf pass)
        // (slot numbers 0 .. RESORT_LEAF_CAP - 2 only: the digest of slot 255 behind 24 set key bits would equal the
        //  hole's ~0u)
        if (j < J && old + incoming[j] >= RESORT_LEAF_CAP) atomicOr(&scalars[1], 1);
    }
    newCount[j] = c;
}

//! [1] |= 2: the leaves of a workgroup of the leaf pass need more LDS slots than it has, |= 4: mover list overflow,
//! |= 8 (no failure): a large quiet tile; [0]: particles carrying the remove marker
__global__ __launch_bounds__(256) void checkTilesKernel(const uint32_t* __restrict__ leafPos,
                                                        const uint32_t* __restrict__ inOffset,
                                                        const uint32_t* __restrict__ layoutNew,
                                                        const uint32_t* __restrict__ numCompact,
                                                        const uint32_t* __restrict__ moverCount, uint32_t moverCap,
                                                        int leavesPerTile, int* __restrict__ scalars)
{
    const uint32_t J = *numCompact;
    uint32_t t       = blockIdx.x * 256 + threadIdx.x;
    if (t == 0)
    {
        scalars[0] = int(inOffset[J + 1] - inOffset[J]);
        if (*moverCount > moverCap) atomicOr(&scalars[1], 4);
    }
    uint32_t j0 = t * uint32_t(leavesPerTile);
    if (j0 >= J) return;
    uint32_t j1    = min(j0 + uint32_t(leavesPerTile), J);
    const uint32_t old = leafPos[j1] - leafPos[j0], arrivals = inOffset[j1] - inOffset[j0];
    if (old + arrivals > RESORT_TILE_SLOTS) atomicOr(&scalars[1], 2);
    // a tile in which nothing moved but with more slots than the quiet instantiation of the leaf pass has: the other
    // instantiation has to be launched for it even without movers
    if (arrivals == 0 && layoutNew[j1] - layoutNew[j0] == old && old > RESORT_QUIET_SLOTS) atomicOr(&scalars[1], 8);
}

template<class K>
__global__ __launch_bounds__(256) void placeMoversKernel(const K* __restrict__ moverKeys,
                                                         const uint32_t* __restrict__ moverIdx,
                                                         const uint32_t* __restrict__ dest,
                                                         const uint32_t* __restrict__ slot, uint32_t M,
                                                         const uint32_t* __restrict__ inOffset, K* __restrict__ binKeys,
                                                         uint32_t* __restrict__ binIdx)
{
    uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    uint32_t at = inOffset[dest[m]] + slot[m];
    binKeys[at] = moverKeys[m];
    binIdx[at]  = moverIdx[m];
}

/*! The leaf pass.  A workgroup takes G consecutive (non-empty) leaves of the previous sync.  Its LDS slots:
 *  [the old positions of its leaves, in place | the movers arriving in its leaves, leaf by leaf].  The encode pass has
 *  replaced the key of every particle that left its leaf by a hole (key ~0, sorts last), so the old positions are
 *  copied as they are.  Then, per tile:
 *   - nothing left or arrived (the common case of a quiet region): one lane per leaf insertion-sorts the leaf in LDS
 *     -- the particles come in the order of the previous sync, i.e. almost sorted: one LDS read per element --, and the
 *     tile is written back slot by slot;
 *   - otherwise every particle counts the elements of its leaf (old positions and arrivals) that sort in front of it,
 *     by (key, old index): LDS broadcast reads, all 256 lanes busy, cost independent of the disorder.  The count is the
 *     particle's place in the leaf; holes rank last and are dropped.
 *  Either way leaf j's new content lands at layoutNew[j]...: ascending keys, ties by old index = the stable sort. */
template<class K, int G, bool COUNTING>
__global__ __launch_bounds__(256) void leafSortKernel(const K* __restrict__ keysIn, const uint64_t* __restrict__ mask,
                                                      const uint32_t* __restrict__ rank, const K* __restrict__ leafLo,
                                                      const uint32_t* __restrict__ leafPos,
                                                      const uint32_t* __restrict__ inOffset,
                                                      const uint32_t* __restrict__ layoutNew,
                                                      const K* __restrict__ binKeys, const uint32_t* __restrict__ binIdx,
                                                      uint32_t J, bool alwaysCount, K* __restrict__ keysOut,
                                                      uint32_t* __restrict__ orderOut)
{
    constexpr int ITER = (RESORT_TILE_SLOTS + 255) / 256;
    constexpr K HOLE   = ~K(0);
    // Two instantiations share this body and the grid: COUNTING = false takes the quiet tiles (keys and old indices in
    // LDS, up to RESORT_QUIET_SLOTS of them = 39 KB: four workgroups per CU), COUNTING = true the tiles in which something
    // moved, and the few quiet ones with more slots than that (digests and the new
    // order's key bits folded to 16, 27 KB: five per CU); a workgroup whose tile is of the other kind leaves after the
    // setup.
    __shared__ __attribute__((aligned(8))) uint32_t sWordsA[COUNTING ? RESORT_TILE_SLOTS / 2 : RESORT_QUIET_SLOTS * sizeof(K) / 4];
    __shared__ uint32_t sWordsB[COUNTING ? RESORT_TILE_SLOTS : RESORT_QUIET_SLOTS / 2];
    K* const sKey        = reinterpret_cast<K*>(sWordsA);        // quiet tiles: the keys ...
    uint16_t* const sIdx = reinterpret_cast<uint16_t*>(sWordsB); // ... and the slots of the tile they came from
    __shared__ uint32_t posK[G + 1], inK[G + 1], outK[G + 1];
    // tiles that count: first key of every leaf, the low key bits a digest leaves out, first 4-slot chunk of every leaf
    __shared__ K loK[G + 1];
    __shared__ uint8_t cutK[G];
    __shared__ uint32_t chunkK[G + 1];
    static_assert(G <= 64, "the chunk prefix of a tile is one wave scan");

    const uint32_t j0 = blockIdx.x * uint32_t(G);
    const uint32_t nl = min(uint32_t(G), J - j0);
    const uint32_t t  = threadIdx.x;
    if (t <= nl)
    {
        posK[t] = leafPos[j0 + t];
        inK[t]  = inOffset[j0 + t];
        outK[t] = layoutNew[j0 + t];
        if constexpr (COUNTING) loK[t] = leafLo[j0 + t];
    }
    __syncthreads();
    if (COUNTING && t < 64)
    {
        // chunks of four old slots, leaf by leaf: a lane of the counting path takes one chunk at a time
        const uint32_t mine = t < nl ? (posK[t + 1] - posK[t] + 3) / 4 : 0u;
        const uint32_t incl = waveInclusiveScan(mine, t);
        if (t < nl) chunkK[t] = incl - mine;
        if (t == 63) chunkK[nl] = incl;
        if (t < nl)
        {
            // 24 leading bits of key - loK[t] tell the particles of a leaf apart in all but a few cases
            const K span   = loK[t + 1] - loK[t] - 1;
            const int bits = span ? int(8 * sizeof(K)) - clzKey(span) : 0;
            cutK[t]        = uint8_t(bits > 24 ? bits - 24 : 0);
        }
    }
    const uint32_t p0 = posK[0], p1 = posK[nl], in0 = inK[0], in1 = inK[nl];
    const uint32_t nOldAll = p1 - p0, slots = nOldAll + (in1 - in0);
    // guarded by checkTilesKernel: a launch only happens when every workgroup fits
    if (slots > RESORT_TILE_SLOTS) return;
    // quiet tile: no arrivals, and every leaf keeps its size (without arrivals: nobody left)
    bool changed = in1 != in0 || alwaysCount;
    if (t < nl) changed = changed || (outK[t + 1] - outK[t]) != (posK[t + 1] - posK[t]);
    // (a quiet tile beyond the slots of the quiet instantiation goes the other way: it can hold RESORT_TILE_SLOTS)
    const bool quiet = !__syncthreads_or(changed) && nOldAll <= RESORT_QUIET_SLOTS;
    if (quiet == COUNTING) return; // the other instantiation's tile

    // Tiles that count keep no keys in LDS, only 32-bit digests: 24 leading bits of (key - first key of the leaf), then
    // the slot in the leaf (old slots first, then arrivals; at most 256).  The digests of a leaf are distinct, and
    // ordered like (key, old index) as long as the leading key bits of its elements differ -- checked afterwards.
    uint32_t* const sDig = sWordsB; // digests by slot
    uint16_t* const sNew = reinterpret_cast<uint16_t*>(sWordsA); // leading key bits in the NEW order, folded to 16 bits
    auto fold            = [](uint32_t dig) { return uint16_t((dig >> 8) ^ (dig >> 24)); }; // equal bits -> equal folds
    auto digest          = [&](K key, uint32_t k, uint32_t slot)
    { return (uint32_t((key - loK[k]) >> cutK[k]) << 8) | slot; };
    // leaf of bin entry m: last k with inK[k] <= m
    auto leafOfArrival = [&](uint32_t m)
    {
        uint32_t lo = 0, hi = nl;
        while (hi - lo > 1)
        {
            uint32_t mid = (lo + hi) / 2;
            if (inK[mid] <= m) lo = mid;
            else hi = mid;
        }
        return lo;
    };
    if constexpr (!COUNTING)
    {
        // old positions: all loads of a thread are issued before the first is used
        K key[ITER];
#pragma unroll
        for (int i = 0; i < ITER; ++i)
        {
            const uint32_t p = p0 + t + 256u * i;
            if (p < p1) key[i] = keysIn[p];
        }
#pragma unroll
        for (int i = 0; i < ITER; ++i)
        {
            const uint32_t p = p0 + t + 256u * i;
            if (p < p1)
            {
                sKey[p - p0] = key[i];
                sIdx[p - p0] = uint16_t(p - p0);
            }
        }
    }
    else
    {
        // old positions -> digests, in two batches (registers: key, leaf-start word and rank of every load in flight)
        constexpr int HALF = (ITER + 1) / 2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            K key[HALF];
            uint64_t word[HALF];
            uint32_t rk[HALF];
#pragma unroll
            for (int i = 0; i < HALF; ++i)
            {
                const uint32_t p = p0 + t + 256u * (h * HALF + i);
                if (p < p1)
                {
                    key[i]  = keysIn[p];
                    word[i] = mask[p >> 6];
                    rk[i]   = rank[p >> 6];
                }
            }
#pragma unroll
            for (int i = 0; i < HALF; ++i)
            {
                const uint32_t p = p0 + t + 256u * (h * HALF + i);
                if (p < p1)
                {
                    const uint32_t k = rk[i] + uint32_t(__popcll(word[i] & ((2ull << (p & 63u)) - 1))) - 1u - j0;
                    sDig[p - p0]     = key[i] == HOLE ? ~0u : digest(key[i], k, p - posK[k]);
                }
            }
        }
    }
    if constexpr (COUNTING)
    {
        for (uint32_t m = in0 + t; m < in1; m += 256)
        {
            const uint32_t k          = leafOfArrival(m);
            sDig[nOldAll + (m - in0)] = digest(binKeys[m], k, (posK[k + 1] - posK[k]) + (m - inK[k]));
        }
    }
    __syncthreads();

    if constexpr (!COUNTING)
    {
        if (t < nl)
        {
            const uint32_t s = posK[t] - p0, nOld = posK[t + 1] - posK[t];
            if (nOld > 1)
            {
                const uint32_t e = s + nOld;
                K pk             = sKey[s];
                K nk             = sKey[s + 1];
                for (uint32_t a = s + 1; a < e; ++a)
                {
                    const K ka = nk;
                    if (a + 1 < e) nk = sKey[a + 1];
                    if (ka > pk)
                    {
                        pk = ka;
                        continue;
                    }
                    const uint32_t ia = sIdx[a];
                    if (ka == pk && ia > sIdx[a - 1]) continue;
                    uint32_t b = a;
                    while (b > s)
                    {
                        const K kb        = sKey[b - 1];
                        const uint32_t ib = sIdx[b - 1];
                        if (kb < ka || (kb == ka && ib < ia)) break;
                        sKey[b] = kb;
                        sIdx[b] = ib;
                        --b;
                    }
                    sKey[b] = ka;
                    sIdx[b] = ia;
                }
            }
        }
        __syncthreads();
        // every leaf kept its size: slot e of the tile goes to layoutNew[first leaf] + e
        const uint32_t out0 = outK[0];
#pragma unroll
        for (int i = 0; i < ITER; ++i)
        {
            const uint32_t e = t + 256u * i;
            if (e < nOldAll)
            {
                keysOut[out0 + e]  = sKey[e];
                orderOut[out0 + e] = p0 + sIdx[e];
            }
        }
        return;
    }

    if constexpr (COUNTING)
    {
    // ---- counting.  A lane takes a chunk of four consecutive old slots of ONE leaf and scans the leaf's digests once
    // for all four: LDS reads / 4, two 32-bit vector instructions per comparison.
    constexpr int CHUNK_ITER = (RESORT_TILE_SLOTS / 4 + G + 255) / 256;
    const uint32_t numChunks = chunkK[nl];
    uint32_t chunkLeaf[CHUNK_ITER], chunkSlot[CHUNK_ITER], place[CHUNK_ITER][4];
#pragma unroll
    for (int i = 0; i < CHUNK_ITER; ++i)
    {
        const uint32_t c = t + 256u * i;
        if (c < numChunks)
        {
            uint32_t lo = 0, hi = nl; // leaf of the chunk: last k with chunkK[k] <= c
            while (hi - lo > 1)
            {
                uint32_t mid = (lo + hi) / 2;
                if (chunkK[mid] <= c) lo = mid;
                else hi = mid;
            }
            const uint32_t k  = lo;
            const uint32_t o0 = posK[k] - p0, nOld = posK[k + 1] - posK[k];
            const uint32_t a0 = nOldAll + (inK[k] - in0), nInc = inK[k + 1] - inK[k];
            const uint32_t s0 = 4 * (c - chunkK[k]); // first slot of the chunk inside the leaf
            chunkLeaf[i] = k, chunkSlot[i] = s0;
            uint32_t d[4], cnt[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                d[j] = s0 + j < nOld ? sDig[o0 + s0 + j] : ~0u; // (slots behind the leaf's end count as holes)
            uint32_t q = 0;
            for (; q + 4 <= nOld; q += 4)
            {
                const uint32_t v0 = sDig[o0 + q], v1 = sDig[o0 + q + 1], v2 = sDig[o0 + q + 2], v3 = sDig[o0 + q + 3];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    cnt[j] += (v0 < d[j]) + (v1 < d[j]) + (v2 < d[j]) + (v3 < d[j]);
            }
            for (; q < nOld; ++q)
            {
                const uint32_t v0 = sDig[o0 + q];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    cnt[j] += v0 < d[j];
            }
            for (q = 0; q < nInc; ++q)
            {
                const uint32_t v0 = sDig[a0 + q];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    cnt[j] += v0 < d[j];
            }
            const uint32_t base = outK[k] - outK[0];
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                place[i][j] = cnt[j];
                if (d[j] != ~0u) sNew[base + cnt[j]] = fold(d[j]);
            }
        }
    }
    auto placeByDigest = [&](uint32_t k, uint32_t dx)
    {
        const uint32_t o0 = posK[k] - p0, nOld = posK[k + 1] - posK[k];
        const uint32_t a0 = nOldAll + (inK[k] - in0), nInc = inK[k + 1] - inK[k];
        uint32_t less = 0;
        for (uint32_t q = 0; q < nOld; ++q)
            less += sDig[o0 + q] < dx;
        for (uint32_t q = 0; q < nInc; ++q)
            less += sDig[a0 + q] < dx;
        return less;
    };
    for (uint32_t m = in0 + t; m < in1; m += 256)
    {
        const uint32_t k = leafOfArrival(m), d = sDig[nOldAll + (m - in0)];
        sNew[(outK[k] - outK[0]) + placeByDigest(k, d)] = fold(d);
    }
    __syncthreads();

    // an element whose neighbour in the new order of its leaf has the same (folded) leading key bits is placed again, by
    // key and old index proper, from global memory (the others are where they belong: leading bits that differ decide)
    auto clashes = [&](uint32_t k, uint32_t pl, uint16_t bitsX)
    {
        const uint32_t base = outK[k] - outK[0], cnt = outK[k + 1] - outK[k];
        return (pl > 0 && sNew[base + pl - 1] == bitsX) || (pl + 1 < cnt && sNew[base + pl + 1] == bitsX);
    };
    auto placeExact = [&](uint32_t k, K kx, uint32_t ix)
    {
        const uint32_t nOld = posK[k + 1] - posK[k], nInc = inK[k + 1] - inK[k];
        uint32_t less = 0;
        for (uint32_t q = 0; q < nOld; ++q)
        {
            const K kq = keysIn[posK[k] + q]; // (a hole is larger than any key)
            less += (kq < kx || (kq == kx && posK[k] + q < ix)) ? 1u : 0u;
        }
        for (uint32_t q = 0; q < nInc; ++q)
        {
            const K kq = binKeys[inK[k] + q];
            less += (kq < kx || (kq == kx && binIdx[inK[k] + q] < ix)) ? 1u : 0u;
        }
        return less;
    };
#pragma unroll
    for (int i = 0; i < CHUNK_ITER; ++i)
    {
        const uint32_t c = t + 256u * i;
        if (c < numChunks)
        {
            const uint32_t k = chunkLeaf[i], nOld = posK[k + 1] - posK[k];
            const uint32_t p = posK[k] + chunkSlot[i]; // position of the chunk's first slot
            K kx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                kx[j] = chunkSlot[i] + j < nOld ? keysIn[p + j] : HOLE;
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                if (kx[j] != HOLE)
                {
                    uint32_t pl = place[i][j];
                    if (clashes(k, pl, fold(sDig[p - p0 + j]))) pl = placeExact(k, kx[j], p + j);
                    keysOut[outK[k] + pl]  = kx[j];
                    orderOut[outK[k] + pl] = p + j;
                }
            }
        }
    }
    for (uint32_t m = in0 + t; m < in1; m += 256)
    {
        const uint32_t k = leafOfArrival(m), d = sDig[nOldAll + (m - in0)];
        const K kx        = binKeys[m];
        const uint32_t ix = binIdx[m];
        uint32_t pl       = placeByDigest(k, d);
        if (clashes(k, pl, fold(d))) pl = placeExact(k, kx, ix);
        keysOut[outK[k] + pl]  = kx;
        orderOut[outK[k] + pl] = ix;
    }
    }
}

/*! The leaf pass for tiles in which something moved, second formulation (round 3).  Ordering a leaf by counting costs
 *  (leaf size)^2 comparisons whatever the lanes do; here every element first drops into one of 8 buckets of its leaf
 *  by the three leading bits of its digest (the octant of the leaf's cell it lies in: one returning LDS atomic), then
 *  counts the smaller digests among the handful of elements of its own bucket only.  The thread that loaded an element
 *  keeps its key in registers through both phases and stores it itself: no key is read twice, no key lives in LDS
 *  (21 KB of LDS: the grouped digests and the bucket tables).
 *  The kernel is a chain of LDS round trips per element (leaf search, atomic, bucket scan): written element by
 *  element, a thread's 17 elements wait for each other's LDS latencies one after the other (that form took the same
 *  0.9 ms as counting over the whole leaf).  So every step is written ACROSS a batch of the thread's elements, branch
 *  free: one search step / one atomic / one scan step for all of them, then the next -- nine independent LDS accesses in
 *  flight per wait.
 *  Digests as in leafSortKernel: 24 leading bits of (key - first key of the leaf), then the slot in the leaf; two
 *  elements of a bucket whose 24 bits agree are placed by key and old index proper (global memory, about one in 10^4).
 *  Same result: leaf j's elements at layoutNew[j]... in the order of (key, old index). */
template<class K, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void leafSortBucketsKernel(
    const K* __restrict__ keysIn, const uint64_t* __restrict__ mask, const uint32_t* __restrict__ rank,
    const K* __restrict__ leafLo, const uint32_t* __restrict__ leafPos, const uint32_t* __restrict__ inOffset,
    const uint32_t* __restrict__ layoutNew, const K* __restrict__ binKeys, const uint32_t* __restrict__ binIdx, uint32_t J,
    bool alwaysCount, K* __restrict__ keysOut, uint32_t* __restrict__ orderOut)
{
    constexpr int ITER  = (RESORT_TILE_SLOTS + 255) / 256;
    constexpr int BATCH = 6;
    constexpr K HOLE    = ~K(0);
    __shared__ uint32_t sGrp[RESORT_TILE_SLOTS]; // digests, grouped by (leaf, bucket)
    __shared__ uint32_t pinK[2 * (G + 1)];       // first old position | first bin entry of every leaf
    __shared__ uint32_t outK[G + 1];
    __shared__ K loK[G + 1];
    __shared__ uint8_t cutK[G];
    __shared__ uint32_t cntK[G * 8 + 1], baseK[G * 8 + 1]; // (the last entry: where elements that are none count)
    uint32_t* const posK = pinK;
    uint32_t* const inK  = pinK + (G + 1);

    const uint32_t j0 = blockIdx.x * uint32_t(G);
    const uint32_t nl = min(uint32_t(G), J - j0);
    const uint32_t t  = threadIdx.x;
    if (t <= nl)
    {
        posK[t] = leafPos[j0 + t];
        inK[t]  = inOffset[j0 + t];
        outK[t] = layoutNew[j0 + t];
        loK[t]  = leafLo[j0 + t];
    }
    for (uint32_t i = t; i < uint32_t(G) * 8 + 1; i += 256)
        cntK[i] = 0, baseK[i] = 0;
    __syncthreads();
    if (t < nl)
    {
        const K span   = loK[t + 1] - loK[t] - 1;
        const int bits = span ? int(8 * sizeof(K)) - clzKey(span) : 0;
        cutK[t]        = uint8_t(bits > 24 ? bits - 24 : 0);
    }
    const uint32_t p0 = posK[0], p1 = posK[nl], in0 = inK[0], in1 = inK[nl];
    const uint32_t nOldAll = p1 - p0, slots = nOldAll + (in1 - in0);
    if (slots > RESORT_TILE_SLOTS) return; // guarded by checkTilesKernel
    bool changed = in1 != in0 || alwaysCount;
    if (t < nl) changed = changed || (outK[t + 1] - outK[t]) != (posK[t + 1] - posK[t]);
    const bool quiet = !__syncthreads_or(changed) && nOldAll <= RESORT_QUIET_SLOTS;
    if (quiet) return; // the quiet instantiation of leafSortKernel takes this tile

    // ---- phase 1: every element (old slots, then arrivals) -> leaf, digest, bucket, place inside the bucket
    K key[ITER];
    uint32_t dig[ITER], info[ITER]; // info: leaf (8 bits) | place in the bucket (24 bits); ~0u: no element
#pragma unroll
    for (int i = 0; i < ITER; ++i)
    {
        const uint32_t e = t + 256u * i;
        key[i]           = HOLE;
        if (e < nOldAll) key[i] = keysIn[p0 + e];
        else if (e < slots) key[i] = binKeys[in0 + (e - nOldAll)];
    }
#pragma unroll
    for (int bs = 0; bs < ITER; bs += BATCH)
    {
        uint32_t val[BATCH], lo[BATCH], off[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
        {
            const int i      = bs + j < ITER ? bs + j : ITER - 1;
            const uint32_t e = t + 256u * i;
            const bool isOld = e < nOldAll, any = e < slots;
            // what is searched for: the old position among the leaves' first positions, or the bin entry among the leaves'
            // first entries (an element that is none searches for the tile's first position: in range, result unused)
            val[j] = !any ? p0 : (isOld ? p0 + e : in0 + (e - nOldAll));
            off[j] = (any && !isOld) ? uint32_t(G + 1) : 0u;
            lo[j]  = 0;
        }
        // last leaf k of the tile with pinK[off + k] <= val, all elements of the batch one search step at a time
        for (uint32_t n = nl; n > 1;)
        {
            const uint32_t half = n >> 1;
#pragma unroll
            for (int j = 0; j < BATCH; ++j)
            {
                const uint32_t probe = pinK[off[j] + lo[j] + half];
                lo[j]                = probe <= val[j] ? lo[j] + half : lo[j];
            }
            n -= half;
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
        {
            if (bs + j >= ITER) continue; // (compile time)
            const int i      = bs + j;
            const uint32_t e = t + 256u * i;
            const bool ok    = e < slots && key[i] != HOLE;
            const uint32_t k = lo[j];
            const uint32_t slot = off[j] ? (posK[k + 1] - posK[k]) + (val[j] - inK[k]) : val[j] - posK[k];
            const uint32_t d    = (uint32_t((key[i] - loK[k]) >> cutK[k]) << 8) | (slot & 0xFFu);
            const uint32_t at   = atomicAdd(&cntK[ok ? k * 8 + (d >> 29) : uint32_t(G) * 8], 1u);
            dig[i]  = ok ? d : ~0u;
            info[i] = ok ? (k << 24) | at : ~0u;
        }
    }
    __syncthreads();
    // ---- bucket starts in the tile's NEW order (leaf k starts at outK[k] - outK[0])
    if (t < nl)
    {
        uint32_t run = outK[t] - outK[0];
#pragma unroll
        for (int b = 0; b < 8; ++b)
        {
            baseK[t * 8 + b] = run;
            run += cntK[t * 8 + b];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITER; ++i)
        if (info[i] != ~0u) sGrp[baseK[(info[i] >> 24) * 8 + (dig[i] >> 29)] + (info[i] & 0xFFFFFFu)] = dig[i];
    __syncthreads();

    // ---- phase 2: rank inside the bucket, store
    auto placeExact = [&](uint32_t k, K kx, uint32_t ix)
    {
        const uint32_t nOld = posK[k + 1] - posK[k], nInc = inK[k + 1] - inK[k];
        uint32_t less = 0;
        for (uint32_t q = 0; q < nOld; ++q)
        {
            const K kq = keysIn[posK[k] + q]; // (a hole is larger than any key)
            less += (kq < kx || (kq == kx && posK[k] + q < ix)) ? 1u : 0u;
        }
        for (uint32_t q = 0; q < nInc; ++q)
        {
            const K kq = binKeys[inK[k] + q];
            less += (kq < kx || (kq == kx && binIdx[inK[k] + q] < ix)) ? 1u : 0u;
        }
        return less;
    };
#pragma unroll
    for (int bs = 0; bs < ITER; bs += BATCH)
    {
        uint32_t bb[BATCH], nb[BATCH], less[BATCH], same[BATCH];
        uint32_t maxNb = 0;
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
        {
            const int i       = bs + j < ITER ? bs + j : ITER - 1;
            const bool ok     = bs + j < ITER && info[i] != ~0u;
            const uint32_t bk = ok ? (info[i] >> 24) * 8 + (dig[i] >> 29) : uint32_t(G) * 8;
            bb[j]   = baseK[bk];
            nb[j]   = ok ? cntK[bk] : 0u;
            less[j] = 0, same[j] = 0;
            maxNb   = max(maxNb, nb[j]);
        }
        // the buckets of all elements of the batch, one entry at a time
        for (uint32_t q = 0; q < maxNb; ++q)
        {
#pragma unroll
            for (int j = 0; j < BATCH; ++j)
            {
                const int i      = bs + j < ITER ? bs + j : ITER - 1;
                const bool in    = q < nb[j];
                const uint32_t v = sGrp[bb[j] + (in ? q : 0u)];
                less[j] += (in && v < dig[i]) ? 1u : 0u;
                same[j] += (in && ((v ^ dig[i]) >> 8) == 0) ? 1u : 0u; // (itself included)
            }
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
        {
            if (bs + j >= ITER) continue; // (compile time)
            const int i = bs + j;
            if (info[i] == ~0u) continue;
            const uint32_t e  = t + 256u * i;
            const uint32_t k  = info[i] >> 24;
            const uint32_t ix = e < nOldAll ? p0 + e : binIdx[in0 + (e - nOldAll)];
            uint32_t at       = outK[0] + bb[j] + less[j];
            if (same[j] > 1) at = outK[k] + placeExact(k, key[i], ix);
            keysOut[at]  = key[i];
            orderOut[at] = ix;
        }
    }
}

/*! Positions of the leaf boundaries of ANOTHER tree (the focus tree after its rebalance) in the keys this re-sort has just
 *  ordered: the leaf table knows where every old leaf starts now (layoutNew), so the search for a boundary key only
 *  covers the particles of the one old leaf whose key range holds it -- a few dozen keys instead of all of them. */
template<class K>
__global__ __launch_bounds__(256) void bracketedPositionsKernel(const K* __restrict__ tree, NodeIdx numNodes,
                                                                const K* __restrict__ keys,
                                                                const K* __restrict__ leafLo,
                                                                const uint32_t* __restrict__ numCompact,
                                                                const uint32_t* __restrict__ coarse,
                                                                const uint32_t* __restrict__ layoutNew,
                                                                uint32_t* __restrict__ pos)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i > numNodes) return;
    const K key      = tree[i];
    const uint32_t J = *numCompact;
    uint32_t lo = 0, hi = J + 1; // last j in [0, J] with leafLo[j] <= key
    if (coarse)
    {
        constexpr int shift = 3 * int(maxLevel<K>()) - RESORT_COARSE_BITS;
        const uint32_t c    = uint32_t(key >> shift);
        lo                  = coarse[c];
        hi                  = c < (1u << RESORT_COARSE_BITS) ? coarse[c + 1] + 1 : J + 1;
    }
    while (hi - lo > 1)
    {
        uint32_t mid = (lo + hi) / 2;
        if (leafLo[mid] <= key) lo = mid;
        else hi = mid;
    }
    // first index with keys[index] >= key, inside that leaf's new range -- its start, without looking at any key, when
    // the boundary is that of the old leaf itself (most leaves of a tree outlive an update)
    uint32_t a = layoutNew[lo], len = leafLo[lo] == key ? 0u : layoutNew[lo + 1] - a;
    while (len > 0)
    {
        uint32_t half = len >> 1;
        if (keys[a + half] < key) { a += half + 1, len -= half + 1; }
        else { len = half; }
    }
    pos[i] = a;
}

__global__ __launch_bounds__(256) void countsOfPositionsKernel(const uint32_t* __restrict__ pos, NodeIdx numNodes,
                                                               uint32_t maxCount, uint32_t* __restrict__ counts)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numNodes) return;
    uint32_t c = pos[i + 1] - pos[i];
    counts[i]  = c < maxCount ? c : maxCount;
}

//! the particles that carry the remove marker: behind every leaf, by ascending old position (idx sorted by the caller)
template<class K>
__global__ __launch_bounds__(256) void placeMarkersKernel(const uint32_t* __restrict__ idx, uint32_t count,
                                                          const uint32_t* __restrict__ layoutNew,
                                                          const uint32_t* __restrict__ numCompact,
                                                          K* __restrict__ keysOut, uint32_t* __restrict__ orderOut)
{
    uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= count) return;
    const uint32_t at = layoutNew[*numCompact] + r;
    keysOut[at]       = endKey<K>();
    orderOut[at]      = idx[r];
}

} // namespace

template<class K>
int LeafResort<K>::prepare(cstone_hip_ctx* ctx, const K* tree, const uint32_t* layout, int numLeaves, size_t n,
                           K* keysOut, bool expectMovers)
{
    StageTimer timer(ctx, CSTONE_STAGE_RESORT_BINS);
    numLeaves_         = numLeaves;
    n_                 = n;
    const size_t words = (n + 63) / 64 + 1;
    const size_t ent   = size_t(numLeaves) + 3; // J + 2 <= numLeaves + 2 table entries, one more for the scans' totals
    const size_t cap   = n / 8 + 1024;
    CS_TRY(mask_.ensure(ctx, words * 8));
    CS_TRY(rank_.ensure(ctx, words * 4));
    CS_TRY(popc_.ensure(ctx, words * 4));
    CS_TRY(leafLo_.ensure(ctx, ent * sizeof(K)));
    CS_TRY(leafPos_.ensure(ctx, ent * 4));
    CS_TRY(outCount_.ensure(ctx, ent * 4));
    CS_TRY(incoming_.ensure(ctx, ent * 4));
    CS_TRY(newCount_.ensure(ctx, ent * 4));
    CS_TRY(layoutNew_.ensure(ctx, ent * 4));
    CS_TRY(inOffset_.ensure(ctx, ent * 4));
    CS_TRY(moverKeys_.ensure(ctx, cap * sizeof(K)));
    CS_TRY(moverIdx_.ensure(ctx, cap * 4));
    CS_TRY(moverDest_.ensure(ctx, cap * 4));
    CS_TRY(moverSlot_.ensure(ctx, cap * 4));
    CS_TRY(binKeys_.ensure(ctx, cap * sizeof(K)));
    CS_TRY(binIdx_.ensure(ctx, cap * 4));

    int* scalars = ctx->devScalars + RESORT_SCALARS;
    CS_HIP(ctx, hipMemsetAsync(scalars, 0, 4 * sizeof(int), ctx->stream)); // [3]: the mover counter
    hipLaunchKernelGGL(leafStartWordsKernel, gridFor(numLeaves, 256), 256, 0, ctx->stream, layout, numLeaves, uint32_t(n),
                       mask_.as<uint64_t>(), popc_.as<uint32_t>(), uint32_t(words));
    // rank of every word and, in scalars[2], the number of non-empty leaves
    CS_TRY(arenaReserve(ctx, scanArenaBytes(words)));
    int rc = scanU32(ctx, popc_.as<uint32_t>(), rank_.as<uint32_t>(), words, 0u, false, (uint32_t*)scalars + 2);
    arenaReset(ctx);
    CS_TRY(rc);
    hipLaunchKernelGGL(fillCompactLeavesKernel<K>, gridFor(size_t(numLeaves) + 3, 256), 256, 0, ctx->stream, tree, layout,
                       numLeaves, mask_.as<uint64_t>(), rank_.as<uint32_t>(), (const uint32_t*)scalars + 2, uint32_t(n),
                       leafLo_.as<K>(), leafPos_.as<uint32_t>(), outCount_.as<uint32_t>(), incoming_.as<uint32_t>());
    // many movers expected (the previous sync had them): the coarse table that shortens their searches
    haveCoarse_ = expectMovers;
    constexpr uint32_t cells = (1u << RESORT_COARSE_BITS) + 1;
    CS_TRY(coarse_.ensure(ctx, size_t(cells) * 4)); // (4 MB, once: no allocation in the middle of a run)
    if (haveCoarse_)
    {
        hipLaunchKernelGGL(coarseLeafTableKernel<K>, gridFor(cells, 256), 256, 0, ctx->stream, leafLo_.as<K>(),
                           (const uint32_t*)scalars + 2, coarse_.as<uint32_t>());
    }
    CS_HIP(ctx, hipGetLastError());

    args_.keysOut    = keysOut;
    args_.leafStart  = mask_.as<uint64_t>();
    args_.leafRank   = rank_.as<uint32_t>();
    args_.leafLo     = leafLo_.as<K>();
    args_.outCount   = outCount_.as<uint32_t>();
    args_.moverKeys  = moverKeys_.as<K>();
    args_.moverIdx   = moverIdx_.as<uint32_t>();
    args_.moverCount = (uint32_t*)scalars + 3;
    args_.moverCap   = uint32_t(cap);
    return CSTONE_OK;
}

template<class K>
int LeafResort<K>::binMovers(cstone_hip_ctx* ctx, int leavesPerTile)
{
    StageTimer timer(ctx, CSTONE_STAGE_RESORT_BINS);
    int* scalars          = ctx->devScalars + RESORT_SCALARS;
    const uint32_t ent    = uint32_t(numLeaves_) + 3;
    const uint32_t* numJ  = (const uint32_t*)scalars + 2;
    const uint32_t* count = (const uint32_t*)scalars + 3;
    hipLaunchKernelGGL(binMoversKernel<K>, unsigned(ctx->numCu) * 16, 256, 0, ctx->stream, moverKeys_.as<K>(), count,
                       args_.moverCap, leafLo_.as<K>(), numJ, haveCoarse_ ? coarse_.as<uint32_t>() : nullptr,
                       incoming_.as<uint32_t>(), moverDest_.as<uint32_t>(), moverSlot_.as<uint32_t>());
    hipLaunchKernelGGL(newLeafSizesKernel, gridFor(ent, 256), 256, 0, ctx->stream, leafPos_.as<uint32_t>(),
                       outCount_.as<uint32_t>(), incoming_.as<uint32_t>(), numJ, ent, newCount_.as<uint32_t>(), scalars);
    CS_TRY(arenaReserve(ctx, 2 * scanArenaBytes(ent)));
    int rc = scanU32Pair(ctx, newCount_.as<uint32_t>(), layoutNew_.as<uint32_t>(), incoming_.as<uint32_t>(),
                         inOffset_.as<uint32_t>(), ent);
    arenaReset(ctx);
    CS_TRY(rc);
    hipLaunchKernelGGL(checkTilesKernel, gridFor(size_t(numLeaves_) / leavesPerTile + 1, 256), 256, 0, ctx->stream,
                       leafPos_.as<uint32_t>(), inOffset_.as<uint32_t>(), layoutNew_.as<uint32_t>(), numJ, count,
                       args_.moverCap, leavesPerTile, scalars);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

template<class K>
int LeafResort<K>::sortLeaves(cstone_hip_ctx* ctx, const K* keysIn, K* keysOut, uint32_t* orderOut, uint32_t numMovers,
                              uint32_t numMarkers, uint32_t J, int leavesPerTile, bool largeQuietTiles)
{
    if (numMovers)
    {
        StageTimer timer(ctx, CSTONE_STAGE_RESORT_BINS);
        hipLaunchKernelGGL(placeMoversKernel<K>, gridFor(numMovers, 256), 256, 0, ctx->stream, moverKeys_.as<K>(),
                           moverIdx_.as<uint32_t>(), moverDest_.as<uint32_t>(), moverSlot_.as<uint32_t>(), numMovers,
                           inOffset_.as<uint32_t>(), binKeys_.as<K>(), binIdx_.as<uint32_t>());
    }
    if (numMarkers)
    {
        // all markers are equal keys: their stable order is that of their old positions
        StageTimer timer(ctx, CSTONE_STAGE_RESORT_BINS);
        // their bin is the last one: it starts at movers - markers
        uint32_t* idx = binIdx_.as<uint32_t>() + (numMovers - numMarkers);
        CS_TRY(cstone_hip_sort_keys(ctx, 32, idx, numMarkers));
        hipLaunchKernelGGL(placeMarkersKernel<K>, gridFor(numMarkers, 256), 256, 0, ctx->stream, idx, numMarkers,
                           layoutNew_.as<uint32_t>(), (const uint32_t*)(ctx->devScalars + RESORT_SCALARS) + 2, keysOut,
                           orderOut);
    }
    if (J == 0) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_RESORT_LEAVES);
    const unsigned grid    = (J + unsigned(leavesPerTile) - 1) / unsigned(leavesPerTile);
    const bool alwaysCount = std::getenv("CSTONE_RESORT_COUNT") != nullptr; // tuning/tests: no quiet-tile shortcut
#define CSTONE_LEAF_SORT(G, COUNTING)                                                                                  \
    hipLaunchKernelGGL((leafSortKernel<K, G, COUNTING>), grid, 256, 0, ctx->stream, keysIn, mask_.as<uint64_t>(),      \
                       rank_.as<uint32_t>(), leafLo_.as<K>(), leafPos_.as<uint32_t>(), inOffset_.as<uint32_t>(),       \
                       layoutNew_.as<uint32_t>(), binKeys_.as<K>(), binIdx_.as<uint32_t>(), J, alwaysCount, keysOut,  \
                       orderOut)
    // quiet tiles and tiles in which something moved: one launch each over all tiles (without movers there are none of
    // the second kind)
    const bool someMoved = numMovers > 0 || alwaysCount || largeQuietTiles;
    // tiles in which something moved: buckets by the leading digest bits, counting inside a bucket (the default), or
    // counting over the whole leaf (the round-2 formulation, kept for comparison: CSTONE_RESORT_SCAN=1)
    static const bool scanLeaves = std::getenv("CSTONE_RESORT_SCAN") != nullptr;
#define CSTONE_LEAF_BUCKETS(G)                                                                                         \
    hipLaunchKernelGGL((leafSortBucketsKernel<K, G>), grid, 256, 0, ctx->stream, keysIn, mask_.as<uint64_t>(),         \
                       rank_.as<uint32_t>(), leafLo_.as<K>(), leafPos_.as<uint32_t>(), inOffset_.as<uint32_t>(),       \
                       layoutNew_.as<uint32_t>(), binKeys_.as<K>(), binIdx_.as<uint32_t>(), J, alwaysCount, keysOut,  \
                       orderOut)
    if (leavesPerTile == 64)
    {
        CSTONE_LEAF_SORT(64, false);
        if (someMoved && scanLeaves) CSTONE_LEAF_SORT(64, true);
        else if (someMoved) CSTONE_LEAF_BUCKETS(64);
    }
    else if (leavesPerTile == 32)
    {
        CSTONE_LEAF_SORT(32, false);
        if (someMoved && scanLeaves) CSTONE_LEAF_SORT(32, true);
        else if (someMoved) CSTONE_LEAF_BUCKETS(32);
    }
    else if (leavesPerTile == 16)
    {
        CSTONE_LEAF_SORT(16, false);
        if (someMoved && scanLeaves) CSTONE_LEAF_SORT(16, true);
        else if (someMoved) CSTONE_LEAF_BUCKETS(16);
    }
    else return fail(ctx, CSTONE_E_INTERNAL, "resort: %d leaves per workgroup not instantiated", leavesPerTile);
#undef CSTONE_LEAF_SORT
#undef CSTONE_LEAF_BUCKETS
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

template<class K>
int LeafResort<K>::countLeaves(cstone_hip_ctx* ctx, const K* tree, int numNodes, const K* keys, uint32_t maxCount,
                               uint32_t* counts)
{
    if (numNodes <= 0) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_NODE_COUNTS);
    CS_TRY(arenaReserve(ctx, alignUp(size_t(numNodes + 1) * sizeof(uint32_t)) + 1024));
    auto* pos = (uint32_t*)arenaTake(ctx, size_t(numNodes + 1) * sizeof(uint32_t));
    hipLaunchKernelGGL(bracketedPositionsKernel<K>, gridFor(size_t(numNodes) + 1, 256), 256, 0, ctx->stream, tree,
                       NodeIdx(numNodes), keys, leafLo_.as<K>(), (const uint32_t*)(ctx->devScalars + RESORT_SCALARS) + 2,
                       haveCoarse_ ? coarse_.as<uint32_t>() : nullptr, layoutNew_.as<uint32_t>(), pos);
    hipLaunchKernelGGL(countsOfPositionsKernel, gridFor(numNodes, 256), 256, 0, ctx->stream, pos, NodeIdx(numNodes),
                       maxCount, counts);
    arenaReset(ctx);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

template class LeafResort<uint32_t>;
template class LeafResort<uint64_t>;

} // namespace cship
