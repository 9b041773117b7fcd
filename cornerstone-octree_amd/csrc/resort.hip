// Incremental re-sort of Domain::sync: leaf table, mover binning and the leaf pass (see resort.hpp for the scheme).
// Replaces, for a sync whose particles mostly stayed in their leaves, the reference's full sort of all keys
// (sortByKeyGpu, R/primitives/primitives_gpu.cu:305-353).  gfx950: wave64, 160 KB LDS per CU.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "device_keys.hpp"
#include "resort.hpp"
#include "scan.hpp"

namespace cship
{
// CSTONE_RESORT_TRACE (tuning builds only, tools/leafpass_trace.py): thread 0 of every workgroup of the leaf pass for
// tiles with movers records wall-clock stamps (100 MHz) at its phase boundaries
#ifdef CSTONE_RESORT_TRACE
constexpr int RESORT_TRACE_SLOTS = 12;
__device__ uint64_t* g_resortTrace;
#define RESORT_TRACE(slot)                                                                                             \
    if (g_resortTrace && threadIdx.x == 0) g_resortTrace[size_t(blockIdx.x) * RESORT_TRACE_SLOTS + (slot)] = wall_clock64();
#else
#define RESORT_TRACE(slot)
#endif
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

// ---------------------------------------------------------------------------------------------------------------------
// The leaf pass for tiles in which something moved (round 3): ONE WAVE PER LEAF, sorted in registers.
//
// What the phase trace of the earlier formulations showed (tools/leafpass_trace.py, CSTONE_RESORT_TRACE): 25-30 us of a
// tile's 32 us went into its LDS phases -- every element looked its leaf up (six dependent, bank-conflicting reads), took
// a place by a returning atomic, scanned its bucket (or its whole leaf) entry by entry; the LDS pipe of a CU serves four
// such tiles at a time and is what those kernels waited for, whatever the arithmetic around it (counting over the leaf
// 0.97-1.07 ms at 10^8 particles, eight buckets per leaf 0.90-0.95 ms, the same with batched accesses 0.99-1.05 ms, with
// stores staged through LDS 1.03-1.10 ms).
// A leaf of the focus tree holds at most a bucket of particles and a wave has 64 lanes.  So: lane s of a wave loads slot
// s of the leaf (old slots, then the arrivals; coalesced) and builds the 32-bit digest (24 leading bits of key - first
// key of the leaf, then the slot number); the wave sorts the digests with a bitonic network in registers (21
// compare-exchange steps for 64 elements, no search: the leaf is the wave's).  The partners come through DPP inside a
// row of 16 lanes and v_permlane16_swap / v_permlane32_swap across rows, all on the VALU: the same network over
// ds_bpermute was bound by the LDS pipe again (0.95 ms).  The place of every slot in the new order goes back to the lane
// that loaded it through a 16-bit table per wave in LDS (two LDS accesses per element), and that lane stores its key and
// old index at layoutNew[leaf] + place: every key is read once and never leaves its register.  Holes (digest ~0) sort
// last.  Leaves with 65..255 slots take two or four elements per lane.  Two sorted neighbours with equal leading bits
// (equal keys among them; about one leaf in 10^4) send the leaf to the exact path: every element counts the elements
// in front of it by (key, old index) proper.  A leaf nothing arrived in and whose remaining keys are still in order goes
// out as it came in, without the network.
// The loop over the wave's leaves is ROLLED (see below), and the wave number is deliberately NOT declared wave-uniform:
// with readfirstlane on it the leaf's table entries travel through scalar registers and every leaf starts with a chain of
// LDS read -> s_waitcnt -> v_readfirstlane -> scalar ALU before the first load can be issued (0.74 / 0.83 ms against
// 0.66 / 0.75 ms, 1 % movers / everything drifting).
// ---------------------------------------------------------------------------------------------------------------------
// ---- cross-lane partners on the VALU (no LDS pipe): DPP within a row of 16 lanes, v_permlane16/32_swap (gfx950) across
//      rows; lane mappings verified with tools/dpp_probe.hip on the hardware
template<int CTRL, int BANK>
__device__ __forceinline__ uint32_t dppMov(uint32_t old, uint32_t v)
{
    return uint32_t(__builtin_amdgcn_update_dpp(int(old), int(v), CTRL, 0xF, BANK, false));
}
//! a permutation inside the row in which every lane has a source: written so that the compiler may fold it into the
//! operation that consumes it (v_min_u32_dpp / v_max_u32_dpp)
template<int CTRL>
__device__ __forceinline__ uint32_t dppAll(uint32_t v)
{
    return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, 0xF, 0xF, true));
}
//! value of lane ^ X
template<int X>
__device__ __forceinline__ uint32_t laneXor(uint32_t v, unsigned lane)
{
    if constexpr (X == 1) return dppAll<0xB1>(v);        // quad_perm [1,0,3,2]
    else if constexpr (X == 2) return dppAll<0x4E>(v);   // quad_perm [2,3,0,1]
    else if constexpr (X == 3) return dppAll<0x1B>(v);   // quad_perm [3,2,1,0]
    else if constexpr (X == 7) return dppAll<0x141>(v);  // row_half_mirror
    else if constexpr (X == 15) return dppAll<0x140>(v); // row_mirror
    else if constexpr (X == 4)
    {
        const uint32_t t = dppMov<0x104, 0x5>(v, v); // row_shl:4 into banks 0 and 2 (lane reads lane + 4)
        return dppMov<0x114, 0xA>(t, v);             // row_shr:4 into banks 1 and 3 (lane reads lane - 4)
    }
    else if constexpr (X == 8)
    {
        const uint32_t t = dppMov<0x108, 0x3>(v, v);
        return dppMov<0x118, 0xC>(t, v);
    }
    else if constexpr (X == 16)
    {
        auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false); // [0]: rows 0,0,2,2   [1]: rows 1,1,3,3
        return (lane & 16u) ? p[0] : p[1];
    }
    else if constexpr (X == 32)
    {
        auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false); // [0]: lower half twice   [1]: upper half twice
        return (lane & 32u) ? p[0] : p[1];
    }
    else if constexpr (X == 31) return laneXor<15>(laneXor<16>(v, lane), lane);
    else
    {
        static_assert(X == 63, "partner not implemented");
        return laneXor<15>(laneXor<16>(laneXor<32>(v, lane), lane), lane);
    }
}

//! compare-exchange with the lane at distance X, for N independent registers at once (one step of N sorts: their
//! instructions interleave, which fills the wait states between a VALU write and a DPP read); the lane whose bit LOWBIT
//! is clear is the lower one and keeps the minimum
template<int X, unsigned LOWBIT, int N>
__device__ __forceinline__ void cmpExchange(uint32_t (&v)[N], unsigned lane)
{
    if constexpr ((X == 4 || X == 8) && LOWBIT == unsigned(X))
    {
        // the lower lanes of such a step are whole DPP banks (four lanes each): the minimum goes to them through a masked
        // row_shl, the maximum to the upper banks through a masked row_shr of the ORIGINAL values; lanes outside a mask
        // keep their operand (old), so no select is needed -- four instructions (two when the moves fold) instead of five
#pragma unroll
        for (int n = 0; n < N; ++n)
        {
            const uint32_t a = v[n];
            uint32_t r       = a;
            // (written as instructions: the compiler does not fold a masked DPP move into its consumer.  s_nop 1: a DPP
            //  operand written by the instruction before needs two wait states, and the assembler text hides the DPP
            //  from the compiler's hazard pass)
            if constexpr (X == 4)
            {
                asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %1, %0 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(r) : "v"(a));
                asm volatile("v_max_u32_dpp %0, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(r) : "v"(a));
            }
            else
            {
                asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0x3" : "+v"(r) : "v"(a));
                asm volatile("v_max_u32_dpp %0, %1, %0 row_shr:8 row_mask:0xf bank_mask:0xc" : "+v"(r) : "v"(a));
            }
            v[n] = r;
        }
        return;
    }
    const bool upper = (lane & LOWBIT) != 0;
#pragma unroll
    for (int n = 0; n < N; ++n)
    {
        const uint32_t o = laneXor<X>(v[n], lane);
        v[n]             = upper ? max(v[n], o) : min(v[n], o);
    }
}

//! the half-cleaners behind a merge step: distances J, J/2, ... 1
template<int J, int N>
__device__ __forceinline__ void halfClean(uint32_t (&v)[N], unsigned lane)
{
    if constexpr (J >= 1)
    {
        cmpExchange<J, unsigned(J), N>(v, lane);
        halfClean<J / 2, N>(v, lane);
    }
}

/*! N independent ascending sorts of 64 values each (register n of every lane: one sort): bitonic network without
 *  direction flags -- every merge of two sorted blocks starts with a compare-exchange against the MIRRORED position
 *  (distance k - 1), followed by half-cleaners at distances k/4 ... 1, the lower lane always keeps the minimum. */
template<int N>
__device__ __forceinline__ void waveSort64(uint32_t (&v)[N], unsigned lane)
{
    cmpExchange<1, 1u, N>(v, lane);                              // blocks of 2
    cmpExchange<3, 2u, N>(v, lane), halfClean<1, N>(v, lane);    // 4
    cmpExchange<7, 4u, N>(v, lane), halfClean<2, N>(v, lane);    // 8
    cmpExchange<15, 8u, N>(v, lane), halfClean<4, N>(v, lane);   // 16
    cmpExchange<31, 16u, N>(v, lane), halfClean<8, N>(v, lane);  // 32
    cmpExchange<63, 32u, N>(v, lane), halfClean<16, N>(v, lane); // 64
}

//! ascending sort of the 64 R values d[r] (element 64 r + lane) of a wave
template<int R>
__device__ __forceinline__ void waveBitonicSort(uint32_t (&d)[R], unsigned lane)
{
    waveSort64<R>(d, lane);
    if constexpr (R >= 2)
    {
        // blocks of 128: element (r, lane) against (r ^ 1, 63 - lane)
#pragma unroll
        for (int r = 0; r < R; r += 2)
        {
            const uint32_t a = d[r], b = d[r + 1];
            d[r]     = min(a, laneXor<63>(b, lane));
            d[r + 1] = max(b, laneXor<63>(a, lane));
        }
        halfClean<32, R>(d, lane);
    }
    if constexpr (R == 4)
    {
        // blocks of 256: (r, lane) against (3 - r, 63 - lane), then distance 64 (registers r, r ^ 1), then within registers
        const uint32_t a0 = d[0], a1 = d[1], a2 = d[2], a3 = d[3];
        d[0] = min(a0, laneXor<63>(a3, lane)), d[3] = max(a3, laneXor<63>(a0, lane));
        d[1] = min(a1, laneXor<63>(a2, lane)), d[2] = max(a2, laneXor<63>(a1, lane));
        const uint32_t b0 = d[0], b1 = d[1], b2 = d[2], b3 = d[3];
        d[0] = min(b0, b1), d[1] = max(b0, b1), d[2] = min(b2, b3), d[3] = max(b2, b3);
        halfClean<32, R>(d, lane);
    }
}

// ---- two elements per lane: lane l of a segment holds the elements 2 l (register 0) and 2 l + 1 (register 1) of its
//      leaf, so that a leaf of up to 64 slots takes HALF a wave and a wave step sorts two leaves (or four of up to 32
//      slots).  Six of the 21 steps of the 64-element network are then compare-exchanges inside a lane (two
//      instructions for both registers, no cross-lane move, no select), the others run on both registers at half the
//      lane distance.
__device__ __forceinline__ void inLaneExchange(uint32_t (&d)[2])
{
    const uint32_t lo = min(d[0], d[1]);
    d[1]              = max(d[0], d[1]);
    d[0]              = lo;
}
//! first step of the merge of two sorted blocks of K/2 elements each: element e against the mirrored element
//! e ^ (K - 1), i.e. (lane, register) against (lane ^ M, other register), M = K/2 - 1; LOWBIT = K/4 in lane space
template<int M, unsigned LOWBIT>
__device__ __forceinline__ void mirrorExchange(uint32_t (&d)[2], unsigned lane)
{
    const uint32_t o1 = laneXor<M>(d[1], lane), o0 = laneXor<M>(d[0], lane);
    const bool upper  = (lane & LOWBIT) != 0;
    d[0]              = upper ? max(d[0], o1) : min(d[0], o1);
    d[1]              = upper ? max(d[1], o0) : min(d[1], o0);
}
//! ascending sort of the 2 * SEG elements of every segment of SEG lanes (SEG = 32: 64 elements, SEG = 16: 32)
template<int SEG>
__device__ __forceinline__ void pairSort(uint32_t (&d)[2], unsigned lane)
{
    inLaneExchange(d);                                                                       // blocks of 2
    mirrorExchange<1, 1u>(d, lane), inLaneExchange(d);                                       // 4
    mirrorExchange<3, 2u>(d, lane), cmpExchange<1, 1u, 2>(d, lane), inLaneExchange(d);       // 8
    mirrorExchange<7, 4u>(d, lane), halfClean<2, 2>(d, lane), inLaneExchange(d);             // 16
    mirrorExchange<15, 8u>(d, lane), halfClean<4, 2>(d, lane), inLaneExchange(d);            // 32
    if constexpr (SEG >= 32) { mirrorExchange<31, 16u>(d, lane), halfClean<8, 2>(d, lane), inLaneExchange(d); } // 64
}

#ifdef CSTONE_WAVE_OCC4
#define CSTONE_WAVE_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define CSTONE_WAVE_OCC
#endif
// ---- the FIELD-CARRYING flavour of the leaf pass (round 4).  A sync used to make four passes over the particle arrays:
//      encode (x, y, z read), leaf pass (keys), gather of h, gather of x, y, z -- 132 bytes per particle.  The lane that
//      holds slot s of a leaf knows where that slot goes as soon as the leaf is ordered, so it can carry x, y, z, h of the
//      slot along: they are loaded with the key (the stayers' from the old arrays at the slot's own position: coalesced;
//      the arrivals' from their bins, where the encode kernel and placeMoversKernel have put them) and stored at the new
//      place with the key.  The gathers disappear: 8 + 32 bytes read, 12 + 32 written per particle, and the maximum of h
//      over the leaf's new content -- what Halos::discover wants per leaf -- is folded by the wave on the way.
template<class T>
struct LeafFields
{
    static constexpr bool on = true;
    using Real               = T;
    const T *x, *y, *z, *h;     // the old order (an arrival's values are fetched from its old position)
    T *ox, *oy, *oz, *oh;       // the new order
    T* hmax;                    // [J]: max h over the new content of every compact leaf (0 for an empty one)
};
struct NoLeafFields
{
    static constexpr bool on = false;
    using Real               = float;
};
template<class T>
struct LeafVal
{
    T x, y, z, h;
};

//! value of lane ^ X for a float or a double (both halves through the VALU permutations above)
template<int X, class T>
__device__ __forceinline__ T laneXorReal(T v, unsigned lane)
{
    if constexpr (sizeof(T) == 4) { return __uint_as_float(laneXor<X>(__float_as_uint(v), lane)); }
    else
    {
        const uint64_t u  = uint64_t(__double_as_longlong(v));
        const uint32_t lo = laneXor<X>(uint32_t(u), lane), hi = laneXor<X>(uint32_t(u >> 32), lane);
        return __longlong_as_double((long long)((uint64_t(hi) << 32) | lo));
    }
}
//! maximum over the 2^BITS lanes of every aligned lane segment
template<int BITS, class T>
__device__ __forceinline__ T segmentMaxLanes(T v, unsigned lane)
{
    v = fmax(v, laneXorReal<1>(v, lane));
    v = fmax(v, laneXorReal<2>(v, lane));
    v = fmax(v, laneXorReal<4>(v, lane));
    v = fmax(v, laneXorReal<8>(v, lane));
    if constexpr (BITS >= 5) v = fmax(v, laneXorReal<16>(v, lane));
    if constexpr (BITS >= 6) v = fmax(v, laneXorReal<32>(v, lane));
    return v;
}

template<class K, int G, int MODE, class F>
__global__ __launch_bounds__(256) CSTONE_WAVE_OCC void leafSortWaveKernel(
    const K* __restrict__ keysIn, const K* __restrict__ leafLo, const uint32_t* __restrict__ leafPos,
    const uint32_t* __restrict__ inOffset, const uint32_t* __restrict__ layoutNew, const K* __restrict__ binKeys,
    const uint32_t* __restrict__ binIdx, uint32_t J, bool alwaysCount, bool allTiles, K* __restrict__ keysOut,
    uint32_t* __restrict__ orderOut, F fl)
{
    constexpr K HOLE = ~K(0);
    using Real       = typename F::Real;
    using Val        = LeafVal<Real>;
    __shared__ uint32_t posK[G + 1], inK[G + 1], outK[G + 1];
    __shared__ K loK[G + 1];
    __shared__ uint32_t slotsK[G + 4]; // old slots + arrivals of every leaf, zeros behind the last one
    __shared__ uint8_t cutK[G];
    __shared__ uint16_t sPlace[4][256];

    RESORT_TRACE(0)
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
    __syncthreads();
    if (t < uint32_t(G) + 4u) slotsK[t] = t < nl ? (posK[t + 1] - posK[t]) + (inK[t + 1] - inK[t]) : 0u;
    if (t < nl)
    {
        // how far a leaf's keys are shifted for the 24 leading bits of the digest: once per leaf here, not once per step
        const K span   = loK[t + 1] - loK[t] - 1;
        const int bits = span ? int(8 * sizeof(K)) - clzKey(span) : 0;
        cutK[t]        = uint8_t(bits > 24 ? bits - 24 : 0);
    }
    const uint32_t p0 = posK[0], p1 = posK[nl], in0 = inK[0], in1 = inK[nl];
    if (!allTiles)
    {
        // the quiet instantiation of leafSortKernel takes the tiles in which nothing moved (same vote as there)
        const uint32_t nOldAll = p1 - p0;
        bool changed           = in1 != in0 || alwaysCount;
        if (t < nl) changed = changed || (outK[t + 1] - outK[t]) != (posK[t + 1] - posK[t]);
        const bool quiet = !__syncthreads_or(changed) && nOldAll <= RESORT_QUIET_SLOTS;
        if (quiet) return;
    }
    __syncthreads(); // (slotsK, cutK)
    const unsigned lane = t & 63u;
    const unsigned wave = t >> 6;

    //! what the loop needs to know about leaf k
    struct Leaf
    {
        uint32_t pk, nOld, ik, nInc, ok, nNew, slots, k;
        K lo;
        unsigned cut;
    };
    auto leafOf = [&](uint32_t k)
    {
        Leaf f;
        f.pk = posK[k], f.nOld = posK[k + 1] - f.pk;
        f.ik = inK[k], f.nInc = inK[k + 1] - f.ik;
        f.ok = outK[k], f.nNew = outK[k + 1] - f.ok;
        f.slots = f.nOld + f.nInc;
        f.lo    = loK[k];
        f.cut   = cutK[k];
        f.k     = k;
        return f;
    };
    // slot s of a leaf: an old position or an arrival (key and old position; requested one wave step ahead)
    auto loadSlot = [&](const Leaf& f, uint32_t s, K& key, uint32_t& idx)
    {
        key = HOLE, idx = 0;
        if (s < f.nOld) key = keysIn[f.pk + s], idx = f.pk + s;
        else if (s < f.slots) key = binKeys[f.ik + (s - f.nOld)], idx = binIdx[f.ik + (s - f.nOld)];
    };
    // x, y, z, h of the particle that sat at old position idx.  They are requested only once the leaf is ordered, right
    // in front of the stores: the values then live in registers for a moment instead of across the network (with them
    // prefetched like the keys the kernel needed 101-116 VGPRs, four waves per SIMD, and the network -- a chain of
    // dependent cross-lane operations -- starved: 2.3-2.6 ms at 1e8 particles); the latency of these loads is covered
    // by the other seven waves of the SIMD
    auto loadFields = [&](uint32_t idx, bool have) -> Val
    {
        Val v{0, 0, 0, 0};
        if constexpr (F::on)
        {
            if (have) v = Val{fl.x[idx], fl.y[idx], fl.z[idx], fl.h[idx]};
        }
        return v;
    };
    // the values requested for the NEXT step are pinned (an empty asm that "uses" them) before the stores of this step go
    // out: the compiler then waits for those loads HERE, where they have had the whole step to return, instead of at the
    // top of the next step behind this step's stores -- vmcnt counts loads and stores in one in-order queue, and a wait
    // placed there has to cover the path of the loop that stores nothing, i.e. it waits for the stores as well
    auto pinNext = [&](K& key, uint32_t& idx) { asm volatile("" : "+v"(key), "+v"(idx)); };
    auto storeAt = [&](uint32_t at, K key, uint32_t idx, const Val& val)
    {
        keysOut[at]  = key;
        orderOut[at] = idx;
        if constexpr (F::on) fl.ox[at] = val.x, fl.oy[at] = val.y, fl.oz[at] = val.z, fl.oh[at] = val.h;
    };
    auto digestOf = [&](const Leaf& f, K key, uint32_t slot) -> uint32_t
    { return key == HOLE ? ~0u : ((uint32_t((key - f.lo) >> f.cut) << 8) | slot); };

    // the maximum of h over the new content of a leaf whose elements lie in the 2^BITS lanes of a segment (hm: this
    // lane's candidate, 0 for a lane without an element -- the reference's segmentMax starts from 0 as well); `writer`:
    // this lane reports for leaf k
    auto reportHmax = [&](auto bitsTag, Real hm, bool writer, uint32_t k)
    {
        if constexpr (F::on)
        {
            const Real m = segmentMaxLanes<decltype(bitsTag)::value>(hm, lane);
            if (writer) fl.hmax[j0 + k] = m;
        }
    };
    // the sorted digests of a leaf -> its new content: the place of every slot in the new order goes back to the lane that
    // loaded the slot (and still holds its key) through a 16-bit table per wave in LDS, two LDS accesses per element; that
    // lane stores key and old index at layoutNew[leaf] + place.  false: two neighbours in the new order have equal leading
    // key bits (the caller then places the leaf by keys and old indices proper)
    auto finishLeaf = [&](const Leaf& f, auto rTag, const K* key, const uint32_t* idx, uint32_t* d) -> bool
    {
        constexpr int R = decltype(rTag)::value;
        bool clash      = false;
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            // the element in front: the lane below, for lane 0 the last lane of the register before (all lanes shuffle)
            const uint32_t below = uint32_t(__shfl_up(int(d[r]), 1));
            const uint32_t last  = R > 1 ? uint32_t(__shfl(int(d[r > 0 ? r - 1 : 0]), 63)) : ~0u;
            const uint32_t prev  = lane != 0 ? below : (r > 0 ? last : ~0u);
            clash = clash || (d[r] != ~0u && prev != ~0u && (d[r] >> 8) == (prev >> 8));
        }
        if (__any(clash)) return false;
        uint16_t* place = sPlace[wave];
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (d[r] != ~0u) place[d[r] & 0xFFu] = uint16_t(lane + 64u * r);
        __builtin_amdgcn_wave_barrier();
        // (register after register: a leaf of more than 64 slots is rare, and four sets of values at once would set the
        //  register budget -- i.e. the occupancy -- of the whole kernel)
        Real hm = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            const Val val = loadFields(idx[r], key[r] != HOLE);
            if (key[r] == HOLE) continue;
            hm                = fmax(hm, val.h);
            const uint32_t at = place[lane + 64u * r];
            if (at < f.nNew) storeAt(f.ok + at, key[r], idx[r], val);
        }
        reportHmax(std::integral_constant<int, 6>{}, hm, lane == 0, f.k);
        __builtin_amdgcn_wave_barrier();
        return true;
    };
    // exact path: every loaded element counts the elements of its leaf in front of it by (key, old index)
    // (rHmax: also fold the leaf's maximum of h -- the caller has not)
    auto exactLeaf = [&](const Leaf& f, int R, const K* key, const uint32_t* idx, bool wholeWave)
    {
        Real hm = 0;
        for (int r = 0; r < R; ++r)
        {
            const Val v = loadFields(idx[r], key[r] != HOLE);
            if (key[r] != HOLE) hm = fmax(hm, v.h);
        }
        if (wholeWave) reportHmax(std::integral_constant<int, 6>{}, hm, lane == 0, f.k);
        for (int r = 0; r < R; ++r)
        {
            if (key[r] == HOLE) continue;
            const Val val_r = loadFields(idx[r], true);
            uint32_t less = 0;
            for (uint32_t q = 0; q < f.nOld; ++q)
            {
                const K kq = keysIn[f.pk + q]; // (a hole is larger than any key)
                less += (kq < key[r] || (kq == key[r] && f.pk + q < idx[r])) ? 1u : 0u;
            }
            for (uint32_t q = 0; q < f.nInc; ++q)
            {
                const K kq = binKeys[f.ik + q];
                less += (kq < key[r] || (kq == key[r] && binIdx[f.ik + q] < idx[r])) ? 1u : 0u;
            }
            storeAt(f.ok + less, key[r], idx[r], val_r);
        }
    };
    auto bigLeaf = [&](const Leaf& f, auto rTag)
    {
        constexpr int R = decltype(rTag)::value;
        K key[R];
        uint32_t idx[R], d[R];
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            loadSlot(f, lane + 64u * r, key[r], idx[r]);
            d[r] = digestOf(f, key[r], lane + 64u * r);
        }
        waveBitonicSort<R>(d, lane);
        if (!finishLeaf(f, rTag, key, idx, d)) exactLeaf(f, R, key, idx, true);
    };

    if constexpr (MODE == 0)
    {
        // The leaves of this wave: wave, wave + 4, ...  A ROLLED loop: the body (load, 64-element network, store) is a few
        // hundred instructions and stays in the instruction cache; unrolled over the wave's 16 leaves the kernel was
        // 24-41 thousand instructions (190-330 KB of code streaming through the instruction cache) and took 0.7-0.9 ms
        // whatever the network cost.  The key of the NEXT leaf is requested before the current one is sorted.
        if (wave >= nl) return;
        Leaf cur = leafOf(wave);
        K keyN        = HOLE;
        uint32_t idxN = 0;
        if (cur.slots <= 64) loadSlot(cur, lane, keyN, idxN);
    #pragma unroll 1
        for (uint32_t k = wave; k < nl; k += 4)
        {
            const Leaf f     = cur;
            K key1[1]        = {keyN};
            uint32_t idx1[1] = {idxN};
            keyN = HOLE, idxN = 0;
            if (k + 4 < nl)
            {
                cur = leafOf(k + 4);
                if (cur.slots <= 64) loadSlot(cur, lane, keyN, idxN);
            }
            if (f.slots == 0)
            {
                if constexpr (F::on)
                    if (lane == 0) fl.hmax[j0 + f.k] = 0;
                continue;
            }
            if (f.slots <= 64)
            {
                // nothing arrived and what stayed is still in order (a departure leaves the largest key behind, so only
                // departures from the end of the leaf pass): the content goes out as it came in
                {
                    const K below     = K(__shfl_up((unsigned long long)key1[0], 1));
                    const bool behind = lane != 0 && key1[0] < below;
                    if (f.nInc == 0 && !__any(behind))
                    {
                        const Val v = loadFields(idx1[0], key1[0] != HOLE);
                        reportHmax(std::integral_constant<int, 6>{}, key1[0] != HOLE ? v.h : Real(0), lane == 0, f.k);
                        pinNext(keyN, idxN);
                        if (lane < f.nNew) storeAt(f.ok + lane, key1[0], idx1[0], v);
                        continue;
                    }
                }
                uint32_t d[1] = {digestOf(f, key1[0], lane)};
                waveBitonicSort<1>(d, lane);
                pinNext(keyN, idxN);
                if (!finishLeaf(f, std::integral_constant<int, 1>{}, key1, idx1, d)) exactLeaf(f, 1, key1, idx1, true);
            }
            else if (f.slots <= 128) { bigLeaf(f, std::integral_constant<int, 2>{}); }
            else { bigLeaf(f, std::integral_constant<int, 4>{}); }
        }
        return;
    }
    if constexpr (MODE == 1)
    {
        // ---- two elements per lane (see pairSort): a step takes TWO consecutive leaves of up to 64 slots (a segment of
        //      32 lanes each) or FOUR of up to 32 slots (16 lanes each); a leaf with more than 64 slots has the wave
        //      to itself as in the other flavours.  The wave's leaves: a contiguous quarter of the tile.
#ifdef CSTONE_RESORT_TRACE
        RESORT_TRACE(1)
        long long trSteps = 0, trIssue = 0, trWait = 0, trSort = 0, trFinish = 0, trT = clock64();
#define WAVE_TRACE(acc)                                                                                                \
    {                                                                                                                  \
        const long long now_ = clock64();                                                                              \
        acc += now_ - trT;                                                                                             \
        trT = now_;                                                                                                    \
    }
#else
#define WAVE_TRACE(acc)
#endif
        const uint32_t per    = (nl + 3u) / 4u;
        const uint32_t kBegin = min(nl, wave * per), kEnd = min(nl, kBegin + per);
        if (kBegin >= kEnd) return;
        // log2 of the segment size in lanes for the step that starts at leaf q; 6: one leaf, more than 64 slots
        auto segBits = [&](uint32_t q) -> unsigned
        {
            const uint32_t s0 = slotsK[q];
            const uint32_t s1 = q + 1 < kEnd ? slotsK[q + 1] : 0u;
            const uint32_t s2 = q + 2 < kEnd ? slotsK[q + 2] : 0u;
            const uint32_t s3 = q + 3 < kEnd ? slotsK[q + 3] : 0u;
            if (s0 > 64u) return 6u;
            const uint32_t m2 = max(s0, s1), m4 = max(m2, max(s2, s3));
            // (a second leaf with more than 64 slots waits for its own step: the segment next to the first stays empty)
            return m4 <= 32u ? 4u : 5u;
        };
        auto leafAt2 = [&](uint32_t q, unsigned bits, uint32_t secondSlots)
        {
            const uint32_t seg = bits == 6 ? 0u : (lane >> bits);
            const uint32_t kk  = q + seg;
            bool mine          = kk < kEnd;
            if (bits == 5 && seg == 1 && secondSlots > 64u) mine = false;
            Leaf f = leafOf(mine ? kk : q);
            if (!mine) f.nOld = f.nInc = f.nNew = f.slots = 0, f.k = ~0u;
            return f;
        };
        //! leaves the step at q with `bits` consumes
        auto stepLeaves = [&](uint32_t q, unsigned bits) -> uint32_t
        {
            if (bits == 6) return 1u;
            if (bits == 4) return 4u;
            return (q + 1 < kEnd && slotsK[q + 1] > 64u) ? 1u : 2u;
        };
        uint32_t k    = kBegin;
        unsigned bits = segBits(k);
        Leaf cur      = leafAt2(k, bits, k + 1 < kEnd ? slotsK[k + 1] : 0u);
        K keyN[2]        = {HOLE, HOLE};
        uint32_t idxN[2] = {0, 0};
        if (bits != 6)
        {
            const uint32_t sl = lane & ((1u << bits) - 1u);
            loadSlot(cur, 2 * sl, keyN[0], idxN[0]);
            loadSlot(cur, 2 * sl + 1, keyN[1], idxN[1]);
        }
#pragma unroll 1
        while (k < kEnd)
        {
            const Leaf f          = cur;
            const unsigned b      = bits;
            const K key[2]        = {keyN[0], keyN[1]};
            const uint32_t idx[2] = {idxN[0], idxN[1]};
            const uint32_t sl     = lane & ((1u << (b == 6 ? 5u : b)) - 1u);
            k += stepLeaves(k, b);
            keyN[0] = keyN[1] = HOLE, idxN[0] = idxN[1] = 0;
            if (k < kEnd)
            {
                bits = segBits(k);
                cur  = leafAt2(k, bits, k + 1 < kEnd ? slotsK[k + 1] : 0u);
                if (bits != 6)
                {
                    const uint32_t sn = lane & ((1u << bits) - 1u);
                    loadSlot(cur, 2 * sn, keyN[0], idxN[0]);
                    loadSlot(cur, 2 * sn + 1, keyN[1], idxN[1]);
                }
            }
#ifdef CSTONE_RESORT_TRACE
            ++trSteps;
            WAVE_TRACE(trIssue)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); // (the keys of this step: all but the loads just issued)
            WAVE_TRACE(trWait)
#endif
            if (b == 6)
            {
                if (f.slots <= 128) { bigLeaf(f, std::integral_constant<int, 2>{}); }
                else { bigLeaf(f, std::integral_constant<int, 4>{}); }
                continue;
            }
            // x, y, z, h of this lane's two elements, and the max h of every leaf of the step (a segment of 16 or 32 lanes
            // each; empty leaves report 0)
            Val val[2];
            auto stepFields = [&]()
            {
                val[0] = loadFields(idx[0], key[0] != HOLE);
                val[1] = loadFields(idx[1], key[1] != HOLE);
                if constexpr (F::on)
                {
                    Real hm = 0;
                    if (key[0] != HOLE) hm = fmax(hm, val[0].h);
                    if (key[1] != HOLE) hm = fmax(hm, val[1].h);
                    const bool writer = sl == 0 && f.k != ~0u;
                    if (b == 5) reportHmax(std::integral_constant<int, 5>{}, hm, writer, f.k);
                    else reportHmax(std::integral_constant<int, 4>{}, hm, writer, f.k);
                }
            };
            if (!__any(f.slots != 0))
            {
                stepFields();
                continue;
            }
            // the element in front of (lane, 0) is (lane - 1, 1), the one in front of (lane, 1) is (lane, 0)
            const K front = K(__shfl_up((unsigned long long)key[1], 1));
            // nothing arrived and what stayed is still in order: the content goes out as it came in
            const bool behind = (sl != 0 && key[0] < front) || key[1] < key[0];
            if (!__any(f.nInc != 0 || behind))
            {
                stepFields();
                pinNext(keyN[0], idxN[0]);
                pinNext(keyN[1], idxN[1]);
#pragma unroll
                for (int r = 0; r < 2; ++r)
                {
                    if (2 * sl + r < f.nNew) storeAt(f.ok + 2 * sl + r, key[r], idx[r], val[r]);
                }
                continue;
            }
            uint32_t d[2] = {digestOf(f, key[0], 2 * sl), digestOf(f, key[1], 2 * sl + 1)};
            if (b == 5) pairSort<32>(d, lane);
            else pairSort<16>(d, lane);
            WAVE_TRACE(trSort)
            pinNext(keyN[0], idxN[0]);
            pinNext(keyN[1], idxN[1]);
            // equal leading bits among neighbours of the new order?  (by segment: ballot masked with the segment's lanes)
            const uint32_t dFront  = uint32_t(__shfl_up(int(d[1]), 1));
            const bool clash0      = sl != 0 && d[0] != ~0u && dFront != ~0u && (d[0] >> 8) == (dFront >> 8);
            const bool clash1      = d[1] != ~0u && d[0] != ~0u && (d[1] >> 8) == (d[0] >> 8);
            const uint64_t clashes = __ballot(clash0 || clash1);
            const unsigned base    = lane - sl; // first lane of my segment
            const uint64_t segment = ((b == 5 ? 0xFFFFFFFFull : 0xFFFFull)) << base;
            const bool exact       = (clashes & segment) != 0;
            uint16_t* place        = sPlace[wave] + 2u * base; // 2 * SEG entries per segment
            if (d[0] != ~0u) place[d[0] & 0xFFu] = uint16_t(2 * sl);
            if (d[1] != ~0u) place[d[1] & 0xFFu] = uint16_t(2 * sl + 1);
            __builtin_amdgcn_wave_barrier();
            stepFields(); // (behind the network: the values live in registers from here to the stores only)
            if (!exact)
            {
#pragma unroll
                for (int r = 0; r < 2; ++r)
                {
                    if (key[r] == HOLE) continue;
                    const uint32_t at = place[2 * sl + r];
                    if (at < f.nNew) storeAt(f.ok + at, key[r], idx[r], val[r]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (exact) exactLeaf(f, 2, key, idx, false);
            WAVE_TRACE(trFinish)
        }
#ifdef CSTONE_RESORT_TRACE
        if (g_resortTrace && threadIdx.x == 0)
        {
            uint64_t* tr = g_resortTrace + size_t(blockIdx.x) * RESORT_TRACE_SLOTS;
            tr[2] = wall_clock64(), tr[3] = uint64_t(trSteps), tr[4] = uint64_t(trIssue), tr[5] = uint64_t(trWait),
            tr[6] = uint64_t(trSort), tr[7] = uint64_t(trFinish);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tr[8] = wall_clock64();
        }
#endif
#undef WAVE_TRACE
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
                                                                uint32_t* __restrict__ pos,
                                                                uint32_t* __restrict__ boundaryLeaf)
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
    const bool onBoundary = leafLo[lo] == key;
    // (for radiiOfLeaves: the compact leaf the boundary falls into, top bit: it IS that leaf's first key)
    if (boundaryLeaf) boundaryLeaf[i] = lo | (onBoundary ? 0x80000000u : 0u);
    uint32_t a = layoutNew[lo], len = onBoundary ? 0u : layoutNew[lo + 1] - a;
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

/*! Halo radii of the leaves of any tree from the per-leaf maxima of h the field-carrying leaf pass folded (hmax, by
 *  compact leaf of the OLD tree).  A leaf whose two boundaries are first keys of compact leaves a and b holds exactly the
 *  new content of the compact leaves [a, b): its maximum is theirs.  Any other leaf (a split old leaf, a leaf that starts
 *  inside the key range of an old one) scans its own particles.  radii = float(max * 2 * ext), 0 for
 *  an empty leaf: cstone_hip_halo_radii's rule (Halos::discover, R/halos/halos.hpp:128-160). */
template<class T>
__global__ __launch_bounds__(256) void radiiFromOldLeavesKernel(const uint32_t* __restrict__ boundaryLeaf, NodeIdx numNodes,
                                                                const uint32_t* __restrict__ layout,
                                                                const T* __restrict__ hmax, const T* __restrict__ hSorted,
                                                                float ext, float* __restrict__ radii)
{
    // one lane per leaf: nearly every leaf of the new tree IS an old leaf (one value to fetch); the others loop
    const NodeIdx i = NodeIdx(blockIdx.x) * 256 + NodeIdx(threadIdx.x);
    if (i >= numNodes) return;
    const uint32_t ba = boundaryLeaf[i], bb = boundaryLeaf[i + 1];
    T m = 0;
    if ((ba & bb & 0x80000000u) != 0)
    {
        for (uint32_t j = ba & 0x7FFFFFFFu; j < (bb & 0x7FFFFFFFu); ++j)
            m = fmax(m, hmax[j]);
    }
    else
    {
        for (uint32_t p = layout[i]; p < layout[i + 1]; ++p)
            m = fmax(m, hSorted[p]);
    }
    radii[i] = float(m * 2 * ext);
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
                           K* keysOut, bool expectMovers, int fieldBits)
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
    if (fieldBits) CS_TRY(hmax_.ensure(ctx, ent * size_t(fieldBits / 8)));
    carryFields_   = fieldBits != 0;
    boundaryNodes_ = -1;

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
    // tiles in which something moved: one wave per leaf (the default), or counting over the whole leaf (the round-2
    // formulation, kept for comparison: CSTONE_RESORT_SCAN=1).  With at least one mover per tile on average hardly a
    // tile is quiet: the wave kernel then takes all tiles and the launch for quiet tiles is left out (at 10^8 particles
    // 0.56 / 0.72 ms instead of 0.59 / 0.77 ms; CSTONE_RESORT_WAVE_ALL=0/1 forces either)
    static const bool scanLeaves  = std::getenv("CSTONE_RESORT_SCAN") != nullptr;
    static const char* waveAllEnv = std::getenv("CSTONE_RESORT_WAVE_ALL");
    const bool waveAll = !scanLeaves && !alwaysCount && (waveAllEnv ? waveAllEnv[0] == '1' : numMovers >= grid);
    // leaves that are less than half full on average: two or four of them per wave step, two elements per lane (the
    // kernel's flavour 1, pairSort; with fuller leaves it saves a few per cent when everything moves and loses more when
    // most leaves are still in order -- a step then sorts two leaves if either needs it; CSTONE_RESORT_PAIRS=0/1 forces)
    const char* pairsEnv        = std::getenv("CSTONE_RESORT_PAIRS"); // (read at every launch: the tests switch it)
    // (... and with fuller leaves when more than 3 % of the particles changed their leaf: hardly a leaf is still in order then)
    const bool pairs = pairsEnv ? pairsEnv[0] == '1' : (n_ < size_t(J) * 32u || size_t(numMovers) * 32u > n_);
#define CSTONE_LEAF_WAVE_MODE(G, MODE)                                                                                 \
    hipLaunchKernelGGL((leafSortWaveKernel<K, G, MODE, NoLeafFields>), grid, 256, 0, ctx->stream, keysIn,              \
                       leafLo_.as<K>(), leafPos_.as<uint32_t>(), inOffset_.as<uint32_t>(), layoutNew_.as<uint32_t>(),  \
                       binKeys_.as<K>(), binIdx_.as<uint32_t>(), J, alwaysCount, waveAll, keysOut, orderOut,           \
                       NoLeafFields{})
#define CSTONE_LEAF_WAVE(G)                                                                                            \
    if (pairs) CSTONE_LEAF_WAVE_MODE(G, 1);                                                                            \
    else CSTONE_LEAF_WAVE_MODE(G, 0)
    if (leavesPerTile == 64)
    {
        if (!waveAll) CSTONE_LEAF_SORT(64, false);
        if (someMoved && scanLeaves) CSTONE_LEAF_SORT(64, true);
        else if (someMoved || waveAll) { CSTONE_LEAF_WAVE(64); }
    }
    else if (leavesPerTile == 32)
    {
        if (!waveAll) CSTONE_LEAF_SORT(32, false);
        if (someMoved && scanLeaves) CSTONE_LEAF_SORT(32, true);
        else if (someMoved || waveAll) { CSTONE_LEAF_WAVE(32); }
    }
    else if (leavesPerTile == 16)
    {
        if (!waveAll) CSTONE_LEAF_SORT(16, false);
        if (someMoved && scanLeaves) CSTONE_LEAF_SORT(16, true);
        else if (someMoved || waveAll) { CSTONE_LEAF_WAVE(16); }
    }
    else return fail(ctx, CSTONE_E_INTERNAL, "resort: %d leaves per workgroup not instantiated", leavesPerTile);
#undef CSTONE_LEAF_SORT
#undef CSTONE_LEAF_WAVE
#undef CSTONE_LEAF_WAVE_MODE
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

template<class K>
int LeafResort<K>::sortLeavesFields(cstone_hip_ctx* ctx, const K* keysIn, K* keysOut, uint32_t* orderOut,
                                    const ResortFields& fields, uint32_t numMovers, uint32_t numMarkers, uint32_t J,
                                    int leavesPerTile)
{
    if (fields.realBits != 32 && fields.realBits != 64) return fail(ctx, CSTONE_E_ARG, "resort: field width %d", fields.realBits);
    if (!carryFields_) return fail(ctx, CSTONE_E_INTERNAL, "resort: prepare() was not asked for the fields");
    auto run = [&](auto tTag) -> int
    {
        using T = decltype(tTag);
        if (numMovers)
        {
            StageTimer timer(ctx, CSTONE_STAGE_RESORT_BINS);
            hipLaunchKernelGGL(placeMoversKernel<K>, gridFor(numMovers, 256), 256, 0, ctx->stream, moverKeys_.as<K>(),
                               moverIdx_.as<uint32_t>(), moverDest_.as<uint32_t>(), moverSlot_.as<uint32_t>(), numMovers,
                               inOffset_.as<uint32_t>(), binKeys_.as<K>(), binIdx_.as<uint32_t>());
        }
        if (numMarkers)
        {
            // (particles that leave the domain: keys and old positions behind the last leaf; their fields stay behind)
            StageTimer timer(ctx, CSTONE_STAGE_RESORT_BINS);
            uint32_t* idx = binIdx_.as<uint32_t>() + (numMovers - numMarkers);
            CS_TRY(cstone_hip_sort_keys(ctx, 32, idx, numMarkers));
            hipLaunchKernelGGL(placeMarkersKernel<K>, gridFor(numMarkers, 256), 256, 0, ctx->stream, idx, numMarkers,
                               layoutNew_.as<uint32_t>(), (const uint32_t*)(ctx->devScalars + RESORT_SCALARS) + 2,
                               keysOut, orderOut);
        }
        if (J == 0) return CSTONE_OK;
        StageTimer timer(ctx, CSTONE_STAGE_RESORT_LEAVES);
        const unsigned grid = (J + unsigned(leavesPerTile) - 1) / unsigned(leavesPerTile);
        LeafFields<T> fl;
        fl.x = static_cast<const T*>(fields.in[0]), fl.y = static_cast<const T*>(fields.in[1]);
        fl.z = static_cast<const T*>(fields.in[2]), fl.h = static_cast<const T*>(fields.in[3]);
        fl.ox = static_cast<T*>(fields.out[0]), fl.oy = static_cast<T*>(fields.out[1]);
        fl.oz = static_cast<T*>(fields.out[2]), fl.oh = static_cast<T*>(fields.out[3]);
        fl.hmax = hmax_.as<T>();
        // the flavour by the fill of the leaves, as in sortLeaves(); every tile goes through this kernel
        const char* pairsEnv = std::getenv("CSTONE_RESORT_PAIRS");
        const bool pairs     = pairsEnv ? pairsEnv[0] == '1' : (n_ < size_t(J) * 32u || size_t(numMovers) * 32u > n_);
#define CSTONE_LEAF_FIELDS_MODE(G, MODE)                                                                               \
    hipLaunchKernelGGL((leafSortWaveKernel<K, G, MODE, LeafFields<T>>), grid, 256, 0, ctx->stream, keysIn,             \
                       leafLo_.as<K>(), leafPos_.as<uint32_t>(), inOffset_.as<uint32_t>(), layoutNew_.as<uint32_t>(),  \
                       binKeys_.as<K>(), binIdx_.as<uint32_t>(), J, false, true, keysOut, orderOut, fl)
#define CSTONE_LEAF_FIELDS(G)                                                                                          \
    if (pairs) CSTONE_LEAF_FIELDS_MODE(G, 1);                                                                          \
    else CSTONE_LEAF_FIELDS_MODE(G, 0)
        if (leavesPerTile == 64) { CSTONE_LEAF_FIELDS(64); }
        else if (leavesPerTile == 32) { CSTONE_LEAF_FIELDS(32); }
        else if (leavesPerTile == 16) { CSTONE_LEAF_FIELDS(16); }
        else return fail(ctx, CSTONE_E_INTERNAL, "resort: %d leaves per workgroup not instantiated", leavesPerTile);
#undef CSTONE_LEAF_FIELDS
#undef CSTONE_LEAF_FIELDS_MODE
        CS_HIP(ctx, hipGetLastError());
        return CSTONE_OK;
    };
    return fields.realBits == 32 ? run(float{}) : run(double{});
}

template<class K>
int LeafResort<K>::radiiOfLeaves(cstone_hip_ctx* ctx, int numNodes, const uint32_t* layout, const void* hSorted,
                                 int realBits, float ext, float* radii)
{
    if (numNodes <= 0) return CSTONE_OK;
    if (boundaryNodes_ != numNodes || !carryFields_)
        return fail(ctx, CSTONE_E_INTERNAL, "resort: radiiOfLeaves without countLeaves for this tree");
    StageTimer timer(ctx, CSTONE_STAGE_HALOS);
    if (realBits == 32)
        hipLaunchKernelGGL(radiiFromOldLeavesKernel<float>, gridFor(size_t(numNodes), 256), 256, 0, ctx->stream,
                           boundaryLeaf_.as<uint32_t>(), NodeIdx(numNodes), layout, hmax_.as<float>(),
                           static_cast<const float*>(hSorted), ext, radii);
    else
        hipLaunchKernelGGL(radiiFromOldLeavesKernel<double>, gridFor(size_t(numNodes), 256), 256, 0, ctx->stream,
                           boundaryLeaf_.as<uint32_t>(), NodeIdx(numNodes), layout, hmax_.as<double>(),
                           static_cast<const double*>(hSorted), ext, radii);
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
    // (the field-carrying pass wants to know where every boundary fell: radiiOfLeaves)
    uint32_t* boundaryLeaf = nullptr;
    if (carryFields_)
    {
        CS_TRY(boundaryLeaf_.ensure(ctx, size_t(numNodes + 1) * sizeof(uint32_t)));
        boundaryLeaf   = boundaryLeaf_.as<uint32_t>();
        boundaryNodes_ = numNodes;
    }
    hipLaunchKernelGGL(bracketedPositionsKernel<K>, gridFor(size_t(numNodes) + 1, 256), 256, 0, ctx->stream, tree,
                       NodeIdx(numNodes), keys, leafLo_.as<K>(), (const uint32_t*)(ctx->devScalars + RESORT_SCALARS) + 2,
                       haveCoarse_ ? coarse_.as<uint32_t>() : nullptr, layoutNew_.as<uint32_t>(), pos, boundaryLeaf);
    hipLaunchKernelGGL(countsOfPositionsKernel, gridFor(numNodes, 256), 256, 0, ctx->stream, pos, NodeIdx(numNodes),
                       maxCount, counts);
    arenaReset(ctx);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

template class LeafResort<uint32_t>;
template class LeafResort<uint64_t>;

#ifdef CSTONE_RESORT_TRACE
//! tuning builds: device buffer of RESORT_TRACE_SLOTS stamps per workgroup (nullptr: off)
extern "C" int cstone_hip_resort_trace_set(void* buffer)
{
    uint64_t* p = static_cast<uint64_t*>(buffer);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_resortTrace), &p, sizeof p) == hipSuccess ? 0 : 1;
}
#endif

} // namespace cship
