// Incremental re-sort of Domain::sync ("leaf re-sort"): the SFC order of a sync, built from the order of the sync before.
//
// The reference sorts all particle keys from scratch in every sync (sortByKeyGpu, R/primitives/primitives_gpu.cu:305-353,
// called from R/sfc/sfc_sorter.hpp).  Between two syncs of a time-stepping code most particles stay inside the leaf cell
// of the focus tree they were in: their positions in the arrays still tell their leaf.  The re-sort uses that:
//   1. the encode pass classifies every particle against the key range of the leaf its POSITION belongs to (previous
//      layout): a "stayer" keeps its leaf, a "mover" is appended to a list (resort.hpp: ResortArgs, sfc.hip:
//      encodeResortKernel)
//   2. the movers are binned by the leaf their new key falls into (binary search, one atomic each)
//   3. one pass over the leaves (leafSortKernel): the stayers of a leaf plus its incoming movers are brought into the
//      order of (key, old index) -- tiles in which nothing moved by one lane per leaf sorting in LDS, the others by
//      counting on 32-bit digests -- and written to the leaf's new place
// The result is the stable sort of the keys (ties by old index), whatever the particles did: a particle is a stayer
// only if its key lies in the range of the leaf its position claims, everything else goes through the bins, and the
// leaf ranges are disjoint and ascending.  Too many movers or an overfull leaf make the caller fall back to the radix sort.
// HBM traffic per particle: encode (3T + 2K) + leaf pass (K read, K + 4 written) instead of encode + 4..8 digit passes
// of 2K + 8 each.
#pragma once

#include "ctx.hpp"
#include "devbuf.hpp"

namespace cship
{

//! what the classifying encode kernel needs (device pointers)
template<class K>
struct ResortArgs
{
    K* keysOut;                // new keys at the particles' old positions; ~0 (a hole) where the particle left its leaf
    const uint64_t* leafStart; // bit p set: a non-empty leaf of the previous sync starts at position p
    const uint32_t* leafRank;  // per 64 positions: number of set bits in front of the word
    const K* leafLo;           // [J + 1]: first key of the j-th non-empty leaf; leafLo[0] = 0, leafLo[J] = endKey
    uint32_t* outCount;        // [J + 1]: particles that left leaf j
    K* moverKeys;              // [moverCap]
    uint32_t* moverIdx;        // [moverCap] old position of the mover
    uint32_t* moverCount;      // movers found (may exceed moverCap: then the list is incomplete)
    uint32_t moverCap;
};

//! the particle arrays a field-carrying leaf pass moves together with the keys: x, y, z, h of the old order (device, n
//! elements of real_bits / 8 bytes) and where the new order goes; hmaxOut / radii: see LeafResort::sortLeavesFields
struct ResortFields
{
    int realBits;
    const void* in[4];
    void* out[4];
};

//! limits of the leaf pass
constexpr uint32_t RESORT_TILE_SLOTS = 4224; // slots (old positions + arrivals) of a workgroup's leaves: key + index in
                                             // LDS for quiet tiles (three workgroups per CU), digests for the others
constexpr uint32_t RESORT_QUIET_SLOTS = 3840; // a tile in which nothing moved is sorted in LDS if it has at most this many
                                              // slots: key + 16-bit slot number, 39 KB, four workgroups per CU
constexpr int RESORT_COARSE_BITS      = 20;  // leading key bits of the movers' search table (binMoversKernel)
constexpr uint32_t RESORT_LEAF_CAP   = 256;  // a leaf must hold FEWER slots than this (the slot number is the low byte of a
                                             // digest, and slot 255 is left to the hole's digest ~0u)

//! device-side results a re-sort attempt reports (ctx->devScalars + RESORT_SCALARS, read back with the box extents)
constexpr int RESORT_SCALARS = 28; // [0] particles with the remove marker, [1] flags (1: leaf too long, 2: tile too
                                   // long, 4: mover list overflow: the attempt fails; 8: a quiet tile beyond
                                   // RESORT_QUIET_SLOTS: both leaf-pass launches needed), [2] J (non-empty leaves), [3] movers

template<class K>
class LeafResort
{
public:
    //! leaves per workgroup of the leaf pass for a focus bucket size; 0: buckets this large are not re-sorted
    static int leavesPerTile(uint32_t bucketFocus)
    {
        return bucketFocus <= 64 ? 64 : bucketFocus <= 128 ? 32 : bucketFocus < RESORT_LEAF_CAP ? 16 : 0;
    }

    /*! compact leaf table of the previous sync (non-empty leaves of `tree` with their first positions) and cleared
     *  counters; afterwards args() is valid.  layout[numLeaves] must equal n.  expectMovers: also a table from the
     *  leading key bits to the leaves, which shortens the search of every mover for its new leaf (worth its 20 us from
     *  some 10^5 movers on) */
    int prepare(cstone_hip_ctx* ctx, const K* tree, const uint32_t* layout, int numLeaves, size_t n, K* keysOut,
                bool expectMovers = false, int fieldBits = 0 /* 32 | 64: the leaf pass will carry x, y, z, h */);
    ResortArgs<K> args() const { return args_; }
    /*! bins the movers, new leaf sizes and offsets, limit checks; everything stays on the device: the three scalars at
     *  ctx->devScalars + RESORT_SCALARS tell the host how it went */
    int binMovers(cstone_hip_ctx* ctx, int leavesPerTile);
    /*! the leaf pass: keysIn = new keys at old positions; keysOut / orderOut = sorted keys and their old positions.
     *  numMovers, numMarkers, numCompactLeaves: the values read back after binMovers */
    int sortLeaves(cstone_hip_ctx* ctx, const K* keysIn, K* keysOut, uint32_t* orderOut, uint32_t numMovers,
                   uint32_t numMarkers, uint32_t numCompactLeaves, int leavesPerTile, bool largeQuietTiles);

    /*! The leaf pass that also MOVES x, y, z, h (one pass over the particle arrays instead of the leaf pass plus a gather
     *  per array): lane s of a leaf's wave loads key and fields of slot s, the wave orders the leaf in registers, every
     *  lane stores key, old index and fields at the leaf's new place.  A mover's fields are fetched from its old position
     *  (known from its bin one wave step before they are needed).  On the way the wave folds the
     *  maximum of h over the leaf's new content (hmax per compact leaf): radiiOfLeaves() turns them into the halo radii
     *  of whatever tree the sync ends up with.  Requires prepare(..., fieldBits). */
    int sortLeavesFields(cstone_hip_ctx* ctx, const K* keysIn, K* keysOut, uint32_t* orderOut, const ResortFields& fields,
                         uint32_t numMovers, uint32_t numMarkers, uint32_t numCompactLeaves, int leavesPerTile);
    /*! radii[i] = float(2 * ext * max h of the particles of leaf i of `tree`) (Halos::discover's rule, as
     *  cstone_hip_halo_radii) from the maxima sortLeavesFields folded: a leaf whose boundaries both are boundaries of
     *  the old (compact) leaf table takes the maximum of the old leaves it is made of, any other leaf scans its
     *  particles in hSorted.  Must follow countLeaves() for the same tree (which notes where every boundary falls);
     *  layout: offsets of the leaves among the sorted particles. */
    int radiiOfLeaves(cstone_hip_ctx* ctx, int numNodes, const uint32_t* layout, const void* hSorted, int realBits,
                      float ext, float* radii);

    /*! computeNodeCounts for any cornerstone leaf array over the keys the last sortLeaves ordered (valid until the next
     *  prepare): every boundary is searched inside the one old leaf that holds its key.  counts[i] = min(#keys in
     *  [tree[i], tree[i + 1]), maxCount) */
    int countLeaves(cstone_hip_ctx* ctx, const K* tree, int numNodes, const K* keys, uint32_t maxCount, uint32_t* counts);

private:
    DevBuf mask_, rank_, popc_, leafLo_, leafPos_, outCount_, incoming_, newCount_, layoutNew_, inOffset_;
    DevBuf moverKeys_, moverIdx_, moverDest_, moverSlot_, binKeys_, binIdx_, coarse_;
    DevBuf hmax_, boundaryLeaf_; // the field-carrying pass: max h per compact leaf, where the new tree's boundaries fall
    bool carryFields_  = false;
    int boundaryNodes_ = -1;     // tree size boundaryLeaf_ was filled for
    bool haveCoarse_ = false;
    ResortArgs<K> args_{};
    int numLeaves_ = 0;
    size_t n_      = 0;
};

//! encode with the box, write the keys to ra.keysOut and classify them (sfc.hip); *done = false: arrays not aligned
//! for the vector kernel, nothing was launched
int computeKeysResort(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* x, const void* y,
                      const void* z, const void* keysIn, size_t n, const cstone_box& box, const void* resortArgs,
                      void* extentsOut, bool* done);

} // namespace cship
