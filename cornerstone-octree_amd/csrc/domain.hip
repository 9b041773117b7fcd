// Device-resident orchestration of cstone::Domain::sync for ONE rank (R/domain/domain.hpp:196-243 and the pieces
// it drives: GlobalAssignment R/domain/assignment.hpp:39-205, FocusedOctree R/focus/octree_focus_mpi.hpp,
// CombinedUpdate::updateFocus R/focus/octree_focus.hpp:83-137, Halos R/halos/halos.hpp:128-222, layout
// R/domain/layout.hpp:150-165).  Everything stays in HBM; per sync the host reads back the bounding box (6 scalars),
// the number of valid particles and, per tree rebalance step, {changed flag, new leaf count}.
//
// Multi-rank operation (peers, MAC-driven focus resolution, treelet / particle / halo exchange over RCCL) is the next
// row to build; the object refuses num_ranks > 1 instead of silently running something else.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <vector>

#include "ctx.hpp"
#include "devbuf.hpp"
#include "device_keys.hpp"
#include "host_tree.hpp"
#include "resort.hpp"
#include "scan.hpp"

namespace cship
{

namespace
{

// ---- focus-tree rebalance decisions for all nodes (leaves and internal), R/focus/rebalance.hpp:50-88.
//      Single rank: the focus is the whole key range, so MAC flags never decide anything (inFringe / inFocus are
//      always true) and the decision reduces to counts.
template<class K>
__global__ __launch_bounds__(256) void focusOpsKernel(const K* __restrict__ prefixes,
                                                      const NodeIdx* __restrict__ childOffsets,
                                                      const NodeIdx* __restrict__ parents,
                                                      const uint32_t* __restrict__ counts, NodeIdx numNodes,
                                                      uint32_t bucket, NodeIdx* __restrict__ ops)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numNodes) return;
    int op = 1;
    if (i > 0 && counts[parents[(i - 1) / 8]] <= bucket) { op = 0; }
    else
    {
        unsigned level = prefixBits(prefixes[i]) / 3;
        if (childOffsets[i] == 0 && level < maxLevel<K>() && counts[i] > bucket) op = 8;
    }
    ops[i] = op;
}

// ---- R/focus/rebalance.hpp:113-184: a merged node inherits the op of its closest kept ancestor iff both start at
//      the same key; reads the ORIGINAL ops and writes a second array (the reference rewrites in place, race-free in
//      outcome).  changed |= 1 if any op != 1.
template<class K>
__global__ __launch_bounds__(256) void protectAncestorsKernel(const K* __restrict__ prefixes,
                                                              const NodeIdx* __restrict__ parents,
                                                              const NodeIdx* __restrict__ opsIn, NodeIdx numNodes,
                                                              NodeIdx* __restrict__ opsOut, int* __restrict__ changed)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    int op    = 1;
    if (i < numNodes)
    {
        if (i == 0) { op = opsIn[0]; }
        else
        {
            NodeIdx a = i;
            while (opsIn[a] == 0)
                a = parents[(a - 1) / 8];
            op = (fromPrefix(prefixes[i]) == fromPrefix(prefixes[a])) ? opsIn[a] : 0;
        }
        opsOut[i] = op;
    }
    // (set once: updates of tens of thousands of waves to one address serialise in the L2, reads of it do not)
    if (__any(op != 1) && (threadIdx.x & 63) == 0 && *reinterpret_cast<volatile int*>(changed) == 0) atomicOr(changed, 1);
}

//! leafOps[i] = ops[leafToInternal[numInternal + i]] for i < numLeaves, leafOps[numLeaves] = 0
__global__ __launch_bounds__(256) void leafOpsKernel(const NodeIdx* __restrict__ ops,
                                                     const NodeIdx* __restrict__ leafToInternal, NodeIdx numInternal,
                                                     NodeIdx numLeaves, uint32_t* __restrict__ leafOps)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i < numLeaves) leafOps[i] = uint32_t(ops[leafToInternal[numInternal + i]]);
    else if (i == numLeaves) leafOps[i] = 0;
}

//! counts[leafToInternal[numInternal + i]] = leafCounts[i]
__global__ __launch_bounds__(256) void scatterLeafCountsKernel(const uint32_t* __restrict__ leafCounts,
                                                               const NodeIdx* __restrict__ leafToInternal,
                                                               NodeIdx numInternal, NodeIdx numLeaves,
                                                               uint32_t* __restrict__ counts)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i < numLeaves) counts[leafToInternal[numInternal + i]] = leafCounts[i];
}

template<class K>
__global__ void countValidKernel(const K* __restrict__ keys, size_t n, int* __restrict__ out)
{
    // number of keys below the end of the curve = index of the first remove marker (keys are sorted)
    size_t lo = 0, len = n;
    while (len > 0)
    {
        size_t half = len >> 1;
        bool right  = keys[lo + half] < endKey<K>();
        lo          = right ? lo + half + 1 : lo;
        len         = right ? len - half - 1 : half;
    }
    out[0] = int(lo);
}

template<class K>
__global__ void initTreeKernel(K* tree, uint32_t* counts, uint32_t c0)
{
    tree[0]   = 0;
    tree[1]   = endKey<K>();
    counts[0] = c0;
}

} // namespace

struct DomainBase
{
    virtual ~DomainBase() = default;
    virtual int sync(void** keys, void** x, void** y, void** z, void** h, size_t n, void** scratch, int numScratch,
                     void** props, const int* propBytes, int numProps) = 0;
    virtual int view(cstone_hip_domain_view* out)        = 0;
    virtual void setHaloFactor(float factor)             = 0;
    virtual void setSortMode(int mode)                   = 0;
    virtual void setSpeculativeBox(bool on)              = 0;
    virtual int reapplySync(const void* in, size_t n, int elemBytes, void* out) = 0;
    virtual int updateExpansionCenters(const void* x, const void* y, const void* z, const void* m, int massBits) = 0;
    virtual void stats(cstone_hip_domain_stats* out)     = 0;
};

template<class K, class T>
class DomainImpl final : public DomainBase
{
public:
    DomainImpl(cstone_hip_ctx* ctx, int curve, uint32_t bucket, uint32_t bucketFocus, float theta, const cstone_box& box)
        : ctx_(ctx)
        , curve_(curve)
        , bucket_(bucket)
        , bucketFocus_(bucketFocus)
        , theta_(theta)
        , box_(box)
    {
    }
    ~DomainImpl() override
    {
        if (pinnedLevels_) (void)hipHostFree(pinnedLevels_);
    }

    int sync(void** keysPP, void** xPP, void** yPP, void** zPP, void** hPP, size_t n, void** scratchAll, int numScratch,
             void** props, const int* propBytes, int numProps) override
    {
        void** scratchPP = scratchAll; // the first scratch buffer: the one the single-scratch rotation works with
        if (n == 0) return fail(ctx_, CSTONE_E_ARG, "domain_sync: no particles");
        if (!firstCall_ && n != bufSize_)
            return fail(ctx_, CSTONE_E_ARG, "Domain sync: input array sizes are inconsistent (%zu != %u)", n, bufSize_);
        K* keys = static_cast<K*>(*keysPP);
        const int kb = 8 * sizeof(K), rb = 8 * sizeof(T);
        // the level ranges of the linked octree the last sync (re)built travelled to a pinned block behind its last
        // kernels; that sync ended with a synchronisation of the stream, so they are here now
        if (levelsPending_)
        {
            CS_HIP(ctx_, hipStreamSynchronize(ctx_->stream)); // (idle unless the last sync failed half way)
            levelRangeHost_.assign(pinnedLevels_, pinnedLevels_ + maxLevel<K>() + 2);
            int deepest = 0;
            for (int l = 0; l <= int(maxLevel<K>()); ++l)
                if (levelRangeHost_[l + 1] > levelRangeHost_[l]) deepest = l;
            if (deepest > deepestBound_)
                return fail(ctx_, CSTONE_E_INTERNAL, "domain_sync: the focus tree has level %d, the bound was %d", deepest,
                            deepestBound_);
            deepestBound_  = deepest;
            levelsPending_ = false;
        }

        // ---- GlobalAssignment::assign (assignment.hpp:57-103): box, keys, sort, one global-tree step
        // The box of a sync follows from the extents of x, y, z (makeGlobalBox + limitBoxShrinking).  After the first call
        // it rarely changes, so the keys are computed SPECULATIVELY with the box of the previous sync while the same
        // pass over x, y, z measures the extents; the box rule is evaluated afterwards and only a box that really
        // changed costs a second encode + sort.  Saves the separate 2.4 GB pass over the coordinates per sync at 1e8.
        auto boxFromExtents = [&](const double* lim, cstone_box& box)
        {
            if (firstCall_) { std::copy(lim, lim + 6, box.lim); }
            else
            {
                // limitBoxShrinking (sfc/box.hpp:415-431), evaluated in T like the reference
                const T shrink = T(0.05);
                for (int d = 0; d < 3; ++d)
                {
                    T lo = T(box.lim[2 * d]), hi = T(box.lim[2 * d + 1]);
                    T len = hi - lo;
                    T a = lo + shrink * len, b = hi - shrink * len;
                    box.lim[2 * d]     = std::min(T(lim[2 * d]), a);
                    box.lim[2 * d + 1] = std::max(T(lim[2 * d + 1]), b);
                }
            }
        };
        const bool anyOpen = box_.bc[0] != 1 || box_.bc[1] != 1 || box_.bc[2] != 1;
        // (adaptive: a sync whose speculation failed -- the outermost particles of an open box move -- makes the following
        //  syncs measure first, until one of them finds the box unchanged again)
        bool speculate     = anyOpen && !firstCall_ && !measureFirst_ && maySpeculate();
        bool boxMoved      = false; // measured first and found a new box: every key changes, nothing to re-sort from
        if (anyOpen && !speculate)
        {
            const cstone_box before = box_;
            double lim[6];
            // extents of the open dimensions in one launch and one read-back (MinMaxGpu x3 in the reference)
            const void* open[3];
            int dims[3], numOpen = 0;
            void* coords[3] = {*xPP, *yPP, *zPP};
            for (int d = 0; d < 3; ++d)
            {
                if (box_.bc[d] == 1) { lim[2 * d] = box_.lim[2 * d], lim[2 * d + 1] = box_.lim[2 * d + 1]; }
                else { open[numOpen] = coords[d], dims[numOpen++] = d; }
            }
            double ext[6];
            CS_TRY(minMaxCoordinates(ctx_, rb, open, numOpen, n, ext));
            for (int i = 0; i < numOpen; ++i)
                lim[2 * dims[i]] = ext[2 * i], lim[2 * dims[i] + 1] = ext[2 * i + 1];
            boxFromExtents(lim, box_);
            for (int k = 0; k < 6; ++k)
                boxMoved = boxMoved || box_.lim[k] != before.lim[k];
            if (!firstCall_) measureFirst_ = boxMoved;
        }
        // computeSfcKeys + setMapFromCodes (assignment.hpp:81-86) in one call: the sort's digits are counted while the keys
        // are still in the encode kernel's registers, its first pass produces the positions instead of reading an iota
        CS_TRY(order_.ensure(ctx_, n * sizeof(uint32_t)));
        CS_TRY(orderAlt_.ensure(ctx_, n * sizeof(uint32_t)));
        CS_TRY(keysAlt_.ensure(ctx_, n * sizeof(K)));
        size_t tb = cstone_hip_sort_pairs_temp_bytes(kb, n);
        CS_TRY(sortTmp_.ensure(ctx_, tb));
        // Which digits need the radix passes: two particles whose keys agree in the digits above the deepest leaf level
        // (+1) of the previous focus tree sit in the same leaf cell, i.e. such runs hold at most a bucket of particles
        // and are ordered by a fix-up pass instead (sort.hip, fixupRunsKernel).  If a run turns out longer (the
        // particles have clustered since), the flag read back below triggers a regular sort of the rest.
        int startPass = 0;
        if (!firstCall_ && !levelRangeHost_.empty() && !allDigits())
        {
            int lmax = 0;
            for (int l = 0; l <= int(maxLevel<K>()); ++l)
                if (levelRangeHost_[l + 1] > levelRangeHost_[l]) lmax = l;
            // one level below the deepest leaves a run holds about bucket / 8 particles; larger buckets get more margin
            // (the fix-up handles runs of up to 192)
            int margin  = 1 + (bucketFocus_ > 128) + (bucketFocus_ > 1024);
            int lowBits = 3 * int(maxLevel<K>()) - 3 * (lmax + margin);
            startPass   = std::max(0, lowBits / 8) & ~1;
        }
        T* extentsDev  = reinterpret_cast<T*>(ctx_->devScalars + 16);
        T* extentsHost = reinterpret_cast<T*>(ctx_->hostScalars + 16);
        auto encodeAndSort = [&](bool measure, bool* measured)
        {
            CS_TRY(sfcKeysAndOrderingHint(ctx_, curve_, kb, rb, *xPP, *yPP, *zPP, keys, order_.as<uint32_t>(), n, box_,
                                          keysAlt_.p, orderAlt_.as<uint32_t>(), sortTmp_.p, tb, startPass,
                                          ctx_->devScalars + 3, true, measure ? extentsDev : nullptr, measured));
            if (startPass == 0) CS_HIP(ctx_, hipMemsetAsync(ctx_->devScalars + 3, 0, sizeof(int), ctx_->stream));
            return CSTONE_OK;
        };
        // the box the extents of this sync ask for (makeGlobalBox + limitBoxShrinking); true: it differs from box_
        auto nextBox = [&](const double* ext, cstone_box& next)
        {
            double lim[6];
            for (int d = 0; d < 3; ++d)
            {
                const bool pbc = box_.bc[d] == 1;
                lim[2 * d]     = pbc ? box_.lim[2 * d] : ext[2 * d];
                lim[2 * d + 1] = pbc ? box_.lim[2 * d + 1] : ext[2 * d + 1];
            }
            next = box_;
            boxFromExtents(lim, next);
            bool changed = false;
            for (int k = 0; k < 6; ++k)
                changed = changed || next.lim[k] != box_.lim[k];
            return changed;
        };

        // ---- the incremental re-sort (resort.hpp): particles that are still inside the leaf their position belonged to
        //      at the previous sync are ordered leaf by leaf, the others are binned into their new leaves.  Same result
        //      as the sort of all keys; a box that changed, too many movers or an overfull leaf take the regular path.
        const int tileLeaves = LeafResort<K>::leavesPerTile(bucketFocus_);
        bool sorted          = false;
        uint32_t resortMarkers = 0; // particles with the remove marker, as the re-sort counted them
        const bool tryResort = !firstCall_ && tileLeaves > 0 && layoutLeaves_ == fLeaves_ && fLeaves_ > 0 &&
                               resortBackoff_ == 0 && !boxMoved && mayResort();
        if (resortBackoff_ > 0) --resortBackoff_;
        // The field-carrying leaf pass (resort.hpp, sortLeavesFields; CSTONE_FUSED_LEAF_PASS=1 and four scratch arrays): x,
        // y, z, h move with the keys in ONE pass over the particle arrays instead of the leaf pass and the gathers of h and
        // of x, y, z.  Measured at 1e8 particles, every particle drifting: 2.14 ms against 0.66 + 0.37 + 0.91 ms for the
        // three passes it replaces (it saves 8 of their 92 bytes per particle, and the ordering network -- bound by
        // instruction issue -- and the data movement do not overlap inside one wave the way separate kernels on two streams
        // do): kept selectable, not the default.
        const bool fusedLeafPass        = std::getenv("CSTONE_FUSED_LEAF_PASS") != nullptr; // (read per sync: tests switch it)
        const bool carryFields          = numScratch >= 4 && fusedLeafPass;
        bool fieldsMoved                = false; // x, y, z, h are in their new order already (and radii follow from hmax)
        // The gather of x, y, z has no consumer inside a sync: with three or more scratch arrays it runs on the context's
        // second stream, next to the tree updates (chains of small, latency-bound kernels with two read-backs in between)
        // and the gather of h, and is joined before the call returns.
        const bool noOverlap = std::getenv("CSTONE_NO_GATHER_OVERLAP") != nullptr; // tuning / tests
        bool xyzForked = false, xyzJoined = false;
        ctx_->auxBusy  = false; // (a sync that failed behind its fork may have left it set)
        auto forkXyzGather = [&](size_t count) -> int
        {
            if (fieldsMoved || numScratch < 3 || noOverlap || xyzForked || count == 0) return CSTONE_OK;
            CS_TRY(ensureAuxStream(ctx_));
            CS_HIP(ctx_, hipEventRecord(ctx_->evFork, ctx_->stream));
            CS_HIP(ctx_, hipStreamWaitEvent(ctx_->aux, ctx_->evFork, 0));
            const void* src[3] = {*xPP, *yPP, *zPP};
            void* dst[3]       = {scratchAll[0], scratchAll[1], scratchAll[2]};
            int rc;
            {
                StreamScope scope(ctx_, ctx_->aux);
                rc = cstone_hip_gather_multi(ctx_, sizeof(T), order_.as<uint32_t>(), count, src, dst, 3);
            }
            CS_TRY(rc);
            CS_HIP(ctx_, hipEventRecord(ctx_->evJoin, ctx_->aux));
            ctx_->auxBusy = true;
            std::swap(*xPP, scratchAll[0]);
            std::swap(*yPP, scratchAll[1]);
            std::swap(*zPP, scratchAll[2]);
            xyzForked = true;
            return CSTONE_OK;
        };
        auto joinXyzGather = [&]() -> int
        {
            if (xyzForked && !xyzJoined) CS_HIP(ctx_, hipStreamWaitEvent(ctx_->stream, ctx_->evJoin, 0));
            xyzJoined     = true;
            ctx_->auxBusy = false;
            return CSTONE_OK;
        };
        if (tryResort)
        {
            CS_TRY(resort_.prepare(ctx_, fTree_.as<K>(), layout_.as<uint32_t>(), fLeaves_, n, keysAlt_.as<K>(), true,
                                   carryFields ? rb : 0));
            const ResortArgs<K> ra = resort_.args();
            bool done              = false;
            CS_TRY(computeKeysResort(ctx_, curve_, kb, rb, *xPP, *yPP, *zPP, keys, n, box_, &ra,
                                     speculate ? extentsDev : nullptr, &done));
            if (done)
            {
                CS_TRY(resort_.binMovers(ctx_, tileLeaves));
                // one read-back: the extents (slots 16..27) and what the re-sort found (28..31)
                CS_TRY(copyToPinned(ctx_, ctx_->hostScalars + 16, ctx_->devScalars + 16, 16 * sizeof(int)));
                CS_HIP(ctx_, hipStreamSynchronize(ctx_->stream));
                bool boxChanged = false;
                cstone_box next = box_;
                if (speculate)
                {
                    double ext[6];
                    for (int k = 0; k < 6; ++k)
                        ext[k] = double(extentsHost[k]);
                    boxChanged = nextBox(ext, next);
                }
                const uint32_t markers = uint32_t(ctx_->hostScalars[RESORT_SCALARS]);
                resortMarkers          = markers;
                const int flags        = ctx_->hostScalars[RESORT_SCALARS + 1];
                const uint32_t J       = uint32_t(ctx_->hostScalars[RESORT_SCALARS + 2]);
                const uint32_t movers  = uint32_t(ctx_->hostScalars[RESORT_SCALARS + 3]);
                if (!boxChanged && (flags & 7) == 0 && movers <= n / 8)
                {
                    if (carryFields)
                    {
                        ResortFields rf{rb, {*xPP, *yPP, *zPP, *hPP}, {scratchAll[0], scratchAll[1], scratchAll[2], scratchAll[3]}};
                        CS_TRY(resort_.sortLeavesFields(ctx_, keysAlt_.as<K>(), keys, order_.as<uint32_t>(), rf, movers,
                                                        markers, J, tileLeaves));
                        std::swap(*xPP, scratchAll[0]);
                        std::swap(*yPP, scratchAll[1]);
                        std::swap(*zPP, scratchAll[2]);
                        std::swap(*hPP, scratchAll[3]);
                        fieldsMoved = true;
                    }
                    else
                    {
                        CS_TRY(resort_.sortLeaves(ctx_, keysAlt_.as<K>(), keys, order_.as<uint32_t>(), movers, markers, J,
                                                  tileLeaves, (flags & 8) != 0));
                        CS_TRY(forkXyzGather(n - markers)); // (the ordering is final: x, y, z follow on the second stream)
                    }
                    CS_HIP(ctx_, hipMemsetAsync(ctx_->devScalars + 3, 0, sizeof(int), ctx_->stream));
                    sorted      = true;
                    lastMovers_ = movers;
                    ++resorts_;
                }
                else
                {
                    // the caller's key array has not been touched: the regular path starts from scratch, with the box
                    // this sync needs (the extents are known by now)
                    if (boxChanged)
                    {
                        box_ = next;
                        ++boxRedos_;
                        measureFirst_ = true;
                    }
                    else
                    {
                        resortBackoff_ = 4;
                        ++resortFallbacks_;
                    }
                    speculate = false;
                }
            }
        }

        resortedThisSync_ = sorted;
        bool measured     = false;
        if (!sorted) CS_TRY(encodeAndSort(speculate, &measured));
        if (!sorted && speculate)
        {
            double lim[6];
            if (measured)
            {
                CS_TRY(copyToPinned(ctx_, extentsHost, extentsDev, 6 * sizeof(T)));
                CS_HIP(ctx_, hipStreamSynchronize(ctx_->stream));
                for (int k = 0; k < 6; ++k)
                    lim[k] = double(extentsHost[k]);
            }
            else
            {
                // unaligned arrays: the plain encode ran, measure the extents the regular way
                const void* all[3] = {*xPP, *yPP, *zPP};
                CS_TRY(minMaxCoordinates(ctx_, rb, all, 3, n, lim));
            }
            cstone_box next = box_;
            if (nextBox(lim, next))
            {
                // the particles have left the box (or shrunk away from it by more than 5 %): keys and order again
                box_ = next;
                ++boxRedos_;
                measureFirst_ = true;
                // the key array is sorted by now: put every entry back to its particle first (the remove markers of the
                // caller must meet their own particles again), then encode with the new box
                CS_TRY(cstone_hip_scatter(ctx_, sizeof(K), order_.as<uint32_t>(), n, keys, keysAlt_.p));
                CS_HIP(ctx_, hipMemcpyAsync(keys, keysAlt_.p, n * sizeof(K), hipMemcpyDeviceToDevice, ctx_->stream));
                CS_TRY(encodeAndSort(false, nullptr));
            }
        }

        if (firstCall_)
        {
            // spanning tree of one rank = the root; counts = bucket - 1 (assignment.hpp:48-53)
            CS_TRY(ensureTree(gTree_, gCounts_, gCap_, 4096));
            hipLaunchKernelGGL(initTreeKernel<K>, 1, 1, 0, ctx_->stream, gTree_.as<K>(), gCounts_.as<uint32_t>(),
                               bucket_ - 1);
            gLeaves_ = 1;
        }
        // A sync that was re-sorted knows its number of valid particles already, and the host holds the (small) global tree
        // and its counts from the last sync: the update step of the global tree is then made on the HOST
        // (globalTreeStepHost), the device only counts, and the counts travel back with the read-back of the focus-tree
        // update -- one stream synchronisation and a handful of launches fewer than cstone_hip_update_octree.
        uint32_t numAssigned = 0;
        int converged        = 0;
        const bool hostStep  = sorted && !firstCall_ && hostGlobalStep_ && int(gLeavesHost_.size()) == gLeaves_ + 1 &&
                              int(gCountsHost_.size()) == gLeaves_;
        if (hostStep)
        {
            numAssigned = uint32_t(n) - resortMarkers;
            if (numAssigned == 0) return fail(ctx_, CSTONE_E_ARG, "domain_sync: all particles removed");
            std::vector<K> fresh;
            const bool same = globalTreeStepHost<K>(gLeavesHost_, gCountsHost_, bucket_, fresh);
            if (!same)
            {
                const int leaves = int(fresh.size()) - 1;
                CS_TRY(ensureTree(gTree_, gCounts_, gCap_, leaves + 1));
                gLeavesHost_.swap(fresh);
                CS_TRY(cstone_hip_upload(ctx_, gTree_.p, gLeavesHost_.data(), gLeavesHost_.size() * sizeof(K)));
                gLeaves_ = leaves;
            }
            converged = same;
            CS_TRY(cstone_hip_compute_node_counts(ctx_, kb, gTree_.p, gCounts_.as<uint32_t>(), gLeaves_, keys, n,
                                                  0xFFFFFFFFu));
            CS_TRY(queueGlobalReadBack(false));
        }
        else
        {
        // particles flagged with the remove marker sort behind the end of the curve and leave the domain: their number
        // is on its way to the host while the global tree is updated (whose own read-back completes the stream)
        hipLaunchKernelGGL(countValidKernel<K>, 1, 1, 0, ctx_->stream, keys, n, ctx_->devScalars + 2);
        CS_TRY(copyToPinned(ctx_, ctx_->hostScalars + 2, ctx_->devScalars + 2, 2 * sizeof(int)));
        CS_TRY(updateGlobal(keys, n, &converged));
        if (firstCall_)
        {
            // `while (!updateOctreeGlobal(...));` (assignment.hpp:95-98): at least one more step, whatever the first said
            int guard = 0;
            do
            {
                CS_TRY(updateGlobal(keys, n, &converged));
                if (++guard > 64) return fail(ctx_, CSTONE_E_INTERNAL, "global tree does not converge");
            } while (!converged);
        }
        // the copy was queued ahead of the update's own read-back (update_octree synchronises for the new leaf count)
        numAssigned = uint32_t(ctx_->hostScalars[2]);
        if (numAssigned == 0) return fail(ctx_, CSTONE_E_ARG, "domain_sync: all particles removed");
        if (ctx_->hostScalars[3] != 0)
        {
            // a run of equal high digits was too long for the fix-up: sort the rest the regular way.  (The global tree
            // above is not affected: its leaf boundaries cannot fall inside such a run.)
            CS_TRY(cstone_hip_sort_pairs(ctx_, kb, keys, order_.as<uint32_t>(), n, keysAlt_.p, orderAlt_.as<uint32_t>(),
                                         sortTmp_.p, tb));
            ++fullSortFallbacks_;
        }
        // (tree and counts for the host's copy: they arrive behind the next synchronisation of the stream)
        CS_TRY(queueGlobalReadBack(true));
        }
        CS_TRY(forkXyzGather(numAssigned)); // (radix path: the ordering is final from here on)

        // ---- GlobalAssignment::distribute on one rank: nothing to exchange; the second sort (assignment.hpp:156) of an
        //      already sorted range is the identity and is skipped.
        // ---- h goes into SFC order for the halo radii (domain.hpp:213-215): gathered further down, in the pass that
        //      takes the maximum per leaf

        // ---- focus tree (octree_focus_mpi.hpp:535-553 on first call, then :225-227)
        if (firstCall_)
        {
            CS_TRY(ensureTree(fTree_, fLeafCounts_, fCap_, 4096));
            hipLaunchKernelGGL(initTreeKernel<K>, 1, 1, 0, ctx_->stream, fTree_.as<K>(), fLeafCounts_.as<uint32_t>(),
                               bucketFocus_ + 1);
            fLeaves_ = 1;
            CS_TRY(buildFocusOctree());
            // counts_ of the single root node = bucketFocus + 1 (octree_focus_mpi.hpp:75)
            CS_TRY(fCounts_.ensure(ctx_, sizeof(uint32_t)));
            CS_HIP(ctx_, hipMemcpyAsync(fCounts_.p, fLeafCounts_.p, sizeof(uint32_t), hipMemcpyDeviceToDevice,
                                        ctx_->stream));
            int conv = 0, guard = 0;
            while (!conv)
            {
                CS_TRY(updateFocus(keys, numAssigned, &conv));
                if (++guard > 64) return fail(ctx_, CSTONE_E_INTERNAL, "focus tree does not converge");
            }
        }
        int conv = 0;
        CS_TRY(updateFocus(keys, numAssigned, &conv));
        takeGlobalReadBack(); // (updateFocus has synchronised the stream: the global counts are here)

        // ---- Halos::discover + computeLayout (halos.hpp:128-222) for the assignment [0, L)
        const NodeIdx L = fLeaves_;
        CS_TRY(layout_.ensure(ctx_, size_t(L + 1) * sizeof(uint32_t)));
        CS_TRY(radii_.ensure(ctx_, size_t(L) * sizeof(float)));
        CS_TRY(flags_.ensure(ctx_, size_t(L) * sizeof(int)));
        CS_HIP(ctx_, hipMemsetAsync(layout_.p, 0, sizeof(uint32_t), ctx_->stream));
        CS_TRY(cstone_hip_inclusive_scan_u32(ctx_, fLeafCounts_.as<uint32_t>(), layout_.as<uint32_t>() + 1, size_t(L)));
        layoutLeaves_ = L;
        // gatherArrays(h) + segmentMax + scale in one pass over h (every leaf is assigned on one rank: the leaves' particles
        // are all the assigned particles)
        static const bool splitGather = std::getenv("CSTONE_SPLIT_H_GATHER") != nullptr; // tuning: the two-pass form
        if (fieldsMoved)
        {
            // h is in SFC order already; the radii follow from the maxima the leaf pass folded per OLD leaf
            CS_TRY(resort_.radiiOfLeaves(ctx_, L, layout_.as<uint32_t>(), *hPP, rb, haloSearchExt_, radii_.as<float>()));
        }
        else
        {
            // where h goes: the first scratch array -- unless x, y, z are still on their way there on the second stream; a
            // fourth scratch array then takes h (no waiting), with three the gather of h waits for them
            void** hDst = scratchPP;
            if (xyzForked && numScratch >= 4) { hDst = &scratchAll[3]; }
            else { CS_TRY(joinXyzGather()); }
            if (splitGather)
            {
                CS_TRY(cstone_hip_gather(ctx_, sizeof(T), order_.as<uint32_t>(), numAssigned, *hPP, *hDst));
                std::swap(*hPP, *hDst);
                CS_TRY(cstone_hip_halo_radii(ctx_, rb, *hPP, layout_.as<uint32_t>(), 0, L, L, haloSearchExt_,
                                             radii_.as<float>()));
            }
            else
            {
                CS_TRY(gatherWithHaloRadii(ctx_, rb, *hPP, order_.as<uint32_t>(), *hDst, layout_.as<uint32_t>(), L,
                                           haloSearchExt_, radii_.as<float>()));
                std::swap(*hPP, *hDst);
            }
        }
        CS_HIP(ctx_, hipMemsetAsync(flags_.p, 0, size_t(L) * sizeof(int), ctx_->stream));
        CS_TRY(cstone_hip_find_halos(ctx_, curve_, kb, rb, fPrefixes_.p, fChild_.as<int32_t>(), fItl_.as<int32_t>(),
                                     fTree_.p, radii_.as<float>(), &box_, 0, L, flags_.as<int32_t>()));
        // one rank: every leaf is assigned, no halo leaves: layout = exclusive scan of the leaf counts (layout.hpp:150-165),
        // which is the inclusive scan written above shifted by one.

        // ---- updateLayout (domain.hpp:542-604): keys already sit at offset 0; gather the unordered arrays
        CS_TRY(joinXyzGather()); // (the property gathers below go through the buffers x, y, z were read from)
        if (fieldsMoved || xyzForked) {} // (x, y, z went with the leaf pass, or on the second stream)
        else if (numScratch >= 3)
        {
            // three free buffers (the caller's scratch tuple, R/domain/domain.hpp:196-206 has three as well): x, y, z go
            // to their new order in ONE pass that reads the ordering once
            const void* src[3] = {*xPP, *yPP, *zPP};
            void* dst[3]       = {scratchAll[0], scratchAll[1], scratchAll[2]};
            CS_TRY(cstone_hip_gather_multi(ctx_, sizeof(T), order_.as<uint32_t>(), numAssigned, src, dst, 3));
            std::swap(*xPP, scratchAll[0]);
            std::swap(*yPP, scratchAll[1]);
            std::swap(*zPP, scratchAll[2]);
        }
        else
        {
            void** arrays[3] = {xPP, yPP, zPP};
            for (auto a : arrays)
            {
                CS_TRY(cstone_hip_gather(ctx_, sizeof(T), order_.as<uint32_t>(), numAssigned, *a, *scratchPP));
                std::swap(*a, *scratchPP);
            }
        }
        for (int p = 0; p < numProps; ++p)
        {
            if (propBytes[p] > int(sizeof(T)))
                return fail(ctx_, CSTONE_E_ARG, "domain_sync: property %d is wider than the scratch element", p);
            CS_TRY(cstone_hip_gather(ctx_, propBytes[p], order_.as<uint32_t>(), numAssigned, props[p], *scratchPP));
            std::swap(props[p], *scratchPP);
        }

        startIndex_ = 0;
        endIndex_   = numAssigned;
        lastN_      = n;
        haveExpansion_ = false; // (the tree and the particles' order may have changed)
        bufSize_    = numAssigned;
        ++syncs_;
        firstCall_  = false;
        // the sticky device-side error word (look-back spin bail-out of the sort, traversal stack overflow ...): a sync
        // that tripped one of those checks must not report success (tests: CSTONE_FORCE_DEVICE_ERROR raises it)
#ifdef CSTONE_TEST_HOOKS
        if (std::getenv("CSTONE_FORCE_DEVICE_ERROR")) CS_HIP(ctx_, hipMemsetAsync(ctx_->devScalars + 63, 1, 1, ctx_->stream));
#endif
        return cstone_hip_ctx_sync(ctx_);
    }

    void setHaloFactor(float factor) override { haloSearchExt_ = factor; }
    void setSortMode(int mode) override { sortMode_ = mode; }
    void setSpeculativeBox(bool on) override { speculativeBox_ = on; }
    // what the client chose (cstone_hip_domain_set_sort_mode / _set_speculative_box); the environment variables of the
    // experiments override it
    bool mayResort() const
    {
        return sortMode_ == CSTONE_SORT_INCREMENTAL && std::getenv("CSTONE_NO_RESORT") == nullptr &&
               std::getenv("CSTONE_FULL_SORT") == nullptr;
    }
    bool allDigits() const { return sortMode_ == CSTONE_SORT_ALL_DIGITS || std::getenv("CSTONE_FULL_SORT") != nullptr; }
    bool maySpeculate() const { return speculativeBox_ && std::getenv("CSTONE_NO_SPECULATIVE_BOX") == nullptr; }

    int view(cstone_hip_domain_view* v) override
    {
        v->start_index           = startIndex_;
        v->end_index             = endIndex_;
        v->num_particles_with_halos = bufSize_;
        v->box                   = box_;
        v->num_global_leaves     = gLeaves_;
        v->global_leaves         = gTree_.p;
        v->global_counts         = gCounts_.as<uint32_t>();
        v->num_focus_leaves      = fLeaves_;
        v->num_focus_nodes       = fLeaves_ + (fLeaves_ - 1) / 7;
        v->focus_leaves          = fTree_.p;
        v->focus_leaf_counts     = fLeafCounts_.as<uint32_t>();
        v->prefixes              = fPrefixes_.p;
        v->child_offsets         = fChild_.as<int32_t>();
        v->parents               = fParents_.as<int32_t>();
        v->level_range           = fLevelRange_.as<int32_t>();
        v->internal_to_leaf      = fItl_.as<int32_t>();
        v->leaf_to_internal      = fLti_.as<int32_t>();
        v->layout                = layout_.as<uint32_t>();
        v->centers               = fCenters_.p;
        v->sizes                 = fSizes_.p;
        v->halo_flags            = flags_.as<int32_t>();
        v->sfc_order             = order_.as<uint32_t>();
        v->halo_radii            = radii_.as<float>();
        v->expansion_centers     = haveExpansion_ ? fExpansion_.p : nullptr;
        return CSTONE_OK;
    }

    /*! Domain::updateExpansionCenters (R/domain/domain.hpp:415-421) on one rank = FocusedOctree::updateCenters without its
     *  exchanges (octree_focus_mpi.hpp:369-449: computeLeafSourceCenter, upsweep with CombineSourceCenter) + setMac for
     *  1 / theta: (centre of mass, MAC radius^2) per node of the focus tree.  x, y, z, m: the arrays of the last sync's
     *  results (all of them are assigned on one rank), device */
    int updateExpansionCenters(const void* x, const void* y, const void* z, const void* m, int massBits) override
    {
        if (lastN_ == 0 || fLeaves_ < 1) return fail(ctx_, CSTONE_E_ARG, "update_expansion_centers: no sync yet");
        if ((massBits != 32 && massBits != 64) || !x || !y || !z || !m)
            return fail(ctx_, CSTONE_E_ARG, "update_expansion_centers: bad argument");
        const NodeIdx L = fLeaves_, I = (L - 1) / 7, M = L + I;
        CS_TRY(fExpansion_.ensure(ctx_, size_t(M) * 4 * sizeof(T)));
        CS_TRY(cstone_hip_leaf_source_centers(ctx_, 8 * sizeof(T), massBits, 8 * sizeof(T), x, y, z, m,
                                              fLti_.as<int32_t>() + I, L, layout_.as<uint32_t>(), fExpansion_.p));
        int32_t levels[maxLevel<K>() + 2];
        CS_TRY(copyToHost(ctx_, levels, fLevelRange_.p, sizeof levels)); // (which levels exist: synchronises the stream)
        CS_TRY(cstone_hip_upsweep_centers(ctx_, 8 * sizeof(T), int(maxLevel<K>()), levels, fChild_.as<int32_t>(),
                                          fExpansion_.p));
        CS_TRY(cstone_hip_set_mac(ctx_, curve_, 8 * sizeof(K), 8 * sizeof(T), fPrefixes_.p, M, fExpansion_.p, 1.0f / theta_,
                                  &box_));
        haveExpansion_ = true;
        return cstone_hip_ctx_sync(ctx_);
    }

    /*! Domain::reapplySync (R/domain/domain.hpp:334-378) without an exchange: the kept particles in SFC order */
    int reapplySync(const void* in, size_t n, int elemBytes, void* out) override
    {
        if (lastN_ == 0) return fail(ctx_, CSTONE_E_ARG, "reapply_sync: no sync yet");
        if (n != lastN_)
            return fail(ctx_, CSTONE_E_ARG, "reapply_sync: array of %zu elements, the last sync took %zu", n, lastN_);
        if (!in || !out || in == out) return fail(ctx_, CSTONE_E_ARG, "reapply_sync: bad array");
        return cstone_hip_gather(ctx_, elemBytes, order_.as<uint32_t>(), endIndex_, in, out);
    }

    void stats(cstone_hip_domain_stats* out) override
    {
        out->syncs               = syncs_;
        out->resorts             = uint32_t(resorts_);
        out->resort_fallbacks    = uint32_t(resortFallbacks_);
        out->box_redos           = uint32_t(boxRedos_);
        out->full_sort_fallbacks = uint32_t(fullSortFallbacks_);
        out->last_movers         = lastMovers_;
    }

    float haloSearchExt_ = 1.0f;

private:
    uint32_t syncs_ = 0, lastMovers_ = 0;
    size_t lastN_ = 0; // input size of the last sync
    int boxRedos_ = 0; // syncs whose speculative keys had to be recomputed because the box changed
    bool measureFirst_ = false; // the last box was not the one before it: measure the extents before encoding
    int sortMode_        = CSTONE_SORT_INCREMENTAL;
    bool speculativeBox_ = true;
    int ensureTree(DevBuf& tree, DevBuf& counts, int& cap, int need)
    {
        if (need <= cap) return CSTONE_OK;
        int newCap = std::max(need, int(cap * 1.5));
        CS_TRY(tree.ensure(ctx_, size_t(newCap + 1) * sizeof(K), true));
        CS_TRY(counts.ensure(ctx_, size_t(newCap) * sizeof(uint32_t), true));
        cap = newCap;
        return CSTONE_OK;
    }

    //! global counts (and, when the device made it, the leaf array) on their way to the pinned block; not waited for
    int queueGlobalReadBack(bool withLeaves)
    {
        CS_TRY(pin_.reserve(ctx_, size_t(gLeaves_) * 4 + size_t(gLeaves_ + 1) * sizeof(K) + 256));
        pinCounts_ = static_cast<uint32_t*>(pin_.take(size_t(gLeaves_) * 4));
        CS_TRY(copyToPinned(ctx_, pinCounts_, gCounts_.p, size_t(gLeaves_) * 4));
        pinLeaves_ = nullptr;
        if (withLeaves)
        {
            pinLeaves_ = static_cast<K*>(pin_.take(size_t(gLeaves_ + 1) * sizeof(K)));
            CS_TRY(copyToPinned(ctx_, pinLeaves_, gTree_.p, size_t(gLeaves_ + 1) * sizeof(K)));
        }
        pinLeafCount_ = gLeaves_;
        return CSTONE_OK;
    }
    void takeGlobalReadBack()
    {
        if (!pinCounts_) return;
        gCountsHost_.assign(pinCounts_, pinCounts_ + pinLeafCount_);
        if (pinLeaves_) gLeavesHost_.assign(pinLeaves_, pinLeaves_ + pinLeafCount_ + 1);
        pinCounts_ = nullptr, pinLeaves_ = nullptr;
    }

    //! updateOctreeGlobal on one rank (tree/update_mpi.hpp:71-94): rebalance with the previous counts, then recount
    int updateGlobal(const K* keys, size_t n, int* converged)
    {
        while (true)
        {
            int leaves = gLeaves_;
            int rc = cstone_hip_update_octree(ctx_, 8 * sizeof(K), keys, n, bucket_, gTree_.p, gCounts_.as<uint32_t>(),
                                              &leaves, gCap_, 0xFFFFFFFFu, converged);
            if (rc == CSTONE_E_CAPACITY)
            {
                CS_TRY(ensureTree(gTree_, gCounts_, gCap_, leaves + 1));
                continue;
            }
            CS_TRY(rc);
            gLeaves_ = leaves;
            return CSTONE_OK;
        }
    }

    int buildFocusOctree()
    {
        const NodeIdx L = fLeaves_, M = L + (L - 1) / 7;
        CS_TRY(fPrefixes_.ensure(ctx_, size_t(M) * sizeof(K)));
        CS_TRY(fChild_.ensure(ctx_, size_t(M + 1) * sizeof(NodeIdx)));
        CS_TRY(fParents_.ensure(ctx_, size_t(std::max(1, (M - 1) / 8)) * sizeof(NodeIdx)));
        CS_TRY(fLevelRange_.ensure(ctx_, (maxLevel<K>() + 2) * sizeof(NodeIdx)));
        CS_TRY(fItl_.ensure(ctx_, size_t(M) * sizeof(NodeIdx)));
        CS_TRY(fLti_.ensure(ctx_, size_t(M) * sizeof(NodeIdx)));
        // a focus update refines by one level at most: the leaves of the new tree are no deeper than the deepest of the
        // tree before + 1.  deepestBound_ is exact at the start of a sync (above) and grows by one per rebuild inside it.
        const int deepest = std::min(int(maxLevel<K>()), deepestBound_ + 1);
        CS_TRY(buildLinkedOctree(ctx_, 8 * sizeof(K), fTree_.p, L, fPrefixes_.p, fChild_.as<int32_t>(),
                                 fParents_.as<int32_t>(), fLevelRange_.as<int32_t>(), fItl_.as<int32_t>(),
                                 fLti_.as<int32_t>(), deepest));
        deepestBound_ = deepest;
        // the exact level ranges go to a pinned block WITHOUT waiting for them: the upsweep of this sync launches the
        // levels up to the bound (one of them may be empty), the next sync reads the block
        if (!pinnedLevels_)
            CS_HIP(ctx_, hipHostMalloc(reinterpret_cast<void**>(&pinnedLevels_), 32 * sizeof(NodeIdx), hipHostMallocDefault));
        CS_TRY(copyToPinned(ctx_, pinnedLevels_, fLevelRange_.p, (maxLevel<K>() + 2) * sizeof(NodeIdx)));
        levelsPending_ = true;
        return CSTONE_OK;
    }

    /*! one FocusedOctree::updateTree + updateCounts + updateGeoCenters step on one rank
     *  (octree_focus_mpi.hpp:108-187,205-273; octree_focus.hpp:83-137; rebalance.hpp:50-184) */
    int updateFocus(const K* keys, size_t n, int* converged)
    {
        const NodeIdx L = fLeaves_, I = (L - 1) / 7, M = L + I;
        CS_TRY(ops_.ensure(ctx_, size_t(M + 1) * sizeof(NodeIdx)));
        CS_TRY(ops2_.ensure(ctx_, size_t(M + 1) * sizeof(NodeIdx)));
        CS_TRY(leafOps_.ensure(ctx_, size_t(L + 1) * sizeof(uint32_t)));
        int* scalars = ctx_->devScalars;
        CS_HIP(ctx_, hipMemsetAsync(scalars, 0, 2 * sizeof(int), ctx_->stream));
        hipLaunchKernelGGL(focusOpsKernel<K>, gridFor(M, 256), 256, 0, ctx_->stream, fPrefixes_.as<K>(),
                           fChild_.as<NodeIdx>(), fParents_.as<NodeIdx>(), fCounts_.as<uint32_t>(), M, bucketFocus_,
                           ops_.as<NodeIdx>());
        // enforceKeys({focusStart, focusEnd}) is a no-op for the keys 0 and 2^(3 maxLevel) (rebalance.hpp:205)
        hipLaunchKernelGGL(protectAncestorsKernel<K>, gridFor(M, 256), 256, 0, ctx_->stream, fPrefixes_.as<K>(),
                           fParents_.as<NodeIdx>(), ops_.as<NodeIdx>(), M, ops2_.as<NodeIdx>(), scalars);
        hipLaunchKernelGGL(leafOpsKernel, gridFor(L + 1, 256), 256, 0, ctx_->stream, ops2_.as<NodeIdx>(),
                           fLti_.as<NodeIdx>(), I, L, leafOps_.as<uint32_t>());
        CS_TRY(arenaReserve(ctx_, scanArenaBytes(size_t(L) + 1)));
        int rc = scanU32(ctx_, leafOps_.as<uint32_t>(), leafOps_.as<uint32_t>(), size_t(L) + 1, 0u, false,
                         (uint32_t*)scalars + 1);
        arenaReset(ctx_);
        CS_TRY(rc);
        CS_TRY(copyToPinned(ctx_, ctx_->hostScalars, scalars, 2 * sizeof(int)));
        CS_HIP(ctx_, hipStreamSynchronize(ctx_->stream));
        *converged           = ctx_->hostScalars[0] == 0;
        const NodeIdx newL   = ctx_->hostScalars[1];
        if (newL < 1) return fail(ctx_, CSTONE_E_INTERNAL, "focus rebalance produced %d leaves", newL);

        if (!*converged)
        {
            CS_TRY(newTree_.ensure(ctx_, size_t(newL + 1) * sizeof(K)));
            CS_TRY(cstone_hip_rebalance_tree(ctx_, 8 * sizeof(K), fTree_.p, L, newL, leafOps_.as<int32_t>(),
                                             newTree_.p));
            // the new leaf array BECOMES the tree (the two buffers change places; the counts are made anew below).  It
            // used to be copied back: 18 MB at 10^8 particles, a blit with a handful of waves that got next to nothing
            // of the memory system while the gather of x, y, z ran on the second stream -- 0.62 ms during which the
            // rest of the tree update waited (profiles/r04: the copy ended with the gather)
            std::swap(fTree_.p, newTree_.p);
            std::swap(fTree_.bytes, newTree_.bytes);
            CS_TRY(fLeafCounts_.ensure(ctx_, size_t(newL) * sizeof(uint32_t)));
            fCap_    = int(std::min(fTree_.bytes / sizeof(K) - 1, fLeafCounts_.bytes / sizeof(uint32_t)));
            fLeaves_ = newL;
            CS_TRY(buildFocusOctree());
        }
        // else: every node op is "keep" -- the leaf array and with it the linked octree are what they were; the reference
        // rebuilds them regardless (octree_focus.hpp:121-134) because it parks its temporaries in the tree's own arrays

        // updateCounts: leaf counts from the particle keys, scattered to the linked layout, summed bottom-up
        const NodeIdx newI = (newL - 1) / 7, newM = newL + newI;
        // the leaf boundaries of an unchanged tree are searched from where they were at the previous sync (layout_)
        const uint32_t* guess = (*converged && layoutLeaves_ == newL) ? layout_.as<uint32_t>() : nullptr;
        // a sync that was re-sorted knows where the particles of every old leaf went: a boundary that an old leaf had
        // already is looked up, any other is searched inside one old leaf (resort.hip, countLeaves)
        if (resortedThisSync_)
            CS_TRY(resort_.countLeaves(ctx_, fTree_.as<K>(), newL, keys, 0xFFFFFFFFu, fLeafCounts_.as<uint32_t>()));
        else
            CS_TRY(cstone_hip_compute_node_counts_guided(ctx_, 8 * sizeof(K), fTree_.p, fLeafCounts_.as<uint32_t>(), newL,
                                                         keys, n, 0xFFFFFFFFu, guess));
        CS_TRY(fCounts_.ensure(ctx_, size_t(newM) * sizeof(uint32_t)));
        hipLaunchKernelGGL(scatterLeafCountsKernel, gridFor(newL, 256), 256, 0, ctx_->stream,
                           fLeafCounts_.as<uint32_t>(), fLti_.as<NodeIdx>(), newI, newL, fCounts_.as<uint32_t>());
        CS_TRY(cstone_hip_upsweep_sum_bounded(ctx_, int(maxLevel<K>()) + 2, fLevelRange_.as<int32_t>(),
                                              fChild_.as<int32_t>(), fCounts_.as<uint32_t>(), deepestBound_));
        // updateGeoCenters: the geometry of the nodes follows from the tree and the box alone -- nothing to do for an
        // unchanged tree inside an unchanged box (the reference recomputes it in every sync, octree_focus_mpi.hpp:259-273)
        bool sameBox = centersNodes_ == newM;
        for (int k = 0; k < 6; ++k)
            sameBox = sameBox && centersBox_.lim[k] == box_.lim[k];
        if (!(*converged && sameBox))
        {
            CS_TRY(fCenters_.ensure(ctx_, size_t(newM) * 3 * sizeof(T)));
            CS_TRY(fSizes_.ensure(ctx_, size_t(newM) * 3 * sizeof(T)));
            CS_TRY(cstone_hip_node_centers(ctx_, curve_, 8 * sizeof(K), 8 * sizeof(T), fPrefixes_.p, newM, &box_,
                                           fCenters_.p, fSizes_.p));
            centersBox_   = box_;
            centersNodes_ = newM;
        }
        CS_HIP(ctx_, hipGetLastError());
        return CSTONE_OK;
    }

    cstone_hip_ctx* ctx_;
    int curve_;
    uint32_t bucket_, bucketFocus_;
    float theta_;
    cstone_box box_;
    bool firstCall_ = true;
    uint32_t startIndex_ = 0, endIndex_ = 0, bufSize_ = 0;

    DevBuf order_, orderAlt_, keysAlt_, sortTmp_;
    LeafResort<K> resort_;
    int resortBackoff_ = 0, resorts_ = 0, resortFallbacks_ = 0;
    bool resortedThisSync_ = false;
    DevBuf gTree_, gCounts_;
    int gCap_ = 0, gLeaves_ = 0;
    std::vector<K> gLeavesHost_;        // host copies of the global tree and its counts (the host makes its update step)
    std::vector<uint32_t> gCountsHost_;
    PinnedBlock pin_;
    uint32_t* pinCounts_ = nullptr;
    K* pinLeaves_        = nullptr;
    int pinLeafCount_    = 0;
    bool hostGlobalStep_ = std::getenv("CSTONE_DEVICE_GLOBAL_STEP") == nullptr; // (tests: the device-side step)
    DevBuf fTree_, fLeafCounts_, fCounts_, newTree_;
    int fCap_ = 0, fLeaves_ = 0;
    int layoutLeaves_ = -1; // number of leaves layout_ was computed for
    cstone_box centersBox_{}; // box and node count fCenters_ / fSizes_ were computed for
    NodeIdx centersNodes_ = -1;
    std::vector<NodeIdx> levelRangeHost_; // of the tree the LAST sync left behind (exact)
    NodeIdx* pinnedLevels_ = nullptr;     // where the level ranges of a rebuilt tree arrive
    bool levelsPending_    = false;
    int deepestBound_      = int(maxLevel<K>()); // no leaf of the current focus tree is deeper
    int fullSortFallbacks_ = 0;
    DevBuf fPrefixes_, fChild_, fParents_, fLevelRange_, fItl_, fLti_, fCenters_, fSizes_;
    DevBuf fExpansion_;          // T[M][4]: centre of mass + MAC radius^2 per node (updateExpansionCenters)
    bool haveExpansion_ = false; // ... of the tree of the last sync
    DevBuf ops_, ops2_, leafOps_, layout_, radii_, flags_;
};

} // namespace cship

using namespace cship;

struct cstone_hip_domain
{
    cstone_hip_ctx* ctx;
    std::unique_ptr<DomainBase> impl;
};

extern "C"
{

int cstone_hip_domain_create(cstone_hip_ctx* ctx, cstone_hip_domain** out, int curve, int key_bits, int real_bits,
                             int rank, int num_ranks, uint32_t bucket_size, uint32_t bucket_size_focus, float theta,
                             const cstone_box* box_host)
{
    if (!ctx || !out || !box_host) return fail(ctx, CSTONE_E_ARG, "domain_create: bad argument");
    *out = nullptr;
    if (bucket_size < bucket_size_focus)
        return fail(ctx, CSTONE_E_ARG,
                    "The bucket size of the global tree must not be smaller than the bucket size of the focused tree");
    if (num_ranks != 1 || rank != 0)
        return fail(ctx, CSTONE_E_ARG, "domain_create: only single-rank domains are implemented in this round");
    if (curve != CSTONE_MORTON && curve != CSTONE_HILBERT) return fail(ctx, CSTONE_E_ARG, "domain_create: bad curve");
    auto* d = new cstone_hip_domain{ctx, nullptr};
    if (key_bits == 32 && real_bits == 32)
        d->impl.reset(new DomainImpl<uint32_t, float>(ctx, curve, bucket_size, bucket_size_focus, theta, *box_host));
    else if (key_bits == 32 && real_bits == 64)
        d->impl.reset(new DomainImpl<uint32_t, double>(ctx, curve, bucket_size, bucket_size_focus, theta, *box_host));
    else if (key_bits == 64 && real_bits == 32)
        d->impl.reset(new DomainImpl<uint64_t, float>(ctx, curve, bucket_size, bucket_size_focus, theta, *box_host));
    else if (key_bits == 64 && real_bits == 64)
        d->impl.reset(new DomainImpl<uint64_t, double>(ctx, curve, bucket_size, bucket_size_focus, theta, *box_host));
    else
    {
        delete d;
        return fail(ctx, CSTONE_E_ARG, "domain_create: unsupported key/real width");
    }
    *out = d;
    return CSTONE_OK;
}

int cstone_hip_domain_destroy(cstone_hip_domain* dom)
{
    if (!dom) return CSTONE_E_ARG;
    // (a domain outliving its context -- a client tearing down in the wrong order: the object and its device buffers are
    //  still released, the dead context is not touched, the caller learns about it)
    const bool alive = ctxAlive(dom->ctx);
    if (alive) (void)hipStreamSynchronize(dom->ctx->stream);
    else (void)hipDeviceSynchronize();
    delete dom;
    return alive ? CSTONE_OK : CSTONE_E_ARG;
}

int cstone_hip_domain_sync(cstone_hip_domain* dom, void** keys, void** x, void** y, void** z, void** h, size_t n,
                           void** scratch, void** props, const int* prop_bytes, int num_props)
{
    if (!dom || !keys || !x || !y || !z || !h || !scratch || !*keys || !*x || !*y || !*z || !*h || !*scratch ||
        num_props < 0 || (num_props && (!props || !prop_bytes)))
        return fail(dom ? dom->ctx : nullptr, CSTONE_E_ARG, "domain_sync: bad argument");
    return dom->impl->sync(keys, x, y, z, h, n, scratch, 1, props, prop_bytes, num_props);
}

int cstone_hip_domain_sync_scratch(cstone_hip_domain* dom, void** keys, void** x, void** y, void** z, void** h, size_t n,
                                   void** scratch, int num_scratch, void** props, const int* prop_bytes, int num_props)
{
    if (!dom || !keys || !x || !y || !z || !h || !scratch || !*keys || !*x || !*y || !*z || !*h || num_scratch < 1 ||
        num_props < 0 || (num_props && (!props || !prop_bytes)))
        return fail(dom ? dom->ctx : nullptr, CSTONE_E_ARG, "domain_sync: bad argument");
    for (int q = 0; q < num_scratch; ++q)
        if (!scratch[q]) return fail(dom->ctx, CSTONE_E_ARG, "domain_sync: null scratch buffer %d", q);
    return dom->impl->sync(keys, x, y, z, h, n, scratch, num_scratch, props, prop_bytes, num_props);
}

int cstone_hip_domain_update_expansion_centers(cstone_hip_domain* dom, const void* x, const void* y, const void* z,
                                               const void* m, int mass_bits)
{
    if (!dom) return CSTONE_E_ARG;
    return dom->impl->updateExpansionCenters(x, y, z, m, mass_bits);
}

int cstone_hip_domain_sync_grav(cstone_hip_domain* dom, void** keys, void** x, void** y, void** z, void** h, void** m,
                                int mass_bits, size_t n, void** scratch, int num_scratch, void** props,
                                const int* prop_bytes, int num_props)
{
    if (!dom || !m || !*m || (mass_bits != 32 && mass_bits != 64) || num_props < 0 || (num_props && (!props || !prop_bytes)))
        return fail(dom ? dom->ctx : nullptr, CSTONE_E_ARG, "domain_sync_grav: bad argument");
    // the masses follow their particles as one more property behind the caller's
    std::vector<void*> pp(props, props + num_props);
    std::vector<int> pb(prop_bytes, prop_bytes + num_props);
    pp.push_back(*m);
    pb.push_back(mass_bits / 8);
    int rc = cstone_hip_domain_sync_scratch(dom, keys, x, y, z, h, n, scratch, num_scratch, pp.data(), pb.data(),
                                            num_props + 1);
    for (int q = 0; q < num_props; ++q)
        props[q] = pp[q];
    *m = pp[num_props];
    if (rc != CSTONE_OK) return rc;
    return dom->impl->updateExpansionCenters(*x, *y, *z, *m, mass_bits);
}

int cstone_hip_domain_view_get(cstone_hip_domain* dom, cstone_hip_domain_view* out)
{
    if (!dom || !out) return CSTONE_E_ARG;
    return dom->impl->view(out);
}

int cstone_hip_domain_stats_get(cstone_hip_domain* dom, cstone_hip_domain_stats* out)
{
    if (!dom || !out) return CSTONE_E_ARG;
    dom->impl->stats(out);
    return CSTONE_OK;
}

int cstone_hip_domain_reapply_sync(cstone_hip_domain* dom, const void* in, size_t n, int elem_bytes, void* out)
{
    if (!dom) return CSTONE_E_ARG;
    return dom->impl->reapplySync(in, n, elem_bytes, out);
}

int cstone_hip_domain_set_halo_factor(cstone_hip_domain* dom, float factor)
{
    if (!dom || !(factor > 0.0f)) return CSTONE_E_ARG;
    dom->impl->setHaloFactor(factor);
    return CSTONE_OK;
}

int cstone_hip_domain_set_sort_mode(cstone_hip_domain* dom, int mode)
{
    if (!dom || mode < CSTONE_SORT_INCREMENTAL || mode > CSTONE_SORT_ALL_DIGITS) return CSTONE_E_ARG;
    dom->impl->setSortMode(mode);
    return CSTONE_OK;
}

int cstone_hip_domain_set_speculative_box(cstone_hip_domain* dom, int on)
{
    if (!dom) return CSTONE_E_ARG;
    dom->impl->setSpeculativeBox(on != 0);
    return CSTONE_OK;
}

int cstone_hip_test_hooks(void)
{
#ifdef CSTONE_TEST_HOOKS
    return 1;
#else
    return 0;
#endif
}

} // extern "C"
