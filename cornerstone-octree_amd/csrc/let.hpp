// The locally essential tree (LET) of one rank and its halo layout: the host state machine of the reference's
// FocusedOctree (R/focus/octree_focus_mpi.hpp) and Halos (R/halos/halos.hpp) on top of the C ABI of cstone_hip.h.
//
// PLAIN HOST C++20: this file includes no HIP header and launches nothing itself.  Every device operation is a call of
// the C ABI (device pointers + sizes), every exchange goes through cstone_hip_comm_ops (RCCL inside the library, or
// whatever the application provides).  What the reference does with MPI_Isend / MPI_Probe loops between peers is done
// here with a count pre-exchange (one all-gather of a row per rank) followed by ONE all-to-all-v per step: xGMI links
// are point to point, a grouped ncclSend/ncclRecv per peer is what an all-to-all-v is there.
//
//   reference step                                   R/ file:line                          here
//   findPeersMac                                     traversal/peers.hpp:63-118            findPeers()
//   FocusedOctree::updateMinMac / updateMacs         focus/octree_focus_mpi.hpp:457-531    updateMinMac()
//   FocusedOctree::updateTree                        :108-187                              updateTree()
//     focusTransfer                                  focus/exchange_focus.hpp:364-433      focusTransfer()
//     CombinedUpdate::updateFocus                    focus/octree_focus.hpp:83-136         updateFocus()
//     macRefine / updateMacRefine                    :218-279                              macRefine()
//     translateAssignment                            domain/domaindecomp.hpp:183-206       translateAssignment()
//     syncTreelets (exchange, check, reject, prune)  focus/exchange_focus.hpp:61-217       syncTreelets()
//     indexTreelets                                  :266-287                              indexTreelets()
//   FocusedOctree::updateCounts (+ peerExchange)     focus/octree_focus_mpi.hpp:205-286    updateCounts()
//   FocusedOctree::converge                          :535-553                              converge()
//   Halos::discover                                  halos/halos.hpp:128-189               discoverHalos()
//   Halos::computeLayout, exchangeRequestKeys        :205-222, domain/exchange_keys.hpp    computeLayout()
//   Halos::exchangeHalos                             :232-253                              exchangeHalos()
//   Domain::syncGrav (the focus-tree part)           domain/domain.hpp:266-318             updateGrav()
//     FocusedOctree::updateCenters                   focus/octree_focus_mpi.hpp:369-449    updateCenters()
//     globalFocusExchange (populate / gather / upsweep / extract)   :288-366,763-784       globalCenterExchange()
//     updateMacs, setMacRadius, addMacs              :479-526,601-610                      updateMacs(), addMacs()
//
// The tree stays on the device between the steps; the host sees a few scalars per step (new leaf counts, status words,
// per-peer counts).  The results -- leaf array, leaf counts, focus assignment, layout, halo flags and halo ranges -- are
// the reference's, bit for bit (tests/test_let.py: this file against the reference's own classes under MPI).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <map>
#include <vector>

#include "cstone_hip.h"

namespace cship
{

#define LET_TRY(expr)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        int rc_ = (expr);                                                                                              \
        if (rc_ != CSTONE_OK) return rc_;                                                                              \
    } while (0)

//! grow-only device buffer on the C ABI
struct LetBuf
{
    cstone_hip_ctx* ctx = nullptr;
    void* p             = nullptr;
    size_t bytes        = 0;

    LetBuf() = default;
    explicit LetBuf(cstone_hip_ctx* c)
        : ctx(c)
    {
    }
    LetBuf(const LetBuf&)            = delete;
    LetBuf& operator=(const LetBuf&) = delete;
    ~LetBuf()
    {
        if (p) (void)cstone_hip_free(ctx, p);
    }
    int ensure(size_t need, bool keep = false)
    {
        if (need <= bytes) return CSTONE_OK;
        size_t want = size_t(double(need) * 1.05) + 256;
        void* q     = nullptr;
        LET_TRY(cstone_hip_malloc(ctx, &q, want));
        if (p)
        {
            if (keep) LET_TRY(cstone_hip_memcpy_d2d(ctx, q, p, bytes));
            LET_TRY(cstone_hip_free(ctx, p)); // synchronises the stream first
        }
        p     = q;
        bytes = want;
        return CSTONE_OK;
    }
    void swap(LetBuf& o)
    {
        std::swap(p, o.p);
        std::swap(bytes, o.bytes);
    }
    template<class V>
    V* as() const
    {
        return static_cast<V*>(p);
    }
};

//! TreeIndexPair of the reference (R/domain/index_ranges.hpp:30-60)
struct LetRange
{
    int32_t start = 0, end = 0;
    int32_t count() const { return end - start; }
};

template<class K, class T>
class FocusLet
{
    static constexpr int kb = 8 * sizeof(K), rb = 8 * sizeof(T);
    static constexpr int maxLevel = sizeof(K) == 4 ? 10 : 21;
    static constexpr K endKey() { return K(1) << (3 * maxLevel); }

public:
    FocusLet(cstone_hip_ctx* ctx, int curve, int rank, int numRanks, uint32_t bucketFocus, float theta,
             const cstone_hip_comm_ops& comm)
        : ctx_(ctx)
        , curve_(curve)
        , rank_(rank)
        , P_(numRanks)
        , bucket_(bucketFocus)
        , theta_(theta)
        , comm_(comm)
        , assignment_(numRanks)
        , globAssignment_(numRanks + 1, K(0))
        , tlCount_(numRanks, 0)
        , tlOffset_(numRanks + 1, 0)
    {
        for (LetBuf* b : allBufs())
            b->ctx = ctx;
        box_.lim[0] = box_.lim[2] = box_.lim[4] = 0.0; // Box<T>{0, 1} (octree_focus_mpi.hpp:689)
        box_.lim[1] = box_.lim[3] = box_.lim[5] = 1.0;
        box_.bc[0] = box_.bc[1] = box_.bc[2] = 0;
        box_.pad_                            = 0;
    }

    /*! The focus-tree part of Domain::sync (R/domain/domain.hpp:217-237) followed by Halos::discover / computeLayout.
     *  keys: the rank's assigned particles, sorted (device); h: their smoothing lengths in the same order (device);
     *  assignment: numRanks + 1 keys (host); globalLeaves / globalCounts: the replicated global tree (device).
     *  externalFailure != 0: the caller has a failure of its own pending that its peers must learn about -- this rank
     *  goes through all the collectives, the status word of the last count exchange carries the failure, and every rank
     *  returns an error from there.  globalTreeSame: the caller knows that the global leaf array is the one of the last
     *  update (then an unchanged assignment and box keep the peers).  Collective: every rank calls it. */
    int update(const cstone_box& box, const K* keys, size_t numKeys, const K* assignment, const K* globalLeaves,
               const uint32_t* globalCounts, int numGlobalLeaves, const T* h, float haloSearchExt,
               int externalFailure = 0, bool globalTreeSame = false)
    {
        const float invThetaEff = 1.0f / theta_ + 0.5f; // invThetaMinMac, R/traversal/macs.hpp:44
        LET_TRY(init());
        LET_TRY(findPeers(assignment, globalLeaves, numGlobalLeaves, box, invThetaEff, globalTreeSame));
        if (firstCall_)
            LET_TRY(converge(box, keys, numKeys, assignment, globalLeaves, globalCounts, numGlobalLeaves, invThetaEff));
        LET_TRY(updateMinMac(assignment, invThetaEff));
        bool converged = false;
        LET_TRY(updateTree(assignment, box, &converged));
        LET_TRY(updateCounts(keys, numKeys, globalLeaves, globalCounts, numGlobalLeaves));
        LET_TRY(discoverHalos(box, h, haloSearchExt));
        LET_TRY(computeLayout(externalFailure));
        firstCall_ = false;
        return CSTONE_OK;
    }

    /*! The focus-tree part of Domain::syncGrav (R/domain/domain.hpp:266-318): like update(), with the tree resolved by the
     *  VECTOR MAC on the mass centres of the nodes (expansion centres) instead of the minimum-distance MAC.
     *  x, y, z, m: the rank's assigned particles in the order of keys (device; m: massBits = 32 | 64);
     *  globalLeavesHost: host copy of the global leaf array (numGlobalLeaves + 1 keys); centerDriftTol: Domain's
     *  centerDriftTol_ (raised by 0.05 whenever halo cells turn out to belong to nobody, like the reference).
     *  Afterwards expansionCenters() holds (centre of mass, MAC radius^2) per node.  Collective. */
    int updateGrav(const cstone_box& box, const K* keys, size_t numKeys, const K* assignment, const K* globalLeaves,
                   const K* globalLeavesHost, const uint32_t* globalCounts, int numGlobalLeaves, const T* x, const T* y,
                   const T* z, const void* m, int massBits, const T* h, float haloSearchExt, float* centerDriftTol,
                   int externalFailure = 0, bool globalTreeSame = false)
    {
        const float invThetaEff = 1.0f / theta_ + std::sqrt(3.0f); // invThetaVecMac, R/traversal/macs.hpp:48
        LET_TRY(init());
        LET_TRY(findPeers(assignment, globalLeaves, numGlobalLeaves, box, invThetaEff, globalTreeSame));
        auto centres = [&]() { return updateCenters(x, y, z, m, massBits, globalLeaves, globalLeavesHost, numGlobalLeaves); };
        if (firstCall_)
        {
            // first rough convergence to avoid computing expansion centres of large nodes with a lot of particles (:270-287)
            LET_TRY(converge(box, keys, numKeys, assignment, globalLeaves, globalCounts, numGlobalLeaves, 1.0f));
            LET_TRY(updateMinMac(assignment, 1.0f));
            uint32_t converged = 0;
            int reps           = 0;
            while (int(converged) != P_ || reps < 2)
            {
                bool conv = false;
                LET_TRY(updateTree(assignment, box, &conv));
                LET_TRY(updateCounts(keys, numKeys, globalLeaves, globalCounts, numGlobalLeaves));
                LET_TRY(centres());
                LET_TRY(updateMacs(assignment, 1.0f / theta_));
                LET_TRY(allReduceSum(conv ? 1u : 0u, &converged));
                if (++reps > 128) return fail(CSTONE_E_INTERNAL, "focus tree (gravity) does not converge");
            }
        }
        uint32_t failed = 0;
        int guard       = 0;
        do
        {
            LET_TRY(updateMacs(assignment, *centerDriftTol / theta_));
            bool conv = false;
            LET_TRY(updateTree(assignment, box, &conv));
            LET_TRY(updateCounts(keys, numKeys, globalLeaves, globalCounts, numGlobalLeaves));
            LET_TRY(centres());
            LET_TRY(updateMacs(assignment, 1.0f / theta_));
            LET_TRY(discoverHalos(box, h, haloSearchExt));
            LET_TRY(addMacs());
            bool unmatched = false;
            LET_TRY(computeLayout(externalFailure, &unmatched));
            // (the rows of computeLayout carry everybody's status: `unmatched` is the same on every rank, the
            //  MPI_Allreduce of :311 is part of that exchange)
            failed = unmatched ? 1u : 0u;
            if (failed)
            {
                *centerDriftTol += 0.05f;
                if (++guard > 64) return fail(CSTONE_E_INTERNAL, "syncGrav: halo cells keep falling outside the peers' ranges");
            }
        } while (failed);
        firstCall_ = false;
        return CSTONE_OK;
    }

    /*! Domain::updateExpansionCenters (R/domain/domain.hpp:415-421): mass centres of the focus tree from the particles as
     *  they are now (x, y, z, m: the assigned particles, device) and the MAC radii for 1 / theta.  Collective. */
    int updateExpansionCenters(const T* x, const T* y, const T* z, const void* m, int massBits, const K* globalLeaves,
                               const K* globalLeavesHost, int numGlobalLeaves)
    {
        LET_TRY(updateCenters(x, y, z, m, massBits, globalLeaves, globalLeavesHost, numGlobalLeaves));
        return cstone_hip_set_mac(ctx_, curve_, kb, rb, prefixes_.p, numNodesOf(L_), centers_.p, 1.0f / theta_, &box_);
    }

    //! (centre of mass, MAC radius^2) of every node of the focus tree: Vec4<T>[numNodes()], after updateGrav
    const T* expansionCenters() const { return centers_.as<T>(); }
    const char* macs() const { return macs_.as<char>(); }

    /*! Halos::exchangeHalos (R/halos/halos.hpp:232-253): array is laid out like the particle buffers of the last
     *  update (numParticlesWithHalos() elements of elemBytes bytes): its assigned range is read, the halo ranges are
     *  overwritten with the owners' values.  Collective. */
    int exchangeHalos(void* array, int elemBytes)
    {
        if (P_ == 1) return CSTONE_OK;
        const size_t e = size_t(elemBytes);
        LET_TRY(haloSend_.ensure(std::max<uint64_t>(sendTotal_, 1) * e));
        LET_TRY(haloRecv_.ensure(std::max<uint64_t>(recvTotal_, 1) * e));
        if (numSendRanges_)
            LET_TRY(cstone_hip_gather_ranges(ctx_, elemBytes, 32, rangeScan_.p, rangeOffsets_.p, numSendRanges_, array,
                                             haloSend_.p, size_t(sendTotal_)));
        std::vector<size_t> sb(P_), rbv(P_);
        for (int p = 0; p < P_; ++p)
            sb[p] = size_t(haloSendCounts_[p]) * e, rbv[p] = size_t(haloRecvCounts_[p]) * e;
        LET_TRY(commCall(comm_.all_to_all_v(comm_.user, haloSend_.p, sb.data(), haloRecv_.p, rbv.data()),
                         "all_to_all_v (halos)"));
        // incoming ranges: [layout[start_p], layout[end_p]) per peer, back to back in rank order on both sides of the
        // assigned block (R/domain/layout.hpp:175-190)
        char* a       = static_cast<char*>(array);
        uint64_t done = 0;
        for (int p = 0; p < P_; ++p)
        {
            if (!haloRecvCounts_[p]) continue;
            LET_TRY(cstone_hip_memcpy_d2d(ctx_, a + size_t(haloRecvOffsets_[p]) * e, haloRecv_.template as<char>() + done * e,
                                          size_t(haloRecvCounts_[p]) * e));
            done += haloRecvCounts_[p];
        }
        return CSTONE_OK;
    }

    /*! The same for up to four arrays of equal element size (4 or 8 bytes) in ONE message per peer: the arrays travel
     *  as rows (cstone_hip_gather_ranges_rows / cstone_hip_scatter_rows).  What arrives from the ranks below me is one
     *  block in front of my particles and what arrives from the ranks above one block behind them, both in rank order
     *  like the receive buffer (R/domain/layout.hpp:175-190), so two launches take the rows apart. */
    int exchangeHalosRows(void* const* arrays, int numArrays, int elemBytes)
    {
        if (P_ == 1) return CSTONE_OK;
        if (numArrays < 1 || numArrays > 4 || (elemBytes != 4 && elemBytes != 8))
            return fail(CSTONE_E_ARG, "exchangeHalosRows: %d arrays of %d-byte elements", numArrays, elemBytes);
        const size_t e = size_t(elemBytes) * size_t(numArrays);
        LET_TRY(haloSend_.ensure(std::max<uint64_t>(sendTotal_, 1) * e));
        LET_TRY(haloRecv_.ensure(std::max<uint64_t>(recvTotal_, 1) * e));
        if (numSendRanges_)
            LET_TRY(cstone_hip_gather_ranges_rows(ctx_, elemBytes, numArrays, rangeScan_.as<uint32_t>(),
                                                  rangeOffsets_.as<uint32_t>(), numSendRanges_, arrays, haloSend_.p,
                                                  size_t(sendTotal_)));
        std::vector<size_t> sb(P_), rbv(P_);
        uint64_t below = 0, above = 0;
        for (int p = 0; p < P_; ++p)
        {
            sb[p] = size_t(haloSendCounts_[p]) * e, rbv[p] = size_t(haloRecvCounts_[p]) * e;
            (p < rank_ ? below : above) += haloRecvCounts_[p];
        }
        LET_TRY(commCall(comm_.all_to_all_v(comm_.user, haloSend_.p, sb.data(), haloRecv_.p, rbv.data()),
                         "all_to_all_v (halos)"));
        if (below != particleStart_ || above != particleTotal_ - particleEnd_)
            return fail(CSTONE_E_INTERNAL, "halo exchange: %llu + %llu incoming particles, the layout has room for %u + %u",
                        (unsigned long long)below, (unsigned long long)above, particleStart_,
                        particleTotal_ - particleEnd_);
        LET_TRY(cstone_hip_scatter_rows(ctx_, elemBytes, numArrays, haloRecv_.p, size_t(below), arrays, 0));
        LET_TRY(cstone_hip_scatter_rows(ctx_, elemBytes, numArrays, haloRecv_.template as<char>() + below * e, size_t(above),
                                        arrays, size_t(particleEnd_)));
        return CSTONE_OK;
    }

    // ---- results of the last update ------------------------------------------------------------------------------
    //! level ranges of the focus tree on the host (from the last update's one read-back of the layout)
    const std::vector<int32_t>& levelRangeHost() const { return levelRangeHost_; }
    int numLeaves() const { return L_; }
    int numNodes() const { return numNodesOf(L_); }
    const K* leaves() const { return leaves_.as<K>(); }
    const uint32_t* leafCounts() const { return leafCounts_.as<uint32_t>(); }
    const uint32_t* layout() const { return layout_.as<uint32_t>(); }
    const int32_t* haloFlags() const { return flags_.as<int32_t>(); }
    const K* prefixes() const { return prefixes_.as<K>(); }
    const int32_t* childOffsets() const { return child_.as<int32_t>(); }
    const int32_t* parents() const { return parents_.as<int32_t>(); }
    const int32_t* levelRange() const { return levelRange_.as<int32_t>(); }
    const int32_t* internalToLeaf() const { return itl_.as<int32_t>(); }
    const int32_t* leafToInternal() const { return lti_.as<int32_t>(); }
    const T* geoCenters() const { return geoCenters_.as<T>(); }
    const T* geoSizes() const { return geoSizes_.as<T>(); }
    const uint32_t* nodeCounts() const { return counts_.as<uint32_t>(); }
    const std::vector<LetRange>& assignment() const { return assignment_; }
    const std::vector<int>& peers() const { return peers_; }
    int startCell() const { return assignment_[rank_].start; }
    int endCell() const { return assignment_[rank_].end; }
    //! Domain::startIndex / endIndex / nParticlesWithHalos (R/domain/domain.hpp:388-397) of the layout just computed
    uint32_t startIndex() const { return particleStart_; }
    uint32_t endIndex() const { return particleEnd_; }
    uint32_t numParticlesWithHalos() const { return particleTotal_; }
    //! which of the rarer paths the updates so far have taken (diagnostics and tests)
    struct Stats
    {
        uint64_t treeUpdates = 0, treeBuilds = 0, focusTransfers = 0, keysTransferred = 0, macRefineSteps = 0,
                 keysInjected = 0, keysRejected = 0, leavesFromGlobal = 0, convergeSteps = 0, peerSearchesSkipped = 0;
    };
    const Stats& stats() const { return stats_; }
    uint64_t halosSent() const { return sendTotal_; }
    uint64_t halosReceived() const { return recvTotal_; }

private:
    static int numNodesOf(int L) { return L + (L - 1) / 7; }
    static int numInternalOf(int L) { return (L - 1) / 7; }

    std::vector<LetBuf*> allBufs()
    {
        return {&leaves_, &leavesNew_, &prefixes_, &child_, &parents_, &levelRange_, &itl_, &lti_, &counts_, &leafCounts_,
                &macs_, &centers_, &geoCenters_, &geoSizes_, &opsAll_, &ops_, &scratchKeys_, &scratchKeys2_, &scratchIdx_,
                &scratchIdx2_, &scratchU64_, &gPrefixes_, &gChild_, &gParents_, &gLevelRange_, &gItl_, &gLti_, &treelets_,
                &treeletIdx_, &tlFlags_, &tlScan_, &sendBuf_, &recvBuf_, &layout_, &flags_, &radii_, &rangeOffsets_,
                &rangeScan_, &haloSend_, &haloRecv_, &rowBuf_, &gSeg_, &gCenters_};
    }

    int fail(int code, const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        return cstone_hip_raise(ctx_, code, buf);
    }

    int commCall(int rc, const char* what)
    {
        if (rc != 0) return fail(CSTONE_E_INTERNAL, "collective %s failed with code %d", what, rc);
        return CSTONE_OK;
    }

    // ---- small helpers on the C ABI --------------------------------------------------------------------------------

    //! result[q] = index of the first of keys[0 .. n) that is >= queries[q] (host in, host out)
    int lowerBounds(const K* keys, size_t n, const std::vector<K>& queries, std::vector<int64_t>& result)
    {
        const size_t m = queries.size();
        result.assign(m, 0);
        if (m == 0) return CSTONE_OK;
        // searches in the leaf array are remembered until it changes: the same few keys (the assignment) are looked up
        // before, during and after an update of the tree, each time a round trip to the device
        const bool inLeaves = keys == leaves_.template as<K>();
        if (inLeaves)
        {
            // the ends of the curve need no search: a cornerstone array starts with 0 and ends with endKey (and holds
            // nothing in between 0 and its second key).  With one rank these are all the keys anybody asks for.
            if (memoVersion_ != treeVersion_) memo_.clear(), memoVersion_ = treeVersion_;
            for (size_t i = 0; i < m; ++i)
            {
                const K q = queries[i];
                int64_t known = -1;
                if (q == 0) known = 0;
                else if (q == 1) known = 1;
                else if (q == endKey()) known = L_;
                else if (q == K(endKey() + 1)) known = int64_t(L_) + 1;
                if (known >= 0 && !memo_.count(q)) memo_[q] = {size_t(L_) + 2, known}; // (valid for any search length)
            }
        }
        if (inLeaves)
        {
            if (memoVersion_ != treeVersion_) memo_.clear(), memoVersion_ = treeVersion_;
            bool all = true;
            for (size_t i = 0; i < m && all; ++i)
            {
                // a search over the first n' keys remembered as (n', v): over n <= n' keys the answer is min(v, n), over
                // more keys it is v if v was found inside the first n'
                auto it = memo_.find(queries[i]);
                all     = it != memo_.end() && (n <= it->second.first || it->second.second < int64_t(it->second.first));
                if (all) result[i] = std::min<int64_t>(it->second.second, int64_t(n));
            }
            if (all) return CSTONE_OK;
        }
        LET_TRY(scratchU64_.ensure(m * (sizeof(K) + 8) + 64));
        K* dq        = scratchU64_.as<K>();
        uint64_t* dr = reinterpret_cast<uint64_t*>(scratchU64_.as<char>() + ((m * sizeof(K) + 63) / 64) * 64);
        LET_TRY(cstone_hip_upload(ctx_, dq, queries.data(), m * sizeof(K)));
        LET_TRY(cstone_hip_lower_bound(ctx_, kb, keys, n, dq, int(m), dr));
        std::vector<uint64_t> r(m);
        LET_TRY(cstone_hip_memcpy_d2h(ctx_, r.data(), dr, m * 8));
        for (size_t i = 0; i < m; ++i)
        {
            result[i] = int64_t(r[i]);
            if (inLeaves) memo_[queries[i]] = {n, result[i]};
        }
        return CSTONE_OK;
    }

    //! one 32-bit value from the device
    template<class V>
    int readBack(const V* dev, V* out, size_t count = 1)
    {
        return cstone_hip_memcpy_d2h(ctx_, out, dev, count * sizeof(V));
    }

    //! the rows of everybody: matrix[src * width + i] = row of rank src (host in, host out)
    int gatherRows(const std::vector<uint64_t>& mine, std::vector<uint64_t>& matrix)
    {
        const size_t w = mine.size();
        matrix.assign(w * P_, 0);
        if (P_ == 1)
        {
            std::copy(mine.begin(), mine.end(), matrix.begin());
            return CSTONE_OK;
        }
        LET_TRY(rowBuf_.ensure(w * 8 * (P_ + 1)));
        uint64_t* send = rowBuf_.as<uint64_t>();
        uint64_t* recv = send + w;
        LET_TRY(cstone_hip_upload(ctx_, send, mine.data(), w * 8));
        LET_TRY(commCall(comm_.all_gather(comm_.user, send, recv, w * 8), "all_gather (counts)"));
        return cstone_hip_memcpy_d2h(ctx_, matrix.data(), recv, w * 8 * P_);
    }

    //! room for a row of w values of V that a device operation writes (send) and the rows of everybody (recv)
    template<class V>
    int rowBuffers(size_t w, V** send, V** recv)
    {
        LET_TRY(rowBuf_.ensure(w * sizeof(V) * (size_t(P_) + 1) + 64));
        *send = rowBuf_.as<V>();
        *recv = *send + w;
        return CSTONE_OK;
    }
    /*! the rows of everybody for a row that IS ON THE DEVICE already (rowBuffers): all-gather and ONE read-back -- a row
     *  computed on the device does not travel to the host and back before the collective.  extra / extraBytes: more
     *  device data that rides on the same read-back (it must lie directly in front of the send row) */
    template<class V>
    int gatherRowsDev(size_t w, std::vector<V>& matrix, void* extraHost = nullptr, size_t extraBytes = 0)
    {
        V* send = rowBuf_.as<V>() + extraBytes / sizeof(V);
        V* recv = send + w;
        matrix.assign(w * P_, V(0));
        LET_TRY(commCall(comm_.all_gather(comm_.user, send, recv, w * sizeof(V)), "all_gather (rows)"));
        if (!extraBytes) return cstone_hip_memcpy_d2h(ctx_, matrix.data(), recv, w * sizeof(V) * P_);
        // [extra | send row | rows of everybody] in one copy
        std::vector<char> all(extraBytes + w * sizeof(V) * (size_t(P_) + 1));
        LET_TRY(cstone_hip_memcpy_d2h(ctx_, all.data(), rowBuf_.p, all.size()));
        std::memcpy(extraHost, all.data(), extraBytes);
        std::memcpy(matrix.data(), all.data() + extraBytes + w * sizeof(V), w * sizeof(V) * P_);
        return CSTONE_OK;
    }

    /*! variable all-to-all of elements of elemBytes: sendCounts[p] elements for rank p lie back to back in send (device);
     *  the counts the others send me are exchanged first (the reference probes the message sizes, MPI_Probe /
     *  MPI_Get_count).  anyGlobal = false on return: nobody sends anything, the data collective was skipped */
    int exchangeV(const void* send, const std::vector<uint64_t>& sendCounts, int elemBytes, LetBuf& recv,
                  std::vector<uint64_t>& recvCounts, bool* anyGlobal = nullptr,
                  const std::vector<uint64_t>* knownMatrix = nullptr)
    {
        std::vector<uint64_t> matrix;
        if (knownMatrix) { matrix = *knownMatrix; } // (the count rows came back with an earlier read-back)
        else { LET_TRY(gatherRows(sendCounts, matrix)); }
        recvCounts.assign(P_, 0);
        uint64_t total = 0, any = 0;
        for (int p = 0; p < P_; ++p)
        {
            recvCounts[p] = matrix[size_t(p) * P_ + rank_];
            total += recvCounts[p];
            for (int q = 0; q < P_; ++q)
                any += matrix[size_t(p) * P_ + q];
        }
        if (anyGlobal) *anyGlobal = any != 0;
        if (any == 0) return CSTONE_OK;
        LET_TRY(recv.ensure(std::max<uint64_t>(total, 1) * elemBytes));
        return allToAll(send, sendCounts, elemBytes, recv.p, recvCounts);
    }

    int allToAll(const void* send, const std::vector<uint64_t>& sendCounts, int elemBytes, void* recv,
                 const std::vector<uint64_t>& recvCounts)
    {
        std::vector<size_t> sb(P_), rbv(P_);
        for (int p = 0; p < P_; ++p)
            sb[p] = size_t(sendCounts[p]) * elemBytes, rbv[p] = size_t(recvCounts[p]) * elemBytes;
        if (P_ == 1) return CSTONE_OK;
        return commCall(comm_.all_to_all_v(comm_.user, send, sb.data(), recv, rbv.data()), "all_to_all_v");
    }

    //! linked octree of leaves_[0 .. L_]
    int buildOctree()
    {
        const int L = L_, M = numNodesOf(L);
        LET_TRY(prefixes_.ensure(size_t(M) * sizeof(K)));
        LET_TRY(child_.ensure(size_t(M + 1) * 4));
        LET_TRY(parents_.ensure(size_t(std::max(1, (M - 1) / 8)) * 4));
        LET_TRY(levelRange_.ensure(size_t(maxLevel + 2) * 4));
        LET_TRY(itl_.ensure(size_t(M) * 4));
        LET_TRY(lti_.ensure(size_t(M) * 4));
        ++stats_.treeBuilds;
        return cstone_hip_build_octree_bounded(ctx_, kb, leaves_.p, L, prefixes_.p, child_.as<int32_t>(),
                                               parents_.as<int32_t>(), levelRange_.as<int32_t>(), itl_.as<int32_t>(),
                                               lti_.as<int32_t>(), levelBound_);
    }

    /*! The leaf array changed.  levelBound_ bounds the level of its deepest leaf from above without a look at the
     *  device: a rebalance step splits a leaf by one level (deeper = true), keys that join the array bring their own
     *  level along, and the exact depth comes back with the read-back at the end of every update (computeLayout).  The
     *  linked octree then sorts its node keys over the digits that can be set only and the upsweeps launch the levels
     *  that can exist only. */
    void treeChanged(bool deeper)
    {
        ++treeVersion_;
        if (deeper) levelBound_ = std::min(levelBound_ + 1, maxLevel);
    }
    //! level of the coarsest octree node that can start at key
    static int levelOfKey(K key)
    {
        if (key == 0) return 0;
        int tz = 0;
        while (((key >> tz) & 1) == 0)
            ++tz;
        return std::max(0, maxLevel - tz / 3);
    }

    //! the leaf-order part of leafToInternal (leafToInternal(tree) of R/tree/octree.hpp:366-375)
    const uint32_t* leafMap() const { return lti_.as<uint32_t>() + numInternalOf(L_); }

    /*! rebalanceTree (R/tree/csarray.hpp:396-409) with the node ops in ops_[0 .. L_] (entry L_ is ignored): exclusive
     *  scan, new leaf array, swap.  The ops array is overwritten by its scan. */
    int rebalanceFromOps()
    {
        const int L = L_;
        LET_TRY(cstone_hip_exclusive_scan_u32(ctx_, ops_.as<uint32_t>(), ops_.as<uint32_t>(), size_t(L) + 1, 0u));
        int32_t newL = 0;
        LET_TRY(readBack(ops_.as<int32_t>() + L, &newL));
        if (newL < 1) return fail(CSTONE_E_INTERNAL, "focus tree: rebalance produced %d leaves", newL);
        LET_TRY(leavesNew_.ensure(size_t(newL + 1) * sizeof(K)));
        LET_TRY(cstone_hip_rebalance_tree(ctx_, kb, leaves_.p, L, newL, ops_.as<int32_t>(), leavesNew_.p));
        leaves_.swap(leavesNew_);
        L_ = newL;
        treeChanged(true);
        return CSTONE_OK;
    }

    // ---- state of a fresh FocusedOctree (octree_focus_mpi.hpp:69-98) ------------------------------------------------
    int init()
    {
        if (L_ != 0) return CSTONE_OK;
        const K root[2] = {K(0), endKey()};
        LET_TRY(leaves_.ensure(2 * sizeof(K)));
        LET_TRY(cstone_hip_upload(ctx_, leaves_.p, root, sizeof root));
        L_ = 1;
        levelBound_ = 0;
        treeChanged(false);
        LET_TRY(buildOctree());
        const uint32_t c0 = bucket_ + 1; // counts_{bucketSize + 1}
        const char m0     = 1;           // macs_{1}
        LET_TRY(counts_.ensure(4));
        LET_TRY(macs_.ensure(1));
        LET_TRY(cstone_hip_upload(ctx_, counts_.p, &c0, 4));
        LET_TRY(cstone_hip_upload(ctx_, macs_.p, &m0, 1));
        haveLeafCounts_ = false;
        return CSTONE_OK;
    }

    // ---- peers ---------------------------------------------------------------------------------------------------
    int findPeers(const K* assignment, const K* globalLeaves, int numGlobalLeaves, const cstone_box& box, float invThetaEff,
                  bool globalTreeSame)
    {
        if (P_ == 1) return CSTONE_OK;
        // the peers are a function of (global tree, assignment, box): a sync that changed none of them keeps its peers
        bool same = globalTreeSame && peersValid_ && numGlobalLeaves == peersGlobalLeaves_;
        for (int r = 0; r <= P_ && same; ++r)
            same = assignment[r] == peersAssignment_[r];
        for (int k = 0; k < 6 && same; ++k)
            same = box.lim[k] == peersBox_.lim[k];
        if (same)
        {
            ++stats_.peerSearchesSkipped;
            return CSTONE_OK;
        }
        peers_.clear();
        peersValid_ = false;
        const int GL = numGlobalLeaves, GM = numNodesOf(GL);
        LET_TRY(gPrefixes_.ensure(size_t(GM) * sizeof(K)));
        LET_TRY(gChild_.ensure(size_t(GM + 1) * 4));
        LET_TRY(gParents_.ensure(size_t(std::max(1, (GM - 1) / 8)) * 4));
        LET_TRY(gLevelRange_.ensure(size_t(maxLevel + 2) * 4));
        LET_TRY(gItl_.ensure(size_t(GM) * 4));
        LET_TRY(gLti_.ensure(size_t(GM) * 4));
        LET_TRY(cstone_hip_build_octree(ctx_, kb, globalLeaves, GL, gPrefixes_.p, gChild_.as<int32_t>(),
                                        gParents_.as<int32_t>(), gLevelRange_.as<int32_t>(), gItl_.as<int32_t>(),
                                        gLti_.as<int32_t>()));
        LET_TRY(readBack(gLevelRange_.as<int32_t>(), gLevelHost_, size_t(maxLevel) + 2)); // (upsweeps over the global tree)
        std::vector<uint64_t> a64(P_ + 1);
        for (int r = 0; r <= P_; ++r)
            a64[r] = uint64_t(assignment[r]);
        std::vector<int32_t> flags(P_, 0);
        LET_TRY(cstone_hip_find_peers_mac(ctx_, curve_, kb, rb, gPrefixes_.p, gChild_.as<int32_t>(),
                                          gLevelRange_.as<int32_t>(), a64.data(), P_, rank_, &box, invThetaEff,
                                          flags.data()));
        for (int r = 0; r < P_; ++r)
            if (flags[r] && r != rank_) peers_.push_back(r);
        peersAssignment_.assign(assignment, assignment + P_ + 1);
        peersBox_          = box;
        peersGlobalLeaves_ = numGlobalLeaves;
        peersValid_        = true;
        return CSTONE_OK;
    }

    // ---- MAC criteria (updateMinMac + updateMacs, octree_focus_mpi.hpp:457-531) ----------------------------------------
    int updateMinMac(const K* assignment, float invThetaEff)
    {
        const int M = numNodesOf(L_);
        LET_TRY(centers_.ensure(size_t(M) * 4 * sizeof(T)));
        // centers_[i] = computeMinMacR2(prefix, invThetaEff, box_) with the box of the LAST updateTree; the radius that
        // setMacRadius then derives from these centres is the same number (distance to the geometric centre = 0)
        LET_TRY(cstone_hip_geo_mac_spheres(ctx_, curve_, kb, rb, prefixes_.p, M, centers_.p, invThetaEff, &box_));
        LET_TRY(macs_.ensure(size_t(M)));
        LET_TRY(cstone_hip_memset(ctx_, macs_.p, 0, size_t(M)));
        // the assignment may have changed: its start and end in the CURRENT leaves, searched among the first L keys
        std::vector<int64_t> idx;
        LET_TRY(lowerBounds(leaves_.as<K>(), size_t(L_), {assignment[rank_], assignment[rank_ + 1]}, idx));
        const int fStart = int(idx[0]), fEnd = int(idx[1]);
        if (fEnd > fStart)
            LET_TRY(cstone_hip_mark_macs(ctx_, curve_, kb, rb, prefixes_.p, child_.as<int32_t>(), centers_.p, &box_,
                                         leaves_.as<K>() + fStart, fEnd - fStart, 0, macs_.as<char>()));
        haveMacs_ = true;
        return CSTONE_OK;
    }

    // ---- updateTree (octree_focus_mpi.hpp:108-187) ---------------------------------------------------------------------
    int updateTree(const K* assignment, const cstone_box& box, bool* convergedOut)
    {
        if (!(haveMacs_ && haveCounts_)) // rebalanceStatus_ != valid (:110-113)
            return fail(CSTONE_E_INTERNAL, "update of criteria required before updating the tree structure");
        const K focusStart = assignment[rank_], focusEnd = assignment[rank_ + 1];
        // (the reference recognises its first update by prevFocusStart == prevFocusEnd == 0; a rank 0 whose range is the
        //  empty [0, 0) would stay "first" for ever and skip the collective of focusTransfer that its peers enter: an
        //  explicit flag, set on every rank by the same update)
        const bool firstUpdate = !updatedOnce_;
        if (firstUpdate) prevFocusStart_ = focusStart, prevFocusEnd_ = focusEnd;

        std::vector<K> enforced;
        LET_TRY(focusTransfer(assignment, firstUpdate, enforced));
        for (int peer : peers_)
        {
            enforced.push_back(assignment[peer]);
            enforced.push_back(assignment[peer + 1]);
        }
        enforced.erase(std::unique(enforced.begin(), enforced.end()), enforced.end());

        bool converged = false;
        LET_TRY(updateFocus(focusStart, focusEnd, enforced, &converged));
        const float invThetaRefine = std::sqrt(3.0f) / 2 + 1e-6f; // octree_focus_mpi.hpp:139: float(sqrt(3)/2 + 1e-6)
        bool refined                = false;
        int guard                   = 0;
        while (!refined)
        {
            LET_TRY(macRefine(prevFocusStart_, prevFocusEnd_, focusStart, focusEnd, invThetaRefine, box, &refined));
            if (++guard > 8 * maxLevel) return fail(CSTONE_E_INTERNAL, "focus tree: MAC refinement does not end");
        }
        // translateAssignment + the count rows of the treelet exchange: when the assignment keys have to be searched in
        // the (changed) leaf array anyway, the device turns the search results into this rank's row of treelet sizes, the
        // rows are all-gathered, and ONE read-back brings the search results and everybody's rows
        std::vector<uint64_t> treeletMatrix;
        bool haveTreeletMatrix = false;
        LET_TRY(translateAssignmentWithCounts(assignment, treeletMatrix, &haveTreeletMatrix));
        LET_TRY(syncTreelets(haveTreeletMatrix ? &treeletMatrix : nullptr));
        LET_TRY(indexTreelets());
        LET_TRY(translateAssignment(assignment)); // (answered from memory unless keys were rejected: lowerBounds)
        std::copy(assignment, assignment + P_ + 1, globAssignment_.begin());

        box_            = box;
        prevFocusStart_ = focusStart;
        prevFocusEnd_   = focusEnd;
        haveMacs_ = haveCounts_ = false;
        updatedOnce_ = true;
        ++stats_.treeUpdates;
        // updateGeoCenters (:614-625)
        const int M = numNodesOf(L_);
        LET_TRY(geoCenters_.ensure(size_t(M) * 3 * sizeof(T)));
        LET_TRY(geoSizes_.ensure(size_t(M) * 3 * sizeof(T)));
        LET_TRY(cstone_hip_node_centers(ctx_, curve_, kb, rb, prefixes_.p, M, &box_, geoCenters_.p, geoSizes_.p));
        *convergedOut = converged;
        return CSTONE_OK;
    }

    /*! focusTransfer (exchange_focus.hpp:364-433): a rank whose range shrank hands the part of its tree that covers the
     *  lost key range to the new owner (one rebalance step with the last leaf counts applied, updateTreelet
     *  R/tree/csarray.hpp:477-488); the keys a rank receives become mandatory keys of its next tree.  Every rank knows
     *  the old and the new assignment of everybody, so all of them agree on whether anything is handed over at all. */
    int focusTransfer(const K* assignment, bool firstUpdate, std::vector<K>& buffer)
    {
        if (firstUpdate || P_ == 1) return CSTONE_OK;
        bool anyChange = false;
        for (int r = 0; r <= P_; ++r)
            anyChange = anyChange || assignment[r] != globAssignment_[r];
        if (!anyChange) return CSTONE_OK;
        if (!haveLeafCounts_) return fail(CSTONE_E_INTERNAL, "focus transfer without leaf counts");

        const K oldStart = prevFocusStart_, oldEnd = prevFocusEnd_;
        const K newStart = assignment[rank_], newEnd = assignment[rank_ + 1];
        std::vector<int64_t> idx;
        LET_TRY(lowerBounds(leaves_.as<K>(), size_t(L_) + 1, {oldStart, newStart, newEnd, oldEnd}, idx));
        // what I lost: [oldStart, newStart) to the rank below, [newEnd, oldEnd) to the rank above
        struct Part
        {
            int dest;
            int64_t first, last;
        };
        std::vector<Part> parts;
        if (oldStart < newStart) parts.push_back({rank_ - 1, idx[0], idx[1]});
        if (newEnd < oldEnd) parts.push_back({rank_ + 1, idx[2], idx[3]});
        std::vector<uint64_t> sendCounts(P_, 0);
        // the rebalanced treelets, back to back in the order [to the rank below | to the rank above] = rank order
        std::vector<std::pair<int, int>> produced; // (dest, number of keys to send)
        size_t used = 0;
        for (const Part& part : parts)
        {
            const int numNodes = int(part.last - part.first);
            int newNum         = 0;
            if (numNodes > 0)
            {
                int conv = 0;
                LET_TRY(ops_.ensure(size_t(numNodes + 2) * 4));
                LET_TRY(cstone_hip_compute_node_ops(ctx_, kb, leaves_.as<K>() + part.first, numNodes,
                                                    leafCounts_.as<uint32_t>() + part.first, bucket_, ops_.as<int32_t>(),
                                                    &newNum, &conv));
                LET_TRY(sendBuf_.ensure((used + size_t(newNum) + 1) * sizeof(K), true));
                LET_TRY(cstone_hip_rebalance_tree(ctx_, kb, leaves_.as<K>() + part.first, numNodes, newNum,
                                                  ops_.as<int32_t>(), sendBuf_.as<K>() + used));
            }
            // the last key of the treelet is not sent (mpiSendAsync(treelet.data(), treelet.size() - 1, ...)): the next
            // treelet overwrites it
            produced.push_back({part.dest, newNum});
            used += size_t(newNum);
        }
        for (auto [dest, count] : produced)
        {
            if (dest < 0 || dest >= P_) return fail(CSTONE_E_INTERNAL, "focus transfer to rank %d", dest);
            sendCounts[dest] = uint64_t(count);
        }
        LET_TRY(sendBuf_.ensure(std::max<size_t>(used, 1) * sizeof(K), true));
        std::vector<uint64_t> recvCounts;
        bool any = false;
        LET_TRY(exchangeV(sendBuf_.p, sendCounts, int(sizeof(K)), recvBuf_, recvCounts, &any));
        if (!any) return CSTONE_OK;
        // received keys: from the rank below first, then from the rank above (rank order = the order of the buffer)
        uint64_t recvTotal = 0;
        for (int p = 0; p < P_; ++p)
        {
            if (recvCounts[p] && p != rank_ - 1 && p != rank_ + 1)
                return fail(CSTONE_E_INTERNAL, "focus transfer from rank %d", p);
            recvTotal += recvCounts[p];
        }
        const size_t at = buffer.size();
        buffer.resize(at + recvTotal);
        if (recvTotal) LET_TRY(readBack(recvBuf_.as<K>(), buffer.data() + at, size_t(recvTotal)));
        ++stats_.focusTransfers;
        stats_.keysTransferred += recvTotal;
        return CSTONE_OK;
    }

    /*! CombinedUpdate::updateFocus (octree_focus.hpp:83-136): ops from counts and MACs, mandatory keys enforced,
     *  ancestors protected, rebalance, keys that one step cannot resolve injected */
    int updateFocus(K focusStart, K focusEnd, const std::vector<K>& mandatory, bool* convergedOut)
    {
        const int L = L_, M = numNodesOf(L), I = numInternalOf(L);
        LET_TRY(opsAll_.ensure(size_t(M + 1) * 4));
        LET_TRY(ops_.ensure(size_t(L + 2) * 4));
        std::vector<K> all{focusStart, focusEnd};
        all.insert(all.end(), mandatory.begin(), mandatory.end());
        LET_TRY(scratchKeys_.ensure(all.size() * sizeof(K)));
        LET_TRY(cstone_hip_upload(ctx_, scratchKeys_.p, all.data(), all.size() * sizeof(K)));
        // decisions from counts and MACs, mandatory keys enforced, ancestors protected, the leaves' ops scanned: one
        // call, one read-back: {status of the enforced keys, converged, every leaf keeps, new number of leaves}
        int res[4] = {0, 0, 0, 0};
        LET_TRY(cstone_hip_focus_update_ops(ctx_, kb, prefixes_.p, child_.as<int32_t>(), parents_.as<int32_t>(),
                                            counts_.as<uint32_t>(), macs_.as<char>(), uint64_t(focusStart),
                                            uint64_t(focusEnd), bucket_, scratchKeys_.p, int(all.size()),
                                            lti_.as<int32_t>() + I, L, M, opsAll_.as<int32_t>(), ops_.as<int32_t>(), res));
        const int status = res[0], newL = res[3];
        // (every op "keep" and nothing to inject: the leaf array and the linked octree stay what they are; the reference
        //  rebuilds regardless because it parks the ops in the tree's arrays)
        const bool allKeep = status != 3 && res[2] != 0;
        if (!allKeep)
        {
            if (newL < 1) return fail(CSTONE_E_INTERNAL, "focus tree: rebalance produced %d leaves", newL);
            LET_TRY(leavesNew_.ensure(size_t(newL + 1) * sizeof(K)));
            LET_TRY(cstone_hip_rebalance_tree(ctx_, kb, leaves_.p, L, newL, ops_.as<int32_t>(), leavesNew_.p));
            leaves_.swap(leavesNew_);
            L_ = newL;
            treeChanged(true);
            if (status == 3)
            {
                LET_TRY(injectKeys(all));
                stats_.keysInjected += all.size();
            }
            LET_TRY(buildOctree());
        }
        *convergedOut = res[1] != 0;
        return CSTONE_OK;
    }

    /*! injectKeys (R/focus/inject.hpp:52-113): the keys join the leaf array, and every gap between two consecutive
     *  keys that is not a power-of-8 node is filled with the coarsest nodes that cover it */
    int injectKeys(const std::vector<K>& keys)
    {
        const size_t n = size_t(L_) + 1 + keys.size();
        LET_TRY(leaves_.ensure(n * sizeof(K), true));
        LET_TRY(cstone_hip_upload(ctx_, leaves_.as<K>() + L_ + 1, keys.data(), keys.size() * sizeof(K)));
        LET_TRY(cstone_hip_sort_keys(ctx_, kb, leaves_.p, n));
        LET_TRY(ops_.ensure((n + 1) * 4));
        LET_TRY(cstone_hip_count_sfc_gaps(ctx_, kb, leaves_.p, int(n) - 1, ops_.as<int32_t>()));
        LET_TRY(cstone_hip_memset(ctx_, ops_.as<int32_t>() + (n - 1), 0, 4));
        LET_TRY(cstone_hip_exclusive_scan_u32(ctx_, ops_.as<uint32_t>(), ops_.as<uint32_t>(), n, 0u));
        int32_t numGap = 0;
        LET_TRY(readBack(ops_.as<int32_t>() + (n - 1), &numGap));
        if (numGap < 1) return fail(CSTONE_E_INTERNAL, "focus tree: key injection produced %d leaves", numGap);
        LET_TRY(leavesNew_.ensure(size_t(numGap + 1) * sizeof(K)));
        LET_TRY(cstone_hip_fill_sfc_gaps(ctx_, kb, leaves_.p, int(n) - 1, ops_.as<int32_t>(), leavesNew_.p));
        leaves_.swap(leavesNew_);
        L_ = numGap;
        for (K key : keys)
            levelBound_ = std::max(levelBound_, levelOfKey(key));
        treeChanged(false);
        return CSTONE_OK;
    }

    /*! macRefine + updateMacRefine (octree_focus.hpp:218-279): when the focus has moved, the leaves that the vector MAC
     *  (relative to the newly gained part of the focus) asks for are split, one level per call */
    int macRefine(K oldStart, K oldEnd, K focusStart, K focusEnd, float invTheta, const cstone_box& box, bool* done)
    {
        *done = true;
        if (oldStart == focusStart && oldEnd == focusEnd) return CSTONE_OK;
        const int L = L_, M = numNodesOf(L), I = numInternalOf(L);
        LET_TRY(centers_.ensure(size_t(M) * 4 * sizeof(T)));
        LET_TRY(cstone_hip_geo_mac_spheres(ctx_, curve_, kb, rb, prefixes_.p, M, centers_.p, invTheta, &box));
        LET_TRY(macs_.ensure(size_t(M)));
        LET_TRY(cstone_hip_memset(ctx_, macs_.p, 0, size_t(M)));
        const K growthLower = focusStart < oldStart ? oldStart : focusStart;
        const K growthUpper = oldEnd < focusEnd ? oldEnd : focusEnd;
        std::vector<int64_t> idx;
        LET_TRY(lowerBounds(leaves_.as<K>(), size_t(L), {growthLower, growthUpper, focusStart, focusEnd}, idx));
        const int fGrowL = int(idx[0]), fGrowU = int(idx[1]), fStart = int(idx[2]), fEnd = int(idx[3]);
        if (fGrowL - fStart > 0)
            LET_TRY(cstone_hip_mark_macs(ctx_, curve_, kb, rb, prefixes_.p, child_.as<int32_t>(), centers_.p, &box,
                                         leaves_.as<K>() + fStart, fGrowL - fStart, 1, macs_.as<char>()));
        if (fEnd - fGrowU > 0)
            LET_TRY(cstone_hip_mark_macs(ctx_, curve_, kb, rb, prefixes_.p, child_.as<int32_t>(), centers_.p, &box,
                                         leaves_.as<K>() + fGrowU, fEnd - fGrowU, 1, macs_.as<char>()));
        LET_TRY(ops_.ensure(size_t(L + 2) * 4));
        LET_TRY(cstone_hip_mac_refine_decision(ctx_, kb, prefixes_.p, macs_.as<char>(), lti_.as<int32_t>() + I, L,
                                               fStart, fEnd, ops_.as<int32_t>()));
        LET_TRY(cstone_hip_memset(ctx_, ops_.as<int32_t>() + L, 0, 4));
        uint64_t ones = 0;
        LET_TRY(cstone_hip_count_equal(ctx_, 32, ops_.p, size_t(L), 1, &ones));
        *done = ones == uint64_t(L);
        if (*done) return CSTONE_OK; // nothing to split: leaves and octree are unchanged
        ++stats_.macRefineSteps;
        LET_TRY(rebalanceFromOps());
        return buildOctree();
    }

    /*! translateAssignment (domaindecomp.hpp:183-206): the leaf index ranges of the peers' and my own key ranges; a range
     *  whose boundary keys are not in the tree is narrowed */
    std::vector<K> assignmentQueries(const K* assignment) const
    {
        std::vector<K> q;
        q.reserve(2 * (P_ + 1));
        for (int r = 0; r <= P_; ++r)
            q.push_back(assignment[r]);
        for (int r = 0; r <= P_; ++r)
            q.push_back(K(assignment[r] + 1)); // upper_bound(key) = lower_bound(key + 1)
        return q;
    }

    /*! translateAssignment, and -- when the searches have to go to the device -- the matrix of treelet sizes of
     *  syncTreelets from the same read-back (cstone_hip_peer_range_counts) */
    int translateAssignmentWithCounts(const K* assignment, std::vector<uint64_t>& matrix, bool* haveMatrix)
    {
        *haveMatrix = false;
        const std::vector<K> q = assignmentQueries(assignment);
        bool known = P_ == 1 || peers_.empty();
        if (!known)
        {
            // (are all searches answered from memory?  lowerBounds() would not go to the device either)
            if (memoVersion_ != treeVersion_) memo_.clear(), memoVersion_ = treeVersion_;
            std::vector<int64_t> idx;
            std::vector<K> none;
            LET_TRY(lowerBounds(leaves_.as<K>(), size_t(L_) + 1, none, idx)); // (seeds the memory with the curve's ends)
            known = true;
            for (K key : q)
                known = known && memo_.count(key) != 0;
        }
        if (known) return translateAssignment(assignment);

        const size_t m = q.size(), w = size_t(P_);
        LET_TRY(rowBuf_.ensure((m + w * (size_t(P_) + 1)) * 8 + 64)); // [bounds (m) | my row (w) | rows of everybody]
        uint64_t* bounds = rowBuf_.as<uint64_t>();
        LET_TRY(scratchU64_.ensure(m * sizeof(K) + 64));
        LET_TRY(cstone_hip_upload(ctx_, scratchU64_.p, q.data(), m * sizeof(K)));
        LET_TRY(cstone_hip_lower_bound(ctx_, kb, leaves_.p, size_t(L_) + 1, scratchU64_.p, int(m), bounds));
        std::vector<uint8_t> isPeer(P_, 0);
        for (int peer : peers_)
            isPeer[peer] = 1;
        LET_TRY(cstone_hip_peer_range_counts(ctx_, bounds, isPeer.data(), P_, bounds + m));
        std::vector<uint64_t> found(m);
        LET_TRY(gatherRowsDev<uint64_t>(w, matrix, found.data(), m * 8));
        for (size_t i = 0; i < m; ++i)
            memo_[q[i]] = {size_t(L_) + 1, int64_t(found[i])};
        *haveMatrix = true;
        return translateAssignment(assignment); // (answered from memory now)
    }

    int translateAssignment(const K* assignment)
    {
        const std::vector<K> q = assignmentQueries(assignment);
        std::vector<int64_t> idx;
        LET_TRY(lowerBounds(leaves_.as<K>(), size_t(L_) + 1, q, idx));
        auto above = [&](int r) { return int32_t(idx[r]); };              // findNodeAbove(assignment[r])
        auto below = [&](int r) { return int32_t(idx[P_ + 1 + r]) - 1; }; // findNodeBelow(assignment[r])
        std::fill(assignment_.begin(), assignment_.end(), LetRange{});
        for (int peer : peers_)
        {
            int32_t s = above(peer), e = below(peer + 1);
            if (e < s) e = s;
            assignment_[peer] = LetRange{s, e};
        }
        assignment_[rank_] = LetRange{above(rank_), below(rank_ + 1)};
        return CSTONE_OK;
    }

    /*! syncTreelets (exchange_focus.hpp:196-217): every peer gets my view of ITS key range (exchangeTreelets :61-96);
     *  keys of such a treelet that the owner does not have are sent back (checkTreelets, exchangeRejectedKeys :98-194)
     *  and removed from the sender's tree; what remains of a treelet (pruneTreelets :118-129) is the node list the owner
     *  serves counts for */
    int syncTreelets(const std::vector<uint64_t>* sizeMatrix = nullptr)
    {
        std::fill(tlCount_.begin(), tlCount_.end(), 0);
        std::fill(tlOffset_.begin(), tlOffset_.end(), 0);
        if (P_ == 1 || peers_.empty())
        {
            // (a rank without peers still takes part in the count exchanges of the others)
            if (P_ == 1) return CSTONE_OK;
        }
        // ---- my leaves over each peer's range, including the upper boundary key
        std::vector<uint64_t> sendCounts(P_, 0), recvCounts;
        uint64_t sendTotal = 0;
        for (int peer : peers_)
        {
            sendCounts[peer] = uint64_t(assignment_[peer].count()) + 1;
            sendTotal += sendCounts[peer];
        }
        LET_TRY(sendBuf_.ensure(std::max<uint64_t>(sendTotal, 1) * sizeof(K)));
        uint64_t at = 0;
        for (int peer : peers_)
        {
            LET_TRY(cstone_hip_memcpy_d2d(ctx_, sendBuf_.as<K>() + at, leaves_.as<K>() + assignment_[peer].start,
                                          size_t(sendCounts[peer]) * sizeof(K)));
            at += sendCounts[peer];
        }
        bool any = false;
        if (sizeMatrix)
        {
            // (what the device computed from the search results must be what the host derives from them)
            for (int p = 0; p < P_; ++p)
                if ((*sizeMatrix)[size_t(rank_) * P_ + p] != sendCounts[p])
                    return fail(CSTONE_E_INTERNAL, "treelet exchange: %llu keys for rank %d, the device counted %llu",
                                (unsigned long long)sendCounts[p], p,
                                (unsigned long long)(*sizeMatrix)[size_t(rank_) * P_ + p]);
        }
        LET_TRY(exchangeV(sendBuf_.p, sendCounts, int(sizeof(K)), treelets_, recvCounts, &any, sizeMatrix));
        if (!any) return CSTONE_OK;
        uint64_t recvTotal = 0;
        std::vector<uint64_t> rOff(P_ + 1, 0);
        for (int p = 0; p < P_; ++p)
        {
            rOff[p + 1] = rOff[p] + recvCounts[p];
            recvTotal += recvCounts[p];
        }

        // ---- keys of the treelets that are not in my tree (all but the last key of every treelet are checked)
        std::vector<uint64_t> rejCounts(P_, 0), keepCounts(recvCounts);
        if (recvTotal)
        {
            LET_TRY(tlFlags_.ensure((recvTotal + 1) * 4));
            LET_TRY(tlScan_.ensure((recvTotal + 1) * 4));
            LET_TRY(cstone_hip_keys_missing(ctx_, kb, leaves_.p, L_, treelets_.p, size_t(recvTotal),
                                            tlFlags_.as<uint32_t>()));
            for (int p = 0; p < P_; ++p)
                if (recvCounts[p]) LET_TRY(cstone_hip_memset(ctx_, tlFlags_.as<uint32_t>() + rOff[p + 1] - 1, 0, 4));
            LET_TRY(cstone_hip_memset(ctx_, tlFlags_.as<uint32_t>() + recvTotal, 0, 4));
            LET_TRY(cstone_hip_exclusive_scan_u32(ctx_, tlFlags_.as<uint32_t>(), tlScan_.as<uint32_t>(),
                                                  size_t(recvTotal) + 1, 0u));
        }
        // rejected keys back to their senders, kept keys packed (both keep the rank order of the treelets); how many
        // keys are rejected is not known on the host yet: room for all of them
        LET_TRY(scratchKeys_.ensure(std::max<uint64_t>(recvTotal, 1) * sizeof(K)));
        LET_TRY(scratchKeys2_.ensure(std::max<uint64_t>(recvTotal, 1) * sizeof(K)));
        if (recvTotal)
            LET_TRY(cstone_hip_partition_keys(ctx_, kb, treelets_.p, tlFlags_.as<uint32_t>(), tlScan_.as<uint32_t>(),
                                              size_t(recvTotal), scratchKeys_.p, scratchKeys2_.p));
        // rejected keys per peer = differences of the scan at the treelet boundaries: this rank's row of the count matrix
        // is made on the device and goes straight into the all-gather; ONE read-back brings everybody's rows
        std::vector<uint64_t> rejRecv(P_, 0);
        bool anyRejected = false;
        {
            uint32_t *send = nullptr, *recv = nullptr;
            LET_TRY(rowBuffers<uint32_t>(size_t(P_), &send, &recv));
            if (recvTotal)
            {
                std::vector<uint32_t> map(P_ + 1);
                for (int p = 0; p <= P_; ++p)
                    map[p] = uint32_t(rOff[p]);
                LET_TRY(scratchIdx_.ensure(size_t(P_ + 1) * 8));
                uint32_t* dmap = scratchIdx_.as<uint32_t>();
                uint32_t* dval = dmap + (P_ + 1);
                LET_TRY(cstone_hip_upload(ctx_, dmap, map.data(), size_t(P_ + 1) * 4));
                LET_TRY(cstone_hip_gather(ctx_, 4, dmap, size_t(P_) + 1, tlScan_.p, dval));
                LET_TRY(cstone_hip_adjacent_difference_u32(ctx_, dval, size_t(P_), send));
            }
            else { LET_TRY(cstone_hip_memset(ctx_, send, 0, size_t(P_) * 4)); }
            std::vector<uint32_t> m32;
            LET_TRY(gatherRowsDev<uint32_t>(size_t(P_), m32));
            std::vector<uint64_t> rejMatrix(m32.begin(), m32.end());
            for (int p = 0; p < P_; ++p)
            {
                rejCounts[p]  = rejMatrix[size_t(rank_) * P_ + p];
                keepCounts[p] = recvCounts[p] - rejCounts[p];
            }
            LET_TRY(exchangeV(scratchKeys_.p, rejCounts, int(sizeof(K)), recvBuf_, rejRecv, &anyRejected, &rejMatrix));
        }
        // the pruned treelets
        treelets_.swap(scratchKeys2_);
        for (int p = 0; p < P_; ++p)
        {
            tlOffset_[p + 1] = tlOffset_[p] + keepCounts[p];
            tlCount_[p]      = keepCounts[p] ? keepCounts[p] - 1 : 0; // nodes = keys - 1
        }
        uint64_t rejRecvTotal = 0;
        for (int p = 0; p < P_; ++p)
            rejRecvTotal += rejRecv[p];
        stats_.keysRejected += rejRecvTotal;
        if (rejRecvTotal)
        {
            // nodeOps (one per key of the leaf array, all 1) with a 0 at every leaf that starts at a rejected key, then
            // rebalanceTree: those leaves merge into their predecessors
            const int L = L_;
            LET_TRY(ops_.ensure(size_t(L + 2) * 4));
            const int32_t one = 1;
            LET_TRY(cstone_hip_fill(ctx_, 4, ops_.p, size_t(L) + 1, &one));
            LET_TRY(cstone_hip_zero_ops_at_keys(ctx_, kb, leaves_.p, L, recvBuf_.p, size_t(rejRecvTotal),
                                                ops_.as<int32_t>()));
            LET_TRY(rebalanceFromOps());
            LET_TRY(buildOctree());
        }
        return CSTONE_OK;
    }

    //! indexTreelets (exchange_focus.hpp:266-287): node index in MY tree of every node of the peers' treelets
    int indexTreelets()
    {
        uint64_t nodes = 0;
        tlIdxOffset_.assign(P_ + 1, 0);
        for (int p = 0; p < P_; ++p)
        {
            tlIdxOffset_[p + 1] = tlIdxOffset_[p] + tlCount_[p];
            nodes += tlCount_[p];
        }
        LET_TRY(treeletIdx_.ensure(std::max<uint64_t>(nodes, 1) * 4));
        for (int peer : peers_)
        {
            if (!tlCount_[peer]) continue;
            LET_TRY(cstone_hip_locate_nodes(ctx_, kb, treelets_.as<K>() + tlOffset_[peer], size_t(tlCount_[peer]) + 1,
                                            prefixes_.p, levelRange_.as<int32_t>(),
                                            treeletIdx_.as<int32_t>() + tlIdxOffset_[peer]));
        }
        return CSTONE_OK;
    }

    // ---- updateCounts (octree_focus_mpi.hpp:205-273) ---------------------------------------------------------------------
    int updateCounts(const K* keys, size_t numKeys, const K* globalLeaves, const uint32_t* globalCounts,
                     int numGlobalLeaves)
    {
        const int L = L_, M = numNodesOf(L), I = numInternalOf(L);
        numKeys_ = numKeys;
        LET_TRY(leafCounts_.ensure(size_t(L) * 4));
        LET_TRY(cstone_hip_compute_node_counts(ctx_, kb, leaves_.p, leafCounts_.as<uint32_t>(), L, keys, numKeys,
                                               0xFFFFFFFFu));
        // leaves that neither I nor a peer own: counts from the global tree (invertRanges + enumerateRanges,
        // R/domain/layout.hpp:57-89, rangeCount R/focus/rebalance.hpp:279-301)
        std::vector<int32_t> fromGlobal;
        {
            int32_t cur = 0;
            for (const LetRange& r : assignment_)
            {
                if (r.start == r.end) continue;
                for (int32_t i = cur; i < r.start; ++i)
                    fromGlobal.push_back(i);
                cur = r.end;
            }
            for (int32_t i = cur; i < L; ++i)
                fromGlobal.push_back(i);
        }
        stats_.leavesFromGlobal += fromGlobal.size();
        if (!fromGlobal.empty())
        {
            LET_TRY(scratchIdx_.ensure(fromGlobal.size() * 4));
            LET_TRY(cstone_hip_upload(ctx_, scratchIdx_.p, fromGlobal.data(), fromGlobal.size() * 4));
            LET_TRY(cstone_hip_range_count(ctx_, kb, globalLeaves, numGlobalLeaves, globalCounts, leaves_.p,
                                           scratchIdx_.as<int32_t>(), int(fromGlobal.size()),
                                           leafCounts_.as<uint32_t>()));
        }
        // first upsweep with local and global data
        LET_TRY(counts_.ensure(size_t(M) * 4));
        LET_TRY(cstone_hip_scatter(ctx_, 4, lti_.as<uint32_t>() + I, size_t(L), leafCounts_.p, counts_.p));
        LET_TRY(cstone_hip_upsweep_sum_bounded(ctx_, maxLevel + 2, levelRange_.as<int32_t>(), child_.as<int32_t>(),
                                               counts_.as<uint32_t>(), levelBound_));
        // counts of the peers' regions from their owners (peerExchange -> exchangeTreeletGeneral, exchange_focus.hpp:289-344)
        if (P_ > 1)
        {
            std::vector<uint64_t> sendCounts(P_, 0), recvCounts(P_, 0);
            uint64_t sendTotal = 0, recvTotal = 0;
            for (int peer : peers_)
            {
                sendCounts[peer] = tlCount_[peer];
                recvCounts[peer] = uint64_t(assignment_[peer].count());
                sendTotal += sendCounts[peer];
                recvTotal += recvCounts[peer];
            }
            LET_TRY(sendBuf_.ensure(std::max<uint64_t>(sendTotal, 1) * 4));
            LET_TRY(recvBuf_.ensure(std::max<uint64_t>(recvTotal, 1) * 4));
            // (the node indices of the peers' treelets lie in rank order, like the send segments)
            if (sendTotal)
                LET_TRY(cstone_hip_gather(ctx_, 4, treeletIdx_.as<uint32_t>(), size_t(sendTotal), counts_.p, sendBuf_.p));
            LET_TRY(allToAll(sendBuf_.p, sendCounts, 4, recvBuf_.p, recvCounts));
            uint64_t at = 0;
            for (int peer : peers_)
            {
                if (recvCounts[peer])
                    LET_TRY(cstone_hip_scatter(ctx_, 4, lti_.as<uint32_t>() + I + assignment_[peer].start,
                                               size_t(recvCounts[peer]), recvBuf_.as<uint32_t>() + at, counts_.p));
                at += recvCounts[peer];
            }
            // second upsweep with the peer data present
            LET_TRY(cstone_hip_upsweep_sum(ctx_, maxLevel + 2, levelRange_.as<int32_t>(), child_.as<int32_t>(),
                                           counts_.as<uint32_t>()));
            // (one rank: no leaf count came from anybody else, leafCounts_ is what this gather would bring back)
            LET_TRY(cstone_hip_gather(ctx_, 4, lti_.as<uint32_t>() + I, size_t(L), counts_.p, leafCounts_.p));
        }
        haveCounts_ = haveLeafCounts_ = true;
        return CSTONE_OK;
    }

    //! sum of one value per rank
    int allReduceSum(uint32_t mine, uint32_t* sum)
    {
        *sum = mine;
        if (P_ == 1) return CSTONE_OK;
        LET_TRY(rowBuf_.ensure(64));
        LET_TRY(cstone_hip_upload(ctx_, rowBuf_.p, &mine, 4));
        LET_TRY(commCall(comm_.all_reduce(comm_.user, rowBuf_.p, 1, 1, 0), "all_reduce (flag)"));
        return readBack(rowBuf_.as<uint32_t>(), sum);
    }

    /*! updateMacs (octree_focus_mpi.hpp:505-526): MAC radii from the expansion centres in centers_ (setMacRadius) and the
     *  marks of the nodes that fail the MAC against my focus */
    int updateMacs(const K* assignment, float invTheta)
    {
        const int M = numNodesOf(L_);
        LET_TRY(cstone_hip_set_mac(ctx_, curve_, kb, rb, prefixes_.p, M, centers_.p, invTheta, &box_));
        LET_TRY(macs_.ensure(size_t(M)));
        LET_TRY(cstone_hip_memset(ctx_, macs_.p, 0, size_t(M)));
        std::vector<int64_t> idx;
        LET_TRY(lowerBounds(leaves_.as<K>(), size_t(L_), {assignment[rank_], assignment[rank_ + 1]}, idx));
        const int fStart = int(idx[0]), fEnd = int(idx[1]);
        if (fEnd > fStart)
            LET_TRY(cstone_hip_mark_macs(ctx_, curve_, kb, rb, prefixes_.p, child_.as<int32_t>(), centers_.p, &box_,
                                         leaves_.as<K>() + fStart, fEnd - fStart, 0, macs_.as<char>()));
        haveMacs_ = true;
        return CSTONE_OK;
    }

    //! addMacs (:601-610): a leaf whose node fails the MAC becomes a halo leaf
    int addMacs() { return cstone_hip_add_macs(ctx_, macs_.as<char>(), lti_.as<int32_t>() + numInternalOf(L_), L_, flags_.as<int32_t>()); }

    /*! updateCenters (octree_focus_mpi.hpp:369-449): mass centres of my leaves from the particles, upsweep, the nodes
     *  that are larger than any rank's domain through the global tree (globalCenterExchange), the peers' regions from
     *  the peers, upsweep */
    int updateCenters(const T* x, const T* y, const T* z, const void* m, int massBits, const K* globalLeaves,
                      const K* globalLeavesHost, int numGlobalLeaves)
    {
        const int L = L_, M = numNodesOf(L), I = numInternalOf(L);
        const int first = assignment_[rank_].start, last = assignment_[rank_].end;
        LET_TRY(centers_.ensure(size_t(M) * 4 * sizeof(T)));
        // temporary pre-halo layout: offsets of MY leaves among my particles, zero-sized ranges everywhere else
        LET_TRY(scratchIdx2_.ensure(size_t(L + 2) * 4));
        uint32_t* lay = scratchIdx2_.as<uint32_t>();
        LET_TRY(cstone_hip_memset(ctx_, lay, 0, size_t(first + 1) * 4));
        if (last > first)
            LET_TRY(cstone_hip_inclusive_scan_u32(ctx_, leafCounts_.as<uint32_t>() + first, lay + first + 1,
                                                  size_t(last - first)));
        if (L > last)
        {
            // (leaves behind mine hold nothing of mine: they all start where my particles end)
            const uint32_t total = uint32_t(numKeys_);
            LET_TRY(cstone_hip_fill(ctx_, 4, lay + last + 1, size_t(L - last), &total));
        }
        LET_TRY(cstone_hip_leaf_source_centers(ctx_, rb, massBits, rb, x, y, z, m, lti_.as<int32_t>() + I, L, lay,
                                               centers_.p));
        LET_TRY(readBack(levelRange_.as<int32_t>(), levelHost_, size_t(maxLevel) + 2));
        LET_TRY(cstone_hip_upsweep_centers(ctx_, rb, maxLevel, levelHost_, child_.as<int32_t>(), centers_.p));
        if (P_ > 1)
        {
            LET_TRY(globalCenterExchange(globalLeaves, globalLeavesHost, numGlobalLeaves));
            // the peers' regions from their owners (peerExchange of SourceCenterType, :436-447), like the counts
            std::vector<uint64_t> sendCounts(P_, 0), recvCounts(P_, 0);
            uint64_t sendTotal = 0, recvTotal = 0;
            for (int peer : peers_)
            {
                sendCounts[peer] = tlCount_[peer];
                recvCounts[peer] = uint64_t(assignment_[peer].count());
                sendTotal += sendCounts[peer];
                recvTotal += recvCounts[peer];
            }
            const int e = int(4 * sizeof(T));
            LET_TRY(sendBuf_.ensure(std::max<uint64_t>(sendTotal, 1) * e));
            LET_TRY(recvBuf_.ensure(std::max<uint64_t>(recvTotal, 1) * e));
            if (sendTotal)
                LET_TRY(cstone_hip_gather(ctx_, e, treeletIdx_.as<uint32_t>(), size_t(sendTotal), centers_.p, sendBuf_.p));
            LET_TRY(allToAll(sendBuf_.p, sendCounts, e, recvBuf_.p, recvCounts));
            uint64_t at = 0;
            for (int peer : peers_)
            {
                if (recvCounts[peer])
                    LET_TRY(cstone_hip_scatter(ctx_, e, lti_.as<uint32_t>() + I + assignment_[peer].start,
                                               size_t(recvCounts[peer]), recvBuf_.template as<char>() + at * e, centers_.p));
                at += recvCounts[peer];
            }
            LET_TRY(cstone_hip_upsweep_centers(ctx_, rb, maxLevel, levelHost_, child_.as<int32_t>(), centers_.p));
        }
        return CSTONE_OK;
    }

    /*! globalFocusExchange for the expansion centres (octree_focus_mpi.hpp:288-366,763-784): every rank contributes the
     *  centres of the global leaves inside its focus (populateGlobal), the segments are gathered (gatherGlobalLeaves; an
     *  all-gather of segments padded to the longest), the global tree is swept up, and the leaves of my focus tree that
     *  neither I nor a peer own take their centres from the global nodes of the same key range (extractGlobal) */
    int globalCenterExchange(const K* globalLeaves, const K* gl, int GL)
    {
        const int L = L_, I = numInternalOf(L);
        const int e = int(4 * sizeof(T));
        const int GM = numNodesOf(GL), GI = numInternalOf(GL);
        // findNodeAbove(globalLeaves, key) for the assignment of the last tree update (host copy of the global leaves)
        auto above = [&](K key) { return int(std::lower_bound(gl, gl + GL + 1, key) - gl); };
        std::vector<int> displ(P_ + 1);
        for (int r = 0; r < P_; ++r)
            displ[r] = above(globAssignment_[r]);
        displ[P_] = GL;
        int longest = 1;
        for (int r = 0; r < P_; ++r)
            longest = std::max(longest, displ[r + 1] - displ[r]);
        const int gFirst = above(prevFocusStart_), gLast = above(prevFocusEnd_);
        const int mine   = gLast - gFirst;
        if (mine != displ[rank_ + 1] - displ[rank_] || gFirst != displ[rank_])
            return fail(CSTONE_E_INTERNAL, "global centre exchange: my global leaves [%d, %d), the assignment says [%d, %d)",
                        gFirst, gLast, displ[rank_], displ[rank_ + 1]);
        // populateGlobal: the focus node of every global leaf of mine
        LET_TRY(gSeg_.ensure(size_t(longest) * e * (size_t(P_) + 1)));
        LET_TRY(gCenters_.ensure(size_t(std::max(GM, GL) + 1) * e * 2));
        char* seg     = gSeg_.as<char>();
        char* all     = seg + size_t(longest) * e;
        char* leafCen = gCenters_.as<char>();                    // [GL] in leaf order
        char* nodeCen = leafCen + size_t(GL + 1) * e;            // [GM] in node order
        LET_TRY(cstone_hip_memset(ctx_, seg, 0, size_t(longest) * e));
        if (mine > 0)
        {
            LET_TRY(scratchIdx_.ensure(size_t(mine + 1) * 4));
            LET_TRY(cstone_hip_locate_nodes(ctx_, kb, globalLeaves + gFirst, size_t(mine) + 1, prefixes_.p,
                                            levelRange_.as<int32_t>(), scratchIdx_.as<int32_t>()));
            LET_TRY(cstone_hip_gather(ctx_, e, scratchIdx_.as<uint32_t>(), size_t(mine), centers_.p, seg));
        }
        LET_TRY(commCall(comm_.all_gather(comm_.user, seg, all, size_t(longest) * e), "all_gather (global leaf centres)"));
        for (int r = 0; r < P_; ++r)
            if (displ[r + 1] > displ[r])
                LET_TRY(cstone_hip_memcpy_d2d(ctx_, leafCen + size_t(displ[r]) * e, all + size_t(r) * longest * e,
                                              size_t(displ[r + 1] - displ[r]) * e));
        // the global tree: leaf quantities to node order, upsweep (the linked octree of the global tree is findPeers')
        LET_TRY(cstone_hip_memset(ctx_, nodeCen, 0, size_t(GM) * e));
        LET_TRY(cstone_hip_scatter(ctx_, e, gLti_.as<uint32_t>() + GI, size_t(GL), leafCen, nodeCen));
        LET_TRY(cstone_hip_upsweep_centers(ctx_, rb, maxLevel, gLevelHost_, gChild_.as<int32_t>(), nodeCen));
        // extractGlobal: the leaves outside my range and the peers' ranges, run by run
        int32_t cur = 0;
        auto extractRun = [&](int32_t a, int32_t b) -> int
        {
            if (b <= a) return CSTONE_OK;
            LET_TRY(scratchIdx_.ensure(size_t(b - a + 1) * 4));
            LET_TRY(cstone_hip_locate_nodes(ctx_, kb, leaves_.as<K>() + a, size_t(b - a) + 1, gPrefixes_.p,
                                            gLevelRange_.as<int32_t>(), scratchIdx_.as<int32_t>()));
            return cstone_hip_gather_scatter(ctx_, e, scratchIdx_.as<uint32_t>(), lti_.as<uint32_t>() + I + a, size_t(b - a),
                                             nodeCen, centers_.p);
        };
        for (const LetRange& r : assignment_)
        {
            if (r.start == r.end) continue;
            LET_TRY(extractRun(cur, r.start));
            cur = r.end;
        }
        return extractRun(cur, L);
    }

    //! converge (octree_focus_mpi.hpp:535-553): update until the tree of EVERY rank has stopped changing
    int converge(const cstone_box& box, const K* keys, size_t numKeys, const K* assignment, const K* globalLeaves,
                 const uint32_t* globalCounts, int numGlobalLeaves, float invThetaEff)
    {
        int guard = 0;
        while (true)
        {
            LET_TRY(updateMinMac(assignment, invThetaEff));
            bool converged = false;
            LET_TRY(updateTree(assignment, box, &converged));
            LET_TRY(updateCounts(keys, numKeys, globalLeaves, globalCounts, numGlobalLeaves));
            // (updateGeoCenters: done at the end of updateTree, the tree has not changed since)
            uint32_t sum = converged ? 1u : 0u;
            if (P_ > 1)
            {
                LET_TRY(rowBuf_.ensure(64));
                LET_TRY(cstone_hip_upload(ctx_, rowBuf_.p, &sum, 4));
                LET_TRY(commCall(comm_.all_reduce(comm_.user, rowBuf_.p, 1, 1, 0), "all_reduce (converged)"));
                LET_TRY(readBack(rowBuf_.as<uint32_t>(), &sum));
            }
            ++stats_.convergeSteps;
            if (int(sum) == P_) return CSTONE_OK;
            if (++guard > 128) return fail(CSTONE_E_INTERNAL, "focus tree does not converge");
        }
    }

    // ---- Halos::discover (halos.hpp:128-189) -----------------------------------------------------------------------------
    int discoverHalos(const cstone_box& box, const T* h, float searchExt)
    {
        const int L = L_, first = assignment_[rank_].start, last = assignment_[rank_].end;
        if (first < 0 || last > L || last < first)
            return fail(CSTONE_E_INTERNAL, "focus tree: bad leaf range [%d, %d) of %d", first, last, L);
        LET_TRY(layout_.ensure(size_t(L + 2) * 4));
        LET_TRY(radii_.ensure(size_t(L) * 4));
        LET_TRY(flags_.ensure(size_t(L) * 4));
        // layout[0 .. last - first] = offsets of the assigned leaves among the assigned particles
        LET_TRY(cstone_hip_offsets_from_counts_u32(ctx_, leafCounts_.as<uint32_t>() + first, layout_.as<uint32_t>(),
                                                   size_t(last - first)));
        LET_TRY(cstone_hip_halo_radii(ctx_, rb, h, layout_.as<uint32_t>(), first, last, L, searchExt, radii_.as<float>()));
        LET_TRY(cstone_hip_memset(ctx_, flags_.p, 0, size_t(L) * 4));
        if (last > first)
            LET_TRY(cstone_hip_find_halos(ctx_, curve_, kb, rb, prefixes_.p, child_.as<int32_t>(), itl_.as<int32_t>(),
                                          leaves_.p, radii_.as<float>(), &box, first, last, flags_.as<int32_t>()));
        return CSTONE_OK;
    }

    /*! Halos::computeLayout (halos.hpp:205-222): offsets of every leaf in the particle buffers (computeNodeLayout,
     *  R/domain/layout.hpp:150-165), the key ranges I want from every peer (exchangeRequestKeys,
     *  R/domain/exchange_keys.hpp:63-119) and the index ranges the peers want from me, the ranges the halos arrive in
     *  (computeHaloRecvList, layout.hpp:175-190) */
    int computeLayout(int externalFailure, bool* unmatchedOut = nullptr)
    {
        if (unmatchedOut) *unmatchedOut = false;
        const int L = L_, first = assignment_[rank_].start, last = assignment_[rank_].end;
        LET_TRY(cstone_hip_node_layout(ctx_, leafCounts_.as<uint32_t>(), flags_.as<int32_t>(), first, last, L,
                                       layout_.as<uint32_t>()));
        // the halo key ranges I request, peer by peer
        std::vector<int32_t> ranges(2 * size_t(P_), 0);
        for (int peer : peers_)
            ranges[2 * peer] = assignment_[peer].start, ranges[2 * peer + 1] = assignment_[peer].end;
        std::vector<uint32_t> pairCounts(P_, 0);
        uint32_t unmatched = 0;
        LET_TRY(scratchKeys_.ensure(size_t(L + 2) * sizeof(K))); // at most (L + 1) / 2 runs of flagged leaves, 2 keys each
        // counts of everybody (+ a status word: a halo cell that no peer owns fails the sync on every rank, checkHalos
        // halos.hpp:59-95).  This rank's row is made on the device (cstone_hip_halo_request_rows) and goes straight into
        // the all-gather: ONE read-back for everybody's rows instead of one for mine and one for theirs
        std::vector<uint64_t> row(P_ + 1, 0), matrix;
        if (P_ > 1)
        {
            uint64_t *send = nullptr, *recv = nullptr;
            LET_TRY(rowBuffers<uint64_t>(size_t(P_) + 1, &send, &recv));
            LET_TRY(cstone_hip_halo_request_rows(ctx_, kb, leaves_.p, flags_.as<int32_t>(), L, first, last, ranges.data(), P_,
                                                 scratchKeys_.p, send, externalFailure));
            LET_TRY(gatherRowsDev<uint64_t>(size_t(P_) + 1, matrix));
            for (int p = 0; p <= P_; ++p)
                row[p] = matrix[size_t(rank_) * (P_ + 1) + p];
        }
        else
        {
            // (one rank: every leaf is mine, nobody to ask -- and no read-back for the answer)
            row[P_] = externalFailure ? 2 : 0;
            matrix  = row;
        }
        (void)pairCounts, (void)unmatched;
        for (int p = 0; p < P_; ++p)
        {
            const uint64_t st = matrix[size_t(p) * (P_ + 1) + P_];
            if (st == 1 && unmatchedOut)
            {
                // syncGrav: the caller loosens its MAC and tries again (on every rank: they all read the same rows)
                *unmatchedOut = true;
                return CSTONE_OK;
            }
            if (st == 1)
                return fail(CSTONE_E_INTERNAL,
                            "halo discovery: rank %d found halo cells that belong to none of its peers (the sync was "
                            "abandoned on every rank)",
                            p);
            if (st != 0)
                return fail(CSTONE_E_INTERNAL, "rank %d reported a failure (the sync was abandoned on every rank)", p);
        }
        std::vector<uint64_t> sendCounts(P_, 0), recvCounts(P_, 0);
        uint64_t recvKeys = 0;
        for (int p = 0; p < P_; ++p)
        {
            sendCounts[p] = row[p];
            recvCounts[p] = matrix[size_t(p) * (P_ + 1) + rank_];
            recvKeys += recvCounts[p];
        }
        LET_TRY(recvBuf_.ensure(std::max<uint64_t>(recvKeys, 1) * sizeof(K)));
        LET_TRY(allToAll(scratchKeys_.p, sendCounts, int(sizeof(K)), recvBuf_.p, recvCounts));

        // what the peers want from me: index ranges of my particle buffers, peer after peer
        numSendRanges_ = int(recvKeys / 2);
        LET_TRY(rangeOffsets_.ensure(size_t(numSendRanges_ + 1) * 4));
        LET_TRY(rangeScan_.ensure(size_t(numSendRanges_ + 2) * 4));
        haloSendCounts_.assign(P_, 0);
        sendTotal_ = 0;
        if (numSendRanges_)
            LET_TRY(cstone_hip_ranges_from_keys(ctx_, kb, leaves_.p, L, layout_.as<uint32_t>(), recvBuf_.p,
                                                size_t(numSendRanges_), rangeOffsets_.as<uint32_t>(),
                                                rangeScan_.as<uint32_t>()));
        // ONE read-back for: particles per peer (differences of the range scan at the peers' first ranges), where the
        // halos arrive and the extent of the buffers (the layout at the assignment boundaries)
        {
            const size_t nA = size_t(P_) + 1, nB = 2 * size_t(P_) + 1;
            std::vector<uint32_t> map(nA + nB, 0), val(nA + nB, 0);
            for (int p = 0; p < P_; ++p)
                map[p + 1] = map[p] + uint32_t(recvCounts[p] / 2);
            for (int p = 0; p < P_; ++p)
            {
                map[nA + 2 * p]     = uint32_t(assignment_[p].start);
                map[nA + 2 * p + 1] = uint32_t(assignment_[p].end);
            }
            map[nA + 2 * P_] = uint32_t(L);
            LET_TRY(scratchIdx_.ensure(map.size() * 8 + (size_t(maxLevel) + 2) * 4));
            uint32_t* dmap = scratchIdx_.as<uint32_t>();
            uint32_t* dval = dmap + map.size();
            LET_TRY(cstone_hip_upload(ctx_, dmap, map.data(), map.size() * 4));
            // ... and the level ranges of the tree: the exact depth for the bounds of the next update (treeChanged)
            const size_t nC = size_t(maxLevel) + 2;
            LET_TRY(cstone_hip_gather_tables_u32(ctx_, dmap, numSendRanges_ ? rangeScan_.as<uint32_t>() : nullptr, nA,
                                                 layout_.as<uint32_t>(), nB, levelRange_.as<uint32_t>(), nC, dval));
            val.resize(nA + nB + nC);
            LET_TRY(readBack(dval, val.data(), val.size()));
            levelRangeHost_.assign(val.begin() + nA + nB, val.end());
            int deepest = 0;
            for (int l = 0; l <= maxLevel; ++l)
                if (levelRangeHost_[l + 1] > levelRangeHost_[l]) deepest = l;
            if (deepest > levelBound_)
                return fail(CSTONE_E_INTERNAL, "focus tree: level %d exists, the bound was %d", deepest, levelBound_);
            levelBound_ = deepest;
            for (int p = 0; p < P_ && numSendRanges_; ++p)
            {
                haloSendCounts_[p] = val[p + 1] - val[p];
                sendTotal_ += haloSendCounts_[p];
            }
            const uint32_t* lay = val.data() + nA;
            haloRecvCounts_.assign(P_, 0);
            haloRecvOffsets_.assign(P_, 0);
            recvTotal_ = 0;
            for (int peer : peers_)
            {
                haloRecvOffsets_[peer] = lay[2 * peer];
                haloRecvCounts_[peer]  = lay[2 * peer + 1] - lay[2 * peer];
                recvTotal_ += haloRecvCounts_[peer];
            }
            particleStart_ = lay[2 * rank_];
            particleEnd_   = lay[2 * rank_ + 1];
            particleTotal_ = lay[2 * P_];
        }
        return CSTONE_OK;
    }

    cstone_hip_ctx* ctx_;
    int curve_, rank_, P_;
    uint32_t bucket_;
    float theta_;
    cstone_hip_comm_ops comm_;

    bool firstCall_ = true;
    cstone_box box_{}; // the box of the last updateTree (octree_focus_mpi.hpp:689)
    K prevFocusStart_ = 0, prevFocusEnd_ = 0;
    bool updatedOnce_ = false; // updateTree has run (on every rank: a collective)
    bool haveMacs_ = true, haveCounts_ = true, haveLeafCounts_ = false; // rebalanceStatus_ (:669-677, :733)
    Stats stats_;

    int L_ = 0; // leaves of the focus tree
    size_t numKeys_ = 0; // assigned particles of the last updateCounts
    int levelBound_ = maxLevel;          // no leaf is deeper than this (treeChanged)
    uint64_t treeVersion_ = 0, memoVersion_ = 0;
    std::map<K, std::pair<size_t, int64_t>> memo_; // lowerBounds in the current leaf array: key -> (keys searched, index)
    std::vector<int32_t> levelRangeHost_;
    LetBuf leaves_, leavesNew_, prefixes_, child_, parents_, levelRange_, itl_, lti_;
    LetBuf counts_, leafCounts_, macs_, centers_, geoCenters_, geoSizes_;
    LetBuf opsAll_, ops_, scratchKeys_, scratchKeys2_, scratchIdx_, scratchIdx2_, scratchU64_;
    LetBuf gPrefixes_, gChild_, gParents_, gLevelRange_, gItl_, gLti_; // linked octree of the global tree (peer search)
    LetBuf gSeg_, gCenters_;                  // global centre exchange: my segment + everybody's, leaf and node centres
    int32_t levelHost_[maxLevel + 2]  = {0};  // level ranges of the focus tree / the global tree on the host (upsweeps)
    int32_t gLevelHost_[maxLevel + 2] = {0};

    std::vector<int> peers_;
    std::vector<K> peersAssignment_; // what the peers were computed for
    cstone_box peersBox_{};
    int peersGlobalLeaves_ = 0;
    bool peersValid_       = false;
    std::vector<LetRange> assignment_; // leaf index range of every peer's (and my) key range
    std::vector<K> globAssignment_;    // the key ranges of the last updateTree

    // treelets: the peers' views of my range, pruned to the keys I have; node counts, key offsets, node-index offsets
    LetBuf treelets_, treeletIdx_, tlFlags_, tlScan_, sendBuf_, recvBuf_, rowBuf_;
    std::vector<uint64_t> tlCount_, tlOffset_, tlIdxOffset_;

    // halo layout
    LetBuf layout_, flags_, radii_, rangeOffsets_, rangeScan_, haloSend_, haloRecv_;
    int numSendRanges_ = 0;
    std::vector<uint64_t> haloSendCounts_, haloRecvCounts_, haloRecvOffsets_;
    uint64_t sendTotal_ = 0, recvTotal_ = 0;
    uint32_t particleStart_ = 0, particleEnd_ = 0, particleTotal_ = 0;
};

} // namespace cship
