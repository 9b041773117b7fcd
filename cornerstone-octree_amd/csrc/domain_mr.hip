// cstone::Domain::sync on SEVERAL ranks, one process per GPU (R/domain/domain.hpp:196-243).  Decomposition logic and
// all device work live here; the three collectives come from the host application through cstone_hip_comm_ops.
//
//   C1  global box            makeGlobalBox (R/sfc/box_mpi.hpp:85-121): local min/max, all_reduce(MIN) of (lo, -hi)
//   C2  global tree           GlobalAssignment (R/domain/assignment.hpp:42-103): spanning tree of numRanks segments,
//                             updateOctreeGlobal = rebalance + recount + all_reduce(SUM) (R/tree/update_mpi.hpp:71-94)
//       assignment            uniformBins / makeSfcAssignment / limitBoundaryShifts (R/domain/domaindecomp.hpp:50-172)
//   C3  particle exchange     createSendRanges (:218-230) on the sorted keys; the fields are NOT reordered first
//                             (assignment.hpp:121-127): leaving particles are packed through the ordering as
//                             (x,y,z,h) rows, ONE all_to_all_v; newcomers are sorted among themselves and merged into
//                             the kept, already sorted range instead of the second full sort of assignment.hpp:156
//   C4  halo discovery        owner side: every rank exports the radius-dilated boxes of its boundary leaves
//                             (all_gather), each owner marks the leaves of its own tree that a foreign box touches
//   C5  halo exchange         ONE all_to_all_v of packed rows
// Result: [halos of lower ranks | assigned, SFC sorted | halos of higher ranks] in domain-owned arrays.
// Box, SFC ranges, global tree and the assigned particles are bit-identical to the reference Domain under MPI
// (tests/golden/ref_domain_mpi_*.npz); the halo set is compared there as well.
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <string>
#include <memory>
#include <numeric>
#include <vector>

#include "ctx.hpp"
#include "devbuf.hpp"
#include "device_keys.hpp"
#include "host_tree.hpp"
#include "let.hpp"
#include "resort.hpp"
#include "scan.hpp"

namespace cship
{

namespace
{

// ---- host-side decomposition rules -------------------------------------------------------------------------------

//! smallest l with 8^l >= n (R/sfc/common.hpp:134-142)
inline unsigned log8ceilHost(uint64_t n)
{
    unsigned l = 0;
    uint64_t p = 1;
    while (p < n)
    {
        p *= 8;
        ++l;
    }
    return l;
}

//! computeSpanningTree(initialDomainSplits(numRanks, level)) (R/tree/csarray.hpp:508-531, domaindecomp.hpp:242-255):
//! per segment the canonical cover by maximal aligned power-of-8 nodes (spanSfcRange, R/sfc/common.hpp:376-438)
template<class K>
void appendCover(std::vector<K>& out, uint64_t a, uint64_t b)
{
    const uint64_t end = uint64_t(endKey<K>());
    while (a < b)
    {
        uint64_t size = end;
        while (size > 1 && (a % size != 0 || size > b - a))
            size /= 8;
        out.push_back(K(a));
        a += size;
    }
}

template<class K>
std::vector<K> initialGlobalTree(int numRanks)
{
    const uint64_t end   = uint64_t(endKey<K>());
    const unsigned level = log8ceilHost(100ull * uint64_t(numRanks));
    const unsigned shift = 3 * (maxLevel<K>() - level);
    const uint64_t delta = end / uint64_t(numRanks);
    std::vector<uint64_t> splits(numRanks + 1, 0);
    for (int i = 1; i < numRanks; ++i)
        splits[i] = ((uint64_t(i) * delta) >> shift) << shift;
    splits[numRanks] = end;
    std::vector<K> tree;
    for (int i = 0; i < numRanks; ++i)
        appendCover(tree, splits[i], splits[i + 1]);
    tree.push_back(K(end));
    return tree;
}

//! uniformBins (R/domain/domaindecomp.hpp:50-75)
inline std::vector<int> uniformBinsHost(const std::vector<uint32_t>& counts, int numBins)
{
    std::vector<uint64_t> scan(counts.size() + 1, 0);
    for (size_t i = 0; i < counts.size(); ++i)
        scan[i + 1] = scan[i] + counts[i];
    double binCount = double(scan.back()) / numBins;
    std::vector<int> bins(numBins + 1, 0);
    bins[numBins] = int(counts.size());
    for (int i = 1; i < numBins; ++i)
    {
        uint64_t target = uint64_t(i * binCount);
        bins[i]         = int(std::lower_bound(scan.begin(), scan.end(), target) - scan.begin());
    }
    return bins;
}

// ---- device helpers ------------------------------------------------------------------------------------------------

//! for each of two keys: the index of the leaf that contains it and that leaf's start and end key (one thread per key);
//! out[3 q] = index, out[3 q + 1] = start, out[3 q + 2] = end.  A key at or behind the end of the curve gives index L.
template<class K>
__global__ void containingLeavesKernel(const K* __restrict__ tree, int numLeaves, K key0, K key1,
                                       uint64_t* __restrict__ out)
{
    const int q = threadIdx.x;
    if (q > 1) return;
    const K key = q == 0 ? key0 : key1;
    int lo = 0, hi = numLeaves + 1; // first entry of tree[0 .. L] greater than key
    while (lo < hi)
    {
        int mid = (lo + hi) >> 1;
        if (tree[mid] <= key) lo = mid + 1;
        else hi = mid;
    }
    const int idx  = lo - 1; // tree[0] = 0 <= key: idx >= 0; key >= tree[L]: idx = L
    out[3 * q]     = uint64_t(idx);
    out[3 * q + 1] = uint64_t(tree[idx]);
    out[3 * q + 2] = idx < numLeaves ? uint64_t(tree[idx + 1]) : uint64_t(tree[idx]);
}


//! counts[i] = max(counts[i], local[i]): a wrapped 32-bit sum over the ranks cannot make a full node look empty
//! (R/tree/update_mpi.hpp:60-64)
__global__ __launch_bounds__(256) void maxWithLocalKernel(uint32_t* __restrict__ counts,
                                                          const uint32_t* __restrict__ local, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) counts[i] = max(counts[i], local[i]);
}

//! rows[i] = {x, y, z, h}[idx[i]]: the fields of a particle travel as one record
template<class T>
__global__ __launch_bounds__(256) void packRowsKernel(const uint32_t* __restrict__ idx, size_t m,
                                                      const T* __restrict__ x, const T* __restrict__ y,
                                                      const T* __restrict__ z, const T* __restrict__ h,
                                                      T* __restrict__ rows)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= m) return;
    uint32_t j      = idx[i];
    rows[4 * i]     = x[j];
    rows[4 * i + 1] = y[j];
    rows[4 * i + 2] = z[j];
    rows[4 * i + 3] = h[j];
}

template<class T>
__global__ __launch_bounds__(256) void unpackRowsKernel(const T* __restrict__ rows, size_t m, T* __restrict__ x,
                                                        T* __restrict__ y, T* __restrict__ z, T* __restrict__ h)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= m) return;
    x[i] = rows[4 * i];
    y[i] = rows[4 * i + 1];
    z[i] = rows[4 * i + 2];
    h[i] = rows[4 * i + 3];
}

/*! the cut points of an assignment in the sorted keys and what follows from them, in one launch (it used to be a search,
 *  a difference and an upload): cut[p] = first key >= assignment[p] for p <= P, send[p] = cut[p + 1] - cut[p] for
 *  p < P (this rank's row of the count matrix) and send[P] = the rank's status word.  A lane searches both ends of its
 *  range itself: nothing to wait for */
template<class K>
__global__ void cutPointsKernel(const K* __restrict__ keys, size_t n, const K* __restrict__ assignment, int P,
                                uint64_t* __restrict__ cut, uint64_t* __restrict__ send, uint64_t status)
{
    auto firstNotBelow = [&](K v)
    {
        size_t lo = 0, len = n;
        while (len > 0)
        {
            size_t half = len / 2;
            if (keys[lo + half] < v) lo += half + 1, len -= half + 1;
            else len = half;
        }
        return uint64_t(lo);
    };
    int p = blockIdx.x * 64 + threadIdx.x;
    if (p > P) return;
    uint64_t mine = firstNotBelow(assignment[p]);
    cut[p]        = mine;
    if (p < P) send[p] = firstNotBelow(assignment[p + 1]) - mine;
    else send[P] = status;
}

//! keys (already sorted) and x, y, z, h (from their input slots order[i]) of the kept particles to their final slots
//! (pos[i], or i when pos is null): the two index maps are read once for the columns of a launch.  WHICH = 0: all five
//! columns; 1: keys and h (what the locally essential tree and the halo discovery need); 2: x, y, z (nobody reads them
//! before the halo exchange: they go out on the context's second stream, next to the tree update)
constexpr int PLACE_PER = 4; // elements per lane, strided by the workgroup: all index loads, then all field loads in flight
template<class K, class T, int WHICH>
__global__ __launch_bounds__(256) void placeColumnsKernel(const uint32_t* __restrict__ order,
                                                          const uint32_t* __restrict__ pos, size_t m,
                                                          const K* __restrict__ keys, const T* __restrict__ x,
                                                          const T* __restrict__ y, const T* __restrict__ z,
                                                          const T* __restrict__ h, K* __restrict__ dk,
                                                          T* __restrict__ dx, T* __restrict__ dy, T* __restrict__ dz,
                                                          T* __restrict__ dh)
{
    // (one element per lane -- three dependent loads in flight -- ran at 0.54 of the HBM peak where gatherMultiKernel, four
    //  elements per lane, reaches 0.71 on the same bytes)
    const size_t base = size_t(blockIdx.x) * (256 * PLACE_PER) + threadIdx.x;
    uint32_t s[PLACE_PER];
    size_t d[PLACE_PER];
    bool ok[PLACE_PER];
#pragma unroll
    for (int k = 0; k < PLACE_PER; ++k)
    {
        const size_t i = base + size_t(k) * 256;
        ok[k]          = i < m;
        s[k]           = ok[k] ? order[i] : 0u;
        d[k]           = (pos && ok[k]) ? size_t(pos[i]) : i;
    }
    if constexpr (WHICH != 2)
    {
        K vk[PLACE_PER];
        T vh[PLACE_PER];
#pragma unroll
        for (int k = 0; k < PLACE_PER; ++k)
            if (ok[k]) vk[k] = keys[base + size_t(k) * 256], vh[k] = h[s[k]]; // the kept keys are already in sorted order
#pragma unroll
        for (int k = 0; k < PLACE_PER; ++k)
            if (ok[k]) dk[d[k]] = vk[k], dh[d[k]] = vh[k];
    }
    if constexpr (WHICH != 1)
    {
        T vx[PLACE_PER], vy[PLACE_PER], vz[PLACE_PER];
#pragma unroll
        for (int k = 0; k < PLACE_PER; ++k)
            if (ok[k]) vx[k] = x[s[k]], vy[k] = y[s[k]], vz[k] = z[s[k]];
#pragma unroll
        for (int k = 0; k < PLACE_PER; ++k)
            if (ok[k]) dx[d[k]] = vx[k], dy[d[k]] = vy[k], dz[d[k]] = vz[k];
    }
}

__global__ __launch_bounds__(256) void boxFlagsKernel(const int32_t* __restrict__ boxes, int n,
                                                      uint32_t* __restrict__ flags)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = boxes[8 * i + 6] != 0;
}

//! keeps the records with flag != 0; scan = exclusive scan of the flags
__global__ __launch_bounds__(256) void compactBoxesKernel(const int32_t* __restrict__ boxes,
                                                          const uint32_t* __restrict__ scan, int n, int owner,
                                                          int32_t* __restrict__ out)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || boxes[8 * i + 6] == 0) return;
    const int4* src = reinterpret_cast<const int4*>(boxes + 8 * size_t(i));
    int4* dst       = reinterpret_cast<int4*>(out + 8 * size_t(scan[i]));
    int4 hi         = src[1];
    hi.w            = owner; // record[7]: who exports the box (find_overlaps marks bit `owner`)
    dst[0]          = src[0];
    dst[1]          = hi;
}

//! cnt[q * nLocal + (i - first)] = particles of leaf i if peer q's boxes touch it (bit `peer` of flags), else 0;
//! q enumerates the peers in rank order without the own rank
__global__ __launch_bounds__(256) void peerCountsKernel(const int32_t* __restrict__ flags,
                                                        const uint32_t* __restrict__ layout, int first, int last,
                                                        int numRanks, int rank, uint32_t* __restrict__ cnt)
{
    int k = blockIdx.x * 256 + threadIdx.x;
    int n = last - first;
    if (k >= n) return;
    uint32_t f = uint32_t(flags[first + k]);
    uint32_t c = layout[first + k + 1] - layout[first + k];
    for (int p = 0, q = 0; p < numRanks; ++p)
    {
        if (p == rank) continue;
        cnt[size_t(q) * n + k] = ((f >> p) & 1u) ? c : 0u;
        ++q;
    }
}

//! totals[q] = particles served to peer q, from the exclusive scan of cnt (grand total in *total)
//! my row of the halo count matrix, one u64 per rank (0 for myself): what the all-gather of the rows takes
__global__ void peerTotalsKernel(const uint32_t* __restrict__ scan, const uint32_t* __restrict__ total, int n,
                                 int numPeers, int rank, uint64_t* __restrict__ row)
{
    int q = threadIdx.x;
    if (q == numPeers) row[rank] = 0;
    if (q >= numPeers) return;
    uint32_t a = scan[size_t(q) * n];
    uint32_t b = (q + 1 < numPeers) ? scan[size_t(q + 1) * n] : *total;
    row[q < rank ? q : q + 1] = b - a;
}

//! particle indices of the leaves each peer needs, grouped by peer in rank order, 16 lanes per (peer, leaf)
__global__ __launch_bounds__(256) void peerFillKernel(const int32_t* __restrict__ flags,
                                                      const uint32_t* __restrict__ layout,
                                                      const uint32_t* __restrict__ scan, int first, int last,
                                                      int numRanks, int rank, uint32_t* __restrict__ out)
{
    const unsigned sub = threadIdx.x & 15u;
    const int n        = last - first;
    size_t item        = size_t(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (item >= size_t(numRanks - 1) * n) return;
    int q = int(item / n), k = int(item % n);
    int p = q < rank ? q : q + 1;
    if (!((uint32_t(flags[first + k]) >> p) & 1u)) return;
    uint32_t a = layout[first + k], b = layout[first + k + 1], o = scan[item];
    uint32_t base = layout[first];
    for (uint32_t j = a + sub; j < b; j += 16)
        out[o + (j - a)] = j - base;
}

//! cnt[i - first] = number of particles of leaf i if it is flagged, else 0
__global__ __launch_bounds__(256) void flaggedCountsKernel(const int32_t* __restrict__ flags,
                                                           const uint32_t* __restrict__ layout, int first, int last,
                                                           uint32_t* __restrict__ cnt)
{
    int i = first + blockIdx.x * 256 + threadIdx.x;
    if (i < last) cnt[i - first] = flags[i] ? layout[i + 1] - layout[i] : 0u;
}

//! particle indices (relative to the first assigned particle) of the flagged leaves, 16 lanes per leaf
__global__ __launch_bounds__(256) void fillIndicesKernel(const int32_t* __restrict__ flags,
                                                         const uint32_t* __restrict__ layout,
                                                         const uint32_t* __restrict__ scan, int first, int last,
                                                         uint32_t* __restrict__ out)
{
    const unsigned sub = threadIdx.x & 15u;
    int i              = first + blockIdx.x * 16 + int(threadIdx.x >> 4);
    if (i >= last || !flags[i]) return;
    uint32_t a = layout[i], b = layout[i + 1], o = scan[i - first];
    uint32_t base = layout[first];
    for (uint32_t j = a + sub; j < b; j += 16)
        out[o + (j - a)] = j - base;
}

//! one set of result arrays
constexpr int MAX_PROPS = 16;
struct Out
{
    DevBuf keys, x, y, z, h;
    DevBuf props[MAX_PROPS];
};

struct MrBase
{
    virtual ~MrBase()                                                                    = default;
    virtual int sync(const void* x, const void* y, const void* z, const void* h, size_t n, const void* const* props,
                     const int* propBytes, int numProps, const void* keysIn, const void* mass = nullptr,
                     int massBits = 0) = 0;
    virtual int updateExpansionCenters(const void* x, const void* y, const void* z, const void* m, int massBits) = 0;
    virtual int view(cstone_hip_domain_mr_view* out)                                     = 0;
    virtual void setHaloFactor(float f)                                                  = 0;
    virtual int exchangeHalos(void* array, int elemBytes)                                = 0;
    virtual int reapplySync(const void* in, size_t n, int elemBytes, void* out)          = 0;
    virtual int octree(cstone_hip_domain_mr_octree* out)                                 = 0;
    virtual int setHaloMode(int mode)                                                    = 0;
    virtual int setTheta(float theta)                                                    = 0;
    virtual void setSortMode(int mode)                                                   = 0;
    virtual void contextGone(bool gone)                                                  = 0;
};

template<class K, class T>
class MultiRankDomain final : public MrBase
{
    static constexpr int kb = 8 * sizeof(K), rb = 8 * sizeof(T);

public:
    MultiRankDomain(cstone_hip_ctx* ctx, int curve, int rank, int numRanks, uint32_t bucket, uint32_t bucketFocus,
                    const cstone_box& box, const cstone_hip_comm_ops& comm)
        : ctx_(ctx)
        , curve_(curve)
        , rank_(rank)
        , P_(numRanks)
        , bucket_(bucket)
        , bucketFocus_(bucketFocus)
        , box_(box)
        , comm_(comm)
    {
    }

    void setHaloFactor(float f) override { haloExt_ = f; }
    void setSortMode(int mode) override { sortMode_ = mode; }
    void contextGone(bool gone) override { ctxGone_ = gone; }
    bool mayResort() const
    {
        return sortMode_ == CSTONE_SORT_INCREMENTAL && std::getenv("CSTONE_NO_RESORT") == nullptr &&
               std::getenv("CSTONE_FULL_SORT") == nullptr;
    }
    bool allDigits() const { return sortMode_ == CSTONE_SORT_ALL_DIGITS || std::getenv("CSTONE_FULL_SORT") != nullptr; }

    //! before the first sync: how halos are found (CSTONE_MR_HALOS_LET: the reference's way, CSTONE_MR_HALOS_OWNER_SIDE)
    int setHaloMode(int mode) override
    {
        if (!firstCall_) return fail(ctx_, CSTONE_E_ARG, "domain_mr_set_halo_mode: only before the first sync");
        if (mode != CSTONE_MR_HALOS_LET && mode != CSTONE_MR_HALOS_OWNER_SIDE)
            return fail(ctx_, CSTONE_E_ARG, "domain_mr_set_halo_mode: unknown mode %d", mode);
        useLet_ = mode == CSTONE_MR_HALOS_LET;
        return CSTONE_OK;
    }

    //! before the first sync: the opening angle of the focus tree's MAC (Domain ctor, R/domain/domain.hpp:95-113)
    int setTheta(float theta) override
    {
        if (!firstCall_ || !(theta > 0.0f)) return fail(ctx_, CSTONE_E_ARG, "domain_mr_set_theta: only before the first sync");
        theta_ = theta;
        return CSTONE_OK;
    }

    /*! Domain::exchangeHalos (R/domain/domain.hpp:381-386, R/halos/halos.hpp:224-257): repeats the halo exchange of the
     *  last sync for another field.  array: device, laid out like the result arrays (num_particles_with_halos elements
     *  of 1, 2, 4, 8, 12, 16, 24 or 32 bytes); its assigned range is read, its halo ranges are overwritten. */
    int exchangeHalos(void* array, int elemBytes) override
    {
        // the element sizes gatherGpu is instantiated for (R/primitives/primitives_gpu.cu:126-148)
        if (elemBytes != 1 && elemBytes != 2 && elemBytes != 4 && elemBytes != 8 && elemBytes != 12 && elemBytes != 16 &&
            elemBytes != 24 && elemBytes != 32)
            return fail(ctx_, CSTONE_E_ARG, "exchange_halos: element size %d", elemBytes);
        if (firstCall_) return fail(ctx_, CSTONE_E_ARG, "exchange_halos: no sync yet");
        if (useLet_) return let_->exchangeHalos(array, elemBytes);
        uint64_t any = 0;
        for (int p = 0; p < P_; ++p)
            any += haloSend_[p] + haloRecv_[p];
        // every rank must take part if anybody exchanges: the totals of the last sync's count matrix decide
        if (P_ == 1 || haloAnyLast_ == 0) return CSTONE_OK;
        (void)any;
        char* a = static_cast<char*>(array);
        std::vector<size_t> sb(P_), rb(P_);
        for (int p = 0; p < P_; ++p)
            sb[p] = haloSend_[p] * elemBytes, rb[p] = haloRecv_[p] * elemBytes;
        CS_TRY(sendRows_.ensure(ctx_, std::max<size_t>(haloSel_, 1) * elemBytes));
        CS_TRY(recvRows_.ensure(ctx_, std::max<size_t>(haloRecvLo_ + haloRecvHi_, 1) * elemBytes));
        if (haloSel_)
            CS_TRY(cstone_hip_gather(ctx_, elemBytes, sel_.as<uint32_t>(), haloSel_, a + haloRecvLo_ * elemBytes,
                                     sendRows_.p));
        CS_TRY(callComm(comm_.all_to_all_v(comm_.user, sendRows_.p, sb.data(), recvRows_.p, rb.data()),
                        "all_to_all_v (exchangeHalos)"));
        if (haloRecvLo_)
            CS_HIP(ctx_, hipMemcpyAsync(a, recvRows_.p, haloRecvLo_ * elemBytes, hipMemcpyDeviceToDevice, ctx_->stream));
        if (haloRecvHi_)
            CS_HIP(ctx_, hipMemcpyAsync(a + (haloRecvLo_ + haloAssigned_) * elemBytes,
                                        recvRows_.as<char>() + haloRecvLo_ * elemBytes, haloRecvHi_ * elemBytes,
                                        hipMemcpyDeviceToDevice, ctx_->stream));
        return CSTONE_OK;
    }

    /*! Domain::reapplySync (R/domain/domain.hpp:334-378): sends one more per-particle field along the routes of the last
     *  sync -- the same particles leave to the same ranks, the kept ones and the newcomers land in the same slots.
     *  in: n elements laid out like the INPUT arrays of the last sync; out: laid out like the result arrays
     *  (num_particles_with_halos elements), only its assigned range is written (exchangeHalos fills the rest). */
    int reapplySync(const void* in, size_t n, int elemBytes, void* out) override
    {
        if (elemBytes != 1 && elemBytes != 2 && elemBytes != 4 && elemBytes != 8 && elemBytes != 12 && elemBytes != 16 &&
            elemBytes != 24 && elemBytes != 32)
            return fail(ctx_, CSTONE_E_ARG, "reapply_sync: element size %d", elemBytes);
        if (firstCall_) return fail(ctx_, CSTONE_E_ARG, "reapply_sync: no sync yet");
        if (n != rsN_) // checkSizesEqual(prevBufDesc_.size, arrays...), R/domain/domain.hpp:341
            return fail(ctx_, CSTONE_E_ARG, "reapply_sync: array of %zu elements, the last sync took %zu", n, size_t(rsN_));
        if ((n && !in) || !out) return fail(ctx_, CSTONE_E_ARG, "reapply_sync: null array");
        const size_t e        = size_t(elemBytes);
        const uint32_t* keptO = order_.as<uint32_t>() + rsKeptOffset_;
        char* dst             = static_cast<char*>(out) + size_t(view_.start_index) * e;
        const void* recvSorted = nullptr;
        if (rsMoved_)
        {
            std::vector<size_t> sb(P_, 0), rbv(P_, 0);
            for (int p = 0; p < P_; ++p)
            {
                if (p == rank_) continue;
                sb[p]  = rsSendCounts_[p] * e;
                rbv[p] = rsRecvCounts_[p] * e;
            }
            // 32-byte elements are moved as 16-byte vectors: keep every staging buffer 16-byte aligned (DevBuf is)
            CS_TRY(sendRows_.ensure(ctx_, std::max<size_t>(rsSend_, 1) * e));
            CS_TRY(recvRows_.ensure(ctx_, std::max<size_t>(rsNb_, 1) * e));
            CS_TRY(moveTmp_.ensure(ctx_, std::max<size_t>(rsNb_, 1) * e));
            if (rsSend_) CS_TRY(cstone_hip_gather(ctx_, elemBytes, leaving_.as<uint32_t>(), rsSend_, in, sendRows_.p));
            CS_TRY(callComm(comm_.all_to_all_v(comm_.user, sendRows_.p, sb.data(), recvRows_.p, rbv.data()),
                            "all_to_all_v (reapplySync)"));
            if (rsNb_)
            {
                CS_TRY(cstone_hip_gather(ctx_, elemBytes, ro_.as<uint32_t>(), rsNb_, recvRows_.p, moveTmp_.p));
                recvSorted = moveTmp_.p;
            }
        }
        if (rsNb_)
        {
            CS_TRY(cstone_hip_gather_scatter(ctx_, elemBytes, keptO, posA_.as<uint32_t>(), rsNa_, in, dst));
            CS_TRY(cstone_hip_scatter(ctx_, elemBytes, posB_.as<uint32_t>(), rsNb_, recvSorted, dst));
        }
        else { CS_TRY(cstone_hip_gather(ctx_, elemBytes, keptO, rsNa_, in, dst)); }
        return CSTONE_OK;
    }

    /*! Domain::octreeProperties() / layout() (R/domain/domain.hpp:388-437) for the result arrays of the last sync: a
     *  cornerstone tree (bucketFocus) over ALL local particles, halos included -- the keys of
     *  [halos of lower ranks | assigned | halos of higher ranks] ascend over the whole array, so the tree machinery of
     *  the single-rank path applies as it is.  Built on the first request after a sync, from the tree of the previous
     *  request; a sync that nobody asks a tree for does not pay for one. */
    int octree(cstone_hip_domain_mr_octree* out) override
    {
        if (firstCall_) return fail(ctx_, CSTONE_E_ARG, "domain_mr_octree: no sync yet");
        if (useLet_)
        {
            // Domain::octreeProperties() is the focus tree itself (R/domain/domain.hpp:425-437): leaves whose particles
            // are not here have empty layout ranges
            const FocusLet<K, T>& t = *let_;
            out->num_leaves = t.numLeaves(), out->num_nodes = t.numNodes();
            out->leaves = t.leaves(), out->leaf_counts = t.leafCounts();
            out->prefixes = t.prefixes(), out->child_offsets = t.childOffsets();
            out->parents = t.parents(), out->level_range = t.levelRange();
            out->internal_to_leaf = t.internalToLeaf(), out->leaf_to_internal = t.leafToInternal();
            out->layout = t.layout();
            out->centers = t.geoCenters(), out->sizes = t.geoSizes();
            out->expansion_centers = haveExpansionCenters_ ? t.expansionCenters() : nullptr;
            return CSTONE_OK;
        }
        if (nsSync_ != syncs_)
        {
            CS_TRY(buildNsTree());
            nsSync_ = syncs_;
        }
        const NodeIdx L = nsLeaves_, M = L + (L - 1) / 7;
        out->num_leaves = L, out->num_nodes = M;
        out->leaves = nsTree_.p, out->leaf_counts = nsCounts_.as<uint32_t>();
        out->prefixes = nsPrefixes_.p, out->child_offsets = nsChild_.as<int32_t>();
        out->parents = nsParents_.as<int32_t>(), out->level_range = nsLevelRange_.as<int32_t>();
        out->internal_to_leaf = nsItl_.as<int32_t>(), out->leaf_to_internal = nsLti_.as<int32_t>();
        out->layout = nsLayout_.as<uint32_t>();
        out->centers = nsCenters_.p, out->sizes = nsSizes_.p;
        out->expansion_centers = nullptr;
        return CSTONE_OK;
    }

    int updateExpansionCenters(const void* x, const void* y, const void* z, const void* m, int massBits) override
    {
        if (firstCall_ || !useLet_ || !let_) return fail(ctx_, CSTONE_E_ARG, "update_expansion_centers: no sync with the LET yet");
        if ((massBits != 32 && massBits != 64) || !x || !y || !z || !m)
            return fail(ctx_, CSTONE_E_ARG, "update_expansion_centers: bad argument");
        const size_t si = view_.start_index;
        int rc = let_->updateExpansionCenters(static_cast<const T*>(x) + si, static_cast<const T*>(y) + si,
                                              static_cast<const T*>(z) + si, static_cast<const char*>(m) + si * size_t(massBits / 8),
                                              massBits, gTree_.as<K>(), gLeavesHost_.data(), gLeaves_);
        haveExpansionCenters_ = rc == CSTONE_OK;
        return rc;
    }

    ~MultiRankDomain() override
    {
        if (hostLevelRange_)
        {
            if (!ctxGone_) (void)hipStreamSynchronize(ctx_->stream); // a copy into the block may still be queued
            (void)hipHostFree(hostLevelRange_);
        }
        if (timing_ && rank_ == 0)
        {
            std::fprintf(stderr, "[cstone_hip_domain_mr] phase times per sync over %d syncs (ms, synchronising):", syncs_);
            for (auto& [name, sec] : phase_)
                std::fprintf(stderr, "  %s %.3f", name.c_str(), sec * 1e3 / std::max(1, syncs_));
            std::fprintf(stderr, "\n");
        }
    }

    int view(cstone_hip_domain_mr_view* v) override
    {
        *v = view_;
        return CSTONE_OK;
    }

    /*! Domain::sync on several ranks (R/domain/domain.hpp:196-243).  Phases, in the order of the tick() marks:
     *    1  global box (min/max + all-reduce)                      R/sfc/box_mpi.hpp:85-121
     *    2  keys + SFC ordering of the present particles            R/domain/assignment.hpp:81-86
     *    3  global tree update + counts all-reduce, SFC assignment  R/domain/assignment.hpp:88-103
     *    4  send ranges, particle all-to-all, merge of the newcomers, every field written once to its final slot
     *                                                               R/domain/assignment.hpp:105-158, domaindecomp_mpi.hpp:86-174
     *    5  this rank's finest tree over its assigned particles, range boundaries enforced, linked octree
     *    6  owner-side halo discovery (boxes all-gathered, one traversal for all peers), halo count matrix
     *    7  room for the halos left and right of the assigned block
     *    8  halo all-to-all, keys of the halo particles             R/halos/halos.hpp:224-257
     *  then the bookkeeping reapplySync / exchangeHalos / octree() work from. */
    int sync(const void* xIn, const void* yIn, const void* zIn, const void* hIn, size_t n, const void* const* propsIn,
             const int* propBytesIn, int numPropsIn, const void* keysIn, const void* mass, int massBits) override
    {
        // syncGrav: the masses travel as one more property behind the caller's
        const bool grav = massBits != 0;
        const void* propList[MAX_PROPS + 1];
        int propSizes[MAX_PROPS + 1];
        const void* const* props = propsIn;
        const int* propBytes     = propBytesIn;
        int numProps             = numPropsIn;
        if (grav)
        {
            if (!useLet_) return fail(ctx_, CSTONE_E_ARG, "domain_mr_sync_grav: needs the locally essential tree (CSTONE_MR_HALOS_LET)");
            if ((massBits != 32 && massBits != 64) || massBits > rb || (n && !mass) || numPropsIn < 0 || numPropsIn >= MAX_PROPS)
                return fail(ctx_, CSTONE_E_ARG, "domain_mr_sync_grav: masses of 32 or 64 bits (not wider than the coordinates), "
                                                "at most %d further properties", MAX_PROPS - 1);
            for (int q = 0; q < numPropsIn; ++q)
                propList[q] = propsIn[q], propSizes[q] = propBytesIn[q];
            propList[numPropsIn] = mass, propSizes[numPropsIn] = massBits / 8;
            props = propList, propBytes = propSizes, numProps = numPropsIn + 1;
        }
        // A rank-local failure must reach the peers: they are about to enter the collectives of this sync and would wait
        // there for ever.  Failures of the arguments (and the failures injected by the tests, CSTONE_MR_FAIL_AT) are
        // therefore kept as a pending status; the rank goes on as an EMPTY rank, the status word rides on the next
        // collective (box all-reduce, count all-gathers) and every rank returns an error behind it.
        pending_ = 0, toggled_ = false;
        if (placeForked_)
        {
            // (a sync that was abandoned behind its fork: whatever it left on the second stream comes first)
            CS_HIP(ctx_, hipStreamWaitEvent(ctx_->stream, ctx_->evJoin, 0));
            placeForked_ = false;
        }
        ctx_->auxBusy = false;
        if (numProps < 0 || numProps > MAX_PROPS) setPending(CSTONE_E_ARG, "domain_mr_sync: at most %d properties", MAX_PROPS);
        for (int q = 0; q < numProps && !pending_; ++q)
        {
            const int e = propBytes[q]; // the element sizes gatherGpu is instantiated for (R/primitives/primitives_gpu.cu:126-148)
            const bool sizeOk = e == 1 || e == 2 || e == 4 || e == 8 || e == 12 || e == 16 || e == 24 || e == 32;
            if ((n && !props[q]) || !sizeOk) // an empty rank may pass null arrays
                setPending(CSTONE_E_ARG, "domain_mr_sync: property %d must have elements of 1, 2, 4, 8, 12, 16, 24 or 32 bytes", q);
        }
        const T* x = static_cast<const T*>(xIn);
        const T* y = static_cast<const T*>(yIn);
        const T* z = static_cast<const T*>(zIn);
        const T* h = static_cast<const T*>(hIn);
        if (n >= (size_t(1) << 30)) setPending(CSTONE_E_ARG, "domain_mr_sync: too many particles per rank");
        injectFailure("start");
        if (pending_)
        {
            if (P_ == 1) return agreed(rank_);
            n = 0, numProps = 0; // limp on as an empty rank until the peers know
        }
        CS_TRY(scal_.ensure(ctx_, 4096 + size_t(P_) * (P_ + 1) * 8 + size_t(P_ + 1) * 16));
        ++syncs_;
        tick(nullptr);

        // ---- the box.  Measuring it costs a pass over x, y, z and a round trip before the first key can be computed.
        //      A sync that is going to re-sort (below) computes its keys with the box of the previous sync instead and
        //      measures the extents in the same pass; the all-reduce of the extents follows the encode, the result comes
        //      back with the re-sort's counters, and only a box that really changed costs a second encode (with the
        //      radix path, as every key changes then).  After such a sync the extents are measured first again until a
        //      sync finds the box unchanged (an open box whose outermost particles move changes every time).
        //      The box all-reduce stays the first collective of the sync on every rank either way.
        const bool anyOpen = !(box_.bc[0] == 1 && box_.bc[1] == 1 && box_.bc[2] == 1);
        bool boxSame       = true;
        for (int k = 0; k < 6; ++k)
            boxSame = boxSame && box_.lim[k] == layoutBox_.lim[k];
        const int tileLeavesSpec = LeafResort<K>::leavesPerTile(bucketFocus_);
        // What the re-sort of THIS sync starts from: the leaves of my range and their layout as the last COMPLETED sync
        // left them.  The members are invalidated here and set again only at the end of the tree update below: a sync
        // that fails half way (a peer's failure, an unmatched halo ...) may have rebalanced, swapped or freed the
        // buffers resortTree_ points into, and the retry must then sort from scratch instead of reading them.
        const K* const resortTree       = resortTree_;
        const int resortLeaves          = resortLeaves_;
        const uint64_t layoutParticles  = layoutParticles_;
        resortTree_ = nullptr, resortLeaves_ = 0, layoutParticles_ = 0;
        bool speculate = anyOpen && !firstCall_ && !measureFirst_ && !pending_ && boxSame && n >= resortMinParticles() &&
                         n == layoutParticles && tileLeavesSpec > 0 && resortLeaves > 0 && resortBackoff_ == 0 &&
                         mayResort() && speculativeBox_;
        if (!speculate)
        {
            const cstone_box before = box_;
            CS_TRY(updateBox(x, y, z, n));
            bool moved = false;
            for (int k = 0; k < 6; ++k)
                moved = moved || box_.lim[k] != before.lim[k];
            if (!firstCall_ && anyOpen) measureFirst_ = moved;
        }
        tick("1 box");

        // ---- keys + SFC ordering of the present particles
        const size_t nAlloc = std::max<size_t>(n, 64);
        CS_TRY(keys_.ensure(ctx_, nAlloc * sizeof(K)));
        CS_TRY(order_.ensure(ctx_, nAlloc * sizeof(uint32_t)));
        CS_TRY(ensureSortScratch(nAlloc));
        // encode leaves entries that hold the remove marker 2^(3 maxLevel) alone (R/sfc/sfc.hpp:284-291): such particles
        // sort behind the end of the curve and leave the domain
        // (without a key array from the caller there are no markers: the key buffer is then pure output)
        if (keysIn && n)
            CS_HIP(ctx_, hipMemcpyAsync(keys_.p, keysIn, n * sizeof(K), hipMemcpyDeviceToDevice, ctx_->stream));
        bool partialSort = false;
        // the level ranges of the previous sync's tree (copied to the pinned block behind its build, many stream
        // synchronisations ago): refreshed at EVERY sync -- buildFocusOctree() bounds its digit passes by it, also on
        // the syncs that re-sort
        if (levelRangePending_)
        {
            prevMaxLeafLevel_ = 0;
            for (int l = 0; l <= int(maxLevel<K>()); ++l)
                if (hostLevelRange_[l + 1] > hostLevelRange_[l]) prevMaxLeafLevel_ = l;
            levelRangePending_ = false;
        }
        // ---- the incremental re-sort (resort.hpp), as in the single-rank domain: the input arrays are the assigned block
        //      the previous sync handed out, ordered by the leaves of this rank's tree (layout_); particles still inside
        //      their leaf are ordered leaf by leaf, the others are binned.  Rank-local: no collective depends on it.
        bool resorted         = false;
        bool boxChecked       = false;
        const int tileLeaves  = LeafResort<K>::leavesPerTile(bucketFocus_);
        bool sameBox          = true;
        for (int k = 0; k < 6; ++k)
            sameBox = sameBox && box_.lim[k] == layoutBox_.lim[k];
        const size_t resortMin = resortMinParticles();
        const bool tryResort = !firstCall_ && n >= resortMin && n == layoutParticles && tileLeaves > 0 && sameBox && resortLeaves > 0 &&
                               resortBackoff_ == 0 && !pending_ && mayResort();
        if (resortBackoff_ > 0) --resortBackoff_;
        if (tryResort)
        {
            CS_TRY(resort_.prepare(ctx_, resortTree, layout_.as<uint32_t>(), resortLeaves, n, keysAlt_.as<K>(),
                                   lastMovers_ > 100000));
            const ResortArgs<K> ra = resort_.args();
            bool done              = false;
            T* extentsDev = reinterpret_cast<T*>(scal_.as<char>() + 256); // {min, max} per axis, measured by the encode
            CS_TRY(computeKeysResort(ctx_, curve_, kb, rb, x, y, z, keysIn ? keys_.p : nullptr, n, box_, &ra,
                                     speculate ? extentsDev : nullptr, &done));
            bool boxHolds = true, foundHere = false;
            int counters[4] = {0, 0, 0, 0};
            if (speculate)
            {
                boxChecked = true;
                // (the keys above were computed with the box of the previous sync: was it still the box?)
                double* dev = scal_.as<double>();
                if (done)
                {
                    CS_TRY(resort_.binMovers(ctx_, tileLeaves));
                    // (the re-sort's counters and this rank's status word become part of the operand: one launch, and
                    //  one copy and one synchronisation bring the reduced extents and the counters)
                    CS_TRY(extentsToReduceOperand(ctx_, rb, extentsDev, dev, statusWord(), ctx_->devScalars + RESORT_SCALARS));
                    foundHere = true;
                }
                else
                {
                    const void* arrays[3] = {x, y, z};
                    CS_TRY(minMaxCoordinatesDev(ctx_, rb, arrays, 3, n, dev));
                }
                cstone_box next;
                CS_TRY(reduceBox(dev, &next, foundHere, foundHere ? counters : nullptr));
                for (int k = 0; k < 6; ++k)
                    boxHolds = boxHolds && next.lim[k] == box_.lim[k];
                if (!boxHolds)
                {
                    box_          = next;
                    measureFirst_ = true;
                    ++boxRedos_;
                }
            }
            else if (done) { CS_TRY(resort_.binMovers(ctx_, tileLeaves)); }
            if (done && boxHolds)
            {
                int found[4];
                if (foundHere) { std::copy(counters, counters + 4, found); }
                else { CS_TRY(toHost(found, ctx_->devScalars + RESORT_SCALARS, sizeof found)); }
                const uint32_t markers = uint32_t(found[0]), J = uint32_t(found[2]), movers = uint32_t(found[3]);
                if ((found[1] & 7) == 0 && movers <= n / 8)
                {
                    CS_TRY(resort_.sortLeaves(ctx_, keysAlt_.as<K>(), keys_.as<K>(), order_.as<uint32_t>(), movers, markers,
                                              J, tileLeaves, (found[1] & 8) != 0));
                    resorted    = true;
                    lastMovers_ = movers;
                    ++resorts_;
                }
                else { resortBackoff_ = 4; }
            }
        }
        if (speculate && !boxChecked)
        {
            // (cannot happen: the conditions of the speculation are those of the attempt above; a box is never left unmeasured)
            CS_TRY(updateBox(x, y, z, n));
        }
        if (n && !resorted)
        {
            // radix passes only over the digits above the leaf level (+1) of the previous tree, runs of equal high digits
            // are finished by a fix-up pass; a run that is too long raises a flag and the regular sort completes the job
            int startPass = 0;
            if (!firstCall_ && prevMaxLeafLevel_ >= 0 && !allDigits())
                startPass = std::max(0, (3 * int(maxLevel<K>()) -
                                         3 * (prevMaxLeafLevel_ + 1 + (bucketFocus_ > 128) + (bucketFocus_ > 1024))) / 8) & ~1;
            int* tooLong = reinterpret_cast<int*>(scal_.as<char>() + 128);
            CS_TRY(sfcKeysAndOrderingHint(ctx_, curve_, kb, rb, x, y, z, keys_.p, order_.as<uint32_t>(), n, box_,
                                          keysAlt_.p, orderAlt_.as<uint32_t>(), sortTmp_.p, sortTmp_.bytes, startPass,
                                          tooLong, keysIn != nullptr));
            // the flag travels to the (pinned) host block behind the sort; the read-back of the global tree update below
            // completes the stream, so no synchronisation of its own is needed (the global leaf boundaries cannot fall
            // inside a run: the update is not affected by an unfinished order)
            partialSort = startPass > 0;
            if (partialSort)
                CS_TRY(copyToPinned(ctx_, ctx_->hostScalars + 3, tooLong, sizeof(int)));
        }

        tick("2 encode+sort");
        CS_TRY(updateGlobalTree(n));
        CS_TRY(queueGlobalTreeReadBack());

        // ---- C3 (first half): send ranges on the sorted keys and the counts of everybody.  The assignment follows from
        //      the global counts that are on their way to the host; it rarely changes from one sync to the next (a
        //      boundary moves by whole leaves of the global tree), so the cut points for the assignment of the LAST sync
        //      are computed and all-gathered right behind the counts, and ONE read-back brings global counts, cut points
        //      and count matrix.  Only a sync whose assignment did change asks again.  (A partially sorted key array --
        //      runs of equal high digits still to be fixed up -- answers these searches correctly: an assignment boundary
        //      is a leaf boundary of the global tree and cannot fall inside a run.)
        std::vector<uint64_t> cut(P_ + 1);
        std::vector<uint64_t> sendCounts(P_), matrix(size_t(P_) * P_, 0);
        std::vector<uint64_t> rows(size_t(P_) * (P_ + 1), 0);
        std::vector<K> cutKeys; // the assignment the queued cut points belong to
        uint64_t *pinRows = nullptr, *pinCut = nullptr;
        auto queueCuts = [&](const std::vector<K>& asg) -> int
        {
            cutKeys      = asg;
            K* dq        = reinterpret_cast<K*>(scal_.as<char>() + 2048);
            uint64_t* dr = reinterpret_cast<uint64_t*>(scal_.as<char>() + 2048 + size_t(P_ + 1) * 8);
            CS_TRY(cstone_hip_upload(ctx_, dq, asg.data(), size_t(P_ + 1) * sizeof(K)));
            // the send counts go from the device into the all-gather; word P of every row: the status of that rank
            // (0 = fine), see the top of sync()
            uint64_t* send = scal_.as<uint64_t>() + 32;
            uint64_t* recv = reinterpret_cast<uint64_t*>(scal_.as<char>() + 4096);
            hipLaunchKernelGGL(cutPointsKernel<K>, gridFor(size_t(P_) + 1, 64), 64, 0, ctx_->stream, keys_.as<K>(), n, dq, P_,
                               dr, send, uint64_t(pending_ ? 1 : 0));
            CS_HIP(ctx_, hipGetLastError());
            if (!pinRows)
            {
                pinRows = static_cast<uint64_t*>(pin_.take(rows.size() * 8));
                pinCut  = static_cast<uint64_t*>(pin_.take(size_t(P_ + 1) * 8));
            }
            if (P_ > 1)
            {
                CS_TRY(callComm(comm_.all_gather(comm_.user, send, recv, size_t(P_ + 1) * 8), "all_gather (counts)"));
                CS_TRY(copyToPinned(ctx_, pinRows, recv, rows.size() * 8));
            }
            CS_TRY(copyToPinned(ctx_, pinCut, dr, size_t(P_ + 1) * 8));
            return CSTONE_OK;
        };
        auto takeCuts = [&]()
        {
            std::copy(pinCut, pinCut + P_ + 1, cut.begin());
            if (P_ > 1) std::copy(pinRows, pinRows + rows.size(), rows.begin());
        };
        injectFailure("assign");
        const bool speculateCuts = !firstCall_ && int(assignment_.size()) == P_ + 1 && speculateCuts_;
        if (speculateCuts) CS_TRY(queueCuts(assignment_));
        CS_HIP(ctx_, hipStreamSynchronize(ctx_->stream)); // global counts (+ leaves), the sort's flag, cut points, matrix
        takeGlobalTreeReadBack();
        if (speculateCuts) takeCuts();
        if (partialSort && ctx_->hostScalars[3] != 0)
            CS_TRY(cstone_hip_sort_pairs(ctx_, kb, keys_.p, order_.as<uint32_t>(), n, keysAlt_.p, orderAlt_.as<uint32_t>(),
                                         sortTmp_.p, sortTmp_.bytes));
        CS_TRY(assign());
        tick("3 global tree+assign");
        if (!speculateCuts || cutKeys != assignment_)
        {
            if (speculateCuts) ++cutRedos_;
            CS_TRY(queueCuts(assignment_));
            CS_HIP(ctx_, hipStreamSynchronize(ctx_->stream));
            takeCuts();
        }
        {
            for (int p = 0; p < P_; ++p)
                sendCounts[p] = cut[p + 1] - cut[p];
            if (P_ == 1) matrix[0] = sendCounts[0];
            for (int p = 0; p < P_ && P_ > 1; ++p)
            {
                if (rows[size_t(p) * (P_ + 1) + P_] != 0) return agreed(p);
                for (int q = 0; q < P_; ++q)
                    matrix[size_t(p) * P_ + q] = rows[size_t(p) * (P_ + 1) + q];
            }
        }
        // conditions every rank derives from the same matrix: all of them return the same error, nobody is left waiting
        for (int q = 0; q < P_; ++q)
        {
            uint64_t arriving = 0;
            for (int p = 0; p < P_; ++p)
                arriving += matrix[size_t(p) * P_ + q];
            if (arriving == 0) return fail(ctx_, CSTONE_E_ARG, "domain_mr_sync: rank %d is left without particles", q);
            if (arriving >= (uint64_t(1) << 30))
                return fail(ctx_, CSTONE_E_ARG, "domain_mr_sync: too many particles for rank %d", q);
        }
        std::vector<size_t> sendBytes(P_, 0), recvBytes(P_, 0);
        uint64_t movedAny = 0, mSend = 0, nb = 0;
        for (int p = 0; p < P_; ++p)
        {
            for (int q = 0; q < P_; ++q)
                if (p != q) movedAny += matrix[size_t(p) * P_ + q];
            if (p == rank_) continue;
            sendBytes[p] = sendCounts[p] * 4 * sizeof(T);
            recvBytes[p] = matrix[size_t(p) * P_ + rank_] * 4 * sizeof(T);
            mSend += sendCounts[p];
            nb += matrix[size_t(p) * P_ + rank_];
        }
        const uint64_t na     = sendCounts[rank_];
        const K* keptKeys     = keys_.as<K>() + cut[rank_];
        const uint32_t* keptO = order_.as<uint32_t>() + cut[rank_];
        const uint64_t nm     = na + nb;

        if (movedAny)
        {
            CS_TRY(leaving_.ensure(ctx_, std::max<size_t>(mSend, 1) * sizeof(uint32_t)));
            CS_TRY(sendRows_.ensure(ctx_, std::max<size_t>(mSend, 1) * 4 * sizeof(T)));
            CS_TRY(recvRows_.ensure(ctx_, std::max<size_t>(nb, 1) * 4 * sizeof(T)));
            size_t nLow = cut[rank_] - cut[0], nHigh = cut[P_] - cut[rank_ + 1];
            if (nLow)
                CS_HIP(ctx_, hipMemcpyAsync(leaving_.p, order_.as<uint32_t>() + cut[0], nLow * 4,
                                            hipMemcpyDeviceToDevice, ctx_->stream));
            if (nHigh)
                CS_HIP(ctx_, hipMemcpyAsync(leaving_.as<uint32_t>() + nLow, order_.as<uint32_t>() + cut[rank_ + 1],
                                            nHigh * 4, hipMemcpyDeviceToDevice, ctx_->stream));
            if (mSend)
                hipLaunchKernelGGL(packRowsKernel<T>, gridFor(mSend, 256), 256, 0, ctx_->stream,
                                   leaving_.as<uint32_t>(), size_t(mSend), x, y, z, h, sendRows_.as<T>());
            CS_TRY(callComm(comm_.all_to_all_v(comm_.user, sendRows_.p, sendBytes.data(), recvRows_.p, recvBytes.data()),
                            "all_to_all_v (particles)"));
        }

        // ---- newcomers: sorted among themselves
        const T* recvSorted[4] = {nullptr, nullptr, nullptr, nullptr};
        if (nb)
        {
            for (int c = 0; c < 4; ++c)
            {
                CS_TRY(rcol_[c].ensure(ctx_, nb * sizeof(T)));
                CS_TRY(rcolS_[c].ensure(ctx_, nb * sizeof(T)));
            }
            hipLaunchKernelGGL(unpackRowsKernel<T>, gridFor(nb, 256), 256, 0, ctx_->stream, recvRows_.as<T>(), size_t(nb),
                               rcol_[0].as<T>(), rcol_[1].as<T>(), rcol_[2].as<T>(), rcol_[3].as<T>());
            CS_TRY(rk_.ensure(ctx_, nb * sizeof(K)));
            CS_TRY(ro_.ensure(ctx_, nb * sizeof(uint32_t)));
            CS_TRY(ensureSortScratch(std::max<size_t>(nAlloc, nb)));
            CS_HIP(ctx_, hipMemsetAsync(rk_.p, 0, nb * sizeof(K), ctx_->stream));
            CS_TRY(cstone_hip_compute_sfc_keys(ctx_, curve_, kb, rb, rcol_[0].p, rcol_[1].p, rcol_[2].p, rk_.p, nb,
                                               &box_));
            CS_TRY(cstone_hip_sort_keys_ordering(ctx_, kb, rk_.p, ro_.as<uint32_t>(), nb, keysAlt_.p,
                                                 orderAlt_.as<uint32_t>(), sortTmp_.p, sortTmp_.bytes));
            for (int c = 0; c < 4; ++c)
            {
                CS_TRY(cstone_hip_gather(ctx_, sizeof(T), ro_.as<uint32_t>(), nb, rcol_[c].p, rcolS_[c].p));
                recvSorted[c] = rcolS_[c].as<T>();
            }
        }

        // ---- further conserved fields travel the same way, one collective per field (the volume is small in the steady
        //      state); received values are brought into the newcomers' sorted order
        DevBuf* propRecvSorted[MAX_PROPS] = {};
        for (int q = 0; q < numProps && movedAny; ++q)
        {
            const size_t e = size_t(propBytes[q]);
            std::vector<size_t> sb(P_, 0), rbv(P_, 0);
            for (int p = 0; p < P_; ++p)
            {
                if (p == rank_) continue;
                sb[p]  = sendCounts[p] * e;
                rbv[p] = matrix[size_t(p) * P_ + rank_] * e;
            }
            CS_TRY(sendRows_.ensure(ctx_, std::max<size_t>(mSend, 1) * e));
            CS_TRY(propRecv_[q].ensure(ctx_, std::max<size_t>(nb, 1) * e));
            CS_TRY(propRecvS_[q].ensure(ctx_, std::max<size_t>(nb, 1) * e));
            if (mSend) CS_TRY(cstone_hip_gather(ctx_, int(e), leaving_.as<uint32_t>(), mSend, props[q], sendRows_.p));
            CS_TRY(callComm(comm_.all_to_all_v(comm_.user, sendRows_.p, sb.data(), propRecv_[q].p, rbv.data()),
                            "all_to_all_v (property)"));
            if (nb) CS_TRY(cstone_hip_gather(ctx_, int(e), ro_.as<uint32_t>(), nb, propRecv_[q].p, propRecvS_[q].p));
            propRecvSorted[q] = &propRecvS_[q];
        }

        // ---- Result arrays.  The assigned block is written ONCE, at an offset M that leaves room for the halos of the
        //      lower ranks (their number is only known after the discovery below; M is generous and follows the
        //      previous sync).  The arrays handed out start at M - (halos of lower ranks).
        cur_ ^= 1; // the inputs may live in the other buffer set
        toggled_ = true;
        Out& o           = out_[cur_];
        const bool margins = P_ > 1 && !noMargin_;
        // (a multiple of 4 elements: the assigned range the client passes back as the next input then starts on a
        //  16-byte boundary, which the fused encode + digit counting kernel needs for its vector loads)
        const uint64_t M   = margins ? (std::max<uint64_t>(2 * prevLo_ + 4096, firstCall_ ? nm / 4 : 0) + 3) & ~uint64_t(3) : 0;
        uint64_t cap       = M + nm + (margins ? std::max<uint64_t>(2 * prevHi_ + 4096, firstCall_ ? nm / 4 : 0) : 0);
        CS_TRY(o.keys.ensure(ctx_, cap * sizeof(K)));
        for (DevBuf* b : {&o.x, &o.y, &o.z, &o.h})
            CS_TRY(b->ensure(ctx_, cap * sizeof(T)));
        for (int q = 0; q < numProps; ++q)
            CS_TRY(o.props[q].ensure(ctx_, cap * size_t(propBytes[q])));

        // ---- merge of the kept, already sorted range with the newcomers: positions, then every field from its input
        //      slot straight to its final slot
        K* keysM = o.keys.as<K>() + M;
        if (nb)
        {
            CS_TRY(posA_.ensure(ctx_, std::max<size_t>(na, 1) * sizeof(uint32_t)));
            CS_TRY(posB_.ensure(ctx_, nb * sizeof(uint32_t)));
            CS_TRY(cstone_hip_merge_positions(ctx_, kb, keptKeys, na, rk_.p, nb, 0, posA_.as<uint32_t>(),
                                              posB_.as<uint32_t>()));
            CS_TRY(cstone_hip_scatter(ctx_, sizeof(K), posB_.as<uint32_t>(), nb, rk_.p, keysM));
        }
        {
            T* dst[4] = {o.x.as<T>() + M, o.y.as<T>() + M, o.z.as<T>() + M, o.h.as<T>() + M};
            // keys and h first: the locally essential tree and the halo discovery work on them.  x, y, z are not read
            // before the halo exchange: they go to their final slots on the context's second stream, next to the tree
            // update (chains of small kernels and read-backs), and are joined in front of the exchange
            const bool overlap = useLet_ && overlapPlace_ && !grav; // (syncGrav reads x, y, z for the mass centres)
            if (overlap) CS_TRY(ensureAuxStream(ctx_));
            if (na)
            {
                StageTimer timer(ctx_, CSTONE_STAGE_PLACE);
                if (overlap)
                    hipLaunchKernelGGL((placeColumnsKernel<K, T, 1>), gridFor(na, 256, PLACE_PER), 256, 0, ctx_->stream, keptO,
                                       nb ? posA_.as<uint32_t>() : nullptr, size_t(na), keptKeys, x, y, z, h, keysM, dst[0],
                                       dst[1], dst[2], dst[3]);
                else
                    hipLaunchKernelGGL((placeColumnsKernel<K, T, 0>), gridFor(na, 256, PLACE_PER), 256, 0, ctx_->stream, keptO,
                                       nb ? posA_.as<uint32_t>() : nullptr, size_t(na), keptKeys, x, y, z, h, keysM, dst[0],
                                       dst[1], dst[2], dst[3]);
            }
            if (nb) CS_TRY(cstone_hip_scatter(ctx_, sizeof(T), posB_.as<uint32_t>(), nb, recvSorted[3], dst[3]));
            if (overlap)
            {
                CS_HIP(ctx_, hipEventRecord(ctx_->evFork, ctx_->stream));
                CS_HIP(ctx_, hipStreamWaitEvent(ctx_->aux, ctx_->evFork, 0));
                int rc = CSTONE_OK;
                {
                    StreamScope scope(ctx_, ctx_->aux);
                    if (na)
                    {
                        StageTimer timer(ctx_, CSTONE_STAGE_PLACE);
                        hipLaunchKernelGGL((placeColumnsKernel<K, T, 2>), gridFor(na, 256, PLACE_PER), 256, 0, ctx_->stream, keptO,
                                           nb ? posA_.as<uint32_t>() : nullptr, size_t(na), keptKeys, x, y, z, h, keysM,
                                           dst[0], dst[1], dst[2], dst[3]);
                    }
                    for (int c = 0; c < 3 && nb && rc == CSTONE_OK; ++c)
                        rc = cstone_hip_scatter(ctx_, sizeof(T), posB_.as<uint32_t>(), nb, recvSorted[c], dst[c]);
                }
                CS_TRY(rc);
                CS_HIP(ctx_, hipEventRecord(ctx_->evJoin, ctx_->aux));
                placeForked_  = true;
                ctx_->auxBusy = true;
            }
            else
            {
                for (int c = 0; c < 3 && nb; ++c)
                    CS_TRY(cstone_hip_scatter(ctx_, sizeof(T), posB_.as<uint32_t>(), nb, recvSorted[c], dst[c]));
            }
        }
        for (int q = 0; q < numProps; ++q)
        {
            const int e = propBytes[q];
            char* dst   = o.props[q].as<char>() + M * e;
            if (nb)
            {
                CS_TRY(cstone_hip_gather_scatter(ctx_, e, keptO, posA_.as<uint32_t>(), na, props[q], dst));
                CS_TRY(cstone_hip_scatter(ctx_, e, posB_.as<uint32_t>(), nb, propRecvSorted[q]->p, dst));
            }
            else { CS_TRY(cstone_hip_gather(ctx_, e, keptO, na, props[q], dst)); }
        }
        tick("4 exchange+merge+place");

        std::vector<uint64_t> hsCounts(P_, 0), hmatrix(size_t(P_) * P_, 0);
        uint64_t numMyBoxes = 0, selTotal = 0;
        uint64_t nlo = 0, nhi = 0, haloAny = 0; // haloAny: the same on every rank, it decides about the collective
        std::vector<size_t> hSendBytes(P_, 0), hRecvBytes(P_, 0);
        if (useLet_)
        {
            // ---- the reference's way (R/domain/domain.hpp:217-237): peers, locally essential tree (focus tree), halo
            //      discovery on it, key-range requests to the owners -- csrc/let.hpp.  h is in SFC order at o.h + M.
            if (!let_) let_ = std::make_unique<FocusLet<K, T>>(ctx_, curve_, rank_, P_, bucketFocus_, theta_, comm_);
            injectFailure("exchange");
            int rc;
            if (grav)
            {
                const char* mSorted = o.props[numProps - 1].as<char>() + M * size_t(massBits / 8);
                rc = let_->updateGrav(box_, keysM, size_t(nm), assignment_.data(), gTree_.as<K>(), gLeavesHost_.data(),
                                      gCounts_.as<uint32_t>(), gLeaves_, o.x.as<T>() + M, o.y.as<T>() + M, o.z.as<T>() + M,
                                      mSorted, massBits, o.h.as<T>() + M, haloExt_, &centerDriftTol_,
                                      pending_ ? rank_ + 1 : 0, gTreeSame_);
                haveExpansionCenters_ = rc == CSTONE_OK;
            }
            else
            {
                rc = let_->update(box_, keysM, size_t(nm), assignment_.data(), gTree_.as<K>(), gCounts_.as<uint32_t>(),
                                  gLeaves_, o.h.as<T>() + M, haloExt_, pending_ ? rank_ + 1 : 0, gTreeSame_);
                haveExpansionCenters_ = false;
            }
            if (rc != CSTONE_OK)
            {
                // (a failure of my own that the status word of the tree's last count exchange has told everybody about:
                //  reported with its own message)
                if (pending_) return agreed(rank_);
                if (toggled_) cur_ ^= 1, toggled_ = false;
                return rc;
            }
            if (uint64_t(let_->endIndex() - let_->startIndex()) != nm)
                return fail(ctx_, CSTONE_E_INTERNAL, "domain_mr_sync: the focus tree counts %u assigned particles, %llu are here",
                            let_->endIndex() - let_->startIndex(), (unsigned long long)nm);
            nlo      = let_->startIndex();
            nhi      = let_->numParticlesWithHalos() - let_->endIndex();
            haloAny  = 1;
            selTotal = let_->halosSent();
            // the leaves of my own range and their offsets among my particles: what the next sync's re-sort starts from
            const int first = let_->startCell(), last = let_->endCell();
            fLeaves_        = let_->numLeaves();
            resortTree_     = let_->leaves() + first;
            resortLeaves_   = last - first;
            CS_TRY(layout_.ensure(ctx_, size_t(resortLeaves_ + 1) * sizeof(uint32_t)));
            CS_TRY(cstone_hip_increment(ctx_, 32, let_->layout() + first, layout_.p, size_t(resortLeaves_) + 1,
                                        uint64_t(uint32_t(0u - uint32_t(nlo)))));
            layoutParticles_ = nm;
            layoutBox_       = box_;
            if (!hostLevelRange_)
                CS_HIP(ctx_, hipHostMalloc(reinterpret_cast<void**>(&hostLevelRange_), 32 * sizeof(NodeIdx), hipHostMallocDefault));
            // (the level ranges came back with the layout: no copy of their own)
            std::copy(let_->levelRangeHost().begin(), let_->levelRangeHost().end(), hostLevelRange_);
            levelRangePending_ = true;
            tick("5 focus tree (LET)");
        }
        else
        {
            // ---- this rank's finest tree over its assigned particles; its SFC range must end on leaf boundaries
            CS_TRY(updateFocusTree(keysM, nm));
            tick("5a focus update");
            int first = 0, last = 0;
            CS_TRY(enforceBoundaries(keysM, nm, &first, &last));
            tick("5b boundaries");
            CS_TRY(buildFocusOctree());
            tick("5c linked octree");
            // the level ranges are only needed by the NEXT sync (how many digits to sort): they travel to a pinned block of
            // this domain now and are looked at then, behind many later synchronisations of the stream
            if (!hostLevelRange_)
                CS_HIP(ctx_, hipHostMalloc(reinterpret_cast<void**>(&hostLevelRange_), 32 * sizeof(NodeIdx), hipHostMallocDefault));
            CS_TRY(copyToPinned(ctx_, hostLevelRange_, fLevelRange_.p, (maxLevel<K>() + 2) * sizeof(NodeIdx)));
            levelRangePending_ = true;
            const int L = fLeaves_;
            CS_TRY(layout_.ensure(ctx_, size_t(L + 1) * sizeof(uint32_t)));
            CS_HIP(ctx_, hipMemsetAsync(layout_.p, 0, sizeof(uint32_t), ctx_->stream));
            CS_TRY(cstone_hip_inclusive_scan_u32(ctx_, fCounts_.as<uint32_t>(), layout_.as<uint32_t>() + 1, size_t(L)));
            layoutParticles_ = nm; // the next sync's re-sort starts from this layout: nm particles in this box
            layoutBox_       = box_;
            resortTree_      = fTree_.as<K>();
            resortLeaves_    = fLeaves_;
            tick("5 focus tree");

            // ---- C4: owner-side halo discovery
            if (P_ > 1)
            {
                const int nLocal = last - first;
                CS_TRY(radii_.ensure(ctx_, size_t(L) * sizeof(float)));
                CS_TRY(boxes_.ensure(ctx_, size_t(std::max(nLocal, 1)) * 32));
                CS_TRY(boxFlags_.ensure(ctx_, size_t(nLocal + 1) * sizeof(uint32_t)));
                CS_TRY(cstone_hip_halo_radii(ctx_, rb, o.h.as<T>() + M, layout_.as<uint32_t>() + first, first, last, L, haloExt_,
                                             radii_.as<float>()));
                // only boxes that really reach a leaf outside my range are exported (a third to a tenth of those the
                // enclosing-node test alone lets through: less to gather, fewer targets for every owner's traversal)
                CS_TRY(cstone_hip_halo_boxes_foreign(ctx_, curve_, kb, rb, fPrefixes_.p, fChild_.as<int32_t>(),
                                                     fItl_.as<int32_t>(), fTree_.p, radii_.as<float>(), &box_, first, last,
                                                     boxes_.as<int32_t>()));
                hipLaunchKernelGGL(boxFlagsKernel, gridFor(nLocal, 256), 256, 0, ctx_->stream, boxes_.as<int32_t>(), nLocal,
                                   boxFlags_.as<uint32_t>());
                uint32_t* total = scal_.as<uint32_t>() + 16;
                CS_TRY(exclusiveScanWithTotal(boxFlags_.as<uint32_t>(), nLocal, total));
                // box counts of everybody straight from the device scalar (one read-back for mine and theirs), then the
                // boxes themselves padded to the longest list
                std::vector<uint64_t> boxCounts(P_);
                {
                    uint32_t* recv = reinterpret_cast<uint32_t*>(scal_.as<char>() + 4096);
                    injectFailure("exchange");
                    statusW32_ = pending_ ? 1u : 0u; // second word: the status of this rank (a member, see above)
                    CS_HIP(ctx_, hipMemcpyAsync(total + 1, &statusW32_, 4, hipMemcpyHostToDevice, ctx_->stream));
                    CS_TRY(callComm(comm_.all_gather(comm_.user, total, recv, 8), "all_gather (box counts)"));
                    std::vector<uint32_t> c32(size_t(P_) * 2);
                    CS_TRY(toHost(c32.data(), recv, c32.size() * 4));
                    for (int p = 0; p < P_; ++p)
                    {
                        if (c32[2 * p + 1] != 0) return agreed(p);
                        boxCounts[p] = c32[2 * p];
                    }
                }
                const uint32_t nbx = uint32_t(boxCounts[rank_]);
                numMyBoxes         = nbx;
                CS_TRY(myBoxes_.ensure(ctx_, size_t(std::max<uint32_t>(nbx, 1)) * 32));
                hipLaunchKernelGGL(compactBoxesKernel, gridFor(nLocal, 256), 256, 0, ctx_->stream, boxes_.as<int32_t>(),
                                   boxFlags_.as<uint32_t>(), nLocal, rank_, myBoxes_.as<int32_t>());
                uint64_t maxBoxes = *std::max_element(boxCounts.begin(), boxCounts.end());
                if (maxBoxes)
                {
                    CS_TRY(myBoxes_.ensure(ctx_, size_t(maxBoxes) * 32, true));
                    if (maxBoxes > numMyBoxes) // padding records must read "no box"
                        CS_HIP(ctx_, hipMemsetAsync(myBoxes_.as<char>() + numMyBoxes * 32, 0, (maxBoxes - numMyBoxes) * 32,
                                                    ctx_->stream));
                    CS_TRY(allBoxes_.ensure(ctx_, size_t(maxBoxes) * 32 * P_));
                    CS_TRY(callComm(comm_.all_gather(comm_.user, myBoxes_.p, allBoxes_.p, size_t(maxBoxes) * 32),
                                    "all_gather (halo boxes)"));
                }
                CS_TRY(oflags_.ensure(ctx_, size_t(L) * sizeof(int32_t)));
                bool haveMatrix = false;
                if (maxBoxes && P_ <= 32 && !peerLoop_)
                {
                    // all peers in one go: the records carry their exporter, find_overlaps sets one bit per exporter
                    // (two calls: the records before and behind my own); then counts, one scan and one fill for all peers
                    const int np = P_ - 1;
                    CS_HIP(ctx_, hipMemsetAsync(oflags_.p, 0, size_t(L) * sizeof(int32_t), ctx_->stream));
                    if (rank_ > 0)
                        CS_TRY(cstone_hip_find_overlaps(ctx_, curve_, kb, fPrefixes_.p, fChild_.as<int32_t>(),
                                                        fItl_.as<int32_t>(), fTree_.p, allBoxes_.as<int32_t>(),
                                                        int(size_t(rank_) * maxBoxes), first, last, oflags_.as<int32_t>()));
                    if (rank_ + 1 < P_)
                        CS_TRY(cstone_hip_find_overlaps(ctx_, curve_, kb, fPrefixes_.p, fChild_.as<int32_t>(),
                                                        fItl_.as<int32_t>(), fTree_.p,
                                                        allBoxes_.as<int32_t>() + size_t(rank_ + 1) * maxBoxes * 8,
                                                        int(size_t(P_ - rank_ - 1) * maxBoxes), first, last,
                                                        oflags_.as<int32_t>()));
                    const size_t items = size_t(np) * nLocal;
                    CS_TRY(cnt_.ensure(ctx_, (items + 1) * sizeof(uint32_t)));
                    hipLaunchKernelGGL(peerCountsKernel, gridFor(nLocal, 256), 256, 0, ctx_->stream, oflags_.as<int32_t>(),
                                       layout_.as<uint32_t>(), first, last, P_, rank_, cnt_.as<uint32_t>());
                    CS_TRY(exclusiveScanWithTotal(cnt_.as<uint32_t>(), int(items), total));
                    // my row of the count matrix goes from the device into the all-gather of the rows; one read-back
                    uint64_t* row  = scal_.as<uint64_t>() + 32;
                    uint64_t* rows = reinterpret_cast<uint64_t*>(scal_.as<char>() + 4096);
                    hipLaunchKernelGGL(peerTotalsKernel, 1, 64, 0, ctx_->stream, cnt_.as<uint32_t>(), total, nLocal, np,
                                       rank_, row);
                    CS_TRY(callComm(comm_.all_gather(comm_.user, row, rows, size_t(P_) * 8), "all_gather (halo counts)"));
                    hmatrix.assign(size_t(P_) * P_, 0);
                    CS_TRY(toHost(hmatrix.data(), rows, size_t(P_) * P_ * 8));
                    haveMatrix = true;
                    for (int p = 0; p < P_; ++p)
                    {
                        hsCounts[p] = hmatrix[size_t(rank_) * P_ + p];
                        selTotal += hsCounts[p];
                    }
                    if (selTotal)
                    {
                        CS_TRY(sel_.ensure(ctx_, selTotal * sizeof(uint32_t)));
                        hipLaunchKernelGGL(peerFillKernel, gridFor(items, 16), 256, 0, ctx_->stream, oflags_.as<int32_t>(),
                                           layout_.as<uint32_t>(), cnt_.as<uint32_t>(), first, last, P_, rank_,
                                           sel_.as<uint32_t>());
                    }
                }
                else
                {
                CS_TRY(cnt_.ensure(ctx_, size_t(nLocal + 1) * sizeof(uint32_t)));
                for (int p = 0; p < P_; ++p)
                {
                    if (p == rank_ || boxCounts[p] == 0) continue;
                    CS_HIP(ctx_, hipMemsetAsync(oflags_.p, 0, size_t(L) * sizeof(int32_t), ctx_->stream));
                    CS_TRY(cstone_hip_find_overlaps(ctx_, curve_, kb, fPrefixes_.p, fChild_.as<int32_t>(),
                                                    fItl_.as<int32_t>(), fTree_.p,
                                                    allBoxes_.as<int32_t>() + size_t(p) * maxBoxes * 8, int(boxCounts[p]),
                                                    first, last, oflags_.as<int32_t>()));
                    hipLaunchKernelGGL(flaggedCountsKernel, gridFor(nLocal, 256), 256, 0, ctx_->stream,
                                       oflags_.as<int32_t>(), layout_.as<uint32_t>(), first, last, cnt_.as<uint32_t>());
                    CS_TRY(exclusiveScanWithTotal(cnt_.as<uint32_t>(), nLocal, total));
                    uint32_t tp = 0;
                    CS_TRY(toHost(&tp, total, 4));
                    if (tp)
                    {
                        CS_TRY(sel_.ensure(ctx_, (selTotal + tp) * sizeof(uint32_t), true));
                        hipLaunchKernelGGL(fillIndicesKernel, gridFor(nLocal, 16), 256, 0, ctx_->stream,
                                           oflags_.as<int32_t>(), layout_.as<uint32_t>(), cnt_.as<uint32_t>(), first, last,
                                           sel_.as<uint32_t>() + selTotal);
                    }
                    hsCounts[p] = tp;
                    selTotal += tp;
                }
                }
                if (!haveMatrix) CS_TRY(countMatrix(hsCounts, hmatrix));
            }
            for (uint64_t v : hmatrix)
                haloAny += v;
            for (int p = 0; p < P_; ++p)
            {
                uint64_t r = hmatrix[size_t(p) * P_ + rank_];
                (p < rank_ ? nlo : nhi) += (p == rank_ ? 0 : r);
                hSendBytes[p] = hsCounts[p] * 4 * sizeof(T);
                hRecvBytes[p] = (p == rank_ ? 0 : r) * 4 * sizeof(T);
            }

        }
        tick("6 halo discovery");
        if (placeForked_)
        {
            // x, y, z of the assigned block are needed from here on (a block that has to be moved, the halo exchange)
            CS_HIP(ctx_, hipStreamWaitEvent(ctx_->stream, ctx_->evJoin, 0));
            placeForked_  = false;
            ctx_->auxBusy = false;
        }
        // ---- room for the halos on both sides of the assigned block
        const uint64_t total = nlo + nm + nhi;
        uint64_t off = M - std::min(M, nlo); // start of the arrays handed out
        if (nlo > M || M + nm + nhi > cap)
        {
            // the margins were too small (first syncs, abrupt changes): move the block once, through a scratch copy
            const uint64_t M2 = (nlo + 3) & ~uint64_t(3), cap2 = M2 + nm + nhi;
            auto shift = [&](DevBuf& buf, size_t elem) -> int
            {
                CS_TRY(moveTmp_.ensure(ctx_, nm * elem));
                CS_HIP(ctx_, hipMemcpyAsync(moveTmp_.p, buf.as<char>() + M * elem, nm * elem, hipMemcpyDeviceToDevice,
                                            ctx_->stream));
                CS_TRY(buf.ensure(ctx_, cap2 * elem));
                CS_HIP(ctx_, hipMemcpyAsync(buf.as<char>() + M2 * elem, moveTmp_.p, nm * elem, hipMemcpyDeviceToDevice,
                                            ctx_->stream));
                return CSTONE_OK;
            };
            CS_TRY(shift(o.keys, sizeof(K)));
            for (DevBuf* b : {&o.x, &o.y, &o.z, &o.h})
                CS_TRY(shift(*b, sizeof(T)));
            for (int q = 0; q < numProps; ++q)
                CS_TRY(shift(o.props[q], size_t(propBytes[q])));
            off = M2 - nlo;
        }
        const uint64_t A = off + nlo; // first assigned slot
        prevLo_ = nlo, prevHi_ = nhi;
        tick("7 margins");
        // ---- C5: halo exchange
        if (useLet_ && P_ > 1)
        {
            // the particle buffers start at `off`: halos below | assigned | halos above, in the order of the focus tree's
            // leaves (R/domain/domain.hpp:522-540: exchangeHalos(x, y, z, h), then the keys of the halo particles)
            // (x, y, z and h travel together: one message per peer instead of four)
            void* xyzh[4] = {o.x.as<T>() + off, o.y.as<T>() + off, o.z.as<T>() + off, o.h.as<T>() + off};
            CS_TRY(let_->exchangeHalosRows(xyzh, 4, int(sizeof(T))));
            if (nlo)
            {
                CS_HIP(ctx_, hipMemsetAsync(o.keys.as<K>() + off, 0, nlo * sizeof(K), ctx_->stream));
                CS_TRY(cstone_hip_compute_sfc_keys(ctx_, curve_, kb, rb, o.x.as<T>() + off, o.y.as<T>() + off,
                                                   o.z.as<T>() + off, o.keys.as<K>() + off, nlo, &box_));
            }
            if (nhi)
            {
                CS_HIP(ctx_, hipMemsetAsync(o.keys.as<K>() + A + nm, 0, nhi * sizeof(K), ctx_->stream));
                CS_TRY(cstone_hip_compute_sfc_keys(ctx_, curve_, kb, rb, o.x.as<T>() + A + nm, o.y.as<T>() + A + nm,
                                                   o.z.as<T>() + A + nm, o.keys.as<K>() + A + nm, nhi, &box_));
            }
        }
        else if (P_ > 1 && haloAny)
        {
            CS_TRY(sendRows_.ensure(ctx_, std::max<size_t>(selTotal, 1) * 4 * sizeof(T)));
            CS_TRY(recvRows_.ensure(ctx_, std::max<size_t>(nlo + nhi, 1) * 4 * sizeof(T)));
            if (selTotal)
                hipLaunchKernelGGL(packRowsKernel<T>, gridFor(selTotal, 256), 256, 0, ctx_->stream, sel_.as<uint32_t>(),
                                   size_t(selTotal), o.x.as<T>() + A, o.y.as<T>() + A, o.z.as<T>() + A,
                                   o.h.as<T>() + A, sendRows_.as<T>());
            CS_TRY(callComm(comm_.all_to_all_v(comm_.user, sendRows_.p, hSendBytes.data(), recvRows_.p,
                                               hRecvBytes.data()),
                            "all_to_all_v (halos)"));
            if (nlo)
                hipLaunchKernelGGL(unpackRowsKernel<T>, gridFor(nlo, 256), 256, 0, ctx_->stream, recvRows_.as<T>(),
                                   size_t(nlo), o.x.as<T>() + off, o.y.as<T>() + off, o.z.as<T>() + off, o.h.as<T>() + off);
            if (nhi)
                hipLaunchKernelGGL(unpackRowsKernel<T>, gridFor(nhi, 256), 256, 0, ctx_->stream,
                                   recvRows_.as<T>() + 4 * nlo, size_t(nhi), o.x.as<T>() + A + nm,
                                   o.y.as<T>() + A + nm, o.z.as<T>() + A + nm, o.h.as<T>() + A + nm);
            // keys of the halo particles (encode skips entries that hold the remove marker: clear first)
            if (nlo)
            {
                CS_HIP(ctx_, hipMemsetAsync(o.keys.as<K>() + off, 0, nlo * sizeof(K), ctx_->stream));
                CS_TRY(cstone_hip_compute_sfc_keys(ctx_, curve_, kb, rb, o.x.as<T>() + off, o.y.as<T>() + off,
                                                   o.z.as<T>() + off, o.keys.as<K>() + off, nlo, &box_));
            }
            if (nhi)
            {
                CS_HIP(ctx_, hipMemsetAsync(o.keys.as<K>() + A + nm, 0, nhi * sizeof(K), ctx_->stream));
                CS_TRY(cstone_hip_compute_sfc_keys(ctx_, curve_, kb, rb, o.x.as<T>() + A + nm, o.y.as<T>() + A + nm,
                                                   o.z.as<T>() + A + nm, o.keys.as<K>() + A + nm, nhi, &box_));
            }
        }
        CS_HIP(ctx_, hipGetLastError());
        tick("8 halo exchange");

        // the particle routes of this sync (reapplySync)
        rsN_ = n, rsNa_ = na, rsNb_ = nb, rsSend_ = mSend, rsMoved_ = movedAny, rsKeptOffset_ = cut[rank_];
        rsSendCounts_ = sendCounts;
        rsRecvCounts_.assign(P_, 0);
        for (int p = 0; p < P_; ++p)
            rsRecvCounts_[p] = p == rank_ ? 0 : matrix[size_t(p) * P_ + rank_];
        haloAnyLast_ = haloAny;
        haloSend_ = hsCounts, haloRecvLo_ = nlo, haloRecvHi_ = nhi, haloAssigned_ = nm, haloSel_ = selTotal;
        haloRecv_.assign(P_, 0);
        for (int p = 0; p < P_; ++p)
            haloRecv_[p] = p == rank_ ? 0 : hmatrix[size_t(p) * P_ + rank_];

        firstCall_                     = false;
        view_.start_index              = uint32_t(nlo);
        view_.end_index                = uint32_t(nlo + nm);
        view_.num_particles_with_halos = uint32_t(total);
        view_.box                      = box_;
        view_.keys = o.keys.as<K>() + off;
        view_.x = o.x.as<T>() + off, view_.y = o.y.as<T>() + off, view_.z = o.z.as<T>() + off, view_.h = o.h.as<T>() + off;
        for (int q = 0; q < MAX_PROPS; ++q)
            view_.props[q] = q < numProps ? static_cast<const void*>(o.props[q].as<char>() + off * propBytes[q]) : nullptr;
        view_.num_global_leaves = gLeaves_, view_.num_focus_leaves = fLeaves_;
        view_.global_leaves = gTree_.p, view_.global_counts = gCounts_.as<uint32_t>();
        view_.focus_leaves = fTree_.p, view_.focus_leaf_counts = fCounts_.as<uint32_t>();
        view_.start_cell = 0, view_.end_cell = fLeaves_, view_.num_peers = 0;
        view_.layout = nullptr, view_.halo_flags = nullptr;
        if (useLet_)
        {
            view_.focus_leaves = let_->leaves(), view_.focus_leaf_counts = let_->leafCounts();
            view_.start_cell = let_->startCell(), view_.end_cell = let_->endCell();
            view_.num_peers  = int32_t(let_->peers().size());
            view_.layout = let_->layout(), view_.halo_flags = let_->haloFlags();
        }
        view_.range_start = uint64_t(assignment_[rank_]), view_.range_end = uint64_t(assignment_[rank_ + 1]);
        view_.particles_sent      = mSend;
        view_.halos_received      = nlo + nhi;
        view_.halos_sent          = selTotal;
        view_.halo_boxes_exported = numMyBoxes;
        view_.resorts             = uint64_t(resorts_);
        // the sticky device-side error word: a sync that tripped a device-side check must not report success
        return cstone_hip_ctx_sync(ctx_);
    }

private:
    //! CSTONE_MR_TIMING=1: synchronising wall clock per phase, printed by rank 0 when the domain is destroyed
    void tick(const char* name)
    {
        if (!timing_) return;
        (void)hipStreamSynchronize(ctx_->stream);
        auto now = std::chrono::steady_clock::now();
        if (name) phase_[name] += std::chrono::duration<double>(now - t0_).count();
        t0_ = now;
    }

    //! a failure of THIS rank that the peers must learn about before anybody returns (see the top of sync())
    void setPending(int code, const char* fmt, ...)
    {
        if (pending_) return;
        char buf[384];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        pending_    = code;
        pendingMsg_ = buf;
    }

    /*! below some 6e6 particles per rank the chain of small launches of the re-sort and its read-back cost more than the
     *  digit passes they replace: measured in round 3, every particle drifting, re-sorted against radix-sorted: 0.73 / 0.60
     *  ms per sync at 1e6, 0.91 / 0.90 at 3e6, 0.90 / 0.95 at 6e6, 1.49 / 1.53 at 1.25e7, 2.62 / 2.66 at 2.5e7, 3.36 / 3.7
     *  at 5e7 (tools/mr_bench.py --rccl) */
    static size_t resortMinParticles()
    {
        static const size_t v = []
        {
            const char* e = std::getenv("CSTONE_MR_RESORT_MIN");
            return e ? size_t(std::strtoull(e, nullptr, 10)) : size_t(6) << 20;
        }();
        return v;
    }

    //! tests: CSTONE_MR_FAIL_AT="<rank>:<point>" makes that rank fail at the named point of sync()
    void injectFailure(const char* point)
    {
#ifdef CSTONE_TEST_HOOKS // (the product library carries no fault injection: lib/libcstone_hip_hooks.so is the tests' build)
        const char* e = std::getenv("CSTONE_MR_FAIL_AT");
        if (!e) return;
        const std::string want = std::to_string(rank_) + ":" + point;
        if (want == e) setPending(CSTONE_E_INTERNAL, "domain_mr_sync: failure injected at '%s'", point);
#else
        (void)point;
#endif
    }

    //! every rank has seen that rank `culprit` failed: all of them return an error from the same point of the sync
    int agreed(int culprit)
    {
        const int code      = culprit == rank_ && pending_ ? pending_ : CSTONE_E_INTERNAL;
        const std::string m = culprit == rank_ ? pendingMsg_ : "rank " + std::to_string(culprit) + " reported a failure";
        pending_            = 0;
        if (toggled_) cur_ ^= 1, toggled_ = false; // the client's arrays of the last good sync stay untouched by the next one
        resortTree_ = nullptr, resortLeaves_ = 0, layoutParticles_ = 0; // an abandoned sync is nothing to re-sort from
        return fail(ctx_, code, "%s (the sync was abandoned on every rank)", m.c_str());
    }

    int callComm(int rc, const char* what)
    {
        if (rc != 0) return fail(ctx_, CSTONE_E_INTERNAL, "collective %s failed with code %d", what, rc);
        return CSTONE_OK;
    }

    int toHost(void* dst, const void* src, size_t bytes)
    {
        return copyToHost(ctx_, dst, src, bytes); // (any host memory; synchronises the stream)
    }

    int ensureSortScratch(size_t n)
    {
        CS_TRY(keysAlt_.ensure(ctx_, n * sizeof(K)));
        CS_TRY(orderAlt_.ensure(ctx_, n * sizeof(uint32_t)));
        CS_TRY(sortTmp_.ensure(ctx_, cstone_hip_sort_pairs_temp_bytes(kb, n)));
        return CSTONE_OK;
    }

    //! in-place exclusive scan of n values, grand total to *totalDev
    int exclusiveScanWithTotal(uint32_t* data, int n, uint32_t* totalDev)
    {
        if (n == 0)
        {
            CS_HIP(ctx_, hipMemsetAsync(totalDev, 0, 4, ctx_->stream));
            return CSTONE_OK;
        }
        CS_TRY(arenaReserve(ctx_, scanArenaBytes(size_t(n))));
        int rc = scanU32(ctx_, data, data, size_t(n), 0u, false, totalDev);
        arenaReset(ctx_);
        return rc;
    }

    //! m[src * P + dst] of every rank's counts
    int countMatrix(const std::vector<uint64_t>& mine, std::vector<uint64_t>& matrix)
    {
        matrix.assign(size_t(P_) * P_, 0);
        if (P_ == 1)
        {
            matrix[0] = mine[0];
            return CSTONE_OK;
        }
        uint64_t* send = scal_.as<uint64_t>() + 32;
        uint64_t* recv = reinterpret_cast<uint64_t*>(scal_.as<char>() + 4096);
        CS_HIP(ctx_, hipMemcpyAsync(send, mine.data(), size_t(P_) * 8, hipMemcpyHostToDevice, ctx_->stream));
        CS_TRY(callComm(comm_.all_gather(comm_.user, send, recv, size_t(P_) * 8), "all_gather (counts)"));
        return toHost(matrix.data(), recv, size_t(P_) * P_ * 8);
    }

    // ---- C1
    int updateBox(const T* x, const T* y, const T* z, size_t n)
    {
        // periodic axes keep their limits (R/sfc/box_mpi.hpp:81-121): with three of them there is nothing to measure
        if (box_.bc[0] == 1 && box_.bc[1] == 1 && box_.bc[2] == 1) return CSTONE_OK;
        // (lo, -hi) per axis stay on the device from the reduction over the particles through the MIN all-reduce over
        // the ranks; one read-back at the end
        static const double nothing[6] = {std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity(),
                                          std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity(),
                                          std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity()};
        double* dev = scal_.as<double>();
        if (n)
        {
            const void* arrays[3] = {x, y, z};
            CS_TRY(minMaxCoordinatesDev(ctx_, rb, arrays, 3, n, dev));
        }
        else { CS_HIP(ctx_, hipMemcpyAsync(dev, nothing, sizeof nothing, hipMemcpyHostToDevice, ctx_->stream)); }
        cstone_box next;
        CS_TRY(reduceBox(dev, &next));
        box_ = next;
        return CSTONE_OK;
    }

    /*! dev: (lo, -hi) of this rank's particles per axis, six doubles on the device, room for a seventh.  All-reduces them
     *  with the status word, reads them back and applies the box rule to box_ -> *next (box_ itself is not changed). */
    //! seventh value of the box reduction: the status of this rank (0, or -(rank + 1) if it has a failure pending)
    double statusWord() const { return pending_ ? -double(rank_ + 1) : 0.0; }

    /*! MIN over the ranks of dev[0..6] = (min, -max) per axis + status.  statusSet: the operand is complete (the kernel
     *  that wrote the extents wrote the status too); counters: four ints behind the operand come back in the same copy */
    int reduceBox(double* dev, cstone_box* next, bool statusSet = false, int* counters = nullptr)
    {
        if (!statusSet)
        {
            const double status = statusWord();
            CS_TRY(cstone_hip_upload(ctx_, dev + 6, &status, sizeof status));
        }
        if (P_ > 1) CS_TRY(callComm(comm_.all_reduce(comm_.user, dev, 7, 0, 1), "all_reduce (box)"));
        double ext[9];
        CS_TRY(toHost(ext, dev, counters ? sizeof ext : 7 * sizeof(double)));
        if (counters) std::memcpy(counters, ext + 7, 4 * sizeof(int));
        if (ext[6] < 0) return agreed(int(-ext[6]) - 1);
        *next = box_;
        double fit[6];
        for (int d = 0; d < 3; ++d)
        {
            fit[2 * d] = ext[2 * d], fit[2 * d + 1] = -ext[2 * d + 1];
            if (box_.bc[d] == 1) fit[2 * d] = box_.lim[2 * d], fit[2 * d + 1] = box_.lim[2 * d + 1];
        }
        if (firstCall_) { std::copy(fit, fit + 6, next->lim); }
        else
        {
            // limitBoxShrinking (R/sfc/box.hpp:415-431), evaluated in T like the reference
            const T shrink = T(0.05);
            for (int d = 0; d < 3; ++d)
            {
                T lo = T(box_.lim[2 * d]), hi = T(box_.lim[2 * d + 1]);
                T len                = hi - lo;
                next->lim[2 * d]     = std::min(T(fit[2 * d]), T(lo + shrink * len));
                next->lim[2 * d + 1] = std::max(T(fit[2 * d + 1]), T(hi - shrink * len));
            }
        }
        return CSTONE_OK;
    }

    int ensureTree(DevBuf& tree, DevBuf& counts, int& cap, int need)
    {
        if (need <= cap) return CSTONE_OK;
        int newCap = std::max(need, int(cap * 1.5));
        CS_TRY(tree.ensure(ctx_, size_t(newCap + 1) * sizeof(K), true));
        CS_TRY(counts.ensure(ctx_, size_t(newCap) * sizeof(uint32_t), true));
        cap = newCap;
        return CSTONE_OK;
    }

    // ---- C2: GlobalAssignment ctor + assign() tree part
    int updateGlobalTree(size_t n)
    {
        if (gLeaves_ == 0)
        {
            std::vector<K> init = initialGlobalTree<K>(P_);
            int leaves          = int(init.size()) - 1;
            CS_TRY(ensureTree(gTree_, gCounts_, gCap_, std::max(4096, 2 * leaves)));
            std::vector<uint32_t> c0(leaves, bucket_ - 1);
            CS_HIP(ctx_, hipMemcpy(gTree_.p, init.data(), init.size() * sizeof(K), hipMemcpyHostToDevice));
            CS_HIP(ctx_, hipMemcpy(gCounts_.p, c0.data(), c0.size() * 4, hipMemcpyHostToDevice));
            gLeaves_ = leaves;
        }
        if (!firstCall_ && hostGlobalStep_ && int(gLeavesHost_.size()) == gLeaves_ + 1 && int(gCountsHost_.size()) == gLeaves_)
        {
            // later calls take exactly ONE update step (assignment.hpp:92-98).  The tree is small and replicated and the
            // host holds its leaves and the all-reduced counts of the last sync (assign() read them): the decision and the
            // new leaf array are made here; the device counts this rank's keys, the counts are reduced, and assign() reads
            // them back together with everything else the sync needs at that point -- no read-back in between
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
            gTreeSame_ = same;
            CS_TRY(cstone_hip_compute_node_counts(ctx_, kb, gTree_.p, gCounts_.as<uint32_t>(), gLeaves_, keys_.p, n,
                                                  0xFFFFFFFFu));
            if (P_ > 1)
            {
                CS_TRY(gLocalCounts_.ensure(ctx_, size_t(gLeaves_) * sizeof(uint32_t)));
                CS_HIP(ctx_, hipMemcpyAsync(gLocalCounts_.p, gCounts_.p, size_t(gLeaves_) * sizeof(uint32_t),
                                            hipMemcpyDeviceToDevice, ctx_->stream));
                CS_TRY(callComm(comm_.all_reduce(comm_.user, gCounts_.p, size_t(gLeaves_), 1, 0), "all_reduce (counts)"));
                hipLaunchKernelGGL(maxWithLocalKernel, gridFor(size_t(gLeaves_), 256), 256, 0, ctx_->stream,
                                   gCounts_.as<uint32_t>(), gLocalCounts_.as<uint32_t>(), gLeaves_);
            }
            gLeavesOnHost_ = true;
            return CSTONE_OK;
        }
        gLeavesOnHost_ = false;
        int steps = 0;
        while (true)
        {
            int leaves = gLeaves_, conv = 0;
            int rc = cstone_hip_update_octree(ctx_, kb, keys_.p, n, bucket_, gTree_.p, gCounts_.as<uint32_t>(), &leaves,
                                              gCap_, 0xFFFFFFFFu, &conv);
            if (rc == CSTONE_E_CAPACITY)
            {
                CS_TRY(ensureTree(gTree_, gCounts_, gCap_, leaves + 1));
                continue;
            }
            CS_TRY(rc);
            gLeaves_ = leaves;
            gTreeSame_ = steps == 0 && !firstCall_ && conv != 0; // the one step of a later sync kept every leaf
            if (P_ > 1)
            {
                CS_TRY(gLocalCounts_.ensure(ctx_, size_t(leaves) * sizeof(uint32_t)));
                CS_HIP(ctx_, hipMemcpyAsync(gLocalCounts_.p, gCounts_.p, size_t(leaves) * sizeof(uint32_t),
                                            hipMemcpyDeviceToDevice, ctx_->stream));
                CS_TRY(callComm(comm_.all_reduce(comm_.user, gCounts_.p, size_t(leaves), 1, 0), "all_reduce (counts)"));
                hipLaunchKernelGGL(maxWithLocalKernel, gridFor(size_t(leaves), 256), 256, 0, ctx_->stream,
                                   gCounts_.as<uint32_t>(), gLocalCounts_.as<uint32_t>(), leaves);
            }
            ++steps;
            // later calls: exactly one step; first call: one step, then `while (!update)` (assignment.hpp:92-98)
            if (!firstCall_ || (steps >= 2 && conv)) break;
            if (steps > 64) return fail(ctx_, CSTONE_E_INTERNAL, "global tree does not converge");
        }
        return CSTONE_OK;
    }

    //! the read-back of the global counts (and of the leaf array, when the device made it): queued into the pinned
    //! block, not waited for; takeGlobalTreeReadBack() behind the synchronisation
    int queueGlobalTreeReadBack()
    {
        CS_TRY(pin_.reserve(ctx_, size_t(gLeaves_) * 4 + size_t(gLeaves_ + 1) * sizeof(K) + size_t(P_ + 1) * (P_ + 2) * 8 + 1024));
        pinCounts_ = static_cast<uint32_t*>(pin_.take(size_t(gLeaves_) * 4));
        CS_TRY(copyToPinned(ctx_, pinCounts_, gCounts_.p, size_t(gLeaves_) * 4));
        pinLeaves_ = nullptr;
        if (!gLeavesOnHost_)
        {
            pinLeaves_ = static_cast<K*>(pin_.take(size_t(gLeaves_ + 1) * sizeof(K)));
            CS_TRY(copyToPinned(ctx_, pinLeaves_, gTree_.p, size_t(gLeaves_ + 1) * sizeof(K)));
        }
        return CSTONE_OK;
    }
    void takeGlobalTreeReadBack()
    {
        gCountsHost_.assign(pinCounts_, pinCounts_ + gLeaves_);
        if (pinLeaves_) gLeavesHost_.assign(pinLeaves_, pinLeaves_ + gLeaves_ + 1);
    }

    //! makeSfcAssignment + limitBoundaryShifts from the host copies of the global tree (after the read-back completed)
    int assign()
    {
        const std::vector<uint32_t>& counts = gCountsHost_;
        const std::vector<K>& leaves        = gLeavesHost_;
        std::vector<int> bins = uniformBinsHost(counts, P_);
        std::vector<K> fresh(P_ + 1);
        for (int r = 0; r <= P_; ++r)
            fresh[r] = leaves[bins[r]];
        if (int(assignment_.size()) == P_ + 1)
        {
            // limitBoundaryShifts (domaindecomp.hpp:140-172): a boundary moves at most into a neighbour's old range
            std::vector<K> old = assignment_;
            for (int r = 1; r < P_; ++r)
                fresh[r] = std::min(std::max(fresh[r], old[r - 1]), old[r + 1]);
        }
        assignment_ = fresh;
        return CSTONE_OK;
    }

    int updateFocusTree(const K* keysM, size_t nm)
    {
        if (fLeaves_ == 0)
        {
            int cap = std::max<int>(4096, int(4 * nm / std::max(1u, bucketFocus_)) + 4096);
            while (true)
            {
                CS_TRY(ensureTree(fTree_, fCounts_, fCap_, cap));
                int leaves = 0, iters = 0;
                int rc = cstone_hip_compute_octree(ctx_, kb, keysM, nm, bucketFocus_, fTree_.p, fCounts_.as<uint32_t>(),
                                                   &leaves, fCap_, 0xFFFFFFFFu, &iters);
                if (rc == CSTONE_E_CAPACITY)
                {
                    cap = leaves + 1;
                    continue;
                }
                CS_TRY(rc);
                fLeaves_   = leaves;
                treeSteps_ = 0;
                return CSTONE_OK;
            }
        }
        ++treeSteps_;
        while (true)
        {
            int leaves = fLeaves_, conv = 0;
            int rc = cstone_hip_update_octree(ctx_, kb, keysM, nm, bucketFocus_, fTree_.p, fCounts_.as<uint32_t>(),
                                              &leaves, fCap_, 0xFFFFFFFFu, &conv);
            if (rc == CSTONE_E_CAPACITY)
            {
                CS_TRY(ensureTree(fTree_, fCounts_, fCap_, leaves + 1));
                continue;
            }
            CS_TRY(rc);
            fLeaves_ = leaves;
            return CSTONE_OK;
        }
    }

    /*! The rank's SFC range must start and end on leaf boundaries of its own tree (the job of enforceKeys in the
     *  reference's focus tree, R/focus/rebalance.hpp:199-266): a leaf that straddles a range boundary is replaced by the
     *  coarsest set of octree nodes that resolves the boundary key.  Such leaves are mostly empty, so the count-driven
     *  update merges them again and the split is redone at every sync: a copy of the leaf array and a recount.
     *  One device query and ONE read-back serve both boundaries; the leaf range [*first, *last) of the rank in the
     *  resulting tree follows on the host. */
    int enforceBoundaries(const K* keysM, size_t nm, int* first, int* last)
    {
        const uint64_t end = uint64_t(endKey<K>());
        const uint64_t b[2] = {uint64_t(assignment_[rank_]), uint64_t(assignment_[rank_ + 1])};
        uint64_t* dq = reinterpret_cast<uint64_t*>(scal_.as<char>() + 1024);
        hipLaunchKernelGGL(containingLeavesKernel<K>, 1, 64, 0, ctx_->stream, fTree_.as<K>(), fLeaves_,
                           K(std::min(b[0], end)), K(std::min(b[1], end)), dq);
        uint64_t q[6];
        CS_TRY(toHost(q, dq, sizeof q));
        const int L = fLeaves_;
        // leaves to replace, in ascending order: (index, start, end, boundary keys strictly inside)
        struct Cut
        {
            int idx;
            uint64_t s, e;
            std::vector<uint64_t> keys;
        };
        std::vector<Cut> cuts;
        for (int k = 0; k < 2; ++k)
        {
            const int idx = int(q[3 * k]);
            if (b[k] == 0 || b[k] >= end || idx >= L || q[3 * k + 1] == b[k]) continue; // already a leaf boundary
            if (!cuts.empty() && cuts.back().idx == idx) { cuts.back().keys.push_back(b[k]); }
            else { cuts.push_back({idx, q[3 * k + 1], q[3 * k + 2], {b[k]}}); }
        }
        // position of a boundary key in the new leaf array: leaves before it in the old array + what the covers add
        coverHost_.clear();
        coverDeepest_ = 0;
        std::vector<int> coverBegin, coverSize;
        for (const Cut& c : cuts)
        {
            coverBegin.push_back(int(coverHost_.size()));
            uint64_t a = c.s;
            for (uint64_t key : c.keys)
            {
                appendCover(coverHost_, a, key);
                a = key;
            }
            appendCover(coverHost_, a, c.e);
            coverSize.push_back(int(coverHost_.size()) - coverBegin.back());
            // how deep the inserted leaves go (the sort of the node keys in buildFocusOctree leaves out the digit passes
            // above the deepest level of the tree)
            for (int i = 0; i < coverSize.back(); ++i)
            {
                const uint64_t lo   = uint64_t(coverHost_[coverBegin.back() + i]);
                const uint64_t hi   = i + 1 < coverSize.back() ? uint64_t(coverHost_[coverBegin.back() + i + 1]) : c.e;
                const uint64_t span = hi - lo;
                int level           = 0;
                while (level < int(maxLevel<K>()) && (uint64_t(1) << (3 * (maxLevel<K>() - level))) > span)
                    ++level;
                coverDeepest_ = std::max(coverDeepest_, level);
            }
        }
        auto newIndexOf = [&](uint64_t key, int containing, bool aligned) -> int
        {
            if (key == 0) return 0;
            int shift = 0;
            for (size_t c = 0; c < cuts.size(); ++c)
            {
                if (cuts[c].idx < containing) { shift += coverSize[c] - 1; }
                else if (cuts[c].idx == containing && !aligned)
                {
                    const K* cb = coverHost_.data() + coverBegin[c];
                    return containing + shift + int(std::find(cb, cb + coverSize[c], K(key)) - cb);
                }
            }
            return containing + shift;
        };
        int extra = 0;
        for (int sz : coverSize)
            extra += sz - 1;
        const int newL = L + extra;
        *first = newIndexOf(b[0], int(q[0]), b[0] == 0 || q[1] == b[0]);
        *last  = b[1] >= end ? newL : newIndexOf(b[1], int(q[3]), q[4] == b[1]);
        if (!cuts.empty())
        {
            CS_TRY(ensureTree(fTree_, fCounts_, fCap_, newL));
            CS_TRY(fTmp_.ensure(ctx_, size_t(newL + 1) * sizeof(K)));
            K* t        = fTmp_.as<K>();
            const K* ft = fTree_.as<K>();
            int src = 0, dst = 0; // old leaves [src, ...) still to copy, new position dst
            for (size_t c = 0; c < cuts.size(); ++c)
            {
                const int keep = cuts[c].idx - src;
                if (keep)
                    CS_HIP(ctx_, hipMemcpyAsync(t + dst, ft + src, size_t(keep) * sizeof(K), hipMemcpyDeviceToDevice,
                                                ctx_->stream));
                dst += keep;
                // coverHost_ is a member: it outlives the copy (the next sync refills it behind a stream sync)
                CS_HIP(ctx_, hipMemcpyAsync(t + dst, coverHost_.data() + coverBegin[c], size_t(coverSize[c]) * sizeof(K),
                                            hipMemcpyHostToDevice, ctx_->stream));
                dst += coverSize[c];
                src = cuts[c].idx + 1;
            }
            CS_HIP(ctx_, hipMemcpyAsync(t + dst, ft + src, size_t(L + 1 - src) * sizeof(K), hipMemcpyDeviceToDevice,
                                        ctx_->stream));
            CS_HIP(ctx_, hipMemcpyAsync(fTree_.p, t, size_t(newL + 1) * sizeof(K), hipMemcpyDeviceToDevice,
                                        ctx_->stream));
            fLeaves_ = newL;
            CS_TRY(cstone_hip_compute_node_counts(ctx_, kb, fTree_.p, fCounts_.as<uint32_t>(), fLeaves_, keysM, nm,
                                                  0xFFFFFFFFu));
        }
        if (*first < 0 || *last > fLeaves_ || *last <= *first)
            return fail(ctx_, CSTONE_E_INTERNAL, "domain_mr_sync: bad leaf range [%d, %d) of %d", *first, *last, fLeaves_);
        return CSTONE_OK;
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
        // leaves of the new tree: at most one level below the deepest of the previous sync's tree (one update step), or as
        // deep as the leaves inserted at the range boundaries; the first tree of a domain may be of any depth
        const int deepest = (treeSteps_ > 0 && prevMaxLeafLevel_ >= 0) ? std::max(prevMaxLeafLevel_ + 1, coverDeepest_)
                                                                       : int(maxLevel<K>());
        return buildLinkedOctree(ctx_, kb, fTree_.p, L, fPrefixes_.p, fChild_.as<int32_t>(), fParents_.as<int32_t>(),
                                 fLevelRange_.as<int32_t>(), fItl_.as<int32_t>(), fLti_.as<int32_t>(), deepest);
    }

    int buildNsTree()
    {
        const K* keysAll = static_cast<const K*>(view_.keys);
        const size_t n   = view_.num_particles_with_halos;
        if (nsLeaves_ == 0)
        {
            int cap = std::max<int>(4096, int(4 * n / std::max(1u, bucketFocus_)) + 4096);
            while (true)
            {
                CS_TRY(ensureTree(nsTree_, nsCounts_, nsCap_, cap));
                int leaves = 0, iters = 0;
                int rc = cstone_hip_compute_octree(ctx_, kb, keysAll, n, bucketFocus_, nsTree_.p,
                                                   nsCounts_.as<uint32_t>(), &leaves, nsCap_, 0xFFFFFFFFu, &iters);
                if (rc == CSTONE_E_CAPACITY)
                {
                    cap = leaves + 1;
                    continue;
                }
                CS_TRY(rc);
                nsLeaves_ = leaves;
                break;
            }
        }
        else
        {
            // the particles moved a little since the last request: a few update steps (each one level of splits or
            // merges, R/tree/csarray.hpp:430-448); the search stays correct if the last one still changed something
            for (int it = 0, conv = 0; it < 4 && !conv;)
            {
                int leaves = nsLeaves_;
                int rc = cstone_hip_update_octree(ctx_, kb, keysAll, n, bucketFocus_, nsTree_.p, nsCounts_.as<uint32_t>(),
                                                  &leaves, nsCap_, 0xFFFFFFFFu, &conv);
                if (rc == CSTONE_E_CAPACITY)
                {
                    CS_TRY(ensureTree(nsTree_, nsCounts_, nsCap_, leaves + 1));
                    continue;
                }
                CS_TRY(rc);
                nsLeaves_ = leaves;
                ++it;
            }
        }
        const NodeIdx L = nsLeaves_, M = L + (L - 1) / 7;
        CS_TRY(nsLayout_.ensure(ctx_, size_t(L + 1) * sizeof(uint32_t)));
        CS_HIP(ctx_, hipMemsetAsync(nsLayout_.p, 0, sizeof(uint32_t), ctx_->stream));
        CS_TRY(cstone_hip_inclusive_scan_u32(ctx_, nsCounts_.as<uint32_t>(), nsLayout_.as<uint32_t>() + 1, size_t(L)));
        CS_TRY(nsPrefixes_.ensure(ctx_, size_t(M) * sizeof(K)));
        CS_TRY(nsChild_.ensure(ctx_, size_t(M + 1) * sizeof(NodeIdx)));
        CS_TRY(nsParents_.ensure(ctx_, size_t(std::max(1, (M - 1) / 8)) * sizeof(NodeIdx)));
        CS_TRY(nsLevelRange_.ensure(ctx_, (maxLevel<K>() + 2) * sizeof(NodeIdx)));
        CS_TRY(nsItl_.ensure(ctx_, size_t(M) * sizeof(NodeIdx)));
        CS_TRY(nsLti_.ensure(ctx_, size_t(M) * sizeof(NodeIdx)));
        CS_TRY(cstone_hip_build_octree(ctx_, kb, nsTree_.p, L, nsPrefixes_.p, nsChild_.as<int32_t>(),
                                       nsParents_.as<int32_t>(), nsLevelRange_.as<int32_t>(), nsItl_.as<int32_t>(),
                                       nsLti_.as<int32_t>()));
        CS_TRY(nsCenters_.ensure(ctx_, size_t(M) * 3 * sizeof(T)));
        CS_TRY(nsSizes_.ensure(ctx_, size_t(M) * 3 * sizeof(T)));
        return cstone_hip_node_centers(ctx_, curve_, kb, rb, nsPrefixes_.p, M, &box_, nsCenters_.p, nsSizes_.p);
    }

    cstone_hip_ctx* ctx_;
    int curve_, rank_, P_;
    uint32_t bucket_, bucketFocus_;
    cstone_box box_;
    cstone_hip_comm_ops comm_;
    float haloExt_  = 1.0f;
    float theta_    = 0.5f;
    int sortMode_   = CSTONE_SORT_INCREMENTAL;
    bool speculativeBox_ = std::getenv("CSTONE_NO_SPECULATIVE_BOX") == nullptr;
    bool measureFirst_   = false; // the last box was not the one before it: measure the extents before encoding
    int boxRedos_        = 0;     // syncs whose speculative keys were thrown away because the box had changed
    bool useLet_    = std::getenv("CSTONE_MR_OWNER_SIDE") == nullptr; // halos through the locally essential tree (default)
    std::unique_ptr<FocusLet<K, T>> let_;
    const K* resortTree_ = nullptr; // the leaves of my own key range (and their number) the next sync's re-sort starts from
    int resortLeaves_    = 0;
    bool firstCall_ = true;
    bool ctxGone_   = false; // the destructor runs behind the context's destruction (cstone_hip_domain_mr_destroy)
    bool toggled_   = false; // this sync has switched to the other output buffer set already
    int pending_    = 0; // status of this rank inside sync(): 0, or the error code the peers have to learn about
    uint64_t statusW64_ = 0; // staging of the status words that ride on the collectives (asynchronous copies read them)
    uint32_t statusW32_ = 0;
    std::string pendingMsg_;
    bool timing_    = std::getenv("CSTONE_MR_TIMING") != nullptr;
    bool noMargin_  = std::getenv("CSTONE_MR_NO_MARGIN") != nullptr; // tests: no room left for halos, the block is moved
    bool peerLoop_  = std::getenv("CSTONE_MR_PEER_LOOP") != nullptr; // tests: take the > 32 ranks path (one traversal per peer)
    int syncs_      = 0;
    std::map<std::string, double> phase_;
    std::chrono::steady_clock::time_point t0_;
    std::vector<K> assignment_;
    cstone_hip_domain_mr_view view_{};

    DevBuf scal_;
    DevBuf keys_, order_, keysAlt_, orderAlt_, sortTmp_;
    DevBuf gTree_, gCounts_, gLocalCounts_;
    int gCap_ = 0, gLeaves_ = 0;
    bool gTreeSame_ = false; // the global leaf array is that of the previous sync
    std::vector<K> gLeavesHost_;        // host copies of the global tree and its all-reduced counts (assign())
    std::vector<uint32_t> gCountsHost_;
    bool gLeavesOnHost_  = false;       // gLeavesHost_ is what gTree_ holds (the host made this sync's update step)
    PinnedBlock pin_;                   // where the read-backs of a sync arrive
    uint32_t* pinCounts_ = nullptr;
    K* pinLeaves_        = nullptr;
    bool hostGlobalStep_ = std::getenv("CSTONE_MR_DEVICE_GLOBAL_STEP") == nullptr; // (tests: the device-side step)
    bool speculateCuts_  = std::getenv("CSTONE_MR_NO_SPECULATIVE_CUTS") == nullptr;  // (tests: always ask after assign())
    int cutRedos_        = 0; // syncs whose assignment changed: cut points asked for twice
    float centerDriftTol_      = 1.05f; // Domain::centerDriftTol_ (R/domain/domain.hpp:665)
    bool haveExpansionCenters_ = false; // the last sync was a syncGrav (or updateExpansionCenters followed it)
    bool overlapPlace_   = std::getenv("CSTONE_MR_NO_PLACE_OVERLAP") == nullptr; // x, y, z placed on the second stream
    bool placeForked_    = false; // ... and not joined yet
    DevBuf fTree_, fCounts_, fTmp_;
    int fCap_ = 0, fLeaves_ = 0;
    DevBuf fPrefixes_, fChild_, fParents_, fLevelRange_, fItl_, fLti_;
    DevBuf leaving_, sendRows_, recvRows_, rcol_[4], rcolS_[4], rk_, ro_, posA_, posB_, moveTmp_;
    DevBuf propRecv_[MAX_PROPS], propRecvS_[MAX_PROPS];
    uint64_t prevLo_ = 0, prevHi_ = 0;
    LeafResort<K> resort_;
    int resortBackoff_ = 0, resorts_ = 0;
    uint32_t lastMovers_ = 0;
    uint64_t layoutParticles_ = 0; // particles and box layout_ was made for
    cstone_box layoutBox_{};
    std::vector<K> coverHost_; // leaf keys inserted at the range boundaries, staged for the copy to the device
    int coverDeepest_ = 0;     // level of the deepest of them
    int treeSteps_    = 0;     // single update steps of the focus tree so far (0: the tree was just built from scratch)
    // tree over all local particles incl. halos (octree()), built on request
    DevBuf nsTree_, nsCounts_, nsLayout_, nsPrefixes_, nsChild_, nsParents_, nsLevelRange_, nsItl_, nsLti_, nsCenters_,
        nsSizes_;
    int nsCap_ = 0, nsLeaves_ = 0, nsSync_ = -1;
    // particle routes of the last sync (reapplySync): input size, kept / received / sent counts, start of the kept range
    uint64_t rsN_ = 0, rsNa_ = 0, rsNb_ = 0, rsSend_ = 0, rsMoved_ = 0, rsKeptOffset_ = 0;
    std::vector<uint64_t> rsSendCounts_, rsRecvCounts_;
    int prevMaxLeafLevel_ = -1; // deepest level of this rank's tree at the previous sync
    NodeIdx* hostLevelRange_ = nullptr; // pinned: the level ranges of the last tree, read at the start of the next sync
    bool levelRangePending_  = false;
    // halo exchange pattern of the last sync (exchangeHalos)
    std::vector<uint64_t> haloSend_, haloRecv_;
    uint64_t haloRecvLo_ = 0, haloRecvHi_ = 0, haloAssigned_ = 0, haloSel_ = 0, haloAnyLast_ = 0;
    DevBuf layout_, radii_, boxes_, boxFlags_, myBoxes_, allBoxes_, oflags_, cnt_, sel_;
    Out out_[2];
    int cur_ = 0;
};

} // namespace

} // namespace cship

using namespace cship;

struct cstone_hip_domain_mr
{
    cstone_hip_ctx* ctx;
    std::unique_ptr<MrBase> impl;
};

extern "C"
{

int cstone_hip_domain_mr_create(cstone_hip_ctx* ctx, cstone_hip_domain_mr** out, int curve, int key_bits, int real_bits,
                                int rank, int num_ranks, uint32_t bucket_size, uint32_t bucket_size_focus,
                                const cstone_box* box_host, const cstone_hip_comm_ops* comm)
{
    if (!ctx || !out || !box_host || !comm) return fail(ctx, CSTONE_E_ARG, "domain_mr_create: null argument");
    if (num_ranks < 1 || rank < 0 || rank >= num_ranks) return fail(ctx, CSTONE_E_ARG, "domain_mr_create: bad rank");
    // the small per-rank tables of a sync (cut points, send counts) live in fixed slots of one scalar block
    if (num_ranks > 96) return fail(ctx, CSTONE_E_ARG, "domain_mr_create: at most 96 ranks in this version");
    if (num_ranks > 1 && (!comm->all_reduce || !comm->all_gather || !comm->all_to_all_v))
        return fail(ctx, CSTONE_E_ARG, "domain_mr_create: missing collective");
    if (curve != CSTONE_MORTON && curve != CSTONE_HILBERT) return fail(ctx, CSTONE_E_ARG, "domain_mr_create: bad curve");
    // Domain ctor (R/domain/domain.hpp:95-113)
    if (bucket_size < bucket_size_focus)
        return fail(ctx, CSTONE_E_ARG, "The bucket size of the global tree must not be smaller than the bucket size of "
                                       "the focused tree");
    auto* d = new cstone_hip_domain_mr{ctx, nullptr};
    if (key_bits == 64 && real_bits == 64)
        d->impl = std::make_unique<MultiRankDomain<uint64_t, double>>(ctx, curve, rank, num_ranks, bucket_size,
                                                                      bucket_size_focus, *box_host, *comm);
    else if (key_bits == 64 && real_bits == 32)
        d->impl = std::make_unique<MultiRankDomain<uint64_t, float>>(ctx, curve, rank, num_ranks, bucket_size,
                                                                     bucket_size_focus, *box_host, *comm);
    else if (key_bits == 32 && real_bits == 64)
        d->impl = std::make_unique<MultiRankDomain<uint32_t, double>>(ctx, curve, rank, num_ranks, bucket_size,
                                                                      bucket_size_focus, *box_host, *comm);
    else if (key_bits == 32 && real_bits == 32)
        d->impl = std::make_unique<MultiRankDomain<uint32_t, float>>(ctx, curve, rank, num_ranks, bucket_size,
                                                                     bucket_size_focus, *box_host, *comm);
    else
    {
        delete d;
        return fail(ctx, CSTONE_E_ARG, "domain_mr_create: unsupported type combination");
    }
    *out = d;
    return CSTONE_OK;
}

int cstone_hip_domain_mr_destroy(cstone_hip_domain_mr* dom)
{
    if (!dom) return CSTONE_E_ARG;
    const bool alive = ctxAlive(dom->ctx); // (see cstone_hip_domain_destroy)
    if (alive) (void)hipStreamSynchronize(dom->ctx->stream);
    else (void)hipDeviceSynchronize();
    dom->impl->contextGone(!alive);
    delete dom;
    return alive ? CSTONE_OK : CSTONE_E_ARG;
}

int cstone_hip_domain_mr_sync_props(cstone_hip_domain_mr* dom, const void* x, const void* y, const void* z,
                                    const void* h, size_t n, const void* const* props, const int* prop_bytes,
                                    int num_props)
{
    if (!dom) return CSTONE_E_ARG;
    if (n && (!x || !y || !z || !h)) return fail(dom->ctx, CSTONE_E_ARG, "domain_mr_sync: null array");
    if (num_props && (!props || !prop_bytes)) return fail(dom->ctx, CSTONE_E_ARG, "domain_mr_sync: null property list");
    return dom->impl->sync(x, y, z, h, n, props, prop_bytes, num_props, nullptr);
}

int cstone_hip_domain_mr_sync_keys(cstone_hip_domain_mr* dom, const void* keys, const void* x, const void* y,
                                   const void* z, const void* h, size_t n, const void* const* props,
                                   const int* prop_bytes, int num_props)
{
    if (!dom) return CSTONE_E_ARG;
    if (n && (!x || !y || !z || !h)) return fail(dom->ctx, CSTONE_E_ARG, "domain_mr_sync: null array");
    if (num_props && (!props || !prop_bytes)) return fail(dom->ctx, CSTONE_E_ARG, "domain_mr_sync: null property list");
    return dom->impl->sync(x, y, z, h, n, props, prop_bytes, num_props, keys);
}

int cstone_hip_domain_mr_sync_grav(cstone_hip_domain_mr* dom, const void* keys, const void* x, const void* y,
                                   const void* z, const void* h, const void* m, int mass_bits, size_t n,
                                   const void* const* props, const int* prop_bytes, int num_props)
{
    if (!dom) return CSTONE_E_ARG;
    if (n && (!x || !y || !z || !h || !m)) return fail(dom->ctx, CSTONE_E_ARG, "domain_mr_sync_grav: null array");
    if (num_props && (!props || !prop_bytes)) return fail(dom->ctx, CSTONE_E_ARG, "domain_mr_sync_grav: null property list");
    if (mass_bits != 32 && mass_bits != 64) return fail(dom->ctx, CSTONE_E_ARG, "domain_mr_sync_grav: mass_bits %d", mass_bits);
    return dom->impl->sync(x, y, z, h, n, props, prop_bytes, num_props, keys, m, mass_bits);
}

int cstone_hip_domain_mr_update_expansion_centers(cstone_hip_domain_mr* dom, const void* x, const void* y, const void* z,
                                                  const void* m, int mass_bits)
{
    if (!dom) return CSTONE_E_ARG;
    return dom->impl->updateExpansionCenters(x, y, z, m, mass_bits);
}

int cstone_hip_domain_mr_sync(cstone_hip_domain_mr* dom, const void* x, const void* y, const void* z, const void* h,
                              size_t n)
{
    return cstone_hip_domain_mr_sync_props(dom, x, y, z, h, n, nullptr, nullptr, 0);
}

int cstone_hip_domain_mr_view_get(cstone_hip_domain_mr* dom, cstone_hip_domain_mr_view* out)
{
    if (!dom || !out) return CSTONE_E_ARG;
    return dom->impl->view(out);
}

int cstone_hip_domain_mr_exchange_halos(cstone_hip_domain_mr* dom, void* array, int elem_bytes)
{
    if (!dom || !array) return CSTONE_E_ARG;
    return dom->impl->exchangeHalos(array, elem_bytes);
}

int cstone_hip_domain_mr_octree_get(cstone_hip_domain_mr* dom, cstone_hip_domain_mr_octree* out)
{
    if (!dom || !out) return CSTONE_E_ARG;
    return dom->impl->octree(out);
}

int cstone_hip_domain_mr_reapply_sync(cstone_hip_domain_mr* dom, const void* in, size_t n, int elem_bytes, void* out)
{
    if (!dom) return CSTONE_E_ARG;
    return dom->impl->reapplySync(in, n, elem_bytes, out);
}

int cstone_hip_domain_mr_set_halo_mode(cstone_hip_domain_mr* dom, int mode)
{
    if (!dom) return CSTONE_E_ARG;
    return dom->impl->setHaloMode(mode);
}

int cstone_hip_domain_mr_set_theta(cstone_hip_domain_mr* dom, float theta)
{
    if (!dom) return CSTONE_E_ARG;
    return dom->impl->setTheta(theta);
}

int cstone_hip_domain_mr_set_sort_mode(cstone_hip_domain_mr* dom, int mode)
{
    if (!dom || mode < CSTONE_SORT_INCREMENTAL || mode > CSTONE_SORT_ALL_DIGITS) return CSTONE_E_ARG;
    dom->impl->setSortMode(mode);
    return CSTONE_OK;
}

int cstone_hip_domain_mr_set_halo_factor(cstone_hip_domain_mr* dom, float factor)
{
    if (!dom || !(factor > 0.0f)) return CSTONE_E_ARG;
    dom->impl->setHaloFactor(factor);
    return CSTONE_OK;
}

} // extern "C"
