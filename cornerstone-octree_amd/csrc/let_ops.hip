// Device work behind the host state machine of the locally essential tree (csrc/let.hpp): the small key-array kernels of
// the treelet exchange (R/focus/exchange_focus.hpp:98-287) and of the halo layout (R/domain/layout.hpp:91-190,
// R/domain/exchange_keys.hpp:63-119).  All of them are node-array work (10^4 .. 10^6 keys per rank): latency and launch
// bound, one lane per key; nothing here touches the particle arrays.
#include <algorithm>
#include <vector>

#include "ctx.hpp"
#include "device_keys.hpp"
#include "scan.hpp"

namespace cship
{
namespace
{

//! index of the first of keys[0 .. n) that is >= key
template<class K>
__device__ __forceinline__ NodeIdx lowerBoundDev(const K* __restrict__ keys, NodeIdx n, K key)
{
    NodeIdx lo = 0, len = n;
    while (len > 0)
    {
        NodeIdx half = len >> 1;
        bool right   = keys[lo + half] < key;
        lo           = right ? lo + half + 1 : lo;
        len          = right ? len - half - 1 : half;
    }
    return lo;
}

//! checkTreelets, R/focus/exchange_focus.hpp:104-115
template<class K>
__global__ __launch_bounds__(256) void keysMissingKernel(const K* __restrict__ leaves, NodeIdx numLeaves,
                                                         const K* __restrict__ keys, size_t n,
                                                         uint32_t* __restrict__ flags)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const K k = keys[i];
    flags[i]  = k != leaves[lowerBoundDev(leaves, numLeaves, k)] ? 1u : 0u; // leaves has numLeaves + 1 keys
}

template<class K>
__global__ __launch_bounds__(256) void partitionKeysKernel(const K* __restrict__ keys, const uint32_t* __restrict__ flags,
                                                           const uint32_t* __restrict__ scan, size_t n,
                                                           K* __restrict__ setOut, K* __restrict__ unsetOut)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const K k = keys[i];
    if (flags[i])
    {
        if (setOut) setOut[scan[i]] = k;
    }
    else if (unsetOut) { unsetOut[i - scan[i]] = k; }
}

//! exchangeRejectedKeys, R/focus/exchange_focus.hpp:186-190 (idempotent stores: the order of the keys is free)
template<class K>
__global__ __launch_bounds__(256) void zeroOpsKernel(const K* __restrict__ leaves, NodeIdx numKeysInLeaves,
                                                     const K* __restrict__ keys, size_t n, int32_t* __restrict__ ops)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const NodeIdx at = lowerBoundDev(leaves, numKeysInLeaves, keys[i]);
    if (at < numKeysInLeaves) ops[at] = 0;
}

//! locateNode(startKey, endKey, prefixes, levelRange), R/tree/octree.hpp:216-241
template<class K>
__global__ __launch_bounds__(256) void locateNodesKernel(const K* __restrict__ keys, size_t numNodes,
                                                         const K* __restrict__ prefixes,
                                                         const NodeIdx* __restrict__ levelRange,
                                                         int32_t* __restrict__ idx)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= numNodes) return;
    const K start = keys[i], end = keys[i + 1];
    const NodeIdx total = levelRange[maxLevel<K>() + 1];
    NodeIdx found       = total;
    if (end > start)
    {
        // prefix length = countLeadingZeros(end - start - 1) - unusedBits (:239)
        const int bits       = clzKey(K(end - start - 1)) - int(KeyInfo<K>::spare);
        const unsigned level = unsigned(bits) / 3u;
        if (bits >= 0 && level <= maxLevel<K>())
        {
            const K want    = toPrefix<K>(start, bits);
            const NodeIdx a = levelRange[level], b = levelRange[level + 1];
            const NodeIdx at = a + lowerBoundDev(prefixes + a, b - a, want);
            if (at != total && prefixes[at] == want) found = at;
        }
    }
    idx[i] = found;
}

//! computeNodeLayout before its scan, R/domain/layout.hpp:156-162
__global__ __launch_bounds__(256) void presentCountsKernel(const uint32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ flags, NodeIdx first, NodeIdx last,
                                                           NodeIdx numLeaves, uint32_t* __restrict__ out)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i > numLeaves) return;
    uint32_t c = 0;
    if (i < numLeaves && ((first <= i && i < last) || flags[i] != 0)) c = counts[i];
    out[i] = c;
}

/*! run boundaries of the flagged leaves inside the peers' leaf ranges (extractMarkedElements, R/domain/layout.hpp:104-139):
 *  starts[i] = 1 if a run starts at leaf i, ends[i] = 1 if one ends there; unmatched counts flagged leaves outside the
 *  own range [first, last) that lie in no peer's range (checkHalos, R/halos/halos.hpp:59-95).
 *  ranges: numRanks (first, last) pairs, ascending, (0, 0) for ranks that are no peers */
__global__ __launch_bounds__(256) void haloRunsKernel(const int32_t* __restrict__ flags, NodeIdx numLeaves, NodeIdx first,
                                                      NodeIdx last, const int32_t* __restrict__ ranges, int numRanks,
                                                      uint32_t* __restrict__ starts, uint32_t* __restrict__ ends,
                                                      uint32_t* __restrict__ unmatched)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i > numLeaves) return;
    uint32_t s = 0, e = 0;
    if (i < numLeaves && flags[i] == 1)
    {
        // the peer range that holds leaf i (a handful of ranges: linear walk)
        int owner = -1;
        for (int r = 0; r < numRanks; ++r)
            if (ranges[2 * r] <= i && i < ranges[2 * r + 1]) owner = r;
        if (owner >= 0)
        {
            const NodeIdx a = ranges[2 * owner], b = ranges[2 * owner + 1];
            s = (i == a || flags[i - 1] != 1) ? 1u : 0u;
            e = (i == b - 1 || flags[i + 1] != 1) ? 1u : 0u;
        }
        else if (i < first || i >= last) { atomicAdd(unmatched, 1u); }
    }
    starts[i] = s;
    ends[i]   = e;
}

template<class K>
__global__ __launch_bounds__(256) void haloPairsKernel(const K* __restrict__ leaves, NodeIdx numLeaves,
                                                       const uint32_t* __restrict__ starts,
                                                       const uint32_t* __restrict__ ends,
                                                       const uint32_t* __restrict__ startScan,
                                                       const uint32_t* __restrict__ endScan, K* __restrict__ pairs)
{
    NodeIdx i = blockIdx.x * 256 + threadIdx.x;
    if (i >= numLeaves) return;
    if (starts[i]) pairs[2 * size_t(startScan[i])] = leaves[i];
    if (ends[i]) pairs[2 * size_t(endScan[i]) + 1] = leaves[i + 1];
}

//! the serving side of exchangeRequestKeys, R/domain/exchange_keys.hpp:98-108
template<class K>
__global__ __launch_bounds__(256) void rangesFromKeysKernel(const K* __restrict__ leaves, NodeIdx numKeysInLeaves,
                                                            const uint32_t* __restrict__ layout,
                                                            const K* __restrict__ pairs, size_t numPairs,
                                                            uint32_t* __restrict__ offsets, uint32_t* __restrict__ lengths)
{
    size_t r = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (r > numPairs) return;
    if (r == numPairs)
    {
        lengths[r] = 0;
        return;
    }
    NodeIdx a = lowerBoundDev(leaves, numKeysInLeaves, pairs[2 * r]);
    NodeIdx b = lowerBoundDev(leaves, numKeysInLeaves, pairs[2 * r + 1]);
    a = min(a, numKeysInLeaves - 1), b = min(b, numKeysInLeaves - 1);
    const uint32_t lo = layout[a], hi = layout[b];
    offsets[r] = lo;
    lengths[r] = hi > lo ? hi - lo : 0u;
}

bool badKeyBits(int kb) { return kb != 32 && kb != 64; }

//! exclusive scan of n values in the arena
int scanInto(cstone_hip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n)
{
    CS_TRY(arenaReserve(ctx, scanArenaBytes(n)));
    int rc = scanU32(ctx, in, out, n, 0u, false, nullptr);
    arenaReset(ctx);
    return rc;
}

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

int cstone_hip_raise(cstone_hip_ctx* ctx, int code, const char* message)
{
    return fail(ctx, code, "%s", message ? message : "");
}

int cstone_hip_keys_missing(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves, const void* keys,
                            size_t num_keys, uint32_t* flags)
{
    if (!ctx || badKeyBits(key_bits) || num_leaves < 1 || !leaves || (num_keys && (!keys || !flags)))
        return fail(ctx, CSTONE_E_ARG, "keys_missing: bad argument");
    if (num_keys == 0) return CSTONE_OK;
    CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(keysMissingKernel<K>, gridFor(num_keys, 256), 256, 0, ctx->stream,
                                                   (const K*)leaves, num_leaves, (const K*)keys, num_keys, flags));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_partition_keys(cstone_hip_ctx* ctx, int key_bits, const void* keys, const uint32_t* flags,
                              const uint32_t* scan, size_t num_keys, void* set_out, void* unset_out)
{
    if (!ctx || badKeyBits(key_bits) || (num_keys && (!keys || !flags || !scan)))
        return fail(ctx, CSTONE_E_ARG, "partition_keys: bad argument");
    if (num_keys == 0) return CSTONE_OK;
    CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(partitionKeysKernel<K>, gridFor(num_keys, 256), 256, 0, ctx->stream,
                                                   (const K*)keys, flags, scan, num_keys, (K*)set_out, (K*)unset_out));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_zero_ops_at_keys(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves, const void* keys,
                                size_t num_keys, int32_t* node_ops)
{
    if (!ctx || badKeyBits(key_bits) || num_leaves < 1 || !leaves || !node_ops || (num_keys && !keys))
        return fail(ctx, CSTONE_E_ARG, "zero_ops_at_keys: bad argument");
    if (num_keys == 0) return CSTONE_OK;
    CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(zeroOpsKernel<K>, gridFor(num_keys, 256), 256, 0, ctx->stream,
                                                   (const K*)leaves, num_leaves + 1, (const K*)keys, num_keys, node_ops));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_locate_nodes(cstone_hip_ctx* ctx, int key_bits, const void* keys, size_t num_keys, const void* prefixes,
                            const int32_t* level_range, int32_t* idx)
{
    if (!ctx || badKeyBits(key_bits) || (num_keys > 1 && (!keys || !prefixes || !level_range || !idx)))
        return fail(ctx, CSTONE_E_ARG, "locate_nodes: bad argument");
    if (num_keys < 2) return CSTONE_OK;
    CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(locateNodesKernel<K>, gridFor(num_keys - 1, 256), 256, 0, ctx->stream,
                                                   (const K*)keys, num_keys - 1, (const K*)prefixes, level_range, idx));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_node_layout(cstone_hip_ctx* ctx, const uint32_t* counts, const int32_t* flags, int first, int last,
                           int num_leaves, uint32_t* layout)
{
    if (!ctx || num_leaves < 0 || !layout || (num_leaves && (!counts || !flags)))
        return fail(ctx, CSTONE_E_ARG, "node_layout: bad argument");
    const size_t n = size_t(num_leaves) + 1;
    hipLaunchKernelGGL(presentCountsKernel, gridFor(n, 256), 256, 0, ctx->stream, counts, flags, first, last, num_leaves,
                       layout);
    CS_TRY(scanInto(ctx, layout, layout, n));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // extern "C"

namespace
{
//! row[p] = 2 * (pairs requested from rank p) as u64, row[numRanks] = status: 2 if the caller has a failure of its own
//! pending, else 1 if halo cells belong to nobody, else 0 -- the row a rank contributes to the all-gather of
//! Halos::computeLayout (let.hpp), written where the collective reads it: no trip to the host in between
__global__ void requestRowKernel(const uint32_t* __restrict__ at, const uint32_t* __restrict__ unmatched, int numRanks,
                                 int externalFailure, uint64_t* __restrict__ row)
{
    const int r = threadIdx.x + blockIdx.x * 64;
    if (r < numRanks) row[r] = 2ull * uint64_t(at[2 * r + 1] - at[2 * r]);
    else if (r == numRanks) row[r] = externalFailure ? 2ull : (*unmatched ? 1ull : 0ull);
}

//! count[p] = number of my leaves over rank p's key range + 1 (its upper boundary key) if p is a peer, else 0: the
//! treelet sizes of syncTreelets (R/focus/exchange_focus.hpp:61-96) from the lower bounds of the assignment keys
//! (bounds[r]: first leaf >= assignment[r]; bounds[P + 1 + r]: first leaf >= assignment[r] + 1) -- translateAssignment
//! (R/domain/domaindecomp.hpp:183-206) evaluated where the numbers are
__global__ void peerRangeCountsKernel(const uint64_t* __restrict__ bounds, const uint8_t* __restrict__ isPeer, int numRanks,
                                      uint64_t* __restrict__ row)
{
    const int p = threadIdx.x + blockIdx.x * 64;
    if (p >= numRanks) return;
    uint64_t c = 0;
    if (isPeer[p])
    {
        const int64_t s = int64_t(bounds[p]);                          // findNodeAbove(assignment[p])
        int64_t e       = int64_t(bounds[numRanks + 1 + p + 1]) - 1;   // findNodeBelow(assignment[p + 1])
        if (e < s) e = s;
        c = uint64_t(e - s) + 1;
    }
    row[p] = c;
}
} // namespace

extern "C"
{

static int haloRequestsImpl(cstone_hip_ctx* ctx, int key_bits, const void* leaves, const int32_t* flags, int num_leaves,
                            int first, int last, const int32_t* ranges_host, int num_ranks, void* pairs_out,
                            uint32_t* pair_counts_host, uint32_t* unmatched_host, uint64_t* row_dev, int external_failure);

int cstone_hip_halo_requests(cstone_hip_ctx* ctx, int key_bits, const void* leaves, const int32_t* flags, int num_leaves,
                             int first, int last, const int32_t* ranges_host, int num_ranks, void* pairs_out,
                             uint32_t* pair_counts_host, uint32_t* unmatched_host)
{
    if (!pair_counts_host || !unmatched_host) return fail(ctx, CSTONE_E_ARG, "halo_requests: bad argument");
    return haloRequestsImpl(ctx, key_bits, leaves, flags, num_leaves, first, last, ranges_host, num_ranks, pairs_out,
                            pair_counts_host, unmatched_host, nullptr, 0);
}

int cstone_hip_halo_request_rows(cstone_hip_ctx* ctx, int key_bits, const void* leaves, const int32_t* flags,
                                 int num_leaves, int first, int last, const int32_t* ranges_host, int num_ranks,
                                 void* pairs_out, uint64_t* row_dev, int external_failure)
{
    if (!row_dev) return fail(ctx, CSTONE_E_ARG, "halo_request_rows: bad argument");
    return haloRequestsImpl(ctx, key_bits, leaves, flags, num_leaves, first, last, ranges_host, num_ranks, pairs_out, nullptr,
                            nullptr, row_dev, external_failure);
}

int cstone_hip_peer_range_counts(cstone_hip_ctx* ctx, const uint64_t* bounds_dev, const uint8_t* is_peer_host,
                                 int num_ranks, uint64_t* row_dev)
{
    if (!ctx || !bounds_dev || !is_peer_host || num_ranks < 1 || !row_dev)
        return fail(ctx, CSTONE_E_ARG, "peer_range_counts: bad argument");
    CS_TRY(arenaReserve(ctx, alignUp(size_t(num_ranks)) + 256));
    auto* dPeer = (uint8_t*)arenaTake(ctx, size_t(num_ranks));
    int rc      = cstone_hip_upload(ctx, dPeer, is_peer_host, size_t(num_ranks));
    if (rc == CSTONE_OK)
    {
        hipLaunchKernelGGL(peerRangeCountsKernel, gridFor(size_t(num_ranks), 64), 64, 0, ctx->stream, bounds_dev, dPeer,
                           num_ranks, row_dev);
        if (hipGetLastError() != hipSuccess) rc = fail(ctx, CSTONE_E_HIP, "peer_range_counts: launch failed");
    }
    arenaReset(ctx);
    return rc;
}

static int haloRequestsImpl(cstone_hip_ctx* ctx, int key_bits, const void* leaves, const int32_t* flags, int num_leaves,
                            int first, int last, const int32_t* ranges_host, int num_ranks, void* pairs_out,
                            uint32_t* pair_counts_host, uint32_t* unmatched_host, uint64_t* row_dev, int external_failure)
{
    if (!ctx || badKeyBits(key_bits) || num_leaves < 1 || num_ranks < 1 || !leaves || !flags || !ranges_host || !pairs_out)
        return fail(ctx, CSTONE_E_ARG, "halo_requests: bad argument");
    const size_t n = size_t(num_leaves) + 1;
    // arena: ranges, the run flags and their scans, the scans at the range boundaries, the counter
    const size_t rangeBytes = alignUp(size_t(num_ranks) * 2 * sizeof(int32_t));
    const size_t atBytes    = alignUp(size_t(num_ranks) * 2 * sizeof(uint32_t));
    CS_TRY(arenaReserve(ctx, rangeBytes + 4 * alignUp(n * 4) + 2 * atBytes + 256 + 2 * scanArenaBytes(n) + 4096));
    auto* dRanges   = (int32_t*)arenaTake(ctx, rangeBytes);
    auto* starts    = (uint32_t*)arenaTake(ctx, n * 4);
    auto* ends      = (uint32_t*)arenaTake(ctx, n * 4);
    auto* startScan = (uint32_t*)arenaTake(ctx, n * 4);
    auto* endScan   = (uint32_t*)arenaTake(ctx, n * 4);
    auto* dMap      = (uint32_t*)arenaTake(ctx, atBytes);
    auto* dAt       = (uint32_t*)arenaTake(ctx, atBytes);
    auto* counter   = (uint32_t*)arenaTake(ctx, 256);
    int rc          = CSTONE_OK;
    std::vector<uint32_t> at(size_t(num_ranks) * 2, 0), map(size_t(num_ranks) * 2, 0);
    for (int r = 0; r < num_ranks; ++r)
    {
        map[2 * r]     = uint32_t(std::clamp(ranges_host[2 * r], 0, num_leaves));
        map[2 * r + 1] = uint32_t(std::clamp(ranges_host[2 * r + 1], 0, num_leaves));
    }
    auto body = [&]() -> int
    {
        // (through the pinned ring: these vectors are locals, and the row variant returns before the copies ran)
        CS_TRY(cstone_hip_upload(ctx, dRanges, ranges_host, size_t(num_ranks) * 2 * sizeof(int32_t)));
        CS_TRY(cstone_hip_upload(ctx, dMap, map.data(), map.size() * 4));
        CS_HIP(ctx, hipMemsetAsync(counter, 0, 4, ctx->stream));
        hipLaunchKernelGGL(haloRunsKernel, gridFor(n, 256), 256, 0, ctx->stream, flags, num_leaves, first, last, dRanges,
                           num_ranks, starts, ends, counter);
        CS_TRY(scanU32(ctx, starts, startScan, n, 0u, false, nullptr));
        CS_TRY(scanU32(ctx, ends, endScan, n, 0u, false, nullptr));
        CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(haloPairsKernel<K>, gridFor(size_t(num_leaves), 256), 256, 0,
                                                       ctx->stream, (const K*)leaves, num_leaves, starts, ends, startScan,
                                                       endScan, (K*)pairs_out));
        CS_TRY(cstone_hip_gather(ctx, 4, dMap, map.size(), startScan, dAt));
        if (row_dev)
        {
            // the counts stay on the device: the row of the all-gather is written where the collective reads it
            hipLaunchKernelGGL(requestRowKernel, gridFor(size_t(num_ranks) + 1, 64), 64, 0, ctx->stream, dAt, counter,
                               num_ranks, external_failure, row_dev);
            CS_HIP(ctx, hipGetLastError());
            return CSTONE_OK;
        }
        CS_TRY(copyToPinned(ctx, ctx->hostScalars + 9, counter, 4));
        CS_TRY(copyToHost(ctx, at.data(), dAt, at.size() * 4)); // (synchronises the stream)
        *unmatched_host = uint32_t(ctx->hostScalars[9]);
        return CSTONE_OK;
    };
    rc = body();
    arenaReset(ctx);
    CS_TRY(rc);
    if (!row_dev)
        for (int r = 0; r < num_ranks; ++r)
            pair_counts_host[r] = at[2 * r + 1] - at[2 * r];
    return CSTONE_OK;
}

int cstone_hip_ranges_from_keys(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int num_leaves,
                                const uint32_t* layout, const void* pairs, size_t num_pairs, uint32_t* range_offsets,
                                uint32_t* range_scan)
{
    if (!ctx || badKeyBits(key_bits) || num_leaves < 1 || !leaves || !layout || !range_scan ||
        (num_pairs && (!pairs || !range_offsets)))
        return fail(ctx, CSTONE_E_ARG, "ranges_from_keys: bad argument");
    CSTONE_KEY_SWITCH(key_bits, hipLaunchKernelGGL(rangesFromKeysKernel<K>, gridFor(num_pairs + 1, 256), 256, 0,
                                                   ctx->stream, (const K*)leaves, num_leaves + 1, layout, (const K*)pairs,
                                                   num_pairs, range_offsets, range_scan));
    CS_TRY(scanInto(ctx, range_scan, range_scan, num_pairs + 1));
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // extern "C"
