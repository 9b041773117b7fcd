// Host-side pieces shared by the single-rank and the multi-rank Domain orchestration (domain.hip, domain_mr.hip):
// the update step of the small GLOBAL tree restated for the host, and a pinned block for the read-backs of a sync.
#pragma once

#include <cstdint>
#include <vector>

#include "ctx.hpp"
#include "device_keys.hpp"

namespace cship
{

/*! One update step of the (small, replicated) GLOBAL tree on the host: the decision of nodeOp (tree.hip,
 *  R/tree/csarray.hpp:270-310) and the expansion of rebalanceKernel (R/tree/csarray.hpp:360-385), restated for the host
 *  copies of the leaf array and the all-reduced counts that the last sync read back anyway.  The device then only counts
 *  (and reduces): the read-back between decision and rebalance of cstone_hip_update_octree disappears from a steady-state
 *  sync.  Returns true if every node op is "keep" (the leaf array is unchanged). */
template<class K>
bool globalTreeStepHost(const std::vector<K>& tree, const std::vector<uint32_t>& counts, uint32_t bucket,
                        std::vector<K>& newTree)
{
    const int numNodes = int(counts.size());
    constexpr unsigned top = maxLevel<K>();
    auto span = [](unsigned level) { return K(1) << (3u * (top - level)); };
    auto levelOf = [&](K s) // level of a node of key span s (a power of 8)
    {
        unsigned level = top;
        while (level > 0 && span(level) < s)
            --level;
        return level;
    };
    std::vector<uint32_t> ops(size_t(numNodes) + 1, 0);
    bool keepAll = true;
    for (int i = 0; i < numNodes; ++i)
    {
        const K start        = tree[i];
        const unsigned level = levelOf(K(tree[i + 1] - start));
        uint32_t op          = 1;
        bool merged          = false;
        if (level > 0)
        {
            const int sib = int((start >> (3u * (top - level))) & 7u);
            if (sib > 0)
            {
                const int first = i - sib;
                if (first >= 0 && first + 8 <= numNodes && tree[first + 8] == K(tree[first] + span(level - 1)))
                {
                    uint64_t parent = 0;
                    for (int k = 0; k < 8; ++k)
                        parent += counts[first + k];
                    merged = parent <= uint64_t(bucket);
                }
            }
        }
        if (merged) { op = 0; }
        else
        {
            const uint32_t c = counts[i];
            if (c > bucket * 512u && level + 3 < top) op = 4096;
            else if (c > bucket * 64u && level + 2 < top) op = 512;
            else if (c > bucket * 8u && level + 1 < top) op = 64;
            else if (c > bucket && level < top) op = 8;
        }
        ops[i]  = op;
        keepAll = keepAll && op == 1;
    }
    if (keepAll) return true;
    newTree.clear();
    for (int i = 0; i < numNodes; ++i)
    {
        const uint32_t cnt = ops[i];
        if (cnt == 0) continue;
        const K start        = tree[i];
        const unsigned level = levelOf(K(tree[i + 1] - start));
        unsigned down = 0; // cnt in {1, 8, 64, 512, 4096}: 0..4 levels down
        for (uint32_t c = cnt; c > 1; c /= 8)
            ++down;
        const K step = span(level + down);
        for (uint32_t j = 0; j < cnt; ++j)
            newTree.push_back(K(start + K(j) * step));
    }
    newTree.push_back(tree[numNodes]);
    return false;
}

//! pinned host block for the read-backs of a sync: an asynchronous copy into PAGEABLE memory makes the host wait for it,
//! which would turn every one of the copies that are meant to travel behind one synchronisation into a round trip
struct PinnedBlock
{
    char* p      = nullptr;
    size_t bytes = 0, used = 0;
    ~PinnedBlock()
    {
        if (p) (void)hipHostFree(p);
    }
    //! room for `need` more bytes (64-byte aligned); grows only while nothing is handed out (used == 0)
    void* take(size_t need)
    {
        const size_t at = (used + 63) & ~size_t(63);
        if (at + need > bytes) return nullptr;
        used = at + need;
        return p + at;
    }
    int reserve(cstone_hip_ctx* ctx, size_t total)
    {
        used = 0;
        if (total <= bytes) return CSTONE_OK;
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (p) CS_HIP(ctx, hipHostFree(p));
        p = nullptr, bytes = 0;
        const size_t want = total + total / 2 + 4096;
        CS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&p), want, hipHostMallocDefault));
        bytes = want;
        return CSTONE_OK;
    }
};

} // namespace cship
