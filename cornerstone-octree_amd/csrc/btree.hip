// Binary radix tree over the keys of a cornerstone leaf array (Karras 2012), the reference's legacy btree
// (R/tree/btree.hpp:104-264, GPU wrapper R/tree/btree.cuh:41-52).  One lane per internal node:
// direction and range of the node by galloping + bisection on common key prefixes, split position by bisection.
// Node layout = the reference's BinaryNode<KeyType>: {int32 child[2]; KeyType prefix}; a child that is a leaf (an index
// into the key array) is stored as index - 2^31 (storeLeafIndex, R/tree/btree.hpp:53-58), the prefix carries the
// placeholder bit.
#include "ctx.hpp"
#include "device_keys.hpp"

namespace cship
{
namespace
{

template<class K>
struct BinaryNode
{
    NodeIdx child[2];
    K prefix;
};
static_assert(sizeof(BinaryNode<uint32_t>) == 12 && sizeof(BinaryNode<uint64_t>) == 16);

template<class K>
__global__ __launch_bounds__(256) void binaryTreeKernel(const K* __restrict__ codes, NodeIdx numCodes,
                                                        BinaryNode<K>* __restrict__ nodes)
{
    const NodeIdx first = blockIdx.x * 256 + threadIdx.x;
    if (first >= numCodes - 1) return;
    auto prefixOf = [&](NodeIdx a, NodeIdx b) { return sharedPrefixBits<K>(codes[a], codes[b]); };

    int d = 1, minPrefix = -1;
    if (first > 0)
    {
        d         = prefixOf(first, first + 1) > prefixOf(first, first - 1) ? 1 : -1;
        minPrefix = prefixOf(first, first - d);
    }
    // how far the node reaches: gallop, then bisect (R/tree/btree.hpp:196-219)
    NodeIdx range = 2, second = first + range * d;
    while (0 <= second && second < numCodes && prefixOf(first, second) > minPrefix)
    {
        range *= 2;
        second = first + range * d;
    }
    second = first;
    do
    {
        range        = (range + 1) / 2;
        NodeIdx cand = second + range * d;
        if (0 <= cand && cand < numCodes && prefixOf(first, cand) > minPrefix) second = cand;
    } while (range > 1);

    const int nbits = prefixOf(first, second);
    const K low     = (K(1) << (3 * maxLevel<K>() - nbits)) - 1;
    BinaryNode<K> out;
    out.prefix = toPrefix<K>(codes[first] & ~low, nbits);

    // split: the last key that shares more than nbits with the first key of the range (findSplit, :104-137)
    const NodeIdx lo = min(first, second), hi = max(first, second);
    NodeIdx split;
    if (codes[lo] == codes[hi]) { split = (lo + hi) >> 1; }
    else
    {
        const int common = sharedPrefixBits<K>(codes[lo], codes[hi]);
        split            = lo;
        NodeIdx step     = hi - lo;
        do
        {
            step         = (step + 1) / 2;
            NodeIdx cand = split + step;
            if (cand < hi && sharedPrefixBits<K>(codes[lo], codes[cand]) > common) split = cand;
        } while (step > 1);
    }
    constexpr NodeIdx leafOffset = NodeIdx(-2147483647 - 1);
    out.child[0] = lo == split ? split + leafOffset : split;
    out.child[1] = hi == split + 1 ? split + 1 + leafOffset : split + 1;
    nodes[first] = out;
}

} // namespace
} // namespace cship

using namespace cship;

extern "C" int cstone_hip_create_binary_tree(cstone_hip_ctx* ctx, int key_bits, const void* tree, int num_nodes,
                                             void* binary_nodes)
{
    if (!ctx || (key_bits != 32 && key_bits != 64) || num_nodes < 0 || (num_nodes && (!tree || !binary_nodes)))
        return fail(ctx, CSTONE_E_ARG, "create_binary_tree: bad argument");
    if (num_nodes == 0) return CSTONE_OK;
    unsigned grid = gridFor(size_t(num_nodes), 256);
    if (key_bits == 32)
        hipLaunchKernelGGL(binaryTreeKernel<uint32_t>, grid, 256, 0, ctx->stream, (const uint32_t*)tree, num_nodes + 1,
                           (BinaryNode<uint32_t>*)binary_nodes);
    else
        hipLaunchKernelGGL(binaryTreeKernel<uint64_t>, grid, 256, 0, ctx->stream, (const uint64_t*)tree, num_nodes + 1,
                           (BinaryNode<uint64_t>*)binary_nodes);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}
