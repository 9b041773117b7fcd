// Neighbor search on gfx950.  Replaces findNeighbors (R/findneighbors.hpp:96-188) on the tree view of
// R/tree/octree.hpp:297-317.  Results follow the reference's CPU path: neighbors j != i with
// |r_i - r_j|^2 < (2 h_i)^2, stored in the order of its depth-first traversal (children 0..7, last
// pushed popped first), true count returned even beyond ngmax.
//
// v0 layout: one lane per target particle, depth-first like the CPU path so the stored lists are
// identical element for element; the per-lane traversal stack (160 entries >= 7*21+1, the deepest a
// depth-first octree walk can get) lives in a scratch slice of the context arena and targets are
// processed in chunks of 2^21 so that scratch stays bounded.  Overflow is reported through the sticky
// error word.  (A wave-cooperative variant with an LDS particle queue is the planned v1.)
// Compiled with -ffp-contract=off: distances must round like the CPU path (no FMA).
#include <algorithm>

#include "ctx.hpp"
#include "device_keys.hpp"

namespace cship
{

namespace
{

constexpr int NB_BLOCK = 128;
constexpr int NB_STACK = 160; // >= 7 * 21 + 1

template<class T, bool PBC>
__device__ __forceinline__ T foldAxis(T dx, T len, T inv, bool periodic)
{
    // dX -= pbc * l * rint(dX * il), R/sfc/box.hpp:195-206
    if (PBC && periodic) return dx - len * rint(dx * inv);
    return dx;
}

template<class T>
__global__ __launch_bounds__(NB_BLOCK) void findNeighborsKernel(
    const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, const T* __restrict__ h, uint32_t first,
    uint32_t last, DBox<T> box, const NodeIdx* __restrict__ childOffsets, const NodeIdx* __restrict__ internalToLeaf,
    const uint32_t* __restrict__ layout, const T* __restrict__ centers, const T* __restrict__ sizes, float ext,
    uint32_t ngmax, uint32_t* __restrict__ neighbors, uint32_t* __restrict__ counts, NodeIdx* __restrict__ stackMem,
    int* __restrict__ errors)
{
    uint32_t tid = blockIdx.x * NB_BLOCK + threadIdx.x;
    uint32_t i   = first + tid;
    if (i >= last) return;
    NodeIdx* stack = stackMem + size_t(tid) * NB_STACK;

    const T xi = x[i], yi = y[i], zi = z[i];
    const T hi = h[i];
    const T radSq  = T(4.0) * hi * hi;
    const T cellSq = radSq * ext * ext;
    const bool px = box.bc[0] == 1, py = box.bc[1] == 1, pz = box.bc[2] == 1;
    const T s = T(2) * hi;
    bool inside = (xi - s >= box.lo[0]) && (yi - s >= box.lo[1]) && (zi - s >= box.lo[2]) && (xi + s <= box.hi[0]) &&
                  (yi + s <= box.hi[1]) && (zi + s <= box.hi[2]);
    const bool usePbc = (px || py || pz) && !inside;

    uint32_t* out = neighbors + size_t(tid) * ngmax;
    uint32_t nn   = 0;

    auto overlaps = [&](NodeIdx n) -> bool
    {
        T dx = centers[3 * n] - xi, dy = centers[3 * n + 1] - yi, dz = centers[3 * n + 2] - zi;
        if (usePbc)
        {
            dx = foldAxis<T, true>(dx, box.len[0], box.inv[0], px);
            dy = foldAxis<T, true>(dy, box.len[1], box.inv[1], py);
            dz = foldAxis<T, true>(dz, box.len[2], box.inv[2], pz);
        }
        dx = fabs(dx) - sizes[3 * n], dy = fabs(dy) - sizes[3 * n + 1], dz = fabs(dz) - sizes[3 * n + 2];
        dx += fabs(dx), dy += fabs(dy), dz += fabs(dz);
        dx *= T(0.5), dy *= T(0.5), dz *= T(0.5);
        return dx * dx + (dy * dy + dz * dz) < cellSq; // right fold, R/util/array.hpp:253-256
    };
    auto searchLeaf = [&](NodeIdx n)
    {
        NodeIdx leaf = internalToLeaf[n];
        for (uint32_t j = layout[leaf]; j < layout[leaf + 1]; ++j)
        {
            if (j == i) continue;
            T dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
            if (usePbc)
            {
                dx = foldAxis<T, true>(dx, box.len[0], box.inv[0], px);
                dy = foldAxis<T, true>(dy, box.len[1], box.inv[1], py);
                dz = foldAxis<T, true>(dz, box.len[2], box.inv[2], pz);
            }
            if (dx * dx + dy * dy + dz * dz < radSq)
            {
                if (nn < ngmax) out[nn] = j;
                ++nn;
            }
        }
    };

    // depth-first walk, R/traversal/traversal.hpp:69-110
    if (overlaps(0))
    {
        if (childOffsets[0] == 0) { searchLeaf(0); }
        else
        {
            int top    = 1;
            stack[0]   = 0;
            NodeIdx node = 0;
            do
            {
                NodeIdx c0 = childOffsets[node];
                for (int oct = 0; oct < 8; ++oct)
                {
                    NodeIdx child = c0 + oct;
                    if (!overlaps(child)) continue;
                    if (childOffsets[child] == 0) { searchLeaf(child); }
                    else if (top < NB_STACK) { stack[top++] = child; }
                    else { atomicOr(errors, 4); }
                }
                node = stack[--top];
            } while (node != 0);
        }
    }
    counts[tid] = nn;
}

} // namespace

} // namespace cship

using namespace cship;

extern "C" int cstone_hip_find_neighbors(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y,
                                         const void* z, const void* h, uint32_t first, uint32_t last,
                                         const cstone_box* box_host, const int32_t* child_offsets,
                                         const int32_t* internal_to_leaf, const uint32_t* layout, const void* centers,
                                         const void* sizes, float ext, uint32_t ngmax, uint32_t* neighbors,
                                         uint32_t* counts)
{
    if (!ctx || !x || !y || !z || !h || !box_host || !child_offsets || !internal_to_leaf || !layout || !centers ||
        !sizes || !counts || (ngmax && !neighbors) || last < first)
        return fail(ctx, CSTONE_E_ARG, "find_neighbors: bad argument");
    if (last == first) return CSTONE_OK;
    if (real_bits != 32 && real_bits != 64) return fail(ctx, CSTONE_E_ARG, "find_neighbors: real_bits %d unsupported", real_bits);
    size_t nw    = last - first;
    size_t chunk = std::min<size_t>(nw, size_t(1) << 21);
    CS_TRY(arenaReserve(ctx, chunk * NB_STACK * sizeof(NodeIdx) + 1024));
    auto* stackMem = (NodeIdx*)arenaTake(ctx, chunk * NB_STACK * sizeof(NodeIdx));
    int* errors    = ctx->devScalars + 63;
    {
        StageTimer timer(ctx, CSTONE_STAGE_NEIGHBORS);
        for (size_t off = 0; off < nw; off += chunk)
        {
            uint32_t f = first + uint32_t(off);
            uint32_t l = uint32_t(std::min<size_t>(size_t(last), size_t(f) + chunk));
            unsigned grid = gridFor(l - f, NB_BLOCK);
            uint32_t* nbOut = neighbors ? neighbors + off * ngmax : nullptr;
            if (real_bits == 32)
                hipLaunchKernelGGL(findNeighborsKernel<float>, grid, NB_BLOCK, 0, ctx->stream, (const float*)x,
                                   (const float*)y, (const float*)z, (const float*)h, f, l,
                                   makeDBox<float>(*box_host), child_offsets, internal_to_leaf, layout,
                                   (const float*)centers, (const float*)sizes, ext, ngmax, nbOut, counts + off,
                                   stackMem, errors);
            else
                hipLaunchKernelGGL(findNeighborsKernel<double>, grid, NB_BLOCK, 0, ctx->stream, (const double*)x,
                                   (const double*)y, (const double*)z, (const double*)h, f, l,
                                   makeDBox<double>(*box_host), child_offsets, internal_to_leaf, layout,
                                   (const double*)centers, (const double*)sizes, ext, ngmax, nbOut, counts + off,
                                   stackMem, errors);
        }
    }
    arenaReset(ctx);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}
