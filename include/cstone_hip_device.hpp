/*! cstone_hip_device.hpp -- the neighbour traversal of libcstone_hip for CLIENT kernels (hipcc, gfx950, wave64).
 *
 * Replaces, for kernels written against it, the device-side interface of the reference:
 *   traverseNeighbors + TravConfig + loadTarget   R/traversal/find_neighbors.cuh:46-230,436-506
 *   OctreeNsView                                  R/tree/octree.hpp:297-317
 *   findNeighbors (the per-particle walk whose results this one reproduces)   R/findneighbors.hpp:96-188
 * (R = include/cstone of the reference).  SPH-EXA-style kernels do not want neighbour LISTS: they want to run their
 * pair interaction while the tree is walked.  cstone_hip::traverseNeighbors does that walk for the 64 targets of a wave and
 * calls the client's functor once per (target, neighbour) pair -- in exactly the order of the reference's depth-first walk
 * (children 0..7, the particles of a leaf in storage order), so a sum accumulated in the functor is bit-identical to the
 * same sum taken over the list that cstone_hip_find_neighbors returns.
 *
 * MI355X shape: ONE traversal per wave of 64 targets that are consecutive in SFC order (their search spheres cover almost
 * the same nodes).  Every stack entry carries the 64-bit mask of the lanes whose own walk would have reached the node;
 * node geometry is wave-uniform (scalar loads), the particles of a leaf are fetched by the wave with one coalesced load
 * per coordinate and handed round with v_readlane, the test loop touches no memory.  The stack (TraversalStack, 1 920
 * bytes) lives in LDS and is supplied by the caller: `__shared__ cstone_hip::TraversalStack stacks[WAVES_PER_BLOCK];`.
 *
 * Compile client kernels with -ffp-contract=off if their results are to be compared bit for bit with a list-based
 * evaluation (the distance test itself is written without contractions either way).
 *
 * Usage (see examples/sph_density.hip):
 *
 *   __global__ void density(cstone_hip::OctreeNsView<double> tree, cstone_hip::DeviceBox<double> box, const double* x, ...)
 *   {
 *       __shared__ cstone_hip::TraversalStack stacks[4];                    // 256 threads = 4 waves
 *       const uint32_t i   = first + blockIdx.x * 256 + threadIdx.x;
 *       const bool valid   = i < last;
 *       double rho         = 0;
 *       uint32_t nc = cstone_hip::traverseNeighbors(valid, valid ? i : last - 1, x, y, z, h, tree, box, 1.0f,
 *                                                   stacks[threadIdx.x / 64], nullptr,
 *                                                   [&](uint32_t j, double dx, double dy, double dz, double d2)
 *                                                   { rho += m[j] * W(sqrt(d2), h[i]); });
 *       if (valid) out[i] = rho;
 *   }
 */
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "cstone_hip.h"

namespace cstone_hip
{

//! device image of cstone::Box<T> (R/sfc/box.hpp:112-191), passed to kernels by value
template<class T>
struct DeviceBox
{
    T lo[3], hi[3], len[3], inv[3];
    int bc[3]; // cstone_box boundary types: 0 open, 1 periodic, 2 fixed
};

//! host: the device box exactly as the Box<T> constructor builds it (lengths and 1 / length in T)
template<class T>
inline DeviceBox<T> makeDeviceBox(const cstone_box& b)
{
    DeviceBox<T> d;
    for (int a = 0; a < 3; ++a)
    {
        d.lo[a]  = T(b.lim[2 * a]);
        d.hi[a]  = T(b.lim[2 * a + 1]);
        d.len[a] = d.hi[a] - d.lo[a];
        d.inv[a] = T(1.) / (d.hi[a] - d.lo[a]); // R/sfc/box.hpp:135
        d.bc[a]  = b.bc[a];
    }
    return d;
}

/*! what the traversal reads of the linked octree: OctreeNsView of the reference (R/tree/octree.hpp:297-317).  The pointers
 *  are the fields of the same names of cstone_hip_domain_view / cstone_hip_domain_mr_octree (device memory). */
template<class T>
struct OctreeNsView
{
    const int32_t* childOffsets;   // [numNodes + 1]: first child of a node, 0 for a leaf
    const int32_t* internalToLeaf; // [numNodes]: index of a leaf node in the cornerstone leaf array
    const uint32_t* layout;        // [numLeaves + 1]: first particle of every leaf
    const T* centers;              // [numNodes][3] geometric centres
    const T* sizes;                // [numNodes][3] half edge lengths
};

//! TravConfig of this traversal (the reference: R/traversal/find_neighbors.cuh:46-68)
struct TravConfig
{
    static constexpr int targetSize = 64;  // targets per traversal = lanes of a wave
    static constexpr int stackSize  = 160; // >= 7 * 21 + 1: the deepest a depth-first walk of an octree of 21 levels gets
};

//! the LDS stack of ONE wave; declare `__shared__ TraversalStack stacks[wavesPerBlock]` and pass stacks[wave]
struct TraversalStack
{
    int32_t node[TravConfig::stackSize];
    uint64_t mask[TravConfig::stackSize];
};

namespace detail
{
__device__ __forceinline__ int32_t uniform(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

//! value of lane k (wave-uniform k) as a wave-uniform scalar
__device__ __forceinline__ float readLane(float v, unsigned k)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), int(k)));
}
__device__ __forceinline__ double readLane(double v, unsigned k)
{
    long long b = __double_as_longlong(v);
    int lo      = __builtin_amdgcn_readlane(int(b), int(k));
    int hi      = __builtin_amdgcn_readlane(int(b >> 32), int(k));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

//! dX -= l * rint(dX * il) on periodic axes (R/sfc/box.hpp:195-206)
template<class T>
__device__ __forceinline__ T foldAxis(T dx, T len, T inv, bool periodic)
{
    if (periodic) return dx - len * rint(dx * inv);
    return dx;
}

struct NoStats
{
    __device__ void leaf(uint32_t, bool) {}
    __device__ void depth(int) {}
};
} // namespace detail

/*! The neighbours of the wave's 64 targets, one traversal for all of them.  COLLECTIVE: every lane of the wave must call it
 *  (lanes without a target pass valid = false and any in-range index i).
 *
 *  valid, i        this lane's target particle (index into x, y, z, h)
 *  x, y, z, h      particle arrays of the domain (device), SFC ordered; the search radius of target i is 2 h[i]
 *  tree, box       OctreeNsView of the domain's octree, the domain's box
 *  ext             Domain::setHaloFactor's search extension for the NODE test (1.0f: none), as cstone_hip_find_neighbors
 *  stack           this wave's LDS stack
 *  errors          device int that gets |= 4 when the stack overflows (cstone's sticky error word), or nullptr
 *  onNeighbor      called as onNeighbor(j, dx, dy, dz, d2) for every particle j != i with d2 = |r_j - r_i|^2 < (2 h_i)^2
 *                  (dx = x_j - x_i after the periodic fold ...), by the lane of target i, in the order of the reference's
 *                  walk
 *  stats           optional observer (detail::NoStats: none)
 *  returns         the number of neighbours of this lane's target */
template<class T, class F, class Stats = detail::NoStats>
__device__ __forceinline__ uint32_t traverseNeighbors(bool valid, uint32_t i, const T* __restrict__ x,
                                                      const T* __restrict__ y, const T* __restrict__ z,
                                                      const T* __restrict__ h, const OctreeNsView<T>& tree,
                                                      const DeviceBox<T>& box, float ext, TraversalStack& stack,
                                                      int* errors, F&& onNeighbor, Stats&& stats = Stats{})
{
    using detail::uniform;
    const unsigned lane = threadIdx.x & 63u;
    int32_t* sNode      = stack.node;
    uint64_t* sMask     = stack.mask;

    const T xi = x[i], yi = y[i], zi = z[i];
    const T hi = h[i];
    const T radSq  = T(4.0) * hi * hi;
    const T cellSq = radSq * ext * ext;
    const bool px = box.bc[0] == 1, py = box.bc[1] == 1, pz = box.bc[2] == 1;
    const T s = T(2) * hi;
    const bool inside = (xi - s >= box.lo[0]) && (yi - s >= box.lo[1]) && (zi - s >= box.lo[2]) && (xi + s <= box.hi[0]) &&
                        (yi + s <= box.hi[1]) && (zi + s <= box.hi[2]);
    const bool usePbc = (px || py || pz) && !inside;
    uint32_t nn       = 0;

    // n is wave-uniform: centres and sizes come through the scalar cache
    auto overlaps = [&](int32_t n) -> bool
    {
        T dx = tree.centers[3 * n] - xi, dy = tree.centers[3 * n + 1] - yi, dz = tree.centers[3 * n + 2] - zi;
        if (usePbc)
        {
            dx = detail::foldAxis<T>(dx, box.len[0], box.inv[0], px);
            dy = detail::foldAxis<T>(dy, box.len[1], box.inv[1], py);
            dz = detail::foldAxis<T>(dz, box.len[2], box.inv[2], pz);
        }
        dx = fabs(dx) - tree.sizes[3 * n], dy = fabs(dy) - tree.sizes[3 * n + 1], dz = fabs(dz) - tree.sizes[3 * n + 2];
        dx += fabs(dx), dy += fabs(dy), dz += fabs(dz);
        dx *= T(0.5), dy *= T(0.5), dz *= T(0.5);
        return dx * dx + (dy * dy + dz * dz) < cellSq; // right fold, R/util/array.hpp:253-256
    };
    // all particles of leaf node n against the lanes that reached it.  The wave fetches up to 64 leaf particles with one
    // coalesced load per coordinate (lane l holds particle base + l) and hands them round by v_readlane: no memory
    // traffic inside the test loop.  Two copies of the loop: the periodic fold is only compiled into the one taken when
    // some lane needs it.
    auto searchLeaf = [&](int32_t n, bool mine)
    {
        const int32_t leaf = uniform(tree.internalToLeaf[n]);
        const uint32_t jb  = uint32_t(uniform(int32_t(tree.layout[leaf])));
        const uint32_t je  = uint32_t(uniform(int32_t(tree.layout[leaf + 1])));
        const bool fold    = __any(mine && usePbc);
        stats.leaf(je - jb, mine);
        for (uint32_t base = jb; base < je; base += 64)
        {
            const uint32_t cnt = min(64u, je - base);
            T xl = T(0), yl = T(0), zl = T(0);
            if (lane < cnt) xl = x[base + lane], yl = y[base + lane], zl = z[base + lane];
            auto test = [&](uint32_t k, bool withFold)
            {
                const uint32_t j = base + k;
                T dx = detail::readLane(xl, k) - xi, dy = detail::readLane(yl, k) - yi, dz = detail::readLane(zl, k) - zi;
                if (withFold && usePbc)
                {
                    dx = detail::foldAxis<T>(dx, box.len[0], box.inv[0], px);
                    dy = detail::foldAxis<T>(dy, box.len[1], box.inv[1], py);
                    dz = detail::foldAxis<T>(dz, box.len[2], box.inv[2], pz);
                }
                const T d2 = dx * dx + dy * dy + dz * dz;
                if (mine && j != i && d2 < radSq)
                {
                    onNeighbor(j, dx, dy, dz, d2);
                    ++nn;
                }
            };
            if (fold)
            {
                for (uint32_t k = 0; k < cnt; ++k)
                    test(k, true);
            }
            else
            {
                uint32_t k = 0;
                for (; k + 4 <= cnt; k += 4)
                {
                    test(k, false);
                    test(k + 1, false);
                    test(k + 2, false);
                    test(k + 3, false);
                }
                for (; k < cnt; ++k)
                    test(k, false);
            }
        }
    };

    // depth-first walk of R/traversal/traversal.hpp:69-110, once per wave
    const bool ov0      = valid && overlaps(0);
    const uint64_t root = __ballot(ov0);
    if (root != 0)
    {
        if (uniform(tree.childOffsets[0]) == 0) { searchLeaf(0, ov0); }
        else
        {
            int top = 1;
            if (lane == 0)
            {
                sNode[0] = 0;
                sMask[0] = root;
            }
            int32_t node  = 0;
            uint64_t mask = root;
            do
            {
                const int32_t c0 = uniform(tree.childOffsets[node]);
                const bool here  = (mask >> lane) & 1ull;
#pragma unroll 1
                for (int oct = 0; oct < 8; ++oct)
                {
                    const int32_t child = c0 + oct;
                    const bool ov       = here && overlaps(child);
                    const uint64_t cm   = __ballot(ov);
                    if (cm == 0) continue;
                    if (uniform(tree.childOffsets[child]) == 0) { searchLeaf(child, ov); }
                    else if (top < TravConfig::stackSize)
                    {
                        if (lane == 0)
                        {
                            sNode[top] = child;
                            sMask[top] = cm;
                        }
                        ++top;
                        stats.depth(top);
                    }
                    else if (lane == 0 && errors) { atomicOr(errors, 4); }
                }
                --top;
                node = uniform(sNode[top]);
                mask = sMask[top];
                mask = (uint64_t(__builtin_amdgcn_readfirstlane(uint32_t(mask >> 32))) << 32) |
                       __builtin_amdgcn_readfirstlane(uint32_t(mask));
            } while (node != 0);
        }
    }
    return nn;
}

} // namespace cstone_hip
