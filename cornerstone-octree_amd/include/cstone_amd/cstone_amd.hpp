/*! @file
 * C++20 host layer over the C ABI of libcstone_hip.so (include/cstone_hip.h).
 *
 * Mirrors the names, argument meaning and error behaviour of the reference's GPU interface
 * (R = /root/reference/include/cstone):
 *   cstone_amd::DeviceVector<T>          R/cuda/device_vector.h:24-63 (data/size/resize/reserve/capacity/swap)
 *   memcpyH2D / memcpyD2H / memcpyD2D    R/cuda/cuda_stubs.h:48-57
 *   computeSfcKeysGpu                    R/sfc/sfc_gpu.h:37-38
 *   sortByKeyGpu, sortByKeyTempStorage, sequenceGpu, gatherGpu, scatterGpu, MinMaxGpu, exclusiveScanGpu,
 *   inclusiveScanGpu                     R/primitives/primitives_gpu.h:36-124
 *   computeNodeCountsGpu, computeNodeOpsGpu, rebalanceTreeGpu     R/tree/csarray_gpu.h:56-88
 *   updateOctreeGpu                      R/tree/update_gpu.cuh:59-82
 *   buildOctreeGpu, upsweepSumGpu        R/tree/octree_gpu.h:47-50
 *   findHalosGpu                         R/traversal/collisions_gpu.h:57-66
 *   Domain<KeyType, T>                   R/domain/domain.hpp:66-699 (GpuTag flavour, one rank in this round)
 * Errors: the reference prints and exits (R/cuda/errorcheck.cuh:30-42) or throws std::runtime_error; every failing C
 * call here throws std::runtime_error carrying cstone_hip_last_error().
 *
 * Header-only, no HIP headers needed: a host compiler (g++/clang++ -std=c++20) and -lcstone_hip are enough.
 */
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <memory>
#include <span>
#include <cmath>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "cstone_hip.h"

namespace cstone_amd
{

using TreeNodeIndex = int;      // R/tree/definitions.h:41
using LocalIndex    = unsigned; // R/tree/definitions.h:43

enum class BoundaryType : char // R/sfc/box.hpp:97-102
{
    open     = 0,
    periodic = 1,
    fixed    = 2
};

//! curve selector; the reference fixes SfcKind = HilbertKey at build time (R/sfc/sfc.hpp:53-55)
enum class Curve : int
{
    morton  = CSTONE_MORTON,
    hilbert = CSTONE_HILBERT
};

//! one context per (host thread, device), created on first use
class Context
{
public:
    static cstone_hip_ctx* get()
    {
        thread_local Context instance;
        return instance.ctx_;
    }

    static void check(int rc, const char* what)
    {
        if (rc != CSTONE_OK)
        {
            throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) +
                                     "): " + cstone_hip_last_error(get()));
        }
    }

private:
    Context()
    {
        if (cstone_hip_ctx_create(&ctx_, 0, nullptr, 1) != CSTONE_OK)
            throw std::runtime_error("cstone_hip_ctx_create failed: no usable MI355X device");
    }
    ~Context() { cstone_hip_ctx_destroy(ctx_); }
    cstone_hip_ctx* ctx_{nullptr};
};

inline void syncGpu() { Context::check(cstone_hip_ctx_sync(Context::get()), "syncGpu"); }

template<class T>
void memcpyH2D(const T* src, std::size_t n, T* dest)
{
    Context::check(cstone_hip_memcpy_h2d(Context::get(), dest, src, n * sizeof(T)), "memcpyH2D");
}
template<class T>
void memcpyD2H(const T* src, std::size_t n, T* dest)
{
    Context::check(cstone_hip_memcpy_d2h(Context::get(), dest, src, n * sizeof(T)), "memcpyD2H");
}
template<class T>
void memcpyD2D(const T* src, std::size_t n, T* dest)
{
    Context::check(cstone_hip_memcpy_d2d(Context::get(), dest, src, n * sizeof(T)), "memcpyD2D");
}

//! uninitialised device array with std::vector-like size management (R/cuda/device_vector.h:24-63)
template<class T>
class DeviceVector
{
public:
    using value_type = T;

    DeviceVector() = default;
    explicit DeviceVector(std::size_t n) { resize(n); }
    DeviceVector(const T* first, const T* last)
    {
        resize(last - first);
        memcpyH2D(first, size_, data_);
    }
    DeviceVector(const DeviceVector& o)
    {
        resize(o.size_);
        memcpyD2D(o.data_, size_, data_);
    }
    DeviceVector(DeviceVector&& o) noexcept { swap(o); }
    DeviceVector& operator=(DeviceVector o)
    {
        swap(o);
        return *this;
    }
    ~DeviceVector()
    {
        if (data_) cstone_hip_free(Context::get(), data_);
    }

    T* data() { return data_; }
    const T* data() const { return data_; }
    std::size_t size() const { return size_; }
    bool empty() const { return size_ == 0; }
    std::size_t capacity() const { return capacity_; }

    void reserve(std::size_t n)
    {
        if (n <= capacity_) return;
        void* p = nullptr;
        Context::check(cstone_hip_malloc(Context::get(), &p, n * sizeof(T)), "DeviceVector::reserve");
        if (size_) memcpyD2D(data_, size_, static_cast<T*>(p));
        if (data_) Context::check(cstone_hip_free(Context::get(), data_), "DeviceVector::reserve");
        data_     = static_cast<T*>(p);
        capacity_ = n;
    }
    void resize(std::size_t n)
    {
        reserve(n);
        size_ = n;
    }
    void swap(DeviceVector& o) noexcept
    {
        std::swap(data_, o.data_);
        std::swap(size_, o.size_);
        std::swap(capacity_, o.capacity_);
    }

    //! take over a buffer handed back by cstone_hip_domain_sync (which exchanges buffers like the reference swaps
    //! vectors); the buffer previously held is NOT freed: it now belongs to another vector of the same call
    void rebind(void* p, std::size_t size, std::size_t capacityBytes)
    {
        data_     = static_cast<T*>(p);
        size_     = size;
        capacity_ = capacityBytes / sizeof(T);
    }
    std::size_t capacityBytes() const { return capacity_ * sizeof(T); }

private:
    T* data_{nullptr};
    std::size_t size_{0}, capacity_{0};
};

template<class T>
T* rawPtr(DeviceVector<T>& v)
{
    return v.data();
}
template<class T>
const T* rawPtr(const DeviceVector<T>& v)
{
    return v.data();
}

template<class T>
std::vector<T> toHost(const DeviceVector<T>& v)
{
    std::vector<T> h(v.size());
    if (!h.empty()) memcpyD2H(v.data(), v.size(), h.data());
    return h;
}

//! global coordinate bounding box (R/sfc/box.hpp:112-191); inverse lengths are derived by the library like Box<T> does
template<class T>
class Box
{
public:
    Box(T xyzMin, T xyzMax, BoundaryType b = BoundaryType::open)
        : Box(xyzMin, xyzMax, xyzMin, xyzMax, xyzMin, xyzMax, b, b, b)
    {
    }
    Box(T xmin, T xmax, T ymin, T ymax, T zmin, T zmax, BoundaryType bx = BoundaryType::open,
        BoundaryType by = BoundaryType::open, BoundaryType bz = BoundaryType::open)
        : pod_{{double(xmin), double(xmax), double(ymin), double(ymax), double(zmin), double(zmax)},
               {int(bx), int(by), int(bz)},
               0}
    {
    }
    explicit Box(const cstone_box& pod)
        : pod_(pod)
    {
    }

    T xmin() const { return T(pod_.lim[0]); }
    T xmax() const { return T(pod_.lim[1]); }
    T ymin() const { return T(pod_.lim[2]); }
    T ymax() const { return T(pod_.lim[3]); }
    T zmin() const { return T(pod_.lim[4]); }
    T zmax() const { return T(pod_.lim[5]); }
    T lx() const { return xmax() - xmin(); }
    T ly() const { return ymax() - ymin(); }
    T lz() const { return zmax() - zmin(); }
    BoundaryType boundaryX() const { return BoundaryType(pod_.bc[0]); }
    BoundaryType boundaryY() const { return BoundaryType(pod_.bc[1]); }
    BoundaryType boundaryZ() const { return BoundaryType(pod_.bc[2]); }
    const cstone_box& pod() const { return pod_; }

    /*! R/sfc/box.hpp:168-176: the box through the client's I/O layer.  Archive: anything with
     *  stepAttribute(name, pointer, count) like the reference's (SPH-EXA's IFileReader / IFileWriter); a reading archive
     *  overwrites the limits and boundary types, a writing one stores them */
    template<class Archive>
    void loadOrStore(Archive* ar)
    {
        T lim[6] = {xmin(), xmax(), ymin(), ymax(), zmin(), zmax()};
        ar->stepAttribute("box", lim, 6);
        char bnd[3] = {char(pod_.bc[0]), char(pod_.bc[1]), char(pod_.bc[2])};
        ar->stepAttribute("boundaryType", bnd, 3);
        for (int k = 0; k < 6; ++k)
            pod_.lim[k] = double(lim[k]);
        for (int d = 0; d < 3; ++d)
            pod_.bc[d] = int(bnd[d]);
    }

private:
    cstone_box pod_;
};

namespace detail
{
template<class K>
constexpr int keyBits()
{
    static_assert(std::is_unsigned_v<K> && (sizeof(K) == 4 || sizeof(K) == 8), "SFC key type: 32- or 64-bit unsigned");
    return 8 * int(sizeof(K));
}
template<class T>
constexpr int realBits()
{
    static_assert(std::is_floating_point_v<T> && (sizeof(T) == 4 || sizeof(T) == 8), "float or double");
    return 8 * int(sizeof(T));
}
} // namespace detail

// ---------------------------------------------------------------------------------------------------------------
// link-seam functions
// ---------------------------------------------------------------------------------------------------------------

template<class KeyType, class T>
void computeSfcKeysGpu(const T* x, const T* y, const T* z, KeyType* keys, std::size_t numKeys, const Box<T>& box,
                       Curve curve = Curve::hilbert)
{
    Context::check(cstone_hip_compute_sfc_keys(Context::get(), int(curve), detail::keyBits<KeyType>(),
                                               detail::realBits<T>(), x, y, z, keys, numKeys, &box.pod()),
                   "computeSfcKeysGpu");
}

template<class KeyType, class ValueType>
std::uint64_t sortByKeyTempStorage(std::uint64_t numElements)
{
    static_assert(sizeof(ValueType) == 4, "the sort payload is a 32-bit LocalIndex");
    return cstone_hip_sort_pairs_temp_bytes(detail::keyBits<KeyType>(), numElements);
}

template<class KeyType, class ValueType>
void sortByKeyGpu(KeyType* first, KeyType* last, ValueType* values, KeyType* keyBuf, ValueType* valueBuf, void* temp,
                  std::uint64_t tempBytes)
{
    static_assert(sizeof(ValueType) == 4, "the sort payload is a 32-bit LocalIndex");
    Context::check(cstone_hip_sort_pairs(Context::get(), detail::keyBits<KeyType>(), first,
                                         reinterpret_cast<std::uint32_t*>(values), std::size_t(last - first), keyBuf,
                                         reinterpret_cast<std::uint32_t*>(valueBuf), temp, tempBytes),
                   "sortByKeyGpu");
}

template<class KeyType, class ValueType>
void sortByKeyGpu(KeyType* first, KeyType* last, ValueType* values)
{
    static_assert(sizeof(ValueType) == 4, "the sort payload is a 32-bit LocalIndex");
    Context::check(cstone_hip_sort_pairs(Context::get(), detail::keyBits<KeyType>(), first,
                                         reinterpret_cast<std::uint32_t*>(values), std::size_t(last - first), nullptr,
                                         nullptr, nullptr, 0),
                   "sortByKeyGpu");
}

inline void sequenceGpu(LocalIndex* input, std::size_t numElements, LocalIndex init)
{
    Context::check(cstone_hip_sequence_u32(Context::get(), input, numElements, init), "sequenceGpu");
}

template<class T, class IndexType>
void gatherGpu(const IndexType* ordering, std::size_t numElements, const T* src, T* buffer)
{
    static_assert(sizeof(IndexType) == 4);
    Context::check(cstone_hip_gather(Context::get(), int(sizeof(T)), reinterpret_cast<const std::uint32_t*>(ordering),
                                     numElements, src, buffer),
                   "gatherGpu");
}

template<class T, class IndexType>
void scatterGpu(const IndexType* ordering, std::size_t numElements, const T* src, T* buffer)
{
    static_assert(sizeof(IndexType) == 4);
    Context::check(cstone_hip_scatter(Context::get(), int(sizeof(T)), reinterpret_cast<const std::uint32_t*>(ordering),
                                      numElements, src, buffer),
                   "scatterGpu");
}

template<class T>
struct MinMaxGpu
{
    std::tuple<T, T> operator()(const T* first, const T* last)
    {
        double mm[2];
        Context::check(cstone_hip_minmax(Context::get(), detail::realBits<T>(), first, std::size_t(last - first), mm),
                       "MinMaxGpu");
        return {T(mm[0]), T(mm[1])};
    }
};

//! lowerBoundGpu, range form (cstone/primitives/primitives_gpu.h:70-71): result[q] = first index with keys[i] >= values[q]
template<class T>
void lowerBoundGpu(const T* first, const T* last, const T* valueFirst, const T* valueLast, std::uint64_t* result)
{
    Context::check(cstone_hip_lower_bound(Context::get(), detail::keyBits<T>(), first, std::size_t(last - first),
                                          valueFirst, int(valueLast - valueFirst), result),
                   "lowerBoundGpu");
}

inline void exclusiveScanGpu(const unsigned* first, const unsigned* last, unsigned* output, unsigned init = 0)
{
    Context::check(cstone_hip_exclusive_scan_u32(Context::get(), first, output, std::size_t(last - first), init),
                   "exclusiveScanGpu");
}

inline void inclusiveScanGpu(const unsigned* first, const unsigned* last, unsigned* output)
{
    Context::check(cstone_hip_inclusive_scan_u32(Context::get(), first, output, std::size_t(last - first)),
                   "inclusiveScanGpu");
}

template<class KeyType>
void computeNodeCountsGpu(const KeyType* tree, unsigned* counts, TreeNodeIndex numNodes, const KeyType* firstKey,
                          const KeyType* lastKey, unsigned maxCount, bool /*useCountsAsGuess*/ = false)
{
    Context::check(cstone_hip_compute_node_counts(Context::get(), detail::keyBits<KeyType>(), tree, counts, numNodes,
                                                  firstKey, std::size_t(lastKey - firstKey), maxCount),
                   "computeNodeCountsGpu");
}

//! returns the new number of nodes; nodeOps holds the exclusive scan of the decisions; *converged as the reference's flag
template<class KeyType>
TreeNodeIndex computeNodeOpsGpu(const KeyType* tree, TreeNodeIndex numNodes, const unsigned* counts,
                                unsigned bucketSize, TreeNodeIndex* nodeOps, bool* converged = nullptr)
{
    int newNumNodes = 0, conv = 0;
    Context::check(cstone_hip_compute_node_ops(Context::get(), detail::keyBits<KeyType>(), tree, numNodes, counts,
                                               bucketSize, nodeOps, &newNumNodes, &conv),
                   "computeNodeOpsGpu");
    if (converged) *converged = conv != 0;
    return newNumNodes;
}

template<class KeyType>
void rebalanceTreeGpu(const KeyType* tree, TreeNodeIndex numNodes, TreeNodeIndex newNumNodes,
                      const TreeNodeIndex* nodeOps, KeyType* newTree)
{
    Context::check(cstone_hip_rebalance_tree(Context::get(), detail::keyBits<KeyType>(), tree, numNodes, newNumNodes,
                                             nodeOps, newTree),
                   "rebalanceTreeGpu");
}

//! one rebalance step + recount on device vectors (R/tree/update_gpu.cuh:59-82); returns the converged flag
template<class KeyType>
bool updateOctreeGpu(const KeyType* firstKey, const KeyType* lastKey, unsigned bucketSize,
                     DeviceVector<KeyType>& tree, DeviceVector<unsigned>& counts,
                     unsigned maxCount = std::numeric_limits<unsigned>::max())
{
    int numLeaves = int(tree.size()) - 1;
    int converged = 0;
    while (true)
    {
        int cap = int(counts.capacity());
        if (int(tree.capacity()) - 1 < cap) cap = int(tree.capacity()) - 1;
        int rc = cstone_hip_update_octree(Context::get(), detail::keyBits<KeyType>(), firstKey,
                                          std::size_t(lastKey - firstKey), bucketSize, tree.data(), counts.data(),
                                          &numLeaves, cap, maxCount, &converged);
        if (rc == CSTONE_E_CAPACITY)
        {
            // numLeaves now holds the required size; grow (contents are preserved) and repeat the step
            std::size_t keep = tree.size();
            tree.reserve(std::size_t(numLeaves * 1.05) + 2);
            counts.reserve(std::size_t(numLeaves * 1.05) + 1);
            numLeaves = int(keep) - 1;
            continue;
        }
        Context::check(rc, "updateOctreeGpu");
        break;
    }
    tree.resize(numLeaves + 1);
    counts.resize(numLeaves);
    return converged != 0;
}

//! R/tree/octree.hpp:280-293
template<class KeyType>
struct OctreeView
{
    TreeNodeIndex numLeafNodes, numInternalNodes, numNodes;
    KeyType* prefixes;
    TreeNodeIndex* childOffsets;
    TreeNodeIndex* parents;
    TreeNodeIndex* levelRange;
    TreeNodeIndex* internalToLeaf;
    TreeNodeIndex* leafToInternal;
};

template<class KeyType>
void buildOctreeGpu(const KeyType* cstoneTree, OctreeView<KeyType> d)
{
    Context::check(cstone_hip_build_octree(Context::get(), detail::keyBits<KeyType>(), cstoneTree, d.numLeafNodes,
                                           d.prefixes, d.childOffsets, d.parents, d.levelRange, d.internalToLeaf,
                                           d.leafToInternal),
                   "buildOctreeGpu");
}

inline void upsweepSumGpu(int numLvl, const TreeNodeIndex* lvlRange, const TreeNodeIndex* childOffsets,
                          LocalIndex* counts)
{
    Context::check(cstone_hip_upsweep_sum(Context::get(), numLvl + 2, lvlRange, childOffsets, counts), "upsweepSumGpu");
}

template<class KeyType, class T>
void findHalosGpu(const KeyType* prefixes, const TreeNodeIndex* childOffsets, const TreeNodeIndex* internalToLeaf,
                  const KeyType* leaves, const float* interactionRadii, const Box<T>& box, TreeNodeIndex firstNode,
                  TreeNodeIndex lastNode, int* collisionFlags, Curve curve = Curve::hilbert)
{
    Context::check(cstone_hip_find_halos(Context::get(), int(curve), detail::keyBits<KeyType>(), detail::realBits<T>(),
                                         prefixes, childOffsets, internalToLeaf, leaves, interactionRadii, &box.pod(),
                                         firstNode, lastNode, collisionFlags),
                   "findHalosGpu");
}

//! R/tree/octree.hpp:297-317: what Domain::octreeProperties() hands to neighbor-search kernels (device pointers)
template<class T, class KeyType>
struct OctreeNsView
{
    TreeNodeIndex numLeafNodes;
    const KeyType* prefixes;
    const TreeNodeIndex* childOffsets;
    const TreeNodeIndex* internalToLeaf;
    const TreeNodeIndex* levelRange;
    const KeyType* leaves;
    const LocalIndex* layout;
    const T* centers; // Vec3<T>[numNodes]
    const T* sizes;   // Vec3<T>[numNodes]
    float searchExtFactor{1.0};
};

//! computeGeoCentersGpu (R/focus/source_center_gpu.h): geometric centers and half sizes of all nodes, Vec3<T>[numNodes] each
template<class KeyType, class T>
void computeGeoCentersGpu(const KeyType* prefixes, TreeNodeIndex numNodes, T* centers, T* sizes, const Box<T>& box,
                          Curve curve = Curve::hilbert)
{
    Context::check(cstone_hip_node_centers(Context::get(), int(curve), detail::keyBits<KeyType>(), detail::realBits<T>(),
                                           prefixes, numNodes, &box.pod(), centers, sizes),
                   "computeGeoCentersGpu");
}

/*! Cornerstone leaf array of a particle CONCENTRATION given as a function (R/tree/continuum.hpp:41-116: continuumCount,
 *  computeContinuumCounts, updateContinuumCsarray, computeContinuumCsarray), for initial conditions.  The rebalance
 *  decisions, the new leaf array and the node geometry are computed on the device by the same kernels the particle
 *  trees use; the concentration -- host code of the caller -- is evaluated on the host at the eight points
 *  center +- size / 2 of every leaf (size = half edge lengths, as centerAndSize returns them) and summed in double like
 *  the reference does: count = round(sum of concentration * sx * sy * sz).
 *  Returns (leaf array, counts) after at most eleven update steps, converged or not (reference: maxIteration = 10). */
template<class KeyType, class F, class T>
std::tuple<std::vector<KeyType>, std::vector<unsigned>>
computeContinuumCsarray(F&& concentration, const Box<T>& box, unsigned bucketSize, Curve curve = Curve::hilbert)
{
    constexpr unsigned maxLevel = sizeof(KeyType) == 8 ? 21 : 10;
    std::vector<KeyType> tree{0, KeyType(1) << (3 * maxLevel)};
    std::vector<unsigned> counts{bucketSize + 1};

    DeviceVector<KeyType> dTree, dNewTree, dPrefixes;
    DeviceVector<unsigned> dCounts;
    DeviceVector<TreeNodeIndex> dOps;
    DeviceVector<T> dCenters, dSizes;
    std::vector<KeyType> prefixes;
    std::vector<T> centers, sizes;

    int maxIteration = 10;
    bool converged   = false;
    do
    {
        const TreeNodeIndex numNodes = TreeNodeIndex(tree.size()) - 1;
        dTree.resize(tree.size());
        dCounts.resize(counts.size());
        dOps.resize(tree.size());
        memcpyH2D(tree.data(), tree.size(), dTree.data());
        memcpyH2D(counts.data(), counts.size(), dCounts.data());
        const TreeNodeIndex newNumNodes =
            computeNodeOpsGpu(dTree.data(), numNodes, dCounts.data(), bucketSize, dOps.data(), &converged);
        dNewTree.resize(std::size_t(newNumNodes) + 1);
        rebalanceTreeGpu(dTree.data(), numNodes, newNumNodes, dOps.data(), dNewTree.data());
        tree.resize(std::size_t(newNumNodes) + 1);
        memcpyD2H(dNewTree.data(), tree.size(), tree.data());

        // node keys of the leaves (level + start, R/sfc/common.hpp:163-171) -> centers and half sizes on the device
        prefixes.resize(newNumNodes);
        for (TreeNodeIndex i = 0; i < newNumNodes; ++i)
        {
            const KeyType span = tree[i + 1] - tree[i];
            unsigned level     = 0;
            while ((KeyType(1) << (3 * (maxLevel - level))) > span)
                ++level;
            prefixes[i] = (KeyType(1) << (3 * level)) | (tree[i] >> (3 * (maxLevel - level)));
        }
        dPrefixes.resize(newNumNodes);
        dCenters.resize(std::size_t(newNumNodes) * 3);
        dSizes.resize(std::size_t(newNumNodes) * 3);
        memcpyH2D(prefixes.data(), prefixes.size(), dPrefixes.data());
        computeGeoCentersGpu(dPrefixes.data(), newNumNodes, dCenters.data(), dSizes.data(), box, curve);
        centers.resize(dCenters.size());
        sizes.resize(dSizes.size());
        memcpyD2H(dCenters.data(), centers.size(), centers.data());
        memcpyD2H(dSizes.data(), sizes.size(), sizes.data());

        counts.resize(newNumNodes);
        for (TreeNodeIndex i = 0; i < newNumNodes; ++i)
        {
            const T* c   = &centers[3 * std::size_t(i)];
            const T* sz  = &sizes[3 * std::size_t(i)];
            const T vol  = sz[0] * sz[1] * sz[2];
            double count = 0;
            for (int ix = -1; ix <= 1; ix += 2)
                for (int iy = -1; iy <= 1; iy += 2)
                    for (int iz = -1; iz <= 1; iz += 2)
                    {
                        const T cx = c[0] + T(0.5) * (T(ix) * sz[0]);
                        const T cy = c[1] + T(0.5) * (T(iy) * sz[1]);
                        const T cz = c[2] + T(0.5) * (T(iz) * sz[2]);
                        count += concentration(cx, cy, cz) * vol;
                    }
            const double r = std::round(count);
            counts[i]      = r >= double(std::numeric_limits<unsigned>::max()) ? std::numeric_limits<unsigned>::max()
                                                                               : unsigned(r);
        }
    } while (!converged && maxIteration--);

    return std::make_tuple(std::move(tree), std::move(counts));
}

//! segmentMax + scaleGpu as Halos::discover uses them (R/halos/halos.hpp:150-160): radii[i] = 2 ext max(h) over leaf i
template<class T>
void haloRadiiGpu(const T* h, const LocalIndex* layout, TreeNodeIndex firstLeaf, TreeNodeIndex lastLeaf,
                  TreeNodeIndex numLeaves, float searchExtFactor, float* radii)
{
    Context::check(cstone_hip_halo_radii(Context::get(), detail::realBits<T>(), h, layout, firstLeaf, lastLeaf, numLeaves,
                                         searchExtFactor, radii),
                   "haloRadiiGpu");
}

/*! findNeighbors (R/findneighbors.hpp:160-188) for the particles [first, last) on an OctreeNsView:
 *  neighborsCount[i - first] = number of j != i with |r_i - r_j|^2 < (2 h_i)^2 (minimum image on periodic axes), the
 *  first ngmax of them in neighbors[(i - first) * ngmax + k] (device arrays; CPU order of the reference) */
template<class T, class KeyType>
void findNeighborsGpu(const T* x, const T* y, const T* z, const T* h, LocalIndex first, LocalIndex last,
                      const Box<T>& box, const OctreeNsView<T, KeyType>& tree, unsigned ngmax, LocalIndex* neighbors,
                      unsigned* neighborsCount)
{
    Context::check(cstone_hip_find_neighbors(Context::get(), detail::realBits<T>(), x, y, z, h, first, last, &box.pod(),
                                             tree.childOffsets, tree.internalToLeaf, tree.layout, tree.centers,
                                             tree.sizes, tree.searchExtFactor, ngmax, neighbors, neighborsCount),
                   "findNeighborsGpu");
}

//! R/traversal/groups.hpp:20-26: groups of target particles, device pointers
struct GroupView
{
    LocalIndex firstBody, lastBody;
    LocalIndex numGroups;
    const LocalIndex* groupStart;
    const LocalIndex* groupEnd;
};

//! R/traversal/groups.hpp:29-57 (the GpuTag flavour: the offsets live in device memory)
class GroupData
{
public:
    GroupData()                 = default;
    GroupData(const GroupView&) = delete;
    GroupView view() const { return {firstBody, lastBody, numGroups, groupStart, groupEnd}; }

    DeviceVector<LocalIndex> data;
    LocalIndex firstBody{0}, lastBody{0};
    LocalIndex numGroups{0};
    LocalIndex* groupStart{nullptr};
    LocalIndex* groupEnd{nullptr};
};

//! R/traversal/groups_gpu.h:46: groups of groupSize consecutive particles
inline void computeFixedGroups(LocalIndex first, LocalIndex last, unsigned groupSize, GroupData& groups)
{
    if (groupSize == 0) throw std::runtime_error("Unsupported spatial group size\n");
    groups.data.resize((last - first + groupSize - 1) / groupSize + 1);
    LocalIndex numGroups = 0;
    Context::check(cstone_hip_compute_fixed_groups(Context::get(), first, last, groupSize, groups.data.data(), &numGroups),
                   "computeFixedGroups");
    groups.firstBody = first, groups.lastBody = last, groups.numGroups = numGroups;
    groups.groupStart = groups.data.data(), groups.groupEnd = groups.data.data() + 1;
}

/*! R/traversal/groups_gpu.h:73-87: groups of at most groupSize particles, split where consecutive particles are farther
 *  apart than tolFactor x the edge of the group's smallest leaf.  h and the scratch vector of the reference's signature
 *  are accepted and not used (the smoothing lengths do not enter the reference's result either). */
template<class Tc, class T, class KeyType>
void computeGroupSplits(LocalIndex first, LocalIndex last, const Tc* x, const Tc* y, const Tc* z, const T* /*h*/,
                        const KeyType* leaves, TreeNodeIndex numLeaves, const LocalIndex* layout, const Box<Tc>& box,
                        unsigned groupSize, float tolFactor, DeviceVector<LocalIndex>& /*numSplitsPerGroup*/,
                        DeviceVector<LocalIndex>& groups)
{
    if (groupSize != 64 && groupSize != 128) throw std::runtime_error("Unsupported spatial group size\n");
    LocalIndex numFixed = (last - first + groupSize - 1) / groupSize;
    groups.resize(std::size_t(numFixed) + numFixed / 8 + 2); // the reference reserves 1.1 x; grow on demand
    LocalIndex numGroups = 0;
    int rc = cstone_hip_compute_group_splits(Context::get(), detail::keyBits<KeyType>(), detail::realBits<Tc>(), first,
                                             last, x, y, z, leaves, numLeaves, layout, &box.pod(), groupSize, tolFactor,
                                             groups.data(), groups.size(), &numGroups);
    if (rc == CSTONE_E_CAPACITY)
    {
        groups.resize(std::size_t(numGroups) + 1);
        rc = cstone_hip_compute_group_splits(Context::get(), detail::keyBits<KeyType>(), detail::realBits<Tc>(), first,
                                             last, x, y, z, leaves, numLeaves, layout, &box.pod(), groupSize, tolFactor,
                                             groups.data(), groups.size(), &numGroups);
    }
    Context::check(rc, "computeGroupSplits");
    groups.resize(std::size_t(numGroups) + 1);
}

//! findNeighbors with the targets of every wavefront taken from one group of a GroupView
template<class T, class KeyType>
void findNeighborsGpu(const T* x, const T* y, const T* z, const T* h, const GroupView& groups, const Box<T>& box,
                      const OctreeNsView<T, KeyType>& tree, unsigned ngmax, LocalIndex* neighbors,
                      unsigned* neighborsCount)
{
    Context::check(cstone_hip_find_neighbors_groups(Context::get(), detail::realBits<T>(), x, y, z, h, groups.firstBody,
                                                    groups.lastBody, groups.groupStart, groups.groupEnd, groups.numGroups,
                                                    &box.pod(), tree.childOffsets, tree.internalToLeaf, tree.layout,
                                                    tree.centers, tree.sizes, tree.searchExtFactor, ngmax, neighbors,
                                                    neighborsCount),
                   "findNeighborsGpu");
}

/*! The transport of the multi-rank Domain on a node of MI355X: RCCL, served from C++ inside libcstone_hip on the context's
 *  stream (csrc/comm_rccl.hip) -- replaces the MPI point-to-point exchanges of the reference
 *  (R/domain/domaindecomp_mpi_gpu.cuh:86-185, R/halos/exchange_halos_gpu.cuh:35-119).  One rank obtains the id
 *  (RcclComm::uniqueId) and passes its 128 bytes to the others by whatever means the application has; every rank then
 *  constructs an RcclComm (collective) and hands ops() to Domain / MultiRankDomain. */
class RcclComm
{
public:
    static std::array<char, 128> uniqueId()
    {
        std::array<char, 128> id{};
        Context::check(cstone_hip_comm_rccl_unique_id(Context::get(), id.data()), "RcclComm::uniqueId");
        return id;
    }
    RcclComm(const std::array<char, 128>& id, int rank, int nRanks)
    {
        Context::check(cstone_hip_comm_rccl_create(Context::get(), id.data(), rank, nRanks, &comm_), "RcclComm");
        Context::check(cstone_hip_comm_rccl_ops(comm_, &ops_), "RcclComm::ops");
    }
    RcclComm(const RcclComm&)            = delete;
    RcclComm& operator=(const RcclComm&) = delete;
    ~RcclComm()
    {
        if (comm_) cstone_hip_comm_rccl_destroy(comm_);
    }
    const cstone_hip_comm_ops& ops() const { return ops_; }

private:
    cstone_hip_comm_rccl* comm_ = nullptr;
    cstone_hip_comm_ops ops_{};
};

/*! cstone::Domain<KeyType, T, GpuTag> on SEVERAL ranks, one process per GPU (cstone_hip_domain_mr_*).
 *  The three collectives of a sync (all-reduce, all-gather, all-to-all-v on device buffers) are supplied by the
 *  application through cstone_hip_comm_ops: RCCL, MPI, or any other transport.  Results live in domain-owned arrays:
 *  [halos of lower ranks | assigned, SFC sorted | halos of higher ranks], valid until the next but one sync, so a
 *  client may update x()[startIndex() .. endIndex()) in place and pass that range back as the next input. */
template<class KeyType, class T>
class MultiRankDomain
{
public:
    /*! theta: the opening angle of the focus tree's MAC (Domain ctor, R/domain/domain.hpp:95-113); ownerSideHalos:
     *  CSTONE_MR_HALOS_OWNER_SIDE instead of the reference's locally essential tree (cstone_hip.h) */
    MultiRankDomain(int rank, int nRanks, unsigned bucketSize, unsigned bucketSizeFocus, const Box<T>& box,
                    const cstone_hip_comm_ops& comm, Curve curve = Curve::hilbert, float theta = 0.5f,
                    bool ownerSideHalos = false)
    {
        int rc = cstone_hip_domain_mr_create(Context::get(), &dom_, int(curve), detail::keyBits<KeyType>(),
                                             detail::realBits<T>(), rank, nRanks, bucketSize, bucketSizeFocus,
                                             &box.pod(), &comm);
        Context::check(rc, "MultiRankDomain");
        Context::check(cstone_hip_domain_mr_set_theta(dom_, theta), "MultiRankDomain (theta)");
        if (ownerSideHalos)
            Context::check(cstone_hip_domain_mr_set_halo_mode(dom_, CSTONE_MR_HALOS_OWNER_SIDE), "MultiRankDomain (halos)");
    }
    MultiRankDomain(const MultiRankDomain&)            = delete;
    MultiRankDomain& operator=(const MultiRankDomain&) = delete;
    ~MultiRankDomain()
    {
        if (dom_) cstone_hip_domain_mr_destroy(dom_);
    }

    //! x, y, z, h: device pointers to this rank's n particles in any order; properties: further conserved fields with
    //! elements of 1..32 bytes that follow their particles (results in view().props[i], halo ranges via exchangeHalos)
    template<class... Props>
    void sync(const T* x, const T* y, const T* z, const T* h, std::size_t n, const Props*... properties)
    {
        static_assert(((sizeof(Props) == 1 || sizeof(Props) == 2 || sizeof(Props) == 4 || sizeof(Props) == 8 ||
                        sizeof(Props) == 12 || sizeof(Props) == 16 || sizeof(Props) == 24 || sizeof(Props) == 32) &&
                       ...));
        constexpr int np = sizeof...(Props);
        const void* pp[np + 1] = {static_cast<const void*>(properties)..., nullptr};
        const int pb[np + 1]   = {int(sizeof(Props))..., 0};
        Context::check(cstone_hip_domain_mr_sync_props(dom_, x, y, z, h, n, pp, pb, np), "MultiRankDomain::sync");
        Context::check(cstone_hip_domain_mr_view_get(dom_, &view_), "MultiRankDomain::view");
    }
    //! the same with the caller's key array, whose remove markers flag particles that leave the domain
    template<class... Props>
    void syncKeys(const KeyType* keys, const T* x, const T* y, const T* z, const T* h, std::size_t n,
                  const Props*... properties)
    {
        constexpr int np = sizeof...(Props);
        const void* pp[np + 1] = {static_cast<const void*>(properties)..., nullptr};
        const int pb[np + 1]   = {int(sizeof(Props))..., 0};
        Context::check(cstone_hip_domain_mr_sync_keys(dom_, keys, x, y, z, h, n, pp, pb, np), "MultiRankDomain::sync");
        Context::check(cstone_hip_domain_mr_view_get(dom_, &view_), "MultiRankDomain::view");
    }
    /*! Domain::syncGrav (R/domain/domain.hpp:246-325): like syncKeys (keys may be null), the masses m follow their
     *  particles as one more property behind the given ones -- masses() afterwards -- and the focus tree is resolved by
     *  the vector MAC on the mass centres of its nodes (expansionCenters()).  Tm: float or double */
    template<class Tm, class... Props>
    void syncGrav(const KeyType* keys, const T* x, const T* y, const T* z, const T* h, const Tm* m, std::size_t n,
                  const Props*... properties)
    {
        static_assert(std::is_same_v<Tm, float> || std::is_same_v<Tm, double>);
        constexpr int np = sizeof...(Props);
        const void* pp[np + 1] = {static_cast<const void*>(properties)..., nullptr};
        const int pb[np + 1]   = {int(sizeof(Props))..., 0};
        Context::check(cstone_hip_domain_mr_sync_grav(dom_, keys, x, y, z, h, m, int(sizeof(Tm)) * 8, n, pp, pb, np),
                       "MultiRankDomain::syncGrav");
        Context::check(cstone_hip_domain_mr_view_get(dom_, &view_), "MultiRankDomain::view");
        massProp_ = np;
    }
    //! the masses of the last syncGrav, laid out like x() (device)
    template<class Tm>
    Tm* masses() const
    {
        return massProp_ < 0 ? nullptr : property<Tm>(massProp_);
    }
    /*! Domain::updateExpansionCenters (R/domain/domain.hpp:415-421): mass centres and MAC radii of the focus tree from
     *  the particles as they are now (arrays laid out like the result arrays; collective) */
    template<class Tm>
    void updateExpansionCenters(const T* x, const T* y, const T* z, const Tm* m)
    {
        Context::check(cstone_hip_domain_mr_update_expansion_centers(dom_, x, y, z, m, int(sizeof(Tm)) * 8),
                       "MultiRankDomain::updateExpansionCenters");
    }
    //! FocusedOctree::expansionCenters(): (centre of mass, MAC radius^2) per node of the focus tree after syncGrav /
    //! updateExpansionCenters, device pointer to 4 values of T per node; null before
    const T* expansionCenters() const
    {
        cstone_hip_domain_mr_octree o;
        Context::check(cstone_hip_domain_mr_octree_get(dom_, &o), "MultiRankDomain::expansionCenters");
        return static_cast<const T*>(o.expansion_centers);
    }
    template<class V>
    V* property(int i) const
    {
        return static_cast<V*>(const_cast<void*>(view_.props[i]));
    }

    LocalIndex startIndex() const { return view_.start_index; }
    LocalIndex endIndex() const { return view_.end_index; }
    LocalIndex nParticles() const { return endIndex() - startIndex(); }
    LocalIndex nParticlesWithHalos() const { return view_.num_particles_with_halos; }
    Box<T> box() const { return Box<T>(view_.box); }
    const KeyType* keys() const { return static_cast<const KeyType*>(view_.keys); }
    T* x() const { return static_cast<T*>(const_cast<void*>(view_.x)); }
    T* y() const { return static_cast<T*>(const_cast<void*>(view_.y)); }
    T* z() const { return static_cast<T*>(const_cast<void*>(view_.z)); }
    T* h() const { return static_cast<T*>(const_cast<void*>(view_.h)); }
    //! the rank's SFC key range [first, second)
    std::pair<KeyType, KeyType> assignedRange() const { return {KeyType(view_.range_start), KeyType(view_.range_end)}; }
    const cstone_hip_domain_mr_view& view() const { return view_; }
    void setHaloFactor(float factor)
    {
        Context::check(cstone_hip_domain_mr_set_halo_factor(dom_, factor), "MultiRankDomain::setHaloFactor");
    }
    //! CSTONE_SORT_INCREMENTAL (default) / _FROM_SCRATCH / _ALL_DIGITS: how a sync orders the particles (same results)
    void setSortMode(int mode)
    {
        Context::check(cstone_hip_domain_mr_set_sort_mode(dom_, mode), "MultiRankDomain::setSortMode");
    }
    //! Domain::exchangeHalos for one more field (device array of nParticlesWithHalos elements of 1..32 bytes)
    template<class V>
    void exchangeHalos(V* field) const
    {
        static_assert(sizeof(V) == 1 || sizeof(V) == 2 || sizeof(V) == 4 || sizeof(V) == 8 || sizeof(V) == 12 ||
                      sizeof(V) == 16 || sizeof(V) == 24 || sizeof(V) == 32);
        Context::check(cstone_hip_domain_mr_exchange_halos(dom_, field, int(sizeof(V))), "MultiRankDomain::exchangeHalos");
    }

    /*! Domain::octreeProperties() (R/domain/domain.hpp:424-437): the tree over all local particles, halos included, for
     *  neighbor searches on the result arrays; built on the first request after a sync */
    OctreeNsView<T, KeyType> octreeProperties() const
    {
        cstone_hip_domain_mr_octree o;
        Context::check(cstone_hip_domain_mr_octree_get(dom_, &o), "MultiRankDomain::octreeProperties");
        return {o.num_leaves,
                static_cast<const KeyType*>(o.prefixes),
                o.child_offsets,
                o.internal_to_leaf,
                o.level_range,
                static_cast<const KeyType*>(o.leaves),
                o.layout,
                static_cast<const T*>(o.centers),
                static_cast<const T*>(o.sizes)};
    }
    //! Domain::layout(): particle offsets of the leaf cells of octreeProperties(), device pointer
    std::span<const LocalIndex> layout() const
    {
        auto o = octreeProperties();
        return {o.layout, std::size_t(o.numLeafNodes) + 1};
    }

    /*! Domain::reapplySync (R/domain/domain.hpp:334-378) for one more field: in holds the n elements of the last sync's
     *  input arrays, out nParticlesWithHalos() elements; the assigned range of out is written (collective call) */
    template<class V>
    void reapplySync(const V* in, std::size_t n, V* out) const
    {
        static_assert(sizeof(V) == 1 || sizeof(V) == 2 || sizeof(V) == 4 || sizeof(V) == 8 || sizeof(V) == 12 ||
                      sizeof(V) == 16 || sizeof(V) == 24 || sizeof(V) == 32);
        Context::check(cstone_hip_domain_mr_reapply_sync(dom_, in, n, int(sizeof(V)), out), "MultiRankDomain::reapplySync");
    }

private:
    cstone_hip_domain_mr* dom_ = nullptr;
    cstone_hip_domain_mr_view view_{};
    int massProp_ = -1;
};

/*! cstone::Domain<KeyType, T, GpuTag> (R/domain/domain.hpp:66-699).
 *  sync() keeps the reference's contract: caller-owned device vectors that are resized and, on one rank, SWAPPED with
 *  the scratch vector (properties then need element sizes <= sizeof(T)).  On several ranks (constructor with a
 *  cstone_hip_comm_ops) the work is done by a MultiRankDomain and the results are copied into the caller's vectors. */
template<class KeyType, class T>
class Domain
{
public:
    using RealType = T;

    Domain(int rank, int nRanks, unsigned bucketSize, unsigned bucketSizeFocus, float theta,
           const Box<T>& box = Box<T>{0, 1}, Curve curve = Curve::hilbert)
    {
        int rc = cstone_hip_domain_create(Context::get(), &dom_, int(curve), detail::keyBits<KeyType>(),
                                          detail::realBits<T>(), rank, nRanks, bucketSize, bucketSizeFocus, theta,
                                          &box.pod());
        Context::check(rc, "Domain");
    }
    /*! several ranks: the same class, with the collectives of the application (cstone_hip_comm_ops).  sync() then keeps
     *  the reference's contract on the caller's vectors -- pass the arrays of the previous layout, get them back resized
     *  to nParticlesWithHalos() with the assigned range [startIndex(), endIndex()) and the halos of x, y, z, h filled --
     *  at the price of one device copy per array out of the domain-owned result buffers; MultiRankDomain is the
     *  zero-copy interface underneath. */
    Domain(int rank, int nRanks, unsigned bucketSize, unsigned bucketSizeFocus, float theta, const Box<T>& box,
           const cstone_hip_comm_ops& comm, Curve curve = Curve::hilbert)
        : mr_(std::make_unique<MultiRankDomain<KeyType, T>>(rank, nRanks, bucketSize, bucketSizeFocus, box, comm, curve,
                                                            theta))
    {
        if (bucketSize < bucketSizeFocus)
            throw std::runtime_error("The bucket size of the global tree must not be smaller than the bucket size"
                                     " of the focused tree\n");
    }
    Domain(const Domain&)            = delete;
    Domain& operator=(const Domain&) = delete;
    ~Domain()
    {
        if (dom_) cstone_hip_domain_destroy(dom_);
    }

    template<class... Props>
    void sync(DeviceVector<KeyType>& keys, DeviceVector<T>& x, DeviceVector<T>& y, DeviceVector<T>& z,
              DeviceVector<T>& h, std::tuple<DeviceVector<Props>&...> properties, DeviceVector<T>& scratch)
    {
        std::size_t n = x.size();
        if (keys.size() != n || y.size() != n || z.size() != n || h.size() != n)
            throw std::runtime_error("Domain sync: input array sizes are inconsistent\n");
        if (mr_)
        {
            syncMultiRank(keys, x, y, z, h, properties);
            return;
        }
        static_assert(((sizeof(Props) <= sizeof(T)) && ...), "properties must not be wider than the scratch element");
        scratch.resize(n);
        bool tooSmall = false;
        std::apply([&](auto&... v) { ((tooSmall = tooSmall || v.capacityBytes() < n * sizeof(T)), ...); }, properties);
        if (tooSmall)
            throw std::runtime_error("Domain sync: a property buffer is smaller than n * sizeof(T) bytes "
                                     "(it is exchanged with the scratch buffer)\n");
        constexpr int np = sizeof...(Props);
        void* pk = keys.data();
        void* px = x.data();
        void* py = y.data();
        void* pz = z.data();
        void* ph = h.data();
        void* ps = scratch.data();
        void* pp[np + 1] = {};
        int pb[np + 1]   = {};
        int i            = 0;
        std::apply([&](auto&... v) { ((pp[i] = v.data(), pb[i] = int(sizeof(*v.data())), ++i), ...); }, properties);
        void* before[5 + np + 1] = {px, py, pz, ph, ps};
        for (int k = 0; k < np; ++k)
            before[5 + k] = pp[k];
        Context::check(cstone_hip_domain_sync(dom_, &pk, &px, &py, &pz, &ph, n, &ps, pp, pb, np), "Domain::sync");
        // the library exchanged buffers among {x, y, z, h, scratch, props...}: rebind the vectors without copying.
        // Every buffer of the set must offer n elements of sizeof(T); remember each buffer's capacity by address.
        cstone_hip_domain_view v;
        Context::check(cstone_hip_domain_view_get(dom_, &v), "Domain::sync");
        const std::size_t m = v.num_particles_with_halos;
        std::size_t caps[5 + np + 1] = {x.capacityBytes(), y.capacityBytes(), z.capacityBytes(), h.capacityBytes(),
                                        scratch.capacityBytes()};
        i = 0;
        std::apply([&](auto&... vec) { ((caps[5 + i++] = vec.capacityBytes()), ...); }, properties);
        auto capOf = [&](void* p)
        {
            for (int k = 0; k < 5 + np; ++k)
                if (before[k] == p) return caps[k];
            throw std::runtime_error("Domain::sync: library returned a foreign buffer");
        };
        x.rebind(px, m, capOf(px));
        y.rebind(py, m, capOf(py));
        z.rebind(pz, m, capOf(pz));
        h.rebind(ph, m, capOf(ph));
        scratch.rebind(ps, n, capOf(ps));
        keys.resize(m);
        i = 0;
        std::apply([&](auto&... vec) { (vec.rebind(pp[i], m, capOf(pp[i])), ..., void(++i)); }, properties);
    }

    /*! the reference's signature (R/domain/domain.hpp:196-203): the scratch buffers come as a tuple; the first one with
     *  elements of type T takes the role of the scratch vector above (the others are left alone: the orderings the
     *  reference parks in them live inside the library) */
    template<class... Props, class... Scratch>
    void sync(DeviceVector<KeyType>& keys, DeviceVector<T>& x, DeviceVector<T>& y, DeviceVector<T>& z,
              DeviceVector<T>& h, std::tuple<DeviceVector<Props>&...> properties, std::tuple<Scratch&...> scratchBuffers)
    {
        static_assert((std::is_same_v<Scratch, DeviceVector<T>> || ...),
                      "one of the scratch buffers must be a DeviceVector<T>");
        DeviceVector<T>* first = nullptr;
        std::apply(
            [&](auto&... s)
            {
                auto pick = [&](auto& v)
                {
                    if constexpr (std::is_same_v<std::decay_t<decltype(v)>, DeviceVector<T>>)
                        if (!first) first = &v;
                };
                (pick(s), ...);
            },
            scratchBuffers);
        sync(keys, x, y, z, h, properties, *first);
    }

    /*! Domain::globalTree() (R/domain/domain.hpp:388-392): the leaves of the global (assignment) tree, device pointer,
     *  numGlobalLeaves + 1 keys; globalCounts(): their particle counts */
    std::span<const KeyType> globalTree() const
    {
        if (mr_)
        {
            const auto& v = mr_->view();
            return {static_cast<const KeyType*>(v.global_leaves), std::size_t(v.num_global_leaves) + 1};
        }
        auto v = view();
        return {static_cast<const KeyType*>(v.global_leaves), std::size_t(v.num_global_leaves) + 1};
    }
    //! what clients read from Domain::focusTree() (R/domain/domain.hpp:394-398): leaves and leaf counts, device pointers
    struct FocusTreeView
    {
        std::span<const KeyType> leaves;
        std::span<const unsigned> counts;
        std::span<const KeyType> treeLeaves() const { return leaves; }
        std::span<const unsigned> leafCounts() const { return counts; }
    };
    FocusTreeView focusTree() const
    {
        if (mr_)
        {
            const auto& v = mr_->view();
            return {{static_cast<const KeyType*>(v.focus_leaves), std::size_t(v.num_focus_leaves) + 1},
                    {v.focus_leaf_counts, std::size_t(v.num_focus_leaves)}};
        }
        auto v = view();
        return {{static_cast<const KeyType*>(v.focus_leaves), std::size_t(v.num_focus_leaves) + 1},
                {v.focus_leaf_counts, std::size_t(v.num_focus_leaves)}};
    }

    //! single-rank mode only: the raw view of the C ABI (multi-rank: multiRank().view())
    cstone_hip_domain_view view() const
    {
        cstone_hip_domain_view v;
        Context::check(cstone_hip_domain_view_get(dom_, &v), "Domain::view");
        return v;
    }
    const MultiRankDomain<KeyType, T>& multiRank() const { return *mr_; }
    LocalIndex startIndex() const { return mr_ ? mr_->startIndex() : view().start_index; }
    LocalIndex endIndex() const { return mr_ ? mr_->endIndex() : view().end_index; }
    LocalIndex nParticles() const { return endIndex() - startIndex(); }
    LocalIndex nParticlesWithHalos() const { return mr_ ? mr_->nParticlesWithHalos() : view().num_particles_with_halos; }
    Box<T> box() const { return mr_ ? mr_->box() : Box<T>(view().box); }
    //! the rank's own cells in focusTree() (R/domain/domain.hpp:403-405)
    TreeNodeIndex startCell() const { return mr_ ? mr_->view().start_cell : 0; }
    TreeNodeIndex endCell() const { return mr_ ? mr_->view().end_cell : view().num_focus_leaves; }
    /*! R/domain/domain.hpp:393: the client has appended particles behind the assigned range of its arrays; the next sync
     *  takes [startIndex(), i) as this rank's present particles */
    void setEndIndex(std::size_t i) { endOverride_ = LocalIndex(i), haveEndOverride_ = true; }
    //! particle offsets of each focus tree leaf cell, device pointer (Domain::layout)
    std::span<const LocalIndex> layout() const
    {
        if (mr_) return mr_->layout();
        auto v = view();
        return {v.layout, std::size_t(v.num_focus_leaves) + 1};
    }
    void setHaloFactor(float factor)
    {
        if (mr_) { mr_->setHaloFactor(factor); }
        else { Context::check(cstone_hip_domain_set_halo_factor(dom_, factor), "Domain::setHaloFactor"); }
    }
    //! CSTONE_SORT_INCREMENTAL (default) / _FROM_SCRATCH / _ALL_DIGITS: how a sync orders the particles (same results)
    void setSortMode(int mode)
    {
        if (mr_) { mr_->setSortMode(mode); }
        else { Context::check(cstone_hip_domain_set_sort_mode(dom_, mode), "Domain::setSortMode"); }
    }
    //! false: the extents of an open box are measured before the keys are computed, every sync (single-rank domain)
    void setSpeculativeBox(bool on)
    {
        if (!mr_) Context::check(cstone_hip_domain_set_speculative_box(dom_, on ? 1 : 0), "Domain::setSpeculativeBox");
    }
    /*! Domain::updateExpansionCenters (R/domain/domain.hpp:415-421): (centre of mass, MAC radius^2) per node of the focus
     *  tree from the particles as they are now; x, y, z, m: arrays laid out like the result arrays of the last sync.
     *  sync(keys, x, y, z, h, std::tie(m, ...), scratch) followed by this call is Domain::syncGrav on one rank (the tree
     *  does not depend on the MACs there); on several ranks: MultiRankDomain::syncGrav */
    template<class Tm>
    void updateExpansionCenters(const DeviceVector<T>& x, const DeviceVector<T>& y, const DeviceVector<T>& z,
                                const DeviceVector<Tm>& m)
    {
        static_assert(std::is_same_v<Tm, float> || std::is_same_v<Tm, double>);
        if (mr_) { mr_->updateExpansionCenters(x.data(), y.data(), z.data(), m.data()); }
        else
        {
            Context::check(cstone_hip_domain_update_expansion_centers(dom_, x.data(), y.data(), z.data(), m.data(),
                                                                      int(sizeof(Tm)) * 8),
                           "Domain::updateExpansionCenters");
        }
    }
    //! FocusedOctree::expansionCenters(): device pointer to 4 values of T per node of the focus tree, null before
    //! updateExpansionCenters / after the next sync
    const T* expansionCenters() const
    {
        return mr_ ? mr_->expansionCenters() : static_cast<const T*>(view().expansion_centers);
    }
    //! R/domain/domain.hpp:411: stores the flag like the reference does (its member convergeTrees, :661, has no reader
    //! there either: the trees converge on the first sync and take one update step per sync afterwards)
    void setTreeConv(bool flag) { convergeTrees_ = flag; }
    bool treeConv() const { return convergeTrees_; }
    /*! R/domain/domain.hpp:381-386: the halo ranges of every array (nParticlesWithHalos() elements) are overwritten with
     *  the owners' values; the send/receive buffers of the reference's signature are not needed.  One rank: no halos. */
    template<class... Vectors, class SendBuffer, class ReceiveBuffer>
    void exchangeHalos(std::tuple<Vectors&...> arrays, SendBuffer&, ReceiveBuffer&) const
    {
        if (!mr_) return;
        std::apply(
            [this](auto&... a)
            {
                ((a.size() == nParticlesWithHalos()
                      ? void()
                      : throw std::runtime_error("Domain exchangeHalos: input array sizes are inconsistent\n")),
                 ...);
                (mr_->exchangeHalos(a.data()), ...);
            },
            arrays);
    }

    /*! R/domain/domain.hpp:334-378 with device vectors (the reference restricts its version to host vectors, :340):
     *  every array of the last sync's input size is brought into the order of the result through the scratch vector,
     *  which it is swapped with; the ordering argument of the reference lives inside the domain */
    template<class... Vectors, class Scratch>
    void reapplySync(std::tuple<Vectors&...> arrays, Scratch& scratch) const
    {
        auto one = [&](auto& a)
        {
            using V = std::decay_t<decltype(*a.data())>;
            static_assert(std::is_same_v<V, std::decay_t<decltype(*scratch.data())>>,
                          "the scratch vector must have the arrays' element type");
            const std::size_t m = nParticlesWithHalos();
            scratch.resize(std::max(m, a.size()));
            if (mr_)
            {
                // the arrays have the layout BEFORE the last sync: their assigned range then is what travelled
                if (a.size() != prevSize_)
                    throw std::runtime_error("Domain reapplySync: input array sizes are inconsistent\n");
                mr_->reapplySync(a.data() + prevStart_, std::size_t(prevEnd_ - prevStart_), scratch.data());
            }
            else
            {
                Context::check(cstone_hip_domain_reapply_sync(dom_, a.data(), a.size(), int(sizeof(V)), scratch.data()),
                               "Domain::reapplySync");
            }
            scratch.resize(m);
            a.swap(scratch);
        };
        std::apply([&](auto&... a) { (one(a), ...); }, arrays);
    }

    OctreeNsView<T, KeyType> octreeProperties() const
    {
        if (mr_) return mr_->octreeProperties();
        auto v = view();
        return {v.num_focus_leaves,
                static_cast<const KeyType*>(v.prefixes),
                v.child_offsets,
                v.internal_to_leaf,
                v.level_range,
                static_cast<const KeyType*>(v.focus_leaves),
                v.layout,
                static_cast<const T*>(v.centers),
                static_cast<const T*>(v.sizes)};
    }

private:
    //! sync() on several ranks: the assigned range of the caller's arrays goes in, the domain-owned results are copied back
    template<class... Props>
    void syncMultiRank(DeviceVector<KeyType>& keys, DeviceVector<T>& x, DeviceVector<T>& y, DeviceVector<T>& z,
                       DeviceVector<T>& h, std::tuple<DeviceVector<Props>&...> properties)
    {
        const std::size_t n = x.size();
        // first call: everything passed is assigned; later: the previous layout with its assigned range
        const LocalIndex first = synced_ ? mr_->startIndex() : 0;
        LocalIndex last        = synced_ ? mr_->endIndex() : LocalIndex(n);
        if (haveEndOverride_ && synced_) last = endOverride_; // setEndIndex(): particles appended behind the assigned range
        haveEndOverride_ = false;
        if (last < first || last > n) throw std::runtime_error("Domain sync: end index outside the arrays\n");
        if (synced_ && n != mr_->nParticlesWithHalos())
            throw std::runtime_error("Domain sync: input array sizes are inconsistent\n");
        bool sizesOk = true;
        std::apply([&](auto&... p) { ((sizesOk = sizesOk && p.size() == n), ...); }, properties);
        if (!sizesOk) throw std::runtime_error("Domain sync: input array sizes are inconsistent\n");
        prevSize_ = n, prevStart_ = first, prevEnd_ = last;
        std::apply([&](auto&... p)
                   { mr_->syncKeys(keys.data() + first, x.data() + first, y.data() + first, z.data() + first,
                                   h.data() + first, std::size_t(last - first), (p.data() + first)...); },
                   properties);
        synced_               = true;
        const std::size_t m   = mr_->nParticlesWithHalos();
        auto back = [m](auto& vec, const auto* src)
        {
            vec.resize(m);
            memcpyD2D(src, m, vec.data());
        };
        back(keys, mr_->keys());
        back(x, static_cast<const T*>(mr_->x()));
        back(y, static_cast<const T*>(mr_->y()));
        back(z, static_cast<const T*>(mr_->z()));
        back(h, static_cast<const T*>(mr_->h()));
        int i = 0;
        std::apply([&](auto&... p)
                   { (back(p, mr_->template property<std::decay_t<decltype(*p.data())>>(i++)), ...); },
                   properties);
    }

    cstone_hip_domain* dom_{nullptr};
    std::unique_ptr<MultiRankDomain<KeyType, T>> mr_;
    bool synced_{false};
    bool convergeTrees_{false};
    LocalIndex endOverride_{0};
    bool haveEndOverride_{false};
    std::size_t prevSize_{0};
    LocalIndex prevStart_{0}, prevEnd_{0};
};

} // namespace cstone_amd
