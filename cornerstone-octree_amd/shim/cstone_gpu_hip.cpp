/* cstone_gpu_hip.cpp -- the reference's GPU link seam on top of libcstone_hip.so
 *
 * A maintainer of cornerstone-octree adds THIS file to the reference tree in place of the .cu translation units of the
 * `cstone_gpu` library (R/CMakeLists.txt:9-20, R = <reference>/include/cstone): it defines every `template<...> extern
 * void fooGpu(...)` that the host headers declare (sfc/sfc_gpu.h, primitives/primitives_gpu.h, tree/csarray_gpu.h,
 * tree/octree_gpu.h, traversal/collisions_gpu.h, focus/rebalance_gpu.h, focus/source_center_gpu.h,
 * halos/gather_halos_gpu.h) and the pimpl of cuda/device_vector.h through the C ABI of include/cstone_hip.h, with the
 * explicit instantiations of the reference's .cu files.  It is plain host C++20: no HIP compiler, no Thrust, no CUDA.
 * With it the reference's own `cstone::Domain<KeyType, T, GpuTag>` (R/domain/domain.hpp, compiled with -DUSE_CUDA
 * against the HIP runtime headers, which R/cuda/cuda_runtime.hpp selects by itself) links and runs on MI355X.
 *
 *   g++ -std=c++20 -D__HIP_PLATFORM_AMD__ -I<reference>/include -I/opt/rocm/include -Iinclude \
 *       -c cornerstone-octree_amd/shim/cstone_gpu_hip.cpp        (then link with -lcstone_hip -lamdhip64)
 *
 * All work goes to the device's default stream, like the reference ("Threading").
 */
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "cstone/cuda/device_vector.h"
#include "cstone/focus/rebalance.hpp"
#include "cstone/focus/rebalance_gpu.h"
#include "cstone/focus/source_center_gpu.h"
#include "cstone/halos/gather_halos_gpu.h"
#include "cstone/primitives/primitives_gpu.h"
#include "cstone/sfc/sfc_gpu.h"
#include "cstone/traversal/collisions_gpu.h"
#include "cstone/traversal/groups_gpu.h"
#include "cstone/tree/csarray_gpu.h"
#include "cstone/tree/octree_gpu.h"

#include "cstone_hip.h"

namespace cstone
{

namespace
{

//! one context per host thread on the default stream
cstone_hip_ctx* hipCtx()
{
    thread_local cstone_hip_ctx* ctx = []
    {
        cstone_hip_ctx* p = nullptr;
        int device        = 0;
        (void)hipGetDevice(&device);
        if (cstone_hip_ctx_create(&p, device, nullptr, 0) != CSTONE_OK) throw std::runtime_error("cstone_hip_ctx_create failed");
        return p;
    }();
    return ctx;
}

void check(int rc)
{
    if (rc != CSTONE_OK) throw std::runtime_error(std::string("libcstone_hip: ") + cstone_hip_last_error(hipCtx()));
}

template<class T>
cstone_box podBox(const Box<T>& b)
{
    return {{double(b.xmin()), double(b.xmax()), double(b.ymin()), double(b.ymax()), double(b.zmin()), double(b.zmax())},
            {int(b.boundaryX()), int(b.boundaryY()), int(b.boundaryZ())},
            0};
}

template<class KeyType>
struct CurveOf
{
    static constexpr int value = CSTONE_HILBERT; // plain integers: SfcKind = HilbertKey, R/sfc/sfc.hpp:53-55
    using Integer              = KeyType;
};
template<class I>
struct CurveOf<MortonKey<I>>
{
    static constexpr int value = CSTONE_MORTON;
    using Integer              = I;
};
template<class I>
struct CurveOf<HilbertKey<I>>
{
    static constexpr int value = CSTONE_HILBERT;
    using Integer              = I;
};

template<class T>
constexpr int bitsOf = 8 * int(sizeof(T));

} // namespace

// ------------------------------------------------------------------------------------------------------------------
// cuda/device_vector.h
// ------------------------------------------------------------------------------------------------------------------
template<class T>
class DeviceVector<T>::Impl
{
public:
    Impl() = default;
    Impl(const Impl& other) { *this = other; }
    ~Impl() { release(); }

    T* data() { return data_; }
    const T* data() const { return data_; }
    std::size_t size() const { return size_; }
    std::size_t capacity() const { return capacity_; }

    void reserve(std::size_t n)
    {
        if (n <= capacity_) return;
        T* fresh = nullptr;
        check(cstone_hip_malloc(hipCtx(), (void**)&fresh, n * sizeof(T)));
        if (size_) check(cstone_hip_memcpy_d2d(hipCtx(), fresh, data_, size_ * sizeof(T)));
        release();
        data_     = fresh;
        capacity_ = n;
    }

    //! contents up to min(old, new) size are kept, new elements are uninitialised (util::uninitialized_allocator)
    void resize(std::size_t n)
    {
        if (n > capacity_)
        {
            std::size_t keep = size_;
            reserve(n);
            size_ = keep;
        }
        size_ = n;
    }

    Impl& operator=(const std::vector<T>& rhs)
    {
        size_ = 0; // nothing to carry over
        resize(rhs.size());
        if (!rhs.empty()) check(cstone_hip_memcpy_h2d(hipCtx(), data_, rhs.data(), rhs.size() * sizeof(T)));
        return *this;
    }

    Impl& operator=(const Impl& rhs)
    {
        if (this == &rhs) return *this;
        size_ = 0;
        resize(rhs.size_);
        if (rhs.size_) check(cstone_hip_memcpy_d2d(hipCtx(), data_, rhs.data_, rhs.size_ * sizeof(T)));
        return *this;
    }

    friend bool operator==(const Impl& lhs, const Impl& rhs)
    {
        if (lhs.size_ != rhs.size_) return false;
        std::vector<char> a(lhs.size_ * sizeof(T)), b(rhs.size_ * sizeof(T));
        if (lhs.size_)
        {
            check(cstone_hip_memcpy_d2h(hipCtx(), a.data(), lhs.data_, a.size()));
            check(cstone_hip_memcpy_d2h(hipCtx(), b.data(), rhs.data_, b.size()));
        }
        return a == b;
    }

private:
    void release()
    {
        if (data_) (void)cstone_hip_free(hipCtx(), data_);
        data_ = nullptr, capacity_ = 0;
    }
    T* data_              = nullptr;
    std::size_t size_     = 0;
    std::size_t capacity_ = 0;
};

template<class T>
DeviceVector<T>::DeviceVector()
    : impl_(new Impl())
{
}
template<class T>
DeviceVector<T>::DeviceVector(std::size_t size)
    : impl_(new Impl())
{
    impl_->resize(size);
}
template<class T>
DeviceVector<T>::DeviceVector(std::size_t size, T init)
    : impl_(new Impl())
{
    *impl_ = std::vector<T>(size, init);
}
template<class T>
DeviceVector<T>::DeviceVector(const DeviceVector<T>& other)
    : impl_(new Impl(*other.impl_))
{
}
template<class T>
DeviceVector<T>::DeviceVector(const std::vector<T>& rhs)
    : impl_(new Impl())
{
    *impl_ = rhs;
}
template<class T>
DeviceVector<T>::DeviceVector(const T* first, const T* last)
    : impl_(new Impl())
{
    *impl_ = std::vector<T>(first, last);
}
template<class T>
DeviceVector<T>::~DeviceVector() = default;
template<class T>
T* DeviceVector<T>::data()
{
    return impl_->data();
}
template<class T>
const T* DeviceVector<T>::data() const
{
    return impl_->data();
}
template<class T>
void DeviceVector<T>::resize(std::size_t size)
{
    impl_->resize(size);
}
template<class T>
void DeviceVector<T>::reserve(std::size_t size)
{
    impl_->reserve(size);
}
template<class T>
std::size_t DeviceVector<T>::size() const
{
    return impl_->size();
}
template<class T>
bool DeviceVector<T>::empty() const
{
    return impl_->size() == 0;
}
template<class T>
std::size_t DeviceVector<T>::capacity() const
{
    return impl_->capacity();
}
template<class T>
DeviceVector<T>& DeviceVector<T>::swap(DeviceVector<T>& rhs)
{
    std::swap(impl_, rhs.impl_);
    return *this;
}
template<class T>
DeviceVector<T>& DeviceVector<T>::operator=(DeviceVector<T> rhs)
{
    this->swap(rhs);
    return *this;
}
template<class T>
DeviceVector<T>& DeviceVector<T>::operator=(const std::vector<T>& rhs)
{
    *impl_ = rhs;
    return *this;
}
template<class T>
bool operator==(const DeviceVector<T>& lhs, const DeviceVector<T>& rhs)
{
    return *lhs.impl_ == *rhs.impl_;
}

// the list of R/cuda/device_vector.cu:162-185
#define DEVICE_VECTOR(T)                                                                                               \
    template class DeviceVector<T>;                                                                                    \
    template bool operator==(const DeviceVector<T>&, const DeviceVector<T>&);
DEVICE_VECTOR(char);
DEVICE_VECTOR(uint8_t);
DEVICE_VECTOR(int);
DEVICE_VECTOR(unsigned);
DEVICE_VECTOR(uint64_t);
DEVICE_VECTOR(float);
DEVICE_VECTOR(double);
#undef DEVICE_VECTOR
template class DeviceVector<util::array<int, 2>>;
template class DeviceVector<util::array<int, 3>>;
template class DeviceVector<util::array<unsigned, 1>>;
template class DeviceVector<util::array<uint64_t, 1>>;
template class DeviceVector<util::array<uint64_t, 2>>;
template class DeviceVector<util::array<unsigned, 2>>;
template class DeviceVector<util::array<float, 3>>;
template class DeviceVector<util::array<float, 4>>;
template class DeviceVector<util::array<float, 8>>;
template class DeviceVector<util::array<float, 12>>;
template class DeviceVector<util::array<double, 3>>;
template class DeviceVector<util::array<double, 4>>;
template class DeviceVector<util::array<double, 8>>;
template class DeviceVector<util::array<double, 12>>;

// ------------------------------------------------------------------------------------------------------------------
// sfc/sfc_gpu.h
// ------------------------------------------------------------------------------------------------------------------
template<class KeyType, class T>
void computeSfcKeysGpu(const T* x, const T* y, const T* z, KeyType* keys, size_t numKeys, const Box<T>& box)
{
    using I      = typename CurveOf<KeyType>::Integer;
    cstone_box b = podBox(box);
    check(cstone_hip_compute_sfc_keys(hipCtx(), CurveOf<KeyType>::value, bitsOf<I>, bitsOf<T>, x, y, z, keys, numKeys, &b));
}
#define SFC_KEYS_GPU(Key, T) template void computeSfcKeysGpu(const T*, const T*, const T*, Key*, size_t, const Box<T>&)
SFC_KEYS_GPU(MortonKey<unsigned>, float);
SFC_KEYS_GPU(MortonKey<unsigned>, double);
SFC_KEYS_GPU(MortonKey<uint64_t>, float);
SFC_KEYS_GPU(MortonKey<uint64_t>, double);
SFC_KEYS_GPU(HilbertKey<unsigned>, float);
SFC_KEYS_GPU(HilbertKey<unsigned>, double);
SFC_KEYS_GPU(HilbertKey<uint64_t>, float);
SFC_KEYS_GPU(HilbertKey<uint64_t>, double);
#undef SFC_KEYS_GPU

// ------------------------------------------------------------------------------------------------------------------
// primitives/primitives_gpu.h
// ------------------------------------------------------------------------------------------------------------------
template<class T>
void fillGpu(T* first, T* last, T value)
{
    check(cstone_hip_fill(hipCtx(), int(sizeof(T)), first, size_t(last - first), &value));
}
template void fillGpu(double*, double*, double);
template void fillGpu(float*, float*, float);
template void fillGpu(int*, int*, int);
template void fillGpu(char*, char*, char);
template void fillGpu(unsigned*, unsigned*, unsigned);
template void fillGpu(uint64_t*, uint64_t*, uint64_t);

template<class T>
void scaleGpu(T* first, T* last, T value)
{
    check(cstone_hip_scale(hipCtx(), bitsOf<T>, first, size_t(last - first), double(value)));
}
template void scaleGpu(double*, double*, double);
template void scaleGpu(float*, float*, float);

template<class T>
void incrementGpu(const T* first, const T* last, T* d_first, T value)
{
    check(cstone_hip_increment(hipCtx(), bitsOf<T>, first, d_first, size_t(last - first), uint64_t(value)));
}
template void incrementGpu(const unsigned*, const unsigned*, unsigned*, unsigned);
template void incrementGpu(const uint64_t*, const uint64_t*, uint64_t*, uint64_t);

template<class T, class IndexType>
void gatherGpu(const IndexType* ordering, size_t numElements, const T* src, T* buffer)
{
    static_assert(sizeof(IndexType) == 4);
    check(cstone_hip_gather(hipCtx(), int(sizeof(T)), reinterpret_cast<const uint32_t*>(ordering), numElements, src, buffer));
}
#define GATHER_GPU(I, T) template void gatherGpu(const I*, size_t, const T*, T*)
GATHER_GPU(int, int);
GATHER_GPU(int, uint32_t);
GATHER_GPU(int, uint64_t);
GATHER_GPU(unsigned, uint8_t);
GATHER_GPU(unsigned, double);
GATHER_GPU(unsigned, float);
GATHER_GPU(unsigned, char);
GATHER_GPU(unsigned, int);
GATHER_GPU(unsigned, long);
GATHER_GPU(unsigned, unsigned);
GATHER_GPU(unsigned, unsigned long);
GATHER_GPU(unsigned, unsigned long long);
using ArrayF1 = util::array<float, 1>;
using ArrayF2 = util::array<float, 2>;
using ArrayF3 = util::array<float, 3>;
using ArrayF4 = util::array<float, 4>;
using ArrayD4 = util::array<double, 4>;
GATHER_GPU(int, ArrayF4);
GATHER_GPU(int, ArrayD4);
GATHER_GPU(unsigned, ArrayF1);
GATHER_GPU(unsigned, ArrayF2);
GATHER_GPU(unsigned, ArrayF3);
GATHER_GPU(unsigned, ArrayF4);
#undef GATHER_GPU

template<class T, class IndexType>
void scatterGpu(const IndexType* ordering, size_t numElements, const T* src, T* buffer)
{
    static_assert(sizeof(IndexType) == 4);
    check(cstone_hip_scatter(hipCtx(), int(sizeof(T)), reinterpret_cast<const uint32_t*>(ordering), numElements, src, buffer));
}
template void scatterGpu(const int*, size_t, const int*, int*);
template void scatterGpu(const int*, size_t, const uint32_t*, uint32_t*);
template void scatterGpu(const int*, size_t, const uint64_t*, uint64_t*);
template void scatterGpu(const int*, size_t, const ArrayF4*, ArrayF4*);
template void scatterGpu(const int*, size_t, const ArrayD4*, ArrayD4*);

template<class T>
std::tuple<T, T> MinMaxGpu<T>::operator()(const T* first, const T* last)
{
    double out[2];
    check(cstone_hip_minmax(hipCtx(), bitsOf<T>, first, size_t(last - first), out));
    return {T(out[0]), T(out[1])};
}
template struct MinMaxGpu<double>;
template struct MinMaxGpu<float>;

template<class T>
T maxNormSquareGpu(const T* x, const T* y, const T* z, size_t numElements)
{
    double out = 0;
    check(cstone_hip_max_norm_square(hipCtx(), bitsOf<T>, x, y, z, numElements, &out));
    return T(out);
}
template float maxNormSquareGpu(const float*, const float*, const float*, size_t);
template double maxNormSquareGpu(const double*, const double*, const double*, size_t);

namespace
{
template<class T>
constexpr int scalarKind()
{
    if constexpr (std::is_same_v<T, float>) return 4;
    else if constexpr (std::is_signed_v<T>) return sizeof(T) == 4 ? 2 : 3;
    else return sizeof(T) == 4 ? 0 : 1;
}
} // namespace

template<class T>
size_t lowerBoundGpu(const T* first, const T* last, T value)
{
    uint64_t idx = 0;
    check(cstone_hip_lower_bound_value(hipCtx(), scalarKind<T>(), first, size_t(last - first), &value, &idx));
    return size_t(idx);
}
template size_t lowerBoundGpu(const unsigned*, const unsigned*, unsigned);
template size_t lowerBoundGpu(const uint64_t*, const uint64_t*, uint64_t);
template size_t lowerBoundGpu(const int*, const int*, int);
template size_t lowerBoundGpu(const int64_t*, const int64_t*, int64_t);
template size_t lowerBoundGpu(const float*, const float*, float);

template<class T, class IndexType>
void lowerBoundGpu(const T* first, const T* last, const T* valueFirst, const T* valueLast, IndexType* result)
{
    int numValues = int(valueLast - valueFirst);
    if (numValues == 0) return;
    if constexpr (sizeof(IndexType) == 8)
    {
        check(cstone_hip_lower_bound(hipCtx(), bitsOf<T>, first, size_t(last - first), valueFirst, numValues,
                                     reinterpret_cast<uint64_t*>(result)));
    }
    else
    {
        check(cstone_hip_lower_bound_u32(hipCtx(), bitsOf<T>, first, size_t(last - first), valueFirst, numValues,
                                         reinterpret_cast<uint32_t*>(result)));
    }
}
template void lowerBoundGpu(const unsigned*, const unsigned*, const unsigned*, const unsigned*, unsigned*);
template void lowerBoundGpu(const uint64_t*, const uint64_t*, const uint64_t*, const uint64_t*, unsigned*);
template void lowerBoundGpu(const unsigned*, const unsigned*, const unsigned*, const unsigned*, uint64_t*);
template void lowerBoundGpu(const uint64_t*, const uint64_t*, const uint64_t*, const uint64_t*, uint64_t*);

template<class Tin, class Tout, class IndexType>
void segmentMax(const Tin* input, const IndexType* segments, size_t numSegments, Tout* output)
{
    check(cstone_hip_segment_max(hipCtx(), bitsOf<Tin>, bitsOf<Tout>, bitsOf<IndexType>, input, segments, numSegments, output));
}
template void segmentMax(const float*, const unsigned*, size_t, float*);
template void segmentMax(const double*, const unsigned*, size_t, float*);
template void segmentMax(const double*, const unsigned*, size_t, double*);
template void segmentMax(const float*, const uint64_t*, size_t, float*);
template void segmentMax(const double*, const uint64_t*, size_t, float*);
template void segmentMax(const double*, const uint64_t*, size_t, double*);

template<class Tin, class Tout>
Tout reduceGpu(const Tin* input, size_t numElements, Tout init)
{
    uint64_t sum = 0;
    check(cstone_hip_reduce_sum(hipCtx(), bitsOf<Tin>, input, numElements, uint64_t(init), &sum));
    return Tout(sum);
}
template size_t reduceGpu(const unsigned*, size_t, size_t);

template<class IndexType>
void sequenceGpu(IndexType* input, size_t numElements, IndexType init)
{
    if constexpr (sizeof(IndexType) == 4)
    {
        check(cstone_hip_sequence_u32(hipCtx(), reinterpret_cast<uint32_t*>(input), numElements, uint32_t(init)));
    }
    else
    {
        check(cstone_hip_sequence_u64(hipCtx(), reinterpret_cast<uint64_t*>(input), numElements, uint64_t(init)));
    }
}
template void sequenceGpu(int*, size_t, int);
template void sequenceGpu(unsigned*, size_t, unsigned);
template void sequenceGpu(uint64_t*, uint64_t, uint64_t);

template<class KeyType>
void sortGpu(KeyType* first, KeyType* last, KeyType* /*keyBuf*/)
{
    check(cstone_hip_sort_keys(hipCtx(), bitsOf<KeyType>, first, size_t(last - first)));
}
template void sortGpu(uint32_t*, uint32_t*, uint32_t*);
template void sortGpu(uint64_t*, uint64_t*, uint64_t*);

template<class KeyType, class ValueType>
uint64_t sortByKeyTempStorage(uint64_t numElements)
{
    return cstone_hip_sort_pairs_temp_bytes(bitsOf<KeyType>, numElements);
}

template<class KeyType, class ValueType>
void sortByKeyGpu(
    KeyType* first, KeyType* last, ValueType* values, KeyType* keyBuf, ValueType* valueBuf, void* tmp, uint64_t tmpBytes)
{
    static_assert(sizeof(ValueType) == 4, "the payload of the pair sort is a 32-bit index");
    check(cstone_hip_sort_pairs(hipCtx(), bitsOf<KeyType>, first, reinterpret_cast<uint32_t*>(values), size_t(last - first),
                                keyBuf, reinterpret_cast<uint32_t*>(valueBuf), tmp, tmpBytes));
}

template<class KeyType, class ValueType>
void sortByKeyGpu(KeyType* first, KeyType* last, ValueType* values)
{
    static_assert(sizeof(ValueType) == 4, "the payload of the pair sort is a 32-bit index");
    check(cstone_hip_sort_pairs(hipCtx(), bitsOf<KeyType>, first, reinterpret_cast<uint32_t*>(values), size_t(last - first),
                                nullptr, nullptr, nullptr, 0));
}
#define SORT_BY_KEY_GPU(KeyType, ValueType)                                                                            \
    template uint64_t sortByKeyTempStorage<KeyType, ValueType>(uint64_t);                                              \
    template void sortByKeyGpu(KeyType*, KeyType*, ValueType*, KeyType*, ValueType*, void*, uint64_t);                 \
    template void sortByKeyGpu(KeyType*, KeyType*, ValueType*)
SORT_BY_KEY_GPU(unsigned, unsigned);
SORT_BY_KEY_GPU(unsigned, int);
SORT_BY_KEY_GPU(uint64_t, unsigned);
SORT_BY_KEY_GPU(uint64_t, int);
#undef SORT_BY_KEY_GPU

namespace
{
//! scans over 32-bit integers with 32- or 64-bit sums, all on the device
template<class IndexType, class SumType>
void scanGpu(const IndexType* first, const IndexType* last, SumType* output, SumType init, bool inclusive)
{
    size_t n = size_t(last - first);
    if (n == 0) return;
    if constexpr (sizeof(IndexType) == 4 && sizeof(SumType) == 4)
    {
        auto* in  = reinterpret_cast<const uint32_t*>(first);
        auto* out = reinterpret_cast<uint32_t*>(output);
        check(inclusive ? cstone_hip_inclusive_scan_u32(hipCtx(), in, out, n)
                        : cstone_hip_exclusive_scan_u32(hipCtx(), in, out, n, uint32_t(init)));
    }
    else
    {
        static_assert(sizeof(IndexType) == 4 && sizeof(SumType) == 8, "scan: 32-bit values, 32- or 64-bit sums");
        check(cstone_hip_scan_u32_to_u64(hipCtx(), reinterpret_cast<const uint32_t*>(first),
                                         reinterpret_cast<uint64_t*>(output), n, uint64_t(init), inclusive ? 1 : 0));
    }
}
} // namespace

template<class IndexType, class SumType>
void exclusiveScanGpu(const IndexType* first, const IndexType* last, SumType* output, SumType init)
{
    scanGpu(first, last, output, init, false);
}
template void exclusiveScanGpu(const int*, const int*, int*, int);
template void exclusiveScanGpu(const int*, const int*, unsigned*, unsigned);
template void exclusiveScanGpu(const int*, const int*, uint64_t*, uint64_t);
template void exclusiveScanGpu(const unsigned*, const unsigned*, unsigned*, unsigned);
template void exclusiveScanGpu(const unsigned*, const unsigned*, uint64_t*, uint64_t);

template<class IndexType, class SumType>
void inclusiveScanGpu(const IndexType* first, const IndexType* last, SumType* output)
{
    scanGpu(first, last, output, SumType(0), true);
}
template void inclusiveScanGpu(const int*, const int*, int*);
template void inclusiveScanGpu(const int*, const int*, unsigned*);
template void inclusiveScanGpu(const unsigned*, const unsigned*, unsigned*);

template<class ValueType>
size_t countGpu(const ValueType* first, const ValueType* last, ValueType v)
{
    uint64_t count = 0;
    uint64_t bits  = 0;
    static_assert(sizeof(ValueType) <= 8);
    std::memcpy(&bits, &v, sizeof(ValueType));
    check(cstone_hip_count_equal(hipCtx(), bitsOf<ValueType>, first, size_t(last - first), bits, &count));
    return size_t(count);
}
template size_t countGpu(const int* first, const int* last, int v);
template size_t countGpu(const unsigned* first, const unsigned* last, unsigned v);
template size_t countGpu(const uint64_t* first, const uint64_t* last, uint64_t v);

// ------------------------------------------------------------------------------------------------------------------
// halos/gather_halos_gpu.h
// ------------------------------------------------------------------------------------------------------------------
template<class T, class IndexType>
void gatherRanges(
    const IndexType* rangeScan, const IndexType* rangeOffsets, int numRanges, const T* src, T* buffer, size_t bufferSize)
{
    check(cstone_hip_gather_ranges(hipCtx(), int(sizeof(T)), bitsOf<IndexType>, rangeScan, rangeOffsets, numRanges, src,
                                   buffer, bufferSize));
}
#define GATHER_RANGES(I, T) template void gatherRanges(const I*, const I*, int, const T*, T*, size_t)
GATHER_RANGES(unsigned, int);
GATHER_RANGES(uint64_t, int);
GATHER_RANGES(unsigned, ArrayF1);
GATHER_RANGES(unsigned, ArrayF2);
GATHER_RANGES(unsigned, ArrayF3);
GATHER_RANGES(unsigned, ArrayF4);
GATHER_RANGES(uint64_t, ArrayF1);
GATHER_RANGES(uint64_t, ArrayF2);
GATHER_RANGES(uint64_t, ArrayF3);
GATHER_RANGES(uint64_t, ArrayF4);
#undef GATHER_RANGES

// ------------------------------------------------------------------------------------------------------------------
// tree/csarray_gpu.h, tree/octree_gpu.h
// ------------------------------------------------------------------------------------------------------------------
template<class KeyType>
void computeNodeCountsGpu(const KeyType* tree,
                          unsigned* counts,
                          TreeNodeIndex numNodes,
                          const KeyType* firstKey,
                          const KeyType* lastKey,
                          unsigned maxCount,
                          bool useCountsAsGuess)
{
    size_t n = size_t(lastKey - firstKey);
    if (useCountsAsGuess && numNodes > 0)
    {
        // the reference seeds each search with the exclusive scan of the previous counts (R/tree/csarray_gpu.cu:117-128)
        uint32_t* guess = nullptr;
        check(cstone_hip_malloc(hipCtx(), (void**)&guess, size_t(numNodes + 1) * sizeof(uint32_t)));
        check(cstone_hip_memset(hipCtx(), guess + numNodes, 0, sizeof(uint32_t)));
        check(cstone_hip_exclusive_scan_u32(hipCtx(), counts, guess, size_t(numNodes) + 1, 0u));
        int rc = cstone_hip_compute_node_counts_guided(hipCtx(), bitsOf<KeyType>, tree, counts, numNodes, firstKey, n,
                                                       maxCount, guess);
        check(cstone_hip_ctx_sync(hipCtx()));
        check(cstone_hip_free(hipCtx(), guess));
        check(rc);
    }
    else { check(cstone_hip_compute_node_counts(hipCtx(), bitsOf<KeyType>, tree, counts, numNodes, firstKey, n, maxCount)); }
}
template void computeNodeCountsGpu(const unsigned*, unsigned*, TreeNodeIndex, const unsigned*, const unsigned*, unsigned, bool);
template void computeNodeCountsGpu(const uint64_t*, unsigned*, TreeNodeIndex, const uint64_t*, const uint64_t*, unsigned, bool);

template<class KeyType>
TreeNodeIndex
computeNodeOpsGpu(const KeyType* tree, TreeNodeIndex numNodes, const unsigned* counts, unsigned bucketSize, TreeNodeIndex* nodeOps)
{
    int newNumNodes = 0, converged = 0;
    check(cstone_hip_compute_node_ops(hipCtx(), bitsOf<KeyType>, tree, numNodes, counts, bucketSize, nodeOps, &newNumNodes,
                                      &converged));
    return newNumNodes;
}
template TreeNodeIndex computeNodeOpsGpu(const unsigned*, TreeNodeIndex, const unsigned*, unsigned, TreeNodeIndex*);
template TreeNodeIndex computeNodeOpsGpu(const uint64_t*, TreeNodeIndex, const unsigned*, unsigned, TreeNodeIndex*);

namespace
{
//! rebalanceTreeGpu reports "nothing changed" = every op is 1 (R/tree/csarray_gpu.cu:133-158,211-224): with scanned ops
//! that is ops[i] == i for all i, i.e. as many new nodes as old ones and the last offset equal to the node count
template<class KeyType>
bool treeUnchanged(const KeyType* tree, TreeNodeIndex numNodes, TreeNodeIndex newNumNodes, const KeyType* newTree)
{
    if (numNodes != newNumNodes) return false;
    std::vector<KeyType> a(size_t(numNodes) + 1), b(size_t(numNodes) + 1);
    check(cstone_hip_memcpy_d2h(hipCtx(), a.data(), tree, a.size() * sizeof(KeyType)));
    check(cstone_hip_memcpy_d2h(hipCtx(), b.data(), newTree, b.size() * sizeof(KeyType)));
    return a == b;
}
} // namespace

template<class KeyType>
bool rebalanceTreeGpu(
    const KeyType* tree, TreeNodeIndex numNodes, TreeNodeIndex newNumNodes, const TreeNodeIndex* nodeOps, KeyType* newTree)
{
    check(cstone_hip_rebalance_tree(hipCtx(), bitsOf<KeyType>, tree, numNodes, newNumNodes, nodeOps, newTree));
    return treeUnchanged(tree, numNodes, newNumNodes, newTree);
}
template bool rebalanceTreeGpu(const unsigned*, TreeNodeIndex, TreeNodeIndex, const TreeNodeIndex*, unsigned*);
template bool rebalanceTreeGpu(const uint64_t*, TreeNodeIndex, TreeNodeIndex, const TreeNodeIndex*, uint64_t*);

template<class KeyType>
void countSfcGapsGpu(const KeyType* tree, TreeNodeIndex numNodes, TreeNodeIndex* nodeOps)
{
    check(cstone_hip_count_sfc_gaps(hipCtx(), bitsOf<KeyType>, tree, numNodes, nodeOps));
}
template void countSfcGapsGpu(const uint32_t*, TreeNodeIndex, TreeNodeIndex*);
template void countSfcGapsGpu(const uint64_t*, TreeNodeIndex, TreeNodeIndex*);

template<class KeyType>
void fillSfcGapsGpu(const KeyType* tree, TreeNodeIndex numNodes, const TreeNodeIndex* nodeOps, KeyType* newTree)
{
    check(cstone_hip_fill_sfc_gaps(hipCtx(), bitsOf<KeyType>, tree, numNodes, nodeOps, newTree));
}
template void fillSfcGapsGpu(const uint32_t*, TreeNodeIndex, const TreeNodeIndex*, uint32_t*);
template void fillSfcGapsGpu(const uint64_t*, TreeNodeIndex, const TreeNodeIndex*, uint64_t*);

template<class KeyType>
void buildOctreeGpu(const KeyType* cstoneTree, OctreeView<KeyType> d)
{
    check(cstone_hip_build_octree(hipCtx(), bitsOf<KeyType>, cstoneTree, d.numLeafNodes, d.prefixes, d.childOffsets, d.parents,
                                  d.levelRange, d.internalToLeaf, d.leafToInternal));
}
template void buildOctreeGpu(const uint32_t*, OctreeView<uint32_t>);
template void buildOctreeGpu(const uint64_t*, OctreeView<uint64_t>);

//! levelRange is a HOST pointer here, as in the reference (R/tree/octree_gpu.cu:196-207)
void upsweepSumGpu(int numLvl, const TreeNodeIndex* lvlRange, const TreeNodeIndex* childOffsets, LocalIndex* counts)
{
    TreeNodeIndex* devRange = nullptr;
    check(cstone_hip_malloc(hipCtx(), (void**)&devRange, size_t(numLvl + 1) * sizeof(TreeNodeIndex)));
    check(cstone_hip_memcpy_h2d(hipCtx(), devRange, lvlRange, size_t(numLvl + 1) * sizeof(TreeNodeIndex)));
    // levels numLvl - 1 ... 0, like the reference's loop
    int rc = cstone_hip_upsweep_sum(hipCtx(), numLvl + 1, devRange, childOffsets, counts);
    check(cstone_hip_ctx_sync(hipCtx()));
    check(cstone_hip_free(hipCtx(), devRange));
    check(rc);
}

// ------------------------------------------------------------------------------------------------------------------
// traversal/collisions_gpu.h
// ------------------------------------------------------------------------------------------------------------------
template<class KeyType, class RadiusType, class T>
void findHalosGpu(const KeyType* prefixes,
                  const TreeNodeIndex* childOffsets,
                  const TreeNodeIndex* internalToLeaf,
                  const KeyType* leaves,
                  const RadiusType* interactionRadii,
                  const Box<T>& box,
                  TreeNodeIndex firstNode,
                  TreeNodeIndex lastNode,
                  int* collisionFlags)
{
    static_assert(std::is_same_v<RadiusType, float>);
    cstone_box b = podBox(box);
    check(cstone_hip_find_halos(hipCtx(), CSTONE_HILBERT, bitsOf<KeyType>, bitsOf<T>, prefixes, childOffsets, internalToLeaf,
                                leaves, interactionRadii, &b, firstNode, lastNode, collisionFlags));
}
#define FIND_HALOS_GPU(KeyType, RadiusType, T)                                                                         \
    template void findHalosGpu(const KeyType*, const TreeNodeIndex*, const TreeNodeIndex*, const KeyType*,             \
                               const RadiusType*, const Box<T>&, TreeNodeIndex, TreeNodeIndex, int*)
FIND_HALOS_GPU(uint32_t, float, float);
FIND_HALOS_GPU(uint32_t, float, double);
FIND_HALOS_GPU(uint64_t, float, float);
FIND_HALOS_GPU(uint64_t, float, double);
#undef FIND_HALOS_GPU

template<class T, class KeyType>
void markMacsGpu(const KeyType* prefixes,
                 const TreeNodeIndex* childOffsets,
                 const Vec4<T>* centers,
                 const Box<T>& box,
                 const KeyType* focusNodes,
                 TreeNodeIndex numFocusNodes,
                 bool limitSource,
                 char* markings)
{
    cstone_box b = podBox(box);
    check(cstone_hip_mark_macs(hipCtx(), CSTONE_HILBERT, bitsOf<KeyType>, bitsOf<T>, prefixes, childOffsets, centers, &b,
                               focusNodes, numFocusNodes, limitSource, markings));
}
#define MARK_MACS_GPU(T, KeyType)                                                                                      \
    template void markMacsGpu(const KeyType*, const TreeNodeIndex*, const Vec4<T>*, const Box<T>&, const KeyType*,     \
                              TreeNodeIndex, bool, char*)
MARK_MACS_GPU(double, uint64_t);
MARK_MACS_GPU(float, uint64_t);
MARK_MACS_GPU(double, unsigned);
MARK_MACS_GPU(float, unsigned);
#undef MARK_MACS_GPU

// ------------------------------------------------------------------------------------------------------------------
// focus/rebalance_gpu.h
// ------------------------------------------------------------------------------------------------------------------
template<class KeyType>
void rebalanceDecisionEssentialGpu(const KeyType* prefixes,
                                   const TreeNodeIndex* childOffsets,
                                   const TreeNodeIndex* parents,
                                   const unsigned* counts,
                                   const char* macs,
                                   KeyType focusStart,
                                   KeyType focusEnd,
                                   unsigned bucketSize,
                                   TreeNodeIndex* nodeOps,
                                   TreeNodeIndex numNodes)
{
    check(cstone_hip_rebalance_decision_essential(hipCtx(), bitsOf<KeyType>, prefixes, childOffsets, parents, counts, macs,
                                                  focusStart, focusEnd, bucketSize, nodeOps, numNodes));
}
template void rebalanceDecisionEssentialGpu(const uint32_t*, const TreeNodeIndex*, const TreeNodeIndex*, const unsigned*,
                                            const char*, uint32_t, uint32_t, unsigned, TreeNodeIndex*, TreeNodeIndex);
template void rebalanceDecisionEssentialGpu(const uint64_t*, const TreeNodeIndex*, const TreeNodeIndex*, const unsigned*,
                                            const char*, uint64_t, uint64_t, unsigned, TreeNodeIndex*, TreeNodeIndex);

template<class KeyType>
void macRefineDecisionGpu(const KeyType* prefixes,
                          const char* macs,
                          const TreeNodeIndex* l2i,
                          TreeNodeIndex numLeafNodes,
                          TreeIndexPair focus,
                          TreeNodeIndex* nodeOps)
{
    check(cstone_hip_mac_refine_decision(hipCtx(), bitsOf<KeyType>, prefixes, macs, l2i, numLeafNodes, focus.start(),
                                         focus.end(), nodeOps));
}
template void macRefineDecisionGpu(const uint32_t*, const char*, const TreeNodeIndex*, TreeNodeIndex, TreeIndexPair, TreeNodeIndex*);
template void macRefineDecisionGpu(const uint64_t*, const char*, const TreeNodeIndex*, TreeNodeIndex, TreeIndexPair, TreeNodeIndex*);

template<class KeyType>
bool protectAncestorsGpu(const KeyType* prefixes, const TreeNodeIndex* parents, TreeNodeIndex* nodeOps, TreeNodeIndex numNodes)
{
    int converged = 1;
    check(cstone_hip_protect_ancestors(hipCtx(), bitsOf<KeyType>, prefixes, parents, nodeOps, numNodes, &converged));
    return converged != 0;
}
template bool protectAncestorsGpu(const uint32_t*, const TreeNodeIndex*, TreeNodeIndex*, TreeNodeIndex);
template bool protectAncestorsGpu(const uint64_t*, const TreeNodeIndex*, TreeNodeIndex*, TreeNodeIndex);

template<class KeyType>
ResolutionStatus enforceKeysGpu(const KeyType* forcedKeys,
                                TreeNodeIndex numForcedKeys,
                                const KeyType* nodeKeys,
                                const TreeNodeIndex* childOffsets,
                                const TreeNodeIndex* parents,
                                TreeNodeIndex* nodeOps)
{
    int status = 0;
    check(cstone_hip_enforce_keys(hipCtx(), bitsOf<KeyType>, forcedKeys, numForcedKeys, nodeKeys, childOffsets, parents,
                                  nodeOps, &status));
    return static_cast<ResolutionStatus>(status);
}
template ResolutionStatus
enforceKeysGpu(const uint32_t*, TreeNodeIndex, const uint32_t*, const TreeNodeIndex*, const TreeNodeIndex*, TreeNodeIndex*);
template ResolutionStatus
enforceKeysGpu(const uint64_t*, TreeNodeIndex, const uint64_t*, const TreeNodeIndex*, const TreeNodeIndex*, TreeNodeIndex*);

template<class KeyType>
void rangeCountGpu(gsl::span<const KeyType> leaves,
                   gsl::span<const unsigned> counts,
                   gsl::span<const KeyType> leavesFocus,
                   gsl::span<const TreeNodeIndex> leavesFocusIdx,
                   gsl::span<unsigned> countsFocus)
{
    check(cstone_hip_range_count(hipCtx(), bitsOf<KeyType>, leaves.data(), int(leaves.size()) - 1, counts.data(),
                                 leavesFocus.data(), leavesFocusIdx.data(), int(leavesFocusIdx.size()), countsFocus.data()));
}
template void rangeCountGpu(gsl::span<const uint32_t>, gsl::span<const unsigned>, gsl::span<const uint32_t>,
                            gsl::span<const TreeNodeIndex>, gsl::span<unsigned>);
template void rangeCountGpu(gsl::span<const uint64_t>, gsl::span<const unsigned>, gsl::span<const uint64_t>,
                            gsl::span<const TreeNodeIndex>, gsl::span<unsigned>);

// ------------------------------------------------------------------------------------------------------------------
// focus/source_center_gpu.h
// ------------------------------------------------------------------------------------------------------------------
template<class Tc, class Tm, class Tf>
void computeLeafSourceCenterGpu(const Tc* x,
                                const Tc* y,
                                const Tc* z,
                                const Tm* m,
                                const TreeNodeIndex* leafToInternal,
                                TreeNodeIndex numLeaves,
                                const LocalIndex* layout,
                                Vec4<Tf>* centers)
{
    check(cstone_hip_leaf_source_centers(hipCtx(), bitsOf<Tc>, bitsOf<Tm>, bitsOf<Tf>, x, y, z, m, leafToInternal, numLeaves,
                                         layout, centers));
}
template void computeLeafSourceCenterGpu(const double*, const double*, const double*, const double*, const TreeNodeIndex*,
                                         TreeNodeIndex, const LocalIndex*, Vec4<double>*);
template void computeLeafSourceCenterGpu(const double*, const double*, const double*, const float*, const TreeNodeIndex*,
                                         TreeNodeIndex, const LocalIndex*, Vec4<double>*);
template void computeLeafSourceCenterGpu(const float*, const float*, const float*, const float*, const TreeNodeIndex*,
                                         TreeNodeIndex, const LocalIndex*, Vec4<float>*);

template<class T>
void upsweepCentersGpu(int numLevels, const TreeNodeIndex* levelRange, const TreeNodeIndex* childOffsets, SourceCenterType<T>* centers)
{
    check(cstone_hip_upsweep_centers(hipCtx(), bitsOf<T>, numLevels, levelRange, childOffsets, centers));
}
template void upsweepCentersGpu(int, const TreeNodeIndex*, const TreeNodeIndex*, SourceCenterType<float>*);
template void upsweepCentersGpu(int, const TreeNodeIndex*, const TreeNodeIndex*, SourceCenterType<double>*);

template<class KeyType, class T>
void computeGeoCentersGpu(const KeyType* prefixes, TreeNodeIndex numNodes, Vec3<T>* centers, Vec3<T>* sizes, const Box<T>& box)
{
    cstone_box b = podBox(box);
    check(cstone_hip_node_centers(hipCtx(), CSTONE_HILBERT, bitsOf<KeyType>, bitsOf<T>, prefixes, numNodes, &b, centers, sizes));
}
template<class KeyType, class T>
void geoMacSpheresGpu(const KeyType* prefixes, TreeNodeIndex numNodes, SourceCenterType<T>* centers, float invTheta, const Box<T>& box)
{
    cstone_box b = podBox(box);
    check(cstone_hip_geo_mac_spheres(hipCtx(), CSTONE_HILBERT, bitsOf<KeyType>, bitsOf<T>, prefixes, numNodes, centers, invTheta, &b));
}
template<class KeyType, class T>
void setMacGpu(const KeyType* prefixes, TreeNodeIndex numNodes, Vec4<T>* macSpheres, float invTheta, const Box<T>& box)
{
    cstone_box b = podBox(box);
    check(cstone_hip_set_mac(hipCtx(), CSTONE_HILBERT, bitsOf<KeyType>, bitsOf<T>, prefixes, numNodes, macSpheres, invTheta, &b));
}
#define NODE_SPHERES_GPU(KeyType, T)                                                                                   \
    template void computeGeoCentersGpu(const KeyType*, TreeNodeIndex, Vec3<T>*, Vec3<T>*, const Box<T>&);              \
    template void geoMacSpheresGpu(const KeyType*, TreeNodeIndex, SourceCenterType<T>*, float, const Box<T>&);         \
    template void setMacGpu(const KeyType*, TreeNodeIndex, Vec4<T>*, float, const Box<T>&)
NODE_SPHERES_GPU(uint32_t, float);
NODE_SPHERES_GPU(uint32_t, double);
NODE_SPHERES_GPU(uint64_t, float);
NODE_SPHERES_GPU(uint64_t, double);
#undef NODE_SPHERES_GPU

template<class T>
void moveCenters(const Vec3<T>* src, TreeNodeIndex numNodes, Vec4<T>* dest)
{
    check(cstone_hip_move_centers(hipCtx(), bitsOf<T>, src, numNodes, dest));
}
template void moveCenters(const Vec3<double>*, TreeNodeIndex, Vec4<double>*);
template void moveCenters(const Vec3<float>*, TreeNodeIndex, Vec4<float>*);

// ---- target particle groups, traversal/groups_gpu.h:46-86 (definitions in traversal/groups_gpu.cu:57-151) --------------
void computeFixedGroups(LocalIndex first, LocalIndex last, unsigned groupSize, GroupData<GpuTag>& groups)
{
    const LocalIndex numBodies = last - first;
    const LocalIndex numGroups = (numBodies + groupSize - 1) / groupSize;
    groups.data.resize(numGroups + 1);
    uint32_t made = 0;
    check(cstone_hip_compute_fixed_groups(hipCtx(), first, last, groupSize, rawPtr(groups.data), &made));
    groups.firstBody  = first;
    groups.lastBody   = last;
    groups.numGroups  = numGroups;
    groups.groupStart = rawPtr(groups.data);
    groups.groupEnd   = rawPtr(groups.data) + 1;
}

template<class Tc, class T, class KeyType>
void computeGroupSplits(LocalIndex first, LocalIndex last, const Tc* x, const Tc* y, const Tc* z, const T* /*h*/,
                        const KeyType* leaves, TreeNodeIndex numLeaves, const LocalIndex* layout, const Box<Tc> box,
                        unsigned groupSize, float tolFactor, DeviceVector<LocalIndex>& numSplitsPerGroup,
                        DeviceVector<LocalIndex>& groups)
{
    if (groupSize != 64 && groupSize != 128) throw std::runtime_error("Unsupported spatial group size\n");
    // sized like the reference's (R/traversal/groups_gpu.cu:103-104: room for a tenth more groups than fixed ones); the
    // entry reports how many entries it needs when that is not enough, and is then called once more
    const size_t numFixed = (size_t(last - first) + groupSize - 1) / groupSize;
    groups.reserve(size_t(double(numFixed) * 1.1) + 1);
    groups.resize(numFixed + 1);
    cstone_box b   = podBox(box);
    uint32_t found = 0;
    int rc = cstone_hip_compute_group_splits(hipCtx(), bitsOf<KeyType>, bitsOf<Tc>, first, last, x, y, z, leaves, numLeaves,
                                             layout, &b, groupSize, tolFactor, rawPtr(groups), groups.capacity(), &found);
    if (rc == CSTONE_E_CAPACITY)
    {
        groups.reserve(size_t(found) + 1);
        rc = cstone_hip_compute_group_splits(hipCtx(), bitsOf<KeyType>, bitsOf<Tc>, first, last, x, y, z, leaves, numLeaves,
                                             layout, &b, groupSize, tolFactor, rawPtr(groups), groups.capacity(), &found);
    }
    check(rc);
    groups.resize(size_t(found) + 1);
    // what the reference leaves in numSplitsPerGroup: the sizes of the new groups (:108-117, `newGroupSizes`)
    numSplitsPerGroup.resize(found);
    check(cstone_hip_adjacent_difference_u32(hipCtx(), rawPtr(groups), found, rawPtr(numSplitsPerGroup)));
}
template void computeGroupSplits(LocalIndex, LocalIndex, const double*, const double*, const double*, const double*,
                                 const uint64_t*, TreeNodeIndex, const LocalIndex*, const Box<double>, unsigned, float,
                                 DeviceVector<LocalIndex>&, DeviceVector<LocalIndex>&);
template void computeGroupSplits(LocalIndex, LocalIndex, const double*, const double*, const double*, const float*,
                                 const uint64_t*, TreeNodeIndex, const LocalIndex*, const Box<double>, unsigned, float,
                                 DeviceVector<LocalIndex>&, DeviceVector<LocalIndex>&);
template void computeGroupSplits(LocalIndex, LocalIndex, const float*, const float*, const float*, const float*,
                                 const uint64_t*, TreeNodeIndex, const LocalIndex*, const Box<float>, unsigned, float,
                                 DeviceVector<LocalIndex>&, DeviceVector<LocalIndex>&);

} // namespace cstone
