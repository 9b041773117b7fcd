// The remaining small primitives of the reference's GPU seam (R/primitives/primitives_gpu.h:36-124,
// R/halos/gather_halos_gpu.h): fill, scale, increment, count, reduce, segment maxima, range gathers, scalar
// lower bound, keys-only sort.  In the reference these are Thrust one-liners (R/primitives/primitives_gpu.cu:51-107,
// 217-283, 296-324, 440-448) and two small kernels (:241-259, R/halos/gather_halos_gpu.cu:26-40).
#include <algorithm>

#include "ctx.hpp"
#include "device_keys.hpp"

namespace cship
{
template<class K>
int sortPairsArena(cstone_hip_ctx* ctx, K* keys, uint32_t* vals, size_t n, int keyBits);

namespace
{

template<class T>
__global__ __launch_bounds__(256) void fillKernel(T* __restrict__ dst, size_t n, T value)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) dst[i] = value;
}

template<class T>
__global__ __launch_bounds__(256) void scaleKernel(T* __restrict__ data, size_t n, T factor)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) data[i] *= factor;
}

//! out[i] = in[i + 1] - in[i] for i < n (in holds n + 1 values): sizes from offsets
__global__ __launch_bounds__(256) void adjacentDifferenceKernel(const uint32_t* __restrict__ in, size_t n,
                                                                uint32_t* __restrict__ out)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] = in[i + 1] - in[i];
}

__global__ __launch_bounds__(256) void gatherTablesKernel(const uint32_t* __restrict__ map, const uint32_t* __restrict__ a,
                                                          size_t nA, const uint32_t* __restrict__ b, size_t nB,
                                                          const uint32_t* __restrict__ c, size_t nC,
                                                          uint32_t* __restrict__ out)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < nA) out[i] = a ? a[map[i]] : 0u;
    else if (i < nA + nB) out[i] = b[map[i]];
    else if (i < nA + nB + nC) out[i] = c[i - nA - nB];
}

template<class T>
__global__ __launch_bounds__(256) void incrementKernel(const T* __restrict__ in, T* __restrict__ out, size_t n, T value)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] + value;
}

//! acc[0] += number of elements equal to value (MODE 0) or the sum of the elements (MODE 1), 64-bit
template<class T, int MODE>
__global__ __launch_bounds__(256) void countOrSumKernel(const T* __restrict__ data, size_t n, T value,
                                                        unsigned long long* __restrict__ acc)
{
    unsigned long long local = 0;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256)
        local += MODE == 0 ? (unsigned long long)(data[i] == value) : (unsigned long long)(data[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        local += __shfl_xor(local, o);
    if ((threadIdx.x & 63u) == 0 && local) atomicAdd(acc, local);
}

//! max(|r|^2) in T, R/primitives/primitives_gpu.cu:194-215 (squares are exact to compare: a max is order-free)
template<class T>
__global__ __launch_bounds__(256) void maxNormSquareKernel(const T* __restrict__ x, const T* __restrict__ y,
                                                           const T* __restrict__ z, size_t n, T* __restrict__ partial)
{
    __shared__ T smax[4];
    T m = T(0);
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256)
    {
        T v = x[i] * x[i] + y[i] * y[i] + z[i] * z[i];
        m   = v > m ? v : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        T t = __shfl_xor(m, o);
        m   = t > m ? t : m;
    }
    if ((threadIdx.x & 63u) == 0) smax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        for (int w = 1; w < 4; ++w)
            m = smax[w] > m ? smax[w] : m;
        partial[blockIdx.x] = m;
    }
}

//! segmentMax, R/primitives/primitives_gpu.cu:241-268: out[s] = max(0, in[seg[s]..seg[s+1])) -- the reference starts
//! every segment at 0, so an empty segment (an empty leaf) gives 0; 16 lanes per segment
template<class Tin, class Tout, class I>
__global__ __launch_bounds__(256) void segmentMaxKernel(const Tin* __restrict__ in, const I* __restrict__ seg,
                                                        size_t numSegments, Tout* __restrict__ out)
{
    const unsigned sub = threadIdx.x & 15u;
    size_t s           = size_t(blockIdx.x) * 16 + (threadIdx.x >> 4);
    if (s >= numSegments) return;
    I a = seg[s], b = seg[s + 1];
    Tin m = 0;
    for (I i = a + sub; i < b; i += 16)
    {
        Tin v = in[i];
        m     = v > m ? v : m;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
    {
        Tin t = __shfl_xor(m, o);
        m     = t > m ? t : m;
    }
    if (sub == 0) out[s] = Tout(m);
}

template<int B>
struct alignas(B >= 16 ? 16 : (B >= 8 ? 8 : B)) Blob
{
    unsigned char b[B];
};
template<>
struct alignas(4) Blob<12>
{
    unsigned char b[12];
};
template<>
struct alignas(8) Blob<24>
{
    unsigned char b[24];
};

//! gatherRanges, R/halos/gather_halos_gpu.cu:26-40: buffer[i] = src[offsets[r] + i - scan[r]], r = the range that
//! holds output slot i (scan[r] <= i < scan[r+1])
template<class E, class I>
__global__ __launch_bounds__(256) void gatherRangesKernel(const I* __restrict__ scan, const I* __restrict__ offsets,
                                                          int numRanges, const E* __restrict__ src,
                                                          E* __restrict__ buffer, size_t bufferSize)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= bufferSize) return;
    int lo = 0, len = numRanges; // upper_bound(scan, scan + numRanges, i) - 1
    while (len > 0)
    {
        int half = len >> 1;
        if (!(I(i) < scan[lo + half])) { lo += half + 1, len -= half + 1; }
        else { len = half; }
    }
    const int r = lo - 1;
    buffer[i]   = src[offsets[r] + I(i) - scan[r]];
}

//! gatherRanges for up to four equally laid out arrays at once, as rows: rows[i * NUM + a] = src[a][offsets[r] + i -
//! scan[r]] (one message per peer carries all arrays of a halo exchange instead of one message per array)
template<class E, int NUM>
__global__ __launch_bounds__(256) void gatherRangesRowsKernel(const uint32_t* __restrict__ scan,
                                                              const uint32_t* __restrict__ offsets, int numRanges,
                                                              const E* __restrict__ s0, const E* __restrict__ s1,
                                                              const E* __restrict__ s2, const E* __restrict__ s3,
                                                              E* __restrict__ rows, size_t numRows)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= numRows) return;
    int lo = 0, len = numRanges;
    while (len > 0)
    {
        int half = len >> 1;
        if (!(uint32_t(i) < scan[lo + half])) { lo += half + 1, len -= half + 1; }
        else { len = half; }
    }
    const int r      = lo - 1;
    const size_t j   = size_t(offsets[r]) + (i - scan[r]);
    const E* src[4]  = {s0, s1, s2, s3};
#pragma unroll
    for (int a = 0; a < NUM; ++a)
        rows[i * NUM + a] = src[a][j];
}

//! the receiving side: dst[a][i] = rows[i * NUM + a]
template<class E, int NUM>
__global__ __launch_bounds__(256) void scatterRowsKernel(const E* __restrict__ rows, size_t numRows, E* __restrict__ d0,
                                                         E* __restrict__ d1, E* __restrict__ d2, E* __restrict__ d3)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= numRows) return;
    E* dst[4] = {d0, d1, d2, d3};
#pragma unroll
    for (int a = 0; a < NUM; ++a)
        dst[a][i] = rows[i * NUM + a];
}

//! index of the first element >= value in a sorted device array, one thread (the arrays are small or the call is rare)
template<class T>
__global__ void lowerBoundOneKernel(const T* __restrict__ data, size_t n, T value, unsigned long long* out)
{
    size_t lo = 0, len = n;
    while (len > 0)
    {
        size_t half = len >> 1;
        if (data[lo + half] < value) { lo += half + 1, len -= half + 1; }
        else { len = half; }
    }
    *out = lo;
}

int readU64(cstone_hip_ctx* ctx, unsigned long long* dev, uint64_t* out)
{
    auto* host = reinterpret_cast<unsigned long long*>(ctx->hostScalars + 12);
    CS_HIP(ctx, hipMemcpyAsync(host, dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = *host;
    return CSTONE_OK;
}

} // namespace

} // namespace cship

using namespace cship;

extern "C"
{

int cstone_hip_fill(cstone_hip_ctx* ctx, int elem_bytes, void* dst, size_t n, const void* value_host)
{
    if (!ctx || !value_host || (n && !dst)) return fail(ctx, CSTONE_E_ARG, "fill: bad argument");
    if (n == 0) return CSTONE_OK;
    unsigned grid = gridFor(n, 256);
    switch (elem_bytes)
    {
        case 1:
            hipLaunchKernelGGL(fillKernel<uint8_t>, grid, 256, 0, ctx->stream, (uint8_t*)dst, n, *(const uint8_t*)value_host);
            break;
        case 4:
            hipLaunchKernelGGL(fillKernel<uint32_t>, grid, 256, 0, ctx->stream, (uint32_t*)dst, n,
                               *(const uint32_t*)value_host);
            break;
        case 8:
            hipLaunchKernelGGL(fillKernel<uint64_t>, grid, 256, 0, ctx->stream, (uint64_t*)dst, n,
                               *(const uint64_t*)value_host);
            break;
        default: return fail(ctx, CSTONE_E_ARG, "fill: element size %d unsupported", elem_bytes);
    }
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_scale(cstone_hip_ctx* ctx, int real_bits, void* data, size_t n, double factor)
{
    if (!ctx || (real_bits != 32 && real_bits != 64) || (n && !data)) return fail(ctx, CSTONE_E_ARG, "scale: bad argument");
    if (n == 0) return CSTONE_OK;
    unsigned grid = gridFor(n, 256);
    if (real_bits == 32) hipLaunchKernelGGL(scaleKernel<float>, grid, 256, 0, ctx->stream, (float*)data, n, float(factor));
    else hipLaunchKernelGGL(scaleKernel<double>, grid, 256, 0, ctx->stream, (double*)data, n, factor);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_gather_tables_u32(cstone_hip_ctx* ctx, const uint32_t* map, const uint32_t* a, size_t n_a,
                                 const uint32_t* b, size_t n_b, const uint32_t* c, size_t n_c, uint32_t* out)
{
    const size_t n = n_a + n_b + n_c;
    if (!ctx || (n && !out) || ((n_a + n_b) && !map) || (n_b && !b) || (n_c && !c))
        return fail(ctx, CSTONE_E_ARG, "gather_tables: bad argument");
    if (n == 0) return CSTONE_OK;
    hipLaunchKernelGGL(gatherTablesKernel, gridFor(n, 256), 256, 0, ctx->stream, map, a, n_a, b, n_b, c, n_c, out);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_adjacent_difference_u32(cstone_hip_ctx* ctx, const uint32_t* in, size_t n, uint32_t* out)
{
    if (!ctx || (n && (!in || !out)) || in == out) return fail(ctx, CSTONE_E_ARG, "adjacent_difference: bad argument");
    if (n == 0) return CSTONE_OK;
    hipLaunchKernelGGL(adjacentDifferenceKernel, gridFor(n, 256), 256, 0, ctx->stream, in, n, out);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_increment(cstone_hip_ctx* ctx, int elem_bits, const void* in, void* out, size_t n, uint64_t value)
{
    if (!ctx || (elem_bits != 32 && elem_bits != 64) || (n && (!in || !out)))
        return fail(ctx, CSTONE_E_ARG, "increment: bad argument");
    if (n == 0) return CSTONE_OK;
    unsigned grid = gridFor(n, 256);
    if (elem_bits == 32)
        hipLaunchKernelGGL(incrementKernel<uint32_t>, grid, 256, 0, ctx->stream, (const uint32_t*)in, (uint32_t*)out, n,
                           uint32_t(value));
    else
        hipLaunchKernelGGL(incrementKernel<uint64_t>, grid, 256, 0, ctx->stream, (const uint64_t*)in, (uint64_t*)out, n,
                           value);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

static int countOrSum(cstone_hip_ctx* ctx, int mode, int elem_bits, const void* data, size_t n, uint64_t value,
                      uint64_t* out_host)
{
    auto* acc = reinterpret_cast<unsigned long long*>(ctx->devScalars + 12);
    CS_HIP(ctx, hipMemsetAsync(acc, 0, sizeof(unsigned long long), ctx->stream));
    unsigned grid = unsigned(std::min<size_t>(size_t(ctx->numCu) * 8, (n + 255) / 256));
    if (elem_bits == 32)
    {
        if (mode == 0)
            hipLaunchKernelGGL((countOrSumKernel<uint32_t, 0>), grid, 256, 0, ctx->stream, (const uint32_t*)data, n,
                               uint32_t(value), acc);
        else
            hipLaunchKernelGGL((countOrSumKernel<uint32_t, 1>), grid, 256, 0, ctx->stream, (const uint32_t*)data, n,
                               uint32_t(value), acc);
    }
    else
    {
        if (mode == 0)
            hipLaunchKernelGGL((countOrSumKernel<uint64_t, 0>), grid, 256, 0, ctx->stream, (const uint64_t*)data, n,
                               value, acc);
        else
            hipLaunchKernelGGL((countOrSumKernel<uint64_t, 1>), grid, 256, 0, ctx->stream, (const uint64_t*)data, n,
                               value, acc);
    }
    CS_HIP(ctx, hipGetLastError());
    return readU64(ctx, acc, out_host);
}

int cstone_hip_count_equal(cstone_hip_ctx* ctx, int elem_bits, const void* data, size_t n, uint64_t value,
                           uint64_t* count_host)
{
    if (!ctx || (elem_bits != 32 && elem_bits != 64) || !count_host || (n && !data))
        return fail(ctx, CSTONE_E_ARG, "count_equal: bad argument");
    *count_host = 0;
    if (n == 0) return CSTONE_OK;
    return countOrSum(ctx, 0, elem_bits, data, n, value, count_host);
}

int cstone_hip_reduce_sum(cstone_hip_ctx* ctx, int elem_bits, const void* data, size_t n, uint64_t init,
                          uint64_t* sum_host)
{
    if (!ctx || (elem_bits != 32 && elem_bits != 64) || !sum_host || (n && !data))
        return fail(ctx, CSTONE_E_ARG, "reduce_sum: bad argument");
    *sum_host = init;
    if (n == 0) return CSTONE_OK;
    uint64_t s = 0;
    CS_TRY(countOrSum(ctx, 1, elem_bits, data, n, 0, &s));
    *sum_host = init + s;
    return CSTONE_OK;
}

int cstone_hip_max_norm_square(cstone_hip_ctx* ctx, int real_bits, const void* x, const void* y, const void* z, size_t n,
                               double* out_host)
{
    if (!ctx || (real_bits != 32 && real_bits != 64) || !out_host || (n && (!x || !y || !z)))
        return fail(ctx, CSTONE_E_ARG, "max_norm_square: bad argument");
    *out_host = 0;
    if (n == 0) return CSTONE_OK;
    unsigned grid = unsigned(std::min<size_t>(size_t(ctx->numCu) * 4, (n + 255) / 256));
    CS_TRY(arenaReserve(ctx, alignUp(size_t(grid) * 8) + 256));
    void* partial = arenaTake(ctx, size_t(grid) * 8);
    std::vector<double> host64(grid);
    std::vector<float> host32(grid);
    hipError_t e;
    if (real_bits == 32)
    {
        hipLaunchKernelGGL(maxNormSquareKernel<float>, grid, 256, 0, ctx->stream, (const float*)x, (const float*)y,
                           (const float*)z, n, (float*)partial);
        e = hipMemcpyAsync(host32.data(), partial, size_t(grid) * 4, hipMemcpyDeviceToHost, ctx->stream);
    }
    else
    {
        hipLaunchKernelGGL(maxNormSquareKernel<double>, grid, 256, 0, ctx->stream, (const double*)x, (const double*)y,
                           (const double*)z, n, (double*)partial);
        e = hipMemcpyAsync(host64.data(), partial, size_t(grid) * 8, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    arenaReset(ctx);
    if (e != hipSuccess) return fail(ctx, CSTONE_E_HIP, "max_norm_square: %s", hipGetErrorString(e));
    double m = 0;
    for (unsigned i = 0; i < grid; ++i)
        m = std::max(m, real_bits == 32 ? double(host32[i]) : host64[i]);
    *out_host = m;
    return CSTONE_OK;
}

int cstone_hip_segment_max(cstone_hip_ctx* ctx, int in_bits, int out_bits, int index_bits, const void* in,
                           const void* segments, size_t num_segments, void* out)
{
    if (!ctx || (num_segments && (!in || !segments || !out)))
        return fail(ctx, CSTONE_E_ARG, "segment_max: bad argument");
    if (num_segments == 0) return CSTONE_OK;
    unsigned grid = gridFor(num_segments, 16);
#define CSTONE_SEGMAX(Tin, Tout, I)                                                                                    \
    hipLaunchKernelGGL((segmentMaxKernel<Tin, Tout, I>), grid, 256, 0, ctx->stream, (const Tin*)in, (const I*)segments, \
                       num_segments, (Tout*)out)
    // the reference's instantiations, R/primitives/primitives_gpu.cu:270-275
    if (in_bits == 32 && out_bits == 32 && index_bits == 32) CSTONE_SEGMAX(float, float, uint32_t);
    else if (in_bits == 64 && out_bits == 32 && index_bits == 32) CSTONE_SEGMAX(double, float, uint32_t);
    else if (in_bits == 64 && out_bits == 64 && index_bits == 32) CSTONE_SEGMAX(double, double, uint32_t);
    else if (in_bits == 32 && out_bits == 32 && index_bits == 64) CSTONE_SEGMAX(float, float, uint64_t);
    else if (in_bits == 64 && out_bits == 32 && index_bits == 64) CSTONE_SEGMAX(double, float, uint64_t);
    else if (in_bits == 64 && out_bits == 64 && index_bits == 64) CSTONE_SEGMAX(double, double, uint64_t);
    else return fail(ctx, CSTONE_E_ARG, "segment_max: unsupported type combination %d/%d/%d", in_bits, out_bits, index_bits);
#undef CSTONE_SEGMAX
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_gather_ranges(cstone_hip_ctx* ctx, int elem_bytes, int index_bits, const void* range_scan,
                             const void* range_offsets, int num_ranges, const void* src, void* buffer,
                             size_t buffer_size)
{
    if (!ctx || (index_bits != 32 && index_bits != 64) || num_ranges < 0 ||
        (buffer_size && (!range_scan || !range_offsets || !src || !buffer || num_ranges == 0)))
        return fail(ctx, CSTONE_E_ARG, "gather_ranges: bad argument");
    if (buffer_size == 0) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_GATHER);
    unsigned grid = gridFor(buffer_size, 256);
#define CSTONE_GR_CASE(B)                                                                                              \
    case B:                                                                                                            \
        if (index_bits == 32)                                                                                          \
            hipLaunchKernelGGL((gatherRangesKernel<Blob<B>, uint32_t>), grid, 256, 0, ctx->stream,                     \
                               (const uint32_t*)range_scan, (const uint32_t*)range_offsets, num_ranges,                \
                               (const Blob<B>*)src, (Blob<B>*)buffer, buffer_size);                                    \
        else                                                                                                           \
            hipLaunchKernelGGL((gatherRangesKernel<Blob<B>, uint64_t>), grid, 256, 0, ctx->stream,                     \
                               (const uint64_t*)range_scan, (const uint64_t*)range_offsets, num_ranges,                \
                               (const Blob<B>*)src, (Blob<B>*)buffer, buffer_size);                                    \
        break
    switch (elem_bytes)
    {
        CSTONE_GR_CASE(1);
        CSTONE_GR_CASE(2);
        CSTONE_GR_CASE(4);
        CSTONE_GR_CASE(8);
        CSTONE_GR_CASE(12);
        CSTONE_GR_CASE(16);
        CSTONE_GR_CASE(24);
        CSTONE_GR_CASE(32);
        default: return fail(ctx, CSTONE_E_ARG, "gather_ranges: element size %d unsupported", elem_bytes);
    }
#undef CSTONE_GR_CASE
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_gather_ranges_rows(cstone_hip_ctx* ctx, int elem_bytes, int num_arrays, const uint32_t* range_scan,
                                  const uint32_t* range_offsets, int num_ranges, const void* const* src, void* rows,
                                  size_t num_rows)
{
    if (!ctx || num_arrays < 1 || num_arrays > 4 || (elem_bytes != 4 && elem_bytes != 8) || num_ranges < 0 ||
        (num_rows && (!range_scan || !range_offsets || !src || !rows || num_ranges == 0)))
        return fail(ctx, CSTONE_E_ARG, "gather_ranges_rows: bad argument");
    if (num_rows == 0) return CSTONE_OK;
    if (num_rows >= (size_t(1) << 32)) return fail(ctx, CSTONE_E_ARG, "gather_ranges_rows: too many rows");
    const void* s[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int a = 0; a < num_arrays; ++a)
    {
        if (!src[a]) return fail(ctx, CSTONE_E_ARG, "gather_ranges_rows: null array");
        s[a] = src[a];
    }
    StageTimer timer(ctx, CSTONE_STAGE_GATHER);
    unsigned grid = gridFor(num_rows, 256);
#define CSTONE_GRR(E, NUM)                                                                                             \
    hipLaunchKernelGGL((gatherRangesRowsKernel<E, NUM>), grid, 256, 0, ctx->stream, range_scan, range_offsets,         \
                       num_ranges, (const E*)s[0], (const E*)s[1], (const E*)s[2], (const E*)s[3], (E*)rows, num_rows)
#define CSTONE_GRR_NUM(E)                                                                                              \
    switch (num_arrays)                                                                                                \
    {                                                                                                                  \
        case 1: CSTONE_GRR(E, 1); break;                                                                               \
        case 2: CSTONE_GRR(E, 2); break;                                                                               \
        case 3: CSTONE_GRR(E, 3); break;                                                                               \
        default: CSTONE_GRR(E, 4); break;                                                                              \
    }
    if (elem_bytes == 4) { CSTONE_GRR_NUM(uint32_t) }
    else { CSTONE_GRR_NUM(uint64_t) }
#undef CSTONE_GRR_NUM
#undef CSTONE_GRR
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_scatter_rows(cstone_hip_ctx* ctx, int elem_bytes, int num_arrays, const void* rows, size_t num_rows,
                            void* const* dst, size_t dst_offset)
{
    if (!ctx || num_arrays < 1 || num_arrays > 4 || (elem_bytes != 4 && elem_bytes != 8) || (num_rows && (!rows || !dst)))
        return fail(ctx, CSTONE_E_ARG, "scatter_rows: bad argument");
    if (num_rows == 0) return CSTONE_OK;
    char* d[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int a = 0; a < num_arrays; ++a)
    {
        if (!dst[a]) return fail(ctx, CSTONE_E_ARG, "scatter_rows: null array");
        d[a] = static_cast<char*>(dst[a]) + dst_offset * size_t(elem_bytes);
    }
    StageTimer timer(ctx, CSTONE_STAGE_GATHER);
    unsigned grid = gridFor(num_rows, 256);
#define CSTONE_SR(E, NUM)                                                                                              \
    hipLaunchKernelGGL((scatterRowsKernel<E, NUM>), grid, 256, 0, ctx->stream, (const E*)rows, num_rows, (E*)d[0],     \
                       (E*)d[1], (E*)d[2], (E*)d[3])
#define CSTONE_SR_NUM(E)                                                                                               \
    switch (num_arrays)                                                                                                \
    {                                                                                                                  \
        case 1: CSTONE_SR(E, 1); break;                                                                                \
        case 2: CSTONE_SR(E, 2); break;                                                                                \
        case 3: CSTONE_SR(E, 3); break;                                                                                \
        default: CSTONE_SR(E, 4); break;                                                                               \
    }
    if (elem_bytes == 4) { CSTONE_SR_NUM(uint32_t) }
    else { CSTONE_SR_NUM(uint64_t) }
#undef CSTONE_SR_NUM
#undef CSTONE_SR
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_lower_bound_value(cstone_hip_ctx* ctx, int kind, const void* data, size_t n, const void* value_host,
                                 uint64_t* index_host)
{
    if (!ctx || !value_host || !index_host || (n && !data)) return fail(ctx, CSTONE_E_ARG, "lower_bound_value: bad argument");
    *index_host = 0;
    if (n == 0) return CSTONE_OK;
    auto* out = reinterpret_cast<unsigned long long*>(ctx->devScalars + 12);
    switch (kind)
    {
        case 0:
            hipLaunchKernelGGL(lowerBoundOneKernel<uint32_t>, 1, 1, 0, ctx->stream, (const uint32_t*)data, n,
                               *(const uint32_t*)value_host, out);
            break;
        case 1:
            hipLaunchKernelGGL(lowerBoundOneKernel<uint64_t>, 1, 1, 0, ctx->stream, (const uint64_t*)data, n,
                               *(const uint64_t*)value_host, out);
            break;
        case 2:
            hipLaunchKernelGGL(lowerBoundOneKernel<int32_t>, 1, 1, 0, ctx->stream, (const int32_t*)data, n,
                               *(const int32_t*)value_host, out);
            break;
        case 3:
            hipLaunchKernelGGL(lowerBoundOneKernel<int64_t>, 1, 1, 0, ctx->stream, (const int64_t*)data, n,
                               *(const int64_t*)value_host, out);
            break;
        case 4:
            hipLaunchKernelGGL(lowerBoundOneKernel<float>, 1, 1, 0, ctx->stream, (const float*)data, n,
                               *(const float*)value_host, out);
            break;
        default: return fail(ctx, CSTONE_E_ARG, "lower_bound_value: kind %d unsupported", kind);
    }
    CS_HIP(ctx, hipGetLastError());
    return readU64(ctx, out, index_host);
}

int cstone_hip_sort_keys(cstone_hip_ctx* ctx, int key_bits, void* keys, size_t n)
{
    if (!ctx || (key_bits != 32 && key_bits != 64) || (n && !keys)) return fail(ctx, CSTONE_E_ARG, "sort_keys: bad argument");
    if (n < 2) return CSTONE_OK;
    // rare and small (injectKeys, R/focus/inject.hpp:97): the pair sort with a throw-away payload
    uint32_t* payload = nullptr;
    CS_HIP(ctx, hipMalloc((void**)&payload, n * sizeof(uint32_t)));
    int rc = key_bits == 32 ? sortPairsArena<uint32_t>(ctx, (uint32_t*)keys, payload, n, 32)
                            : sortPairsArena<uint64_t>(ctx, (uint64_t*)keys, payload, n, 64);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(payload);
    if (rc == CSTONE_OK && e != hipSuccess) return fail(ctx, CSTONE_E_HIP, "sort_keys: %s", hipGetErrorString(e));
    return rc;
}

} // extern "C"
