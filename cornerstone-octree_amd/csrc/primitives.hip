// Streaming primitives for gfx950: gather / scatter, min-max reduction, prefix sums.
// Replace the Thrust/CUB one-liners and small kernels of R/primitives/primitives_gpu.cu
// (gatherGpu :110-148, scatterGpu :151-175, MinMaxGpu :178-187, exclusive/inclusiveScanGpu :395-437).
#include <algorithm>

#include "ctx.hpp"
#include "device_keys.hpp"
#include "scan.hpp"

namespace cship
{

namespace
{

template<int B>
struct alignas(B >= 16 ? 16 : (B >= 8 ? 8 : B)) Elem
{
    unsigned char b[B];
};
template<>
struct alignas(4) Elem<12>
{
    unsigned char b[12];
};
template<>
struct alignas(8) Elem<24>
{
    unsigned char b[24];
};

// result[q] = index of the first key >= values[q] (unsigned comparison), one lane per query
template<class K, class I = uint64_t>
__global__ __launch_bounds__(256) void lowerBoundKernel(const K* __restrict__ keys, size_t n,
                                                        const K* __restrict__ values, int numValues,
                                                        I* __restrict__ result)
{
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= numValues) return;
    K v       = values[q];
    size_t lo = 0, len = n;
    while (len > 0)
    {
        size_t half = len >> 1;
        if (keys[lo + half] < v)
        {
            lo += half + 1;
            len -= half + 1;
        }
        else { len = half; }
    }
    result[q] = I(lo);
}

//! out[i] = init + i (64-bit flavour of sequenceGpu)
__global__ __launch_bounds__(256) void sequence64Kernel(uint64_t* __restrict__ out, size_t n, uint64_t init)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < n) out[i] = init + i;
}

// ---- prefix sums of 32-bit values in 64 bits (exclusiveScanGpu<unsigned, uint64_t> of the reference's list): the same
//      three launches as csrc/scan.hip with 64-bit partial sums
constexpr int SCAN64_ITEMS = 8;
__device__ __forceinline__ uint64_t blockExclusiveScan256u64(uint64_t v, uint64_t* waveSums /*LDS[4]*/, uint64_t* total)
{
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
        const uint64_t t = uint64_t(__shfl_up((unsigned long long)inc, o));
        if (lane >= unsigned(o)) inc += t;
    }
    __syncthreads();
    if (lane == 63) waveSums[w] = inc;
    __syncthreads();
    uint64_t off = 0, tot = 0;
#pragma unroll
    for (unsigned i = 0; i < 4; ++i)
    {
        const uint64_t s = waveSums[i];
        if (i < w) off += s;
        tot += s;
    }
    if (total) *total = tot;
    return off + inc - v;
}
__global__ __launch_bounds__(256) void blockSum64Kernel(const uint32_t* __restrict__ in, size_t n, uint64_t* __restrict__ sums)
{
    __shared__ uint64_t ws[4];
    const size_t base = (size_t(blockIdx.x) * 256 + threadIdx.x) * SCAN64_ITEMS;
    uint64_t s        = 0;
#pragma unroll
    for (int k = 0; k < SCAN64_ITEMS; ++k)
        if (base + k < n) s += in[base + k];
    uint64_t total;
    blockExclusiveScan256u64(s, ws, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void scanSums64Kernel(uint64_t* __restrict__ sums, unsigned m, uint64_t init)
{
    __shared__ uint64_t ws[4];
    uint64_t carry = init;
    for (unsigned base = 0; base < m; base += 256)
    {
        const unsigned i = base + threadIdx.x;
        const uint64_t v = i < m ? sums[i] : 0;
        uint64_t total;
        const uint64_t ex = blockExclusiveScan256u64(v, ws, &total);
        if (i < m) sums[i] = carry + ex;
        carry += total;
    }
}
__global__ __launch_bounds__(256) void blockScan64Kernel(const uint32_t* __restrict__ in, size_t n,
                                                         const uint64_t* __restrict__ sums, uint64_t* __restrict__ out,
                                                         bool inclusive)
{
    __shared__ uint64_t ws[4];
    const size_t base = (size_t(blockIdx.x) * 256 + threadIdx.x) * SCAN64_ITEMS;
    uint32_t v[SCAN64_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN64_ITEMS; ++k)
    {
        v[k] = base + k < n ? in[base + k] : 0u;
        s += v[k];
    }
    uint64_t run = sums[blockIdx.x] + blockExclusiveScan256u64(s, ws, nullptr);
#pragma unroll
    for (int k = 0; k < SCAN64_ITEMS; ++k)
    {
        if (base + k < n) out[base + k] = inclusive ? run + v[k] : run;
        run += v[k];
    }
}

// dst[i] = src[map[i]]: map and dst are streamed, src is a random read (4 + 2E bytes per element)
template<class E, int PER>
__global__ __launch_bounds__(256) void gatherKernel(const uint32_t* __restrict__ map, size_t n,
                                                    const E* __restrict__ src, E* __restrict__ dst)
{
    size_t base = size_t(blockIdx.x) * (256 * PER) + threadIdx.x;
    uint32_t idx[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k)
    {
        size_t i = base + size_t(k) * 256;
        idx[k]   = i < n ? map[i] : 0u;
    }
    E v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k)
    {
        size_t i = base + size_t(k) * 256;
        if (i < n) v[k] = src[idx[k]];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k)
    {
        size_t i = base + size_t(k) * 256;
        if (i < n) dst[i] = v[k];
    }
}

template<class E, int PER>
__global__ __launch_bounds__(256) void scatterKernel(const uint32_t* __restrict__ map, size_t n,
                                                     const E* __restrict__ src, E* __restrict__ dst)
{
    size_t base = size_t(blockIdx.x) * (256 * PER) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < PER; ++k)
    {
        size_t i = base + size_t(k) * 256;
        if (i < n) dst[map[i]] = src[i];
    }
}

//! dst[mapOut[i]] = src[mapIn[i]]: a gather and a scatter in one pass (gatherScatter, R/primitives/gather.hpp:120-131)
template<class E, int PER>
__global__ __launch_bounds__(256) void gatherScatterKernel(const uint32_t* __restrict__ mapIn,
                                                           const uint32_t* __restrict__ mapOut, size_t n,
                                                           const E* __restrict__ src, E* __restrict__ dst)
{
    size_t base = size_t(blockIdx.x) * (256 * PER) + threadIdx.x;
    uint32_t in[PER], out[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k)
    {
        size_t i = base + size_t(k) * 256;
        in[k]    = i < n ? mapIn[i] : 0u;
        out[k]   = i < n ? mapOut[i] : 0u;
    }
    E v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k)
    {
        size_t i = base + size_t(k) * 256;
        if (i < n) v[k] = src[in[k]];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k)
    {
        size_t i = base + size_t(k) * 256;
        if (i < n) dst[out[k]] = v[k];
    }
}

//! positions of the elements of two sorted runs in their stable merge (ties: run A first)
template<class K>
__global__ __launch_bounds__(256) void mergePositionsKernel(const K* __restrict__ a, size_t na, const K* __restrict__ b,
                                                            size_t nb, uint32_t offset, uint32_t* __restrict__ posA,
                                                            uint32_t* __restrict__ posB)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i < na)
    {
        // elements of B strictly below a[i]
        K v       = a[i];
        size_t lo = 0, len = nb;
        while (len > 0)
        {
            size_t half = len >> 1;
            if (b[lo + half] < v) { lo += half + 1, len -= half + 1; }
            else { len = half; }
        }
        posA[i] = offset + uint32_t(i + lo);
    }
    else if (i < na + nb)
    {
        // elements of A below or equal to b[j]
        size_t j  = i - na;
        K v       = b[j];
        size_t lo = 0, len = na;
        while (len > 0)
        {
            size_t half = len >> 1;
            if (!(v < a[lo + half])) { lo += half + 1, len -= half + 1; }
            else { len = half; }
        }
        posB[j] = offset + uint32_t(j + lo);
    }
}

//! dst[a][i] = src[a][map[i]] for NUM arrays of equal element size: the map is read ONCE for all of them (4 + 2 E NUM
//! bytes per element instead of NUM (4 + 2 E)), all loads of a thread are in flight before the first store
template<class E, int PER, int NUM>
__global__ __launch_bounds__(256) void gatherMultiKernel(const uint32_t* __restrict__ map, size_t n,
                                                         const E* __restrict__ s0, const E* __restrict__ s1,
                                                         const E* __restrict__ s2, const E* __restrict__ s3,
                                                         E* __restrict__ d0, E* __restrict__ d1, E* __restrict__ d2,
                                                         E* __restrict__ d3)
{
    size_t base = size_t(blockIdx.x) * (256 * PER) + threadIdx.x;
    uint32_t in[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k)
    {
        size_t i = base + size_t(k) * 256;
        in[k]    = i < n ? map[i] : 0u;
    }
    const E* src[4] = {s0, s1, s2, s3};
    E* dst[4]       = {d0, d1, d2, d3};
    E v[NUM][PER];
#pragma unroll
    for (int a = 0; a < NUM; ++a)
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (base + size_t(k) * 256 < n) v[a][k] = src[a][in[k]];
#pragma unroll
    for (int a = 0; a < NUM; ++a)
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (base + size_t(k) * 256 < n) dst[a][base + size_t(k) * 256] = v[a][k];
}

template<int B, int NUM>
void launchGatherMulti(cstone_hip_ctx* ctx, const uint32_t* map, size_t n, const void* const* src, void* const* dst)
{
    constexpr int PER = 4;
    using E           = Elem<B>;
    const E* s[4]     = {nullptr, nullptr, nullptr, nullptr};
    E* d[4]           = {nullptr, nullptr, nullptr, nullptr};
    for (int a = 0; a < NUM; ++a)
        s[a] = (const E*)src[a], d[a] = (E*)dst[a];
    hipLaunchKernelGGL((gatherMultiKernel<E, PER, NUM>), gridFor(n, 256, PER), 256, 0, ctx->stream, map, n, s[0], s[1],
                       s[2], s[3], d[0], d[1], d[2], d[3]);
}

template<bool GATHER, int B>
void launchPermute(cstone_hip_ctx* ctx, const uint32_t* map, size_t n, const void* src, void* dst)
{
    constexpr int PER = 4;
    unsigned grid     = gridFor(n, 256, PER);
    if (GATHER)
        hipLaunchKernelGGL((gatherKernel<Elem<B>, PER>), grid, 256, 0, ctx->stream, map, n, (const Elem<B>*)src,
                           (Elem<B>*)dst);
    else
        hipLaunchKernelGGL((scatterKernel<Elem<B>, PER>), grid, 256, 0, ctx->stream, map, n, (const Elem<B>*)src,
                           (Elem<B>*)dst);
}

template<bool GATHER>
int permute(cstone_hip_ctx* ctx, int elemBytes, const uint32_t* map, size_t n, const void* src, void* dst)
{
    if (!ctx) return CSTONE_E_ARG;
    if (n == 0) return CSTONE_OK;
    if (!map || !src || !dst) return fail(ctx, CSTONE_E_ARG, "gather/scatter: null array");
    StageTimer timer(ctx, CSTONE_STAGE_GATHER);
    // alignment the element type is accessed with: Vec3<float> = 12 bytes at 4, Vec3<double> = 24 bytes at 8
    // (R/util/array.hpp:42-58), 16- and 32-byte elements as 16-byte vectors
    int natural = elemBytes == 12 ? 4 : (elemBytes == 24 ? 8 : (elemBytes >= 16 ? 16 : elemBytes));
    if ((uintptr_t(src) % natural) || (uintptr_t(dst) % natural))
        return fail(ctx, CSTONE_E_ARG, "gather/scatter: arrays must be aligned to %d bytes", natural);
    switch (elemBytes)
    {
        case 1: launchPermute<GATHER, 1>(ctx, map, n, src, dst); break;
        case 2: launchPermute<GATHER, 2>(ctx, map, n, src, dst); break;
        case 4: launchPermute<GATHER, 4>(ctx, map, n, src, dst); break;
        case 8: launchPermute<GATHER, 8>(ctx, map, n, src, dst); break;
        case 12: launchPermute<GATHER, 12>(ctx, map, n, src, dst); break;
        case 16: launchPermute<GATHER, 16>(ctx, map, n, src, dst); break;
        case 24: launchPermute<GATHER, 24>(ctx, map, n, src, dst); break;
        case 32: launchPermute<GATHER, 32>(ctx, map, n, src, dst); break;
        default: return fail(ctx, CSTONE_E_ARG, "gather/scatter: element size %d unsupported", elemBytes);
    }
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

// ---- min/max: per-block partials, then one workgroup folds them (two launches: the launch boundary
//      provides the cross-XCD visibility, no in-kernel hand-off needed)
template<class T>
__device__ __forceinline__ void blockMinMax(T& lo, T& hi, T* smin, T* smax)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        T a = __shfl_xor(lo, o), b = __shfl_xor(hi, o);
        lo  = a < lo ? a : lo;
        hi  = b > hi ? b : hi;
    }
    unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (lane == 0) smin[w] = lo, smax[w] = hi;
    __syncthreads();
    for (int i = 0; i < 4; ++i)
    {
        lo = smin[i] < lo ? smin[i] : lo;
        hi = smax[i] > hi ? smax[i] : hi;
    }
}

//! grid-stride min/max with 16-byte loads, four of them in flight per lane
template<class T>
struct MinMaxArrays
{
    const T* x[3];
};

template<class T>
__global__ __launch_bounds__(256) void minMaxPartialKernel(MinMaxArrays<T> arrays, size_t n, T* __restrict__ partialAll)
{
    const T* __restrict__ x = arrays.x[blockIdx.y];
    T* __restrict__ partial = partialAll + size_t(blockIdx.y) * gridDim.x * 2;
    constexpr int VEC = 16 / sizeof(T);
    struct alignas(16) Pack
    {
        T v[VEC];
    };
    __shared__ T smin[4], smax[4];
    T lo = x[0], hi = x[0];
    // head up to the first 16-byte boundary and tail beyond the last full pack: first block, scalar
    size_t head      = (VEC - (reinterpret_cast<uintptr_t>(x) / sizeof(T)) % VEC) % VEC;
    head             = head < n ? head : n;
    const size_t nPk = (n - head) / VEC;
    const Pack* px   = reinterpret_cast<const Pack*>(x + head);
    const size_t stride = size_t(gridDim.x) * 256;
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    for (; i + 3 * stride < nPk; i += 4 * stride)
    {
        Pack p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            p[k] = px[i + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < VEC; ++j)
            {
                lo = p[k].v[j] < lo ? p[k].v[j] : lo;
                hi = p[k].v[j] > hi ? p[k].v[j] : hi;
            }
    }
    for (; i < nPk; i += stride)
    {
        Pack p = px[i];
#pragma unroll
        for (int j = 0; j < VEC; ++j)
        {
            lo = p.v[j] < lo ? p.v[j] : lo;
            hi = p.v[j] > hi ? p.v[j] : hi;
        }
    }
    if (blockIdx.x == 0)
    {
        for (size_t j = threadIdx.x; j < head; j += 256)
        {
            lo = x[j] < lo ? x[j] : lo;
            hi = x[j] > hi ? x[j] : hi;
        }
        for (size_t j = head + nPk * VEC + threadIdx.x; j < n; j += 256)
        {
            lo = x[j] < lo ? x[j] : lo;
            hi = x[j] > hi ? x[j] : hi;
        }
    }
    blockMinMax(lo, hi, smin, smax);
    if (threadIdx.x == 0) partial[2 * blockIdx.x] = lo, partial[2 * blockIdx.x + 1] = hi;
}

template<class T>
__global__ __launch_bounds__(256) void minMaxFinalKernel(const T* __restrict__ partialAll, unsigned m,
                                                         T* __restrict__ outAll)
{
    const T* __restrict__ partial = partialAll + size_t(blockIdx.x) * m * 2;
    T* __restrict__ out           = outAll + 2 * blockIdx.x;
    __shared__ T smin[4], smax[4];
    T lo = partial[0], hi = partial[1];
    for (unsigned b = threadIdx.x; b < m; b += 256)
    {
        T a = partial[2 * b], c = partial[2 * b + 1];
        lo  = a < lo ? a : lo;
        hi  = c > hi ? c : hi;
    }
    blockMinMax(lo, hi, smin, smax);
    if (threadIdx.x == 0) out[0] = lo, out[1] = hi;
}

//! min and max of up to three arrays of n elements each: one launch pair, one read-back
template<class T>
int minMaxArrays(cstone_hip_ctx* ctx, const T* const* xs, int numArrays, size_t n, double* out)
{
    if (n == 0) return fail(ctx, CSTONE_E_ARG, "minmax: empty range");
    unsigned grid = unsigned(std::min<size_t>(size_t(ctx->numCu) * 8, (n + 255) / 256));
    CS_TRY(arenaReserve(ctx, alignUp(size_t(grid) * 6 * sizeof(T)) + 1024));
    T* partial = (T*)arenaTake(ctx, size_t(grid) * 6 * sizeof(T));
    T* res     = (T*)arenaTake(ctx, 6 * sizeof(T));
    MinMaxArrays<T> arrays{{xs[0], xs[numArrays > 1 ? 1 : 0], xs[numArrays > 2 ? 2 : 0]}};
    {
        StageTimer timer(ctx, CSTONE_STAGE_MINMAX);
        hipLaunchKernelGGL(minMaxPartialKernel<T>, dim3(grid, numArrays), 256, 0, ctx->stream, arrays, n, partial);
        hipLaunchKernelGGL(minMaxFinalKernel<T>, numArrays, 256, 0, ctx->stream, partial, grid, res);
    }
    T host[6];
    hipError_t e = hipMemcpyAsync(host, res, size_t(2 * numArrays) * sizeof(T), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    arenaReset(ctx);
    if (e != hipSuccess) return fail(ctx, CSTONE_E_HIP, "minmax: %s", hipGetErrorString(e));
    for (int i = 0; i < 2 * numArrays; ++i)
        out[i] = host[i];
    return CSTONE_OK;
}

template<class T>
int minMax(cstone_hip_ctx* ctx, const T* x, size_t n, double* out2)
{
    return minMaxArrays<T>(ctx, &x, 1, n, out2);
}

//! (min, -max) of each array as doubles: the operand of a MIN all-reduce over the ranks
template<class T>
__global__ void minNegMaxKernel(const T* __restrict__ res, int numArrays, double* __restrict__ out)
{
    int i = threadIdx.x;
    if (i < numArrays) out[2 * i] = double(res[2 * i]), out[2 * i + 1] = -double(res[2 * i + 1]);
}

//! the extents stay on the device (no read-back): devOut[2 d] = min, devOut[2 d + 1] = -max of array d
template<class T>
int minMaxArraysDev(cstone_hip_ctx* ctx, const T* const* xs, int numArrays, size_t n, double* devOut)
{
    if (n == 0) return fail(ctx, CSTONE_E_ARG, "minmax: empty range");
    unsigned grid = unsigned(std::min<size_t>(size_t(ctx->numCu) * 8, (n + 255) / 256));
    CS_TRY(arenaReserve(ctx, alignUp(size_t(grid) * 6 * sizeof(T)) + 1024));
    T* partial = (T*)arenaTake(ctx, size_t(grid) * 6 * sizeof(T));
    T* res     = (T*)arenaTake(ctx, 6 * sizeof(T));
    MinMaxArrays<T> arrays{{xs[0], xs[numArrays > 1 ? 1 : 0], xs[numArrays > 2 ? 2 : 0]}};
    {
        StageTimer timer(ctx, CSTONE_STAGE_MINMAX);
        hipLaunchKernelGGL(minMaxPartialKernel<T>, dim3(grid, numArrays), 256, 0, ctx->stream, arrays, n, partial);
        hipLaunchKernelGGL(minMaxFinalKernel<T>, numArrays, 256, 0, ctx->stream, partial, grid, res);
        hipLaunchKernelGGL(minNegMaxKernel<T>, 1, 64, 0, ctx->stream, res, numArrays, devOut);
    }
    arenaReset(ctx); // later calls reuse the slices behind these launches in stream order
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // namespace

int minMaxCoordinatesDev(cstone_hip_ctx* ctx, int real_bits, const void* const* xs, int numArrays, size_t n, double* devOut)
{
    if (real_bits == 32) return minMaxArraysDev<float>(ctx, (const float* const*)xs, numArrays, n, devOut);
    return minMaxArraysDev<double>(ctx, (const double* const*)xs, numArrays, n, devOut);
}

namespace
{
//! the whole operand of the box reduction of a multi-rank sync in one launch (see extentsToReduceOperand)
template<class T>
__global__ void reduceOperandKernel(const T* __restrict__ res, double* __restrict__ out, double status,
                                    const int* __restrict__ counters)
{
    int i = threadIdx.x;
    if (i < 3) out[2 * i] = double(res[2 * i]), out[2 * i + 1] = -double(res[2 * i + 1]);
    if (i == 3) out[6] = status;
    if (counters && i >= 4 && i < 8) reinterpret_cast<int*>(out + 7)[i - 4] = counters[i - 4];
}
} // namespace

/*! {min, max} per axis as T (what the encode kernels measure on the way) -> (min, -max) as doubles at devOut[0..5],
 *  the rank's status word at devOut[6] (it used to be a host-to-device copy of its own) and, behind the operand, four
 *  counters of the caller (the re-sort's: they travel to the host in the same copy as the reduced extents) */
int extentsToReduceOperand(cstone_hip_ctx* ctx, int real_bits, const void* extents, double* devOut, double status,
                           const int* counters)
{
    if (real_bits == 32)
        hipLaunchKernelGGL(reduceOperandKernel<float>, 1, 64, 0, ctx->stream, (const float*)extents, devOut, status, counters);
    else
        hipLaunchKernelGGL(reduceOperandKernel<double>, 1, 64, 0, ctx->stream, (const double*)extents, devOut, status, counters);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int minMaxCoordinates(cstone_hip_ctx* ctx, int real_bits, const void* const* xs, int numArrays, size_t n, double* out)
{
    if (real_bits == 32) return minMaxArrays<float>(ctx, (const float* const*)xs, numArrays, n, out);
    return minMaxArrays<double>(ctx, (const double* const*)xs, numArrays, n, out);
}

namespace
{
} // namespace

} // namespace cship

using namespace cship;

extern "C"
{

int cstone_hip_gather(cstone_hip_ctx* ctx, int elem_bytes, const uint32_t* map, size_t n, const void* src, void* dst)
{
    return permute<true>(ctx, elem_bytes, map, n, src, dst);
}

int cstone_hip_gather_multi(cstone_hip_ctx* ctx, int elem_bytes, const uint32_t* map, size_t n, const void* const* src,
                            void* const* dst, int num_arrays)
{
    if (!ctx) return CSTONE_E_ARG;
    if (num_arrays < 1 || num_arrays > 4 || !src || !dst) return fail(ctx, CSTONE_E_ARG, "gather_multi: 1..4 arrays");
    if (n == 0) return CSTONE_OK;
    if (num_arrays == 1) return cstone_hip_gather(ctx, elem_bytes, map, n, src[0], dst[0]);
    if (!map) return fail(ctx, CSTONE_E_ARG, "gather_multi: null map");
    if (elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16)
        return fail(ctx, CSTONE_E_ARG, "gather_multi: elements of 4, 8 or 16 bytes");
    for (int a = 0; a < num_arrays; ++a)
    {
        if (!src[a] || !dst[a] || (uintptr_t(src[a]) % elem_bytes) || (uintptr_t(dst[a]) % elem_bytes))
            return fail(ctx, CSTONE_E_ARG, "gather_multi: null or misaligned array %d", a);
        for (int b = 0; b < num_arrays; ++b)
            if (dst[a] == src[b]) return fail(ctx, CSTONE_E_ARG, "gather_multi: a destination aliases a source");
    }
    StageTimer timer(ctx, CSTONE_STAGE_GATHER);
#define CSTONE_GM(B)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        if (num_arrays == 2) launchGatherMulti<B, 2>(ctx, map, n, src, dst);                                           \
        else if (num_arrays == 3) launchGatherMulti<B, 3>(ctx, map, n, src, dst);                                      \
        else launchGatherMulti<B, 4>(ctx, map, n, src, dst);                                                           \
    } while (0)
    if (elem_bytes == 4) CSTONE_GM(4);
    else if (elem_bytes == 8) CSTONE_GM(8);
    else CSTONE_GM(16);
#undef CSTONE_GM
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_scatter(cstone_hip_ctx* ctx, int elem_bytes, const uint32_t* map, size_t n, const void* src, void* dst)
{
    return permute<false>(ctx, elem_bytes, map, n, src, dst);
}

int cstone_hip_gather_scatter(cstone_hip_ctx* ctx, int elem_bytes, const uint32_t* map_in, const uint32_t* map_out,
                              size_t n, const void* src, void* dst)
{
    if (!ctx) return CSTONE_E_ARG;
    if (n == 0) return CSTONE_OK;
    if (!map_in || !map_out || !src || !dst) return fail(ctx, CSTONE_E_ARG, "gather_scatter: null array");
    StageTimer timer(ctx, CSTONE_STAGE_GATHER);
    unsigned grid = gridFor(n, 256, 4);
    int natural   = elem_bytes == 12 ? 4 : (elem_bytes == 24 ? 8 : (elem_bytes >= 16 ? 16 : elem_bytes));
    if (natural > 0 && ((uintptr_t(src) % natural) || (uintptr_t(dst) % natural)))
        return fail(ctx, CSTONE_E_ARG, "gather_scatter: arrays must be aligned to %d bytes", natural);
#define CSTONE_GS_CASE(B)                                                                                              \
    case B:                                                                                                            \
        hipLaunchKernelGGL((gatherScatterKernel<Elem<B>, 4>), grid, 256, 0, ctx->stream, map_in, map_out, n,           \
                           (const Elem<B>*)src, (Elem<B>*)dst);                                                        \
        break
    switch (elem_bytes)
    {
        CSTONE_GS_CASE(1);
        CSTONE_GS_CASE(2);
        CSTONE_GS_CASE(4);
        CSTONE_GS_CASE(8);
        CSTONE_GS_CASE(12);
        CSTONE_GS_CASE(16);
        CSTONE_GS_CASE(24);
        CSTONE_GS_CASE(32);
        default: return fail(ctx, CSTONE_E_ARG, "gather_scatter: element size %d unsupported", elem_bytes);
    }
#undef CSTONE_GS_CASE
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_merge_positions(cstone_hip_ctx* ctx, int key_bits, const void* a, size_t na, const void* b, size_t nb,
                               uint32_t offset, uint32_t* pos_a, uint32_t* pos_b)
{
    if (!ctx || (key_bits != 32 && key_bits != 64) || (na && (!a || !pos_a)) || (nb && (!b || !pos_b)))
        return fail(ctx, CSTONE_E_ARG, "merge_positions: bad argument");
    if (na + nb == 0) return CSTONE_OK;
    unsigned grid = gridFor(na + nb, 256);
    if (key_bits == 32)
        hipLaunchKernelGGL(mergePositionsKernel<uint32_t>, grid, 256, 0, ctx->stream, (const uint32_t*)a, na,
                           (const uint32_t*)b, nb, offset, pos_a, pos_b);
    else
        hipLaunchKernelGGL(mergePositionsKernel<uint64_t>, grid, 256, 0, ctx->stream, (const uint64_t*)a, na,
                           (const uint64_t*)b, nb, offset, pos_a, pos_b);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_minmax(cstone_hip_ctx* ctx, int real_bits, const void* x, size_t n, double* out2_host)
{
    if (!ctx || !x || !out2_host) return fail(ctx, CSTONE_E_ARG, "minmax: bad argument");
    if (real_bits == 32) return minMax<float>(ctx, (const float*)x, n, out2_host);
    if (real_bits == 64) return minMax<double>(ctx, (const double*)x, n, out2_host);
    return fail(ctx, CSTONE_E_ARG, "minmax: real_bits %d unsupported", real_bits);
}

int cstone_hip_minmax_arrays(cstone_hip_ctx* ctx, int real_bits, const void* const* arrays, int num_arrays, size_t n,
                             double* out_host)
{
    if (!ctx || !arrays || !out_host || num_arrays < 1 || num_arrays > 3 || (real_bits != 32 && real_bits != 64))
        return fail(ctx, CSTONE_E_ARG, "minmax_arrays: bad argument");
    for (int i = 0; i < num_arrays; ++i)
        if (!arrays[i]) return fail(ctx, CSTONE_E_ARG, "minmax_arrays: null array");
    return minMaxCoordinates(ctx, real_bits, arrays, num_arrays, n, out_host);
}

int cstone_hip_exclusive_scan_u32(cstone_hip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n, uint32_t init)
{
    if (!ctx || (n && (!in || !out))) return fail(ctx, CSTONE_E_ARG, "exclusive_scan: bad argument");
    CS_TRY(arenaReserve(ctx, scanArenaBytes(n)));
    int rc = scanU32(ctx, in, out, n, init, false);
    arenaReset(ctx);
    return rc;
}

int cstone_hip_inclusive_scan_u32(cstone_hip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n)
{
    if (!ctx || (n && (!in || !out))) return fail(ctx, CSTONE_E_ARG, "inclusive_scan: bad argument");
    CS_TRY(arenaReserve(ctx, scanArenaBytes(n)));
    int rc = scanU32(ctx, in, out, n, 0u, true);
    arenaReset(ctx);
    return rc;
}

int cstone_hip_offsets_from_counts_u32(cstone_hip_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n)
{
    if (!ctx || !out || (n && (!in || in == out))) return fail(ctx, CSTONE_E_ARG, "offsets_from_counts: bad argument");
    CS_TRY(arenaReserve(ctx, scanArenaBytes(n)));
    int rc = scanU32(ctx, in, out, n, 0u, false, out + n);
    arenaReset(ctx);
    return rc;
}

int cstone_hip_lower_bound(cstone_hip_ctx* ctx, int key_bits, const void* keys, size_t n, const void* values,
                           int num_values, uint64_t* result)
{
    if (!ctx || (key_bits != 32 && key_bits != 64) || num_values < 0 || (num_values && (!values || !result)) ||
        (n && !keys))
        return fail(ctx, CSTONE_E_ARG, "lower_bound: bad argument");
    if (num_values == 0) return CSTONE_OK;
    unsigned grid = (unsigned(num_values) + 255) / 256;
    if (key_bits == 32)
        hipLaunchKernelGGL(lowerBoundKernel<uint32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*)keys, n,
                           (const uint32_t*)values, num_values, result);
    else
        hipLaunchKernelGGL(lowerBoundKernel<uint64_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t*)keys, n,
                           (const uint64_t*)values, num_values, result);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_lower_bound_u32(cstone_hip_ctx* ctx, int key_bits, const void* keys, size_t n, const void* values,
                               int num_values, uint32_t* result)
{
    if (!ctx || (key_bits != 32 && key_bits != 64) || num_values < 0 || n >= (size_t(1) << 32) ||
        (num_values && (!values || !result)) || (n && !keys))
        return fail(ctx, CSTONE_E_ARG, "lower_bound_u32: bad argument");
    if (num_values == 0) return CSTONE_OK;
    unsigned grid = (unsigned(num_values) + 255) / 256;
    if (key_bits == 32)
        hipLaunchKernelGGL((lowerBoundKernel<uint32_t, uint32_t>), dim3(grid), dim3(256), 0, ctx->stream,
                           (const uint32_t*)keys, n, (const uint32_t*)values, num_values, result);
    else
        hipLaunchKernelGGL((lowerBoundKernel<uint64_t, uint32_t>), dim3(grid), dim3(256), 0, ctx->stream,
                           (const uint64_t*)keys, n, (const uint64_t*)values, num_values, result);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_sequence_u64(cstone_hip_ctx* ctx, uint64_t* out, size_t n, uint64_t init)
{
    if (!ctx || (n && !out)) return fail(ctx, CSTONE_E_ARG, "sequence_u64: bad argument");
    if (n == 0) return CSTONE_OK;
    hipLaunchKernelGGL(sequence64Kernel, gridFor(n, 256), 256, 0, ctx->stream, out, n, init);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_scan_u32_to_u64(cstone_hip_ctx* ctx, const uint32_t* in, uint64_t* out, size_t n, uint64_t init,
                               int inclusive)
{
    if (!ctx || (n && (!in || !out))) return fail(ctx, CSTONE_E_ARG, "scan_u32_to_u64: bad argument");
    if (n == 0) return CSTONE_OK;
    if (static_cast<const void*>(in) == static_cast<const void*>(out))
        return fail(ctx, CSTONE_E_ARG, "scan_u32_to_u64: in place is not possible (the elements grow)");
    const unsigned blocks = unsigned((n + 256 * SCAN64_ITEMS - 1) / (256 * SCAN64_ITEMS));
    CS_TRY(arenaReserve(ctx, alignUp(size_t(blocks) * 8) + 256));
    auto* sums = (uint64_t*)arenaTake(ctx, size_t(blocks) * 8);
    hipLaunchKernelGGL(blockSum64Kernel, blocks, 256, 0, ctx->stream, in, n, sums);
    hipLaunchKernelGGL(scanSums64Kernel, 1, 256, 0, ctx->stream, sums, blocks, init);
    hipLaunchKernelGGL(blockScan64Kernel, blocks, 256, 0, ctx->stream, in, n, sums, out, inclusive != 0);
    arenaReset(ctx);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // extern "C"
