// SFC key encode for gfx950.  Replaces computeSfcKeysGpu (R/sfc/sfc_gpu.cu:39-57; arithmetic
// R/sfc/sfc.hpp:158-194).  HBM-bound streaming kernel: (3T + 2K) bytes per particle.
//   - SoA x/y/z are read with 16-byte lane loads (2 doubles / 4 floats per lane) when the pointers
//     allow it, keys written the same way
//   - Hilbert keys come from a 24-state transducer table in LDS (device_keys.hpp): one ds_read_u16
//     per level instead of ~25 data-dependent VALU ops; VEC independent chains per lane hide the
//     LDS latency
// THIS FILE MUST BE COMPILED WITH -ffp-contract=off: int(floor(x*m) - lo*m) is evaluated as two
// roundings on the reference's CPU path and must not become an FMA.
#include "ctx.hpp"
#include "device_keys.hpp"

namespace cship
{

template<class K, class T>
__device__ __forceinline__ K gridMorton(T x, T y, T z, T mx, T my, T mz, T sx, T sy, T sz)
{
    constexpr int top = (1u << maxLevel<K>()) - 1;
    int ix = int(floor(x * mx) - sx);
    int iy = int(floor(y * my) - sy);
    int iz = int(floor(z * mz) - sz);
    ix = min(ix, top), iy = min(iy, top), iz = min(iz, top);
    return mortonEncode<K>(unsigned(ix), unsigned(iy), unsigned(iz));
}

template<class K, class T, int VEC, bool HILBERT>
__global__ __launch_bounds__(256) void encodeKernel(const T* __restrict__ x, const T* __restrict__ y,
                                                    const T* __restrict__ z, K* __restrict__ keys, size_t n,
                                                    DBox<T> box, const uint16_t* __restrict__ encTable)
{
    __shared__ uint16_t enc[24 * 8];
    if (HILBERT)
    {
        if (threadIdx.x < 24 * 8) enc[threadIdx.x] = encTable[threadIdx.x];
        __syncthreads();
    }
    constexpr unsigned g = 1u << maxLevel<K>();
    const T mx = g * box.inv[0], my = g * box.inv[1], mz = g * box.inv[2]; // R/sfc/sfc.hpp:188-194
    const T sx = box.lo[0] * mx, sy = box.lo[1] * my, sz = box.lo[2] * mz;

    size_t base = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) * VEC;
    if (base >= n) return;

    if (VEC > 1 && base + VEC <= n)
    {
        T vx[VEC], vy[VEC], vz[VEC];
        K vk[VEC];
        __builtin_memcpy(vx, __builtin_assume_aligned(x + base, sizeof(T) * VEC), sizeof vx);
        __builtin_memcpy(vy, __builtin_assume_aligned(y + base, sizeof(T) * VEC), sizeof vy);
        __builtin_memcpy(vz, __builtin_assume_aligned(z + base, sizeof(T) * VEC), sizeof vz);
        __builtin_memcpy(vk, __builtin_assume_aligned(keys + base, sizeof(K) * VEC), sizeof vk);
        K out[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            out[v] = gridMorton<K, T>(vx[v], vy[v], vz[v], mx, my, mz, sx, sy, sz);
        if (HILBERT)
        {
            // VEC interleaved transducer chains
            K h[VEC];
            unsigned st[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                h[v] = 0, st[v] = 0;
#pragma unroll
            for (int level = int(maxLevel<K>()) - 1; level >= 0; --level)
            {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                {
                    unsigned e = enc[st[v] * 8 + (unsigned(out[v] >> (3 * level)) & 7u)];
                    h[v]       = (h[v] << 3) | K(e & 7u);
                    st[v]      = e >> 3;
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                out[v] = h[v];
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            if (vk[v] == endKey<K>()) out[v] = vk[v]; // particles flagged for removal keep their marker
        __builtin_memcpy(__builtin_assume_aligned(keys + base, sizeof(K) * VEC), out, sizeof out);
    }
    else
    {
        for (size_t i = base; i < min(base + size_t(VEC), n); ++i)
        {
            K old = keys[i];
            K m   = gridMorton<K, T>(x[i], y[i], z[i], mx, my, mz, sx, sy, sz);
            if (HILBERT) m = hilbertFromMorton<K>(m, enc);
            if (old != endKey<K>()) keys[i] = m;
        }
    }
}

template<class K, class T>
int computeKeys(cstone_hip_ctx* ctx, int curve, const T* x, const T* y, const T* z, K* keys, size_t n,
                const cstone_box& hostBox)
{
    if (n == 0) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_ENCODE);
    DBox<T> box   = makeDBox<T>(hostBox);
    auto* enc     = (const uint16_t*)ctx->hilbertTables;
    constexpr int VEC = 16 / sizeof(T);
    // 16-byte lane accesses need aligned bases; keys vectors are VEC*sizeof(K) wide (16 or 32 B)
    bool aligned = (uintptr_t(x) % 16 == 0) && (uintptr_t(y) % 16 == 0) && (uintptr_t(z) % 16 == 0) &&
                   (uintptr_t(keys) % (sizeof(K) * VEC) == 0);
    constexpr unsigned block = 256;
    if (aligned)
    {
        unsigned grid = gridFor(n, block, VEC);
        if (curve == CSTONE_HILBERT)
            hipLaunchKernelGGL((encodeKernel<K, T, VEC, true>), grid, block, 0, ctx->stream, x, y, z, keys, n, box, enc);
        else
            hipLaunchKernelGGL((encodeKernel<K, T, VEC, false>), grid, block, 0, ctx->stream, x, y, z, keys, n, box,
                               enc);
    }
    else
    {
        unsigned grid = gridFor(n, block, 1);
        if (curve == CSTONE_HILBERT)
            hipLaunchKernelGGL((encodeKernel<K, T, 1, true>), grid, block, 0, ctx->stream, x, y, z, keys, n, box, enc);
        else
            hipLaunchKernelGGL((encodeKernel<K, T, 1, false>), grid, block, 0, ctx->stream, x, y, z, keys, n, box, enc);
    }
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // namespace cship

using namespace cship;

extern "C" int cstone_hip_compute_sfc_keys(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* x,
                                           const void* y, const void* z, void* keys, size_t n,
                                           const cstone_box* box_host)
{
    if (!ctx || !box_host || (curve != CSTONE_MORTON && curve != CSTONE_HILBERT))
        return fail(ctx, CSTONE_E_ARG, "compute_sfc_keys: bad argument");
    if (n && (!x || !y || !z || !keys)) return fail(ctx, CSTONE_E_ARG, "compute_sfc_keys: null array");
    if (key_bits == 32 && real_bits == 32)
        return computeKeys<uint32_t, float>(ctx, curve, (const float*)x, (const float*)y, (const float*)z,
                                            (uint32_t*)keys, n, *box_host);
    if (key_bits == 32 && real_bits == 64)
        return computeKeys<uint32_t, double>(ctx, curve, (const double*)x, (const double*)y, (const double*)z,
                                             (uint32_t*)keys, n, *box_host);
    if (key_bits == 64 && real_bits == 32)
        return computeKeys<uint64_t, float>(ctx, curve, (const float*)x, (const float*)y, (const float*)z,
                                            (uint64_t*)keys, n, *box_host);
    if (key_bits == 64 && real_bits == 64)
        return computeKeys<uint64_t, double>(ctx, curve, (const double*)x, (const double*)y, (const double*)z,
                                             (uint64_t*)keys, n, *box_host);
    return fail(ctx, CSTONE_E_ARG, "compute_sfc_keys: key_bits %d / real_bits %d unsupported", key_bits, real_bits);
}
