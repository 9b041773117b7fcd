// What the memory system gives for the access pattern of one onesweep digit pass, with no ranking at all:
// every workgroup reads a tile of 16 Ki keys (8 B) and values (4 B) coalesced and writes them as 256 runs of 64
// elements, run d of tile t at d * (n / 256) + 64 t -- where a pass over uniformly distributed digits puts them.
//   hipcc --offload-arch=gfx950 -O3 tools/scatter_pattern.hip -o gpurun_out/scatter_pattern && gpurun_out/scatter_pattern
// Variants: LDS bytes per workgroup (occupancy 1 or more per CU), threads per workgroup.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                                       \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess)                                                                                           \
        {                                                                                                              \
            std::printf("%s: %s\n", #x, hipGetErrorString(e));                                                         \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

template<int BLOCK, int ITEMS, bool THROUGH_LDS>
__global__ __launch_bounds__(BLOCK) void pattern(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                 uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                 uint32_t binStride)
{
    extern __shared__ uint64_t stage[];
    constexpr int TILE  = BLOCK * ITEMS;
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    uint64_t k[ITEMS];
    uint32_t v[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        k[r] = kin[size_t(tile) * TILE + r * BLOCK + tid];
        v[r] = vin[size_t(tile) * TILE + r * BLOCK + tid];
    }
    if (THROUGH_LDS)
    {
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
            stage[r * BLOCK + tid] = k[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
            k[r] = stage[r * BLOCK + (tid ^ 64)];
    }
    constexpr uint32_t RUN = TILE / 256; // elements per digit run of this tile
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        uint32_t i   = r * BLOCK + tid;
        uint32_t d   = i / RUN;
        uint32_t dst = d * binStride + tile * RUN + (i % RUN);
        kout[dst]    = k[r];
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        uint32_t i   = r * BLOCK + tid;
        uint32_t d   = i / RUN;
        uint32_t dst = d * binStride + tile * RUN + (i % RUN);
        vout[dst]    = v[r];
    }
}

template<int BLOCK, int ITEMS, bool LDS>
void run(const char* name, size_t n, size_t ldsBytes, uint64_t* kin, uint32_t* vin, uint64_t* kout, uint32_t* vout)
{
    constexpr int TILE = BLOCK * ITEMS;
    uint32_t tiles     = uint32_t(n / TILE);
    uint32_t binStride = uint32_t(size_t(tiles) * (TILE / 256));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pattern<BLOCK, ITEMS, LDS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f, sum = 0;
    const int reps = 12;
    for (int i = 0; i < reps + 2; ++i)
    {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((pattern<BLOCK, ITEMS, LDS>), tiles, BLOCK, ldsBytes, 0, kin, vin, kout, vout, binStride);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (i >= 2) best = ms < best ? ms : best, sum += ms;
    }
    double bytes = 24.0 * double(tiles) * TILE;
    std::printf("%-44s tiles %6u  avg %.4f ms  best %.4f ms  -> %.0f GB/s (avg)\n", name, tiles, sum / reps, best,
                bytes / (sum / reps * 1e-3) / 1e9);
}

int main()
{
    size_t n = 100000000;
    uint64_t *kin, *kout;
    uint32_t *vin, *vout;
    CHECK(hipMalloc(&kin, n * 8));
    CHECK(hipMalloc(&kout, n * 8));
    CHECK(hipMalloc(&vin, n * 4));
    CHECK(hipMalloc(&vout, n * 4));
    CHECK(hipMemset(kin, 1, n * 8));
    CHECK(hipMemset(vin, 2, n * 4));
    run<1024, 16, false>("16Ki tile, 1024 thr, no LDS limit", n, 0, kin, vin, kout, vout);
    run<1024, 16, false>("16Ki tile, 1024 thr, 1 workgroup/CU", n, 140 * 1024, kin, vin, kout, vout);
    run<1024, 16, true>("16Ki tile, 1024 thr, 1 wg/CU, keys via LDS", n, 140 * 1024, kin, vin, kout, vout);
    run<512, 16, false>("8Ki tile, 512 thr, 2 workgroups/CU", n, 70 * 1024, kin, vin, kout, vout);
    run<256, 16, false>("4Ki tile, 256 thr, no LDS limit", n, 0, kin, vin, kout, vout);
    // plain copy for reference: same bytes, no scatter
    {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        float sum = 0;
        for (int i = 0; i < 12; ++i)
        {
            CHECK(hipEventRecord(a));
            CHECK(hipMemcpyAsync(kout, kin, n * 8, hipMemcpyDeviceToDevice, 0));
            CHECK(hipMemcpyAsync(vout, vin, n * 4, hipMemcpyDeviceToDevice, 0));
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (i >= 2) sum += ms;
        }
        std::printf("%-44s avg %.4f ms -> %.0f GB/s\n", "hipMemcpy D2D of the same 2.4 GB", sum / 10,
                    24.0 * n / (sum / 10 * 1e-3) / 1e9);
    }
    return 0;
}
