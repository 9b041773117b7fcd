// Ceiling of the memory system for variants of the digit-pass access pattern (no ranking): tile size, number of
// bins (digit width), workgroups per CU, non-temporal loads/stores.  Run d of tile t of RUN = TILE / BINS elements
// goes to d * (n / BINS) + RUN * t.  hipcc --offload-arch=gfx950 -O3 tools/scatter_variants.hip -o gpurun_out/scatter_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

template<int BLOCK, int ITEMS, int BINS, bool NTLOAD, bool NTSTORE>
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
        size_t i = size_t(tile) * TILE + r * BLOCK + tid;
        k[r] = NTLOAD ? __builtin_nontemporal_load(kin + i) : kin[i];
        v[r] = NTLOAD ? __builtin_nontemporal_load(vin + i) : vin[i];
    }
    constexpr uint32_t RUN = TILE / BINS;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        uint32_t i   = r * BLOCK + tid;
        uint32_t d   = i / RUN;
        uint32_t dst = d * binStride + tile * RUN + (i % RUN);
        if (NTSTORE) __builtin_nontemporal_store(k[r], kout + dst); else kout[dst] = k[r];
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        uint32_t i   = r * BLOCK + tid;
        uint32_t d   = i / RUN;
        uint32_t dst = d * binStride + tile * RUN + (i % RUN);
        if (NTSTORE) __builtin_nontemporal_store(v[r], vout + dst); else vout[dst] = v[r];
    }
}

template<int BLOCK, int ITEMS, int BINS, bool NTL, bool NTS>
void run(const char* name, size_t n, size_t ldsBytes, uint64_t* kin, uint32_t* vin, uint64_t* kout, uint32_t* vout)
{
    constexpr int TILE = BLOCK * ITEMS;
    uint32_t tiles     = uint32_t(n / TILE);
    uint32_t binStride = uint32_t(size_t(tiles) * (TILE / BINS));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pattern<BLOCK, ITEMS, BINS, NTL, NTS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e9f, sum = 0;
    const int reps = 10;
    for (int i = 0; i < reps + 2; ++i)
    {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((pattern<BLOCK, ITEMS, BINS, NTL, NTS>), tiles, BLOCK, ldsBytes, 0, kin, vin, kout, vout, binStride);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (i >= 2) best = ms < best ? ms : best, sum += ms;
    }
    double bytes = 24.0 * double(tiles) * TILE;
    std::printf("%-60s tiles %6u  avg %.4f ms  best %.4f ms  -> %.0f GB/s (avg)\n", name, tiles, sum / reps, best,
                bytes / (sum / reps * 1e-3) / 1e9);
    std::fflush(stdout);
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
    const size_t one = 140 * 1024, two = 75 * 1024;
    run<1024, 16, 256, false, false>("16Ki 256 bins 1024thr 1wg/CU", n, one, kin, vin, kout, vout);
    run<1024, 16, 256, true, false>("16Ki 256 bins 1024thr 1wg/CU ntload", n, one, kin, vin, kout, vout);
    run<1024, 16, 256, false, true>("16Ki 256 bins 1024thr 1wg/CU ntstore", n, one, kin, vin, kout, vout);
    run<1024, 16, 256, true, true>("16Ki 256 bins 1024thr 1wg/CU ntload ntstore", n, one, kin, vin, kout, vout);
    run<512, 32, 256, false, false>("16Ki 256 bins 512thr x32 2wg/CU", n, two, kin, vin, kout, vout);
    run<512, 32, 256, true, true>("16Ki 256 bins 512thr x32 2wg/CU nt nt", n, two, kin, vin, kout, vout);
    run<512, 16, 256, false, false>("8Ki 256 bins 512thr 2wg/CU", n, two, kin, vin, kout, vout);
    run<1024, 16, 512, false, false>("16Ki 512 bins 1024thr 1wg/CU", n, one, kin, vin, kout, vout);
    run<1024, 16, 128, false, false>("16Ki 128 bins 1024thr 1wg/CU", n, one, kin, vin, kout, vout);
    run<1024, 32, 256, false, false>("32Ki 256 bins 1024thr x32 1wg/CU", n, one, kin, vin, kout, vout);
    run<1024, 32, 512, false, false>("32Ki 512 bins 1024thr x32 1wg/CU", n, one, kin, vin, kout, vout);
    run<1024, 24, 256, false, false>("24Ki 256 bins 1024thr x24 1wg/CU", n, one, kin, vin, kout, vout);
    run<1024, 16, 256, false, false>("16Ki 256 bins 1024thr 1wg/CU (again)", n, one, kin, vin, kout, vout);
    return 0;
}
