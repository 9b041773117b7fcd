// Experiment: does ds_add_rtn_u32 hand out return values in ascending lane order among the lanes of ONE wave
// instruction that hit the same LDS address?  (needed for a stable rank by LDS atomics instead of ballots)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const unsigned* digits, unsigned* ranks, int items)
{
    __shared__ unsigned hist[16][256];
    unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 256; i += blockDim.x)
        (&hist[0][0])[i] = 0;
    __syncthreads();
    size_t base = (size_t(blockIdx.x) * blockDim.x + wave * 64) * items;
    for (int r = 0; r < items; ++r)
    {
        unsigned d = digits[base + r * 64 + lane];
        ranks[base + r * 64 + lane] = atomicAdd(&hist[wave][d], 1u);
    }
}

int main()
{
    const int blocks = 2048, threads = 1024, items = 16;
    size_t n = size_t(blocks) * threads * items;
    std::vector<unsigned> h(n), r(n);
    unsigned *dd, *dr;
    hipMalloc(&dd, n * 4);
    hipMalloc(&dr, n * 4);
    long bad = 0;
    for (int mode = 0; mode < 5; ++mode)
    {
        srand(mode + 1);
        for (size_t i = 0; i < n; ++i)
        {
            unsigned v = rand();
            h[i] = mode == 0 ? 7u : mode == 1 ? v & 255u : mode == 2 ? v & 3u : mode == 3 ? (v & 1u) * 32u + (v >> 8 & 1u) : (i & 63u) >> 2;
        }
        hipMemcpy(dd, h.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, blocks, threads, 0, 0, dd, dr, items);
        hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
        long modeBad = 0;
        // reference: stable rank in (r, lane) order per wave
        for (size_t w = 0; w < n / (64 * items); ++w)
        {
            unsigned cnt[256] = {0};
            for (int k = 0; k < 64 * items; ++k)
            {
                size_t i = w * 64 * items + k;
                if (r[i] != cnt[h[i]]++) ++modeBad;
            }
        }
        printf("mode %d: %ld of %zu ranks differ from the stable lane-ascending rank\n", mode, modeBad, n);
        bad += modeBad;
    }
    printf(bad ? "LDS_ORDER: NOT lane-ascending\n" : "LDS_ORDER: lane-ascending in every trial\n");
    return 0;
}
