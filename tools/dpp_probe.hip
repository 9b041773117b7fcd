// which lane does a DPP / permlane-swap operation read from?  Prints, for every candidate control word, the source lane
// of each of the 64 lanes (input: the lane's own number).  gfx950.  Build: hipcc --offload-arch=gfx950 tools/dpp_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template<int CTRL, int BANK>
__device__ unsigned dpp(unsigned old, unsigned v)
{
    return unsigned(__builtin_amdgcn_update_dpp(int(old), int(v), CTRL, 0xF, BANK, false));
}

__global__ void probe(unsigned* out)
{
    const unsigned lane = threadIdx.x;
    unsigned v          = lane;
    int r               = 0;
    out[64 * r++ + lane] = dpp<0xB1, 0xF>(99, v);  // quad_perm [1,0,3,2]
    out[64 * r++ + lane] = dpp<0x4E, 0xF>(99, v);  // quad_perm [2,3,0,1]
    out[64 * r++ + lane] = dpp<0x1B, 0xF>(99, v);  // quad_perm [3,2,1,0]
    out[64 * r++ + lane] = dpp<0x141, 0xF>(99, v); // row_half_mirror
    out[64 * r++ + lane] = dpp<0x140, 0xF>(99, v); // row_mirror
    out[64 * r++ + lane] = dpp<0x104, 0xF>(99, v); // row_shl:4
    out[64 * r++ + lane] = dpp<0x114, 0xF>(99, v); // row_shr:4
    out[64 * r++ + lane] = dpp<0x104, 0x5>(99, v); // row_shl:4, banks 0 and 2
    out[64 * r++ + lane] = dpp<0x114, 0xA>(99, v); // row_shr:4, banks 1 and 3
    out[64 * r++ + lane] = dpp<0x108, 0x3>(99, v); // row_shl:8, banks 0 and 1
    out[64 * r++ + lane] = dpp<0x118, 0xC>(99, v); // row_shr:8, banks 2 and 3
    {
        auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        out[64 * r++ + lane] = p[0];
        out[64 * r++ + lane] = p[1];
    }
    {
        auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        out[64 * r++ + lane] = p[0];
        out[64 * r++ + lane] = p[1];
    }
}

int main()
{
    unsigned* d;
    const int rows = 15;
    hipMalloc(&d, rows * 64 * 4);
    hipLaunchKernelGGL(probe, 1, 64, 0, 0, d);
    std::vector<unsigned> h(rows * 64);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    const char* names[] = {"quad_perm[1,0,3,2]", "quad_perm[2,3,0,1]", "quad_perm[3,2,1,0]", "row_half_mirror", "row_mirror",
                           "row_shl:4", "row_shr:4", "row_shl:4 bank 0x5", "row_shr:4 bank 0xA", "row_shl:8 bank 0x3",
                           "row_shr:8 bank 0xC", "permlane16_swap[0]", "permlane16_swap[1]", "permlane32_swap[0]",
                           "permlane32_swap[1]"};
    for (int r = 0; r < rows; ++r)
    {
        std::printf("%-22s", names[r]);
        for (int l = 0; l < 64; ++l)
            std::printf(" %2u", h[r * 64 + l]);
        std::printf("\n");
    }
    return 0;
}
