// Target particle groups on gfx950.  Replaces computeFixedGroups (R/traversal/groups_gpu.cu:41-71) and
// computeGroupSplits (R/traversal/groups_gpu.cu:74-151, kernels R/traversal/groups_gpu.cuh:57-232): the
// [first,last) range of SFC-sorted particles is cut into runs of group_size particles, and each run is cut
// again behind every particle whose successor in the same run is farther away (in unit-cube coordinates)
// than tol_factor x the edge of the smallest leaf cell of the run's particles.
//
// Layout: one WAVE per run; lane l holds particles l, l + 64 (group_size 128) of the run, its successor comes
// over a DPP row shift.  The split bits of a run are one or two 64-bit ballots.  Pass 1 stores them with their
// pop count + 1; one scan later pass 2 writes the group starts straight from the bits (start of the run, then
// the index behind every set bit) -- the reference turns the bits into run lengths with one thread per run and
// scans a second time; the resulting offsets are the same.
//
// Leaves are cubes, so the cube root of the leaf volume in the unit box is 2^-level exactly; it is built from
// the level instead of calling cbrt.  The critical distance follows the reference in taking the leaves of the
// run's FIRST 64 particles only (groups_gpu.cuh:194-201 indexes leafIdx[0] in every round of its loop).
// Compiled with -ffp-contract=off: |dr|^2 is evaluated as x^2 + (y^2 + z^2) without FMA, like the CPU oracle.
#include <algorithm>

#include "ctx.hpp"
#include "device_keys.hpp"
#include "scan.hpp"

namespace cship
{

namespace
{

constexpr int GR_BLOCK = 256;
constexpr int GR_WAVES = GR_BLOCK / 64;

__global__ __launch_bounds__(256) void fixedGroupsKernel(uint32_t first, uint32_t last, uint32_t groupSize,
                                                         uint32_t numGroups, uint32_t* __restrict__ groups)
{
    uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g < numGroups) groups[g] = first + g * groupSize;
    if (g == numGroups) groups[g] = last;
}

//! index of the leaf whose particle range holds body b: upper_bound(layout, layout + numLeaves, b) - 1
__device__ __forceinline__ int leafOfBody(const uint32_t* __restrict__ layout, int numLeaves, uint32_t b)
{
    int lo = 0, hi = numLeaves;
    while (lo < hi)
    {
        int mid = (lo + hi) >> 1;
        if (layout[mid] <= b) lo = mid + 1;
        else hi = mid;
    }
    return lo - 1;
}

template<class T>
__device__ __forceinline__ T shiftDown1(T v)
{
    return __shfl_down(v, 1);
}

template<class K, class T, int NWT>
__global__ __launch_bounds__(GR_BLOCK) void groupSplitsKernel(uint32_t first, uint32_t last, const T* __restrict__ x,
                                                              const T* __restrict__ y, const T* __restrict__ z,
                                                              const K* __restrict__ leaves, int numLeaves,
                                                              const uint32_t* __restrict__ layout, T ilx, T ily, T ilz,
                                                              float tolFactor, uint32_t numRuns,
                                                              uint64_t* __restrict__ masks,
                                                              uint32_t* __restrict__ numSplits)
{
    constexpr uint32_t G = 64u * NWT;
    const unsigned lane  = threadIdx.x & 63u;
    const uint32_t run   = blockIdx.x * GR_WAVES + (threadIdx.x >> 6);
    if (run >= numRuns) return;

    uint32_t body[NWT];
#pragma unroll
    for (int k = 0; k < NWT; ++k)
        body[k] = min(first + run * G + k * 64u + lane, last - 1);

    // deepest leaf among the first 64 particles of the run
    int leaf        = leafOfBody(layout, numLeaves, body[0]);
    unsigned level  = leaf >= 0 ? levelOfSpan<K>(leaves[leaf + 1] - leaves[leaf]) : 0u;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1)
        level = max(level, unsigned(__shfl_xor(int(level), s)));
    // cbrt(min(8 sx sy sz, 1)) with s = half the leaf edge in the unit box = 2^-level; times the tolerance in T
    const T distCrit   = T(ldexpf(1.0f, -int(level))) * T(tolFactor);
    const T distCritSq = distCrit * distCrit;

    T px[NWT], py[NWT], pz[NWT];
#pragma unroll
    for (int k = 0; k < NWT; ++k)
    {
        px[k] = x[body[k]] * ilx;
        py[k] = y[body[k]] * ily;
        pz[k] = z[body[k]] * ilz;
    }
    uint32_t count = 1;
#pragma unroll
    for (int k = 0; k < NWT; ++k)
    {
        // successor: the next lane; behind lane 63 the first particle of the next 64, behind the last of the run itself
        T nx = shiftDown1(px[k]), ny = shiftDown1(py[k]), nz = shiftDown1(pz[k]);
        if (k + 1 < NWT)
        {
            T sx = __shfl(px[k + 1 < NWT ? k + 1 : k], 0), sy = __shfl(py[k + 1 < NWT ? k + 1 : k], 0),
              sz = __shfl(pz[k + 1 < NWT ? k + 1 : k], 0);
            if (lane == 63) nx = sx, ny = sy, nz = sz;
        }
        T dx = nx - px[k], dy = ny - py[k], dz = nz - pz[k];
        bool split    = dx * dx + (dy * dy + dz * dz) > distCritSq;
        uint64_t bits = __ballot(split);
        count += __popcll(bits);
        if (lane == 0) masks[size_t(run) * NWT + k] = bits;
    }
    if (lane == 0) numSplits[run] = count;
}

template<int NWT>
__global__ __launch_bounds__(GR_BLOCK) void fillGroupsKernel(uint32_t first, uint32_t last, uint32_t numRuns,
                                                             const uint64_t* __restrict__ masks,
                                                             const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ total,
                                                             uint32_t capacity, uint32_t* __restrict__ groups)
{
    constexpr uint32_t G = 64u * NWT;
    const unsigned lane  = threadIdx.x & 63u;
    const uint32_t run   = blockIdx.x * GR_WAVES + (threadIdx.x >> 6);
    if (run >= numRuns) return;
    const uint32_t numGroups = *total;
    if (numGroups + 1 > capacity) return; // the host reports it
    uint32_t slot = offsets[run];
    if (lane == 0) groups[slot] = first + run * G;
    ++slot;
#pragma unroll
    for (int k = 0; k < NWT; ++k)
    {
        const uint64_t bits = masks[size_t(run) * NWT + k];
        if ((bits >> lane) & 1ull)
            groups[slot + __popcll(bits & ((1ull << lane) - 1ull))] = first + run * G + k * 64u + lane + 1;
        slot += __popcll(bits);
    }
    if (run + 1 == numRuns && lane == 0) groups[numGroups] = last;
}

template<class K, class T>
int groupSplits(cstone_hip_ctx* ctx, uint32_t first, uint32_t last, const T* x, const T* y, const T* z, const K* leaves,
                int numLeaves, const uint32_t* layout, const cstone_box& boxHost, uint32_t groupSize, float tolFactor,
                uint32_t* groups, size_t capacity, uint32_t* numGroupsOut)
{
    const uint32_t numRuns = (last - first + groupSize - 1) / groupSize;
    const int nwt          = int(groupSize / 64);
    const size_t maskBytes = alignUp(size_t(numRuns) * nwt * sizeof(uint64_t));
    const size_t cntBytes  = alignUp(size_t(numRuns) * sizeof(uint32_t));
    CS_TRY(arenaReserve(ctx, maskBytes + cntBytes + scanArenaBytes(numRuns) + 1024));
    auto* masks  = (uint64_t*)arenaTake(ctx, maskBytes);
    auto* counts = (uint32_t*)arenaTake(ctx, cntBytes);
    auto* total  = (uint32_t*)(ctx->devScalars + 8);
    DBox<T> box  = makeDBox<T>(boxHost);
    unsigned grid = gridFor(numRuns, GR_WAVES);
    int rc        = CSTONE_OK;
    {
        StageTimer timer(ctx, CSTONE_STAGE_NEIGHBORS);
        if (nwt == 1)
            hipLaunchKernelGGL((groupSplitsKernel<K, T, 1>), grid, GR_BLOCK, 0, ctx->stream, first, last, x, y, z, leaves,
                               numLeaves, layout, box.inv[0], box.inv[1], box.inv[2], tolFactor, numRuns, masks, counts);
        else
            hipLaunchKernelGGL((groupSplitsKernel<K, T, 2>), grid, GR_BLOCK, 0, ctx->stream, first, last, x, y, z, leaves,
                               numLeaves, layout, box.inv[0], box.inv[1], box.inv[2], tolFactor, numRuns, masks, counts);
        rc = scanU32(ctx, counts, counts, numRuns, 0u, false, total);
        if (rc == CSTONE_OK)
        {
            if (nwt == 1)
                hipLaunchKernelGGL(fillGroupsKernel<1>, grid, GR_BLOCK, 0, ctx->stream, first, last, numRuns, masks,
                                   counts, total, uint32_t(std::min<size_t>(capacity, 0xffffffffu)), groups);
            else
                hipLaunchKernelGGL(fillGroupsKernel<2>, grid, GR_BLOCK, 0, ctx->stream, first, last, numRuns, masks,
                                   counts, total, uint32_t(std::min<size_t>(capacity, 0xffffffffu)), groups);
            hipError_t e = hipMemcpyAsync(ctx->hostScalars + 8, total, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = fail(ctx, CSTONE_E_HIP, "compute_group_splits: %s", hipGetErrorString(e));
        }
    }
    arenaReset(ctx);
    CS_TRY(rc);
    const uint32_t numGroups = uint32_t(ctx->hostScalars[8]);
    *numGroupsOut            = numGroups;
    if (size_t(numGroups) + 1 > capacity)
        return fail(ctx, CSTONE_E_CAPACITY, "compute_group_splits: %u groups need %u entries, capacity %zu", numGroups,
                    numGroups + 1, capacity);
    return CSTONE_OK;
}

} // namespace

} // namespace cship

using namespace cship;

extern "C" int cstone_hip_compute_fixed_groups(cstone_hip_ctx* ctx, uint32_t first, uint32_t last, uint32_t group_size,
                                               uint32_t* groups, uint32_t* num_groups)
{
    if (!ctx || !groups || !num_groups || last < first || group_size == 0)
        return fail(ctx, CSTONE_E_ARG, "compute_fixed_groups: bad argument");
    const uint32_t n = (last - first + group_size - 1) / group_size;
    *num_groups      = n;
    hipLaunchKernelGGL(fixedGroupsKernel, gridFor(size_t(n) + 1, 256), 256, 0, ctx->stream, first, last, group_size, n,
                       groups);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

extern "C" int cstone_hip_compute_group_splits(cstone_hip_ctx* ctx, int key_bits, int real_bits, uint32_t first,
                                               uint32_t last, const void* x, const void* y, const void* z,
                                               const void* leaves, int num_leaves, const uint32_t* layout,
                                               const cstone_box* box_host, uint32_t group_size, float tol_factor,
                                               uint32_t* groups, size_t capacity, uint32_t* num_groups)
{
    if (!ctx || !x || !y || !z || !leaves || !layout || !box_host || !groups || !num_groups || last < first ||
        num_leaves < 1)
        return fail(ctx, CSTONE_E_ARG, "compute_group_splits: bad argument");
    // the reference accepts one or two times its warp size (64 on AMD hardware) and throws otherwise
    if (group_size != 64 && group_size != 128)
        return fail(ctx, CSTONE_E_ARG, "compute_group_splits: unsupported spatial group size %u", group_size);
    if ((key_bits != 32 && key_bits != 64) || (real_bits != 32 && real_bits != 64))
        return fail(ctx, CSTONE_E_ARG, "compute_group_splits: key_bits %d / real_bits %d unsupported", key_bits, real_bits);
    if (last == first)
    {
        if (capacity < 1) return fail(ctx, CSTONE_E_CAPACITY, "compute_group_splits: capacity 0");
        *num_groups = 0;
        CS_HIP(ctx, hipMemcpyAsync(groups, &last, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return CSTONE_OK;
    }
#define CSTONE_GROUPS_CASE(K, T)                                                                                       \
    return groupSplits<K, T>(ctx, first, last, (const T*)x, (const T*)y, (const T*)z, (const K*)leaves, num_leaves,    \
                             layout, *box_host, group_size, tol_factor, groups, capacity, num_groups)
    if (key_bits == 32)
    {
        if (real_bits == 32) CSTONE_GROUPS_CASE(uint32_t, float);
        CSTONE_GROUPS_CASE(uint32_t, double);
    }
    if (real_bits == 32) CSTONE_GROUPS_CASE(uint64_t, float);
    CSTONE_GROUPS_CASE(uint64_t, double);
#undef CSTONE_GROUPS_CASE
}
