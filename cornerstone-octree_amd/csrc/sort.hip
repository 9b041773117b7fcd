// Stable LSD radix sort of (key, uint32 value) pairs for gfx950 -- "onesweep" formulation.
// Replaces sortByKeyGpu / cub::DeviceRadixSort::SortPairs (R/primitives/primitives_gpu.cu:328-369).
//
// Traffic per pair: one read of the keys for ALL digit histograms (K bytes), then per 8-bit digit
// pass one read and one write of key+value: K + P*2*(K+4) bytes, P = K passes (200 B for 64-bit
// keys, 68 B for 32-bit keys).  HBM-bound; no MFMA.
//
// Pass kernel, one workgroup per tile of TILE = BLOCK*ITEMS pairs:
//   1. tile index from an atomic ticket (earlier tiles are therefore running or done: the
//      look-back below cannot deadlock whatever the dispatch order or residency)
//   2. keys loaded wave-striped (64 consecutive keys per wave instruction)
//   3. stable rank of every key among equal digits of its wave: 8 ballots build the mask of
//      lanes holding the same digit (wave64 "match-any"), popcount below the lane gives the
//      rank, the first lane of each group bumps the wave's private LDS digit counter
//   4. per-digit thread: exclusive prefix over the waves, tile total -> published as AGGREGATE
//      in the tile's 256-entry status row; tile-local exclusive scan over digits
//   5. keys are permuted into tile-sorted order through LDS
//   6. decoupled look-back: thread d sums the status words of preceding tiles for digit d until
//      it meets an INCLUSIVE entry, then publishes its own inclusive prefix.  Status words are
//      32-bit {2-bit state, 30-bit count} granules moved with relaxed agent-scope atomics
//      (value and flag in one word: no fence needed, coherent across the 8 XCD L2s).  Spins are bounded.
//   7. keys, then values, are streamed from LDS to their global slots: consecutive lanes write
//      consecutive addresses within each digit run
#include <algorithm>
#include <utility>

#include "ctx.hpp"
#include "device_keys.hpp"

namespace cship
{

namespace
{

constexpr int RADIX_BITS = 8;
constexpr int RADIX      = 1 << RADIX_BITS;

constexpr uint32_t STATE_AGG = 1u << 30;
constexpr uint32_t STATE_INC = 2u << 30;
constexpr uint32_t COUNT_MASK = (1u << 30) - 1;

constexpr int HIST_BLOCK = 256;

template<class K>
struct SortCfg
{
    static constexpr int BLOCK = 256;
    static constexpr int ITEMS = 16;
    static constexpr int TILE  = BLOCK * ITEMS;
    static constexpr int WAVES = BLOCK / 64;
    static constexpr int PASSES = sizeof(K);
};

struct SortTemp
{
    uint32_t* hist;     // [PASSES][RADIX] counts, then exclusive bases
    uint32_t* tickets;  // [PASSES]
    uint32_t* errors;   // [1]
    uint32_t* status;   // [PASSES][numTiles][RADIX]
};

__host__ __device__ inline size_t headerWords(int passes) { return size_t(passes) * RADIX + 64; }

// -------------------------------------------------------------------------------------------------
// all digit histograms in one read of the keys
// -------------------------------------------------------------------------------------------------
template<class K>
__global__ __launch_bounds__(HIST_BLOCK) void histogramKernel(const K* __restrict__ keys, size_t n,
                                                              uint32_t* __restrict__ hist)
{
    constexpr int P = SortCfg<K>::PASSES;
    __shared__ uint32_t lh[P * RADIX];
    for (int i = threadIdx.x; i < P * RADIX; i += HIST_BLOCK)
        lh[i] = 0;
    __syncthreads();

    constexpr int VEC = 16 / sizeof(K);
    const size_t nVec   = n / VEC;
    const size_t stride = size_t(gridDim.x) * HIST_BLOCK;
    const unsigned lane = threadIdx.x & 63u;

    auto add = [&](K key, bool valid)
    {
#pragma unroll
        for (int p = 0; p < P; ++p)
        {
            unsigned d = unsigned(key >> (p * RADIX_BITS)) & (RADIX - 1);
            // nearly sorted input makes the high digits wave-uniform: one add instead of a 64-way LDS conflict
            uint64_t vmask = __ballot(valid);
            if (vmask == 0) continue;
            unsigned d0   = __builtin_amdgcn_readfirstlane(__shfl(d, __ffsll((unsigned long long)vmask) - 1));
            uint64_t same = __ballot(valid && d == d0);
            if (same == vmask)
            {
                if (lane == unsigned(__ffsll((unsigned long long)vmask) - 1))
                    atomicAdd(&lh[p * RADIX + d0], unsigned(__popcll(vmask)));
            }
            else if (valid) { atomicAdd(&lh[p * RADIX + d], 1u); }
        }
    };

    // every wave walks whole iterations together so the ballots above see all 64 lanes
    size_t iters = (nVec + stride - 1) / stride;
    size_t vi    = size_t(blockIdx.x) * HIST_BLOCK + threadIdx.x;
    for (size_t it = 0; it < iters; ++it, vi += stride)
    {
        bool valid = vi < nVec;
        K v[VEC];
        if (valid) { __builtin_memcpy(v, __builtin_assume_aligned(keys + vi * VEC, 16), 16); }
        else
        {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                v[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            add(v[j], valid);
    }
    // tail elements (n not a multiple of VEC): first wave of block 0
    if (blockIdx.x == 0 && threadIdx.x < 64)
    {
        size_t i   = nVec * VEC + threadIdx.x;
        bool valid = i < n;
        K key      = valid ? keys[i] : K(0);
        add(key, valid);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P * RADIX; i += HIST_BLOCK)
    {
        uint32_t c = lh[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

//! hist[p][:] <- exclusive scan; one block of RADIX threads per pass
__global__ __launch_bounds__(RADIX) void scanHistogramKernel(uint32_t* __restrict__ hist)
{
    __shared__ uint32_t waveSum[RADIX / 64];
    uint32_t* h   = hist + size_t(blockIdx.x) * RADIX;
    unsigned d    = threadIdx.x;
    unsigned lane = d & 63u, w = d >> 6;
    uint32_t v    = h[d];
    uint32_t inc  = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
        uint32_t t = __shfl_up(inc, o);
        if (lane >= unsigned(o)) inc += t;
    }
    if (lane == 63) waveSum[w] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (unsigned i = 0; i < w; ++i)
        off += waveSum[i];
    h[d] = off + inc - v;
}

// -------------------------------------------------------------------------------------------------
// one digit pass
// -------------------------------------------------------------------------------------------------
template<class K>
__global__ __launch_bounds__(SortCfg<K>::BLOCK) void onesweepKernel(const K* __restrict__ keysIn,
                                                                    const uint32_t* __restrict__ valsIn,
                                                                    K* __restrict__ keysOut,
                                                                    uint32_t* __restrict__ valsOut, size_t n,
                                                                    int pass, uint32_t numTiles,
                                                                    const uint32_t* __restrict__ bases,
                                                                    uint32_t* __restrict__ ticket,
                                                                    uint32_t* __restrict__ status,
                                                                    uint32_t* __restrict__ errors)
{
    using Cfg           = SortCfg<K>;
    constexpr int BLOCK = Cfg::BLOCK, ITEMS = Cfg::ITEMS, TILE = Cfg::TILE, WAVES = Cfg::WAVES;

    __shared__ uint32_t waveHist[WAVES * RADIX]; // per-wave digit counts -> exclusive prefix over waves
    __shared__ uint32_t digitStart[RADIX];       // tile-local exclusive scan over digits
    __shared__ uint32_t binOffset[RADIX];        // global slot of the digit's first tile element minus digitStart
    __shared__ uint32_t scanTmp[WAVES];
    __shared__ uint32_t tileShared;
    __shared__ K stage[TILE];

    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int shift = pass * RADIX_BITS;

    if (tid == 0) tileShared = atomicAdd(ticket, 1u);
    for (int i = tid; i < WAVES * RADIX; i += BLOCK)
        waveHist[i] = 0;
    __syncthreads();
    const uint32_t tile = tileShared;
    if (tile >= numTiles) return; // cannot happen (grid == numTiles); keeps a stray launch harmless

    const size_t tileBase = size_t(tile) * TILE;
    const unsigned tileCount = unsigned(min(size_t(TILE), n - tileBase));
    const unsigned segBase   = wave * (64 * ITEMS);

    // ---- 2. load (wave-striped)
    K key[ITEMS];
    uint32_t val[ITEMS]; // fetched now so their latency hides behind the ranking
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        unsigned idx = segBase + r * 64 + lane;
        key[r]       = idx < tileCount ? keysIn[tileBase + idx] : K(~K(0));
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        unsigned idx = segBase + r * 64 + lane;
        val[r]       = idx < tileCount ? valsIn[tileBase + idx] : 0u;
    }

    // ---- 3. stable in-wave ranking
    unsigned rank[ITEMS];
    uint32_t* myHist = waveHist + wave * RADIX;
    const uint64_t ltMask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        unsigned idx = segBase + r * 64 + lane;
        bool valid   = idx < tileCount;
        unsigned d   = unsigned(key[r] >> shift) & (RADIX - 1);
        uint64_t m   = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b)
        {
            bool bit    = (d >> b) & 1u;
            uint64_t v  = __ballot(bit);
            m &= bit ? v : ~v;
        }
        unsigned below  = __popcll(m & ltMask);
        unsigned leader = __ffsll((unsigned long long)m) - 1; // lowest lane of my group (m has my own bit if valid)
        unsigned base   = 0;
        if (valid && below == 0)
        {
            base      = myHist[d];
            myHist[d] = base + unsigned(__popcll(m));
        }
        base    = __shfl(base, valid ? int(leader) : int(lane));
        rank[r] = base + below;
    }
    __syncthreads();

    // ---- 4. per-digit: prefix over waves, publish aggregate, scan over digits
    uint32_t total = 0;
    if (tid < RADIX)
    {
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
        {
            uint32_t c            = waveHist[w * RADIX + tid];
            waveHist[w * RADIX + tid] = total;
            total += c;
        }
        uint32_t word = (tile == 0 ? STATE_INC : STATE_AGG) | total;
        __hip_atomic_store(status + size_t(tile) * RADIX + tid, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        uint32_t inc = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            uint32_t t = __shfl_up(inc, o);
            if (lane >= unsigned(o)) inc += t;
        }
        if (lane == 63) scanTmp[wave] = inc;
        digitStart[tid] = inc - total; // wave-local for now
    }
    __syncthreads();
    if (tid < RADIX)
    {
        uint32_t off = 0;
        for (unsigned w = 0; w < wave; ++w)
            off += scanTmp[w];
        digitStart[tid] += off;
    }
    __syncthreads();

    // ---- 5. permute keys into tile-sorted order through LDS
    unsigned pos[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        unsigned idx = segBase + r * 64 + lane;
        unsigned d   = unsigned(key[r] >> shift) & (RADIX - 1);
        pos[r]       = digitStart[d] + waveHist[wave * RADIX + d] + rank[r];
        if (idx < tileCount) stage[pos[r]] = key[r];
    }

    // ---- 6. decoupled look-back, one thread per digit
    if (tid < RADIX)
    {
        uint32_t exclusive = 0;
        if (tile > 0)
        {
            int64_t t = int64_t(tile) - 1;
            unsigned spins = 0;
            while (t >= 0)
            {
                uint32_t w = __hip_atomic_load(status + size_t(t) * RADIX + tid, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                uint32_t st = w & ~COUNT_MASK;
                if (st == 0)
                {
                    if (++spins > (1u << 24)) // ~ seconds: something is badly wrong, do not hang the GPU
                    {
                        atomicOr(errors, 1u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                exclusive += w & COUNT_MASK;
                if (st == STATE_INC) break;
                --t;
            }
            __hip_atomic_store(status + size_t(tile) * RADIX + tid, STATE_INC | ((exclusive + total) & COUNT_MASK),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        binOffset[tid] = bases[tid] + exclusive - digitStart[tid];
    }
    __syncthreads();

    // ---- 7. stream out keys (remember each slot for the values)
    uint32_t dst[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
    {
        unsigned i = k * BLOCK + tid;
        if (i < tileCount)
        {
            K kk       = stage[i];
            unsigned d = unsigned(kk >> shift) & (RADIX - 1);
            dst[k]     = binOffset[d] + i;
            keysOut[dst[k]] = kk;
        }
    }
    __syncthreads();
    uint32_t* vstage = reinterpret_cast<uint32_t*>(stage);
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        unsigned idx = segBase + r * 64 + lane;
        if (idx < tileCount) vstage[pos[r]] = val[r];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
    {
        unsigned i = k * BLOCK + tid;
        if (i < tileCount) valsOut[dst[k]] = vstage[i];
    }
}

__global__ void sequenceKernel(uint32_t* out, size_t n, uint32_t init)
{
    size_t i = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n && (uintptr_t(out) & 15) == 0)
    {
        uint4 v = make_uint4(init + uint32_t(i), init + uint32_t(i) + 1, init + uint32_t(i) + 2, init + uint32_t(i) + 3);
        *reinterpret_cast<uint4*>(out + i) = v;
    }
    else
    {
        for (size_t j = i; j < min(i + 4, n); ++j)
            out[j] = init + uint32_t(j);
    }
}

template<class K>
size_t sortTempBytes(size_t n)
{
    using Cfg       = SortCfg<K>;
    size_t numTiles = (n + Cfg::TILE - 1) / Cfg::TILE;
    return alignUp((headerWords(Cfg::PASSES) + size_t(Cfg::PASSES) * numTiles * RADIX) * sizeof(uint32_t));
}

template<class K>
int sortPairs(cstone_hip_ctx* ctx, K* keys, uint32_t* vals, size_t n, K* keysAlt, uint32_t* valsAlt, void* temp,
              size_t tempBytes)
{
    using Cfg = SortCfg<K>;
    if (n == 0) return CSTONE_OK;
    if (n >= (size_t(1) << 30)) return fail(ctx, CSTONE_E_ARG, "sort_pairs: n = %zu exceeds 2^30 - 1", n);
    size_t need = sortTempBytes<K>(n);
    if (tempBytes < need) return fail(ctx, CSTONE_E_CAPACITY, "sort_pairs: temp %zu < %zu bytes", tempBytes, need);

    constexpr int P   = Cfg::PASSES;
    uint32_t numTiles = uint32_t((n + Cfg::TILE - 1) / Cfg::TILE);
    auto* words       = (uint32_t*)temp;
    SortTemp t;
    t.hist    = words;
    t.tickets = words + size_t(P) * RADIX;
    t.errors  = (uint32_t*)ctx->devScalars + 63; // sticky; reported by cstone_hip_ctx_sync
    t.status  = words + headerWords(P);

    CS_HIP(ctx, hipMemsetAsync(temp, 0, need, ctx->stream));
    {
        StageTimer timer(ctx, CSTONE_STAGE_SORT_HIST);
        size_t nVec   = n / (16 / sizeof(K));
        unsigned grid = unsigned(std::min<size_t>(size_t(ctx->numCu) * 8, (nVec + HIST_BLOCK - 1) / HIST_BLOCK));
        grid          = std::max(grid, 1u);
        hipLaunchKernelGGL(histogramKernel<K>, grid, HIST_BLOCK, 0, ctx->stream, keys, n, t.hist);
        hipLaunchKernelGGL(scanHistogramKernel, P, RADIX, 0, ctx->stream, t.hist);
    }
    K* kIn         = keys;
    uint32_t* vIn  = vals;
    K* kOut        = keysAlt;
    uint32_t* vOut = valsAlt;
    for (int p = 0; p < P; ++p)
    {
        StageTimer timer(ctx, CSTONE_STAGE_SORT_PASS);
        hipLaunchKernelGGL(onesweepKernel<K>, numTiles, Cfg::BLOCK, 0, ctx->stream, kIn, vIn, kOut, vOut, n, p,
                           numTiles, t.hist + size_t(p) * RADIX, t.tickets + p,
                           t.status + size_t(p) * numTiles * RADIX, t.errors);
        std::swap(kIn, kOut);
        std::swap(vIn, vOut);
    }
    static_assert(P % 2 == 0, "an even number of passes leaves the result in the caller's buffers");
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // namespace

template<class K>
int sortPairsArena(cstone_hip_ctx* ctx, K* keys, uint32_t* vals, size_t n)
{
    size_t tb = sortTempBytes<K>(n);
    CS_TRY(arenaReserve(ctx, alignUp(n * sizeof(K)) + alignUp(n * sizeof(uint32_t)) + tb + 1024));
    K* ka        = (K*)arenaTake(ctx, n * sizeof(K));
    uint32_t* va = (uint32_t*)arenaTake(ctx, n * sizeof(uint32_t));
    void* tmp    = arenaTake(ctx, tb);
    int rc       = sortPairs<K>(ctx, keys, vals, n, ka, va, tmp, tb);
    arenaReset(ctx);
    return rc;
}
template int sortPairsArena<uint32_t>(cstone_hip_ctx*, uint32_t*, uint32_t*, size_t);
template int sortPairsArena<uint64_t>(cstone_hip_ctx*, uint64_t*, uint32_t*, size_t);

} // namespace cship

using namespace cship;

extern "C"
{

size_t cstone_hip_sort_pairs_temp_bytes(int key_bits, size_t n)
{
    if (key_bits == 32) return sortTempBytes<uint32_t>(n);
    if (key_bits == 64) return sortTempBytes<uint64_t>(n);
    return 0;
}

int cstone_hip_sort_pairs(cstone_hip_ctx* ctx, int key_bits, void* keys, uint32_t* values, size_t n, void* keys_alt,
                          uint32_t* values_alt, void* temp, size_t temp_bytes)
{
    if (!ctx || (key_bits != 32 && key_bits != 64)) return fail(ctx, CSTONE_E_ARG, "sort_pairs: bad key_bits");
    if (n && (!keys || !values)) return fail(ctx, CSTONE_E_ARG, "sort_pairs: null array");
    bool own = !keys_alt && !values_alt && !temp;
    if (!own && (!keys_alt || !values_alt || !temp))
        return fail(ctx, CSTONE_E_ARG, "sort_pairs: pass all of keys_alt/values_alt/temp or none");
    if (own)
    {
        return key_bits == 32 ? sortPairsArena<uint32_t>(ctx, (uint32_t*)keys, values, n)
                              : sortPairsArena<uint64_t>(ctx, (uint64_t*)keys, values, n);
    }
    return key_bits == 32
               ? sortPairs<uint32_t>(ctx, (uint32_t*)keys, values, n, (uint32_t*)keys_alt, values_alt, temp, temp_bytes)
               : sortPairs<uint64_t>(ctx, (uint64_t*)keys, values, n, (uint64_t*)keys_alt, values_alt, temp,
                                     temp_bytes);
}

int cstone_hip_sequence_u32(cstone_hip_ctx* ctx, uint32_t* out, size_t n, uint32_t init)
{
    if (!ctx || (n && !out)) return fail(ctx, CSTONE_E_ARG, "sequence_u32: bad argument");
    if (n == 0) return CSTONE_OK;
    hipLaunchKernelGGL(sequenceKernel, gridFor(n, 256, 4), 256, 0, ctx->stream, out, n, init);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // extern "C"
