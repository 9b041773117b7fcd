// Stable LSD radix sort of (key, uint32 value) pairs for gfx950 -- "onesweep" formulation.
// Replaces sortByKeyGpu / cub::DeviceRadixSort::SortPairs (R/primitives/primitives_gpu.cu:328-369).
//
// Traffic per pair: one read of the keys for ALL digit histograms (K bytes), then per 8-bit digit
// pass one read and one write of key+value: K + P*2*(K+4) bytes, P = K passes (200 B for 64-bit
// keys, 68 B for 32-bit keys).  HBM-bound; no MFMA.
//
// Pass kernel, one workgroup per tile of TILE = BLOCK*16 pairs (BLOCK = 1024 for large inputs: 16 Ki
// pairs per tile, the whole 160 KB LDS of a CU; BLOCK = 256 for small inputs):
//   1. tile index from an atomic ticket (earlier tiles are therefore running or done: the
//      look-back below cannot deadlock whatever the dispatch order or residency)
//   2. keys loaded wave-striped (64 consecutive keys per wave instruction)
//   3. stable rank of every key among equal digits of its wave: a returning LDS atomic on the wave's
//      private digit counter (ds_add_rtn serves the lanes of one instruction in ascending lane order on
//      gfx950 -- probed once per context; a wave's LDS instructions execute in order)
//   4. digit threads: tile totals, tile-local exclusive scan over digits; wave 0 publishes the tile's
//      AGGREGATE row
//   5. keys are permuted into tile-sorted order through LDS
//   6. decoupled look-back by 4 waves, one digit quarter each, 16 status rows per round trip (see
//      quarterLookBack), then the INCLUSIVE row is published
//   7. keys, then values, are streamed from LDS to their global slots: consecutive lanes write
//      consecutive addresses within each digit run
// The last, partial tile needs no look-back; the workgroup that draws the ticket behind the last full tile takes it
// (phase-trace builds keep it in a one-workgroup kernel of its own).
// Measured phase budget of a 16 Ki tile (CSTONE_SORT_TRACE build, tools/sort_trace.py, 1e8 random 64-bit pairs):
// ticket 0.7 us, key load 4.6-5.2, rank 3.6-4.1, digit scans 1.2, permute 1.1, look-back 4.5, key store 2.1,
// value stage + store 2.9: about 23 us per tile and CU, i.e. the kernel is bound by the tile's serial phases with
// ONE resident workgroup per CU (LDS), not by HBM.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <utility>

#include "ctx.hpp"
#include "device_keys.hpp"

namespace cship
{

namespace
{

constexpr int RADIX_BITS = 8;
constexpr int RADIX      = 1 << RADIX_BITS;

constexpr uint32_t STATE_AGG  = 1u << 30;
constexpr uint32_t STATE_INC  = 2u << 30;
constexpr uint32_t COUNT_MASK = (1u << 30) - 1;

constexpr int HIST_BLOCK   = 256;
constexpr int SMALL_BLOCK  = 256;  // 4 Ki pairs per tile
#ifndef CSTONE_LARGE_BLOCK
#define CSTONE_LARGE_BLOCK 1024
#endif
// tuning switches of the pass kernel (tools/build_variant.sh); the defaults are what measured best on MI355X:
//   CSTONE_SORT_EARLY_LB  look-back started right behind the ranking barrier, next to the digit scan (r2: 3 % SLOWER)
//   CSTONE_SORT_NO_TICKET tile = blockIdx.x instead of an atomic ticket (relies on in-order dispatch per XCD)
#ifndef CSTONE_SORT_EARLY_LB
#define CSTONE_SORT_EARLY_LB 0
#endif
#ifndef CSTONE_SORT_NO_TICKET
#define CSTONE_SORT_NO_TICKET 0
#endif
constexpr int LARGE_BLOCK  = CSTONE_LARGE_BLOCK; // 16 Ki pairs per tile

// CSTONE_SORT_TRACE (tuning builds only, tools/sort_trace.py): wave 0 of every tile records wall-clock stamps
#ifdef CSTONE_SORT_TRACE
constexpr int TRACE_SLOTS = 16;
__device__ uint64_t* g_sortTrace;
#define SORT_TRACE(slot)                                                                                               \
    if (g_sortTrace && threadIdx.x == 0) g_sortTrace[(size_t(traceRow)) * TRACE_SLOTS + (slot)] = wall_clock64();
#define SORT_TRACE_VAL(slot, v)                                                                                        \
    if (g_sortTrace && threadIdx.x == 0) g_sortTrace[(size_t(traceRow)) * TRACE_SLOTS + (slot)] = (v);
#else
#define SORT_TRACE(slot)
#define SORT_TRACE_VAL(slot, v)
#endif

template<class K, int BLOCK_>
struct SortCfg
{
    static constexpr int BLOCK  = BLOCK_;
    static constexpr int ITEMS  = 16;
    static constexpr int TILE   = BLOCK * ITEMS;
    static constexpr int WAVES  = BLOCK / 64;
    static constexpr int PASSES = sizeof(K);
};

struct SortTemp
{
    uint32_t* hist;    // [PASSES][RADIX] counts, then exclusive bases
    uint32_t* tickets; // [PASSES]
    uint32_t* errors;  // [1]
    uint32_t* status;  // [PASSES][numTiles][RADIX]
};

__host__ __device__ inline size_t headerWords(int passes) { return size_t(passes) * RADIX + 64; }

// -------------------------------------------------------------------------------------------------
// all digit histograms in one read of the keys
// -------------------------------------------------------------------------------------------------
template<class K>
__global__ __launch_bounds__(HIST_BLOCK) void histogramKernel(const K* __restrict__ keys, size_t n,
                                                              uint32_t* __restrict__ hist)
{
    constexpr int P = int(sizeof(K));
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

//! mask (two 32-bit halves) of the lanes of this wave whose digit equals mine: wave64 match-any by 8 ballots
__device__ __forceinline__ void matchDigit(unsigned d, uint32_t& mlo, uint32_t& mhi)
{
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b)
    {
        int32_t sb = int32_t(d << (31 - b)) >> 31; // 0 or -1
        uint64_t v = __ballot(sb != 0);
        mlo &= ~(uint32_t(v) ^ uint32_t(sb));
        mhi &= ~(uint32_t(v >> 32) ^ uint32_t(sb));
    }
}

/*! one-time probe of the property the ranking relies on: returning LDS atomics of one wave instruction are served in
 *  ascending lane order.  Compares against the ballot-based stable rank for random, few-valued and uniform digits. */
__global__ __launch_bounds__(1024) void ldsOrderProbeKernel(uint32_t* __restrict__ mismatches)
{
    __shared__ uint32_t hist[16 * RADIX];
    __shared__ uint32_t ref[16 * RADIX];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned bad = 0;
    for (int mode = 0; mode < 4; ++mode)
    {
        for (int i = tid; i < 16 * RADIX; i += 1024)
            hist[i] = ref[i] = 0;
        __syncthreads();
        for (int r = 0; r < 32; ++r)
        {
            unsigned hsh = (tid * 2654435761u) ^ ((r + 1) * 40503u * (mode + 7u));
            hsh ^= hsh >> 13;
            hsh *= 0x5bd1e995u;
            hsh ^= hsh >> 15;
            unsigned d   = mode == 0 ? hsh & 255u : mode == 1 ? hsh & 3u : mode == 2 ? (lane >> 2) + (r & 1) : 9u;
            unsigned got = atomicAdd(&hist[wave * RADIX + d], 1u);
            uint32_t mlo = ~0u, mhi = ~0u;
            matchDigit(d, mlo, mhi);
            unsigned below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
            unsigned base  = ref[wave * RADIX + d];
            if (below == 0) ref[wave * RADIX + d] = base + unsigned(__popc(mlo) + __popc(mhi));
            bad += (got != base + below);
        }
        __syncthreads();
    }
    if (bad) atomicAdd(mismatches, bad);
}

//! workgroup barrier that orders LDS traffic only: __syncthreads() would also wait for every outstanding global
//! load and store (vmcnt(0)) and thereby serialise the value loads and the draining stores
__device__ __forceinline__ void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// streaming hints of the pass: every pair is read once and written once per pass (tuning switches)
template<class V>
__device__ __forceinline__ V streamLoad(const V* p)
{
#ifdef CSTONE_SORT_NT_LOAD
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template<class V>
__device__ __forceinline__ void streamStore(V* p, V v)
{
#ifdef CSTONE_SORT_NT_STORE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

//! 16-byte agent-scope (sc1) row accesses for the look-back: one wave moves a whole 256-digit status row with a
//! single instruction (dword-sized sc1 accesses cost one fabric transaction each)
__device__ __forceinline__ void storeRowSc1(uint32_t* p, u32x4 v)
{
    // The wait states are part of the statement: a store of more than 64 bits reads its data registers over several
    // cycles, and an instruction that overwrites them must keep its distance (the compiler's hazard recogniser inserts
    // the s_nop for stores it emits itself, but it does not look inside inline assembly).  Found the hard way: with a
    // v_mbcnt writing the first data register right behind the store, dword 0 of every 16th lane went out as zero.
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 4" ::"v"(p), "v"(v) : "memory");
}

template<class K, int BLOCK>
struct alignas(16) SortSmem
{
    using Cfg = SortCfg<K, BLOCK>;
    uint32_t waveHist[Cfg::WAVES * RADIX]; // per-wave digit counts, later: tile-local slot of (wave, digit)
    uint32_t digitStart[RADIX];            // tile-local exclusive scan over digits
    uint32_t binOffset[RADIX];             // global slot of a digit's first tile element minus digitStart
    uint32_t total[RADIX];                 // digit counts of the tile
    uint32_t scanTmp[RADIX / 64];
    uint32_t tileShared[4];
    K stage[Cfg::TILE];
};

//! sixteen quarter-rows in flight: 4 instructions, each lane group of 16 lanes addresses its own row.  The wait is
//! part of the statement because the compiler does not track loads issued from inline asm.
__device__ __forceinline__ void loadQuarterRows32Sc1(const uint32_t* const (&p)[8], u32x4 (&w)[8])
{
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\t"
                 "global_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\t"
                 "global_load_dwordx4 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
                 : "memory");
}

__device__ __forceinline__ void loadQuarterRows16Sc1(const uint32_t* p0, const uint32_t* p1, const uint32_t* p2,
                                                     const uint32_t* p3, u32x4 (&w)[4])
{
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
}

#ifndef CSTONE_LB_INSTR
#define CSTONE_LB_INSTR 4
#endif
constexpr int LB_WAVES = 4;                   // look-back waves: wave w owns the digits 64w .. 64w+63
constexpr int LB_INSTR = CSTONE_LB_INSTR;     // 4 or 8 load instructions per round
constexpr int LB_ROWS  = 4 * LB_INSTR;        // status rows per look-back round
constexpr unsigned LB_NONE = 255;

__device__ __forceinline__ void loadQuarterRows(const uint32_t* col, int32_t t, int32_t g, u32x4 (&w)[4])
{
    loadQuarterRows16Sc1(col + size_t(max(t - g, 0)) * RADIX, col + size_t(max(t - 4 - g, 0)) * RADIX,
                         col + size_t(max(t - 8 - g, 0)) * RADIX, col + size_t(max(t - 12 - g, 0)) * RADIX, w);
}
__device__ __forceinline__ void loadQuarterRows(const uint32_t* col, int32_t t, int32_t g, u32x4 (&w)[8])
{
    const uint32_t* p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        p[i] = col + size_t(max(t - 4 * i - g, 0)) * RADIX;
    loadQuarterRows32Sc1(p, w);
} // "no INCLUSIVE word met"

/*! @brief decoupled look-back of one digit quarter by one wave, 16 status rows per round trip
 *
 *  Lane l = 16 g + q reads the digits 64 wave + 4q .. +3 (16 bytes, agent scope: sc1 is coherent across the 8 XCD
 *  L2s) of the rows t-g, t-4-g, t-8-g, t-12-g: one instruction covers 4 rows, 4 instructions the 16 tiles before t.
 *  Rows are consumed strictly in order up to the first one this quarter of which is not published yet; per digit
 *  the counts are added up to and including the first INCLUSIVE word.  Status words are 32-bit {2-bit state, 30-bit
 *  count}: value and flag travel in one word, so no fence is needed and tearing between words is harmless.
 *  Why 16 rows at once: the tiles in flight start faster (about 10 per microsecond) than one agent-scope round trip
 *  (about 1.2 us), so the nearest INCLUSIVE row is typically 15-20 tiles back; walking there 4 rows per round trip
 *  left the CU idle for 5-6 round trips per tile.  The four quarters proceed independently, no barrier inside.
 *  @return exclusive prefix (count of the digit in all earlier tiles), identical in the 4 lanes that share q */
__device__ __forceinline__ u32x4 quarterLookBack(const uint32_t* __restrict__ status, uint32_t tile, unsigned wave,
                                                 unsigned lane, uint32_t* __restrict__ errors
#ifdef CSTONE_SORT_TRACE
                                                 , unsigned& rounds, unsigned& rowsUsed
#endif
)
{
    const unsigned g = lane >> 4, q = lane & 15u;
    const uint32_t* col = status + 64 * wave + 4 * q;
    u32x4 excl     = {0, 0, 0, 0};
    bool done[4]   = {false, false, false, false};
    int32_t t      = int32_t(tile) - 1;
    unsigned spins = 0;
    bool finished  = t < 0;
    while (!finished)
    {
        u32x4 w[LB_INSTR];
        loadQuarterRows(col, t, int32_t(g), w);
        // first row (0 = tile t) whose quarter is not completely published; rows before tile 0 do not exist and
        // are never reached (row 0 is INCLUSIVE for every digit)
        unsigned stop = LB_ROWS;
#pragma unroll
        for (int i = 0; i < LB_INSTR; ++i)
        {
            bool exists = t - 4 * i - int32_t(g) >= 0;
            bool ready  = true;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                ready = ready && (w[i][c] & ~COUNT_MASK) != 0;
            uint64_t m = __ballot(ready || !exists);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                if (stop == LB_ROWS && ((m >> (16 * gg)) & 0xFFFFull) != 0xFFFFull) stop = 4 * i + gg;
        }
        // per digit: nearest usable row holding an INCLUSIVE word
        unsigned first[4] = {LB_NONE, LB_NONE, LB_NONE, LB_NONE};
#pragma unroll
        for (int i = LB_INSTR - 1; i >= 0; --i)
        {
            unsigned rho = 4 * i + g;
            bool usable  = rho < stop && t - int32_t(rho) >= 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (usable && (w[i][c] & ~COUNT_MASK) == STATE_INC) first[c] = rho;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
            first[c] = min(first[c], unsigned(__shfl_xor(int(first[c]), 16)));
            first[c] = min(first[c], unsigned(__shfl_xor(int(first[c]), 32)));
        }
        u32x4 sum = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < LB_INSTR; ++i)
        {
            unsigned rho = 4 * i + g;
            bool usable  = rho < stop && t - int32_t(rho) >= 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (usable && rho <= first[c]) sum[c] += w[i][c] & COUNT_MASK;
        }
        bool allDone = true;
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
            uint32_t v = sum[c];
            v += uint32_t(__shfl_xor(int(v), 16));
            v += uint32_t(__shfl_xor(int(v), 32));
            if (!done[c])
            {
                excl[c] += v;
                done[c] = first[c] != LB_NONE;
            }
            allDone = allDone && done[c];
        }
#ifdef CSTONE_SORT_TRACE
        ++rounds;
        rowsUsed += stop;
#endif
        if (__all(allDone)) { finished = true; }
        else
        {
            t -= int32_t(stop);
            if (t < 0) finished = true; // row 0 is always inclusive: cannot be reached with open digits
            if (stop == 0)
            {
#ifdef CSTONE_SORT_DEBUG
                if (++spins > (1u << 16))
                {
                    if (lane == 0 || lane == 16 || lane == 63)
                        printf("look-back stuck: tile %u wave %u lane %u t %d words %08x %08x %08x %08x | %08x\n", tile, wave,
                               lane, t, w[0][0], w[0][1], w[0][2], w[0][3], w[1][0]);
#else
                if (++spins > (1u << 22)) // seconds: something is badly wrong, do not hang the GPU
                {
#endif
                    if (lane == 0) atomicOr(errors, 1u);
                    finished = true;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    return excl;
}

/*! @brief rank, permute and scatter one tile whose keys are already in registers
 *
 *  TAIL = false: a full tile of TILE pairs inside the look-back chain (the hot path, no validity predicates)
 *  TAIL = true : the last, partial tile. It sits at the END of every digit bin (it is the last tile in input
 *                order), so its slots follow from the global digit bases alone and it needs no look-back:
 *                first slot of digit d = end(d) - (tail count of d), end(d) = bases[d+1] (n for the last digit) */
template<class K, int BLOCK, bool TAIL, bool BALLOT>
__device__ __forceinline__ void sortTile(SortSmem<K, BLOCK>& sm, K (&key)[SortCfg<K, BLOCK>::ITEMS],
                                         uint32_t tile, unsigned tileCount,
                                         const uint32_t* __restrict__ valsIn,
                                         K* __restrict__ keysOut, uint32_t* __restrict__ valsOut, int shift,
                                         const uint32_t* __restrict__ bases, uint32_t* __restrict__ status,
                                         uint32_t* __restrict__ errors, uint32_t n)
{
    using Cfg           = SortCfg<K, BLOCK>;
    constexpr int ITEMS = Cfg::ITEMS, TILE = Cfg::TILE, WAVES = Cfg::WAVES;
    constexpr bool FULL = !TAIL;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tileBase = tile * uint32_t(TILE); // n < 2^30: 32-bit index arithmetic throughout
    const unsigned segBase  = wave * (64 * ITEMS);
#ifdef CSTONE_SORT_TRACE
    const size_t traceRow = size_t(shift / RADIX_BITS) * (n / TILE + 1) + tile;
#endif

    // values are fetched now so that their latency hides behind the ranking
#if CSTONE_SORT_EARLY_LB
    constexpr bool EARLY = FULL && WAVES > LB_WAVES;
#else
    constexpr bool EARLY = false;
#endif
    // EARLY: the look-back waves fetch their values behind the look-back (their registers hold keys and ranks until then)
    const bool lateValues = EARLY && wave < unsigned(LB_WAVES);
    uint32_t val[ITEMS];
    // valsIn == nullptr: the values are the positions 0..n-1 (first pass of a sort that starts from the identity
    // ordering, sequenceGpu + sortByKeyGpu in one): nothing to read
#define CSTONE_LOAD_VALUES()                                                                                           \
    _Pragma("unroll") for (int r = 0; r < ITEMS; ++r)                                                                  \
    {                                                                                                                  \
        unsigned idx = segBase + r * 64 + lane;                                                                        \
        val[r]       = valsIn == nullptr ? tileBase + idx                                                              \
                                         : ((FULL || idx < tileCount) ? streamLoad(valsIn + tileBase + idx) : 0u);     \
    }
    if (!lateValues) { CSTONE_LOAD_VALUES() }
#ifdef CSTONE_SORT_TRACE
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); // the keys have arrived, the 16 value loads may still fly
    SORT_TRACE(2)
#endif

    // ---- 1. stable in-wave ranking by returning LDS atomics on the wave's private digit counters: ds_add_rtn hands
    //      the lanes of one instruction that hit the same counter their values in ascending lane order on gfx950
    //      (probed once per context, ldsOrderProbeKernel), and a wave's LDS instructions execute in order, so
    //      rank = number of equal digits before this key in (item, lane) order.  A wave-uniform digit (the high
    //      digits of nearly sorted input) would serialise 64 ways: it takes the arithmetic shortcut instead.
    unsigned rank[ITEMS];
    uint32_t* myHist = sm.waveHist + wave * RADIX;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        unsigned d = unsigned(key[r] >> shift) & (RADIX - 1);
        if (BALLOT)
        {
            // the ranking that does not rely on the service order of LDS atomics (selected when the one-time probe of
            // that order fails, or by CSTONE_SORT_BALLOT_RANK): wave64 match-any of the digit by 8 ballots; the lowest
            // lane of every digit class advances the class counter, all read it before (LDS executes a wave in order)
            const bool valid = FULL || segBase + r * 64 + lane < tileCount;
            const uint64_t vm = __ballot(valid);
            uint32_t mlo = uint32_t(vm), mhi = uint32_t(vm >> 32);
            matchDigit(d, mlo, mhi);
            const unsigned below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
            const unsigned base  = myHist[d];
            if (valid && below == 0) myHist[d] = base + unsigned(__popc(mlo) + __popc(mhi));
            rank[r] = valid ? base + below : 0u;
        }
        else if (FULL)
        {
            unsigned d0 = __builtin_amdgcn_readfirstlane(d);
            if (__all(d == d0))
            {
                unsigned base = myHist[d0];
                if (lane == 0) myHist[d0] = base + 64;
                rank[r] = base + lane;
            }
            else { rank[r] = atomicAdd(&myHist[d], 1u); }
        }
        else
        {
            bool valid = segBase + r * 64 + lane < tileCount;
            rank[r]    = valid ? atomicAdd(&myHist[d], 1u) : 0u;
        }
    }
    SORT_TRACE(3)
    ldsBarrier();
    SORT_TRACE(4)

    unsigned pos[ITEMS];
    if constexpr (EARLY)
    {
        // ---- 2-5 (16-wave tiles inside the chain). The look-back of the predecessors needs nothing of THIS tile, so
        //      the four look-back waves start it right behind the ranking barrier, while ONE other wave (4 digits per
        //      lane) adds up the tile totals, publishes the AGGREGATE row and turns the per-wave counts into slots; the
        //      remaining waves pick the slots up through an LDS flag (a wave's LDS instructions execute in order) and
        //      permute their keys meanwhile.  No workgroup barrier between the ranking and the stores.
        constexpr unsigned SCAN_WAVE = LB_WAVES;
#ifdef CSTONE_SORT_FLAT_FLAG
        volatile uint32_t* slotsReady = &sm.tileShared[1];
#else
        // an LDS (address space 3) pointer: a generic volatile pointer would turn the polls into FLAT loads that wait
        // for the wave's global memory operations as well (vmcnt)
        auto* slotsReady = (__attribute__((address_space(3))) volatile uint32_t*)&sm.tileShared[1];
#endif
        u32x4 excl = {0, 0, 0, 0}, base4 = {0, 0, 0, 0};
        SORT_TRACE(5)
        SORT_TRACE(6)
        SORT_TRACE(7)
        if (wave == SCAN_WAVE)
        {
            u32x4 tot = {0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < WAVES; ++w)
                tot += *reinterpret_cast<const u32x4*>(&sm.waveHist[w * RADIX + 4 * lane]);
            // this wave's value loads are retired before its AGGREGATE store goes out (one vmcnt queue, see below)
#pragma unroll
            for (int r = 0; r < ITEMS; ++r)
                asm volatile("" : "+v"(val[r]));
            storeRowSc1(status + size_t(tile) * RADIX + 4 * lane, tot | (tile == 0 ? STATE_INC : STATE_AGG));
            *reinterpret_cast<u32x4*>(&sm.total[4 * lane]) = tot;
            uint32_t s4 = tot[0] + tot[1] + tot[2] + tot[3], inc4 = s4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
            {
                uint32_t t = __shfl_up(inc4, o);
                if (lane >= unsigned(o)) inc4 += t;
            }
            u32x4 run;
            run[0] = inc4 - s4;
            run[1] = run[0] + tot[0];
            run[2] = run[1] + tot[1];
            run[3] = run[2] + tot[2];
            *reinterpret_cast<u32x4*>(&sm.digitStart[4 * lane]) = run;
#pragma unroll
            for (int w = 0; w < WAVES; ++w)
            {
                u32x4* slot = reinterpret_cast<u32x4*>(&sm.waveHist[w * RADIX + 4 * lane]);
                const u32x4 c = *slot;
                *slot         = run;
                run += c;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) *slotsReady = 1u;
        }
        else if (wave < unsigned(LB_WAVES))
        {
            base4 = *reinterpret_cast<const u32x4*>(bases + 64 * wave + 4 * (lane & 15u));
#ifdef CSTONE_SORT_TRACE
            unsigned rounds = 0, rowsUsed = 0;
            excl = quarterLookBack(status, tile, wave, lane, errors, rounds, rowsUsed);
            if (wave == 0)
            {
                SORT_TRACE_VAL(13, rounds)
                SORT_TRACE_VAL(14, rowsUsed)
                SORT_TRACE(8)
            }
#else
            excl = quarterLookBack(status, tile, wave, lane, errors);
#endif
            CSTONE_LOAD_VALUES()
        }
        while (*slotsReady == 0u)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        if (wave < unsigned(LB_WAVES))
        {
            asm volatile("" : "+v"(base4)); // pins the wait for base4 here, ahead of the store
            if (lane < 16)
            {
                const unsigned d0 = 64 * wave + 4 * lane;
                const u32x4 tot   = *reinterpret_cast<const u32x4*>(&sm.total[d0]);
                if (tile > 0) storeRowSc1(status + size_t(tile) * RADIX + d0, ((excl + tot) & COUNT_MASK) | STATE_INC);
                const u32x4 start4 = *reinterpret_cast<const u32x4*>(&sm.digitStart[d0]);
                *reinterpret_cast<u32x4*>(&sm.binOffset[d0]) = base4 + excl - start4;
            }
        }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
        {
            unsigned d        = unsigned(key[r] >> shift) & (RADIX - 1);
            pos[r]            = myHist[d] + rank[r];
            sm.stage[pos[r]]  = key[r];
        }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
            asm volatile("" : "+v"(val[r]));
        ldsBarrier();
        SORT_TRACE(9)
    }
    else
    {
        // ---- 2. digit threads (the first RADIX threads): tile totals and their scan over the digits
        uint32_t total = 0, inc = 0;
        if (tid < RADIX)
        {
    #pragma unroll
            for (int w = 0; w < WAVES; ++w)
                total += sm.waveHist[w * RADIX + tid];
            sm.total[tid] = total;
            inc           = total;
    #pragma unroll
            for (int o = 1; o < 64; o <<= 1)
            {
                uint32_t t = __shfl_up(inc, o);
                if (lane >= unsigned(o)) inc += t;
            }
            if (lane == 63) sm.scanTmp[wave] = inc;
        }
        ldsBarrier();
        SORT_TRACE(5)

        u32x4 base4 = {0, 0, 0, 0};
        if (!TAIL && wave < unsigned(LB_WAVES))
        {
            // global digit bases of this wave's digit quarter: fetched now, so that no load has to be waited for behind
            // the INCLUSIVE store below (vmcnt counts stores too: a wait for a younger load would also sit out the
            // store's round trip)
            base4 = *reinterpret_cast<const u32x4*>(bases + 64 * wave + 4 * (lane & 15u));
            if (wave == 0)
            {
                // publish the tile aggregate (tile 0: already the inclusive prefix) as one 1 KiB row
                u32x4 t4 = *reinterpret_cast<const u32x4*>(&sm.total[4 * lane]);
                storeRowSc1(status + size_t(tile) * RADIX + 4 * lane, t4 | (tile == 0 ? STATE_INC : STATE_AGG));
            }
        }

        // ---- 3. tile-local slot of the first element of every (wave, digit)
        if (tid < RADIX)
        {
            uint32_t off = 0;
            for (unsigned w = 0; w < wave; ++w)
                off += sm.scanTmp[w];
            uint32_t run       = off + inc - total;
            sm.digitStart[tid] = run;
    #pragma unroll
            for (int w = 0; w < WAVES; ++w)
            {
                uint32_t c                   = sm.waveHist[w * RADIX + tid];
                sm.waveHist[w * RADIX + tid] = run;
                run += c;
            }
        }
        ldsBarrier();
        SORT_TRACE(6)

        // ---- 4. permute keys into tile-sorted order through LDS
    #pragma unroll
        for (int r = 0; r < ITEMS; ++r)
        {
            unsigned d = unsigned(key[r] >> shift) & (RADIX - 1);
            pos[r]     = myHist[d] + rank[r];
            if (FULL || segBase + r * 64 + lane < tileCount) sm.stage[pos[r]] = key[r];
        }
        SORT_TRACE(7)
        // The value loads are retired HERE, before any store of this tile is issued (the INCLUSIVE row, the keys): vmcnt
        // counts loads and stores in one queue, a wait for val[] behind a store would sit out its round trip as well.
    #pragma unroll
        for (int r = 0; r < ITEMS; ++r)
            asm volatile("" : "+v"(val[r]));

        // ---- 5. global slot of every digit run of this tile
        if (TAIL)
        {
            if (tid < RADIX)
            {
                uint32_t end      = (tid == RADIX - 1) ? n : bases[tid + 1];
                sm.binOffset[tid] = end - total - sm.digitStart[tid];
            }
        }
        else if (wave < unsigned(LB_WAVES))
        {
    #ifdef CSTONE_SORT_TRACE
            unsigned rounds = 0, rowsUsed = 0;
            u32x4 excl = quarterLookBack(status, tile, wave, lane, errors, rounds, rowsUsed);
    #else
            u32x4 excl = quarterLookBack(status, tile, wave, lane, errors);
    #endif
            asm volatile("" : "+v"(base4)); // pins the wait for base4 here, ahead of the store
            if (lane < 16)
            {
                const unsigned d0 = 64 * wave + 4 * lane;
                const u32x4 tot   = *reinterpret_cast<const u32x4*>(&sm.total[d0]);
                if (tile > 0) storeRowSc1(status + size_t(tile) * RADIX + d0, ((excl + tot) & COUNT_MASK) | STATE_INC);
                const u32x4 start4 = *reinterpret_cast<const u32x4*>(&sm.digitStart[d0]);
                *reinterpret_cast<u32x4*>(&sm.binOffset[d0]) = base4 + excl - start4;
            }
    #ifdef CSTONE_SORT_TRACE
            if (wave == 0)
            {
                SORT_TRACE_VAL(13, rounds)
                SORT_TRACE_VAL(14, rowsUsed)
                SORT_TRACE(8)
            }
    #endif
        }
        ldsBarrier();
        SORT_TRACE(9)

    }

    // ---- 6. stream out keys (remember each slot for the values), then values through the same LDS block
    uint32_t dst[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
    {
        unsigned i = k * BLOCK + tid;
        dst[k]     = n; // marks "nothing to store"
        if (FULL || i < tileCount)
        {
            K kk       = sm.stage[i];
            unsigned d = unsigned(kk >> shift) & (RADIX - 1);
            dst[k]     = sm.binOffset[d] + i;
            if (dst[k] < n) { streamStore(keysOut + dst[k], kk); }
            else { atomicOr(errors, 8u); } // cannot happen; turns a would-be wild store into a reported error
        }
    }
    SORT_TRACE(10)
    ldsBarrier();
    uint32_t* vstage = reinterpret_cast<uint32_t*>(sm.stage);
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        if (FULL || segBase + r * 64 + lane < tileCount) vstage[pos[r]] = val[r];
    }
    ldsBarrier();
    SORT_TRACE(11)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
    {
        unsigned i = k * BLOCK + tid;
        if (dst[k] < n) streamStore(valsOut + dst[k], vstage[i]);
    }
    SORT_TRACE(12)
}

/*! Full tiles, one per workgroup, tile index by ticket: every lower tile is owned by a workgroup that has started,
 *  so the look-back cannot deadlock whatever the dispatch order or residency.  Workgroups retire and start
 *  continuously, which keeps their phases (load / rank / look-back / store) staggered across the chip; persistent
 *  variants (ticket per tile with key prefetch, or round-robin tiles) measured 8-40 % slower on MI355X because they
 *  synchronise the look-back rounds. */
template<class K, int BLOCK, bool BALLOT>
__global__ __launch_bounds__(BLOCK) void onesweepKernel(const K* __restrict__ keysIn,
                                                        const uint32_t* __restrict__ valsIn, K* __restrict__ keysOut,
                                                        uint32_t* __restrict__ valsOut, uint32_t n, int pass,
                                                        uint32_t numFullTiles, const uint32_t* __restrict__ bases,
                                                        uint32_t* __restrict__ ticket, uint32_t* __restrict__ status,
                                                        uint32_t* __restrict__ errors)
{
    using Cfg           = SortCfg<K, BLOCK>;
    constexpr int ITEMS = Cfg::ITEMS, TILE = Cfg::TILE, WAVES = Cfg::WAVES;
    __shared__ SortSmem<K, BLOCK> sm;

    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned segBase = wave * (64 * ITEMS);
#ifdef CSTONE_SORT_TRACE
    uint64_t tEntry = wall_clock64();
#endif

#if CSTONE_SORT_NO_TICKET
    // workgroups are dispatched in the order of their index on every XCD and lower tiles never wait for higher ones, so
    // the lowest unfinished tile is always resident: the look-back cannot deadlock; every wave clears its own counters,
    // nothing has to be agreed before the key loads go out
    const uint32_t tile = blockIdx.x;
    if (tid == 64) sm.tileShared[1] = 0; // "slots ready" flag of sortTile
    for (int i = lane; i < RADIX; i += 64)
        sm.waveHist[wave * RADIX + i] = 0;
    (void)ticket;
#else
    if (tid == 0) sm.tileShared[0] = atomicAdd(ticket, 1u);
    if (tid == 64) sm.tileShared[1] = 0; // "slots ready" flag of sortTile
    for (int i = tid; i < WAVES * RADIX; i += BLOCK)
        sm.waveHist[i] = 0;
    __syncthreads();
    const uint32_t tile = sm.tileShared[0];
#endif
    if (tile > numFullTiles) return;
    // Long passes (many generations of tiles per CU): the first generation starts spread over about 16 us instead of
    // all at once, so that the load / rank / store phases of the CUs are out of step from the beginning (measured
    // 1.5-3 % per pass at 1e8 pairs; 24 or 48 us spreads gain nothing, short passes would only lose the delay).
    if (numFullTiles >= 2048u && tile < 256u)
    {
        unsigned units = (tile * 600u) >> 8; // units of 64 clocks
        for (; units >= 100; units -= 100)
            __builtin_amdgcn_s_sleep(100);
        for (; units > 0; --units)
            __builtin_amdgcn_s_sleep(1);
    }
#ifndef CSTONE_SORT_TRACE
    if (tile == numFullTiles)
    {
        // the partial last tile (launched only when there is one): no look-back, see sortTile<TAIL = true>.  Handling it
        // here lets it run next to the full tiles instead of in a one-workgroup launch of its own behind them.
        const uint32_t tileBase  = numFullTiles * uint32_t(TILE);
        const unsigned tileCount = n - tileBase;
        K tkey[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
        {
            unsigned idx = segBase + r * 64 + lane;
            tkey[r]      = idx < tileCount ? keysIn[tileBase + idx] : K(~K(0));
        }
        sortTile<K, BLOCK, true, BALLOT>(sm, tkey, numFullTiles, tileCount, valsIn, keysOut, valsOut, pass * RADIX_BITS, bases,
                                 nullptr, errors, n);
        return;
    }
#else
    if (tile == numFullTiles) return;
#endif
#ifdef CSTONE_SORT_TRACE
    const size_t traceRow = size_t(pass) * (n / TILE + 1) + tile;
    SORT_TRACE_VAL(0, tEntry)
    SORT_TRACE(1)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    SORT_TRACE_VAL(15, (uint64_t(xcc) << 32) | hwid)
#endif
    K key[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
        key[r] = streamLoad(keysIn + tile * uint32_t(TILE) + segBase + r * 64 + lane);
    sortTile<K, BLOCK, false, BALLOT>(sm, key, tile, unsigned(TILE), valsIn, keysOut, valsOut, pass * RADIX_BITS,
                              bases, status, errors, n);
}

//! the last, partial tile (n % TILE pairs) as a kernel of its own: only used by -DCSTONE_SORT_TRACE builds, the regular
//! build handles it inside onesweepKernel (see sortTile<TAIL = true>)
template<class K, int BLOCK>
__global__ __launch_bounds__(BLOCK) void onesweepTailKernel(const K* __restrict__ keysIn,
                                                            const uint32_t* __restrict__ valsIn,
                                                            K* __restrict__ keysOut, uint32_t* __restrict__ valsOut,
                                                            uint32_t n, int pass, uint32_t numFullTiles,
                                                            const uint32_t* __restrict__ bases,
                                                            uint32_t* __restrict__ errors)
{
    using Cfg           = SortCfg<K, BLOCK>;
    constexpr int ITEMS = Cfg::ITEMS, TILE = Cfg::TILE, WAVES = Cfg::WAVES;
    __shared__ SortSmem<K, BLOCK> sm;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned segBase   = wave * (64 * ITEMS);
    const uint32_t tileBase  = numFullTiles * uint32_t(TILE);
    const unsigned tileCount = n - tileBase;
    for (int i = tid; i < WAVES * RADIX; i += BLOCK)
        sm.waveHist[i] = 0;
    K key[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
    {
        unsigned idx = segBase + r * 64 + lane;
        key[r]       = idx < tileCount ? keysIn[tileBase + idx] : K(~K(0));
    }
    ldsBarrier();
    sortTile<K, BLOCK, true, false>(sm, key, numFullTiles, tileCount, valsIn, keysOut, valsOut, pass * RADIX_BITS, bases,
                                    nullptr, errors, n);
}

/*! @brief orders the low key bits inside runs of equal high bits (keys >> shift), stably
 *
 *  After a sort on the digits above `shift` only, the elements of a run sit in input order.  The thread of a run's first
 *  element insertion-sorts the run in place (strict comparison: equal keys keep their order), so the result equals the
 *  full stable sort.  Meant for keys whose high bits almost identify them (SFC keys with the digits above the leaf
 *  level of the octree sorted: a run lies inside one leaf cell); runs longer than RUN_LIMIT raise *tooLong instead and
 *  the caller sorts the remaining digits the regular way.  Remove markers (all equal) are left alone. */
constexpr uint32_t RUN_LIMIT = 192;
constexpr int FIX_VEC = 4; // consecutive keys per lane: a stream this light is otherwise bound by wave launches
template<class K>
__global__ __launch_bounds__(256) void fixupRunsKernel(K* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t n,
                                                       int shift, int* __restrict__ tooLong)
{
    const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * FIX_VEC;
    const unsigned lane = threadIdx.x & 63u;
    // FIX_VEC + 3 keys around my slots: k[1 + e] = keys[base + e]; k[0] = the key before, k[FIX_VEC + 1], k[FIX_VEC + 2]
    // the two behind (runs of two are decided and swapped from registers).  Neighbouring lanes supply the edges, the
    // wave's edge lanes fetch theirs.
    K k[FIX_VEC + 3];
#pragma unroll
    for (int e = 0; e < FIX_VEC; ++e)
        k[1 + e] = base + e < n ? keys[base + e] : endKey<K>();
    k[0]           = __shfl_up(k[FIX_VEC], 1);
    k[FIX_VEC + 1] = __shfl_down(k[1], 1);
    k[FIX_VEC + 2] = __shfl_down(k[2], 1);
    if (lane == 0) k[0] = base > 0 && base - 1 < n ? keys[base - 1] : endKey<K>();
    if (lane == 63)
    {
        k[FIX_VEC + 1] = base + FIX_VEC < n ? keys[base + FIX_VEC] : endKey<K>();
        k[FIX_VEC + 2] = base + FIX_VEC + 1 < n ? keys[base + FIX_VEC + 1] : endKey<K>();
    }
#pragma unroll
    for (int e = 0; e < FIX_VEC; ++e)
    {
        const uint32_t i = base + e;
        const K k0       = k[1 + e];
        if (i + 1 >= n || k0 == endKey<K>()) continue;
        const K top = k0 >> shift;
        if (i > 0 && k[e] != endKey<K>() && (k[e] >> shift) == top) continue; // not the first of its run
        const K next = k[2 + e];
        if (next == endKey<K>() || (next >> shift) != top) continue;           // a run of one
        const K next2 = k[3 + e];
        if (i + 2 >= n || next2 == endKey<K>() || (next2 >> shift) != top)
        {
            // by far the most frequent case, a run of two: at most one swap, decided from registers
            if (k0 > next)
            {
                uint32_t v0 = vals[i], v1 = vals[i + 1];
                keys[i] = next, keys[i + 1] = k0;
                vals[i] = v1, vals[i + 1] = v0;
            }
            continue;
        }
        uint32_t end = i + 3;
        while (end < n && (keys[end] >> shift) == top && end - i <= RUN_LIMIT)
            ++end;
        if (end - i > RUN_LIMIT)
        {
            atomicOr(tooLong, 1);
            continue;
        }
        for (uint32_t a = i + 1; a < end; ++a)
        {
            const K ka        = keys[a];
            const uint32_t va = vals[a];
            uint32_t b        = a;
            while (b > i && keys[b - 1] > ka)
            {
                keys[b] = keys[b - 1];
                vals[b] = vals[b - 1];
                --b;
            }
            if (b != a)
            {
                keys[b] = ka;
                vals[b] = va;
            }
        }
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

//! inputs of at least this many pairs use the 16 Ki-pair tiles (CSTONE_SORT_LARGE_MIN overrides, for tuning runs)
size_t largeTileThreshold()
{
    static const size_t value = []
    {
        const char* e = std::getenv("CSTONE_SORT_LARGE_MIN");
        return e ? size_t(std::strtoull(e, nullptr, 10)) : size_t(1) << 20;
    }();
    return value;
}

template<class K>
size_t sortTempBytes(size_t n)
{
    // sized for the small tiles: an upper bound on the number of status rows of either configuration
    constexpr int TILE = SortCfg<K, SMALL_BLOCK>::TILE;
    constexpr int P    = SortCfg<K, SMALL_BLOCK>::PASSES;
    size_t numTiles    = (n + TILE - 1) / TILE;
    return alignUp((headerWords(P) + size_t(P) * numTiles * RADIX) * sizeof(uint32_t));
}

template<class K, int BLOCK, bool BALLOT>
void launchPasses(cstone_hip_ctx* ctx, const SortTemp& t, K* keys, uint32_t* vals, size_t n, K* keysAlt,
                  uint32_t* valsAlt, bool iotaValues, int startPass, int endPass)
{
    using Cfg             = SortCfg<K, BLOCK>;
    constexpr int P       = Cfg::PASSES;
    uint32_t numFullTiles = uint32_t(n / Cfg::TILE);
    bool haveTail         = (n % Cfg::TILE) != 0;
    K* kIn         = keys;
    uint32_t* vIn  = iotaValues ? nullptr : vals;
    K* kOut        = keysAlt;
    uint32_t* vOut = valsAlt;
    for (int p = startPass; p < std::min(P, endPass); ++p)
    {
        StageTimer timer(ctx, vIn == nullptr ? CSTONE_STAGE_SORT_PASS_IOTA : CSTONE_STAGE_SORT_PASS);
        const uint32_t* bases = t.hist + size_t(p) * RADIX;
#ifndef CSTONE_SORT_TRACE
        // one launch: the workgroup that draws ticket numFullTiles takes the partial last tile
        hipLaunchKernelGGL((onesweepKernel<K, BLOCK, BALLOT>), numFullTiles + (haveTail ? 1u : 0u), BLOCK, 0, ctx->stream, kIn, vIn,
                           kOut, vOut, uint32_t(n), p, numFullTiles, bases, t.tickets + p,
                           t.status + size_t(p) * numFullTiles * RADIX, t.errors);
#else
        if (numFullTiles)
            hipLaunchKernelGGL((onesweepKernel<K, BLOCK, BALLOT>), numFullTiles, BLOCK, 0, ctx->stream, kIn, vIn, kOut, vOut,
                               uint32_t(n), p, numFullTiles, bases, t.tickets + p,
                               t.status + size_t(p) * numFullTiles * RADIX, t.errors);
        if (haveTail)
            hipLaunchKernelGGL((onesweepTailKernel<K, BLOCK>), 1, BLOCK, 0, ctx->stream, kIn, vIn, kOut, vOut,
                               uint32_t(n), p, numFullTiles, bases, t.errors);
#endif
        std::swap(kIn, kOut);
        if (p == startPass && iotaValues) { vIn = vOut, vOut = vals; }
        else { std::swap(vIn, vOut); }
    }
    static_assert(P % 2 == 0, "an even number of passes leaves the result in the caller's buffers");
}

template<class K>
int sortPairs(cstone_hip_ctx* ctx, K* keys, uint32_t* vals, size_t n, K* keysAlt, uint32_t* valsAlt, void* temp,
              size_t tempBytes, bool iotaValues = false, int histogramState = 0 /* 0: do all, 1: only clear the
              temp (the caller counts into it next), 2: temp cleared and digit counts present */,
              int startPass = 0 /* even: digits below 8 * startPass bits are left to the caller (fixupRuns) */,
              int endPass = 64 /* even: the caller knows that all keys are zero from bit 8 * endPass on */)
{
    if (n == 0) return CSTONE_OK;
    if (n >= (size_t(1) << 30)) return fail(ctx, CSTONE_E_ARG, "sort_pairs: n = %zu exceeds 2^30 - 1", n);
    size_t need = sortTempBytes<K>(n);
    if (tempBytes < need) return fail(ctx, CSTONE_E_CAPACITY, "sort_pairs: temp %zu < %zu bytes", tempBytes, need);

    if (ctx->ldsOrderOk < 0)
    {
        uint32_t* flag = (uint32_t*)ctx->devScalars + 62;
        CS_HIP(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
        hipLaunchKernelGGL(ldsOrderProbeKernel, 8, 1024, 0, ctx->stream, flag);
        ctx->hostScalars[62] = 1;
        CS_HIP(ctx, hipMemcpyAsync(ctx->hostScalars + 62, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->ldsOrderOk = ctx->hostScalars[62] == 0;
    }
    // a device that failed the probe (none has so far) sorts with the ballot-based ranking: slower, order-independent
    const bool ballot = !ctx->ldsOrderOk || std::getenv("CSTONE_SORT_BALLOT_RANK") != nullptr;

    constexpr int P = sizeof(K);
    auto* words     = (uint32_t*)temp;
    SortTemp t;
    t.hist    = words;
    t.tickets = words + size_t(P) * RADIX;
    t.errors  = (uint32_t*)ctx->devScalars + 63; // sticky; reported by cstone_hip_ctx_sync
    t.status  = words + headerWords(P);

    // (16 Ki-pair tiles are workgroups of 1024 lanes: while a bandwidth kernel of the same context fills the CUs from the
    //  second stream -- the gather of x, y, z, placeColumnsKernel -- four of its workgroups never finish on one CU at the
    //  same moment, and the first digit pass of the tree's node sort sat there until that kernel had drained: 0.28 ms
    //  (single-rank sync) and 0.73 ms (multi-rank) instead of 0.02 ms at 2.6 million node keys)
    bool large       = n >= largeTileThreshold() && !ctx->auxBusy;
    size_t tile      = large ? SortCfg<K, LARGE_BLOCK>::TILE : SortCfg<K, SMALL_BLOCK>::TILE;
    size_t usedBytes = (headerWords(P) + size_t(P) * (n / tile) * RADIX) * sizeof(uint32_t);
    if (histogramState != 2) CS_HIP(ctx, hipMemsetAsync(temp, 0, usedBytes, ctx->stream));
    if (histogramState == 1) return CSTONE_OK;
    {
        StageTimer timer(ctx, CSTONE_STAGE_SORT_HIST);
        size_t nVec   = n / (16 / sizeof(K));
        unsigned grid = unsigned(std::min<size_t>(size_t(ctx->numCu) * 8, (nVec + HIST_BLOCK - 1) / HIST_BLOCK));
        grid          = std::max(grid, 1u);
        if (histogramState == 0) hipLaunchKernelGGL(histogramKernel<K>, grid, HIST_BLOCK, 0, ctx->stream, keys, n, t.hist);
        hipLaunchKernelGGL(scanHistogramKernel, P, RADIX, 0, ctx->stream, t.hist);
    }
#ifdef CSTONE_SORT_TRACE
    const char* traceFile = std::getenv("CSTONE_SORT_TRACE_FILE");
    uint64_t* traceDev    = nullptr;
    size_t traceWords     = size_t(P) * (n / tile + 1) * TRACE_SLOTS;
    if (traceFile)
    {
        CS_HIP(ctx, hipMalloc((void**)&traceDev, traceWords * 8));
        CS_HIP(ctx, hipMemsetAsync(traceDev, 0, traceWords * 8, ctx->stream));
    }
    CS_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sortTrace), &traceDev, sizeof(traceDev), 0, hipMemcpyHostToDevice,
                                       ctx->stream));
#endif
    if (ballot)
    {
        if (large) launchPasses<K, LARGE_BLOCK, true>(ctx, t, keys, vals, n, keysAlt, valsAlt, iotaValues, startPass, endPass);
        else launchPasses<K, SMALL_BLOCK, true>(ctx, t, keys, vals, n, keysAlt, valsAlt, iotaValues, startPass, endPass);
    }
    else if (large) launchPasses<K, LARGE_BLOCK, false>(ctx, t, keys, vals, n, keysAlt, valsAlt, iotaValues, startPass, endPass);
    else launchPasses<K, SMALL_BLOCK, false>(ctx, t, keys, vals, n, keysAlt, valsAlt, iotaValues, startPass, endPass);
    CS_HIP(ctx, hipGetLastError());
#ifdef CSTONE_SORT_TRACE
    if (traceFile)
    {
        std::vector<uint64_t> host(traceWords);
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        CS_HIP(ctx, hipMemcpy(host.data(), traceDev, traceWords * 8, hipMemcpyDeviceToHost));
        CS_HIP(ctx, hipFree(traceDev));
        if (FILE* f = std::fopen(traceFile, "wb"))
        {
            uint64_t hdr[4] = {uint64_t(P), uint64_t(n / tile + 1), uint64_t(TRACE_SLOTS), uint64_t(tile)};
            std::fwrite(hdr, 8, 4, f);
            std::fwrite(host.data(), 8, traceWords, f);
            std::fclose(f);
        }
    }
#endif
    return CSTONE_OK;
}

} // namespace

template<class K>
int sortPairsArena(cstone_hip_ctx* ctx, K* keys, uint32_t* vals, size_t n, int keyBits)
{
    size_t tb = sortTempBytes<K>(n);
    CS_TRY(arenaReserve(ctx, alignUp(n * sizeof(K)) + alignUp(n * sizeof(uint32_t)) + tb + 1024));
    K* ka        = (K*)arenaTake(ctx, n * sizeof(K));
    uint32_t* va = (uint32_t*)arenaTake(ctx, n * sizeof(uint32_t));
    void* tmp    = arenaTake(ctx, tb);
    // keyBits: the caller's bound on the significant bits of the keys (digit passes above them would be the identity)
    const int endPass = std::min(int(sizeof(K)), ((keyBits + 7) / 8 + 1) & ~1);
    int rc            = sortPairs<K>(ctx, keys, vals, n, ka, va, tmp, tb, false, 0, 0, endPass);
    arenaReset(ctx);
    return rc;
}
template int sortPairsArena<uint32_t>(cstone_hip_ctx*, uint32_t*, uint32_t*, size_t, int);
template int sortPairsArena<uint64_t>(cstone_hip_ctx*, uint64_t*, uint32_t*, size_t, int);

} // namespace cship

namespace cship
{
/*! computeSfcKeys + setMapFromCodes with a hint on the key structure: only the digits at or above bit 8 * startPass
 *  go through the radix passes, the order inside runs of equal high bits is finished by fixupRunsKernel.
 *  *tooLongDev (device int, zeroed here) != 0 afterwards means a run was too long for that: the caller then completes
 *  the job with a regular sort of (keys, ordering), which yields the same result as if nothing had been skipped. */
int sfcKeysAndOrderingHint(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* x, const void* y,
                           const void* z, void* keys, uint32_t* ordering, size_t n, const cstone_box& box,
                           void* keys_alt, uint32_t* values_alt, void* temp, size_t temp_bytes, int startPass,
                           int* tooLongDev, bool honourMarkers, void* extentsOut, bool* extentsMeasured)
{
    if (extentsMeasured) *extentsMeasured = false;
    if (n == 0) return CSTONE_OK;
    startPass &= ~1; // an even number of passes leaves the result in the caller's buffers
    auto run = [&](int state)
    {
        return key_bits == 32 ? sortPairs<uint32_t>(ctx, (uint32_t*)keys, ordering, n, (uint32_t*)keys_alt, values_alt,
                                                    temp, temp_bytes, true, state, startPass)
                              : sortPairs<uint64_t>(ctx, (uint64_t*)keys, ordering, n, (uint64_t*)keys_alt, values_alt,
                                                    temp, temp_bytes, true, state, startPass);
    };
    CS_TRY(run(1));
    bool fused = false;
    CS_TRY(computeKeysAndHistogram(ctx, curve, key_bits, real_bits, x, y, z, keys, n, box, (uint32_t*)temp, &fused,
                                   startPass, honourMarkers, extentsOut));
    if (extentsMeasured) *extentsMeasured = fused && extentsOut != nullptr;
    CS_TRY(run(fused ? 2 : 0));
    if (startPass > 0)
    {
        StageTimer timer(ctx, CSTONE_STAGE_SORT_HIST);
        CS_HIP(ctx, hipMemsetAsync(tooLongDev, 0, sizeof(int), ctx->stream));
        if (key_bits == 32)
            hipLaunchKernelGGL(fixupRunsKernel<uint32_t>, gridFor(n, 256, FIX_VEC), 256, 0, ctx->stream, (uint32_t*)keys, ordering,
                               uint32_t(n), 8 * startPass, tooLongDev);
        else
            hipLaunchKernelGGL(fixupRunsKernel<uint64_t>, gridFor(n, 256, FIX_VEC), 256, 0, ctx->stream, (uint64_t*)keys, ordering,
                               uint32_t(n), 8 * startPass, tooLongDev);
        CS_HIP(ctx, hipGetLastError());
    }
    return CSTONE_OK;
}
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
        return key_bits == 32 ? sortPairsArena<uint32_t>(ctx, (uint32_t*)keys, values, n, 32)
                              : sortPairsArena<uint64_t>(ctx, (uint64_t*)keys, values, n, 64);
    }
    return key_bits == 32
               ? sortPairs<uint32_t>(ctx, (uint32_t*)keys, values, n, (uint32_t*)keys_alt, values_alt, temp, temp_bytes)
               : sortPairs<uint64_t>(ctx, (uint64_t*)keys, values, n, (uint64_t*)keys_alt, values_alt, temp,
                                     temp_bytes);
}

int cstone_hip_sort_keys_ordering(cstone_hip_ctx* ctx, int key_bits, void* keys, uint32_t* ordering, size_t n,
                                  void* keys_alt, uint32_t* values_alt, void* temp, size_t temp_bytes)
{
    if (!ctx || (key_bits != 32 && key_bits != 64)) return fail(ctx, CSTONE_E_ARG, "sort_keys_ordering: bad key_bits");
    if (n && (!keys || !ordering || !keys_alt || !values_alt || !temp))
        return fail(ctx, CSTONE_E_ARG, "sort_keys_ordering: null array");
    return key_bits == 32 ? sortPairs<uint32_t>(ctx, (uint32_t*)keys, ordering, n, (uint32_t*)keys_alt, values_alt,
                                                temp, temp_bytes, true)
                          : sortPairs<uint64_t>(ctx, (uint64_t*)keys, ordering, n, (uint64_t*)keys_alt, values_alt,
                                                temp, temp_bytes, true);
}

int cstone_hip_sfc_keys_and_ordering(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* x,
                                     const void* y, const void* z, void* keys, uint32_t* ordering, size_t n,
                                     const cstone_box* box_host, void* keys_alt, uint32_t* values_alt, void* temp,
                                     size_t temp_bytes)
{
    if (!ctx || !box_host || (key_bits != 32 && key_bits != 64) || (curve != CSTONE_MORTON && curve != CSTONE_HILBERT))
        return fail(ctx, CSTONE_E_ARG, "sfc_keys_and_ordering: bad argument");
    if (n == 0) return CSTONE_OK;
    if (!x || !y || !z || !keys || !ordering || !keys_alt || !values_alt || !temp)
        return fail(ctx, CSTONE_E_ARG, "sfc_keys_and_ordering: null array");
    // clear the sort's workspace, count the digits while encoding, then the digit passes
    auto run = [&](int state)
    {
        return key_bits == 32 ? sortPairs<uint32_t>(ctx, (uint32_t*)keys, ordering, n, (uint32_t*)keys_alt, values_alt,
                                                    temp, temp_bytes, true, state)
                              : sortPairs<uint64_t>(ctx, (uint64_t*)keys, ordering, n, (uint64_t*)keys_alt, values_alt,
                                                    temp, temp_bytes, true, state);
    };
    CS_TRY(run(1));
    bool fused = false;
    CS_TRY(computeKeysAndHistogram(ctx, curve, key_bits, real_bits, x, y, z, keys, n, *box_host, (uint32_t*)temp,
                                   &fused));
    return run(fused ? 2 : 0);
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
