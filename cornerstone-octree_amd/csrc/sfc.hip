// SFC key encode for gfx950.  Replaces computeSfcKeysGpu (R/sfc/sfc_gpu.cu:39-57; arithmetic
// R/sfc/sfc.hpp:158-194).  HBM-bound streaming kernel: (3T + 2K) bytes per particle.
//   - SoA x/y/z are read with 16-byte lane loads (2 doubles / 4 floats per lane) when the pointers
//     allow it, keys written the same way
//   - Hilbert keys come from a 24-state transducer table in LDS (device_keys.hpp): one ds_read_u16
//     per level instead of ~25 data-dependent VALU ops; VEC independent chains per lane hide the
//     LDS latency
// THIS FILE MUST BE COMPILED WITH -ffp-contract=off: int(floor(x*m) - lo*m) is evaluated as two
// roundings on the reference's CPU path and must not become an FMA.
#include <algorithm>
#include <cstdlib>

#include <limits>

#include "ctx.hpp"
#include "device_keys.hpp"
#include "resort.hpp"

namespace cship
{

//! the integer cell of a position (R/sfc/sfc.hpp:188-194), without the interleave
template<class K, class T>
__device__ __forceinline__ void gridCell(T x, T y, T z, T mx, T my, T mz, T sx, T sy, T sz, unsigned& ix, unsigned& iy,
                                         unsigned& iz)
{
    constexpr int top = (1u << maxLevel<K>()) - 1;
    ix = unsigned(min(int(floor(x * mx) - sx), top));
    iy = unsigned(min(int(floor(y * my) - sy), top));
    iz = unsigned(min(int(floor(z * mz) - sz), top));
}

//! keys of VEC positions; enc2: the two-level Hilbert table in LDS (HilbertTables::enc2, unused for Morton keys)
template<class K, class T, int VEC, bool HILBERT>
__device__ __forceinline__ void keysOfPositions(const T (&vx)[VEC], const T (&vy)[VEC], const T (&vz)[VEC], T mx, T my,
                                                T mz, T sx, T sy, T sz, const uint16_t* enc2, K (&out)[VEC])
{
    unsigned ix[VEC], iy[VEC], iz[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        gridCell<K, T>(vx[v], vy[v], vz[v], mx, my, mz, sx, sy, sz, ix[v], iy[v], iz[v]);
    if constexpr (HILBERT) { hilbertFromGrid<K, VEC>(ix, iy, iz, enc2, out); }
    else
    {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            out[v] = mortonEncode<K>(ix[v], iy[v], iz[v]);
    }
}
template<class K, class T, bool HILBERT>
__device__ __forceinline__ K keyOfPosition(T x, T y, T z, T mx, T my, T mz, T sx, T sy, T sz, const uint16_t* enc2)
{
    const T ax[1] = {x}, ay[1] = {y}, az[1] = {z};
    K out[1];
    keysOfPositions<K, T, 1, HILBERT>(ax, ay, az, mx, my, mz, sx, sy, sz, enc2, out);
    return out[0];
}
//! the table into LDS (all 256 threads of the workgroup; the caller synchronises)
__device__ __forceinline__ void loadHilbertPairs(uint16_t* enc2, const uint16_t* __restrict__ tables)
{
    const uint16_t* src = tables + 2 * 48 * 8; // HilbertTables::enc2 follows enc and dec
    for (unsigned i = threadIdx.x; i < HILBERT_STATES * 64; i += 256)
        enc2[i] = src[i];
}

template<class K, class T, int VEC, bool HILBERT>
__global__ __launch_bounds__(256) void encodeKernel(const T* __restrict__ x, const T* __restrict__ y,
                                                    const T* __restrict__ z, K* __restrict__ keys, size_t n,
                                                    DBox<T> box, const uint16_t* __restrict__ encTable)
{
    __shared__ uint16_t enc2[HILBERT ? HILBERT_STATES * 64 : 1];
    if (HILBERT)
    {
        loadHilbertPairs(enc2, encTable);
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
        keysOfPositions<K, T, VEC, HILBERT>(vx, vy, vz, mx, my, mz, sx, sy, sz, enc2, out);
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
            K m   = keyOfPosition<K, T, HILBERT>(x[i], y[i], z[i], mx, my, mz, sx, sy, sz, enc2);
            if (old != endKey<K>()) keys[i] = m;
        }
    }
}

//! extents of a 256-thread workgroup: ext = {xmin, xmax, ymin, ymax, zmin, zmax} per lane -> partials[blockIdx.x][6]
template<class T>
__device__ __forceinline__ void foldBlockExtents(const T (&ext)[6], T* __restrict__ partials)
{
    __shared__ T wext[4][6];
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int k = 0; k < 6; ++k)
    {
        T v = ext[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
        {
            T t = __shfl_xor(v, o);
            v   = (k & 1) ? (t > v ? t : v) : (t < v ? t : v);
        }
        if (lane == 0) wext[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6)
    {
        const int k = threadIdx.x;
        T v         = wext[0][k];
        for (int w = 1; w < 4; ++w)
            v = (k & 1) ? (wext[w][k] > v ? wext[w][k] : v) : (wext[w][k] < v ? wext[w][k] : v);
        partials[size_t(blockIdx.x) * 6 + k] = v;
    }
}

/*! Encode fused with the digit histograms of the radix sort that follows it in Domain::sync (computeSfcKeys +
 *  setMapFromCodes, R/domain/assignment.hpp:81-86): the keys are counted while they are still in registers, which
 *  saves the sort its read of all keys, and the LDS atomics of the counting overlap with the HBM streams of the
 *  encode.  Grid-stride (few, long-lived workgroups: each flushes its 2 KiB-per-digit LDS histogram once).
 *  hist: [sizeof(K)][256] counters, zeroed by the caller. */
template<class K, class T, int VEC, bool HILBERT>
__global__ __launch_bounds__(256) void encodeHistogramKernel(const T* __restrict__ x, const T* __restrict__ y,
                                                             const T* __restrict__ z, K* __restrict__ keys, size_t n,
                                                             DBox<T> box, const uint16_t* __restrict__ encTable,
                                                             uint32_t* __restrict__ hist, int firstDigit,
                                                             bool honourMarkers, T* __restrict__ extentPartials)
{
    constexpr int P = int(sizeof(K));
    __shared__ uint16_t enc2[HILBERT ? HILBERT_STATES * 64 : 1];
    __shared__ uint32_t lh[P * 256];
    // extentPartials != nullptr: the coordinates' extents are measured on the way (Domain::sync encodes with the box of
    // the previous sync and checks afterwards that the box has not changed: no separate pass over x, y, z)
    T ext[6] = {std::numeric_limits<T>::infinity(),  -std::numeric_limits<T>::infinity(),
                std::numeric_limits<T>::infinity(),  -std::numeric_limits<T>::infinity(),
                std::numeric_limits<T>::infinity(),  -std::numeric_limits<T>::infinity()};
    auto widen = [&](T xv, T yv, T zv)
    {
        ext[0] = xv < ext[0] ? xv : ext[0], ext[1] = xv > ext[1] ? xv : ext[1];
        ext[2] = yv < ext[2] ? yv : ext[2], ext[3] = yv > ext[3] ? yv : ext[3];
        ext[4] = zv < ext[4] ? zv : ext[4], ext[5] = zv > ext[5] ? zv : ext[5];
    };
    if (HILBERT) loadHilbertPairs(enc2, encTable);
    for (int i = threadIdx.x; i < P * 256; i += 256)
        lh[i] = 0;
    __syncthreads();
    constexpr unsigned g = 1u << maxLevel<K>();
    const T mx = g * box.inv[0], my = g * box.inv[1], mz = g * box.inv[2]; // R/sfc/sfc.hpp:188-194
    const T sx = box.lo[0] * mx, sy = box.lo[1] * my, sz = box.lo[2] * mz;
    const unsigned lane = threadIdx.x & 63u;

    auto count = [&](K key, bool valid)
    {
#pragma unroll
        for (int p = 0; p < P; ++p)
        {
            if (p < firstDigit) continue; // digits the sort will not pass over need no counts
            unsigned d = unsigned(key >> (p * 8)) & 255u;
            // nearly sorted input makes the high digits wave-uniform: one add instead of a 64-way LDS conflict
            uint64_t vmask = __ballot(valid);
            if (vmask == 0) continue;
            unsigned d0   = __builtin_amdgcn_readfirstlane(__shfl(d, __ffsll((unsigned long long)vmask) - 1));
            uint64_t same = __ballot(valid && d == d0);
            if (same == vmask)
            {
                if (lane == unsigned(__ffsll((unsigned long long)vmask) - 1))
                    atomicAdd(&lh[p * 256 + d0], unsigned(__popcll(vmask)));
            }
            else if (valid) { atomicAdd(&lh[p * 256 + d], 1u); }
        }
    };
    auto encodeOne = [&](T xv, T yv, T zv, K old) -> K
    {
        K m = keyOfPosition<K, T, HILBERT>(xv, yv, zv, mx, my, mz, sx, sy, sz, enc2);
        return old == endKey<K>() ? old : m;
    };

    const size_t nVec   = n / VEC;
    const size_t stride = size_t(gridDim.x) * 256;
    const size_t iters  = (nVec + stride - 1) / stride; // whole waves walk the iterations together (ballots above)
    size_t vi           = size_t(blockIdx.x) * 256 + threadIdx.x;
    for (size_t it = 0; it < iters; ++it, vi += stride)
    {
        const bool valid = vi < nVec;
        K out[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            out[v] = 0;
        if (valid)
        {
            const size_t base = vi * VEC;
            T vx[VEC], vy[VEC], vz[VEC];
            K vk[VEC];
            __builtin_memcpy(vx, __builtin_assume_aligned(x + base, sizeof(T) * VEC), sizeof vx);
            __builtin_memcpy(vy, __builtin_assume_aligned(y + base, sizeof(T) * VEC), sizeof vy);
            __builtin_memcpy(vz, __builtin_assume_aligned(z + base, sizeof(T) * VEC), sizeof vz);
            // honourMarkers == false: the key array holds nothing of the caller's (no remove markers): it is not read
            if (honourMarkers) { __builtin_memcpy(vk, __builtin_assume_aligned(keys + base, sizeof(K) * VEC), sizeof vk); }
            else
            {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    vk[v] = 0;
            }
            keysOfPositions<K, T, VEC, HILBERT>(vx, vy, vz, mx, my, mz, sx, sy, sz, enc2, out);
            if (extentPartials)
            {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    widen(vx[v], vy[v], vz[v]);
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (vk[v] == endKey<K>()) out[v] = vk[v]; // particles flagged for removal keep their marker
            __builtin_memcpy(__builtin_assume_aligned(keys + base, sizeof(K) * VEC), out, sizeof out);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            count(out[v], valid);
    }
    // elements behind the last full vector: first wave of block 0
    if (blockIdx.x == 0 && threadIdx.x < 64)
    {
        size_t i   = nVec * VEC + threadIdx.x;
        bool valid = i < n;
        K key      = 0;
        if (valid)
        {
            key     = encodeOne(x[i], y[i], z[i], honourMarkers ? keys[i] : K(0));
            keys[i] = key;
            if (extentPartials) widen(x[i], y[i], z[i]);
        }
        count(key, valid);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P * 256; i += 256)
    {
        uint32_t c = lh[i];
        if (c) atomicAdd(&hist[i], c);
    }
    if (extentPartials) foldBlockExtents<T>(ext, extentPartials);
}

//! folds the per-workgroup extents of encodeHistogramKernel: out = {xmin, xmax, ymin, ymax, zmin, zmax}
template<class T>
__global__ __launch_bounds__(256) void foldExtentsKernel(const T* __restrict__ partials, unsigned numBlocks,
                                                         T* __restrict__ out)
{
    __shared__ T wext[4][6];
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int k = 0; k < 6; ++k)
    {
        T v = (k & 1) ? -std::numeric_limits<T>::infinity() : std::numeric_limits<T>::infinity();
        for (unsigned b = threadIdx.x; b < numBlocks; b += 256)
        {
            T t = partials[size_t(b) * 6 + k];
            v   = (k & 1) ? (t > v ? t : v) : (t < v ? t : v);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
        {
            T t = __shfl_xor(v, o);
            v   = (k & 1) ? (t > v ? t : v) : (t < v ? t : v);
        }
        if (lane == 0) wext[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6)
    {
        const int k = threadIdx.x;
        T v         = wext[0][k];
        for (int w = 1; w < 4; ++w)
            v = (k & 1) ? (wext[w][k] > v ? wext[w][k] : v) : (wext[w][k] < v ? wext[w][k] : v);
        out[k] = v;
    }
}

/*! Encode for the incremental re-sort (resort.hpp): the keys go to ra.keysOut (the caller's key array is only read, for
 *  its remove markers) and every particle is checked against the key range of the leaf its position belonged to at the
 *  previous sync.  Particles that left their leaf are counted per leaf and appended to the mover list -- staged per wave
 *  in LDS and flushed with one atomic per 128..256 movers; in ra.keysOut a hole (~0) takes their place.  The extents of x, y, z are measured like in
 *  encodeHistogramKernel.  Grid-stride, whole waves walk the iterations together. */
template<class K, class T, int VEC, bool HILBERT>
__global__ __launch_bounds__(256) void encodeResortKernel(const T* __restrict__ x, const T* __restrict__ y,
                                                          const T* __restrict__ z, const K* __restrict__ keysIn,
                                                          size_t n, DBox<T> box, const uint16_t* __restrict__ encTable,
                                                          ResortArgs<K> ra, T* __restrict__ extentPartials)
{
    // (larger stages -- fewer atomics on the list's counter -- cost more in residency than they save: 4x the stage
    //  +0.04 ms, 8x +0.24 ms at 1e8 particles)
    constexpr unsigned STAGE = 64 * VEC * 2;
    __shared__ uint16_t enc2[HILBERT ? HILBERT_STATES * 64 : 1];
    __shared__ K stageKey[4][STAGE];
    __shared__ uint32_t stageIdx[4][STAGE];
    T ext[6] = {std::numeric_limits<T>::infinity(),  -std::numeric_limits<T>::infinity(),
                std::numeric_limits<T>::infinity(),  -std::numeric_limits<T>::infinity(),
                std::numeric_limits<T>::infinity(),  -std::numeric_limits<T>::infinity()};
    auto widen = [&](T xv, T yv, T zv)
    {
        ext[0] = xv < ext[0] ? xv : ext[0], ext[1] = xv > ext[1] ? xv : ext[1];
        ext[2] = yv < ext[2] ? yv : ext[2], ext[3] = yv > ext[3] ? yv : ext[3];
        ext[4] = zv < ext[4] ? zv : ext[4], ext[5] = zv > ext[5] ? zv : ext[5];
    };
    if (HILBERT) loadHilbertPairs(enc2, encTable);
    __syncthreads();
    constexpr unsigned g = 1u << maxLevel<K>();
    const T mx = g * box.inv[0], my = g * box.inv[1], mz = g * box.inv[2]; // R/sfc/sfc.hpp:188-194
    const T sx = box.lo[0] * mx, sy = box.lo[1] * my, sz = box.lo[2] * mz;
    const unsigned lane   = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint64_t below  = (1ull << lane) - 1;
    unsigned staged       = 0; // wave-uniform

    auto flush = [&]()
    {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(ra.moverCount, staged);
        base = __builtin_amdgcn_readfirstlane(base);
        for (unsigned k = lane; k < staged; k += 64)
        {
            if (base + k < ra.moverCap)
            {
                ra.moverKeys[base + k] = stageKey[w][k];
                ra.moverIdx[base + k]  = stageIdx[w][k];
            }
        }
        staged = 0;
    };
    // is the particle at position p still inside the leaf that position belonged to?  If not: count and stage it;
    // returns what goes to keysOut: the key of a stayer, a hole (~0: sorts behind everything) for a mover
    // (word, rank: the leaf-start bits of the 64 positions around p and the number of starts in front of them)
    auto classify = [&](K key, size_t p, bool valid, uint64_t word, uint32_t rank) -> K
    {
        bool mover = false;
        if (valid)
        {
            const uint32_t j = rank + uint32_t(__popcll(word & ((2ull << (p & 63)) - 1))) - 1u;
            const K lo = ra.leafLo[j], hi = ra.leafLo[j + 1];
            mover = !(key >= lo && key < hi);
            if (mover) atomicAdd(&ra.outCount[j], 1u);
        }
        const uint64_t mm = __ballot(mover);
        if (mm)
        {
            // flush only when these movers would not fit any more: the stage is nearly full at every flush, which halves
            // the number of atomics on the list's ONE counter (tens of thousands of them serialise in the L2)
            if (staged + unsigned(__popcll(mm)) > STAGE) flush();
            if (mover)
            {
                const unsigned off = staged + unsigned(__popcll(mm & below));
                stageKey[w][off]   = key;
                stageIdx[w][off]   = uint32_t(p);
            }
            staged += unsigned(__popcll(mm));
        }
        return mover ? ~K(0) : key;
    };

    const size_t nVec   = n / VEC;
    const size_t stride = size_t(gridDim.x) * 256;
    const size_t iters  = (nVec + stride - 1) / stride;
    size_t vi           = size_t(blockIdx.x) * 256 + threadIdx.x;
    // what an iteration reads: requested one iteration ahead, so that a wave computes the keys of one vector while
    // the loads of the next are in flight.  (Worth 1-2 % only: four fifths of this kernel's traffic are reads, and
    // read-dominated streams top out near 4.7 TB/s on this machine -- the pure read of minMaxPartialKernel runs at 4.6 --
    // whatever the arithmetic costs; halving the VALU and LDS work with the two-level Hilbert table did not move it.)
    struct In
    {
        T x[VEC], y[VEC], z[VEC];
        K k[VEC];
        uint64_t word;
        uint32_t rank;
    };
    auto fetch = [&](size_t at, In& in)
    {
        const size_t base = at * VEC;
        __builtin_memcpy(in.x, __builtin_assume_aligned(x + base, sizeof(T) * VEC), sizeof in.x);
        __builtin_memcpy(in.y, __builtin_assume_aligned(y + base, sizeof(T) * VEC), sizeof in.y);
        __builtin_memcpy(in.z, __builtin_assume_aligned(z + base, sizeof(T) * VEC), sizeof in.z);
        // keysIn == nullptr: the caller has no key array, i.e. no remove markers
        if (keysIn) { __builtin_memcpy(in.k, __builtin_assume_aligned(keysIn + base, sizeof(K) * VEC), sizeof in.k); }
        else
        {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                in.k[v] = 0;
        }
        // the VEC positions of a lane lie in one 64-position word of the leaf table (base is a multiple of VEC)
        in.word = ra.leafStart[base >> 6];
        in.rank = ra.leafRank[base >> 6];
    };
    In next{};
    if (vi < nVec) fetch(vi, next);
    for (size_t it = 0; it < iters; ++it, vi += stride)
    {
        const bool valid  = vi < nVec;
        const size_t base = vi * VEC;
        const In in       = next;
        if (vi + stride < nVec) fetch(vi + stride, next);
        K out[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            out[v] = 0;
        if (valid)
        {
            keysOfPositions<K, T, VEC, HILBERT>(in.x, in.y, in.z, mx, my, mz, sx, sy, sz, enc2, out);
            if (extentPartials)
            {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    widen(in.x[v], in.y[v], in.z[v]);
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (in.k[v] == endKey<K>()) out[v] = in.k[v]; // particles flagged for removal keep their marker
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            out[v] = classify(out[v], base + v, valid, valid ? in.word : 0, valid ? in.rank : 0);
        if (valid) __builtin_memcpy(__builtin_assume_aligned(ra.keysOut + base, sizeof(K) * VEC), out, sizeof out);
    }
    // elements behind the last full vector: first wave of block 0
    if (blockIdx.x == 0 && threadIdx.x < 64)
    {
        size_t i   = nVec * VEC + threadIdx.x;
        bool valid = i < n;
        K key      = 0;
        if (valid)
        {
            K m = keyOfPosition<K, T, HILBERT>(x[i], y[i], z[i], mx, my, mz, sx, sy, sz, enc2);
            key = (keysIn && keysIn[i] == endKey<K>()) ? endKey<K>() : m;
            if (extentPartials) widen(x[i], y[i], z[i]);
        }
        key = classify(key, i, valid, valid ? ra.leafStart[i >> 6] : 0, valid ? ra.leafRank[i >> 6] : 0);
        if (valid) ra.keysOut[i] = key;
    }
    if (staged) flush();
    if (extentPartials) foldBlockExtents<T>(ext, extentPartials);
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

//! workgroups per CU of the grid-stride encode kernels: as many as a CU holds at a time (the registers of
//! encodeHistogramKernel allow six waves per SIMD, those of encodeResortKernel with its loads in flight four) -- with
//! more, the second round of workgroups fills the CUs only partly, and every workgroup pays its set-up (Hilbert table,
//! extents fold, mover flush): 0.19 -> 0.15 ms at 1.25e7 particles, the same at 1e8.  CSTONE_ENCODE_BLOCKS overrides.
inline size_t encodeBlocksPerCu(size_t resident)
{
    static const size_t v = []
    {
        const char* e = std::getenv("CSTONE_ENCODE_BLOCKS");
        return e ? size_t(std::strtoull(e, nullptr, 10)) : size_t(0);
    }();
    return v ? v : resident;
}

template<class K, class T>
int computeKeysHist(cstone_hip_ctx* ctx, int curve, const T* x, const T* y, const T* z, K* keys, size_t n,
                    const cstone_box& hostBox, uint32_t* hist, bool* fused, int firstDigit, bool honourMarkers,
                    void* extentsOut /* device T[6] or nullptr; only written when *fused */)
{
    constexpr int VEC = 16 / sizeof(T);
    bool aligned = (uintptr_t(x) % 16 == 0) && (uintptr_t(y) % 16 == 0) && (uintptr_t(z) % 16 == 0) &&
                   (uintptr_t(keys) % (sizeof(K) * VEC) == 0);
    *fused = aligned && n > 0;
    if (!*fused)
    {
        // the plain encode kernel always looks for remove markers: without any, give it a cleared array
        if (!honourMarkers && n) CS_HIP(ctx, hipMemsetAsync(keys, 0, n * sizeof(K), ctx->stream));
        return computeKeys<K, T>(ctx, curve, x, y, z, keys, n, hostBox); // the sort counts on its own
    }
    StageTimer timer(ctx, CSTONE_STAGE_ENCODE);
    DBox<T> box   = makeDBox<T>(hostBox);
    auto* enc     = (const uint16_t*)ctx->hilbertTables;
    size_t nVec   = n / VEC;
    unsigned grid = unsigned(std::max<size_t>(1, std::min<size_t>(size_t(ctx->numCu) * encodeBlocksPerCu(6), (nVec + 255) / 256)));
    T* partials   = nullptr;
    if (extentsOut)
    {
        CS_TRY(arenaReserve(ctx, alignUp(size_t(grid) * 6 * sizeof(T)) + 256));
        partials = (T*)arenaTake(ctx, size_t(grid) * 6 * sizeof(T));
    }
    if (curve == CSTONE_HILBERT)
        hipLaunchKernelGGL((encodeHistogramKernel<K, T, VEC, true>), grid, 256, 0, ctx->stream, x, y, z, keys, n, box,
                           enc, hist, firstDigit, honourMarkers, partials);
    else
        hipLaunchKernelGGL((encodeHistogramKernel<K, T, VEC, false>), grid, 256, 0, ctx->stream, x, y, z, keys, n, box,
                           enc, hist, firstDigit, honourMarkers, partials);
    if (extentsOut)
    {
        hipLaunchKernelGGL(foldExtentsKernel<T>, 1, 256, 0, ctx->stream, partials, grid, (T*)extentsOut);
        arenaReset(ctx); // later launches reuse the slice behind these in stream order
    }
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int computeKeysAndHistogram(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* x, const void* y,
                            const void* z, void* keys, size_t n, const cstone_box& box, uint32_t* hist, bool* fused,
                            int firstDigit, bool honourMarkers, void* extentsOut)
{
    if (key_bits == 32 && real_bits == 32)
        return computeKeysHist<uint32_t, float>(ctx, curve, (const float*)x, (const float*)y, (const float*)z,
                                                (uint32_t*)keys, n, box, hist, fused, firstDigit, honourMarkers, extentsOut);
    if (key_bits == 32 && real_bits == 64)
        return computeKeysHist<uint32_t, double>(ctx, curve, (const double*)x, (const double*)y, (const double*)z,
                                                 (uint32_t*)keys, n, box, hist, fused, firstDigit, honourMarkers, extentsOut);
    if (key_bits == 64 && real_bits == 32)
        return computeKeysHist<uint64_t, float>(ctx, curve, (const float*)x, (const float*)y, (const float*)z,
                                                (uint64_t*)keys, n, box, hist, fused, firstDigit, honourMarkers, extentsOut);
    if (key_bits == 64 && real_bits == 64)
        return computeKeysHist<uint64_t, double>(ctx, curve, (const double*)x, (const double*)y, (const double*)z,
                                                 (uint64_t*)keys, n, box, hist, fused, firstDigit, honourMarkers, extentsOut);
    return fail(ctx, CSTONE_E_ARG, "sfc_keys_and_ordering: unsupported type combination");
}

template<class K, class T>
static int computeKeysResortT(cstone_hip_ctx* ctx, int curve, const T* x, const T* y, const T* z, const K* keysIn,
                              size_t n, const cstone_box& hostBox, const ResortArgs<K>& ra, void* extentsOut, bool* done)
{
    constexpr int VEC = 16 / sizeof(T);
    bool aligned = (uintptr_t(x) % 16 == 0) && (uintptr_t(y) % 16 == 0) && (uintptr_t(z) % 16 == 0) &&
                   (uintptr_t(keysIn) % (sizeof(K) * VEC) == 0) && (uintptr_t(ra.keysOut) % (sizeof(K) * VEC) == 0);
    // (keysIn may be null: no key array of the caller's, no remove markers)
    *done = aligned && n > 0;
    if (!*done) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_ENCODE);
    DBox<T> box   = makeDBox<T>(hostBox);
    auto* enc     = (const uint16_t*)ctx->hilbertTables;
    size_t nVec   = n / VEC;
    unsigned grid = unsigned(std::max<size_t>(1, std::min<size_t>(size_t(ctx->numCu) * encodeBlocksPerCu(4), (nVec + 255) / 256)));
    T* partials   = nullptr;
    if (extentsOut)
    {
        CS_TRY(arenaReserve(ctx, alignUp(size_t(grid) * 6 * sizeof(T)) + 256));
        partials = (T*)arenaTake(ctx, size_t(grid) * 6 * sizeof(T));
    }
    if (curve == CSTONE_HILBERT)
        hipLaunchKernelGGL((encodeResortKernel<K, T, VEC, true>), grid, 256, 0, ctx->stream, x, y, z, keysIn, n, box, enc,
                           ra, partials);
    else
        hipLaunchKernelGGL((encodeResortKernel<K, T, VEC, false>), grid, 256, 0, ctx->stream, x, y, z, keysIn, n, box,
                           enc, ra, partials);
    if (extentsOut)
    {
        hipLaunchKernelGGL(foldExtentsKernel<T>, 1, 256, 0, ctx->stream, partials, grid, (T*)extentsOut);
        arenaReset(ctx);
    }
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int computeKeysResort(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* x, const void* y,
                      const void* z, const void* keysIn, size_t n, const cstone_box& box, const void* resortArgs,
                      void* extentsOut, bool* done)
{
    if (key_bits == 32 && real_bits == 32)
        return computeKeysResortT<uint32_t, float>(ctx, curve, (const float*)x, (const float*)y, (const float*)z,
                                                   (const uint32_t*)keysIn, n, box,
                                                   *(const ResortArgs<uint32_t>*)resortArgs, extentsOut, done);
    if (key_bits == 32 && real_bits == 64)
        return computeKeysResortT<uint32_t, double>(ctx, curve, (const double*)x, (const double*)y, (const double*)z,
                                                    (const uint32_t*)keysIn, n, box,
                                                    *(const ResortArgs<uint32_t>*)resortArgs, extentsOut, done);
    if (key_bits == 64 && real_bits == 32)
        return computeKeysResortT<uint64_t, float>(ctx, curve, (const float*)x, (const float*)y, (const float*)z,
                                                   (const uint64_t*)keysIn, n, box,
                                                   *(const ResortArgs<uint64_t>*)resortArgs, extentsOut, done);
    if (key_bits == 64 && real_bits == 64)
        return computeKeysResortT<uint64_t, double>(ctx, curve, (const double*)x, (const double*)y, (const double*)z,
                                                    (const uint64_t*)keysIn, n, box,
                                                    *(const ResortArgs<uint64_t>*)resortArgs, extentsOut, done);
    return fail(ctx, CSTONE_E_ARG, "resort encode: unsupported type combination");
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
