// Device-side key arithmetic shared by the kernels of libcstone_hip (gfx950).
// Semantics follow the reference's HOST_DEVICE_FUN helpers (R = /root/reference/include/cstone);
// the formulations are our own (wave64 / VALU friendly) and are parity-tested against the oracle.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "cstone_hip.h"

namespace cship
{

using NodeIdx  = int32_t;  // TreeNodeIndex, R/tree/definitions.h:41
using LocalIdx = uint32_t; // LocalIndex,    R/tree/definitions.h:43

template<class K>
struct KeyInfo;
template<>
struct KeyInfo<uint32_t>
{
    static constexpr unsigned levels = 10, spare = 2; // R/tree/definitions.h:46-72
};
template<>
struct KeyInfo<uint64_t>
{
    static constexpr unsigned levels = 21, spare = 1;
};

template<class K>
__host__ __device__ constexpr unsigned maxLevel()
{
    return KeyInfo<K>::levels;
}

template<class K>
__host__ __device__ constexpr K nodeSpan(unsigned level) // R/sfc/common.hpp:97-104
{
    return K(1) << (3u * (maxLevel<K>() - level));
}

template<class K>
__host__ __device__ constexpr K endKey() // also the remove marker, R/tree/definitions.h:87-91
{
    return nodeSpan<K>(0);
}

__device__ __forceinline__ int clzKey(uint32_t x) { return x ? __clz(x) : 32; }
__device__ __forceinline__ int clzKey(uint64_t x) { return x ? __clzll(x) : 64; }

template<class K>
__device__ __forceinline__ int sharedPrefixBits(K a, K b) // R/sfc/common.hpp:131-135
{
    return clzKey(K(a ^ b)) - int(KeyInfo<K>::spare);
}

template<class K>
__device__ __forceinline__ unsigned levelOfSpan(K span) // R/sfc/common.hpp:143-148
{
    return (clzKey(K(span - 1)) - KeyInfo<K>::spare) / 3;
}

template<class K>
__device__ __forceinline__ K toPrefix(K key, int nbits) // R/sfc/common.hpp:163-171
{
    return (K(1) << nbits) | (key >> (3 * maxLevel<K>() - nbits));
}

template<class K>
__device__ __forceinline__ unsigned prefixBits(K prefix) // R/sfc/common.hpp:183-187
{
    return 8 * sizeof(K) - 1 - clzKey(prefix);
}

template<class K>
__device__ __forceinline__ K fromPrefix(K prefix) // R/sfc/common.hpp:190-198
{
    unsigned nb = prefixBits(prefix);
    return (prefix ^ (K(1) << nb)) << (3 * maxLevel<K>() - nb);
}

template<class K>
__device__ __forceinline__ unsigned octDigit(K key, unsigned pos) // R/sfc/common.hpp:236-240
{
    return unsigned(key >> (3u * (maxLevel<K>() - pos))) & 7u;
}

// ---------------------------------------------------------------------------------------------------
// bit interleave, R/sfc/morton.hpp:52-128
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread3(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0xFF0000FFu;
    v = (v | (v << 8)) & 0x0F00F00Fu;
    v = (v | (v << 4)) & 0xC30C30C3u;
    v = (v | (v << 2)) & 0x49249249u;
    return v;
}

__device__ __forceinline__ uint64_t spread3(uint64_t v)
{
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x001f00000000ffffull;
    v = (v | v << 16) & 0x001f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

__device__ __forceinline__ uint32_t squeeze3(uint32_t v)
{
    v &= 0x09249249u;
    v = (v ^ (v >> 2)) & 0x030c30c3u;
    v = (v ^ (v >> 4)) & 0x0300f00fu;
    v = (v ^ (v >> 8)) & 0xff0000ffu;
    v = (v ^ (v >> 16)) & 0x000003ffu;
    return v;
}

__device__ __forceinline__ uint32_t squeeze3(uint64_t v)
{
    v &= 0x1249249249249249ull;
    v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ull;
    v = (v ^ (v >> 4)) & 0x100f00f00f00f00full;
    v = (v ^ (v >> 8)) & 0x001f0000ff0000ffull;
    v = (v ^ (v >> 16)) & 0x001f00000000ffffull;
    v = (v ^ (v >> 32)) & 0x00000000001fffffull;
    return uint32_t(v);
}

template<class K>
__device__ __forceinline__ K mortonEncode(unsigned ix, unsigned iy, unsigned iz)
{
    return spread3(K(ix)) * 4 + spread3(K(iy)) * 2 + spread3(K(iz));
}

template<class K>
__device__ __forceinline__ void mortonDecode(K key, unsigned& ix, unsigned& iy, unsigned& iz)
{
    ix = squeeze3(K(key >> 2));
    iy = squeeze3(K(key >> 1));
    iz = squeeze3(key);
}

// ---------------------------------------------------------------------------------------------------
// Hilbert curve as a finite-state transducer over Morton octants.
//
// The reference (R/sfc/hilbert.hpp:58-107) walks the levels top-down and, after emitting a digit,
// reflects/permutes the remaining low bits of (x,y,z).  The accumulated effect of those operations
// is an element of the signed-permutation group of the cube, so the whole curve is the transducer
//     (state, morton octant of the ORIGINAL coordinates at this level) -> (hilbert digit, next state)
// There are at most 48 states; the table is generated on the host by running the reference
// recurrence symbolically (hilbert_tables.hpp) and lives in LDS / constant memory: one table lookup
// per level instead of ~25 data-dependent bit operations.
// Entry layout: bits 0..2 digit, bits 3..8 next state.
// ---------------------------------------------------------------------------------------------------
constexpr int HILBERT_STATES = 24; // reachable states of the transducer (checked when the tables are built)

struct HilbertTables
{
    uint16_t enc[48 * 8]; // [state][morton octant]   -> digit | next<<3
    uint16_t dec[48 * 8]; // [state][hilbert digit]   -> morton octant | next<<3
    // two levels per lookup, straight from the grid coordinates (24 states are reachable):
    // [state][xx | yy << 2 | zz << 4], xx = the coordinate's bits of the two levels -> two digits | next << 6
    uint16_t enc2[HILBERT_STATES * 64];
};

template<class K>
__host__ __device__ __forceinline__ K hilbertFromMorton(K morton, const uint16_t* enc)
{
    K key          = 0;
    unsigned state = 0;
#pragma unroll
    for (int level = int(maxLevel<K>()) - 1; level >= 0; --level)
    {
        unsigned oct = unsigned(morton >> (3 * level)) & 7u;
        unsigned e   = enc[state * 8 + oct];
        key          = (key << 3) | K(e & 7u);
        state        = e >> 3;
    }
    return key;
}

/*! Hilbert keys of VEC grid cells, two levels per table lookup (enc2, in LDS) and straight from the cell coordinates: no
 *  bit interleave, no 64-bit shifts -- the one-level form above costs about ten vector instructions per level and key on
 *  64-bit keys, which made the encode kernels VALU-bound (85 % VALU issue at 60 % of the HBM peak); this one five per two
 *  levels.  The VEC chains are independent and interleave.  Same keys as iHilbert (R/sfc/hilbert.hpp:58-107). */
template<class K, int VEC>
__device__ __forceinline__ void hilbertFromGrid(const unsigned (&ix)[VEC], const unsigned (&iy)[VEC],
                                                const unsigned (&iz)[VEC], const uint16_t* enc2, K (&keys)[VEC])
{
    constexpr int L = int(maxLevel<K>());
    uint32_t acc[VEC][2];
    unsigned st[VEC]; // next state, already multiplied by 64
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        acc[v][0] = acc[v][1] = 0, st[v] = 0;
#pragma unroll
    for (int p = 0; p < L / 2; ++p)
    {
        const int s = L - 2 - 2 * p; // the lower of the two levels
#pragma unroll
        for (int v = 0; v < VEC; ++v)
        {
            const unsigned idx = ((ix[v] >> s) & 3u) | (((iy[v] >> s) & 3u) << 2) | (((iz[v] >> s) & 3u) << 4);
            const unsigned e   = enc2[st[v] | idx];
            acc[v][p / 5]      = (acc[v][p / 5] << 6) | (e & 63u);
            st[v]              = e & 0x7C0u;
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v)
    {
        if constexpr (L & 1)
        {
            // the last level on its own: as the upper level of a pair whose lower level is not looked at
            const unsigned idx = ((ix[v] & 1u) << 1) | ((iy[v] & 1u) << 3) | ((iz[v] & 1u) << 5);
            const unsigned e   = enc2[st[v] | idx];
            keys[v]            = (K(acc[v][0]) << 33) | (K(acc[v][1]) << 3) | K((e >> 3) & 7u);
        }
        else { keys[v] = K(acc[v][0]); }
    }
}

template<class K>
__host__ __device__ __forceinline__ K mortonFromHilbert(K hilbert, const uint16_t* dec)
{
    K key          = 0;
    unsigned state = 0;
#pragma unroll
    for (int level = int(maxLevel<K>()) - 1; level >= 0; --level)
    {
        unsigned dig = unsigned(hilbert >> (3 * level)) & 7u;
        unsigned e   = dec[state * 8 + dig];
        key          = (key << 3) | K(e & 7u);
        state        = e >> 3;
    }
    return key;
}

// ---------------------------------------------------------------------------------------------------
// boxes
// ---------------------------------------------------------------------------------------------------
template<class T>
struct DBox // device image of cstone::Box<T>, R/sfc/box.hpp:112-191, passed to kernels by value
{
    T lo[3], hi[3], len[3], inv[3];
    int bc[3];
};

//! host: build the device box exactly like the Box<T> constructor does (lengths and 1/length in T)
template<class T>
inline DBox<T> makeDBox(const cstone_box& b)
{
    DBox<T> d;
    for (int a = 0; a < 3; ++a)
    {
        d.lo[a]  = T(b.lim[2 * a]);
        d.hi[a]  = T(b.lim[2 * a + 1]);
        d.len[a] = d.hi[a] - d.lo[a];
        d.inv[a] = T(1.) / (d.hi[a] - d.lo[a]); // R/sfc/box.hpp:135
        d.bc[a]  = b.bc[a];
    }
    return d;
}

struct IBox // R/sfc/box.hpp:272-321
{
    int lo[3], hi[3];
};

} // namespace cship
