// Halo discovery on gfx950.  Replaces segmentMax + scaleGpu as used by Halos::discover
// (R/halos/halos.hpp:128-189) and findHalosGpu (R/traversal/collisions_gpu.cu:40-104; semantics
// R/traversal/collisions.hpp:40-105, boxoverlap.hpp:42-182, traversal.hpp:69-110).
//
// The reference walks the tree with ONE THREAD per local leaf and a 128-entry private stack, and
// decodes a Hilbert key per visited node with a 21-iteration bit loop.  Here:
//   * a wave takes 64 consecutive local leaves; each lane builds its leaf's halo box and runs the
//     "is the halo box inside my own key range?" rejection (2 transducer encodes)
//   * for every leaf that survives (ballot), the WHOLE WAVE traverses cooperatively: up to 8 nodes
//     are popped from an LDS stack per step and their 64 children are tested in parallel, one per
//     lane; children to descend into are pushed back with a ballot/popcount compaction
//   * node keys are decoded with the Hilbert transducer table in LDS, only `level` digits deep
// Flags are idempotent stores of 1, so the traversal order is free (results are order-independent).
#include "ctx.hpp"
#include "device_keys.hpp"

namespace cship
{

namespace
{

constexpr int HALO_WAVES = 4;    // waves per workgroup
constexpr int STACK_CAP  = 1024; // node indices per wave (4 KB of LDS)

__device__ __forceinline__ bool rangesOverlap(int a, int b, int c, int d) { return b > c && d > a; }

//! periodic overlap on a ring of circumference R, R/traversal/boxoverlap.hpp:57-71
__device__ __forceinline__ bool ringOverlap(int R, int a, int b, int c, int d)
{
    return rangesOverlap(a, b, c, d) || rangesOverlap(a + R, b + R, c, d) || rangesOverlap(a, b, c + R, d + R);
}

//! lower corner of the level-`level` node starting at `key`; decodes only the digits that matter
template<class K, bool HILBERT>
__device__ __forceinline__ void nodeCorner(K key, unsigned level, const uint16_t* dec, int& x, int& y, int& z)
{
    K morton = key;
    if (HILBERT)
    {
        morton         = 0;
        unsigned state = 0;
        for (unsigned l = 1; l <= level; ++l)
        {
            unsigned e = dec[state * 8 + octDigit(key, l)];
            morton |= K(e & 7u) << (3u * (maxLevel<K>() - l));
            state = e >> 3;
        }
    }
    unsigned ix, iy, iz;
    mortonDecode<K>(morton, ix, iy, iz);
    x = int(ix), y = int(iy), z = int(iz);
}

template<class K, bool HILBERT>
__device__ __forceinline__ K encodeCurve(unsigned ix, unsigned iy, unsigned iz, const uint16_t* enc)
{
    K m = mortonEncode<K>(ix, iy, iz);
    return HILBERT ? hilbertFromMorton<K>(m, enc) : m;
}

//! x in [0,1) -> grid units, rounding up, clamped: toNBitIntCeil, R/sfc/common.hpp:78-88
template<class K, class T>
__device__ __forceinline__ int toGridCeil(T x)
{
    constexpr unsigned nb = maxLevel<K>();
    unsigned r            = unsigned(ceil(x * T(1u << nb)));
    return int(min(r, (1u << nb) - 1u));
}

/*! MODE 0 (the reference's findHalos): targets are the radius-dilated boxes of leaves [first,last); tree nodes inside
 *         the own key range [leaves[first], leaves[last]) are skipped, every other overlapped leaf is flagged
 *  MODE 1 (export): only compute the dilated boxes of leaves [first,last): boxes[k][0..5] = lo/hi per axis,
 *         boxes[k][6] = 1 if the box is NOT contained in the own key range (it needs halos from other ranks), [7] = 0
 *  MODE 3 (export, proven): like MODE 1, but boxes[k][6] = 1 only if the box really overlaps a leaf OUTSIDE the own key
 *         range of this tree (the walk of MODE 0, stopped at the first such leaf): the enclosing-node test of MODE 1
 *         also exports every box that straddles a coarse octree boundary deep inside the own domain
 *  MODE 2 (serve): targets are numTargets foreign boxes (same 8-int records, record[6] == 0 are ignored); only nodes
 *         that intersect the own key range are visited and own leaves overlapped by a target are flagged: the owner
 *         of the particles answers "which of my leaves does your halo region touch" on its own, finest tree */
template<class K, class T, bool HILBERT, int MODE>
__global__ __launch_bounds__(HALO_WAVES * 64) void findHalosKernel(
    const K* __restrict__ prefixes, const NodeIdx* __restrict__ childOffsets, const NodeIdx* __restrict__ internalToLeaf,
    const K* __restrict__ leaves, const float* __restrict__ radii, DBox<T> box, NodeIdx first, NodeIdx last,
    int* __restrict__ flags, const uint16_t* __restrict__ tables, int* __restrict__ errors,
    const int* __restrict__ boxesIn, NodeIdx numTargets, int* __restrict__ boxesOut)
{
    __shared__ uint16_t enc[24 * 8];
    __shared__ uint16_t dec[24 * 8];
    __shared__ NodeIdx stacks[HALO_WAVES][STACK_CAP];
    if (threadIdx.x < 24 * 8)
    {
        enc[threadIdx.x] = tables[threadIdx.x];
        dec[threadIdx.x] = tables[48 * 8 + threadIdx.x];
    }
    __syncthreads();

    constexpr int R     = 1 << maxLevel<K>();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    NodeIdx* stack      = stacks[wave];
    const K lowest = leaves[first], highest = leaves[last];

    const NodeIdx slot0 = NodeIdx(blockIdx.x * (HALO_WAVES * 64) + threadIdx.x);
    NodeIdx leaf = first + slot0;
    bool active  = MODE == 2 ? slot0 < numTargets : leaf < last;

    // ---- per lane: halo box of my leaf + containment rejection (collisions.hpp:91-98)
    int lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
    int mark  = 1; // MODE 2: the bit a record sets in the flags = 1 << record[7] (the exporting rank, 0 by default)
    if (MODE == 2)
    {
        if (active)
        {
            const int* rec = boxesIn + size_t(slot0) * 8;
#pragma unroll
            for (int d = 0; d < 3; ++d)
                lo[d] = rec[2 * d], hi[d] = rec[2 * d + 1];
            active = rec[6] != 0;
            mark   = 1 << (rec[7] & 31);
        }
    }
    else if (active)
    {
        K start        = leaves[leaf];
        unsigned level = levelOfSpan<K>(leaves[leaf + 1] - start);
        int c[3];
        nodeCorner<K, HILBERT>(start, level, dec, c[0], c[1], c[2]);
        int edge     = 1 << (maxLevel<K>() - level);
        float radius = radii[leaf];
#pragma unroll
        for (int d = 0; d < 3; ++d)
        {
            int delta = toGridCeil<K>(radius * box.inv[d]); // float * T, evaluated in T like the reference
            int a = c[d] - delta, b = c[d] + edge + delta;
            bool pbc = box.bc[d] == 1;
            lo[d]    = pbc ? a : min(max(0, a), R);
            hi[d]    = pbc ? b : min(max(0, b), R);
        }
        bool inside;
        if (min(min(lo[0], lo[1]), lo[2]) < 0 || max(max(hi[0], hi[1]), hi[2]) > R)
        {
            inside = lowest == 0 && highest == endKey<K>();
        }
        else
        {
            K kLo = encodeCurve<K, HILBERT>(lo[0], lo[1], lo[2], enc);
            K kHi = encodeCurve<K, HILBERT>(hi[0] - 1, hi[1] - 1, hi[2] - 1, enc);
            unsigned common = unsigned(sharedPrefixBits<K>(kLo, kHi)) / 3u;
            K nodeStart     = kLo & ~K(nodeSpan<K>(common) - 1);
            inside          = nodeStart >= lowest && nodeStart + nodeSpan<K>(common) <= highest;
        }
        active = !inside;
        if (MODE == 1 || MODE == 3)
        {
            int* rec = boxesOut + size_t(slot0) * 8;
#pragma unroll
            for (int d = 0; d < 3; ++d)
                rec[2 * d] = lo[d], rec[2 * d + 1] = hi[d];
            rec[6] = (MODE == 1 && active) ? 1 : 0; // MODE 3: set by the walk below
            rec[7] = 0;
        }
    }
    if (MODE == 1) return;

    // ---- cooperative traversal, one surviving leaf at a time
    uint64_t todo = __ballot(active);
    while (todo)
    {
        int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        int tlo[3], thi[3];
#pragma unroll
        for (int d = 0; d < 3; ++d)
        {
            tlo[d] = __shfl(lo[d], src);
            thi[d] = __shfl(hi[d], src);
        }
        const int tmark = __shfl(mark, src);

        auto descend = [&](NodeIdx n, bool& isLeaf) -> bool
        {
            K prefix       = prefixes[n];
            K start        = fromPrefix(prefix);
            unsigned level = prefixBits(prefix) / 3;
            isLeaf         = childOffsets[n] == 0;
            K end          = start + nodeSpan<K>(level);
            if ((MODE == 0 || MODE == 3) && !(start < lowest || end > highest)) return false; // inside my own range: nothing to find
            if (MODE == 2 && (end <= lowest || start >= highest)) return false; // not mine: the owner serves it
            int c[3];
            nodeCorner<K, HILBERT>(start, level, dec, c[0], c[1], c[2]);
            int edge = 1 << (maxLevel<K>() - level);
            return ringOverlap(R, c[0], c[0] + edge, tlo[0], thi[0]) && ringOverlap(R, c[1], c[1] + edge, tlo[1], thi[1]) &&
                   ringOverlap(R, c[2], c[2] + edge, tlo[2], thi[2]);
        };

        // root (traversal.hpp:71-78)
        int top = 0;
        {
            bool rootLeaf = false;
            bool go       = false;
            if (lane == 0) go = descend(0, rootLeaf);
            go       = __shfl(int(go), 0);
            rootLeaf = __shfl(int(rootLeaf), 0);
            if (!go) continue;
            if (rootLeaf)
            {
                if (lane == 0)
                {
                    if (MODE == 2) atomicOr(&flags[internalToLeaf[0]], tmark);
                    else if (MODE == 3) boxesOut[size_t(blockIdx.x * (HALO_WAVES * 64) + wave * 64 + src) * 8 + 6] = 1;
                    else flags[internalToLeaf[0]] = 1;
                }
                continue;
            }
            if (lane == 0) stack[0] = 0;
            top = 1;
        }
        while (top > 0)
        {
            int take    = min(top, 8);
            int slot    = int(lane >> 3);
            bool mine   = slot < take;
            NodeIdx par = mine ? stack[top - 1 - slot] : 0; // LDS reads of the current wave: in program order
            top -= take;
            bool isLeaf = false, go = false;
            NodeIdx child = 0;
            if (mine)
            {
                child = childOffsets[par] + NodeIdx(lane & 7u);
                go    = descend(child, isLeaf);
            }
            if (MODE == 3)
            {
                // one foreign leaf is proof enough: mark the record and leave this target
                if (__any(go && isLeaf))
                {
                    if (lane == 0) boxesOut[size_t(blockIdx.x * (HALO_WAVES * 64) + wave * 64 + src) * 8 + 6] = 1;
                    top = 0;
                    break;
                }
            }
            else if (go && isLeaf)
            {
                if (MODE == 2) atomicOr(&flags[internalToLeaf[child]], tmark);
                else flags[internalToLeaf[child]] = 1;
            }
            bool push     = go && !isLeaf;
            uint64_t pm   = __ballot(push);
            int numPush   = __popcll(pm);
            if (top + numPush > STACK_CAP)
            {
                if (lane == 0) atomicOr(errors, 2);
                top = 0; // give up on this leaf: the sticky error word makes the call fail
                break;
            }
            if (push) stack[top + __popcll(pm & ((1ull << lane) - 1ull))] = child;
            top += numPush;
        }
    }
}

//! radii[leaf] for leaf in [first,last): float(max(h[layout[k]..layout[k+1])) * 2 * ext); 16 lanes per leaf (a leaf
//! holds a few dozen particles: whole waves per leaf would mostly idle and launch 4x as many waves)
constexpr int RADII_LEAVES_PER_BLOCK = 16;
template<class Th>
__global__ __launch_bounds__(256) void haloRadiiKernel(const Th* __restrict__ h, const uint32_t* __restrict__ layout,
                                                       NodeIdx first, NodeIdx last, float ext, float* __restrict__ radii,
                                                       NodeIdx numLeaves)
{
    // the leaves outside [first, last) have no particles here: radius 0 (what used to be a memset of the whole array
    // in front of this launch), spread over the workgroups
    const NodeIdx outside = numLeaves - (last - first);
    for (NodeIdx o = NodeIdx(blockIdx.x) * 256 + NodeIdx(threadIdx.x); o < outside; o += NodeIdx(gridDim.x) * 256)
        radii[o < first ? o : o + (last - first)] = 0.0f;
    const unsigned sub = threadIdx.x & 15u;
    NodeIdx k = NodeIdx(blockIdx.x) * RADII_LEAVES_PER_BLOCK + NodeIdx(threadIdx.x >> 4);
    if (first + k >= last) return; // whole 16-lane groups leave together: the shuffles below stay inside a group
    uint32_t a = layout[k], b = layout[k + 1];
    float out = 0.0f;
    if (b > a)
    {
        // four slots per lane at a time (a leaf holds at most a bucket): the loads of a leaf go out together
        Th m = h[a];
        for (uint32_t i = a + sub; i < b; i += 64)
        {
            const uint32_t i1 = i + 16, i2 = i + 32, i3 = i + 48;
            const Th v0 = h[i];
            const Th v1 = i1 < b ? h[i1] : v0;
            const Th v2 = i2 < b ? h[i2] : v0;
            const Th v3 = i3 < b ? h[i3] : v0;
            const Th m01 = v0 > v1 ? v0 : v1, m23 = v2 > v3 ? v2 : v3;
            const Th m4  = m01 > m23 ? m01 : m23;
            m            = m4 > m ? m4 : m;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
        {
            Th t = __shfl_xor(m, o);
            m    = t > m ? t : m;
        }
        out = float(m * 2 * ext); // Th*int -> Th, then *float in the common type, halos.hpp:176
    }
    if (sub == 0) radii[first + k] = out;
}

//! the same with the gather of h into SFC order in the same pass: hOut[i] = h[order[i]] for the particles of the leaves
//! [first, last) (layout = their offsets, hOut indexed like layout), radii from the values on their way through
template<class Th>
__global__ __launch_bounds__(256) void gatherHaloRadiiKernel(const Th* __restrict__ h, const uint32_t* __restrict__ order,
                                                             Th* __restrict__ hOut, const uint32_t* __restrict__ layout,
                                                             NodeIdx first, NodeIdx last, float ext,
                                                             float* __restrict__ radii)
{
    const unsigned sub = threadIdx.x & 15u;
    NodeIdx k = NodeIdx(blockIdx.x) * RADII_LEAVES_PER_BLOCK + NodeIdx(threadIdx.x >> 4);
    if (first + k >= last) return;
    uint32_t a = layout[k], b = layout[k + 1];
    float out = 0.0f;
    if (b > a)
    {
        // four slots per lane at a time (a leaf holds at most a bucket, typically 30-60 particles): all index loads go
        // out together, then all value loads -- three dependent memory round trips per leaf instead of one per slot
        Th m          = 0;
        bool haveAny  = false;
        for (uint32_t base = a + sub; base < b; base += 64)
        {
            uint32_t idx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                idx[u] = base + 16 * u < b ? order[base + 16 * u] : 0u;
            Th v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + 16 * u < b) v[u] = h[idx[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + 16 * u < b)
                {
                    hOut[base + 16 * u] = v[u];
                    m                   = (!haveAny || v[u] > m) ? v[u] : m;
                    haveAny             = true;
                }
        }
        // lanes without a particle of this leaf hold no candidate: take the first lane's (it always has one)
        Th m0 = __shfl(m, int(threadIdx.x & 63u & ~15u));
        if (!haveAny) m = m0;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
        {
            Th t = __shfl_xor(m, o);
            m    = t > m ? t : m;
        }
        out = float(m * 2 * ext);
    }
    if (sub == 0) radii[first + k] = out;
}

} // namespace

int gatherWithHaloRadii(cstone_hip_ctx* ctx, int h_bits, const void* h, const uint32_t* order, void* hOut,
                        const uint32_t* layout, int numLeaves, float ext, float* radii)
{
    if (numLeaves == 0) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_GATHER_H);
    unsigned grid = gridFor(size_t(numLeaves), RADII_LEAVES_PER_BLOCK);
    if (h_bits == 32)
        hipLaunchKernelGGL(gatherHaloRadiiKernel<float>, grid, 256, 0, ctx->stream, (const float*)h, order, (float*)hOut,
                           layout, 0, numLeaves, ext, radii);
    else
        hipLaunchKernelGGL(gatherHaloRadiiKernel<double>, grid, 256, 0, ctx->stream, (const double*)h, order,
                           (double*)hOut, layout, 0, numLeaves, ext, radii);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // namespace cship

using namespace cship;

extern "C"
{

int cstone_hip_halo_radii(cstone_hip_ctx* ctx, int h_bits, const void* h, const uint32_t* layout, int first, int last,
                          int num_leaves, float ext, float* radii)
{
    if (!ctx || !radii || first < 0 || last < first || last > num_leaves || (last > first && (!h || !layout)))
        return fail(ctx, CSTONE_E_ARG, "halo_radii: bad argument");
    if (num_leaves == 0) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_HALOS);
    if (last == first)
    {
        CS_HIP(ctx, hipMemsetAsync(radii, 0, size_t(num_leaves) * sizeof(float), ctx->stream));
        return CSTONE_OK;
    }
    unsigned grid = gridFor(size_t(last - first), RADII_LEAVES_PER_BLOCK);
    if (h_bits == 32)
        hipLaunchKernelGGL(haloRadiiKernel<float>, grid, 256, 0, ctx->stream, (const float*)h, layout, first, last, ext,
                           radii, num_leaves);
    else if (h_bits == 64)
        hipLaunchKernelGGL(haloRadiiKernel<double>, grid, 256, 0, ctx->stream, (const double*)h, layout, first, last,
                           ext, radii, num_leaves);
    else
        return fail(ctx, CSTONE_E_ARG, "halo_radii: h_bits %d unsupported", h_bits);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_find_halos(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                          const int32_t* child_offsets, const int32_t* internal_to_leaf, const void* leaves,
                          const float* radii, const cstone_box* box_host, int first, int last, int32_t* flags)
{
    if (!ctx || !prefixes || !child_offsets || !internal_to_leaf || !leaves || !radii || !box_host || !flags ||
        first < 0 || last < first)
        return fail(ctx, CSTONE_E_ARG, "find_halos: bad argument");
    if (last == first) return CSTONE_OK;
    StageTimer timer(ctx, CSTONE_STAGE_HALOS);
    unsigned grid = gridFor(size_t(last - first), HALO_WAVES * 64);
    auto* tables  = (const uint16_t*)ctx->hilbertTables;
    int* errors   = ctx->devScalars + 63;
#define CS_LAUNCH_HALOS(K, T)                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        if (curve == CSTONE_HILBERT)                                                                                   \
            hipLaunchKernelGGL((findHalosKernel<K, T, true, 0>), grid, HALO_WAVES * 64, 0, ctx->stream,                \
                               (const K*)prefixes, child_offsets, internal_to_leaf, (const K*)leaves, radii,           \
                               makeDBox<T>(*box_host), first, last, flags, tables, errors, nullptr, 0, nullptr);       \
        else                                                                                                           \
            hipLaunchKernelGGL((findHalosKernel<K, T, false, 0>), grid, HALO_WAVES * 64, 0, ctx->stream,               \
                               (const K*)prefixes, child_offsets, internal_to_leaf, (const K*)leaves, radii,           \
                               makeDBox<T>(*box_host), first, last, flags, tables, errors, nullptr, 0, nullptr);       \
    } while (0)
    if (curve != CSTONE_MORTON && curve != CSTONE_HILBERT) return fail(ctx, CSTONE_E_ARG, "find_halos: bad curve");
    if (key_bits == 32 && real_bits == 32) CS_LAUNCH_HALOS(uint32_t, float);
    else if (key_bits == 32 && real_bits == 64) CS_LAUNCH_HALOS(uint32_t, double);
    else if (key_bits == 64 && real_bits == 32) CS_LAUNCH_HALOS(uint64_t, float);
    else if (key_bits == 64 && real_bits == 64) CS_LAUNCH_HALOS(uint64_t, double);
    else return fail(ctx, CSTONE_E_ARG, "find_halos: unsupported type combination");
#undef CS_LAUNCH_HALOS
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

/* ---- building blocks of the multi-rank halo exchange (owner-side discovery) ---- */

int cstone_hip_halo_boxes_foreign(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* prefixes,
                                  const int32_t* child_offsets, const int32_t* internal_to_leaf, const void* leaves,
                                  const float* radii, const cstone_box* box_host, int first, int last, int32_t* boxes)
{
    if (!ctx || !prefixes || !child_offsets || !internal_to_leaf || !leaves || !radii || !box_host || !boxes ||
        first < 0 || last < first)
        return fail(ctx, CSTONE_E_ARG, "halo_boxes_foreign: bad argument");
    if (last == first) return CSTONE_OK;
    if (curve != CSTONE_MORTON && curve != CSTONE_HILBERT) return fail(ctx, CSTONE_E_ARG, "halo_boxes_foreign: bad curve");
    StageTimer timer(ctx, CSTONE_STAGE_HALOS);
    unsigned grid = gridFor(size_t(last - first), HALO_WAVES * 64);
    auto* tables  = (const uint16_t*)ctx->hilbertTables;
    int* errors   = ctx->devScalars + 63;
#define CS_LAUNCH_BOXES3(K, T)                                                                                         \
    do                                                                                                                 \
    {                                                                                                                  \
        if (curve == CSTONE_HILBERT)                                                                                   \
            hipLaunchKernelGGL((findHalosKernel<K, T, true, 3>), grid, HALO_WAVES * 64, 0, ctx->stream,                \
                               (const K*)prefixes, child_offsets, internal_to_leaf, (const K*)leaves, radii,           \
                               makeDBox<T>(*box_host), first, last, nullptr, tables, errors, nullptr, 0, boxes);       \
        else                                                                                                           \
            hipLaunchKernelGGL((findHalosKernel<K, T, false, 3>), grid, HALO_WAVES * 64, 0, ctx->stream,               \
                               (const K*)prefixes, child_offsets, internal_to_leaf, (const K*)leaves, radii,           \
                               makeDBox<T>(*box_host), first, last, nullptr, tables, errors, nullptr, 0, boxes);       \
    } while (0)
    if (key_bits == 32 && real_bits == 32) CS_LAUNCH_BOXES3(uint32_t, float);
    else if (key_bits == 32 && real_bits == 64) CS_LAUNCH_BOXES3(uint32_t, double);
    else if (key_bits == 64 && real_bits == 32) CS_LAUNCH_BOXES3(uint64_t, float);
    else if (key_bits == 64 && real_bits == 64) CS_LAUNCH_BOXES3(uint64_t, double);
    else return fail(ctx, CSTONE_E_ARG, "halo_boxes_foreign: unsupported type combination");
#undef CS_LAUNCH_BOXES3
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_halo_boxes(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* leaves,
                          const float* radii, const cstone_box* box_host, int first, int last, int32_t* boxes)
{
    if (!ctx || !leaves || !radii || !box_host || !boxes || first < 0 || last < first)
        return fail(ctx, CSTONE_E_ARG, "halo_boxes: bad argument");
    if (last == first) return CSTONE_OK;
    if (curve != CSTONE_MORTON && curve != CSTONE_HILBERT) return fail(ctx, CSTONE_E_ARG, "halo_boxes: bad curve");
    StageTimer timer(ctx, CSTONE_STAGE_HALOS);
    unsigned grid = gridFor(size_t(last - first), HALO_WAVES * 64);
    auto* tables  = (const uint16_t*)ctx->hilbertTables;
    int* errors   = ctx->devScalars + 63;
#define CS_LAUNCH_BOXES(K, T)                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        if (curve == CSTONE_HILBERT)                                                                                   \
            hipLaunchKernelGGL((findHalosKernel<K, T, true, 1>), grid, HALO_WAVES * 64, 0, ctx->stream, nullptr,       \
                               nullptr, nullptr, (const K*)leaves, radii, makeDBox<T>(*box_host), first, last,         \
                               nullptr, tables, errors, nullptr, 0, boxes);                                            \
        else                                                                                                           \
            hipLaunchKernelGGL((findHalosKernel<K, T, false, 1>), grid, HALO_WAVES * 64, 0, ctx->stream, nullptr,      \
                               nullptr, nullptr, (const K*)leaves, radii, makeDBox<T>(*box_host), first, last,         \
                               nullptr, tables, errors, nullptr, 0, boxes);                                            \
    } while (0)
    if (key_bits == 32 && real_bits == 32) CS_LAUNCH_BOXES(uint32_t, float);
    else if (key_bits == 32 && real_bits == 64) CS_LAUNCH_BOXES(uint32_t, double);
    else if (key_bits == 64 && real_bits == 32) CS_LAUNCH_BOXES(uint64_t, float);
    else if (key_bits == 64 && real_bits == 64) CS_LAUNCH_BOXES(uint64_t, double);
    else return fail(ctx, CSTONE_E_ARG, "halo_boxes: unsupported type combination");
#undef CS_LAUNCH_BOXES
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int cstone_hip_find_overlaps(cstone_hip_ctx* ctx, int curve, int key_bits, const void* prefixes,
                             const int32_t* child_offsets, const int32_t* internal_to_leaf, const void* leaves,
                             const int32_t* boxes, int num_boxes, int first, int last, int32_t* flags)
{
    if (!ctx || !prefixes || !child_offsets || !internal_to_leaf || !leaves || !flags || num_boxes < 0 || first < 0 ||
        last < first || (num_boxes && !boxes))
        return fail(ctx, CSTONE_E_ARG, "find_overlaps: bad argument");
    if (num_boxes == 0 || last == first) return CSTONE_OK;
    if (curve != CSTONE_MORTON && curve != CSTONE_HILBERT) return fail(ctx, CSTONE_E_ARG, "find_overlaps: bad curve");
    StageTimer timer(ctx, CSTONE_STAGE_HALOS);
    unsigned grid = gridFor(size_t(num_boxes), HALO_WAVES * 64);
    auto* tables  = (const uint16_t*)ctx->hilbertTables;
    int* errors   = ctx->devScalars + 63;
    DBox<float> unused{};
#define CS_LAUNCH_SERVE(K)                                                                                             \
    do                                                                                                                 \
    {                                                                                                                  \
        if (curve == CSTONE_HILBERT)                                                                                   \
            hipLaunchKernelGGL((findHalosKernel<K, float, true, 2>), grid, HALO_WAVES * 64, 0, ctx->stream,            \
                               (const K*)prefixes, child_offsets, internal_to_leaf, (const K*)leaves, nullptr, unused, \
                               first, last, flags, tables, errors, boxes, num_boxes, nullptr);                         \
        else                                                                                                           \
            hipLaunchKernelGGL((findHalosKernel<K, float, false, 2>), grid, HALO_WAVES * 64, 0, ctx->stream,           \
                               (const K*)prefixes, child_offsets, internal_to_leaf, (const K*)leaves, nullptr, unused, \
                               first, last, flags, tables, errors, boxes, num_boxes, nullptr);                         \
    } while (0)
    if (key_bits == 32) CS_LAUNCH_SERVE(uint32_t);
    else if (key_bits == 64) CS_LAUNCH_SERVE(uint64_t);
    else return fail(ctx, CSTONE_E_ARG, "find_overlaps: key_bits %d unsupported", key_bits);
#undef CS_LAUNCH_SERVE
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

} // extern "C"
