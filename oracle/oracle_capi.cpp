// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  C entry points (ctypes) onto the CPU restatement in
// cstone_oracle.hpp.  All pointers are HOST pointers.  key_bits in {32,64}, real_bits in {32,64},
// curve 0 = Morton, 1 = Hilbert.  box = {xmin,xmax,ymin,ymax,zmin,zmax}, bc = boundary type per axis.
#include <random>

#include "cstone_oracle.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

namespace
{
template<class T>
Box<T> mkBox(const double* lim, const int* bc)
{
    return Box<T>(T(lim[0]), T(lim[1]), T(lim[2]), T(lim[3]), T(lim[4]), T(lim[5]), bc[0], bc[1], bc[2]);
}

template<class F>
int withKey(int keyBits, F&& f)
{
    if (keyBits == 32) { f(uint32_t{}); }
    else if (keyBits == 64) { f(uint64_t{}); }
    else { return -1; }
    return 0;
}
template<class F>
int withReal(int realBits, F&& f)
{
    if (realBits == 32) { f(float{}); }
    else if (realBits == 64) { f(double{}); }
    else { return -1; }
    return 0;
}
} // namespace

extern "C"
{

int cstone_oracle_compute_sfc_keys(int curve, int key_bits, int real_bits, const void* x, const void* y, const void* z,
                                   void* keys, size_t n, const double* lim, const int* bc)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       withReal(real_bits,
                                [&](auto t)
                                {
                                    using T = decltype(t);
                                    computeKeys<K, T>(Curve(curve), (const T*)x, (const T*)y, (const T*)z, (K*)keys, n,
                                                      mkBox<T>(lim, bc));
                                });
                   });
}

//! scalar helpers for known-answer tests
uint64_t cstone_oracle_encode(int curve, int key_bits, unsigned ix, unsigned iy, unsigned iz)
{
    return key_bits == 32 ? uint64_t(sfcEncode<uint32_t>(Curve(curve), ix, iy, iz))
                          : sfcEncode<uint64_t>(Curve(curve), ix, iy, iz);
}

void cstone_oracle_decode(int curve, int key_bits, uint64_t key, unsigned* out3)
{
    if (key_bits == 32) { sfcDecode<uint32_t>(Curve(curve), uint32_t(key), out3[0], out3[1], out3[2]); }
    else { sfcDecode<uint64_t>(Curve(curve), key, out3[0], out3[1], out3[2]); }
}

void cstone_oracle_node_ibox(int curve, int key_bits, uint64_t key, unsigned level, int* out6)
{
    IBox b = key_bits == 32 ? nodeIBox<uint32_t>(Curve(curve), uint32_t(key), level)
                            : nodeIBox<uint64_t>(Curve(curve), key, level);
    for (int d = 0; d < 3; ++d)
        out6[2 * d] = b.lo[d], out6[2 * d + 1] = b.hi[d];
}

int cstone_oracle_sort_pairs(int key_bits, void* keys, unsigned* vals, size_t n)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       sortByKey<K, unsigned>((K*)keys, vals, n);
                   });
}

int cstone_oracle_gather(int elem_bytes, const unsigned* map, size_t n, const void* src, void* dst)
{
    const char* s = (const char*)src;
    char* d       = (char*)dst;
    for (size_t i = 0; i < n; ++i)
        std::memcpy(d + i * elem_bytes, s + size_t(map[i]) * elem_bytes, elem_bytes);
    return 0;
}

int cstone_oracle_node_counts(int key_bits, const void* tree, unsigned* counts, int num_nodes, const void* keys,
                              size_t n, unsigned max_count)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       nodeCounts<K>((const K*)tree, counts, num_nodes, (const K*)keys, n, max_count);
                   });
}

//! node_ops[num_nodes+1]; returns converged flag through *converged; node_ops left UNSCANNED
int cstone_oracle_node_ops(int key_bits, const void* tree, int num_nodes, const unsigned* counts, unsigned bucket,
                           int* node_ops, int* converged)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K    = decltype(k);
                       *converged = rebalanceDecision<K>((const K*)tree, counts, num_nodes, bucket, node_ops);
                   });
}

/*! one updateOctree step. tree_io holds *num_leaves+1 keys on entry and must have room for cap_leaves+1;
 *  counts_io likewise (cap_leaves). Returns -2 if capacity is too small. */
int cstone_oracle_update_octree(int key_bits, const void* keys, size_t n, unsigned bucket, void* tree_io,
                                unsigned* counts_io, int* num_leaves, int cap_leaves, unsigned max_count,
                                int* converged)
{
    int rc = 0;
    int st = withKey(key_bits,
                     [&](auto k)
                     {
                         using K = decltype(k);
                         std::vector<K> tree((K*)tree_io, (K*)tree_io + *num_leaves + 1);
                         std::vector<unsigned> counts(counts_io, counts_io + *num_leaves);
                         *converged = updateOctree<K>((const K*)keys, n, bucket, tree, counts, max_count);
                         if (int(counts.size()) > cap_leaves)
                         {
                             rc = -2;
                             return;
                         }
                         std::copy(tree.begin(), tree.end(), (K*)tree_io);
                         std::copy(counts.begin(), counts.end(), counts_io);
                         *num_leaves = int(counts.size());
                     });
    return st ? st : rc;
}

//! build from the root until converged; returns number of update iterations via *iterations
int cstone_oracle_compute_octree(int key_bits, const void* keys, size_t n, unsigned bucket, void* tree_out,
                                 unsigned* counts_out, int* num_leaves, int cap_leaves, unsigned max_count,
                                 int* iterations)
{
    int rc = 0;
    int st = withKey(key_bits,
                     [&](auto k)
                     {
                         using K = decltype(k);
                         std::vector<K> tree;
                         std::vector<unsigned> counts;
                         *iterations = computeOctree<K>((const K*)keys, n, bucket, tree, counts, max_count);
                         *num_leaves = int(counts.size());
                         if (int(counts.size()) > cap_leaves)
                         {
                             rc = -2;
                             return;
                         }
                         std::copy(tree.begin(), tree.end(), (K*)tree_out);
                         std::copy(counts.begin(), counts.end(), counts_out);
                     });
    return st ? st : rc;
}

int cstone_oracle_spanning_tree(int key_bits, const void* span_keys, int num_keys, void* tree_out, int cap,
                                int* num_leaves)
{
    int rc = 0;
    int st = withKey(key_bits,
                     [&](auto k)
                     {
                         using K = decltype(k);
                         auto t  = spanningTree<K>((const K*)span_keys, num_keys);
                         *num_leaves = int(t.size()) - 1;
                         if (int(t.size()) > cap + 1)
                         {
                             rc = -2;
                             return;
                         }
                         std::copy(t.begin(), t.end(), (K*)tree_out);
                     });
    return st ? st : rc;
}

/*! linked octree. Array sizes (M = L + (L-1)/7): prefixes[M], child_offsets[M+1], parents[max(1,(M-1)/8)],
 *  level_range[maxLevel+2], internal_to_leaf[M], leaf_to_internal[M] */
int cstone_oracle_build_octree(int key_bits, const void* leaves, int num_leaves, void* prefixes, int* child_offsets,
                               int* parents, int* level_range, int* internal_to_leaf, int* leaf_to_internal)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       LinkedOctree<K> o;
                       buildLinkedOctree<K>((const K*)leaves, num_leaves, o);
                       std::copy(o.prefixes.begin(), o.prefixes.end(), (K*)prefixes);
                       std::copy(o.childOffsets.begin(), o.childOffsets.end(), child_offsets);
                       std::copy(o.parents.begin(), o.parents.end(), parents);
                       std::copy(o.levelRange.begin(), o.levelRange.end(), level_range);
                       std::copy(o.internalToLeaf.begin(), o.internalToLeaf.end(), internal_to_leaf);
                       std::copy(o.leafToInternal.begin(), o.leafToInternal.end(), leaf_to_internal);
                   });
}

int cstone_oracle_upsweep_counts(const int* level_range, int num_level_entries, const int* child_offsets,
                                 unsigned* counts)
{
    upsweepCounts(level_range, num_level_entries, child_offsets, counts);
    return 0;
}

//! layout = (last-first+1) offsets starting at 0 for leaf `first`; radii[num_leaves]
int cstone_oracle_halo_radii(int h_bits, const void* h, const unsigned* layout, int first, int last, int num_leaves,
                             float ext, float* radii)
{
    return withReal(h_bits,
                    [&](auto t)
                    {
                        using Th = decltype(t);
                        haloRadii<Th>((const Th*)h, layout, first, last, num_leaves, ext, radii);
                    });
}

int cstone_oracle_find_halos(int curve, int key_bits, int real_bits, const void* prefixes, const int* child_offsets,
                             const int* internal_to_leaf, const void* leaves, const float* radii, const double* lim,
                             const int* bc, int first, int last, int* flags)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       withReal(real_bits,
                                [&](auto t)
                                {
                                    using T = decltype(t);
                                    findHalos<K, T, float>(Curve(curve), (const K*)prefixes, child_offsets,
                                                           internal_to_leaf, (const K*)leaves, radii,
                                                           mkBox<T>(lim, bc), first, last, flags);
                                });
                   });
}

//! centers/sizes: [num_nodes][3] of real_bits reals
int cstone_oracle_node_centers(int curve, int key_bits, int real_bits, const void* prefixes, int num_nodes,
                               const double* lim, const int* bc, void* centers, void* sizes)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       withReal(real_bits,
                                [&](auto t)
                                {
                                    using T = decltype(t);
                                    nodeCenters<K, T>(Curve(curve), (const K*)prefixes, num_nodes, mkBox<T>(lim, bc),
                                                      (T*)centers, (T*)sizes);
                                });
                   });
}

/*! neighbor search for particles [first,last): coordinates and h share real_bits.
 *  neighbors[(last-first)*ngmax] row-major, counts[last-first] (true count, may exceed ngmax) */
int cstone_oracle_find_neighbors(int real_bits, const void* x, const void* y, const void* z, const void* h,
                                 unsigned first, unsigned last, const double* lim, const int* bc,
                                 const int* child_offsets, const int* internal_to_leaf, const unsigned* layout,
                                 const void* centers, const void* sizes, float ext, unsigned ngmax,
                                 unsigned* neighbors, unsigned* counts)
{
    return withReal(real_bits,
                    [&](auto t)
                    {
                        using T = decltype(t);
                        NsTree<T> tree{child_offsets, internal_to_leaf, layout, (const T*)centers, (const T*)sizes,
                                       ext};
                        findNeighbors<T, T>((const T*)x, (const T*)y, (const T*)z, (const T*)h, first, last,
                                            mkBox<T>(lim, bc), tree, ngmax, neighbors, counts);
                    });
}

/* ---- independent (brute-force) checkers for the owner-side halo discovery building blocks of the HIP library
 *      (cstone_hip_halo_boxes / cstone_hip_find_overlaps); they restate R/traversal/boxoverlap.hpp:42-182 only ---- */
int cstone_oracle_halo_boxes(int curve, int key_bits, int real_bits, const void* leaves, const float* radii,
                             const double* lim, const int* bc, int first, int last, int* boxes)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       withReal(real_bits,
                                [&](auto t)
                                {
                                    using T       = decltype(t);
                                    const K* lv   = (const K*)leaves;
                                    Box<T> box    = mkBox<T>(lim, bc);
                                    K lo = lv[first], hi = lv[last];
                                    for (int i = first; i < last; ++i)
                                    {
                                        unsigned level = levelOfSpan<K>(lv[i + 1] - lv[i]);
                                        IBox b = haloBox<K, T, float>(nodeIBox<K>(Curve(curve), lv[i], level), radii[i], box);
                                        int* rec = boxes + size_t(i - first) * 8;
                                        for (int d = 0; d < 3; ++d)
                                            rec[2 * d] = b.lo[d], rec[2 * d + 1] = b.hi[d];
                                        rec[6] = boxInsideKeyRange<K>(Curve(curve), lo, hi, b) ? 0 : 1;
                                        rec[7] = 0;
                                    }
                                });
                   });
}

int cstone_oracle_find_overlaps(int curve, int key_bits, const void* leaves, const int* boxes, int num_boxes, int first,
                                int last, int* flags)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K     = decltype(k);
                       const K* lv = (const K*)leaves;
#pragma omp parallel for schedule(dynamic, 16)
                       for (int l = first; l < last; ++l)
                       {
                           unsigned level = levelOfSpan<K>(lv[l + 1] - lv[l]);
                           IBox nb        = nodeIBox<K>(Curve(curve), lv[l], level);
                           for (int j = 0; j < num_boxes && !flags[l]; ++j)
                           {
                               const int* rec = boxes + size_t(j) * 8;
                               if (!rec[6]) continue;
                               IBox tb{{rec[0], rec[2], rec[4]}, {rec[1], rec[3], rec[5]}};
                               if (boxesOverlap<K>(nb, tb)) flags[l] = 1;
                           }
                       }
                   });
}

/* ---- target particle groups, R/traversal/groups_gpu.cu:41-151 ---- */
int cstone_oracle_fixed_groups(unsigned first, unsigned last, unsigned group_size, unsigned* groups, int capacity)
{
    auto g = fixedGroups(first, last, group_size);
    if (int(g.size()) > capacity) return -int(g.size());
    std::copy(g.begin(), g.end(), groups);
    return int(g.size()) - 1;
}

/*! returns the number of groups (groups holds one entry more) or -(entries needed) if capacity is too small */
int cstone_oracle_group_splits(int key_bits, int real_bits, unsigned first, unsigned last, const void* x, const void* y,
                               const void* z, const void* leaves, int num_leaves, const unsigned* layout,
                               const double* lim, const int* bc, unsigned group_size, float tol_factor,
                               unsigned* groups, int capacity)
{
    int ret = 0;
    int rc  = withKey(key_bits,
                     [&](auto k)
                     {
                         using K = decltype(k);
                         return withReal(real_bits,
                                         [&](auto t)
                                         {
                                             using T = decltype(t);
                                             auto g  = groupSplits<K, T>(first, last, (const T*)x, (const T*)y,
                                                                        (const T*)z, (const K*)leaves, num_leaves,
                                                                        layout, mkBox<T>(lim, bc), group_size,
                                                                        tol_factor);
                                             if (int(g.size()) > capacity) { ret = -int(g.size()); }
                                             else
                                             {
                                                 std::copy(g.begin(), g.end(), groups);
                                                 ret = int(g.size()) - 1;
                                             }
                                         });
                     });
    return rc ? -1 : ret;
}

/*! split bits of n = 64 * words positions (x,y,z interleaved), R/traversal/groups_gpu.cuh:57-93 */
void cstone_oracle_find_splits(const double* pos3, int words, double dist_crit_sq, uint64_t* splits)
{
    std::vector<std::array<double, 3>> pos(size_t(words) * 64);
    for (size_t i = 0; i < pos.size(); ++i)
        pos[i] = {pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]};
    auto s = findSplits(pos, dist_crit_sq, 64);
    std::copy(s.begin(), s.end(), splits);
}

/*! R/traversal/groups_gpu.cuh:107-130 for a stream of `words` masks of `width` (32 or 64) bits; returns the count */
int cstone_oracle_make_splits(const uint64_t* masks, int words, int width, unsigned* lengths)
{
    auto l = makeSplits(std::vector<uint64_t>(masks, masks + words), width);
    std::copy(l.begin(), l.end(), lengths);
    return int(l.size());
}

// ---- focus tree (locally essential tree)
int cstone_oracle_essential_ops(int key_bits, const void* prefixes, const int* child_offsets, const int* parents,
                                const unsigned* counts, const char* macs, uint64_t focus_start, uint64_t focus_end,
                                unsigned bucket, int* ops, int num_nodes)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       essentialOps<K>((const K*)prefixes, child_offsets, parents, counts, macs, K(focus_start),
                                       K(focus_end), bucket, ops, num_nodes);
                   });
}

int cstone_oracle_mac_refine_ops(int key_bits, const void* prefixes, const char* macs, const int* leaf_to_internal,
                                 int num_leaves, int focus_first, int focus_last, int* ops)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       macRefineOps<K>((const K*)prefixes, macs, leaf_to_internal, num_leaves, focus_first, focus_last, ops);
                   });
}

//! returns 1 if converged, 0 if not, < 0 on a bad argument
int cstone_oracle_protect_ancestors(int key_bits, const void* prefixes, const int* parents, int* ops, int num_nodes)
{
    int converged = 0;
    int rc        = withKey(key_bits,
                            [&](auto k)
                            {
                         using K   = decltype(k);
                         converged = protectAncestors<K>((const K*)prefixes, parents, ops, num_nodes);
                     });
    return rc ? rc : converged;
}

//! returns the ResolutionStatus (0..3), < 0 on a bad argument
int cstone_oracle_enforce_keys(int key_bits, const void* forced_keys, int num_keys, const void* prefixes,
                               const int* child_offsets, const int* parents, int* ops)
{
    int status = 0;
    int rc     = withKey(key_bits,
                         [&](auto k)
                         {
                         using K = decltype(k);
                         status  = enforceKeys<K>((const K*)forced_keys, num_keys, (const K*)prefixes, child_offsets,
                                                  parents, ops);
                     });
    return rc ? rc : status;
}

int cstone_oracle_range_count(int key_bits, const void* leaves, int num_leaves, const unsigned* counts,
                              const void* leaves_focus, int num_focus_leaves, const int* focus_idx, int num_idx,
                              unsigned* counts_focus)
{
    (void)num_focus_leaves;
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       rangeCount<K>((const K*)leaves, num_leaves, counts, (const K*)leaves_focus, focus_idx, num_idx,
                                     counts_focus);
                   });
}

//! mode 0: geoMacSpheres, mode 1: setMac; spheres = Vec4<T>[num_nodes]
int cstone_oracle_mac_spheres(int curve, int mode, int key_bits, int real_bits, const void* prefixes, int num_nodes,
                              void* spheres, float inv_theta, const double* lim, const int* bc)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       withReal(real_bits,
                                [&](auto t)
                                {
                                    using T = decltype(t);
                                    macSpheres<K, T>(Curve(curve), mode, (const K*)prefixes, num_nodes, (T*)spheres,
                                                     inv_theta, mkBox<T>(lim, bc));
                                });
                   });
}

int cstone_oracle_mark_macs(int curve, int key_bits, int real_bits, const void* prefixes, const int* child_offsets,
                            const void* centers, const double* lim, const int* bc, const void* focus_nodes,
                            int num_focus_nodes, int limit_source, char* markings)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       withReal(real_bits,
                                [&](auto t)
                                {
                                    using T = decltype(t);
                                    markMacs<K, T>(Curve(curve), (const K*)prefixes, child_offsets, (const T*)centers,
                                                   mkBox<T>(lim, bc), (const K*)focus_nodes, num_focus_nodes,
                                                   limit_source != 0, markings);
                                });
                   });
}

//! out == NULL: only count; returns the number of keys of the coarsest cornerstone tiling of [a, b)
int cstone_oracle_span_sfc_range(int key_bits, uint64_t a, uint64_t b, void* out)
{
    int num = 0;
    int rc  = withKey(key_bits,
                      [&](auto k)
                      {
                         using K = decltype(k);
                         num     = spanRange<K>(K(a), K(b), (K*)out);
                     });
    return rc ? rc : num;
}

//! bits = coordinate / mass / centre precision: 64/64/64, 64/32/64 or 32/32/32
int cstone_oracle_leaf_source_centers(int coord_bits, int mass_bits, int center_bits, const void* x, const void* y,
                                      const void* z, const void* m, const int* leaf_to_internal, int num_leaves,
                                      const unsigned* layout, void* centers)
{
    if (coord_bits == 64 && mass_bits == 64 && center_bits == 64)
        leafSourceCenters((const double*)x, (const double*)y, (const double*)z, (const double*)m, leaf_to_internal,
                          num_leaves, layout, (double*)centers);
    else if (coord_bits == 64 && mass_bits == 32 && center_bits == 64)
        leafSourceCenters((const double*)x, (const double*)y, (const double*)z, (const float*)m, leaf_to_internal,
                          num_leaves, layout, (double*)centers);
    else if (coord_bits == 32 && mass_bits == 32 && center_bits == 32)
        leafSourceCenters((const float*)x, (const float*)y, (const float*)z, (const float*)m, leaf_to_internal,
                          num_leaves, layout, (float*)centers);
    else
        return -1;
    return 0;
}

int cstone_oracle_upsweep_centers(int real_bits, int num_levels, const int* level_range, const int* child_offsets,
                                  void* centers)
{
    return withReal(real_bits,
                    [&](auto t)
                    {
                        using T = decltype(t);
                        upsweepCenters<T>(num_levels, level_range, child_offsets, (T*)centers);
                    });
}

int cstone_oracle_segment_max(int in_bits, int out_bits, const void* in, const unsigned* segments, size_t num_segments,
                              void* out)
{
    if (in_bits == 32 && out_bits == 32) segmentMax((const float*)in, segments, num_segments, (float*)out);
    else if (in_bits == 64 && out_bits == 32) segmentMax((const double*)in, segments, num_segments, (float*)out);
    else if (in_bits == 64 && out_bits == 64) segmentMax((const double*)in, segments, num_segments, (double*)out);
    else return -1;
    return 0;
}

/*! the uniform cloud of the reference's tests and benchmarks (test/coord_samples/random.hpp:93-113): std::mt19937(seed),
 *  x drawn completely, then y, then z from std::uniform_real_distribution<T>(lo, hi); restated here so that the GPU box
 *  can regenerate the 1e7-particle input of BASELINE configs[1] instead of shipping 240 MB */
int cstone_oracle_random_uniform(int real_bits, unsigned seed, size_t n, const double* lim, void* x, void* y, void* z)
{
    return withReal(real_bits,
                    [&](auto t)
                    {
                        using T = decltype(t);
                        std::mt19937 gen(seed);
                        T* out[3] = {(T*)x, (T*)y, (T*)z};
                        for (int d = 0; d < 3; ++d)
                        {
                            std::uniform_real_distribution<T> dis{T(lim[2 * d]), T(lim[2 * d + 1])};
                            for (size_t i = 0; i < n; ++i)
                                out[d][i] = dis(gen);
                        }
                    });
}

/*! the Plummer sphere of the reference's benchmarks (test/coord_samples/plummer.hpp:17-79), restated: drand48 behind
 *  srand48(42), R = 1 / sqrt(u^(-2/3) - 1) kept while R < 100, Z = (1 - 2u) R, theta = 2 pi u, everything scaled by
 *  3 pi / 16 and moved so that the centre of mass (equal masses 1/n, accumulated in T like the reference) is the origin.
 *  Serial by construction (one random stream); 1e8 particles take about 15 s. */
int cstone_oracle_plummer(int real_bits, size_t n, void* x, void* y, void* z)
{
    return withReal(real_bits,
                    [&](auto t)
                    {
                        using T = decltype(t);
                        srand48(42);
                        T* pos[3] = {(T*)x, (T*)y, (T*)z};
                        const T conv = T(3.0 * M_PI / 16.0);
                        size_t i = 0;
                        while (i < n)
                        {
                            T R = T(1.0 / std::sqrt(std::pow(drand48(), -2.0 / 3.0) - 1.0));
                            if (R < T(100.0))
                            {
                                T Z     = T((1.0 - 2.0 * drand48()) * R);
                                T theta = T(2 * M_PI * drand48());
                                // (the reference calls the C functions sqrt / cos / sin unqualified: double
                                //  arithmetic on the T-valued operands, rounded to T at the assignment)
                                T X     = T(std::sqrt(double(R * R - Z * Z)) * std::cos(double(theta)));
                                T Y     = T(std::sqrt(double(R * R - Z * Z)) * std::sin(double(theta)));
                                pos[0][i] = X * conv, pos[1][i] = Y * conv, pos[2][i] = Z * conv;
                                ++i;
                            }
                        }
                        T mcm = 0, mass = T(1) / T(n), xcm[3] = {0, 0, 0};
                        for (i = 0; i < n; ++i)
                        {
                            mcm += mass;
                            for (int k = 0; k < 3; ++k)
                                xcm[k] += mass * pos[k][i];
                        }
                        for (int k = 0; k < 3; ++k)
                            xcm[k] /= mcm;
                        for (i = 0; i < n; ++i)
                            for (int k = 0; k < 3; ++k)
                                pos[k][i] -= xcm[k];
                    });
}

//! child[2 * num_nodes], prefix[num_nodes] of the binary radix tree over tree[0 .. num_nodes]
int cstone_oracle_binary_tree(int key_bits, const void* tree, int num_nodes, int* child, void* prefix)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       binaryTree<K>((const K*)tree, num_nodes + 1, child, (K*)prefix);
                   });
}

int cstone_oracle_num_threads()
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

} // extern "C"
