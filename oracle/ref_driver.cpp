// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
//
// Thin C entry points onto the REFERENCE's own CPU implementation, compiled from the headers where
// they lie under /root/reference/include (never copied into this repository).  Built by
// oracle/Makefile into oracle/_ref/libcstone_ref.so (git-ignored).  Used to pin the restatement in
// cstone_oracle.hpp and, optionally, as bench.py's cpu_baseline (kind = "reference").
//
// Signatures mirror oracle_capi.cpp (prefix cstone_ref_).  The reference hard-wires
// SfcKind = HilbertKey (sfc/sfc.hpp:53-55), so everything that goes through sfcIBox/sfcKey
// (halos, node centers) exists for curve = 1 (Hilbert) only and returns -3 for Morton.
#include <cstring>
#include <vector>

#include "cstone/findneighbors.hpp"
#include "cstone/focus/rebalance.hpp"
#include "cstone/focus/source_center.hpp"
#include "cstone/primitives/gather.hpp"
#include "cstone/sfc/sfc.hpp"
#include "cstone/traversal/collisions.hpp"
#include "cstone/traversal/macs.hpp"
#include "cstone/tree/btree.hpp"
#include "cstone/tree/continuum.hpp"
#include "cstone/tree/csarray.hpp"
#include "cstone/tree/octree.hpp"
// the reference's own Plummer generator, included where it lies (test/coord_samples/plummer.hpp)
#include "../test/coord_samples/plummer.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace cstone;

namespace
{
template<class T>
Box<T> mkBox(const double* lim, const int* bc)
{
    return Box<T>(T(lim[0]), T(lim[1]), T(lim[2]), T(lim[3]), T(lim[4]), T(lim[5]), BoundaryType(bc[0]),
                  BoundaryType(bc[1]), BoundaryType(bc[2]));
}

template<class F>
int withKey(int keyBits, F&& f)
{
    if (keyBits == 32) { f(unsigned{}); }
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

int cstone_ref_compute_sfc_keys(int curve, int key_bits, int real_bits, const void* x, const void* y, const void* z,
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
                                    auto box = mkBox<T>(lim, bc);
                                    if (curve == 0)
                                        computeSfcKeys((const T*)x, (const T*)y, (const T*)z, (MortonKey<K>*)keys, n,
                                                       box);
                                    else
                                        computeSfcKeys((const T*)x, (const T*)y, (const T*)z, (HilbertKey<K>*)keys, n,
                                                       box);
                                });
                   });
}

uint64_t cstone_ref_encode(int curve, int key_bits, unsigned ix, unsigned iy, unsigned iz)
{
    if (key_bits == 32) return curve == 0 ? iMorton<unsigned>(ix, iy, iz) : iHilbert<unsigned>(ix, iy, iz);
    return curve == 0 ? iMorton<uint64_t>(ix, iy, iz) : iHilbert<uint64_t>(ix, iy, iz);
}

void cstone_ref_decode(int curve, int key_bits, uint64_t key, unsigned* out3)
{
    util::tuple<unsigned, unsigned, unsigned> t;
    if (key_bits == 32) t = curve == 0 ? decodeMorton<unsigned>(unsigned(key)) : decodeHilbert<unsigned>(unsigned(key));
    else t = curve == 0 ? decodeMorton<uint64_t>(key) : decodeHilbert<uint64_t>(key);
    out3[0] = util::get<0>(t), out3[1] = util::get<1>(t), out3[2] = util::get<2>(t);
}

void cstone_ref_node_ibox(int curve, int key_bits, uint64_t key, unsigned level, int* out6)
{
    IBox b;
    if (key_bits == 32) b = curve == 0 ? mortonIBox<unsigned>(unsigned(key), level) : hilbertIBox<unsigned>(unsigned(key), level);
    else b = curve == 0 ? mortonIBox<uint64_t>(key, level) : hilbertIBox<uint64_t>(key, level);
    out6[0] = b.xmin(), out6[1] = b.xmax(), out6[2] = b.ymin(), out6[3] = b.ymax(), out6[4] = b.zmin(), out6[5] = b.zmax();
}

int cstone_ref_sort_pairs(int key_bits, void* keys, unsigned* vals, size_t n)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       sort_by_key((K*)keys, (K*)keys + n, vals);
                   });
}

int cstone_ref_node_counts(int key_bits, const void* tree, unsigned* counts, int num_nodes, const void* keys, size_t n,
                           unsigned max_count)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       computeNodeCounts((const K*)tree, counts, num_nodes, (const K*)keys, (const K*)keys + n,
                                         max_count, false);
                   });
}

int cstone_ref_node_ops(int key_bits, const void* tree, int num_nodes, const unsigned* counts, unsigned bucket,
                        int* node_ops, int* converged)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K    = decltype(k);
                       *converged = rebalanceDecision((const K*)tree, counts, num_nodes, bucket, node_ops);
                   });
}

int cstone_ref_update_octree(int key_bits, const void* keys, size_t n, unsigned bucket, void* tree_io,
                             unsigned* counts_io, int* num_leaves, int cap_leaves, unsigned max_count, int* converged)
{
    int rc = 0;
    int st = withKey(key_bits,
                     [&](auto k)
                     {
                         using K = decltype(k);
                         std::vector<K> tree((K*)tree_io, (K*)tree_io + *num_leaves + 1);
                         std::vector<unsigned> counts(counts_io, counts_io + *num_leaves);
                         *converged =
                             updateOctree((const K*)keys, (const K*)keys + n, bucket, tree, counts, max_count);
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

int cstone_ref_compute_octree(int key_bits, const void* keys, size_t n, unsigned bucket, void* tree_out,
                              unsigned* counts_out, int* num_leaves, int cap_leaves, unsigned max_count,
                              int* iterations)
{
    int rc = 0;
    int st = withKey(key_bits,
                     [&](auto k)
                     {
                         using K = decltype(k);
                         auto [tree, counts] = computeOctree((const K*)keys, (const K*)keys + n, bucket, max_count);
                         *iterations = -1;
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

int cstone_ref_spanning_tree(int key_bits, const void* span_keys, int num_keys, void* tree_out, int cap,
                             int* num_leaves)
{
    int rc = 0;
    int st = withKey(key_bits,
                     [&](auto k)
                     {
                         using K = decltype(k);
                         auto t  = computeSpanningTree<K>({(const K*)span_keys, size_t(num_keys)});
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

int cstone_ref_build_octree(int key_bits, const void* leaves, int num_leaves, void* prefixes, int* child_offsets,
                            int* parents, int* level_range, int* internal_to_leaf, int* leaf_to_internal)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       int numInternal = (num_leaves - 1) / 7;
                       int numNodes    = num_leaves + numInternal;
                       // the reference does not write childOffsets[numNodes]; keep the caller's buffer defined
                       child_offsets[numNodes] = 0;
                       buildOctreeCpu((const K*)leaves, num_leaves, numInternal, (K*)prefixes, child_offsets, parents,
                                      level_range, internal_to_leaf, leaf_to_internal);
                   });
}

int cstone_ref_upsweep_counts(const int* level_range, int num_level_entries, const int* child_offsets,
                              unsigned* counts)
{
    upsweep({level_range, size_t(num_level_entries)}, {child_offsets, size_t(level_range[num_level_entries - 1])},
            counts, NodeCount<unsigned>{});
    return 0;
}

int cstone_ref_find_halos(int curve, int key_bits, int real_bits, const void* prefixes, const int* child_offsets,
                          const int* internal_to_leaf, const void* leaves, const float* radii, const double* lim,
                          const int* bc, int first, int last, int* flags)
{
    if (curve != 1) return -3;
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       withReal(real_bits,
                                [&](auto t)
                                {
                                    using T = decltype(t);
                                    findHalos((const K*)prefixes, child_offsets, internal_to_leaf, (const K*)leaves,
                                              radii, mkBox<T>(lim, bc), first, last, flags);
                                });
                   });
}

int cstone_ref_node_centers(int curve, int key_bits, int real_bits, const void* prefixes, int num_nodes,
                            const double* lim, const int* bc, void* centers, void* sizes)
{
    if (curve != 1) return -3;
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       withReal(real_bits,
                                [&](auto t)
                                {
                                    using T = decltype(t);
                                    static_assert(sizeof(Vec3<T>) == 3 * sizeof(T));
                                    nodeFpCenters<K>({(const K*)prefixes, size_t(num_nodes)}, (Vec3<T>*)centers,
                                                     (Vec3<T>*)sizes, mkBox<T>(lim, bc));
                                });
                   });
}

int cstone_ref_find_neighbors(int real_bits, const void* x, const void* y, const void* z, const void* h,
                              unsigned first, unsigned last, const double* lim, const int* bc,
                              const int* child_offsets, const int* internal_to_leaf, const unsigned* layout,
                              const void* centers, const void* sizes, float ext, unsigned ngmax, unsigned* neighbors,
                              unsigned* counts)
{
    return withReal(real_bits,
                    [&](auto t)
                    {
                        using T = decltype(t);
                        OctreeNsView<T, uint64_t> view{0,
                                                       nullptr,
                                                       child_offsets,
                                                       internal_to_leaf,
                                                       nullptr,
                                                       nullptr,
                                                       layout,
                                                       (const Vec3<T>*)centers,
                                                       (const Vec3<T>*)sizes,
                                                       ext};
                        findNeighbors((const T*)x, (const T*)y, (const T*)z, (const T*)h, first, last,
                                      mkBox<T>(lim, bc), view, ngmax, neighbors, counts);
                    });
}

// ---- focus tree: the reference's own CPU functions (R/focus/rebalance.hpp, R/traversal/macs.hpp, R/focus/source_center.hpp)
int cstone_ref_essential_ops(int key_bits, const void* prefixes, const int* child_offsets, const int* parents,
                             const unsigned* counts, const char* macs, uint64_t focus_start, uint64_t focus_end,
                             unsigned bucket, int* ops, int num_nodes)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       rebalanceDecisionEssential<K>({(const K*)prefixes, size_t(num_nodes)}, child_offsets, parents,
                                                     counts, macs, K(focus_start), K(focus_end), bucket, ops);
                   });
}

int cstone_ref_mac_refine_ops(int key_bits, const void* prefixes, const char* macs, const int* leaf_to_internal,
                              int num_leaves, int focus_first, int focus_last, int* ops)
{
    // the reference has this loop only as a GPU kernel (R/focus/rebalance_gpu.cu:88-101); its per-leaf rule is macRefineOp
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       for (int i = 0; i < num_leaves; ++i)
                       {
                           int n  = leaf_to_internal[i];
                           ops[i] = (i < focus_first || i >= focus_last) ? macRefineOp(((const K*)prefixes)[n], macs[n]) : 1;
                       }
                   });
}

int cstone_ref_protect_ancestors(int key_bits, const void* prefixes, const int* parents, int* ops, int num_nodes)
{
    int converged = 0;
    int rc        = withKey(key_bits,
                            [&](auto k)
                            {
                         using K   = decltype(k);
                         converged = protectAncestors<K>({(const K*)prefixes, size_t(num_nodes)}, parents, ops);
                     });
    return rc ? rc : converged;
}

int cstone_ref_enforce_keys(int key_bits, const void* forced_keys, int num_keys, const void* prefixes,
                            const int* child_offsets, const int* parents, int* ops)
{
    int status = 0;
    int rc     = withKey(key_bits,
                         [&](auto k)
                         {
                         using K = decltype(k);
                         status  = int(enforceKeys<K>({(const K*)forced_keys, size_t(num_keys)}, (const K*)prefixes,
                                                      child_offsets, parents, ops));
                     });
    return rc ? rc : status;
}

int cstone_ref_range_count(int key_bits, const void* leaves, int num_leaves, const unsigned* counts,
                           const void* leaves_focus, int num_focus_leaves, const int* focus_idx, int num_idx,
                           unsigned* counts_focus)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       rangeCount<K>({(const K*)leaves, size_t(num_leaves) + 1}, {counts, size_t(num_leaves)},
                                     {(const K*)leaves_focus, size_t(num_focus_leaves) + 1},
                                     {focus_idx, size_t(num_idx)}, {counts_focus, size_t(num_focus_leaves)});
                   });
}

int cstone_ref_mac_spheres(int curve, int mode, int key_bits, int real_bits, const void* prefixes, int num_nodes,
                           void* spheres, float inv_theta, const double* lim, const int* bc)
{
    if (curve != 1) return -3;
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       withReal(real_bits,
                                [&](auto t)
                                {
                                    using T  = decltype(t);
                                    auto box = mkBox<T>(lim, bc);
                                    gsl::span<const K> keys{(const K*)prefixes, size_t(num_nodes)};
                                    auto* c = (SourceCenterType<T>*)spheres;
                                    if (mode == 0) geoMacSpheres<K, T>(keys, c, inv_theta, box);
                                    else setMac<T, K>(keys, {c, size_t(num_nodes)}, inv_theta, box);
                                });
                   });
}

int cstone_ref_mark_macs(int curve, int key_bits, int real_bits, const void* prefixes, const int* child_offsets,
                         const void* centers, const double* lim, const int* bc, const void* focus_nodes,
                         int num_focus_nodes, int limit_source, char* markings)
{
    if (curve != 1) return -3;
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       withReal(real_bits,
                                [&](auto t)
                                {
                                    using T = decltype(t);
                                    markMacs<T, K>((const K*)prefixes, child_offsets, (const Vec4<T>*)centers,
                                                   mkBox<T>(lim, bc), (const K*)focus_nodes, num_focus_nodes,
                                                   limit_source != 0, markings);
                                });
                   });
}

int cstone_ref_span_sfc_range(int key_bits, uint64_t a, uint64_t b, void* out)
{
    int num = 0;
    int rc  = withKey(key_bits,
                      [&](auto k)
                      {
                         using K = decltype(k);
                         num     = out ? spanSfcRange<K>(K(a), K(b), (K*)out) : spanSfcRange<K>(K(a), K(b));
                     });
    return rc ? rc : num;
}

int cstone_ref_leaf_source_centers(int coord_bits, int mass_bits, int center_bits, const void* x, const void* y,
                                   const void* z, const void* m, const int* leaf_to_internal, int num_leaves,
                                   const unsigned* layout, void* centers)
{
    auto run = [&](auto tc, auto tm, auto tf)
    {
        using Tc = decltype(tc);
        using Tm = decltype(tm);
        using Tf = decltype(tf);
        size_t n = layout[num_leaves];
        computeLeafMassCenter<Tc, Tm, Tf>({(const Tc*)x, n}, {(const Tc*)y, n}, {(const Tc*)z, n}, {(const Tm*)m, n},
                                          {leaf_to_internal, size_t(num_leaves)}, layout, (SourceCenterType<Tf>*)centers);
    };
    if (coord_bits == 64 && mass_bits == 64 && center_bits == 64) run(double{}, double{}, double{});
    else if (coord_bits == 64 && mass_bits == 32 && center_bits == 64) run(double{}, float{}, double{});
    else if (coord_bits == 32 && mass_bits == 32 && center_bits == 32) run(float{}, float{}, float{});
    else return -1;
    return 0;
}

int cstone_ref_upsweep_centers(int real_bits, int num_levels, const int* level_range, const int* child_offsets,
                               void* centers)
{
    return withReal(real_bits,
                    [&](auto t)
                    {
                        using T = decltype(t);
                        // the generic bottom-up pass of R/tree/octree.hpp:584-628 with the combination rule of
                        // R/focus/source_center.hpp:79-95, as the CPU branch of FocusedOctree::updateCenters uses it
                        upsweep({level_range, size_t(num_levels) + 1}, {child_offsets, size_t(level_range[num_levels])},
                                (SourceCenterType<T>*)centers, CombineSourceCenter<T>{});
                    });
}

int cstone_ref_binary_tree(int key_bits, const void* tree, int num_nodes, int* child, void* prefix)
{
    return withKey(key_bits,
                   [&](auto k)
                   {
                       using K = decltype(k);
                       std::vector<BinaryNode<K>> nodes(num_nodes);
                       createBinaryTree((const K*)tree, num_nodes, nodes.data());
                       for (int i = 0; i < num_nodes; ++i)
                       {
                           child[2 * i]     = nodes[i].child[0];
                           child[2 * i + 1] = nodes[i].child[1];
                           ((K*)prefix)[i]  = nodes[i].prefix;
                       }
                   });
}

/*! computeContinuumCsarray (R/tree/continuum.hpp:103-116) for two concentration functions a C interface can name:
 *  kind 0: constant n0 / box volume; kind 1: n0 / (2 pi r) inside the unit sphere, 0 outside, r floored at one grid cell
 *  (the function of T/unit/tree/continuum.cpp:54-75).  Box = [lo, hi]^3, double coordinates, Hilbert keys.
 *  Returns the number of leaves (tree: that many + 1 keys) or -1 if cap is too small. */
int cstone_ref_continuum(int key_bits, int kind, double n0, unsigned bucket, double lo, double hi, void* tree,
                         unsigned* counts, int cap)
{
    int leaves = -1;
    withKey(key_bits,
            [&](auto k)
            {
                using K = decltype(k);
                Box<double> box(lo, hi);
                const double vol = box.lx() * box.ly() * box.lz();
                const double eps = box.lx() / double(1u << maxTreeLevel<K>{});
                auto constant    = [=](double, double, double) { return n0 / vol; };
                auto oneOverR    = [=](double x, double y, double z)
                {
                    double r = std::max(std::sqrt(x * x + y * y + z * z), eps);
                    return r > 1.0 ? 0.0 : n0 / (2 * M_PI * r);
                };
                std::vector<K> t;
                std::vector<unsigned> c;
                if (kind == 0) std::tie(t, c) = computeContinuumCsarray<K>(constant, box, bucket);
                else std::tie(t, c) = computeContinuumCsarray<K>(oneOverR, box, bucket);
                if (int(c.size()) > cap) return;
                std::copy(t.begin(), t.end(), (K*)tree);
                std::copy(c.begin(), c.end(), counts);
                leaves = int(c.size());
            });
    return leaves;
}

//! plummer<T>(n) of test/coord_samples/plummer.hpp (pins cstone_oracle_plummer)
int cstone_ref_plummer(int real_bits, size_t n, void* x, void* y, void* z)
{
    if (real_bits == 64)
    {
        auto pos = plummer<double>(n);
        std::copy(pos[0].begin(), pos[0].end(), (double*)x);
        std::copy(pos[1].begin(), pos[1].end(), (double*)y);
        std::copy(pos[2].begin(), pos[2].end(), (double*)z);
    }
    else if (real_bits == 32)
    {
        auto pos = plummer<float>(n);
        std::copy(pos[0].begin(), pos[0].end(), (float*)x);
        std::copy(pos[1].begin(), pos[1].end(), (float*)y);
        std::copy(pos[2].begin(), pos[2].end(), (float*)z);
    }
    else return -1;
    return 0;
}

int cstone_ref_num_threads()
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

} // extern "C"
