// Instantiates EVERY function of the C++20 host layer (the reference's link-seam names on top of the C ABI) for both
// key widths and both real types, runs them as one pipeline on a small cloud and checks the results against plain host
// code: keys sorted and consistent, permutation, tree counts, linked octree, halo flags of a two-part split, neighbour
// counts against an O(n^2) loop, target groups.  Exit code 0 = every check passed.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#include "cstone_amd/cstone_amd.hpp"

using namespace cstone_amd;

template<class KeyType, class T>
bool runAll(const char* name, Curve curve)
{
    const std::size_t n = 3000;
    std::mt19937 gen(7);
    std::uniform_real_distribution<T> dis(0, 1);
    std::vector<T> hx(n), hy(n), hz(n), hh(n, T(0.04));
    for (auto& v : hx) v = dis(gen);
    for (auto& v : hy) v = dis(gen);
    for (auto& v : hz) v = dis(gen);
    Box<T> box(0, 1, 0, 1, 0, 1, BoundaryType::open, BoundaryType::periodic, BoundaryType::open);
    DeviceVector<T> x(hx.data(), hx.data() + n), y(hy.data(), hy.data() + n), z(hz.data(), hz.data() + n),
        h(hh.data(), hh.data() + n), xs(n), ys(n), zs(n), back(n);
    bool ok = true;
    auto expect = [&](bool cond, const char* what)
    {
        if (!cond) std::printf("  %s: FAILED %s\n", name, what);
        ok = ok && cond;
    };

    // box extents, keys, sort, gather / scatter
    auto [xmin, xmax] = MinMaxGpu<T>{}(x.data(), x.data() + n);
    expect(xmin == *std::min_element(hx.begin(), hx.end()) && xmax == *std::max_element(hx.begin(), hx.end()), "MinMaxGpu");
    DeviceVector<KeyType> keys(n);
    Context::check(cstone_hip_memset(Context::get(), keys.data(), 0, n * sizeof(KeyType)), "memset");
    computeSfcKeysGpu(x.data(), y.data(), z.data(), keys.data(), n, box, curve);
    auto unsortedKeys = toHost(keys);
    DeviceVector<LocalIndex> order(n);
    sequenceGpu(order.data(), n, 0);
    DeviceVector<KeyType> keyBuf(n);
    DeviceVector<LocalIndex> valBuf(n);
    DeviceVector<char> temp(sortByKeyTempStorage<KeyType, LocalIndex>(n));
    sortByKeyGpu(keys.data(), keys.data() + n, order.data(), keyBuf.data(), valBuf.data(), temp.data(), temp.size());
    auto hk = toHost(keys);
    auto ho = toHost(order);
    std::vector<LocalIndex> ref(n);
    std::iota(ref.begin(), ref.end(), 0);
    std::stable_sort(ref.begin(), ref.end(), [&](LocalIndex a, LocalIndex b) { return unsortedKeys[a] < unsortedKeys[b]; });
    expect(std::is_sorted(hk.begin(), hk.end()) && ho == ref, "sortByKeyGpu (stable permutation)");
    {
        DeviceVector<KeyType> k2(unsortedKeys.data(), unsortedKeys.data() + n);
        DeviceVector<LocalIndex> o2(n);
        sequenceGpu(o2.data(), n, 0);
        sortByKeyGpu(k2.data(), k2.data() + n, o2.data());
        expect(toHost(o2) == ref, "sortByKeyGpu (arena scratch)");
    }
    gatherGpu(order.data(), n, x.data(), xs.data());
    gatherGpu(order.data(), n, y.data(), ys.data());
    gatherGpu(order.data(), n, z.data(), zs.data());
    scatterGpu(order.data(), n, xs.data(), back.data());
    expect(toHost(back) == hx, "gatherGpu / scatterGpu round trip");

    // cornerstone tree: from the root with update steps, counts, node ops, rebalance
    const unsigned bucket = 16;
    std::vector<KeyType> rootTree{0, KeyType(1) << (3 * (sizeof(KeyType) == 8 ? 21 : 10))};
    DeviceVector<KeyType> tree(rootTree.data(), rootTree.data() + 2);
    std::vector<unsigned> rootCount{unsigned(n)};
    DeviceVector<unsigned> counts(rootCount.data(), rootCount.data() + 1);
    int steps = 0;
    while (!updateOctreeGpu(keys.data(), keys.data() + n, bucket, tree, counts) && steps < 30)
        ++steps;
    auto hc = toHost(counts);
    auto ht = toHost(tree);
    const TreeNodeIndex L = TreeNodeIndex(hc.size());
    expect(std::accumulate(hc.begin(), hc.end(), std::size_t(0)) == n && *std::max_element(hc.begin(), hc.end()) <= bucket,
           "updateOctreeGpu (converged counts)");
    DeviceVector<unsigned> counts2(L);
    computeNodeCountsGpu(tree.data(), counts2.data(), L, keys.data(), keys.data() + n, 0xFFFFFFFFu);
    expect(toHost(counts2) == hc, "computeNodeCountsGpu");
    DeviceVector<TreeNodeIndex> ops(L + 1);
    bool conv = false;
    TreeNodeIndex newL = computeNodeOpsGpu(tree.data(), L, counts.data(), bucket, ops.data(), &conv);
    expect(conv && newL == L, "computeNodeOpsGpu (converged tree)");
    DeviceVector<KeyType> tree2(newL + 1);
    rebalanceTreeGpu(tree.data(), L, newL, ops.data(), tree2.data());
    expect(toHost(tree2) == ht, "rebalanceTreeGpu (identity on a converged tree)");
    std::vector<KeyType> q{ht[L / 2], ht[L]};
    DeviceVector<KeyType> dq(q.data(), q.data() + 2);
    DeviceVector<std::uint64_t> pos(2);
    lowerBoundGpu(keys.data(), keys.data() + n, dq.data(), dq.data() + 2, pos.data());
    auto hp = toHost(pos);
    expect(hp[0] == std::size_t(std::lower_bound(hk.begin(), hk.end(), q[0]) - hk.begin()) && hp[1] == n, "lowerBoundGpu");

    // layout by scans, linked octree, upsweep, geometry
    DeviceVector<LocalIndex> layout(L + 1), layoutIncl(L);
    exclusiveScanGpu(counts.data(), counts.data() + L, layout.data());
    inclusiveScanGpu(counts.data(), counts.data() + L, layoutIncl.data());
    {
        auto le = toHost(layout);
        auto li = toHost(layoutIncl);
        bool good = li[L - 1] == n;
        for (TreeNodeIndex i = 0; i + 1 < L; ++i)
            good = good && le[i + 1] == li[i];
        expect(good, "exclusiveScanGpu / inclusiveScanGpu");
        memcpyD2D(layoutIncl.data() + (L - 1), 1, layout.data() + L);
    }
    const TreeNodeIndex I = (L - 1) / 7, M = L + I;
    DeviceVector<KeyType> prefixes(M);
    DeviceVector<TreeNodeIndex> child(M + 1), parents(std::max(1, (M - 1) / 8)), levelRange(24), itl(M), lti(M);
    buildOctreeGpu(tree.data(), OctreeView<KeyType>{L, I, M, prefixes.data(), child.data(), parents.data(),
                                                    levelRange.data(), itl.data(), lti.data()});
    {
        auto hlti = toHost(lti);
        auto hitl = toHost(itl);
        bool inv = true;
        // leafToInternal: node i of the cornerstone order (internal nodes first, then the leaves) -> linked layout;
        // internalToLeaf: back, shifted so that the leaves come out as 0 .. L-1 (R/tree/octree.hpp:160-165)
        for (TreeNodeIndex i = 0; i < M; ++i)
            inv = inv && hitl[hlti[i]] == i - I;
        expect(inv, "buildOctreeGpu (internalToLeaf inverts leafToInternal)");
        // node counts in the linked layout: leaves scattered by leafToInternal, internal nodes by the upsweep
        std::vector<LocalIndex> nodeCounts(M, 0);
        for (TreeNodeIndex i = 0; i < L; ++i)
            nodeCounts[hlti[I + i]] = hc[i];
        DeviceVector<LocalIndex> dc(nodeCounts.data(), nodeCounts.data() + M);
        upsweepSumGpu(sizeof(KeyType) == 8 ? 21 : 10, levelRange.data(), child.data(), dc.data());
        expect(toHost(dc)[0] == n, "upsweepSumGpu (root count)");
    }
    DeviceVector<T> centers(3 * M), sizes(3 * M);
    computeGeoCentersGpu(prefixes.data(), M, centers.data(), sizes.data(), box, curve);
    expect(toHost(sizes)[0] == T(0.5) && toHost(centers)[0] == T(0.5), "computeGeoCentersGpu (root cell)");

    // halos of the lower half of the leaves among the upper half
    DeviceVector<T> hs(n);
    gatherGpu(order.data(), n, h.data(), hs.data());
    DeviceVector<float> radii(L);
    haloRadiiGpu(hs.data(), layout.data(), 0, L / 2, L, 1.0f, radii.data());
    DeviceVector<int> flags(L);
    Context::check(cstone_hip_memset(Context::get(), flags.data(), 0, L * sizeof(int)), "memset");
    findHalosGpu(prefixes.data(), child.data(), itl.data(), tree.data(), radii.data(), box, 0, L / 2, flags.data(), curve);
    {
        auto hf = toHost(flags);
        int inside = std::accumulate(hf.begin(), hf.begin() + L / 2, 0), outside = std::accumulate(hf.begin() + L / 2, hf.end(), 0);
        expect(inside == 0 && outside > 0, "haloRadiiGpu / findHalosGpu");
    }

    // neighbours against the O(n^2) loop (minimum image in y), fixed targets and target groups
    OctreeNsView<T, KeyType> view{L, prefixes.data(), child.data(), itl.data(), levelRange.data(), tree.data(),
                                  layout.data(), centers.data(), sizes.data()};
    const unsigned ngmax = 64;
    DeviceVector<LocalIndex> nidx(n * ngmax);
    DeviceVector<unsigned> nc(n);
    findNeighborsGpu(xs.data(), ys.data(), zs.data(), hs.data(), 0, LocalIndex(n), box, view, ngmax, nidx.data(), nc.data());
    auto sx = toHost(xs);
    auto sy = toHost(ys);
    auto sz = toHost(zs);
    std::vector<unsigned> brute(n, 0);
    for (std::size_t i = 0; i < n; ++i)
        for (std::size_t j = 0; j < n; ++j)
        {
            T dx = sx[j] - sx[i], dy = sy[j] - sy[i], dz = sz[j] - sz[i];
            dy -= std::rint(dy); // periodic axis of length 1
            if (j != i && dx * dx + dy * dy + dz * dz < T(4) * T(0.04) * T(0.04)) ++brute[i];
        }
    expect(toHost(nc) == brute, "findNeighborsGpu (counts)");
    DeviceVector<LocalIndex> splitScratch, groupOffsets;
    computeGroupSplits(0, LocalIndex(n), xs.data(), ys.data(), zs.data(), hs.data(), tree.data(), L, layout.data(), box, 64,
                       1.5f, splitScratch, groupOffsets);
    GroupData fixed;
    computeFixedGroups(0, LocalIndex(n), 64, fixed);
    GroupView gv{0, LocalIndex(n), LocalIndex(groupOffsets.size() - 1), groupOffsets.data(), groupOffsets.data() + 1};
    DeviceVector<unsigned> nc2(n);
    findNeighborsGpu(xs.data(), ys.data(), zs.data(), hs.data(), gv, box, view, ngmax, nidx.data(), nc2.data());
    expect(toHost(nc2) == brute && gv.numGroups >= fixed.view().numGroups, "computeGroupSplits / findNeighborsGpu over groups");
    syncGpu();
    std::printf("%s: %s\n", name, ok ? "ok" : "FAILED");
    return ok;
}

//! computeContinuumCsarray for the two concentration functions the oracle's reference build can name (oracle/ref_driver.cpp,
//! cstone_ref_continuum): leaves, particles and an FNV-1a digest of leaf keys and counts, for tests/test_cpp_layer.py
template<class KeyType>
void continuumDigest(const char* name, int kind)
{
    const double n0 = 1e6, lo = -1, hi = 1;
    Box<double> box(lo, hi);
    const double vol = box.lx() * box.ly() * box.lz();
    const double eps = box.lx() / double(1u << (sizeof(KeyType) == 8 ? 21 : 10));
    auto constant    = [=](double, double, double) { return n0 / vol; };
    auto oneOverR    = [=](double x, double y, double z)
    {
        double r = std::max(std::sqrt(x * x + y * y + z * z), eps);
        return r > 1.0 ? 0.0 : n0 / (2 * M_PI * r);
    };
    std::vector<KeyType> tree;
    std::vector<unsigned> counts;
    if (kind == 0) std::tie(tree, counts) = computeContinuumCsarray<KeyType>(constant, box, 64u);
    else std::tie(tree, counts) = computeContinuumCsarray<KeyType>(oneOverR, box, 64u);
    std::uint64_t h = 1469598103934665603ull, sum = 0;
    auto mix = [&](std::uint64_t v)
    {
        for (int b = 0; b < 8; ++b)
            h = (h ^ ((v >> (8 * b)) & 0xff)) * 1099511628211ull;
    };
    for (auto k : tree)
        mix(std::uint64_t(k));
    for (auto c : counts)
        mix(c), sum += c;
    std::printf("continuum %s kind %d: leaves %zu particles %llu digest %016llx\n", name, kind, counts.size(),
                (unsigned long long)sum, (unsigned long long)h);
}

int main()
{
    continuumDigest<std::uint64_t>("u64", 0);
    continuumDigest<std::uint64_t>("u64", 1);
    continuumDigest<std::uint32_t>("u32", 0);
    continuumDigest<std::uint32_t>("u32", 1);
    bool ok = runAll<std::uint64_t, double>("u64/f64 hilbert", Curve::hilbert);
    ok      = runAll<std::uint64_t, float>("u64/f32 morton", Curve::morton) && ok;
    ok      = runAll<std::uint32_t, double>("u32/f64 morton", Curve::morton) && ok;
    ok      = runAll<std::uint32_t, float>("u32/f32 hilbert", Curve::hilbert) && ok;
    std::printf("seam check: %s\n", ok ? "all passed" : "FAILED");
    return ok ? 0 : 1;
}
