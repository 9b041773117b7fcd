// Minimal client of the C++20 host layer: the reference README's usage pattern (R/README.md:64-103) on one MI355X.
//   g++ -std=c++20 -I include -I cornerstone-octree_amd/include examples/domain_example.cpp \
//       -L cornerstone-octree_amd/lib -lcstone_hip -Wl,-rpath,$PWD/cornerstone-octree_amd/lib -o domain_example
#include <cstdio>
#include <random>
#include <vector>

#include "cstone_amd/cstone_amd.hpp"

using namespace cstone_amd;

int main(int argc, char** argv)
{
    using KeyType = std::uint64_t;
    using T       = double;
    std::size_t n = argc > 1 ? std::stoul(argv[1]) : 1000000;

    std::mt19937 gen(42);
    std::uniform_real_distribution<T> dis(0, 1);
    std::vector<T> hx(n), hy(n), hz(n), hh(n, 0.01);
    for (auto& v : hx) v = dis(gen);
    for (auto& v : hy) v = dis(gen);
    for (auto& v : hz) v = dis(gen);
    std::vector<float> hmass(n, 1.0f);

    DeviceVector<T> x(hx.data(), hx.data() + n), y(hy.data(), hy.data() + n), z(hz.data(), hz.data() + n),
        h(hh.data(), hh.data() + n), scratch;
    // properties share the scratch buffer, which is sized for T: reserve the same byte capacity
    DeviceVector<float> mass;
    mass.reserve(n * sizeof(T) / sizeof(float));
    mass.resize(n);
    memcpyH2D(hmass.data(), n, mass.data());
    DeviceVector<KeyType> keys(n);
    Context::check(cstone_hip_memset(Context::get(), keys.data(), 0, n * sizeof(KeyType)), "memset");
    x.reserve(n), scratch.reserve(n);

    Domain<KeyType, T> domain(0, 1, /*bucketSize*/ 1024, /*bucketSizeFocus*/ 64, /*theta*/ 0.5f);
    for (int step = 0; step < 3; ++step)
    {
        domain.sync(keys, x, y, z, h, std::tie(mass), scratch);
        syncGpu();
        auto v = domain.view();
        std::printf("step %d: particles [%u, %u) of %u, %d global leaves, %d focus leaves\n", step, v.start_index,
                    v.end_index, v.num_particles_with_halos, v.num_global_leaves, v.num_focus_leaves);
    }
    // a field that was not part of the sync follows later (Domain::reapplySync): one more sync with "id" left out,
    // then id is brought into the new order and must name the particle that now sits in each slot
    std::vector<T> hid(x.size());
    for (std::size_t i = 0; i < hid.size(); ++i)
        hid[i] = T(i);
    DeviceVector<T> id(hid.data(), hid.data() + hid.size()), idScratch;
    auto xBefore = toHost(x);
    // (the reference's signature: the scratch buffers as a tuple)
    DeviceVector<KeyType> keyScratch;
    domain.sync(keys, x, y, z, h, std::tie(mass), std::tie(keyScratch, scratch));
    domain.reapplySync(std::tie(id), idScratch);
    {
        // Domain::globalTree() / focusTree(): cornerstone arrays from 0 to the end of the key space whose counts add up
        auto fromDevice = [](auto span)
        {
            std::vector<std::remove_const_t<typename decltype(span)::element_type>> v(span.size());
            memcpyD2H(span.data(), span.size(), v.data());
            return v;
        };
        auto ft = domain.focusTree();
        auto gt = fromDevice(domain.globalTree());
        auto fl = fromDevice(ft.treeLeaves());
        auto fc = fromDevice(ft.leafCounts());
        unsigned long long total = 0;
        for (auto c : fc)
            total += c;
        bool fine = gt.front() == 0 && fl.front() == 0 && gt.back() == fl.back() && std::is_sorted(fl.begin(), fl.end()) &&
                    total == domain.nParticles();
        std::printf("globalTree / focusTree: %zu and %zu leaves, counts add up: %s\n", gt.size() - 1, fl.size() - 1,
                    fine ? "yes" : "NO");
    }
    auto xAfter = toHost(x);
    auto idAfter = toHost(id);
    bool followed = idAfter.size() == xAfter.size();
    for (std::size_t i = 0; followed && i < idAfter.size(); ++i)
        followed = xBefore[std::size_t(idAfter[i])] == xAfter[i];
    std::printf("reapplySync: field followed its particles: %s\n", followed ? "yes" : "NO");

    // target groups for neighbor kernels (computeGroupSplits) on the domain's tree
    auto v = domain.view();
    DeviceVector<LocalIndex> splitScratch, groupOffsets;
    computeGroupSplits(domain.startIndex(), domain.endIndex(), x.data(), y.data(), z.data(), h.data(),
                       static_cast<const KeyType*>(v.focus_leaves), v.num_focus_leaves, v.layout, domain.box(), 64, 1.5f,
                       splitScratch, groupOffsets);
    GroupData fixed;
    computeFixedGroups(domain.startIndex(), domain.endIndex(), 64, fixed);
    auto go = toHost(groupOffsets);
    bool groupsOk = go.front() == domain.startIndex() && go.back() == domain.endIndex() &&
                    go.size() - 1 >= fixed.numGroups;
    for (std::size_t g = 1; g < go.size(); ++g)
        groupsOk = groupsOk && go[g] > go[g - 1] && go[g] - go[g - 1] <= 64;
    std::printf("target groups: %zu (fixed: %u): %s\n", go.size() - 1, fixed.numGroups, groupsOk ? "ok" : "BAD");

    // Domain::syncGrav on one rank = sync with the masses among the properties (above) + updateExpansionCenters
    domain.updateExpansionCenters(x, y, z, mass);
    std::vector<T> rootCentre(4, T(-1));
    if (domain.expansionCenters())
        Context::check(cstone_hip_memcpy_d2h(Context::get(), rootCentre.data(), domain.expansionCenters(), 4 * sizeof(T)),
                       "root centre");
    std::printf("expansion centre of the root: (%.4f, %.4f, %.4f), MAC radius^2 %.4f\n", double(rootCentre[0]),
                double(rootCentre[1]), double(rootCentre[2]), double(rootCentre[3]));
    followed = followed && rootCentre[0] > T(0.4) && rootCentre[0] < T(0.6) && rootCentre[3] > T(0);

    auto k = toHost(keys);
    bool sorted = true;
    for (std::size_t i = 1; i < k.size(); ++i)
        sorted = sorted && k[i - 1] <= k[i];
    std::printf("keys sorted: %s\n", sorted ? "yes" : "NO");

    // The multi-rank Domain with the collectives of a one-rank "communicator" (nothing to exchange).  A real client
    // fills cstone_hip_comm_ops with RCCL / MPI calls on the device buffers it is handed.
    cstone_hip_comm_ops self{};
    self.all_reduce   = [](void*, void*, std::size_t, int, int) { return 0; };
    self.all_gather   = [](void*, const void* s, void* r, std::size_t bytes)
    { return cstone_hip_memcpy_d2d(Context::get(), r, s, bytes); };
    self.all_to_all_v = [](void*, const void*, const std::size_t*, void*, const std::size_t*) { return 0; };
    MultiRankDomain<KeyType, T> mr(0, 1, 1024, 64, Box<T>{0, 1}, self);
    mr.sync(x.data(), y.data(), z.data(), h.data(), x.size(), mass.data());
    syncGpu();
    std::printf("multi-rank domain on one rank: particles [%u, %u) of %u, range [%llu, %llu)\n", mr.startIndex(),
                mr.endIndex(), mr.nParticlesWithHalos(), (unsigned long long)mr.assignedRange().first,
                (unsigned long long)mr.assignedRange().second);
    auto nsView = mr.octreeProperties();
    std::printf("its tree over local + halo particles: %d leaves\n", nsView.numLeafNodes);
    bool same = mr.nParticles() == x.size() && nsView.numLeafNodes > 0;
    // Domain::syncGrav: the masses follow their particles, the focus tree carries mass centres and MAC radii
    const auto first = mr.startIndex();
    mr.syncGrav(static_cast<const KeyType*>(nullptr), mr.x() + first, mr.y() + first, mr.z() + first, mr.h() + first,
                mr.property<float>(0) + first, mr.nParticles());
    syncGpu();
    const T* centers = mr.expansionCenters();
    std::vector<T> root(4, T(-1));
    if (centers) Context::check(cstone_hip_memcpy_d2h(Context::get(), root.data(), centers, 4 * sizeof(T)), "centre of the root");
    std::printf("syncGrav: centre of mass of the root (%.4f, %.4f, %.4f), %u particles, masses %s\n", double(root[0]),
                double(root[1]), double(root[2]), mr.nParticles(), mr.masses<float>() ? "attached" : "MISSING");
    same = same && centers && mr.masses<float>() && mr.nParticles() == x.size() && root[0] > T(0.4) && root[0] < T(0.6);
    return sorted && same && followed && groupsOk ? 0 : 1;
}
