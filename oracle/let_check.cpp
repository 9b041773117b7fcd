// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
// The product's host state machine of the locally essential tree (cornerstone-octree_amd/csrc/let.hpp, plain C++ over
// the C ABI) against the REFERENCE's own classes, rank by rank under mpiexec: the reference's
// cstone::Domain<KeyType, T, CpuTag> runs a sync (its FocusedOctree, Halos and GlobalAssignment members do the work,
// compiled from the headers where they lie under /root/reference/include), then FocusLet::update gets the same inputs
// -- box, assigned keys, smoothing lengths, global tree and assignment -- and must arrive at the same peers, focus
// leaves, leaf and node counts, focus assignment, halo flags, layout, start / end index, buffer size, node centres
// and, through exchangeHalos, the same halo particles.  The harness only talks to the C ABI (cstone_hip_malloc /
// memcpy for every array it hands over or looks at), so the same source gives two programs (oracle/Makefile):
//   oracle/_ref/let_check      the ABI served by oracle/cabi_on_oracle.cpp (CPU restatement on host memory): runs in
//                              this container, which has no GPU (tests/test_let.py, -m "not gpu")
//   oracle/_ref/let_check_hip  linked against libcstone_hip.so: the HIP kernels behind the same state machine on the
//                              MI355X box, the ranks of mpiexec sharing the GPU (tests/test_let.py, -m gpu)
//
// usage: let_check <k64f64|k32f32|k64f32> <numParticles> <syncs> <bucket> <bucketFocus> <bcx> <bcy> <bcz> <kind> <seed> [grav]
//        kind: 0 uniform, 1 blobs (imbalanced), 2 drifting blob (the assignment moves every sync)
#include <mpi.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <numeric>
#include <random>
#include <string>
#include <tuple>
#include <vector>

// the members the comparison needs (GlobalAssignment, Halos, counts of the focus tree) are private
#define private public
#define protected public
#include "cstone/domain/domain.hpp"
#undef private
#undef protected

#include "cstone_hip.h"
#include "../cornerstone-octree_amd/csrc/let.hpp"


using namespace cstone;

namespace
{

//! cstone_hip_comm_ops over MPI: the buffers are DEVICE buffers of the ABI, staged through the host
struct MpiComm
{
    int P;
    cstone_hip_ctx* ctx;
};

int mpiAllReduce(void* user, void* buf, size_t count, int dtype, int op)
{
    auto* c        = static_cast<MpiComm*>(user);
    const size_t e = dtype == 0 ? 8 : 4;
    std::vector<char> h(count * e);
    if (cstone_hip_memcpy_d2h(c->ctx, h.data(), buf, h.size())) return 1;
    MPI_Datatype t = dtype == 0 ? MPI_DOUBLE : MPI_UINT32_T;
    MPI_Op o       = op == 0 ? MPI_SUM : MPI_MIN;
    if (MPI_Allreduce(MPI_IN_PLACE, h.data(), int(count), t, o, MPI_COMM_WORLD) != MPI_SUCCESS) return 1;
    return cstone_hip_memcpy_h2d(c->ctx, buf, h.data(), h.size());
}
int mpiAllGather(void* user, const void* send, void* recv, size_t bytes)
{
    auto* c = static_cast<MpiComm*>(user);
    std::vector<char> s(bytes), r(bytes * c->P);
    if (cstone_hip_memcpy_d2h(c->ctx, s.data(), send, bytes)) return 1;
    if (MPI_Allgather(s.data(), int(bytes), MPI_BYTE, r.data(), int(bytes), MPI_BYTE, MPI_COMM_WORLD) != MPI_SUCCESS)
        return 1;
    return cstone_hip_memcpy_h2d(c->ctx, recv, r.data(), r.size());
}
int mpiAllToAllV(void* user, const void* send, const size_t* sb, void* recv, const size_t* rb)
{
    auto* c = static_cast<MpiComm*>(user);
    int P   = c->P;
    std::vector<int> sc(P), sd(P), rc(P), rd(P);
    int so = 0, ro = 0;
    for (int p = 0; p < P; ++p)
    {
        sc[p] = int(sb[p]), sd[p] = so, so += sc[p];
        rc[p] = int(rb[p]), rd[p] = ro, ro += rc[p];
    }
    std::vector<char> s(std::max(so, 1)), r(std::max(ro, 1));
    if (so && cstone_hip_memcpy_d2h(c->ctx, s.data(), send, size_t(so))) return 1;
    if (MPI_Alltoallv(s.data(), sc.data(), sd.data(), MPI_BYTE, r.data(), rc.data(), rd.data(), MPI_BYTE,
                      MPI_COMM_WORLD) != MPI_SUCCESS)
        return 1;
    if (ro && cstone_hip_memcpy_h2d(c->ctx, recv, r.data(), size_t(ro))) return 1;
    return 0;
}

//! a host array on the device of the ABI
template<class V>
struct OnDevice
{
    cstone_hip_ctx* ctx;
    void* p = nullptr;
    OnDevice(cstone_hip_ctx* c, const V* host, size_t n)
        : ctx(c)
    {
        if (cstone_hip_malloc(ctx, &p, std::max<size_t>(n, 1) * sizeof(V)) ||
            cstone_hip_memcpy_h2d(ctx, p, host, n * sizeof(V)))
            std::abort();
    }
    ~OnDevice() { cstone_hip_free(ctx, p); }
    const V* get() const { return static_cast<const V*>(p); }
    V* get() { return static_cast<V*>(p); }
};

//! a device array of the ABI on the host
template<class V>
std::vector<V> fetch(cstone_hip_ctx* ctx, const V* dev, size_t n)
{
    std::vector<V> out(n);
    if (n && cstone_hip_memcpy_d2h(ctx, out.data(), dev, n * sizeof(V))) std::abort();
    return out;
}

int failures = 0;
int rankG    = 0;

template<class A, class B>
void expectEqual(const char* what, int sync, const A* a, size_t na, const B* b, size_t nb)
{
    bool ok = na == nb;
    size_t at = 0;
    for (; ok && at < na; ++at)
        if (!(a[at] == b[at]))
        {
            ok = false;
            break;
        }
    if (!ok)
    {
        ++failures;
        std::fprintf(stderr, "[rank %d sync %d] %s DIFFERS (sizes %zu / %zu, first difference at %zu)\n", rankG, sync,
                     what, na, nb, at);
    }
}

int gArgc = 0;

template<class K, class T>
int run(int rank, int P, char** argv)
{
    const size_t N     = std::strtoull(argv[2], nullptr, 10);
    const int syncs    = std::atoi(argv[3]);
    const unsigned bucket = unsigned(std::atoi(argv[4])), bucketFocus = unsigned(std::atoi(argv[5]));
    const int bc[3]    = {std::atoi(argv[6]), std::atoi(argv[7]), std::atoi(argv[8])};
    const int kind     = std::atoi(argv[9]);
    const unsigned seed = unsigned(std::atoi(argv[10]));
    const float theta  = 0.5f;
    // 11th argument "grav": Domain::syncGrav against FocusLet::updateGrav (expansion centres and MAC radii compared as well)
    const bool grav = gArgc > 11 && std::string(argv[11]) == "grav";

    // the same cloud on every rank, every rank keeps a random share
    std::mt19937 gen(seed);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    std::normal_distribution<double> nor(0.0, 1.0);
    std::array<std::array<double, 3>, 4> centers;
    for (auto& c : centers)
        for (double& v : c)
            v = 0.2 + 0.6 * uni(gen);
    const T top = T(1) - T(1) / T(1 << (sizeof(T) == 8 ? 30 : 20));
    std::vector<T> x, y, z, h, m;
    for (size_t i = 0; i < N; ++i)
    {
        double p[3];
        bool blob = kind != 0 && uni(gen) < 0.5;
        int c     = int(uni(gen) * 4) & 3;
        for (int d = 0; d < 3; ++d)
            p[d] = blob ? centers[c][d] + (kind == 2 ? 0.02 : 0.04) * nor(gen) : uni(gen);
        double hh = 0.02 * (0.5 + 0.5 * uni(gen));
        int owner = int(uni(gen) * P) % P;
        if (owner != rank) continue;
        x.push_back(std::min(std::max(T(p[0]), T(0)), top));
        y.push_back(std::min(std::max(T(p[1]), T(0)), top));
        z.push_back(std::min(std::max(T(p[2]), T(0)), top));
        h.push_back(T(hh));
        m.push_back(T(0.5 + hh * 25.0) / T(N)); // (masses that differ from particle to particle)
    }
    std::vector<K> keys(x.size());
    std::vector<T> s1, s2, s3;

    Box<T> box(T(0), T(1), T(0), T(1), T(0), T(1), BoundaryType(bc[0]), BoundaryType(bc[1]), BoundaryType(bc[2]));
    Domain<K, T, CpuTag> dom(rank, P, bucket, bucketFocus, theta, box);

    cstone_hip_ctx* ctx = nullptr;
    if (cstone_hip_ctx_create(&ctx, 0, nullptr, 1) != 0)
    {
        std::fprintf(stderr, "[rank %d] no context: %s\n", rank, cstone_hip_last_error(nullptr));
        return 1;
    }
    MpiComm mc{P, ctx};
    cstone_hip_comm_ops ops{&mc, mpiAllReduce, mpiAllGather, mpiAllToAllV};
    auto letOwner = std::make_unique<cship::FocusLet<K, T>>(ctx, CSTONE_HILBERT, rank, P, bucketFocus, theta, ops);
    cship::FocusLet<K, T>& let = *letOwner;

    float driftTol = 1.05f; // Domain::centerDriftTol_ (R/domain/domain.hpp:665)
    for (int s = 0; s < syncs; ++s)
    {
        if (grav) dom.syncGrav(keys, x, y, z, h, m, std::tuple{}, std::tie(s1, s2, s3));
        else dom.sync(keys, x, y, z, h, std::tuple{}, std::tie(s1, s2, s3));
        MPI_Barrier(MPI_COMM_WORLD);
        const LocalIndex st = dom.startIndex(), en = dom.endIndex();

        // ---- the same inputs for the product's state machine
        cstone_box cb;
        const Box<T>& b = dom.box();
        cb.lim[0] = b.xmin(), cb.lim[1] = b.xmax(), cb.lim[2] = b.ymin(), cb.lim[3] = b.ymax(), cb.lim[4] = b.zmin(),
        cb.lim[5] = b.zmax();
        cb.bc[0] = bc[0], cb.bc[1] = bc[1], cb.bc[2] = bc[2], cb.pad_ = 0;
        const auto& ga = dom.global_;
        std::vector<K> assignment(P + 1);
        for (int r = 0; r <= P; ++r)
            assignment[r] = ga.assignment()[r];
        auto gl = ga.treeLeaves();
        auto gc = ga.nodeCounts();
        OnDevice<K> dKeys(ctx, keys.data() + st, en - st), dGl(ctx, gl.data(), gl.size());
        OnDevice<unsigned> dGc(ctx, gc.data(), gc.size());
        OnDevice<T> dH(ctx, h.data() + st, en - st);
        int rc;
        if (grav)
        {
            OnDevice<T> dX(ctx, x.data() + st, en - st), dY(ctx, y.data() + st, en - st), dZ(ctx, z.data() + st, en - st),
                dM(ctx, m.data() + st, en - st);
            rc = let.updateGrav(cb, dKeys.get(), en - st, assignment.data(), dGl.get(), gl.data(), dGc.get(),
                                int(gl.size()) - 1, dX.get(), dY.get(), dZ.get(), dM.get(), int(8 * sizeof(T)), dH.get(), 1.0f,
                                &driftTol);
        }
        else
        {
            rc = let.update(cb, dKeys.get(), en - st, assignment.data(), dGl.get(), dGc.get(), int(gl.size()) - 1, dH.get(),
                            1.0f);
        }
        int rcAll = rc != 0;
        MPI_Allreduce(MPI_IN_PLACE, &rcAll, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
        if (rcAll)
        {
            if (rc) std::fprintf(stderr, "[rank %d sync %d] FocusLet::update failed: %s\n", rank, s, cstone_hip_last_error(ctx));
            ++failures;
            break;
        }

        // ---- comparison
        std::vector<int> peers = findPeersMac(rank, ga.assignment(), ga.octree(), dom.box(),
                                              grav ? invThetaVecMac(theta) : invThetaMinMac(theta));
        expectEqual("peers", s, peers.data(), peers.size(), let.peers().data(), let.peers().size());
        const auto& ft = dom.focusTree_;
        auto fl        = ft.treeLeaves();
        const int L    = let.numLeaves(), M = let.numNodes();
        expectEqual("focus leaves", s, fl.data(), fl.size(), fetch(ctx, let.leaves(), size_t(L) + 1).data(), size_t(L) + 1);
        expectEqual("focus leaf counts", s, ft.leafCounts().data(), ft.leafCounts().size(),
                    fetch(ctx, let.leafCounts(), size_t(L)).data(), size_t(L));
        expectEqual("focus node counts", s, ft.counts_.data(), ft.counts_.size(),
                    fetch(ctx, let.nodeCounts(), size_t(M)).data(), size_t(M));
        expectEqual("prefixes", s, ft.treeData_.prefixes.data(), size_t(ft.treeData_.numNodes),
                    fetch(ctx, let.prefixes(), size_t(M)).data(), size_t(M));
        expectEqual("child offsets", s, ft.treeData_.childOffsets.data(), size_t(ft.treeData_.numNodes),
                    fetch(ctx, let.childOffsets(), size_t(M)).data(), size_t(M));
        {
            std::vector<int> a, c;
            for (auto pr : ft.assignment())
                a.push_back(pr.start()), a.push_back(pr.end());
            for (auto pr : let.assignment())
                c.push_back(pr.start), c.push_back(pr.end);
            expectEqual("focus assignment", s, a.data(), a.size(), c.data(), c.size());
        }
        auto hf = dom.halos_.haloFlags();
        expectEqual("halo flags", s, hf.data(), size_t(L), fetch(ctx, let.haloFlags(), size_t(L)).data(), size_t(L));
        auto lay = dom.layout();
        expectEqual("layout", s, lay.data(), lay.size(), fetch(ctx, let.layout(), size_t(L) + 1).data(), size_t(L) + 1);
        const uint32_t idx[3]  = {uint32_t(st), uint32_t(en), uint32_t(dom.nParticlesWithHalos())};
        const uint32_t mine[3] = {let.startIndex(), let.endIndex(), let.numParticlesWithHalos()};
        expectEqual("start / end / size", s, idx, 3, mine, 3);
        expectEqual("node centres", s, reinterpret_cast<const T*>(ft.geoCentersAcc_.data()), ft.geoCentersAcc_.size() * 3,
                    fetch(ctx, let.geoCenters(), size_t(M) * 3).data(), size_t(M) * 3);
        expectEqual("node sizes", s, reinterpret_cast<const T*>(ft.geoSizesAcc_.data()), ft.geoSizesAcc_.size() * 3,
                    fetch(ctx, let.geoSizes(), size_t(M) * 3).data(), size_t(M) * 3);
        if (grav)
        {
            // expansion centres (centre of mass) and MAC radii^2 of EVERY node, the MAC marks, the drift tolerance
            expectEqual("expansion centres + MAC radii", s, reinterpret_cast<const T*>(ft.centers_.data()),
                        ft.centers_.size() * 4, fetch(ctx, let.expansionCenters(), size_t(M) * 4).data(), size_t(M) * 4);
            expectEqual("MAC marks", s, ft.macs_.data(), ft.macs_.size(), fetch(ctx, let.macs(), size_t(M)).data(), size_t(M));
            expectEqual("centerDriftTol", s, &dom.centerDriftTol_, 1, &driftTol, 1);
        }
        // the halo exchange: x with its halo ranges wiped must come back as the reference left it (8-byte elements);
        // a 3-byte field derived from the keys as well
        if (let.numParticlesWithHalos() == x.size())
        {
            std::vector<T> xx(x);
            std::fill(xx.begin(), xx.begin() + st, T(-7));
            std::fill(xx.begin() + en, xx.end(), T(-7));
            {
                OnDevice<T> dx(ctx, xx.data(), xx.size());
                if (let.exchangeHalos(dx.get(), int(sizeof(T))) != 0)
                {
                    ++failures;
                    std::fprintf(stderr, "[rank %d sync %d] exchangeHalos failed: %s\n", rank, s, cstone_hip_last_error(ctx));
                }
                xx = fetch(ctx, dx.get(), xx.size());
            }
            expectEqual("halo x", s, x.data(), x.size(), xx.data(), xx.size());
            std::vector<uint16_t> tag(x.size()), want(x.size());
            for (size_t i = 0; i < x.size(); ++i)
                want[i] = uint16_t(keys[i] >> 7), tag[i] = (i >= st && i < en) ? want[i] : uint16_t(0xFFFF);
            {
                OnDevice<uint16_t> dt(ctx, tag.data(), tag.size());
                let.exchangeHalos(dt.get(), 2);
                tag = fetch(ctx, dt.get(), tag.size());
            }
            expectEqual("halo tags", s, want.data(), want.size(), tag.data(), tag.size());
            // x, y, z in one message per peer (exchangeHalosRows): the same particles
            {
                std::vector<T> cx(x), cy(y), cz(z);
                for (auto* v : {&cx, &cy, &cz})
                {
                    std::fill(v->begin(), v->begin() + st, T(-7));
                    std::fill(v->begin() + en, v->end(), T(-7));
                }
                OnDevice<T> dx(ctx, cx.data(), cx.size()), dy(ctx, cy.data(), cy.size()), dz(ctx, cz.data(), cz.size());
                void* arrays[3] = {dx.get(), dy.get(), dz.get()};
                if (let.exchangeHalosRows(arrays, 3, int(sizeof(T))) != 0)
                {
                    ++failures;
                    std::fprintf(stderr, "[rank %d sync %d] exchangeHalosRows failed: %s\n", rank, s,
                                 cstone_hip_last_error(ctx));
                }
                cx = fetch(ctx, dx.get(), cx.size()), cy = fetch(ctx, dy.get(), cy.size()), cz = fetch(ctx, dz.get(), cz.size());
                expectEqual("halo rows x", s, x.data(), x.size(), cx.data(), cx.size());
                expectEqual("halo rows y", s, y.data(), y.size(), cy.data(), cy.size());
                expectEqual("halo rows z", s, z.data(), z.size(), cz.data(), cz.size());
            }
        }
        if (rank == 0)
            std::printf("sync %d: leaves %d, peers %zu, halos in %u, assigned %u\n", s, L, peers.size(),
                        let.numParticlesWithHalos() - (let.endIndex() - let.startIndex()), let.endIndex() - let.startIndex());

        // ---- the client keeps its assigned particles and moves them (as oracle/ref_domain_mpi.cpp; kind 2 pushes
        //      everything along x so that the assignment boundaries move)
        const T c = T(0.01), half = T(0.5);
        for (size_t i = st; i < en; ++i)
        {
            T xo = x[i], yo = y[i], zo = z[i];
            T xn = xo + c * (yo - half), yn = yo + c * (zo - half), zn = zo + c * (xo - half);
            if (kind == 2) xn = xo + T(0.03) * (T(1) - xo);
            x[i] = std::min(std::max(xn, T(0)), top);
            y[i] = std::min(std::max(yn, T(0)), top);
            z[i] = std::min(std::max(zn, T(0)), top);
        }
    }
    {
        // which of the rarer paths this run has exercised, summed over the ranks
        const auto& st = let.stats();
        uint64_t v[9]  = {st.treeUpdates, st.treeBuilds, st.focusTransfers, st.keysTransferred, st.macRefineSteps,
                          st.keysInjected, st.keysRejected, st.leavesFromGlobal, st.convergeSteps};
        MPI_Allreduce(MPI_IN_PLACE, v, 9, MPI_UINT64_T, MPI_SUM, MPI_COMM_WORLD);
        if (rank == 0)
            std::printf("LET_PATHS treeUpdates=%llu treeBuilds=%llu focusTransfers=%llu keysTransferred=%llu "
                        "macRefineSteps=%llu keysInjected=%llu keysRejected=%llu leavesFromGlobal=%llu convergeSteps=%llu\n",
                        (unsigned long long)v[0], (unsigned long long)v[1], (unsigned long long)v[2],
                        (unsigned long long)v[3], (unsigned long long)v[4], (unsigned long long)v[5],
                        (unsigned long long)v[6], (unsigned long long)v[7], (unsigned long long)v[8]);
    }
    letOwner.reset(); // its buffers go before the context does
    cstone_hip_ctx_destroy(ctx);
    return failures;
}

} // namespace

int main(int argc, char** argv)
{
    MPI_Init(&argc, &argv);
    int rank = 0, P = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    rankG = rank;
    gArgc = argc;
    if (argc < 11)
    {
        if (rank == 0) std::fprintf(stderr, "usage: see the head of oracle/let_check.cpp\n");
        MPI_Finalize();
        return 2;
    }
    std::string types = argv[1];
    int bad           = 1;
    if (types == "k64f64") bad = run<uint64_t, double>(rank, P, argv);
    if (types == "k32f32") bad = run<unsigned, float>(rank, P, argv);
    if (types == "k64f32") bad = run<uint64_t, float>(rank, P, argv);
    MPI_Allreduce(MPI_IN_PLACE, &bad, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
    if (rank == 0) std::printf("LET_CHECK %s ranks=%d mismatches=%d\n", bad == 0 ? "OK" : "FAILED", P, bad);
    MPI_Finalize();
    return bad == 0 ? 0 : 1;
}
