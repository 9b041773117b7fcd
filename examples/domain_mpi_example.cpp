// The multi-rank Domain driven from C++ with MPI as the transport: shows how an application fills
// cstone_hip_comm_ops.  The MPICH of this image is not GPU-aware, so the three collectives stage through host
// buffers; with a GPU-aware MPI (or RCCL) the device pointers are passed on as they are.
//   mpiexec -n 2 ./domain_mpi_example [particles per rank]
// Every rank starts with its own random particles in the unit cube; after each sync the example checks what the
// reference's integration tests check first (T/integration_mpi/domain_nranks.cpp:117-131): particle count conserved
// over the ranks, keys sorted, assigned keys inside the rank's SFC range, ranges ascending with the rank.
#include <mpi.h>

#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#include "cstone_amd/cstone_amd.hpp"

using namespace cstone_amd;

namespace
{
std::vector<char> stageSend, stageRecv;

int allReduce(void*, void* buf, std::size_t count, int dtype, int op)
{
    const std::size_t bytes = count * (dtype == 0 ? 8 : 4);
    stageSend.resize(bytes);
    if (cstone_hip_memcpy_d2h(Context::get(), stageSend.data(), buf, bytes)) return 1;
    int rc = MPI_Allreduce(MPI_IN_PLACE, stageSend.data(), int(count), dtype == 0 ? MPI_DOUBLE : MPI_UINT32_T,
                           op == 0 ? MPI_SUM : MPI_MIN, MPI_COMM_WORLD);
    if (rc != MPI_SUCCESS) return 1;
    return cstone_hip_memcpy_h2d(Context::get(), buf, stageSend.data(), bytes);
}

int allGather(void*, const void* send, void* recv, std::size_t bytes)
{
    int P;
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    stageSend.resize(bytes);
    stageRecv.resize(bytes * P);
    if (bytes && cstone_hip_memcpy_d2h(Context::get(), stageSend.data(), send, bytes)) return 1;
    if (MPI_Allgather(stageSend.data(), int(bytes), MPI_BYTE, stageRecv.data(), int(bytes), MPI_BYTE, MPI_COMM_WORLD) !=
        MPI_SUCCESS)
        return 1;
    return bytes ? cstone_hip_memcpy_h2d(Context::get(), recv, stageRecv.data(), bytes * P) : 0;
}

int allToAllV(void*, const void* send, const std::size_t* sendBytes, void* recv, const std::size_t* recvBytes)
{
    int P;
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    std::vector<int> sc(P), sd(P), rc(P), rd(P);
    std::size_t st = 0, rt = 0;
    for (int p = 0; p < P; ++p)
    {
        sc[p] = int(sendBytes[p]), sd[p] = int(st), st += sendBytes[p];
        rc[p] = int(recvBytes[p]), rd[p] = int(rt), rt += recvBytes[p];
    }
    stageSend.resize(st);
    stageRecv.resize(rt);
    if (st && cstone_hip_memcpy_d2h(Context::get(), stageSend.data(), send, st)) return 1;
    if (MPI_Alltoallv(stageSend.data(), sc.data(), sd.data(), MPI_BYTE, stageRecv.data(), rc.data(), rd.data(), MPI_BYTE,
                      MPI_COMM_WORLD) != MPI_SUCCESS)
        return 1;
    return rt ? cstone_hip_memcpy_h2d(Context::get(), recv, stageRecv.data(), rt) : 0;
}
} // namespace

int main(int argc, char** argv)
{
    MPI_Init(&argc, &argv);
    int rank = 0, P = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    using KeyType = std::uint64_t;
    using T       = double;
    const std::size_t n = argc > 1 ? std::stoul(argv[1]) : 200000;

    std::mt19937 gen(42 + rank);
    std::uniform_real_distribution<T> dis(0, 1);
    std::vector<T> hx(n), hy(n), hz(n), hh(n, 0.01);
    for (auto& v : hx) v = dis(gen);
    for (auto& v : hy) v = dis(gen);
    for (auto& v : hz) v = dis(gen);
    std::vector<float> hid(n);
    std::iota(hid.begin(), hid.end(), float(rank) * 1e6f);
    DeviceVector<T> x(hx.data(), hx.data() + n), y(hy.data(), hy.data() + n), z(hz.data(), hz.data() + n),
        h(hh.data(), hh.data() + n);
    DeviceVector<float> id(hid.data(), hid.data() + n);

    cstone_hip_comm_ops comm{nullptr, allReduce, allGather, allToAllV};
    MultiRankDomain<KeyType, T> domain(rank, P, /*bucketSize*/ unsigned(std::max<std::size_t>(64, n / 100)),
                                       /*bucketSizeFocus*/ 64, Box<T>{0, 1}, comm);
    bool ok = true;
    const T *px = x.data(), *py = y.data(), *pz = z.data(), *ph = h.data();
    const float* pid = id.data();
    std::size_t count = n;
    for (int step = 0; step < 3; ++step)
    {
        domain.sync(px, py, pz, ph, count, pid);
        syncGpu();
        const LocalIndex first = domain.startIndex(), last = domain.endIndex();
        unsigned long long mine = last - first, total = 0;
        MPI_Allreduce(&mine, &total, 1, MPI_UNSIGNED_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
        std::vector<KeyType> keys(domain.nParticlesWithHalos());
        memcpyD2H(domain.keys(), keys.size(), keys.data());
        auto [lo, hi] = domain.assignedRange();
        bool sorted = std::is_sorted(keys.begin(), keys.end());
        bool inside = mine == 0 || (keys[first] >= lo && keys[last - 1] < hi);
        std::vector<unsigned long long> los(P);
        unsigned long long myLo = lo;
        MPI_Allgather(&myLo, 1, MPI_UNSIGNED_LONG_LONG, los.data(), 1, MPI_UNSIGNED_LONG_LONG, MPI_COMM_WORLD);
        bool ascending = std::is_sorted(los.begin(), los.end());
        ok = ok && total == (unsigned long long)n * P && sorted && inside && ascending;
        if (rank == 0)
            std::printf("step %d: %llu particles on %d ranks, rank 0 holds [%u, %u) of %u (halos included): %s\n", step, total,
                        P, first, last, domain.nParticlesWithHalos(), (sorted && inside && ascending) ? "ok" : "BAD");
        // the next step works on the assigned particles in place (the arrays stay valid over the next sync)
        px = domain.x() + first, py = domain.y() + first, pz = domain.z() + first, ph = domain.h() + first;
        pid   = domain.template property<float>(0) + first;
        count = last - first;
    }
    // The same through the reference's own class interface: Domain<KeyType, T> with caller-owned device vectors that
    // come back resized to [halos | assigned | halos] (domain.hpp:196-243), exchangeHalos and reapplySync on top.
    {
        Domain<KeyType, T> dom(rank, P, unsigned(std::max<std::size_t>(64, n / 100)), 64, 0.5f, Box<T>{0, 1}, comm);
        DeviceVector<KeyType> keys(n);
        DeviceVector<T> dx(hx.data(), hx.data() + n), dy(hy.data(), hy.data() + n), dz(hz.data(), hz.data() + n),
            dh(hh.data(), hh.data() + n), scratch;
        DeviceVector<float> tag(hid.data(), hid.data() + n), tagScratch;
        Context::check(cstone_hip_memset(Context::get(), keys.data(), 0, n * sizeof(KeyType)), "memset");
        for (int step = 0; step < 2; ++step)
        {
            // a field that is NOT part of the sync: its values follow later through reapplySync
            auto xin = toHost(dx);
            std::vector<T> hlate(xin.size());
            for (std::size_t i = 0; i < xin.size(); ++i)
                hlate[i] = 3 * xin[i] - 1;
            DeviceVector<T> late(hlate.data(), hlate.data() + hlate.size()), lateScratch;

            dom.sync(keys, dx, dy, dz, dh, std::tie(tag), scratch);
            dom.reapplySync(std::tie(late), lateScratch);
            syncGpu();
            const LocalIndex first = dom.startIndex(), last = dom.endIndex(), all = dom.nParticlesWithHalos();
            bool sizes = dx.size() == all && keys.size() == all && tag.size() == all && late.size() == all;
            auto xs = toHost(dx);
            auto ls = toHost(late);
            bool lateOk = true;
            for (LocalIndex i = first; i < last; ++i)
                lateOk = lateOk && ls[i] == 3 * xs[i] - 1;
            // exchangeHalos: a field known on the assigned range only gets the owners' values in the halo ranges
            std::vector<T> hf(all, T(-7));
            for (LocalIndex i = first; i < last; ++i)
                hf[i] = 2 * xs[i] + 1;
            DeviceVector<T> f(hf.data(), hf.data() + all);
            dom.exchangeHalos(std::tie(f), scratch, scratch);
            auto fs = toHost(f);
            bool halosOk = true;
            for (LocalIndex i = 0; i < all; ++i)
                halosOk = halosOk && fs[i] == 2 * xs[i] + 1;
            auto ns = dom.octreeProperties();
            ok = ok && sizes && lateOk && halosOk && ns.numLeafNodes > 0;
            if (rank == 0)
                std::printf("Domain class, step %d: [%u, %u) of %u, reapplySync %s, exchangeHalos %s, %d tree leaves\n", step,
                            first, last, all, lateOk ? "ok" : "BAD", halosOk ? "ok" : "BAD", ns.numLeafNodes);
        }
    }
    int all = ok ? 1 : 0, every = 0;
    MPI_Allreduce(&all, &every, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
    if (rank == 0) std::printf("multi-rank domain over MPI: %s\n", every ? "all checks passed" : "FAILED");
    MPI_Finalize();
    return every ? 0 : 1;
}
