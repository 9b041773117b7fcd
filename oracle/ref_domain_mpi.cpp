// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
// The reference's own cstone::Domain<uint64_t, double, CpuTag> on SEVERAL MPI ranks (mpiexec -n P), compiled from the
// headers where they lie under /root/reference/include against the MPICH of this image (/opt/conda).  Built by
// oracle/Makefile into oracle/_ref/ref_domain_mpi (git-ignored); run by tests/golden/make_golden_domain_mpi.py to
// generate the multi-rank Domain::sync fixtures (decomposition, assigned particles, global tree, box per sync).
//
// usage: ref_domain_mpi <input.bin> <output-prefix>
//   input : int64 {N, P, syncs, bucket, bucketFocus, bcx, bcy, bcz}, double lim[6], double x[N], y[N], z[N], h[N],
//           int32 owner[N]
//   motion between syncs (assigned particles only, old values on the right-hand side, IEEE double, no FMA):
//           x' = clamp(x + c (y - 0.5)), y' = clamp(y + c (z - 0.5)), z' = clamp(z + c (x - 0.5)), c = 0.01,
//           clamp to [0, 1 - 2^-30]  (the box is the unit cube in every fixture)
//   output <prefix>.rank<r>.bin, per sync: int64 {start, end, withHalos, numGlobalLeaves, P+1}, double lim[6],
//           uint64 assignment[P+1], uint64 globalLeaves[L+1], uint32 globalCounts[L] (padded to 8 bytes),
//           uint64 keys[end-start], double x[end-start], h[end-start], double haloX[], haloY[], haloZ[] (withHalos-(end-start) each)
#include <mpi.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cstone/domain/domain.hpp"

using namespace cstone;

template<class V>
static void put(FILE* f, const V* p, size_t n)
{
    if (n && std::fwrite(p, sizeof(V), n, f) != n) std::abort();
}

int main(int argc, char** argv)
{
    MPI_Init(&argc, &argv);
    int rank = 0, P = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    if (argc < 3) return 2;

    FILE* in = std::fopen(argv[1], "rb");
    if (!in) return 3;
    long hdr[8];
    double lim[6];
    if (std::fread(hdr, sizeof(long), 8, in) != 8 || std::fread(lim, sizeof(double), 6, in) != 6) return 4;
    const size_t N = hdr[0];
    if (hdr[1] != P) return 5;
    const int syncs = int(hdr[2]);
    std::vector<double> gx(N), gy(N), gz(N), gh(N);
    std::vector<int> owner(N);
    if (std::fread(gx.data(), 8, N, in) != N || std::fread(gy.data(), 8, N, in) != N ||
        std::fread(gz.data(), 8, N, in) != N || std::fread(gh.data(), 8, N, in) != N ||
        std::fread(owner.data(), 4, N, in) != N)
        return 6;
    std::fclose(in);

    std::vector<double> x, y, z, h;
    for (size_t i = 0; i < N; ++i)
        if (owner[i] == rank) x.push_back(gx[i]), y.push_back(gy[i]), z.push_back(gz[i]), h.push_back(gh[i]);
    std::vector<uint64_t> keys(x.size());
    std::vector<double> s1, s2, s3;

    Box<double> box(lim[0], lim[1], lim[2], lim[3], lim[4], lim[5], BoundaryType(hdr[5]), BoundaryType(hdr[6]),
                    BoundaryType(hdr[7]));
    Domain<uint64_t, double, CpuTag> dom(rank, P, unsigned(hdr[3]), unsigned(hdr[4]), 0.5f, box);

    char name[512];
    std::snprintf(name, sizeof name, "%s.rank%d.bin", argv[2], rank);
    FILE* out = std::fopen(name, "wb");
    if (!out) return 7;

    const double c = 0.01, top = 1.0 - 1.0 / double(1 << 30);
    for (int s = 0; s < syncs; ++s)
    {
        dom.sync(keys, x, y, z, h, std::tuple{}, std::tie(s1, s2, s3));
        const long st = dom.startIndex(), en = dom.endIndex();
        auto gl   = dom.globalTree().treeLeaves();
        long info[5] = {st, en, long(dom.nParticlesWithHalos()), long(gl.size()) - 1, P + 1};
        put(out, info, 5);
        auto b        = dom.box();
        double bl[6]  = {b.xmin(), b.xmax(), b.ymin(), b.ymax(), b.zmin(), b.zmax()};
        put(out, bl, 6);
        // the assignment is not exposed by Domain: rank r's range starts at its first assigned leaf, i.e. at the
        // lowest key >= which everything is assigned to r; it is recovered by the fixture generator from the
        // assigned keys of all ranks, here we store the focus-tree view of it (startCell/endCell leaf keys)
        auto fl = dom.focusTree().treeLeaves();
        uint64_t mine[2] = {fl[dom.startCell()], fl[dom.endCell()]};
        put(out, mine, 2);
        put(out, gl.data(), gl.size());
        std::vector<unsigned> gc(dom.globalTree().numLeafNodes() + (dom.globalTree().numLeafNodes() & 1), 0u);
        // global leaf counts are private to GlobalAssignment: recomputed by the generator from all assigned keys
        put(out, gc.data(), gc.size());
        put(out, keys.data() + st, size_t(en - st));
        put(out, x.data() + st, size_t(en - st));
        put(out, h.data() + st, size_t(en - st));
        // halo particles: [0, start) and [end, withHalos)
        const long wh = long(dom.nParticlesWithHalos());
        for (auto* a : {&x, &y, &z})
        {
            put(out, a->data(), size_t(st));
            put(out, a->data() + en, size_t(wh - en));
        }

        // the client keeps only its assigned particles and moves them
        std::vector<double> nx(x.begin() + st, x.begin() + en), ny(y.begin() + st, y.begin() + en),
            nz(z.begin() + st, z.begin() + en), nh(h.begin() + st, h.begin() + en);
        for (size_t i = 0; i < nx.size(); ++i)
        {
            double xo = nx[i], yo = ny[i], zo = nz[i];
            double xn = xo + c * (yo - 0.5), yn = yo + c * (zo - 0.5), zn = zo + c * (xo - 0.5);
            nx[i] = std::min(std::max(xn, 0.0), top);
            ny[i] = std::min(std::max(yn, 0.0), top);
            nz[i] = std::min(std::max(zn, 0.0), top);
        }
        // Domain::sync expects arrays of the previous size nParticlesWithHalos with the assigned range in place
        std::copy(nx.begin(), nx.end(), x.begin() + st);
        std::copy(ny.begin(), ny.end(), y.begin() + st);
        std::copy(nz.begin(), nz.end(), z.begin() + st);
    }
    std::fclose(out);
    MPI_Finalize();
    return 0;
}
