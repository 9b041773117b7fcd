// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
// The reference's own cstone::Domain<KeyType, T, CpuTag> on SEVERAL MPI ranks (mpiexec -n P), compiled from the
// headers where they lie under /root/reference/include against the MPICH of this image (/opt/conda).  Built by
// oracle/Makefile into oracle/_ref/ref_domain_mpi (git-ignored); run by tests/golden/make_golden_domain_mpi.py to
// generate the multi-rank Domain::sync fixtures (decomposition, assigned particles, global tree, box per sync).
//
// usage: ref_domain_mpi <input.bin> <output-prefix> [k64f64 | k32f32 | k64f32 | k32f64]   (default k64f64)
//   input : int64 {N, P, syncs, bucket, bucketFocus, bcx, bcy, bcz}, double lim[6], double x[N], y[N], z[N], h[N],
//           int32 owner[N]   (coordinates are narrowed to T on reading)
//   motion between syncs (assigned particles only, old values on the right-hand side, IEEE double, no FMA):
//           x' = clamp(x + c (y - 0.5)), y' = clamp(y + c (z - 0.5)), z' = clamp(z + c (x - 0.5)), c = 0.01,
//           clamp to [0, 1 - 2^-30] (T = double) resp. [0, 1 - 2^-20] (T = float); all arithmetic in T
//           (the box is the unit cube in every fixture)
//   output <prefix>.rank<r>.bin, per sync: int64 {start, end, withHalos, numGlobalLeaves, P+1}, double lim[6],
//           K range[2], K globalLeaves[L+1], uint32 zeros[L] (padded to an even count),
//           K keys[end-start], T x[end-start], h[end-start], T haloX[], haloY[], haloZ[] (withHalos-(end-start) each),
//           int64 {Lf, startCell, endCell}, K focusLeaves[Lf+1], uint32 focusCounts[Lf], uint32 layout[Lf+1] (each padded
//           to an even count), K haloKeys[], T haloH[]
#include <mpi.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "cstone/domain/domain.hpp"

using namespace cstone;

template<class V>
static void put(FILE* f, const V* p, size_t n)
{
    if (n && std::fwrite(p, sizeof(V), n, f) != n) std::abort();
}

template<class K, class T>
int run(int argc, char** argv, int rank, int P)
{

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

    std::vector<T> x, y, z, h;
    for (size_t i = 0; i < N; ++i)
        if (owner[i] == rank)
            x.push_back(static_cast<T>(gx[i])), y.push_back(static_cast<T>(gy[i])), z.push_back(static_cast<T>(gz[i])),
                h.push_back(static_cast<T>(gh[i]));
    std::vector<K> keys(x.size());
    std::vector<T> s1, s2, s3;

    Box<T> box(static_cast<T>(lim[0]), static_cast<T>(lim[1]), static_cast<T>(lim[2]), static_cast<T>(lim[3]),
               static_cast<T>(lim[4]), static_cast<T>(lim[5]), static_cast<BoundaryType>(hdr[5]),
               static_cast<BoundaryType>(hdr[6]), static_cast<BoundaryType>(hdr[7]));
    Domain<K, T, CpuTag> dom(rank, P, unsigned(hdr[3]), unsigned(hdr[4]), 0.5f, box);

    char name[512];
    std::snprintf(name, sizeof name, "%s.rank%d.bin", argv[2], rank);
    FILE* out = std::fopen(name, "wb");
    if (!out) return 7;

    const T c = T(0.01), half = T(0.5), top = T(1) - T(1) / T(1 << (sizeof(T) == 8 ? 30 : 20));
    for (int s = 0; s < syncs; ++s)
    {
        dom.sync(keys, x, y, z, h, std::tuple{}, std::tie(s1, s2, s3));
        const long st = dom.startIndex(), en = dom.endIndex();
        auto gl   = dom.globalTree().treeLeaves();
        long info[5] = {st, en, long(dom.nParticlesWithHalos()), long(gl.size()) - 1, P + 1};
        put(out, info, 5);
        auto b        = dom.box();
        double bl[6]  = {double(b.xmin()), double(b.xmax()), double(b.ymin()),
                         double(b.ymax()), double(b.zmin()), double(b.zmax())};
        put(out, bl, 6);
        // the assignment is not exposed by Domain: rank r's range starts at its first assigned leaf, i.e. at the
        // lowest key >= which everything is assigned to r; it is recovered by the fixture generator from the
        // assigned keys of all ranks, here we store the focus-tree view of it (startCell/endCell leaf keys)
        auto fl = dom.focusTree().treeLeaves();
        K mine[2] = {fl[dom.startCell()], fl[dom.endCell()]};
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
        // the locally essential (focus) tree and the layout of the particle buffers: int64 {Lf, startCell, endCell},
        // K focusLeaves[Lf + 1], uint32 leafCounts[Lf], uint32 layout[Lf + 1] (both padded to an even count), then the
        // keys and smoothing lengths of the halo particles
        {
            auto flv      = dom.focusTree().treeLeaves();
            auto fcv      = dom.focusTree().leafCounts();
            auto lay      = dom.layout();
            const long Lf = long(flv.size()) - 1;
            long finfo[3] = {Lf, long(dom.startCell()), long(dom.endCell())};
            put(out, finfo, 3);
            put(out, flv.data(), flv.size());
            std::vector<unsigned> c(fcv.begin(), fcv.end());
            if (c.size() & 1) c.push_back(0u);
            put(out, c.data(), c.size());
            std::vector<unsigned> l(lay.begin(), lay.begin() + Lf + 1);
            if (l.size() & 1) l.push_back(0u);
            put(out, l.data(), l.size());
            put(out, keys.data(), size_t(st));
            put(out, keys.data() + en, size_t(wh - en));
            put(out, h.data(), size_t(st));
            put(out, h.data() + en, size_t(wh - en));
        }

        // the client keeps only its assigned particles and moves them
        std::vector<T> nx(x.begin() + st, x.begin() + en), ny(y.begin() + st, y.begin() + en),
            nz(z.begin() + st, z.begin() + en), nh(h.begin() + st, h.begin() + en);
        for (size_t i = 0; i < nx.size(); ++i)
        {
            T xo = nx[i], yo = ny[i], zo = nz[i];
            T xn = xo + c * (yo - half), yn = yo + c * (zo - half), zn = zo + c * (xo - half);
            nx[i] = std::min(std::max(xn, T(0)), top);
            ny[i] = std::min(std::max(yn, T(0)), top);
            nz[i] = std::min(std::max(zn, T(0)), top);
        }
        // Domain::sync expects arrays of the previous size nParticlesWithHalos with the assigned range in place
        std::copy(nx.begin(), nx.end(), x.begin() + st);
        std::copy(ny.begin(), ny.end(), y.begin() + st);
        std::copy(nz.begin(), nz.end(), z.begin() + st);
    }
    std::fclose(out);
    return 0;
}

int main(int argc, char** argv)
{
    MPI_Init(&argc, &argv);
    int rank = 0, P = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    if (argc < 3) return 2;
    std::string types = argc > 3 ? argv[3] : "k64f64";
    int rc            = 8;
    if (types == "k64f64") rc = run<uint64_t, double>(argc, argv, rank, P);
    if (types == "k32f32") rc = run<unsigned, float>(argc, argv, rank, P);
    if (types == "k64f32") rc = run<uint64_t, float>(argc, argv, rank, P);
    if (types == "k32f64") rc = run<unsigned, double>(argc, argv, rank, P);
    MPI_Finalize();
    return rc;
}
