// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
// C entry points onto the reference's own cstone::Domain<KeyType, T, CpuTag> (single MPI rank, no launcher needed),
// compiled from the headers where they lie under /root/reference/include against the MPICH that ships in this
// image (/opt/conda).  Built by oracle/Makefile into oracle/_ref/libcstone_ref_domain.so (git-ignored).
// Used by tests/golden/make_golden_domain.py to generate Domain::sync fixtures.
#include <mpi.h>

#include <vector>

#include "cstone/domain/domain.hpp"

using namespace cstone;

namespace
{
template<class K, class T>
struct Holder
{
    Domain<K, T, CpuTag> dom;
    std::vector<K> keys;
    std::vector<T> x, y, z, h;
    std::vector<T> s1, s2, s3;
    Holder(unsigned bucket, unsigned bucketFocus, float theta, const Box<T>& box)
        : dom(0, 1, bucket, bucketFocus, theta, box)
    {
    }
};
using H64 = Holder<uint64_t, double>;

void ensureMpi()
{
    int init = 0;
    MPI_Initialized(&init);
    if (!init) MPI_Init(nullptr, nullptr);
}
} // namespace

extern "C"
{

void* cstone_refdom_create(unsigned bucket, unsigned bucketFocus, float theta, const double* lim, const int* bc)
{
    ensureMpi();
    Box<double> box(lim[0], lim[1], lim[2], lim[3], lim[4], lim[5], BoundaryType(bc[0]), BoundaryType(bc[1]),
                    BoundaryType(bc[2]));
    return new H64(bucket, bucketFocus, theta, box);
}

void cstone_refdom_destroy(void* p) { delete (H64*)p; }

//! set the particle arrays (size n); keys_in may carry remove markers (nullptr = all zero)
void cstone_refdom_set(void* p, size_t n, const double* x, const double* y, const double* z, const double* h,
                       const uint64_t* keys_in)
{
    auto* d = (H64*)p;
    d->x.assign(x, x + n), d->y.assign(y, y + n), d->z.assign(z, z + n), d->h.assign(h, h + n);
    if (keys_in) d->keys.assign(keys_in, keys_in + n);
    else d->keys.assign(n, 0);
}

void cstone_refdom_sync(void* p)
{
    auto* d = (H64*)p;
    d->dom.sync(d->keys, d->x, d->y, d->z, d->h, std::tuple{}, std::tie(d->s1, d->s2, d->s3));
}

//! out[0..] = startIndex, endIndex, nParticlesWithHalos, numGlobalLeaves, numFocusLeaves
void cstone_refdom_info(void* p, long* out)
{
    auto* d = (H64*)p;
    out[0]  = d->dom.startIndex();
    out[1]  = d->dom.endIndex();
    out[2]  = d->dom.nParticlesWithHalos();
    out[3]  = d->dom.globalTree().numLeafNodes();
    out[4]  = d->dom.focusTree().treeLeaves().size() - 1;
    auto b  = d->dom.box();
    double lim[6] = {b.xmin(), b.xmax(), b.ymin(), b.ymax(), b.zmin(), b.zmax()};
    std::memcpy(out + 8, lim, sizeof lim);
}

void cstone_refdom_get(void* p, uint64_t* keys, double* x, double* y, double* z, double* h, uint64_t* globalLeaves,
                       uint64_t* focusLeaves, unsigned* focusCounts, unsigned* layout)
{
    auto* d = (H64*)p;
    std::copy(d->keys.begin(), d->keys.end(), keys);
    std::copy(d->x.begin(), d->x.end(), x);
    std::copy(d->y.begin(), d->y.end(), y);
    std::copy(d->z.begin(), d->z.end(), z);
    std::copy(d->h.begin(), d->h.end(), h);
    auto gl = d->dom.globalTree().treeLeaves();
    std::copy(gl.begin(), gl.end(), globalLeaves);
    auto fl = d->dom.focusTree().treeLeaves();
    std::copy(fl.begin(), fl.end(), focusLeaves);
    auto fc = d->dom.focusTree().leafCounts();
    std::copy(fc.begin(), fc.end(), focusCounts);
    auto lo = d->dom.layout();
    std::copy(lo.begin(), lo.end(), layout);
}

} // extern "C"
