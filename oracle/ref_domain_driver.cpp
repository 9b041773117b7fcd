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
//! type-erased access to Domain<K, T, CpuTag>; arrays cross the C boundary as raw bytes of K resp. T
struct HolderBase
{
    virtual ~HolderBase() = default;
    virtual void set(size_t n, const void* x, const void* y, const void* z, const void* h, const void* keysIn) = 0;
    virtual void sync()                                                                                        = 0;
    virtual void info(long* out)                                                                               = 0;
    virtual void get(void* keys, void* x, void* y, void* z, void* h, void* globalLeaves, void* focusLeaves,
                     unsigned* focusCounts, unsigned* layout)                                                  = 0;
};

template<class K, class T>
struct Holder final : HolderBase
{
    Domain<K, T, CpuTag> dom;
    std::vector<K> keys;
    std::vector<T> x, y, z, h;
    std::vector<T> s1, s2, s3;
    Holder(unsigned bucket, unsigned bucketFocus, float theta, const Box<T>& box)
        : dom(0, 1, bucket, bucketFocus, theta, box)
    {
    }
    void set(size_t n, const void* xi, const void* yi, const void* zi, const void* hi, const void* keysIn) override
    {
        auto* px = (const T*)xi;
        auto* py = (const T*)yi;
        auto* pz = (const T*)zi;
        auto* ph = (const T*)hi;
        x.assign(px, px + n), y.assign(py, py + n), z.assign(pz, pz + n), h.assign(ph, ph + n);
        if (keysIn) keys.assign((const K*)keysIn, (const K*)keysIn + n);
        else keys.assign(n, 0);
    }
    void sync() override { dom.sync(keys, x, y, z, h, std::tuple{}, std::tie(s1, s2, s3)); }
    void info(long* out) override
    {
        out[0] = dom.startIndex();
        out[1] = dom.endIndex();
        out[2] = dom.nParticlesWithHalos();
        out[3] = dom.globalTree().numLeafNodes();
        out[4] = dom.focusTree().treeLeaves().size() - 1;
        auto b = dom.box();
        double lim[6] = {double(b.xmin()), double(b.xmax()), double(b.ymin()),
                         double(b.ymax()), double(b.zmin()), double(b.zmax())};
        std::memcpy(out + 8, lim, sizeof lim);
    }
    void get(void* ko, void* xo, void* yo, void* zo, void* ho, void* globalLeaves, void* focusLeaves,
             unsigned* focusCounts, unsigned* layout) override
    {
        std::copy(keys.begin(), keys.end(), (K*)ko);
        std::copy(x.begin(), x.end(), (T*)xo);
        std::copy(y.begin(), y.end(), (T*)yo);
        std::copy(z.begin(), z.end(), (T*)zo);
        std::copy(h.begin(), h.end(), (T*)ho);
        auto gl = dom.globalTree().treeLeaves();
        std::copy(gl.begin(), gl.end(), (K*)globalLeaves);
        auto fl = dom.focusTree().treeLeaves();
        std::copy(fl.begin(), fl.end(), (K*)focusLeaves);
        auto fc = dom.focusTree().leafCounts();
        std::copy(fc.begin(), fc.end(), focusCounts);
        auto lo = dom.layout();
        std::copy(lo.begin(), lo.end(), layout);
    }
};

void ensureMpi()
{
    int init = 0;
    MPI_Initialized(&init);
    if (!init) MPI_Init(nullptr, nullptr);
}

template<class K, class T>
HolderBase* make(unsigned bucket, unsigned bucketFocus, float theta, const double* lim, const int* bc)
{
    Box<T> box(static_cast<T>(lim[0]), static_cast<T>(lim[1]), static_cast<T>(lim[2]), static_cast<T>(lim[3]),
               static_cast<T>(lim[4]), static_cast<T>(lim[5]), static_cast<BoundaryType>(bc[0]),
               static_cast<BoundaryType>(bc[1]), static_cast<BoundaryType>(bc[2]));
    return new Holder<K, T>(bucket, bucketFocus, theta, box);
}
} // namespace

extern "C"
{

//! key_bits in {32, 64}, real_bits in {32, 64}: the instantiation of cstone::Domain to run
void* cstone_refdom_create_typed(int key_bits, int real_bits, unsigned bucket, unsigned bucketFocus, float theta,
                                 const double* lim, const int* bc)
{
    ensureMpi();
    if (key_bits == 64 && real_bits == 64) return make<uint64_t, double>(bucket, bucketFocus, theta, lim, bc);
    if (key_bits == 64 && real_bits == 32) return make<uint64_t, float>(bucket, bucketFocus, theta, lim, bc);
    if (key_bits == 32 && real_bits == 64) return make<unsigned, double>(bucket, bucketFocus, theta, lim, bc);
    if (key_bits == 32 && real_bits == 32) return make<unsigned, float>(bucket, bucketFocus, theta, lim, bc);
    return nullptr;
}

void* cstone_refdom_create(unsigned bucket, unsigned bucketFocus, float theta, const double* lim, const int* bc)
{
    return cstone_refdom_create_typed(64, 64, bucket, bucketFocus, theta, lim, bc);
}

void cstone_refdom_destroy(void* p) { delete (HolderBase*)p; }

//! set the particle arrays (size n, elements of the domain's real type); keys_in may carry remove markers (nullptr =
//! all zero)
void cstone_refdom_set(void* p, size_t n, const void* x, const void* y, const void* z, const void* h,
                       const void* keys_in)
{
    ((HolderBase*)p)->set(n, x, y, z, h, keys_in);
}

void cstone_refdom_sync(void* p) { ((HolderBase*)p)->sync(); }

//! out[0..] = startIndex, endIndex, nParticlesWithHalos, numGlobalLeaves, numFocusLeaves; out[8..13] = box as doubles
void cstone_refdom_info(void* p, long* out) { ((HolderBase*)p)->info(out); }

void cstone_refdom_get(void* p, void* keys, void* x, void* y, void* z, void* h, void* globalLeaves, void* focusLeaves,
                       unsigned* focusCounts, unsigned* layout)
{
    ((HolderBase*)p)->get(keys, x, y, z, h, globalLeaves, focusLeaves, focusCounts, layout);
}

/*! The reference's own Halos::discover + computeLayout (R/halos/halos.hpp:128-222, CPU branch: halo radii = max h per
 *  leaf * 2 * ext, then findHalos) for a pretended assignment [first, last) on ONE rank.  layout_out[numLeaves + 1] is the
 *  layout computeLayout produces: a leaf outside [first, last) has a non-empty range there iff it was flagged as a halo
 *  (and holds particles) -- this pins the halo radius rule, which lives inline inside the member function.
 *  h_bits in {32, 64}; h holds the particles of the leaves [first, last) back to back. */
int cstone_refdom_halo_discover(int key_bits, int real_bits, int h_bits, const void* prefixes, const int* child_offsets,
                                const int* internal_to_leaf, const void* leaves, const unsigned* counts, int num_leaves,
                                int first, int last, const double* lim, const int* bc, const void* h, float ext,
                                unsigned* layout_out)
{
    ensureMpi();
    auto run = [&](auto k, auto t, auto th)
    {
        using K  = decltype(k);
        using T  = decltype(t);
        using Th = decltype(th);
        Box<T> box(static_cast<T>(lim[0]), static_cast<T>(lim[1]), static_cast<T>(lim[2]), static_cast<T>(lim[3]),
                   static_cast<T>(lim[4]), static_cast<T>(lim[5]), static_cast<BoundaryType>(bc[0]),
                   static_cast<BoundaryType>(bc[1]), static_cast<BoundaryType>(bc[2]));
        Halos<K, CpuTag> halos(0);
        std::vector<TreeIndexPair> assignment{TreeIndexPair(first, last)};
        std::vector<LocalIndex> layout(num_leaves + 1, 0);
        std::vector<int> scratch;
        halos.discover((const K*)prefixes, child_offsets, internal_to_leaf, (const K*)leaves,
                       gsl::span<const unsigned>(counts, num_leaves), assignment, layout, box, (const Th*)h, ext, scratch);
        std::vector<int> peers;
        // returns 1 when a halo has no peer to come from (always the case on one rank): the layout is complete before
        halos.computeLayout(gsl::span<const K>((const K*)leaves, num_leaves + 1), gsl::span<const unsigned>(counts, num_leaves),
                            assignment, peers, layout);
        std::copy(layout.begin(), layout.end(), layout_out);
    };
    auto withH = [&](auto k, auto t)
    {
        if (h_bits == 32) run(k, t, float{});
        else run(k, t, double{});
    };
    if (key_bits == 64 && real_bits == 64) withH(uint64_t{}, double{});
    else if (key_bits == 64 && real_bits == 32) withH(uint64_t{}, float{});
    else if (key_bits == 32 && real_bits == 64) withH(unsigned{}, double{});
    else if (key_bits == 32 && real_bits == 32) withH(unsigned{}, float{});
    else return -1;
    return 0;
}

} // extern "C"
