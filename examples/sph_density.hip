// A client kernel written against include/cstone_hip_device.hpp: SPH density summation while the tree is walked
// (cstone_hip::traverseNeighbors, the equivalent of the reference's traverseNeighbors, R/traversal/find_neighbors.cuh:
// 436-506), compared BIT FOR BIT with the same sum taken over the neighbour lists of cstone_hip_find_neighbors.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -I include examples/sph_density.hip \
//         -L cornerstone-octree_amd/lib -lcstone_hip -o sph_density
//
// Prints one line per configuration and "sph_density OK"; exit code 1 on any mismatch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "cstone_hip.h"
#include "cstone_hip_device.hpp"

#define CHECK(call)                                                                                                    \
    do                                                                                                                 \
    {                                                                                                                  \
        int rc_ = (call);                                                                                              \
        if (rc_ != 0)                                                                                                  \
        {                                                                                                              \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, cstone_hip_last_error(ctx));                      \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

//! cubic spline kernel (3D, support 2h)
template<class T>
__device__ __forceinline__ T splineW(T r, T h)
{
    const T q     = r / h;
    const T sigma = T(1.0 / 3.14159265358979323846) / (h * h * h);
    if (q < T(1)) return sigma * (T(1) - T(1.5) * q * q + T(0.75) * q * q * q);
    if (q < T(2))
    {
        const T t = T(2) - q;
        return sigma * T(0.25) * t * t * t;
    }
    return T(0);
}

//! density while walking the tree: the pair interaction is the functor of traverseNeighbors
template<class T>
__global__ __launch_bounds__(256) void densityTraversal(cstone_hip::OctreeNsView<T> tree, cstone_hip::DeviceBox<T> box,
                                                        const T* x, const T* y, const T* z, const T* h, const T* m,
                                                        uint32_t first, uint32_t last, T* rho, uint32_t* nc)
{
    __shared__ cstone_hip::TraversalStack stacks[4];
    const uint32_t t = first + blockIdx.x * 256 + threadIdx.x;
    const bool valid = t < last;
    const uint32_t i = valid ? t : last - 1;
    const T hi       = h[i];
    T sum            = m[i] * splineW<T>(T(0), hi);
    const uint32_t n = cstone_hip::traverseNeighbors(valid, i, x, y, z, h, tree, box, 1.0f, stacks[threadIdx.x / 64], nullptr,
                                                     [&](uint32_t j, T, T, T, T d2) { sum += m[j] * splineW<T>(T(sqrt(d2)), hi); });
    if (valid) rho[i] = sum, nc[i] = n;
}

//! the same sum over the stored lists (what a client without the device header would do)
template<class T>
__global__ __launch_bounds__(256) void densityFromLists(cstone_hip::DeviceBox<T> box, const T* x, const T* y, const T* z,
                                                        const T* h, const T* m, uint32_t first, uint32_t last,
                                                        const uint32_t* lists, const uint32_t* counts, uint32_t ngmax,
                                                        T* rho)
{
    const uint32_t i = first + blockIdx.x * 256 + threadIdx.x;
    if (i >= last) return;
    const T hi = h[i], xi = x[i], yi = y[i], zi = z[i];
    const T s  = T(2) * hi;
    const bool px = box.bc[0] == 1, py = box.bc[1] == 1, pz = box.bc[2] == 1;
    const bool inside = (xi - s >= box.lo[0]) && (yi - s >= box.lo[1]) && (zi - s >= box.lo[2]) && (xi + s <= box.hi[0]) &&
                        (yi + s <= box.hi[1]) && (zi + s <= box.hi[2]);
    const bool usePbc = (px || py || pz) && !inside;
    T sum             = m[i] * splineW<T>(T(0), hi);
    const uint32_t c  = min(counts[i - first], ngmax);
    for (uint32_t k = 0; k < c; ++k)
    {
        const uint32_t j = lists[size_t(i - first) * ngmax + k];
        T dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
        if (usePbc)
        {
            dx = cstone_hip::detail::foldAxis<T>(dx, box.len[0], box.inv[0], px);
            dy = cstone_hip::detail::foldAxis<T>(dy, box.len[1], box.inv[1], py);
            dz = cstone_hip::detail::foldAxis<T>(dz, box.len[2], box.inv[2], pz);
        }
        sum += m[j] * splineW<T>(T(sqrt(dx * dx + dy * dy + dz * dz)), hi);
    }
    rho[i] = sum;
}

template<class T>
int run(cstone_hip_ctx* ctx, size_t n, int bcx, int bcy, int bcz)
{
    constexpr int rb = 8 * sizeof(T);
    // a cloud with structure: half uniform, half in two blobs
    std::vector<T> x(n), y(n), z(n), h(n), m(n, T(1.0) / T(n));
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd   = [&]()
    {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        return double(s >> 11) / double(1ull << 53);
    };
    for (size_t i = 0; i < n; ++i)
    {
        if (i % 2)
        {
            x[i] = T(rnd()), y[i] = T(rnd()), z[i] = T(rnd());
        }
        else
        {
            const double c = (i % 4) ? 0.3 : 0.7;
            auto blob      = [&]() { return std::min(0.999999, std::max(0.0, c + 0.08 * (rnd() + rnd() + rnd() - 1.5))); };
            x[i] = T(blob()), y[i] = T(blob()), z[i] = T(blob());
        }
        h[i] = T(0.005 + 0.003 * rnd());
    }
    cstone_box box{};
    for (int d = 0; d < 3; ++d)
        box.lim[2 * d] = 0.0, box.lim[2 * d + 1] = 1.0;
    box.bc[0] = bcx, box.bc[1] = bcy, box.bc[2] = bcz;

    void *dk = nullptr, *dx = nullptr, *dy = nullptr, *dz = nullptr, *dh = nullptr, *ds = nullptr, *dm = nullptr;
    CHECK(cstone_hip_malloc(ctx, &dk, n * 8));
    for (void** p : {&dx, &dy, &dz, &dh, &ds, &dm})
        CHECK(cstone_hip_malloc(ctx, p, n * sizeof(T)));
    CHECK(cstone_hip_memset(ctx, dk, 0, n * 8));
    CHECK(cstone_hip_memcpy_h2d(ctx, dx, x.data(), n * sizeof(T)));
    CHECK(cstone_hip_memcpy_h2d(ctx, dy, y.data(), n * sizeof(T)));
    CHECK(cstone_hip_memcpy_h2d(ctx, dz, z.data(), n * sizeof(T)));
    CHECK(cstone_hip_memcpy_h2d(ctx, dh, h.data(), n * sizeof(T)));
    CHECK(cstone_hip_memcpy_h2d(ctx, dm, m.data(), n * sizeof(T)));

    cstone_hip_domain* dom = nullptr;
    CHECK(cstone_hip_domain_create(ctx, &dom, CSTONE_HILBERT, 64, rb, 0, 1, uint32_t(n / 50), 32, 0.5f, &box));
    CHECK(cstone_hip_domain_sync(dom, &dk, &dx, &dy, &dz, &dh, n, &ds, nullptr, nullptr, 0));
    cstone_hip_domain_view v;
    CHECK(cstone_hip_domain_view_get(dom, &v));
    const uint32_t first = v.start_index, last = v.end_index, nt = last - first;

    cstone_hip::OctreeNsView<T> tree{v.child_offsets, v.internal_to_leaf, v.layout, static_cast<const T*>(v.centers),
                                     static_cast<const T*>(v.sizes)};
    const auto dbox = cstone_hip::makeDeviceBox<T>(v.box);
    void *rhoA = nullptr, *rhoB = nullptr, *ncA = nullptr, *lists = nullptr, *counts = nullptr;
    const uint32_t ngmax = 1500;
    CHECK(cstone_hip_malloc(ctx, &rhoA, n * sizeof(T)));
    CHECK(cstone_hip_malloc(ctx, &rhoB, n * sizeof(T)));
    CHECK(cstone_hip_malloc(ctx, &ncA, n * 4));
    CHECK(cstone_hip_malloc(ctx, &lists, size_t(nt) * ngmax * 4));
    CHECK(cstone_hip_malloc(ctx, &counts, size_t(nt) * 4));
    // (the context was created on the null stream: the client's launches are ordered with the library's)
    hipLaunchKernelGGL(densityTraversal<T>, (nt + 255) / 256, 256, 0, nullptr, tree, dbox, (const T*)dx, (const T*)dy,
                       (const T*)dz, (const T*)dh, (const T*)dm, first, last, (T*)rhoA, (uint32_t*)ncA);
    CHECK(cstone_hip_find_neighbors(ctx, rb, dx, dy, dz, dh, first, last, &v.box, v.child_offsets, v.internal_to_leaf,
                                    v.layout, v.centers, v.sizes, 1.0f, ngmax, (uint32_t*)lists, (uint32_t*)counts));
    hipLaunchKernelGGL(densityFromLists<T>, (nt + 255) / 256, 256, 0, nullptr, dbox, (const T*)dx, (const T*)dy,
                       (const T*)dz, (const T*)dh, (const T*)dm, first, last, (const uint32_t*)lists,
                       (const uint32_t*)counts, ngmax, (T*)rhoB);
    CHECK(cstone_hip_ctx_sync(ctx));
    std::vector<T> a(n), b(n);
    std::vector<uint32_t> ca(n), cb(nt);
    CHECK(cstone_hip_memcpy_d2h(ctx, a.data(), rhoA, n * sizeof(T)));
    CHECK(cstone_hip_memcpy_d2h(ctx, b.data(), rhoB, n * sizeof(T)));
    CHECK(cstone_hip_memcpy_d2h(ctx, ca.data(), ncA, n * 4));
    CHECK(cstone_hip_memcpy_d2h(ctx, cb.data(), counts, size_t(nt) * 4));
    size_t bad = 0, overflow = 0;
    double sumRho = 0;
    uint64_t sumNc = 0;
    for (uint32_t i = first; i < last; ++i)
    {
        if (cb[i - first] > ngmax) ++overflow;
        if (ca[i] != cb[i - first] || std::memcmp(&a[i], &b[i], sizeof(T)) != 0) ++bad;
        sumRho += double(a[i]);
        sumNc += ca[i];
    }
    std::printf("f%d bc(%d,%d,%d) n=%zu: mean neighbours %.1f, mean density %.6g, mismatches %zu, list overflows %zu\n", rb, bcx,
                bcy, bcz, n, double(sumNc) / nt, sumRho / nt, bad, overflow);
    for (void* p : {dk, dx, dy, dz, dh, ds, dm, rhoA, rhoB, ncA, lists, counts})
        (void)cstone_hip_free(ctx, p);
    (void)cstone_hip_domain_destroy(dom);
    return (bad == 0 && overflow == 0 && sumNc > 0) ? 0 : 1;
}

int main()
{
    cstone_hip_ctx* ctx = nullptr;
    if (cstone_hip_ctx_create(&ctx, 0, nullptr, 0) != 0)
    {
        std::fprintf(stderr, "no context: %s\n", cstone_hip_last_error(nullptr));
        return 1;
    }
    int rc = 0;
    rc |= run<double>(ctx, 200000, 0, 0, 0);
    rc |= run<double>(ctx, 120000, 1, 1, 0);
    rc |= run<float>(ctx, 150000, 1, 1, 1);
    (void)cstone_hip_ctx_destroy(ctx);
    std::printf(rc == 0 ? "sph_density OK\n" : "sph_density FAILED\n");
    return rc;
}
