// Internal runtime of libcstone_hip: context, workspace arena, stage timers, launch helpers.
// gfx950 only (wave64, 256 CUs); no CUDA/HIP dual paths.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "cstone_hip.h"

struct cstone_hip_ctx
{
    int device         = 0;
    hipStream_t stream = nullptr;
    bool ownStream     = false;
    int numCu          = 256;
    std::string lastError;

    // grow-only device workspace handed out in 256-byte aligned slices for the duration of one API call
    char* arena       = nullptr;
    size_t arenaBytes = 0;
    size_t arenaUsed  = 0;

    // pinned host block for scalar results (one sync per read-back instead of symbol copies)
    int* hostScalars = nullptr; // [64]
    int* devScalars  = nullptr; // [64]

    // pinned staging ring for small host-to-device uploads that must not synchronise the stream (cstone_hip_upload)
    char* downloadStage  = nullptr; // pinned, 64 KiB: copyToHost() into pageable memory goes through it
    char* uploadStage    = nullptr;
    size_t uploadBytes   = 0;
    size_t uploadCursor  = 0;

    // a second stream of the context's own (ensureAuxStream) for work of one call that does not depend on the rest of
    // it -- Domain::sync gathers x, y, z there while the trees are updated on `stream` --, and the events that order the
    // two; whatever runs there is joined back into `stream` before the call returns
    hipStream_t aux    = nullptr;
    bool auxBusy       = false; // between a fork onto the second stream and its join: a bandwidth kernel may be running
                                // there, and workgroups of 1024 lanes would not find a CU until it has drained (sort.hip)
    hipEvent_t evFork  = nullptr;
    hipEvent_t evJoin  = nullptr;

    // -1 unknown, else result of the one-time LDS atomic ordering probe of the radix sort (sort.hip)
    int ldsOrderOk = -1;

    // Hilbert transducer tables (device_keys.hpp), device copy
    void* hilbertTables = nullptr;

    // stage timers
    int profiling  = 0; // 0 off, 1 every stage, 2 only the stages of the kernels that move the particle arrays
    bool markers   = false; // roctx ranges around every stage (cstone_hip_profile_markers; rocprofv3 --marker-trace)
    int timerDepth = 0; // only the outermost StageTimer of a call records (nested helper launches are part of it)
    struct Bracket
    {
        int stage;
        hipEvent_t a, b;
    };
    std::vector<Bracket> brackets;
    std::vector<hipEvent_t> eventPool;
    double stageMs[CSTONE_NUM_STAGES]   = {0};
    int stageLaunches[CSTONE_NUM_STAGES] = {0};
    std::vector<float> stageSamples[CSTONE_NUM_STAGES]; // individual brackets (bounded), for min / median / max
};

namespace cship
{

//! is ctx a context cstone_hip_ctx_create handed out and cstone_hip_ctx_destroy has not seen yet?  Entry points that
//! may be called during a client's tear-down (free, the destroy functions) ask before they touch the context: a dead or
//! unknown pointer gives CSTONE_E_ARG instead of a use-after-free
bool ctxAlive(const cstone_hip_ctx* ctx);

inline int fail(cstone_hip_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->lastError = buf;
    return code;
}

#define CS_HIP(ctx, call)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return cship::fail(ctx, CSTONE_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,     \
                               __LINE__);                                                                              \
    } while (0)

#define CS_TRY(expr)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        int rc_ = (expr);                                                                                              \
        if (rc_ != CSTONE_OK) return rc_;                                                                              \
    } while (0)

//! creates ctx->aux and its events on first use
int ensureAuxStream(cstone_hip_ctx* ctx);

/*! a few bytes from the device to PINNED host memory (hipHostMalloc: the context's hostScalars, a PinnedBlock ...) on the
 *  context's stream, by an ordinary kernel that stores through the mapped host pointer; the host reads them behind the
 *  next synchronisation of the stream.  hipMemcpyAsync does this with a blit kernel, and blit kernels that ran next to a
 *  bandwidth kernel on the context's second stream were seen to end only when that kernel had drained (0.3 ms, 0.85 ms
 *  for 16 bytes; profiles/r04 traces), with everything queued behind them.  More than 64 KiB go the runtime's way. */
int copyToPinned(cstone_hip_ctx* ctx, void* pinnedDst, const void* devSrc, size_t bytes);
//! the same into ANY host memory (through a pinned block of the context) -- synchronises the stream
int copyToHost(cstone_hip_ctx* ctx, void* dst, const void* devSrc, size_t bytes);
//! the launches (and stage timers) of a scope go to another stream of the context
struct StreamScope
{
    cstone_hip_ctx* ctx;
    hipStream_t saved;
    StreamScope(cstone_hip_ctx* c, hipStream_t s)
        : ctx(c)
        , saved(c->stream)
    {
        ctx->stream = s;
    }
    ~StreamScope() { ctx->stream = saved; }
};

//! reserve bytes from the arena; grows (with a stream sync) when too small. Pointers stay valid until arenaReset.
int arenaReserve(cstone_hip_ctx* ctx, size_t totalBytes);
void* arenaTake(cstone_hip_ctx* ctx, size_t bytes);
inline void arenaReset(cstone_hip_ctx* ctx) { ctx->arenaUsed = 0; }
inline size_t alignUp(size_t b, size_t a = 256) { return (b + a - 1) / a * a; }

//! RAII stage timer: records an event pair around the enclosed launches when profiling is enabled
struct StageTimer
{
    cstone_hip_ctx* ctx;
    int idx = -1;
    StageTimer(cstone_hip_ctx* c, int stage);
    ~StageTimer();
};

//! min/max of up to three equally long coordinate arrays, out = {min0, max0, min1, max1, ...} (primitives.hip)
int minMaxCoordinates(cstone_hip_ctx* ctx, int real_bits, const void* const* xs, int numArrays, size_t n, double* out);
//! the same without the read-back: devOut[2 d] = min, devOut[2 d + 1] = -max as doubles on the device (primitives.hip)
int minMaxCoordinatesDev(cstone_hip_ctx* ctx, int real_bits, const void* const* xs, int numArrays, size_t n,
                         double* devOut);
//! {min, max} per axis as T on the device -> (min, -max) as doubles: the operand of the box all-reduce
int extentsToReduceOperand(cstone_hip_ctx* ctx, int real_bits, const void* extents, double* devOut, double status,
                           const int* counters);

//! encode + the sort's digit histograms in one kernel (sfc.hip); *fused = false: hist untouched (unaligned input)
//! extentsOut (device, 6 reals {xmin, xmax, ymin, ...}, or nullptr): the extents of x, y, z measured by the same pass;
//! written only when *fused comes back true
int computeKeysAndHistogram(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* x, const void* y,
                            const void* z, void* keys, size_t n, const cstone_box& box, uint32_t* hist, bool* fused,
                            int firstDigit = 0, bool honourMarkers = true, void* extentsOut = nullptr);

inline unsigned gridFor(size_t n, unsigned block, unsigned perThread = 1)
{
    size_t per = size_t(block) * perThread;
    return unsigned((n + per - 1) / per);
}

//! encode + SFC ordering with only the digits at or above bit 8 * startPass radix-sorted (sort.hip); honourMarkers =
//! false: keys is pure output (nothing of the caller's in it), it is neither cleared nor read
int sfcKeysAndOrderingHint(cstone_hip_ctx* ctx, int curve, int key_bits, int real_bits, const void* x, const void* y,
                           const void* z, void* keys, uint32_t* ordering, size_t n, const cstone_box& box,
                           void* keys_alt, uint32_t* values_alt, void* temp, size_t temp_bytes, int startPass,
                           int* tooLongDev, bool honourMarkers = true, void* extentsOut = nullptr,
                           bool* extentsMeasured = nullptr);

//! hOut[i] = h[order[i]] for the particles of the leaves [0, numLeaves) (layout: their offsets) and, from the same pass,
//! radii[leaf] = float(max h of the leaf * 2 * ext) (halos.hip): the gather of h and Halos::discover's radii in one
int gatherWithHaloRadii(cstone_hip_ctx* ctx, int h_bits, const void* h, const uint32_t* order, void* hOut,
                        const uint32_t* layout, int numLeaves, float ext, float* radii);

//! cstone_hip_build_octree with a bound on the level of the leaves (tree.hip): the sort of the node keys then skips the
//! digit passes above 3 * deepestLevel + 1 bits
int buildLinkedOctree(cstone_hip_ctx* ctx, int key_bits, const void* leaves, int numLeaves, void* prefixes,
                      int32_t* childOffsets, int32_t* parents, int32_t* levelRange, int32_t* internalToLeaf,
                      int32_t* leafToInternal, int deepestLevel);

//! bottom-up saturating sum over the linked octree, launching only the levels that exist (tree.hip)
int upsweepSumLevels(cstone_hip_ctx* ctx, int numLevelsPlus2, const int32_t* levelRangeHost, const int32_t* levelRange,
                     const int32_t* childOffsets, uint32_t* counts);

} // namespace cship
