// runtime part of the C ABI: context, memory, copies, stage timers
// replaces R/cuda/device_vector.{h,cu}, cuda_stubs.h:48-57, errorcheck.cuh (R = reference include/cstone)
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_set>

#include "ctx.hpp"
#include "hilbert_tables.hpp"

namespace cship
{

namespace
{
std::mutex gRegistryMutex;
std::unordered_set<const cstone_hip_ctx*>& registry()
{
    static std::unordered_set<const cstone_hip_ctx*> live;
    return live;
}

// roctx ranges (the reference's tracing hooks): librocprofiler-sdk-roctx (what rocprofv3
// --marker-trace listens to), else the older libroctx64; opened on first use, absent libraries just leave no ranges
using RangePush = int (*)(const char*);
using RangePop  = int (*)();
RangePush gRangePush = nullptr;
RangePop gRangePop   = nullptr;
bool loadRoctx()
{
    static const bool ok = []
    {
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so",
                                 "libroctx64.so.4"})
        {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))
            {
                gRangePush = reinterpret_cast<RangePush>(dlsym(h, "roctxRangePushA"));
                gRangePop  = reinterpret_cast<RangePop>(dlsym(h, "roctxRangePop"));
                if (gRangePush && gRangePop) return true;
            }
        }
        return false;
    }();
    return ok;
}

const char* stageName(int stage)
{
    static const char* names[CSTONE_NUM_STAGES] = {"cstone:encode",      "cstone:sort_hist",   "cstone:sort_pass",
                                                   "cstone:gather",      "cstone:node_counts", "cstone:rebalance",
                                                   "cstone:link_octree", "cstone:halos",       "cstone:neighbors",
                                                   "cstone:minmax",      "cstone:sort_pass_iota", "cstone:resort_bins",
                                                   "cstone:resort_leaves", "cstone:gather_h",  "cstone:place",
                                                   "cstone:stage15"};
    return stage >= 0 && stage < CSTONE_NUM_STAGES ? names[stage] : "cstone:?";
}
} // namespace

bool ctxAlive(const cstone_hip_ctx* ctx)
{
    if (!ctx) return false;
    std::lock_guard<std::mutex> lock(gRegistryMutex);
    return registry().count(ctx) != 0;
}

int ensureAuxStream(cstone_hip_ctx* ctx)
{
    if (ctx->aux) return CSTONE_OK;
    // lowest priority: what runs there (bandwidth-bound bulk moves) must not keep the short kernels of the main stream
    // from the compute units
    int least = 0, greatest = 0;
    CS_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    CS_HIP(ctx, hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, least));
    CS_HIP(ctx, hipEventCreateWithFlags(&ctx->evFork, hipEventDisableTiming));
    CS_HIP(ctx, hipEventCreateWithFlags(&ctx->evJoin, hipEventDisableTiming));
    return CSTONE_OK;
}

int arenaReserve(cstone_hip_ctx* ctx, size_t totalBytes)
{
    totalBytes = alignUp(totalBytes) + 4096;
    if (totalBytes <= ctx->arenaBytes) return CSTONE_OK;
    if (ctx->arenaUsed != 0) return fail(ctx, CSTONE_E_INTERNAL, "arena grow while slices are live");
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->arena) CS_HIP(ctx, hipFree(ctx->arena));
    ctx->arena      = nullptr;
    ctx->arenaBytes = 0;
    size_t want     = totalBytes + totalBytes / 16;
    CS_HIP(ctx, hipMalloc((void**)&ctx->arena, want));
    ctx->arenaBytes = want;
    return CSTONE_OK;
}

void* arenaTake(cstone_hip_ctx* ctx, size_t bytes)
{
    size_t off = alignUp(ctx->arenaUsed);
    if (off + bytes > ctx->arenaBytes) return nullptr;
    ctx->arenaUsed = off + bytes;
    return ctx->arena + off;
}

static hipEvent_t takeEvent(cstone_hip_ctx* ctx)
{
    if (!ctx->eventPool.empty())
    {
        hipEvent_t e = ctx->eventPool.back();
        ctx->eventPool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

StageTimer::StageTimer(cstone_hip_ctx* c, int stage)
    : ctx(c)
{
    if (ctx->markers) (void)gRangePush(stageName(stage)); // (nested ranges nest: every timer pops its own)
    if (!ctx->profiling) return;
    if (ctx->timerDepth++ > 0) return;
    // level 2: only the kernels that move the particle arrays get their two event records (a record costs a few
    // microseconds of stream time; a sync has about forty brackets, eight of them around such kernels)
    const bool heavy = stage == CSTONE_STAGE_ENCODE || stage == CSTONE_STAGE_SORT_PASS ||
                       stage == CSTONE_STAGE_SORT_PASS_IOTA || stage == CSTONE_STAGE_GATHER ||
                       stage == CSTONE_STAGE_RESORT_LEAVES || stage == CSTONE_STAGE_HALOS || stage == CSTONE_STAGE_GATHER_H || stage == CSTONE_STAGE_PLACE ||
                       stage == CSTONE_STAGE_NEIGHBORS;
    if (ctx->profiling == 2 && !heavy) return;
    cstone_hip_ctx::Bracket b{stage, takeEvent(ctx), takeEvent(ctx)};
    (void)hipEventRecord(b.a, ctx->stream);
    idx = int(ctx->brackets.size());
    ctx->brackets.push_back(b);
}

StageTimer::~StageTimer()
{
    if (ctx->markers) (void)gRangePop();
    if (!ctx->profiling) return;
    --ctx->timerDepth;
    if (idx >= 0) (void)hipEventRecord(ctx->brackets[idx].b, ctx->stream);
}

static int drainBrackets(cstone_hip_ctx* ctx)
{
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& b : ctx->brackets)
    {
        float ms = 0;
        CS_HIP(ctx, hipEventElapsedTime(&ms, b.a, b.b));
        ctx->stageMs[b.stage] += ms;
        ctx->stageLaunches[b.stage] += 1;
        if (ctx->stageSamples[b.stage].size() < 8192) ctx->stageSamples[b.stage].push_back(ms);
        ctx->eventPool.push_back(b.a);
        ctx->eventPool.push_back(b.b);
    }
    ctx->brackets.clear();
    return CSTONE_OK;
}

} // namespace cship

using namespace cship;

static thread_local char gCreateError[256] = "null context";

extern "C"
{

int cstone_hip_ctx_create(cstone_hip_ctx** out, int device, void* stream, int private_stream)
{
    if (!out) return CSTONE_E_ARG;
    *out      = nullptr;
    auto* ctx = new cstone_hip_ctx;
    ctx->device = device;
    // no context to hold the message yet: cstone_hip_last_error(NULL) returns it
    auto giveUp = [&](const char* what, hipError_t e)
    {
        std::snprintf(gCreateError, sizeof gCreateError, "cstone_hip_ctx_create: %s on device %d: %s", what, device,
                      hipGetErrorString(e));
        (void)hipGetLastError();
        if (ctx->hilbertTables) (void)hipFree(ctx->hilbertTables);
        if (ctx->devScalars) (void)hipFree(ctx->devScalars);
        if (ctx->hostScalars) (void)hipHostFree(ctx->hostScalars);
        if (ctx->ownStream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return CSTONE_E_HIP;
    };
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return giveUp("hipSetDevice", e);
    if (!private_stream) { ctx->stream = (hipStream_t)stream; }
    else
    {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) return giveUp("hipStreamCreateWithFlags", e);
        ctx->ownStream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->numCu = prop.multiProcessorCount;
    e = hipHostMalloc((void**)&ctx->hostScalars, 64 * sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) return giveUp("hipHostMalloc of the scalar page", e);
    e = hipMalloc((void**)&ctx->devScalars, 64 * sizeof(int));
    if (e != hipSuccess) return giveUp("hipMalloc of the device scalars", e);
    e = hipMemset(ctx->devScalars, 0, 64 * sizeof(int));
    if (e != hipSuccess) return giveUp("hipMemset of the device scalars", e);
    {
        int numStates     = 0;
        HilbertTables t   = makeHilbertTables(&numStates);
        if (numStates != HILBERT_STATES) return giveUp("Hilbert transducer: unexpected number of states", hipErrorUnknown);
        e = hipMalloc(&ctx->hilbertTables, sizeof t);
        if (e != hipSuccess) return giveUp("hipMalloc of the Hilbert tables", e);
        e = hipMemcpy(ctx->hilbertTables, &t, sizeof t, hipMemcpyHostToDevice);
        if (e != hipSuccess) return giveUp("upload of the Hilbert tables", e);
    }
    {
        std::lock_guard<std::mutex> lock(gRegistryMutex);
        registry().insert(ctx);
    }
    *out = ctx;
    return CSTONE_OK;
}

int cstone_hip_ctx_destroy(cstone_hip_ctx* ctx)
{
    if (!ctx) return CSTONE_E_ARG;
    {
        // (a second destroy of the same pointer, or a pointer that never was a context: refused, not dereferenced)
        std::lock_guard<std::mutex> lock(gRegistryMutex);
        if (registry().erase(ctx) == 0) return CSTONE_E_ARG;
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& b : ctx->brackets)
    {
        (void)hipEventDestroy(b.a);
        (void)hipEventDestroy(b.b);
    }
    for (auto e : ctx->eventPool)
        (void)hipEventDestroy(e);
    if (ctx->aux)
    {
        (void)hipStreamSynchronize(ctx->aux);
        (void)hipEventDestroy(ctx->evFork);
        (void)hipEventDestroy(ctx->evJoin);
        (void)hipStreamDestroy(ctx->aux);
    }
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->devScalars) (void)hipFree(ctx->devScalars);
    if (ctx->hilbertTables) (void)hipFree(ctx->hilbertTables);
    if (ctx->hostScalars) (void)hipHostFree(ctx->hostScalars);
    if (ctx->uploadStage) (void)hipHostFree(ctx->uploadStage);
    if (ctx->downloadStage) (void)hipHostFree(ctx->downloadStage);
    if (ctx->ownStream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CSTONE_OK;
}

int cstone_hip_ctx_sync(cstone_hip_ctx* ctx)
{
    if (!ctx) return CSTONE_E_ARG;
    // sticky device-side error word (bounded spins, traversal stack overflow ...)
    CS_TRY(copyToPinned(ctx, ctx->hostScalars + 63, ctx->devScalars + 63, sizeof(int)));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->hostScalars[63] != 0)
    {
        // report once, then re-arm
        (void)hipMemsetAsync(ctx->devScalars + 63, 0, sizeof(int), ctx->stream);
        return fail(ctx, CSTONE_E_INTERNAL, "device-side check failed, code 0x%x", unsigned(ctx->hostScalars[63]));
    }
    return CSTONE_OK;
}

const char* cstone_hip_last_error(cstone_hip_ctx* ctx) { return ctx ? ctx->lastError.c_str() : gCreateError; }

int cstone_hip_device_info(cstone_hip_ctx* ctx, int* num_cu, int* wave_size)
{
    if (!ctx) return CSTONE_E_ARG;
    if (num_cu) *num_cu = ctx->numCu;
    if (wave_size) *wave_size = 64;
    return CSTONE_OK;
}

int cstone_hip_malloc(cstone_hip_ctx* ctx, void** ptr, size_t bytes)
{
    if (!ctxAlive(ctx) || !ptr) return CSTONE_E_ARG;
    *ptr = nullptr;
    if (bytes == 0) return CSTONE_OK;
    CS_HIP(ctx, hipMalloc(ptr, bytes));
    return CSTONE_OK;
}

int cstone_hip_free(cstone_hip_ctx* ctx, void* ptr)
{
    // a client that tears down in the wrong order (buffers freed through a context it has destroyed already) gets an
    // error code, not a use-after-free; the buffer itself is still released (device memory does not belong to the context)
    if (!ctxAlive(ctx))
    {
        if (ptr) (void)hipFree(ptr);
        return CSTONE_E_ARG;
    }
    if (!ptr) return CSTONE_OK;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CS_HIP(ctx, hipFree(ptr));
    return CSTONE_OK;
}

int cstone_hip_memcpy_h2d(cstone_hip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return CSTONE_E_ARG;
    if (bytes == 0) return CSTONE_OK;
    CS_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream)); // pageable source must stay valid: complete before returning
    return CSTONE_OK;
}

namespace
{
/*! a few bytes from the pinned ring to the device with an ordinary kernel (one lane per byte, or per word when both
 *  ends allow).  hipMemcpyAsync does the same with a blit kernel that ends on a system-scope release: while a bandwidth
 *  kernel runs on the second stream (placeColumnsKernel of the multi-rank sync) that release did not complete before
 *  the other kernel had finished (profiles/r04_mr_sync_api_sequence.json: a 264-byte copy that lasted 102 us) and
 *  everything queued behind the copy waited with it */
__global__ __launch_bounds__(256) void smallCopyKernel(char* __restrict__ dst, const char* __restrict__ src, size_t bytes,
                                                           bool words)
{
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (words)
    {
        if (i * 4 < bytes) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
    }
    else if (i < bytes) { dst[i] = src[i]; }
}
} // namespace

} // extern "C"

namespace cship
{
static bool runtimeCopies()
{
    static const bool v = std::getenv("CSTONE_D2H_BLIT") != nullptr; // (A/B: the runtime's copies instead)
    return v;
}

int copyToPinned(cstone_hip_ctx* ctx, void* pinnedDst, const void* devSrc, size_t bytes)
{
    if (bytes == 0) return CSTONE_OK;
    if (bytes > (size_t(64) << 10) || runtimeCopies())
    {
        CS_HIP(ctx, hipMemcpyAsync(pinnedDst, devSrc, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return CSTONE_OK;
    }
    const bool words = (reinterpret_cast<uintptr_t>(pinnedDst) % 4 == 0) && (reinterpret_cast<uintptr_t>(devSrc) % 4 == 0) &&
                       (bytes % 4 == 0);
    const size_t lanes = words ? bytes / 4 : bytes;
    hipLaunchKernelGGL(smallCopyKernel, dim3(unsigned((lanes + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<char*>(pinnedDst), static_cast<const char*>(devSrc), bytes, words);
    CS_HIP(ctx, hipGetLastError());
    return CSTONE_OK;
}

int copyToHost(cstone_hip_ctx* ctx, void* dst, const void* devSrc, size_t bytes)
{
    if (bytes == 0) return CSTONE_OK;
    constexpr size_t stageBytes = size_t(64) << 10;
    if (bytes > stageBytes || runtimeCopies())
    {
        CS_HIP(ctx, hipMemcpyAsync(dst, devSrc, bytes, hipMemcpyDeviceToHost, ctx->stream));
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return CSTONE_OK;
    }
    if (!ctx->downloadStage)
        CS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->downloadStage), stageBytes, hipHostMallocDefault));
    CS_TRY(copyToPinned(ctx, ctx->downloadStage, devSrc, bytes));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(dst, ctx->downloadStage, bytes);
    return CSTONE_OK;
}
} // namespace cship

extern "C"
{

int cstone_hip_upload(cstone_hip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return CSTONE_E_ARG;
    if (bytes == 0) return CSTONE_OK;
    constexpr size_t stageBytes = size_t(1) << 20;
    if (!ctx->uploadStage)
    {
        CS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->uploadStage), stageBytes, hipHostMallocDefault));
        ctx->uploadBytes = stageBytes, ctx->uploadCursor = 0;
    }
    if (bytes > ctx->uploadBytes / 4) return cstone_hip_memcpy_h2d(ctx, dst, src, bytes); // large: the plain, synchronising copy
    size_t off = (ctx->uploadCursor + 63) & ~size_t(63);
    if (off + bytes > ctx->uploadBytes)
    {
        // wrap around: the copies queued from the ring so far must have left it
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        off = 0;
    }
    std::memcpy(ctx->uploadStage + off, src, bytes);
    static const bool blit = std::getenv("CSTONE_UPLOAD_BLIT") != nullptr; // (A/B: the runtime's copy instead)
    if (blit) { CS_HIP(ctx, hipMemcpyAsync(dst, ctx->uploadStage + off, bytes, hipMemcpyHostToDevice, ctx->stream)); }
    else
    {
        const bool words = (reinterpret_cast<uintptr_t>(dst) % 4 == 0) && (bytes % 4 == 0);
        const size_t lanes = words ? bytes / 4 : bytes;
        hipLaunchKernelGGL(smallCopyKernel, dim3(unsigned((lanes + 255) / 256)), dim3(256), 0, ctx->stream,
                           static_cast<char*>(dst), ctx->uploadStage + off, bytes, words);
        CS_HIP(ctx, hipGetLastError());
    }
    ctx->uploadCursor = off + bytes;
    return CSTONE_OK;
}

int cstone_hip_memcpy_d2h(cstone_hip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return CSTONE_E_ARG;
    if (bytes == 0) return CSTONE_OK;
    return copyToHost(ctx, dst, src, bytes);
}

int cstone_hip_memcpy_d2d(cstone_hip_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return CSTONE_E_ARG;
    if (bytes == 0) return CSTONE_OK;
    CS_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return CSTONE_OK;
}

int cstone_hip_memset(cstone_hip_ctx* ctx, void* dst, int value, size_t bytes)
{
    if (!ctx) return CSTONE_E_ARG;
    if (bytes == 0) return CSTONE_OK;
    CS_HIP(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return CSTONE_OK;
}

int cstone_hip_profile_enable(cstone_hip_ctx* ctx, int on)
{
    if (!ctx) return CSTONE_E_ARG;
    CS_TRY(drainBrackets(ctx));
    ctx->profiling = on < 0 ? 0 : (on > 2 ? 1 : on);
    return CSTONE_OK;
}

int cstone_hip_profile_markers(cstone_hip_ctx* ctx, int on)
{
    if (!ctx) return CSTONE_E_ARG;
    if (on && !loadRoctx()) return fail(ctx, CSTONE_E_INTERNAL, "profile_markers: no roctx library could be opened");
    ctx->markers = on != 0;
    return CSTONE_OK;
}

int cstone_hip_profile_reset(cstone_hip_ctx* ctx)
{
    if (!ctx) return CSTONE_E_ARG;
    CS_TRY(drainBrackets(ctx));
    for (int s = 0; s < CSTONE_NUM_STAGES; ++s)
        ctx->stageMs[s] = 0, ctx->stageLaunches[s] = 0, ctx->stageSamples[s].clear();
    return CSTONE_OK;
}

int cstone_hip_profile_get(cstone_hip_ctx* ctx, int stage, double* total_ms, int* launches)
{
    if (!ctx || stage < 0 || stage >= CSTONE_NUM_STAGES) return CSTONE_E_ARG;
    CS_TRY(drainBrackets(ctx));
    if (total_ms) *total_ms = ctx->stageMs[stage];
    if (launches) *launches = ctx->stageLaunches[stage];
    return CSTONE_OK;
}

int cstone_hip_profile_get_spread(cstone_hip_ctx* ctx, int stage, double* min_ms, double* median_ms, double* max_ms)
{
    if (!ctx || stage < 0 || stage >= CSTONE_NUM_STAGES) return CSTONE_E_ARG;
    CS_TRY(drainBrackets(ctx));
    std::vector<float> v = ctx->stageSamples[stage];
    std::sort(v.begin(), v.end());
    if (min_ms) *min_ms = v.empty() ? 0.0 : v.front();
    if (median_ms) *median_ms = v.empty() ? 0.0 : v[v.size() / 2];
    if (max_ms) *max_ms = v.empty() ? 0.0 : v.back();
    return CSTONE_OK;
}

} // extern "C"
