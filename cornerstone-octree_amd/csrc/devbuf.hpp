// grow-only device buffer shared by the Domain orchestration files
#pragma once

#include "ctx.hpp"

namespace cship
{

//! grow-only device buffer, growth factor like the reference's reallocate() (R/util/reallocate.hpp:37-47)
struct DevBuf
{
    void* p      = nullptr;
    size_t bytes = 0;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    int ensure(cstone_hip_ctx* ctx, size_t need, bool keep = false)
    {
        if (need <= bytes) return CSTONE_OK;
        size_t want = size_t(double(need) * 1.05) + 256;
        void* q     = nullptr;
        CS_HIP(ctx, hipMalloc(&q, want));
        if (p)
        {
            if (keep) CS_HIP(ctx, hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
            CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            CS_HIP(ctx, hipFree(p));
        }
        p     = q;
        bytes = want;
        return CSTONE_OK;
    }
    template<class V>
    V* as() const
    {
        return static_cast<V*>(p);
    }
};

} // namespace cship
