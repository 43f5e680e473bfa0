// A launch attribute (hipFuncSetAttribute: MaxDynamicSharedMemorySize) belongs to ONE device, not to the process:
// the "already set" flag of a launcher is therefore kept per device.  Written so that the launchers keep their
// shape --   static GGDevOnce done;  if (!done) { ...set the attributes...; done = true; }
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

struct GGDevOnce {
    std::atomic<unsigned long long> mask{0};          // bit d: device d has the attributes (devices >= 64: set every time)
    static int dev()
    {
        int d = 0;
        return hipGetDevice(&d) == hipSuccess ? d : 64;
    }
    bool operator!() const
    {
        const int d = dev();
        return d >= 64 || !((mask.load(std::memory_order_acquire) >> d) & 1ull);
    }
    GGDevOnce &operator=(bool v)
    {
        const int d = dev();
        if (v && d < 64) mask.fetch_or(1ull << d, std::memory_order_release);
        return *this;
    }
};
