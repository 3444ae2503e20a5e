// device.hip -- process-level launch policy of the library: the CU budget of its persistent grids (grid_cus.hpp).
// The ONE piece of state in the library (a tuning knob, not data): set once before workspaces are sized, read by every launch.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <atomic>

#include "grid_cus.hpp"
#include "shapeclipper_hip.h"

namespace sc {
static std::atomic<int> g_reserved{-1};        // -1: not set yet -> SHAPECLIPPER_RESERVE_CUS (or 0)

static int reserved_cus() {
    int r = g_reserved.load(std::memory_order_relaxed);
    if (r < 0) {
        const char* e = getenv("SHAPECLIPPER_RESERVE_CUS");
        r = e && atoi(e) > 0 ? atoi(e) : 0;
        g_reserved.store(r, std::memory_order_relaxed);
    }
    return r;
}

int grid_cus() {
    static int cu_cache[64] = {0};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev >= 0 && dev < 64 && cu_cache[dev]) cus = cu_cache[dev];
    else {
        if (dev < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (dev >= 0 && dev < 64) cu_cache[dev] = cus;
    }
    // whole XCD rows stay usable: never below 8 CUs, and the reservation is taken off the top
    const int g = cus - reserved_cus();
    return g >= 8 ? g : 8;
}
}  // namespace sc

extern "C" int sc_set_reserved_cus(int n) {
    if (n < 0) return (int)hipErrorInvalidValue;
    sc::g_reserved.store(n, std::memory_order_relaxed);
    return 0;
}

extern "C" int sc_grid_cus(void) { return sc::grid_cus(); }
