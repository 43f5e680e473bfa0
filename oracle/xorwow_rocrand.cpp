// TEST INFRASTRUCTURE.  rocRAND's XORWOW engine (host-callable, header only: /opt/rocm/include/rocrand/
// rocrand_xorwow.h of the ROCm image) as a C entry point: n-th raw output of xorwow_engine(seed, 0, 0).
// tests/test_oracle.py compares the oracle's restated generator skeleton with it (oracle/gridgcn_oracle.c:
// gridgcn_oracle_xorwow_raw with rocRAND's four scramble constants).  Built by `make -C oracle` when the header exists.
#include <rocrand/rocrand_xorwow.h>
#include <cstdint>
extern "C" uint32_t gridgcn_rocrand_xorwow_raw(uint64_t seed, int n)
{
    rocrand_device::xorwow_engine e(seed, 0, 0);
    uint32_t x = 0;
    for (int i = 0; i < n; i++) x = e();
    return x;
}
