// driver.cpp -- the one entry the emulated library adds to the C ABI of include/gridgcn.h (which it exports in full:
// gridgcn_capi.hip is compiled for the host like every other source): the emulator's counters.
#include "simt_hip.h"

extern long long simt_buf_oob;

extern "C" void simt_counters(long long *out)
{
    out[0] = simt_launches;                // kernel launches
    out[1] = simt_rendezvous;              // cross-lane operations resolved
    out[2] = simt_foreign_reads;           // ... that read a lane outside the set executing the operation
    out[3] = simt_divergent_rendezvous;    // ... reached in divergent control flow (the emulator had to guess)
    out[4] = simt_buf_oob;                 // buffer dwords outside their descriptor's range (partial tiles: legitimate)
}

// schedule of the launches that follow (simt_hip.h: simt_order, a bit mask): 0 ascending; 1 / 2 / 4 workgroups / waves /
// lanes descending; 8 workgroups permuted
extern "C" void simt_set_order(int order) { simt_order = order; }
// ... restricted to launches whose kernel expression contains `substr` ("" = all): which kernel is it that depends on order?
extern "C" void simt_set_order_filter(const char *substr)
{
    strncpy(simt_order_filter, substr ? substr : "", sizeof simt_order_filter - 1);
}

// which kernels have run: "kernel expression\tlaunches\n" per kernel (the expression as the launcher spells it, template
// arguments by name); returns the number of kernels
extern "C" int simt_kernel_coverage(char *buf, size_t n)
{
    std::string s;
    for (auto &kv : simt_kernel_launches) s += kv.first + "\t" + std::to_string(kv.second) + "\n";
    if (buf && n) {
        const size_t m = std::min(n - 1, s.size());
        memcpy(buf, s.data(), m);
        buf[m] = 0;
    }
    return (int)simt_kernel_launches.size();
}
