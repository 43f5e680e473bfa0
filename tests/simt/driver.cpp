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
