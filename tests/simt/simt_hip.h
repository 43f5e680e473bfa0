// simt_hip.h -- a wave64 SIMT emulator for the HOST: the HIP kernels of grid_gcn_amd/csrc, compiled by g++ and run on
// the CPU, one fibre per work-item.  TEST INFRASTRUCTURE (tests/test_simt_index.py): it lets the CPU-tier suite execute
// the PRODUCT's kernel source -- not a restatement of it -- against the oracle when no GPU is at hand, and lets a
// kernel change be checked bit for bit before a GPU session is spent on it.  It proves the arithmetic and the
// data flow of a kernel; it says nothing about its speed, its occupancy or the memory model of the real machine.
//
// Execution model
//   * a launch runs its workgroups one after the other; a workgroup is blockDim fibres (ucontext), scheduled wave
//     by wave, lane by lane, each until it blocks;
//   * every cross-lane operation (__ballot, __shfl*, readlane / readfirstlane, __builtin_amdgcn_wave_barrier) is a
//     RENDEZVOUS of the wave: a lane deposits its operand and waits; when no lane of the wave can run any more, the
//     waiting lanes -- all at the same operation in wave-uniform control flow: the lanes the EXEC mask would hold
//     there -- are resolved together (simt_resolve_wave).  Lanes run one after the other BETWEEN rendezvous points, so LDS traffic between the lanes
//     of a wave must be ordered by one -- exactly where the GPU code needs its wave_barrier for the compiler;
//   * __syncthreads is the rendezvous of every live fibre of the workgroup;
//   * atomics are plain read-modify-writes (one fibre runs at a time); fences are nothing; s_waitcnt is a rendezvous of
//     the wave (the hardware executes it per wave: what the other lanes issued before it has been issued);
//   * `__shared__` is a function-local static (workgroups are sequential), `extern __shared__` is rewritten by
//     tests/simt/build.py into a pointer to the launch's dynamic LDS block, `k<<<g, b, lds, s>>>(args)` into
//     simt_launch(g, b, lds, [&] { k(args); }).
//   * a shuffle that reads a lane outside its group is counted (simt_foreign_reads): a kernel relying on it would
//     read a stale register on the GPU as well.
#pragma once
#include <time.h>
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

using std::isfinite;
using std::isnan;
using std::isinf;

// ---- vector types -----------------------------------------------------------------------------------------------
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- qualifiers -------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// tests/simt/build.py --race (-DSIMT_RACE, -fsanitize=thread instrumentation, tests/simt/simt_race.cpp as its runtime):
// the emulator's own code is not instrumented, `__shared__` variables sit in one section (their addresses are LDS)
#ifdef SIMT_RACE
#define SIMT_NOSAN __attribute__((no_sanitize("thread")))
#define __shared__ static __attribute__((section("simt_lds")))
extern "C" void simt_race_atomic(const void *p, int size, int kind, const void *pc);   // 1 read-modify-write, 2 load, 3 store
extern "C" void simt_race_access(const void *p, int size, int write, const void *pc);
extern "C" void simt_race_launch_end();
// (the atomics are calls there: the address they return to is the kernel's line)
#define SIMT_ATOMIC_FN __attribute__((no_sanitize("thread"), noinline))
#define SIMT_RACE_ATOMIC(p, kind) simt_race_atomic((const void *)(p), (int)sizeof(*(p)), (kind), __builtin_return_address(0))
#else
#define SIMT_NOSAN
#define SIMT_ATOMIC_FN
#define __shared__ static
#define SIMT_RACE_ATOMIC(p, kind) ((void)0)
#endif
#define HIP_SYMBOL(x) x

// ---- the host runtime calls the launchers make ------------------------------------------------------------------
typedef void *hipStream_t;
enum hipError_t { hipSuccess = 0, hipErrorUnknown = 999 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
// events hold the host's clock (launches are synchronous): an elapsed time is the emulation's wall time -- never zero, so
// that the library's *_timed entries and bench.py's arithmetic on them can be executed here (tools/simt_bench.py)
typedef void *hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new double(0.0); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete (double *)e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    *(double *)e = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = (float)(*(double *)b - *(double *)a);
    if (!(*ms > 0.f)) *ms = 1e-6f;
    return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

// ---- scheduler --------------------------------------------------------------------------------------------------
enum { SIMT_RUN = 0, SIMT_WAVE = 1, SIMT_BLOCK = 2, SIMT_DONE = 3, SIMT_SPIN = 4 };
enum { SIMT_OP_BALLOT, SIMT_OP_SHFL, SIMT_OP_UP, SIMT_OP_DOWN, SIMT_OP_XOR, SIMT_OP_BAR, SIMT_OP_FIRST, SIMT_OP_XCHG };

struct SimtFiber {
    ucontext_t ctx;
    int state;
    dim3 tidx;
    int op, arg, width;
    const void *site;
    int lane;               // lane of the wave (flat work-item number & 63)
    long long seq;          // cross-lane operations executed so far in this launch
    int bar;                // workgroup barriers passed so far
    uint64_t val, res;
    uint64_t wide[4];       // 32-byte operand of the bf16 matrix instruction
};

inline SimtFiber *simt_cur = nullptr;
inline ucontext_t simt_sched;
inline dim3 simt_blockIdx, simt_blockDim, simt_gridDim;
inline char *simt_dyn_lds = nullptr;
inline long long simt_foreign_reads = 0, simt_launches = 0, simt_rendezvous = 0, simt_divergent_rendezvous = 0;
inline const std::function<void()> *simt_body = nullptr;
// operands of the wave's last rendezvous, as deposited (SIMT_OP_XCHG: matrix instructions read them lane by lane; it
// stays valid until the wave's next rendezvous, i.e. until every lane has consumed it) and who took part
inline uint64_t simt_xchg[64];
inline uint64_t simt_xchg_wide[64][4];
inline bool simt_xchg_in[64];
// Schedule of a launch (simt_set_order, tests only), a bit mask.  0: workgroups, waves and lanes in ascending order.
// 1: workgroups descending; 2: the waves of a workgroup descending; 4: the lanes of a wave descending; 8: workgroups in
// a pseudo-random permutation (seeded by the launch number).  A kernel whose result is meant to be independent of
// arrival order gives the same bytes under all of them.
inline int simt_order = 0;
inline char simt_order_filter[128] = "";
inline std::map<std::string, long long> simt_kernel_launches;   // kernel expression -> launches (coverage: driver.cpp)       // ... only for launches whose kernel expression contains this
inline long long simt_wg_serial = 0;            // workgroups run so far (all launches)
inline size_t simt_lds_bytes = 0;               // size of the running launch's dynamic LDS block
inline const char *simt_kernel_name = nullptr;  // ... its kernel expression
inline const void *simt_kernarg = nullptr;      // __builtin_amdgcn_kernarg_segment_ptr(): first argument of the launch

#define threadIdx (simt_cur->tidx)
#define blockIdx simt_blockIdx
#define blockDim simt_blockDim
#define gridDim simt_gridDim

// AddressSanitizer build (tests/simt/build.py --asan: the CPU build is where a sanitizer can look at these kernels):
// the fibre switches are announced, so that ASan knows which stack it is on
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define SIMT_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#endif
inline const void *simt_sched_stack = nullptr;
inline size_t simt_sched_stack_size = 0;

SIMT_NOSAN static inline void simt_yield(int st)
{
    SimtFiber *f = simt_cur;
    f->state = st;
#ifdef SIMT_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(st == SIMT_DONE ? nullptr : &fake, simt_sched_stack, simt_sched_stack_size);
#endif
    swapcontext(&f->ctx, &simt_sched);
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(fake, &simt_sched_stack, &simt_sched_stack_size);
#endif
}

SIMT_NOSAN static void simt_entry()
{
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &simt_sched_stack, &simt_sched_stack_size);
#endif
    (*simt_body)();
    simt_yield(SIMT_DONE);
}

// scheduler -> fibre
SIMT_NOSAN static inline void simt_resume(SimtFiber *f, char *stack, size_t size)
{
    simt_cur = f;
#ifdef SIMT_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, stack, size);
#endif
    swapcontext(&simt_sched, &f->ctx);
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

SIMT_NOSAN __attribute__((noinline)) static uint64_t simt_collective(int op, uint64_t val, int arg, int width)
{
    SimtFiber *f = simt_cur;
    f->op = op; f->val = val; f->arg = arg; f->width = width;
    f->site = __builtin_return_address(0);
    f->seq++;
    simt_yield(SIMT_WAVE);
    return f->res;
}

// Resolve ONE group of waiting lanes of a wave (lanes[0..n): the wave's fibres).  Called when no lane of the wave can
// run.  Every lane counts the cross-lane operations it has executed in this launch; in wave-uniform control flow --
// the only place the kernels under test put such an operation -- the k-th operation of one lane IS the k-th of every
// other, so the waiting lanes all carry the same count and the same opcode and form one group: the lanes the EXEC
// mask would hold there.  (The call SITE is not the identity of an operation: g++ duplicates a call into both arms
// of an `if (lane == 0)` in front of it.)  If the waiting lanes disagree -- a cross-lane operation inside divergent
// control flow: the hardware would run the branches one after the other and reconverge -- the group that is furthest
// behind (lowest count, then lowest code address) goes on alone, the others wait for it, and the event is counted:
// simt_divergent_rendezvous.  The tests assert that the count is 0, i.e. that no guess was ever made.
SIMT_NOSAN static inline void simt_resolve_wave(SimtFiber *lanes, int n)
{
    int lead = -1, nkeys = 0;
    for (int i = 0; i < n; i++) {
        if (lanes[i].state != SIMT_WAVE) continue;
        bool seen = false;
        for (int j = 0; j < i; j++)
            seen |= lanes[j].state == SIMT_WAVE && lanes[j].seq == lanes[i].seq && lanes[j].op == lanes[i].op;
        if (seen) continue;
        nkeys++;
        if (lead < 0 || lanes[i].seq < lanes[lead].seq ||
            (lanes[i].seq == lanes[lead].seq && (uintptr_t)lanes[i].site < (uintptr_t)lanes[lead].site))
            lead = i;
    }
    const long long seq = lanes[lead].seq;
    const int op = lanes[lead].op;
    if (nkeys > 1) {
        if (simt_divergent_rendezvous < 4 && getenv("SIMT_DEBUG")) {
            fprintf(stderr, "simt: wave blocked at %d different operations:", nkeys);
            for (int i = 0; i < n; i++)
                if (lanes[i].state == SIMT_WAVE)
                    fprintf(stderr, " %d:#%lld/%d@%p", i, lanes[i].seq, lanes[i].op, lanes[i].site);
            fprintf(stderr, "\n");
        }
        simt_divergent_rendezvous++;
    }
    bool in[64] = {false};
    uint64_t ballot = 0;
    int first = -1;
    for (int j = 0; j < n; j++)
        if (lanes[j].state == SIMT_WAVE && lanes[j].seq == seq && lanes[j].op == op) {
            in[j] = true;
            if (first < 0) first = j;
            if (op == SIMT_OP_BALLOT && lanes[j].val) ballot |= 1ull << j;
        }
    simt_rendezvous++;
    for (int j = 0; j < n; j++) {
        if (!in[j]) continue;
        SimtFiber &f = lanes[j];
        const int w = f.width > 0 ? f.width : 64, base = j - (j % w);
        int src = j;
        switch (op) {
        case SIMT_OP_BALLOT: f.res = ballot; break;
        case SIMT_OP_BAR: f.res = 0; break;
        case SIMT_OP_FIRST: f.res = lanes[first].val; break;
        case SIMT_OP_SHFL: src = base + (((f.arg % w) + w) % w); break;
        case SIMT_OP_UP: src = (j % w) >= f.arg ? j - f.arg : j; break;
        case SIMT_OP_DOWN: src = (j % w) + f.arg < w ? j + f.arg : j; break;
        case SIMT_OP_XOR: src = base + ((j % w) ^ f.arg); if (src - base >= w) src = j; break;
        }
        if (op == SIMT_OP_SHFL || op == SIMT_OP_UP || op == SIMT_OP_DOWN || op == SIMT_OP_XOR) {
            if (src < 0 || src >= n) src = j;
            if (!in[src]) simt_foreign_reads++;
            f.res = lanes[src].val;
        }
    }
    for (int j = 0; j < 64; j++) {
        simt_xchg_in[j] = j < n && in[j];
        simt_xchg[j] = j < n ? lanes[j].val : 0;
        if (op == SIMT_OP_XCHG && j < n) memcpy(simt_xchg_wide[j], lanes[j].wide, 32);
    }
    for (int j = 0; j < n; j++)
        if (in[j]) lanes[j].state = SIMT_RUN;
}

inline std::vector<char> simt_stacks;
inline std::vector<SimtFiber> simt_fibers;
inline std::vector<char> simt_lds_buf;
#ifndef SIMT_STACK
#define SIMT_STACK (96 * 1024)
#endif

inline long long simt_spin_rounds = 0;
SIMT_NOSAN static inline void simt_run_block(int nthreads, const dim3 &bd)
{
    simt_spin_rounds = 0;
    simt_wg_serial++;
    if ((int)simt_fibers.size() < nthreads) simt_fibers.resize(nthreads);
    if (simt_stacks.size() < (size_t)nthreads * SIMT_STACK) simt_stacks.resize((size_t)nthreads * SIMT_STACK);
    for (int t = 0; t < nthreads; t++) {
        SimtFiber &f = simt_fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = simt_stacks.data() + (size_t)t * SIMT_STACK;
        f.ctx.uc_stack.ss_size = SIMT_STACK;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, simt_entry, 0);
        f.state = SIMT_RUN;
        f.tidx = dim3(t % bd.x, (t / bd.x) % bd.y, t / (bd.x * bd.y));
        f.val = f.res = 0;
        f.seq = 0;
        f.bar = 0;
        f.lane = t & 63;
    }
    const int nwave = (nthreads + 63) / 64;
    for (;;) {
        bool progress = false;
        for (int wi = 0; wi < nwave; wi++) {
            const int w = (simt_order & 2) ? nwave - 1 - wi : wi;
            SimtFiber *lanes = &simt_fibers[w * 64];
            const int n = std::min(64, nthreads - w * 64);
            for (;;) {
                bool ran = false;
                for (int li = 0; li < n; li++) {
                    const int l = (simt_order & 4) ? n - 1 - li : li;
                    if (lanes[l].state == SIMT_RUN) {
                        simt_resume(&lanes[l], simt_stacks.data() + (size_t)(w * 64 + l) * SIMT_STACK, SIMT_STACK);
                        ran = true;
                    }
                }
                bool waiting = false, spinning = false;
                for (int l = 0; l < n; l++) {
                    waiting |= lanes[l].state == SIMT_WAVE;
                    if (lanes[l].state == SIMT_SPIN) { lanes[l].state = SIMT_RUN; spinning = true; }
                }
                if (ran) progress = true;
                // (a lane polling memory -- s_sleep -- gives the other waves a turn: whatever it waits for is theirs)
                if (!waiting || spinning) break;
                simt_resolve_wave(lanes, n);
                progress = true;
            }
        }
        int nblock = 0, ndone = 0;
        for (int t = 0; t < nthreads; t++) {
            nblock += simt_fibers[t].state == SIMT_BLOCK;
            ndone += simt_fibers[t].state == SIMT_DONE;
        }
        if (ndone == nthreads) break;
        if (nblock + ndone == nthreads) {
            for (int t = 0; t < nthreads; t++)
                if (simt_fibers[t].state == SIMT_BLOCK) simt_fibers[t].state = SIMT_RUN;
            continue;
        }
        if (++simt_spin_rounds > (1ll << 26)) {
            fprintf(stderr, "simt: livelock (a lane polls for something nobody writes)\n");
            abort();
        }
        if (!progress) {
            fprintf(stderr, "simt: deadlock in block (%u,%u,%u): %d at the barrier, %d done of %d\n", simt_blockIdx.x,
                    simt_blockIdx.y, simt_blockIdx.z, nblock, ndone, nthreads);
            abort();
        }
    }
    simt_cur = nullptr;
}

// (the kernel-argument segment: the address of the launch's first argument -- one kernel reads its own argument
//  struct through __builtin_amdgcn_kernarg_segment_ptr())
template <class A, class... R> static inline const void *simt_first_arg(const A &a, const R &...) { return &a; }
static inline const void *simt_first_arg() { return nullptr; }

SIMT_NOSAN static inline void simt_launch_impl(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body,
                                    const void *kernarg, const char *name)
{
    simt_kernarg = kernarg;
    static const bool trace = getenv("SIMT_TRACE") != nullptr;
    if (trace)
        fprintf(stderr, "simt: %s <<<(%u,%u,%u), (%u,%u,%u), %zu>>>\n", name, grid.x, grid.y, grid.z, block.x, block.y,
                block.z, lds_bytes);
    simt_launches++;
    const int order_asked = simt_order;
    if (simt_order_filter[0] && !strstr(name, simt_order_filter)) simt_order = 0;
    simt_kernel_name = name;
    simt_kernel_launches[name]++;
    simt_lds_bytes = lds_bytes;
    if (simt_lds_buf.size() < lds_bytes + 64) simt_lds_buf.resize(lds_bytes + 64);
    simt_dyn_lds = (char *)(((uintptr_t)simt_lds_buf.data() + 63) & ~(uintptr_t)63);
    simt_gridDim = grid;
    simt_blockDim = block;
    simt_body = &body;
    const int nthreads = (int)(block.x * block.y * block.z);
    const unsigned long long nwg = (unsigned long long)grid.x * grid.y * grid.z;
    // order 2: i -> (a i + c) mod nwg with a odd multiplier coprime to nwg (a bijection), seeded by the launch number
    unsigned long long mul = 1, add = 0;
    if ((simt_order & 8) && nwg > 1) {
        mul = (0x9E3779B97F4A7C15ull * (unsigned long long)simt_launches) % nwg | 1;
        auto gcd = [](unsigned long long a, unsigned long long b) { while (b) { const unsigned long long t = a % b; a = b; b = t; } return a; };
        while (gcd(mul, nwg) != 1) mul += 2;
        add = (0xD1B54A32D192ED03ull * (unsigned long long)simt_launches) % nwg;
    }
    for (unsigned long long i = 0; i < nwg; i++) {
        unsigned long long k = i;
        if (simt_order & 8) k = (unsigned long long)(((unsigned __int128)mul * i + add) % nwg);
        if (simt_order & 1) k = nwg - 1 - k;
        simt_blockIdx = dim3((unsigned)(k % grid.x), (unsigned)((k / grid.x) % grid.y), (unsigned)(k / ((unsigned long long)grid.x * grid.y)));
        simt_run_block(nthreads, block);
    }
    simt_body = nullptr;
    simt_order = order_asked;
#ifdef SIMT_RACE
    simt_race_launch_end();
#endif
}

// (the launch statement's closure stays where the compiler put it -- the launching function's frame -- and the
//  std::function holds one reference to it: a closure of more than 16 bytes would otherwise be copied to the heap, and the
//  traffic accounting of the race build would take the work-items' reads of the by-value kernel arguments for device
//  memory traffic; the launching thread's stack it knows)
template <class F>
SIMT_NOSAN static inline void simt_launch(dim3 grid, dim3 block, size_t lds_bytes, const F &fn, const void *kernarg = nullptr,
                               const char *name = "?")
{
    const std::function<void()> body = [&fn]() { fn(); };
    simt_launch_impl(grid, block, lds_bytes, body, kernarg, name);
}

#define SIMT_INL inline __attribute__((always_inline))
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    simt_launch((grid), (block), (lds), [&]() { kernel(__VA_ARGS__); }, simt_first_arg(__VA_ARGS__), #kernel)

// ---- device intrinsics ------------------------------------------------------------------------------------------
SIMT_NOSAN static inline void __syncthreads() { simt_cur->bar++; simt_yield(SIMT_BLOCK); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_barrier() __syncthreads()

template <class T> static inline uint64_t simt_pack(T v)
{
    static_assert(sizeof(T) <= 8, "operand wider than 64 bits");
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    return u;
}
template <class T> static inline T simt_unpack(uint64_t u)
{
    T v;
    memcpy(&v, &u, sizeof(T));
    return v;
}
static SIMT_INL unsigned long long __ballot(int pred) { return simt_collective(SIMT_OP_BALLOT, pred ? 1 : 0, 0, 64); }
static SIMT_INL int __any(int pred) { return simt_collective(SIMT_OP_BALLOT, pred ? 1 : 0, 0, 64) != 0; }
static SIMT_INL int __all(int pred) { return simt_collective(SIMT_OP_BALLOT, pred ? 0 : 1, 0, 64) == 0; }
template <class T> static SIMT_INL T __shfl(T v, int src, int width = 64)
{
    return simt_unpack<T>(simt_collective(SIMT_OP_SHFL, simt_pack(v), src, width));
}
template <class T> static SIMT_INL T __shfl_up(T v, unsigned d, int width = 64)
{
    return simt_unpack<T>(simt_collective(SIMT_OP_UP, simt_pack(v), (int)d, width));
}
template <class T> static SIMT_INL T __shfl_down(T v, unsigned d, int width = 64)
{
    return simt_unpack<T>(simt_collective(SIMT_OP_DOWN, simt_pack(v), (int)d, width));
}
template <class T> static SIMT_INL T __shfl_xor(T v, int m, int width = 64)
{
    return simt_unpack<T>(simt_collective(SIMT_OP_XOR, simt_pack(v), m, width));
}
static SIMT_INL void __builtin_amdgcn_wave_barrier() { (void)simt_collective(SIMT_OP_BAR, 0, 0, 64); }
static SIMT_INL void simt_waitcnt() { (void)simt_collective(SIMT_OP_BAR, 0, 0, 64); }
static SIMT_INL int __builtin_amdgcn_readfirstlane(int v)
{
    return simt_unpack<int>(simt_collective(SIMT_OP_FIRST, simt_pack(v), 0, 64));
}
static SIMT_INL int __builtin_amdgcn_readlane(int v, int lane)
{
    return simt_unpack<int>(simt_collective(SIMT_OP_SHFL, simt_pack(v), lane, 64));
}

SIMT_NOSAN static inline void simt_sleep() { simt_yield(SIMT_SPIN); }
#define __builtin_amdgcn_s_sleep(n) simt_sleep()
#define __builtin_amdgcn_kernarg_segment_ptr() ((void *)simt_kernarg)
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) simt_atomic_load((p))
#define __hip_atomic_store(p, v, order, scope) simt_atomic_store((p), (v))
#define __hip_atomic_fetch_add(p, v, order, scope) simt_fetch_add((p), (v))
template <class T> SIMT_ATOMIC_FN static inline T simt_atomic_load(T *p) { SIMT_RACE_ATOMIC(p, 2); return *p; }
template <class T, class V> SIMT_ATOMIC_FN static inline void simt_atomic_store(T *p, V v) { SIMT_RACE_ATOMIC(p, 3); *p = (T)v; }
template <class T, class V> SIMT_ATOMIC_FN static inline T simt_fetch_add(T *p, V v) { SIMT_RACE_ATOMIC(p, 1); T o = *p; *p = (T)(o + (T)v); return o; }

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] B[2x32].  Lane l supplies A[l & 31][l >> 5] and B[l >> 5][l & 31] and owns
// D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31], r = 0..15.  The guide (cdna_hip_programming.md, "FP32-input MFMA"):
// bit for bit a k-ordered chain of fused multiply-adds, one rounding per product.
typedef float simt_f32x16 __attribute__((ext_vector_type(16)));
SIMT_NOSAN static SIMT_INL simt_f32x16 simt_mfma_f32_32x32x2f32(float a, float b, simt_f32x16 c, int, int, int)
{
    uint64_t v = 0;
    memcpy(&v, &a, 4);
    memcpy((char *)&v + 4, &b, 4);
    (void)simt_collective(SIMT_OP_XCHG, v, 0, 64);
    const int l = simt_cur->lane, col = l & 31, h = l >> 5;
    float b0, b1;
    memcpy(&b0, (const char *)&simt_xchg[col] + 4, 4);
    memcpy(&b1, (const char *)&simt_xchg[col + 32] + 4, 4);
    if (!simt_xchg_in[col] || !simt_xchg_in[col + 32]) simt_foreign_reads++;
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        float a0, a1;
        memcpy(&a0, &simt_xchg[row], 4);
        memcpy(&a1, &simt_xchg[row + 32], 4);
        if (!simt_xchg_in[row] || !simt_xchg_in[row + 32]) simt_foreign_reads++;
        c[r] = __builtin_fmaf(a1, b1, __builtin_fmaf(a0, b0, c[r]));
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 simt_mfma_f32_32x32x2f32

// v_mfma_f32_32x32x16_bf16: D[32x32] += A[32x16] B[16x32], bf16 operands, fp32 accumulate.  Lane l supplies
// A[l & 31][8 (l >> 5) + j] and B[8 (l >> 5) + j][l & 31], j = 0..7.  The products are exact in fp32; the hardware's
// order of adding the sixteen of them is not documented -- ascending k with one rounding per addition here.
static inline float simt_bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
template <class V> SIMT_NOSAN static SIMT_INL simt_f32x16 simt_mfma_f32_32x32x16_bf16(V a, V b, simt_f32x16 c, int, int, int)
{
    static_assert(sizeof(V) == 16, "bf16x8 operands");
    memcpy(simt_cur->wide, &a, 16);
    memcpy(simt_cur->wide + 2, &b, 16);
    (void)simt_collective(SIMT_OP_XCHG, 0, 0, 64);
    const int l = simt_cur->lane, col = l & 31, h = l >> 5;
    uint16_t bk[16];
    memcpy(bk, &simt_xchg_wide[col][2], 16);
    memcpy(bk + 8, &simt_xchg_wide[col + 32][2], 16);
    if (!simt_xchg_in[col] || !simt_xchg_in[col + 32]) simt_foreign_reads++;
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        uint16_t ak[16];
        memcpy(ak, &simt_xchg_wide[row][0], 16);
        memcpy(ak + 8, &simt_xchg_wide[row + 32][0], 16);
        if (!simt_xchg_in[row] || !simt_xchg_in[row + 32]) simt_foreign_reads++;
        float acc = c[r];
        for (int k = 0; k < 16; k++) acc = __builtin_fmaf(simt_bf16_to_f32(ak[k]), simt_bf16_to_f32(bk[k]), acc);
        c[r] = acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 simt_mfma_f32_32x32x16_bf16

static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }   // (the GPU's is an approximation within 1 ulp)
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __mul24(int a, int b) { return (int)((long long)a * b); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __int_as_float(int x) { return simt_unpack<float>((uint64_t)(uint32_t)x); }
static inline int __float_as_int(float x) { return (int)(uint32_t)simt_pack(x); }
static inline float __uint_as_float(unsigned x) { return simt_unpack<float>((uint64_t)x); }
static inline unsigned __float_as_uint(float x) { return (unsigned)simt_pack(x); }
static inline long long wall_clock64() { return 0; }
static inline long long clock64() { return 0; }

template <class T> SIMT_ATOMIC_FN static inline T atomicAdd(T *p, T v) { SIMT_RACE_ATOMIC(p, 1); T o = *p; *p = o + v; return o; }
SIMT_ATOMIC_FN static inline unsigned long long atomicAdd(unsigned long long *p, unsigned v) { SIMT_RACE_ATOMIC(p, 1); unsigned long long o = *p; *p = o + v; return o; }
template <class T> SIMT_ATOMIC_FN static inline T atomicMax(T *p, T v) { SIMT_RACE_ATOMIC(p, 1); T o = *p; *p = o > v ? o : v; return o; }
template <class T> SIMT_ATOMIC_FN static inline T atomicMin(T *p, T v) { SIMT_RACE_ATOMIC(p, 1); T o = *p; *p = o < v ? o : v; return o; }
template <class T> SIMT_ATOMIC_FN static inline T atomicOr(T *p, T v) { SIMT_RACE_ATOMIC(p, 1); T o = *p; *p = o | v; return o; }
template <class T> SIMT_ATOMIC_FN static inline T atomicAnd(T *p, T v) { SIMT_RACE_ATOMIC(p, 1); T o = *p; *p = o & v; return o; }
template <class T> SIMT_ATOMIC_FN static inline T atomicExch(T *p, T v) { SIMT_RACE_ATOMIC(p, 1); T o = *p; *p = v; return o; }
template <class T> SIMT_ATOMIC_FN static inline T atomicCAS(T *p, T c, T v) { SIMT_RACE_ATOMIC(p, 1); T o = *p; if (o == c) *p = v; return o; }
