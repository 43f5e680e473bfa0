// race_selftest.hip -- kernels with KNOWN races (and their repaired forms) for the detector of tests/simt/simt_race.cpp.
// TEST INFRASTRUCTURE: compiled only for the host, by tests/simt/build.py: build_selftest(), through the same rewrites
// and the same instrumentation as the product's sources; tests/test_simt_race.py says which reports must appear.

// LDS written by one wave and read by another: with / without the barrier
__global__ void rk_barrier(int *out, int fix)
{
    __shared__ int buf[128];
    const int t = threadIdx.x;
    buf[t] = t + 1;
    if (fix) __syncthreads();
    out[blockIdx.x * 128 + t] = buf[(t + 64) & 127];
}

// LDS exchanged between the lanes of ONE wave: with / without a wave-level ordering point
__global__ void rk_lockstep(int *out, int fix)
{
    __shared__ int buf[64];
    const int t = threadIdx.x;
    buf[t] = t + 1;
    if (fix) __builtin_amdgcn_wave_barrier();
    out[t] = buf[t ^ 1];
}

// write-after-read: the second phase overwrites what another wave may still be reading
__global__ void rk_war(int *out, int fix)
{
    __shared__ int buf[128];
    const int t = threadIdx.x;
    buf[t] = t;
    __syncthreads();
    const int v = buf[(t + 64) & 127];
    if (fix) __syncthreads();
    buf[t] = v + 1;
    __syncthreads();
    out[t] = buf[t];
}

// partial results of every workgroup summed by the last arriver.
//   mode 1: the hand-off as the product writes it (stores, barrier, fence, ticket; ticket, barrier, loads)
//   mode 0: no ticket at all -- every workgroup reads workgroup 0's slice
//   mode 2: the ticket is taken BEFORE the slice is written
//   mode 3: ticket in place, but the last arriver's other waves do not wait for the work-item that took it
__global__ void rk_ticket(int *part, int *ticket, int *out, int mode)
{
    __shared__ int last;
    const int t = threadIdx.x, nb = gridDim.x;
    if (mode == 2 && t == 0) last = atomicAdd(ticket, 1) == nb - 1;
    part[blockIdx.x * 128 + t] = t + (int)blockIdx.x;
    if (mode == 0) {
        out[blockIdx.x * 128 + t] = part[t];
        return;
    }
    __syncthreads();
    if (mode != 2 && t == 0) {
        __threadfence();
        last = atomicAdd(ticket, 1) == nb - 1;
    }
    if (mode == 3) {
        if (blockIdx.x == nb - 1 && t >= 64) {        // (the emulator runs the workgroups in order: this IS the last)
            int s = 0;
            for (int b = 0; b < nb; b++) s += part[b * 128 + t];
            out[t] = s;
        }
        return;
    }
    __syncthreads();
    if (last) {
        int s = 0;
        for (int b = 0; b < nb; b++) s += part[b * 128 + t];
        out[t] = s;
    }
}

// the idiom "whoever sees it raises the flag": conflicting stores of one value
__global__ void rk_flag(int *out)
{
    __shared__ int flag;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    if (threadIdx.x & 1) flag = 1;
    __syncthreads();
    out[threadIdx.x] = flag;
}

// accumulation by atomics, read back in the same launch without a hand-off: unordered
__global__ void rk_atomic_then_plain(int *acc, int *out)
{
    // (a work-item that took part in no atomic on the location and is not ordered behind one that did)
    if (blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x == 63) out[0] = *acc;
    } else if (threadIdx.x < 32) {
        atomicAdd(acc, 1);
    }
}

// a per-wave counter every lane reads and ONE lane then advances (the shape of a ballot-ranked placement): in lockstep the
// loads are one instruction and the store a later one; lane by lane the store of one lane meets the loads of the others
__global__ void rk_counter(int *out, int fix)
{
    __shared__ int ctr;
    const int t = threadIdx.x;
    if (t == 0) ctr = 5;
    __syncthreads();
    const int base = ctr;
    if (fix) __builtin_amdgcn_wave_barrier();
    if (t == 63) ctr = base + 64;
    out[t] = base + t;
}

// LDS nobody of this workgroup has written (the GPU hands over the last tenant's bytes), and a store beyond the launch's
// dynamic LDS block (dropped silently by the GPU); a float4 stored as a whole is a memory intrinsic to the compiler
struct rk_f4 { float x, y, z, w; };
__global__ void rk_lds(int *out, int mode)
{
    extern __shared__ int dyn[];
    __shared__ rk_f4 quad[64];
    const int t = threadIdx.x;
    if (mode == 0) {                       // read what nobody wrote
        out[blockIdx.x * 64 + t] = dyn[t];
    } else if (mode == 1) {                // 64 ints were asked for at the launch
        dyn[64 + t] = t;
        out[blockIdx.x * 64 + t] = t;
    } else {                               // whole-struct store, element loads: initialised, and ordered by the barrier
        const rk_f4 v = {1.f * t, 2.f, 3.f, 4.f};
        quad[t] = v;
        __syncthreads();
        out[blockIdx.x * 64 + t] = (int)(quad[63 - t].x + quad[63 - t].w);
    }
}

// a barrier of PART of a workgroup built from an LDS counter (the shape of gridgcn_bwdfused.hip: half_barrier): waves 0 and
// 1 exchange a buffer, waves 2 and 3 do not take part.  mode 0: no synchronisation at all (a race); mode 1: arrivals
// counted by an atomic add after the wave's LDS stores have drained, then a poll -- ordered; mode 2: the poll without the
// drain in front of the arrival is still a release in the emulator's model (documented blind spot: s_waitcnt is what makes
// it one on the GPU), so it is not part of the test
__global__ void rk_ldsbar(int *out, int mode)
{
    __shared__ int buf[128];
    __shared__ int cnt;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) cnt = 0;
    __syncthreads();
    if (wave < 2) {
        buf[t] = t + 1;
        if (mode == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2) __builtin_amdgcn_s_sleep(1);
        }
        out[blockIdx.x * 128 + t] = buf[(t + 64) % 128];
    }
}

// traffic accounting (simt_traffic_enable): 16 workgroups (two per XCD: workgroup w on XCD w mod 8) of 64 work-items.
//   every work-item reads ONE table of 64 ints (256 bytes = 2 lines) -> fetched once per XCD, not per workgroup: 8 x 256
//   work-item t of workgroup w reads a[w * 64 + t] (streamed: 16 x 256 bytes, each line by one XCD)
//   and stores b[(w * 64 + t) * 16]: 4 bytes per 64-byte stride -> 1024 distinct 32-byte sectors
//   workgroup w >= 8 also reads c[(w - 8) * 64 + t], which workgroup w - 8 (same XCD) has WRITTEN: no fetch
__global__ void rk_traffic(const int *a, int *b, int *c, const int *table)
{
    const int w = blockIdx.x, t = threadIdx.x;
    int v = table[t] + a[w * 64 + t];
    if (w < 8) c[w * 64 + t] = v;
    else v += c[(w - 8) * 64 + t];
    b[(w * 64 + t) * 16] = v;
}

extern "C" void rk_run(int which, int arg, int *a, int *b, int *c)
{
    hipStream_t st = nullptr;
    switch (which) {
    case 0: rk_barrier<<<4, 128, 0, st>>>(a, arg); break;
    case 1: rk_lockstep<<<1, 64, 0, st>>>(a, arg); break;
    case 2: rk_war<<<1, 128, 0, st>>>(a, arg); break;
    case 3: rk_ticket<<<6, 128, 0, st>>>(a, b, c, arg); break;
    case 4: rk_flag<<<1, 128, 0, st>>>(a); break;
    case 5: rk_atomic_then_plain<<<3, 64, 0, st>>>(a, b); break;
    case 6: rk_counter<<<1, 64, 0, st>>>(a, arg); break;
    case 7:
        // (the emulator's LDS buffer only grows: a first launch with a large block, so that the out-of-block store of
        //  mode 1 lands inside the buffer, as it lands inside the CU's LDS on the GPU)
        rk_lds<<<1, 64, 4096, st>>>(b, 2);
        rk_lds<<<2, 64, 64 * sizeof(int), st>>>(a, arg);
        break;
    case 8: rk_traffic<<<16, 64, 0, st>>>(a, b, c, a + 2048); break;
    case 9: rk_ldsbar<<<2, 256, 0, st>>>(a, arg); break;
    }
}
