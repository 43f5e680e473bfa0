// simt_race.cpp -- a data-race detector for the product's kernels on the CPU build.  TEST INFRASTRUCTURE.
//
// tests/simt/build.py --race compiles every kernel source with clang's ThreadSanitizer INSTRUMENTATION
// (-fsanitize=thread: a call to __tsan_read<N> / __tsan_write<N> in front of every load and store that is not a private
// stack slot) but links THIS file instead of the ThreadSanitizer runtime.  ThreadSanitizer's own runtime models
// threads and mutexes; what orders two accesses on the GPU is something else, and the emulator knows it exactly:
//
//   two accesses to the same location, at least one a store, not both atomic, are ORDERED iff
//     * they belong to different launches (a kernel boundary orders everything), or
//     * same work-item, or
//     * same workgroup and a __syncthreads lies between them (the first was made before the k-th barrier of its
//       work-item, the second after the k-th barrier of its own: barrier counts differ), or
//     * same wave and a cross-lane operation of the wave (ballot, shuffle, readlane, wave_barrier, s_waitcnt, a matrix
//       instruction) lies between them (the lanes' counts of such operations differ) -- what the hardware's in-order
//       LDS / memory pipeline gives a wave once the compiler is told not to move the accesses across that point, or
//     * same workgroup, different waves, and the first wave RELEASED (an atomic read-modify-write or atomic store on an LDS
//       location, executed after the access in the sense above) where the second ACQUIRED (an atomic load or read-modify-
//       write of that location executed before its access): barriers of PART of a workgroup built from an LDS counter
//       (gridgcn_bwdfused.hip: the four waves of a half count their arrivals and poll), or
//     * different workgroups, and the first workgroup RELEASED (an atomic read-modify-write or atomic store executed
//       after the access: a barrier, or a wave operation of the same wave, or the same work-item in between) on a
//       location on which the second ACQUIRED (an atomic read-modify-write or atomic load executed before its access)
//       -- the ticket / last-arriver hand-offs; transitive through a third workgroup.
//   Everything else is reported, once per (kernel, kind, pair of code addresses):
//     INTRA  same workgroup, different waves, no barrier between       -- a missing __syncthreads
//     WAVE   same wave, different lanes, no wave operation between      -- the kernel leans on lockstep there
//     INTER  different workgroups of one launch                         -- unordered global traffic
//   Two more things the same shadow state shows, reported in the same table:
//     UNINIT a load of an LDS word no work-item of THIS workgroup has stored (LDS is not cleared between workgroups: the
//            GPU hands over whatever the last tenant left)
//     LDSOOB an access beyond the launch's dynamic LDS block (the GPU drops such a store and reads 0: silently)
//
// LDS (the launch's dynamic block and every `__shared__` variable: section "simt_lds") belongs to one workgroup: an
// entry left by an earlier workgroup is stale, not a conflict.  The fibres' stacks are private.
//
// What it cannot see: accesses the compiler kept in registers or removed (it is the -O1 host build that runs), the
// hardware's memory model beyond "atomics synchronise" (a relaxed ticket counts as release + acquire here: the fences
// that make it so on the GPU are checked by tests/test_isa_handoff.py on the GPU ISA), and anything the tests do not
// execute.
//
// TRAFFIC ACCOUNTING (simt_traffic_enable(1); tools/simt_traffic.py).  The same hooks see every global load and store
// with its address, so a launch's memory footprint can be COUNTED instead of estimated.  Per launch, with workgroup
// (x, y, z) placed on XCD  linear index mod 8  (the dispatch order of MI355X: cdna_hip_programming.md):
//   requested   bytes the work-items asked for (sum of access sizes): what an L1-less, L2-less machine would move
//   fetched     128-byte lines x 128 that an XCD reads WITHOUT having read or written them earlier in the launch, summed
//               over the eight XCDs: the L2-miss traffic of eight private, unbounded L2s that start the launch cold --
//               what TCC_EA_RDREQ (FETCH_SIZE) counts when nothing of a launch's working set is evicted inside the
//               launch and nothing is inherited from the launch in front
//   written     distinct 32-byte sectors x 32 stored to: the write-back of an L2 that merges all stores to a sector
// Exact for what they define, hardware-independent, and a bracket for the counters: PMC fetch <= `fetched` when a
// launch inherits lines from its predecessor in the same XCD's L2 (the index build relies on it), PMC fetch >= `fetched`
// when the working set of the workgroups in flight exceeds 4 MB per XCD.  Checked against the round-5 counter tables
// (profiles/traffic.json) in profiles/r6_emulated_traffic.txt.
#include "simt_hip.h"

#include <dlfcn.h>
#include <link.h>
#include <pthread.h>

#include <map>
#include <string>
#include <unordered_map>

extern "C" {
extern char __start_simt_lds[] __attribute__((weak));
extern char __stop_simt_lds[] __attribute__((weak));
}

namespace {

struct Acc {                 // one recorded access: 20 bytes
    uint32_t gen;            // launch serial (0 = empty)
    uint32_t wg_tid;         // workgroup serial of the launch << 10 | work-item
    uint32_t seq;            // the work-item's count of cross-lane operations
    uint32_t pc_fl;          // code address index << 2 | atomic << 1 | write
    uint16_t bar;            // ... of workgroup barriers
};
struct Cell { Acc w, r0, r1, r2; };   // last store; latest load, latest load of another lane of r0's wave, of another wave

enum { PAGE_SHIFT = 12, GRAN = 1024 };
struct Shadow {
    std::unordered_map<uintptr_t, Cell *> pages;
    uintptr_t last_page = ~(uintptr_t)0;
    Cell *last = nullptr;
    int shift;               // log2 of the granule in bytes (2: words, 0: bytes)
    explicit Shadow(int s) : shift(s) {}
    Cell *cell(uintptr_t a)
    {
        const uintptr_t g = a >> shift, page = g / GRAN;
        if (page != last_page) {
            auto it = pages.find(page);
            if (it == pages.end()) it = pages.emplace(page, (Cell *)calloc(GRAN, sizeof(Cell))).first;
            last_page = page;
            last = it->second;
        }
        return last + (g % GRAN);
    }
    void clear()
    {
        for (auto &p : pages) free(p.second);
        pages.clear();
        last_page = ~(uintptr_t)0;
        last = nullptr;
    }
};
Shadow sh_word(2), sh_byte(0);

struct Rel { uint32_t wg, tid, seq; uint16_t bar; };                  // a release: who, and where in its program
struct Edge { Rel rel; uint32_t acq_tid, acq_seq; uint16_t acq_bar; };  // ... acquired by the current workgroup
std::unordered_map<uintptr_t, std::map<uint32_t, Rel>> released;       // atomic location -> latest release per workgroup
std::map<uint32_t, Edge> acquired;                                     // of the workgroup that is running
uint32_t acquired_wg = ~0u, acquired_gen = 0;
// ... and inside the running workgroup, through LDS atomics: location -> latest release per wave; per acquiring work-item
// the latest edge from every releasing wave (checks are made when the later access happens, so the latest edge is the one
// that covers the most)
struct RelW { uint32_t tid, seq; uint16_t bar; };
struct EdgeW { RelW rel; uint32_t acq_seq; uint16_t acq_bar; };
std::unordered_map<uintptr_t, std::unordered_map<uint32_t, RelW>> lds_released;
std::unordered_map<uint32_t, std::unordered_map<uint32_t, EdgeW>> lds_acquired;    // [work-item][releasing wave]

std::vector<const void *> pcs(1, nullptr);
std::unordered_map<const void *, uint32_t> pc_index;
uint32_t pc_of(const void *pc)
{
    auto it = pc_index.find(pc);
    if (it != pc_index.end()) return it->second;
    if (pcs.size() >= (1u << 29)) return 0;
    pc_index[pc] = (uint32_t)pcs.size();
    pcs.push_back(pc);
    return (uint32_t)(pcs.size() - 1);
}

struct Report {
    long long count = 0;
    long long same = 0;      // store/store conflicts in which the second store wrote the value that was there
    uintptr_t addr = 0;
    uint32_t wg0 = 0, tid0 = 0, wg1 = 0, tid1 = 0;
    bool w0 = false, w1 = false, lds = false;
};
struct Key {
    std::string kernel;
    int kind;
    const void *pc0, *pc1;
    bool operator<(const Key &o) const
    {
        if (kernel != o.kernel) return kernel < o.kernel;
        if (kind != o.kind) return kind < o.kind;
        if (pc0 != o.pc0) return pc0 < o.pc0;
        return pc1 < o.pc1;
    }
};
std::map<Key, Report> reports;
long long n_access = 0, n_conflict = 0;
bool enabled = true;
uint32_t cur_gen = 0;
const char *KIND[] = {"INTRA", "WAVE", "INTER", "UNINIT", "LDSOOB"};

// ---- traffic accounting ----------------------------------------------------------------------------------------------
bool traffic_on = false;
struct TLine { uint8_t touched_xcd = 0, wsect = 0; };      // per 128-byte line of the current launch
std::unordered_map<uintptr_t, TLine> t_lines;
struct TPc { long long fetched = 0, written = 0, req_ld = 0, req_st = 0; };
struct TStat {
    long long launches = 0, wgs = 0, req_ld = 0, req_st = 0, fetched = 0, written = 0;
    std::unordered_map<const void *, TPc> by_pc;
};
std::map<std::string, TStat> t_stats;
TStat *t_cur = nullptr;
uint32_t t_cur_gen = 0;

// the emulator's own state is not device memory: the library's data segments (threadIdx & co., option flags) and the
// fibre records
uintptr_t t_obj_lo = 0, t_obj_hi = 0;
int t_find_obj(struct dl_phdr_info *info, size_t, void *)
{
    const uintptr_t me = (uintptr_t)&t_obj_lo;
    for (int i = 0; i < info->dlpi_phnum; i++) {
        const ElfW(Phdr) &ph = info->dlpi_phdr[i];
        if (ph.p_type != PT_LOAD) continue;
        const uintptr_t lo = info->dlpi_addr + ph.p_vaddr, hi = lo + ph.p_memsz;
        if (me >= lo && me < hi) {
            for (int j = 0; j < info->dlpi_phnum; j++) {
                const ElfW(Phdr) &q = info->dlpi_phdr[j];
                if (q.p_type != PT_LOAD) continue;
                const uintptr_t l2 = info->dlpi_addr + q.p_vaddr, h2 = l2 + q.p_memsz;
                if (!t_obj_lo || l2 < t_obj_lo) t_obj_lo = l2;
                if (h2 > t_obj_hi) t_obj_hi = h2;
            }
            return 1;
        }
    }
    return 0;
}
uintptr_t t_host_lo = 0, t_host_hi = 0;      // stack of the launching thread: kernel arguments (SGPRs on the GPU)
inline bool is_emulator_state(uintptr_t a)
{
    if (!t_obj_hi) {
        dl_iterate_phdr(t_find_obj, nullptr);
        pthread_attr_t at;
        void *sa = nullptr;
        size_t sz = 0;
        if (pthread_getattr_np(pthread_self(), &at) == 0) {
            pthread_attr_getstack(&at, &sa, &sz);
            pthread_attr_destroy(&at);
            t_host_lo = (uintptr_t)sa;
            t_host_hi = t_host_lo + sz;
        }
    }
    if (a >= t_obj_lo && a < t_obj_hi) return true;
    if (a >= t_host_lo && a < t_host_hi) return true;
    return a - (uintptr_t)simt_fibers.data() < simt_fibers.size() * sizeof(SimtFiber);
}

inline void traffic(uintptr_t a, unsigned size, bool write, const void *pc)
{
    if (is_emulator_state(a)) return;
    if (t_cur_gen != (uint32_t)simt_launches || !t_cur) {
        t_cur = &t_stats[simt_kernel_name ? simt_kernel_name : "?"];
        t_cur_gen = (uint32_t)simt_launches;
    }
    const unsigned lin = simt_blockIdx.x + simt_gridDim.x * (simt_blockIdx.y + simt_gridDim.y * simt_blockIdx.z);
    const uint8_t xbit = (uint8_t)(1u << (lin & 7));
    TPc &P = t_cur->by_pc[pc];
    (write ? t_cur->req_st : t_cur->req_ld) += size;
    (write ? P.req_st : P.req_ld) += size;
    for (uintptr_t s = a >> 5; s <= (a + size - 1) >> 5; s++) {       // 32-byte sectors
        TLine &L = t_lines[s >> 2];
        if (write) {
            const uint8_t sb = (uint8_t)(1u << (s & 3));
            if (!(L.wsect & sb)) { L.wsect |= sb; t_cur->written += 32; P.written += 32; }
        } else if (!(L.touched_xcd & xbit)) {
            t_cur->fetched += 128; P.fetched += 128;
        }
        L.touched_xcd |= xbit;
    }
}

inline bool is_stack(uintptr_t a)
{
    const uintptr_t s = (uintptr_t)simt_stacks.data();
    return a - s < simt_stacks.size();
}
inline bool is_lds(uintptr_t a)
{
    if (a - (uintptr_t)simt_dyn_lds < simt_lds_bytes) return true;
    return a - (uintptr_t)__start_simt_lds < (uintptr_t)(__stop_simt_lds - __start_simt_lds);
}

inline void start_wg_if_new(uint32_t wg)
{
    if (acquired_wg != wg || acquired_gen != cur_gen) {
        acquired.clear();
        lds_released.clear();
        lds_acquired.clear();
        acquired_wg = wg;
        acquired_gen = cur_gen;
    }
}

// was access A (of another workgroup) released to the current workgroup before the access at (tid, bar, seq)?
bool ordered_by_atomics(const Acc &A, uint32_t tid, uint16_t bar, uint32_t seq)
{
    auto it = acquired.find(A.wg_tid >> 10);
    if (it == acquired.end()) return false;
    const Edge &e = it->second;
    const uint32_t atid = A.wg_tid & 1023;
    const bool before_rel = A.bar < e.rel.bar || atid == e.rel.tid || ((atid >> 6) == (e.rel.tid >> 6) && A.seq < e.rel.seq);
    const bool after_acq = bar > e.acq_bar || tid == e.acq_tid || ((tid >> 6) == (e.acq_tid >> 6) && seq > e.acq_seq);
    return before_rel && after_acq;
}

// same workgroup, different waves: was access A released through an LDS atomic that work-item `tid` (or, in front of a
// wave operation, another lane of its wave) has acquired?
bool ordered_by_lds_atomics(const Acc &A, uint32_t tid, uint32_t seq)
{
    const uint32_t atid = A.wg_tid & 1023, aw = atid >> 6;
    auto covers = [&](const EdgeW &e) {
        return A.bar < e.rel.bar || atid == e.rel.tid || A.seq < e.rel.seq;       // (A's wave is the releasing wave)
    };
    auto it = lds_acquired.find(tid);
    if (it != lds_acquired.end()) {
        auto jt = it->second.find(aw);
        if (jt != it->second.end() && covers(jt->second)) return true;
    }
    for (uint32_t l = (tid & ~63u); l < (tid & ~63u) + 64; l++) {                  // a lane of my wave, then a wave operation
        if (l == tid) continue;
        auto lt = lds_acquired.find(l);
        if (lt == lds_acquired.end()) continue;
        auto jt = lt->second.find(aw);
        if (jt != lt->second.end() && seq > jt->second.acq_seq && covers(jt->second)) return true;
    }
    return false;
}

// a store that conflicts with an earlier store: what does it write?  The hook runs BEFORE the store, so the location is
// looked at again when the next hook runs (nothing but the store lies between): the same bytes = the idiom "every
// work-item stores the same value" (flags, clamped staging stores), counted apart.
struct Pending { Report *r; uintptr_t addr; unsigned size; uint32_t old; };
std::vector<Pending> pending;
inline void flush_pending()
{
    for (const Pending &p : pending) {
        uint32_t now = 0;
        memcpy(&now, (const void *)p.addr, p.size);
        if (now == p.old) p.r->same++;
    }
    pending.clear();
}

void report(int kind, const Acc &A, const Acc &B, uintptr_t addr, bool lds, unsigned gran)
{
    n_conflict++;
    Key k{simt_kernel_name ? simt_kernel_name : "?", kind, pcs[A.pc_fl >> 2], pcs[B.pc_fl >> 2]};
    Report &r = reports[k];
    if (r.count++ == 0) {
        r.addr = addr;
        r.wg0 = A.wg_tid >> 10; r.tid0 = A.wg_tid & 1023; r.w0 = A.pc_fl & 1;
        r.wg1 = B.wg_tid >> 10; r.tid1 = B.wg_tid & 1023; r.w1 = B.pc_fl & 1;
        r.lds = lds;
    }
    if ((A.pc_fl & 1) && (B.pc_fl & 1)) {
        Pending p{&r, addr, gran, 0};
        memcpy(&p.old, (const void *)addr, gran);
        pending.push_back(p);
    }
}

void report_one(int kind, const Acc &B, uintptr_t addr)
{
    n_conflict++;
    Key k{simt_kernel_name ? simt_kernel_name : "?", kind, pcs[B.pc_fl >> 2], pcs[B.pc_fl >> 2]};
    Report &r = reports[k];
    if (r.count++ == 0) {
        r.addr = addr;
        r.wg0 = r.wg1 = B.wg_tid >> 10; r.tid0 = r.tid1 = B.wg_tid & 1023; r.w0 = r.w1 = B.pc_fl & 1;
        r.lds = true;
    }
}

inline void check(const Acc &A, const Acc &B, uintptr_t addr, bool lds, unsigned gran)
{
    if (A.gen != B.gen) return;
    const uint32_t wgA = A.wg_tid >> 10, wgB = B.wg_tid >> 10;
    if ((A.pc_fl & 2) && (B.pc_fl & 2)) return;            // both atomic
    if (wgA == wgB) {
        const uint32_t ta = A.wg_tid & 1023, tb = B.wg_tid & 1023;
        if (ta == tb || A.bar != B.bar) return;            // (B runs later in emulator time: its count is the larger)
        if ((ta >> 6) == (tb >> 6)) {
            if (A.seq != B.seq) return;
            report(1, A, B, addr, lds, gran);
        } else {
            if (ordered_by_lds_atomics(A, tb, B.seq)) return;
            report(0, A, B, addr, lds, gran);
        }
        return;
    }
    if (lds) return;                                       // another workgroup's LDS contents: stale
    if (ordered_by_atomics(A, B.wg_tid & 1023, B.bar, B.seq)) return;
    report(2, A, B, addr, lds, gran);
}

inline bool written_by_wg(const Acc &w, const Acc &cur)
{
    return w.gen == cur.gen && (w.wg_tid >> 10) == (cur.wg_tid >> 10);
}
// the word shadow and the byte shadow are separate tables: a word stored as a word and loaded as two halves (or the
// other way round) is initialised all the same
inline bool written_other_granule(uintptr_t addr, unsigned gran, const Acc &cur)
{
    if (gran == 1) return written_by_wg(sh_word.cell(addr & ~(uintptr_t)3)->w, cur);
    for (unsigned o = 0; o < 4; o++)
        if (!written_by_wg(sh_byte.cell(addr + o)->w, cur)) return false;
    return true;
}

inline void touch(Cell *c, const Acc &cur, uintptr_t addr, bool lds, unsigned gran)
{
    if (cur.pc_fl & 1) {
        check(c->w, cur, addr, lds, gran);
        check(c->r0, cur, addr, lds, gran);
        check(c->r1, cur, addr, lds, gran);
        check(c->r2, cur, addr, lds, gran);
        c->w = cur;
        c->r0.gen = c->r1.gen = c->r2.gen = 0;   // (accesses ordered after this store are ordered after those loads: a
                                                 //  load that is NOT ordered before this store has just been reported)
    } else {
        if (lds && !written_by_wg(c->w, cur) && !written_other_granule(addr, gran, cur)) report_one(3, cur, addr);
        check(c->w, cur, addr, lds, gran);
        // r0 = the latest reader; when it is replaced it moves to r1 (replaced by another lane of its wave: a wave that
        // reads a word lane by lane and then lets ONE lane store it) or to r2 (replaced by another wave / workgroup)
        if (c->r0.gen == cur.gen && c->r0.wg_tid != cur.wg_tid) {
            if ((c->r0.wg_tid >> 6) == (cur.wg_tid >> 6)) c->r1 = c->r0;
            else c->r2 = c->r0;
        }
        c->r0 = cur;
    }
}

void access(const void *p, unsigned size, bool write, bool atomic, const void *pc)
{
    SimtFiber *f = simt_cur;
    if (!pending.empty()) flush_pending();
    if (!(enabled || traffic_on) || !f || f->state != SIMT_RUN) return;    // host code, the scheduler
    const uintptr_t a = (uintptr_t)p;
    if (is_stack(a)) return;
    if (traffic_on && size && !is_lds(a) &&
        !(a - (uintptr_t)simt_dyn_lds < simt_lds_buf.size() - (size_t)(simt_dyn_lds - simt_lds_buf.data())))
        traffic(a, size, write, pc);
    if (!enabled) return;
    // the emulator's own state (the exchange buffers of the emulated cross-lane and matrix operations, threadIdx & co.,
    // the fibre records) is not device memory: the bf16 matrix instruction's operand exchange showed as INTER rows of
    // every bf16 kernel.  (The product has no __device__ globals; static __shared__ variables are recognised first.)
    if (!is_lds(a) && is_emulator_state(a)) return;
    n_access++;
    cur_gen = (uint32_t)simt_launches;
    const uint32_t wg = (uint32_t)simt_wg_serial & 0x3fffff, tid = (uint32_t)(f - simt_fibers.data());
    start_wg_if_new(wg);
    Acc cur{cur_gen, wg << 10 | tid, (uint32_t)f->seq, pc_of(pc) << 2 | (atomic ? 2u : 0u) | (write ? 1u : 0u),
            (uint16_t)f->bar};
    const bool lds = is_lds(a);
    if (!lds && a - (uintptr_t)simt_dyn_lds < simt_lds_buf.size() - (size_t)(simt_dyn_lds - simt_lds_buf.data())) {
        report_one(4, cur, a);
        return;
    }
    unsigned o = 0;
    if ((a & 3) == 0)
        for (; o + 4 <= size; o += 4) touch(sh_word.cell(a + o), cur, a + o, lds, 4);
    for (; o < size; o++) touch(sh_byte.cell(a + o), cur, a + o, lds, 1);
}

}  // namespace

extern "C" {

// kind: 1 read-modify-write (acquire + release), 2 load (acquire), 3 store (release)
__attribute__((noinline)) void simt_race_atomic(const void *p, int size, int kind, const void *pc)
{
    SimtFiber *f = simt_cur;
    if (!(enabled || traffic_on) || !f || f->state != SIMT_RUN) return;
    if (!enabled) {                                         // traffic only
        if (kind != 3) access(p, size, false, true, pc);
        if (kind != 2) access(p, size, true, true, pc);
        return;
    }
    if (kind != 3) access(p, size, false, true, pc);
    if (kind != 2) access(p, size, true, true, pc);
    const uint32_t wg = (uint32_t)simt_wg_serial & 0x3fffff, tid = (uint32_t)(f - simt_fibers.data());
    start_wg_if_new(wg);
    if (is_lds((uintptr_t)p)) {
        auto &relw = lds_released[(uintptr_t)p];
        const uint32_t w = tid >> 6;
        if (kind != 3) {
            auto &mine = lds_acquired[tid];
            for (auto &kv : relw)
                if (kv.first != w) mine[kv.first] = EdgeW{kv.second, (uint32_t)f->seq, (uint16_t)f->bar};
            // transitive: what the releasing work-items had acquired themselves travels with their release
            for (auto &kv : relw) {
                if (kv.first == w) continue;
                auto rt = lds_acquired.find(kv.second.tid);
                if (rt == lds_acquired.end() || rt->first == tid) continue;
                for (auto &e : rt->second)
                    if (e.first != w && !mine.count(e.first)) mine[e.first] = EdgeW{e.second.rel, (uint32_t)f->seq, (uint16_t)f->bar};
            }
        }
        if (kind != 2) relw[w] = RelW{tid, (uint32_t)f->seq, (uint16_t)f->bar};
        return;
    }
    auto &rel = released[(uintptr_t)p];
    if (kind != 3)
        for (auto &kv : rel) {
            if (kv.first == wg) continue;
            Edge e{kv.second, tid, (uint32_t)f->seq, (uint16_t)f->bar};
            auto it = acquired.find(kv.first);
            // (keep the edge that covers more: the later release; of equal ones the earlier acquire)
            if (it == acquired.end() || it->second.rel.bar < e.rel.bar) acquired[kv.first] = e;
        }
    if (kind != 2) {
        rel[wg] = Rel{wg, tid, (uint32_t)f->seq, (uint16_t)f->bar};
        for (auto &kv : acquired)                          // transitive: what this workgroup had acquired travels on,
            if (!rel.count(kv.first)) rel[kv.first] = kv.second.rel;   // as released where ITS releaser said
    }
}

__attribute__((noinline)) void simt_race_access(const void *p, int size, int write, const void *pc)
{
    access(p, (unsigned)size, write != 0, false, pc);
}

// a launch is over: nothing recorded in it can conflict with anything later
void simt_race_launch_end()
{
    if (traffic_on) {
        TStat &k = t_stats[simt_kernel_name ? simt_kernel_name : "?"];
        k.launches++;
        k.wgs += (long long)simt_gridDim.x * simt_gridDim.y * simt_gridDim.z;
        t_lines.clear();
        t_cur = nullptr;
    }
    flush_pending();
    released.clear();
    acquired.clear();
    lds_released.clear();
    lds_acquired.clear();
    acquired_wg = ~0u;
    if (sh_word.pages.size() + sh_byte.pages.size() > 16384) {   // ~ 0.8 GB of shadow: start afresh
        sh_word.clear();
        sh_byte.clear();
    }
}

void simt_race_enable(int on) { enabled = on != 0; }
void simt_traffic_enable(int on) { traffic_on = on != 0; }
void simt_traffic_reset()
{
    t_stats.clear();
    t_lines.clear();
    t_cur = nullptr;
}
// one line per kernel:  K <tab> kernel <tab> launches <tab> workgroups <tab> requested load / store <tab> fetched <tab> written
// and, with per_pc != 0, one line per code address of that kernel:  P <tab> kernel <tab> library <tab> offset <tab> the same four
int simt_traffic_report(char *buf, size_t n, int per_pc)
{
    std::string s;
    char line[1024];
    for (auto &kv : t_stats) {
        const TStat &k = kv.second;
        snprintf(line, sizeof line, "K\t%s\t%lld\t%lld\t%lld\t%lld\t%lld\t%lld\n", kv.first.c_str(), k.launches, k.wgs,
                 k.req_ld, k.req_st, k.fetched, k.written);
        s += line;
        if (!per_pc) continue;
        for (auto &pv : k.by_pc) {
            Dl_info i0{};
            dladdr(pv.first, &i0);
            snprintf(line, sizeof line, "P\t%s\t%s\t0x%zx\t%lld\t%lld\t%lld\t%lld\n", kv.first.c_str(),
                     i0.dli_fname ? i0.dli_fname : "?", (size_t)((const char *)pv.first - (const char *)i0.dli_fbase),
                     pv.second.req_ld, pv.second.req_st, pv.second.fetched, pv.second.written);
            s += line;
        }
    }
    if (buf && n) {
        const size_t m = std::min(n - 1, s.size());
        memcpy(buf, s.data(), m);
        buf[m] = 0;
    }
    return (int)s.size();
}
void simt_race_reset()
{
    reports.clear();
    n_access = n_conflict = 0;
}
void simt_race_counters(long long *out)
{
    out[0] = n_access;
    out[1] = n_conflict;
    out[2] = (long long)reports.size();
}

// text report: one line per distinct (kernel, kind, code pair); returns the number of lines
int simt_race_report(char *buf, size_t n)
{
    flush_pending();
    std::string s;
    char line[1024];
    for (auto &kv : reports) {
        const Key &k = kv.first;
        const Report &r = kv.second;
        Dl_info i0{}, i1{};
        dladdr(k.pc0, &i0);
        dladdr(k.pc1, &i1);
        snprintf(line, sizeof line, "%s\t%s\t%s\t0x%zx\t0x%zx\t%lld\t%s%s %s wg %u item %u -> %s wg %u item %u\n",
                 KIND[k.kind], k.kernel.c_str(), i0.dli_fname ? i0.dli_fname : "?",
                 (size_t)((const char *)k.pc0 - (const char *)i0.dli_fbase),
                 (size_t)((const char *)k.pc1 - (const char *)i1.dli_fbase), r.count, r.same == r.count ? "same-value stores: " : "", r.lds ? "lds" : "global",
                 r.w0 ? "store" : "load", r.wg0, r.tid0, r.w1 ? "store" : "load", r.wg1, r.tid1);
        s += line;
    }
    if (buf && n) {
        const size_t m = std::min(n - 1, s.size());
        memcpy(buf, s.data(), m);
        buf[m] = 0;
    }
    return (int)reports.size();
}

// ---- the instrumentation's entry points (clang -fsanitize=thread) ------------------------------------------------
#define RD(N)                                                                                                   \
    void __tsan_read##N(void *p) { access(p, N, false, false, __builtin_return_address(0)); }                   \
    void __tsan_unaligned_read##N(void *p) { access(p, N, false, false, __builtin_return_address(0)); }         \
    void __tsan_write##N(void *p) { access(p, N, true, false, __builtin_return_address(0)); }                   \
    void __tsan_unaligned_write##N(void *p) { access(p, N, true, false, __builtin_return_address(0)); }         \
    void __tsan_read_write##N(void *p)                                                                          \
    {                                                                                                           \
        access(p, N, false, false, __builtin_return_address(0));                                                \
        access(p, N, true, false, __builtin_return_address(0));                                                 \
    }                                                                                                           \
    void __tsan_unaligned_read_write##N(void *p)                                                                \
    {                                                                                                           \
        access(p, N, false, false, __builtin_return_address(0));                                                \
        access(p, N, true, false, __builtin_return_address(0));                                                 \
    }
RD(1) RD(2) RD(4) RD(8) RD(16)
void __tsan_read_range(void *p, unsigned long n) { access(p, (unsigned)n, false, false, __builtin_return_address(0)); }
void __tsan_write_range(void *p, unsigned long n) { access(p, (unsigned)n, true, false, __builtin_return_address(0)); }
// struct copies and fills the compiler lowers to the memory intrinsics (a float4 assigned as a whole)
void *__tsan_memcpy(void *d, const void *s, unsigned long n)
{
    const void *pc = __builtin_return_address(0);
    access(s, (unsigned)n, false, false, pc);
    access(d, (unsigned)n, true, false, pc);
    return memcpy(d, s, n);
}
void *__tsan_memmove(void *d, const void *s, unsigned long n)
{
    const void *pc = __builtin_return_address(0);
    access(s, (unsigned)n, false, false, pc);
    access(d, (unsigned)n, true, false, pc);
    return memmove(d, s, n);
}
void *__tsan_memset(void *d, int v, unsigned long n)
{
    access(d, (unsigned)n, true, false, __builtin_return_address(0));
    return memset(d, v, n);
}
void __tsan_init() {}
void __tsan_func_entry(void *) {}
void __tsan_func_exit() {}
void __tsan_vptr_update(void **, void *) {}
void __tsan_vptr_read(void **) {}
void __tsan_ignore_thread_begin() {}
void __tsan_ignore_thread_end() {}
}
