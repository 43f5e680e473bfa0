// simt_buf.cpp -- host definitions of the raw buffer intrinsics the training kernels declare by their LLVM names
// (grid_gcn_amd/csrc/gridgcn_mma.h: `... __asm("llvm.amdgcn.raw.buffer.load.v4f32")`; tests/simt/build.py drops the
// label -- LLVM treats a function of that NAME as the intrinsic on every target -- and the calls link to these).
// Raw buffer, stride 0: address = base (descriptor words 0-1) + lane offset + uniform offset; words beyond
// num_records (descriptor word 2; 0xffffffff = no limit) read 0 and are not stored -- checked per dword, as the
// hardware does, which is what the kernels' range-checked partial tiles rely on.
#include "simt_hip.h"

#include "gridgcn_mma.h"

long long simt_buf_oob = 0;      // dwords outside their descriptor's range (legitimate: partial tiles)

static inline char *simt_base(gg_rsrc r)
{
    return (char *)(((uint64_t)(uint32_t)r.y << 32 | (uint64_t)(uint32_t)r.x) & 0x0000ffffffffffffull);
}
static inline bool simt_in(gg_rsrc r, uint64_t off, unsigned size)
{
    const uint32_t n = (uint32_t)r.z;
    if (n == 0xffffffffu || off + size <= (uint64_t)n) return true;
    simt_buf_oob++;
    return false;
}
// race build (tests/simt/simt_race.cpp): this unit is compiled WITHOUT the instrumentation and announces its accesses
// itself, under the code address of the kernel's buffer instruction
#ifdef SIMT_RACE
#define SIMT_BUF_PC const void *const simt_pc = __builtin_return_address(0)
#define SIMT_BUF_RD(p, n) simt_race_access((p), (n), 0, simt_pc)
#define SIMT_BUF_WR(p, n) simt_race_access((p), (n), 1, simt_pc)
#else
#define SIMT_BUF_PC const void *const simt_pc = nullptr
#define SIMT_BUF_RD(p, n) ((void)0)
#define SIMT_BUF_WR(p, n) ((void)0)
#endif
template <class T> static inline T simt_ld(gg_rsrc r, uint64_t off, const void *simt_pc)
{
    T v = 0;
    (void)simt_pc;
    if (simt_in(r, off, sizeof(T))) {
        SIMT_BUF_RD(simt_base(r) + off, (int)sizeof(T));
        memcpy(&v, simt_base(r) + off, sizeof(T));
    }
    return v;
}

gg_f32x4 gg_buf_ld4(gg_rsrc r, unsigned v, unsigned s, int)
{
    SIMT_BUF_PC;
    gg_f32x4 o;
    const uint64_t off = (uint64_t)v + s;
    o.x = simt_ld<float>(r, off, simt_pc); o.y = simt_ld<float>(r, off + 4, simt_pc);
    o.z = simt_ld<float>(r, off + 8, simt_pc); o.w = simt_ld<float>(r, off + 12, simt_pc);
    return o;
}
gg_f32x2 gg_buf_ld2(gg_rsrc r, unsigned v, unsigned s, int)
{
    SIMT_BUF_PC;
    gg_f32x2 o;
    const uint64_t off = (uint64_t)v + s;
    o.x = simt_ld<float>(r, off, simt_pc); o.y = simt_ld<float>(r, off + 4, simt_pc);
    return o;
}
float gg_buf_ld(gg_rsrc r, unsigned v, unsigned s, int) { SIMT_BUF_PC; return simt_ld<float>(r, (uint64_t)v + s, simt_pc); }
unsigned char gg_buf_ld_u8(gg_rsrc r, unsigned v, unsigned s, int) { SIMT_BUF_PC; return simt_ld<unsigned char>(r, (uint64_t)v + s, simt_pc); }
unsigned short gg_buf_ld_u16(gg_rsrc r, unsigned v, unsigned s, int) { SIMT_BUF_PC; return simt_ld<unsigned short>(r, (uint64_t)v + s, simt_pc); }
unsigned gg_buf_ld_u32(gg_rsrc r, unsigned v, unsigned s, int) { SIMT_BUF_PC; return simt_ld<unsigned>(r, (uint64_t)v + s, simt_pc); }
void gg_buf_st(float x, gg_rsrc r, unsigned v, unsigned s, int)
{
    SIMT_BUF_PC;
    (void)simt_pc;
    const uint64_t off = (uint64_t)v + s;
    if (simt_in(r, off, 4)) {
        SIMT_BUF_WR(simt_base(r) + off, 4);
        memcpy(simt_base(r) + off, &x, 4);
    }
}

// declared by single kernels (gridgcn_attbwd_nz.hip, gridgcn_attfwd.hip)
gg_i32x4 gg_buf_ld4i(gg_rsrc r, unsigned v, unsigned s, int)
{
    SIMT_BUF_PC;
    gg_i32x4 o;
    const uint64_t off = (uint64_t)v + s;
    o.x = simt_ld<int>(r, off, simt_pc); o.y = simt_ld<int>(r, off + 4, simt_pc);
    o.z = simt_ld<int>(r, off + 8, simt_pc); o.w = simt_ld<int>(r, off + 12, simt_pc);
    return o;
}
void gg_buf_st_u8(unsigned char x, gg_rsrc r, unsigned v, unsigned s, int)
{
    SIMT_BUF_PC;
    (void)simt_pc;
    const uint64_t off = (uint64_t)v + s;
    if (simt_in(r, off, 1)) {
        SIMT_BUF_WR(simt_base(r) + off, 1);
        simt_base(r)[off] = (char)x;
    }
}
