"""Build the HOST emulation of the index kernels: tests/simt/_build/libsimt_index.so.

The kernel sources are taken from grid_gcn_amd/csrc AS THEY ARE, passed through three textual rewrites that only
concern launch syntax and GPU-only spellings (what they compute is untouched), and compiled for the HOST (clang++, x86) against
tests/simt/simt_hip.h:

    k<<<grid, block, lds, stream>>>(args);            ->  simt_launch(grid, block, lds, [&]() { k(args); }, &first arg);
    extern __shared__ [attrs] T name[];               ->  T *name = (T *)simt_dyn_lds;
    asm volatile("s_waitcnt ...")                     ->  simt_waitcnt(): a rendezvous of the wave -- the instruction is
                                                          executed by the WAVE, so every lane's earlier memory
                                                          operations have been issued when it returns (the last-
                                                          arriver hand-offs drain a neighbour lane's atomic with it)
    asm volatile("s_waitcnt ...\n\ts_barrier")        ->  __syncthreads()
    asm volatile("" : "+s"(x)) and the like           ->  (nothing: register-class pins, scheduling fences)
    __attribute__((amdgpu_...(..)))                   ->  (nothing)

    python tests/simt/build.py [--force]
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "grid_gcn_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libgridgcn_simt.so")
def _product_sources():
    """every kernel source of the product's own build (grid_gcn_amd/build.py: SOURCES)"""
    sys.path.insert(0, ROOT)
    from grid_gcn_amd import build as product_build
    return list(product_build.SOURCES)


SOURCES = _product_sources()
CXX = "/opt/rocm/lib/llvm/bin/clang++"      # host compile: the kernels use clang's ext_vector_type / elementwise builtins


def _balanced(s, i, open_ch, close_ch):
    """index just past the bracket that closes the one at s[i]"""
    assert s[i] == open_ch, (s[i - 20:i + 20])
    depth = 0
    while True:
        c = s[i]
        if c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1


def rewrite_launches(s):
    out, pos = [], 0
    while True:
        i = s.find("<<<", pos)
        if i < 0:
            out.append(s[pos:])
            return "".join(out)
        # kernel expression: identifier with an optional <template arguments> directly in front of <<<
        j = i
        if s[j - 1] == ">":
            depth, j = 0, j - 1
            while True:
                if s[j] == ">":
                    depth += 1
                elif s[j] == "<":
                    depth -= 1
                    if depth == 0:
                        break
                j -= 1
        k = j
        while k > 0 and (s[k - 1].isalnum() or s[k - 1] == "_"):
            k -= 1
        kern = s[k:i]
        e = s.index(">>>", i)
        cfg = s[i + 3:e]
        # split the launch configuration at top-level commas
        parts, depth, cur = [], 0, ""
        for c in cfg:
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            if c == "," and depth == 0:
                parts.append(cur.strip())
                cur = ""
            else:
                cur += c
        parts.append(cur.strip())
        grid, block = parts[0], parts[1]
        lds = parts[2] if len(parts) > 2 else "0"
        a0 = e + 3
        while s[a0].isspace():
            a0 += 1
        a1 = _balanced(s, a0, "(", ")")
        args = s[a0:a1]
        out.append(s[pos:k])
        out.append("simt_launch(%s, %s, %s, [&]() { %s%s; }, simt_first_arg%s, \"%s\")" % (
            grid, block, lds, kern, args, args, kern.replace('"', "")))
        pos = a1


def rewrite(text):
    text = re.sub(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([\w ]+?)\s+(\w+)\[\];",
                  lambda m: "%s *%s = (%s *)simt_dyn_lds;" % (m.group(1), m.group(2), m.group(1)), text)
    text = re.sub(r'asm volatile\("s_waitcnt [^"]*\\n\\ts_barrier"[^;]*;', "__syncthreads();", text)
    text = re.sub(r'asm volatile\("s_waitcnt [^"]*"[^;]*;', "simt_waitcnt();", text)
    text = re.sub(r"__attribute__\(\(amdgpu_\w+\([^)]*\)\)\)", "", text)
    # empty asm statements that only pin a value to a GPU register class ("+s" / "v": scheduling fences)
    text = re.sub(r'asm volatile\(""\s*:[^;]*\);', ";", text)
    return rewrite_launches(rewrite_header(text))


def rewrite_header(text):
    """the raw buffer intrinsics are declared by their LLVM names (`... __asm("llvm.amdgcn.raw.buffer.load.f32")`):
    the label goes, the emulator defines the functions under their C++ names (tests/simt/simt_buf.cpp)"""
    return re.sub(r'\s*__asm\("llvm\.amdgcn\.raw\.buffer\.[^"]+"\)', "", text)


def _stamp():
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h") or f in SOURCES:
            h.update(open(os.path.join(CSRC, f), "rb").read())
    for f in ("simt_hip.h", "driver.cpp", "simt_buf.cpp", "simt_race.cpp", "build.py"):
        h.update(open(os.path.join(HERE, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "gridgcn.h"), "rb").read())
    return h.hexdigest()


RACE_FLAGS = ["-DSIMT_RACE=1", "-fsanitize=thread", "-mllvm", "-tsan-instrument-func-entry-exit=0", "-mllvm",
              "-tsan-instrument-atomics=0"]


UBSAN_CHECKS = ("signed-integer-overflow,shift,bounds,alignment,integer-divide-by-zero,float-cast-overflow,null,"
                "vla-bound,unreachable,return,pointer-overflow,bool")


def _ubsan_runtime():
    import glob
    return sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so"))[0]


def build(force=False, verbose=False, asan=False, race=False, ubsan=False):
    """one translation unit per kernel source, as the product's own build (grid_gcn_amd/build.py), plus the driver.
    asan=True: libgridgcn_simt_asan.so, every unit with -fsanitize=address (run the python process with the runtime
    preloaded: tests/simt/asan.sh)"""
    import concurrent.futures
    global OUT, LIB
    if asan:
        saved = (OUT, LIB)
        OUT = os.path.join(HERE, "_build", "asan")
        LIB = os.path.join(OUT, "libgridgcn_simt_asan.so")
        try:
            return _build(force, verbose, ["-fsanitize=address", "-fno-omit-frame-pointer", "-shared-libasan"])
        finally:
            OUT, LIB = saved
    if ubsan:
        # UndefinedBehaviorSanitizer over the kernel units (tests/simt/ubsan.sh): index arithmetic that overflows, shifts
        # out of range, register arrays indexed past their end, misaligned vector accesses, float -> int conversions
        # out of range; the runtime (a shared library of the toolchain) is linked in, reports go to stderr / log_path
        saved = (OUT, LIB)
        OUT = os.path.join(HERE, "_build", "ubsan")
        LIB = os.path.join(OUT, "libgridgcn_simt_ubsan.so")
        try:
            return _build(force, verbose, ["-fsanitize=" + UBSAN_CHECKS, "-fno-sanitize-recover=unreachable,return"],
                          link_extra=[_ubsan_runtime(), "-Wl,-rpath," + os.path.dirname(_ubsan_runtime())])
        finally:
            OUT, LIB = saved
    if race:
        # clang's ThreadSanitizer INSTRUMENTATION of the kernel units, tests/simt/simt_race.cpp as its runtime
        saved = (OUT, LIB)
        OUT = os.path.join(HERE, "_build", "race")
        LIB = os.path.join(OUT, "libgridgcn_simt_race.so")
        try:
            return _build(force, verbose, RACE_FLAGS, race=True)
        finally:
            OUT, LIB = saved
    return _build(force, verbose, [])


def _build(force, verbose, extra, race=False, link_extra=None):
    import concurrent.futures
    os.makedirs(OUT, exist_ok=True)
    stamp_file = os.path.join(OUT, "stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    flags = ["-std=c++17", "-O1", "-g", "-fPIC", "-ffp-contract=off", "-fno-strict-aliasing", "-Wno-attributes",
             "-Wno-unused-variable", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-int-to-pointer-cast",
             "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-Wno-pass-failed", "-Wno-unused-value",
             "-include", os.path.join(HERE, "simt_hip.h"), "-D__HIPCC__=1", "-DGG_SIMT=1",
             "-I" + HERE, "-I" + OUT, "-I" + CSRC, "-I" + os.path.join(ROOT, "include")] + extra
    units = []
    for h in sorted(os.listdir(CSRC)):          # headers: found in _build first (-I order)
        if h.endswith(".h"):
            with open(os.path.join(OUT, h), "w") as f:
                f.write(rewrite_header(open(os.path.join(CSRC, h)).read()))
    for src in SOURCES:
        text = rewrite(open(os.path.join(CSRC, src)).read())
        cpp = os.path.join(OUT, src.replace(".hip", ".simt.cpp"))
        with open(cpp, "w") as f:
            f.write("// generated by tests/simt/build.py from grid_gcn_amd/csrc/%s -- do not edit\n" % src)
            f.write(text)
        units.append(cpp)
    units += [os.path.join(HERE, "driver.cpp"), os.path.join(HERE, "simt_buf.cpp")]
    plain = set()         # units of the race build that are NOT instrumented: the detector and the buffer intrinsics
    if race:
        plain = {os.path.join(HERE, "simt_buf.cpp"), os.path.join(HERE, "simt_race.cpp")}
        units.append(os.path.join(HERE, "simt_race.cpp"))

    def compile_one(cpp):
        obj = os.path.join(OUT, os.path.basename(cpp) + ".o")
        fl = [f for f in flags if not (cpp in plain and (f.startswith("-fsanitize") or f.startswith("-tsan") or
                                                          f == "-mllvm"))]
        cmd = [CXX] + fl + ["-c", cpp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("simt build failed (%s):\n%s" % (os.path.basename(cpp), r.stderr[-6000:]))
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, units))
    cmd = [CXX, "-shared", "-fPIC"] + ([] if (race or link_extra) else extra) + objs + (["-ldl"] if race else []) + \
        (link_extra or []) + ["-o", LIB + ".tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("simt link failed:\n" + r.stderr[-6000:])
    os.replace(LIB + ".tmp", LIB)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB


def build_selftest():
    """tests/simt/race_selftest.hip through the same rewrites and the same instrumentation -> a library of its own
    with the detector linked in (tests/test_simt_race.py)"""
    out = os.path.join(HERE, "_build", "race_selftest")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "librace_selftest.so")
    srcs = [os.path.join(HERE, f) for f in ("race_selftest.hip", "simt_race.cpp", "simt_hip.h", "build.py")]
    if os.path.exists(lib) and all(os.path.getmtime(lib) > os.path.getmtime(f) for f in srcs):
        return lib
    cpp = os.path.join(out, "race_selftest.simt.cpp")
    with open(cpp, "w") as f:
        f.write("// generated by tests/simt/build.py from tests/simt/race_selftest.hip -- do not edit\n")
        f.write(rewrite(open(srcs[0]).read()))
    base = ["-std=c++17", "-O1", "-g", "-fPIC", "-ffp-contract=off", "-Wno-unused-value", "-Wno-unknown-attributes",
            "-include", os.path.join(HERE, "simt_hip.h"), "-DGG_SIMT=1", "-DSIMT_RACE=1", "-I" + HERE]
    for unit, extra in ((cpp, RACE_FLAGS), (srcs[1], ["-DSIMT_RACE=1"])):
        r = subprocess.run([CXX] + base + extra + ["-c", unit, "-o", unit + ".o" if unit == cpp else
                            os.path.join(out, "simt_race.o")], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("race selftest build failed:\n" + r.stderr[-4000:])
    r = subprocess.run([CXX, "-shared", "-fPIC", cpp + ".o", os.path.join(out, "simt_race.o"), "-ldl", "-o", lib],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("race selftest link failed:\n" + r.stderr[-4000:])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, asan="--asan" in sys.argv, race="--race" in sys.argv,
                ubsan="--ubsan" in sys.argv))
