"""gfx950 assembly of a csrc/*.hip file, per kernel (no GPU needed).

    from tools import isa;  k = isa.kernels("gridgcn_direct.hip");  body = k["gg_k_dw_reduce_direct"]
    python tools/isa.py gridgcn_direct.hip [substring]      # kernel names (+ instruction mix of the matches)

The compile (-O3, the library's flags, device only) is cached under /tmp/isa by source + header mtimes.
Used by tests/test_isa_handoff.py, tools/isa_chains.py-style audits and by hand when a kernel is tuned."""
import collections
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "grid_gcn_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only"]


def asm_path(src, extra=()):
    path = src if os.path.isabs(src) else os.path.join(CSRC, src)
    deps = [path] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    stamp = hashlib.sha1(("|".join("%s:%d" % (d, os.stat(d).st_mtime_ns) for d in sorted(deps)) +
                          "|".join(extra)).encode()).hexdigest()[:16]
    os.makedirs("/tmp/isa", exist_ok=True)
    out = "/tmp/isa/%s.%s.s" % (os.path.basename(path), stamp)
    if not os.path.exists(out):
        r = subprocess.run([HIPCC] + FLAGS + list(extra) + ["-I" + CSRC, "-I" + os.path.join(ROOT, "include"), path,
                            "-o", out + ".tmp"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-3000:])
        os.replace(out + ".tmp", out)
    return out


def kernels(src, extra=()):
    """{demangled kernel name without its argument list: [instruction lines]}"""
    lines = [l for l in open(asm_path(src, extra)).read().splitlines() if not l.strip().startswith(";")]
    heads = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+|gg_\w+):", l)] if m]
    names = subprocess.run(["c++filt"] + [h for _, h in heads], capture_output=True, text=True).stdout.splitlines()
    out = {}
    for (i, _), n, nxt in zip(heads, names, [h[0] for h in heads[1:]] + [len(lines)]):
        body = []
        for l in lines[i + 1:nxt]:
            if l.startswith("\t.section") or l.startswith(".Lfunc_end"):
                break
            body.append(l)
        out[re.sub(r"^void ", "", re.sub(r"\(.*", "", n))] = body
    return out


def mix(body):
    c = collections.Counter()
    for l in body:
        t = l.strip().split()
        if not t or t[0].endswith(":") or t[0].startswith("."):
            continue
        op = t[0]
        key = ("mfma" if "mfma" in op else "vmem_ld" if re.match(r"(global|buffer|flat)_load", op) else
               "vmem_st" if re.match(r"(global|buffer|flat)_store", op) else "atomic" if "atomic" in op else
               "lds" if op.startswith("ds_") else "readlane" if "readlane" in op or "writelane" in op else
               "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "other")
        c[key] += 1
    return dict(c)


if __name__ == "__main__":
    ks = kernels(sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else None
    for n, b in ks.items():
        if sub is None:
            print(n)
        elif sub in n:
            print(n, len(b), mix(b))
