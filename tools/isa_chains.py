"""Chains of single loads in the ISA of every kernel.

Compiles each csrc/*.hip to gfx950 assembly and reports, per kernel,
  loops   inner loops (<= 120 instructions) with one or two global / buffer loads and a full `s_waitcnt vmcnt(0)`:
          one memory round trip per iteration unless the trip count is tiny;
  sunk    places where a load is followed by a full wait and a store within six instructions: the shape a
          "batched" copy takes when the compiler sinks each load into the bounds check around its store.
    python tools/isa_chains.py [file.hip ...]
"""
import collections
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "grid_gcn_amd", "csrc")
files = [os.path.join(CSRC, f) for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
LOAD = re.compile(r"\b(global_load|buffer_load|flat_load)")
MEM = re.compile(r"global_load|buffer_load|flat_load|s_waitcnt vmcnt|ds_write|global_store|buffer_store")
loops, sunk = collections.Counter(), collections.Counter()
for f in files:
    name = os.path.basename(f)
    if name == "gridgcn_capi.hip":
        continue
    asm = "/tmp/isa_chains_%s.s" % name
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                    "-S", "--cuda-device-only", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), f, "-o", asm],
                   capture_output=True)
    if not os.path.exists(asm):
        continue
    lines = [l for l in open(asm).read().splitlines() if not l.strip().startswith(";")]
    func, labels = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func, labels = m.group(1), {}
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
        m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i]
            inner = not any(re.match(r"^\.LBB", b) for b in body[1:])
            nl = sum(1 for b in body if LOAD.search(b))
            if inner and len(body) <= 120 and 1 <= nl <= 2 and any("s_waitcnt vmcnt(0)" in b for b in body):
                loops[(name, func)] += 1
        if LOAD.search(l):
            seq = [x for x in lines[i + 1:i + 7] if MEM.search(x)]
            if len(seq) >= 2 and "s_waitcnt vmcnt(0)" in seq[0] and re.search(r"ds_write|_store", seq[1]):
                sunk[(name, func)] += 1
for title, cnt in (("loops", loops), ("sunk", sunk)):
    print("---- %s" % title)
    for (name, func), c in cnt.most_common():
        d = subprocess.run(["c++filt", func], capture_output=True, text=True).stdout.strip()
        print("%3d %-24s %s" % (c, name, re.sub(r"\(.*", "", d)[:90]))
