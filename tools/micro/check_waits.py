"""s_waitcnt vmcnt(N) values inside the loops of tools/micro/mfma_mem.hip's register-ring kernels (modes A, C):
the ring is in flight only if the waits sit at N = loads issued since (DS - 1 steps' worth), not at 0.
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/micro/mfma_mem.hip -o /tmp/mm.s
    python tools/micro/check_waits.py /tmp/mm.s"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read().split("\n")
for name in [l.split(":")[0] for l in t if re.match(r"^_Z3k_[ac]\w+:", l)]:
    st = [i for i, l in enumerate(t) if l.startswith(name + ":")][0]
    en = [i for i in range(st, len(t)) if "s_endpgm" in t[i]][0]
    body, res = t[st:en], []
    for j, l in enumerate(body):
        if "Inner Loop Header" in l:
            lab = l.split(":")[0]
            end = [e for e in range(j + 1, len(body)) if re.search(r"s_cbranch\w+ " + re.escape(lab) + r"\b", body[e])]
            if not end:
                continue
            seg = body[j:end[0]]
            nl, nm = sum("global_load" in x for x in seg), sum("v_mfma" in x for x in seg)
            w = [int(re.search(r"vmcnt\((\d+)\)", x).group(1)) for x in seg if "vmcnt(" in x]
            if nl:
                res.append("loads=%d mfma=%d waits=%d min=%d top=%s" % (nl, nm, len(w), min(w), sorted(set(w))[-3:]))
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:24]
    print(dem, "|", " || ".join(res), "| SPILLS (cell reported n/a)" if any("scratch_" in l for l in body) else "")
