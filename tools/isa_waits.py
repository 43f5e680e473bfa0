"""For each kernel of an ISA listing: its inner loops that load from memory, with the s_waitcnt vmcnt(N)
values inside them and their load / MFMA counts.  vmcnt(0) in a loop that issues loads itself = nothing
stays in flight across the compute (how the hot loops of round 2 were found).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Igrid_gcn_amd/csrc \\
          -S --cuda-device-only grid_gcn_amd/csrc/gridgcn_direct.hip -o /tmp/direct.s
    python tools/isa_waits.py /tmp/direct.s | less

(Fully unrolled bodies are not loops: read those with
    grep -n "s_waitcnt vmcnt\|global_load\|v_mfma" on the kernel's slice of the listing.)"""
import re, sys
txt = open(sys.argv[1]).read().split("\n")
name = None
kern = {}
for i, l in enumerate(txt):
    m = re.match(r"^(_Z\w+):", l)
    if m: name = m.group(1); kern[name] = [i, None]
    if "s_endpgm" in l and name: kern[name][1] = i
for k, (a, b) in kern.items():
    if b is None: continue
    body = txt[a:b]
    # loop headers
    loops = [j for j, l in enumerate(body) if "Inner Loop Header" in l]
    out = []
    for j in loops:
        lab = body[j].split(":")[0]
        # loop end: the backward branch to this label
        end = None
        for e in range(j + 1, len(body)):
            if re.search(r"s_cbranch\w+ " + re.escape(lab) + r"\b", body[e]): end = e; break
        if end is None: continue
        seg = body[j:end]
        nl = sum("global_load" in x or "buffer_load" in x for x in seg)
        nm = sum("v_mfma" in x for x in seg)
        w = [re.search(r"vmcnt\((\d+)\)", x).group(1) for x in seg if "vmcnt(" in x]
        if nl:
            out.append("   loop@%d len=%d loads=%d mfma=%d vmcnt=%s" % (j, end - j, nl, nm, ",".join(w)))
    if out:
        print(k[:90]); print("\n".join(out))
