#!/bin/bash
# issue / wait counters of the backward GEMM kernels (tools/time_dw.py shapes)
R=$PWD; O=$R/gpurun_out/pmc_dw; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/a -o p -- python $R/tools/time_dw.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -o p -- python $R/tools/time_dw.py > $O/b.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for d in ("a", "b"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:46], r["Counter_Name"])
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
        for (k, c), v in sorted(agg.items()):
            if "gg_k_linear_d" in k or "att_bwd" in k:
                print("%-48s %-26s n=%2d avg=%14.0f" % (k, c, v[0], v[1] / v[0]))
PY
tail -3 $O/a.log $O/b.log | grep -i "error\|fail" | head
