#!/bin/bash
# one GPU session: parity of the index operators, then timings (new split build vs legacy build)
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r2a
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/parity.log 2>&1
echo "parity rc=$?"; tail -15 $OUT/parity.log
for cfg in seg80k synth200k seg8k cls; do
  B=8; [ $cfg = seg8k ] && B=16; [ $cfg = cls ] && B=32
  echo "== $cfg new"; timeout 300 python tools/prof_index.py --cfg $cfg --B $B --iters 50 2>&1 | tee $OUT/time_${cfg}_new.log
  echo "== $cfg legacy"; GG_INDEX_LEGACY=1 timeout 300 python tools/prof_index.py --cfg $cfg --B $B --iters 50 2>&1 | tee $OUT/time_${cfg}_legacy.log
done
cd /tmp && export TMPDIR=/tmp
for cfg in seg80k synth200k; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_$cfg -o p -- python $GRAFT_REPO_ROOT/tools/prof_index.py --cfg $cfg --iters 20 > $GRAFT_REPO_ROOT/$OUT/prof_$cfg.log 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$OUT/prof_$cfg/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -z "$f" ] && f=$(find $GRAFT_REPO_ROOT/$OUT/prof_$cfg -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats $cfg ($f)"; head -20 "$f" | cut -c1-160
done
