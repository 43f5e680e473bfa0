#!/bin/bash
# Data-race detector over the product's kernels on the CPU build:  tests/simt/race.sh [report file] [pytest arguments]
#   default: every emulator suite (index operators, attention kernels, the product's host code with the long cases:
#   GG_SIMT_FULL=1) -> profiles/r6_race_report.txt.  See tests/simt/simt_race.cpp for what is and is not a report.
cd "$(dirname "$0")/../.."
OUT=${1:-profiles/r6_race_report.txt}; shift
python tests/simt/build.py --race > /dev/null || exit 1
export GG_SIMT_RACE=1 GG_SIMT_FULL=${GG_SIMT_FULL:-1}
: > "$OUT"
for t in ${@:-tests/test_simt_index.py tests/test_simt_train.py tests/test_simt_product.py}; do
  GG_SIMT_RACE_REPORT=/tmp/simt_race_part.txt python -m pytest -q -p no:cacheprovider "$t" | tail -1 > /tmp/simt_race_pytest.txt
  { echo "== $t: $(cat /tmp/simt_race_pytest.txt)"; cat /tmp/simt_race_part.txt; } >> "$OUT"
done
cat "$OUT"
