#!/bin/bash
# UndefinedBehaviorSanitizer over the product's kernels on the CPU build:  tests/simt/ubsan.sh [report] [pytest arguments]
#   default: every emulator suite -> profiles/r6_ubsan_report.txt (one line per source location; empty = nothing found)
cd "$(dirname "$0")/../.."
OUT=${1:-profiles/r6_ubsan_report.txt}; shift
python tests/simt/build.py --ubsan > /dev/null || exit 1
rm -f /tmp/simt_ubsan.log.*
export GG_SIMT_UBSAN=1 UBSAN_OPTIONS=print_stacktrace=0:log_path=/tmp/simt_ubsan.log
: > "$OUT"
for t in ${@:-tests/test_simt_index.py tests/test_simt_train.py tests/test_simt_product.py}; do
  echo "== $t: $(python -m pytest -q -p no:cacheprovider "$t" | tail -1)" >> "$OUT"
done
cat /tmp/simt_ubsan.log.* 2>/dev/null | grep "runtime error" | sed -e 's#.*/_build/ubsan/##' -e 's#\.simt\.cpp#.hip(+1)#' | sort | uniq -c | sort -rn >> "$OUT"
cat "$OUT"
