#!/bin/bash
# Data-race detector over the product's kernels on the CPU build:  tests/simt/race.sh [report file] [pytest arguments]
#   default: every emulator suite (index operators, attention kernels, the product's host code with the long cases:
#   GG_SIMT_FULL=1) -> profiles/r6_race_report.txt, and which kernels those suites launched ->
#   profiles/r6_simt_kernel_coverage.txt.  See tests/simt/simt_race.cpp for what is and is not a report.
cd "$(dirname "$0")/../.."
OUT=${1:-profiles/r6_race_report.txt}; shift
COV=${OUT%/*}/r6_simt_kernel_coverage.txt
python tests/simt/build.py --race > /dev/null || exit 1
export GG_SIMT_RACE=1 GG_SIMT_FULL=${GG_SIMT_FULL:-1} GG_SIMT_COVERAGE=/tmp/simt_cov_raw.txt
FINAL=$OUT; OUT=/tmp/simt_race_report.$$.txt      # (an hour of runs: the report appears under profiles/ when it is whole)
: > "$OUT"; : > $GG_SIMT_COVERAGE
for t in ${@:-tests/test_simt_index.py tests/test_simt_train.py tests/test_simt_product.py}; do
  GG_SIMT_RACE_REPORT=/tmp/simt_race_part.txt python -m pytest -q -p no:cacheprovider "$t" | tail -1 > /tmp/simt_race_pytest.txt
  { echo "== $t: $(cat /tmp/simt_race_pytest.txt)"; cat /tmp/simt_race_part.txt; } >> "$OUT"
done
python - "$GG_SIMT_COVERAGE" > "$COV" <<'PY'
import collections, glob, re, sys
cov = collections.Counter()
for ln in open(sys.argv[1]):
    k, n = ln.rstrip("\n").split("\t")
    cov[re.sub(r"<.*", "", k).strip()] += int(n)
allk = {}
for f in sorted(glob.glob("grid_gcn_amd/csrc/*.hip")):
    for m in re.finditer(r"__global__[^;{]*?\bvoid\s+(gg_k_\w+)\s*\(", open(f).read(), re.S):
        allk.setdefault(m.group(1), f.split("/")[-1])
miss = [k for k in allk if k not in cov]
print("# kernels of grid_gcn_amd/csrc launched by the emulator suites (tests/simt/race.sh): %d of %d" % (len(allk) - len(miss), len(allk)))
for k in sorted(allk, key=lambda k: (allk[k], k)):
    print("%-26s %-34s %d" % (allk[k], k, cov.get(k, 0)))
PY
mv "$OUT" "$FINAL"
cat "$FINAL"; head -1 "$COV"
