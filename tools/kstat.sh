#!/bin/bash
# tools/kstat.sh <tag> <pattern> -- <python command...>: rocprofv3 kernel-trace stats of a command, rows matching pattern
tag=$1; pat=$2; shift 3
export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag --output-format csv -- "$@" > gpurun_out/ks_$tag.log 2>&1 || { tail -5 gpurun_out/ks_$tag.log; exit 1; }
f=$(find /tmp/ks_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/ks_${tag}_kernel_stats.csv
python3 - "$f" "$pat" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Name']:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
tail -1 gpurun_out/ks_$tag.log
