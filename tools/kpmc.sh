#!/bin/bash
# tools/kpmc.sh <tag> <pattern> "<counters>" -- <command...>: one rocprofv3 --pmc pass, per-kernel means of the counters
tag=$1; pat=$2; ctr=$3; shift 4
export TMPDIR=/tmp
rm -rf /tmp/kp_$tag
rocprofv3 --pmc $ctr -d /tmp/kp_$tag --output-format csv -- "$@" > gpurun_out/kp_$tag.log 2>&1 || { tail -5 gpurun_out/kp_$tag.log; exit 1; }
python3 tools/pmc_summary.py /tmp/kp_$tag "$pat"
