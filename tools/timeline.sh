#!/bin/bash
# Launch-by-launch timeline of the last of a few back-to-back solves (through gpurun):  bash tools/timeline.sh <tag> <workload> [bits]
set -o pipefail
TAG=$1; W=${2:-cfg4}; B=${3:-64}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_tl -- python3 $R/tools/solve_loop.py $W 8 $B > $OUT/${TAG}_tl.txt 2> $OUT/${TAG}_tl.err
python3 $R/tools/profile_summary.py timeline $OUT/${TAG}_tl $OUT/${TAG}_timeline.txt > /dev/null
rm -rf $OUT/${TAG}_tl
head -5 $OUT/${TAG}_timeline.txt; cd $R
