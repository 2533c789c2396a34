#!/bin/bash
# In-solve per-kernel times of two settings of one debug option, under rocprofv3 --kernel-trace --stats (through gpurun):
#   bash tools/ab_trace.sh <tag> <workload> <storage_bits> <option> [solves]
set -o pipefail
TAG=$1; W=$2; B=$3; OPT=$4; N=${5:-8}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  SFMBA_DEBUG=$OPT=$V rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${OPT}${V} -- python3 $R/tools/solve_loop.py $W $N $B > $OUT/${TAG}_${OPT}${V}.txt 2> $OUT/${TAG}_${OPT}${V}.err
  python3 $R/tools/profile_summary.py stats $OUT/${TAG}_${OPT}${V} $OUT/${TAG}_${OPT}${V}_kernel_stats.md > /dev/null
  rm -rf $OUT/${TAG}_${OPT}${V}
done
cat $OUT/${TAG}_${OPT}0.txt $OUT/${TAG}_${OPT}1.txt
