#!/bin/bash
# device-side camera-major sort kernels under the tracer: bash tools/sort_time.sh <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=$1
cd /tmp && export TMPDIR=/tmp
for W in cfg4 cfg5; do
  B=64; [ $W = cfg5 ] && B=32
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sort_$W -- python3 $R/tools/solve_loop.py $W 2 $B > /dev/null 2> $OUT/${TAG}_sort_$W.err
  python3 $R/tools/profile_summary.py stats $OUT/${TAG}_sort_$W $OUT/${TAG}_sort_${W}_kernel_stats.md > /dev/null
  rm -rf $OUT/${TAG}_sort_$W
  grep "k_cam_hist\|k_cam_offsets\|k_cam_scatter\|k_unpack\|k_expand\|k_xcd" $OUT/${TAG}_sort_${W}_kernel_stats.md | cut -d'|' -f2,3,5 | cut -c1-40,90-
done
