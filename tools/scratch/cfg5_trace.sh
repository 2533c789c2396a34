set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=r04
cd /tmp && export TMPDIR=/tmp
SFMBA_HIP_RUNTIME=system rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_cfg5 -- python3 $R/tools/solve_loop.py cfg5 6 32 > $OUT/${TAG}_cfg5_f32_solve_loop.txt 2> $OUT/${TAG}_stats_cfg5.err
python3 $R/tools/profile_summary.py stats $OUT/${TAG}_stats_cfg5 $OUT/${TAG}_cfg5_f32_kernel_stats.md > /dev/null
rm -rf $OUT/${TAG}_stats_cfg5
cut -c1-64,100-200 $OUT/${TAG}_cfg5_f32_kernel_stats.md | head -8; cat $OUT/${TAG}_cfg5_f32_solve_loop.txt
