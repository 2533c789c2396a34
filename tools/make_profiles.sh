#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ on the GPU box:  bash tools/make_profiles.sh r02   (through gpurun)
# Writes gpurun_out/<tag>_*: the default bench line, the same command under --kernel-trace --stats, the two PMC passes
# (FETCH_SIZE, WRITE_SIZE -> <tag>_traffic.json), cfg2 under the tracer, the cfg3 / cfg5 bench lines, the eighth-size
# shard under the tracer, back-to-back kernel timings and solve loops.  Raw CSV directories are deleted at the end;
# the summaries are copied to profiles/ by hand afterwards.
set -o pipefail
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --no-per-call > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-per-call --steps 10 --warmup 5 --settle 0.1 > $OUT/${TAG}_pmc_fetch.json 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-per-call --steps 10 --warmup 5 --settle 0.1 > $OUT/${TAG}_pmc_write.json 2> $OUT/${TAG}_pmc_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_cfg2 -- python3 $R/bench.py --workload cfg2 --no-cpu-baseline --no-per-call > $OUT/${TAG}_bench_cfg2.json 2> $OUT/${TAG}_stats_cfg2.err
cd $R
python3 bench.py --workload cfg3 --no-cpu-baseline --no-per-call > $OUT/${TAG}_bench_cfg3.json 2>> $OUT/${TAG}_bench.err
python3 bench.py --workload cfg5 --storage-bits 32 --no-cpu-baseline --no-per-call --steps 10 > $OUT/${TAG}_bench_cfg5_f32.json 2>> $OUT/${TAG}_bench.err
python3 bench.py --workload cfg5 --no-cpu-baseline --no-per-call --steps 10 > $OUT/${TAG}_bench_cfg5_f64.json 2>> $OUT/${TAG}_bench.err
python3 tools/profile_summary.py stats $OUT/${TAG}_stats $OUT/${TAG}_bench_kernel_stats.md > /dev/null
python3 tools/profile_summary.py stats $OUT/${TAG}_stats_cfg2 $OUT/${TAG}_bench_cfg2_kernel_stats.md > /dev/null
python3 tools/profile_summary.py pmc $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_traffic.json "cfg4 (1000 cameras / 100k points / 1M observations), fp64" > /dev/null
for f in bench_line bench_under_rocprof bench_cfg2 bench_cfg3 bench_cfg5_f32 bench_cfg5_f64; do echo "== $f"; head -c 600 $OUT/${TAG}_$f.json; echo; done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_eighth -- python3 $R/tools/solve_loop.py 1000,12500,125000 > $OUT/${TAG}_eighth.log 2>&1
cd $R
python3 tools/profile_summary.py stats $OUT/${TAG}_eighth $OUT/${TAG}_eighth_kernel_stats.md > /dev/null
python3 tools/time_kernels.py cfg4 > $OUT/${TAG}_time_kernels.txt 2>&1
for c in cfg4 cfg2 cfg3 1000,12500,125000; do python3 tools/solve_loop.py $c; done > $OUT/${TAG}_solve_loop.txt 2>&1
python3 tools/solve_loop.py cfg5 6 32 >> $OUT/${TAG}_solve_loop.txt 2>&1
python3 tools/solve_loop.py cfg5 6 64 >> $OUT/${TAG}_solve_loop.txt 2>&1
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_stats_cfg2 $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_eighth
cat $OUT/${TAG}_time_kernels.txt $OUT/${TAG}_solve_loop.txt
