#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ on the GPU box:  bash tools/make_profiles.sh r04   (through gpurun)
# Writes gpurun_out/<tag>_*: the default bench line, the same command under --kernel-trace --stats, the two PMC passes
# (FETCH_SIZE, WRITE_SIZE -> <tag>_traffic.json), cfg2 under the tracer, the cfg3 / cfg5 bench lines, cfg5 (fp32 storage)
# under the tracer, SQ / TA counters of the residual+Jacobian kernel at cfg4 and cfg5, the per-rank critical path of an
# 8-way sharding (tools/shard_profiles.sh), back-to-back kernel timings, solve loops and the grid-barrier probe.  Raw CSV
# directories are deleted at the end; the summaries are copied to profiles/ by hand afterwards.
set -o pipefail
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py 2> $OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --no-per-call 2> $OUT/${TAG}_stats.err | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-per-call --steps 10 --warmup 5 --settle 0.1 > $OUT/${TAG}_pmc_fetch.json 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-per-call --steps 10 --warmup 5 --settle 0.1 > $OUT/${TAG}_pmc_write.json 2> $OUT/${TAG}_pmc_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_cfg2 -- python3 $R/bench.py --workload cfg2 --no-cpu-baseline --no-per-call 2> $OUT/${TAG}_stats_cfg2.err | tail -1 > $OUT/${TAG}_bench_cfg2.json
# (SFMBA_HIP_RUNTIME=system: this tool does not import torch, and with the wheel's ROCm 7.0 runtime under the 7.2 tracer the
# 24 MB copies of the trial point to its host mirror run as blit kernels beside K1 instead of on the SDMA engines -- K1 then
# reads 750 us in the trace; untraced, and in bench.py with either runtime, they do not)
SFMBA_HIP_RUNTIME=system rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_cfg5 -- python3 $R/tools/solve_loop.py cfg5 6 32 > $OUT/${TAG}_cfg5_f32_solve_loop.txt 2> $OUT/${TAG}_stats_cfg5.err
# what gates the residual+Jacobian kernel: SQ / TA counters, two passes per size, the kernel launched back to back
for W in cfg4 cfg5; do
  B=64; [ $W = cfg5 ] && B=32
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/${TAG}_k1c_a_$W -- python3 $R/tools/time_kernels.py $W 0 $B > /dev/null 2>> $OUT/${TAG}_k1c.err
  rocprofv3 --pmc TA_BUSY_avr GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $OUT/${TAG}_k1c_b_$W -- python3 $R/tools/time_kernels.py $W 0 $B > /dev/null 2>> $OUT/${TAG}_k1c.err
done
cd $R
python3 bench.py --workload cfg3 --no-cpu-baseline --no-per-call 2>> $OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench_cfg3.json
python3 bench.py --workload cfg5 --storage-bits 32 --no-cpu-baseline --no-per-call --steps 10 2>> $OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench_cfg5_f32.json
python3 bench.py --workload cfg5 --no-cpu-baseline --no-per-call --steps 10 2>> $OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench_cfg5_f64.json
python3 tools/profile_summary.py stats $OUT/${TAG}_stats $OUT/${TAG}_bench_kernel_stats.md > /dev/null
python3 tools/profile_summary.py stats $OUT/${TAG}_stats_cfg2 $OUT/${TAG}_bench_cfg2_kernel_stats.md > /dev/null
python3 tools/profile_summary.py stats $OUT/${TAG}_stats_cfg5 $OUT/${TAG}_cfg5_f32_kernel_stats.md > /dev/null
python3 tools/profile_summary.py pmc $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_traffic.json "cfg4 (1000 cameras / 100k points / 1M observations), fp64" > /dev/null
python3 tools/profile_summary.py counters $OUT/${TAG}_k1_counters_cfg4.json "k_resjac back to back, cfg4 fp64" $OUT/${TAG}_k1c_a_cfg4 $OUT/${TAG}_k1c_b_cfg4 > /dev/null
python3 tools/profile_summary.py counters $OUT/${TAG}_k1_counters_cfg5_f32.json "k_resjac back to back, cfg5 fp32 storage" $OUT/${TAG}_k1c_a_cfg5 $OUT/${TAG}_k1c_b_cfg5 > /dev/null
for f in bench_line bench_under_rocprof bench_cfg2 bench_cfg3 bench_cfg5_f32 bench_cfg5_f64; do echo "== $f"; head -c 400 $OUT/${TAG}_$f.json; echo; done
bash tools/shard_profiles.sh $TAG > $OUT/${TAG}_shard.log 2>&1
python3 tools/time_kernels.py cfg4 > $OUT/${TAG}_time_kernels.txt 2>&1
python3 tools/time_kernels.py cfg5 0,7,1,2,4,5,8 32 >> $OUT/${TAG}_time_kernels.txt 2>&1
for c in cfg4 cfg2 cfg3 1000,12500,125000; do python3 tools/solve_loop.py $c; done > $OUT/${TAG}_solve_loop.txt 2>&1
python3 tools/solve_loop.py cfg5 6 32 >> $OUT/${TAG}_solve_loop.txt 2>&1
python3 tools/solve_loop.py cfg5 6 64 >> $OUT/${TAG}_solve_loop.txt 2>&1
# round 4: the mixed-precision product (kernel times, accuracy, solves; counters of pass A / pass B at cfg5 in both forms),
# the sharded path with and without the per-camera exchange inside the producers, the J-free iteration, the N > 1 bench
# path on one device, per-call overhead, the randomised parity run
python3 tools/ab_mixed.py cfg5 32 > $OUT/${TAG}_ab_mixed.txt 2>&1
python3 tools/ab_mixed.py cfg4 64 >> $OUT/${TAG}_ab_mixed.txt 2>&1
cd /tmp
for M in 1 0; do
  SFMBA_DEBUG=pcg_mixed=$M rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/${TAG}_abc_a_$M -- python3 $R/tools/time_kernels.py cfg5 4,5 32 > /dev/null 2>> $OUT/${TAG}_k1c.err
  SFMBA_DEBUG=pcg_mixed=$M rocprofv3 --pmc TA_BUSY_avr GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $OUT/${TAG}_abc_b_$M -- python3 $R/tools/time_kernels.py cfg5 4,5 32 > /dev/null 2>> $OUT/${TAG}_k1c.err
  python3 $R/tools/profile_summary.py counters $OUT/${TAG}_cfg5_passAB_counters_mixed$M.json "passes A and B of the Schur product back to back, cfg5 fp32 storage, pcg_mixed=$M" $OUT/${TAG}_abc_a_$M $OUT/${TAG}_abc_b_$M > /dev/null
  rm -rf $OUT/${TAG}_abc_a_$M $OUT/${TAG}_abc_b_$M
done
cd $R
bash tools/ab_shard.sh $TAG pcg_inline > $OUT/${TAG}_ab_shard_inline.txt 2>&1
bash tools/ab_option.sh cfg4 64 jfree 20 > $OUT/${TAG}_ab_jfree.txt 2>&1
bash tools/ab_trace.sh ${TAG}j cfg4 64 jfree 12 >> $OUT/${TAG}_ab_jfree.txt 2>&1
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --one-device --dist-backend gloo --exchange torch --steps 20 --warmup 5 --no-cpu-baseline 2> $OUT/${TAG}_bench_2rank_one_device.err | tail -1 > $OUT/${TAG}_bench_2rank_one_device.json
# round 4, late: the launch-by-launch timeline of one solve, memory-side counters of the camera-major passes, and the A/B
# of the launch pair a speculative PCG batch no longer enqueues
bash tools/timeline.sh $TAG cfg4 > /dev/null 2>&1
bash tools/cam_counters.sh $TAG cfg4 > /dev/null 2>&1
cd $R
bash tools/ab_option.sh cfg4 64 pcg_skip_last 20 > $OUT/${TAG}_ab_skip_last.txt 2>&1
python3 tools/call_overhead.py > $OUT/${TAG}_call_overhead.txt 2>&1
python3 tools/call_overhead.py cfg2 >> $OUT/${TAG}_call_overhead.txt 2>&1
python3 tools/fuzz_parity.py 300 0 > $OUT/${TAG}_fuzz.txt 2>&1
python3 tools/fuzz_parity.py 300 7 >> $OUT/${TAG}_fuzz.txt 2>&1
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_stats_cfg2 $OUT/${TAG}_stats_cfg5 $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_k1c_a_cfg4 $OUT/${TAG}_k1c_b_cfg4 $OUT/${TAG}_k1c_a_cfg5 $OUT/${TAG}_k1c_b_cfg5
cat $OUT/${TAG}_time_kernels.txt $OUT/${TAG}_solve_loop.txt $OUT/${TAG}_ab_mixed.txt $OUT/${TAG}_ab_shard_inline.txt; tail -8 $OUT/${TAG}_shard.log
