#!/bin/bash
# copy the summaries of `bash tools/make_profiles.sh <tag>` (gpurun_out/, scratch) into profiles/ (tracked)
TAG=${1:-r04}
cd "$(dirname "$0")/.."
for f in bench_line.json bench_under_rocprof.json bench_kernel_stats.md traffic.json bench_cfg2.json bench_cfg2_kernel_stats.md \
         bench_cfg3.json bench_cfg5_f32.json bench_cfg5_f64.json cfg5_f32_kernel_stats.md cfg5_f32_solve_loop.txt \
         k1_counters_cfg4.json k1_counters_cfg5_f32.json cfg5_passAB_counters_mixed0.json cfg5_passAB_counters_mixed1.json \
         shard8_cfg4_local.json shard8_cfg4_sharded.json shard8_cfg4_sharded_kernel_stats.md shard8_cfg4_sharded_under_rocprof.json \
         shard8_cfg5_local.json shard8_cfg5_sharded.json shard8_cfg5_sharded_kernel_stats.md shard8_cfg5_sharded_under_rocprof.json \
         shard.log time_kernels.txt solve_loop.txt ab_mixed.txt ab_shard_inline.txt ab_jfree.txt bench_2rank_one_device.json \
         call_overhead.txt fuzz.txt timeline.txt cam_counters_cfg4.json ab_skip_last.txt; do
  cp gpurun_out/${TAG}_$f profiles/${TAG}_$f
done
cp gpurun_out/${TAG}j_jfree0_kernel_stats.md profiles/${TAG}_jfree0_kernel_stats.md
cp gpurun_out/${TAG}j_jfree1_kernel_stats.md profiles/${TAG}_jfree1_kernel_stats.md
cp gpurun_out/${TAG}_traffic.json profiles/k1_traffic.json
ls profiles | grep -c "^${TAG}_"
