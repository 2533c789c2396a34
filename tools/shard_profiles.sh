#!/bin/bash
# Per-rank critical path of an 8-way strong-scaling run, measured on ONE GPU:
#   bash tools/shard_profiles.sh r03      (through gpurun)
# For cfg4 (fp64) and cfg5 (fp32 storage): rank 0's share of the 8-way sharding (all cameras, 1/8 of the points and
# observations) solved (a) as a single-rank problem -- the local form of the PCG, no collectives -- and (b) through the
# SHARDED code path (--force-exchange: world-size-1 communicator + direct all-reduce kernels over the rank's own staging
# buffer; every collective of an 8-GPU run is launched, only the xGMI hops are missing), the latter also under
# rocprofv3 --kernel-trace --stats.
set -o pipefail
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for W in cfg4 cfg5; do
  B=64; [ $W = cfg5 ] && B=32
  S=20; [ $W = cfg5 ] && S=10
  python3 $R/bench.py --workload $W --storage-bits $B --shard-of 8 --steps $S --no-cpu-baseline --no-per-call > $OUT/${TAG}_shard8_${W}_local.json 2>> $OUT/${TAG}_shard.err
  python3 $R/bench.py --workload $W --storage-bits $B --shard-of 8 --steps $S --no-cpu-baseline --no-per-call --force-exchange > $OUT/${TAG}_shard8_${W}_sharded.json 2>> $OUT/${TAG}_shard.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_shard8_${W} -- python3 $R/bench.py --workload $W --storage-bits $B --shard-of 8 --steps $S --no-cpu-baseline --no-per-call --force-exchange > $OUT/${TAG}_shard8_${W}_sharded_under_rocprof.json 2>> $OUT/${TAG}_shard.err
  python3 $R/tools/profile_summary.py stats $OUT/${TAG}_shard8_${W} $OUT/${TAG}_shard8_${W}_sharded_kernel_stats.md > /dev/null
  rm -rf $OUT/${TAG}_shard8_${W}
  for f in local sharded; do python3 - <<PY
import json
d=json.loads(open("$OUT/${TAG}_shard8_${W}_$f.json").read().strip().split("\n")[-1])   # (RCCL prints a banner first)
print("$W $f", round(d["value"],1), "it/s", round(1e3*d["ms_per_step"],1), "us/iteration", "launches", d["launches_per_iteration"], "collectives", d["collectives_per_iteration"], d.get("transport"))
PY
  done
done
tail -5 $OUT/${TAG}_shard.err
