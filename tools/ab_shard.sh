#!/bin/bash
# Per-rank critical path of an 8-way sharding on one GPU (bench.py --shard-of 8 --force-exchange), A/B over a debug option:
#   bash tools/ab_shard.sh <tag> <option>          (through gpurun)
set -o pipefail
TAG=$1; OPT=$2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd $R
for W in cfg4 cfg5; do
  B=64; [ $W = cfg5 ] && B=32
  S=20; [ $W = cfg5 ] && S=10
  for V in 0 1 0 1; do
    SFMBA_DEBUG=$OPT=$V python3 bench.py --workload $W --storage-bits $B --shard-of 8 --steps $S --no-cpu-baseline --no-per-call --force-exchange 2>> $OUT/${TAG}_shard.err | tail -1 > $OUT/${TAG}_shard8_${W}_${OPT}${V}.json
    python3 - <<PY
import json
d=json.loads(open("$OUT/${TAG}_shard8_${W}_${OPT}${V}.json").read().strip().split("\n")[-1])
print("$W $OPT=$V", round(d["value"],1), "it/s", round(1e3*d["ms_per_step"],1), "us/iteration", "launches", d["launches_per_iteration"], "collectives", d["collectives_per_iteration"], d.get("transport"), flush=True)
PY
  done
  python3 bench.py --workload $W --storage-bits $B --shard-of 8 --steps $S --no-cpu-baseline --no-per-call 2>> $OUT/${TAG}_shard.err | tail -1 > $OUT/${TAG}_shard8_${W}_local.json
  python3 - <<PY
import json
d=json.loads(open("$OUT/${TAG}_shard8_${W}_local.json").read().strip().split("\n")[-1])
print("$W local", round(d["value"],1), "it/s", round(1e3*d["ms_per_step"],1), "us/iteration", "launches", d["launches_per_iteration"], flush=True)
PY
done
tail -3 $OUT/${TAG}_shard.err
