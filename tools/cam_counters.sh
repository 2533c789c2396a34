#!/bin/bash
# Memory-side counters of the camera-major passes (K3, pass B, rhs + preconditioner blocks) back to back at one workload
# (through gpurun):  bash tools/cam_counters.sh <tag> [workload] [bits]
set -o pipefail
TAG=$1; W=${2:-cfg4}; B=${3:-64}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_cc_a -- python3 $R/tools/time_kernels.py $W 2,5,8 $B > /dev/null 2> $OUT/${TAG}_cc.err
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCC_EA0_RDREQ_sum --output-format csv -d $OUT/${TAG}_cc_b -- python3 $R/tools/time_kernels.py $W 2,5,8 $B > /dev/null 2>> $OUT/${TAG}_cc.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/${TAG}_cc_c -- python3 $R/tools/time_kernels.py $W 2,5,8 $B > /dev/null 2>> $OUT/${TAG}_cc.err
python3 $R/tools/profile_summary.py counters $OUT/${TAG}_cam_counters_$W.json "camera-major passes back to back, $W, $B-bit storage" $OUT/${TAG}_cc_a $OUT/${TAG}_cc_b $OUT/${TAG}_cc_c > /dev/null
rm -rf $OUT/${TAG}_cc_a $OUT/${TAG}_cc_b $OUT/${TAG}_cc_c
python3 - <<PY
import json
d=json.load(open("$OUT/${TAG}_cam_counters_$W.json"))["kernels"]
for k,v in d.items():
    if any(s in k for s in ("k_cam_blocks","k_cam_schur","k_cam_rhs_diag")):
        print(k[:60], {c: round(x["avg"]) for c,x in v.items()})
PY
