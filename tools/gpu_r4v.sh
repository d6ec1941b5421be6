#!/bin/bash
mkdir -p gpurun_out/r4v; O=gpurun_out/r4v
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $O/issue_rate tools/issue_rate.hip > $O/issue_build.log 2>&1; timeout 120 $O/issue_rate 39 > $O/issue_rate_f64.json 2>$O/issue_rate.err; rm -f $O/issue_rate
for v in "far=6" "packed far=6" "packed" ""; do timeout 200 python tools/step_timeline.py $v > "$O/timeline_$(echo $v | tr ' =' '__').txt" 2>&1; done
tail -4 $O/timeline_*.txt; tail -3 $O/issue_build.log; head -c 600 $O/issue_rate_f64.json
