#!/bin/bash
# diagnostic: the shader clock while the default bench runs (a frame of launch-floor kernels keeps the chip mostly idle), and the
# headline with the performance level pinned to "high" (rocm-smi) -- what the DVFS governor costs a latency-bound workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-clocks}
mkdir -p $O
cd $R
sample() {  # $1 = out file
  for i in $(seq 1 60); do
    for f in /sys/class/drm/card*/device/pp_dpm_sclk; do [ -r $f ] && grep '\*' $f | tr '\n' ' '; done
    echo
    sleep 0.1
  done > $1
}
rocm-smi --showperflevel --showclocks > $O/smi_before.txt 2>&1
python bench.py --no-cpu-baseline --no-extras --steps 3000 --warmup 20 > $O/line_auto.json 2> $O/line_auto.err &
BP=$!
sleep 12; sample $O/sclk_auto.txt
wait $BP
rocm-smi --setperflevel high > $O/set_high.txt 2>&1
rocm-smi --showperflevel --showclocks > $O/smi_high.txt 2>&1
python bench.py --no-cpu-baseline --no-extras --steps 3000 --warmup 20 > $O/line_high.json 2> $O/line_high.err &
BP=$!
sleep 12; sample $O/sclk_high.txt
wait $BP
rocm-smi --setperflevel auto > /dev/null 2>&1
python - <<PY
import json
for n in ("auto", "high"):
    try:
        d = json.loads(open("$O/line_%s.json" % n).read().strip().splitlines()[-1])
        print(n, "fps", d["value"], "ms", d["ms_per_step"], "icp us", d["roofline"]["avg_us"], "frac", d["roofline"]["frac"])
    except Exception as e:
        print(n, "ERR", e)
    import collections
    c = collections.Counter(l.strip() for l in open("$O/sclk_%s.txt" % n))
    print("  sclk samples:", c.most_common(4))
PY
