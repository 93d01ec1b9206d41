#!/bin/bash
# usage: tools/ab_step.sh "ENV=val ..." ...   -> median / min per-step device time of the forward workload (40 steps) per setting
cd $GRAFT_REPO_ROOT
for setting in "$@"; do
  env $setting RPE_BENCH_STEP_TIMES=1 python bench.py --steps 40 --no-cpu-baseline --no-corr-microbench --eval-batches 0 2> /tmp/ab.err > /tmp/ab.out
  python - "$setting" <<'PY'
import sys, re, json, statistics
err = open('/tmp/ab.err').read()
m = re.findall(r"step ms: \[([^\]]*)\]", err)
first = [float(x) for x in m[0].split(",")] if m else []
d = json.loads(open('/tmp/ab.out').read().strip().splitlines()[-1])
print("%-40s median %.3f min %.3f max %.3f | line %.3f ms | epe %s %s" % (sys.argv[1], statistics.median(first), min(first), max(first), d['ms_per_step'], d['epe_delta']['epe2d'], d['epe_delta']['epe3d']))
PY
done
