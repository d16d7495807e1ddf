# A/B of several values of one environment variable on the headline step, alternating, one box: bash tools/diag/ab_vals.sh VAR "v1 v2 ..." [rounds]
var=$1; vals=$2; n=${3:-2}
mkdir -p gpurun_out
for i in $(seq 1 $n); do for v in $vals; do
  env $var=$v timeout -k 10 150 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/abv.log 2>&1 || exit 1
  python - <<PY
import json
for l in open("gpurun_out/abv.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$var=$v", d["ms_per_step"], d["value"], d["final_loss"])
PY
done; done
