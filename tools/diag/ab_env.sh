# A/B of one environment switch on the headline step, alternating, one box: bash tools/diag/ab_env.sh VAR [n_pairs]
var=$1; n=${2:-2}
mkdir -p gpurun_out
for i in $(seq 1 $n); do for v in 0 1; do
  env $var=$v timeout -k 10 150 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/abe_$v.log 2>&1 || exit 1
  python - <<PY
import json
for l in open("gpurun_out/abe_$v.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$var=$v", d["ms_per_step"], d["value"], d["final_loss"])
PY
done; done
