# every bench configuration once (eager, and captured where the step is launch-bound): bash tools/diag/all_configs.sh
mkdir -p gpurun_out
for args in "--config vit_tiny" "--config vit_tiny --graph" "--config deit_tiny" "--config deit_tiny --graph" "--config cvt" "--config cvt --graph" "--config coatnet" "--config vit_b16 --dropout 0.3"; do
  timeout -k 10 200 python bench.py $args --no-cpu-baseline --no-kernel-timing > gpurun_out/cfg.log 2>&1 || { echo "FAILED: $args"; tail -5 gpurun_out/cfg.log; continue; }
  python - <<PY
import json
for l in open("gpurun_out/cfg.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$args:", d["ms_per_step"], "ms/step", d["value"], "img/s  loss", d["first_loss"], "->", d["final_loss"])
PY
done
