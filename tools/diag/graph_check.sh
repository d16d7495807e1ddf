for e in "X=1" "VITK_CLASS_ROWS_LAST_BLOCK=0" "VITK_WGRAD_OVERLAP=0"; do
  for extra in "" "--no-pipelined-step"; do
    env $e timeout -k 10 200 python bench.py --config vit_tiny --graph $extra --no-cpu-baseline --no-kernel-timing > gpurun_out/cfg.log 2>&1 || { echo "FAILED: $e $extra"; tail -3 gpurun_out/cfg.log; continue; }
    python - <<PY
import json
for l in open("gpurun_out/cfg.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$e $extra:", d["ms_per_step"], "loss", d["first_loss"], "->", d["final_loss"])
PY
  done
done
