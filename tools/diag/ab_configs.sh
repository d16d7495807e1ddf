# A/B of the second-stream weight gradients on the secondary configurations (one box): bash tools/diag/ab_configs.sh
mkdir -p gpurun_out
for cfg in coatnet cvt deit_tiny vit_tiny; do
  for ov in 0 1; do
    VITK_WGRAD_OVERLAP=$ov timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --no-kernel-timing > gpurun_out/abc_${cfg}_$ov.log 2>&1 || exit 1
    python - <<PY
import json
for l in open("gpurun_out/abc_${cfg}_$ov.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$cfg overlap=$ov", d["ms_per_step"], d["value"], d["final_loss"])
PY
  done
done
