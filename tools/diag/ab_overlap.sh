mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/ab_$name.log 2>&1 || return 1
  python - <<PY
import json
for l in open("gpurun_out/ab_$name.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$name", d["ms_per_step"], d["value"])
PY
}
run ov0_ev0 VITK_WGRAD_OVERLAP=0 VITK_P8_EVEN_ROUNDS=0 && run ov1_ev0 VITK_WGRAD_OVERLAP=1 VITK_P8_EVEN_ROUNDS=0 && run ov0_ev1 VITK_WGRAD_OVERLAP=0 VITK_P8_EVEN_ROUNDS=1 && run ov1_ev1 VITK_WGRAD_OVERLAP=1 VITK_P8_EVEN_ROUNDS=1 && run ov0_ev0b VITK_WGRAD_OVERLAP=0 VITK_P8_EVEN_ROUNDS=0 && run ov1_ev1b VITK_WGRAD_OVERLAP=1 VITK_P8_EVEN_ROUNDS=1
