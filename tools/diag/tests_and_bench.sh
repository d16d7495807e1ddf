# the round-end sequence on one box: GPU suite, smoke, default bench line (outputs under gpurun_out/)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest -x -q --timeout 180 --timeout-method thread tests -m gpu > gpurun_out/full_gpu_tests.log 2>&1 || { tail -30 gpurun_out/full_gpu_tests.log; exit 1; }
tail -2 gpurun_out/full_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.log").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["mfma_util_pct"], d["roofline"]["frac"])
for s in d["secondary"]:
	print(s["config"], "graph" if s["hip_graph"] else "eager", s["dropout"], s["ms_per_step"], s["value"])
PY
