set -o pipefail
timeout -k 10 900 env VITK_DEEP2=1 python -u -m pytest -x -q --timeout 180 --timeout-method thread tests/test_gpu_kernels.py -k "gemm" > gpurun_out/group_tests.log 2>&1 || { tail -40 gpurun_out/group_tests.log; exit 1; }
tail -2 gpurun_out/group_tests.log
for c in "cvt --graph" "vit_tiny --graph" "vit_b16"; do
  for v in base VITK_DEEP2 base VITK_DEEP2; do
    echo -n "$c $v: "
    env $v=1 timeout -k 10 200 python bench.py --config $c --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" || exit 1
  done
done
