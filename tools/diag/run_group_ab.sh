set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/group_ab.log
timeout -k 10 900 python -u -m pytest -x -q --timeout 180 --timeout-method thread tests/test_gpu_kernels.py -k "gemm_group or wgrad_with_bias or gemm_layouts" > gpurun_out/group_tests.log 2>&1 || { tail -30 gpurun_out/group_tests.log; exit 1; }
tail -3 gpurun_out/group_tests.log
for cfg in "vit_tiny --graph" "cvt --graph" "deit_tiny"; do
  for g in 1 0 1 0; do
    echo "== $cfg no_dma_mn=$g" >> gpurun_out/group_ab.log
    if [ $g = 1 ]; then export VITK_NO_DMA128_MN=1; else unset VITK_NO_DMA128_MN; fi
    timeout -k 10 200 python bench.py --config $cfg --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> gpurun_out/group_ab.log || exit 1
  done
done
cat gpurun_out/group_ab.log
