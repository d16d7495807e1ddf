set -o pipefail
for c in "cvt --graph" "vit_tiny --graph"; do
  for v in 32 64 100 140 200 300 32; do
    echo -n "$c P8_MIN_TILES=$v: "
    VITK_P8_MIN_TILES=$v timeout -k 10 200 python bench.py --config $c --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" || exit 1
  done
done
