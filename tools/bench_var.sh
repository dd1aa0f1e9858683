#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
for i in 1 2; do
  for S in 4 6; do
    (cd "$R" && EZKL_MSM_SLOTS=$S python bench.py --no-cpu-baseline) 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('run $i SLOTS=$S value %.4g ms/step %.4f acc %.4f msm_dev %.4f ntt_dev %.4f modmul29 %.4g copy %.0f' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['extra']['msm_device_ms'], j['extra']['ntt_device_ms'], j['extra']['modmul29_per_s'], j['extra']['hbm_copy_GBs']))"
  done
done
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6
