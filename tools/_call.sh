rocm-smi --showperflevel 2>&1 | grep -i "perf" | head -3
echo "--- auto"; AB_COOP=7 timeout 200 python tools/msm_ab.py 2>&1 | tail -1
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showperflevel 2>&1 | grep -i "perf" | head -3; rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -3
echo "--- high"; AB_COOP=7 timeout 200 python tools/msm_ab.py 2>&1 | tail -1
WHAT=ntt timeout 100 python tools/msm_steps.py 2>&1 | tail -1
rocm-smi --setperflevel auto 2>&1 | tail -2
WHAT=ntt timeout 100 python tools/msm_steps.py 2>&1 | tail -1
