import os, sys, subprocess, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for L in (20, 26, 35, 44, 52, 60, 70, 88, 104, 140):
    env = dict(os.environ, EZKL_MSM_L=str(L))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
    j = json.loads(out.strip().splitlines()[-1])
    print(L, "step %.3f ms  dev %.3f  acc %.3f" % (j["ms_per_step"], j["extra"]["msm_device_ms"], j["roofline"]["avg_launch_ms"]), flush=True)
