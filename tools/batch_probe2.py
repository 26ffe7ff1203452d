"""Probe: does GPU_MAX_HW_QUEUES change multi-stream batch throughput? (env must be set before HIP init)"""
import os, sys, subprocess
for q in ("", "8", "16"):
    env = dict(os.environ)
    if q: env["GPU_MAX_HW_QUEUES"] = q
    print(f"--- GPU_MAX_HW_QUEUES={q or 'default'}", flush=True)
    subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "batch_probe.py"), "256", "64", "8,16"], env=env)
