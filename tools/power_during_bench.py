"""rocm-smi (shader clock, package power) sampled beside a long run of the driver's bench command: the step is power-capped under its chain kernels, so its
sustained time is a function of the box's thermal / power state.   python tools/power_during_bench.py [steps=200]  -> one JSON line"""
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
samples, stop = [], threading.Event()


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
        c = json.loads(r.stdout)
        c = c[sorted(c)[0]]
        sclk = [v for k, v in c.items() if "sclk" in k.lower()]
        pw = [v for k, v in c.items() if "power" in k.lower() and "(W)" in k]
        mhz = int(re.findall(r"(\d+)\s*Mhz", sclk[0])[0]) if sclk else None
        return mhz, float(pw[0]) if pw else None
    except Exception:
        return None, None


def sampler(t0):
    while not stop.is_set():
        m, w = smi()
        samples.append((round(time.time() - t0, 2), m, w))
        time.sleep(0.25)


t0 = time.time()
th = threading.Thread(target=sampler, args=(t0,))
th.start()
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline", "--no-extras"],
                   capture_output=True, text=True)
stop.set()
th.join()
line = json.loads(r.stdout.strip().splitlines()[-1])
busy = [(m, w) for _, m, w in samples if w is not None and w > 900]
out = {"what": "rocm-smi sampled every ~0.3 s beside `bench.py --steps %d --warmup 5 --no-extras` (whole training step, bf16); busy = samples above 900 W" % steps,
       "bench": {k: line[k] for k in ("value", "ms_per_step", "steps", "loss_last_chunk", "params_finite", "steps_discarded_by_check_grad", "peak_hbm_gib")},
       "busy_samples": len(busy), "sclk_mhz_mean": round(sum(m for m, _ in busy) / max(len(busy), 1), 1), "power_w_mean": round(sum(w for _, w in busy) / max(len(busy), 1), 1),
       "sclk_mhz_min_max": [min((m for m, _ in busy), default=None), max((m for m, _ in busy), default=None)],
       "power_w_min_max": [min((w for _, w in busy), default=None), max((w for _, w in busy), default=None)],
       "samples_every_4th": samples[::4]}
print(json.dumps(out))
