"""Shader clock and power of the GPU while one chain-kernel family runs back to back (rocm-smi sampled beside a launch loop).
Behind it: round 4 found that the weights-stationary kernels take ~22 % fewer shader cycles than the wave-resident ones (GRBM_GUI_ACTIVE,
profiles/r04_ws_sq_counters.txt) but only ~10 % less time -- the denser kernel is clocked lower.
  python tools/clock_under_load.py [S=4194304] [seconds=4]"""
import ctypes, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
spec = importlib.util.spec_from_file_location("ws_compare", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ws_compare.py"))
W = importlib.util.module_from_spec(spec); spec.loader.exec_module(W)
from lab4d_amd import _lib, mlp

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
        d = json.loads(r.stdout)
        c = d[sorted(d)[0]]
        sclk = [v for k, v in c.items() if "sclk" in k.lower()]
        pw = [v for k, v in c.items() if "power" in k.lower() and "(W)" in k]
        return (sclk[0] if sclk else None, pw[0] if pw else None)
    except Exception as e:
        return (repr(e)[:60], None)


c = W.make_case(mlp.NET_FG_BASE, S, S // 2, 7)
out = {"S": S, "idle": smi()}
for ws in (False, True):
    f = W.run_fwd(c, ws)
    b = W.run_bwd(c, f, ws)
    for what, fn, args in (("fwd", _lib.lib().lab4d_mlp_forward, f["args"]), ("bwd", _lib.lib().lab4d_mlp_backward, b["args"])):
        os.environ["LAB4D_WS"] = "1" if ws else "0"
        samples, stop = [], False

        def sampler():
            while not stop:
                samples.append(smi())
                time.sleep(0.15)
        th = threading.Thread(target=sampler)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        t0 = time.time()
        th.start()
        e0.record()
        while time.time() - t0 < secs:
            for _ in range(20):
                fn(ctypes.byref(args), _lib.stream())
            n += 20
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop = True
        th.join()
        ms = e0.elapsed_time(e1) / n
        key = "%s_%s" % (what, "ws" if ws else "wave")
        out[key] = {"ms_per_launch_sustained": round(ms, 3), "launches": n, "smi_samples (sclk, W)": samples[2:][:12]}
        print(key, out[key], flush=True)
    del f, b
    torch.cuda.empty_cache()
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_clock_under_load.json"), "w"), indent=1)
