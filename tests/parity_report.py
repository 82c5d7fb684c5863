"""Measured parity errors next to their bounds: every device parity test that has a stated bound reports what it measured."""
import os


def report(tag, measured):
    """Measured errors go to stdout (pytest -s / -rP) and to gpurun_out/parity_<tag>.json so that the bound and the measurement
    can be read side by side."""
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_%s.json" % tag), "w") as f:
        json.dump({k: float("%.3e" % v) for k, v in measured.items()}, f, indent=1, sort_keys=True)
    print(tag, {k: "%.2e" % v for k, v in measured.items()})
