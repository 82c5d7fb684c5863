"""Summarise rocprofv3 --pmc passes into per-kernel HBM traffic per launch.

usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> [source note]
Each dir holds one rocprofv3 `--pmc X --kernel-trace --output-format csv` run (FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Units and corrections per
that guide: both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    """rocprof kernel symbol -> the name bench.py reports (k_mlp_fwd<FgBase>, k_mlp_wgrad_dma<8,4>, ...)."""
    m = re.match(r"(?:void )?(?:lab4d::)?(k_\w+)(<.*>)?", name)
    if not m:
        return name.split("(")[0][:60]
    base, targs = m.group(1), m.group(2) or ""
    net = re.search(r"Net(\w+?)[,>]", targs)
    if base in ("k_mlp_fwd", "k_mlp_bwd") and net:
        flags = re.findall(r"\b(true|false)\b", targs)  # k_mlp_fwd<Net, P, TAN, ST>
        tan = base == "k_mlp_fwd" and len(flags) >= 1 and flags[0] == "true"
        return "%s<%s>%s" % (base, net.group(1), "@tangent" if tan else "")
    if base in ("k_mlp_fwd_ws", "k_mlp_bwd_ws") and net:  # the weights-stationary chains (csrc/mlp_kernels_ws.hpp)
        return "%s<%s>" % (base, net.group(1))
    if base == "k_mlp_wgrad_dma":
        nums = re.findall(r"\d+", targs)
        return "k_mlp_wgrad_dma<%s>" % ",".join(nums[:2])
    if base == "k_mlp_wgrad":
        tm = re.search(r"(\d+)\s*>$", targs)
        return "k_mlp_wgrad<%s>" % (tm.group(1) if tm else "?")
    return base


def collect(d, counter):
    """kernel -> [launches, sum] over the FULL-SIZE launches of each kernel: the bench also launches the MLP kernels on
    the small eikonal subsample (bench.py reports those under '<kernel>@eik'); launches whose counter value is below half
    of the kernel's largest are dropped so they do not dilute the per-launch figure."""
    vals = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                vals[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    acc = {}
    for k, v in vals.items():
        keep = [x for x in v if x >= 0.5 * max(v)]
        acc[k] = [len(keep), sum(keep)]
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    F, W = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    res = {"_source": note, "_units": "bytes per launch; read = 2*FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB"}
    for k in sorted(set(F) | set(W)):
        nf, sf = F.get(k, [0, 0.0]); nw, sw = W.get(k, [0, 0.0])
        rd = 2.0 * 1024.0 * sf / max(nf, 1); wr = 1024.0 * sw / max(nw, 1)
        res[k] = {"launches": max(nf, nw), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    top = sorted(((v["hbm_bytes_per_launch"] * v["launches"], k) for k, v in res.items() if isinstance(v, dict)), reverse=True)[:25]
    for tot, k in top:
        v = res[k]
        print("%-34s n=%6d  read %10.1f MB  write %10.1f MB per launch" % (k, v["launches"], v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
