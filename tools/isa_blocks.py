"""Static instruction mix of a kernel's basic blocks (from `hipcc -S --cuda-device-only`): which blocks hold the MFMAs and how many
VALU / LDS / VMEM / SALU instructions the single wave per SIMD has to issue between them.
    python tools/isa_blocks.py file.s <mangled-name-substring> [min_mfma]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "accmov"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"
blocks, cur, name = [], collections.Counter(), "entry"
ops = collections.defaultdict(collections.Counter)
for i in range(start + 1, end + 1):
    l = lines[i].split(";")[0].strip()
    if not l or l.startswith("."): 
        if l.startswith(".LBB") and l.endswith(":"):
            blocks.append((name, cur)); cur, name = collections.Counter(), l[:-1]
        continue
    if l.endswith(":"):
        blocks.append((name, cur)); cur, name = collections.Counter(), l[:-1]
        continue
    op = l.split()[0]
    c = cls(op)
    cur[c] += 1
    if c == "nop":
        m = re.search(r"s_nop (\d+)", l); cur["nop_cycles"] += int(m.group(1)) + 1 if m else 1
    ops[name][op] += 1
    if c == "branch":  # fall-through code after a branch is a block of its own
        blocks.append((name, cur)); cur, name = collections.Counter(), name + "+"
blocks.append((name, cur))
tot = collections.Counter()
for n, c in blocks: tot.update(c)
print("kernel total:", dict(tot))
for n, c in blocks:
    if c["mfma"] >= min_mfma:
        print(f"{n:12s} mfma {c['mfma']:4d} valu {c['valu']:4d} accmov {c['accmov']:3d} lds {c['lds']:3d} vmem {c['vmem']:3d} salu {c['salu']:4d} smem {c['smem']:3d} wait {c['wait']:3d} "
              f"barrier {c['barrier']} nop {c['nop']}({c['nop_cycles']}cy)  valu/mfma {c['valu']/c['mfma']:.2f}")
        if "-v" in sys.argv:
            print("     ", ", ".join(f"{o} {k}" for o, k in ops[n].most_common(28) if not o.startswith("v_mfma")))

# run lengths of vector / memory instructions between consecutive MFMAs of the blocks listed above (the in-order issue picture)
if "-r" in sys.argv:
    name = "entry"; runs = collections.defaultdict(list); run = 0
    for i in range(start + 1, end + 1):
        l = lines[i].split(";")[0].strip()
        if not l: continue
        if l.endswith(":"):
            runs[name].append(run); run = 0; name = l[:-1]; continue
        if l.startswith("."): continue
        op = l.split()[0]
        c = cls(op)
        if c == "branch":
            runs[name].append(run); run = 0; name = name + "+"; continue
        if c == "mfma":
            runs[name].append(run); run = 0
        elif c in ("valu", "accmov", "lds", "vmem", "nop", "wait", "barrier"):
            run += 1
    runs[name].append(run)
    for n, c in blocks:
        if c["mfma"] >= min_mfma:
            r = runs[n]
            # excess = instructions beyond the ~7 issue slots one 32x32x16 MFMA covers: the matrix pipe idles for about 4 cycles each
            print(n, "runs:", r, " excess(>7):", sum(x - 7 for x in r if x > 7), " zero-gaps:", sum(1 for x in r[1:-1] if x == 0))
