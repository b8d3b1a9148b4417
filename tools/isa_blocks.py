"""Per-basic-block opcode histogram of one kernel in a hipcc -S listing (whole function, up to .Lfunc_end).
usage: isa_blocks.py file.s kernel_substring [min_mfma]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); name, cur = m.group(1), []
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur.append(t.split(";")[0].strip())
blocks.append((name, cur))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): return "trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")) or "dpp" in op: return "xlane"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith(("s_barrier", "s_sched", "s_setprio", "s_sleep")): return "sync"
    if op.startswith("s_"): return "salu"
    return "other"
for name, ins in blocks:
    ops = [i.split()[0] for i in ins]
    nm = sum(o.startswith("v_mfma") for o in ops)
    if nm < min_mfma: continue
    c = collections.Counter(cls(o) for o in ops)
    print(f"{name} n={len(ops)} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
    d = collections.Counter(o for o in ops if cls(o) in ("valu", "valu_pk", "trans", "cvt", "xlane", "acc_mov", "scratch", "nop"))
    print("   ", ", ".join(f"{k}:{v}" for k, v in d.most_common(30)))
