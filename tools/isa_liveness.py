"""Approximate VGPR liveness over one kernel of a hipcc -S listing: per basic block the peak number of live arch VGPRs (backward dataflow over the CFG;
destination = the first operand of loads / VALU / MFMA, everything else a use).  Used to find WHERE a kernel that sits on the 256-register budget peaks.
usage: isa_liveness.py file.s kernel_substring [top_n]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
L = open(path).read().split("\n")
st = next(i for i, l in enumerate(L) if l.startswith("_Z") and key in l and ":" in l)
en = next(i for i in range(st, len(L)) if L[i].startswith(".Lfunc_end"))
blocks, order, cur = {}, [], "entry"
blocks[cur] = []; order.append(cur)
for l in L[st + 1:en]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = m.group(1); blocks[cur] = []; order.append(cur); continue
    t = l.split(";")[0].strip()
    if not t or t.startswith("."): continue
    blocks[cur].append(t)
def vregs(tok):
    s = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok): s.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![\w\[])v(\d+)\b", tok): s.add(int(m.group(1)))
    return s
NODEF = ("ds_write", "scratch_store", "global_store", "buffer_store", "buffer_load", "v_cmp", "v_readlane", "v_readfirstlane", "s_", "ds_append", "global_load_lds")
def defuse(ins):
    op, _, rest = ins.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    d, u = set(), set()
    if op.startswith("buffer_load") and "lds" in ins: nodef = True
    else: nodef = op.startswith(NODEF) and not op.startswith("buffer_load")
    if op.startswith("buffer_load") and "lds" not in ins: nodef = False
    for i, o in enumerate(ops):
        r = vregs(o)
        if i == 0 and not nodef: d |= r
        else: u |= r
    if op.startswith(("v_writelane", "v_mfma", "v_fmac", "v_mac", "v_bfi", "v_permlane32_swap", "v_dot2c")):  # read-modify-write / partial writes / tied
        if op.startswith("v_mfma"): pass
        else: u |= d
    if op.startswith("v_permlane32_swap"): d |= vregs(ops[1]) if len(ops) > 1 else set()
    return d, u
succ = {}
for i, b in enumerate(order):
    ins = blocks[b]; s = []
    fall = True
    for t in ins:
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", t)
        if m: s.append(m.group(1))
        m = re.match(r"s_branch\s+(\.LBB\d+_\d+)", t)
        if m: s.append(m.group(1)); fall = False
        if t.startswith("s_endpgm"): fall = False
    if fall and i + 1 < len(order): s.append(order[i + 1])
    succ[b] = s
live_in = {b: set() for b in order}
changed = True
while changed:
    changed = False
    for b in reversed(order):
        live = set()
        for s_ in succ[b]: live |= live_in.get(s_, set())
        for t in reversed(blocks[b]):
            d, u = defuse(t)
            live = (live - d) | u
        if live != live_in[b]: live_in[b] = live; changed = True
res = []
for b in order:
    live = set()
    for s_ in succ[b]: live |= live_in.get(s_, set())
    peak, at = len(live), len(blocks[b])
    for idx in range(len(blocks[b]) - 1, -1, -1):
        d, u = defuse(blocks[b][idx])
        live = (live - d) | u
        if len(live) > peak: peak, at = len(live), idx
    res.append((peak, b, len(blocks[b]), at, sum("v_mfma" in t for t in blocks[b])))
for peak, b, n, at, nm in sorted(res, reverse=True)[:top]:
    print(f"{b:12s} peak live VGPRs {peak:4d} (at instr {at} of {n}; {nm} MFMAs)  e.g. {blocks[b][min(at, n - 1)] if n else ''}")
