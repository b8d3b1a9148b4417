"""Wait-state check of the hand-placed kernels: every MFMA result register against the first non-MFMA instruction that touches it.

The 64-per-wave kernels issue their MFMAs from inline asm (operand register classes are part of the design), so hipcc's hazard recognizer pads nothing around
them: "matrix pipe writes a VGPR / AGPR, another unit reads or overwrites it" needs passes + 3 wait states (CDNA3 ISA guide 4.5, "XDL write VGPR -> VALU
read / write, VMEM / LDS / FLAT read": 5 / 11 / 19 for 2 / 8 / 16 passes) and nothing in the tool chain inserts them.  In a full step the distance is there by
construction (a dozen MFMAs between a score chain's end and its row-max tree); a variant that REMOVES instructions can lose it -- round 5's peeled first
iteration read a chain 3 instructions behind its last MFMA and the row maxima came from whatever the registers held before (results within tolerance,
but not reproducible: 17 run-to-run failures in the GPU suite, none standalone).

Counts along the fall-through order of the disassembly AND across every branch to its target (loop back edges, the jumps into and out of cold code; three branches
deep): a non-MFMA instruction = 1 wait state, s_nop N = N + 1, an MFMA in between = its passes (the matrix pipe takes the next one that many quads later).  A dependent MFMA
(accumulating into the same tuple, or the register as its A / B operand) is not checked: back-to-back accumulation is interlocked, and no kernel here feeds
an MFMA result to an A / B operand without a VALU conversion in between.

Second rule, the other direction (scan_operands): a VALU result as an MFMA operand needs 2 wait states.  Third (scan_valu_pairs): a VALU result as an operand of
v_permlane*_swap needs 2, a transcendental's result in a non-transcendental VALU instruction 1 -- an asm statement gets neither from hipcc.  Fourth (scan_sgpr_vmem):
an SGPR written by a VALU instruction (v_readfirstlane ...) as the descriptor / offset of a memory instruction needs 5.  Fifth (scan_lds_waits): every register an LDS
read returns into is covered by a counted s_waitcnt lgkmcnt before its first use (the asm-issued transposed reads have no other protection; weakening any one of the
hand-placed waits of the forward by 2 is flagged in 12 of 15 sampled places, the rest have slack).

usage: isa_mfma_hazards.py [file.o | file.s | file.dis ...]      (default: every object of flash-attention_amd/csrc)
exit code 1 on any finding."""
import functools, glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    """gfx950 code object of a hipcc host object -> llvm-objdump text."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(td, "copy.o")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)


def passes_of(op):
    m = re.match(r"v_mfma_\w+?_(\d+)x(\d+)x(\d+)_?(\w*)", op)
    if not m:
        return 16
    mm, _, kk, ty = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)
    if ty in ("bf16", "f16"):   # gfx950: 32x32x16 = 8 passes, 16x16x32 = 4; the half-K forms of gfx942 take as long for half the work
        return 8 if mm == 32 else 4
    if ty.startswith(("f8", "bf8", "fp8")) or "f8f6f4" in op:
        return 8 if mm == 32 else 4
    return 16 if mm == 32 else 8    # f32 / xf32 / f64 / i8: priced at the slow end (none in the hot kernels)


def regs_of(text):
    """{('v', n), ('a', n), ...} named in an operand string."""
    out = set()
    for kind, lo, hi in re.findall(r"\b([va])\[(\d+):(\d+)\]", text):
        out.update((kind, r) for r in range(int(lo), int(hi) + 1))
    for kind, n in re.findall(r"\b([va])(\d+)\b", text):
        out.add((kind, int(n)))
    return out


@functools.lru_cache(maxsize=2)
def parse(text):
    """-> [(function, [(op, operands, branch target index or None)])]: llvm-objdump -d text (targets from the instruction addresses) or a hipcc -S listing (labels)."""
    funcs, cur, labels, addrs = [], None, {}, {}
    for raw in text.split("\n"):
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:", raw)
        if m:
            cur, labels, addrs = [], {}, {}
            funcs.append((m.group(1), cur, labels, addrs))
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.L\w+):", raw)
        if m:
            labels[m.group(1)] = len(cur)
            continue
        line = re.split(r"//|;", raw)[0].strip()
        if not line or line.startswith(".") or line.endswith(":") or not re.match(r"^[a-z]", line):
            continue
        m = re.search(r"//\s*([0-9A-Fa-f]+):", raw)
        if m:
            addrs[int(m.group(1), 16)] = len(cur)
        op, _, rest = line.partition(" ")
        cur.append([op, rest.strip(), int(m.group(1), 16) if m else None])
    out = []
    for name, ins, labels, addrs in funcs:
        res = []
        for op, rest, addr in ins:
            tgt = None
            if op.startswith(("s_branch", "s_cbranch")):
                if rest in labels:
                    tgt = labels[rest]
                elif addr is not None and re.match(r"^-?\d+$", rest):
                    off = int(rest)
                    tgt = addrs.get(addr + 4 + 4 * (off - 65536 if off >= 32768 else off))
            res.append((op, rest, tgt))
        out.append((name, res))
    return out


def scan(text):
    """-> [(function, wait states, needed, mfma line, consumer line)] for every MFMA result touched too early.  The walk follows the fall-through order; at every
    branch (conditional or not) the registers still pending are also carried to the branch target (and over further branches, three deep)."""
    found, seen = [], set()

    def step(func, ins, i, pending):
        """one instruction against the pending list [dst regs, waited, needed, mfma text, mfma index]; returns the new list"""
        op, rest, _ = ins[i]
        line = f"{op} {rest}"
        if op.startswith(("v_mfma", "v_smfmac")):
            p = passes_of(op)
            dst = regs_of(rest.split(",")[0])
            touched = regs_of(rest.split(",", 1)[1] if "," in rest else "")
            keep = []
            for ent in pending:
                ent = [ent[0], ent[1] + p, ent[2], ent[3], ent[4]]
                if ent[0] & dst:
                    continue                         # accumulating on in the same tuple: interlocked
                if ent[1] < ent[2] and (ent[0] & touched):   # an MFMA result as another tuple's operand: reported, it is not interlocked either
                    if (ent[4], i) not in seen:
                        seen.add((ent[4], i)); found.append((func, ent[1] - p, ent[2], ent[3], line))
                    continue
                if ent[1] < ent[2]:
                    keep.append(ent)
            return keep + [[dst, 0, p + 3, line, i]]
        if op == "s_nop":
            n = int(rest or 0) + 1
            return [[e[0], e[1] + n, e[2], e[3], e[4]] for e in pending if e[1] + n < e[2]]
        touched = regs_of(rest) if not op.startswith("s_") else set()
        keep = []
        for ent in pending:
            if ent[0] & touched:
                if (ent[4], i) not in seen:
                    seen.add((ent[4], i)); found.append((func, ent[1], ent[2], ent[3], line))
                continue
            if ent[1] + 1 < ent[2]:
                keep.append([ent[0], ent[1] + 1, ent[2], ent[3], ent[4]])
        return keep

    def side_walk(func, ins, i, pending, depth):
        while pending and i < len(ins):
            op, _, tgt = ins[i]
            pending = step(func, ins, i, pending)
            if op.startswith(("s_branch", "s_cbranch")) and tgt is not None and depth < 3 and pending:
                side_walk(func, ins, tgt, [list(e) for e in pending], depth + 1)
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                return
            i += 1

    for func, ins in parse(text):
        pending = []
        for i, (op, _, tgt) in enumerate(ins):
            pending = step(func, ins, i, pending)
            if op.startswith(("s_branch", "s_cbranch")) and tgt is not None and pending:
                side_walk(func, ins, tgt, [list(e) for e in pending], 1)
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                pending = []
    return found


def scan_operands(text, need=2):
    """The other direction: a VALU instruction writes a VGPR, an MFMA reads it as an operand fewer than `need` wait states later (hipcc keeps 2 in the kernels it
    schedules itself -- fa_bwd.hip's dQ kernel has 68 operand pairs at exactly 2 and none below -- and knows nothing about the inline-asm ones).
    -> [(function, wait states, needed, writer, mfma)]"""
    found = []
    for func, ins in parse(text):
        recent = []   # [registers written, wait states since, text]
        for op, rest, _ in ins:
            line = f"{op} {rest}"
            if op.startswith(("v_mfma", "v_smfmac")):
                srcs = regs_of(rest.split(",", 1)[1] if "," in rest else "")
                found += [(func, age, need, w, line) for regs, age, w in recent if regs & srcs]
                recent = []
                continue
            n = int(rest or 0) + 1 if op == "s_nop" else 1
            recent = [[r, a + n, w] for r, a, w in recent if a + n < need]
            if op.startswith("v_") and not op.startswith("v_cmp") and "," in rest:
                recent.append([regs_of(rest.split(",")[0]), 0, line])
    return found


def scan_valu_pairs(text):
    """Two VALU -> VALU distances hipcc keeps by itself and an asm statement does not get: a VALU result as a v_permlane*_swap operand (2 wait states; the compiler-
    scheduled forwards hold 135 / 176 swaps at exactly 2, none below) and a transcendental's result in a non-transcendental VALU instruction (1).
    -> [(function, wait states, needed, writer, reader)]"""
    TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")
    found = []
    for func, ins in parse(text):
        recent = []   # [registers written, wait states since, text, is transcendental]
        for op, rest, _ in ins:
            line = f"{op} {rest}"
            if op.startswith("v_") and "," in rest and not op.startswith(("v_mfma", "v_smfmac")):
                swap = op.startswith(("v_permlane16_swap", "v_permlane32_swap"))
                reads = regs_of(rest) if swap else regs_of(rest.split(",", 1)[1])
                for regs, age, w, tr in recent:
                    if regs & reads:
                        if swap and age < 2:
                            found.append((func, age, 2, w, line))
                        elif tr and age < 1 and not op.startswith(TRANS):
                            found.append((func, age, 1, w, line))
            n = int(rest or 0) + 1 if op == "s_nop" else 1
            recent = [[r, a + n, w, t] for r, a, w, t in recent if a + n < 2]
            if op.startswith("v_") and not op.startswith("v_cmp") and "," in rest:
                written = regs_of(rest.split(",")[0])
                if op.startswith(("v_permlane16_swap", "v_permlane32_swap", "v_swap")):
                    written |= regs_of(rest.split(",")[1])
                recent.append([written, 0, line, op.startswith(TRANS)])
    return found


def sregs_of(text):
    out = set()
    for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(lo), int(hi) + 1))
    out.update(int(n) for n in re.findall(r"\bs(\d+)\b", text))
    if "vcc" in text:
        out.update((106, 107))
    return out


def scan_sgpr_vmem(text, need=5):
    """A VALU instruction writes an SGPR (v_readfirstlane, v_readlane, a compare, a carry out), a memory instruction reads it as descriptor / offset fewer than 5 wait
    states later (hipcc: 44 pairs at exactly 5 in fa_bwd.hip's dK/dV kernel, none below; the tile DMAs of the 64-per-wave kernels are asm statements).
    -> [(function, wait states, needed, writer, reader)]"""
    found = []
    for func, ins in parse(text):
        recent = []
        for op, rest, _ in ins:
            line = f"{op} {rest}"
            if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
                rd = sregs_of(rest)
                found += [(func, age, need, w, line) for regs, age, w in recent if regs & rd]
            n = int(rest or 0) + 1 if op == "s_nop" else 1
            recent = [[r, a + n, w] for r, a, w in recent if a + n < need]
            if op.startswith("v_") and "," in rest:
                written = sregs_of(rest.split(",")[0])
                if op.startswith("v_cmp") and not written:
                    written = {106, 107}   # the e32 form writes vcc
                if written:
                    recent.append([written, 0, line])
    return found


def scan_lds_waits(text):
    """Every register an LDS read returns into, against its first use: the counted s_waitcnt lgkmcnt(N) in front of the use must cover the read.  hipcc inserts these
    waits for the loads it knows; the transposed V-fragment reads of the 64-per-wave forward (ds_read_b64_tr_b16) and a few others are asm statements, and the paired,
    counted waits of the hand-placed steps are then the only thing between an MFMA and a register LDS has not written yet.
    Model: cnt = an upper bound of the hardware counter (+1 per LDS / scalar-memory / flat instruction, min(cnt, N) at a wait); LDS operations return in order, so
    the read with sequence number q among n issued is certainly back once n - q >= cnt (scalar and flat loads share the counter and return out of order: they can only
    take slots away from LDS operations, the bound stays valid).  Walks the fall-through order and, like scan(), carries the state over every branch to its target.
    -> [(function, reads possibly in flight, 0, the read, the use)]"""
    found, seen = [], set()

    def step(func, ins, i, st):
        cnt, n_lds, pending = st
        op, rest, _ = ins[i]
        line = f"{op} {rest}"
        if op.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                cnt = min(cnt, int(m.group(1)))
            elif re.match(r"^(0x[0-9a-fA-F]+|\d+)$", rest.strip()):
                cnt = min(cnt, (int(rest.strip(), 0) >> 8) & 0xF)
            pending = [p for p in pending if n_lds - p[0] < cnt]
            return [cnt, n_lds, pending]
        touched = regs_of(rest) if not op.startswith("s_") else set()
        keep = []
        for pnd in pending:
            if pnd[1] & touched:
                if (pnd[3], i) not in seen:
                    seen.add((pnd[3], i)); found.append((func, min(cnt, n_lds - pnd[0] + 1), 0, pnd[2], line))
                continue
            keep.append(pnd)
        pending = keep
        if op.startswith("ds_"):
            n_lds += 1; cnt += 1
            first = rest.split(",")[0]
            if re.match(r"^\s*v", first) and any(k in op for k in ("read", "load", "permute", "swizzle", "consume", "append")) :
                pending = pending + [[n_lds, regs_of(first), line, i]]
        elif op.startswith(("s_load", "s_buffer_load", "s_scratch_load", "flat_")):
            cnt += 1
        return [cnt, n_lds, pending]

    def side_walk(func, ins, i, st, depth, budget=600):
        while st[2] and i < len(ins) and budget > 0:
            op, _, tgt = ins[i]
            st = step(func, ins, i, st)
            if op.startswith(("s_branch", "s_cbranch")) and tgt is not None and depth < 2 and st[2]:
                side_walk(func, ins, tgt, [st[0], st[1], [list(p) for p in st[2]]], depth + 1, budget)
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                return
            i += 1; budget -= 1

    for func, ins in parse(text):
        st = [0, 0, []]
        for i, (op, _, tgt) in enumerate(ins):
            st = step(func, ins, i, st)
            if op.startswith(("s_branch", "s_cbranch")) and tgt is not None and st[2]:
                side_walk(func, ins, tgt, [st[0], st[1], [list(p) for p in st[2]]], 1)
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                st = [st[0], st[1], []]
    return found


def main(argv):
    files = argv or sorted(glob.glob(os.path.join(ROOT, "flash-attention_amd", "csrc", "*.o")))
    bad = 0
    for f in files:
        if f.endswith(".o"):
            if b".hip_fatbin" not in subprocess.check_output([f"{LLVM}/llvm-objdump", "-h", f]):
                continue   # a host-only object
            text = disassemble(f)
        else:
            text = open(f).read()
        n_mfma = len(re.findall(r"\bv_mfma", text))
        hits, ops, pairs, lds = scan(text), scan_operands(text), scan_valu_pairs(text) + scan_sgpr_vmem(text), scan_lds_waits(text)
        print(f"{os.path.basename(f)}: {n_mfma} MFMAs, {len(hits)} result registers touched early, {len(ops)} operands written late, {len(pairs)} swap / transcendental / scalar-operand pairs too close, "
              f"{len(lds)} LDS reads used before their wait")
        ops = ops + pairs + lds
        for func, ws, need, first, second in hits + ops:
            print(f"    {func[:70]}: {ws} of {need} wait states\n        {first}\n        {second}")
        bad += len(hits) + len(ops)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
