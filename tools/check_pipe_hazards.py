"""Static check of the pipelined screen kernel's ISA: no instruction may touch a VGPR that an inline-asm load has
written but that has not been covered by a later s_waitcnt yet (the compiler does not know those registers are pending;
a live-range split or copy inserted between the load and the wait would read stale data).
usage: python tools/check_pipe_hazards.py <kernel.s>   (text of ONE kernel, in layout order)"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pending = []   # list of (set(regs), line) in issue order
def regs_of(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None: out.add(int(m.group(3)))
        else: out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out
in_asm = False
bad = 0
for n, l in enumerate(lines, 1):
    t = l.strip()
    if "#ASMSTART" in t: in_asm = True; continue
    if "#ASMEND" in t: in_asm = False; continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
    op = t.split()[0]
    if in_asm:
        if op.startswith("global_load"):
            dst = t.split(None, 1)[1].split(",")[0]
            pending.append((regs_of(dst), n))
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            k = int(m.group(1))
            pending = pending[len(pending) - k:] if k and len(pending) > k else ([] if k == 0 else pending)
        continue
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", t)
        if m and int(m.group(1)) == 0: pending = []
        continue
    used = regs_of(t.split(None, 1)[1]) if " " in t else set()
    for rs, ln in pending:
        if used & rs:
            print("line %d touches v%s pending since line %d: %s" % (n, sorted(used & rs), ln, t))
            bad += 1
            break
print("hazards:", bad)
