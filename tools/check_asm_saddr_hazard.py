"""Static check on device assembly (hipcc --cuda-device-only -S file.hip -o file.s): an inline-asm global load whose scalar base
was written by a VALU instruction (v_readlane / v_readfirstlane, e.g. an SGPR spill reload) within the last 5 instructions reads
a stale base — gfx9 needs wait states there and the compiler's hazard recogniser does not look inside inline asm.  bam.hip puts
an s_nop in front of its load for this reason; screen.hip's loads take a base that is computed once per kernel.
usage: python tools/check_asm_saddr_hazard.py file.s"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
bad = total = 0
for i, l in enumerate(lines):
    if "ASMSTART" not in l:
        continue
    j, blk = i + 1, []
    while "ASMEND" not in lines[j]:
        blk.append(lines[j])
        j += 1
    if any(b.strip().startswith("s_nop") for b in blk[:1]):
        continue
    sregs = set()
    for b in blk:
        m = re.search(r"global_load\w*\s+\S+,\s*\S+,\s*s\[(\d+):(\d+)\]", b)
        if m:
            sregs.update([int(m.group(1)), int(m.group(2))])
    if not sregs:
        continue
    total += 1
    k, cnt = i - 1, 0
    while k > 0 and cnt < 6:
        t = lines[k].strip()
        if t and not t.startswith(";") and not t.startswith("."):
            cnt += 1
            m = re.match(r"v_read(?:first)?lane_b32\s+s(\d+)", t)
            if m and int(m.group(1)) in sregs:
                bad += 1
                print("hazard before line %d: %s -> %s" % (i + 1, t, blk[0].strip()))
                break
        k -= 1
print("%d inline-asm loads with a scalar base, %d possible hazards" % (total, bad))
sys.exit(1 if bad else 0)
