"""Audit of the inline-asm patch loads of conv_x6c.hip (cdna_hip_programming.md 5.7 item 1: hipcc does not know that an asm load's
destination lands later).  For every kernel in the .s: each `global_load_dword vN, vM, s[..]` inside an ASMSTART block must not have
its destination register mentioned again (read, copied, overwritten) before an asm `s_waitcnt vmcnt` in linear order, and must be
preceded by `s_nop 4` when its SGPR base was written by v_readfirstlane right before.
usage: python tools/check_asm_loads.py /tmp/x6c.s"""
import re
import sys

t = open(sys.argv[1]).read().split("\n")
kern = None
bad = 0
i = 0
while i < len(t):
    l = t[i]
    m = re.match(r"^(_ZN3p2i\w+):", l)
    if m:
        kern = m.group(1)
    if "global_load_dword " in l and i >= 1 and ("ASMSTART" in t[i - 1] or "s_nop 4" in t[i - 1]):
        dst = re.search(r"global_load_dword (v\d+),", l).group(1)
        n = int(dst[1:])
        j = i + 1
        waited = False
        while j < len(t) and "s_endpgm" not in t[j]:
            lj = t[j].strip()
            if lj.startswith("s_waitcnt vmcnt"):
                waited = True
                break
            if not lj.startswith(";") and "global_load_dword" not in lj:
                regs = set(int(x) for x in re.findall(r"\bv(\d+)\b", lj))
                for a, b in re.findall(r"v\[(\d+):(\d+)\]", lj):
                    regs.update(range(int(a), int(b) + 1))
                if n in regs:
                    print("HAZARD %s line %d: %s touched before a vmcnt wait: %s" % (kern, j, dst, lj[:90]))
                    bad += 1
                    break
            j += 1
    i += 1
print("hazards:", bad)
sys.exit(1 if bad else 0)
