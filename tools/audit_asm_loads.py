"""Audit of hand-counted asm loads (extrema_kernel): hipcc treats an `asm volatile("global_load ...")` destination as written
at the asm statement, so it may read, copy or reuse that VGPR before the data has landed (cdna_hip_programming.md 5.7).
This walks siftmi_api.s (make -C siftmetal_amd/csrc asm) in program order, keeps the list of asm-issued loads that the
asm-issued `s_waitcnt vmcnt(N)` statements have not yet retired (all but the N youngest retire), and reports every
instruction OUTSIDE an asm statement that names a still-pending destination register.
usage: python tools/audit_asm_loads.py [siftmi_api.s]   (exit code 1 on a violation)"""
import os
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "siftmetal_amd", "csrc", "siftmi_api.s")
lines = open(path).read().split("\n")


def regs_of(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


bad = total_loads = kernels = 0
name, in_asm, pending = None, False, []
for ln, raw in enumerate(lines, 1):
    t = raw.strip()
    m = re.match(r"^(_ZN6siftmi\w+):", raw)
    if m:
        name, pending, in_asm = m.group(1), [], False
        continue
    if name is None or not t or t.startswith(";;#ASMSTART"):
        in_asm = in_asm or t.startswith(";;#ASMSTART")
        continue
    if t.startswith(";;#ASMEND"):
        in_asm = False
        continue
    if t.startswith("s_endpgm"):
        if pending:
            print("%s: %d asm loads never retired before s_endpgm" % (name[:60], len(pending)))
            bad += 1
        name = None
        continue
    if t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    if in_asm:
        m = re.match(r"global_load_dword\s+v(\d+),", t)
        if m:
            pending.append((int(m.group(1)), ln))
            total_loads += 1
            if len(pending) == 1:
                kernels += 1
        m = re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", t)
        if m:
            keep = int(m.group(1))
            pending = pending[len(pending) - keep:] if keep else []
        continue
    if pending:
        hit = regs_of(t) & {r for r, _ in pending}
        if hit:
            print("%s line %d: `%s` touches v%s while its asm load (line %d) is still in flight" %
                  (name[:50], ln, t, sorted(hit), [l for r, l in pending if r in hit][0]))
            bad += 1
print("audited %d asm loads; %d violation(s)" % (total_loads, bad))
sys.exit(1 if bad else 0)
