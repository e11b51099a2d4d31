#!/usr/bin/env python3
"""The LCE variants of pgx_find_mems_pairs_kernel load their lines with asm statements the compiler does not see through, and wait for them by hand
(see the declaration of `row` in pgx_kernels.hip).  That is only right while no instruction of the compiler's touches the loaded registers between
such a load and the hand-written wait that follows it in the instruction stream (a copy there would read them before the data is in).  This script
checks exactly that on the device assembly (hipcc --save-temps): usage isa_lint.py <device .s file>; exit status 1 and the offending lines if not."""
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
import isa_extract as I  # noqa: E402


def regs_of(text):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", text):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def lint(lines):
    bad, pending, n_loads = [], set(), 0
    i = 0
    while i < len(lines):
        ln = lines[i]
        if "#ASMSTART" in ln:
            j = i + 1
            body = []
            while "#ASMEND" not in lines[j]:
                body.append(lines[j].strip())
                j += 1
            for b in body:
                if b.startswith("global_load") and "lds" not in b.split()[0]:
                    pending |= regs_of(b.split(",")[0])  # the destination
                    n_loads += 1
                elif b.startswith("s_waitcnt") and "vmcnt(0)" in b:
                    pending = set()
            i = j + 1
            continue
        s = ln.strip()
        if pending and s and not s.startswith((";", ".", "//")) and not s.endswith(":"):
            hit = regs_of(s) & pending
            if hit:
                bad.append((i, s, sorted(hit)))
        i += 1
    return bad, n_loads


if __name__ == "__main__":
    ks = I.kernels(sys.argv[1])
    rc = 0
    for k, body in ks.items():
        if "pgx_find_mems_pairs_kernel" not in k:
            continue
        bad, n = lint(body)
        if n:
            print("%s: %d asm loads, %d instructions between a load and its wait touch the loaded registers" % (k[:60], n, len(bad)))
        for b in bad:
            print("   line %d: %s  (registers %s)" % b)
            rc = 1
    sys.exit(rc)
