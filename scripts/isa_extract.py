#!/usr/bin/env python3
"""Cut one kernel out of a hipcc --save-temps device .s file and summarise it: registers, scratch, instruction mix (whole kernel and per
labelled region between two marker comments).  Usage: isa_extract.py <file.s> <substring of the mangled name> [-o out.s]"""
import collections
import re
import sys


def kernels(path):
    cur, body, out = None, [], {}
    for ln in open(path):
        m = re.match(r"^(_Z\w+):\s", ln)
        if m and cur is None:
            cur, body = m.group(1), []
        if cur is not None:
            body.append(ln)
            if re.match(r"^\s*\.size\s+" + re.escape(cur), ln) or ln.startswith(".Lfunc_end"):
                out[cur] = body
                cur = None
    return out


def meta(path, name):
    txt = open(path).read()
    d = {}
    for key in ("num_vgpr", "num_agpr", "numbered_sgpr", "private_seg_size"):
        m = re.search(r"\.set %s\.%s, (\S+)" % (re.escape(name), key), txt)
        if m:
            d[key] = m.group(1)
    m = re.search(r"; ScratchSize: (\d+)[^\n]*\n(?:[^\n]*\n){0,12}?", txt[txt.find(name + ":"):])
    return d


def classify(op):
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def mix(lines):
    c = collections.Counter()
    ops = collections.Counter()
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c[classify(op)] += 1
        ops[op] += 1
    return c, ops


if __name__ == "__main__":
    path, pat = sys.argv[1], sys.argv[2]
    ks = kernels(path)
    hits = [k for k in ks if pat in k]
    if not hits:
        sys.exit("no kernel matches %r; have %d kernels" % (pat, len(ks)))
    for k in hits:
        c, ops = mix(ks[k])
        print(k)
        print("   ", meta(path, k), dict(c))
        print("    top ops:", ops.most_common(14))
    if "-o" in sys.argv:
        open(sys.argv[sys.argv.index("-o") + 1], "w").writelines(ks[hits[0]])
