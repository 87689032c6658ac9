#!/usr/bin/env python3
"""Diffs the prototypes of integration/Interface_thx.cpp against the reference's gpu/interface/Interface.h:16-528.

BUILD-CONTAINER ONLY: reads /root/reference (absent on the GPU box); tests/test_abi_cpu.py runs it when the reference is
there.  Interface.h cannot be compiled in this image (it includes cuthunder.h and, through Volume.h, boost), so the check is
textual: every free function the stub DEFINES must have a declaration in Interface.h with the same name, return type and
parameter TYPE list (names and whitespace ignored; overloads matched by their type list), and every 3-D entry of Interface.h
must be defined by the stub (the 2-D mode entries listed in SKIP_2D are out of scope).
Exit status 0 = no difference.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/gpu/interface/Interface.h"
STUB = os.path.join(ROOT, "integration", "Interface_thx.cpp")
SKIP_2D = {"ExpectGlobal2D", "ExpectLocalV2D", "ExpectLocalPreI2D", "InsertI2D", "ExposePT2D", "ExposeWT2D", "ExposePF2D",
           "ExposeCorrF2D"}


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    return re.sub(r"//[^\n]*", " ", s)


def param_types(plist):
    """'Complex* traP, const int *iCol' -> ('Complex*', 'const int*')"""
    out = []
    depth, cur = 0, ""
    for ch in plist:
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            depth += ch in "<(" 
            depth -= ch in ">)"
            cur += ch
    if cur.strip():
        out.append(cur)
    types = []
    for p in out:
        p = " ".join(p.split())
        p = re.sub(r"\s*([*&])\s*", r"\1 ", p).strip()          # 'double *offS' -> 'double* offS'
        m = re.match(r"^(.*?[\w>*&])\s+(\w+)$", p)              # drop the parameter name
        t = m.group(1) if m and not re.match(r"^(const|unsigned|int|bool|double|float|void)$", m.group(2)) else p
        t = re.sub(r"\s+", " ", t).replace(" *", "*").replace(" &", "&").replace("std::", "")
        types.append(t)
    return tuple(types)


def prototypes(text, definitions):
    """free functions: 'void Name(args);' (declarations) or 'void Name(args) {' (definitions)"""
    text = strip_comments(text)
    end = r"\{" if definitions else r";"
    out = []
    for m in re.finditer(r"(?m)^\s*(void|int|bool)\s+(\w+)\s*\(([^;{}]*?)\)\s*" + end, text):
        out.append((m.group(2), m.group(1), param_types(m.group(3))))
    return out


def main():
    if not os.path.exists(REF):
        print("iface_check: %s not present (GPU box) -- nothing to check" % REF)
        return 0
    ref = prototypes(open(REF).read(), False)
    stub = prototypes(open(STUB).read(), True)
    ref_set = {(n, r, t) for n, r, t in ref}
    bad = 0
    for n, r, t in stub:
        if (n, r, t) not in ref_set:
            cands = [tt for nn, rr, tt in ref if nn == n]
            print("MISMATCH %s: stub has (%s)" % (n, ", ".join(t)))
            for c in cands:
                diff = [(i, a, b) for i, (a, b) in enumerate(zip(c, t)) if a != b]
                print("   Interface.h: (%s)%s" % (", ".join(c), "  differs at %s" % diff if len(c) == len(t) else "  (%d vs %d parameters)" % (len(c), len(t))))
            if not cands:
                print("   no function of that name in Interface.h")
            bad += 1
    stub_set = {(n, t) for n, r, t in stub}
    for n, r, t in ref:
        if n in SKIP_2D:
            continue
        if (n, t) not in stub_set:
            print("MISSING  %s(%s) is declared in Interface.h and not defined by the stub" % (n, ", ".join(t)))
            bad += 1
    print("iface_check: %d declarations in Interface.h (%d 2-D entries out of scope), %d definitions in the stub, %d problems"
          % (len(ref), sum(1 for n, _, _ in ref if n in SKIP_2D), len(stub), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
