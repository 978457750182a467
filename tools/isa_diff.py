#!/usr/bin/env python
"""isa_diff.py A.s B.s [kernel-substring ...] -- are the gfx950 bodies of the named kernels identical in two device assemblies (labels renumbered)?
The guard used while refactoring shared headers in round 3: the headline kernels' assembly must not change unless the change is meant to."""
import re, sys
def body(path, name):
    L = open(path).read().split("\n"); out = []; on = False
    for ln in L:
        if not on and ln.startswith("_ZN") and name in ln.split(":")[0] and ":" in ln:
            on = True; continue
        if on:
            if ln.startswith(".Lfunc_end") or ln.strip().startswith(".amdhsa_kernel"): break
            t = ln.split(";")[0].rstrip()
            if not t.strip() or t.lstrip().startswith("."): continue
            out.append(re.sub(r"\.LBB\d+_", ".LBB_", t))
    return out
a, b = sys.argv[1], sys.argv[2]
names = sys.argv[3:] or ["clouds_kernelILi3ELi1ENS_6TexSetE", "clouds_kernel_persistentILi3E", "clouds_kernelILi3ELi2E", "clouds_kernelILi3ELi4E", "clouds_kernelILi1ELi1E", "clouds_kernelILi0ELi1E"]
bad = 0
for k in names:
    x, y = body(a, k), body(b, k)
    same = x == y and len(x) > 0
    bad += 0 if same else 1
    print("%-36s %5d %5d  %s" % (k, len(x), len(y), "identical" if same else "DIFFERENT"))
sys.exit(1 if bad else 0)
