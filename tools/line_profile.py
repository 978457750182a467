#!/usr/bin/env python
"""line_profile.py [counts.json] [--kernel plain|persistent] [--top N] -- executed VALU instructions and issue cycles PER SOURCE LINE.

Block execution counts of a census run (tools/isa_profile.py run -> profiles/rNN/census_counts_*.json) x the per-block instructions of the
CURRENT device assembly built with line tables (isa_census.build_asm: -gline-tables-only, same code generation).  The block structure must
match the census build's (same number of blocks and the same instruction kinds per block); the tool refuses otherwise.  Where the kernel's
issue time goes, by line of cloud_core.h / kernels.hip: the map the instruction diet of round 4 was planned with."""
import argparse, collections, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_census as IC
import isa_profile as IP
ap = argparse.ArgumentParser()
ap.add_argument("counts", nargs="?", default=os.path.join(ROOT, "profiles", "r03", "census_counts_C3.json"))
ap.add_argument("--kernel", default="plain")
ap.add_argument("--top", type=int, default=45)
ap.add_argument("--asm")
ap.add_argument("--csrc", help="source directory to build the assembly from (default: the tree's csrc)")
ap.add_argument("--force", action="store_true", help="apply the counts although some blocks' instruction kinds differ")
a = ap.parse_args()
d = json.load(open(a.counts))
counts = d["kernels"][a.kernel]["counts"]
static = d["static"][a.kernel]["blocks"]
if a.csrc:
    IC.CSRC = os.path.abspath(a.csrc)
asm = a.asm or IC.build_asm(tempfile.mkdtemp(prefix="line_profile_"))
name, blocks = IC.parse_kernel(asm, IP.KERNELS[a.kernel])
if len(blocks) != len(static):
    raise SystemExit("block structure differs: %d blocks now, %d in the census run" % (len(blocks), len(static)))
import re
def norm(mn):
    return re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
bad = 0
for b, s in zip(blocks, static):
    k = collections.Counter(norm(i[0]) for i in b["insns"])
    if dict(k) != s["kinds"]:
        bad += 1
if bad and not a.force:
    raise SystemExit("%d of %d blocks have different instructions than in the census run (use --force to apply the counts anyway)" % (bad, len(blocks)))
cal = json.load(open(IP.calibration_path()))
per_line = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
tot_n = tot_c = 0
for b, c in zip(blocks, counts):
    for mn, ops, f, ln in b["insns"]:
        if IC.classify(mn) not in ("full", "half", "trans"):
            continue
        mn = norm(mn)
        cost = IP.kind_cost(mn, cal)[0]
        e = per_line[(f, ln)]
        e[0] += c; e[1] += c * cost; e[2][mn] += c
        tot_n += c; tot_c += c * cost
print("%s: %d blocks (%d differ), executed VALU %.4g, issue cycles %.4g" % (a.kernel, len(blocks), bad, tot_n, tot_c))
rows = sorted(per_line.items(), key=lambda kv: -kv[1][1])
cum = 0.0
for (f, ln), (n, c, kinds) in rows[:a.top]:
    cum += c
    top = ", ".join("%s %.3g" % (k, v) for k, v in kinds.most_common(4))
    print("%-16s:%-4s  %9.4g instr  %5.2f %% of issue (cum %5.1f)   %s" % (f, ln, n, 100 * c / tot_c, 100 * cum / tot_c, top))
