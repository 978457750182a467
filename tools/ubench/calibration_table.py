#!/usr/bin/env python
"""Turns the rocprofv3 csv files of tools/ubench/calibrate.sh into the calibration bench.py prices the cloud kernel with
(profiles/r02/issue_cost_calibration.json) and prints the tables (profiles/r02/valu_issue_costs_gfx950.txt, gather_access_costs_gfx950.txt).
All cycle figures are SQ_BUSY_CYCLES / 32 (one count per shader engine): the SQ's own clock, no frequency assumed."""
import collections
import csv
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03/calib"
N_SE, N_SIMD, N_CU = 32, 1024, 256


def dispatches(prefix, sub):
    kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(d, sub, prefix + "_kernel_trace.csv")))}
    cnt = collections.defaultdict(dict)
    for r in csv.DictReader(open(os.path.join(d, sub, prefix + "_counter_collection.csv"))):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        cnt[r["Dispatch_Id"]]["_name"] = r["Kernel_Name"].split("(")[0].split("::")[-1]
        cnt[r["Dispatch_Id"]]["_grid"] = int(r["Grid_Size"])
        cnt[r["Dispatch_Id"]]["_ns"] = int(kt[r["Dispatch_Id"]]["End_Timestamp"]) - int(kt[r["Dispatch_Id"]]["Start_Timestamp"])
    return [cnt[k] for k in sorted(cnt, key=int)]


out = {"units": "cycles = SQ_BUSY_CYCLES / 32 shader engines (the SQ's own clock); per wave64 instruction per SIMD, 8 waves per SIMD resident"}
# ---- VALU issue classes
rows = []
seen = collections.Counter()
for c in dispatches("v", "valu"):
    key = (c["_name"], c["_grid"])
    seen[key] += 1
    waves_per_simd = c["_grid"] // 64 // N_SIMD
    if seen[key] != 2 or waves_per_simd != 8:          # the timed launch at full occupancy
        continue
    cyc = c["SQ_BUSY_CYCLES"] / N_SE
    n = c["SQ_INSTS_VALU"]
    per = cyc / (n / N_SIMD)
    cls = {k[len("SQ_INSTS_VALU_"):]: c.get(k, 0.0) / n for k in c if k.startswith("SQ_INSTS_VALU_")}
    rows.append((c["_name"], per, c["_ns"] / 1e3, cyc / c["_ns"] * 1e3, cls))
print("# gfx950 VALU issue cost per wave64 instruction per SIMD (8 waves/SIMD resident), cycles = SQ_BUSY_CYCLES/32; counter = share of the kind's")
print("# instructions each SQ_INSTS_VALU_<class> counter saw (tools/ubench/valu_rates2.hip under rocprofv3 --pmc)")
for name, per, us, mhz, cls in rows:
    tags = " ".join("%s=%.2f" % (k, v) for k, v in sorted(cls.items()) if v > 0.01)
    print("%-20s %5.2f cycles   %8.1f us  %4.0f MHz   %s" % (name[2:], per, us, mhz, tags))
cost = {r[0][2:]: r[1] for r in rows}
full = [cost[k] for k in ("fma_3src", "fma_sgpr_const", "fmac", "mul", "add", "mov", "and", "add_u32", "mul_e64_2vgpr") if k in cost]
half = [cost[k] for k in ("fma_mix", "max_min", "med3", "lshl", "lshl_or", "and_or", "bfe", "bfi", "lshl_add", "cvt_flr", "fract", "floor", "cvt_f32_i32", "cvt_f32_f16",
                          "cmp_sgpr", "cndmask_sgpr", "mul_lo_u32", "mad_u32_u24") if k in cost]
trans = [cost[k] for k in ("rcp", "exp", "log", "sqrt") if k in cost]
out["valu"] = {"full_rate_cycles": sum(full) / len(full), "half_rate_cycles": sum(half) / len(half), "transcendental_cycles": sum(trans) / len(trans),
               "full_rate_kinds": "v_fma_f32 v_fmac_f32 v_mul_f32 v_add_f32 v_sub_f32 v_mov_b32 v_and_b32 v_add_u32",
               "half_rate_kinds": "v_fma_mix_f32 v_max/min/med3_f32 v_lshl* v_and_or v_bfe/bfi v_lshl_add v_cvt_* v_fract/floor v_cmp_* v_cndmask v_mul_lo_u32 v_mad_u32_u24 v_pk_*_f32",
               "per_kind_cycles": cost,
               "counter_classes": {r[0][2:]: {k: round(v, 3) for k, v in r[4].items() if v > 0.01} for r in rows}}
print("# classes: full rate %.2f, half rate %.2f, transcendental %.2f cycles" % (out["valu"]["full_rate_cycles"], out["valu"]["half_rate_cycles"], out["valu"]["transcendental_cycles"]))
# ---- TCP access cost
print("# gfx950 vector-L1 (TCP) cost of 16-byte gathers (tools/ubench/gather_rates.hip): per wave-level load instruction")
g = dispatches("g", "gather")
best = None
seen = collections.Counter()
names = ["coalesced", "same-line", "lines-4", "lines-16", "random"]
sizes = ["16 KiB", "1 MiB", "32 MiB", "1 GiB"]
idx = 0
table = []
for c in g:
    if "gather" not in c["_name"]:
        continue
    seen[c["_name"]] += 1
    # launches come in (warm-up, timed) pairs: 5 patterns x 4 footprints
    idx += 1
    if idx % 2:
        continue
    k = idx // 2 - 1
    pat, size = names[k % 5], sizes[k // 5]
    cyc = c["SQ_BUSY_CYCLES"] / N_SE
    loads = c["SQ_INSTS_VMEM_RD"]
    acc = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    row = dict(pattern=pat, footprint=size, cycles_per_load_per_cu=cyc / (loads / N_CU), tcp_accesses_per_load=acc / loads, cycles_per_tcp_access=cyc / (acc / N_CU),
               ta_busy_frac=c["TA_TA_BUSY_sum"] / (N_CU * cyc), tcc_req_per_load=c["TCP_TCC_READ_REQ_sum"] / loads)
    table.append(row)
    print("%-10s %-7s %6.1f cycles/load/CU  %5.1f TCP accesses/load  %5.2f cycles/access  TA busy %.2f  %5.1f L2 requests/load" % (
        pat, size, row["cycles_per_load_per_cu"], row["tcp_accesses_per_load"], row["cycles_per_tcp_access"], row["ta_busy_frac"], row["tcc_req_per_load"]))
l1 = [r for r in table if r["footprint"] == "16 KiB"]
out["tcp"] = {"table": table,
              "cycles_per_access_distinct_lines": max(r["cycles_per_tcp_access"] for r in l1 if r["pattern"] != "random"),
              "cycles_per_access_best_case": min(r["cycles_per_tcp_access"] for r in l1),
              "note": "one CU's TCP retires one cache access (one 64-byte sector of one line) per cycle when the lanes of a quad fall in distinct lines or the wave is "
                      "coalesced (16 accesses per 1 KiB wave load), up to 1.67 per cycle for fully random lanes; TA_TA_BUSY is 0.93-1.00 of the kernel cycles "
                      "in every saturated pattern, so bench.py reports TA_TA_BUSY / (CUs x kernel cycles) as the L1-gather fraction"}
# ---- FETCH_SIZE against known byte counts (1 GiB footprint: every line of a wave load misses the L2)
try:
    f = dispatches("f", "fetch")
    big_only = True
    print("# FETCH_SIZE calibration (1 GiB and 32 MiB footprints): bytes per wave-level load instruction as FETCH_SIZE reports them vs what the pattern must bring in")
    idx, ftab = 0, []
    expect = {"coalesced": 1024.0, "same-line": 128.0, "lines-4": 512.0, "lines-16": 2048.0, "random": None}
    for c in f:
        if "gather" not in c["_name"]:
            continue
        idx += 1
        if idx % 2:
            continue
        k = idx // 2 - 1
        pat, size = names[k % 5], sizes[2 + k // 5]           # the pass runs `gather_rates big`: 32 MiB and 1 GiB footprints only
        loads = c["SQ_INSTS_VMEM_RD"]
        rep = c["FETCH_SIZE"] * 1024.0 / loads
        miss = c["TCC_MISS_sum"] / loads
        row = dict(pattern=pat, footprint=size, fetch_size_bytes_per_load=rep, tcc_misses_per_load=miss, fetch_size_bytes_per_miss=c["FETCH_SIZE"] * 1024.0 / max(1.0, c["TCC_MISS_sum"]),
                   ea_rdreq_per_load=c.get("TCC_EA_RDREQ_sum", 0.0) / loads, ea_rdreq_32b_per_load=c.get("TCC_EA_RDREQ_32B_sum", 0.0) / loads,
                   known_line_bytes_per_load=expect[pat], reported_over_known=(rep / expect[pat]) if expect[pat] else None)
        ftab.append(row)
        print("%-10s %-7s FETCH_SIZE %7.1f B/load  %5.2f TCC misses/load  %5.1f B per miss  EA rdreq %5.2f (32B: %5.2f) per load   known %s B/load  reported/known %s" % (
            pat, size, rep, miss, row["fetch_size_bytes_per_miss"], row["ea_rdreq_per_load"], row["ea_rdreq_32b_per_load"], expect[pat], "%.2f" % row["reported_over_known"] if expect[pat] else "-"))
    out["fetch_size"] = {"table": ftab}
except (OSError, KeyError) as e:
    print("# no FETCH_SIZE calibration pass:", e)
json.dump(out, open(os.path.join(os.path.dirname(d.rstrip("/")), "issue_cost_calibration.json"), "w"), indent=1)
