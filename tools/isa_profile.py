#!/usr/bin/env python
"""isa_profile.py -- basic-block execution counts of the cloud kernels by ASSEMBLY REWRITING, and the instruction census they give.

Why: the roofline of bench.py prices VALU instructions by kind.  Hardware class counters see only part of the mix (28 % of this kernel's
instructions are in no class counter), PC sampling is "not supported on any of the agents" on this pool
(profiles/r03/pc_sampling_probe.txt), and a thread trace is a separate tool chain.  So the profiler is built here, at ISA level:

    build   hipcc -save-temps -> the device assembly of kernels.hip (the SAME flags as the product build); every basic block of the chosen
            kernels gets a counter: lane k of VGPR c of three extra VGPRs (64 counters per register), bumped by
                s_mov_b64 s[a:b], exec ; s_mov_b32 exec_lo/hi, 1 << k ; v_add_u32 vC, 1, vC ; s_mov_b64 exec, s[a:b]
            (no memory access, no SCC/VCC, only registers above the kernel's own allocation); before every s_endpgm the 256 lane counters
            are added to the kernel's `stats` buffer with global atomics.  The rewritten assembly is assembled, linked and bundled
            back into a complete libcloudsky (godot-volumetric-cloud-demo-v2_amd/libcloudsky_census.so).
    run     (GPU box) renders the workload with the census library (CSKY_LIBRARY), checks the frame is byte-identical to the product
            library's, reads the block counts (csky_census_clouds) and writes counts + static census to a JSON file.
    report  block counts x static per-block instruction histogram -> executed wave-instructions by kind, priced with the measured gfx950
            issue costs (profiles/r02/issue_cost_calibration.json: per KIND, not per class) -> VALU issue cycles per SIMD; compared with
            the hardware counters of the same workload (SQ_INSTS_VALU, _SALU, VMEM_RD, LDS) as a cross-check of the census itself.

Counts are wave-level (one per wavefront entering the block, whatever its EXEC mask: that is what occupies an issue slot).
"""
import argparse
import collections
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_census as IC  # noqa: E402

CSRC = os.path.join(ROOT, "godot-volumetric-cloud-demo-v2_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNELS = {"plain": "_ZN4csky13clouds_kernelILi3ELi1ENS_6TexSetE", "persistent": "_ZN4csky24clouds_kernel_persistentILi3E"}
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-pass-failed"]
N_CTR_VGPR = 4                      # 256 block counters per kernel
CENSUS_BYTE_OFFSET = 16             # the kernel's own two 64-bit tallies come first in the stats buffer


def kernel_span(lines, mangled_prefix):
    """(index of the label line, index of .Lfunc_end line, index of '.amdhsa_kernel' line, index of '.end_amdhsa_kernel')"""
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and l.rstrip().split(":")[0].startswith(mangled_prefix) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    kd0 = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel " + mangled_prefix))   # (sits between the code and .Lfunc_end)
    kd1 = next(i for i in range(kd0, len(lines)) if lines[i].strip() == ".end_amdhsa_kernel")
    return start, min(end, kd0), kd0, kd1


def block_starts(lines, start, end):
    """Line indices of the first instruction of every basic block, with the SAME splitting rule as isa_census.parse_kernel (labels start a
    block, branches and s_endpgm end one; empty blocks are dropped)."""
    starts, cur_first, n_in_block = [], None, 0
    endpgm = []
    for i in range(start + 1, end):
        ln = lines[i]
        m = IC.LABEL.match(ln)
        if m:
            lab = m.group(1)
            if lab.startswith(".LBB") or lab.startswith("BB"):
                if n_in_block:
                    starts.append(cur_first)
                cur_first, n_in_block = None, 0
            continue
        if ln.lstrip().startswith(".") or ln.lstrip().startswith(";") or not ln.strip():
            continue
        m = IC.INSN.match(ln)
        if not m:
            continue
        if cur_first is None:
            cur_first = i
        n_in_block += 1
        mn = m.group(1)
        if mn == "s_endpgm":
            endpgm.append(i)
        if mn.startswith("s_cbranch") or mn == "s_branch" or mn == "s_endpgm":
            starts.append(cur_first)
            cur_first, n_in_block = None, 0
    if n_in_block:
        starts.append(cur_first)
    return starts, endpgm


def explicit_arg_offsets(lines, mangled_prefix):
    """kernarg offsets of the kernel's explicit arguments (by_value / global_buffer; hidden_* arguments follow them) from the amdhsa metadata"""
    name = next(i for i, l in enumerate(lines) if l.strip().startswith(".name:") and l.split(":", 1)[1].strip().startswith(mangled_prefix))
    a0 = max(i for i in range(name) if lines[i].strip() == ".args:")
    offs, cur = [], {}
    for i in range(a0 + 1, name):
        t = lines[i].strip()
        if t.startswith("- "):
            if cur:
                offs.append(cur)
            cur = {}
            t = t[2:]
        if not t.startswith("."):
            if t and not t.startswith("-"):
                break
            continue
        k, _, v = t.partition(":")
        cur[k.strip()] = v.strip()
        if k.strip() in (".group_segment_fixed_size", ".kernarg_segment_align"):
            cur.pop(k.strip()); break
    if cur:
        offs.append(cur)
    return [int(a[".offset"]) for a in offs if a.get(".value_kind") in ("by_value", "global_buffer")]


def instrument(lines, mangled_prefix):
    start, end, kd0, kd1 = kernel_span(lines, mangled_prefix)
    kd = {}
    for i in range(kd0, kd1):
        m = re.match(r"\s*\.amdhsa_(\w+)\s+(\S+)", lines[i])
        if m:
            kd[m.group(1)] = (i, m.group(2))
    nv, ns = int(kd["next_free_vgpr"][1]), int(kd["next_free_sgpr"][1])
    kernarg = int(kd["kernarg_size"][1])
    if int(kd["user_sgpr_kernarg_segment_ptr"][1]) != 1 or int(kd["user_sgpr_count"][1]) != 2:
        raise SystemExit("unexpected user SGPR layout: the kernarg pointer is expected in s[0:1]")
    first = next(i for i in range(start + 1, end) if IC.INSN.match(lines[i]) and not lines[i].lstrip().startswith(("." , ";")))
    if "s[0:1]" not in lines[first]:
        raise SystemExit("first instruction does not read the kernarg pointer s[0:1]: %s" % lines[first])
    F = (ns + 1) & ~1                       # s[F:F+1] = census buffer, s[F+2:F+3] = saved EXEC
    if F + 4 > 102:
        raise SystemExit("no free SGPRs for the instrumentation")
    C0 = (nv + 3) & ~3                      # counters v[C0 .. C0+2], temporary v[C0+3]
    T = C0 + N_CTR_VGPR
    starts, endpgm = block_starts(lines, start, end)
    if len(starts) > 64 * N_CTR_VGPR:
        raise SystemExit("%d basic blocks: more than %d counters" % (len(starts), 64 * N_CTR_VGPR))
    stats_off = explicit_arg_offsets(lines, mangled_prefix)[-2]      # (..., stats, wg_cost): the second-to-last explicit argument
    del kernarg
    ins = collections.defaultdict(list)
    ins[first] += ["\ts_load_dwordx2 s[%d:%d], s[0:1], 0x%x" % (F, F + 1, stats_off)] + ["\tv_mov_b32_e32 v%d, 0" % (C0 + k) for k in range(N_CTR_VGPR)]
    for b, li in enumerate(starts):
        c, lane = divmod(b, 64)
        lo, hi = (1 << lane, 0) if lane < 32 else (0, 1 << (lane - 32))
        ins[li] += ["\ts_mov_b64 s[%d:%d], exec" % (F + 2, F + 3), "\ts_mov_b32 exec_lo, 0x%x" % lo, "\ts_mov_b32 exec_hi, 0x%x" % hi,
                    "\tv_add_u32_e32 v%d, 1, v%d" % (C0 + c, C0 + c), "\ts_mov_b64 exec, s[%d:%d]" % (F + 2, F + 3), "\ts_nop 4"]
    flush = ["\ts_waitcnt vmcnt(0) lgkmcnt(0)", "\ts_mov_b64 exec, -1", "\tv_mbcnt_lo_u32_b32 v%d, -1, 0" % T, "\tv_mbcnt_hi_u32_b32 v%d, -1, v%d" % (T, T),
             "\tv_lshlrev_b32_e32 v%d, 2, v%d" % (T, T)]
    for k in range(N_CTR_VGPR):
        flush.append("\tglobal_atomic_add v%d, v%d, s[%d:%d] offset:%d" % (T, C0 + k, F, F + 1, CENSUS_BYTE_OFFSET + 256 * k))
    flush.append("\ts_waitcnt vmcnt(0)")
    for li in endpgm:
        ins[li] = ins[li] + flush           # (the block counter of the s_endpgm's own block, if it starts there, comes first)
    out = []
    for i, ln in enumerate(lines):
        if i in ins:
            # the first instruction of the kernel keeps its place in front of its own block counter only for the entry load
            out += ins[i]
        if kd0 < i < kd1:
            ln = re.sub(r"(\.amdhsa_next_free_vgpr)\s+\d+", r"\1 %d" % (T + 1), ln)
            ln = re.sub(r"(\.amdhsa_next_free_sgpr)\s+\d+", r"\1 %d" % (F + 4), ln)
            ln = re.sub(r"(\.amdhsa_accum_offset)\s+\d+", r"\1 %d" % ((T + 1 + 3) & ~3), ln)
        out.append(ln)
    return out, len(starts)


def cmd_build(args):
    import tempfile
    own = args.work is None                                    # a private work directory per build: concurrent builds on one box never share one
    work = tempfile.mkdtemp(prefix="isa_profile_") if own else args.work
    if not own:
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(work)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-save-temps", "-c", os.path.join(CSRC, "kernels.hip"), "-o", os.path.join(work, "k.o")], cwd=work,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = os.path.join(work, "kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
    shutil.copy(asm, os.path.join(work, "product.s"))
    lines = open(asm).read().split("\n")
    info = {}
    for tag, pre in KERNELS.items():
        lines, nb = instrument(lines, pre)
        info[tag] = nb
    open(os.path.join(work, "census.s"), "w").write("\n".join(lines))
    run = lambda c: subprocess.check_call(c, cwd=work)
    run([LLVM + "/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-main-file-name", "kernels.hip", "-target-cpu", "gfx950", "-mrelocation-model", "pic",
         "-o", "dev.o", "census.s"])
    run([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-plugin-opt=-amdgpu-internalize-symbols", "-plugin-opt=mcpu=gfx950", "-o", "dev.out", "dev.o"])
    run([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null",
         "-input=dev.out", "-output=dev.hipfb"])
    run(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", "dev.hipfb", "-c", os.path.join(CSRC, "kernels.hip"), "-o", "kernels_host.o"])
    out = os.path.join(ROOT, "godot-volumetric-cloud-demo-v2_amd", "libcloudsky_census.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-shared", "-o", out, os.path.join(work, "kernels_host.o"), "bc7enc.hip", "api.cpp", "assets.cpp", "godot_import.cpp"], cwd=CSRC)
    # static census of the PRODUCT assembly (block indices are shared with the instrumentation: same splitting rule)
    static = {}
    for tag, pre in KERNELS.items():
        name, blocks = IC.parse_kernel(os.path.join(work, "product.s"), pre)
        cen, loops = IC.census(blocks)
        if len(cen) != info[tag]:
            raise SystemExit("block count mismatch for %s: census %d vs instrumentation %d" % (tag, len(cen), info[tag]))
        static[tag] = {"kernel": name, "blocks": [{"n": b["n"], "classes": b["classes"], "kinds": b["kinds"], "depth": b["depth"]} for b in cen]}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_collect
    static["source_hash"] = pmc_collect.source_hash()
    json.dump(static, open(os.path.join(ROOT, "godot-volumetric-cloud-demo-v2_amd", "libcloudsky_census.json"), "w"))
    print("built %s (%s basic blocks instrumented), static census -> libcloudsky_census.json" % (out, info))
    if own:
        shutil.rmtree(work, ignore_errors=True)


def census_available():
    """(ok, why): the census library exists and was built from the kernel sources as they are now"""
    lib = os.path.join(ROOT, "godot-volumetric-cloud-demo-v2_amd", "libcloudsky_census.so")
    js = os.path.join(ROOT, "godot-volumetric-cloud-demo-v2_amd", "libcloudsky_census.json")
    if not (os.path.exists(lib) and os.path.exists(js)):
        return False, "libcloudsky_census.so not built (python tools/isa_profile.py build)"
    import pmc_collect
    if json.load(open(js)).get("source_hash") != pmc_collect.source_hash():
        return False, "libcloudsky_census.so was built from other kernel sources"
    return True, ""


def run_counts(config, quiet=False):
    """GPU box: counts of one frame of `config` for the plain kernel and (CSKY_PERSISTENT=2) the persistent form."""
    lib = os.path.join(ROOT, "godot-volumetric-cloud-demo-v2_amd", "libcloudsky_census.so")
    static = json.load(open(os.path.join(ROOT, "godot-volumetric-cloud-demo-v2_amd", "libcloudsky_census.json")))
    res = {"config": config, "source_hash": static["source_hash"], "kernels": {}}
    args = argparse.Namespace(config=config)
    for tag in ("plain", "persistent"):
        env = dict(os.environ, CSKY_LIBRARY=lib, CSKY_PERSISTENT="2" if tag == "persistent" else "0")
        code = ("import sys, json, hashlib, numpy as np; sys.path.insert(0, %r); import gvcd_amd\n"
                "W,H,ps,ls,sun = {'C2':(512,256,64,4,(0,1,0)),'C3':(2048,1024,128,6,(1,1,0)),'C5frame':(4096,2048,128,6,(1,1,0))}[%r]\n"
                "s=np.asarray(sun,np.float64); s=(s/np.linalg.norm(s)).astype(np.float32)\n"
                "p=np.array([W,H,0,0,0,0,0,0,0,0,0,0,0.270588,0.188235,0.027451,1.0,s[0],s[1],s[2],1.0,1.0,1.0,1.0,0.0,0.0,0.05,0.2,0.0],np.float32)\n"
                "c=gvcd_amd.Context(0); c.set_noise(*gvcd_amd.assets.load_default_noise()); c.set_march(ps,ls); c.set_segments(1); c.render_transmittance(256,64); c.render_sky_lut(s,200,100,readback=False)\n"
                "img=c.render_clouds(p); st=c.cloud_stats()\n"
                "cnt=c.census_clouds(p, W, (8,0,1,H//8), 256)\n"
                "print(json.dumps({'frame_sha': hashlib.sha256(img.tobytes()).hexdigest()[:16], 'stats': {k:int(v) for k,v in st.items()}, 'counts': [int(x) for x in cnt], 'lib': gvcd_amd.library_path()}))\n"
                % (ROOT, args.config))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            raise SystemExit("census run (%s) failed: %s" % (tag, r.stderr[-1500:]))
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        # the product library's frame of the same workload
        env2 = dict(os.environ, CSKY_PERSISTENT="2" if tag == "persistent" else "0")
        env2.pop("CSKY_LIBRARY", None)
        code2 = code.replace("cnt=c.census_clouds(p, W, (8,0,1,H//8), 256)\n", "cnt=[]\n")
        r2 = subprocess.run([sys.executable, "-c", code2], env=env2, capture_output=True, text=True, timeout=300)
        d2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
        d["product_frame_sha"] = d2["frame_sha"]
        d["frame_identical_to_product"] = d["frame_sha"] == d2["frame_sha"]
        nb = len(static[tag]["blocks"])
        d["counts"] = d["counts"][:nb]
        res["kernels"][tag] = d
        if not quiet:
            print("%s: %d blocks, frame identical to the product library: %s, entry block executed %d times" % (tag, nb, d["frame_identical_to_product"], d["counts"][0]))
    res["static"] = static
    return res


def cmd_run(args):
    res = run_counts(args.config)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(res, open(args.out, "w"))
    print("wrote", args.out)


def kind_cost(kind, cal):
    """issue cycles per wave64 instruction of one VALU kind (per-kind table measured on gfx950; classes as fall-back)"""
    pk = cal["valu"]["per_kind_cycles"]
    full, half, trans = cal["valu"]["full_rate_cycles"], cal["valu"]["half_rate_cycles"], cal["valu"]["transcendental_cycles"]
    table = {"v_fma_f32": pk["fma_3src"], "v_fmac_f32": pk["fmac"], "v_mul_f32": pk["mul"], "v_add_f32": pk["add"], "v_sub_f32": pk["add"], "v_subrev_f32": pk["add"],
             "v_mov_b32": pk["mov"], "v_and_b32": pk["and"], "v_or_b32": pk["and"], "v_xor_b32": pk["and"], "v_add_u32": pk["add_u32"], "v_sub_u32": pk["add_u32"],
             "v_subrev_u32": pk["add_u32"], "v_fma_mix_f32": pk["fma_mix"], "v_max_f32": pk["max_min"], "v_min_f32": pk["max_min"], "v_med3_f32": pk["med3"],
             "v_lshlrev_b32": pk["lshl"], "v_lshrrev_b32": pk["lshl"], "v_ashrrev_i32": pk["lshl"], "v_lshl_or_b32": pk["lshl_or"], "v_and_or_b32": pk["and_or"],
             "v_bfe_u32": pk["bfe"], "v_bfi_b32": pk["bfi"], "v_lshl_add_u32": pk["lshl_add"], "v_cvt_flr_i32_f32": pk["cvt_flr"], "v_fract_f32": pk["fract"],
             "v_floor_f32": pk["floor"], "v_cvt_f32_i32": pk["cvt_f32_i32"], "v_cvt_f32_u32": pk["cvt_f32_i32"], "v_cvt_f32_f16": pk["cvt_f32_f16"],
             "v_cndmask_b32": pk["cndmask_sgpr"], "v_rcp_f32": pk["rcp"], "v_exp_f32": pk["exp"], "v_log_f32": pk["log"], "v_sqrt_f32": pk["sqrt"], "v_rsq_f32": pk["sqrt"],
             "v_mul_lo_u32": pk["mul_lo_u32"], "v_mad_u32_u24": pk["mad_u32_u24"], "v_fmaak_f32": pk.get("fmamk", pk["fma_sgpr_const"]), "v_fmamk_f32": pk.get("fmamk", pk["fma_sgpr_const"])}
    for kk, key in (("v_add_lshl_u32", "add_lshl"), ("v_or3_b32", "or3"), ("v_readfirstlane_b32", "readfirstlane"), ("v_mbcnt_lo_u32_b32", "mbcnt"), ("v_mbcnt_hi_u32_b32", "mbcnt"),
                    ("v_sub_f32", "sub"), ("v_subrev_f32", "sub")):
        if key in pk:
            table[kk] = pk[key]
    if kind in table:
        return table[kind], "measured"
    if kind.startswith("v_cmp") or kind.startswith("v_cmpx"):
        return pk["cmp_sgpr"], "measured"
    c = IC.classify(kind)
    return {"full": full, "half": half, "trans": trans, "quarter": 2 * half, "lane": half}.get(c, half), "class:" + c


def calibration_path():
    for r in ("r03", "r02"):
        p = os.path.join(ROOT, "profiles", r, "issue_cost_calibration.json")
        if os.path.exists(p):
            return p
    raise SystemExit("no issue-cost calibration under profiles/")


def mixed_stream_factor(cal):
    """measured cost of the 16-instruction stream in the census's proportions / the sum of its kinds' costs (tools/ubench/valu_rates2.hip mix16,
    mix16_trans): how much a MIXED stream costs more than the additive per-kind pricing.  The kernel runs one transcendental per ~21 VALU
    instructions, between the two streams (none, 1 in 16): both are returned."""
    pk = cal["valu"]["per_kind_cycles"]
    if "mix16" not in pk:
        return None
    base = 3 * pk["mul"] + 2 * pk["fma_3src"] + 2 * pk["fmac"] + pk["add"] + pk.get("sub", pk["add"]) + 2 * pk["fma_mix"] + pk["and"] + pk["cvt_flr"] + pk["fract"] + pk["lshl"]
    return {"no_transcendental": pk["mix16"] / ((base + pk["mov"]) / 16.0), "one_transcendental_in_16": pk["mix16_trans"] / ((base + pk["rcp"]) / 16.0)}


# ---- the two pricings of round 4
# (1) the guide's issue rates (/opt/skills/guides/MI355X_MICROARCH.md: a SIMD is 32 lanes wide, a wave64 VALU instruction issues over 2 cycles; half-rate
#     kinds 4, transcendentals 8), per instruction CLASS: what bench.py's top-level roofline.frac uses, recomputable from the class totals in one line.
GUIDE_CYCLES = {"full": 2.0, "half": 4.0, "trans": 8.0, "quarter": 8.0, "lane": 4.0}
# (2) the TIME a kind costs, measured differentially (tools/ubench/valu_rates2 diff -> profiles/r04/valu_issue_time_gfx950.json: slope of the event-timed
#     launch time between N and 3N loop iterations at 8 waves per SIMD, so launch ramp, prologue and tail drop out).  In nanoseconds per wave64 instruction
#     per SIMD: no clock reading enters at all -- a pure stream of one kind draws enough power for the chip to run at 2.0-2.35 GHz, lower for the full-rate
#     kinds, and the hwmon clock node under-reads those states (full-rate kinds come out at 1.8-1.9 "cycles" with it) -- and the denominator of the
#     fraction is the kernel's own duration in the same unit.  An UPPER estimate of the VALU's busy share: the cloud kernel itself runs at 2.38 GHz.
def time_calibration_path():
    p = os.path.join(ROOT, "profiles", "r04", "valu_issue_time_gfx950.json")
    return p if os.path.exists(p) else None


def kind_ns(kind, tcal):
    ns = {k: v["ns_per_instr"] for k, v in tcal.items()}
    full = sum(ns[k] for k in ("v_mul_f32", "v_add_f32", "v_fmac_f32", "v_mov_b32", "v_and_b32", "v_fma_f32 3 vgpr src")) / 6.0
    half = sum(ns[k] for k in ("v_fma_mix_f32", "v_lshlrev_b32", "v_cvt_flr_i32_f32", "v_fract_f32", "v_cndmask_b32 sgpr", "v_lshl_or_b32")) / 6.0
    trans = sum(ns[k] for k in ("v_rcp_f32", "v_exp_f32", "v_log_f32", "v_sqrt_f32")) / 4.0
    table = {"v_fma_f32": ns["v_fma_f32 3 vgpr src"], "v_fmac_f32": ns["v_fmac_f32"], "v_mul_f32": ns["v_mul_f32"], "v_add_f32": ns["v_add_f32"], "v_sub_f32": ns["v_sub_f32"],
             "v_subrev_f32": ns["v_sub_f32"], "v_mov_b32": ns["v_mov_b32"], "v_and_b32": ns["v_and_b32"], "v_or_b32": ns["v_and_b32"], "v_xor_b32": ns["v_and_b32"],
             "v_add_u32": ns["v_add_u32"], "v_sub_u32": ns["v_add_u32"], "v_subrev_u32": ns["v_add_u32"], "v_fma_mix_f32": ns["v_fma_mix_f32"], "v_max_f32": ns["v_max/min_f32"],
             "v_min_f32": ns["v_max/min_f32"], "v_med3_f32": ns["v_med3_f32"], "v_lshlrev_b32": ns["v_lshlrev_b32"], "v_lshrrev_b32": ns["v_lshlrev_b32"],
             "v_lshl_or_b32": ns["v_lshl_or_b32"], "v_and_or_b32": ns["v_and_or_b32"], "v_bfe_u32": ns["v_bfe_u32"], "v_bfi_b32": ns["v_bfi_b32"], "v_lshl_add_u32": ns["v_lshl_add_u32"],
             "v_cvt_flr_i32_f32": ns["v_cvt_flr_i32_f32"], "v_fract_f32": ns["v_fract_f32"], "v_floor_f32": ns["v_floor_f32"], "v_cvt_f32_i32": ns["v_cvt_f32_i32"],
             "v_cvt_f32_u32": ns["v_cvt_f32_i32"], "v_cndmask_b32": ns["v_cndmask_b32 sgpr"], "v_rcp_f32": ns["v_rcp_f32"], "v_exp_f32": ns["v_exp_f32"], "v_log_f32": ns["v_log_f32"],
             "v_sqrt_f32": ns["v_sqrt_f32"], "v_rsq_f32": ns["v_sqrt_f32"], "v_mul_lo_u32": ns["v_mul_lo_u32"], "v_mad_u32_u24": ns["v_mad_u32_u24"], "v_fmaak_f32": ns["v_fmamk_f32"],
             "v_fmamk_f32": ns["v_fmamk_f32"], "v_add_lshl_u32": ns["v_add_lshl_u32"], "v_or3_b32": ns["v_or3_b32"], "v_readfirstlane_b32": ns["v_readfirstlane_b32"],
             "v_mbcnt_lo_u32_b32": ns["v_mbcnt_lo/hi"], "v_mbcnt_hi_u32_b32": ns["v_mbcnt_lo/hi"]}
    if kind in table:
        return table[kind], "measured"
    if kind.startswith("v_cmp"):
        return ns["v_cmp_gt_f32 -> sgpr"], "measured"
    c = IC.classify(kind)
    return {"full": full, "half": half, "trans": trans, "quarter": 2 * half, "lane": half}.get(c, half), "class:" + c


def report(d, quiet=False):
    cal = json.load(open(calibration_path()))
    tpath = time_calibration_path()
    tcal = json.load(open(tpath)) if tpath else None
    out = {"config": d["config"], "source_hash": d["source_hash"], "calibration": os.path.relpath(calibration_path(), ROOT), "mixed_stream_factor": mixed_stream_factor(cal),
           "guide_cycles_per_class": GUIDE_CYCLES, "time_calibration": os.path.relpath(tpath, ROOT) if tpath else None, "kernels": {}}
    for tag, k in d["kernels"].items():
        blocks = d["static"][tag]["blocks"]
        counts = k["counts"]
        kinds, classes = collections.Counter(), collections.Counter()
        for b, n in zip(blocks, counts):
            for kk, v in b["kinds"].items():
                kinds[kk] += v * n
            for cc, v in b["classes"].items():
                classes[cc] += v * n
        valu = {kk: v for kk, v in kinds.items() if kk.startswith("v_") and IC.classify(kk) not in ()}
        cycles, by_source = 0.0, collections.Counter()
        per_kind = []
        for kk, v in sorted(valu.items(), key=lambda x: -x[1]):
            c, src = kind_cost(kk, cal)
            cycles += c * v
            by_source["measured" if src == "measured" else "class"] += v
            per_kind.append({"kind": kk, "wave_instructions": v, "cycles_each": c, "source": src})
        n_valu = sum(valu.values())
        n_waves = counts[0]
        by_class = {c: sum(v for kk, v in valu.items() if IC.classify(kk) == c) for c in ("full", "half", "trans", "quarter", "lane")}
        guide_cycles = sum(GUIDE_CYCLES[c] * v for c, v in by_class.items())
        time_ns = sum(kind_ns(kk, tcal)[0] * v for kk, v in valu.items()) if tcal else None
        o = {"frame_identical_to_product": k["frame_identical_to_product"], "wavefronts": n_waves, "stats": k["stats"],
             "executed": {"valu": n_valu, "salu": classes["salu"], "smem": classes["smem"], "vmem_load": classes["vmem_load"], "vmem_store": classes["vmem_store"],
                          "vmem_atomic": classes["vmem_atomic"], "lds": classes["lds"], "branch": classes["branch"], "wait": classes["wait"], "scratch": classes["scratch"]},
             "valu_by_class": by_class,
             "valu_issue_cycles_guide_per_simd": guide_cycles / 1024.0,                       # sum over classes of count x (2, 4, 8) / 1024 SIMDs
             "valu_issue_time_ms_per_simd": (time_ns / 1024.0 * 1e-6) if time_ns is not None else None,   # sum over kinds of count x measured ns / 1024 SIMDs
             "valu_issue_cycles_total": cycles, "valu_issue_cycles_per_simd": cycles / 1024.0,   # round 3's per-kind cycles (SQ_BUSY_CYCLES/32 of the calibration launches): kept for comparison
             "valu_priced_by_measured_kind_fraction": by_source["measured"] / max(1, n_valu),
             "scratch_instructions_per_wavefront": classes["scratch"] / max(1, n_waves),
             "top_kinds": per_kind[:40]}
        out["kernels"][tag] = o
        if quiet:
            continue
        print("== %s: %d wavefronts, frame identical to product: %s" % (tag, n_waves, k["frame_identical_to_product"]))
        print("   executed wave-instructions: VALU %.4g (full %.4g, half %.4g, trans %.4g)  SALU %.4g  SMEM %.4g  VMEM loads %.4g  LDS %.4g  branches %.4g  waits %.4g  scratch %.4g (%.1f per wavefront)"
              % (n_valu, o["valu_by_class"]["full"], o["valu_by_class"]["half"], o["valu_by_class"]["trans"], classes["salu"], classes["smem"], classes["vmem_load"], classes["lds"],
                 classes["branch"], classes["wait"], classes["scratch"], o["scratch_instructions_per_wavefront"]))
        print("   VALU issue per SIMD: %.4g cycles at the guide's rates (full 2, half 4, transcendental 8)   %s ms at the measured per-kind times   (round 3's calibrated cycles: %.4g)"
              % (guide_cycles / 1024.0, "%.4f" % o["valu_issue_time_ms_per_simd"] if time_ns is not None else "n/a", cycles / 1024.0))
    return out


def cmd_report(args):
    out = report(json.load(open(args.counts)))
    print("mixed-stream factor (measured / additive):", out["mixed_stream_factor"])
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    b = sub.add_parser("build"); b.add_argument("--work", default=None, help="keep the intermediate files here (default: a private temporary directory, removed afterwards)")
    r = sub.add_parser("run"); r.add_argument("--config", default="C3"); r.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "census_counts.json"))
    p = sub.add_parser("report"); p.add_argument("counts"); p.add_argument("--out", default=None)
    a = ap.parse_args()
    {"build": cmd_build, "run": cmd_run, "report": cmd_report}[a.cmd](a)


if __name__ == "__main__":
    main()
