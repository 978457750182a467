"""The ISA census tooling (tools/isa_census.py, tools/isa_profile.py) on the product assembly: hipcc cross-compiles without a GPU.
Guards two facts DESIGN.md §5 states: the cloud kernels' scratch (spill) accesses sit outside the sampling loops of the march, and every basic block of both
kernels has a counter in the census build (<= 256 blocks)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def product_asm(tmp_path_factory):
    import isa_census as IC
    return IC.build_asm(str(tmp_path_factory.mktemp("isa")))


def test_scratch_accesses_sit_outside_the_march_loops(product_asm):
    import isa_census as IC
    for kernel, max_depth in (("_ZN4csky13clouds_kernelILi3ELi1ENS_6TexSetE", 0), ("_ZN4csky24clouds_kernel_persistentILi3E", 2)):
        name, blocks = IC.parse_kernel(product_asm, kernel)
        cen, loops = IC.census(blocks)
        assert len(cen) > 100 and sum(b["n"] for b in cen) > 1500, name
        deepest = max(b["depth"] for b in cen)
        assert deepest >= 3                                               # the march loops are there (primary, flush, light march, replay)
        scratch_blocks = [b for b in cen if b["classes"].get("scratch")]
        # Round 4: the plain kernel spills NOTHING (69 VGPRs: the per-ray constants and the running state of the ray rest in LDS between flushes);
        # the persistent form keeps a few spills per TILE (the outer pop loops, depth <= 2), never inside the march
        if max_depth == 0:
            assert not scratch_blocks, (name, [(b["depth"], b["classes"]) for b in scratch_blocks])
        inside = [b for b in scratch_blocks if b["depth"] > max_depth]
        assert not inside, (name, [(b["depth"], b["classes"]) for b in inside])
        assert sum(b["classes"].get("scratch", 0) for b in cen) <= 24


def test_every_basic_block_gets_a_counter(product_asm):
    import isa_census as IC
    import isa_profile as IP
    lines = open(product_asm).read().split("\n")
    for tag, pre in IP.KERNELS.items():
        out, nb = IP.instrument(list(lines), pre)
        name, blocks = IC.parse_kernel(product_asm, pre)
        assert nb == len(IC.census(blocks)[0]) <= 64 * IP.N_CTR_VGPR, (tag, nb)
        body = "\n".join(out)
        assert body.count("global_atomic_add v") >= IP.N_CTR_VGPR            # the flush in front of every s_endpgm
        offs = IP.explicit_arg_offsets(lines, pre)
        assert offs[-1] - offs[-2] == 8 and len(offs) in (7, 9)             # (..., stats, wg_cost) close the explicit arguments


def test_classification_covers_the_kinds_the_kernel_executes(product_asm):
    import isa_census as IC
    name, blocks = IC.parse_kernel(product_asm, "_ZN4csky13clouds_kernelILi3ELi1ENS_6TexSetE")
    kinds = set()
    for b in IC.census(blocks)[0]:
        kinds.update(k for k in b["kinds"] if k.startswith("v_"))
    assert IC.classify("v_fma_mix_f32") == "half" and IC.classify("v_rcp_f32") == "trans" and IC.classify("v_fmac_f32") == "full"
    assert {"v_fma_mix_f32", "v_cvt_flr_i32_f32", "v_fract_f32", "v_rsq_f32"} <= kinds       # the layouts' and the exact sqrt's instructions are in the binary
    assert not [k for k in kinds if k.startswith("v_pk_")], "the SLP vectoriser must stay off (Makefile: -fno-slp-vectorize)"
