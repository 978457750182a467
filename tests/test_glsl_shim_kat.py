"""Known-answer tests of oracle/glsl_exec/glsl_shim.hpp against the GLSL specification (oracle/glsl_exec/shim_kat.cpp): the stand-in the reference's shader text
is executed under must itself mean what GLSL means -- constructors, swizzle reads and writes, column-major matrices, the built-ins' definitions, texel-centre
sampling with CLAMP_TO_EDGE / REPEAT / LOD clamping, discarded out-of-image stores, and the constant-folding model in both variants.  The GLSL part of that file
goes through the same mechanical rewrites as the reference's text (make_glsl_fixtures.REWRITES), so those are under test too.  Needs g++ only."""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GX = os.path.join(ROOT, "oracle", "glsl_exec")


def _rewrites():
    spec = importlib.util.spec_from_file_location("make_glsl_fixtures", os.path.join(GX, "make_glsl_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("variant", ["fold", "float"])
def test_shim_means_what_glsl_means(oracle, tmp_path, variant):
    m = _rewrites()
    src = open(os.path.join(GX, "shim_kat.cpp")).read()
    a, b = src.index("/* GLSL-BEGIN"), src.index("/* GLSL-END */")
    glsl, counts = m.rewrite(src[a:b])
    assert dict(counts)["out-parameters become references"] == 2 and dict(counts)["strip the float-literal suffix"] > 50
    tu = tmp_path / "kat.cpp"
    tu.write_text(src[:a] + glsl + src[b:])
    exe = str(tmp_path / "kat")
    flags = ["-std=c++17", "-O1", "-ffp-contract=off", "-fno-fast-math", "-I", GX, "-Wall", "-Wno-unused-variable", "-Wno-unused-function"]
    flags += ["-DGX_FOLD_DOUBLE=1"] if variant == "fold" else ["-DGX_FOLD_DOUBLE=0", "-fsingle-precision-constant"]
    subprocess.check_call(["g++"] + flags + ["-o", exe, str(tu), os.path.join(ROOT, "oracle", "libcskoracle.so"), "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "shim KAT ok" in r.stdout, r.stdout[-3000:]
