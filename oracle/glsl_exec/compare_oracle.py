#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY.  Prints, fixture by fixture, how many RGBA16F halfs of the C oracle differ from what the reference's shader
text wrote when executed under the shim (tests/golden/glslexec.npz) -- the table behind tests/test_oracle_glslexec.py.  Each stage is fed
the fixture's own upstream output, so a difference can only come from that stage.  Runs anywhere (reads only the committed arrays)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from conftest import SUNS, norm  # noqa: E402
from glslexec_fixture import COMPOSITES, GlslExec, SKY_OF  # noqa: E402
from oracle import oracle as O  # noqa: E402
import gvcd_amd  # noqa: E402


def rep(name, a, b, extra=""):
    a = np.ascontiguousarray(a).view(np.int16).astype(np.int32)
    b = np.ascontiguousarray(b).view(np.int16).astype(np.int32)
    d = np.abs(a - b)
    print("%-16s oracle vs %-5s : %6d of %7d halfs differ, max %d fp16 ulp %s" % (name[0], name[1], (d > 0).sum(), d.size, d.max(), extra))


def main():
    gx = GlslExec()
    otex = O.OracleTextures(*gvcd_amd.assets.load_default_noise())
    t = gx.fold("trans")
    for v, get in (("fold", gx.fold), ("float", gx.flt)):
        rep(("trans", v), O.transmittance_lut(), get("trans"))
    suns = {k: norm(s) for k, s in SUNS.items()}
    suns["windy"] = gx.z["windy_params"][16:19]
    suns["below"] = norm((0.3, -0.2, 0.5))
    for k in gx.extra:
        if k not in SKY_OF:
            suns[k] = gx.z[k + "_params"][16:19]
    for k, s in suns.items():
        o = O.sky_lut(s, t)
        for v, get in (("fold", gx.fold), ("float", gx.flt)):
            rep(("sky_" + k, v), o, get("sky_" + k))
    for k, (pc, rect, sky) in gx.cloud_cases(SUNS).items():
        img, st = O.clouds(otex, pc, gx.fold("sky_" + sky), rect=rect, return_stats=True)
        for v, get in (("fold", gx.fold), ("float", gx.flt)):
            rep(("clouds_" + k, v), img, get("clouds_" + k), "(alpha mean %.3f, %d in-cloud samples)" % (img[..., 3].astype(np.float32).mean(), st["incloud_samples"]))
    for k, c in COMPOSITES.items():
        w, h = c["size"]
        o = O.composite(gx.fold("clouds_" + c["from"]), gx.fold("clouds_" + c["to"]), gx.fold("sky_" + c["from"]), gx.fold("sky_" + c["to"]), t, norm(SUNS[c["sun"]]),
                        blend_amount=c["blend"], sun_disk_scale=c["disk"], out_w=w, out_h=h)
        rep(("composite_" + k, "fold"), o, gx.fold("composite_" + k), "(clouds.gdshader sky(), %d x %d panorama)" % (w, h))


if __name__ == "__main__":
    main()
