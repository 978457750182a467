"""Loader of tests/golden/glslexec.npz: what the reference's own shader TEXT wrote when it was executed on the CPU under
oracle/glsl_exec/glsl_shim.hpp (generator: oracle/glsl_exec/make_glsl_fixtures.py; build container only).  Arrays only -- no
shader text is stored.  `fold` = constant expressions folded in double (glslang's behaviour, the variant the oracle models);
`float` = every constant narrowed to fp32 at once (stored as a sparse difference from `fold`)."""
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glslexec.npz")
SKY_OF = {"cov50": "deg45", "fine": "deg45", "c3edge": "deg45", "c3mid": "deg45"}          # cases that share an earlier case's sun (and therefore its sky LUT)
# clouds.gdshader sky() over a panorama (oracle/glsl_exec/make_glsl_fixtures.py COMPOSITES): textures blended, light, blend_amount, sun_disk_scale, size
COMPOSITES = {"blend35": dict(**{"from": "zenith", "to": "deg45"}, sun="deg45", blend=0.35, disk=2.0, size=(256, 128)),
              "demo": dict(**{"from": "demo", "to": "demo"}, sun="demo", blend=0.0, disk=1.0, size=(192, 96))}


class GlslExec:
    def __init__(self):
        self.z = np.load(PATH)
        self.extra = [str(k) for k in self.z["extra_names"]]

    def fold(self, key):
        return self.z["fold_" + key].view(np.float16)

    def flt(self, key):
        a = self.z["fold_" + key].copy()
        a.reshape(-1)[self.z["float_" + key + "_idx"]] = self.z["float_" + key + "_val"]
        return a.view(np.float16)

    def cloud_cases(self, suns):
        """name -> (28-float push-constant block, dispatched rectangle, name of the sky LUT fixture it was rendered with)"""
        from oracle import oracle as O
        c = {k: (O.default_params(64, 32, s), (0, 0, 64, 32), k) for k, s in suns.items()}
        c["windy"] = (self.z["windy_params"], tuple(int(v) for v in self.z["windy_rect"]), "windy")
        for k in self.extra:
            c[k] = (self.z[k + "_params"], tuple(int(v) for v in self.z[k + "_rect"]), SKY_OF.get(k, k))
        return c
