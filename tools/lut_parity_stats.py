import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import gvcd_amd
from oracle import oracle as O
from conftest import ulp_diff, norm
ctx=gvcd_amd.Context(0)
tr=ctx.render_transmittance(256,64); tro=O.transmittance_lut(256,64)
d=ulp_diff(tr,tro); print("transmittance max ulp",d.max(),(d==0).mean())
worst=0
for th in list(np.linspace(-20,200,23))+[45.,90.]:
    for z in (0.0,0.1,-0.4):
        sun=norm((np.cos(np.radians(th)),np.sin(np.radians(th)),z))
        d=ulp_diff(ctx.render_sky_lut(sun,200,100),O.sky_lut(sun,tro))
        worst=max(worst,int(d.max()))
        if d.max()>1: print(th,z,d.max(),(d<=1).mean(),(d==0).mean())
print("sky worst",worst)
