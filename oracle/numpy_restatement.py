"""Independent numpy-fp32 restatement of the three reference shaders + fixture generator.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference (GLSL for Vulkan + GDScript) cannot be executed
and ships no golden vectors, so the C oracle (cloudsky_oracle.c) is cross-checked against THIS second,
separately written restatement (vectorised numpy, float32 arrays throughout, written from the GLSL text:
/root/reference/cloud_sky/{clouds,sky-lut,transmittance-lut}.glsl, cited C:/S:/T:line).  Agreement of two
independent restatements catches transcription slips; it cannot catch a shared misreading of the GLSL.

Run `python oracle/numpy_restatement.py` to (re)generate tests/golden/*.npz.  The script needs nothing from
/root/reference at run time: its inputs are the repo's bitmaps and the deterministic shape-noise generator.
"""
import os
import sys

import numpy as np

f32 = np.float32
F = lambda x: np.asarray(x, dtype=np.float32)


# ----------------------------------------------------------------------------- GLSL builtins on fp32 arrays
def clamp(x, lo, hi):
    return np.minimum(np.maximum(x, f32(lo)), f32(hi))


def mix(a, b, t):
    return a * (f32(1.0) - t) + b * t


def smoothstep(e0, e1, x):
    t = clamp((x - e0) / (e1 - e0), 0.0, 1.0)
    return t * t * (f32(3.0) - f32(2.0) * t)


def fract(x):
    return x - np.floor(x)


def length(v):  # v: [...,3]
    return np.sqrt(v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1] + v[..., 2] * v[..., 2])


def dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def normalize(v):
    return v / length(v)[..., None]


# ----------------------------------------------------------------------------- samplers (Vulkan spec rules)
def mip_chain(level0):
    """2x2x2 box mips, (sum+4)>>3.  level0: uint8 [n,n,n,ch].  Returns list of levels."""
    levels = [np.ascontiguousarray(level0, np.uint8)]
    while levels[-1].shape[0] > 1:
        s = levels[-1].astype(np.uint32)
        n = s.shape[0] // 2
        s = s.reshape(n, 2, n, 2, n, 2, -1).sum(axis=(1, 3, 5))
        levels.append(((s + 4) >> 3).astype(np.uint8))
    return levels


def tex3d_repeat(levels, s, lod):
    """REPEAT/LINEAR 3-D tap at integer LOD (clamped).  s: [...,3] fp32 normalised coords -> [...,ch] fp32."""
    lod = int(min(max(lod, 0), len(levels) - 1))
    t = levels[lod]
    n = t.shape[0]
    u = s * f32(n) - f32(0.5)
    i0f = np.floor(u)
    a = (u - i0f).astype(f32)
    i0 = np.mod(i0f.astype(np.int64), n)
    i1 = np.mod(i0 + 1, n)
    tf = t.astype(f32) / f32(255.0)

    def tx(ix, iy, iz):
        return tf[iz, iy, ix]  # volume index [z][y][x]

    x0, y0, z0 = i0[..., 0], i0[..., 1], i0[..., 2]
    x1, y1, z1 = i1[..., 0], i1[..., 1], i1[..., 2]
    ax, ay, az = a[..., 0:1], a[..., 1:2], a[..., 2:3]
    lerp = lambda p, q, w: p + (q - p) * w
    c00 = lerp(tx(x0, y0, z0), tx(x1, y0, z0), ax)
    c10 = lerp(tx(x0, y1, z0), tx(x1, y1, z0), ax)
    c01 = lerp(tx(x0, y0, z1), tx(x1, y0, z1), ax)
    c11 = lerp(tx(x0, y1, z1), tx(x1, y1, z1), ax)
    return lerp(lerp(c00, c10, ay), lerp(c01, c11, ay), az)


def tex2d(t, s, repeat):
    """LINEAR 2-D tap; t: float32 [h,w,ch]; s: [...,2] (x,y) normalised."""
    h, w = t.shape[0], t.shape[1]
    u = s * F([w, h]) - f32(0.5)
    i0f = np.floor(u)
    a = (u - i0f).astype(f32)
    i0 = i0f.astype(np.int64)
    i1 = i0 + 1
    if repeat:
        x0, x1 = np.mod(i0[..., 0], w), np.mod(i1[..., 0], w)
        y0, y1 = np.mod(i0[..., 1], h), np.mod(i1[..., 1], h)
    else:
        x0, x1 = np.clip(i0[..., 0], 0, w - 1), np.clip(i1[..., 0], 0, w - 1)
        y0, y1 = np.clip(i0[..., 1], 0, h - 1), np.clip(i1[..., 1], 0, h - 1)
    ax, ay = a[..., 0:1], a[..., 1:2]
    lerp = lambda p, q, w_: p + (q - p) * w_
    return lerp(lerp(t[y0, x0], t[y0, x1], ax), lerp(t[y1, x0], t[y1, x1], ax), ay)


# ============================================================================= atmosphere (T:45-145 == S:44-202)
EARTH_RADIUS = f32(6371.0)
ATMOSPHERE_THICKNESS = f32(100.0)
ATMOSPHERE_RADIUS = f32(6471.0)
SUN_IRR = F([1.679, 1.828, 1.986, 1.307])
MOL_SCAT_BASE = F([6.605e-3, 1.067e-2, 1.842e-2, 3.156e-2])
OZONE_X_DOBSON = F(np.array([3.472e-21, 3.914e-21, 1.349e-21, 11.03e-23]) * 1e-4 * 350.0)
AER_ABS = F([2.8722e-24, 4.6168e-24, 7.9706e-24, 1.3578e-23])
AER_SCAT = F([1.5908e-22, 1.7711e-22, 2.0942e-22, 2.4033e-22])
AER_BASE = f32(1.3681e20)
AER_BG_DIV_BASE = f32(2e6 / 1.3681e20)
AER_HSCALE = f32(0.73)


def ray_sphere_intersection(ro, rd, radius):  # T:89-98
    b = dot(ro, rd)
    c = dot(ro, ro) - radius * radius
    d = b * b - c
    with np.errstate(invalid="ignore"):
        sq = np.sqrt(np.maximum(d, f32(0)))
    res = np.where(d > b * b, -b + sq, -b - sq)
    res = np.where(d < 0, f32(-1.0), res)
    res = np.where((c > 0) & (b > 0), f32(-1.0), res)
    return res.astype(f32)


def collision_coefficients(h):  # T:131-145; returns (aer_scat, mol_scat, extinction) as [...,4]
    h = np.maximum(h, f32(0.0))
    aer_density = AER_BASE * (np.exp(-h / AER_HSCALE) + AER_BG_DIV_BASE)
    aa = AER_ABS * aer_density[..., None]
    asc = AER_SCAT * aer_density[..., None]
    h2 = h + f32(1e-4)
    t = np.log(h2) - f32(3.22261)
    dens = f32(3.78547397e20) * (f32(1.0) / h2) * np.exp(-t * t * f32(5.55555555))
    ma = OZONE_X_DOBSON * dens[..., None]
    ms = MOL_SCAT_BASE * np.exp(f32(-0.07771971) * np.power(h, f32(1.16364243)))[..., None]
    ext = aa + asc + ma + ms
    return asc.astype(f32), ms.astype(f32), ext.astype(f32)


def transmittance_lut(w=256, h=64):  # T:157-196
    px, py = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32))
    uvx, uvy = px / f32(w), py / f32(h)
    c = uvx * f32(2.0) - f32(1.0)
    sun_dir = np.stack([-np.sqrt(f32(1.0) - c * c), np.zeros_like(c), c], -1)
    d = mix(EARTH_RADIUS, ATMOSPHERE_RADIUS, uvy)
    ro = np.stack([np.zeros_like(d), np.zeros_like(d), d], -1)
    t_d = ray_sphere_intersection(ro, sun_dir, ATMOSPHERE_RADIUS)
    dt = t_d / f32(40.0)
    result = np.zeros((h, w, 4), f32)
    for i in range(40):
        t = (f32(i) + f32(0.5)) * dt
        x_t = ro + sun_dir * t[..., None]
        alt = length(x_t) - EARTH_RADIUS
        _, _, ext = collision_coefficients(alt)
        result = result + ext * dt[..., None]
    return np.exp(-result).astype(np.float16)


S_PI = 3.14159265358979323846
M = F([[137.672389239975, -8.632904716299537, -1.7181567391931372], [32.549094028629234, 91.29801417199785, -12.005406444382531],
       [-38.91428392614275, 34.31665471469816, 29.89044807197628], [8.572844237945445, -11.103384660054624, 117.47585277566478]])


def sky_lut(sun, trans, w=200, h=100):  # S:278-315
    T = np.asarray(trans).astype(f32)
    sun = F(sun)
    px, py = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32))
    uvx, uvy = px / f32(w), py / f32(h)
    az = f32(2.0 * S_PI) * uvx
    l = uvy * f32(2.0) - f32(1.0)
    elev = l * l * np.sign(l) * f32(S_PI) * f32(0.5)
    rd = np.stack([np.cos(elev) * np.cos(az), np.cos(elev) * np.sin(az), np.sin(elev)], -1).astype(f32)
    ro = np.broadcast_to(F([0, 0, 6371.5]), rd.shape)
    atmos = ray_sphere_intersection(ro, rd, ATMOSPHERE_RADIUS)
    ground = ray_sphere_intersection(ro, rd, EARTH_RADIUS)
    t_d = np.where(ground < 0, atmos, ground)
    sd = F([-sun[0], -sun[2], sun[1]])  # S:221-223
    cos_theta = dot(-rd, sd)
    mol_phase = f32((3.0 / 16.0) / S_PI) * (f32(1.0) + cos_theta * cos_theta)
    den = f32(1.0 + 0.64) + f32(1.6) * cos_theta
    aer_phase = f32(0.25 / S_PI) * (f32(1.0) - f32(0.64)) / (den * np.sqrt(den))
    dt = t_d / f32(30.0)
    L = np.zeros((h, w, 4), f32)
    Tr = np.ones((h, w, 4), f32)

    def tlut(c, hn):
        u = clamp(c * f32(0.5) + f32(0.5), 0, 1)
        v = clamp(hn, 0, 1)
        return tex2d(T, np.stack([u, v], -1), repeat=False)

    for i in range(30):
        t = (f32(i) + f32(0.5)) * dt
        x_t = ro + rd * t[..., None]
        d = length(x_t)
        zen = x_t / d[..., None]
        alt = d - EARTH_RADIUS
        nalt = alt / ATMOSPHERE_THICKNESS
        sc = dot(zen, sd)
        asc, msc, ext = collision_coefficients(alt)
        t_sun = tlut(sc, nalt)
        # S:144-164
        omega = f32(2.0 * S_PI) * (f32(1.0) - np.sqrt(d * d - EARTH_RADIUS * EARTH_RADIUS) / d)
        T_to_ground = tlut(sc, np.zeros_like(sc))
        one = np.ones_like(sc)
        T_g2s = tlut(one, np.zeros_like(sc)) / tlut(one, nalt)
        L_ground = (f32(0.25 / S_PI) * omega * f32(0.3 / S_PI))[..., None] * T_to_ground * T_g2s * sc[..., None]
        L_ms = F([0.02 * 0.217, 0.02 * 0.347, 0.02 * 0.594, 0.02]) * (f32(1.0) / (f32(1.0) + f32(5.0) * np.exp(f32(-17.92) * sc)))[..., None]
        ms = L_ms + L_ground
        S = SUN_IRR * (msc * (mol_phase[..., None] * t_sun + ms) + asc * (aer_phase[..., None] * t_sun + ms))
        stepT = np.exp(-dt[..., None] * ext)
        S_int = (S - S * stepT) / np.maximum(ext, f32(1e-7))
        L = L + Tr * S_int
        Tr = Tr * stepT
    rgb = M[0] * L[..., 0:1] + M[1] * L[..., 1:2] + M[2] * L[..., 2:3] + M[3] * L[..., 3:4]
    out = np.concatenate([rgb, np.ones((h, w, 1), f32)], -1)
    return out.astype(np.float16)


# ============================================================================= clouds.glsl
C_PI = f32(3.141592)
RANDOM_VECTORS = F([[0.38051305, 0.92453449, -0.02111345], [-0.50625799, -0.03590792, -0.86163418],
                    [-0.32509218, -0.94557439, 0.01428793], [0.09026238, -0.27376545, 0.95755165],
                    [0.28128598, 0.42443639, -0.86065785], [-0.16852403, 0.14748697, 0.97460106]])


class CloudInputs:
    def __init__(self, large_rgba8, small_rgb8, weather_rgb8):
        self.large = mip_chain(np.asarray(large_rgba8).reshape(128, 128, 128, 4))
        self.small = mip_chain(np.asarray(small_rgb8).reshape(32, 32, 32, 3))
        self.weather = np.asarray(weather_rgb8).reshape(512, 512, 3).astype(f32) / f32(255.0)


def sky_lookup(sky, d):  # C:49-57; d: [3]
    d = F(d)
    phi = np.arctan2(d[2], d[0]).astype(f32)
    theta = np.arcsin(d[1]).astype(f32)
    u = phi / C_PI * f32(0.5) + f32(0.5)
    v = np.sqrt(np.abs(theta) / (C_PI * f32(0.5))) * np.sign(theta) * f32(0.5) + f32(0.5)
    return tex2d(np.asarray(sky).astype(f32), F([[u, v]]), repeat=False)[0, :3]


def height_fraction(r):  # C:77-80
    return clamp((r - f32(6001500.0)) / (f32(6004000.0) - f32(6001500.0)), 0, 1)


def density_height_gradient(hf, ctype):  # C:82-95
    ST, SC, CU = F([0.02, 0.05, 0.09, 0.11]), F([0.02, 0.2, 0.48, 0.625]), F([0.01, 0.0625, 0.78, 1.0])
    stratus = f32(1.0) - clamp(ctype * f32(2.0), 0, 1)
    stratocu = f32(1.0) - np.abs(ctype - f32(0.5)) * f32(2.0)
    cumulus = clamp(ctype - f32(0.5), 0, 1) * f32(2.0)
    g = ST * stratus[..., None] + SC * stratocu[..., None] + CU * cumulus[..., None]
    return smoothstep(g[..., 0], g[..., 1], hf) - smoothstep(g[..., 2], g[..., 3], hf)


def remap(v, omin, omax, nmin, nmax):  # C:67-69
    return nmin + (((v - omin) / (omax - omin)) * (nmax - nmin))


def density(inp, P, p, weather, mip):  # C:109-137; p [n,3], weather [n,3]
    p = p.copy()
    hf = height_fraction(length(p))
    p[:, 0] += f32(20.0) * P["cloud_pos"][0] * f32(0.6)
    p[:, 2] += f32(20.0) * P["cloud_pos"][1] * f32(0.6)
    n = tex3d_repeat(inp.large, p * f32(0.00008), mip - 2)
    fbm = n[:, 1] * f32(0.625) + n[:, 2] * f32(0.25) + n[:, 3] * f32(0.125)
    g = density_height_gradient(hf, weather[:, 0])
    with np.errstate(divide="ignore", invalid="ignore"):
        base = remap(n[:, 0], -(f32(1.0) - fbm), f32(1.0), f32(0.0), f32(1.0))
        wc = P["cloud_coverage"] * weather[:, 2]
        base = remap(base * g, f32(1.0) - wc, f32(1.0), f32(0.0), f32(1.0))
        base = base * wc
        p[:, 0] -= P["detailed_pos"][0] * f32(40.0)
        p[:, 2] -= P["detailed_pos"][1] * f32(40.0)
        p[:, 1] -= P["time"] * f32(40.0)
        hn = tex3d_repeat(inp.small, p * f32(0.001), mip)
        hfbm = hn[:, 0] * f32(0.625) + hn[:, 1] * f32(0.25) + hn[:, 2] * f32(0.125)
        hfbm = mix(hfbm, f32(1.0) - hfbm, clamp(hf * f32(4.0), 0, 1))
        base = remap(base, hfbm * f32(0.4) * hf, f32(1.0), f32(0.0), f32(1.0))
        base = np.where(np.isnan(base), f32(0.0), base)  # GLSL clamp(NaN) is undefined; defined as 0
        return np.power(clamp(base, 0, 1), (f32(1.0) - hf) * f32(0.8) + f32(0.5)).astype(f32)


def hg(c, g):  # C:72-75
    g = f32(g)
    return f32(0.0795774715459) * (f32(1.0) - g * g) / np.power(f32(1.0) + g * g - f32(2.0) * g * c, f32(1.5))


def unpack_params(params):
    p = F(params)
    return dict(texture_size=p[0:2], update_position=p[2:4], cloud_pos=p[4:6], detailed_pos=p[6:8], weather_pos=p[8:10],
                ground_color=p[12:16], LIGHT_DIRECTION=p[16:19], LIGHT_ENERGY=p[19], LIGHT_COLOR=p[20:23], time=p[23],
                density=p[25], cloud_coverage=p[26], time_offset=p[27])


def weather_tap(inp, p, wpos):
    s = np.stack([p[:, 0] * f32(0.00006) + f32(0.5) + wpos[0], p[:, 2] * f32(0.00006) + f32(0.5) + wpos[1]], -1)
    return tex2d(inp.weather, s, repeat=True)


def clouds(inp, params, sky, rect=None, primary_steps=128, light_steps=6):  # C:258-266
    P = unpack_params(params)
    W, H = int(P["texture_size"][0]), int(P["texture_size"][1])
    gx0, gy0, w, h = rect if rect is not None else (0, 0, W, H)
    gx, gy = np.meshgrid(np.arange(gx0, gx0 + w), np.arange(gy0, gy0 + h))
    px = (gx + int(P["update_position"][0])).astype(f32).reshape(-1)
    py = (gy + int(P["update_position"][1])).astype(f32).reshape(-1)
    ex, ey = px / P["texture_size"][0], py / P["texture_size"][1]
    nx = ex - ey
    ny = (ex + ey) - f32(1.0)
    nz = f32(1.0) - np.abs(nx) - np.abs(ny)
    assert (nz >= 0).all()  # oct_wrap (C:239-244) is dead for uv in [0,1)^2
    nrm = normalize(np.stack([nx, ny, nz], -1))
    dirs = np.stack([nrm[:, 0], nrm[:, 2], nrm[:, 1]], -1)  # .xzy
    out = np.zeros((h * w, 4), f32)
    up = np.where(dirs[:, 1] > 0)[0]
    d = dirs[up]
    cam = F([0.0, 6000000.0, 0.0])

    def isect(r):  # C:97-105
        a = dot(d, d)
        b = f32(2.0) * dot(d, cam[None, :])
        c = dot(cam, cam) - (f32(r) * f32(r))
        dd = np.sqrt((b * b) - f32(4.0) * a * c)
        return np.maximum(-b - dd, -b + dd) / (f32(2.0) * a)

    start = cam + d * isect(6001500.0)[:, None]
    end = cam + d * isect(6004000.0)[:, None]
    shelldist = length(end - start)
    raystep = d * shelldist[:, None] / f32(primary_steps)
    ss = length(raystep)
    dn = raystep / ss[:, None]
    # C:60-64,145 hash(pos*10)
    hp = fract(start * f32(10.0) * f32(0.3183099) + f32(0.1)) * f32(17.0)
    hsh = fract(hp[:, 0] * hp[:, 1] * hp[:, 2] * (hp[:, 0] + hp[:, 1] + hp[:, 2]))
    p = start + dn * hsh[:, None] * ss[:, None]
    lss = f32(2500.0 / 64.0)
    ldir = normalize(P["LIGHT_DIRECTION"][None, :])[0]
    costheta = dot(dn, ldir[None, :])
    phase = np.maximum(np.maximum(hg(costheta, 0.6), hg(costheta, f32(0.4) - f32(1.4) * ldir[1])), hg(costheta, -0.2))
    sun_c = sky_lookup(sky, P["LIGHT_DIRECTION"]) * f32(0.1) * P["LIGHT_ENERGY"] * P["LIGHT_COLOR"]
    amb = sky_lookup(sky, normalize(F([[1.0, 1.0, 0.0]]))[0]) * f32(0.05)
    amb = mix(amb, np.full(3, length(amb), f32), f32(0.5))
    gnd = sky_lookup(sky, normalize(F([[1.0, -1.0, 0.0]]))[0]) * f32(5.0) * f32(0.05)
    gnd = mix(gnd, P["ground_color"][:3] * np.full(3, length(gnd), f32), f32(0.5))
    n = len(up)
    T = np.ones(n, f32)
    alpha = np.zeros(n, f32)
    L = np.zeros((n, 3), f32)
    incloud = 0
    stepv = dn * ss[:, None]
    for i in range(primary_steps):
        p = p + stepv
        wsmp = weather_tap(inp, p, P["weather_pos"])
        hf = height_fraction(length(p))
        t = density(inp, P, p, wsmp, 0)
        dt = np.exp(-P["density"] * t * ss)
        m = np.where(t > 0)[0]
        if len(m) == 0:
            continue
        incloud += len(m)
        pm = p[m]
        lp = pm.copy()
        cd = np.zeros(len(m), f32)
        for j in range(light_steps):
            lp = lp + (ldir + RANDOM_VECTORS[j] * f32(j)) * lss
            lw = weather_tap(inp, lp, P["weather_pos"])
            cd = cd + density(inp, P, lp, lw, j)
        lp = pm + ldir * f32(18.0) * lss
        lhf = height_fraction(length(lp))
        lw = weather_tap(inp, lp, F([0, 0]))  # C:197: no weather_pos
        cd = cd + np.power(density(inp, P, lp, lw, 5), (f32(1.0) - lhf) * f32(0.8) + f32(0.5))
        beers = np.exp(-P["density"] * cd * lss * f32(3.0))
        powder = f32(1.0) - np.exp(-P["density"] * cd * lss * f32(3.0) * f32(2.0))
        bt = f32(2.0) * beers * powder
        ambient = mix(gnd[None, :], amb[None, :], smoothstep(f32(0.0), f32(1.0), hf[m])[:, None])
        alpha[m] = alpha[m] + (f32(1.0) - dt[m]) * (f32(1.0) - alpha[m])
        rad = (ambient + bt[:, None] * sun_c[None, :] * phase[m][:, None]) * t[m][:, None]
        L[m] = L[m] + T[m][:, None] * (rad - rad * dt[m][:, None]) / np.maximum(f32(1e-7), t[m])[:, None]
        T[m] = T[m] * dt[m]
    out[up, :3] = L
    out[up, 3] = clamp(alpha, 0, 1)
    return out.reshape(h, w, 4).astype(np.float16), dict(incloud_samples=incloud, primary_samples=n * primary_steps, hash_max=float(hsh.max()) if n else 0.0)


def default_params(w, h, sun, coverage=0.2, density=0.05):
    s = np.asarray(sun, np.float64)
    s = (s / np.linalg.norm(s)).astype(np.float32)
    return np.array([w, h, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0,
                     1.0, 0.0, 0.0, density, coverage, 0.0], np.float32)


SUNS = {"zenith": (0.0, 1.0, 0.0), "deg45": (1.0, 1.0, 0.0), "demo": (-0.998773, 0.0495291, 2.69869e-07)}


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import gvcd_amd  # product asset layer only (bitmaps + deterministic noise generator): inputs, not results

    large, small, weather = gvcd_amd.assets.load_default_noise()
    inp = CloudInputs(large, small, weather)
    gold = os.path.join(root, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    tr = transmittance_lut()
    np.savez_compressed(os.path.join(gold, "transmittance_lut_np.npz"), lut=tr.view(np.uint16))
    skies = {}
    for name, sun in SUNS.items():
        s = np.asarray(sun, np.float64)
        s = (s / np.linalg.norm(s)).astype(np.float32)
        skies[name] = sky_lut(s, tr)
    np.savez_compressed(os.path.join(gold, "sky_lut_np.npz"), **{k: v.view(np.uint16) for k, v in skies.items()})
    cl = {}
    for name, sun in SUNS.items():
        img, st = clouds(inp, default_params(64, 32, sun), skies[name])
        cl[name] = img.view(np.uint16)
        cl[name + "_incloud"] = np.int64(st["incloud_samples"])
        print(name, st, float(img[..., 3].astype(np.float32).mean()))
    # one windy / offset-tile / reduced-step case: exercises every push-constant field (C:18-40)
    pw = default_params(96, 48, (0.3, 0.6, -0.2), coverage=0.3, density=0.08)
    pw[2:4] = (16, 8)
    pw[4:6] = (3.5, -1.25)
    pw[6:8] = (0.75, 2.5)
    pw[8:10] = (0.031, -0.017)
    pw[19] = 1.7
    pw[20:23] = (1.0, 0.9, 0.8)
    pw[23] = 2.25
    sun_w = pw[16:19]
    sk_w = sky_lut(sun_w, tr)
    img, st = clouds(inp, pw, sk_w, rect=(8, 4, 48, 24), primary_steps=64, light_steps=4)
    cl["windy"] = img.view(np.uint16)
    cl["windy_params"] = pw
    cl["windy_sky"] = sk_w.view(np.uint16)
    print("windy", st)
    np.savez_compressed(os.path.join(gold, "clouds_np.npz"), **cl)
    import hashlib

    with open(os.path.join(gold, "INPUTS.txt"), "w") as f:
        f.write("fixtures generated by oracle/numpy_restatement.py (independent numpy-fp32 restatement; NOT reference output)\n")
        f.write("shape noise: csky_generate_shape_noise(seed=1, n=128) sha256=%s\n" % hashlib.sha256(large.tobytes()).hexdigest())
        f.write("worlnoise.bmp volume sha256=%s\nweather.bmp rgb sha256=%s\n" % (hashlib.sha256(small.tobytes()).hexdigest(), hashlib.sha256(weather.tobytes()).hexdigest()))


if __name__ == "__main__":
    main()
