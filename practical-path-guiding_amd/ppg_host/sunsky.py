"""`sunsky` emitter → latitude-longitude radiance map, at scene-conversion time.

Mitsuba's sunsky plug-in (emitters/sunsky.cpp:105-235) has no run-time behaviour of its own: its constructor rasterises the
Hosek-Wilkie sky (emitters/sky.cpp:386-447, sunsky/skymodel.cpp) into a resolution x resolution/2 bitmap, splats the sun's disc
(Preetham's spectral sun radiance, sunsky/sunmodel.h:255-365) on top with a few thousand (0,2)-sequence samples
(sunsky.cpp:170-205) and instantiates an `envmap` emitter on that bitmap.  `envmap` is part of the hot path's boundary
(ppg_scene.envmap), so the plug-in is restated here as the same bake, done by the scene loader.

The model's coefficient tables — datasetRGB1..3 / datasetRGBRad1..3 (sunsky/skymodeldata.h) and the k_o / k_g / k_wa / solar
tables of sunmodel.h — are NOT copied into this repository: they are parsed from the operator's Mitsuba source tree when a scene
with a sunsky emitter is converted (like data/ior and data/microfacet, which are read from the same tree).

All of this is double / float arithmetic on a 512 x 256 grid, once per scene; nothing of it runs on the GPU.
"""
import math
import os
import re

import numpy as np

from . import spectrum

SUN_APP_RADIUS = 0.5358  # sunsky.cpp:34


def parse_c_arrays(path, names):
    """`double name[] = { ... };` / `Float name[N] = { ... };` initialisers of a C source file → {name: float64 array}."""
    text = open(path, errors="replace").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for name in names:
        m = re.search(r"\b(?:double|Float|float)\s+%s\s*\[\s*\d*\s*\]\s*=\s*\{(.*?)\}\s*;" % re.escape(name), text, flags=re.S)
        if not m:
            raise ValueError("%s: array %s not found" % (path, name))
        out[name] = np.array([float(t) for t in m.group(1).replace("\n", " ").split(",") if t.strip()], np.float64)
    return out


def sun_coordinates(p):
    """computeSunCoordinates(props), sunmodel.h:217-251 → (elevation = zenith angle, azimuth) as float32-rounded floats.
    The date / time / place branch is the PSA algorithm (Blanco-Muriel et al.), sunmodel.h:103-205, in double."""
    if "sunDirection" in p:
        raise NotImplementedError("sunsky: sunDirection")
    lat, lon, tz = float(np.float32(p.get("latitude", 35.6894))), float(np.float32(p.get("longitude", 139.6917))), float(np.float32(p.get("timezone", 9)))
    year, month, day = int(p.get("year", 2010)), int(p.get("month", 7)), int(p.get("day", 10))
    hour, minute, second = float(np.float32(p.get("hour", 15.0))), float(np.float32(p.get("minute", 0.0))), float(np.float32(p.get("second", 0.0)))
    dec_hours = hour - tz + (minute + second / 60.0) / 60.0

    def cdiv(a, b):  # C integer division truncates towards zero
        return int(a / b) if a * b < 0 else a // b
    aux1 = cdiv(month - 14, 12)
    aux2 = cdiv(1461 * (year + 4800 + aux1), 4) + cdiv(367 * (month - 2 - 12 * aux1), 12) - cdiv(3 * cdiv(year + 4900 + aux1, 100), 4) + day - 32075
    elapsed = float(aux2) - 0.5 + dec_hours / 24.0 - 2451545.0
    omega = 2.1429 - 0.0010394594 * elapsed
    mean_longitude = 4.8950630 + 0.017202791698 * elapsed
    anomaly = 6.2400600 + 0.0172019699 * elapsed
    ecl_long = mean_longitude + 0.03341607 * math.sin(anomaly) + 0.00034894 * math.sin(2 * anomaly) - 0.0001134 - 0.0000203 * math.sin(omega)
    ecl_obl = 0.4090928 - 6.2140e-9 * elapsed + 0.0000396 * math.cos(omega)
    sin_el = math.sin(ecl_long)
    ra = math.atan2(math.cos(ecl_obl) * sin_el, math.cos(ecl_long))
    if ra < 0:
        ra += 2 * math.pi
    decl = math.asin(math.sin(ecl_obl) * sin_el)
    gmst = 6.6974243242 + 0.0657098283 * elapsed + dec_hours
    lmst = float(np.float32(math.radians(float(np.float32(gmst * 15 + lon)))))  # degToRad((Float) ...)
    lat_r = float(np.float32(math.radians(lat)))
    hour_angle = lmst - ra
    elevation = math.acos(math.cos(lat_r) * math.cos(hour_angle) * math.cos(decl) + math.sin(decl) * math.sin(lat_r))
    azimuth = math.atan2(-math.sin(hour_angle), math.tan(decl) * math.cos(lat_r) - math.sin(lat_r) * math.cos(hour_angle))
    if azimuth < 0:
        azimuth += 2 * math.pi
    elevation += (6371.01 / 149597890) * math.sin(elevation)  # parallax
    return float(np.float32(elevation)), float(np.float32(azimuth))


def _bezier5(t, m):  # the quintic Bezier curve over the six elevation control points (skymodel.cpp:106-113)
    s = 1.0 - t
    return s ** 5 * m[0] + 5 * s ** 4 * t * m[1] + 10 * s ** 3 * t ** 2 * m[2] + 10 * s ** 2 * t ** 3 * m[3] + 5 * s * t ** 4 * m[4] + t ** 5 * m[5]


def hosek_rgb_state(tables, turbidity, albedo, solar_elevation):
    """arhosek_rgb_skymodelstate_alloc_init (skymodel.cpp:346-374): per channel the nine distribution parameters
    (ArHosekSkyModel_CookConfiguration, :80-161) and the radiance scale (CookRadianceConfiguration, :163-224) — quintic Bezier in
    (elevation / 90 deg)^(1/3), linear in turbidity and albedo."""
    it = int(turbidity)
    assert 1 <= it <= 10, "turbidity must be in [1, 10]"
    rem = turbidity - it
    t = (solar_elevation / (math.pi / 2.0)) ** (1.0 / 3.0)
    configs, radiances = np.zeros((3, 9)), np.zeros(3)
    for ch in range(3):
        ds, rad = tables["datasetRGB%d" % (ch + 1)], tables["datasetRGBRad%d" % (ch + 1)]
        for alb, w_alb in ((0, 1.0 - albedo), (1, albedo)):
            for turb, w_t in ((it - 1, 1.0 - rem), (it, rem)):
                if turb > 9:  # int_turbidity == 10: the function returns before the high-turbidity terms
                    continue
                m = ds[9 * 6 * 10 * alb + 9 * 6 * turb:][:54].reshape(6, 9)
                configs[ch] += w_alb * w_t * np.array([_bezier5(t, m[:, i]) for i in range(9)])
                r = rad[6 * 10 * alb + 6 * turb:][:6]
                radiances[ch] += w_alb * w_t * _bezier5(t, r)
    return configs, radiances


def hosek_radiance(config, radiance, theta, gamma):
    """arhosek_tristim_skymodel_radiance / ArHosekSkyModel_GetRadianceInternal (skymodel.cpp:226-239, 383-397), vectorised."""
    cg, ct = np.cos(gamma), np.cos(theta)
    exp_m = np.exp(config[4] * gamma)
    ray_m = cg * cg
    mie_m = (1.0 + cg * cg) / np.power(1.0 + config[8] * config[8] - 2.0 * config[8] * cg, 1.5)
    zenith = np.sqrt(ct)
    return (1.0 + config[0] * np.exp(config[1] / (ct + 0.01))) * (config[2] + config[3] * exp_m + config[5] * ray_m + config[6] * mie_m + config[7] * zenith) * radiance


def sun_radiance_rgb(tables, theta, turbidity):
    """computeSunRadiance (sunmodel.h:317-365): solar spectrum x Rayleigh, aerosol, ozone, mixed-gas and water-vapour transmittances on
    350..800 nm in 5 nm steps, then Spectrum::fromContinuousSpectrum of the RGB build (CIE matching functions → XYZ → Rec.709)."""
    lam = np.arange(350.0, 801.0, 5.0)
    k_o = spectrum._eval_interp(tables["k_oWavelengths"], tables["k_oAmplitudes"][:64], lam)
    k_g = spectrum._eval_interp(tables["k_gWavelengths"], tables["k_gAmplitudes"], lam)
    k_wa = spectrum._eval_interp(tables["k_waWavelengths"], tables["k_waAmplitudes"], lam)
    sol = spectrum._eval_interp(tables["solWavelengths"], tables["solAmplitudes"], lam)
    beta = 0.04608365822050 * turbidity - 0.04586025928522
    m = 1.0 / (math.cos(theta) + 0.15 * (93.885 - theta / math.pi * 180.0) ** -1.253)
    tau_r = np.exp(-m * 0.008735 * (lam / 1000.0) ** -4.08)
    tau_a = np.exp(-m * beta * (lam / 1000.0) ** -1.3)
    tau_o = np.exp(-m * k_o * 0.35)
    tau_g = np.exp(-1.41 * k_g * m / (1 + 118.93 * k_g * m) ** 0.45)
    tau_wa = np.exp(-0.2385 * k_wa * 2.0 * m / (1 + 20.07 * k_wa * 2.0 * m) ** 0.45)
    data = sol * tau_r * tau_a * tau_o * tau_g * tau_wa
    return spectrum.interpolated_to_rgb(list(zip(lam, data)), zero_extend=False, clamp=True).astype(np.float64)


def _sample02(n):
    """(0,2)-sequence point n (qmc.h:43-59, 82-87, 115-120): van der Corput and Sobol' radical inverses in base 2, 24 / 32 bits."""
    i = np.arange(n, dtype=np.uint64)
    x = np.zeros(n, np.uint64)
    for b in range(32):
        x |= ((i >> np.uint64(b)) & np.uint64(1)) << np.uint64(31 - b)
    u = (x >> np.uint64(8)).astype(np.float32) / np.float32(1 << 24)
    y = np.zeros(n, np.uint64)
    v, k = 1 << 31, i.copy()
    while k.any():
        y ^= np.where(k & np.uint64(1), np.uint64(v), np.uint64(0))
        k >>= np.uint64(1)
        v ^= v >> 1
    return u, (y.astype(np.float64) / float(1 << 32)).astype(np.float32)


def load_tables(mitsuba_src):
    d = os.path.join(mitsuba_src, "src", "emitters", "sunsky")
    t = parse_c_arrays(os.path.join(d, "skymodeldata.h"), ["datasetRGB1", "datasetRGB2", "datasetRGB3", "datasetRGBRad1", "datasetRGBRad2", "datasetRGBRad3"])
    t.update(parse_c_arrays(os.path.join(d, "sunmodel.h"), ["k_oWavelengths", "k_oAmplitudes", "k_gWavelengths", "k_gAmplitudes", "k_waWavelengths", "k_waAmplitudes",
                                                          "solWavelengths", "solAmplitudes"]))
    return t


def bake(props, mitsuba_src):
    """SunSkyEmitter(props) (sunsky.cpp:100-235) → float32 radiance map [resolution / 2, resolution, 3] in the envmap plug-in's
    latitude-longitude layout (row 0 = +y, azimuth from -z towards +x: toSphere / fromSphere, sunmodel.h:83-100 = envmap.cpp's)."""
    tables = load_tables(mitsuba_src)
    scale = float(props.get("scale", 1.0))
    sun_scale, sky_scale = float(props.get("sunScale", scale)), float(props.get("skyScale", scale))
    sun_radius_scale = float(props.get("sunRadiusScale", 1.0))
    turbidity, stretch = float(props.get("turbidity", 3.0)), float(props.get("stretch", 1.0))
    albedo = np.broadcast_to(np.asarray(props.get("albedo", 0.2), np.float64), (3,))  # sky.cpp:224: Spectrum(0.2f)
    res = int(props.get("resolution", 512))
    w, h = res, res // 2
    sun_zenith, sun_azimuth = sun_coordinates(props)
    sun_elevation = 0.5 * math.pi - sun_zenith
    if sun_elevation < 0:
        raise ValueError("sunsky: the sun is below the horizon (sky.cpp:239-240)")
    if props.get("extend"):
        raise NotImplementedError("sunsky: extend")
    # the sky, sky.cpp:409-447 (one model state per channel, each with its own albedo)
    theta = ((np.arange(h) + 0.5) * (math.pi / h))[:, None] / stretch
    phi = ((np.arange(w) + 0.5) * (2 * math.pi / w))[None, :]
    cos_gamma = np.cos(theta) * math.cos(sun_zenith) + np.sin(theta) * math.sin(sun_zenith) * np.cos(phi - sun_azimuth)
    gamma = np.arccos(np.clip(cos_gamma, -1.0, 1.0))
    img = np.zeros((h, w, 3))
    up = (np.cos(theta) > 0)[:, 0]
    for ch in range(3):
        configs, radiances = hosek_rgb_state(tables, turbidity, float(albedo[ch]), sun_elevation)
        th = np.broadcast_to(theta, (h, w))[up]
        img[up, :, ch] = hosek_radiance(configs[ch], radiances[ch], th, gamma[up]) / 106.856980
    img = np.maximum(img, 0.0) * sky_scale
    img = img.astype(np.float32).astype(np.float64)
    # the sun, sunsky.cpp:165-205
    sun_rad = sun_radiance_rgb(tables, sun_zenith, turbidity) * sun_scale
    sz = sun_zenith * stretch
    n = np.array([math.sin(sun_azimuth) * math.sin(sz), math.cos(sz), -math.cos(sun_azimuth) * math.sin(sz)])  # toSphere
    if abs(n[0]) > abs(n[1]):  # coordinateSystem, util.cpp:592-601
        inv = 1.0 / math.sqrt(n[0] * n[0] + n[2] * n[2])
        c = np.array([n[2] * inv, 0.0, -n[0] * inv])
    else:
        inv = 1.0 / math.sqrt(n[1] * n[1] + n[2] * n[2])
        c = np.array([0.0, n[2] * inv, -n[1] * inv])
    b = np.cross(c, n)
    th0 = math.radians(SUN_APP_RADIUS * 0.5)
    if sun_radius_scale == 0:
        raise NotImplementedError("sunsky: sunRadiusScale = 0 (directional sun)")
    cos_cut = math.cos(th0 * sun_radius_scale)
    n_samples = int(max(100.0, (res * res // 2) * (0.5 * (1 - cos_cut)) * 1000))
    value = sun_rad * (2 * math.pi * (1 - math.cos(th0))) * float(w * h) / (2 * math.pi * math.pi * n_samples)
    u, v = _sample02(n_samples)
    ct = (1 - u.astype(np.float64)) + u.astype(np.float64) * cos_cut  # squareToUniformCone, warp.cpp:54-63
    st = np.sqrt(np.maximum(0.0, 1.0 - ct * ct))
    ph = 2.0 * math.pi * v.astype(np.float64)
    local = np.stack([np.cos(ph) * st, np.sin(ph) * st, ct], 1)
    d = local[:, :1] * b[None] + local[:, 1:2] * c[None] + local[:, 2:3] * n[None]  # Frame(n): s = b, t = c
    sin_theta = np.sqrt(np.maximum(0.0, 1.0 - d[:, 1] ** 2))
    az = np.arctan2(d[:, 0], -d[:, 2])
    az = np.where(az < 0, az + 2 * math.pi, az)
    el = np.arccos(np.clip(d[:, 1], -1.0, 1.0))
    px = np.clip((az * (w / (2 * math.pi))).astype(np.int64), 0, w - 1)
    py = np.clip((el * (h / math.pi)).astype(np.int64), 0, h - 1)
    contrib = value[None, :] / np.maximum(1e-3, sin_theta)[:, None]
    np.add.at(img, (py, px), contrib)
    return img.astype(np.float32), dict(sun_zenith=sun_zenith, sun_azimuth=sun_azimuth, sun_radiance=sun_rad, n_sun_samples=n_samples)
