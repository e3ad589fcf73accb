"""TacView ACMI 2.0 text recording of a batch (the on-disk format the reference's `render` emits:
envs/env_base.py:111-151, envs/singlecombat_env.py:276-321) and the local-tangent-plane -> WGS-84 conversion it
needs (envs/utils/utils.py:74-142).

This is a consumer of the hot path, not part of it: one device->host copy of six state columns per rendered
frame, text formatting on the host.
"""
import math
import os

import numpy as np

WGS84_A = 6378137.0          # semi-major axis [m]            (utils.py:5)
WGS84_B = 6356752.3142       # semi-minor axis [m]            (utils.py:6)
_F = (WGS84_A - WGS84_B) / WGS84_A
_E2 = _F * (2.0 - _F)        # first eccentricity squared     (utils.py:7-8)
_REF_PI = 3.14159265359      # the reference converts rad -> deg with this truncated pi (utils.py:9,133-134)
FT = 0.3048


def enu_to_ecef(east, north, up, lat_ref=0.0, lon_ref=0.0, h_ref=0.0):
    """Local east/north/up [m] about the geodetic reference point -> ECEF [m] (numpy, vectorised)."""
    east, north, up = (np.asarray(v, dtype=np.float64) for v in (east, north, up))
    la, lo = math.radians(lat_ref), math.radians(lon_ref)
    sl, cl, so, co = math.sin(la), math.cos(la), math.sin(lo), math.cos(lo)
    nu = WGS84_A / math.sqrt(1.0 - _E2 * sl * sl)                  # prime-vertical radius at the reference
    x0, y0, z0 = (h_ref + nu) * cl * co, (h_ref + nu) * cl * so, (h_ref + (1.0 - _E2) * nu) * sl
    radial = cl * up - sl * north                                   # component towards the spin axis' normal
    return co * radial - so * east + x0, so * radial + co * east + y0, sl * up + cl * north + z0


def ecef_to_geodetic(x, y, z):
    """ECEF [m] -> (lat [deg], lon [deg], h [m]); closed form of Heikkinen (1982) / Zhu (1994), as the reference uses."""
    x, y, z = (np.asarray(v, dtype=np.float64) for v in (x, y, z))
    a, b = WGS84_A, WGS84_B
    e2 = 1.0 - (b / a) ** 2
    ep2 = e2 * (a / b) ** 2
    r = np.hypot(x, y)
    big_e2 = a * a - b * b
    f = 54.0 * b * b * z * z
    g = r * r + (1.0 - e2) * z * z - e2 * big_e2
    c = e2 * e2 * f * r * r / (g * g * g)
    s = np.cbrt(1.0 + c + np.sqrt(c * c + 2.0 * c))
    p = f / (3.0 * (s + 1.0 / s + 1.0) ** 2 * g * g)
    q = np.sqrt(1.0 + 2.0 * e2 * e2 * p)
    r0 = -(p * e2 * r) / (1.0 + q) + np.sqrt(0.5 * a * a * (1.0 + 1.0 / q) - p * (1.0 - e2) * z * z / (q * (1.0 + q)) - 0.5 * p * r * r)
    t = (r - e2 * r0) ** 2
    u = np.sqrt(t + z * z)
    v = np.sqrt(t + (1.0 - e2) * z * z)
    z0 = b * b * z / (a * v)
    h = u * (1.0 - b * b / (a * v))
    lat = np.arctan((z + ep2 * z0) / r)
    lon = np.arctan2(y, x)
    k = _REF_PI / 180.0
    return lat / k, lon / k, h


def enu_to_geodetic(east, north, up, lat_ref=0.0, lon_ref=0.0, h_ref=0.0):
    return ecef_to_geodetic(*enu_to_ecef(east, north, up, lat_ref, lon_ref, h_ref))


class AcmiRecorder:
    """Appends frames to a TacView text recording.  `frame()` takes the [k, >=6] state rows (ft, rad) of the aircraft to
    draw; ids start at 100 as in the reference."""

    HEADER = ('FileType=text/acmi/tacview\n', 'FileVersion=2.0\n', '0,ReferenceTime=2023-04-01T00:00:00Z\n')

    def __init__(self, path):
        self.path = path
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        with open(path, 'w', encoding='utf-8') as f:
            f.writelines(self.HEADER)

    def frame(self, timestamp, states, names='F16', colors='Red', first_id=100):
        s = np.asarray(states, dtype=np.float64)
        if s.ndim == 1:
            s = s[None]
        k = s.shape[0]
        names = [names] * k if isinstance(names, str) else list(names)
        colors = [colors] * k if isinstance(colors, str) else list(colors)
        lat, lon, alt = enu_to_geodetic(s[:, 1] * FT, s[:, 0] * FT, s[:, 2] * FT)   # (east, north, up) = (epos, npos, alt)
        deg = s[:, 3:6] * 180.0 / np.pi
        with open(self.path, 'a', encoding='utf-8') as f:
            f.write(f'#{timestamp:.2f}\n')
            for i in range(k):
                f.write(f'{first_id + i},T={lon[i]}|{lat[i]}|{alt[i]}|{deg[i, 0]}|{deg[i, 1]}|{deg[i, 2]},'
                        f'Name={names[i]},Color={colors[i]}\n')


def parse_acmi(text):
    """-> (header lines, [(timestamp, [(id, [lon, lat, alt, roll, pitch, yaw], {props})...])...]) — for tests and tooling."""
    lines = text.splitlines()
    header, frames = [], []
    for ln in lines:
        if ln.startswith('#'):
            frames.append((float(ln[1:]), []))
        elif not frames:
            header.append(ln)
        else:
            oid, rest = ln.split(',', 1)
            fields = rest.split(',')
            nums = [float(v) for v in fields[0][2:].split('|')]
            props = dict(p.split('=', 1) for p in fields[1:])
            frames[-1][1].append((int(oid), nums, props))
    return header, frames
