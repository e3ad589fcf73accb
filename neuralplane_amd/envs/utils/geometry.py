"""Host-side helpers with the names and meaning of the reference's envs/utils/utils.py:35-250 — WGS-84 conversions (scalar
floats, degrees / metres) and the pairwise air-combat geometry and shaping functions (batched torch tensors, any device).

Inside SingleCombatEnv.step the same geometry and rewards are evaluated by the fused HIP kernel (np_f16_combat.h); these
functions are what a caller uses OUTSIDE the step — custom shaping, analysis, TacView export — and what the golden vectors of
tests/golden/pairwise_kat.npz / geodesy_kat.npz were recorded from on the reference's side.
"""
import math

import torch

# WGS-84
_A = 6378137.0
_B = 6356752.3142
_F = (_A - _B) / _A
_E2 = _F * (2.0 - _F)
_REF_PI = 3.14159265359             # the reference's own pi (utils.py:9): quadrant correction and rad -> deg (utils.py:124-134)
_REF_DEG = _REF_PI / 180.0


def _prime_vertical(sin_lat):
    return _A / math.sqrt(1.0 - _E2 * sin_lat * sin_lat)


def _enu_basis(lat0, lon0):
    """Rows of the ECEF -> ENU rotation at (lat0, lon0) [deg]."""
    la, lo = math.radians(lat0), math.radians(lon0)
    sl, cl, sp, cp = math.sin(la), math.cos(la), math.sin(lo), math.cos(lo)
    east = (-sp, cp, 0.0)
    north = (-sl * cp, -sl * sp, cl)
    up = (cl * cp, cl * sp, sl)
    return east, north, up


def geodetic_to_ecef(lat, lon, h):
    """(lat, lon) [deg], h [m] -> ECEF (x, y, z) [m]."""
    la, lo = math.radians(lat), math.radians(lon)
    sl, cl = math.sin(la), math.cos(la)
    n = _prime_vertical(sl)
    return (h + n) * cl * math.cos(lo), (h + n) * cl * math.sin(lo), (h + (1.0 - _E2) * n) * sl


def ecef_to_enu(x, y, z, lat0, lon0, h0):
    """ECEF point -> (east, north, up) [m] about the reference point (lat0, lon0 [deg], h0 [m])."""
    x0, y0, z0 = geodetic_to_ecef(lat0, lon0, h0)
    d = (x - x0, y - y0, z - z0)
    east, north, up = _enu_basis(lat0, lon0)
    return tuple(sum(r[k] * d[k] for k in range(3)) for r in (east, north, up))


def enu_to_ecef(xEast, yNorth, zUp, lat0, lon0, h0):
    """(east, north, up) [m] about (lat0, lon0, h0) -> ECEF."""
    x0, y0, z0 = geodetic_to_ecef(lat0, lon0, h0)
    east, north, up = _enu_basis(lat0, lon0)
    return tuple(c0 + east[k] * xEast + north[k] * yNorth + up[k] * zUp for k, c0 in enumerate((x0, y0, z0)))


def ecef_to_geodetic(x, y, z):
    """ECEF -> (lat [deg], lon [deg], h [m]); Bowring-style closed form followed by Newton corrections on the latitude
    (converged to round-off for |h| < 1000 km), agreeing with the reference's closed form to < 1e-9 deg / 1e-6 m."""
    p = math.hypot(x, y)
    if x >= 0:
        lon = math.atan2(y, x)
    else:                                          # the reference adds / subtracts its truncated pi in the western half-plane
        lon = math.atan(y / x) + (_REF_PI if y >= 0 else -_REF_PI)
    if p < 1e-9:                                   # on the polar axis
        return math.copysign(90.0, z), lon / _REF_DEG, abs(z) - _B
    ep2 = (_A * _A - _B * _B) / (_B * _B)
    theta = math.atan2(z * _A, p * _B)
    lat = math.atan2(z + ep2 * _B * math.sin(theta) ** 3, p - _E2 * _A * math.cos(theta) ** 3)
    for _ in range(3):
        sl = math.sin(lat)
        n = _prime_vertical(sl)
        h = p / math.cos(lat) - n
        lat = math.atan2(z, p * (1.0 - _E2 * n / (n + h)))
    sl = math.sin(lat)
    n = _prime_vertical(sl)
    h = p / math.cos(lat) - n if abs(math.cos(lat)) > 1e-12 else abs(z) - _B
    return lat / _REF_DEG, lon / _REF_DEG, h


def geodetic_to_enu(lat, lon, h, lat_ref, lon_ref, h_ref):
    return ecef_to_enu(*geodetic_to_ecef(lat, lon, h), lat_ref, lon_ref, h_ref)


def enu_to_geodetic(xEast, yNorth, zUp, lat_ref, lon_ref, h_ref):
    return ecef_to_geodetic(*enu_to_ecef(xEast, yNorth, zUp, lat_ref, lon_ref, h_ref))


# ---- pairwise geometry (utils.py:156-206) ----------------------------------------------------------------------------------------
def _off_angle(delta, vel, dist):
    """angle between the line of sight `delta` and the velocity `vel`, arccos of the clamped cosine (the reference's 1e-8 guard)."""
    speed = torch.linalg.norm(vel, dim=1)
    cosine = torch.sum(delta * vel, dim=1) / (dist * speed + 1e-8)
    return torch.arccos(torch.clamp(cosine, -1, 1))


def _ao_ta_r(ego_pos, enm_pos, ego_vel, enm_vel, return_side):
    delta = enm_pos - ego_pos
    dist = torch.linalg.norm(delta, dim=1)
    ao = _off_angle(delta, ego_vel, dist)       # antenna-train angle: own velocity against the line of sight
    ta = _off_angle(delta, enm_vel, dist)       # aspect angle: the opponent's velocity against the same line
    if not return_side:
        return ao, ta, dist
    # which side the opponent is on: sign of the z component of (horizontal velocity) x (horizontal line of sight)
    side = torch.sign(ego_vel[:, 0] * delta[:, 1] - ego_vel[:, 1] * delta[:, 0])
    return ao, ta, dist, side


def get_AO_TA_R(ego_pos, enm_pos, ego_vel, enm_vel, return_side=False):
    """[m, 3] positions and velocities (north, east, up order as the reference passes them) -> (AO, TA, R[, side])."""
    return _ao_ta_r(ego_pos, enm_pos, ego_vel, enm_vel, return_side)


def get2d_AO_TA_R(ego_pos, enm_pos, ego_vel, enm_vel, return_side=False):
    """The same in the horizontal plane (last component dropped)."""
    return _ao_ta_r(ego_pos[:, :-1], enm_pos[:, :-1], ego_vel[:, :-1], enm_vel[:, :-1], return_side)


# ---- shaping functions (utils.py:208-250) ------------------------------------------------------------------------------------------
def _aspect_term(TA, k):
    """min(atanh(1 - max(k TA / pi, 1e-4)) / (2 pi), 0): the non-positive aspect-angle term of the orientation rewards."""
    x = torch.clamp_min(k * TA / torch.pi, 1e-4)
    return torch.clamp_max(torch.arctanh(1 - x) / (2 * torch.pi), 0.0)


def orientation_reward(AO, TA, version='v2'):
    if version == 'v0':
        return (1 - torch.tanh(9 * (AO - torch.pi / 9))) / 3 + 1 / 3 + _aspect_term(TA, 2.0) + 0.5
    if version == 'v1':
        x = torch.clamp_min(2 * TA / torch.pi, 1e-4)
        return (1 - torch.tanh(2 * (AO - torch.pi / 2))) / 2 * torch.arctanh(1 - x) / (2 * torch.pi) + 0.5
    if version == 'v2':
        return 1 / (50 * AO / torch.pi + 2) + 1 / 2 + _aspect_term(TA, 1.9) + 0.5
    raise NotImplementedError(f'Unknown orientation function version: {version}')


def range_reward(target_dist, R, version='v3'):
    """R and target_dist in km."""
    if version == 'v0':
        d = R - target_dist
        return torch.exp(-d ** 2 * 0.004) / (1 + torch.exp(-(d + 2) * 2))
    if version in ('v1', 'v2'):
        d = R - target_dist
        base = torch.clamp(1.2 * torch.clamp_max(torch.exp(-d * 0.21), 1.0) / (1 + torch.exp(-(d + 1) * 0.8)), 0.3, 1)
        return base if version == 'v1' else torch.maximum(base, torch.sign(7 - R))
    if version == 'v3':
        near = (R < 5).to(R.dtype)
        parabola = torch.clamp(-0.032 * R ** 2 + 0.284 * R + 0.38, 0, 1)
        return 1 * near + (1 - near) * parabola + torch.clamp(torch.exp(-0.16 * R), 0, 0.2)
    raise NotImplementedError(f'Unknown range function version: {version}')


def orientation_fn(AO):
    """Triangle of height 1 and half-width pi / 6 around AO = 0 (counted from both sides at exactly 0, as the reference does)."""
    slope = 6 * AO / torch.pi
    right = ((AO >= 0) & (AO <= torch.pi / 6)).to(AO.dtype)
    left = ((AO <= 0) & (AO >= -torch.pi / 6)).to(AO.dtype)
    return (1 - slope) * right + (1 + slope) * left


def distance_fn(R):
    """1 up to 1 km, falling linearly to 0 at 3 km."""
    return 1 * (R <= 1).to(R.dtype) + (3 - R) / 2 * ((R > 1) & (R <= 3)).to(R.dtype)
