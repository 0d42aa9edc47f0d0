"""Synthetic wind-field generators for the BASELINE configs (inputs, not kernels).

* :func:`ideal_vortex` -- vectorised restatement of the reference's analytic
  moving-vortex generator, ``examples/ideal_vortex.py:130-208``;
  configs 1 and 2.  Checked against a loop-faithful copy of the reference's
  triple loop in ``tests/test_flows.py``.
* :func:`era5_like` -- the "ERA5-like" 0.25 degree global flow of configs 3-5
  (SURVEY section 8d); not in the reference, defined here.

Arrays are returned ``(time, latitude, longitude)``, coordinates ascending.
"""
from __future__ import annotations

import numpy as np

__all__ = ["ideal_vortex", "vortex_config_subtropical", "config1", "config2",
           "era5_like", "seed_grid"]

# examples/ideal_vortex.py:220-223
vortex_config_subtropical = {
    'lat_min': -88, 'lat_max': 89, 'lon_min': -180, 'lon_max': 180, 'dx': 2,
    'dy': 2, 'u_c': 0, 'k': 0, 'diag_factor': 1, 'v_c': 0, 'nt': 8,
    'radius': 2, 'max_intensity': 60, 'center': [-55, -20], 'basic_zonal': 0}


def _vortex_on(lats, lons, nt, max_intensity, radius, center, u_c, v_c,
               basic_zonal, k):
    t = np.arange(nt, dtype=np.float64)[:, None, None]
    new_x = lons[None, None, :] - center[0] - u_c * t           # ideal_vortex.py:178
    if k > 0:                                                    # :179-180
        new_y = lats[None, :, None] - center[1] - v_c * np.sin(k * 2 * np.pi * t / nt)
    elif k == 0:                                                 # :181-182
        new_y = lats[None, :, None] - center[1] - v_c * t
    else:
        raise ValueError('Meridional wavenumber k must be greater than zero.')
    new_x, new_y = np.broadcast_arrays(new_x, new_y)
    distance = np.sqrt(new_x ** 2 + new_y ** 2)                  # :185
    theta = np.arccos(new_y / (distance + 1e-8))                 # :191
    with np.errstate(divide='ignore', invalid='ignore'):
        mag = np.where(distance > radius,                        # :192-195
                       max_intensity * radius ** 2 / (2 * distance),
                       max_intensity * 0.5 * distance)
    u = np.cos(theta) * mag + basic_zonal                        # :197
    v = np.where(new_x < 0, np.sin(theta) * mag,                 # :198-201
                 np.sin(theta + np.pi) * mag)
    return np.ascontiguousarray(u), np.ascontiguousarray(v)


def ideal_vortex(lat_min, lat_max, lon_min, lon_max, dx, dy, nt,
                 max_intensity=10, radius=5, center=None, u_c=0, v_c=0,
                 diag_factor=0, basic_zonal=2, k=0):
    """Same signature and defaults as examples/ideal_vortex.py:130-133.

    Returns ``(u, v, lats, lons)``; u, v are ``(nt, ny, nx)`` float64.
    """
    lats = np.arange(lat_min, lat_max, dy)                       # :159
    lons = np.arange(lon_min, lon_max, dx)                       # :160
    u, v = _vortex_on(lats, lons, nt, max_intensity, radius, center, u_c, v_c,
                      basic_zonal, k)
    return u, v, lats, lons


def config1():
    """BASELINE config 1: the example's actual values (89x180, nt=8, fp64)."""
    return ideal_vortex(**vortex_config_subtropical)


def config2(n=1024, nt=201):
    """BASELINE config 2: n x n nodes, moving vortex, fp64.

    lats = -88 + j*176/n, lons = -180 + i*360/n; u_c=0.02, v_c=0.01 degrees
    per time index.  Use with ``timestep=-900``.
    """
    lats = -88.0 + np.arange(n) * (176.0 / n)
    lons = -180.0 + np.arange(n) * (360.0 / n)
    u, v = _vortex_on(lats, lons, nt, max_intensity=60, radius=2,
                      center=[-55, -20], u_c=0.02, v_c=0.01, basic_zonal=0, k=0)
    return u, v, lats, lons


def config2_on_device(torch, device, n=1024, nt=201):
    """:func:`config2` evaluated with torch on ``device`` (float64 tensors ``u, v`` of shape ``(nt, n, n)`` plus the numpy
    coordinates): the same formula term by term -- the device's ``arccos / sin / cos`` differ from numpy's in the last bits,
    so the field equals :func:`config2` to ~1e-12 m/s, not bit for bit.  For benchmarks: the numpy triple-broadcast takes
    a minute on one host core, this a few milliseconds."""
    lats = -88.0 + np.arange(n) * (176.0 / n)
    lons = -180.0 + np.arange(n) * (360.0 / n)
    f64 = torch.float64
    t = torch.arange(nt, dtype=f64, device=device)[:, None, None]
    la = torch.as_tensor(lats, dtype=f64, device=device)[None, :, None]
    lo = torch.as_tensor(lons, dtype=f64, device=device)[None, None, :]
    max_intensity, radius, center, u_c, v_c = 60.0, 2.0, (-55.0, -20.0), 0.02, 0.01
    new_x = (lo - center[0] - u_c * t).expand(nt, n, n)
    new_y = (la - center[1] - v_c * t).expand(nt, n, n)
    distance = torch.sqrt(new_x ** 2 + new_y ** 2)
    theta = torch.arccos(new_y / (distance + 1e-8))
    mag = torch.where(distance > radius, max_intensity * radius ** 2 / (2 * distance), max_intensity * 0.5 * distance)
    u = torch.cos(theta) * mag
    v = torch.where(new_x < 0, torch.sin(theta) * mag, torch.sin(theta + np.pi) * mag)
    return u.contiguous(), v.contiguous(), lats, lons


def _era5_like_terms(nt, ny, nx, seed, dt_seconds, n_modes):
    """The separable factors of :func:`era5_like`: u[n] = base[:, None] + Pu.T @ lu[n], v[n] = Pv.T @ lv[n]."""
    lats = -90.0 + 180.0 / ny / 2 + (180.0 / ny) * np.arange(ny)   # -89.875 ... 89.875
    lons = -180.0 + (360.0 / nx) * np.arange(nx)                    # -180 ... 179.75
    rng = np.random.default_rng(seed)
    km = rng.integers(1, 9, n_modes)
    lm = rng.integers(1, 7, n_modes)
    A = rng.uniform(3, 12, n_modes)
    B = rng.uniform(3, 12, n_modes)
    om = rng.uniform(-2, 2, n_modes) * 2 * np.pi / 86400.0
    al = rng.uniform(0, 2 * np.pi, n_modes)
    be = rng.uniform(0, 2 * np.pi, n_modes)
    phi = np.deg2rad(lats)
    lam = np.deg2rad(lons)
    t = dt_seconds * np.arange(nt)
    cphi = np.cos(phi)
    Pu = A[:, None] * cphi[None, :] * np.cos(lm[:, None] * phi[None, :])   # (m, ny)
    Pv = B[:, None] * cphi[None, :] * np.sin(lm[:, None] * phi[None, :])
    ct = np.cos(om[:, None] * t[None, :])                                  # (m, nt)
    st = np.sin(om[:, None] * t[None, :])
    su = np.sin(km[:, None] * lam[None, :] + al[:, None])                  # (m, nx)
    cu = np.cos(km[:, None] * lam[None, :] + al[:, None])
    sv = np.sin(km[:, None] * lam[None, :] + be[:, None])
    cv = np.cos(km[:, None] * lam[None, :] + be[:, None])
    return lats, lons, 25.0 * cphi ** 2, Pu, Pv, ct, st, su, cu, sv, cv


def era5_like(nt=97, ny=720, nx=1440, seed=20260355, dtype=np.float32,
              dt_seconds=900.0, n_modes=12):
    """Configs 3-5: smooth synthetic global flow on a 0.25 degree grid.

    u = 25 cos^2(phi) + sum_m A_m cos(phi) sin(k_m lam + w_m t + a_m) cos(l_m phi)
    v =                 sum_m B_m cos(phi) cos(k_m lam + w_m t + b_m) sin(l_m phi)

    evaluated through the angle-sum identity so the cost is 2*n_modes rank-1
    updates per component.  ``rng = default_rng(seed)`` draws, in this order:
    k, l, A, B, omega, alpha, beta.  A shorter series is a prefix of a longer one.
    """
    lats, lons, base, Pu, Pv, ct, st, su, cu, sv, cv = _era5_like_terms(nt, ny, nx, seed, dt_seconds, n_modes)
    u = np.empty((nt, ny, nx), dtype=dtype)
    v = np.empty((nt, ny, nx), dtype=dtype)
    for n in range(nt):
        # sin(x + wt) = sin x cos wt + cos x sin wt ; cos(x + wt) = cos x cos wt - sin x sin wt
        lu = su * ct[:, n:n + 1] + cu * st[:, n:n + 1]                     # (m, nx)
        lv = cv * ct[:, n:n + 1] - sv * st[:, n:n + 1]
        u[n] = (base[:, None] + Pu.T @ lu).astype(dtype)
        v[n] = (Pv.T @ lv).astype(dtype)
    return u, v, lats.astype(dtype), lons.astype(dtype)


def era5_like_on_device(torch, device, nt=97, ny=720, nx=1440, seed=20260355, dtype=np.float32, dt_seconds=900.0, n_modes=12):
    """:func:`era5_like` evaluated with torch on ``device``: the same factors (computed on the host in float64), the
    ``nt`` rank-12 products on the device in float64, cast to ``dtype``.  The device's matrix product sums in another
    order than the host BLAS, so a value may differ from :func:`era5_like` in the last float32 bit -- for benchmarks of
    the long series (385 levels take 25 s of host time, milliseconds here), not for parity tests.  Returns device tensors
    ``u, v`` of shape ``(nt, ny, nx)`` and the numpy coordinates."""
    lats, lons, base, Pu, Pv, ct, st, su, cu, sv, cv = _era5_like_terms(nt, ny, nx, seed, dt_seconds, n_modes)
    f64 = torch.float64
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=f64, device=device)
    base, PuT, PvT, ct, st, su, cu, sv, cv = T(base), T(Pu.T), T(Pv.T), T(ct), T(st), T(su), T(cu), T(sv), T(cv)
    out_dtype = getattr(torch, np.dtype(dtype).name)
    u = torch.empty((nt, ny, nx), dtype=out_dtype, device=device)
    v = torch.empty((nt, ny, nx), dtype=out_dtype, device=device)
    for n in range(nt):
        lu = su * ct[:, n:n + 1] + cu * st[:, n:n + 1]
        lv = cv * ct[:, n:n + 1] - sv * st[:, n:n + 1]
        u[n] = (base[:, None] + PuT @ lu).to(out_dtype)
        v[n] = (PvT @ lv).to(out_dtype)
    return u, v, lats.astype(dtype), lons.astype(dtype)


def seed_grid(ny, nx, lats, lons, dtype=None):
    """Uniform ``ny x nx`` seed grid spanning the field coordinates inclusively
    (``linspace``), SURVEY section 8d configs 3-4."""
    dtype = dtype or lats.dtype
    return (np.linspace(float(lats[0]), float(lats[-1]), ny).astype(dtype),
            np.linspace(float(lons[0]), float(lons[-1]), nx).astype(dtype))
