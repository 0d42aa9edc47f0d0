"""CPU oracle for the LagrangianCoherence FTLE hot path -- TEST INFRASTRUCTURE ONLY.

This module is a numpy + scipy restatement, on plain arrays, of the reference's
parcel-advection -> flow-map-gradient -> sigma_max path.  It is the *checker*
for the HIP kernels: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``lagrangiancoherence_amd/`` imports it and the product has no CPU fallback.

PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, fixtures or
golden vectors and cannot be imported in this image
(``xarray``, ``numba``, ``dask``, ``cftime``, ``windspharm``, ``xr_tools`` are
absent -- an ordinary ModuleNotFoundError).  What *is*
pinned: every arithmetic kernel the reference delegates to is called here at
the same call site with the same arguments (``scipy.ndimage.map_coordinates``,
``scipy.ndimage.gaussian_filter``, ``numpy.linalg.norm(ord=2)`` via
``scipy.linalg.norm``) on scipy 1.15.3 / numpy 2.2.6, and the hand-restated
parts (the numba stencil, the clamps, the SETTLS accumulation) are checked by
the known-answer tests in ``tests/test_oracle_kat.py``.

All ``file:line`` citations are relative to the reference checkout
(``/root/reference``).  Arrays are ``(time, latitude, longitude)`` with
latitude and longitude ascending, i.e. the layout the reference reaches after
its ``sortby`` calls (LCS/trajectory.py:49-52, LCS/LCS.py:101-104).

numpy type promotion is deliberately left to numpy: the reference's xarray
arithmetic defers to numpy ufuncs, so feeding this module float32 (or mixed)
arrays reproduces the dtype each intermediate has in the reference (Q10, Q11).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import norm as _scipy_norm
from scipy.ndimage import gaussian_filter, map_coordinates

EARTH_R = 6371000  # LCS/trajectory.py:54, LCS/LCS.py:23,194, LCS/tools.py:249

__all__ = [
    "xr_map_coordinates",
    "parcel_propagation",
    "fourth_order_derivative",
    "derivative_spherical_coords",
    "flowmap_gradient",
    "sigma_max",
    "sigma_max_closed_form",
    "lcs",
    "ideal_vortex_loops",
    "interp_restated",
    "spline_prefilter_mirror",
]


# --------------------------------------------------------------------------
# a2  tools.xr_map_coordinates  (LCS/tools.py:11-41)
# --------------------------------------------------------------------------
def xr_map_coordinates(field, lat_f, lon_f, new_x, new_y, order=1):
    """Interpolate one 2-D field at every seed.  LCS/tools.py:11-41.

    ``field`` is ``(nlat_f, nlon_f)``; ``lat_f``/``lon_f`` its coordinates;
    ``new_x``/``new_y`` are ``(ny, nx)`` seed positions in degrees.  The row
    classes (interior vs pole) are by *seed row index* (tools.py:24-33).  In
    the reference seeds are the field nodes, so ``ny == nlat_f``; the
    generalisation to a separate seed grid keeps the rule on the seed rows.
    """
    if order < 1:
        # slice(0, -0) is empty and the reshape at tools.py:30 raises.
        raise ValueError("interp_order=0 fails in the reference (tools.py:24-30)")
    field = np.asarray(field)
    x_size = lon_f.shape[0]                                    # tools.py:19
    y_size = lat_f.shape[0]                                    # tools.py:20
    lon_min = np.asarray(lon_f.min())                          # 0-d arrays, as
    lon_max = np.asarray(lon_f.max())                          # ``.min().values``
    lat_min = np.asarray(lat_f.min())
    lat_max = np.asarray(lat_f.max())
    new_x = x_size * (new_x - lon_min) / (lon_max - lon_min)   # tools.py:21 (Q2)
    new_y = y_size * (new_y - lat_min) / (lat_max - lat_min)   # tools.py:22
    ny = new_y.shape[0]
    out = np.empty(new_y.shape, dtype=field.dtype)             # Q10
    idxs = np.arange(order, ny - order)                        # tools.py:25
    if idxs.size:
        out[idxs, :] = map_coordinates(                        # tools.py:26-30
            field,
            np.array([new_y[idxs, :].ravel(), new_x[idxs, :].ravel()]),
            order=order, mode="wrap").reshape(idxs.size, -1)
    pole_idxs = np.hstack([np.arange(0, order), np.arange(-order, 0)])  # :31-33
    out[pole_idxs, :] = map_coordinates(                       # tools.py:35-39 (Q3)
        field,
        np.array([new_y[pole_idxs, :].ravel(), new_x[pole_idxs, :].ravel()]),
        order=1, mode="constant").reshape(pole_idxs.size, -1)
    return out


# --------------------------------------------------------------------------
# a1  trajectory.parcel_propagation  (LCS/trajectory.py:41-142)
# --------------------------------------------------------------------------
def _clamp(positions_x, positions_y, x_min, x_max, y_min, y_max,
           cyclic_xboundary, noncyclic_clamp):
    # trajectory.py:89-90 / 115-116 (Q8: NaN > y_min is False -> y_min)
    positions_y = np.where(positions_y > y_min, positions_y, y_min)
    positions_y = np.where(positions_y < y_max, positions_y, y_max)
    if cyclic_xboundary:
        # trajectory.py:93-94 / 119-120 (Q7: python floor-mod, hard-coded 180)
        positions_x = np.where(positions_x > -180, positions_x, positions_x % 180)
        positions_x = np.where(positions_x < 180, positions_x,
                               -180 + (positions_x % 180))
    elif noncyclic_clamp == "pointwise":
        positions_x = positions_x.copy()
        positions_x[positions_x < x_min] = x_min
        positions_x[positions_x > x_max] = x_max
    elif noncyclic_clamp == "reference_outer":
        # trajectory.py:96-97 / 122-123 (Q9): a tuple of 1-D integer arrays is
        # *orthogonal* indexing on a DataArray: rows x cols cross product.
        positions_x = positions_x.copy()
        r, c = np.where(positions_x < x_min)
        if r.size:
            positions_x[np.ix_(np.unique(r), np.unique(c))] = x_min
        r, c = np.where(positions_x > x_max)
        if r.size:
            positions_x[np.ix_(np.unique(r), np.unique(c))] = x_max
    else:
        raise ValueError(noncyclic_clamp)
    return positions_x, positions_y


def parcel_propagation(U, V, lat, lon, timestep=1, SETTLS_order=0,
                       interp_order=3, cyclic_xboundary=False,
                       return_traj=False, seed_lat=None, seed_lon=None,
                       t0=0, nsteps=None, noncyclic_clamp="reference_outer"):
    """Euler + K accumulate-"SETTLS" sub-steps.  LCS/trajectory.py:41-142.

    ``U, V``: ``(nt, nlat, nlon)``.  Returns ``(x, y)`` of shape ``(ny, nx)``
    (or ``(nsteps+1, ny, nx)`` with ``return_traj``; index 0 is the seed grid,
    trajectory.py:73-74,138-139).  Fields are always consumed in stored order
    whatever the sign of ``timestep`` (Q6, trajectory.py:58-60,80-84).

    Extensions that the reference cannot express (it seeds at field nodes,
    trajectory.py:68-70): ``seed_lat``/``seed_lon`` (separate seed grid, used
    by BASELINE configs 3-5) and ``t0``/``nsteps`` (ensemble start index,
    config 5).  With the defaults they vanish.
    """
    U = np.asarray(U)
    V = np.asarray(V)
    lat = np.asarray(lat)
    lon = np.asarray(lon)
    seed_lat = lat if seed_lat is None else np.asarray(seed_lat)
    seed_lon = lon if seed_lon is None else np.asarray(seed_lon)
    nt = U.shape[0]
    if nsteps is None:
        nsteps = nt - 1 - t0

    conversion_y = 180 / (EARTH_R * np.pi)                     # trajectory.py:55
    conversion_x = 180 / (np.pi * EARTH_R *                    # trajectory.py:56 (Q5)
                          np.abs(np.cos(seed_lat * np.pi / 180)))
    conversion_x = np.broadcast_to(conversion_x[:, None],      # trajectory.py:57
                                   (seed_lat.size, seed_lon.size))
    y_min = lat.min()                                          # trajectory.py:63-66
    y_max = lat.max()
    x_min = lon.min()
    x_max = lon.max()
    positions_x, positions_y = np.meshgrid(seed_lon, seed_lat)  # trajectory.py:68-70
    pos_list_x = [positions_x]
    pos_list_y = [positions_y]

    def interp(F, px, py):
        return xr_map_coordinates(F, lat, lon, px, py, order=interp_order)

    for time_idx in range(t0, t0 + nsteps):                    # trajectory.py:80
        va = interp(V[time_idx], positions_x, positions_y)     # trajectory.py:82
        ua = interp(U[time_idx], positions_x, positions_y)     # trajectory.py:84
        positions_y = positions_y + timestep * conversion_y * va   # :86
        positions_x = positions_x + timestep * conversion_x * ua   # :87
        positions_x, positions_y = _clamp(positions_x, positions_y, x_min, x_max,
                                          y_min, y_max, cyclic_xboundary,
                                          noncyclic_clamp)
        k = 0
        while k < SETTLS_order:                                # trajectory.py:100
            v_t = interp(V[time_idx], positions_x, positions_y)        # :105
            v_tp = interp(V[time_idx + 1], positions_x, positions_y)   # :106
            u_t = interp(U[time_idx], positions_x, positions_y)        # :107
            u_tp = interp(U[time_idx + 1], positions_x, positions_y)   # :108
            positions_y = positions_y + 0.5 * timestep * conversion_y * (
                va + 2 * v_t - v_tp)                           # :110 (Q4: accumulates)
            positions_x = positions_x + 0.5 * timestep * conversion_x * (
                ua + 2 * u_t - u_tp)                           # :112
            positions_x, positions_y = _clamp(positions_x, positions_y, x_min,
                                              x_max, y_min, y_max,
                                              cyclic_xboundary, noncyclic_clamp)
            k += 1
        pos_list_x.append(positions_x)                         # :125-126
        pos_list_y.append(positions_y)

    if return_traj:                                            # :128-139
        return np.stack(pos_list_x), np.stack(pos_list_y)
    return pos_list_x[-1], pos_list_y[-1]                      # :141-142


# --------------------------------------------------------------------------
# a4  tools.fourth_order_derivative (numba)  LCS/tools.py:190-245
#     tools.derivative_spherical_coords     LCS/tools.py:248-267
# --------------------------------------------------------------------------
def fourth_order_derivative(arr, dim=0, isglobal=True):
    """Vectorised restatement of the numba kernel.  LCS/tools.py:190-245.

    numba typing of ``(4/3)*(a[i+1]-a[i-1])/2 - (1/3)*(a[i+2]-a[i-2])/4`` on a
    float32 array: the differences are float32, the float64 constants promote
    the scaling to float64, the store into ``zeros_like(arr)`` rounds back to
    ``arr.dtype`` (Q11).  That is what the explicit ``astype`` calls below do.
    """
    arr = np.asarray(arr)
    out = np.zeros_like(arr)
    f64 = np.float64

    def centred(p1, m1, p2, m2):
        d1 = (p1 - m1).astype(f64)          # arr.dtype difference, then widened
        d2 = (p2 - m2).astype(f64)
        return (4 / 3) * d1 / 2 - (1 / 3) * d2 / 4

    if dim == 0:
        n = arr.shape[0]
        out[2:n - 2] = centred(arr[3:n - 1], arr[1:n - 3], arr[4:n], arr[0:n - 4])  # :202-207
        for i in (0, 1):                                                            # :210-213
            out[i] = (arr[i + 1] - arr[i]).astype(f64) / 2
        for i in (-1, -2):                                                          # :214-217
            out[i] = (arr[i] - arr[i - 1]).astype(f64) / 2
    elif dim == 1:
        if isglobal:                                                                # :220-228
            out[:] = centred(np.roll(arr, -1, 1), np.roll(arr, 1, 1),
                             np.roll(arr, -2, 1), np.roll(arr, 2, 1))
        else:                                                                       # :229-244
            n = arr.shape[1]
            out[:, 2:n - 2] = centred(arr[:, 3:n - 1], arr[:, 1:n - 3],
                                      arr[:, 4:n], arr[:, 0:n - 4])
            for j in (0, 1):
                out[:, j] = (arr[:, j + 1] - arr[:, j]).astype(f64) / 2
            for j in (-1, -2):
                out[:, j] = (arr[:, j] - arr[:, j - 1]).astype(f64) / 2
    else:
        raise ValueError("Dim must be either 0 or 1.")
    return out


def derivative_spherical_coords(values, lat, lon, dim=0, isglobal=True,
                                fd_fp32_cast=True, dlat=None, dlon=None):
    """LCS/tools.py:248-267.  ``values`` is ``(nlat, nlon)`` ascending.

    ``fd_fp32_cast=False`` is NOT reference behaviour; it exists so tests can
    bound the float32 noise of Q11 against a clean float64 stencil.
    ``dlat``/``dlon`` (extension, default = the reference's ``coord[1]-coord[0]``):
    the grid spacing when ``lat``/``lon`` are a WINDOW of a larger grid whose
    first two coordinates define it (tests/_fullsize.py).
    """
    y = lat * np.pi / 180                                                    # :254
    dlon = (lon[1] - lon[0]) if dlon is None else dlon
    dlat = (lat[1] - lat[0]) if dlat is None else dlat
    dx = (np.pi / 180) * dlon * EARTH_R * np.cos(y)                          # :255
    dy = (np.pi / 180) * dlat * EARTH_R                                      # :256
    src = values.astype("float32") if fd_fp32_cast else np.asarray(values)   # :258 (Q11)
    deriv = fourth_order_derivative(src, dim=dim, isglobal=isglobal)
    if dim == 0:
        return deriv / dy                                                    # :262
    if dim == 1:
        return deriv / dx[:, None]                                           # :264
    raise ValueError("Dim must be either 0 or 1.")


# --------------------------------------------------------------------------
# a3  LCS.flowmap_gradient  (LCS/LCS.py:171-225)
# --------------------------------------------------------------------------
def flowmap_gradient(x_departure, y_departure, lat, lon, sigma=None,
                     fd_fp32_cast=True, dlat=None, dlon=None):
    """Returns the ``(9, ny, nx)`` "def_tensor" in the reference's merge order
    ``dXdx,dXdy,dYdx,dYdy,dZdx,dZdy,dXdr,dYdr,dZdr``.  LCS/LCS.py:187-223."""
    if isinstance(sigma, (float, int)):                                      # :187-190
        x_departure = gaussian_filter(x_departure, sigma=sigma)
        y_departure = gaussian_filter(y_departure, sigma=sigma)
    LON = x_departure * np.pi / 180                                          # :195
    LAT = (y_departure - 90) * np.pi / 180                                   # :196
    X = EARTH_R * np.sin(LAT) * np.cos(LON)                                  # :197
    Y = EARTH_R * np.sin(LAT) * np.sin(LON)                                  # :198
    Z = EARTH_R * np.cos(LAT)                                                # :199
    kw = dict(lat=lat, lon=lon, fd_fp32_cast=fd_fp32_cast, dlat=dlat, dlon=dlon)
    dXdx = derivative_spherical_coords(X, dim=1, **kw)                       # :200
    dXdy = derivative_spherical_coords(X, dim=0, **kw)
    dYdx = derivative_spherical_coords(Y, dim=1, **kw)
    dYdy = derivative_spherical_coords(Y, dim=0, **kw)
    dZdx = derivative_spherical_coords(Z, dim=1, **kw)
    dZdy = derivative_spherical_coords(Z, dim=0, **kw)                       # :205
    zero = np.zeros_like(dXdx)                                               # :206-208
    comps = [dXdx, dXdy, dYdx, dYdy, dZdx, dZdy, zero, zero, zero]           # :220
    return np.stack([np.asarray(c) for c in comps])                          # to_array


# --------------------------------------------------------------------------
# a5  eigen step of LCS.__call__  (LCS/LCS.py:145-157)
# --------------------------------------------------------------------------
def sigma_max(def_tensor, tensor_layout="reference"):
    """Largest singular value per seed, NaN where any component is NaN.

    ``tensor_layout='reference'``: the 9 components reshaped row-major to 3x3
    (LCS.py:152-153, Q13) and ``scipy.linalg.norm(ord=2)`` (LCS.py:154).
    ``'physical'`` (NOT reference behaviour): the Jacobian
    ``[[dXdx,dXdy],[dYdx,dYdy],[dZdx,dZdy]]``.
    """
    nine, ny, nx = def_tensor.shape
    flat = def_tensor.reshape(9, ny * nx)                     # stack(points)  :145
    keep = ~np.isnan(flat).any(axis=0)                        # dropna         :146
    vals = flat[:, keep]
    if tensor_layout == "reference":
        vals = vals.reshape([3, 3, vals.shape[-1]])           # :153
    elif tensor_layout == "physical":
        vals = np.stack([vals[0:2], vals[2:4], vals[4:6]])    # (3,2,N)
    else:
        raise ValueError(tensor_layout)
    out = np.full(ny * nx, np.nan, dtype=vals.dtype)
    if vals.shape[-1]:
        out[keep] = _scipy_norm(vals, axis=(0, 1), ord=2)     # :154
    return out.reshape(ny, nx)                                # unstack        :157


def sigma_max_closed_form(def_tensor, tensor_layout="reference"):
    """Closed-form 2x2 eigen solve the HIP kernel uses; float64.

    reference layout: rows r1=(a,b,c), r2=(d,e,f), r3=0 with
    (a..f)=(dXdx,dXdy,dYdx,dYdy,dZdx,dZdy); sigma^2 is the larger eigenvalue
    of the Gram matrix [[p,r],[r,q]].
    """
    a, b, c, d, e, f = (np.asarray(def_tensor[i], dtype=np.float64) for i in range(6))
    if tensor_layout == "reference":
        p = a * a + b * b + c * c
        q = d * d + e * e + f * f
        r = a * d + b * e + c * f
    elif tensor_layout == "physical":
        p = a * a + c * c + e * e
        q = b * b + d * d + f * f
        r = a * b + c * d + e * f
    else:
        raise ValueError(tensor_layout)
    disc = np.sqrt((p - q) ** 2 + 4 * r * r)
    return np.sqrt(0.5 * ((p + q) + disc))


# --------------------------------------------------------------------------
# a6  LCS.__call__ arithmetic, array level  (LCS/LCS.py:129-157)
# --------------------------------------------------------------------------
def lcs(U, V, lat, lon, timestep=1, SETTLS_order=0, interp_order=3,
        cyclic_xboundary=False, gauss_sigma=None, seed_lat=None, seed_lon=None,
        t0=0, nsteps=None, tensor_layout="reference", fd_fp32_cast=True,
        noncyclic_clamp="reference_outer"):
    """advect -> flowmap_gradient -> sigma_max; returns (sigma, x_dep, y_dep)."""
    x_dep, y_dep = parcel_propagation(U, V, lat, lon, timestep=timestep,
                                      SETTLS_order=SETTLS_order,
                                      interp_order=interp_order,
                                      cyclic_xboundary=cyclic_xboundary,
                                      seed_lat=seed_lat, seed_lon=seed_lon,
                                      t0=t0, nsteps=nsteps,
                                      noncyclic_clamp=noncyclic_clamp)       # :129-134
    slat = lat if seed_lat is None else seed_lat
    slon = lon if seed_lon is None else seed_lon
    dt = flowmap_gradient(x_dep, y_dep, np.asarray(slat), np.asarray(slon),
                          sigma=gauss_sigma, fd_fp32_cast=fd_fp32_cast)      # :142
    return sigma_max(dt, tensor_layout=tensor_layout), x_dep, y_dep          # :145-157


# --------------------------------------------------------------------------
# a7  examples/ideal_vortex.py:130-208 -- loop-faithful generator (small sizes)
# --------------------------------------------------------------------------
def ideal_vortex_loops(lat_min, lat_max, lon_min, lon_max, dx, dy, nt,
                       max_intensity=10, radius=5, center=None, u_c=0, v_c=0,
                       diag_factor=0, basic_zonal=2, k=0):
    """Triple loop exactly as examples/ideal_vortex.py:159-201 (u, v only).

    Returns ``(u, v, lats, lons)`` with u, v shaped ``(nt, ny, nx)``.
    Pure-Python loops: small grids only; pins the vectorised product
    generator ``lagrangiancoherence_amd.flows.ideal_vortex``.
    """
    lats = np.arange(lat_min, lat_max, dy)                    # :159
    lons = np.arange(lon_min, lon_max, dx)                    # :160
    nx = lons.shape[0]
    ny = lats.shape[0]
    u = np.zeros([ny, nx, nt])
    v = np.zeros([ny, nx, nt])
    for t in range(nt):                                       # :175
        for x in range(nx):
            for y in range(ny):
                new_x = lons[x] - center[0] - u_c * t         # :178
                if k > 0:
                    new_y = lats[y] - center[1] - v_c * np.sin(k * 2 * np.pi * t / nt)
                elif k == 0:
                    new_y = lats[y] - center[1] - v_c * t     # :182
                else:
                    raise ValueError("Meridional wavenumber k must be greater than zero.")
                distance = np.sqrt(new_x ** 2 + new_y ** 2)  # :185
                theta = np.arccos(new_y / (distance + 1e-8))  # :191
                if distance > radius:                         # :192-195
                    mag = max_intensity * radius ** 2 / (2 * distance)
                else:
                    mag = max_intensity * 0.5 * distance
                u[y, x, t] = np.cos(theta) * mag + basic_zonal            # :197
                if new_x < 0:                                              # :198-201
                    v[y, x, t] = np.sin(theta) * mag
                else:
                    v[y, x, t] = np.sin(theta + np.pi) * mag
    return (np.ascontiguousarray(u.transpose(2, 0, 1)),
            np.ascontiguousarray(v.transpose(2, 0, 1)), lats, lons)


# --------------------------------------------------------------------------
# Restatement of scipy's interpolation semantics (what the HIP gather does).
# Pinned against scipy.ndimage itself by KAT-3 / KAT-4.
# --------------------------------------------------------------------------
def _wrap_coord(c, n):
    """scipy NI_EXTEND_WRAP coordinate map: identity on [0, n-1], else period n-1."""
    c = np.array(c, dtype=np.float64)
    sz = n - 1
    lo = c < 0
    c[lo] += sz * (np.trunc(-c[lo] / sz) + 1)
    hi = c > n - 1
    c[hi] -= sz * np.trunc(c[hi] / sz)
    return c


def _mirror_index(i, n):
    s2 = 2 * n - 2
    i = np.abs(i) % s2
    return np.where(i > n - 1, s2 - i, i)


def _spline_poles(order):
    """scipy ni_splines.c get_filter_poles."""
    sq = np.sqrt
    return {2: [sq(8.0) - 3.0], 3: [sq(3.0) - 2.0],
            4: [sq(664.0 - sq(438976.0)) + sq(304.0) - 19.0, sq(664.0 + sq(438976.0)) - sq(304.0) - 19.0],
            5: [sq(67.5 - sq(4436.25)) + sq(26.25) - 6.5, sq(67.5 + sq(4436.25)) - sq(26.25) - 6.5]}[order]


def spline_prefilter_mirror(field, order=3):
    """B-spline coefficients of ``order`` (2..5), mirror boundary, both axes.

    Equals ``scipy.ndimage.spline_filter(field, order, mode='mirror')`` -- the
    prefilter scipy runs inside ``map_coordinates(order, mode='wrap')`` -- to 4e-15 for
    orders 2 and 3, to 1e-12 for orders 4 and 5 (scipy's pole constants differ in the last bit).
    """
    poles = _spline_poles(order)
    gain = 1.0
    for z in poles:
        gain *= (1 - z) * (1 - 1 / z)
    c = np.array(field, dtype=np.float64)
    for axis in (0, 1):
        c = np.moveaxis(c, axis, 0).copy()
        n = c.shape[0]
        c *= gain
        for z in poles:
            zn1 = z ** (n - 1)
            c0 = c[0] + zn1 * c[n - 1]
            zi = z
            for i in range(1, n - 1):
                c0 = c0 + zi * (c[i] + zn1 * c[n - 1 - i])
                zi *= z
            c[0] = c0 / (1 - zn1 * zn1)
            for i in range(1, n):
                c[i] += z * c[i - 1]
            c[n - 1] = (z / (z * z - 1)) * (c[n - 1] + z * c[n - 2])
            for i in range(n - 2, -1, -1):
                c[i] = z * (c[i + 1] - c[i])
        c = np.moveaxis(c, 0, axis)
    return c


def _bspline_centred(n, t):
    """beta_n(t) = 1/n! sum_k (-1)^k C(n+1, k) (t + (n+1)/2 - k)_+^n."""
    from math import comb, factorial
    s = 0.0
    for k in range(n + 2):
        a = t + (n + 1) / 2.0 - k
        s = s + (-1) ** k * comb(n + 1, k) * np.where(a > 0, a, 0.0) ** n
    return s / factorial(n)


def interp_restated(field, cy, cx, order, mode):
    """Index-space interpolation with scipy's semantics, no scipy call.

    ``mode='wrap'``: coordinate wrapped with period n-1, out-of-range taps
    mirrored; ``mode='constant'`` (order 1 only): exactly 0 outside [0, n-1].
    """
    field = np.asarray(field, dtype=np.float64)
    ny, nx = field.shape
    cy = np.asarray(cy, dtype=np.float64)
    cx = np.asarray(cx, dtype=np.float64)
    if mode == "constant":
        assert order == 1
        inside = (cy >= 0) & (cy <= ny - 1) & (cx >= 0) & (cx <= nx - 1)
        cy = np.where(inside, cy, 0.0)
        cx = np.where(inside, cx, 0.0)
    elif mode == "wrap":
        cy = _wrap_coord(cy, ny)
        cx = _wrap_coord(cx, nx)
        inside = np.ones(cy.shape, dtype=bool)
    else:
        raise ValueError(mode)
    y0 = np.floor(cy).astype(np.int64)
    x0 = np.floor(cx).astype(np.int64)
    ty = cy - y0
    tx = cx - x0
    if order == 1:
        wy = [1 - ty, ty]
        wx = [1 - tx, tx]
        off = 0
        coeff = field
    elif order == 3:
        def w(t):
            zc = 1 - t
            w0 = zc * zc * zc / 6
            w1 = (t * t * (t - 2) * 3 + 4) / 6
            w2 = (zc * zc * (zc - 2) * 3 + 4) / 6
            return [w0, w1, w2, 1.0 - w0 - w1 - w2]
        wy = w(ty)
        wx = w(tx)
        off = -1
        coeff = spline_prefilter_mirror(field)
    elif order in (2, 4, 5):
        # first tap: floor(c) - order//2 (odd) or floor(c + 1/2) - order//2 (even); centred B-spline weights
        if order % 2 == 0:
            y0 = np.floor(cy + 0.5).astype(np.int64)
            x0 = np.floor(cx + 0.5).astype(np.int64)
        off = -(order // 2)
        wy = [_bspline_centred(order, cy - (y0 + off + a)) for a in range(order + 1)]
        wx = [_bspline_centred(order, cx - (x0 + off + b)) for b in range(order + 1)]
        coeff = spline_prefilter_mirror(field, order)
    else:
        raise ValueError(order)
    out = np.zeros(cy.shape)
    for a, wa in enumerate(wy):
        yi = _mirror_index(y0 + off + a, ny)
        for b, wb in enumerate(wx):
            xi = _mirror_index(x0 + off + b, nx)
            out = out + wa * wb * coeff[yi, xi]
    return np.where(inside, out, 0.0)
