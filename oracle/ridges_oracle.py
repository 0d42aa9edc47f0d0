"""CPU oracle for tools.find_ridges_spherical_hessian (LCS/tools.py:52-155) -- TEST INFRASTRUCTURE ONLY.

SURVEY.md section 8f rank 4: the consumer of the FTLE field.  Restated on plain (latitude, longitude)
arrays with numpy/scipy, calling ``scipy.ndimage.gaussian_filter`` and ``numpy.linalg.eig`` exactly where
the reference does, and reproducing its quirks:

  R1  the Hessian's eigenVECTOR is taken as ``eig[1][argmin(eig[0])]`` -- a ROW of the eigenvector matrix,
      not a column (tools.py:107), so the value depends on LAPACK dgeev's ordering and sign conventions
      for a symmetric 2x2 (dlanv2: V = [[cs,-sn],[sn,cs]], w = (a', d'));
  R2  every derivative is taken of ``values.astype('float32')`` (tools.py:258), so first derivatives are
      float32-rounded before they are differenced again;
  R3  inf/NaN Hessian entries are zeroed (tools.py:92-93) but a NaN gradient is not: |NaN| <= tol and
      |NaN| > tol are both False, so such a point ends up flagged 1 (tools.py:136-137) if eigmin < 0;
  R4  ``eigmin`` is the eigenvalue of LARGEST magnitude (tools.py:114), despite its name.

Also here: ``dlanv2_sym`` -- the closed form of what dgeev returns for a symmetric 2x2, which the HIP
kernel implements; pinned against ``numpy.linalg.eig`` itself in tests/test_ridges.py.
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import gaussian_filter

from .lcs_oracle import derivative_spherical_coords

__all__ = ["find_ridges_spherical_hessian", "dlanv2_sym"]


def find_ridges_spherical_hessian(values, lat, lon, sigma=.5, tolerance_threshold=0.0005e-3, return_eigvectors=False,
                                  isglobal=True):
    """values: (nlat, nlon), lat/lon ascending.  Returns (ridge mask, eigmin, dt_prod_raw), each (nlat, nlon);
    with ``return_eigvectors`` also (eigvectors (2, nlat, nlon) zeroed where eigmin >= 0, gradient (2, nlat, nlon),
    angle (nlat, nlon)) -- the extra members of the reference's six-tuple (tools.py:123-133,140-147).
    tools.py:67-155; ``isglobal`` goes to every derivative_spherical_coords call as in tools.py:77-81."""
    da = np.asarray(values, dtype=np.float64)
    if isinstance(sigma, (float, int)):
        da = gaussian_filter(da, sigma=sigma)                                     # tools.py:74-75

    def D(a, dim):
        return derivative_spherical_coords(a, lat, lon, dim=dim, isglobal=isglobal)   # float32 cast inside (R2)
    ddadx, ddady = D(da, 1), D(da, 0)                                             # tools.py:77-78
    d2dadx2, d2dady2, d2dadxdy = D(ddadx, 1), D(ddady, 0), D(ddadx, 0)            # tools.py:79-81
    hess = np.stack([d2dadx2, d2dadxdy, d2dadxdy, d2dady2]).reshape(4, -1)        # tools.py:87-90
    grad = np.stack([ddadx, ddady]).reshape(2, -1)
    hess = np.where(np.abs(hess) != np.inf, hess, 0)                              # tools.py:92
    hess = np.where(~np.isnan(hess), hess, 0)                                     # tools.py:93
    H = hess.reshape(2, 2, -1).transpose(2, 0, 1)                                 # (N, 2, 2), tools.py:99
    w, V = np.linalg.eig(H)                                                       # per-point dgeev, tools.py:106
    n = np.arange(H.shape[0])
    eigvector = V[n, np.argmin(w, axis=1), :]                                     # ROW of V (R1), tools.py:107
    dt_angle = np.einsum("ni,in->n", eigvector, grad)                             # tools.py:115
    eigmin = w[n, np.argmax(np.abs(w), axis=1)]                                   # tools.py:118 (R4)
    dt = dt_angle
    mask = np.where(np.abs(dt) <= tolerance_threshold, dt, 0)                     # tools.py:136
    mask = np.where(np.abs(dt) > tolerance_threshold, mask, 1)                    # tools.py:137 (R3)
    mask = np.where(np.sign(eigmin) == -1, mask, 0)                               # tools.py:138
    shp = da.shape
    if return_eigvectors:
        ev = eigvector.T                                                          # (2, N), tools.py:123
        with np.errstate(divide="ignore", invalid="ignore"):
            angle = 180 / np.pi * np.arctan(ev[0] / ev[1])                        # tools.py:125 (before the masking)
        ev_masked = np.where(eigmin[None] < 0, ev, 0)                             # tools.py:133
        return (mask.reshape(shp), eigmin.reshape(shp), dt.reshape(shp), ev_masked.reshape((2,) + shp),
                grad.reshape((2,) + shp), angle.reshape(shp))
    return mask.reshape(shp), eigmin.reshape(shp), dt.reshape(shp)


def dlanv2_sym(a, b, d):
    """(w0, w1, V) with V[..., 2, 2] such that numpy.linalg.eig([[a,b],[b,d]]) == ((w0, w1), V).

    LAPACK dgeev on a symmetric 2x2 (no balancing, trivial Hessenberg reduction), i.e. dlahqr on a 2x2:
      * deflation test (Ahues & Tisseur, dlahqr): if |b| <= ulp*(|a|+|d|) and
        |b|*(|b|/s) <= max(smlnum, ulp*(bb*(aa/s))) the subdiagonal is zeroed: w = (a, d) and dtrevc
        solves the triangular [[a,b],[0,d]]: V = [[1, x/n],[0, 1/n]], x = b/(d-a), n = hypot(x,1);
      * otherwise one dlanv2 standardisation; T comes out diagonal so V = [[cs,-sn],[sn,cs]].
    dlanv2's other branch ("complex or real (almost) equal eigenvalues": z < 4*eps, i.e. |b| and |a-d|
    below ~1e-15 in ABSOLUTE terms) equalises the diagonal with one rotation and, b and c having equal
    signs, reduces to triangular form with a second one; dtrevc then solves [[A,B],[0,D]]."""
    a, b, d = (np.asarray(v, dtype=np.float64) for v in (a, b, d))
    ulp = np.finfo(np.float64).eps                     # dlamch('P')
    safmin = np.finfo(np.float64).tiny
    smlnum = safmin * (2.0 / ulp)                      # dlahqr: safmin*(nh/ulp), nh = 2
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        ab = np.abs(b)
        tst = np.abs(a) + np.abs(d)
        aa = np.maximum(np.abs(d), np.abs(a - d))
        bb = np.minimum(np.abs(d), np.abs(a - d))
        s_ = aa + ab
        deflate = (ab <= smlnum) | ((ab <= ulp * tst) & (ab * (ab / s_) <= np.maximum(smlnum, ulp * (bb * (aa / s_)))))
        # --- deflated: triangular eigenvectors (dtrevc + dlaln2 + dgeev normalisation)
        smin = np.maximum(ulp * np.abs(d), safmin * (2.0 / ulp))
        csr = a - d
        csr = np.where(np.abs(csr) < smin, smin, csr)
        x1 = (-b) / csr
        emax = np.maximum(np.abs(x1), 1.0)
        v0, v1 = x1 / emax, 1.0 / emax
        nrm = np.hypot(v0, v1)
        Vd = np.stack([np.stack([np.ones_like(a), v0 / nrm], -1), np.stack([np.zeros_like(a), v1 / nrm], -1)], -2)
        # --- dlanv2, real-eigenvalue branch
        p = 0.5 * (a - d)
        scale = np.maximum(np.abs(p), ab)
        z0 = (p / scale) * p + (ab / scale) * ab
        z = p + np.copysign(np.sqrt(scale) * np.sqrt(z0), p)
        w0 = d + z
        w1 = d - (ab / z) * ab
        big, small = np.maximum(np.abs(b), np.abs(z)), np.minimum(np.abs(b), np.abs(z))
        tau = big * np.sqrt(1.0 + (small / big) ** 2)  # dlapy2
        cs, sn = z / tau, b / tau
        Vr = np.stack([np.stack([cs, -sn], -1), np.stack([sn, cs], -1)], -2)
        # --- dlanv2, "almost equal eigenvalues" branch (symmetric input: c == b)
        sigma = b + b
        temp = a - d
        pp = 0.5 * temp
        big2, small2 = np.maximum(np.abs(sigma), np.abs(temp)), np.minimum(np.abs(sigma), np.abs(temp))
        tau2 = big2 * np.sqrt(1.0 + (small2 / big2) ** 2)
        cs2 = np.sqrt(0.5 * (1.0 + np.abs(sigma) / tau2))
        sn2 = -(pp / (tau2 * cs2)) * np.copysign(1.0, sigma)
        AA = a * cs2 + b * sn2
        BB = -a * sn2 + b * cs2
        CC = b * cs2 + d * sn2
        DD = -b * sn2 + d * cs2
        A2 = AA * cs2 + CC * sn2
        B2 = BB * cs2 + DD * sn2
        C2 = -AA * sn2 + CC * cs2
        D2 = -BB * sn2 + DD * cs2
        mid = 0.5 * (A2 + D2)
        same = (C2 != 0) & (B2 != 0) & (np.copysign(1.0, B2) == np.copysign(1.0, C2))
        sab, sac = np.sqrt(np.abs(B2)), np.sqrt(np.abs(C2))
        p3 = np.copysign(sab * sac, C2)
        tau3 = 1.0 / np.sqrt(np.abs(B2 + C2))
        cs1, sn1 = sab * tau3, sac * tau3
        # C2 != 0, B2 == 0: swap rows/columns;  C2 == 0: already triangular
        swap = (C2 != 0) & (B2 == 0)
        Af = np.where(same, mid + p3, mid)
        Df = np.where(same, mid - p3, mid)
        Bf = np.where(same, B2 - C2, np.where(swap, -C2, B2))
        csf = np.where(same, cs2 * cs1 - sn2 * sn1, np.where(swap, -sn2, cs2))
        snf = np.where(same, cs2 * sn1 + sn2 * cs1, np.where(swap, cs2, sn2))
        # eigenvectors of T = [[Af,Bf],[0,Df]] (dtrevc), back-transformed by Z = [[cs,-sn],[sn,cs]]
        smin3 = np.maximum(ulp * np.abs(Df), smlnum)
        csr3 = Af - Df
        csr3 = np.where(np.abs(csr3) < smin3, smin3, csr3)
        y1 = (-Bf) / csr3
        c0 = np.stack([csf, snf], -1)                                  # Z[:,0]
        c1 = np.stack([csf * y1 - snf, snf * y1 + csf], -1)            # Z[:,0]*y1 + Z[:,1]
        c0 = c0 / np.max(np.abs(c0), axis=-1, keepdims=True)
        c1 = c1 / np.max(np.abs(c1), axis=-1, keepdims=True)
        c0 = c0 / np.hypot(c0[..., 0], c0[..., 1])[..., None]
        c1 = c1 / np.hypot(c1[..., 0], c1[..., 1])[..., None]
        Vq = np.stack([c0, c1], -1)
    rare = ~deflate & ~(z0 >= 4.0 * ulp)
    w0 = np.where(deflate, a, np.where(rare, Af, w0))
    w1 = np.where(deflate, d, np.where(rare, Df, w1))
    V = np.where(deflate[..., None, None], Vd, np.where(rare[..., None, None], Vq, Vr))
    return w0, w1, V
