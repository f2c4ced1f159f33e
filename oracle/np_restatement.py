"""Independent numpy restatement of the escape-time math, used ONLY to pin the C oracle.

TEST INFRASTRUCTURE ONLY (see oracle/fr_oracle.h, which states what pins the oracle to the reference).
Written separately from fr_oracle.c (vectorised, masked updates instead of a scalar
loop with break) so that an error in one restatement is unlikely to be repeated in the
other.  Follows shaders/mandelbrot.comp:147-177, shaders/julia.comp:216-249,325 and
shaders/burning_ship.comp:217-256,322-325.
numpy performs one IEEE operation per ufunc call, so no contraction can occur.
"""
from __future__ import annotations

import numpy as np


def mandelbrot(W, H, center_x, center_y, zoom, max_iter, bailout=4.0, dtype=np.float64):
    T = dtype
    px, py = np.meshgrid(np.arange(W, dtype=T), np.arange(H, dtype=T))
    resx, resy = T(W), T(H)
    uvx = (px - T(0.5) * resx) / resy                       # mandelbrot.comp:150
    uvy = (py - T(0.5) * resy) / resy
    cx = T(center_x) + uvx * T(zoom)                        # :151
    cy = T(center_y) + uvy * T(zoom)
    return _iterate(np.zeros_like(cx), np.zeros_like(cy), cx, cy, max_iter, T(bailout), T, julia=False)


def julia(W, H, center_x, center_y, zoom, max_iter, c_re, c_im, bailout=4.0, dtype=np.float64):
    T = dtype
    px, py = np.meshgrid(np.arange(W, dtype=T), np.arange(H, dtype=T))
    uvx, uvy = px / T(W), py / T(H)                         # julia.comp:325
    aspect = T(W) / T(H)                                    # :221
    zx = T(center_x) + (uvx - T(0.5)) * T(zoom) * aspect    # :223
    zy = T(center_y) + (uvy - T(0.5)) * T(zoom)             # :224
    cx = np.full_like(zx, T(c_re))
    cy = np.full_like(zy, T(c_im))
    return _iterate(zx, zy, cx, cy, max_iter, T(bailout), T, julia=True)


def burning_ship(W, H, center_x, center_y, zoom, max_iter, bailout=4.0, dtype=np.float64):
    T = dtype
    px, py = np.meshgrid(np.arange(W, dtype=T), np.arange(H, dtype=T))
    uvx, uvy = px / T(W), py / T(H)                         # burning_ship.comp:393
    aspect = T(W) / T(H)                                    # :322
    cx = T(center_x) + (uvx - T(0.5)) * T(zoom) * aspect    # :324
    cy = T(center_y) + (uvy - T(0.5)) * T(zoom)             # :325
    return _iterate(np.zeros_like(cx), np.zeros_like(cy), cx, cy, max_iter, T(bailout), T, julia=True, fold=True)


def _iterate(zx, zy, cx, cy, max_iter, bailout, T, julia, fold=False):
    B2 = bailout * bailout
    it = np.full(zx.shape, max_iter, np.int32)
    ezx = zx.copy()
    ezy = zy.copy()
    live = np.ones(zx.shape, bool)
    with np.errstate(over="ignore", invalid="ignore"):
        for i in range(max_iter):
            if not live.any():
                break
            ax, ay = (np.abs(zx), np.abs(zy)) if fold else (zx, zy)   # burning_ship.comp:241
            x = ax * ax - ay * ay + cx                      # as written, one rounding per op
            y = T(2.0) * ax * ay + cy
            zx = np.where(live, x, zx)
            zy = np.where(live, y, zy)
            esc = live & (zx * zx + zy * zy > B2)
            it[esc] = i
            ezx[esc] = zx[esc]
            ezy[esc] = zy[esc]
            live &= ~esc
    ezx[live] = zx[live]
    ezy[live] = zy[live]
    r2 = ezx * ezx + ezy * ezy
    nu = it.astype(T)
    e = it < max_iter
    with np.errstate(divide="ignore", invalid="ignore"):
        if julia:                                           # julia.comp:238, burning_ship.comp:252
            sm = it.astype(T) + T(1.0) - np.log(np.log(r2) / np.log(bailout)) / np.log(T(2.0))
        else:                                               # mandelbrot.comp:174-176
            log_zn = np.log(r2) / T(2.0)
            mu = np.log(log_zn / np.log(T(2.0))) / np.log(T(2.0))
            sm = it.astype(T) + T(1.0) - mu
    nu = np.where(e, sm, nu).astype(T)
    return it, nu, ezx, ezy
