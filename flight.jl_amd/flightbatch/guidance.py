"""Host-side helpers for segment guidance scenarios: the `Segment` constructors of the reference
(lib/FlightApps/src/c172/c172x/guidance/c172x_gdc.jl:33-89) in numpy, vectorised over aircraft. Points are
(latitude, longitude, ellipsoidal altitude) triples, arrays [3] or [3, n]. The guidance law itself runs on the GPU
(k_x2_ctl); these functions only build its inputs."""
from __future__ import annotations

import numpy as np

A, F = 6378137.0, 1.0 / 298.257223563          # WGS-84 (FP/geodesy.jl:15-35)
E2 = 2 * F - F * F


def ecef(p):
    lat, lon, h = np.asarray(p, dtype=np.float64)
    n = np.array([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)])
    N = A / np.sqrt(1 - E2 * n[2] ** 2)
    return np.array([(N + h) * n[0], (N + h) * n[1], (N * (1 - E2) + h) * n[2]])


def ned_axes(lat, lon):
    """Unit vectors of the local-level frame (north, east, down) in ECEF coordinates."""
    sl, cl, so, co = np.sin(lat), np.cos(lat), np.sin(lon), np.cos(lon)
    north = np.array([-sl * co, -sl * so, cl]); east = np.array([-so, co, 0 * so]); down = np.array([-cl * co, -cl * so, -sl])
    return north, east, down


def latlon_of(r):
    """Geodetic latitude / longitude of an ECEF position (Bowring's iteration, converged to double precision)."""
    x, y, z = r
    p = np.hypot(x, y)
    lat = np.arctan2(z, p * (1 - E2))
    for _ in range(6):
        N = A / np.sqrt(1 - E2 * np.sin(lat) ** 2)
        lat = np.arctan2(z + E2 * N * np.sin(lat), p)
    return lat, np.arctan2(y, x)


def segment_end(p1, s, χ, γ=None, Δh=None):
    """Segment(p1; s, χ, γ | Δh).p2 (c172x_gdc.jl:56-83): s metres along azimuth χ in the local-level frame of p1."""
    p1 = np.asarray(p1, dtype=np.float64)
    if (γ is None) == (Δh is None):
        raise ValueError("give either γ (flight path angle) or Δh (altitude increment)")
    dh = s * np.tan(γ) if Δh is None else Δh
    north, east, _ = ned_axes(p1[0], p1[1])
    r2 = ecef(p1) + s * np.cos(χ) * north + s * np.sin(χ) * east
    lat, lon = latlon_of(r2)
    return np.array([lat, lon, p1[2] + dh + 0 * lat])


class Segment:
    """Segment(p1, p2); `-seg` swaps the end points (c172x_gdc.jl:87)."""

    def __init__(self, p1, p2):
        self.p1, self.p2 = np.asarray(p1, dtype=np.float64), np.asarray(p2, dtype=np.float64)

    @classmethod
    def from_origin(cls, p1, s, χ, γ=None, Δh=None):
        return cls(p1, segment_end(p1, s, χ, γ, Δh))

    def __neg__(self):
        return Segment(self.p2, self.p1)
