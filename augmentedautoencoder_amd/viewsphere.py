"""Codebook row -> rotation mapping (host NumPy, float64).

Mirrors what the reference derives in ``Dataset.viewsphere_for_embedding``
(/root/reference/auto_pose/ae/dataset.py:39-58) from the sixd-toolkit view
sampler (/root/reference/auto_pose/ae/pysixd_stuff/view_sampler.py:19-188):
row ``i`` of the codebook is view ``i // num_cyclo`` of a refined icosahedron,
rotated in-plane by ``-linspace(0, 2*pi, num_cyclo)[i % num_cyclo]``.

The *order* of the icosphere points is part of the contract (it decides which
rotation a codebook index means).  It depends on a breadth-first walk whose
frontier is de-duplicated through a Python ``set`` and then stably sorted by
azimuth (view_sampler.py:92-105), so the same container operations are kept
here.  tests/test_viewsphere.py pins the result against fixtures produced by
the reference's own importable view_sampler.py (tests/golden/).
"""
from __future__ import annotations

import math

import numpy as np

_TWO_PI = 2.0 * math.pi


def _icosahedron():
    g = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [(-1.0, g, 0.0), (1.0, g, 0.0), (-1.0, -g, 0.0), (1.0, -g, 0.0),
             (0.0, -1.0, g), (0.0, 1.0, g), (0.0, -1.0, -g), (0.0, 1.0, -g),
             (g, 0.0, -1.0), (g, 0.0, 1.0), (-g, 0.0, -1.0), (-g, 0.0, 1.0)]
    tris = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
            (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
            (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    return verts, tris


def icosphere_points(min_n_pts, radius=1.0):
    """Refine an icosahedron (each face -> 4) until it has >= min_n_pts vertices,
    project to the sphere, and order the vertices top-down by a breadth-first
    walk with every frontier sorted by azimuth (view_sampler.py:19-120).
    Returns (points [P,3] float64, refinement level per point)."""
    pts, faces = _icosahedron()
    level_of = [0] * len(pts)
    level = 0
    while len(pts) < min_n_pts:
        level += 1
        midpoint_id = {}
        refined = []
        for tri in faces:
            ids = list(tri)
            for e in range(3):
                a, b = tri[e], tri[(e + 1) % 3]
                key = (a, b) if a < b else (b, a)
                if key not in midpoint_id:
                    midpoint_id[key] = len(pts)
                    mid = 0.5 * (np.array(pts[key[0]]) + np.array(pts[key[1]]))
                    pts.append(mid.tolist())
                    level_of.append(level)
                ids.append(midpoint_id[key])
            refined += [(ids[0], ids[3], ids[5]), (ids[3], ids[1], ids[4]),
                        (ids[3], ids[4], ids[5]), (ids[5], ids[4], ids[2])]
        faces = refined

    P = np.array(pts)
    P *= np.reshape(radius / np.linalg.norm(P, axis=1), (P.shape[0], 1))

    neighbours = {}
    for tri in faces:
        for e in range(3):
            neighbours.setdefault(tri[e], set()).add(tri[(e + 1) % 3])
            neighbours[tri[e]].add(tri[(e + 2) % 3])

    def azimuth_of(i):
        return (math.atan2(P[i][1], P[i][0]) + _TWO_PI) % _TWO_PI

    order = []
    visited = [False] * P.shape[0]
    frontier = [np.argmax(P[:, 2])]
    while len(order) != P.shape[0]:
        frontier = sorted(frontier, key=azimuth_of)      # stable: ties keep set order
        reached = []
        for i in frontier:
            order.append(i)
            visited[i] = True
            reached += [j for j in neighbours[i]]
        frontier = [j for j in set(reached) if not visited[j]]

    order = np.array(order)
    return P[order, :], [level_of[i] for i in order]


def _rotation_about(angle, axis):
    """Axis-angle rotation, same arithmetic as the Gohlke helper the sampler calls
    (pysixd_stuff/transform.py:327-336): diag(cos) + outer(d,d)(1-cos) + sin*[d]x."""
    s, c = math.sin(angle), math.cos(angle)
    d = np.array(axis[:3], dtype=np.float64, copy=True)
    d /= math.sqrt(np.dot(d, d))
    R = np.diag([c, c, c])
    R += np.outer(d, d) * (1.0 - c)
    d *= s
    R += np.array([[0.0, -d[2], d[1]],
                   [d[2], 0.0, -d[0]],
                   [-d[1], d[0], 0.0]])
    return R


def sample_views(min_n_views, radius=1.0,
                 azimuth_range=(0, _TWO_PI), elev_range=(-0.5 * math.pi, 0.5 * math.pi)):
    """Look-at rotations (OpenCV convention) for the icosphere points
    (view_sampler.py:122-188).  Returns (list of {'R','t'}, levels)."""
    pts, levels = icosphere_points(min_n_views, radius=radius)
    flip_yz = _rotation_about(math.pi, [1, 0, 0])
    views = []
    for p in pts:
        az = math.atan2(p[1], p[0])
        if az < 0:
            az += _TWO_PI
        el = math.acos(np.linalg.norm([p[0], p[1], 0]) / np.linalg.norm(p))
        if p[2] < 0:
            el = -el
        if not (azimuth_range[0] <= az <= azimuth_range[1] and elev_range[0] <= el <= elev_range[1]):
            continue
        fwd = -np.array(p)
        fwd /= np.linalg.norm(fwd)
        up = np.array([0.0, 0.0, 1.0])
        side = np.cross(fwd, up)
        if np.count_nonzero(side) == 0:          # looking straight along +-Z
            side = np.array([1.0, 0.0, 0.0])
        side /= np.linalg.norm(side)
        up = np.cross(side, fwd)
        R = np.array([[side[0], side[1], side[2]],
                      [up[0], up[1], up[2]],
                      [-fwd[0], -fwd[1], -fwd[2]]])
        R = flip_yz.dot(R)                        # OpenGL -> OpenCV camera frame
        t = -R.dot(np.array(p).reshape((3, 1)))
        views.append({'R': R, 't': t})
    return views, levels


def viewsphere_for_embedding(min_n_views, radius, num_cyclo):
    """[N,3,3] float64, N = n_views*num_cyclo (dataset.py:39-58).  Note that
    linspace(0, 2pi, num_cyclo) includes both endpoints, so rows 36k and 36k+35
    are the same rotation (structural near-ties in every real codebook)."""
    views, _ = sample_views(int(min_n_views), float(radius))
    num_cyclo = int(num_cyclo)
    Rs = np.empty((len(views) * num_cyclo, 3, 3))
    i = 0
    for v in views:
        for cyclo in np.linspace(0, 2. * np.pi, num_cyclo):
            c, s = np.cos(-cyclo), np.sin(-cyclo)
            rot_z = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
            Rs[i, :, :] = rot_z.dot(v['R'])
            i += 1
    return Rs
