"""The CLASSIC 256-case marching cubes (Lorensen & Cline 1987 as tabulated by C. G. Bloyd / P. Bourke, "Polygonising a
scalar field", 1994) - TEST INFRASTRUCTURE ONLY.

Why it is here: above 256^3 the reference extracts its mesh with PyMCubes (``mcubes.marching_cubes(final.numpy(), 0.5)``,
lib/common/seg3d_lossless.py:592) and at 256^3 with kaolin's ``voxelgrids_to_trianglemeshes`` (:599); neither package is in
this image and the reference tree holds no output of either, so the TRIANGULATION of the product's marching cubes stays
"parity unpinned" (DESIGN.md section 4.5).  PyMCubes' C++ core is the classic algorithm with Bourke's published edgeTable /
triTable (public domain) - this file restates that published algorithm so that the gap becomes a LIST: which of the 256 cube
configurations the product's generated table triangulates with a different triangle set, a different surface-loop structure,
or identically (tests/test_mesh_tools.py::test_product_table_vs_classic_table).

The table below is written out from the published one and is VALIDATED MECHANICALLY by validate_table() rather than trusted:
every triangle corner lies on a cut edge of its case, every cut edge is used, every case's triangles form closed fans (each
triangle edge is shared by two triangles or lies in a cube face), the segments a case leaves in a cube face depend only on
that face's four corner bits (so neighbouring cubes agree and the surface of a volume is closed), plus closedness of the
extracted surface on random volumes (tests).

Conventions (Bourke): corner m of a cube at (x, y, z) offsets
    0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0) 4:(0,0,1) 5:(1,0,1) 6:(1,1,1) 7:(0,1,1)
edge e joins EDGE_CORNERS[e]; bit m of the case index is set when corner m is "set".  Bourke sets a corner whose value lies
BELOW the level; PyMCubes' core does the same as far as recalled (its source is not available here) - compare with the
product under BOTH readings (``set_is_inside``), the listing reports both.
"""
from __future__ import annotations

import numpy as np

CORNERS = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.int64)
EDGE_CORNERS = np.array([[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6], [6, 7], [7, 4], [0, 4], [1, 5], [2, 6], [3, 7]], np.int64)
# the six faces as corner quadruples in cyclic order
FACES = np.array([[0, 1, 2, 3], [4, 5, 6, 7], [0, 1, 5, 4], [3, 2, 6, 7], [0, 3, 7, 4], [1, 2, 6, 5]], np.int64)

_T = """
-
0 8 3
0 1 9
1 8 3 9 8 1
1 2 10
0 8 3 1 2 10
9 2 10 0 2 9
2 8 3 2 10 8 10 9 8
3 11 2
0 11 2 8 11 0
1 9 0 2 3 11
1 11 2 1 9 11 9 8 11
3 10 1 11 10 3
0 10 1 0 8 10 8 11 10
3 9 0 3 11 9 11 10 9
9 8 10 10 8 11
4 7 8
4 3 0 7 3 4
0 1 9 8 4 7
4 1 9 4 7 1 7 3 1
1 2 10 8 4 7
3 4 7 3 0 4 1 2 10
9 2 10 9 0 2 8 4 7
2 10 9 2 9 7 2 7 3 7 9 4
8 4 7 3 11 2
11 4 7 11 2 4 2 0 4
9 0 1 8 4 7 2 3 11
4 7 11 9 4 11 9 11 2 9 2 1
3 10 1 3 11 10 7 8 4
1 11 10 1 4 11 1 0 4 7 11 4
4 7 8 9 0 11 9 11 10 11 0 3
4 7 11 4 11 9 9 11 10
9 5 4
9 5 4 0 8 3
0 5 4 1 5 0
8 5 4 8 3 5 3 1 5
1 2 10 9 5 4
3 0 8 1 2 10 4 9 5
5 2 10 5 4 2 4 0 2
2 10 5 3 2 5 3 5 4 3 4 8
9 5 4 2 3 11
0 11 2 0 8 11 4 9 5
0 5 4 0 1 5 2 3 11
2 1 5 2 5 8 2 8 11 4 8 5
10 3 11 10 1 3 9 5 4
4 9 5 0 8 1 8 10 1 8 11 10
5 4 0 5 0 11 5 11 10 11 0 3
5 4 8 5 8 10 10 8 11
9 7 8 5 7 9
9 3 0 9 5 3 5 7 3
0 7 8 0 1 7 1 5 7
1 5 3 3 5 7
9 7 8 9 5 7 10 1 2
10 1 2 9 5 0 5 3 0 5 7 3
8 0 2 8 2 5 8 5 7 10 5 2
2 10 5 2 5 3 3 5 7
7 9 5 7 8 9 3 11 2
9 5 7 9 7 2 9 2 0 2 7 11
2 3 11 0 1 8 1 7 8 1 5 7
11 2 1 11 1 7 7 1 5
9 5 8 8 5 7 10 1 3 10 3 11
5 7 0 5 0 9 7 11 0 1 0 10 11 10 0
11 10 0 11 0 3 10 5 0 8 0 7 5 7 0
11 10 5 7 11 5
10 6 5
0 8 3 5 10 6
9 0 1 5 10 6
1 8 3 1 9 8 5 10 6
1 6 5 2 6 1
1 6 5 1 2 6 3 0 8
9 6 5 9 0 6 0 2 6
5 9 8 5 8 2 5 2 6 3 2 8
2 3 11 10 6 5
11 0 8 11 2 0 10 6 5
0 1 9 2 3 11 5 10 6
5 10 6 1 9 2 9 11 2 9 8 11
6 3 11 6 5 3 5 1 3
0 8 11 0 11 5 0 5 1 5 11 6
3 11 6 0 3 6 0 6 5 0 5 9
6 5 9 6 9 11 11 9 8
5 10 6 4 7 8
4 3 0 4 7 3 6 5 10
1 9 0 5 10 6 8 4 7
10 6 5 1 9 7 1 7 3 7 9 4
6 1 2 6 5 1 4 7 8
1 2 5 5 2 6 3 0 4 3 4 7
8 4 7 9 0 5 0 6 5 0 2 6
7 3 9 7 9 4 3 2 9 5 9 6 2 6 9
3 11 2 7 8 4 10 6 5
5 10 6 4 7 2 4 2 0 2 7 11
0 1 9 4 7 8 2 3 11 5 10 6
9 2 1 9 11 2 9 4 11 7 11 4 5 10 6
8 4 7 3 11 5 3 5 1 5 11 6
5 1 11 5 11 6 1 0 11 7 11 4 0 4 11
0 5 9 0 6 5 0 3 6 11 6 3 8 4 7
6 5 9 6 9 11 4 7 9 7 11 9
10 4 9 6 4 10
4 10 6 4 9 10 0 8 3
10 0 1 10 6 0 6 4 0
8 3 1 8 1 6 8 6 4 6 1 10
1 4 9 1 2 4 2 6 4
3 0 8 1 2 9 2 4 9 2 6 4
0 2 4 4 2 6
8 3 2 8 2 4 4 2 6
10 4 9 10 6 4 11 2 3
0 8 2 2 8 11 4 9 10 4 10 6
3 11 2 0 1 6 0 6 4 6 1 10
6 4 1 6 1 10 4 8 1 2 1 11 8 11 1
9 6 4 9 3 6 9 1 3 11 6 3
8 11 1 8 1 0 11 6 1 9 1 4 6 4 1
3 11 6 3 6 0 0 6 4
6 4 8 11 6 8
7 10 6 7 8 10 8 9 10
0 7 3 0 10 7 0 9 10 6 7 10
10 6 7 1 10 7 1 7 8 1 8 0
10 6 7 10 7 1 1 7 3
1 2 6 1 6 8 1 8 9 8 6 7
2 6 9 2 9 1 6 7 9 0 9 3 7 3 9
7 8 0 7 0 6 6 0 2
7 3 2 6 7 2
2 3 11 10 6 8 10 8 9 8 6 7
2 0 7 2 7 11 0 9 7 6 7 10 9 10 7
1 8 0 1 7 8 1 10 7 6 7 10 2 3 11
11 2 1 11 1 7 10 6 1 6 7 1
8 9 6 8 6 7 9 1 6 11 6 3 1 3 6
0 9 1 11 6 7
7 8 0 7 0 6 3 11 0 11 6 0
7 11 6
7 6 11
3 0 8 11 7 6
0 1 9 11 7 6
8 1 9 8 3 1 11 7 6
10 1 2 6 11 7
1 2 10 3 0 8 6 11 7
2 9 0 2 10 9 6 11 7
6 11 7 2 10 3 10 8 3 10 9 8
7 2 3 6 2 7
7 0 8 7 6 0 6 2 0
2 7 6 2 3 7 0 1 9
1 6 2 1 8 6 1 9 8 8 7 6
10 7 6 10 1 7 1 3 7
10 7 6 1 7 10 1 8 7 1 0 8
0 3 7 0 7 10 0 10 9 6 10 7
7 6 10 7 10 8 8 10 9
6 8 4 11 8 6
3 6 11 3 0 6 0 4 6
8 6 11 8 4 6 9 0 1
9 4 6 9 6 3 9 3 1 11 3 6
6 8 4 6 11 8 2 10 1
1 2 10 3 0 11 0 6 11 0 4 6
4 11 8 4 6 11 0 2 9 2 10 9
10 9 3 10 3 2 9 4 3 11 3 6 4 6 3
8 2 3 8 4 2 4 6 2
0 4 2 4 6 2
1 9 0 2 3 4 2 4 6 4 3 8
1 9 4 1 4 2 2 4 6
8 1 3 8 6 1 8 4 6 6 10 1
10 1 0 10 0 6 6 0 4
4 6 3 4 3 8 6 10 3 0 3 9 10 9 3
10 9 4 6 10 4
4 9 5 7 6 11
0 8 3 4 9 5 11 7 6
5 0 1 5 4 0 7 6 11
11 7 6 8 3 4 3 5 4 3 1 5
9 5 4 10 1 2 7 6 11
6 11 7 1 2 10 0 8 3 4 9 5
7 6 11 5 4 10 4 2 10 4 0 2
3 4 8 3 5 4 3 2 5 10 5 2 11 7 6
7 2 3 7 6 2 5 4 9
9 5 4 0 8 6 0 6 2 6 8 7
3 6 2 3 7 6 1 5 0 5 4 0
6 2 8 6 8 7 2 1 8 4 8 5 1 5 8
9 5 4 10 1 6 1 7 6 1 3 7
1 6 10 1 7 6 1 0 7 8 7 0 9 5 4
4 0 10 4 10 5 0 3 10 6 10 7 3 7 10
7 6 10 7 10 8 5 4 10 4 8 10
6 9 5 6 11 9 11 8 9
3 6 11 0 6 3 0 5 6 0 9 5
0 11 8 0 5 11 0 1 5 5 6 11
6 11 3 6 3 5 5 3 1
1 2 10 9 5 11 9 11 8 11 5 6
0 11 3 0 6 11 0 9 6 5 6 9 1 2 10
11 8 5 11 5 6 8 0 5 10 5 2 0 2 5
6 11 3 6 3 5 2 10 3 10 5 3
5 8 9 5 2 8 5 6 2 3 8 2
9 5 6 9 6 0 0 6 2
1 5 8 1 8 0 5 6 8 3 8 2 6 2 8
1 5 6 2 1 6
1 3 6 1 6 10 3 8 6 5 6 9 8 9 6
10 1 0 10 0 6 9 5 0 5 6 0
0 3 8 5 6 10
10 5 6
11 5 10 7 5 11
11 5 10 11 7 5 8 3 0
5 11 7 5 10 11 1 9 0
10 7 5 10 11 7 9 8 1 8 3 1
11 1 2 11 7 1 7 5 1
0 8 3 1 2 7 1 7 5 7 2 11
9 7 5 9 2 7 9 0 2 2 11 7
7 5 2 7 2 11 5 9 2 3 2 8 9 8 2
2 5 10 2 3 5 3 7 5
8 2 0 8 5 2 8 7 5 10 2 5
9 0 1 5 10 3 5 3 7 3 10 2
9 8 2 9 2 1 8 7 2 10 2 5 7 5 2
1 3 5 3 7 5
0 8 7 0 7 1 1 7 5
9 0 3 9 3 5 5 3 7
9 8 7 5 9 7
5 8 4 5 10 8 10 11 8
5 0 4 5 11 0 5 10 11 11 3 0
0 1 9 8 4 10 8 10 11 10 4 5
10 11 4 10 4 5 11 3 4 9 4 1 3 1 4
2 5 1 2 8 5 2 11 8 4 5 8
0 4 11 0 11 3 4 5 11 2 11 1 5 1 11
0 2 5 0 5 9 2 11 5 4 5 8 11 8 5
9 4 5 2 11 3
2 5 10 3 5 2 3 4 5 3 8 4
5 10 2 5 2 4 4 2 0
3 10 2 3 5 10 3 8 5 4 5 8 0 1 9
5 10 2 5 2 4 1 9 2 9 4 2
8 4 5 8 5 3 3 5 1
0 4 5 1 0 5
8 4 5 8 5 3 9 0 5 0 3 5
9 4 5
4 11 7 4 9 11 9 10 11
0 8 3 4 9 7 9 11 7 9 10 11
1 10 11 1 11 4 1 4 0 7 4 11
3 1 4 3 4 8 1 10 4 7 4 11 10 11 4
4 11 7 9 11 4 9 2 11 9 1 2
9 7 4 9 11 7 9 1 11 2 11 1 0 8 3
11 7 4 11 4 2 2 4 0
11 7 4 11 4 2 8 3 4 3 2 4
2 9 10 2 7 9 2 3 7 7 4 9
9 10 7 9 7 4 10 2 7 8 7 0 2 0 7
3 7 10 3 10 2 7 4 10 1 10 0 4 0 10
1 10 2 8 7 4
4 9 1 4 1 7 7 1 3
4 9 1 4 1 7 0 8 1 8 7 1
4 0 3 7 4 3
4 8 7
9 10 8 10 11 8
3 0 9 3 9 11 11 9 10
0 1 10 0 10 8 8 10 11
3 1 10 11 3 10
1 2 11 1 11 9 9 11 8
3 0 9 3 9 11 1 2 9 2 11 9
0 2 11 8 0 11
3 2 11
2 3 8 2 8 10 10 8 9
9 10 2 0 9 2
2 3 8 2 8 10 0 1 8 1 10 8
1 10 2
1 3 8 9 1 8
0 9 1
0 3 8
-
"""


def _parse():
    rows = [ln.strip() for ln in _T.strip().splitlines()]
    assert len(rows) == 256, len(rows)
    out = []
    for ln in rows:
        v = [] if ln == "-" else [int(x) for x in ln.split()]
        assert len(v) % 3 == 0 and len(v) <= 15
        out.append(np.array(v, np.int64).reshape(-1, 3))
    return out


TRI_TABLE = _parse()          # [256] arrays [T,3] of cube-edge ids
_FLAT_TABLE = _EDGE_LO = _EDGE_AXIS = None      # filled below, once _flat_table is defined


def cut_edges(case: int):
    """edge ids whose two corners differ in `case`"""
    b = (case >> EDGE_CORNERS) & 1
    return set(np.nonzero(b[:, 0] != b[:, 1])[0].tolist())


def _edge_faces():
    """cube edge -> the two cube faces (indices into FACES) it lies in"""
    out = {}
    for e, (a, b) in enumerate(EDGE_CORNERS):
        out[e] = [k for k, f in enumerate(FACES) if a in f and b in f]
        assert len(out[e]) == 2
    return out


EDGE_FACES = _edge_faces()


def face_segments(tris: np.ndarray):
    """the triangle sides that lie IN a cube face (both end edges belong to that face) and are not shared by two triangles:
    -> {face: sorted list of (edge, edge)}; and the list of interior sides that are not matched (must be empty)"""
    from collections import Counter
    sides = Counter()
    for t in tris.tolist():
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            sides[(min(a, b), max(a, b))] += 1
    seg, bad = {}, []
    for (a, b), n in sides.items():
        common = set(EDGE_FACES[a]) & set(EDGE_FACES[b])
        if n == 2:
            continue
        if n == 1 and len(common) == 1:
            seg.setdefault(next(iter(common)), []).append((a, b))
        else:
            bad.append((a, b, n))
    return {k: sorted(v) for k, v in seg.items()}, bad


def validate_table(table=None):
    """Mechanical validation (see the module docstring).  -> dict of findings; ``ok`` when nothing is wrong."""
    table = TRI_TABLE if table is None else table
    problems = []
    face_rule = {}                      # (face id, 4 corner bits in the face's cyclic order) -> segments as (corner pair, corner pair)
    for c in range(256):
        tris = np.asarray(table[c], np.int64).reshape(-1, 3)
        cut = cut_edges(c)
        used = set(tris.reshape(-1).tolist())
        if not used <= cut:
            problems.append((c, "triangle corner on an edge that is not cut", sorted(used - cut)))
        if used != cut:
            problems.append((c, "cut edge without a vertex", sorted(cut - used)))
        if any(len(set(t)) != 3 for t in tris.tolist()):
            problems.append((c, "degenerate triangle", None))
        seg, bad = face_segments(tris)
        if bad:
            problems.append((c, "triangle side neither shared by two triangles nor inside one cube face", bad))
        for k, f in enumerate(FACES):
            bits = tuple(int((c >> m) & 1) for m in f)
            # a segment as the pair of face SIDES it joins: side j = the cube edge between face corners j and j+1
            side_of = {}
            for j in range(4):
                a, b = f[j], f[(j + 1) % 4]
                e = next(e for e, (p, q) in enumerate(EDGE_CORNERS) if {p, q} == {a, b})
                side_of[e] = j
            s = sorted(tuple(sorted((side_of[a], side_of[b]))) for a, b in seg.get(k, []))
            n_cut = sum(bits[j] != bits[(j + 1) % 4] for j in range(4))
            if 2 * len(s) != n_cut:
                problems.append((c, f"face {k}: {len(s)} segments for {n_cut} cut sides", s))
            key = bits
            if key in face_rule and face_rule[key] != s:
                problems.append((c, f"face {k} corner pattern {bits}: segments {s} here, {face_rule[key]} elsewhere (neighbouring cubes would disagree)", None))
            face_rule.setdefault(key, s)
    # which corners does the table separate on an ambiguous face (1,0,1,0)?
    amb = face_rule.get((1, 0, 1, 0))
    return {"ok": not problems, "problems": problems, "ambiguous_face_segments": amb,
            "separates_set_corners": amb == [(0, 3), (1, 2)]}


def loops(tris: np.ndarray):
    """the closed polygons a case's surface cuts out of the cube faces, as frozensets of cube-edge ids - what the triangulation
    of a case cannot change (it only picks diagonals): the TOPOLOGICAL content of a table row"""
    seg, _ = face_segments(np.asarray(tris, np.int64).reshape(-1, 3))
    adj = {}
    for lst in seg.values():
        for a, b in lst:
            adj.setdefault(a, []).append(b)
            adj.setdefault(b, []).append(a)
    seen, out = set(), []
    for s in sorted(adj):
        if s in seen:
            continue
        comp, stack = set(), [s]
        while stack:
            u = stack.pop()
            if u in comp:
                continue
            comp.add(u)
            stack.extend(adj[u])
        seen |= comp
        out.append(frozenset(comp))
    return sorted(out, key=lambda s: sorted(s))


def _cell_triangles(v: np.ndarray, level: float, set_below: bool, origin, dims, equal_is_set: bool = True):
    """the triangles of the cells of the sub-volume ``v`` (float64, its element [0,0,0] is lattice point ``origin`` of a volume
    of ``dims`` points): [T,3] GLOBAL keys of their vertices - a cut lattice edge is named by (its lower end point, its axis)"""
    n0, n1, n2 = v.shape
    # a value EXACTLY at the level: PyMCubes sets the bit (marchingcubes.h, as recalled: `if (v[m] <= isovalue) cubeindex |= 1 << m`),
    # Bourke's text does not (`<`).  It matters for a handful of the 1.4e8 values of a 513^3 volume (degenerate triangles there)
    if set_below:
        s = (v <= level) if equal_is_set else (v < level)
    else:
        s = (v >= level) if equal_is_set else (v > level)
    idx = np.zeros((n0 - 1, n1 - 1, n2 - 1), np.int64)
    for m, (dx, dy, dz) in enumerate(CORNERS):
        idx |= s[dx:n0 - 1 + dx, dy:n1 - 1 + dy, dz:n2 - 1 + dz].astype(np.int64) << m
    cells = np.argwhere((idx > 0) & (idx < 255))
    if len(cells) == 0:
        return np.zeros((0, 3), np.int64)
    cases = idx[cells[:, 0], cells[:, 1], cells[:, 2]]
    tri_e = _FLAT_TABLE[cases]                                # [cells, 5, 3] cube-edge ids
    valid = tri_e[..., 0] >= 0
    cell_of = np.repeat(np.arange(len(cells)), 5).reshape(-1, 5)[valid]
    e = tri_e[valid]                                          # [T, 3]
    p = cells[cell_of][:, None, :] + _EDGE_LO[e] + np.asarray(origin, np.int64)      # [T, 3, 3] lower end point, global
    ax = _EDGE_AXIS[e]                                        # [T, 3]
    return ((p[..., 0] * dims[1] + p[..., 1]) * dims[2] + p[..., 2]) * 3 + ax


def _flat_table():
    ntri = np.array([len(t) for t in TRI_TABLE])
    flat = np.full((256, 5, 3), -1, np.int64)
    for c in range(256):
        flat[c, : ntri[c]] = TRI_TABLE[c]
    c0 = CORNERS[EDGE_CORNERS[:, 0]]
    c1 = CORNERS[EDGE_CORNERS[:, 1]]
    return flat, np.minimum(c0, c1), np.argmax(np.abs(c1 - c0), axis=1)


def _mesh_from_keys(v: np.ndarray, key: np.ndarray, level: float):
    n0, n1, n2 = v.shape
    uk, inv = np.unique(key.reshape(-1), return_inverse=True)
    faces = inv.reshape(-1, 3)
    pk, ak = uk // 3, uk % 3
    q = np.stack([pk // (n1 * n2), (pk // n2) % n1, pk % n2], 1)
    q1 = q.copy()
    q1[np.arange(len(q)), ak] += 1
    va = v[q[:, 0], q[:, 1], q[:, 2]].astype(np.float64)
    vb = v[q1[:, 0], q1[:, 1], q1[:, 2]].astype(np.float64)
    t = (level - va) / (vb - va)
    verts = q.astype(np.float64)
    verts[np.arange(len(q)), ak] += t
    return verts, faces


_FLAT_TABLE, _EDGE_LO, _EDGE_AXIS = _flat_table()


def marching_cubes(vol: np.ndarray, level: float = 0.5, set_below: bool = True, equal_is_set: bool = True):
    """The classic algorithm on ``vol`` [n0, n1, n2] with the cube's (x, y, z) = array axes (0, 1, 2), as PyMCubes reads a
    numpy array.  -> (verts [Nv,3] float64 in array-index coordinates (axis0, axis1, axis2), faces [Nf,3] int64); one vertex per
    cut lattice edge (shared between the cubes around it), linear interpolation.  ``set_below``: a corner is "set" when its value
    is below the level (Bourke; False: above); ``equal_is_set``: ... or exactly at it (PyMCubes' `<=`)."""
    v = np.asarray(vol, np.float64)
    key = _cell_triangles(v, level, set_below, (0, 0, 0), v.shape, equal_is_set)
    return _mesh_from_keys(v, key, level)


def marching_cubes_blocks(vol: np.ndarray, level: float = 0.5, set_below: bool = True, block: int = 64, equal_is_set: bool = True):
    """``marching_cubes`` on a volume too large to hold its case array at once (513^3: 1 GiB of int64): the cells in blocks of
    ``block``^3, blocks whose values all lie on one side of the level skipped (they hold no triangle) - the same triangles in
    cell order within a block, blocks in array order; vertices numbered by their global edge key as in ``marching_cubes``."""
    vol = np.asarray(vol)
    n0, n1, n2 = vol.shape
    keys = []
    for a0 in range(0, n0 - 1, block):
        for a1 in range(0, n1 - 1, block):
            for a2 in range(0, n2 - 1, block):
                sub = vol[a0:a0 + block + 1, a1:a1 + block + 1, a2:a2 + block + 1]
                if set_below:
                    bit = (sub <= level) if equal_is_set else (sub < level)
                else:
                    bit = (sub >= level) if equal_is_set else (sub > level)
                if bit.all() or not bit.any():
                    continue
                keys.append(_cell_triangles(sub.astype(np.float64), level, set_below, (a0, a1, a2), vol.shape, equal_is_set))
    key = np.concatenate(keys) if keys else np.zeros((0, 3), np.int64)
    return _mesh_from_keys(vol, key, level)


def case_tris_from_mesher(mesher, set_is_inside: bool):
    """Ask ANY marching-cubes implementation for its triangles of every one of the 256 single-cube configurations, in Bourke's
    numbering.  ``mesher(vol [2,2,2] float32, level)`` -> (verts [Nv,3] in ARRAY-INDEX coordinates (axis0, axis1, axis2), faces);
    the corner values are 0 / 1 around level 0.5, so every vertex is an edge midpoint.  ``set_is_inside``: bit m of the case index
    means "corner m is above the level" (else below).  -> list[256] of [T,3] arrays of cube-edge ids (orientation as returned)."""
    mid = {tuple(((CORNERS[a] + CORNERS[b]) * 0.5).tolist()): e for e, (a, b) in enumerate(EDGE_CORNERS)}
    out = []
    for c in range(256):
        vol = np.zeros((2, 2, 2), np.float32)
        for m, (x, y, z) in enumerate(CORNERS):
            bit = (c >> m) & 1
            vol[x, y, z] = float(bit) if set_is_inside else float(1 - bit)
        verts, faces = mesher(vol, 0.5)
        verts = np.asarray(verts, np.float64).reshape(-1, 3)
        faces = np.asarray(faces, np.int64).reshape(-1, 3)
        e = np.array([mid[tuple((np.round(np.asarray(p) * 2) / 2).tolist())] for p in verts.tolist()], np.int64) if len(verts) else np.zeros(0, np.int64)
        out.append(e[faces] if len(faces) else np.zeros((0, 3), np.int64))
    return out


def compare_tables(a, b):
    """per case: 'same' (same triangle set, as unordered vertex triples), 'triangulation' (same surface loops, other diagonals),
    'topology' (different loops: another decision on an ambiguous face - a different surface)"""
    res = {"same": [], "triangulation": [], "topology": []}
    for c in range(256):
        ta = {frozenset(t) for t in np.asarray(a[c]).reshape(-1, 3).tolist()}
        tb = {frozenset(t) for t in np.asarray(b[c]).reshape(-1, 3).tolist()}
        if ta == tb:
            res["same"].append(c)
        elif loops(a[c]) == loops(b[c]):
            res["triangulation"].append(c)
        else:
            res["topology"].append(c)
    return res
