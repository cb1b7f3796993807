"""Seeded procedural room meshes standing in for NKSR reconstructions of S3DIS rooms.

No S3DIS data or reconstructed mesh exists in the build container or on the GPU box, so the
benchmark scenes are generated: an inward-facing room box plus furniture
boxes, every face tessellated on a 2 cm grid (the reference reconstructs with voxel_size=0.02,
s3dis_nksr_reconstructor.py:75) into two triangles per cell, vertices jittered by N(0, 2 mm),
triangle rows shuffled (NKSR output is not spatially sorted).  Per-triangle semantic ids follow the
S3DIS class list used by the reference (s3dis_annotation_loader.py:51-65): 0 ceiling, 1 floor,
2 wall, 7 table, 8 chair, 10 bookcase.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class TriangleMesh:
    """Minimal stand-in for open3d.geometry.TriangleMesh (legacy): float64 vertices, int32 triangles."""
    vertices: np.ndarray
    triangles: np.ndarray
    triangle_sem: np.ndarray = None
    triangle_ins: np.ndarray = None

    def __len__(self):
        return len(self.triangles)


def _grid_face(origin, eu, ev, lu, lv, cell, flip, rng=None):
    """Tessellate the rectangle origin + s*eu + t*ev (0<=s<=lu, 0<=t<=lv) into 2 triangles per cell.  With ``rng`` the
    diagonal of every cell is chosen at random (irregular connectivity)."""
    nu, nv = max(1, int(round(lu / cell))), max(1, int(round(lv / cell)))
    s = np.linspace(0.0, lu, nu + 1)
    t = np.linspace(0.0, lv, nv + 1)
    S, T = np.meshgrid(s, t, indexing="ij")
    verts = origin[None, :] + S.reshape(-1, 1) * eu[None, :] + T.reshape(-1, 1) * ev[None, :]
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).reshape(-1)
    b = a + (nv + 1)
    c = b + 1
    d = a + 1
    if rng is None:
        tri = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0)
    else:
        other = rng.random(len(a)) < 0.5                  # the other diagonal: (a,b,d), (b,c,d)
        t1 = np.where(other[:, None], np.stack([a, b, d], 1), np.stack([a, b, c], 1))
        t2 = np.where(other[:, None], np.stack([b, c, d], 1), np.stack([a, c, d], 1))
        tri = np.concatenate([t1, t2], 0)
    if flip:
        tri = tri[:, ::-1]
    return verts, tri


def _box_faces(lo, hi, cell, inward, rng=None):
    """Six tessellated faces of an axis-aligned box; yields (verts, tris, face_id 0..5 = -x,+x,-y,+y,-z,+z)."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    ext = hi - lo
    E = np.eye(3)
    for axis in range(3):
        u, v = (axis + 1) % 3, (axis + 2) % 3
        for side in (0, 1):
            origin = lo.copy()
            if side:
                origin[axis] = hi[axis]
            # outward normal of the (eu, ev) grid is +axis when (u, v, axis) is a right-handed cycle
            flip = (side == 0) != inward
            yield _grid_face(origin, E[u], E[v], ext[u], ext[v], cell, flip, rng) + (2 * axis + side,)


def make_room(size=(5.0, 4.0, 3.0), num_boxes=8, seed=6, cell=0.02, jitter=0.002, corridor=0.6, rough=False):
    """Room of ``size`` with ``num_boxes`` furniture boxes kept clear of the line y = size_y/2 (the
    benchmark trajectory) by ``corridor`` metres on each side.  Deterministic in ``seed``.

    ``rough=True`` makes the unfriendly variant of the same room (what an NKSR reconstruction looks like rather than
    a CAD model): every object tessellated at its own cell size (shell 2.5 cm, furniture 1.5-4 cm: mixed triangle
    sizes), the diagonal of every cell chosen at random, coincident vertices of adjacent faces WELDED (one vertex
    row, so the room shell is a closed manifold: no seams), every vertex moved by up to 0.3 cell in a random 3-D
    direction (irregular shapes and sizes) and the whole surface displaced by a smooth cm-scale noise field (non-planar
    walls); ``jitter`` is ignored."""
    rng = np.random.default_rng(seed)
    frng = rng if rough else None
    vcell = []                                     # rough: cell size of the object each vertex came from
    Lx, Ly, Lz = size
    verts, tris, sem, ins = [], [], [], []
    base = 0

    def add(v, t, s, inst, c=cell):
        nonlocal base
        verts.append(v)
        vcell.append(np.full(len(v), c))
        tris.append(t + base)
        sem.append(np.full(len(t), s, np.uint16))
        ins.append(np.full(len(t), inst, np.uint16))
        base += len(v)

    face_sem = {0: 2, 1: 2, 2: 2, 3: 2, 4: 1, 5: 0}   # walls, floor (-z), ceiling (+z)
    shell_cell = 0.025 if rough else cell
    for v, t, fid in _box_faces((0, 0, 0), (Lx, Ly, Lz), shell_cell, inward=True, rng=frng):
        add(v, t, face_sem[fid], {0: 1, 1: 2, 2: 3, 3: 4, 4: 5, 5: 6}[fid], shell_cell)

    kinds = [(7, (0.8, 1.6), (0.6, 0.9), (0.70, 0.80)),    # table
             (8, (0.4, 0.5), (0.4, 0.5), (0.45, 0.90)),    # chair
             (10, (0.8, 1.2), (0.3, 0.4), (1.6, 2.0))]     # bookcase
    placed = []
    inst = 7
    tries = 0
    while len(placed) < num_boxes and tries < 10000:
        tries += 1
        s, rx, ry, rz = kinds[int(rng.integers(len(kinds)))]
        wx, wy, wz = rng.uniform(*rx), rng.uniform(*ry), rng.uniform(*rz)
        if rng.random() < 0.5:
            wx, wy = wy, wx
        # snap sizes to the grid so that every face tessellates into whole cells
        wx, wy, wz = (max(cell, round(w / cell) * cell) for w in (wx, wy, wz))
        x0 = round(rng.uniform(0.1, Lx - wx - 0.1) / cell) * cell
        side = rng.random() < 0.5
        y_lo, y_hi = (0.1, Ly / 2 - corridor - wy) if side else (Ly / 2 + corridor, Ly - wy - 0.1)
        if y_hi <= y_lo:
            continue
        y0 = round(rng.uniform(y_lo, y_hi) / cell) * cell
        lo, hi = np.array([x0, y0, 0.0]), np.array([x0 + wx, y0 + wy, wz])
        if any(np.all(lo < phi + 0.05) and np.all(plo < hi + 0.05) for plo, phi in placed):
            continue
        placed.append((lo, hi))
        bcell = cell
        if rough:                                  # a cell size of this object's own that divides its extents
            bcell = float(rng.choice([0.015, 0.02, 0.03, 0.04]))
            hi = lo + np.maximum(np.round((hi - lo) / bcell), 1) * bcell
        for v, t, fid in _box_faces(lo, hi, bcell, inward=False, rng=frng):
            if fid == 4:
                continue   # bottom face rests on the floor
            add(v, t, s, inst, bcell)
        inst += 1

    V = np.concatenate(verts, 0)
    F = np.concatenate(tris, 0)
    S = np.concatenate(sem, 0)
    I = np.concatenate(ins, 0)
    if rough:
        # weld: vertices of adjacent faces that coincide become one row (the shell is then a closed manifold)
        key = np.round(V / 1e-6).astype(np.int64)
        _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
        V, C = V[first], np.concatenate(vcell)[first]
        F = inv.reshape(-1)[F]
        # irregular triangles: every vertex moves by up to 0.3 of its object's cell in a random 3-D direction
        V = V + rng.uniform(-0.3, 0.3, V.shape) * C[:, None]
        # non-planar surfaces: a smooth displacement field, six plane waves of 20-70 cm wavelength, ~1 cm amplitude
        for _ in range(6):
            kvec = rng.normal(size=3)
            kvec *= 2 * np.pi / rng.uniform(0.2, 0.7) / np.linalg.norm(kvec)
            amp = rng.normal(size=3)
            amp *= rng.uniform(0.002, 0.006) / np.linalg.norm(amp)
            V = V + np.sin(V @ kvec + rng.uniform(0, 2 * np.pi))[:, None] * amp[None, :]
    elif jitter > 0:
        V = V + rng.normal(0.0, jitter, V.shape)
    perm = rng.permutation(len(F))
    return TriangleMesh(vertices=np.ascontiguousarray(V, dtype=np.float64),
                        triangles=np.ascontiguousarray(F[perm], dtype=np.int32),
                        triangle_sem=S[perm], triangle_ins=I[perm])


# named stand-ins used by bench.py / tests (sizes and seeds from SURVEY.md section 8(d))
SCENES = {
    "synth_A1_office":  dict(size=(8.0, 6.0, 3.0), num_boxes=12, seed=1),
    "synth_A2_office":  dict(size=(6.0, 5.0, 3.0), num_boxes=10, seed=2),
    "synth_A3_office":  dict(size=(4.0, 4.0, 3.0), num_boxes=6, seed=3),
    "synth_A4_office":  dict(size=(7.0, 4.0, 3.0), num_boxes=9, seed=4),
    "synth_A5_office":  dict(size=(6.0, 6.0, 3.0), num_boxes=11, seed=5),
    "synth_A6_office2": dict(size=(5.0, 4.0, 3.0), num_boxes=8, seed=6),
}
# the same rooms, unfriendly: welded, non-planar, irregular and mixed triangle sizes (see make_room(rough=True))
ROUGH_SCENES = {
    "synth_rough_A1": dict(size=(8.0, 6.0, 3.0), num_boxes=12, seed=1, rough=True),
    "synth_rough_A6": dict(size=(5.0, 4.0, 3.0), num_boxes=8, seed=6, rough=True),
}


# a large hall (~3 M triangles, ~0.3 GB of scene data: past the 256 MiB Infinity Cache) for the regime where the scene no
# longer sits in cache; not one of the BASELINE scenes
EXTRA_SCENES = {
    "synth_hall": dict(size=(16.0, 12.0, 3.0), num_boxes=40, seed=11),
}


def _spec(name):
    return ROUGH_SCENES.get(name) or EXTRA_SCENES.get(name) or SCENES[name]


def make_scene(name, cell=0.02):
    return make_room(cell=cell, **_spec(name))


def scene_size(name):
    return _spec(name)["size"]


def unit_cube(lo=-1.0, hi=1.0):
    """12-triangle cube with inward-facing winding (known-answer scene)."""
    v = np.array([[x, y, z] for x in (lo, hi) for y in (lo, hi) for z in (lo, hi)], dtype=np.float64)
    q = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = []
    for a, b, c, d in q:
        f += [(a, b, c), (a, c, d)]
    return TriangleMesh(vertices=v, triangles=np.array(f, dtype=np.int32))


def quad(z=2.0, half=1.0):
    """Two triangles forming the square |x|,|y| <= half in the plane z."""
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], dtype=np.float64)
    return TriangleMesh(vertices=v, triangles=np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32))
