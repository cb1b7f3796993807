"""Minimal triangle-mesh PLY reader (ascii / binary little endian), so Open3D stays optional.
Stands in for o3d.io.read_triangle_mesh at reference s3dis_simulator.py:90."""
import numpy as np

from .synth import TriangleMesh

_T = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
      "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
      "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def read_triangle_mesh(path) -> TriangleMesh:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, elems = None, []
        while True:
            tok = f.readline().decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elems.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                elems[-1][2].append(tok[1:])
            elif tok[0] == "end_header":
                break
        verts, faces = None, None
        for name, count, props in elems:
            if name == "vertex":
                dt = np.dtype([(p[-1], "<" + _T[p[0]]) for p in props])
                if fmt == "ascii":
                    names = [p[-1] for p in props]
                    a = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(names)))
                    verts = np.stack([a[:, names.index(c)] for c in "xyz"], 1)
                else:
                    a = np.fromfile(f, dtype=dt, count=count)
                    verts = np.stack([a["x"], a["y"], a["z"]], 1).astype(np.float64)
            elif name == "face":
                if fmt == "ascii":
                    a = np.loadtxt(f, max_rows=count, ndmin=2).astype(np.int64) if count else np.zeros((0, 4), np.int64)
                    faces = a[:, 1:4]
                else:
                    p = props[0]          # list <count type> <index type> vertex_indices
                    ct, it = np.dtype("<" + _T[p[1]]), np.dtype("<" + _T[p[2]])
                    rec = np.dtype([("n", ct), ("i", it, (3,))])
                    a = np.fromfile(f, dtype=rec, count=count)
                    if count and not (a["n"] == 3).all():
                        raise ValueError("only triangle faces are supported")
                    faces = a["i"]
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError(f"unsupported PLY format {fmt}")
    if verts is None:
        raise ValueError("PLY has no vertex element")
    if faces is None:
        faces = np.zeros((0, 3), np.int32)
    return TriangleMesh(vertices=np.ascontiguousarray(verts, np.float64),
                        triangles=np.ascontiguousarray(faces, np.int32))


def write_triangle_mesh(path, mesh):
    v = np.asarray(mesh.vertices, dtype="<f4")
    t = np.asarray(mesh.triangles, dtype="<i4")
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\n"
                b"property float y\nproperty float z\nelement face %d\n"
                b"property list uchar int vertex_indices\nend_header\n" % (len(v), len(t)))
        v.tofile(f)
        rec = np.empty(len(t), dtype=np.dtype([("n", "u1"), ("i", "<i4", (3,))]))
        rec["n"], rec["i"] = 3, t
        rec.tofile(f)


def read_point_cloud(path) -> np.ndarray:
    """(N,3) float64 vertex coordinates of any PLY this module can parse (point clouds written by
    containers.S3DISSimScene, Open3D-style double/float clouds, meshes).  Stands in for
    o3d.io.read_point_cloud at reference evaluate_single_scene.py:15-23."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, count, props, current, seen_vertex = None, 0, [], None, False
        while True:
            tok = f.readline().decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                current = tok[1]
                if current == "vertex":
                    count, seen_vertex = int(tok[2]), True
                elif not seen_vertex:
                    raise ValueError("the vertex element must come first")
            elif tok[0] == "property" and current == "vertex":
                props.append(tok[1:])
            elif tok[0] == "end_header":
                break
        if not seen_vertex:
            raise ValueError("PLY has no vertex element")
        names = [p[-1] for p in props]
        if fmt == "ascii":
            a = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(names)))
            return np.stack([a[:, names.index(c)] for c in "xyz"], 1).astype(np.float64)
        if fmt != "binary_little_endian":
            raise ValueError(f"unsupported PLY format {fmt}")
        dt = np.dtype([(p[-1], "<" + _T[p[0]]) for p in props])
        a = np.fromfile(f, dtype=dt, count=count)
        return np.stack([a["x"], a["y"], a["z"]], 1).astype(np.float64)
