"""PLY files as the reference reads and writes them, without the ``plyfile`` package.

The reference stores two kinds of PLY, both with a single ``vertex`` element of scalar properties written by
``plyfile`` in its default binary little-endian form:

* point clouds (scene/dataset_readers.py:108-147): ``x y z nx ny nz`` as float plus ``red green blue`` as uchar;
* trained models (scene/gaussian_model.py:289-329): all-float rows ``x y z nx ny nz f_dc_* f_rest_* features_*
  opacity scale_* rot_*``.

``write_ply`` produces that header layout (``property float x`` ...); ``read_ply`` accepts binary little/big endian and
ASCII files, the classic (``float``) and the sized (``float32``) type names, and skips any element after ``vertex``.
"""
import collections
import os

import numpy as np

_PLY_TO_NP = {
    "char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
    "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4",
    "float64": "f8",
}
_NP_TO_PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
              "f8": "double"}

BasicPointCloud = collections.namedtuple("BasicPointCloud", ["points", "colors", "normals"])  # utils/graphics_utils.py:17-20


def write_ply(path, vertices):
    """Write a numpy structured array as the ``vertex`` element of a binary little-endian PLY file."""
    vertices = np.asarray(vertices)
    if vertices.dtype.names is None or vertices.ndim != 1:
        raise ValueError("write_ply expects a one-dimensional structured array")
    lines = ["ply", "format binary_little_endian 1.0", "element vertex %d" % vertices.shape[0]]
    fields = []
    for name in vertices.dtype.names:
        kind = vertices.dtype[name].str.lstrip("<>=|")
        if kind not in _NP_TO_PLY:
            raise ValueError("property %r has type %s, which PLY cannot store" % (name, vertices.dtype[name]))
        lines.append("property %s %s" % (_NP_TO_PLY[kind], name))
        fields.append((name, "<" + kind))
    lines.append("end_header")
    packed = np.empty(vertices.shape[0], dtype=np.dtype(fields))  # packed, little endian, no padding
    for name in vertices.dtype.names:
        packed[name] = vertices[name]
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("ascii"))
        packed.tofile(f)


def read_ply(path):
    """-> structured array holding the ``vertex`` element (which must come first, as in every file the reference
    handles).  Raises ValueError on anything that is not a PLY file with scalar vertex properties."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s is not a PLY file" % path)
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: header ends before end_header" % path)
            words = line.decode("ascii", "replace").split()
            if not words or words[0] in ("comment", "obj_info"):
                continue
            if words[0] == "format":
                fmt = words[1]
            elif words[0] == "element":
                elements.append((words[1], int(words[2]), []))
            elif words[0] == "property":
                if not elements:
                    raise ValueError("%s: property before any element" % path)
                if words[1] == "list":
                    elements[-1][2].append((words[-1], None))
                else:
                    if words[1] not in _PLY_TO_NP:
                        raise ValueError("%s: unknown property type %r" % (path, words[1]))
                    elements[-1][2].append((words[2], _PLY_TO_NP[words[1]]))
            elif words[0] == "end_header":
                break
        if not elements or elements[0][0] != "vertex":
            raise ValueError("%s: the first element must be 'vertex'" % path)
        _, count, props = elements[0]
        if any(kind is None for _, kind in props):
            raise ValueError("%s: list properties in the vertex element are not supported" % path)
        if fmt in ("binary_little_endian", "binary_big_endian"):
            order = "<" if fmt == "binary_little_endian" else ">"
            dtype = np.dtype([(name, order + kind) for name, kind in props])
            data = np.fromfile(f, dtype=dtype, count=count)
            if data.shape[0] != count:
                raise ValueError("%s: file ends after %d of %d vertices" % (path, data.shape[0], count))
            return data.astype(np.dtype([(name, "<" + kind) for name, kind in props]), copy=False)
        if fmt == "ascii":
            dtype = np.dtype([(name, "<" + kind) for name, kind in props])
            data = np.empty(count, dtype=dtype)
            for i in range(count):
                words = f.readline().split()
                if len(words) < len(props):
                    raise ValueError("%s: vertex %d has %d of %d values" % (path, i, len(words), len(props)))
                data[i] = tuple(float(w) if kind[0] == "f" else int(w) for w, (_, kind) in zip(words, props))
            return data
        raise ValueError("%s: unsupported PLY format %r" % (path, fmt))


def fetchPly(path, rng=None):
    """Point cloud file -> BasicPointCloud (scene/dataset_readers.py:108-130): colours scaled to [0,1]; missing colours
    are replaced by uniform random ones and missing normals by random unit vectors, as the reference does."""
    v = read_ply(path)
    rng = np.random if rng is None else rng
    positions = np.vstack([v["x"], v["y"], v["z"]]).T
    n = positions.shape[0]
    names = v.dtype.names
    if all(c in names for c in ("red", "green", "blue")):
        colors = np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0
    else:
        colors = rng.rand(n, 3)
    if all(c in names for c in ("nx", "ny", "nz")):
        normals = np.vstack([v["nx"], v["ny"], v["nz"]]).T
    else:
        r = rng.randn(n, 3)
        normals = r / np.linalg.norm(r, axis=1, keepdims=True)
    return BasicPointCloud(points=positions, colors=colors, normals=normals)


def storePly(path, xyz, rgb):
    """Positions [N,3] and colours 0..255 [N,3] -> point cloud file with zero normals (scene/dataset_readers.py:132-147)."""
    xyz = np.asarray(xyz)
    rgb = np.asarray(rgb)
    v = np.zeros(xyz.shape[0], dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"),
                                      ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    v["x"], v["y"], v["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    v["red"], v["green"], v["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    write_ply(path, v)


def mkdir_p(folder):  # utils/system_utils.py:16-26
    if folder:
        os.makedirs(folder, exist_ok=True)
