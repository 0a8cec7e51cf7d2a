"""CPU checks of the scene-initialisation pieces: the checker's restatement of the reference's nearest-neighbour
routine against a brute-force search and a k-d tree, and the PLY reader/writer.

Parity status: unpinned -- the reference holds no outputs or fixtures for simple-knn and no PLY files, and neither
``simple_knn`` nor ``plyfile`` is importable here.  The nearest-neighbour result is defined by the inputs alone (the
search is exact), which is what the independent searches below verify; the file layout is checked against the
header ``plyfile`` is documented to write for the reference's dtype lists."""
import os

import numpy as np
import pytest
import torch

from neuralgaussiansplatting_amd import ply_io
from neuralgaussiansplatting_amd.gaussian_model import GaussianModel


def knn_points(kind, P, seed=0):
    rng = np.random.RandomState(seed)
    if kind == "uniform":
        return rng.uniform(-1.3, 1.3, (P, 3)).astype(np.float32)
    if kind == "clusters":  # dense blobs far apart: rejection radii differ by orders of magnitude
        centres = rng.uniform(-5, 5, (8, 3))
        sigma = 10.0 ** rng.uniform(-3, 0, 8)
        which = rng.randint(0, 8, P)
        return (centres[which] + rng.randn(P, 3) * sigma[which, None]).astype(np.float32)
    if kind == "plane":  # one coordinate constant and non-zero: degenerate Morton axis
        p = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
        p[:, 2] = 0.75
        return p
    if kind == "duplicates":  # every point three times, plus a few singles
        base = rng.uniform(-1, 1, (P // 3, 3)).astype(np.float32)
        return np.concatenate([base, base, base, rng.uniform(-1, 1, (P - 3 * (P // 3), 3)).astype(np.float32)])
    if kind == "positive_octant":  # bounds are reduced together with the origin
        return rng.uniform(2.0, 3.0, (P, 3)).astype(np.float32)
    raise KeyError(kind)


def brute_force(points):
    """float32 arithmetic in the reference's order: dx*dx + dy*dy + dz*dz, three smallest, (a + b + c) / 3."""
    P = points.shape[0]
    out = np.empty(P, np.float32)
    for i in range(P):
        d = points - points[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        d2[i] = np.float32(np.finfo(np.float32).max)
        b = np.sort(d2)[:3] if P >= 4 else np.sort(np.concatenate([np.delete(d2, i), np.full(3, np.finfo(np.float32).max, np.float32)]))[:3]
        with np.errstate(over="ignore"):
            out[i] = (b[0] + b[1] + b[2]) / np.float32(3.0)
    return out


@pytest.mark.parametrize("kind", ["uniform", "clusters", "plane", "duplicates", "positive_octant"])
def test_knn_oracle_equals_brute_force(kind, oracle):
    pts = knn_points(kind, 1500, seed=3)
    got = oracle.knn_mean_dist2(pts)
    np.testing.assert_array_equal(got, brute_force(pts))


def test_knn_oracle_vs_kdtree(oracle):
    from scipy.spatial import cKDTree
    pts = knn_points("clusters", 40000, seed=5)
    got = oracle.knn_mean_dist2(pts)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    want = (d[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-30)


@pytest.mark.parametrize("P", [1, 2, 3])
def test_knn_oracle_too_few_points_gives_sentinel_mean(P, oracle):
    # the FLT_MAX sentinels of the missing neighbours enter the mean: two or three of them overflow to +inf,
    # one gives FLT_MAX / 3
    got = oracle.knn_mean_dist2(knn_points("uniform", P))
    assert np.all(got >= np.float32(np.finfo(np.float32).max) / np.float32(3.0))
    assert np.all(np.isinf(got)) == (P < 3)


def test_knn_oracle_morton_order_is_a_permutation_sorted_by_code(oracle):
    pts = knn_points("uniform", 5000, seed=1)
    _, order = oracle.knn_mean_dist2(pts, return_order=True)
    assert np.array_equal(np.sort(order), np.arange(5000, dtype=np.uint32))
    lo, hi = np.minimum(pts.min(0), 0), np.maximum(pts.max(0), 0)
    cell = (((pts - lo) / (hi - lo)) * np.float32(1023.0)).astype(np.uint32)

    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        return (x | (x << 2)) & 0x09249249

    code = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
    assert np.all(np.diff(code[order].astype(np.int64)) >= 0)


# --------------------------------------------------------------------------- PLY

def test_ply_roundtrip_mixed_types(tmp_path):
    v = np.zeros(7, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("idx", "i4"), ("w", "f8")])
    rng = np.random.RandomState(0)
    v["x"], v["y"], v["z"] = rng.randn(3, 7)
    v["red"], v["idx"], v["w"] = rng.randint(0, 255, 7), rng.randint(-99, 99, 7), rng.randn(7)
    path = str(tmp_path / "m.ply")
    ply_io.write_ply(path, v)
    back = ply_io.read_ply(path)
    assert back.dtype.names == v.dtype.names
    for n in v.dtype.names:
        np.testing.assert_array_equal(back[n], v[n])
    header = open(path, "rb").read().split(b"end_header\n")[0].decode()
    assert header == ("ply\nformat binary_little_endian 1.0\nelement vertex 7\nproperty float x\nproperty float y\n"
                      "property float z\nproperty uchar red\nproperty int idx\nproperty double w\n")
    assert os.path.getsize(path) == len(header) + len("end_header\n") + 7 * (12 + 1 + 4 + 8)


def test_ply_reads_ascii_and_big_endian_and_sized_type_names(tmp_path):
    a = tmp_path / "a.ply"
    a.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float32 x\nproperty float32 y\n"
                 "property float32 z\nproperty uint8 red\nelement face 1\nproperty list uchar int vertex_indices\n"
                 "end_header\n0.5 1.5 -2 255\n1 2 3 7\n3 0 1 1\n")
    v = ply_io.read_ply(str(a))
    np.testing.assert_array_equal(v["x"], np.float32([0.5, 1]))
    np.testing.assert_array_equal(v["red"], np.uint8([255, 7]))
    b = tmp_path / "b.ply"
    with open(b, "wb") as f:
        f.write(b"ply\nformat binary_big_endian 1.0\nelement vertex 3\nproperty float x\nproperty short s\nend_header\n")
        np.array([(1.25, -3), (2.5, 4), (-7.0, 5)], dtype=[("x", ">f4"), ("s", ">i2")]).tofile(f)
    v = ply_io.read_ply(str(b))
    np.testing.assert_array_equal(v["x"], np.float32([1.25, 2.5, -7.0]))
    np.testing.assert_array_equal(v["s"], np.int16([-3, 4, 5]))


def test_ply_rejects_garbage(tmp_path):
    p = tmp_path / "x.ply"
    p.write_bytes(b"not a ply\n")
    with pytest.raises(ValueError):
        ply_io.read_ply(str(p))
    p.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float x\nend_header\n\0\0\0\0")
    with pytest.raises(ValueError, match="ends after"):
        ply_io.read_ply(str(p))
    with pytest.raises(ValueError):
        ply_io.write_ply(str(p), np.zeros((3, 3), np.float32))


def test_point_cloud_store_and_fetch(tmp_path):
    rng = np.random.RandomState(4)
    xyz, rgb = rng.randn(50, 3), rng.uniform(0, 255, (50, 3))
    path = str(tmp_path / "points3d.ply")
    ply_io.storePly(path, xyz, rgb)
    pc = ply_io.fetchPly(path)
    np.testing.assert_array_equal(pc.points, xyz.astype(np.float32))
    np.testing.assert_array_equal(pc.colors, np.floor(rgb) / 255.0)
    assert not pc.normals.any()
    # a bare x/y/z file: colours and normals are made up, as the reference does
    ply_io.write_ply(path, np.zeros(5, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4")]))
    pc = ply_io.fetchPly(path, rng=np.random.RandomState(0))
    assert pc.colors.shape == (5, 3) and np.allclose(np.linalg.norm(pc.normals, axis=1), 1.0)


@pytest.mark.parametrize("degree", [0, 3])
def test_model_save_and_load_ply(degree, tmp_path):
    P, M = 33, (degree + 1) ** 2
    g = torch.Generator().manual_seed(degree)
    r = lambda *s: torch.randn(*s, generator=g)
    m = GaussianModel.from_activated(r(P, 3), r(P, M, 3), torch.rand(P, 1, generator=g) * 0.8 + 0.1,
                                     torch.rand(P, 3, generator=g) + 0.1, r(P, 4), degree)
    m._features.data.copy_(r(P, 64))
    m._normals.data.copy_(r(P, 3))
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    m.save_ply(path)
    names = ply_io.read_ply(path).dtype.names
    assert names == tuple(m.construct_list_of_attributes())
    assert len(names) == 6 + 3 + 3 * (M - 1) + 64 + 1 + 3 + 4
    # SH columns are channel-major: f_rest_0 is coefficient 1 of the red channel, f_rest_(M-1) coefficient 1 of green
    if M > 1:
        v = ply_io.read_ply(path)
        np.testing.assert_array_equal(v["f_rest_0"], m._features_rest[:, 0, 0].detach().numpy())
        np.testing.assert_array_equal(v["f_rest_%d" % (M - 1)], m._features_rest[:, 0, 1].detach().numpy())
    back = GaussianModel(degree).load_ply(path, device="cpu")
    assert back.active_sh_degree == degree
    for a, b in zip(m.parameters() + [m._normals, m._features], back.parameters() + [back._normals, back._features]):
        assert a.shape == b.shape and torch.equal(a, b)
    with pytest.raises(AssertionError):
        GaussianModel(degree + 1).load_ply(path, device="cpu")


def test_knn_binding_refuses_cpu_tensors():
    from neuralgaussiansplatting_amd.simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="GPU only"):
        distCUDA2(torch.zeros(4, 3))
