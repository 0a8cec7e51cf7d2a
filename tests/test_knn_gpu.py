"""GPU parity of the nearest-neighbour routine (through the C ABI) against the checker's restatement of the
reference's simple-knn.  Bar: bit-exact -- both evaluate dx*dx + dy*dy + dz*dz without contraction, the search is
exact on both sides, and the mean is formed in the same order."""
import os

import numpy as np
import pytest
import torch

from test_knn_cpu import knn_points

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def dist2():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from neuralgaussiansplatting_amd.simple_knn._C import _lib, distCUDA2
    _lib()
    return distCUDA2


@pytest.mark.parametrize("kind,P", [("uniform", 20000), ("clusters", 100000), ("plane", 30000), ("duplicates", 10000),
                                    ("positive_octant", 5000)])
def test_knn_matches_oracle(kind, P, dist2, oracle):
    pts = knn_points(kind, P, seed=11)
    got = dist2(torch.tensor(pts, device=DEV)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.knn_mean_dist2(pts))


@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 63, 64, 65, 127, 4095, 4096, 4097, 4161])
def test_knn_ragged_sizes(P, dist2, oracle):
    pts = knn_points("uniform", P, seed=P)
    got = dist2(torch.tensor(pts, device=DEV)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.knn_mean_dist2(pts))


# NGS_RANDOM_CONFIGS=N widens the sweep for a soak run
@pytest.mark.parametrize("seed", range(int(os.environ.get("NGS_RANDOM_CONFIGS", "6"))))
def test_knn_random_sweep(seed, dist2, oracle):
    rng = np.random.RandomState(5000 + seed)
    kind = ("uniform", "clusters", "plane", "duplicates", "positive_octant")[rng.randint(0, 5)]
    P = int(rng.choice([rng.randint(1, 300), rng.randint(300, 9000), rng.randint(9000, 60000)]))
    pts = knn_points(kind, P, seed=seed)
    if rng.rand() < 0.3:  # anisotropic extent: one axis dominates the Morton code
        pts = pts * np.array([1.0, 10.0 ** rng.uniform(-3, 0), 10.0 ** rng.uniform(-3, 0)], np.float32)
    got = dist2(torch.tensor(pts, device=DEV)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.knn_mean_dist2(pts))


def test_knn_empty_and_bad_shapes(dist2):
    assert dist2(torch.zeros((0, 3), device=DEV)).shape == (0,)
    with pytest.raises(RuntimeError, match="num_points, 3"):
        dist2(torch.zeros((5, 4), device=DEV))
    with pytest.raises(RuntimeError, match="float32"):
        dist2(torch.zeros((5, 3), device=DEV, dtype=torch.float64))


def test_knn_all_points_identical(dist2, oracle):
    pts = np.full((1000, 3), 0.25, np.float32)
    got = dist2(torch.tensor(pts, device=DEV)).cpu().numpy()
    assert not got.any()
    np.testing.assert_array_equal(got, oracle.knn_mean_dist2(pts))


def test_knn_full_size_scene(dist2, oracle):
    """3 M points, the benchmark scene's positions: against the checker on every point, and against a k-d tree on a
    sample (independent of both implementations)."""
    import os
    from scipy.spatial import cKDTree
    pts = knn_points("uniform", 3_000_000, seed=1234)
    got = dist2(torch.tensor(pts, device=DEV)).cpu().numpy()
    oracle.set_threads(min(16, os.cpu_count() or 1))  # per-point results do not depend on the thread count
    try:
        want = oracle.knn_mean_dist2(pts)
    finally:
        oracle.set_threads(1)
    np.testing.assert_array_equal(got, want)
    sample = np.random.RandomState(0).choice(pts.shape[0], 20000, replace=False)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts[sample].astype(np.float64), k=4)
    np.testing.assert_allclose(got[sample], (d[:, 1:] ** 2).mean(axis=1), rtol=2e-6)


def test_create_from_pcd(dist2):
    from neuralgaussiansplatting_amd.gaussian_model import GaussianModel
    from neuralgaussiansplatting_amd.ply_io import BasicPointCloud
    from neuralgaussiansplatting_amd.sh_utils import SH2RGB
    rng = np.random.RandomState(2)
    pcd = BasicPointCloud(points=rng.uniform(-1.3, 1.3, (5000, 3)), colors=rng.rand(5000, 3), normals=None)
    m = GaussianModel(3).create_from_pcd(pcd, 1.0, device=DEV)
    want = torch.log(torch.sqrt(torch.clamp_min(dist2(m.get_xyz.detach()), 1e-7)))
    assert torch.equal(m._scaling[:, 0], want) and torch.equal(m._scaling[:, 1], m._scaling[:, 2])
    assert m.get_features.shape == (5000, 16, 3) and not m._features_rest.any()
    np.testing.assert_allclose(SH2RGB(m._features_dc[:, 0, :]).detach().cpu().numpy(), pcd.colors, atol=1e-6)
    assert torch.allclose(m.get_opacity, torch.full((5000, 1), 0.1, device=DEV))
    assert torch.equal(m.get_rotation, torch.tensor([[1.0, 0, 0, 0]], device=DEV).expand(5000, 4))


def test_knn_nonfinite_points_do_not_stall_the_search(dist2, oracle):
    """A NaN or infinite coordinate has no defined neighbours; such points must not make their wave walk every leaf
    (thousands of them would take minutes), and the finite points' results must not change."""
    pts = knn_points("uniform", 200000, seed=4)
    bad = np.random.RandomState(1).choice(pts.shape[0], 3000, replace=False)
    dirty = pts.copy()
    dirty[bad[:1500], 0] = np.nan
    dirty[bad[1500:], 2] = np.inf
    got = dist2(torch.tensor(dirty, device=DEV)).cpu().numpy()
    keep = np.ones(pts.shape[0], bool)
    keep[bad] = False
    want = oracle.knn_mean_dist2(pts[keep])
    np.testing.assert_array_equal(got[keep], want)
