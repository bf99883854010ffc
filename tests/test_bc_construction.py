"""Boundary conditions built from the transferred field (reference material_field.py:364-550).  CPU: the
oracle's restatement of the density clustering must label exactly like sklearn.cluster.DBSCAN (the reference's call).
GPU: the product's device clustering (csrc/field_transfer.hip) must give the oracle's labels, and the two builders must
register the cuboids the reference's arithmetic gives."""
import numpy as np
import pytest

from oracle.field_oracle import dbscan_labels as oracle_dbscan
from pixie_amd.material_field import STATIONARY_ID, fix_to_ground, handle_stationary_clusters

sklearn_cluster = pytest.importorskip("sklearn.cluster")


class _Recorder:
    def __init__(self):
        self.calls = []

    def set_velocity_on_cuboid(self, **kw):
        self.calls.append(kw)


def _blobs(seed, n_blobs=5, per=300, noise=200, spread=0.03):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(0.2, 1.8, size=(n_blobs, 3))
    pts = [c + spread * rng.standard_normal((per + 40 * i, 3)) for i, c in enumerate(centres)]
    pts.append(rng.uniform(0.0, 2.0, size=(noise, 3)))
    pts = np.concatenate(pts).astype(np.float32)
    return pts[rng.permutation(len(pts))]


@pytest.mark.parametrize("seed,eps,min_samples", [(0, 0.03, 10), (1, 0.05, 5), (2, 0.02, 20), (3, 0.08, 3), (4, 0.03, 1)])
def test_oracle_dbscan_labels_equal_sklearn(seed, eps, min_samples):
    pts = _blobs(seed)
    want = sklearn_cluster.DBSCAN(eps=eps, min_samples=min_samples).fit_predict(pts)
    assert np.array_equal(oracle_dbscan(pts, eps, min_samples), want)


def test_oracle_dbscan_edge_cases():
    assert oracle_dbscan(np.zeros((0, 3)), 0.03, 10).shape == (0,)
    assert np.array_equal(oracle_dbscan(np.random.default_rng(0).uniform(0, 100, (50, 3)), 0.01, 10), np.full(50, -1))
    chain = np.stack([np.arange(30) * 0.01, np.zeros(30), np.zeros(30)], axis=1)   # one chain: ends are border points
    want = sklearn_cluster.DBSCAN(eps=0.0151, min_samples=3).fit_predict(chain)
    assert np.array_equal(oracle_dbscan(chain, 0.0151, 3), want)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,eps,min_samples", [(0, 0.03, 10), (1, 0.05, 5), (2, 0.02, 20), (3, 0.08, 3), (4, 0.03, 1)])
def test_device_dbscan_equals_oracle_and_sklearn(hip_device, seed, eps, min_samples):
    import torch
    from pixie_amd.material_field import dbscan_labels
    pts = _blobs(seed)
    want = oracle_dbscan(pts, eps, min_samples)
    assert np.array_equal(want, sklearn_cluster.DBSCAN(eps=eps, min_samples=min_samples).fit_predict(pts))
    got = dbscan_labels(torch.from_numpy(pts).to(hip_device), eps, min_samples).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_device_dbscan_edge_cases_and_a_dense_cluster(hip_device):
    import torch
    from pixie_amd.material_field import dbscan_labels
    assert dbscan_labels(torch.zeros((0, 3), device=hip_device), 0.03, 10).shape == (0,)
    far = np.random.default_rng(0).uniform(0, 100, (50, 3)).astype(np.float32)       # extent >> 256 eps: the lattice is capped
    assert np.array_equal(dbscan_labels(far, 0.01, 10).cpu().numpy(), np.full(50, -1))
    chain = np.stack([np.arange(30) * 0.01, np.zeros(30), np.zeros(30)], axis=1).astype(np.float32)
    assert np.array_equal(dbscan_labels(chain, 0.0151, 3).cpu().numpy(), oracle_dbscan(chain, 0.0151, 3))
    # a stationary part as the pipeline sees it: 60 000 particles, ~500 neighbours within eps each, plus stray ones
    rng = np.random.default_rng(5)
    dense = np.concatenate([rng.uniform(0.8, 1.2, (40000, 3)), rng.uniform(0.2, 0.5, (15000, 3)) * [1, 1, 0.3], rng.uniform(0, 2, (5000, 3))]).astype(np.float32)
    got = dbscan_labels(dense, 0.03, 10).cpu().numpy()
    assert np.array_equal(got, oracle_dbscan(dense, 0.03, 10))


@pytest.mark.gpu
def test_stationary_cluster_cuboids(hip_device):
    pts = _blobs(7, n_blobs=3, per=400, noise=50)
    mat = np.zeros(len(pts), dtype=np.int32)
    mat[::2] = STATIONARY_ID
    stat = pts[mat == STATIONARY_ID]
    labels = sklearn_cluster.DBSCAN(eps=0.03, min_samples=10).fit_predict(stat)
    ids, counts = np.unique(labels[labels >= 0], return_counts=True)
    rec = _Recorder()
    out = handle_stationary_clusters(rec, pts, mat, buffer=0.01)
    big = ids[np.argmax(counts)]
    c = stat[labels == big]
    assert len(out) == 1 and len(rec.calls) == 1 and out[0]["cluster_size"] == counts.max() and out[0]["cluster_id"] == big
    np.testing.assert_allclose(rec.calls[0]["point"], 0.5 * (c.min(0) + c.max(0)), rtol=0, atol=0)
    np.testing.assert_allclose(rec.calls[0]["size"], 0.5 * (c.max(0) - c.min(0)) + 0.01, rtol=0, atol=0)
    assert rec.calls[0]["reset"] == 1 and rec.calls[0]["velocity"] == [0.0, 0.0, 0.0]
    rec = _Recorder()
    out = handle_stationary_clusters(rec, pts, mat, only_handle_largest_cluster=False)
    assert len(out) == len(ids) == len(rec.calls)
    assert handle_stationary_clusters(_Recorder(), pts, np.zeros(len(pts), dtype=np.int32)) == []


def test_fix_to_ground_matches_reference_arithmetic():   # (host arithmetic on six extrema: runs without a device)
    import torch
    pos = np.random.default_rng(3).uniform(0.4, 1.6, size=(1000, 3)).astype(np.float32)
    for arg in (pos, torch.from_numpy(pos)):
        for pct in (1, 5):
            rec = _Recorder()
            out = fix_to_ground(rec, arg, delta_z=0.05, buffer_xy=0.5, min_z_percentile=pct)
            mn, mx = pos[:, :2].min(0), pos[:, :2].max(0)
            z = np.percentile(pos[:, 2], pct) if pct > 1 else pos[:, 2].min()
            np.testing.assert_allclose(rec.calls[0]["point"], [(mn[0] + mx[0]) / 2, (mn[1] + mx[1]) / 2, z + 0.025], rtol=1e-6)
            np.testing.assert_allclose(rec.calls[0]["size"], [(mx[0] - mn[0]) / 2 + 0.5, (mx[1] - mn[1]) / 2 + 0.5, 0.025], rtol=1e-6)
            assert out[0]["type"] == "ground" and rec.calls[0]["reset"] == 1
