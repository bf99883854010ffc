"""oracle/field_oracle.py against the golden vectors produced by the reference's own code
(tests/golden/make_field_golden.py executes unscale_prediction / MaterialProperties cut out of the reference files)."""
import os

import numpy as np
import pytest

from oracle import field_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "field_transfer.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_unscale_and_point_cloud_match_reference(gold):
    un = field_oracle.unscale_prediction(gold["pred"])
    assert np.array_equal(un, gold["unscaled"])
    cloud = field_oracle.voxel_point_cloud(un, gold["mask"], gold["min_bounds"], gold["max_bounds"])
    for key in ("pos", "density", "E", "nu", "material_id", "part_labels", "conf"):
        assert cloud[key].dtype == gold["cloud_" + key].dtype
        assert np.array_equal(cloud[key], gold["cloud_" + key]), key


@pytest.mark.parametrize("weighted", [False, True])
def test_knn_assignment_matches_reference(gold, weighted):
    out = field_oracle.field_to_particles(gold["pred"], gold["mask"], gold["min_bounds"], gold["max_bounds"], gold["particle_pos"],
                                          k=10, nn_distance_threshold=0.1, weighted=weighted)
    tag = "w_" if weighted else "u_"
    assert out["too_far"].sum() == gold[tag + "too_far"].sum() > 0
    for key in ("part_labels", "density", "E", "nu", "material_id", "conf", "nearest_dist", "too_far"):
        assert np.array_equal(out[key], gold[tag + key]), key
    far = out["too_far"]
    assert (out["material_id"][far] == field_oracle.STATIONARY_ID).all() and (out["part_labels"][far] == 0).all()
