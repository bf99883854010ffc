"""pixie_amd/ply_io.py: the PLY point files between the two programs (CPU only; plyfile is not installed, so the header
is checked against the PLY specification and the layout plyfile writes for `PlyElement.describe(structured_array)`)."""
import numpy as np
import pytest

from pixie_amd.ply_io import read_ply, write_ply

DTYPE = [("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1"), ("alpha", "u1"),
         ("part_label", "i4"), ("density", "f4"), ("E", "f4"), ("nu", "f4"), ("material_id", "i4"), ("conf", "f4")]


def vertices(n, seed=0):
    rng = np.random.default_rng(seed)
    v = np.zeros(n, dtype=DTYPE)
    for name in ("x", "y", "z", "density", "E", "nu", "conf"):
        v[name] = rng.normal(size=n).astype(np.float32) * (1e6 if name == "E" else 1.0)
    for name in ("red", "green", "blue", "alpha"):
        v[name] = rng.integers(0, 256, n)
    v["part_label"] = rng.integers(-3, 9, n); v["material_id"] = rng.integers(0, 8, n)
    return v


@pytest.mark.parametrize("text", [False, True])
@pytest.mark.parametrize("n", [0, 1, 257])
def test_round_trip(tmp_path, text, n):
    v = vertices(n)
    path = str(tmp_path / "p.ply")
    write_ply(path, v, text=text)
    got, elements = read_ply(path)
    assert list(elements) == ["vertex"] and got.dtype.names == v.dtype.names and len(got) == n
    for name in v.dtype.names:
        assert np.array_equal(got[name], v[name]), name      # '%.18g' round-trips float32 exactly in the ascii form


def test_header_and_body_layout(tmp_path):
    v = vertices(5)
    path = str(tmp_path / "p.ply")
    write_ply(path, v)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert head.decode().splitlines() == ["ply", "format binary_little_endian 1.0", "element vertex 5", "property float x", "property float y",
                                          "property float z", "property uchar red", "property uchar green", "property uchar blue",
                                          "property uchar alpha", "property int part_label", "property float density", "property float E",
                                          "property float nu", "property int material_id", "property float conf"]
    assert len(body) == 5 * (3 * 4 + 4 + 4 + 3 * 4 + 4 + 4)      # packed records, no padding
    rec = np.frombuffer(body, dtype=np.dtype([(n, "<" + t) for n, t in DTYPE]))
    assert np.array_equal(rec["E"], v["E"]) and np.array_equal(rec["alpha"], v["alpha"])


def test_reads_foreign_headers(tmp_path):
    """comments, obj_info, sized type names and big-endian bodies (what other writers produce)"""
    path = str(tmp_path / "f.ply")
    body = np.array([(1.5, 2, 7), (-2.5, 3, 9)], dtype=[("x", ">f8"), ("k", ">i2"), ("c", ">u1")])
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_big_endian 1.0\ncomment made by hand\nobj_info x\nelement vertex 2\nproperty float64 x\n"
                b"property int16 k\nproperty uint8 c\nend_header\n" + body.tobytes())
    got, _ = read_ply(path)
    assert got["x"].tolist() == [1.5, -2.5] and got["k"].tolist() == [2, 3] and got["c"].tolist() == [7, 9]
    with open(path, "wb") as f:
        f.write(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty list uchar int idx\nend_header\n3 0 1 2\n")
    with pytest.raises(ValueError):
        read_ply(path)
    with pytest.raises(ValueError):
        write_ply(path, np.zeros(3))
