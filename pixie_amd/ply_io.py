"""Minimal PLY reader / writer for the point files that sit between the two halves of the pipeline.

The reference writes them with `plyfile` (`PlyData([PlyElement.describe(vertex_data, 'vertex')], text=False).write(path)`,
pixie/voxel/map_pred_to_coords.py:262-263; text=True at PhysGaussian/gs_simulation.py:103) and reads them back with
`PlyData.read` (gs_simulation.py:123, :248).  plyfile is a third-party package that is not part of this tree; what is
needed of it is small: one element of scalar properties described by a numpy structured dtype.  Files written here have
the header plyfile writes for such an element (`format binary_little_endian 1.0` / `format ascii 1.0`, one
`property <type> <name>` line per field, plyfile's type names), so either side can read the other's files.
Host-side file I/O only: no compute lives here.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

# numpy kind+size -> PLY scalar type name, as plyfile spells them (its _data_type_reverse table)
_NP_TO_PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}
_PLY_TO_NP = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
              "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def write_ply(path: str, vertex: np.ndarray, text: bool = False, element: str = "vertex") -> None:
    """Write one element of scalar properties.  `vertex` is a 1-D structured array, e.g. the 13-field vertex_data of
    map_pred_to_ply (x, y, z f4; red, green, blue, alpha u1; part_label i4; density, E, nu f4; material_id i4; conf f4)."""
    vertex = np.asarray(vertex)
    if vertex.dtype.names is None or vertex.ndim != 1:
        raise ValueError("write_ply expects a 1-D numpy structured array")
    lines = ["ply", "format ascii 1.0" if text else "format binary_little_endian 1.0", f"element {element} {len(vertex)}"]
    le_fields = []
    for name in vertex.dtype.names:
        dt = vertex.dtype.fields[name][0]
        if dt.shape != () or dt.kind not in "iuf":
            raise ValueError(f"property {name!r}: only scalar integer / float properties are supported")
        key = dt.kind + str(dt.itemsize)
        if key not in _NP_TO_PLY:
            raise ValueError(f"property {name!r}: unsupported dtype {dt}")
        lines.append(f"property {_NP_TO_PLY[key]} {name}")
        le_fields.append((name, "<" + key))
    lines.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("ascii"))
        if text:
            # plyfile prints with numpy.savetxt: '%d' for integer kinds, '%.18g' for floats, one space between fields
            fmt = " ".join("%d" if vertex.dtype.fields[n][0].kind in "iu" else "%.18g" for n in vertex.dtype.names)
            for row in vertex:
                f.write((fmt % tuple(row.tolist()) + "\n").encode("ascii"))
        else:
            packed = np.empty(len(vertex), dtype=np.dtype(le_fields))   # packed, little-endian, declaration order
            for name in vertex.dtype.names:
                packed[name] = vertex[name]
            f.write(packed.tobytes())


def read_ply(path: str) -> Tuple[np.ndarray, Dict[str, np.ndarray]]:
    """Read a PLY file; returns (the first element as a structured array, {element name: array}) for every element whose
    properties are all scalars (list properties -- faces -- are not needed on this path and raise)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt = None
        elements = []   # (name, count, [(prop name, numpy code)])
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                elements[-1][2].append((tok[2], _PLY_TO_NP[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unknown PLY format {fmt!r}")
        out: Dict[str, np.ndarray] = {}
        for name, count, props in elements:
            native = np.dtype([(p, "=" + c) for p, c in props])
            if fmt == "ascii":
                arr = np.empty(count, dtype=native)
                for i in range(count):
                    vals = f.readline().split()
                    arr[i] = tuple((int(v) if native[j].kind in "iu" else float(v)) for j, v in enumerate(vals[:len(props)]))
            else:
                order = "<" if fmt == "binary_little_endian" else ">"
                disk = np.dtype([(p, order + c) for p, c in props])
                raw = f.read(disk.itemsize * count)
                if len(raw) != disk.itemsize * count:
                    raise ValueError(f"{path}: truncated element {name!r}")
                arr = np.frombuffer(raw, dtype=disk, count=count).astype(native)
            out[name] = arr
    first = elements[0][0] if elements else None
    return (out[first] if first else np.empty(0)), out
