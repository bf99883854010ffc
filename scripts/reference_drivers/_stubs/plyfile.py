"""The slice of `plyfile` the reference's drivers use (gs_simulation.py:123-200 `load_point_cloud`, material_field.py
debug dumps), on top of pixie_amd.ply_io: PlyData.read(path)['vertex'][name], .properties[i].name, `name in element`,
PlyElement.describe(array, 'vertex'), PlyData([element], text=...).write(path)."""
import numpy as np

from pixie_amd.ply_io import read_ply, write_ply


class _Prop:
    def __init__(self, name):
        self.name = name


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data
        self.properties = [_Prop(n) for n in data.dtype.names]

    @staticmethod
    def describe(data, name):
        return PlyElement(name, np.asarray(data))

    def __getitem__(self, key):
        return self.data[key]

    def __contains__(self, key):
        return key in self.data.dtype.names

    def __len__(self):
        return len(self.data)


class PlyData:
    def __init__(self, elements, text=False):
        self.elements, self.text = list(elements), text

    @staticmethod
    def read(path):
        _, allel = read_ply(path)
        return PlyData([PlyElement(k, v) for k, v in allel.items()])

    def __getitem__(self, name):
        return next(e for e in self.elements if e.name == name)

    def write(self, path):
        write_ply(path, self.elements[0].data, text=self.text, element=self.elements[0].name)
