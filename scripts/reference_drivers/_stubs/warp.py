"""Stand-in for `import warp as wp` in the reference's DRIVER files (material_field.py:5, utils/decode_param.py:2).
The drivers use Warp only to wrap a torch tensor before handing it to the solver (gs_simulation.py:528:
`mpm_model.E = wp.from_torch(t)`); with pixie_amd the tensor is handed over as it is."""


def from_torch(t, dtype=None):
    return t


def init():
    pass
