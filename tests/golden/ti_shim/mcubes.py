"""PyMCubes is not installed in this image (and not vendored by the reference: `import mcubes`, filling.py:5).  The module
exists so that filling.py imports; the one call the reference makes raises, so nothing unpinned can enter a fixture."""


def smooth(*args, **kwargs):
    raise NotImplementedError("mcubes.smooth: PyMCubes is absent; fill_particles(smooth=True) cannot be run from the reference's code here")
