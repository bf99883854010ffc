"""Import stub: engine_utils.py imports h5py at module level; the solver path never calls it."""
