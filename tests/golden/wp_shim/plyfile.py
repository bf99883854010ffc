"""Import stub: engine_utils.py imports PlyData / PlyElement at module level; the solver path never calls them."""
PlyData = PlyElement = None
