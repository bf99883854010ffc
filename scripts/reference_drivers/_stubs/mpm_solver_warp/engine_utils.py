"""`from mpm_solver_warp.engine_utils import *` (utils/decode_param.py:4): h5/ply dump helpers, not on the path."""
