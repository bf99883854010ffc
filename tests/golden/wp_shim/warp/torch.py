"""`import warp.torch` (warp_utils.py:2) -- nothing from it is used on the solver path."""
