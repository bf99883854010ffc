"""THE import swap of INTEGRATION.md section 1, realised as a module alias so that the reference's driver files run unedited:
`from mpm_solver_warp.mpm_solver_warp import MPM_Simulator_WARP, get_material_name` resolves to pixie_amd."""
from pixie_amd.mpm_solver import *  # noqa: F401,F403
from pixie_amd.mpm_solver import MPM_Simulator_WARP, get_material_id, get_material_name  # noqa: F401
