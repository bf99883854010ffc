"""One scene of `material_mode=neural`, device-resident from the voxel feature grid to the simulated particles
(BASELINE configs[2]).

The reference chains three programs through files (pixie/utils.py:736-779, PhysGaussian/gs_simulation.py:483-531,633-634):

    inference_combined.py   feature grid -> SegmentationUNet + RegressionUNet -> sample_0_pred.npy (11, D, D, D)
    map_pred_to_coords.py   pred.npy -> unscale_prediction -> masked voxel list -> PLY
    gs_simulation.py        PLY -> K-NN onto the particles -> MPM_Simulator_WARP set-up -> p2g2p() x substeps per frame

`neural_scene_rollout` is the same sequence with every hand-over a device tensor: the network outputs feed
pixie_field_to_particles (un-scaling, masked lattice, K = 10 nearest voxels: csrc/field_transfer.hip) directly, its
per-particle arrays feed the solver's set_per_particle, and the substeps run as one fused step loop.  Nothing here is new
arithmetic -- each stage is the function its own tests hold to the reference -- so the only thing to check is the plumbing:
tests/test_pipeline_hip.py compares the particle state with the staged route (files and host arrays between the stages, one
p2g2p() call per substep) bit for bit.  There is no CPU path.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import torch

from .material_field import apply_material_field_to_solver
from .mpm_solver import MPM_Simulator_WARP
from .unet import predict_material_field


def neural_scene_rollout(seg_network, cont_network, feat_grid: torch.Tensor, mask: torch.Tensor, particle_x: torch.Tensor,
                         particle_vol: torch.Tensor, *, n_grid: int, grid_lim: float, dt: float, n_substeps: int,
                         params: Dict, min_bounds: Sequence[float], max_bounds: Sequence[float],
                         to_field_frame: Callable[[torch.Tensor], torch.Tensor], configure: Optional[Callable] = None,
                         particle_cov: Optional[torch.Tensor] = None, k: int = 10, nn_distance_threshold: float = 0.1, ranges: Optional[Dict[str, float]] = None,
                         timings: Optional[Dict[str, float]] = None):
    """feat_grid (1, C, D, H, W) float32 and mask (D, H, W) on the networks' device; particle_x (n, 3) in the simulation frame
    (gs_simulation.py leaves the particles in [0, grid_lim]^3), `to_field_frame` maps them into the frame of the field's
    [min_bounds, max_bounds] lattice (transform_to_original_coordinates, PhysGaussian/material_field.py:81-87).
    `params`: the scene's physics dict (set_parameters_dict); `configure(solver)`: boundary conditions / modifiers.
    Returns (solver, combined prediction (11, D, H, W), per-particle confidence).  With `timings` (a dict), HIP-event
    milliseconds of the stages are left in it (synchronises once at the end)."""
    dev = feat_grid.device
    if dev.type != "cuda":
        raise RuntimeError("neural_scene_rollout runs on a HIP device only (no CPU fallback)")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if timings is not None else None
    mark = (lambda i: ev[i].record()) if ev else (lambda i: None)
    mark(0)
    with torch.no_grad():
        combined, _, _, _ = predict_material_field(seg_network, cont_network, feat_grid)     # inference_combined.py:122-126,186-195
    pred = combined[0]
    mark(1)
    dev = str(feat_grid.device)      # the GPU the field lives on (one process per GPU: not necessarily the signature's default "cuda:0")
    solver = MPM_Simulator_WARP(10, device=dev)                                                # gs_simulation.py:483-489
    solver.load_initial_data_from_torch(particle_x, particle_vol, particle_cov, n_grid=n_grid, grid_lim=grid_lim, device=dev)
    solver.set_parameters_dict(params)
    if configure is not None:
        configure(solver)
    mark(2)
    conf = apply_material_field_to_solver(solver, pred, mask, min_bounds, max_bounds, to_field_frame(particle_x.to(dev)),   # material_field.py:303-363
                                          k_smoothing_neighbors=k, nn_distance_threshold=nn_distance_threshold, ranges=ranges)
    mark(3)
    solver.run(dt, n_substeps)                                                                 # gs_simulation.py:633-634
    mark(4)
    if ev:
        torch.cuda.synchronize()
        for name, a, b in (("unet_ms", 0, 1), ("solver_setup_ms", 1, 2), ("field_to_particles_ms", 2, 3), ("rollout_ms", 3, 4), ("total_ms", 0, 4)):
            timings[name] = ev[a].elapsed_time(ev[b])
    return solver, pred, conf


def neural_scene_batch(seg_network, cont_network, scenes, *, n_grid: int, grid_lim: float, dt: float, n_substeps: int,
                       params: Dict, min_bounds: Sequence[float], max_bounds: Sequence[float],
                       to_field_frame: Callable[[torch.Tensor], torch.Tensor], configure: Optional[Callable] = None,
                       k: int = 10, nn_distance_threshold: float = 0.1, ranges: Optional[Dict[str, float]] = None):
    """A batch of scenes through `neural_scene_rollout`'s stages, software-pipelined across two kinds of work that do not compete:
    the rollout of scene i (a 100 k-particle step loop: two latency-bound launches per substep that fill 40 % of the chip) runs on a
    side stream, issued from a host thread of its own, WHILE the networks of scene i + 1 (matrix-core bound) run on the caller's
    stream.  `scenes`: an iterable of (feat_grid, mask, particle_x, particle_vol) device tensors.  Returns the solvers, in order;
    each ends in exactly the state `neural_scene_rollout` leaves it in (tests/test_pipeline_hip.py).  The caller's stream is ordered
    after every rollout when this returns.  (BASELINE configs[3]: a batch of scenes per GPU; measured in bench.py: pipeline leg.)"""
    import threading
    solvers, threads, errors = [], [], []
    side = None
    cur = None

    def roll(solver, stream):
        try:
            torch.cuda.set_device(solver.device)
            with torch.cuda.stream(stream):
                solver.run(dt, n_substeps)
        except Exception as exc:
            errors.append(exc)

    # (an exception inside the loop -- the 10 %-too-far assertion of the field transfer, the device check, `configure` -- must
    # not leave rollout threads running or the caller's stream unordered behind the side streams while their solvers are released)
    try:
        for i, (feat_grid, mask, particle_x, particle_vol) in enumerate(scenes):
            dev = feat_grid.device
            if dev.type != "cuda":
                raise RuntimeError("neural_scene_batch runs on a HIP device only (no CPU fallback)")
            if side is None:
                cur = torch.cuda.current_stream(dev)
                side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
            with torch.no_grad():
                combined, _, _, _ = predict_material_field(seg_network, cont_network, feat_grid)
            solver = MPM_Simulator_WARP(10, device=str(dev))
            solver.load_initial_data_from_torch(particle_x, particle_vol, None, n_grid=n_grid, grid_lim=grid_lim, device=str(dev))
            solver.set_parameters_dict(params)
            if configure is not None:
                configure(solver)
            apply_material_field_to_solver(solver, combined[0], mask, min_bounds, max_bounds, to_field_frame(particle_x.to(dev)),
                                           k_smoothing_neighbors=k, nn_distance_threshold=nn_distance_threshold, ranges=ranges)
            solver.flush()
            st = side[i % 2]
            if len(threads) >= 2:
                threads[-2].join()          # the previous user of this side stream has issued all its launches
            st.wait_stream(cur)             # set-up and field transfer of this scene are in front of its rollout
            t = threading.Thread(target=roll, args=(solver, st))
            t.start()
            threads.append(t)
            solvers.append(solver)
    finally:
        for t in threads:
            t.join()
        if side is not None:
            for st in side:
                cur.wait_stream(st)
    if errors:
        raise errors[0]
    return solvers
