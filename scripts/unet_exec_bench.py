"""Host cost of one network forward per executor: Python plan walk (one ctypes call per operator), the C handle
(pixie_unet_forward: one call per network), and the C handle replayed as a HIP graph.  Usage: unet_exec_bench.py [D ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixie_amd.synthetic import feature_grid
from pixie_amd.unet import SegmentationUNet
from pixie_amd.unet_plan import synthetic_state_dict


def timed(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n * 1e3      # time until the last launch was queued
    torch.cuda.synchronize()
    return host, (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    for D in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]:
        net = SegmentationUNet(64, 32, 64, 3, (1, 1, 2, 4), (8,), D, 8)
        net.load_numpy_state(synthetic_state_dict(net.cfg, 0))
        net = net.to(dev).eval()
        feat = torch.from_numpy(feature_grid(D, 64, seed=1)).to(dev)
        n = 20 if D <= 64 else 5
        row = {}
        for name, executor, graph in (("python", "python", False), ("c", "c", False), ("c+graph", "c", True), ("python+graph", "python", True)):
            net.executor, net.use_graph = executor, graph
            net._graphs.clear()
            row[name] = timed(lambda: net(feat), n)
        ws = net._handle.workspace_bytes(D, D, D) / 2 ** 20
        print(f"D={D:4d} workspace {ws:8.1f} MiB | " + " | ".join(f"{k}: host {v[0]:7.3f} ms, total {v[1]:7.3f} ms" for k, v in row.items()), flush=True)


if __name__ == "__main__":
    main()
