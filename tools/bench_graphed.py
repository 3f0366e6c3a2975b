"""Perception -> planning loop (encoder + 64-sample shooting) eager vs one hipGraph launch (monoforce_amd/graphed.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_problem
from monoforce_amd import synthetic as syn
from monoforce_amd.graphed import GraphedTerrainPlanner
from monoforce_amd.terrain_encoder import LiftSplatShoot
dev = torch.device('cuda')
for n_trajs in (64, 4096):
    cfg, dp, pts, masks, z, mu, ctrl = build_problem(n_trajs, 500, 4, dev, 1)
    torch.manual_seed(0)
    gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
    enc = LiftSplatShoot(gc, dict(final_dim=(256, 512))).to(dev)
    gp = GraphedTerrainPlanner(enc, dp, syn.lss_camera_rig(1), (4, 3, 256, 512), n_trajs=n_trajs, cost='force')
    imgs = torch.randn(4, 3, 256, 512, device=dev)
    def timeit(fn, n=20):
        for _ in range(3): fn(imgs)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): int(fn(imgs)['best'])          # the node reads the winner every frame
        return (time.perf_counter() - t) / n * 1e3
    print(f'n_trajs={n_trajs}: eager {timeit(gp.eager):.2f} ms/frame, graph {timeit(gp):.2f} ms/frame', flush=True)
