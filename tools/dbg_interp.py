import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests import helpers as hp
from tests.test_rollout_gpu import make_dphysics, run_hip
from tests.test_parity_net_gpu import probe_body
from monoforce_amd import synthetic as syn
g = hp.load('interp'); dt=torch.float32
res, d_max = float(g['grid_res']), float(g['d_max'])
grid = torch.as_tensor(g['f32/grid']); qx, qy = g['f32/qx'][0], g['f32/qy'][0]; nq=16
pts, masks = syn.robot_points_4()
for ppl in (0,1):
    dp = probe_body(make_dphysics(pts, masks, 0, res, d_max, points_per_lane=ppl), dt); dp.dphys_cfg.robot_mass=1e6
    x0 = torch.zeros(nq,3); x0[:,0]=torch.as_tensor(qx); x0[:,1]=torch.as_tensor(qy)
    xd = torch.zeros(nq,3); xd[:,2]=-1
    st=(x0,xd,torch.eye(3).repeat(nq,1,1),torch.zeros(nq,3))
    outs, sd = run_hip(dp, grid[0:1], torch.zeros(nq,1,2), st, None)
    z=sd[0][:,2].cpu().numpy(); F=outs[4][:,0,0].numpy().astype(np.float64); n=F/np.linalg.norm(F,axis=-1,keepdims=True)
    print(ppl, 'z', z[:4], g['f32/z'][0][:4]); print('F', F[:2]); print('n', n[:2], g['f32/n'][0][:2])
