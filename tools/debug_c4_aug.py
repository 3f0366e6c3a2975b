"""Debug: the c4_aug step (plan rebuilt inside the captured train step) replayed many times with a changing augmentation; after every
replay the plan workspace is checked for consistency (kept points == CSR total, lists are permutations, sorted inside voxels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_problem
from monoforce_amd.terrain_encoder import LiftSplatShoot
from monoforce_amd.train import EncoderTrainStep, synthetic_encoder_batch
dev = torch.device('cuda', 0)
R = int(os.environ.get('DBG_ROLLOUTS', '1024'))
cfg, dp, pts, masks, z, mu, ctrl = build_problem(R, 500, 4, dev, 1)
gc = dict(xbound=[-6.4, 6.4, 0.05], ybound=[-6.4, 6.4, 0.05], zbound=[-3.2, 3.2, 6.4], dbound=[0.6, 6.4, 0.1])
torch.manual_seed(0)
enc = LiftSplatShoot(gc, dict(final_dim=(256, 512))).to(dev).train()
enc.cache_plan = False
batch = synthetic_encoder_batch(enc, dp, n_rollouts=R, device=dev)
step = EncoderTrainStep(enc, dp, lr=1e-4, graph=os.environ.get('DBG_GRAPH', '1') == '1')
ga = torch.Generator().manual_seed(1234)
pool = []
post_rots, post_trans = batch[0][4], batch[0][5]
for _ in range(8):
    sc = 0.9 + 0.2 * torch.rand(post_rots.shape[:2], generator=ga)
    pr = torch.eye(3).repeat(*post_rots.shape[:2], 1, 1); pr[..., 0, 0] = sc; pr[..., 1, 1] = sc
    pt = torch.zeros(post_trans.shape); pt[..., :2] = (torch.rand(*post_trans.shape[:2], 2, generator=ga) - 0.5) * 30.0
    pool.append((pr.to(dev), pt.to(dev)))
P, V = 4 * 59 * 16 * 32, 256 * 256
a256 = lambda n: (n * 4 + 255) // 256 * 256


def check(i):
    torch.cuda.synchronize()
    w = enc._plan_ws_slots[0]['ws'].cpu().numpy()
    keys = w[0:4 * P].view(np.int32)
    count = w[a256(P):a256(P) + 4 * V].view(np.int32)
    off = w[a256(P) + 2 * a256(V):a256(P) + 2 * a256(V) + 4 * (V + 1)].view(np.int32)
    o_list = a256(P) + 2 * a256(V) + a256(V + 1)
    lst = w[o_list:o_list + 4 * P].view(np.int32)
    kept = int((keys >= 0).sum())
    ok_keys = bool((keys < V).all())
    hist = np.bincount(keys[keys >= 0], minlength=V)
    ok_count = bool((count == hist).all())
    ok_off = bool((np.diff(off.astype(np.int64)) == hist).all()) and off[0] == 0 and off[-1] == kept
    l = lst[:kept]
    ok_perm = ok_off and bool((np.sort(l) == np.flatnonzero(keys >= 0)).all())
    ok_sorted = ok_perm and bool((keys[l] == np.repeat(np.arange(V), hist)).all())
    print(i, 'kept', kept, 'keys', ok_keys, 'count', ok_count, 'offsets', ok_off, 'perm', ok_perm, 'voxel order', ok_sorted, flush=True)


for i in range(int(os.environ.get('DBG_STEPS', '40'))):
    pr, pt = pool[i % 8]
    batch[0][4].copy_(pr); batch[0][5].copy_(pt)
    eager = (i % 7 == 3)
    loss = step.step(batch, eager=eager)[0]
    if os.environ.get('DBG_CHECK', '1') == '1':
        check(i)
print('done', float(loss), 'graph' if step.graph else 'eager')
