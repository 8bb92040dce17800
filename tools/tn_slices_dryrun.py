"""Dry build of the training plans of every bench configuration ON THE HOST (no launch is made; the HIP-device check of ops._dev is bypassed
for the table-building code only) to list, per grouped weight-gradient launch, the token slices the rule of train_plan.tn_token_slices
chooses next to what the rule before the round-3 fix chose (profiles/r03_tn_slices.txt).

    python tools/tn_slices_dryrun.py          # ~2 minutes, ~25 GB of host memory for the B=256, N=80 plan
"""
import contextlib, io, json, os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import weights as W
from diffuscene_amd import ops, train_plan
from diffuscene_amd._lib import SS_PER_SLOT, SS_PER_TOKEN
from diffuscene_amd.flat import FlatStorage
from diffuscene_amd.networks.denoise_net import Unet1D
from diffuscene_amd.networks.diffusion_ddpm import DiffusionPoint
ops._dev = lambda t, name=None: t
real = train_plan.tn_token_slices
def old_rule(groups, tile_n, target):
    m_max = max(m for m, _, _ in groups)
    lt = sum(((n + tile_n - 1) // tile_n) * ((k + 127) // 128) for m, n, k in groups if 2 * m >= m_max)
    return max(1, min(32, -(-target // max(lt, 1))))
log = []
def spy(groups, tile_n, target, tile_k=128):
    r = real(groups, tile_n, target, tile_k)
    ms = sorted({m for m, _, _ in groups}, reverse=True)
    nt = sum(((n + tile_n - 1) // tile_n) * ((k + tile_k - 1) // tile_k) for m, n, k in groups if 2 * m >= r[1])
    log.append("   launch of %3d groups, token lengths %s: old rule %2d slices -> new %2d (m_ref %d; %d long tiles of %d x %d, target %d)" % (len(groups), ms, old_rule(groups, tile_n, target), r[0], r[1], nt, tile_n, tile_k, target))
    return r
train_plan.tn_token_slices = spy
stats = os.path.join(tempfile.mkdtemp(), "s.txt"); open(stats, "w").write(json.dumps(W.DATASET_STATS))
CASES = [("living80", W.UNCOND_LIVING, 256, 80, 128, 0), ("bedroom21", W.UNCOND_BEDROOM, 256, 21, 128, 0), ("text", W.TEXT_BEDROOM, 128, 12, 128, 32),
         ("complete", W.UNCOND_LIVING, 128, 80, 128, 0), ("arrange", W.REARRANGE_LIVING, 128, 80, 512, 0), ("living80 shard B=32", W.UNCOND_LIVING, 32, 80, 128, 0)]
for name, kw, B, N, ctx, L in CASES:
    del log[:]
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet1D(**kw)
        arrange = kw["channels"] == 5
        cfg = dict(objectness_dim=0, class_dim=kw["class_dim"], angle_dim=2, objfeat_dim=32)
        if arrange: cfg["room_arrange_condition"] = True
        d = DiffusionPoint(net, cfg, time_num=1000, model_mean_type="v", loss_separate=True, loss_iou=not arrange, train_stats_file=stats).diffusion
    flat = FlatStorage(net)
    tb = {n: getattr(d, n).float() for n in d._TABLE_NAMES}
    be = train_plan.HipBackend(torch.device("cpu"))
    plan = train_plan.TrainPlan(net, flat, d, B, N, SS_PER_TOKEN if ctx == 512 else SS_PER_SLOT, ctx, L, 512 if L else 0, be, tables=tb)
    print("%s (B=%d, N=%d): %d fwd + %d bwd launches" % (name, B, N, len(plan.fwd), len(plan.bwd)))
    print("\n".join(log))
    del plan, flat, net
