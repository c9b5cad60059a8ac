import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from crop2seg_amd import engine as E, _lib as L
dev = torch.device("cuda")
C = 64
p = {"n.weight": torch.ones(C, device=dev), "n.bias": torch.zeros(C, device=dev)}
ctx = E.Ctx(p, {}, None, E.Workspace(dev), True, None)
x = torch.randn(128, C, 128, 128, device=dev)
for i in range(3):
    y = E.norm_act(ctx, x, "n", L.NORM_GROUP, 4, True, None, None, 0.0)
    torch.cuda.synchronize()
x.normal_()          # fresh write by another kernel
y = E.norm_act(ctx, x, "n", L.NORM_GROUP, 4, True, None, None, 0.0)
torch.cuda.synchronize()
