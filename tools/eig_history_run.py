import sys, torch
sys.path.insert(0, "/root/repo")
from aline_amd.tasks import HiddenLocation
dev = torch.device("cuda")
task = HiddenLocation(device=dev)
L, B, T = 1_000_000, 200, 30
theta = torch.rand(L + 1, B, 1, 2, device=dev)
xs, ys = torch.rand(B, T, 2, device=dev), torch.randn(B, T, 1, device=dev)
for _ in range(3):
    task.native_eig_history(theta, xs, ys)
torch.cuda.synchronize()
