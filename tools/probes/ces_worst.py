"""Debug probe: worst element of the CES step kernel against the fp64 oracle at a given seed (prints its inputs)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import aline_oracle as orc
from aline_amd.tasks import CESTask
task = CESTask(); torch.manual_seed(3)
L, B, T = 100_000, 20, 2
th0 = task.sample_theta(B)
thetas = torch.cat([th0.unsqueeze(0), task.sample_theta((L, B))], 0).contiguous()
x = task.sample_data(B, T); x[..., 3:] = (x[..., :3] + 0.3 * torch.randn(B, T, 3, device="cuda")).clamp(0.5, 99.5)
y = task.forward(x, th0.unsqueeze(1))
worst = (0, None)
for t in range(T):
    got = task.log_likelihood(y[:, t].unsqueeze(0), x[:, t].unsqueeze(0), thetas).squeeze(-1).cpu().double()
    ref = orc.ces_log_likelihood(y[:, t].cpu().double().unsqueeze(0), x[:, t].cpu().double().unsqueeze(0), thetas.cpu().double()).squeeze(-1)
    r32 = orc.ces_log_likelihood(y[:, t].cpu().unsqueeze(0), x[:, t].cpu().unsqueeze(0), thetas.cpu()).squeeze(-1).double()
    ok = torch.isfinite(ref) & torch.isfinite(got)
    err = (got - ref).abs() / (ref.abs() + 1); err[~ok] = 0
    l, b = divmod(int(err.argmax()), B)
    print(f"t={t} max rel err {float(err.max()):.3e} at l={l} b={b}: hip {float(got[l,b]):.6f} fp64 {float(ref[l,b]):.6f} torch-fp32 {float(r32[l,b]):.6f}")
    print("   theta", [float(v) for v in thetas[l, b]], "x", [float(v) for v in x[b, t]], "y", repr(float(y[b, t, 0])))
    e32 = (r32 - ref).abs() / (ref.abs() + 1); e32[~(torch.isfinite(r32) & torch.isfinite(ref))] = 0
    print(f"   torch-fp32 oracle: max rel err {float(e32.max()):.3e};  frac(hip err > 1e-3) = {float((err > 1e-3).float().mean()):.2e}, frac(fp32 err > 1e-3) = {float((e32 > 1e-3).float().mean()):.2e}")
