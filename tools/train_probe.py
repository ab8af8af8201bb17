import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
from conftest import Fixture
from helpers import native_model, to_dev
from aline_amd.train import train_step
# usage: train_probe.py [t_chunk ...]   (with arguments: timing only, at those chunk sizes)
TCS = [int(v) for v in sys.argv[1:]]
fx = Fixture("cfg2_location_d32")
model, _ = native_model(fx.meta["dims"], fx.meta["wseed"])
if not TCS:
    terms, ro = train_step(model, to_dev(fx.batch()), 30, forced_idx=fx.forced_idx("train"), clip_grads=False)
for k, p in (model.named_parameters() if not TCS else ()):
    ref = fx.t("train.grad." + k); got = p.grad.cpu()
    print(f"{k:55s} max|ref|={float(ref.abs().max()):.3e} relerr={float((got-ref).abs().max())/(float(ref.abs().max())+1e-9):.2e}")
# timing at B=1000
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.tasks import HiddenLocation
m = Aline(Embedder(2,1,32,128,2,"theta"), Encoder(32,128,4,0.0,3), OutputHead(2,1,32,128)).cuda()
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
task = HiddenLocation()
batch = task.sample_batch(1000)
for tc in (TCS or (10, 30)):
    train_step(m, batch, 30, optimizer=opt, t_chunk=tc); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): train_step(m, batch, 30, optimizer=opt, t_chunk=tc)
    torch.cuda.synchronize(); dt = (time.time()-t0)/3
    print(f"t_chunk={tc}: train step {dt*1e3:.1f} ms -> {1000*30*200/dt:.3e} designs/s; ws GB = {torch.cuda.max_memory_allocated()/1e9:.1f}")
