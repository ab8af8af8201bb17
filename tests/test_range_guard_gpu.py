"""GPU: the f16 range guard of the F16X3 (3-term split-f16) mode (include/aline_hip.h: aline_f16_range_status).  Operands of
that mode must stay below 65504; a model / input that leaves the range must not produce inf / NaN silently: the kernels raise
a sticky device flag, `Rollout.run_checked()` / `Aline.forward` re-run in exact fp32 and warn, the training step raises."""
import warnings

import pytest
import torch

import aline_oracle as orc
from helpers import maxdiff, native_model, to_dev

pytestmark = pytest.mark.gpu

D32 = {"dim_x": 2, "dim_y": 1, "d": 32, "F": 128, "n_head": 4, "L": 3, "C": 10, "n_theta": 2, "embedding_type": "theta",
       "time_token": False}
D256 = dict(D32, d=256, F=256, n_head=8, L=2)
CFG = dict(embedding_type="theta", num_components=10, std_min=1e-4, n_target_theta=2)


def _batch(B=6, nq=40, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(context_x=torch.rand(B, 1, 2, generator=g), context_y=torch.randn(B, 1, 1, generator=g),
                query_x=torch.rand(B, nq, 2, generator=g), query_y=torch.randn(B, nq, 1, generator=g),
                target_all=torch.rand(B, 2, 1, generator=g))


def _scaled(model, key, factor):
    with torch.no_grad():
        dict(model.named_parameters())[key].mul_(factor)
    return orc.cast_state_dict(model.state_dict())


@pytest.mark.parametrize("dims,path,key,factor,bit", [
    (D32, "s3::step_kernel", "embedder.x_embedder.2.weight", 3e5, 1),            # layer-0 input rows beyond 65504
    (D32, "s3::step_kernel", "encoder.encoder.layers.1.linear1.weight", 400.0, 2),   # |w| * 2^8 beyond f16
    (D32, "s3::step_kernel", "encoder.encoder.layers.0.self_attn.in_proj_weight", 250.0, 1),   # q / k / v operands overflow (weights fit)
    (D256, "x3::layer_kernel", "embedder.x_embedder.2.weight", 3e5, 1),
    (D256, "x3::layer_kernel", "encoder.encoder.layers.0.linear2.weight", 4000.0, 2),
    (dict(D32, n_head=8), "generic pipeline", "embedder.y_embedder.2.weight", 3e5, 1),      # the F16X3 GEMM policy
])
def test_out_of_range_operands_raise_the_flag_and_fall_back_to_f32(dims, path, key, factor, bit):
    from aline_amd.rollout import Rollout
    model, _ = native_model(dims, 7, "f16x3")
    T = 5
    batch = _batch()
    cfg = dict(CFG, n_head=dims["n_head"], num_layers=dims["L"])
    ok = Rollout(model, to_dev(batch), T, select="argmax").run()
    assert ok.path == path and ok.range_status() == 0                       # the unscaled model is clean
    forced = ok.idx.clone()
    sd = _scaled(model, key, factor)
    ro = Rollout(model, to_dev(batch), T, select="forced", forced_idx=forced)
    ro.run()
    st = ro.range_status()
    assert st & bit, (st, bit)
    with pytest.raises(RuntimeError, match="f16 range"):
        ro.check_range()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ro.run_checked()
    assert ro.fell_back and any("f16's range" in str(x.message) for x in w)
    torch.cuda.synchronize()
    # the f32 re-run is the reference's arithmetic on the same (extreme) weights: finite, and equal to the oracle's
    ref = orc.rollout(sd, batch, cfg, T, forced_idx=forced.cpu())
    rll = torch.stack(ref["target_ll"])
    assert torch.isfinite(ro.target_ll).all() == torch.isfinite(rll).all()
    fin = torch.isfinite(rll)
    assert float(((ro.target_ll.cpu() - rll)[fin].abs() / (1 + rll[fin].abs())).max()) < 1e-3


def test_step_api_reruns_in_f32_with_a_warning():
    """The drop-in `model.forward(batch)` in f16x3: the status word is read back after the step (one 4-byte copy) and an
    overflowing step is repeated in f32 -- the caller never sees inf / NaN the fp32 reference would not produce."""
    model, _ = native_model(D32, 7, "f16x3")
    model.eval()
    batch = to_dev(_batch())
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        a = model(batch)
    assert not w
    _scaled(model, "embedder.x_embedder.2.weight", 3e5)
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = model(batch)
    assert any("f16's range" in str(x.message) for x in w)
    model.set_precision("f32")
    with torch.no_grad():
        ref = model(batch)
    assert torch.isfinite(out.design_out.zt).all()
    assert maxdiff(out.design_out.zt, ref.design_out.zt.cpu()) == 0.0
    assert maxdiff(out.posterior_out.mixture_means, ref.posterior_out.mixture_means.cpu()) == 0.0
    assert (a.design_out.idx.shape == out.design_out.idx.shape)


def test_training_step_stops_on_overflow():
    from aline_amd import train as tr
    from aline_amd.tasks import HiddenLocation
    model, _ = native_model(D32, 7, "f16x3")
    batch = HiddenLocation(n_query_init=40).sample_batch(8)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    tr.train_step(model, batch, 4, optimizer=opt)
    tr.check_range_async(block=True)                                       # clean
    _scaled(model, "embedder.x_embedder.2.weight", 3e5)
    tr.train_step(model, batch, 4, optimizer=None)
    with pytest.raises(RuntimeError, match="f16's range"):
        tr.check_range_async(block=True)


def test_driver_train_stops_before_the_update_and_the_checkpoint(tmp_path):
    """`driver.train` must stop in the epoch whose rollout overflowed -- before optimizer.step() turns every weight
    into NaN (inf-norm clip of a NaN gradient) and before a checkpoint of them is written."""
    import os
    import random
    from aline_amd.driver import train
    from aline_amd.tasks import HiddenLocation
    from test_driver import _Cfg, _cfg
    torch.manual_seed(0); random.seed(0)
    model, _ = native_model(D32, 7, "f16x3")
    task = HiddenLocation(n_query_init=40, device=torch.device("cuda"))
    cfg = _cfg(tmp_path, max_epoch=3, burning_epoch=0, checkpoint=1, T=4, min_T=4, batch_size=8,
               task=_Cfg(mask_type=["all"], embedding_type="theta", n_target_data=0, n_target_theta=2, n_query_init=40))
    _scaled(model, "embedder.x_embedder.2.weight", 3e5)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with pytest.raises(RuntimeError, match="f16's range"):
        train(cfg, model, task)
    assert all(torch.equal(v, before[k]) for k, v in model.state_dict().items())      # no update was applied
    assert not os.path.exists(tmp_path / "ckpt_1.tar")                                   # and nothing was saved
