"""Per-parameter gradient errors of the native backward against a gradient fixture of tests/golden (reference autograd;
round-4 fixtures also carry the reference's fp64 run).  python tools/grad_check.py grad_cfg2_d256 [precision] [debug flags ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch  # noqa: E402
from conftest import Fixture  # noqa: E402
from helpers import grad_errors, native_model, to_dev  # noqa: E402


def main():
    name = sys.argv[1]
    prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
    flags = sys.argv[3:]
    from aline_amd import _lib
    from aline_amd.train import train_step
    fx = Fixture(name)
    dims, T = fx.meta["dims"], fx.meta["T"]
    model, _ = native_model(dims, fx.meta["wseed"], prec)
    with _lib.debug(*flags):
        terms, ro = train_step(model, to_dev(fx.batch()), T, optimizer=None, embedding_type=dims["embedding_type"],
                               mask_type=fx.meta["mask_type"], forced_idx=fx.forced_idx("train"), clip_grads=False)
        torch.cuda.synchronize()
    print("path", ro.path, "predict_loss", float(terms["predict_loss"]), float(fx.np("train.predict_loss")),
          "design_loss", float(terms["design_loss"]), float(fx.np("train.design_loss")))
    named = [(k, p.grad) for k, p in model.named_parameters()]
    for prefix in (("train64", "train") if "train64.design_loss" in fx else ("train",)):
        errs = grad_errors(fx, named, prefix)
        print(prefix, "worst:", sorted(errs.items(), key=lambda kv: -kv[1])[:12])
    for k, g in named:
        if "acquisition" in k:
            print(k, "got max", float(g.abs().max()), "ref max", float(fx.np("train.gmax." + k)) if "train.gmax." + k in fx else None)


if __name__ == "__main__":
    main()
