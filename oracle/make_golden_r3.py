"""Round-3 fixtures, produced by the REFERENCE itself (imported from /root/reference as in make_golden.py; run in the build
container only, the fixtures travel):
  aux_timetoken_train  data mode + time token (model/head.py:342-345, train_aline.py:80-82), train-mode rollout WITH the
                       gradients of design_loss + predict_loss wrt every parameter -- pins the time-token backward.
    python oracle/make_golden_r3.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg          # noqa: E402  (sets up the reference imports)


def main():
    d6 = mg.dims_of(1, n_theta=0, embedding_type="data", time_token=True)
    gp6 = dict(dim_x=1, embedding_type="data", n_context_init=2, n_query_init=20, n_target_theta=0, n_target_data=7,
               design_scale=5, noise_scale=0.01)
    mg.gen_model_fixture("aux_timetoken_train", mg.GPTask(**gp6), d6, B=5, T=6, with_grads=True,
                         mask_kwargs=mg.mask_kw("all", "data", 7, 0), seed=321, wseed=9)


if __name__ == "__main__":
    main()
