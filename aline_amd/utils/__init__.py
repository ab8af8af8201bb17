from .attrdict import AttrDict
from .target_mask import create_target_mask, select_targets_by_mask
from .eval import (compute_ll, get_traces, compute_EIG_from_history, eval_boed, calculate_gmm_variance, save_bounds)

__all__ = ["AttrDict", "create_target_mask", "select_targets_by_mask", "compute_ll", "get_traces",
           "compute_EIG_from_history", "eval_boed", "calculate_gmm_variance", "save_bounds"]
