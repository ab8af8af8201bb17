"""Host-side mirror of the reference's utils/target_mask.py (create_target_mask :5-104,
select_targets_by_mask :107-125): tiny boolean logic run once per epoch, kept in Python."""
import random

import torch


def create_target_mask(mask_type, embedding_type, n_target_data, n_target_theta, n_selected_targets,
                       predefined_masks, predefined_mask_weights, mask_index, attend_to):
    n_target = n_target_data + n_target_theta
    mask = torch.zeros(n_target, dtype=torch.bool)
    if mask_type == "all":
        mask.fill_(True)
    elif mask_type == "none":
        pass
    elif mask_type == "partial":
        if embedding_type in ("data", "theta"):
            mask[torch.randperm(n_target)[:n_selected_targets]] = True
    elif mask_type == "predefined":
        if mask_index is not None:
            chosen = predefined_masks[mask_index]
        elif predefined_mask_weights is not None and len(predefined_mask_weights) == len(predefined_masks):
            w = torch.tensor(predefined_mask_weights, dtype=torch.float)
            chosen = predefined_masks[torch.multinomial(w / w.sum(), 1).item()]
        else:
            chosen = random.choice(predefined_masks)
        for i, on in enumerate(chosen):
            if i < n_target and on:
                mask[i] = True
    elif mask_type == "split":
        if embedding_type == "mix":
            to_data = (attend_to == "data") if attend_to is not None else random.choice([True, False])
            if to_data:
                mask[:n_target_data] = True
            else:
                mask[n_target_data:] = True
    return mask


def select_targets_by_mask(target_results, target_mask):
    sel = torch.where(target_mask.to(target_results.device))[0]
    return target_results[:, sel]
