"""Permutation / multi-branch combination of per-pair objectives (aps/task/objf.py:238-369).  The
pair losses are [N] vectors: host-side glue over a handful of values."""
from itertools import permutations
from typing import Any, Callable, List, Optional

import torch as th


def multiple_objf(inp: List[Any], ref: List[Any], objf: Callable,
                  weight: Optional[List[float]] = None, transform: Optional[Callable] = None,
                  batchmean: bool = False) -> th.Tensor:
    """weighted sum of the pair losses (objf.py:238-275)"""
    if len(inp) != len(ref):
        raise ValueError(f"Size mismatch between #inp and #ref: {len(inp)} vs {len(ref)}")
    if weight is None:
        weight = [1 / len(inp)] * len(inp)
    if len(weight) != len(inp):
        raise RuntimeError(f"Missing weight ({len(weight)}) for {len(inp)} tasks")
    if transform:
        inp, ref = [transform(i) for i in inp], [transform(r) for r in ref]
    loss = sum(w * objf(o, r) for w, o, r in zip(weight, inp, ref))
    return th.mean(loss) if batchmean else loss


def permu_invarint_objf(inp: List[Any], ref: List[Any], objf: Callable,
                        transform: Optional[Callable] = None, batchmean: bool = False,
                        return_permutation: bool = False):
    """permutation-invariant loss per utterance (objf.py:278-325)"""
    num_spks = len(inp)
    if num_spks != len(ref):
        raise ValueError(f"Size mismatch between #inp and #ref: {num_spks} vs {len(ref)}")
    if transform:
        inp, ref = [transform(i) for i in inp], [transform(r) for r in ref]
    if num_spks == 1:
        return objf(inp[0], ref[0])
    loss_mat = th.stack([sum(objf(inp[s], ref[t]) for s, t in enumerate(p)) / len(p)
                         for p in permutations(range(num_spks))])
    loss, index = th.min(loss_mat, dim=0)
    if batchmean:
        loss = th.mean(loss)
    return (loss, index) if return_permutation else loss


def hybrid_permu_objf(out: List[Any], ref: List[Any], objf: Callable,
                      transform: Optional[Callable] = None, weight: Optional[List[float]] = None,
                      permute: bool = True, permu_num_spks: int = 2) -> th.Tensor:
    """pair-wise, permuted, or permuted + pair-wise residual branches (objf.py:328-369)"""
    num_branch = len(out)
    if num_branch != len(ref):
        raise RuntimeError(f"Got {len(ref)} references but with {num_branch} outputs")
    if not permute:
        return multiple_objf(out, ref, objf, weight=weight, transform=transform)
    loss = permu_invarint_objf(out[:permu_num_spks], ref[:permu_num_spks], objf,
                               transform=transform)
    if num_branch > permu_num_spks:
        num_weight = num_branch - (permu_num_spks - 1)
        if weight is None:
            weight = [1 / num_weight] * num_weight
        other = multiple_objf(out[permu_num_spks:], ref[permu_num_spks:], objf, weight=weight[1:])
        loss = weight[0] * loss + other
    return loss
