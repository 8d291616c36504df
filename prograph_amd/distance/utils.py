"""
Input normalisation shared by the distance operators.

Mirrors `clean_input` of the reference (prograph/distance/utils.py:7-39): empty operands are
an error, everything becomes a 2-D torch tensor, and the operand with the shorter second
dimension is right-padded with zeros.
"""
import torch
import torch.nn.functional as F

_EMPTY_MSG = ("An operand with zero rows was given.  Padding it would silently measure the distance of every "
              "sequence to the all-zero sequence; pass an explicit zero tensor of the right shape if that is intended.")


def clean_input(X, Y, verbose=False):
    if X.shape[0] == 0 or Y.shape[0] == 0:
        raise ValueError(_EMPTY_MSG)
    X = torch.atleast_2d(torch.as_tensor(X))
    Y = torch.atleast_2d(torch.as_tensor(Y))
    dx, dy = X.shape[1], Y.shape[1]
    if dx != dy:
        if verbose:
            print("X and Y have different sequence lengths (dimension 1)")
        if dy > dx:
            X = F.pad(X, (0, dy - dx))
        else:
            Y = F.pad(Y, (0, dx - dy))
    return X, Y
