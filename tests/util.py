"""Shared helpers for the tests: golden loading and state-dict extraction."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def sub(d, prefix, tensor=True):
    """All entries of npz `d` under 'prefix/' as {rest: tensor}."""
    out = {}
    p = prefix + "/"
    for k in d.files:
        if k.startswith(p):
            v = d[k]
            out[k[len(p):]] = torch.tensor(v) if tensor else v
    return out


def opt(d, key):
    return torch.tensor(d[key]) if key in d.files else None


def cfg(d, key):
    return json.loads(str(d[key]))


def split_batches(rows, sizes):
    out, o = [], 0
    for s in sizes:
        out.append(np.asarray(rows[o:o + s]))
        o += s
    return out
