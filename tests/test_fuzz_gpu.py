"""A slice of the randomised parity sweeps inside the driver-run suite (`-m gpu`).

tools/fuzz_parity.py (shapes, head dims, dtypes, layouts — BNHD views, padded row pitches, head slices —, scales, causal; forward + backward
against float64 attention on the device, held to the contract fa2_fwd_plan names per head range) found the round-2 row-pitch bug of the
hand-scheduled kernels once it drew grids wide enough for them; tools/fuzz_mask.py does the same for the bias / mask path.  One fixed seed of
each runs here, plus forced draws of the two shape classes the random draw reaches rarely: grids of more 256-row workgroups than CUs (the
persistent hand-scheduled kernels, the split of a partly filled last round) and long causal sequences (pair units through the item seam).
The committed long sweeps live under profiles/*fuzz*.json."""
import importlib.util
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location("_" + name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_randomised_parity_sweep_slice():
    fz = _tool("fuzz_parity")
    from rocwmma_fattn import _fa2_lib
    rng = random.Random(2024)
    gen = torch.Generator(device="cuda").manual_seed(2024)
    bad, n_asm, n_split, n_padded_wide, n_tail = [], 0, 0, 0, 0
    plan_of = []
    for i in range(110):
        d = fz.one_case(i, rng, gen, want_bwd=(i % 3 == 0))
        plan_of.append(d)
    for j in range(16):                                      # wide grids: every second one with the backward
        plan_of.append(fz.one_case(1000 + j, rng, gen, want_bwd=(j % 2 == 0), force="wide"))
    for j in range(3):
        plan_of.append(fz.one_case(2000 + j, rng, gen, want_bwd=(j == 0), force="longcausal"))
    for d in plan_of:
        if d["fails"]:
            bad.append(d)
        kernel, contract, heads_main, kernel_tail, contract_tail, nsplit = d["plan"]
        n_asm += kernel == _fa2_lib.FA2_KERNEL_ASM
        n_split += nsplit > 1
        n_tail += kernel_tail != 0
        n_padded_wide += d["i"] >= 1000 and "rowpad" in d["layouts"]
    assert not bad, bad[:3]
    # the slice must reach what it is here for: the hand-scheduled kernels, the split, a padded row pitch on a wide grid
    assert n_asm >= 5 and n_split >= 1 and n_padded_wide >= 1, (n_asm, n_split, n_padded_wide, n_tail)


def test_randomised_mask_sweep_slice():
    fm = _tool("fuzz_mask")
    rng = random.Random(77)
    gen = torch.Generator(device="cuda").manual_seed(77)
    bad = []
    for i in range(90):
        d = fm.one_case(i, rng, gen, want_bwd=(i % 2 == 0))
        if d["fails"]:
            bad.append(d)
    assert not bad, bad[:3]
