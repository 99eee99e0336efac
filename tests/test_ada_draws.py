"""Host side of ADA: the seeded transformation draws of vspbfr_amd.non_leaking against the reference's own draws
(tests/golden/ada_draws.npz, tools/make_golden.py --only ada_draws; reference non_leaking.py:660-760).  CPU only."""
import numpy as np
import torch


def test_seeded_draws_match_reference(golden):
    from vspbfr_amd import non_leaking as NL
    g = golden("ada_draws")
    for seed, p_aug, size in g["cases"]:
        seed, size = int(seed), int(size)
        torch.manual_seed(seed)
        G = NL.sample_affine(float(p_aug), 32, size, size)
        C = NL.sample_color(float(p_aug), 32)
        assert np.abs(G.numpy() - g[f"G_{seed}"]).max() < 1e-6
        assert np.abs(C.numpy() - g[f"C_{seed}"]).max() < 1e-6
    # (the quarter-turn categories (0, 3) of non_leaking.py:673 are inside these draws: with {0, 1, 2, 3} the 64 matrices differ)
