"""Checkpoint-layout contract (CPU): the vspbfr_amd modules expose exactly the state-dict keys/shapes/dtypes recorded from
the reference's modules (tests/golden/state_specs.json, written by tools/make_golden.py), so the published checkpoints load
with strict=True."""
import json
import os
from argparse import Namespace

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPECS = json.load(open(os.path.join(ROOT, "tests", "golden", "state_specs.json")))


def build(name):
    from vspbfr_amd.diffusion import Code_diffuser
    from vspbfr_amd.e4e import Encoder4Editing, Generator
    from vspbfr_amd.restorenet import Restoration_net
    return {
        "restorenet512": lambda: Restoration_net(512, 512, 8, channel_multiplier=2),
        "restorenet64": lambda: Restoration_net(64, 512, 8),
        "diffuser": lambda: Code_diffuser(timesteps=4),
        "e4e_encoder": lambda: Encoder4Editing(50, "ir_se", Namespace(input_channel=3, stylegan_size=1024)),
        "e4e_decoder1024": lambda: Generator(1024, 512, 8, channel_multiplier=2),
        "e4e_decoder64": lambda: Generator(64, 512, 8, channel_multiplier=2),
    }[name]()


@pytest.mark.parametrize("name", sorted(SPECS))
def test_state_dict_layout(name):
    got = {k: (list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in build(name).state_dict().items()}
    ref = {k: (s, d) for k, s, d in SPECS[name]}
    assert set(got) == set(ref), (sorted(set(ref) - set(got))[:5], sorted(set(got) - set(ref))[:5])
    assert all(got[k] == ref[k] for k in ref)


def test_ddpm_buffers_and_schedule():
    import numpy as np

    from vspbfr_amd.diffusion import Code_diffuser, My_DDPM
    d = My_DDPM(Code_diffuser(4), timesteps=4, linear_start=0.1, linear_end=0.99)
    bufs = [k for k in d.state_dict() if not k.startswith("model.")]
    assert bufs == ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                    "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "diffuser.npz")))
    np.testing.assert_array_equal(d.posterior_mean_coef1.numpy(), g["ddpm_T4/coef1"])  # bit-exact vs the reference's buffers
    np.testing.assert_array_equal(d.posterior_mean_coef2.numpy(), g["ddpm_T4/coef2"])


def test_shard_range_covers_batch():
    from vspbfr_amd.pipeline import shard_range
    for n in (0, 1, 7, 8, 128, 131):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
