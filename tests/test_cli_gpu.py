"""End-to-end run of the CLI counterpart of restoration_test.py on the GPU box: checkpoint files in the reference's layout
(g_ema / att_mapper / pSp dict) -> PNGs with the reference's names."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cli_end_to_end(tmp_path):
    from PIL import Image
    from vspbfr_amd import restoration_test as cli
    from vspbfr_amd.diffusion import Code_diffuser
    from vspbfr_amd.e4e import Encoder4Editing, Generator
    from vspbfr_amd.restorenet import Restoration_net
    torch.manual_seed(0)
    ck = tmp_path / "ckpt"
    ck.mkdir()
    torch.save({"g_ema": Restoration_net(512, 512, 8).state_dict()}, ck / "restoration_net.pt")
    torch.save({"att_mapper": Code_diffuser(timesteps=4).state_dict()}, ck / "code_diffuser.pt")
    enc = Encoder4Editing(50, "ir_se", Namespace(input_channel=3, stylegan_size=1024))
    dec = Generator(1024, 512, 8)
    sd = {"encoder." + k: v for k, v in enc.state_dict().items()}
    sd.update({"decoder." + k: v for k, v in dec.state_dict().items()})
    torch.save({"state_dict": sd, "latent_avg": torch.zeros(18, 512),
                "opts": {"encoder_type": "Encoder4Editing", "stylegan_size": 1024, "start_from_latent_avg": True}},
               ck / "style_encoder_decoder.pt")
    lq = tmp_path / "lq"
    lq.mkdir()
    rng = np.random.default_rng(1)
    for i, (w, h) in enumerate([(512, 512), (640, 600), (300, 400)]):
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(lq / f"face_{i}.png")
    out = tmp_path / "eval"
    cli.main(["--batch", "2", "--ckpt", str(ck / "restoration_net.pt"), "--ddpm_ckpt", str(ck / "code_diffuser.pt"),
              "--psp_checkpoint_path", str(ck / "style_encoder_decoder.pt"), "--eval_dir", str(out),
              "--lq_data_list", str(lq), "--hq_data_list", "None", "--data_name_list", "demo"])
    d = out / "restoration_net" / "0" / "demo"
    names = sorted(os.listdir(d))
    assert names == sorted(f"{i:06d}_0_demo_{k}.png" for i in range(3) for k in ("restore", "low", "sample"))
    for n in names:
        im = Image.open(d / n)
        assert im.size == (512, 512) and im.mode == "RGB"
    # the *_low.png is the loader's output re-quantised: identical to the centre-cropped LANCZOS input
    low0 = np.asarray(Image.open(d / "000000_0_demo_low.png"))
    assert np.array_equal(low0, np.asarray(Image.open(lq / "face_0.png").convert("RGB")))
