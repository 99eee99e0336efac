"""Counterpart of the reference's inference CLI (restoration_test.py) on the MI355X path: same flags (:211-226), same
output naming (:140-156), eval mode + no_grad, the per-batch `torch.cuda.empty_cache()` sync dropped.

    python -m vspbfr_amd.restoration_test --batch 8 --ckpt pre-train/restoration_net.pt \\
        --ddpm_ckpt pre-train/code_diffuser.pt --psp_checkpoint_path pre-train/style_encoder_decoder.pt \\
        --lq_data_list dirA,dirB --hq_data_list None,None --data_name_list a,b --eval_dir ./eval_dir

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N -m vspbfr_amd.restoration_test ...`; the file list
of every dataset is split contiguously over the ranks (independent images, no collective on the path); each rank writes
its own PNGs (rank appears in the file name, as in the reference)."""
import argparse
import os

import torch

from .e4e import E4e_embedding
from .imageio import PngWriter, RestoreTestSet, output_name
from .pipeline import RestorationPipeline, load_ddpm, shard_range
from .restorenet import Restoration_net


def get_store_data(lq, hq, names):
    lq, hq, names = (str(v).strip().split(",") for v in (lq, hq, names))
    return [{"lq": a, "hq": b, "name": c} for a, b, c in zip(lq, hq, names)]


def tester_restore_ddpm(args, pipe, lq_root, hq_root, eval_dict, data_name, device, rank=0, world=1):
    data = RestoreTestSet(lq_root, None if hq_root == "None" else hq_root, (args.size, args.size))
    lo, hi = shard_range(len(data), rank, world)
    os.makedirs(eval_dict, exist_ok=True)
    writer = PngWriter()
    print("testing!!! len:%d (rank %d handles %d..%d)" % (len(data), rank, lo, hi))
    with torch.no_grad():
        for start in range(lo, hi, args.batch):
            idx = list(range(start, min(start + args.batch, hi)))
            if args.debug and (start - lo) // args.batch > 10:
                break
            items = [data[i] for i in idx]
            gts = None
            if data.hq is not None:
                gts = torch.stack([it[1] for it in items])
                items = [it[0] for it in items]
            low = torch.stack(items).to(device, non_blocking=True)
            out = pipe(low)
            for kind, t in (("restore", out["restored"]), ("low", low), ("sample", out["style_sample"]), ("gt", gts)):
                if t is not None:
                    writer.submit(t.to(device) if kind == "gt" else t, [output_name(eval_dict, i, rank, data_name, kind) for i in idx])
    writer.drain()
    return eval_dict


def main(argv=None):
    ap = argparse.ArgumentParser(description="Visual Style prompt restoration test (MI355X path)")
    ap.add_argument("--batch", type=int, default=1, help="batch sizes for each gpu")
    ap.add_argument("--size", type=int, default=512, help="image sizes for the models")
    ap.add_argument("--mixing", type=float, default=0.5, help="probability of latent code mixing")
    ap.add_argument("--channel_multiplier", type=int, default=2)
    ap.add_argument("--debug", type=bool, default=False, help="for debugging")
    ap.add_argument("--ckpt", type=str, default=None)
    ap.add_argument("--ddpm_ckpt", type=str, default="pre-train/code_diffuser.pt")
    ap.add_argument("--psp_checkpoint_path", type=str, default="pre-train/style_encoder_decoder.pt")
    ap.add_argument("--eval_dir", type=str, default="./eval_dir")
    ap.add_argument("--lq_data_list", type=str, default="")
    ap.add_argument("--hq_data_list", type=str, default="")
    ap.add_argument("--data_name_list", type=str, default="")
    ap.add_argument("--timesteps", type=int, default=4, help="extension: DDPM steps (the reference hard-codes 4, :35-38)")
    ap.add_argument("--no_sample", action="store_true", help="extension: skip the 1024^2 tail and the *_sample.png output")
    ap.add_argument("--conv_dtype", choices=["f32", "bf16", "bf16x3"], default="f32",
                    help="extension: bf16 = the bf16-kernel configuration (vsp_conv2d_bf16; not the parity path); "
                         "bf16x3 = split-precision operands on the bf16 pipe (fp32-grade)")
    args = ap.parse_args(argv)
    args.latent, args.n_mlp = 512, 8
    from . import hip_ops
    hip_ops.BF16_CONV = {"f32": False, "bf16": True, "bf16x3": "x3"}[args.conv_dtype]

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    g_ema = Restoration_net(args.size, args.latent, args.n_mlp, channel_multiplier=args.channel_multiplier)
    if args.ckpt is not None:
        print("load models:", args.ckpt)
        try:
            g_ema.load_state_dict(torch.load(args.ckpt, map_location="cpu")["g_ema"])
        except RuntimeError as e:  # the reference prints and carries on with the initial weights (:246-250)
            print(str(e))
    g_ema = g_ema.to(device).eval()
    name_ = os.path.basename(str(args.ckpt)).strip().split(".")[0]
    eval_root = os.path.join(args.eval_dir, name_)
    psp = E4e_embedding(args.psp_checkpoint_path, out_size=args.size, size=1024, device=device, use_generator=True)
    store = get_store_data(args.lq_data_list, args.hq_data_list, args.data_name_list)
    for k, d in enumerate(store):
        diffusion = load_ddpm(args.ddpm_ckpt, device=device, timesteps=args.timesteps)
        pipe = RestorationPipeline(g_ema, psp, diffusion, mixing=args.mixing, with_sample=not args.no_sample)
        eval_dict = os.path.join(eval_root, str(len(store) - 1), d["name"])  # the reference's `str(i)` is the last index (:174)
        tester_restore_ddpm(args, pipe, d["lq"], d["hq"], eval_dict, d["name"], device, rank, world)


if __name__ == "__main__":
    main()
