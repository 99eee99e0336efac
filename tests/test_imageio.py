"""Loader / writer either side of the path.  CPU parts here; the device quantiser is in the gpu test."""
import numpy as np
import pytest
import torch


def _make_images(tmp_path, sizes):
    from PIL import Image
    rng = np.random.default_rng(0)
    paths = []
    for i, (w, h) in enumerate(sizes):
        p = tmp_path / ("sub" if i % 2 else "") / f"img_{i:02d}.{'png' if i % 2 else 'jpg'}"
        p.parent.mkdir(exist_ok=True)
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(p)
        paths.append(str(p))
    (tmp_path / "notes.txt").write_text("not an image")
    return paths


def test_loader_matches_reference_recipe(tmp_path):
    from PIL import Image
    from vspbfr_amd.imageio import RestoreTestSet, list_images, load_image
    _make_images(tmp_path, [(600, 512), (512, 512), (300, 500), (1024, 700)])
    files = list_images(str(tmp_path))
    assert len(files) == 4 and files == sorted(files)
    ds = RestoreTestSet(str(tmp_path))
    for i, f in enumerate(files):
        t = ds[i]
        assert t.shape == (3, 512, 512) and t.dtype == torch.float32 and -1.0 <= t.min() and t.max() <= 1.0
        # restatement of dataset.py:470-495 + ToTensor/Normalize(0.5, 0.5)
        img = Image.open(f).convert("RGB")
        w, h = img.size
        if (h, w) != (512, 512):
            r = max(512 / h, 512 / w)
            nw, nh = int(r * w), int(r * h)
            img = img.resize((nw, nh), Image.Resampling.LANCZOS)
            hi, wi = max(nh - 512, 0) // 2, max(nw - 512, 0) // 2
            img = img.crop((wi, hi, wi + 512, hi + 512))
        ref = (torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float() / 255 - 0.5) / 0.5
        assert torch.equal(t, ref)
    assert torch.equal(load_image(files[1]), ds[1])


def test_loader_matches_reference_classes(golden):
    """tests/golden/loader.npz = what the reference's ImageFolder_restore_test_no_gt / ImageFolder_restore_test return
    (tools/make_golden.py::gen_loader, transform=None -> PIL image) for the committed folder tests/golden/loader_images;
    ToTensor + Normalize(0.5, 0.5) is the caller's transform (restoration_test.py:89-94): x / 255, then (x - 0.5) / 0.5."""
    import os
    from vspbfr_amd.imageio import RestoreTestSet
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loader_images")
    g = golden("loader")

    def tens(u8):
        return (torch.from_numpy(u8.copy()).permute(2, 0, 1).float() / 255 - 0.5) / 0.5
    ds = RestoreTestSet(os.path.join(root, "lq"), im_size=(64, 64))
    assert [os.path.relpath(f, root) for f in ds.lq] == [str(f) for f in g["no_gt/files"]]   # sorted, recursive, .txt filtered
    for i in range(len(ds)):
        assert torch.equal(ds[i], tens(g[f"no_gt/{i}"])), i
    ds2 = RestoreTestSet(os.path.join(root, "lq"), os.path.join(root, "hq"), im_size=(64, 64))
    assert [os.path.relpath(f, root) for f in ds2.hq] == [str(f) for f in g["gt/hq_files"]]
    assert len(ds2) == 4                                      # the LQ list decides the length (dataset.py:406-407); hq has 4 too
    for i in range(len(ds2)):
        lq, hq = ds2[i]
        assert torch.equal(lq, tens(g[f"gt/{i}/lq"])) and torch.equal(hq, tens(g[f"gt/{i}/hq"])), i


def test_output_names():
    from vspbfr_amd.imageio import output_name
    assert output_name("out/x", 7, 0, "celeba", "restore") == "out/x/000007_0_celeba_restore.png"


@pytest.mark.gpu
def test_device_quantiser_and_png_roundtrip(tmp_path):
    from PIL import Image
    from oracle import models as OM
    from vspbfr_amd import hip_ops as H
    from vspbfr_amd.imageio import PngWriter
    x = torch.randn(3, 3, 37, 53) * 0.9
    x[0, 0, 0, :8] = torch.tensor([-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 3.0, 0.999])
    q = H.quantize_u8_nhwc(x.cuda())
    ref = OM.save_image_quantize(x).permute(0, 2, 3, 1)
    assert torch.equal(q.cpu(), ref)  # bit-exact: integer output
    w = PngWriter(workers=2)
    paths = [str(tmp_path / f"o{i}.png") for i in range(3)]
    w.submit(x.cuda(), paths)
    w.drain()
    for i, p in enumerate(paths):
        assert np.array_equal(np.asarray(Image.open(p)), ref[i].numpy())
