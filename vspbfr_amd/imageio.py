"""Image I/O either side of the hot path: the test-time loader of the reference
(dataset.py:376-495 + restoration_test.py:89-94) and its PNG writer (restoration_test.py:134-157, torchvision
save_image(normalize=True, range=(-1, 1))).

Decode/resize stay on the host (PIL, exactly the reference's LANCZOS-resize-to-cover + centre crop, so pixels match);
the quantiser to 8 bits + NCHW->NHWC transpose runs on the device (vsp_quantize_u8_nhwc) so a restored batch crosses
PCIe as 0.75 MB/image of uint8 instead of 3 MB of fp32, and PNG encoding happens on a thread pool while the next batch
is already on the GPU."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

IMG_EXT = (".JPG", ".jpg", ".png", ".jpeg")


def list_images(root):
    """Recursive, sorted listing with the reference's extension filter (op/utils_train.py:8-25, dataset.py:454-463)."""
    out = []
    for dp, dn, fn in os.walk(root):
        dn.sort()
        for f in fn:
            if f.endswith(IMG_EXT):
                out.append(os.path.join(dp, f))
    out.sort()
    return out


def _to_tensor(img):
    """ToTensor + Normalize(0.5, 0.5) (restoration_test.py:89-94): uint8 HWC -> float32 CHW in [-1, 1]."""
    a = np.asarray(img, dtype=np.uint8)
    t = torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32).div_(255.0)
    return t.sub_(0.5).div_(0.5)


def _cover_and_crop(imgs, size_of, im_size):
    """LANCZOS resize so that the image `size_of` covers im_size (h, w), then the same centre crop for every image in
    `imgs` (dataset.py:410-429, 470-492: the resize target and the crop window come from ONE image's size)."""
    from PIL import Image
    w, h = size_of.size
    if h == im_size[0] and w == im_size[1]:
        return imgs
    ratio = max(1.0 * im_size[0] / h, 1.0 * im_size[1] / w)
    new_w, new_h = int(ratio * w), int(ratio * h)
    h_idx = (new_h - im_size[0]) // 2 if new_h - im_size[0] > 0 else 0
    w_idx = (new_w - im_size[1]) // 2 if new_w - im_size[1] > 0 else 0
    box = (w_idx, h_idx, int(w_idx + im_size[1]), int(h_idx + im_size[0]))
    return [im.resize((new_w, new_h), Image.Resampling.LANCZOS).crop(box) for im in imgs]


def load_image(path, im_size=(512, 512)):
    """PIL RGB -> LANCZOS resize so the image covers im_size (h, w) -> centre crop -> float32 CHW in [-1, 1]
    (dataset.py:470-495 followed by ToTensor + Normalize(0.5, 0.5), restoration_test.py:89-94)."""
    from PIL import Image
    img = Image.open(path).convert("RGB")
    return _to_tensor(_cover_and_crop([img], img, im_size)[0])


def load_pair(lq_path, hq_path, im_size=(512, 512)):
    """ImageFolder_restore_test.__getitem__ (dataset.py:408-436): the HQ image's size decides the resize and the crop of BOTH
    images (an LQ file of another size is stretched to the HQ's scaled size)."""
    from PIL import Image
    lq, hq = Image.open(lq_path).convert("RGB"), Image.open(hq_path).convert("RGB")
    lq, hq = _cover_and_crop([lq, hq], hq, im_size)
    return _to_tensor(lq), _to_tensor(hq)


class RestoreTestSet:
    """ImageFolder_restore_test / _no_gt: LQ files (and HQ files when a ground-truth root is given, paired by sorted
    order; the HQ image's size decides the resize, dataset.py:415-417)."""

    def __init__(self, lq_root, hq_root=None, im_size=(512, 512)):
        self.lq = list_images(lq_root)
        self.hq = list_images(hq_root) if hq_root not in (None, "None", "") else None
        self.im_size = im_size

    def __len__(self):
        return len(self.lq)

    def __getitem__(self, idx):
        if self.hq is None:
            return load_image(self.lq[idx], self.im_size)
        return load_pair(self.lq[idx], self.hq[idx], self.im_size)


def output_name(eval_dir, index, rank, data_name, kind):
    """`{index:06d}_{rank}_{name}_{kind}.png`, kind in restore / low / sample / gt (restoration_test.py:140-156)."""
    return f"{str(eval_dir)}/{str(index).zfill(6)}_{str(rank)}_{data_name}_{kind}.png"


class PngWriter:
    """Device-side quantisation + asynchronous PNG encoding."""

    def __init__(self, workers=8):
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.pending = []

    def submit(self, batch, paths):
        """batch: (B, 3, H, W) fp32 on the device in [-1, 1] (values outside are clamped like save_image does)."""
        from . import hip_ops as H
        u8 = H.quantize_u8_nhwc(batch.contiguous(), -1.0, 1.0)
        host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(u8, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append(self.pool.submit(self._encode, host, ev, list(paths)))

    @staticmethod
    def _encode(host, ev, paths):
        from PIL import Image
        ev.synchronize()
        arr = host.numpy()
        for i, p in enumerate(paths):
            Image.fromarray(arr[i]).save(p)

    def drain(self):
        for f in self.pending:
            f.result()
        self.pending = []
