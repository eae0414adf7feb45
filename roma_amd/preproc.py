"""Device-side pre-processing: the reference's PIL bicubic resize + /255 + ImageNet normalise
(romatch/utils/utils.py:165-261) reproduced BIT-EXACTLY on the GPU.

PIL's 8-bit resampling is fixed-point integer arithmetic (Pillow src/libImaging/Resample.c).  The per-output-sample
tables are computed here in float64 with exactly the operation order of `precompute_coeffs` and
`normalize_coeffs_8bpc`; `roma_resample_u8` applies them (horizontal pass to uint8, then vertical pass) and
`roma_normalize_u8` does the final scaling.  `resample_tables` and the numpy restatement `resample_numpy` are host logic
(tested against PIL itself without a GPU); `preprocess_device` is the product path."""
from __future__ import annotations

import functools
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _bicubic(x: np.ndarray) -> np.ndarray:
    """bicubic_filter of Resample.c (a = -0.5), elementwise in float64."""
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


@functools.lru_cache(maxsize=64)
def resample_tables(in_size: int, out_size: int):
    """(bounds int32 (out,2) = [xmin, count], coef int32 (out, ksize), ksize) for PIL's BICUBIC, box = the whole axis."""
    scale = filterscale = in_size / out_size                      # (in1 - in0) / outSize in double
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale                                   # bicubic support = 2
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)                        # C (int) cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = _bicubic((np.arange(xmax, dtype=np.float64) + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in k:                                               # sequential sum, like the C loop
            ww += float(v)
        if ww != 0.0:
            k = k / ww
        fixed = np.where(k < 0, -0.5 + k * (1 << PRECISION_BITS), 0.5 + k * (1 << PRECISION_BITS))
        coef[xx, :xmax] = np.trunc(fixed).astype(np.int64).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return bounds, coef, ksize


def resample_numpy(img: np.ndarray, size) -> np.ndarray:
    """uint8 (H,W,C) -> uint8 (h,w,C): the two integer passes in numpy (reference for the kernel, checked against PIL)."""
    h, w = size
    H, W, C = img.shape
    cur = img.astype(np.int64)
    if W != w:
        b, k, ks = resample_tables(W, w)
        out = np.empty((H, w, C), dtype=np.int64)
        for xx in range(w):
            x0, n = b[xx]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(cur[:, x0:x0 + n, :], k[xx, :n].astype(np.int64), axes=([1], [0]))
            out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        cur = out
    if H != h:
        b, k, ks = resample_tables(H, h)
        out = np.empty((h, cur.shape[1], C), dtype=np.int64)
        for yy in range(h):
            y0, n = b[yy]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(k[yy, :n].astype(np.int64), cur[y0:y0 + n], axes=([0], [0]))
            out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255)
        cur = out
    return cur.astype(np.uint8)


_TABLE_CACHE = {}


def _device_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    hit = _TABLE_CACHE.get(key)
    if hit is None:
        b, k, ks = resample_tables(in_size, out_size)
        hit = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
        _TABLE_CACHE[key] = hit
    return hit


def resize_device(img_u8: torch.Tensor, size) -> torch.Tensor:
    """uint8 (H,W,3) CUDA tensor -> uint8 (h,w,3), bit-identical to `PIL.Image.resize((w,h), BICUBIC)`."""
    from . import ops
    h, w = size
    H, W, C = img_u8.shape
    cur = img_u8.contiguous()
    if W != w:                                                    # PIL: horizontal pass first
        cur = ops.resample_u8(cur, w, 1, *_device_tables(W, w, cur.device))
    if H != h:
        cur = ops.resample_u8(cur, h, 0, *_device_tables(H, h, cur.device))
    return cur


def preprocess_device(im, size, device) -> torch.Tensor:
    """PIL RGB image (or uint8 (H,W,3) array/tensor) -> normalised (3,h,w) fp32 CUDA tensor; one uint8 upload."""
    from . import ops
    if isinstance(im, torch.Tensor):
        src = im.to(device)
    else:
        src = torch.from_numpy(np.array(im, dtype=np.uint8)).to(device)
    return ops.normalize_u8(resize_device(src, size), IMAGENET_MEAN, IMAGENET_STD)


def decode_jpeg_device(data, device):
    """JPEG bytes (or a path) -> uint8 (H, W, 3) RGB tensor on `device`, bit-identical to `np.array(Image.open(..).convert("RGB"))`
    (matcher.py:606-637, 667-676).  The Huffman decoding runs on the host (roma_jpeg_entropy_decode), de-quantisation + inverse DCT +
    chroma up-sampling + colour conversion on the GPU (roma_jpeg_reconstruct): only the quantised coefficients cross PCIe.
    Sequential and progressive Huffman streams, grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0.  Returns None for what is not covered (CMYK / RGB-stored,
    12-bit, arithmetic coding, lossless): decode those with PIL."""
    import os
    from . import _lib
    from ._lib import check
    if isinstance(data, (str, os.PathLike)):
        with open(data, "rb") as fh:
            data = fh.read()
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    info = np.zeros(8, dtype=np.int32)
    rc = lib.roma_jpeg_info(buf.ctypes.data, len(data), info.ctypes.data)
    if rc == _lib.ROMA_E_UNSUPPORTED or rc == _lib.ROMA_E_ARG:
        return None
    check(rc, "roma_jpeg_info")
    W, H = int(info[0]), int(info[1])
    nblocks = int(info[4]) * int(info[5]) + 2 * int(info[6]) * int(info[7])
    device = torch.device(device)
    # coefficients + tables in ONE pinned staging tensor from torch's caching host allocator (blocks are recycled once the copy that
    # used them has run: no per-call pinning, and the host never waits for the GPU — a hand-rolled grow-only buffer guarded by an event
    # made match(path, path) wait for the previous match before decoding the next image: 53.2 instead of 55+ pairs/s from files)
    need = (nblocks + 3) * 64
    stage = torch.empty((need,), dtype=torch.int16, pin_memory=True)
    coef = stage[:nblocks * 64].view(nblocks, 64)
    qt = stage[nblocks * 64:].view(3, 64)                             # uint16 bit patterns
    rc = lib.roma_jpeg_entropy_decode(buf.ctypes.data, len(data), coef.data_ptr(), qt.data_ptr())
    if rc == _lib.ROMA_E_ARG:                                        # a corrupt stream: let PIL produce its own error / partial image
        return None
    check(rc, "roma_jpeg_entropy_decode")
    both = stage.to(device, non_blocking=True)
    coef_d, qt_d = both[:nblocks * 64], both[nblocks * 64:]
    planes = torch.empty((nblocks * 64,), dtype=torch.uint8, device=device)
    rgb = torch.empty((H, W, 3), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        check(lib.roma_jpeg_reconstruct(coef_d.data_ptr(), qt_d.data_ptr(), planes.data_ptr(), rgb.data_ptr(), info.ctypes.data,
                                        torch.cuda.current_stream().cuda_stream), "roma_jpeg_reconstruct")
    return rgb
