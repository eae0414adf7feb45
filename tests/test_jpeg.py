"""JPEG decoding for match() on file paths (matcher.py:606-637, 667-676: `Image.open(path).convert("RGB")`).
Host part (roma_jpeg_info / roma_jpeg_entropy_decode, no GPU) + the numpy restatement of libjpeg's reconstruction
(oracle/jpeg_oracle.py) against PIL itself: bit-identical — that pins the restatement; the GPU test then holds roma_jpeg_reconstruct
to the same images."""
import glob
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image

from roma_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "assets", "*.jpg")))


def _host_decode(data):
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    info = np.zeros(8, np.int32)
    rc = lib.roma_jpeg_info(buf.ctypes.data, len(data), info.ctypes.data)
    if rc != 0:
        return rc, None, None, None
    nb = int(info[4]) * int(info[5]) + 2 * int(info[6]) * int(info[7])
    coef = np.zeros((nb, 64), np.int16)
    qt = np.zeros((3, 64), np.uint16)
    rc = lib.roma_jpeg_entropy_decode(buf.ctypes.data, len(data), coef.ctypes.data, qt.ctypes.data)
    return rc, info, coef, qt


def _synthetic_jpegs():
    """Streams the bundled photographs do not cover: 4:4:4, 4:2:2, grey, odd sizes, restart intervals, optimised Huffman tables, and
    PROGRESSIVE streams (spectral selection + successive approximation; a bundled photograph re-encoded that way)."""
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:97, 0:131]
    base = np.stack([127 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 127 + 90 * np.cos(xx / 5.0 + yy / 11.0), 40 + 1.5 * xx], axis=2)
    img = np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)
    out = {}
    for name, im, kw in [("444", Image.fromarray(img), dict(subsampling=0, quality=90)),
                         ("420_odd", Image.fromarray(img[:95, :129]), dict(subsampling=2, quality=75)),
                         ("grey", Image.fromarray(img[..., 0]), dict(quality=85)),
                         ("420_opt", Image.fromarray(img), dict(subsampling=2, quality=60, optimize=True)),
                         ("420_rst", Image.fromarray(img), dict(subsampling=2, quality=80, restart_marker_blocks=3)),
                         ("tiny", Image.fromarray(img[:5, :3]), dict(subsampling=2, quality=95)),
                         ("422", Image.fromarray(img), dict(subsampling=1, quality=85)),
                         ("422_odd", Image.fromarray(img[:33, :61]), dict(subsampling=1, quality=70)),
                         ("422_tiny", Image.fromarray(img[:9, :4]), dict(subsampling=1, quality=90)),
                         ("prog_420", Image.fromarray(img), dict(progressive=True, quality=80)),
                         ("prog_444", Image.fromarray(img), dict(progressive=True, subsampling=0, quality=92)),
                         ("prog_422_odd", Image.fromarray(img[:95, :129]), dict(progressive=True, subsampling=1, quality=60)),
                         ("prog_grey", Image.fromarray(img[..., 0]), dict(progressive=True, quality=85)),
                         ("prog_rst", Image.fromarray(img), dict(progressive=True, quality=75, restart_marker_blocks=4)),
                         ("prog_photo", Image.open(ASSETS[1]).convert("RGB"), dict(progressive=True, quality=88, optimize=True))]:
        bio = io.BytesIO()
        try:
            im.save(bio, "JPEG", **kw)
        except TypeError:                                       # an older Pillow without restart_marker_blocks
            continue
        out[name] = bio.getvalue()
    return out


def test_host_entropy_decoder_and_restatement_are_bit_identical_to_pil():
    from oracle import jpeg_oracle
    streams = {os.path.basename(f): open(f, "rb").read() for f in ASSETS}
    streams.update(_synthetic_jpegs())
    assert len(streams) >= 16
    for name, data in streams.items():
        rc, info, coef, qt = _host_decode(data)
        assert rc == 0, (name, rc, _lib.load().roma_last_error())
        ref = np.array(Image.open(io.BytesIO(data)).convert("RGB"))
        mine = jpeg_oracle.reconstruct(coef, qt, info)
        assert mine.shape == ref.shape and np.array_equal(mine, ref), name


def test_streams_outside_the_supported_subset_are_refused_not_misdecoded():
    img = Image.fromarray((np.arange(64 * 64 * 3) % 251).astype(np.uint8).reshape(64, 64, 3))
    bio = io.BytesIO()
    img.convert("CMYK").save(bio, "JPEG")                       # four components
    rc, *_ = _host_decode(bio.getvalue())
    assert rc == _lib.ROMA_E_UNSUPPORTED
    bio = io.BytesIO()
    try:
        img.save(bio, "JPEG", keep_rgb=True)                    # three components stored as RGB (Adobe marker, no YCbCr transform)
        rc, *_ = _host_decode(bio.getvalue())
        assert rc == _lib.ROMA_E_UNSUPPORTED
    except TypeError:                                           # an older Pillow without keep_rgb
        pass
    rc, *_ = _host_decode(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    assert rc == _lib.ROMA_E_ARG
    data = open(ASSETS[0], "rb").read()
    rc, *_ = _host_decode(data[:600])                           # truncated inside the tables
    assert rc < 0


@pytest.mark.gpu
def test_device_reconstruction_is_bit_identical_to_pil():
    from roma_amd.preproc import decode_jpeg_device
    streams = {os.path.basename(f): open(f, "rb").read() for f in ASSETS}
    streams.update(_synthetic_jpegs())
    for name, data in streams.items():
        rgb = decode_jpeg_device(data, "cuda")
        assert rgb is not None and rgb.dtype == torch.uint8, name
        ref = torch.from_numpy(np.array(Image.open(io.BytesIO(data)).convert("RGB")))
        assert rgb.shape == ref.shape and torch.equal(rgb.cpu(), ref), name
    bio = io.BytesIO()
    Image.fromarray(np.zeros((32, 32, 3), np.uint8)).convert("CMYK").save(bio, "JPEG")
    assert decode_jpeg_device(bio.getvalue(), "cuda") is None  # the caller decodes such a stream with PIL
    assert torch.equal(decode_jpeg_device(ASSETS[1], "cuda").cpu(), torch.from_numpy(np.array(Image.open(ASSETS[1]).convert("RGB"))))   # a path


@pytest.mark.gpu
def test_match_on_jpeg_paths_feeds_the_same_bits_as_pil():
    """What match(path_A, path_B) feeds the network with the JPEGs decoded on the device — both resolutions, resized and normalised on
    the device — is BIT-identical to what it feeds with PIL decoding on the host; and match() runs on paths either way (two runs of
    the same inputs differ at the 1e-4 level through the library GEMMs' atomics, so outputs are compared with a tolerance; the e2e
    fixtures of test_gpu_model.py, generated from PIL-decoded inputs, go through the device decoder as well)."""
    from roma_amd.model_zoo import build_roma
    from roma_amd.preproc import decode_jpeg_device, preprocess_device
    for f in ASSETS[:2]:
        dev = decode_jpeg_device(f, "cuda")
        host = Image.open(f).convert("RGB")
        for size in ((112, 112), (560, 560), (864, 864)):
            assert torch.equal(preprocess_device(dev, size, "cuda"), preprocess_device(host, size, "cuda")), (f, size)
    torch.manual_seed(0)
    model = build_roma((112, 112), upsample_preds=True, amp_dtype=torch.float32)
    model.upsample_res = (168, 168)
    model = model.to("cuda").eval()
    assert model.device_jpeg
    w1, c1 = model.match(ASSETS[0], ASSETS[1], device="cuda")
    model.device_jpeg = False
    w2, c2 = model.match(ASSETS[0], ASSETS[1], device="cuda")
    assert torch.isfinite(w1).all() and torch.isfinite(c1).all()
    assert float((w1 - w2).abs().median()) < 1e-4 and float((c1 - c2).abs().median()) < 1e-4
