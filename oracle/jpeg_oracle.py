"""CPU restatement (numpy) of what PIL / libjpeg-turbo does AFTER entropy decoding with its defaults — the checker of
roma_jpeg_reconstruct (roma_amd/csrc/jpeg.hip).  TEST INFRASTRUCTURE: imported by tests/ only; the product path never touches it.

Reference behaviour: romatch/models/matcher.py:606-637, 667-676 (`Image.open(path).convert("RGB")`); the decoder behind it is a
third-party dependency that is not vendored in /root/reference (Pillow's bundled libjpeg-turbo, JDCT_ISLOW + fancy up-sampling), so its
published algorithm is restated here and PINNED by comparing with PIL itself on the bundled photographs (tests/test_host_logic.py):

* inverse DCT: the "islow" integer transform (Loeffler-Ligtenberg-Moschytz, 13-bit constants, first pass scaled by 2^2), +128, clamp;
* h2v2 "fancy" up-sampling: 3/4 nearer + 1/4 farther sample in each direction, biases 8 / 7 alternating along a row, edges replicated,
  over ceil(W/2) x ceil(H/2) chroma samples (plain replication when there are at most two chroma columns); for 4:2:2 the horizontal half
  of it alone (biases 1 / 2, the first and last output column are the samples themselves);
* YCbCr -> RGB with the 16-bit fixed-point constants 1.40200, 1.77200, 0.71414, 0.34414.
"""
import numpy as np

F0_298, F0_390, F0_541, F0_765, F0_899, F1_175 = 2446, 3196, 4433, 6270, 7373, 9633
F1_501, F1_847, F1_961, F2_053, F2_562, F3_072 = 12299, 15137, 16069, 16819, 20995, 25172


def _idct8(v, shift):
    """8-point pass along axis 0 of an int64 array (8, ...)."""
    z2, z3 = v[2], v[6]
    z1 = (z2 + z3) * F0_541
    t2 = z1 - z3 * F1_847
    t3 = z1 + z2 * F0_765
    z2, z3 = v[0], v[4]
    t0, t1 = (z2 + z3) << 13, (z2 - z3) << 13
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    o0, o1, o2, o3 = v[7], v[5], v[3], v[1]
    z1, z2, z3, z4 = o0 + o3, o1 + o2, o0 + o2, o1 + o3
    z5 = (z3 + z4) * F1_175
    o0, o1, o2, o3 = o0 * F0_298, o1 * F2_053, o2 * F3_072, o3 * F1_501
    z1, z2, z3, z4 = -z1 * F0_899, -z2 * F2_562, -z3 * F1_961 + z5, -z4 * F0_390 + z5
    o0, o1, o2, o3 = o0 + z1 + z3, o1 + z2 + z4, o2 + z2 + z3, o3 + z1 + z4
    r = 1 << (shift - 1)
    return np.stack([t10 + o3, t11 + o2, t12 + o1, t13 + o0, t13 - o0, t12 - o1, t11 - o2, t10 - o3]) + r >> shift


def idct_plane(coef, qt, blocks_w, blocks_h):
    """coef (nblocks, 64) int16 natural order, qt (64,) -> uint8 plane (blocks_h * 8, blocks_w * 8)."""
    c = coef.astype(np.int64).reshape(-1, 8, 8) * qt.astype(np.int64).reshape(1, 8, 8)
    ws = _idct8(c.transpose(1, 0, 2), 11)                      # pass 1 down the columns: axis 0 = row index
    out = _idct8(ws.transpose(2, 1, 0), 18)                    # pass 2 along the rows: axis 0 = column index -> (col, block, row)
    px = np.clip(out.transpose(1, 2, 0) + 128, 0, 255).astype(np.uint8)     # (block, row, col)
    return px.reshape(blocks_h, blocks_w, 8, 8).transpose(0, 2, 1, 3).reshape(blocks_h * 8, blocks_w * 8)


def upsample_h2v2(p, W, H):
    cw, ch = (W + 1) // 2, (H + 1) // 2
    p = p[:ch, :cw].astype(np.int64)
    if cw <= 2:                                                 # libjpeg selects the triangle filter only for more than two chroma columns;
        return np.repeat(np.repeat(p, 2, axis=0), 2, axis=1)[:H, :W]   # narrower images get plain replication
    up, dn = np.vstack([p[:1], p[:-1]]), np.vstack([p[1:], p[-1:]])
    rows = np.empty((2 * ch, cw), np.int64)
    rows[0::2], rows[1::2] = 3 * p + up, 3 * p + dn            # column sums of output rows 2r, 2r + 1
    left, right = np.hstack([rows[:, :1], rows[:, :-1]]), np.hstack([rows[:, 1:], rows[:, -1:]])
    out = np.empty((2 * ch, 2 * cw), np.int64)
    out[:, 0::2], out[:, 1::2] = (3 * rows + left + 8) >> 4, (3 * rows + right + 7) >> 4
    return out[:H, :W]


def upsample_h2v1(p, W, H):
    """4:2:2: horizontal triangle filter only (biases 1 / 2; the first and last output column are the samples themselves)."""
    cw = (W + 1) // 2
    p = p[:H, :cw].astype(np.int64)
    if cw <= 2:
        return np.repeat(p, 2, axis=1)[:, :W]
    left, right = np.hstack([p[:, :1], p[:, :-1]]), np.hstack([p[:, 1:], p[:, -1:]])
    out = np.empty((H, 2 * cw), np.int64)
    out[:, 0::2], out[:, 1::2] = (3 * p + left + 1) >> 2, (3 * p + right + 2) >> 2
    out[:, 0], out[:, 2 * cw - 1] = p[:, 0], p[:, -1]
    return out[:, :W]


def reconstruct(coef, qt, info):
    """What roma_jpeg_reconstruct computes: coef (nblocks, 64) int16, qt (3, 64) uint16, info from roma_jpeg_info -> uint8 (H, W, 3)."""
    W, H, nc, sub, ybw, ybh, cbw, cbh = [int(v) for v in info]
    ny, ncb = ybw * ybh, cbw * cbh
    Y = idct_plane(coef[:ny], qt[0], ybw, ybh)[:H, :W].astype(np.int64)
    if nc == 1:
        return np.repeat(Y[..., None], 3, axis=2).astype(np.uint8)
    cb = idct_plane(coef[ny:ny + ncb], qt[1], cbw, cbh)
    cr = idct_plane(coef[ny + ncb:ny + 2 * ncb], qt[2], cbw, cbh)
    if sub == 1:
        cb, cr = upsample_h2v2(cb, W, H), upsample_h2v2(cr, W, H)
    elif sub == 2:
        cb, cr = upsample_h2v1(cb, W, H), upsample_h2v1(cr, W, H)
    else:
        cb, cr = cb[:H, :W].astype(np.int64), cr[:H, :W].astype(np.int64)
    xb, xr = cb - 128, cr - 128
    R = Y + ((91881 * xr + 32768) >> 16)
    B = Y + ((116130 * xb + 32768) >> 16)
    G = Y + ((-22554 * xb + 32768 - 46802 * xr) >> 16)
    return np.clip(np.stack([R, G, B], axis=2), 0, 255).astype(np.uint8)
