"""CPU restatement of the face-crop resize of the reference's `face_rec` (model/pred_func.py:67-92):

    face_image = cv2.resize(frame[top:bottom, left:right], (224, 224), interpolation=cv2.INTER_AREA)

TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

**Parity unpinned.**  The arithmetic lives in OpenCV (`opencv-python`, a dependency the reference imports as `cv2`;
not vendored under /root/reference and not installed in the build image), and the reference holds no fixture for it.
What follows restates OpenCV 4.x `modules/imgproc/src/resize.cpp` as published, for 8-bit 3-channel images:

  * both dimensions shrink by whole factors      -> `ResizeAreaFast_`: integer block sums, `cvRound(sum * (1.f / area))`,
                                                    and the 2x2 special case `(a + b + c + d + 2) >> 2`
  * both dimensions shrink (any factor >= 1)     -> `ResizeArea_` over `computeResizeAreaTab` tables: fp32 running sums,
                                                    x first then y, one multiply and one add per term, `cvRound` at the end
  * otherwise (a dimension grows)                -> the bilinear path in "area mode": coefficients from
                                                    `fx = (dx + 1) - (sx + 1) * inv_scale`, 11-bit fixed point,
                                                    `HResizeLinear` / `VResizeLinear` integer arithmetic

It is tied (tests/test_oracle.py) to the definition — at most half an LSB from the exact area mean (float64 brute force)
on shrinking sizes, exact block means on whole factors, the integer passes of the growing branch against the same weights
in floating point — and to Pillow's BOX filter, an independent area-averaging resampler, within 1 LSB on whole factors
(on fractional factors Pillow's BOX counts whole source pixels only, a different function).  Bit-equality with a given
OpenCV build (IPP / OpenCL / HAL overrides differ between builds) is NOT claimed.
"""
import math

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS
DBL_EPSILON = 2.220446049250313e-16


def _cv_round(v):
    """cvRound: round half to even (lrint in the default rounding mode)."""
    return np.rint(v)


def _area_tab(ssize, dsize, scale):
    """computeResizeAreaTab: list of (dst index, src index, fp32 weight), in table order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _area_tab_by_rank(ssize, dsize, scale):
    """the table of `_area_tab` regrouped as (K, dsize) arrays: k-th (source index, weight) of every destination"""
    per = [[] for _ in range(dsize)]
    for (d, s_, a) in _area_tab(ssize, dsize, scale):
        per[d].append((s_, a))
    K = max(len(p) for p in per)
    idx = np.zeros((K, dsize), dtype=np.int64)
    w = np.zeros((K, dsize), dtype=np.float32)
    for d, p in enumerate(per):
        for k, (s_, a) in enumerate(p):
            idx[k, d], w[k, d] = s_, a
    return idx, w


def _linear_tab(ssize, dsize, scale, inv_scale):
    """area-mode bilinear coefficients of cv::resize: per dst index (src index, a0, a1 as 11-bit fixed point) and the
    first dst index whose right neighbour falls outside (`xmax`)."""
    ofs, coef, xmax = [], [], dsize
    for dx in range(dsize):
        sx = math.floor(dx * scale)
        fx = np.float32((dx + 1) - (sx + 1) * inv_scale)
        fx = np.float32(0.0) if fx <= 0 else np.float32(fx - np.floor(fx))
        if sx < 0:
            fx, sx = np.float32(0.0), 0
        if sx + 1 >= ssize:
            xmax = min(xmax, dx)
            if sx >= ssize - 1:
                fx, sx = np.float32(0.0), ssize - 1
        a0 = int(np.clip(_cv_round(np.float32(np.float32(1.0) - fx) * np.float32(COEF_SCALE)), -32768, 32767))
        a1 = int(np.clip(_cv_round(np.float32(fx) * np.float32(COEF_SCALE)), -32768, 32767))
        ofs.append(sx)
        coef.append((a0, a1))
    return ofs, coef, xmax


def resize_area_u8(src, dw=224, dh=224):
    """src: (h, w, 3) uint8 -> (dh, dw, 3) uint8 with cv2.INTER_AREA semantics (see the module docstring)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw, cn = src.shape
    inv_sx, inv_sy = dw / sw, dh / sh
    scale_x, scale_y = 1.0 / inv_sx, 1.0 / inv_sy
    isx, isy = int(_cv_round(scale_x)), int(_cv_round(scale_y))
    fast = abs(scale_x - isx) < DBL_EPSILON and abs(scale_y - isy) < DBL_EPSILON
    if scale_x >= 1 and scale_y >= 1:
        if fast:
            blk = src[:dh * isy, :dw * isx].reshape(dh, isy, dw, isx, cn).astype(np.int64)
            s = blk.sum(axis=(1, 3))
            if isx == 2 and isy == 2:
                return ((s + 2) >> 2).astype(np.uint8)
            scale = np.float32(1.0) / np.float32(isx * isy)
            return np.clip(_cv_round(s.astype(np.float32) * scale), 0, 255).astype(np.uint8)
        # ResizeArea_: buf[dx] = buf[dx] + S * alpha over a destination column's table entries in table order, then
        # sum[dx] = sum[dx] + beta * buf[dx] over a destination row's entries in table order (fp32, one multiply and
        # one add per term).  Vectorised over destinations by "k-th entry of every destination"; a destination with
        # fewer entries adds +0.0, which leaves an fp32 value unchanged.
        (xi, xa_), (yi, ya_) = _area_tab_by_rank(sw, dw, scale_x), _area_tab_by_rank(sh, dh, scale_y)
        S = src.astype(np.float32)
        buf = np.zeros((sh, dw, cn), dtype=np.float32)
        for k in range(xi.shape[0]):
            buf = buf + S[:, xi[k], :] * xa_[k][None, :, None]
        total = np.zeros((dh, dw, cn), dtype=np.float32)
        for k in range(yi.shape[0]):
            total = total + ya_[k][:, None, None] * buf[yi[k]]
        return np.clip(_cv_round(total), 0, 255).astype(np.uint8)
    xofs, xa, xmax = _linear_tab(sw, dw, scale_x, inv_sx)
    yofs, ya, _ = _linear_tab(sh, dh, scale_y, inv_sy)
    S = src.astype(np.int64)
    rows = np.zeros((sh, dw, cn), dtype=np.int64)               # HResizeLinear of every source row
    for dx in range(dw):
        sx = xofs[dx]
        if dx < xmax:
            rows[:, dx] = S[:, sx] * xa[dx][0] + S[:, sx + 1] * xa[dx][1]
        else:
            rows[:, dx] = S[:, sx] * COEF_SCALE
    out = np.zeros((dh, dw, cn), dtype=np.uint8)
    for dy in range(dh):
        s0 = min(max(yofs[dy], 0), sh - 1)
        s1 = min(max(yofs[dy] + 1, 0), sh - 1)
        b0, b1 = ya[dy]
        v = (((b0 * (rows[s0] >> 4)) >> 16) + ((b1 * (rows[s1] >> 4)) >> 16) + 2) >> 2
        out[dy] = v.astype(np.uint8)                             # `uchar(...)`: a plain narrowing cast
    return out


def face_crops(frames, boxes, size=224):
    """frames: (F, H, W, 3) uint8; boxes: rows of (frame, top, right, bottom, left) as face_recognition returns them
    (pred_func.py:79-81).  The reference's RGB<->BGR swaps around the resize cancel (the resize is per channel)."""
    out = np.zeros((len(boxes), size, size, 3), dtype=np.uint8)
    for i, (f, top, right, bottom, left) in enumerate(boxes):
        out[i] = resize_area_u8(frames[f][top:bottom, left:right], size, size)
    return out
