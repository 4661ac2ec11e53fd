"""Image preprocessing of the LLaVA SigLIP tower's `image_processor`.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Call site in the reference: test/inference.py:203 (`image_processor.preprocess(frames, return_tensors='pt')['pixel_values']`).
The processor itself is LLaVA-NeXT's SigLipImageProcessor ([3P-recalled], llava/model/multimodal_encoder/siglip_encoder.py):
  convert_to_rgb -> resize(size=(384,384), resample=BICUBIC) through PIL on the uint8 image -> rescale 1/255 ->
  normalize(mean .5, std .5) -> channels-first float32.
Frames from test/datasets.py:85 are uint8 [T,3,R,R].  When R == size the PIL resize is the identity.
"""
import numpy as np
import torch


def siglip_preprocess(frames, size=384):
    from PIL import Image
    arr = np.asarray(frames)
    out = np.empty((arr.shape[0], 3, size, size), dtype=np.float32)
    for t in range(arr.shape[0]):
        a = arr[t].transpose(1, 2, 0)
        if a.shape[0] != size or a.shape[1] != size:
            a = np.asarray(Image.fromarray(a.astype(np.uint8)).resize((size, size), resample=Image.BICUBIC))
        f = a.astype(np.float32) * np.float32(1 / 255)
        out[t] = ((f - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1)
    return torch.from_numpy(out)


# ---- explicit restatement of Pillow's 8-bit separable resampler (src/libImaging/Resample.c) -----------------------
# Used to pin the HIP preprocess kernel bit-exactly without needing PIL semantics to be re-derived on the device side.
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size, out_size):
    """precompute_coeffs() + normalize_coeffs_8bpc(): per output index (xmin, int32 taps[ksize]).  support = 2 (bicubic);
    for upscaling filterscale = 1."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        ww = 0.0
        for x in range(xmax):
            wgt = _bicubic((x + xmin - center + 0.5) * ss)
            k[x] = wgt
            ww += wgt
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        ki = []
        for x in range(ksize):
            v = k[x]
            ki.append(int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)))
        bounds.append((xmin, xmax))
        kk.append(ki)
    return bounds, np.asarray(kk, dtype=np.int64), ksize


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_bicubic_u8(img_hwc, out_size):
    """ImagingResample for 8-bit images: horizontal pass into a uint8 temp, then the vertical pass
    (ImagingResampleInner; both passes round with `ss = 1 << (PRECISION_BITS-1)` then clip8)."""
    Hin, Win, C = img_hwc.shape
    src = img_hwc.astype(np.int64)
    bx, kx, _ = pil_bicubic_coeffs(Win, out_size)
    tmp = np.empty((Hin, out_size, C), dtype=np.uint8)
    for xx in range(out_size):
        xmin, xmax = bx[xx]
        acc = np.full((Hin, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(xmax):
            acc += src[:, xmin + x, :] * kx[xx, x]
        tmp[:, xx, :] = _clip8(acc)
    by, ky, _ = pil_bicubic_coeffs(Hin, out_size)
    t64 = tmp.astype(np.int64)
    out = np.empty((out_size, out_size, C), dtype=np.uint8)
    for yy in range(out_size):
        ymin, ymax = by[yy]
        acc = np.full((out_size, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for y in range(ymax):
            acc += t64[ymin + y] * ky[yy, y]
        out[yy] = _clip8(acc)
    return out
