"""convolve(): image formation by convolution with a point spread function, on the GPU.

Drop-in for `optrace.tracer.convolve.convolve` (convolve.py:49-454), the second half of SURVEY section 8f rank 2.
The contract is the reference's: four image / PSF type combinations, magnification `m` (scale and flip), padding
modes of numpy.pad, `keep_size`, result extent = image extent + PSF extent, linear-sRGB arithmetic with out-of-gamut
values kept until the final colour mapping.  What runs where:

  host    type / size checks, sRGB gamma removal and padding of the (<= 4 MP) image      NumPy, once per call
  device  PSF -> linear sRGB, area resize to the image's pixel pitch (two small f64 GEMMs with the exact
          pixel-overlap weights), zero padding, real 2-D FFTs of every distinct plane (rocFFT through torch.fft),
          spectral products, inverse FFTs, slicing, linear sRGB -> XYZ
  device  XYZ -> sRGB with the rendering intent, gamut mapping, normalisation and gamma: `ot_image_convert`
          (csrc/ot_image.hpp), the kernel behind RenderImage.get

scipy.signal.fftconvolve(mode="full") pads to the next 5-smooth length and multiplies real FFTs; the same is done here,
so results agree to rounding (tests/test_gpu_convolve.py against fixtures computed by the reference itself).
The reference resizes the PSF with cv2.resize(INTER_AREA) (convolve.py:335); here the area average is formed from
the pixel overlaps directly (identity when PSF and image share a pixel pitch, block means for integer ratios).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi
from ._device import require_device, stream_ptr, ptr
from ._warn import warning
from .base import check_type, check_above, check_not_below
from .image import RGBImage, GrayscaleImage, srgb_to_srgb_linear
from .render_image import RenderImage

# Bruce Lindbloom's sRGB (D65) matrices as the reference spells them (color/srgb.py:60-62, 106-108)
_RGBL_TO_XYZ = [[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]]
_XYZ_TO_RGBL = [[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]]


def _fast_len(n: int) -> int:
    """Smallest 5-smooth integer >= n (scipy.fft.next_fast_len's choice for real transforms)."""
    best = 1 << max(n - 1, 0).bit_length()
    p5 = 1
    while p5 < best:
        p35 = p5
        while p35 < best:
            m = p35
            while m < n:
                m *= 2
            best = min(best, m)
            p35 *= 3
        p5 *= 5
    return best


def _area_weights(n_in: int, n_out: int, dev) -> torch.Tensor:
    """(n_out, n_in) matrix of the area-average resize along one axis: output pixel j covers the source interval
    [j, j + 1) * n_in / n_out and takes every source pixel with the share of the interval it overlaps (rows sum to 1).
    Enlarging works the same way (an output pixel then lies inside one or two source pixels)."""
    scale = n_in / n_out
    j = torch.arange(n_out, dtype=torch.float64, device=dev)[:, None]
    i = torch.arange(n_in, dtype=torch.float64, device=dev)[None, :]
    lo, hi = j * scale, (j + 1) * scale
    overlap = (torch.minimum(hi, i + 1) - torch.maximum(lo, i)).clamp_(min=0.0)
    return overlap / scale


def _gamma(v: torch.Tensor) -> torch.Tensor:
    """color.srgb_linear_to_srgb (srgb.py:358-376), odd in its argument."""
    a = 0.055
    av = v.abs()
    return torch.where(av <= 0.0031308, v * 12.92, torch.sign(v) * ((1 + a) * av ** (1 / 2.4) - a))


def convolve(img, psf, m: float = 1, keep_size: bool = False, padding_mode: str = "constant", padding_value=None,
             cargs: dict = {}):
    """Convolve an image with a point spread function (same signature, cases and errors as the reference's).

    1. GrayscaleImage with a GrayscaleImage PSF -> GrayscaleImage
    2. GrayscaleImage with a RenderImage PSF (colour information in the PSF) -> RGBImage
    3. RGBImage with a GrayscaleImage PSF -> RGBImage
    4. RGBImage with a list of three RenderImage PSFs, rendered for the sRGB R, G, B primaries -> RGBImage

    `m` scales the image before the convolution (|m| > 1 enlarges, m < 0 flips it).  `padding_mode` is one of
    numpy.pad's modes, `padding_value` (three values for an RGBImage, one otherwise) belongs to "constant".
    `keep_size` crops the result back to the pixel count of the input.  `cargs` overrides the arguments of the final
    colour mapping, by default rendering_intent="Absolute", normalize=True, clip=True, L_th=0, chroma_scale=None.
    """
    check_type("m", m, (int, float))
    check_type("cargs", cargs, dict)
    check_above("abs(m)", abs(m), 0)
    check_type("keep_size", keep_size, bool)
    check_type("img", img, (RGBImage, GrayscaleImage))

    img_color = isinstance(img, RGBImage)
    gray_only = isinstance(psf, GrayscaleImage) and not img_color  # one plane in, one plane out
    three_psf = isinstance(psf, list) and len(psf) == 3
    psf_color = isinstance(psf, RenderImage) or three_psf
    dev = require_device()

    # ---- image: padding value and gamma removal (convolve.py:156-203) ----------------------------------------
    if img_color:
        check_type("padding_value", padding_value, (list, np.ndarray, type(None)))
        pval = np.zeros(3) if padding_value is None else np.asarray(padding_value, dtype=np.float64)
        if pval.ndim != 1 or pval.shape[0] != 3:
            raise ValueError(f"padding_value must be a 3 element array/list, but has shape {pval.shape}")
        if np.any(pval < 0):
            raise ValueError("value in 'padding_value' needs to be non-negative.")
    else:
        check_type("padding_value", padding_value, (int, float, type(None)))
        pv = 0. if padding_value is None else float(padding_value)
        check_not_below("padding_value", pv, 0)
        pval = np.full(3, pv)
    pval_lin = srgb_to_srgb_linear(pval.copy())
    planes = srgb_to_srgb_linear(img.data)  # (Ny, Nx, 3) or (Ny, Nx)
    custom_padding = not (padding_mode == "constant" and float(pval_lin.sum()) == 0)

    # ---- PSF checks (convolve.py:206-253) ----------------------------------------------------------------------
    if psf_color:
        psfs = psf if three_psf else [psf]
        if img_color and not three_psf:
            raise TypeError("A list of a R, G, B RenderImage PSF is required for convolving"
                            " a colored image with a colored PSF.")
        if not img_color and three_psf:
            raise TypeError("A single colored RenderImage is sufficient for a grayscale image.")
        for i, p_i in enumerate(psfs):
            check_type(f"psf[{i}]", p_i, RenderImage)
            if not np.all(np.asarray(psfs[0].extent) == np.asarray(p_i.extent)):
                raise ValueError("All PSF sizes need to be the same. Render the detector image with"
                                 " the same manual extent option.")
    else:
        check_type("psf", psf, GrayscaleImage)
        psfs = [psf]

    # ---- sizes (convolve.py:256-339): pixel centres span the side lengths, so pitch = s / (n - 1) ------------
    iN = np.array([img.shape[1], img.shape[0]])
    pN = np.array([psfs[0].shape[1], psfs[0].shape[0]])
    is_ = np.array(img.s) * abs(m)
    ps_ = np.array(psfs[0].s)
    ip, pp = is_ / (iN - 1), ps_ / (pN - 1)
    if ps_[0] > 2 * is_[0] or ps_[1] > 2 * is_[1]:
        raise ValueError(f"m-scaled image size [{is_[0]:.5g}, {is_[1]:.5g}] is more than two times "
                         f"smaller than PSF size [{ps_[0]:.5g}, {ps_[1]:.5g}].")
    if pN[0] * pN[1] > 4e6:
        raise ValueError("PSF needs to be smaller than 4MP")
    if iN[0] * iN[1] > 4e6:
        raise ValueError("Image needs to be smaller than 4MP")
    if pp[0] > ip[0] or pp[1] > ip[1]:
        warning(f"PSF pixel sizes [{pp[0]:.5g}, {pp[1]:.5g}] larger than image pixel sizes"
                f" [{ip[0]:.5g}, {ip[1]:.5g}], generally you want a PSF in a higher resolution")
    if pN[0] < 50 or pN[1] < 50:
        raise ValueError(f"PSF too small with shape {psfs[0].shape}, "
                         "needs to have at least 50 values in each dimension.")
    if iN[0] < 50 or iN[1] < 50:
        raise ValueError(f"Image too small with shape {img.shape}, needs to have at least 50 values in each dimension.")
    if iN[0] * iN[1] < 2e4:
        warning("Low resolution image.")
    if pN[0] * pN[1] < 2e4:
        warning("Low resolution PSF.")
    if not (0.2 < pp[0] / pp[1] < 5):
        warning(f"Pixels of PSF are strongly non-square with side lengths [{pp[0]}mm, {pp[1]}mm]")
    if not (0.2 < ip[0] / ip[1] < 5):
        warning(f"Pixels of image are strongly non-square with side lengths [{ip[0]}mm, {ip[1]}mm]")

    sc = pp / ip
    ppad = 4                                                            # zero rim around the resized PSF
    p2N = np.where(pN * sc < 1, 1, np.round(pN * sc).astype(int))       # PSF pixels at the image's pitch
    p3N = p2N + 2 * ppad
    ipad = p3N if custom_padding else np.array([0, 0])                  # image rim for the padding mode
    i2N = iN + 2 * ipad
    i3N = i2N + p3N - 1                                                 # "full" convolution
    i4N = iN if keep_size else iN + p3N - 1
    i4s = (i4N - 1) * ip
    ext = np.asarray(img.extent, dtype=np.float64) + np.asarray(psfs[0].extent, dtype=np.float64)
    xm, ym = (ext[0] + ext[1]) / 2, (ext[2] + ext[3]) / 2
    i4e = [xm - i4s[0] / 2, xm + i4s[0] / 2, ym - i4s[1] / 2, ym + i4s[1] / 2]

    # ---- image planes on the device: pad for the mode, flip for m < 0 (convolve.py:342-371) -------------------
    if custom_padding:
        width = ((int(ipad[1]),) * 2, (int(ipad[0]),) * 2) + (((0, 0),) if planes.ndim == 3 else ())
        if padding_mode == "constant":
            if planes.ndim == 3:
                padded = np.empty((planes.shape[0] + 2 * ipad[1], planes.shape[1] + 2 * ipad[0], 3))
                padded[:] = pval_lin
                padded[ipad[1]:ipad[1] + planes.shape[0], ipad[0]:ipad[0] + planes.shape[1]] = planes
                planes = padded
            else:
                planes = np.pad(planes, width, mode="constant", constant_values=pval_lin[0])
        else:
            planes = np.pad(planes, width, mode=padding_mode)
    if m < 0:
        planes = planes[::-1, ::-1]
    d_img = torch.from_numpy(np.ascontiguousarray(planes)).to(dev)
    img_planes = [d_img] if d_img.ndim == 2 else [d_img[:, :, c] for c in range(3)]
    if not img_color and not gray_only:
        img_planes = img_planes * 3  # a grey image has three equal linear-sRGB channels

    # ---- PSF planes: linear sRGB, area resize, zero rim (convolve.py:228-250, 373-397) ------------------------
    Wy = _area_weights(int(pN[1]), int(p2N[1]), dev)
    Wx = _area_weights(int(pN[0]), int(p2N[0]), dev)
    keep = float(pN[0] * pN[1]) / float(p2N[0] * p2N[1])  # an area average times this keeps the PSF's sum
    t_x2r = torch.tensor(_XYZ_TO_RGBL, dtype=torch.float64, device=dev)

    def psf_planes(p) -> list:
        if isinstance(p, RenderImage):
            p._check_for_image()
            xyz = (p._dev if p._dev is not None else torch.from_numpy(p._data).to(dev))[:, :, :3]
            lin = xyz.reshape(-1, 3) @ t_x2r.T  # color.xyz_to_srgb_linear(rendering_intent="Ignore", normalize=False)
            chans = [lin[:, c].reshape(xyz.shape[0], xyz.shape[1]) for c in range(3)]
        else:
            g = torch.from_numpy(srgb_to_srgb_linear(p.data)).to(dev)
            tot = g.sum()
            if float(tot):
                g = g * (1 / tot)
            chans = [g]
        out = []
        for ch in chans:
            small = (Wy @ ch @ Wx.T) * keep
            out.append(torch.nn.functional.pad(small, (ppad, ppad, ppad, ppad)))
        return out

    psf_sets = [psf_planes(p) for p in psfs]  # per PSF: one plane (grey) or three (R, G, B response)

    # ---- convolution: real FFTs of every distinct plane, products, inverse transforms (convolve.py:400-437) ---
    Ly, Lx = _fast_len(int(i3N[1])), _fast_len(int(i3N[0]))
    f_img = [torch.fft.rfft2(pl, s=(Ly, Lx)) for pl in (img_planes if img_color or gray_only else img_planes[:1])]
    if not img_color and not gray_only:
        f_img = f_img * 3
    f_psf = [[torch.fft.rfft2(pl, s=(Ly, Lx)) for pl in planes_] for planes_ in psf_sets]

    def back(spec) -> torch.Tensor:
        return torch.fft.irfft2(spec, s=(Ly, Lx))[:int(i3N[1]), :int(i3N[0])]

    if gray_only:
        res = back(f_img[0] * f_psf[0][0])                                            # (Ny, Nx)
    elif three_psf:  # image channel i spreads into all three output channels through PSF i
        res = torch.stack([back(sum(f_img[i] * f_psf[i][j] for i in range(3))) for j in range(3)], dim=2)
    else:            # channel by channel (a grey PSF serves all three)
        one = f_psf[0]
        res = torch.stack([back(f_img[c] * one[c if len(one) == 3 else 0]) for c in range(3)], dim=2)

    # ---- slicing (convolve.py:440-452) ------------------------------------------------------------------------
    if custom_padding:
        res = res[ipad[1]:res.shape[0] - ipad[1], ipad[0]:res.shape[1] - ipad[0]]
    if keep_size:
        off = (i3N - i2N) // 2
        res = res[off[1]:off[1] + iN[1], off[0]:off[0] + iN[0]]

    # ---- back to sRGB ------------------------------------------------------------------------------------------
    if gray_only:
        if cargs.get("normalize", True):
            mx = float(res.max())
            if mx:
                res = res * (1 / mx)
        out = _gamma(res.clamp(0, 1)).cpu().numpy()
        return GrayscaleImage(out, extent=i4e)

    opts = dict(rendering_intent="Absolute", normalize=True, clip=True, L_th=0, chroma_scale=None) | cargs
    ny, nx = int(res.shape[0]), int(res.shape[1])
    t_r2x = torch.tensor(_RGBL_TO_XYZ, dtype=torch.float64, device=dev)
    xyz = res.reshape(-1, 3) @ t_r2x.T
    intent = opts["rendering_intent"]
    if intent == "Ignore":  # plain matrix conversion, out-of-gamut values only clipped (srgb.py:318-319 path)
        lin = xyz @ t_x2r.T
        if opts["normalize"]:
            mx = float(torch.nan_to_num(lin, nan=-np.inf).max())
            if mx:
                lin = lin * (1 / mx)
        if opts["clip"]:
            lin = lin.clamp(0, 1)
        return RGBImage(_gamma(lin).reshape(ny, nx, 3).cpu().numpy(), extent=i4e)
    if intent not in ("Absolute", "Perceptual"):
        raise ValueError(f"Invalid rendering_intent '{intent}'.")

    lib = _capi.load_library()
    hist = torch.zeros(ny * nx, 4, dtype=torch.float64, device=dev)
    hist[:, :3] = xyz
    out = torch.empty(ny * nx * 3, dtype=torch.float64, device=dev)
    ws = torch.empty(4 * nx * ny + 8, dtype=torch.float64, device=dev)
    mode = 2 if intent == "Absolute" else 3  # OT_IMG_SRGB_ABSOLUTE / OT_IMG_SRGB_PERCEPTUAL
    if not opts["normalize"]:
        mode |= 0x100  # OT_IMG_FLAG_NO_NORMALIZE
    if not opts["clip"]:
        mode |= 0x200  # OT_IMG_FLAG_NO_CLIP
    cs = float("nan") if opts["chroma_scale"] is None else float(opts["chroma_scale"])
    _capi.check(lib.ot_image_convert(ptr(hist), nx, ny, 1, mode, 1.0, 683.0, float(opts["L_th"]), cs, ptr(out), ptr(ws),
                                     stream_ptr()))
    return RGBImage(out.cpu().numpy().reshape(ny, nx, 3), extent=i4e)
