"""Inputs of the convolve() fixtures, shared by the generator (tests/golden/generate_golden.py, which runs them through
the reference) and by tests/test_gpu_convolve.py (which runs them through optrace_amd)."""
import numpy as np


def convolve_cases():
    """Inputs of the convolve() fixtures (shared with tests/test_gpu_convolve.py): name -> dict(img, img_s, psf,
    psf_s, kwargs).  Images are multiples of 1/255 (stored as uint8), colour PSFs XYZW float32; every PSF has the pixel
    pitch of the (m-scaled) image, so the reference's cv2.resize call is the identity (oracle/refload.py)."""
    rng = np.random.default_rng(77)
    ny, nx, pn = 120, 150, 61

    def image(color: bool) -> np.ndarray:
        yy, xx = np.mgrid[0:ny, 0:nx]
        base = ((xx // 15 + yy // 12) % 2) * 0.7 + 0.15 + 0.1 * np.sin(xx / 7.0) * np.cos(yy / 5.0)
        if not color:
            return np.round(np.clip(base, 0, 1) * 255).astype(np.uint8)
        rgb = np.stack([base, np.roll(base, 9, axis=1) * (yy / ny), np.roll(base, 5, axis=0) * (1 - xx / nx)], axis=2)
        return np.round(np.clip(rgb + rng.uniform(0, 0.05, rgb.shape), 0, 1) * 255).astype(np.uint8)

    def gray_psf() -> np.ndarray:  # an off-centre blob with a ring: values in [0, 1] as uint8
        yy, xx = np.mgrid[0:pn, 0:pn]
        r2 = (xx - 33.0) ** 2 + (yy - 27.0) ** 2
        v = np.exp(-r2 / 40.0) + 0.3 * np.exp(-(np.sqrt(r2) - 14.0) ** 2 / 6.0)
        return np.round(v / v.max() * 255).astype(np.uint8)

    def color_psf(shift: float, tint) -> np.ndarray:  # XYZW with a wavelength-like lateral colour shift
        yy, xx = np.mgrid[0:pn, 0:pn]
        out = np.zeros((pn, pn, 4), dtype=np.float32)
        for c, (dx, amp) in enumerate(zip((-shift, 0.0, shift), tint)):
            out[:, :, c] = amp * np.exp(-((xx - 30.0 - dx) ** 2 + (yy - 30.0) ** 2) / (18.0 + 6 * c))
        out[:, :, 3] = out[:, :, 1]
        return out

    def sizes(m: float):  # image side lengths, and the PSF side lengths that give the same pitch after scaling by |m|
        img_s = [0.149 * 10, 0.119 * 10]                       # pitch 0.01 * 10 / ... = 0.1 mm / 10
        ip = np.array(img_s) * abs(m) / (np.array([nx, ny]) - 1)
        return img_s, list(ip * (pn - 1))

    cases = {}
    for name, color, psf_kind, m, kw in [
            ("gray_gray", False, "gray", 1.0, {}),
            ("gray_gray_flip_scale", False, "gray", -2.0, dict(keep_size=True, cargs=dict(normalize=False))),
            ("rgb_gray_edge_keep", True, "gray", 1.0, dict(padding_mode="edge", keep_size=True)),
            ("rgb_gray_padvalue", True, "gray", 0.5, dict(padding_value=[0.2, 0.1, 0.3])),
            ("gray_colorpsf", False, "color", 1.0, {}),
            ("gray_colorpsf_perceptual", False, "color", -1.0, dict(padding_mode="reflect", padding_value=None,
                                                                     cargs=dict(rendering_intent="Perceptual", L_th=0.01))),
            ("rgb_threepsf", True, "three", 1.0, dict(cargs=dict(normalize=False))),
            ("rgb_threepsf_ignore", True, "three", 1.0, dict(keep_size=True, cargs=dict(rendering_intent="Ignore"))),
    ]:
        img_s, psf_s = sizes(m)
        if psf_kind == "gray":
            psf = gray_psf()
        elif psf_kind == "color":
            psf = color_psf(3.0, (0.9, 1.0, 0.4))
        else:
            psf = np.stack([color_psf(2.0, (1.0, 0.5, 0.02)), color_psf(0.0, (0.35, 1.0, 0.12)), color_psf(-2.5, (0.2, 0.08, 1.0))])
        cases[name] = dict(img=image(color), img_s=img_s, psf=psf, psf_s=psf_s, psf_kind=psf_kind, m=m, kwargs=kw)
    return cases


def build_convolve_inputs(otm, case):
    """(img, psf) objects of package `otm` (the reference here, optrace_amd in the tests) for one case."""
    data = case["img"].astype(np.float64) / 255
    img = (otm.RGBImage if data.ndim == 3 else otm.GrayscaleImage)(data, case["img_s"])
    ps = case["psf_s"]
    ext = [-ps[0] / 2 + 0.02, ps[0] / 2 + 0.02, -ps[1] / 2 - 0.01, ps[1] / 2 - 0.01]  # off-centre PSF extent

    def render_image(xyzw):
        r = otm.RenderImage(ext)
        if hasattr(r, "_dev"):      # optrace_amd keeps the histogram in device memory
            import torch
            r._dev = torch.from_numpy(xyzw.astype(np.float64)).cuda()
        else:
            r._data = xyzw.astype(np.float64)
        return r

    if case["psf_kind"] == "gray":
        psf = otm.GrayscaleImage(case["psf"].astype(np.float64) / 255, extent=ext)
    elif case["psf_kind"] == "color":
        psf = render_image(case["psf"])
    else:
        psf = [render_image(p) for p in case["psf"]]
    return img, psf
