import sys, pathlib, time
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np, torch
import optrace_amd as ot, scenes
from optrace_amd.ray_storage import RayStorage

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
mono = ot.LightSpectrum("Monochromatic", wl=550.)
img = ot.RGBImage(scenes.synthetic_rgb_image(), [4, 3])
cases = {
    "point none mono": dict(surface=ot.Point(), divergence="None", spectrum=mono),
    "point iso mono": dict(surface=ot.Point(), divergence="Isotropic", div_angle=5, spectrum=mono),
    "point lamb mono": dict(surface=ot.Point(), divergence="Lambertian", div_angle=5, spectrum=mono),
    "point iso d65": dict(surface=ot.Point(), divergence="Isotropic", div_angle=5, spectrum=ot.presets.light_spectrum.d65),
    "point iso lines": dict(surface=ot.Point(), divergence="Isotropic", div_angle=5, spectrum=ot.LightSpectrum("Lines", lines=[486.1327, 589.2938, 656.272], line_vals=[1, 1, 1])),
    "rect none mono": dict(surface=ot.RectangularSurface(dim=[2, 2]), divergence="None", spectrum=mono),
    "disc none mono": dict(surface=ot.CircularSurface(r=2), divergence="None", spectrum=mono),
    "rect iso mono": dict(surface=ot.RectangularSurface(dim=[2, 2]), divergence="Isotropic", div_angle=5, spectrum=mono),
    "rgb none": dict(surface=img, divergence="None"),
    "rgb iso conv": dict(surface=img, divergence="Isotropic", div_angle=5, orientation="Converging", conv_pos=[0, 0, 12]),
}
for no_pol in (True, False):
    for name, kw in cases.items():
        rs = ot.RaySource(pos=[0, 0, 0], **kw)
        st = RayStorage()
        st.init([rs], N, 1, no_pol)
        st.generate(seed=3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            st.generate(seed=4)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"{'no_pol' if no_pol else 'pol   '} {name:18s} {ms:7.3f} ms  {ms * 1e9 / N:6.1f} ps/ray", flush=True)
