#!/usr/bin/env python3
"""Pack the CIE standard tables used by the tracer into optrace_amd/data/cie_tables.npz.

Inputs are published CIE data sets (not code):
  * CIE 1931 2-degree colour-matching functions, 360-830 nm, 1 nm (CIE 2018, DOI 10.25039/CIE.DS.xvudnb9b)
  * CIE standard illuminants A, C, D50, D55, D65, D75, F2, F7, F11 (CIE Colorimetry, 3rd ed., 2004) and
    LED illuminants (CIE 2018, DOI 10.25039/CIE.DS.vgssnyfg), 300-780 nm, 5 nm
read here from the CSV copies the reference keeps under optrace/resources/ (same parsing as
optrace/tracer/color/observers.py:11 and illuminants.py:13: empty cells -> 0).
Run once in the build container; the .npz is committed.
"""
import pathlib
import sys

import numpy as np

res = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/optrace/resources")
out = pathlib.Path(__file__).resolve().parent.parent / "optrace_amd" / "data" / "cie_tables.npz"

obs = np.genfromtxt(res / "observers.csv", skip_header=1, delimiter=",", filling_values=0, dtype=np.float64)
ill = np.genfromtxt(res / "illuminants.csv", skip_header=1, delimiter=",", filling_values=0, dtype=np.float64)
names = ["wl", "A", "C", "D50", "D55", "D65", "D75", "F2", "F7", "F11", "LED-B1", "LED-B2", "LED-B3",
         "LED-B4", "LED-B5", "LED-BH1", "LED-RGB1", "LED-V1", "LED-V2"]
assert obs.shape == (471, 4) and ill.shape[1] == len(names)
np.savez_compressed(out, observers=obs, illuminants=ill, illuminant_names=np.array(names))
print(out, obs.shape, ill.shape)
