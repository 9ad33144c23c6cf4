"""Loader for the upstream NumPy reference (optrace @ /root/reference) -- TEST INFRASTRUCTURE ONLY.

Used only by the fixture generators under tests/golden/ in the build container: the reference
never travels to the GPU box.  Nothing in the product path (optrace_amd/) imports this file.

Recipe from SURVEY.md section 8c: the reference needs three GUI/IO modules that are absent from
this image (traits, cv2, chardet) and two `typing` names that only exist on Python >= 3.11.  None
of them are touched by the tracing hot path, so empty stand-in modules are registered before import.
"""
import sys
import types
import typing

REFERENCE_ROOT = "/root/reference"


def load(seed: int | None = None):
    """Import the reference package and (optionally) seed both of its RNG streams.

    Returns the imported ``optrace`` module.  With ``seed`` given, ``np.random`` (HURB normals,
    ray-count remainder: raytracer.py:468-469, ray_storage.py:67) and the module generator
    ``optrace.tracer.random._random`` (random.py:5) are seeded and multithreading is turned off so
    runs are bit-reproducible.
    """
    import numpy as np
    import typing_extensions

    sys.dont_write_bytecode = True
    if not hasattr(typing, "assert_never"):
        typing.assert_never = typing_extensions.assert_never
    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self

    if "traits" not in sys.modules:
        traits = types.ModuleType("traits")
        ets = types.ModuleType("traits.etsconfig")
        api = types.ModuleType("traits.etsconfig.api")

        class ETSConfig:  # noqa: D401 - attribute bag only
            toolkit = ""

        api.ETSConfig = ETSConfig
        traits.etsconfig = ets
        ets.api = api
        sys.modules["traits"] = traits
        sys.modules["traits.etsconfig"] = ets
        sys.modules["traits.etsconfig.api"] = api
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.INTER_AREA = 3

        def resize(src, dsize, interpolation=None):
            """The one call on a numeric path (convolve.py:335, PSF -> image pixel pitch): the fixtures use PSFs
            that already have the image's pitch, where OpenCV returns a copy.  Anything else is not reproduced
            here -- no stand-in arithmetic enters a golden vector."""
            if tuple(int(v) for v in dsize) != (src.shape[1], src.shape[0]):
                raise NotImplementedError("cv2.resize stand-in: equal sizes only")
            return src.copy()

        cv2.resize = resize
        sys.modules["cv2"] = cv2
    if "chardet" not in sys.modules:
        sys.modules["chardet"] = types.ModuleType("chardet")

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import optrace as ot  # noqa: E402

    ot.global_options.show_progress_bar = False
    ot.global_options.show_warnings = False
    if seed is not None:
        reseed(ot, seed)
    return ot


def reseed(ot, seed: int) -> None:
    import numpy as np
    import optrace.tracer.random as orandom

    np.random.seed(seed)
    orandom._random = np.random.Generator(np.random.SFC64(seed))
    ot.global_options.multithreading = False
