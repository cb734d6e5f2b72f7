#!/opt/conda/bin/python3.9
"""Golden vectors for the intensity augmentations whose arithmetic lives in scikit-image / scikit-learn (reference
fetal_net/augment.py:87-128): the REFERENCE's own functions, imported from /root/reference and run over the real libraries.

Runs ONLY in the build container, under the stray conda interpreter (scikit-image 0.18.3, scikit-learn 0.24.2; the main interpreter has
neither):
    /opt/conda/bin/python3.9 tests/golden/make_skimage_fixture.py          ->  tests/golden/skimage_golden.npz
Per case: the input volume, the arguments, the numpy seed set right before the call, and the reference's output.
    contrast_augment    = skimage.exposure.rescale_intensity(in_range=(lo, hi), out_range='image')
    add_gaussian_noise  = MinMaxScaler((0,1)) -> skimage.util.random_noise('gaussian', clip=True, var=sigma^2) -> inverse_transform
    add_speckle_noise   = ... 'speckle' ...
    shot_noise          = MinMaxScaler -> floor(x * 1023) / 1023 -> random_noise('poisson', clip=True) -> inverse_transform
    apply_gaussian_filter = skimage.filters.gaussian(data, sigma)
"""
import importlib.util
import os
import sys
import types
import warnings

warnings.simplefilter("ignore")
import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference_augment():
    import sklearn.preprocessing._data as skdata
    import sklearn.utils.validation as skval
    real_check_array = skval.check_array

    def check_array(*args, **kw):
        kw.pop("warn_on_dtype", None)       # removed from scikit-learn in 0.23; it only ever controlled a warning
        return real_check_array(*args, **kw)

    skval.check_array = check_array
    _mod("sklearn.preprocessing.data", _handle_zeros_in_scale=skdata._handle_zeros_in_scale)       # pre-0.22 module path the reference imports
    _mod("nibabel", Nifti1Image=object)
    _mod("nilearn")
    _mod("nilearn.image", reorder_img=None, new_img_like=None, resample_to_img=None)
    _mod("imgaug", augmenters=None)
    _mod("imgaug.augmenters")
    pkg = _mod("fetal_net")
    pkg.__path__ = [os.path.join(REF, "fetal_net")]
    up = _mod("fetal_net.utils")
    up.__path__ = [os.path.join(REF, "fetal_net/utils")]
    _mod("fetal_net.utils.nilearn_custom_utils")
    _mod("fetal_net.utils.nilearn_custom_utils.nilearn_utils", crop_img_to=None)
    _mod("fetal_net.utils.sitk_utils", resample_to_spacing=None, calculate_origin_offset=None)
    for name, rel in (("fetal_net.utils.utils", "fetal_net/utils/utils.py"), ("fetal_net.augment", "fetal_net/augment.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
    return sys.modules["fetal_net.augment"]


def main():
    A = load_reference_augment()
    import skimage
    import sklearn
    out = {"versions": np.array(["skimage " + skimage.__version__, "sklearn " + sklearn.__version__, "numpy " + np.__version__])}
    rs = np.random.RandomState(11)
    vols = {"mri": rs.randn(8, 7, 6) * 120 + 400, "neg": -np.abs(rs.randn(5, 6, 4)) * 3 - 1, "unit": rs.rand(6, 6, 6),
            "flat": np.full((4, 4, 4), 2.5)}
    for k, v in vols.items():
        out["in_" + k] = v
    n = 0
    for vk, (lo, hi) in [("mri", (250.0, 600.0)), ("mri", (-50.0, 2000.0)), ("neg", (-6.0, -2.0)), ("unit", (0.2, 0.2001)), ("unit", (0.3, 0.9))]:
        out["contrast_%d" % n] = A.contrast_augment(vols[vk].copy(), lo, hi)
        out["contrast_%d_args" % n] = np.array([lo, hi])
        out["contrast_%d_in" % n] = np.array(vk)
        n += 1
    n = 0
    for vk, sigma, seed in [("mri", 0.05, 3), ("neg", 0.2, 4), ("unit", 0.01, 5), ("mri", 0.5, 6)]:
        for kind, fn in (("gaussian", A.add_gaussian_noise), ("speckle", A.add_speckle_noise)):
            np.random.seed(seed)
            out["%s_%d" % (kind, n)] = fn(vols[vk].copy(), sigma)
            out["%s_%d_args" % (kind, n)] = np.array([sigma, seed])
            out["%s_%d_in" % (kind, n)] = np.array(vk)
        n += 1
    n = 0
    for vk, seed in [("mri", 7), ("unit", 8), ("neg", 9)]:
        np.random.seed(seed)
        out["shot_%d" % n] = A.shot_noise(vols[vk].copy())
        out["shot_%d_args" % n] = np.array([seed])
        out["shot_%d_in" % n] = np.array(vk)
        n += 1
    n = 0
    for vk, sigma in [("mri", 0.7), ("mri", 2.0), ("unit", 0.3), ("neg", 1.3), ("flat", 1.0)]:
        out["gfilter_%d" % n] = A.apply_gaussian_filter(vols[vk].copy(), sigma)
        out["gfilter_%d_args" % n] = np.array([sigma])
        out["gfilter_%d_in" % n] = np.array(vk)
        n += 1
    np.savez_compressed(os.path.join(HERE, "skimage_golden.npz"), **out)
    print("skimage_golden.npz:", len(out), "arrays;", list(out["versions"]))


if __name__ == "__main__":
    main()
