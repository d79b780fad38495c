"""Top-level `data` - the inference-path names of the reference's `data` package (reference g2vlm_utils.py:4, 15-16,
g2vlm.py:16-20), resolved to this repository's host code (g2vlm_amd/host.py, g2vlm_amd/g2vlm_utils.py):

    data.data_utils        add_special_tokens (:278-313), pil_img2rgb (:254-263)
    data.transforms        QwenVL2ImageTransform (:151-178)
    data.transforms_vggt   DinoImageNormalizeTransform (:27-45), load_images (:411-451), load_and_resize14 (:454-462),
                           load_and_resize16 (:464-471)

The datasets, augmentations and the other transform classes there belong to training (SURVEY §2 OUT OF SCOPE) and are not
provided.  Aliases only: every name is the object defined under g2vlm_amd/."""
import sys
import types

from g2vlm_amd import g2vlm_utils as _u
from g2vlm_amd import host as _h


def _module(name, doc, **names):
    m = types.ModuleType(name, doc)
    m.__dict__.update(names)
    m.__all__ = sorted(names)
    return sys.modules.setdefault(name, m)


data_utils = _module("data.data_utils", "reference data/data_utils.py (inference names)",
                     add_special_tokens=_u.add_special_tokens, pil_img2rgb=_u.pil_img2rgb)
transforms = _module("data.transforms", "reference data/transforms.py (inference names)", QwenVL2ImageTransform=_h.QwenVL2ImageTransform)
transforms_vggt = _module("data.transforms_vggt", "reference data/transforms_vggt.py (inference names)",
                          DinoImageNormalizeTransform=_h.DinoImageNormalizeTransform, load_images=_h.load_images,
                          load_and_resize14=_h.load_and_resize14, load_and_resize16=_h.load_and_resize16)
