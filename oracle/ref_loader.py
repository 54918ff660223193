"""Load the *reference* NeuS modules from /root/reference (only exists in the build container).

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py to generate the committed fixtures under
tests/golden/ and by tests that pin oracle/neus_oracle.py against the reference when it is present.
Nothing on the product path (avatarclip_amd/) may import this file.

Recipe follows SURVEY.md Appendix C: stub `mcubes` / `icecream` (imported at
AvatarGen/AppearanceGen/models/renderer.py:6-7 but unused on the hot path).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("AVATARCLIP_REFERENCE", "/root/reference")
REF_AG = os.path.join(REF_ROOT, "AvatarGen", "AppearanceGen")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_AG, "models"))


def load_reference():
    """Returns a namespace with SDFNetwork, RenderingNetwork, SingleVarianceNetwork, NeuSRenderer,
    sample_pdf, get_embedder -- the unmodified reference classes."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_AG)
    for m in ("mcubes", "icecream"):
        if m not in sys.modules:
            mod = types.ModuleType(m)
            mod.ic = lambda *a, **k: None
            sys.modules[m] = mod
    # the reference uses a top-level package called `models`; import it under an isolated sys.path
    saved = list(sys.path)
    saved_models = sys.modules.pop("models", None)
    sys.path.insert(0, REF_AG)
    try:
        import importlib
        fields = importlib.import_module("models.fields")
        renderer = importlib.import_module("models.renderer")
        embedder = importlib.import_module("models.embedder")
    finally:
        sys.path[:] = saved
    ns = types.SimpleNamespace(
        SDFNetwork=fields.SDFNetwork,
        RenderingNetwork=fields.RenderingNetwork,
        SingleVarianceNetwork=fields.SingleVarianceNetwork,
        NeuSRenderer=renderer.NeuSRenderer,
        sample_pdf=renderer.sample_pdf,
        get_embedder=embedder.get_embedder,
        small_ckpt=os.path.join(REF_AG, "pretrained_models", "zero_beta_stand_pose_small.pth"),
        ref_ag=REF_AG,
    )
    return ns


SMALL_SDF = dict(d_out=129, d_in=3, d_hidden=128, n_layers=3, skip_in=[3], multires=6, bias=0.5,
                 scale=1.0, geometric_init=True, weight_norm=True)
SMALL_COLOR = dict(d_feature=128, mode="no_view_dir", d_in=6, d_out=3, d_hidden=128, n_layers=1,
                   weight_norm=True, multires_view=0, squeeze_out=True, extra_color=True)
FULL_SDF = dict(d_out=257, d_in=3, d_hidden=256, n_layers=4, skip_in=[4], multires=6, bias=0.5,
                scale=1.0, geometric_init=True, weight_norm=True)
FULL_COLOR = dict(d_feature=256, mode="no_view_dir", d_in=6, d_out=3, d_hidden=256, n_layers=2,
                  weight_norm=True, multires_view=0, squeeze_out=True, extra_color=True)
