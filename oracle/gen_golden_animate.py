"""tests/golden/animate.npz by RUNNING THE REFERENCE'S OWN AvatarAnimate code (build container only: needs /root/reference).
    python oracle/gen_golden_animate.py                                               TEST INFRASTRUCTURE ONLY.

AvatarAnimate/models/{pose_generation,motion_generation}.py import clip, smplx, neural_renderer and human_body_prior at module level (all absent
here), so the modules cannot be imported; what IS executed, unmodified, extracted with `ast` (the way gen_golden.py takes dataset.py's functions):
  * models/utils.py as a module (pure torch): axis_angle_to_matrix, matrix_to_rotation_6d, rotation_6d_to_matrix, matrix_to_quaternion,
    quaternion_to_axis_angle;
  * pose_padding; the classes SinusoidalPositionalEncoding, MotionXTransformerEncoder, MotionXTransformerDecoder;
  * VPoserRealNVP: the network-building statements of __init__ (everything but super().__init__, torch.load, load_state_dict) and the methods decode /
    encode; VPoserCodebook.suppress_duplicated_poses / get_topk_poses; MotionInterpolation.encode_poses / decode_poses / get_motion;
    MotionOptimizer: the network-building statements of __init__, decode and get_motion (clip_coef = 0: the reference's motion_ablation/baseline conf) --
    bound to a stand-in `self` that carries the attributes those methods read (device, vp = oracle/animate_standins.StandInVPoser, text features).
The fixture holds seeds, inputs and OUTPUTS; weights are re-created in the tests from the same seeds (same construction order = same draws)."""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import distributions

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from oracle.animate_standins import StandInVPoser, text_feature_of  # noqa: E402

AA = os.path.join(ref_loader.REF_ROOT, "AvatarAnimate", "models")
GOLD = os.path.join(ROOT, "tests", "golden")


def load_utils():
    spec = importlib.util.spec_from_file_location("aa_utils", os.path.join(AA, "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def namespace(U):
    ns = {"torch": torch, "nn": nn, "F": F, "np": np, "distributions": distributions, "Tensor": torch.Tensor, "tqdm": lambda it: it,
          "Optional": None, "Tuple": None, "Union": None}
    for k in ("rotation_6d_to_matrix", "matrix_to_quaternion", "quaternion_to_axis_angle", "axis_angle_to_matrix", "matrix_to_rotation_6d"):
        ns[k] = getattr(U, k)
    return ns


def extract(path, ns, classes=(), functions=(), methods=()):
    """exec the named top-level classes / functions of `path` in ns; return {(class, method): function} for `methods` = [(class, name, keep)] where keep
    filters the statements of the method body (None: all)"""
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name in classes) or (isinstance(n, ast.FunctionDef) and n.name in functions)]
    for n in ast.walk(ast.Module(body=body, type_ignores=[])):          # the annotations name typing objects we do not import
        if isinstance(n, ast.FunctionDef):
            n.returns = None
            for a in n.args.args + n.args.kwonlyargs:
                a.annotation = None
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    out = {}
    for cname, mname, keep in methods:
        cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cname][0]
        fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == mname][0]
        fn.returns = None
        for a in fn.args.args + fn.args.kwonlyargs:
            a.annotation = None
        fn.decorator_list = []
        if keep is not None:
            fn.body = [s for s in fn.body if keep(ast.unparse(s))]
        loc = {}
        exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns, loc)
        out[(cname, mname)] = loc[mname]
    return out


NET_ONLY = lambda src: not (src.startswith("super().__init__") or "torch.load" in src or "load_state_dict" in src)


class Self(nn.Module):
    """the stand-in `self` the reference's methods are bound to"""

    def __init__(self):
        super().__init__()
        self.device = "cpu"


def bind(obj, fns, cname, *names):
    for n in names:
        setattr(obj, n, types.MethodType(fns[(cname, n)], obj))


def main():
    U = load_utils()
    ns = namespace(U)
    P = os.path.join(AA, "pose_generation.py")
    M = os.path.join(AA, "motion_generation.py")
    fp = extract(P, ns, functions=("pose_padding",), methods=[("VPoserRealNVP", "__init__", NET_ONLY), ("VPoserRealNVP", "decode", None),
                                                              ("VPoserRealNVP", "encode", None), ("VPoserCodebook", "suppress_duplicated_poses", None),
                                                              ("VPoserCodebook", "get_topk_poses", None)])
    fm = extract(M, ns, classes=("SinusoidalPositionalEncoding", "MotionXTransformerEncoder", "MotionXTransformerDecoder"), functions=("pose_padding",),
                 methods=[("MotionInterpolation", "encode_poses", None), ("MotionInterpolation", "decode_poses", None), ("MotionInterpolation", "get_motion", None),
                          ("MotionOptimizer", "__init__", NET_ONLY), ("MotionOptimizer", "decode", None), ("MotionOptimizer", "get_motion", None)])
    rec = {}
    g = torch.Generator().manual_seed(0)
    # ---- rotations
    aa = torch.randn(64, 3, generator=g) * 1.2
    aa[0] = 0.0
    aa[1] = torch.tensor([1e-8, -2e-8, 1e-8])
    aa[2] = torch.tensor([3.0, 0.3, -0.2])                     # close to pi
    rm = U.axis_angle_to_matrix(aa)
    d6 = torch.randn(64, 6, generator=g)
    rm6 = U.rotation_6d_to_matrix(d6)
    rec.update(rot_aa=aa, rot_matrix=rm, rot_6d_of_matrix=U.matrix_to_rotation_6d(rm), rot_d6=d6, rot_matrix_of_6d=rm6,
               rot_aa_of_matrix=U.quaternion_to_axis_angle(U.matrix_to_quaternion(rm6)), pad_in=aa.reshape(-1)[:63][None].clone(),
               pad_out=ns["pose_padding"](aa.reshape(-1)[:63][None].clone()))
    # ---- conditional RealNVP (small sizes: dim 32 is VPoser's, hdim 48, 3 blocks)
    torch.manual_seed(11)
    nvp = Self()
    fp[("VPoserRealNVP", "__init__")](nvp, dim=32, hdim=48, num_block=3, num_sample=4, num_batch=2, ckpt_path=None)
    bind(nvp, fp, "VPoserRealNVP", "decode", "encode")
    z = torch.randn(7, 32, generator=g)
    feat = torch.nn.functional.normalize(torch.randn(7, 512, generator=g), dim=-1)
    with torch.no_grad():
        x = nvp.decode(z, feat)
        zi, ld = nvp.encode(x, feat)
    rec.update(nvp_seed=np.int64(11), nvp_z=z, nvp_feat=feat, nvp_x=x, nvp_z_back=zi, nvp_log_det=ld, nvp_mask=nvp.mask,
               nvp_keys=np.array(sorted(nvp.state_dict().keys())))
    # ---- codebook retrieval
    vp = StandInVPoser(0)
    cb = Self()
    cb.vp, cb.pre_topk, cb.topk, cb.filter_threshold = vp, 12, 5, 0.07
    cb.codebook = torch.randn(200, 32, generator=g)
    cb.codebook[37] = cb.codebook[5] + 1e-3                       # a near-duplicate that ranks right behind its twin
    cb.codebook_embedding = torch.nn.functional.normalize(torch.randn(200, 512, generator=g), dim=-1)
    tf = text_feature_of("a rendered 3d man is arguing")
    cb.codebook_embedding[5] = torch.nn.functional.normalize(tf + 0.3 * cb.codebook_embedding[5], dim=0)
    cb.codebook_embedding[37] = torch.nn.functional.normalize(tf + 0.31 * cb.codebook_embedding[37], dim=0)
    cb.get_text_feature = lambda text: text_feature_of(text)
    bind(cb, fp, "VPoserCodebook", "suppress_duplicated_poses", "get_topk_poses")
    poses = cb.get_topk_poses("a rendered 3d man is arguing")
    rec.update(cb_codebook=cb.codebook, cb_embedding=cb.codebook_embedding, cb_poses=poses)
    # ---- interpolation through the latent space
    mi = Self()
    mi.vp, mi.num_frame, mi.anchor_position = vp, 60, (0, 14, 29, 44, 59)
    bind(mi, fm, "MotionInterpolation", "encode_poses", "decode_poses", "get_motion")
    cand = ns["pose_padding"](poses)
    with torch.no_grad():
        rec.update(mi_poses=cand, mi_motion=mi.get_motion("x", cand))
    # ---- the motion VAE's decoder and the latent optimisation (small: 12 frames, width 64, 2 layers)
    torch.manual_seed(21)
    mo = Self()
    mo.num_frame = 12
    fm[("MotionOptimizer", "__init__")](mo, latent_dim=64, num_layers=2, num_heads=4, ckpt_path=None, optim_name="Adam", optim_cfg={"lr": 0.01},
                                        num_iteration=25, recon_coef=(1, 0.8, 0.6, 0.4, 0.2), clip_coef=0.0, delta_coef=0.01, clip_num_part=30)
    mo.eval()
    bind(mo, fm, "MotionOptimizer", "decode", "get_motion")
    mo.get_text_feature = lambda text: text_feature_of(text)
    lat = torch.randn(2, 64, generator=g)
    with torch.no_grad():
        rec.update(mo_seed=np.int64(21), mo_latent=lat, mo_rot6d=mo.decoder(lat), mo_decoded=mo.decode(lat[0]),
                   mo_enc=mo.encoder(torch.randn(2, 12, 55, 6, generator=torch.Generator().manual_seed(5))))
    torch.manual_seed(22)                                          # get_motion draws its initial latent with torch.randn
    motion = mo.get_motion("a rendered 3d man is arguing", cand[:, :63])
    rec.update(mo_init_seed=np.int64(22), mo_motion=motion, mo_keys=np.array(sorted(mo.state_dict().keys())))
    out = {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in rec.items()}
    np.savez_compressed(os.path.join(GOLD, "animate.npz"), **out)
    print("animate.npz", os.path.getsize(os.path.join(GOLD, "animate.npz")), {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith(("mo_m", "cb_p", "mi_m"))})


if __name__ == "__main__":
    main()
