"""AvatarAnimate's generators (avatarclip_amd/animate.py) against tests/golden/animate.npz, which oracle/gen_golden_animate.py
produced by RUNNING THE REFERENCE'S OWN classes and methods (extracted with `ast`: the modules import clip / smplx / neural_renderer / human_body_prior
at the top and cannot be imported here): rotation conversions, pose padding, the conditional RealNVP (network construction under a seed, decode, encode,
log-determinant), the codebook retrieval with duplicate suppression, the latent interpolation, the motion VAE's encoder / decoder and 25 Adam steps of the
latent optimisation (clip_coef = 0).  Third-party blobs are seeded stand-ins on both sides (oracle/animate_standins.py): what is pinned is the ARITHMETIC
around them; parity with the real VPoser / RealNVP / motion-VAE weights stays unpinned.
GPU: the CLIP-scored paths (rendering through the HIP rasteriser, embeddings through the HIP ViT) and the conf-driven run, on stand-ins."""
import os

import numpy as np
import pytest
import torch

from oracle.animate_standins import StandInVPoser, text_feature_of

gpu = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "animate.npz")


def _gold():
    z = np.load(GOLD)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind == "f" else z[k]) for k in z.files}


def _ctx(device="cpu", **kw):
    from avatarclip_amd import animate as A
    return A.AnimateContext(kw.pop("perceptor", None), text_feature_of, kw.pop("smpl", None), StandInVPoser(0).to(device), device=device, **kw)


def test_rotation_conversions_and_padding_match_the_reference_functions():
    from avatarclip_amd import animate as A
    g = _gold()
    assert torch.allclose(A.axis_angle_to_matrix(g["rot_aa"]), g["rot_matrix"], atol=1e-6)
    assert torch.equal(A.matrix_to_rotation_6d(g["rot_matrix"]), g["rot_6d_of_matrix"])
    assert torch.allclose(A.rotation_6d_to_matrix(g["rot_d6"]), g["rot_matrix_of_6d"], atol=1e-6)
    assert torch.allclose(A.matrix_to_axis_angle(g["rot_matrix_of_6d"]), g["rot_aa_of_matrix"], atol=2e-5)
    assert torch.equal(A.pose_padding(g["pad_in"]), g["pad_out"]) and A.pose_padding(g["pad_out"]) is g["pad_out"]
    with pytest.raises(ValueError):
        A.pose_padding(torch.zeros(2, 60))
    # and they are rotations: R R^T = 1, det = +1, axis-angle round trip
    R = A.axis_angle_to_matrix(g["rot_aa"])
    assert torch.allclose(R @ R.transpose(-1, -2), torch.eye(3).expand_as(R), atol=1e-5) and torch.allclose(torch.linalg.det(R), torch.ones(64), atol=1e-5)
    back = A.matrix_to_axis_angle(R[3:])
    assert torch.allclose(A.axis_angle_to_matrix(back), R[3:], atol=1e-5)


def test_conditional_realnvp_matches_the_reference_class():
    from avatarclip_amd import animate as A
    g = _gold()
    torch.manual_seed(int(g["nvp_seed"]))
    nvp = A.VPoserRealNVP(_ctx(), dim=32, hdim=48, num_block=3, num_sample=4, num_batch=2)
    assert sorted(nvp.state_dict().keys()) == list(g["nvp_keys"])                     # the reference's parameter names: its checkpoint loads unmodified
    assert torch.equal(nvp.mask, g["nvp_mask"])                                        # same draws in the same order
    with torch.no_grad():
        x = nvp.decode(g["nvp_z"], g["nvp_feat"])
        z, ld = nvp.encode(x, g["nvp_feat"])
    assert torch.allclose(x, g["nvp_x"], atol=1e-5) and torch.allclose(z, g["nvp_z_back"], atol=1e-5) and torch.allclose(ld, g["nvp_log_det"], atol=1e-5)
    assert torch.allclose(z, g["nvp_z"], atol=1e-4)                                    # the flow inverts
    torch.manual_seed(3)
    s = nvp.sample(6, g["nvp_feat"][0])
    assert s.shape == (6, 32) and torch.isfinite(s).all()


def test_codebook_retrieval_and_duplicate_suppression_match_the_reference_methods():
    from avatarclip_amd import animate as A
    g = _gold()
    gen = A.VPoserCodebook(_ctx(), codebook=g["cb_codebook"], codebook_embedding=g["cb_embedding"], topk=5, pre_topk=12, filter_threshold=0.07)
    poses = gen.get_topk_poses("a rendered 3d man is arguing")
    assert poses.shape == (5, 63) and torch.allclose(poses, g["cb_poses"], atol=1e-6)
    # the planted near-duplicate (entry 37 = entry 5 + 1e-3, ranked right behind it) is gone; without the filter it would be the second pose
    raw = gen.ctx.vp.decode(g["cb_codebook"][[5, 37]])["pose_body"].reshape(2, -1)
    assert torch.allclose(poses[0], raw[0], atol=1e-6) and (poses[1:] - raw[1]).abs().mean(-1).min() > 0.07
    kept = A.VPoserCodebook.suppress_duplicated_poses(torch.tensor([[0.0] * 4, [0.05] * 4, [0.2] * 4, [0.26] * 4, [0.5] * 4]), 0.07)
    assert kept[:, 0].tolist() == pytest.approx([0.0, 0.2, 0.5])


def test_latent_interpolation_matches_the_reference_method():
    from avatarclip_amd import animate as A
    g = _gold()
    motion = A.MotionInterpolation(_ctx()).get_motion("x", g["mi_poses"])
    assert motion.shape == (60, 69) and torch.allclose(motion, g["mi_motion"], atol=1e-6)
    assert torch.equal(motion[:, 63:], torch.zeros(60, 6))
    with pytest.raises(ValueError):
        A.MotionInterpolation(_ctx(), num_frame=60, anchor_position=(0, 10, 58))
    with pytest.raises(TypeError):          # a mistyped conf key is an error, as with the reference's explicit signatures
        A.build_motion_generator({"type": "MotionInterpolation", "anchor_positon": (0, 59)}, _ctx())
    assert A.build_motion_generator({"type": "MotionInterpolation", "smpl_path": "../smpl_models"}, _ctx()).num_frame == 60


def test_motion_vae_and_latent_optimisation_match_the_reference_class():
    from avatarclip_amd import animate as A
    g = _gold()
    torch.manual_seed(int(g["mo_seed"]))
    mo = A.MotionOptimizer(_ctx(), num_frame=12, latent_dim=64, num_layers=2, num_heads=4, num_iteration=25, clip_coef=0.0, delta_coef=0.01)
    assert sorted(mo.state_dict().keys()) == list(g["mo_keys"])                        # data/motion_vae.pth's keys
    with torch.no_grad():
        assert torch.allclose(mo.decoder(g["mo_latent"]), g["mo_rot6d"], atol=1e-5)
        assert torch.allclose(mo.decode(g["mo_latent"][0]), g["mo_decoded"], atol=2e-5)
        enc_in = torch.randn(2, 12, 55, 6, generator=torch.Generator().manual_seed(5))
        assert torch.allclose(mo.encoder(enc_in), g["mo_enc"], atol=1e-5)
    torch.manual_seed(int(g["mo_init_seed"]))
    motion = mo.get_motion("a rendered 3d man is arguing", g["mi_poses"])
    assert motion.shape == (12, 69) and torch.allclose(motion, g["mo_motion"], atol=2e-4), (motion - g["mo_motion"]).abs().max()
    # the losses do what the docstring says: the reconstruction term is zero for a motion that passes through every candidate
    recon, delta = mo.losses(g["mi_poses"][:, :63].repeat(3, 1)[:12], g["mi_poses"][:, :63])
    assert float(recon) < 1e-10 and float(delta) > 0


def test_generators_that_need_the_rasterisers_gradient_say_so():
    from avatarclip_amd import animate as A
    for make in (lambda: A.PoseOptimizer(_ctx()), lambda: A.VPoserOptimizer(_ctx()), lambda: A.MotionOptimizer(_ctx(), clip_coef=0.001),
                 lambda: A.build_pose_generator({"type": "PoseOptimizer"}, _ctx()), lambda: A.build_motion_generator({"type": "MotionOptimizer"}, _ctx())):
        with pytest.raises(NotImplementedError, match="neural_renderer's backward"):
            make()


def _synthetic_smpl(dev):
    """SMPL-shaped arrays over the real template mesh of tests/golden/smpl_views.npz: 24 pseudo-joints on the body's height axis, smooth skinning
    weights by distance, no pose blend shapes (the licensed model is an input; this one only has its interface and a plausible articulation)"""
    z = np.load(os.path.join(os.path.dirname(GOLD), "smpl_views.npz"))
    v, f = z["mesh_v"].astype(np.float32), z["mesh_f"].astype(np.int32)
    rs = np.random.RandomState(0)
    centres = v[rs.choice(len(v), 24, replace=False)]
    d = np.linalg.norm(v[:, None] - centres[None], axis=-1)
    w = np.exp(-(d / 0.15) ** 2) + 1e-6
    w /= w.sum(1, keepdims=True)
    jreg = np.exp(-(d.T / 0.05) ** 2) + 1e-9
    jreg /= jreg.sum(1, keepdims=True)
    parents = np.array([-1] + [int(rs.randint(0, i)) for i in range(1, 24)], np.int64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return dict(v_template=t(v), posedirs=torch.zeros(23 * 9, len(v) * 3, device=dev), J_regressor=t(jreg.astype(np.float32)), parents=torch.from_numpy(parents),
                lbs_weights=t(w.astype(np.float32)), faces=f)


CONF = """
general {{ base_exp_dir = {out}
          mode = {mode}
          text = a rendered 3d man is arguing }}
pose_generator {{ type = {pose} }}
motion_generator {{ type = {motion}
{extra} }}
"""


@gpu
def test_clip_scored_generators_and_the_conf_driven_run_on_stand_in_assets(tmp_path):
    """pose features through the HIP rasteriser + HIP ViT (5 cameras, numpy's elevation draws in the reference's order), RealNVP's sample-score-keep loop,
    score-sorted candidates, and main.py's flow on the reference's conf layouts (motion_ablation/interpolation, motion_ablation/baseline, pose_ablation/*)."""
    from avatarclip_amd import animate as A
    from avatarclip_amd import clip_vit as V
    from avatarclip_amd.conf import ConfigFactory
    from oracle import clip_vit_oracle as C
    dev = torch.device("cuda")
    ctx = A.AnimateContext(V.ClipVisionB32(C.random_state_dict(0), dev), text_feature_of, _synthetic_smpl(dev), StandInVPoser(0).to(dev), device=dev)
    g = _gold()
    poses = g["mi_poses"][:3].to(dev)
    np.random.seed(4)
    f1 = ctx.get_pose_feature(poses)
    np.random.seed(4)
    f2 = ctx.get_pose_feature(poses)
    assert f1.shape == (3, 512) and torch.isfinite(f1).all() and torch.equal(f1, f2)              # deterministic given numpy's generator
    assert (f1[0] - f1[1]).abs().max() > 1e-4                                                      # different poses, different embeddings
    np.random.seed(4)
    single = ctx.get_pose_feature(poses[1], angles=(150,))
    assert single.shape == (1, 512)
    s = ctx.calculate_pose_score("a rendered 3d man is arguing", poses[0])
    assert -1.0 <= s <= 1.0
    ranked = ctx.sort_poses_by_score("a rendered 3d man is arguing", [p for p in poses])
    np.random.seed(9)
    scores = [ctx.calculate_pose_score("a rendered 3d man is arguing", p) for p in ranked]
    assert len(ranked) == 3
    # RealNVP: two rounds of four samples, best kept
    torch.manual_seed(1)
    nvp = A.VPoserRealNVP(ctx, hdim=48, num_block=3, num_sample=4, num_batch=2, topk=2)
    best = nvp.get_topk_poses("a rendered 3d man is arguing")
    assert best.shape == (2, 63) and torch.isfinite(best).all()
    # the conf-driven run (main.py): codebook -> interpolation; codebook -> motion VAE without the CLIP term; pose mode; a generator that cannot run
    assets = dict(codebook=g["cb_codebook"], codebook_embedding=g["cb_embedding"])
    conf = ConfigFactory.parse_string(CONF.format(out=str(tmp_path / "interp"), mode="motion", pose="VPoserCodebook", motion="MotionInterpolation", extra=""))
    poses_i, motion_i = A.run(conf, ctx, pose_assets=assets)
    assert poses_i.shape == (5, 63) and motion_i.shape == (60, 69) and torch.allclose(poses_i.cpu(), g["cb_poses"], atol=1e-5)
    assert sorted(os.listdir(str(tmp_path / "interp"))) == ["candidate_%d.npy" % i for i in range(5)] + ["motion.npy"]
    assert np.allclose(np.load(str(tmp_path / "interp" / "motion.npy")), g["mi_motion"].numpy(), atol=1e-5)
    conf = ConfigFactory.parse_string(CONF.format(out=str(tmp_path / "base"), mode="motion", pose="VPoserCodebook", motion="MotionOptimizer",
                                                  extra="    recon_coef = [1.0, 1.0, 1.0, 1.0, 1.0]\n    clip_coef = 0.0\n    delta_coef = 0.0\n    num_iteration = 10\n    latent_dim = 64\n    num_layers = 2"))
    _, motion_b = A.run(conf, ctx, pose_assets=assets)
    assert motion_b.shape == (60, 69) and torch.isfinite(motion_b).all() and motion_b.is_cuda
    conf = ConfigFactory.parse_string(CONF.format(out=str(tmp_path / "pose"), mode="pose", pose="VPoserCodebook", motion="MotionOptimizer", extra=""))
    assert A.run(conf, ctx, pose_assets=assets)[1] is None
    conf = ConfigFactory.parse_string(CONF.format(out=str(tmp_path / "opt"), mode="pose", pose="VPoserOptimizer", motion="MotionOptimizer", extra=""))
    with pytest.raises(NotImplementedError):
        A.run(conf, ctx)
