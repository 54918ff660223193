"""Pins of the rows whose dependency the reference does not vendor (a16, f-1, f-2, f-3), against golden files written by
scripts/pin_third_party.py FROM THE THIRD-PARTY PACKAGES THEMSELVES.  Offline none of those files exists and every pin test SKIPS -- the
rows stay "parity unpinned" (DESIGN.md section 2).  Whoever holds the assets runs

    python scripts/pin_third_party.py --clip ViT-B-32.pt --bpe bpe_simple_vocab_16e6.txt.gz --smpl SMPL_NEUTRAL.pkl --neural-renderer --pymcubes
    AVC_CLIP_WEIGHTS=ViT-B-32.pt AVC_CLIP_BPE=bpe_simple_vocab_16e6.txt.gz AVC_SMPL_MODEL=SMPL_NEUTRAL.pkl python -m pytest tests/test_third_party_pins.py

and the same tests compare the CPU oracles (and, -m gpu, the HIP kernels) with what OpenAI CLIP / smplx / neural_renderer / PyMCubes computed.
The plumbing itself (file format, hashing, the refusal of stand-in files) is tested here on synthetic assets, always.

Tolerances: CLIP embeddings cosine >= 0.9999 (oracle, fp32) / >= 0.999 (HIP, bf16 MFMA operands), token ids exact; SMPL vertices 1e-5 m;
neural_renderer silhouettes IoU >= 0.999 and grey values 2e-3 inside; PyMCubes: the same surface -- vertex set equal to 1e-5 after sorting,
equal triangle count, area and enclosed volume to 1e-6 relative (the triangulation of ambiguous cells may legitimately differ by a flip; a
count mismatch is reported with both numbers).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "scripts"))
gpu = pytest.mark.gpu


def _golden(name, gold_dir=GOLD):
    path = os.path.join(gold_dir, name)
    if not os.path.exists(path):
        pytest.skip("%s not present: run scripts/pin_third_party.py where the third-party assets are (row stays unpinned)" % name)
    z = np.load(path, allow_pickle=False)
    if str(z["source"]) == "stand-in":
        pytest.skip("%s was written with --stand-in (this repo's own oracle): not a pin" % name)
    return z


def _asset(env, sha, what):
    import pin_third_party as P
    path = os.environ.get(env)
    if not path or not os.path.exists(path):
        pytest.skip("$%s not set: the %s the golden file was made from is needed to run our side" % (env, what))
    assert P.sha256_of(path) == str(sha), "$%s is not the %s the golden file was made from (SHA-256 differs)" % (env, what)
    return path


# ------------------------------------------------------------------------------------------------ the comparisons (shared by pins and plumbing test)
def check_clip_oracle(z, weights, bpe):
    import pin_third_party as P
    from oracle import clip_vit_oracle as C, clip_text_oracle as T
    from avatarclip_amd import clip_vit as V, tokenizer as TK
    sd = {k: v.float() for k, v in V.load_state_dict(weights).items()}
    imgs = P.seeded_images(int(z["n_images"]), int(z["image_seed"]))
    emb = C.encode_image(sd, imgs).detach()
    cos = torch.cosine_similarity(emb, torch.from_numpy(z["image_emb"]), dim=-1)
    assert cos.min() > 0.9999, cos
    tok = torch.from_numpy(z["tokens"])
    if bpe:
        mine = TK.tokenize([str(p) for p in z["prompts"]], TK.SimpleTokenizer(bpe))
        assert torch.equal(mine, tok), "token ids differ from %s" % str(z["tokenizer_source"])
    te = T.encode_text(sd, tok).detach()
    cos_t = torch.cosine_similarity(te, torch.from_numpy(z["text_emb"]), dim=-1)
    assert cos_t.min() > 0.9999, cos_t
    # the quantity the loss uses (main.py:514-516): cos(image embedding, text embedding), every pair
    n = lambda x: torch.nn.functional.normalize(x.float(), dim=-1)
    assert (n(emb) @ n(te).T - n(torch.from_numpy(z["image_emb"])) @ n(torch.from_numpy(z["text_emb"])).T).abs().max() < 1e-3
    return sd, imgs, tok


def check_smpl(z, model_path):
    from avatarclip_amd import smpl_lbs
    a = smpl_lbs.load_smpl_arrays(model_path)
    pose = torch.from_numpy(z["pose_axis_angle"]).float()
    assert np.abs(z["betas"]).max() == 0
    rot = smpl_lbs.batch_rodrigues(pose.reshape(-1, 3)).reshape(pose.shape[0], -1, 3, 3)
    v, _ = smpl_lbs.lbs(a["v_template"][None].expand(pose.shape[0], -1, -1), rot, a["posedirs"], a["J_regressor"], a["parents"], a["lbs_weights"])
    assert np.abs(v.numpy() - z["vertices"]).max() < 1e-5
    assert np.array_equal(np.asarray(a["faces"]).astype(np.int64), z["faces"])


def check_nr_oracle(z):
    from oracle import nr_oracle as NO
    g = np.load(os.path.join(GOLD, "smpl_views.npz"))
    for (eye, at), ref in zip(z["cameras"], z["images"]):
        img = NO.render_one_batch(g["mesh_v"], g["mesh_f"], eye, at)
        a, b = img[..., 0] > 0, ref[..., 0] > 0
        iou = (a & b).sum() / max(1, (a | b).sum())
        assert iou >= 0.999, (eye, iou)
        both = a & b
        assert np.abs(img[..., 0][both] - ref[..., 0][both]).max() < 2e-3


def _surface_stats(v, t):
    p0, p1, p2 = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    return 0.5 * np.linalg.norm(np.cross(p1 - p0, p2 - p0), axis=1).sum(), np.einsum("ij,ij->i", p0, np.cross(p1, p2)).sum() / 6.0


def check_mcubes(z, march):
    import pin_third_party as P
    u = P.mc_volume(int(z["n"]), int(z["seed"]))
    v, t = march(u, float(z["threshold"]))
    v, t = np.asarray(v, np.float64), np.asarray(t, np.int64)
    rv, rt = z["vertices"], z["triangles"]
    assert v.shape == rv.shape, ("vertex count", v.shape, rv.shape)
    key = lambda x: x[np.lexsort((x[:, 2], x[:, 1], x[:, 0]))]
    assert np.abs(key(np.round(v, 6)) - key(np.round(rv, 6))).max() < 1e-5
    assert t.shape == rt.shape, ("triangle count", t.shape, rt.shape)
    (a0, v0), (a1, v1) = _surface_stats(v, t), _surface_stats(rv, rt)
    assert abs(a0 - a1) <= 1e-6 * abs(a1) and abs(v0 - v1) <= 1e-6 * max(abs(v1), 1e-9)


# ------------------------------------------------------------------------------------------------ pins (skip without the golden files)
def test_clip_towers_and_tokenizer_against_openai():
    z = _golden("third_party_clip.npz")
    check_clip_oracle(z, _asset("AVC_CLIP_WEIGHTS", z["weights_sha256"], "CLIP checkpoint"), os.environ.get("AVC_CLIP_BPE"))


@gpu
def test_hip_clip_towers_against_openai():
    z = _golden("third_party_clip.npz")
    weights = _asset("AVC_CLIP_WEIGHTS", z["weights_sha256"], "CLIP checkpoint")
    sd, imgs, tok = check_clip_oracle(z, weights, os.environ.get("AVC_CLIP_BPE"))
    from avatarclip_amd import clip_vit as V
    model = V.ClipVisionB32(sd, torch.device("cuda"))
    for lo in range(0, imgs.shape[0], 2):            # B = 2: the per-iteration path; all at once: the batched scoring path
        e = model.encode_image(imgs[lo:lo + 2].cuda()).float().cpu()
        assert torch.cosine_similarity(e, torch.from_numpy(z["image_emb"][lo:lo + 2]), dim=-1).min() > 0.999
    with torch.no_grad():
        e = model.encode_image(imgs.cuda()).float().cpu()
    assert torch.cosine_similarity(e, torch.from_numpy(z["image_emb"]), dim=-1).min() > 0.999
    te = model.encode_text(tok).float().cpu()
    assert torch.cosine_similarity(te, torch.from_numpy(z["text_emb"]), dim=-1).min() > 0.999


def test_smpl_skinning_against_smplx():
    z = _golden("third_party_smpl.npz")
    check_smpl(z, _asset("AVC_SMPL_MODEL", z["model_sha256"], "SMPL model file"))


def test_rasteriser_restatement_against_neural_renderer():
    check_nr_oracle(_golden("third_party_nr.npz"))


@gpu
def test_hip_rasteriser_against_neural_renderer():
    z = _golden("third_party_nr.npz")
    from avatarclip_amd.smpl_prior import MeshPrior
    g = np.load(os.path.join(GOLD, "smpl_views.npz"))
    prior = MeshPrior(g["mesh_v"], g["mesh_f"], device=torch.device("cuda"))
    for (eye, at), ref in zip(z["cameras"], z["images"]):
        img = prior(np.asarray(eye, np.float32), np.asarray(at, np.float32)).cpu().numpy()
        a, b = img[..., 0] > 0, ref[..., 0] > 0
        assert (a & b).sum() / max(1, (a | b).sum()) >= 0.999
        assert np.abs(img[..., 0][a & b] - ref[..., 0][a & b]).max() < 2e-3


def test_marching_cubes_restatement_against_pymcubes():
    from oracle import mcubes_oracle as MO
    check_mcubes(_golden("third_party_mcubes.npz"), MO.marching_cubes)


@gpu
def test_hip_marching_cubes_against_pymcubes():
    from avatarclip_amd import mesh
    z = _golden("third_party_mcubes.npz")

    def march(u, iso):
        v, t = mesh.marching_cubes(torch.from_numpy(u).cuda(), iso)
        return v.cpu().numpy(), t.cpu().numpy()
    check_mcubes(z, march)


# ------------------------------------------------------------------------------------------------ the plumbing, on synthetic assets (always runs)
def test_pin_script_plumbing_on_synthetic_assets(tmp_path, monkeypatch):
    """scripts/pin_third_party.py --stand-in on a seeded CLIP state dict and a synthetic skeleton: every section writes its file, the
    comparison code above accepts what the oracles themselves produced, the asset hash is enforced, stand-in files are refused as pins,
    and a section whose package is missing is reported through the exit status instead of being written from a restatement."""
    import pin_third_party as P
    from oracle import clip_vit_oracle as C, clip_text_oracle as T
    sd = dict(C.random_state_dict(0))
    sd.update(T.random_state_dict(0))
    wpath = str(tmp_path / "clip_sd.pt")
    torch.save({k: v.clone() for k, v in sd.items()}, wpath)
    import gzip
    from tests.test_clip_text import _synthetic_merges
    bpe = str(tmp_path / "bpe.txt.gz")
    with gzip.open(bpe, "wt", encoding="utf-8") as fp:
        fp.write("#version: test\n" + "\n".join(" ".join(m) for m in _synthetic_merges(" ".join(P.PROMPTS).lower(), 50)) + "\n")
    rng = np.random.RandomState(0)
    V_, J = 60, 24
    parents = np.concatenate([[-1], rng.randint(0, np.arange(1, J))]).astype(np.int64)
    jreg = np.abs(rng.rand(J, V_)); jreg /= jreg.sum(1, keepdims=True)
    w = np.abs(rng.rand(V_, J)) ** 3; w /= w.sum(1, keepdims=True)
    spath = str(tmp_path / "smpl_synth.npz")
    np.savez(spath, v_template=rng.randn(V_, 3).astype(np.float32) * 0.3, posedirs=(rng.randn((J - 1) * 9, V_ * 3) * 0.01).astype(np.float32),
             J_regressor=jreg.astype(np.float32), parents=parents, lbs_weights=w.astype(np.float32), faces=rng.randint(0, V_, (40, 3)).astype(np.int32))
    out = str(tmp_path / "gold")
    rc = P.main(["--clip", wpath, "--bpe", bpe, "--smpl", spath, "--neural-renderer", "--pymcubes", "--out", out, "--stand-in"])
    assert rc == 0 and sorted(os.listdir(out)) == ["third_party_clip.npz", "third_party_mcubes.npz", "third_party_nr.npz", "third_party_smpl.npz"]
    zc = np.load(os.path.join(out, "third_party_clip.npz"))
    assert str(zc["source"]) == "stand-in" and zc["image_emb"].shape == (4, 512) and zc["tokens"].shape == (len(P.PROMPTS), 77)
    check_clip_oracle(zc, wpath, bpe)
    check_smpl(np.load(os.path.join(out, "third_party_smpl.npz")), spath)
    from oracle import mcubes_oracle as MO
    check_mcubes(np.load(os.path.join(out, "third_party_mcubes.npz")), MO.marching_cubes)
    zn = np.load(os.path.join(out, "third_party_nr.npz"))
    assert zn["images"].shape == (5, 256, 256, 3) and (zn["images"][0, ..., 0] > 0).mean() > 0.02
    # stand-in files are not pins; a wrong asset is refused by its hash
    with pytest.raises(pytest.skip.Exception):
        _golden("third_party_clip.npz", out)
    other = str(tmp_path / "other.pt")
    torch.save({k: v + 1 for k, v in sd.items()}, other)
    monkeypatch.setenv("AVC_CLIP_WEIGHTS", other)
    with pytest.raises(AssertionError, match="SHA-256"):
        _asset("AVC_CLIP_WEIGHTS", zc["weights_sha256"], "CLIP checkpoint")
    monkeypatch.setenv("AVC_CLIP_WEIGHTS", wpath)
    assert _asset("AVC_CLIP_WEIGHTS", zc["weights_sha256"], "CLIP checkpoint") == wpath
    # without --stand-in the third-party packages are required: absent here, so nothing is written and the exit status says so
    out2 = str(tmp_path / "gold2")
    missing = [m for m in ("smplx", "neural_renderer", "mcubes") if isinstance(P._try_import(m), Exception)]
    rc2 = P.main(["--smpl", spath, "--neural-renderer", "--pymcubes", "--out", out2])
    assert rc2 == len(missing) and len(os.listdir(out2)) == 3 - len(missing)
