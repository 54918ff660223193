"""ShapeGen's LinearVAE / codebook / prompt helpers (AvatarGen/ShapeGen/main.py:22-91, utils.py:35-57) against tests/golden/shapegen.npz,
which oracle/gen_golden_shapegen.py produced by running the reference's OWN class (extracted with ast) under a fixed seed; plus the GPU
run of BASELINE config 5's chain ShapeGen -> dataset writer -> Runner.train -> Runner.train_clip on synthetic VAE weights."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shapegen.npz")
gpu = pytest.mark.gpu


def test_linear_vae_matches_the_reference_class():
    from avatarclip_amd import shapegen as SG
    z = np.load(GOLD)
    torch.manual_seed(0)          # the reference class was built under this seed: same construction order, same draws
    model = SG.LinearVAE(6890 * 3, 16, torch.from_numpy(z["v_template"])).eval().requires_grad_(False)
    assert sorted(model.state_dict().keys()) == sorted(p + s for p in ("enc1.", "enc2.", "dec1.", "dec2.") for s in ("weight", "bias"))
    assert np.array_equal(model.dec2.bias.numpy()[:64], z["dec2_bias_head"])
    with torch.no_grad():
        dec = model.decode(torch.from_numpy(z["latents"]))
        torch.manual_seed(5)
        out, mu, log_var = model(torch.from_numpy(z["fwd_in"]))
    assert dec.shape == (3, 6890, 3)
    assert np.abs(dec.numpy() - z["decoded"]).max() < 1e-5
    assert np.abs(mu.numpy() - z["fwd_mu"]).max() < 1e-5 and np.abs(log_var.numpy() - z["fwd_log_var"]).max() < 1e-5
    assert np.abs(out.numpy()[:, :256] - z["fwd_out"]).max() < 1e-4
    for p, (txt, w, stop) in zip(z["prompts"], z["parsed"]):
        got = SG.parse_prompt(str(p))
        assert (got[0], repr(got[1]), repr(got[2])) == (str(txt), str(w), str(stop))


def test_codebook_file_with_a_tensor_key_and_obj_round_trip(tmp_path):
    from avatarclip_amd import shapegen as SG
    from avatarclip_amd.smpl_prior import read_obj
    codes, emb = torch.randn(5, 16), torch.randn(5, 512)
    torch.save({codes: emb}, tmp_path / "codebook.pth")                 # the reference's layout (main.py:86-91)
    c, e = SG.load_codebook(str(tmp_path / "codebook.pth"), device="cpu")
    assert torch.equal(c, codes) and torch.equal(e, emb)
    torch.save({"codebook": codes, "clip_codebook": emb}, tmp_path / "cb2.pth")
    c, e = SG.load_codebook(str(tmp_path / "cb2.pth"), device="cpu")
    assert torch.equal(c, codes) and torch.equal(e, emb)
    V = np.random.RandomState(0).randn(7, 3).astype(np.float32)
    F = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6]], np.int32)
    SG.writeOBJ(str(tmp_path / "m.obj"), V, F)
    v2, f2 = read_obj(str(tmp_path / "m.obj"))
    assert np.allclose(v2, V, atol=1e-6) and np.array_equal(f2, F)
    small = SG.LinearVAE(30, 4, torch.zeros(10, 3))
    torch.save(small.state_dict(), tmp_path / "vae.pth")
    again = SG.create_load_AE(30, 4, np.zeros((10, 3), np.float32), str(tmp_path / "vae.pth"), device="cpu")
    lat = torch.randn(2, 4)
    assert torch.equal(again.decode(lat), small.decode(lat)) and not any(p.requires_grad for p in again.parameters())


@gpu
def test_config5_chain_shapegen_to_train_clip(tmp_path):
    """scripts/pipeline_config5.py: synthetic LinearVAE weights + a codebook whose CLIP embeddings are made by the pipeline itself
    -> codebook search finds the planted code -> .obj -> 108-view dataset -> Runner.train -> checkpoint -> Runner.train_clip"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pipeline_config5", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                   "scripts", "pipeline_config5.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run(str(tmp_path), n_codes=6, planted=4, train_iters=25, clip_iters=4, res=64, small=True)
    assert res["best"] == 4
    assert os.path.exists(res["obj"]) and len(res["frames"]) == 108
    assert res["train_losses"][-1] < res["train_losses"][0]
    assert all(np.isfinite(res["clip_losses"]))
    assert res["pretrain_loaded"]
