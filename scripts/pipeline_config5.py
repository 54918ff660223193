"""BASELINE config 5's AvatarGen chain in one scripted run (development aid + GPU test driver):
    ShapeGen (LinearVAE decode + CLIP codebook search, ShapeGen/main.py:93-123) -> .obj -> ShapeGen/render.py's 108-view set on the HIP
    rasteriser -> AppearanceGen Runner.train (NeuS init, main.py:180-256) -> checkpoint -> Runner.train_clip from that pretrain
    (main.py:337-566).
The licensed / downloaded inputs (VAE + codebook blobs, SMPL template, CLIP weights) are replaced by seeded stand-ins: LinearVAE
weights scaled so that decoded bodies stay bodies, the template mesh of tests/golden/smpl_views.npz, seeded CLIP weights, a codebook
whose CLIP embeddings the script computes itself, and text embeddings planted so that code `planted` is the answer.
    python scripts/pipeline_config5.py [workdir]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(workdir, n_codes=8, planted=3, train_iters=200, clip_iters=20, res=64, small=True, seed=0):
    import bench
    from avatarclip_amd import clip_score, clip_vit, shapegen as SG, shapegen_render as SR
    from avatarclip_amd.runner import Runner, clip_vit_random_state_dict
    from avatarclip_amd.smpl_prior import ROT_MAT
    dev = torch.device("cuda")
    z = np.load(os.path.join(ROOT, "tests", "golden", "smpl_views.npz"))
    v_template, faces = z["mesh_v"].astype(np.float32), z["mesh_f"]
    # ---- ShapeGen: VAE + codebook files as the reference lays them out
    torch.manual_seed(seed)
    vae = SG.LinearVAE(v_template.size, 16, torch.from_numpy(v_template))
    with torch.no_grad():
        vae.dec2.weight.mul_(0.02)            # offsets of a few centimetres
        vae.dec2.bias.zero_()
    os.makedirs(workdir, exist_ok=True)
    torch.save(vae.state_dict(), os.path.join(workdir, "model_VAE_16.pth"))
    model_AE = SG.create_load_AE(v_template.size, 16, v_template, os.path.join(workdir, "model_VAE_16.pth"), dev)
    perceptor = clip_vit.ClipVisionB32(clip_vit_random_state_dict(seed), dev)
    codes = torch.randn(n_codes, 16, generator=torch.Generator().manual_seed(seed + 1)).to(dev)
    with torch.no_grad():
        emb = torch.cat([clip_score.render_embedding(perceptor, SG.render_body(model_AE.decode(c[None])[0].cpu().numpy(), faces, device=dev)).mean(0, keepdim=True)
                         for c in codes])
        zero_emb = clip_score.render_embedding(perceptor, SG.render_body(model_AE.decode(torch.zeros(1, 16, device=dev))[0].cpu().numpy(), faces, device=dev)).mean(0)
    torch.save({codes.cpu(): emb.cpu()}, os.path.join(workdir, "codebook.pth"))
    codebook, clip_codebook = SG.load_codebook(os.path.join(workdir, "codebook.pth"), dev)
    n_txt = torch.randn(1, 512, generator=torch.Generator().manual_seed(seed + 2)).to(dev)
    t_txt = n_txt + (emb[planted] - zero_emb)[None]        # "target - neutral" points from the neutral body to the planted code
    v, v0, best, cos = SG.shape_gen(model_AE, faces, perceptor, codebook, clip_codebook, n_txt, t_txt)
    obj = os.path.join(workdir, "coarse_shape.obj")
    SG.writeOBJ(obj, v, faces)
    # ---- ShapeGen/render.py: the NeuS-init set (the template is a T pose: turned like the shipped T-pose set, see tests/test_smpl_prior.py)
    t = np.array([0.0, 0.288, 0.211])
    data_dir = os.path.join(workdir, "render")
    u8, transforms = SR.write_nerf_dataset(data_dir, ((v + t) @ np.asarray(ROT_MAT).T).astype(np.float32), faces, device=dev, camera_distance=2.0)
    # ---- AppearanceGen stage 1: Runner.train
    conf = bench.make_conf(256, 32, small=small)
    conf.put("general.base_exp_dir", os.path.join(workdir, "exp_init"))
    conf.put("dataset.data_dir", data_dir)
    conf.put("train.batch_size", 1024)
    conf.put("train.warm_up_end", 0)
    conf.put("train.end_iter", train_iters)
    conf.put("train.save_freq", train_iters)
    r = Runner(None, mode="train", conf=conf, device=dev)
    perm = r.get_image_perm()
    r.update_learning_rate()
    train_losses = []
    for _ in range(train_iters):
        batch = r.dataset.gen_random_rays_at(perm[r.iter_step % len(perm)], r.batch_size)
        train_losses.append(float(r.train_iteration(batch)))
        r.update_learning_rate()
    r.save_checkpoint()
    ckpt = os.path.join(conf.get_string("general.base_exp_dir"), "checkpoints", "ckpt_{:0>6d}.pth".format(r.iter_step))
    # ---- AppearanceGen stage 2: Runner.train_clip from that pretrain, the prior = the chosen shape
    conf2 = bench.make_conf(res, 32, small=small)
    conf2.put("general.base_exp_dir", os.path.join(workdir, "exp_clip"))
    conf2.put("general.smpl_mesh", obj)
    conf2.put("train.pretrain", ckpt)
    conf2.put("train.warm_up_end", 0)
    r2 = Runner(None, mode="train_clip", conf=conf2, device=dev)
    w0 = r.sdf_network.lin0.weight_v.detach()
    pretrain_loaded = bool(torch.equal(r2.sdf_network.lin0.weight_v.detach(), w0))
    r2.init_clip()
    r2.init_smpl()
    r2.update_learning_rate()
    clip_losses = []
    for i in range(clip_iters):
        clip_losses.append(float(r2.train_clip_iteration(i)))
        r2.update_learning_rate()
    return dict(best=best, cos=cos.cpu().numpy(), obj=obj, frames=transforms, train_losses=train_losses, clip_losses=clip_losses,
                pretrain_loaded=pretrain_loaded)


if __name__ == "__main__":
    out = run(sys.argv[1] if len(sys.argv) > 1 else "/tmp/avc_config5")
    print("code", out["best"], "cosines", np.round(out["cos"], 3))
    print("NeuS init loss %.4f -> %.4f" % (out["train_losses"][0], np.mean(out["train_losses"][-10:])))
    print("train_clip losses", np.round(out["clip_losses"], 4))
