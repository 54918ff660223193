"""ShapeGen's text -> coarse body shape stage (AvatarGen/ShapeGen/main.py), the first link of BASELINE
config 5:  decode the 16-D shape codebook with the LinearVAE decoder (main.py:22-68), render the zero-beta body, embed it with the HIP
CLIP image encoder, pick the code whose pre-computed CLIP embedding moves from the neutral body's embedding in the direction "target
text - neutral text" (main.py:93-123), decode that code into a mesh and write it as an .obj (utils.py:35-57).  The .obj then goes to
`avatarclip_amd.shapegen_render` (108-view NeuS-init set) -> `Runner.train` -> `Runner.train_clip`.

    python -m avatarclip_amd.shapegen --AE_path_fname data/model_VAE_16.pth --codebook_fname data/codebook.pth \\
        --template_obj data/zero_beta_smpl.obj --clip_weights ViT-B-32.pt --clip_bpe bpe_simple_vocab_16e6.txt.gz \\
        --target_txt "a 3d rendering of a strong man in unreal engine" --output_folder output/coarse_shape

Inputs the reference downloads or ships outside the repository (the VAE / codebook blobs of ShapeGen/data, the SMPL template, the CLIP
weights + merge table) are arguments here.  The body render that is embedded uses the HIP rasteriser with neural_renderer's lighting
of a white body (`smpl_prior.MeshPrior`), camera distance 2, elevation 0, azimuth 150 degrees (utils.py:10-27); the reference textures
the body with `data/smpl_uv.obj` first, which is not part of the repository either -- `render_fn` takes any replacement."""
import argparse
import os

import numpy as np
import torch
from torch import nn

from . import clip_score
from .shapegen_render import get_points_from_angles

N_VERTS = 6890


class LinearVAE(nn.Module):
    """main.py:22-68 (same attribute names: the reference's `model_VAE_16.pth` state dict loads as it is).  Two plain linears each
    way, no activation between them; the decoder adds the body template."""

    def __init__(self, in_dim, latent_dim, v_template):
        super().__init__()
        self.v_template = v_template
        self.latent_dim = latent_dim
        self.enc1 = nn.Linear(in_features=in_dim, out_features=8192)
        self.enc2 = nn.Linear(in_features=8192, out_features=latent_dim * 2)
        self.dec1 = nn.Linear(in_features=latent_dim, out_features=8192)
        self.dec2 = nn.Linear(in_features=8192, out_features=in_dim)

    def reparameterize(self, mu, log_var):
        std = torch.exp(0.5 * log_var)
        return mu + torch.randn_like(std) * std

    def forward(self, x):
        latent_param = self.enc2(self.enc1(x)).view(-1, 2, self.latent_dim)
        mu, log_var = latent_param[:, 0, :], latent_param[:, 1, :]
        z = self.reparameterize(mu, log_var)
        return self.dec2(self.dec1(z)), mu, log_var

    def _template(self):
        return self.v_template.reshape(1, -1, 3).to(self.dec2.weight.device)

    def sample_z(self):
        return torch.tensor(np.random.normal(0., 1., size=(1, self.latent_dim))).float().to(self.dec2.weight.device)

    def sample(self):
        zgen = self.sample_z()
        return self.decode(zgen), zgen

    def decode(self, latent):
        """[B, latent_dim] -> [B, V, 3] vertices (main.py:67-68)"""
        return self.dec2(self.dec1(latent)).reshape(latent.shape[0], -1, 3) + self._template()


def parse_prompt(prompt):
    """main.py:70-73: 'text[:weight[:stop]]'"""
    vals = prompt.rsplit(':', 2)
    vals = vals + ['', '1', '-inf'][len(vals):]
    return vals[0], float(vals[1]), float(vals[2])


def create_load_AE(in_dim, latent_dim, v_template, pth_fname, device="cuda"):
    """main.py:75-79 (tensors-only load: the file is a plain state dict)"""
    model_AE = LinearVAE(in_dim, latent_dim, torch.as_tensor(v_template).float())
    model_AE.load_state_dict(torch.load(pth_fname, map_location="cpu", weights_only=True))
    return model_AE.eval().requires_grad_(False).to(device)


def load_codebook(fname, device="cuda"):
    """main.py:86-91: the file holds ONE dict entry {codes [N, 16]: their CLIP image embeddings [N, 512]} (a tensor as the key);
    also accepted: a dict with the string keys 'codebook' / 'clip_codebook'."""
    d = torch.load(fname, map_location="cpu", weights_only=True)
    if "codebook" in d and "clip_codebook" in d:
        return d["codebook"].to(device), d["clip_codebook"].to(device)
    for k, v in d.items():
        return k.to(device), v.to(device)
    raise ValueError("empty codebook file %s" % fname)


def render_body(vertices, faces, device="cuda", image_size=256, camera_distance=2.0, elevation=0.0, angles=(150,)):
    """utils.py:10-27 `render_one_batch` on the HIP rasteriser: [len(angles), 3, S, S] in [0, 1] (white body, neural_renderer's
    default light, camera_mode 'look_at' the origin)"""
    from .smpl_prior import MeshPrior
    prior = MeshPrior(np.asarray(vertices, np.float32), faces, device=device, image_size=image_size, apply_rot_mat=False)
    imgs = []
    for a in angles:
        eye = get_points_from_angles(camera_distance, elevation, a)
        grey = prior.render_grey(eye, -eye / np.linalg.norm(eye))
        imgs.append(grey[None].repeat(3, 1, 1))
    return torch.stack(imgs)


def writeOBJ(file, V, F):
    """utils.py:35-57 (vertices + 1-based triangle indices)"""
    with open(file, 'w') as fh:
        for v in V:
            fh.write('v ' + ' '.join(str(float(x)) for x in v) + '\n')
        for f in F:
            fh.write('f ' + ' '.join(str(int(i) + 1) for i in f) + '\n')


@torch.no_grad()
def shape_gen(model_AE, faces, perceptor, codebook, clip_codebook, neutral_text_embed, target_text_embed, render_fn=None):
    """main.py:93-123 -> (vertices [V,3] of the chosen shape, vertices of the zero-beta shape, index of the code, cosine per code).
    `perceptor`: avatarclip_amd.clip_vit.ClipVisionB32; the text embeddings [1,512] come from its text tower (encode_text of the
    tokenised prompts) or from a cache."""
    dev = codebook.device
    render_fn = render_fn or (lambda v: render_body(v, faces, device=dev))
    zero = model_AE.decode(torch.zeros(1, model_AE.latent_dim, device=dev))
    images = render_fn(zero[0].cpu().numpy())
    neutral_image_embed = clip_score.render_embedding(perceptor, images).mean(0)          # main.py:104-108
    best, cos = clip_score.shape_codebook_search(clip_codebook, neutral_image_embed, neutral_text_embed, target_text_embed)
    v = model_AE.decode(codebook[best].reshape(1, -1))
    return v[0].cpu().numpy(), zero[0].cpu().numpy(), best, cos


def main(argv=None):
    from . import clip_vit, tokenizer
    from .smpl_prior import read_obj
    ap = argparse.ArgumentParser()
    ap.add_argument('--AE_path_fname', type=str, default='./data/model_VAE_16.pth')
    ap.add_argument('--codebook_fname', type=str, default='./data/codebook.pth')
    ap.add_argument('--template_obj', type=str, required=True, help="the SMPL template (v_template, faces) as an .obj, e.g. AppearanceGen/data/zero_beta_smpl.obj")
    ap.add_argument('--clip_weights', type=str, default=os.environ.get("AVC_CLIP_WEIGHTS"))
    ap.add_argument('--clip_bpe', type=str, default=os.environ.get("AVC_CLIP_BPE"))
    ap.add_argument('--neutral_txt', type=str, default='a 3d rendering of a person in unreal engine')
    ap.add_argument('--target_txt', type=str, default='a 3d rendering of a strong man in unreal engine')
    ap.add_argument('--output_folder', type=str, default='./output/coarse_shape')
    args = ap.parse_args(argv)
    if not args.clip_weights or not args.clip_bpe:
        raise SystemExit("the CLIP ViT-B/32 checkpoint and its BPE merge table are inputs (--clip_weights / --clip_bpe)")
    dev = torch.device("cuda")
    v_template, faces = read_obj(args.template_obj)
    model_AE = create_load_AE(v_template.size, 16, v_template, args.AE_path_fname, dev)
    codebook, clip_codebook = load_codebook(args.codebook_fname, dev)
    perceptor = clip_vit.ClipVisionB32(clip_vit.load_state_dict(args.clip_weights), dev)
    tok = tokenizer.SimpleTokenizer(args.clip_bpe)
    ntxt, ttxt = parse_prompt(args.neutral_txt)[0], parse_prompt(args.target_txt)[0]
    nembed = perceptor.encode_text(tokenizer.tokenize([ntxt], tok).to(dev)).float()
    tembed = perceptor.encode_text(tokenizer.tokenize([ttxt], tok).to(dev)).float()
    print("Start generating coarse body shape given the target text: {}".format(args.target_txt))
    v, _, best, _ = shape_gen(model_AE, faces, perceptor, codebook, clip_codebook, nembed, tembed)
    os.makedirs(args.output_folder, exist_ok=True)
    output_fname = os.path.join(args.output_folder, '_'.join(args.target_txt.split(' ')) + '.obj')
    writeOBJ(output_fname, v, faces)
    print("code {} -> results saved in {}".format(best, output_fname))


if __name__ == '__main__':
    main()
