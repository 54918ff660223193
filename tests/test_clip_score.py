"""CLIP scoring of rendered bodies (ShapeGen codebook search, AvatarAnimate pose / motion
scoring).  CPU: the scoring arithmetic against the numpy restatement (with a stub encoder).  GPU: batched encode_image of
5-view renders on the HIP ViT kernels against the fp32 ViT oracle, and the rankings that follow from it."""
import numpy as np
import pytest
import torch

gpu = pytest.mark.gpu


class _StubPerceptor:
    def __init__(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.w = torch.randn(3 * 224 * 224, 512, generator=g) * 1e-2

    def encode_image(self, x):
        return x.reshape(x.shape[0], -1) @ self.w


def _renders(n, res, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(n, 3, res, res, generator=g)
    yy, xx = torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")
    body = (((xx - res / 2) / (0.18 * res)) ** 2 + ((yy - res / 2) / (0.42 * res)) ** 2) < 1
    return img * body


def test_scoring_arithmetic_matches_numpy_restatement():
    from avatarclip_amd import clip_score as S
    from oracle import clip_score_oracle as SO
    P = _StubPerceptor()
    num_cam, bs = 5, 3
    imgs = _renders(num_cam * bs, 256, 1)
    assert np.abs(S.preprocess_renders(imgs).numpy() - SO.preprocess(imgs.numpy())).max() < 1e-5
    emb = S.render_embedding(P, imgs)
    pf = S.pose_feature(P, imgs, num_cam)
    assert np.abs(pf.numpy() - SO.pose_feature(emb.numpy(), num_cam)).max() < 1e-5
    g = torch.Generator().manual_seed(3)
    text = torch.randn(1, 512, generator=g)
    sc = S.pose_score(text, pf)
    assert np.abs(sc.numpy() - SO.cosine(text.numpy(), pf.numpy())).max() < 1e-6
    order, top = S.rank_poses(text, pf, 2)
    assert order.tolist() == np.argsort(-SO.cosine(text.numpy(), pf.numpy()), kind="stable")[:2].tolist()
    loss = S.motion_clip_loss(pf, text, st_idx=2, clip_num_part=5, num_frame=60)
    assert abs(float(loss) - SO.motion_clip_loss(pf.numpy(), text.numpy(), 2, 5, 60)) < 1e-6
    codebook = torch.randn(64, 512, generator=g)
    ntxt, ttxt = torch.randn(1, 512, generator=g), torch.randn(1, 512, generator=g)
    best, cos = S.shape_codebook_search(codebook, emb.mean(0), ntxt, ttxt)
    b2, c2 = SO.shape_codebook_search(codebook.numpy().astype(np.float64), emb.mean(0).numpy().astype(np.float64), ntxt.numpy(), ttxt.numpy())
    assert best == b2 and np.abs(cos.numpy() - c2).max() < 1e-5


@gpu
def test_batched_render_embeddings_and_rankings_on_the_hip_vit():
    from avatarclip_amd import clip_score as S
    from avatarclip_amd.clip_vit import ClipVisionB32
    from avatarclip_amd.runner import clip_vit_random_state_dict
    from oracle import clip_vit_oracle as C
    from oracle import clip_score_oracle as SO
    sd = clip_vit_random_state_dict(0)
    dev = torch.device("cuda")
    P = ClipVisionB32(sd, dev)
    num_cam, bs = 5, 4                         # 20 images = 1000 tokens: eight 128-row GEMM launches per linear
    imgs = _renders(num_cam * bs, 256, 2)
    pf = S.pose_feature(P, imgs.to(dev), num_cam).cpu()
    ref_emb = C.encode_image(sd, torch.from_numpy(SO.preprocess(imgs.numpy())).float())
    ref_pf = torch.from_numpy(SO.pose_feature(ref_emb.detach().numpy(), num_cam))
    cosines = torch.nn.functional.cosine_similarity(pf, ref_pf)
    print("pose-feature cosine vs fp32 oracle:", cosines.tolist())
    assert cosines.min() > 0.9995
    text = torch.nn.functional.normalize(torch.randn(1, 512, generator=torch.Generator().manual_seed(5)), dim=-1)
    s_hip, s_ref = S.pose_score(text, pf), SO.cosine(text.numpy(), ref_pf.numpy())
    assert np.abs(s_hip.numpy() - s_ref).max() < 1e-3           # SURVEY 8d gate on CLIP cosine
    # the codebook search picks the same code when the winner's margin exceeds the embedding tolerance
    g = torch.Generator().manual_seed(9)
    codebook = ref_emb.detach().mean(0, keepdim=True) + 0.3 * torch.randn(256, 512, generator=g)
    ntxt, ttxt = torch.randn(1, 512, generator=g), torch.randn(1, 512, generator=g)
    emb_hip = S.render_embedding(P, imgs[:num_cam].to(dev)).mean(0).cpu()
    best, cos = S.shape_codebook_search(codebook, emb_hip, ntxt, ttxt)
    b2, c2 = SO.shape_codebook_search(codebook.numpy().astype(np.float64), ref_emb[:num_cam].detach().mean(0).numpy().astype(np.float64),
                                      ntxt.numpy(), ttxt.numpy())
    assert np.abs(cos.numpy() - c2).max() < 2e-3
    top2 = np.sort(c2)[-2:]
    assert best == b2 or top2[1] - top2[0] < 4e-3
