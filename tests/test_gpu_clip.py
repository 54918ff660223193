"""GPU parity of the CLIP ViT-B/32 image encoder kernels vs the fp32 CPU oracle (seeded weights, see
oracle/clip_vit_oracle.py for the parity status).  Tolerances: bf16 GEMM operands with fp32 accumulate ->
cosine similarity of the embeddings >= 0.9995 and |d cos(emb, text)| <= 1e-3 (north-star: CLIP cosine tolerance);
pixel gradient relative L2 error <= 3e-2."""
import pytest
import torch

from oracle import clip_vit_oracle as C

gpu = pytest.mark.gpu


@gpu
def test_vit_linear_and_attention_kernels():
    from avatarclip_amd import clip_vit as V
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    for (M, N, K, act) in ((50, 768, 768, 0), (100, 3072, 768, 1), (2, 512, 768, 0), (130, 768, 3072, 0)):
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) * K ** -0.5
        b = torch.randn(N, generator=g) * 0.1
        r = torch.randn(M, N, generator=g)
        lin = V._Lin(w, b, dev)
        xd = x.to(dev).requires_grad_(True)
        y = V.LinearFn.apply(xd, lin, act, r.to(dev))
        xb, wb = x.bfloat16().float(), w.bfloat16().float()
        pre = xb @ wb.t() + b
        ref = (C.quick_gelu(pre) if act else pre) + r
        assert torch.allclose(y.detach().cpu(), ref, atol=2e-3, rtol=2e-3), (M, N, K, (y.detach().cpu() - ref).abs().max())
        dy = torch.randn(M, N, generator=g)
        y.backward(dy.to(dev))
        xr = x.clone().requires_grad_(True)
        pre_r = xr @ w.t() + b
        ((C.quick_gelu(pre_r) if act else pre_r) * dy).sum().backward()
        rel = (xd.grad.cpu() - xr.grad).norm() / xr.grad.norm()
        assert rel < 1e-2, (M, N, K, rel)
    qkv = torch.randn(2, 50, 2304, generator=g)
    qd = qkv.to(dev).requires_grad_(True)
    out = V.AttentionFn.apply(qd)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr.split(768, dim=-1)
    sh = lambda t: t.reshape(2, 50, 12, 64).permute(0, 2, 1, 3)
    att = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) / 8.0, dim=-1)
    ref = (att @ sh(v)).permute(0, 2, 1, 3).reshape(2, 50, 768)
    assert torch.allclose(out.detach().cpu(), ref.detach(), atol=1e-4)
    do = torch.randn(2, 50, 768, generator=g)
    out.backward(do.to(dev))
    (ref * do).sum().backward()
    assert torch.allclose(qd.grad.cpu(), qr.grad, atol=1e-4, rtol=1e-3)


@gpu
def test_encode_image_matches_oracle():
    from avatarclip_amd import clip_vit as V
    dev = torch.device("cuda")
    sd = C.random_state_dict(0)
    model = V.ClipVisionB32(sd, dev)
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, 224, 224, generator=g)
    text = torch.randn(1, 512, generator=g)
    xd = img.to(dev).requires_grad_(True)
    emb = model.encode_image(xd)
    cos = torch.cosine_similarity(emb.mean(0), text.to(dev).mean(0), dim=0)
    cos.backward()
    xr = img.clone().requires_grad_(True)
    ref = C.encode_image(sd, xr)
    cos_r = torch.cosine_similarity(ref.mean(0), text.mean(0), dim=0)
    cos_r.backward()
    sim = torch.cosine_similarity(emb.detach().cpu(), ref.detach(), dim=-1)
    rel = (xd.grad.cpu() - xr.grad).norm() / xr.grad.norm()
    print("embedding cosine", sim.tolist(), "cos(emb,text)", cos.item(), cos_r.item(), "pixel-grad rel err", rel.item())
    assert sim.min() > 0.9995
    assert abs(cos.item() - cos_r.item()) < 1e-3
    assert rel < 3e-2
