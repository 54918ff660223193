"""Batched image-encoder throughput (row f-4: ShapeGen codebook search / pose retrieval encode hundreds of renders per call):
    python scripts/score_bench.py [B]         AVC_VIT_LIBRARY_GEMM=1 -> the linears through torch.mm (hipBLASLt) instead of the hand-written kernel"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import clip_vit as V
from avatarclip_amd.runner import clip_vit_random_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = V.ClipVisionB32(clip_vit_random_state_dict(0), "cuda")
x = torch.randn(B, 3, 224, 224, device="cuda")
with torch.no_grad():
    for _ in range(2):
        e = m.encode_image(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        e = m.encode_image(x)
    torch.cuda.synchronize()
dt = (time.time() - t0) / 5
print("B=%d  %s linears: %.2f ms per call = %.0f images/s = %.0f TF/s (8.8 GFLOP per image)  |emb| %.3f"
      % (B, "library (hipBLASLt)" if V.LIBRARY_GEMM else "hand-written", dt * 1e3, B / dt, B * 8.8e9 / dt / 1e12, e.norm(dim=-1).mean().item()))
