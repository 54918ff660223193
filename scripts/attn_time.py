"""Average duration of the CLIP attention kernels (forward, backward) at B images (development aid): python scripts/attn_time.py [B ...]
A/B against the VALU kernels: AVC_LIB_NAME=libavc_attnvalu.so (scripts/build_variant.sh attnvalu -DVIT_ATTN_MFMA=0)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avatarclip_amd import clip_vit as V

for B in [int(a) for a in sys.argv[1:]] or [2, 512]:
    qkv = torch.randn(B, 50, 2304, device="cuda", requires_grad=True)
    do = torch.randn(B, 50, 768, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 200
    for rep in range(2):
        outs = []
        ev[0].record()
        for _ in range(n):
            outs.append(V.AttentionFn.apply(qkv))
        ev[1].record()
        for o in outs[:n]:
            o.backward(do)
        ev[2].record()
        torch.cuda.synchronize()
        qkv.grad = None
    print("B=%d  %s  forward %.1f us  backward (incl. autograd) %.1f us" % (B, os.environ.get("AVC_LIB_NAME", "libavc.so"),
                                                                           1e3 * ev[0].elapsed_time(ev[1]) / n, 1e3 * ev[1].elapsed_time(ev[2]) / n))
