"""CLIP text tower: oracle pinned against transformers, tokenizer against transformers' BPE."""
import numpy as np
import pytest
import torch

from oracle import clip_text_oracle as T


def _tokens(B, seed):
    g = torch.Generator().manual_seed(seed)
    tok = torch.zeros(B, T.CTX, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(3, 40, (1,), generator=g))
        tok[b, 0] = 49406
        tok[b, 1:1 + n] = torch.randint(0, 49000, (n,), generator=g)
        tok[b, 1 + n] = 49407                      # EOT: the largest id -> argmax pooling position
    return tok


def test_text_oracle_matches_transformers_clip_text_model():
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    sd = T.random_state_dict(0)
    cfg = CLIPTextConfig(attn_implementation="eager")
    hf = CLIPTextModelWithProjection(cfg).eval()
    missing, unexpected = hf.load_state_dict(T.to_hf_state_dict(sd), strict=False)
    assert not [m for m in missing if "position_ids" not in m] and not unexpected
    tok = _tokens(3, 1)
    with torch.no_grad():
        ref = hf(input_ids=tok).text_embeds
        out = T.encode_text(sd, tok)
    assert out.shape == (3, 512)
    assert (out - ref).abs().max() < 2e-4 * ref.abs().max().clamp(min=1.0)


def _synthetic_merges(corpus, n):
    """a small BPE merge table learnt from `corpus` with the standard greedy procedure (test fixture)"""
    from avatarclip_amd.tokenizer import bytes_to_unicode
    be = bytes_to_unicode()
    words = {}
    for w in corpus.lower().split():
        sym = tuple(be[b] for b in w.encode("utf-8"))
        sym = sym[:-1] + (sym[-1] + "</w>",)
        words[sym] = words.get(sym, 0) + 1
    merges = []
    for _ in range(n):
        counts = {}
        for sym, c in words.items():
            for a, b in zip(sym[:-1], sym[1:]):
                counts[(a, b)] = counts.get((a, b), 0) + c
        if not counts:
            break
        best = max(sorted(counts), key=lambda p: counts[p])
        merges.append(best)
        new = {}
        for sym, c in words.items():
            out, i = [], 0
            while i < len(sym):
                if i + 1 < len(sym) and (sym[i], sym[i + 1]) == best:
                    out.append(sym[i] + sym[i + 1]); i += 2
                else:
                    out.append(sym[i]); i += 1
            new[tuple(out)] = new.get(tuple(out), 0) + c
        words = new
    return merges


def test_tokenizer_matches_transformers_bpe_on_a_synthetic_merge_table(tmp_path):
    import gzip, json
    from transformers import CLIPTokenizer
    from avatarclip_amd import tokenizer as TK
    corpus = ("a 3d rendering of the iron man in unreal engine the face of the back of a tall and skinny female soldier "
              "that is arguing rendering render engine man woman iron the the of of in in")
    merges = _synthetic_merges(corpus, 60)
    bpe = tmp_path / "bpe.txt.gz"
    with gzip.open(bpe, "wt", encoding="utf-8") as fp:
        fp.write("#version: test\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    tk = TK.SimpleTokenizer(str(bpe), n_merges=len(merges))
    vocab = {t: i for t, i in tk.encoder.items()}
    (tmp_path / "vocab.json").write_text(json.dumps(vocab))
    (tmp_path / "merges.txt").write_text("#version: test\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    hf = CLIPTokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"))
    prompts = ["a 3D rendering of the Iron Man in unreal engine", "the face of a tall, skinny soldier!", "it's 42 men's render-engine",
               "  spaces   and\ttabs  "]
    for p in prompts:
        mine = tk.encode(p)
        ref = hf(p, add_special_tokens=False)["input_ids"]
        assert mine == ref, (p, mine, ref)
    toks = TK.tokenize(prompts, tk)
    assert toks.shape == (4, 77) and (toks[:, 0] == tk.sot).all()
    for i, p in enumerate(prompts):
        n = len(tk.encode(p))
        assert toks[i, n + 1] == tk.eot and (toks[i, n + 2:] == 0).all() and toks[i].argmax() == n + 1
    assert tk.decode(tk.encode("iron man")).strip() == "iron man"
    with pytest.raises(RuntimeError):
        TK.tokenize("man " * 100, tk)
    with pytest.raises(FileNotFoundError):
        TK.SimpleTokenizer(str(tmp_path / "missing.gz"))


# ------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@gpu
def test_hip_text_tower_matches_oracle():
    """bf16-MFMA linears + causal attention kernel vs the fp32 oracle: cosine >= 0.9995, |cos(text, image-like vector)| within 1e-3"""
    from avatarclip_amd.clip_text import ClipTextB32
    sd = T.random_state_dict(0)
    tok = _tokens(3, 5)
    enc = ClipTextB32(sd, "cuda")
    out = enc.encode_text(tok).cpu()
    ref = T.encode_text(sd, tok)
    assert out.shape == (3, 512)
    cos = torch.nn.functional.cosine_similarity(out, ref, dim=-1)
    print("text tower cosine", cos.tolist(), "max abs", (out - ref).abs().max().item(), "ref scale", ref.abs().max().item())
    assert (cos > 0.9995).all()
    probe = torch.nn.functional.normalize(torch.randn(1, 512, generator=torch.Generator().manual_seed(9)), dim=-1)
    c1 = torch.nn.functional.cosine_similarity(out, probe.expand_as(out), dim=-1)
    c2 = torch.nn.functional.cosine_similarity(ref, probe.expand_as(ref), dim=-1)
    assert (c1 - c2).abs().max() < 1e-3


@gpu
def test_runner_encodes_prompts_with_the_text_tower(tmp_path):
    """Runner.init_clip (main.py:258-288) end to end: tokenizer (synthetic merge table) -> encode_text -> cached embeddings"""
    import gzip
    import bench
    from oracle import clip_vit_oracle as CV
    from avatarclip_amd import tokenizer as TK
    from avatarclip_amd.runner import Runner
    merges = _synthetic_merges("a 3d rendering of the iron man in unreal engine the face of the back of iron man", 40)
    bpe = tmp_path / "bpe.txt.gz"
    with gzip.open(bpe, "wt", encoding="utf-8") as fp:
        fp.write("#version: test\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    # a full CLIP state dict: vision tower + text tower; the synthetic vocabulary is smaller than 49408, ids stay in range
    sd = dict(CV.random_state_dict(0))
    sd.update(T.random_state_dict(1))
    conf = bench.make_conf(32, 32, small=True)
    conf.put("clip.bpe_path", str(bpe))
    conf.put("clip.face_prompt", "the face of the Iron Man")
    conf.put("clip.back_prompt", "the back of the Iron Man")
    runner = Runner(None, mode="train_clip", conf=conf, device=torch.device("cuda"))
    # the merge table of this test has 40 merges, the tokenizer default expects the real table's 48894
    orig = TK.SimpleTokenizer.__init__
    TK.SimpleTokenizer.__init__ = lambda self, path=None, n_merges=len(merges): orig(self, path, n_merges)
    try:
        runner.init_clip(clip_state_dict=sd)
    finally:
        TK.SimpleTokenizer.__init__ = orig
    tk = TK.SimpleTokenizer(str(bpe), n_merges=len(merges))
    for prompt, emb in ((conf.get_string("clip.prompt"), runner.encoded_text), ("the face of the Iron Man", runner.encoded_face_text),
                        ("the back of the Iron Man", runner.encoded_back_text)):
        ref = T.encode_text(sd, TK.tokenize([prompt], tk))
        cos = torch.nn.functional.cosine_similarity(emb.cpu(), ref, dim=-1).item()
        assert emb.shape == (1, 512) and cos > 0.9995, (prompt, cos)
