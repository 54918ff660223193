"""Golden vectors for ShapeGen's LinearVAE (test infrastructure; run in the build container, reads /root/reference):
the reference's own class -- AvatarGen/ShapeGen/main.py:22-68, extracted with `ast` because the module imports neural_renderer, smplx
and clip at its top -- built under torch.manual_seed(0) (nn.Linear's default initialisation, construction order enc1, enc2, dec1,
dec2), decoding three latents on the CPU.  The weights themselves (169 M floats) are not stored: the test rebuilds
avatarclip_amd.shapegen.LinearVAE under the same seed, which consumes the generator identically."""
import ast
import os
import sys

import numpy as np
import torch

REF = "/root/reference/AvatarGen/ShapeGen/main.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "shapegen.npz")


def reference_class(name):
    tree = ast.parse(open(REF).read())
    node = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name == name][0]
    ns = {"nn": torch.nn, "torch": torch, "np": np}
    exec(compile(ast.Module([node], []), REF, "exec"), ns)
    return ns[name]


def main():
    LinearVAE = reference_class("LinearVAE")
    parse_prompt = reference_class("parse_prompt")
    g = torch.Generator().manual_seed(7)
    v_template = torch.randn(6890, 3, generator=g)
    torch.manual_seed(0)
    model = LinearVAE(6890 * 3, 16, v_template).eval().requires_grad_(False)
    lat = torch.randn(3, 16, generator=g)
    with torch.no_grad():
        dec = model.decode(lat)                          # main.py:67-68 (CPU: nothing in decode() needs .cuda())
        x = torch.randn(2, 6890 * 3, generator=g) * 0.1
        torch.manual_seed(5)
        out, mu, log_var = model(x)                      # main.py:46-57 (the reparameterisation draws from the global generator)
    prompts = ["a strong man", "a strong man:2", "tall:0.5:-3", "a:b:c:1:2"]
    np.savez_compressed(OUT, v_template=v_template.numpy(), latents=lat.numpy(), decoded=dec.numpy(), fwd_in=x.numpy(),
                        fwd_out=out.numpy()[:, :256], fwd_mu=mu.numpy(), fwd_log_var=log_var.numpy(),
                        dec2_bias_head=model.dec2.bias.numpy()[:64], prompts=np.array(prompts),
                        parsed=np.array([[p[0], repr(p[1]), repr(p[2])] for p in map(parse_prompt, prompts)]))
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
