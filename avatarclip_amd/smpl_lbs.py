"""Linear blend skinning of the SMPL template (the `my_lbs` of AvatarGen/AppearanceGen/models/utils.py:176-224 and the
smplx.lbs helpers it calls: batch_rodrigues :72-106, vertices2joints, batch_rigid_transform), as plain torch on whatever device
the arrays live on.  The licensed SMPL arrays (v_template, posedirs, J_regressor, parents, lbs_weights, faces) are INPUTS:
`load_smpl_arrays` reads them from an .npz export or from the official pickle."""
import os

import numpy as np
import torch


def batch_rodrigues(rot_vecs: torch.Tensor, epsilon: float = 1e-8) -> torch.Tensor:
    """models/utils.py:72-106: axis-angle [N,3] -> rotation matrices [N,3,3]"""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + epsilon, dim=1, keepdim=True, p=2)
    rot_dir = rot_vecs / angle
    cos = torch.unsqueeze(torch.cos(angle), dim=1)
    sin = torch.unsqueeze(torch.sin(angle), dim=1)
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=rot_vecs.dtype, device=rot_vecs.device)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view((n, 3, 3))
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device).unsqueeze(dim=0)
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def batch_rigid_transform(rot_mats, joints, parents):
    """smplx.lbs.batch_rigid_transform: world transforms of the kinematic chain and their rest-pose-relative form A."""
    B, J = joints.shape[0], joints.shape[1]
    joints = joints.unsqueeze(-1)
    rel = joints.clone()
    rel[:, 1:] = rel[:, 1:] - joints[:, parents[1:]]
    T = torch.zeros(B, J, 4, 4, dtype=rot_mats.dtype, device=rot_mats.device)
    T[:, :, :3, :3] = rot_mats
    T[:, :, :3, 3] = rel[..., 0]
    T[:, :, 3, 3] = 1
    chain = [T[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[int(parents[i])], T[:, i]))
    world = torch.stack(chain, dim=1)
    posed_joints = world[:, :, :3, 3]
    jh = torch.cat([joints, torch.zeros(B, J, 1, 1, dtype=joints.dtype, device=joints.device)], dim=2)
    A = world - torch.nn.functional.pad(torch.matmul(world, jh), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed_joints, A


def lbs(v_shaped, rot_mats, posedirs, J_regressor, parents, lbs_weights):
    """models/utils.py:176-224 with pose2rot=False: v_shaped [B,V,3] (shape already applied), rot_mats [B,J,3,3] -> (verts, joints)"""
    B = rot_mats.shape[0]
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)
    ident = torch.eye(3, dtype=rot_mats.dtype, device=rot_mats.device)
    pose_feature = (rot_mats[:, 1:] - ident).reshape(B, -1)
    v_posed = torch.matmul(pose_feature, posedirs).view(B, -1, 3) + v_shaped
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents)
    nj = J_regressor.shape[0]
    T = torch.matmul(lbs_weights.unsqueeze(0).expand(B, -1, -1), A.view(B, nj, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=v_posed.dtype, device=v_posed.device)], dim=2)
    verts = torch.matmul(T, vh.unsqueeze(-1))[:, :, :3, 0]
    return verts, J_transformed


def load_smpl_arrays(path, device="cpu"):
    """.npz with the SMPL field names, or the official SMPL_*.pkl (chumpy objects are read through their `.r` array when the
    chumpy package is importable; posedirs is reshaped to [(J-1)*9, V*3] like smplx does)."""
    if path.endswith(".npz"):
        try:
            d = dict(np.load(path, allow_pickle=False))      # plain arrays: nothing from the file is executed
        except ValueError:
            if os.environ.get("AVC_ALLOW_UNSAFE_PICKLE") != "1":
                raise RuntimeError("%s holds pickled objects; loading it would execute code from the file (set AVC_ALLOW_UNSAFE_PICKLE=1 "
                                   "to allow that, or re-save the arrays with np.savez)" % path)
            d = dict(np.load(path, allow_pickle=True))
    else:
        # the official SMPL_*.pkl IS a pickle (of chumpy objects): reading it executes code from the file, as smplx / the reference do
        import logging
        import pickle
        logging.warning("%s is a pickle: loading it executes code from the file (convert it to .npz to avoid that)", path)
        with open(path, "rb") as f:
            d = pickle.load(f, encoding="latin1")
    g = lambda k: np.asarray(d[k].r if hasattr(d[k], "r") else (d[k].todense() if hasattr(d[k], "todense") else d[k]))
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(device)
    posedirs = g("posedirs")
    if posedirs.ndim == 3:
        posedirs = posedirs.reshape(-1, posedirs.shape[-1]).T
    parents = np.asarray(g("kintree_table"))[0].astype(np.int64) if "kintree_table" in d else np.asarray(g("parents")).astype(np.int64)
    parents[0] = -1
    return dict(v_template=t(g("v_template")), posedirs=t(posedirs), J_regressor=t(g("J_regressor")), parents=torch.as_tensor(parents),
                lbs_weights=t(g("weights") if "weights" in d else g("lbs_weights")), faces=np.asarray(g("f") if "f" in d else g("faces")).astype(np.int32))
