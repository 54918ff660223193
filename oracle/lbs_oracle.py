"""numpy restatement of SMPL linear blend skinning as AppearanceGen uses it (models/utils.py:176-224 `my_lbs`, pose2rot=False;
smplx.lbs.batch_rigid_transform / vertices2joints) -- TEST INFRASTRUCTURE ONLY.  Explicit loops over joints and vertices."""
import numpy as np


def rodrigues(r, eps=1e-8):
    """models/utils.py:72-106 for one axis-angle vector"""
    angle = np.linalg.norm(r + eps)
    d = r / angle
    K = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def lbs(v_shaped, rot_mats, posedirs, J_regressor, parents, lbs_weights):
    """v_shaped [V,3], rot_mats [J,3,3], posedirs [(J-1)*9, V*3], J_regressor [J,V], parents [J], lbs_weights [V,J]"""
    V, J = v_shaped.shape[0], rot_mats.shape[0]
    joints = J_regressor @ v_shaped
    pose_feature = (rot_mats[1:] - np.eye(3)).reshape(-1)
    v_posed = v_shaped + (pose_feature @ posedirs).reshape(V, 3)
    world = [None] * J
    for j in range(J):
        T = np.eye(4)
        T[:3, :3] = rot_mats[j]
        T[:3, 3] = joints[j] - (joints[parents[j]] if j > 0 else 0)
        world[j] = T if j == 0 else world[parents[j]] @ T
    A = []
    for j in range(J):
        a = world[j].copy()
        a[:3, 3] = world[j][:3, 3] - world[j][:3, :3] @ joints[j]
        A.append(a)
    out = np.zeros((V, 3))
    for i in range(V):
        T = sum(lbs_weights[i, j] * A[j] for j in range(J))
        out[i] = (T @ np.append(v_posed[i], 1.0))[:3]
    return out, np.stack([w[:3, 3] for w in world])
