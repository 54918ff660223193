"""Host-side camera sampling helpers with the reference's semantics (AvatarGen/AppearanceGen/models/utils.py:9-70).
Plain numpy on the host RNG, exactly like the reference (np.random.* draws in the same order)."""
import numpy as np


def norm_np_arr(arr):
    return arr / np.linalg.norm(arr)


def lookat(eye, at, up):
    """camera-to-world matrix [x y z eye] (utils.py:9-27)."""
    zaxis = norm_np_arr(eye - at)
    xaxis = norm_np_arr(np.cross(up, zaxis))
    yaxis = np.cross(zaxis, xaxis)
    return np.array([
        [xaxis[0], yaxis[0], zaxis[0], eye[0]],
        [xaxis[1], yaxis[1], zaxis[1], eye[1]],
        [xaxis[2], yaxis[2], zaxis[2], eye[2]],
        [0, 0, 0, 1]])


def sphere_coord(theta, phi, r=1.0):
    return np.array([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)])


def random_eye_normal():
    """utils.py:29-41: distance U(1,2), phi U(0,2pi), theta N(0, pi/3)."""
    camera_distance = np.random.uniform(1, 2)
    phi = np.random.uniform(0, 2 * np.pi)
    theta = np.random.normal(0, np.pi / 3)
    is_front = 0 if (theta > np.pi / 2 or theta < -np.pi / 2) else 1
    return sphere_coord(theta, phi, camera_distance), theta, phi, is_front


def random_eye(is_front=None, distance=None, theta_std=None):
    """utils.py:43-57."""
    camera_distance = np.random.uniform(1, 2) if distance is None else distance
    phi = np.random.uniform(0, 2 * np.pi)
    if theta_std is None:
        theta_std = np.pi / 6
    theta = np.random.normal(0, theta_std)
    theta = np.clip(theta, -np.pi / 2, np.pi / 2)
    is_front = np.random.choice(2) if is_front is None else is_front
    if is_front == 0:
        theta += np.pi
    return sphere_coord(theta, phi, camera_distance), theta, phi, is_front


def random_at():
    """utils.py:66-70."""
    return np.random.normal(np.array([0, 0, 0]), np.array([0.1, 0.1, 0.1])).clip(-0.3, 0.3)
