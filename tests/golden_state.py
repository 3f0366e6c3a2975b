"""A fixed non-trivial initial state (yawed / pitched, moving, spinning) for tests that pass `state=`."""
import numpy as np
import torch


def given_state(B, dtype=torch.float64):
    x = torch.zeros(B, 3, dtype=dtype)
    R = torch.zeros(B, 3, 3, dtype=dtype)
    for b in range(B):
        yaw, pitch = 0.4 * (b + 1) * (-1) ** b, 0.05 * (b + 1)
        Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(pitch), 0, np.sin(pitch)], [0, 1, 0], [-np.sin(pitch), 0, np.cos(pitch)]])
        R[b] = torch.as_tensor(Rz @ Ry)
        x[b] = torch.tensor([0.2 * b - 0.1, -0.15 * b, 0.0])
    xd = torch.tensor([[0.3, 0.05, 0.0]], dtype=dtype).repeat(B, 1) * torch.arange(1, B + 1).unsqueeze(1)
    w = torch.tensor([[0.02, -0.03, 0.3]], dtype=dtype).repeat(B, 1)
    return (x, xd, R, w)
