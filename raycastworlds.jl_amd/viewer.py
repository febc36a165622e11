"""Headless stand-in for `play!` / `copy_image_to_frame_buffer!` (SR:488-568, utils.jl:64-73).

The reference shows `camera_view` in a MiniFB window; MiniFB needs a display and is out of scope.
What remains useful for a batch of agents is getting a frame out as an ordinary image:
`frame_to_rgb` applies the same transpose the reference applies before blitting
(frame_buffer[j, i] = image[i, j], utils.jl:68-70) and unpacks 0x00RRGGBB, `save_ppm` writes it.
Host-side only; nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np


def frame_to_rgb(frame: np.ndarray) -> np.ndarray:
    """One agent's camera_view as returned by the engine — uint32 (N, H_cam), i.e. Julia's
    (H_cam, N) column-major image — to uint8 (H_cam, N, 3) with row 0 at the top."""
    frame = np.asarray(frame, dtype=np.uint32)
    if frame.ndim != 2:
        raise ValueError("expected one frame of shape (num_rays, height_camera_view_pu)")
    img = frame.T                                   # [row, column]
    return np.stack([(img >> 16) & 0xFF, (img >> 8) & 0xFF, img & 0xFF], axis=-1).astype(np.uint8)


def save_ppm(frame: np.ndarray, path: str) -> None:
    """Write one frame as a binary PPM (P6)."""
    rgb = frame_to_rgb(frame)
    with open(path, "wb") as f:
        f.write(f"P6\n{rgb.shape[1]} {rgb.shape[0]}\n255\n".encode())
        f.write(rgb.tobytes())


def save_agent_ppm(env, agent: int, path: str) -> None:
    """Copy agent `agent`'s current camera view off the device and write it."""
    save_ppm(env.camera_view_host(agent, 1)[0], path)
