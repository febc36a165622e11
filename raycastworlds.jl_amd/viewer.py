"""Headless stand-in for `play!` / `copy_image_to_frame_buffer!` (SR:488-568, utils.jl:64-73).

The reference shows `camera_view` in a MiniFB window; MiniFB needs a display and is out of scope.
What remains useful for a batch of agents is getting a frame out as an ordinary image:
`frame_to_rgb` applies the same transpose the reference applies before blitting
(frame_buffer[j, i] = image[i, j], utils.jl:68-70) and unpacks 0x00RRGGBB, `save_ppm` writes it;
`play_keys` replays `play!`'s keyboard callback (SR:521-555) for a scripted key sequence and returns / dumps
the MiniFB frame buffer after every key.  Host-side only: the images come off the device through the getters.
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


ACTION_KEYS = ("w", "s", "a", "d")   # RCW.get_action_keys(env) SR:485: W, S, A, D -> actions 1..4


def frame_buffer_of(image: np.ndarray, width_image: int, height_image: int) -> np.ndarray:
    """`copy_image_to_frame_buffer!` (utils.jl:64-73) into a fresh zeroed MiniFB frame buffer
    `zeros(UInt32, width_image, height_image)` (SR:508): frame_buffer[j, i] = image[i, j].  `image` is one view as
    the engine returns it — numpy (W_img, H_img) in C order == Julia (H_img, W_img) — so the result, a numpy
    (height_image, width_image) array in C order == Julia's (width_image, height_image), has the image in its
    top-left corner."""
    image = np.asarray(image, dtype=np.uint32)
    fb = np.zeros((height_image, width_image), dtype=np.uint32)
    fb[: image.shape[1], : image.shape[0]] = image.T
    return fb


def play_keys(env, keys: str, agent: int = 0, frame_dir=None):
    """Headless replay of `play!`'s keyboard callback (SR:521-555) for a scripted key sequence: W / S / A / D act
    (the same action for every agent of the batch), R resets, V toggles between the camera view and the top view
    (needs render_top_view=True), Q closes, anything else warns ("No keybinding exists", SR:540).  After every key
    the shown agent's current view is blitted into the frame buffer exactly as the reference does (SR:543-547; the
    buffer is cleared when the view changes, SR:533).  Returns one record per key: (key, steps_taken, reward, done,
    frame_buffer); with `frame_dir` every frame buffer is also written as a PPM."""
    import os
    import warnings

    from .single_room import CAMERA_VIEW, NUM_VIEWS, TOP_VIEW, act_, reset_

    cfg = env.cfg
    has_top = bool(cfg.render_top_view)
    h_top, w_top = cfg.height_tile_map_tu * cfg.pu_per_tu, cfg.width_tile_map_tu * cfg.pu_per_tu
    h_cam, w_cam = cfg.height_camera_view_pu, cfg.num_rays
    height_image, width_image = max(h_top, h_cam), max(w_top, w_cam)              # SR:503-506

    def blit(view):
        image = env.camera_view_host(agent, 1)[0] if view == CAMERA_VIEW else env.top_view_host(agent, 1)[0]
        return frame_buffer_of(image, width_image, height_image)

    current_view, steps_taken, out = CAMERA_VIEW, 0, []
    for n, key in enumerate(keys.lower()):
        if key == "q":                                                            # SR:526-528
            break
        if key == "r":                                                            # SR:529-531
            reset_(env)
            steps_taken = 0
        elif key == "v":                                                          # SR:532-534
            current_view = (current_view % NUM_VIEWS) + 1                         # mod1(current_view + 1, NUM_VIEWS)
            if current_view == TOP_VIEW and not has_top:
                raise ValueError("the top view needs an env built with render_top_view=True")
        elif key in ACTION_KEYS:                                                  # SR:535-538
            act_(env, ACTION_KEYS.index(key) + 1)
            steps_taken += 1
        else:
            warnings.warn(f"No keybinding exists for {key}")                      # SR:540
        fb = blit(current_view)                                                   # SR:543-547
        w = env.world
        out.append((key, steps_taken, w.reward[agent], bool(w.done[agent]), fb))
        if frame_dir is not None:
            os.makedirs(frame_dir, exist_ok=True)
            rgb = np.stack([(fb >> 16) & 0xFF, (fb >> 8) & 0xFF, fb & 0xFF], axis=-1).astype(np.uint8)
            with open(os.path.join(frame_dir, f"frame_{n:04d}.ppm"), "wb") as f:
                f.write(f"P6\n{width_image} {height_image}\n255\n".encode())
                f.write(rgb.tobytes())
    return out
