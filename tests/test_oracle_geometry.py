"""An independent, brute-force check of what cast_ray computes (CPU only).

The oracle restates RayCaster.cast_ray as a grid DDA (oracle/rcw_oracle.c: orc_cast_ray; the package itself is not
vendored in the reference).  Here the same question — which obstacle tile does the ray enter first,
through which face, how far along the ray — is answered WITHOUT a march: every obstacle tile is intersected as an
axis-aligned box (slab method) in Float64 and the nearest entry wins.  The two must agree wherever the answer does not
hinge on a tie (a ray through a tile corner), which is exactly what the UNPINNED tie-break switch is about; the
distance must agree to rounding.  This guards the restatements against a shared misreading of the algorithm's
geometry (hit tile, hit dimension, Euclidean distance along the ray), not against the tie-break itself."""
import numpy as np
import pytest


def slab_nearest(obst, x, y, dx, dy):
    """(i, j, dim, t, margin): nearest obstacle tile entered by the ray (1-based tile, 1 = entered through an i-face,
    2 = through a j-face), the ray parameter of the entry, and how far the runner-up / the other axis is (tie margin)."""
    H, W = obst.shape
    best = (None, None, None, np.inf)
    second = np.inf
    axis_margin = np.inf
    for i in range(1, H + 1):
        for j in range(1, W + 1):
            if not obst[i - 1, j - 1]:
                continue
            with np.errstate(divide="ignore", invalid="ignore"):
                tx = sorted(((i - 1 - x) / dx, (i - x) / dx)) if dx != 0 else ((-np.inf, np.inf) if i - 1 <= x <= i else (np.inf, -np.inf))
                ty = sorted(((j - 1 - y) / dy, (j - y) / dy)) if dy != 0 else ((-np.inf, np.inf) if j - 1 <= y <= j else (np.inf, -np.inf))
            t_in, t_out = max(tx[0], ty[0]), min(tx[1], ty[1])
            if t_in > t_out or t_out < 0 or t_in < 0:
                continue
            if t_in < best[3]:
                second = best[3]
                best = (i, j, 1 if tx[0] > ty[0] else 2, t_in)
                axis_margin = abs(tx[0] - ty[0])
            elif t_in < second:
                second = t_in
    return best + (min(second - best[3], axis_margin),)


@pytest.mark.parametrize("bits", [32, 64])
def test_dda_equals_brute_force_box_intersection(oracle, bits):
    rng = np.random.default_rng(11 + bits)
    real = np.float64 if bits == 64 else np.float32
    checked = ties = 0
    for _ in range(60):
        H, W = int(rng.integers(4, 14)), int(rng.integers(4, 14))
        obst = np.zeros((H, W), dtype=bool)
        obst[0, :] = obst[-1, :] = True; obst[:, 0] = obst[:, -1] = True          # the wall ring SR:57-60
        obst[rng.random((H, W)) < 0.12] = True                                      # + obstacles anywhere (goal tiles count as obstacles SR:209)
        free = np.argwhere(~obst)
        if len(free) == 0:
            continue
        for _ in range(40):
            ti, tj = free[rng.integers(len(free))]
            x, y = real(ti + rng.uniform(0.02, 0.98)), real(tj + rng.uniform(0.02, 0.98))
            th = rng.uniform(0, 2 * np.pi)
            if rng.random() < 0.15:
                th = np.round(th / (np.pi / 2)) * (np.pi / 2) + rng.choice([0.0, 1e-3, -1e-3])   # near-axis rays
            dx, dy = real(np.cos(th)), real(np.sin(th))
            n = real(np.hypot(np.float64(dx), np.float64(dy)))
            dx, dy = real(dx / n), real(dy / n)
            if dx == 0 or dy == 0:
                continue                                                            # (1/0: the reference's own Inf arithmetic, covered by the hand-derived cases)
            i, j, dim, t, margin = slab_nearest(obst, np.float64(x), np.float64(y), np.float64(dx), np.float64(dy))
            assert i is not None                                                    # the wall ring closes the map
            if margin < 1e-4:
                ties += 1
                continue                                                            # a corner: the tie-break decides, not geometry
            for tie in (0, 1):
                for dist_mode in (0, 1):
                    gi, gj, gdim, gdist = oracle.cast_ray(obst, x, y, dx, dy, tie_break=tie, dist_mode=dist_mode, bits=bits)
                    assert (gi, gj, gdim) == (i, j, dim), (obst.astype(int), x, y, dx, dy, tie, dist_mode)
                    assert abs(np.float64(gdist) - t) <= (1e-12 if bits == 64 else 2e-5) * max(1.0, t), (gdist, t)
            checked += 1
    assert checked > 1500 and ties < checked // 10
