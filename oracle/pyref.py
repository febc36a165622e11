"""Second, independent CPU restatement of the SingleRoom step/render path (pure Python + numpy
Float32 scalars), written from the reference text function by function.

TEST INFRASTRUCTURE ONLY (same rules as rcw_oracle.c).  Its purpose is to pin the C oracle:
two restatements written separately from the same Julia source must agree bit for bit, and
this one is small and slow enough to be read against the reference line by line.  Single
agent, 1-based indices exactly as in Julia (arrays carry an unused row/column 0).

PARITY UNPINNED for RayCaster.cast_ray / StaticArrays.normalize / LinRange (see rcw_oracle.c).
SR = src/single_room.jl, CD = src/collision_detection.jl, UT = src/utils.jl.
"""
from __future__ import annotations

import math

import numpy as np





WALL, GOAL = 1, 2           # SR:17-18
NUM_ACTIONS = 4             # SR:19

COLOURS = dict(floor=0x00404040, ceiling=0x00FFFFFF, wall_dim_1=0x00808080, wall_dim_2=0x00C0C0C0,
               goal_dim_1=0x00800000, goal_dim_2=0x00C00000)   # SR:291-296


def wu_to_tu(x):            # UT:5
    return int(math.floor(float(x))) + 1


def turn_left(d, nd):       # UT:13 (Julia mod is floored)
    return (d + 1) % nd


def turn_right(d, nd):      # UT:14
    return (d - 1) % nd


def is_player_colliding(obstacle_map, pos, radius, T=np.float32):
    """CD:21-42.  obstacle_map[i][j] 1-based; raises IndexError where Julia raises BoundsError."""
    H, W = len(obstacle_map) - 1, len(obstacle_map[1]) - 1
    half = T(0.5)
    it, jt = wu_to_tu(pos[0]), wu_to_tu(pos[1])
    for j in range(jt - 1, jt + 2):             # CD:30
        for i in range(it - 1, it + 2):         # CD:31
            tile = (T(i) - half, T(j) - half)   # CD:33-34
            if not (1 <= i <= H and 1 <= j <= W):
                raise IndexError("BoundsError")
            if obstacle_map[i][j]:              # CD:35 (&& short-circuits)
                q = (T(pos[0]) - tile[0], T(pos[1]) - tile[1])
                proj = tuple(min(max(c, -half), half) for c in q)   # clamp.(q, -h, h) CD:11
                v = (q[0] - proj[0], q[1] - proj[1])                # CD:16
                if v[0] * v[0] + v[1] * v[1] < T(radius) * T(radius):   # CD:18
                    return True
    return False


def cast_ray(obstacle_map, x, y, dx, dy, tie_le=False, dist_pre=False, T=np.float32):
    """RayCaster.cast_ray (external, call site SR:223): canonical grid DDA.  UNPINNED."""
    H, W = len(obstacle_map) - 1, len(obstacle_map[1]) - 1
    x, y, dx, dy = T(x), T(y), T(dx), T(dy)
    i, j = wu_to_tu(x), wu_to_tu(y)
    with np.errstate(divide="ignore", invalid="ignore"):
        ddx = abs(T(1) / dx)
        ddy = abs(T(1) / dy)
        if dx < 0:
            si, sx = -1, (x - T(i - 1)) * ddx
        else:
            si, sx = 1, (T(i) - x) * ddx
        if dy < 0:
            sj, sy = -1, (y - T(j - 1)) * ddy
        else:
            sj, sy = 1, (T(j) - y) * ddy
        dim, dist = 0, T(0)
        while True:
            if not (1 <= i <= H and 1 <= j <= W):
                raise IndexError("BoundsError")
            if obstacle_map[i][j]:
                break
            if (sx <= sy) if tie_le else (sx < sy):
                dist = sx; sx = sx + ddx; i += si; dim = 1
            else:
                dist = sy; sy = sy + ddy; j += sj; dim = 2
        if not dist_pre:
            if dim == 1:
                dist = sx - ddx
            elif dim == 2:
                dist = sy - ddy
    return i, j, dim, T(dist)


class World:
    """SingleRoomWorld + the camera half of SingleRoom, one agent (SR:21-40, SR:241-256)."""

    def __init__(self, H=8, W=16, nd=128, radius=1 / 8, inc=1 / 8, fov=2 / 3, num_rays=512,
                 camera_height=1.0, Hc=256, tie_le=False, dist_pre=False, normalize_divide=False, T=np.float32,
                 R=np.float32):
        self.T = T                                   # the reference's world-unit type (SR:259)
        self.R = R                                   # the reward type (SR:266): reward = zero(R), goal_reward = one(R) SR:81-82
        self.H, self.W, self.nd, self.N, self.Hc = H, W, nd, num_rays, Hc
        self.radius, self.inc, self.fov = self.T(radius), self.T(inc), self.T(fov)
        self.camh = self.T(camera_height)
        self.tie_le, self.dist_pre, self.normalize_divide = tie_le, dist_pre, normalize_divide
        # tile_map[o][i][j], 1-based (SR:54-60)
        self.tile_map = [None] + [[[False] * (W + 1) for _ in range(H + 1)] for _ in range(2)]
        for i in range(1, H + 1):
            self.tile_map[WALL][i][1] = True
            self.tile_map[WALL][i][W] = True
        for j in range(1, W + 1):
            self.tile_map[WALL][1][j] = True
            self.tile_map[WALL][H][j] = True
        # directions_wu SR:65-69
        self.directions = []
        for i in range(1, nd + 1):
            theta = (i - 1) * 2 * math.pi / nd
            self.directions.append((self.T(math.cos(theta)), self.T(math.sin(theta))))
        self.goal = (2, 2)
        self.tile_map[GOAL][2][2] = True
        self.pos = (self.T(1.5), self.T(1.5))
        self.dir = 0
        self.reward = self.R(0)
        self.done = False
        self.camera_view = np.zeros((self.N, Hc), dtype=np.uint32)   # [k-1][row-1] == Julia [row, k]

    def set_state(self, goal, pos, d):
        """What reset!(world) leaves behind (SR:118-132) with the draws chosen by the caller."""
        self.tile_map[GOAL][self.goal[0]][self.goal[1]] = False     # SR:118
        self.goal = (int(goal[0]), int(goal[1]))
        self.tile_map[GOAL][self.goal[0]][self.goal[1]] = True      # SR:122
        self.pos = (self.T(pos[0]), self.T(pos[1]))
        self.dir = int(d)
        self.reward = self.R(0)
        self.done = False
        self.cast_rays()
        self.update_camera_view()

    def act(self, action):
        """act!(world, action) SR:139-191."""
        assert action in range(1, NUM_ACTIONS + 1), f"Invalid action: {action}"   # SR:140
        goal_map, wall_map = self.tile_map[GOAL], self.tile_map[WALL]
        if action in (1, 2):
            d = self.directions[self.dir]
            if action == 1:   # UT:16
                new = (self.pos[0] + self.inc * d[0], self.pos[1] + self.inc * d[1])
            else:             # UT:17
                new = (self.pos[0] - self.inc * d[0], self.pos[1] - self.inc * d[1])
            g = is_player_colliding(goal_map, new, self.radius, self.T)     # SR:162
            w = is_player_colliding(wall_map, new, self.radius, self.T)     # SR:163
            if g or w:
                if g:
                    self.reward, self.done = self.R(1), True           # SR:166-168
                else:
                    self.reward, self.done = self.R(0), False          # SR:170-171
            else:
                self.pos = new                                      # SR:174
                self.reward, self.done = self.R(0), False
        else:
            self.dir = turn_left(self.dir, self.nd) if action == 3 else turn_right(self.dir, self.nd)
            self.reward, self.done = self.R(0), False                  # SR:186-187

    def ray_fan(self):
        """SR:214-221: normalized ray directions for the current heading."""
        d = self.directions[self.dir]
        cam = (d[1], -d[0])                                         # rotate_minus_90 SR:193
        first = (d[0] + self.fov * cam[0], d[1] + self.fov * cam[1])   # SR:216
        last = (d[0] - self.fov * cam[0], d[1] - self.fov * cam[1])    # SR:217
        lendiv = max(self.N - 1, 1)
        rays = []
        for i in range(1, self.N + 1):
            t = (i - 1) / lendiv                                    # Float64 (lerpi)
            u = (self.T((1 - t) * float(first[0]) + t * float(last[0])),
                 self.T((1 - t) * float(first[1]) + t * float(last[1])))
            n = np.sqrt(u[0] * u[0] + u[1] * u[1])                  # norm
            if self.normalize_divide:
                rays.append((u[0] / n, u[1] / n))
            else:
                inv = self.T(1) / n                                    # inv(norm(a)) * a
                rays.append((inv * u[0], inv * u[1]))
        return rays

    def cast_rays(self):
        """cast_rays!(world) SR:195-231."""
        H, W = self.H, self.W
        obst = [[False] * (W + 1) for _ in range(H + 1)]            # any(tile_map, dims=1) SR:209
        for i in range(1, H + 1):
            for j in range(1, W + 1):
                obst[i][j] = self.tile_map[WALL][i][j] or self.tile_map[GOAL][i][j]
        self.ray_dirs = self.ray_fan()
        self.ray_hits = [cast_ray(obst, self.pos[0], self.pos[1], r[0], r[1], self.tie_le, self.dist_pre, self.T)
                         for r in self.ray_dirs]                    # SR:223

    def update_camera_view(self):
        """update_camera_view!(env) SR:374-444."""
        d = self.directions[self.dir]
        Hc, N = self.Hc, self.N
        self.col_height = [0] * N
        self.col_colour = [0] * N
        for i in range(1, N + 1):
            r = self.ray_dirs[i - 1]
            ih, jh, dim, dist = self.ray_hits[i - 1]
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                projected = dist * (d[0] * r[0] + d[1] * r[1])      # SR:404
                height_line = self.camh * self.T(N) / (self.T(2) * self.fov * projected)   # SR:406
            if np.isfinite(height_line):                             # SR:407-411
                h = int(math.floor(float(height_line)))
            else:
                h = Hc
            if self.tile_map[WALL][ih][jh]:                          # SR:417
                colour, cid = (COLOURS["wall_dim_1"], 0) if dim == 1 else (COLOURS["wall_dim_2"], 1)
            else:
                colour, cid = (COLOURS["goal_dim_1"], 2) if dim == 1 else (COLOURS["goal_dim_2"], 3)
            k = N - i + 1                                            # SR:431
            self.col_height[k - 1] = h
            self.col_colour[k - 1] = cid
            col = self.camera_view[k - 1]
            if h >= Hc - 1:                                          # SR:433
                col[:] = colour
            else:
                pad = (Hc - h) // 2                                  # SR:436
                col[:pad] = COLOURS["ceiling"]                       # rows 1:pad
                col[pad:Hc - pad] = colour                           # rows pad+1 : end-pad
                col[Hc - pad:] = COLOURS["floor"]                    # rows end-pad+1 : end

    # ---- top view (SR:342-372, SR:446-483); SimpleDraw 0.3 rasterisers ASSUMED (unpinned) ----
    def update_top_view(self, pu=32):
        H, W = self.H, self.W
        Ht, Wt = H * pu, W * pu
        img = np.zeros((Ht + 1, Wt + 1), dtype=np.uint32)        # 1-based [i][j]

        def put(i, j, c):
            if 1 <= i <= Ht and 1 <= j <= Wt:
                img[i, j] = c

        pu_per_tu = Ht // H                                       # SR:346
        colors = (0x00FFFFFF, 0x00FF0000, 0x00000000)             # SR:288
        for j in range(1, W + 1):                                 # draw_tile_map! SR:348-369
            for i in range(1, H + 1):
                i0, j0 = (i - 1) * pu_per_tu + 1, (j - 1) * pu_per_tu + 1
                obj = 1 if self.tile_map[WALL][i][j] else (2 if self.tile_map[GOAL][i][j] else None)
                color = colors[-1] if obj is None else colors[obj - 1]
                img[i0:i0 + pu_per_tu, j0:j0 + pu_per_tu] = color             # FilledRectangle
                img[i0, j0:j0 + pu_per_tu] = 0x00CCCCCC                        # SR:364
                img[i0 + pu_per_tu - 1, j0:j0 + pu_per_tu] = 0x00CCCCCC        # SR:365
                img[i0:i0 + pu_per_tu, j0] = 0x00CCCCCC                        # SR:366
                img[i0:i0 + pu_per_tu, j0 + pu_per_tu - 1] = 0x00CCCCCC        # SR:367

        def wu_to_pu(x):                                          # UT:6
            return int(math.floor(float(self.T(x) * self.T(pu_per_tu)))) + 1

        ip, jp = wu_to_pu(self.pos[0]), wu_to_pu(self.pos[1])     # SR:468
        rp = wu_to_pu(self.radius)                                # SR:469
        for r, (_, _, _, dist) in zip(self.ray_dirs, self.ray_hits):          # SR:473-477
            end = (self.pos[0] + dist * r[0], self.pos[1] + dist * r[1])
            i1, j1, i2, j2 = ip, jp, wu_to_pu(end[0]), wu_to_pu(end[1])
            di, dj = abs(i2 - i1), -abs(j2 - j1)                  # Bresenham (assumed SD.Line)
            si, sj = (1 if i1 < i2 else -1), (1 if j1 < j2 else -1)
            err = di + dj
            while True:
                put(i1, j1, 0x00808080)
                if i1 == i2 and j1 == j2:
                    break
                e2 = 2 * err
                if e2 >= dj:
                    err += dj; i1 += si
                if e2 <= di:
                    err += di; j1 += sj
        # SD.Circle(Point(ip - rp, jp - rp), 2 rp + 1): midpoint circle (assumed)  SR:480
        ci, cj, x, y, d = ip, jp, 0, rp, 1 - rp
        while x <= y:
            for a, b in ((x, y), (-x, y), (x, -y), (-x, -y), (y, x), (-y, x), (y, -x), (-y, -x)):
                put(ci + a, cj + b, 0x00C0C0C0)
            x += 1
            if d < 0:
                d += 2 * x + 1
            else:
                y -= 1
                d += 2 * (x - y) + 1
        self.top_view = img[1:, 1:].T.copy()                      # [j-1][i-1] == the engine's (Wt, Ht)

    def step(self, action):
        """act!(env, action) SR:333-340 (top view on request only)."""
        self.act(action)
        self.cast_rays()
        self.update_camera_view()
