"""Host-side logic of the Python mirror that needs no GPU."""
import numpy as np
import pytest


def test_host_actions_mirror_the_assert(rcw):
    from raycastworlds_jl_amd.single_room import host_actions

    np.testing.assert_array_equal(host_actions(4, 3), np.full(4, 3, np.uint8))
    np.testing.assert_array_equal(host_actions(3, [1, 2, 4]), np.array([1, 2, 4], np.uint8))
    np.testing.assert_array_equal(host_actions(2, np.array([1.0, 4.0])), np.array([1, 4], np.uint8))
    for bad in (0, 5, -1, [1, 2, 0], np.array([1.5, 2.0]), np.array([1, 9], dtype=np.uint8)):
        n = 1 if np.isscalar(bad) else len(bad)
        with pytest.raises(AssertionError, match="Invalid action"):
            host_actions(n, bad)
    with pytest.raises(ValueError):
        host_actions(3, [1, 2])


def test_unpack_tile_map_is_julia_bitarray_layout(rcw, oracle):
    from raycastworlds_jl_amd.single_room import unpack_tile_map

    orc = oracle.OracleBatch(3, seed=5, height_tile_map_tu=8, width_tile_map_tu=16, num_rays=8)
    tm = unpack_tile_map(orc.tile_map_chunks(), 8, 16)
    assert tm.shape == (3, 2, 8, 16)
    for b in range(3):
        wall, goal = tm[b, 0], tm[b, 1]
        assert wall[0, :].all() and wall[-1, :].all() and wall[:, 0].all() and wall[:, -1].all()
        assert wall.sum() == 2 * 8 + 2 * 16 - 4
        gi, gj = orc.goal[b]
        assert goal.sum() == 1 and goal[gi - 1, gj - 1]
        assert 2 <= gi <= 7 and 2 <= gj <= 15


def test_action_names_and_spaces(rcw):
    assert rcw.get_action_names(None) == ("MOVE_FORWARD", "MOVE_BACKWARD", "TURN_LEFT", "TURN_RIGHT")
    assert list(rcw.RLBase.action_space(None)) == [1, 2, 3, 4]
    assert rcw.RLBase.state_space(None) is None
    assert rcw.SingleRoomModule.NUM_ACTIONS == 4 and rcw.SingleRoomModule.WALL == 1 and rcw.SingleRoomModule.GOAL == 2


def test_shard_range(rcw):
    assert rcw.shard_range(65536, 8, 0) == (0, 8192)
    assert rcw.shard_range(65536, 8, 7) == (57344, 8192)
    assert rcw.shard_range(8, 1, 0) == (0, 8)
    covered = []
    for r in range(4):
        f, c = rcw.shard_range(4096, 4, r)
        covered += list(range(f, f + c))
    assert covered == list(range(4096))
    with pytest.raises(ValueError):
        rcw.shard_range(10, 4, 0)
    with pytest.raises(ValueError):
        rcw.shard_range(8, 2, 2)


def test_reset_generator_is_sharding_invariant(oracle):
    """The reset stream is keyed by GLOBAL agent id: a batch of 8 equals two shards of 4."""
    kw = dict(height_tile_map_tu=8, width_tile_map_tu=8, num_rays=16)
    whole = oracle.OracleBatch(8, seed=42, **kw)
    lo = oracle.OracleBatch(4, seed=42, agent_id_offset=0, **kw)
    hi = oracle.OracleBatch(4, seed=42, agent_id_offset=4, **kw)
    np.testing.assert_array_equal(whole.goal, np.concatenate([lo.goal, hi.goal]))
    np.testing.assert_array_equal(whole.position, np.concatenate([lo.position, hi.position]))
    np.testing.assert_array_equal(whole.direction, np.concatenate([lo.direction, hi.direction]))
    assert len({tuple(g) for g in whole.goal} | {tuple(p) for p in whole.position}) > 4   # not all equal


def test_reset_distributions(oracle):
    """Same distributions as reset!(world) SR:110-137: goal on interior tiles, player at the
    centre of an empty tile, heading on 0..nd-1."""
    orc = oracle.OracleBatch(4096, seed=1, render=False, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=4)
    g, p, d = orc.goal, orc.position, orc.direction
    assert g.min() >= 2 and g.max() <= 7
    assert ((p - 0.5) == np.floor(p - 0.5)).all()
    tile = (p + 0.5).astype(int)
    assert tile.min() >= 2 and tile.max() <= 7                   # never on the wall ring
    assert not (tile == g).all(axis=1).any()                     # never on the goal tile
    assert d.min() >= 0 and d.max() <= 127
    # roughly uniform: each of the 36 interior goal tiles gets 4096/36 = 114 +- 5 sigma
    counts = np.bincount((g[:, 0] - 2) * 6 + (g[:, 1] - 2), minlength=36)
    assert counts.min() > 60 and counts.max() < 170
    assert len(np.unique(d)) == 128


def test_frame_dump_matches_the_reference_blit(rcw, oracle, tmp_path):
    """frame_to_rgb applies copy_image_to_frame_buffer!'s transpose (utils.jl:64-73)."""
    orc = oracle.OracleBatch(1, height_tile_map_tu=8, width_tile_map_tu=8, num_rays=64)
    orc.set_state([[2, 2]], [[4.5, 4.5]], [0])
    frame = orc.camera_view[0]                       # (N, H_cam)
    rgb = rcw.frame_to_rgb(frame)
    assert rgb.shape == (256, 64, 3)
    assert (rgb[0] == 255).all()                     # ceiling 0x00FFFFFF on top
    assert (rgb[128] == 0x80).all()                  # wall 0x00808080 in the middle
    assert (rgb[255] == 0x40).all()                  # floor 0x00404040 at the bottom
    p = tmp_path / "frame.ppm"
    rcw.save_ppm(frame, str(p))
    data = p.read_bytes()
    assert data.startswith(b"P6\n64 256\n255\n") and len(data) == len(b"P6\n64 256\n255\n") + 64 * 256 * 3


def test_line_closed_form_equals_the_error_term_walk():
    """The write-once top view kernel (rcw_top_draw.hip, top_draw) steps a line with the remainder of
    floor((2 b k + a) / (2 a)) instead of the error term of SD.Line as the oracle restates it (sd_line in
    oracle/rcw_oracle.c).  Exhaustive check that both visit the same pixels, for every end point within +-48 px."""
    def error_term_walk(i2, j2):
        i1 = j1 = 0
        di, dj = abs(i2), -abs(j2)
        si, sj = (1 if i1 < i2 else -1), (1 if j1 < j2 else -1)
        err, out = di + dj, []
        while True:
            out.append((i1, j1))
            if i1 == i2 and j1 == j2:
                return out
            e2 = 2 * err
            if e2 >= dj:
                err += dj; i1 += si
            if e2 <= di:
                err += di; j1 += sj

    def remainder_walk(i2, j2):
        di, dj = abs(i2), abs(j2)
        si, sj = (1 if 0 < i2 else -1), (1 if 0 < j2 else -1)
        imaj = di >= dj
        a, b = (di, dj) if imaj else (dj, di)
        acc, i, j, out = a, 0, 0, []
        for _ in range(a + 1):
            out.append((i, j))
            acc += 2 * b
            t = acc >= 2 * a
            if t:
                acc -= 2 * a
            if imaj:
                i += si; j += sj if t else 0
            else:
                j += sj; i += si if t else 0
        return out

    for i2 in range(-48, 49):
        for j2 in range(-48, 49):
            assert error_term_walk(i2, j2) == remainder_walk(i2, j2), (i2, j2)


def test_line_carry_walk_equals_the_closed_form():
    """What top_draw actually carries: the 32-bit FRACTION of k * slope / 2^32 + 1/2 + 2^-18 with
    slope = floor(2^32 b / a) (computed in Float64 as the kernel does), the minor axis stepping on the carry, started
    at an arbitrary pixel k0 with one 64-bit multiply-add.  Against floor((2 b k + a) / (2 a)) for every (a, b) up to
    96, for the extreme slopes, and for random lines up to the 16,384 pixels the kernels accept (rcw_api.hip)."""
    frac0 = 0x80000000 + (1 << 14)

    def check(a, b, ks):
        slope = 0xFFFFFFFF if b >= a else int(np.floor(np.float64(b) * 4294967296.0 / np.float64(a)))
        assert slope == (0xFFFFFFFF if b >= a else (b << 32) // a)          # the Float64 quotient's floor is exact
        ks = np.asarray(ks, dtype=np.uint64)
        at_k = ks * np.uint64(slope) + np.uint64(frac0)                      # v_mad_u64_u32 (k < 2^14: no overflow)
        minor = (at_k >> np.uint64(32)).astype(np.int64)
        want = (2 * b * ks.astype(np.int64) + a) // (2 * a) if a else np.zeros(len(ks), np.int64)
        np.testing.assert_array_equal(minor, want, err_msg=f"a={a} b={b}")
        # the loop: frac += slope, a carry steps the minor axis — same as the closed form one pixel on
        frac = (at_k & np.uint64(0xFFFFFFFF)) + np.uint64(slope)
        stepped = minor + (frac >> np.uint64(32)).astype(np.int64)
        want1 = (2 * b * (ks.astype(np.int64) + 1) + a) // (2 * a) if a else np.zeros(len(ks), np.int64)
        np.testing.assert_array_equal(stepped[ks < a], want1[ks < a], err_msg=f"step a={a} b={b}")

    for a in range(1, 97):
        for b in range(0, a + 1):
            check(a, b, range(a + 1))
    rng = np.random.default_rng(4)
    for a in [16383, 16382, 8191, 4097, 4096, 1023, 1024, 513] + [int(v) for v in rng.integers(97, 16384, 300)]:
        for b in {0, 1, 2, a // 3, a // 2, a // 2 + 1, a - 2, a - 1, a, int(rng.integers(0, a + 1)), int(rng.integers(0, a + 1))}:
            if 0 <= b <= a:
                check(a, b, range(a + 1))


def test_frame_buffer_blit_is_the_reference_transpose(rcw):
    """frame_buffer_of == copy_image_to_frame_buffer! (utils.jl:64-73) into zeros(UInt32, width_image, height_image)
    (SR:508), checked element by element against the reference loop restated literally."""
    rng = np.random.default_rng(0)
    H_img, W_img, height_image, width_image = 5, 3, 7, 6
    image_julia = rng.integers(0, 2**32, (H_img, W_img), dtype=np.uint64).astype(np.uint32)     # Julia image[i, j]
    frame_buffer = np.zeros((width_image, height_image), dtype=np.uint32)                        # Julia [j, i]
    for j in range(W_img):
        for i in range(H_img):
            frame_buffer[j, i] = image_julia[i, j]                                               # utils.jl:68-70
    engine_layout = np.ascontiguousarray(image_julia.T)          # what the engine returns: (W_img, H_img) in C order
    got = rcw.frame_buffer_of(engine_layout, width_image, height_image)
    # Julia's (width_image, height_image) column-major buffer is numpy (height_image, width_image) in C order
    np.testing.assert_array_equal(got, frame_buffer.T)


def _fast_div(n, d):
    """rcw_device.h `fast_div`, operation for operation in Float32 / int32 (numpy, vectorised over n)."""
    inv = np.float32(1.0) / np.float32(d)
    q = (n.astype(np.float32) * inv).astype(np.int64)          # v_cvt_f32_i32, v_mul_f32, v_cvt_i32_f32 (truncation)
    q = q - (q * d > n)
    q = q + ((q + 1) * d <= n)
    return q


def test_fast_div_is_exact_where_the_kernels_use_it():
    """`fast_div(n, d, 1/d)` replaces the integer division in the store kernels.  Its argument — the Float32 quotient is
    off by at most one, which the two corrections repair — needs n < 2^31 / d (the products stay in int32) and a quotient
    whose Float32 error stays below one: n·2^-23 < 1 is sufficient but not necessary.  The uses: rcw_fill_flat_kernel /
    rcw_top_store_flat_kernel (n < H_cam + 256 or H·pu + 256 <= 2^20 + 256, every d from 37 resp. 9), and
    rcw_fill_frame_kernel, whose flat index runs to N·H_cam < 2^25 with d = H_cam or H_cam / 4 — beyond 2^23, where
    (float)n is no longer exact: checked here over the whole launcher-admitted range (N <= 8192, N·H_cam < 2^25)."""
    rng = np.random.default_rng(5)
    for d in list(range(1, 70)) + [84, 100, 117, 250, 255, 256, 257, 1000, 4095, 4096, 16383, 16384, (1 << 20) - 1, 1 << 20]:
        hi = min(d + 256, 1 << 21) if d >= 37 else 70000
        n = np.arange(0, hi + 1, dtype=np.int64)
        np.testing.assert_array_equal(_fast_div(n, d), n // d, err_msg=f"d={d}")
    # the frame kernel: v < N * vpc (VEC, vpc = H_cam / 4) or N * H_cam, N <= 8192, below 2^25
    for hc in (1, 2, 3, 5, 7, 21, 25, 33, 35, 36, 84 // 4, 100 // 4, 999, 4093, 4097):
        top = min(8192 * hc, 1 << 25)
        n = np.unique(np.concatenate([np.arange(max(0, top - 200000), top, dtype=np.int64),
                                      rng.integers(0, top, 200000), (np.arange(1, 8193, dtype=np.int64) * hc - 1) % top,
                                      (np.arange(0, 8192, dtype=np.int64) * hc) % top]))
        np.testing.assert_array_equal(_fast_div(n, hc), n // hc, err_msg=f"frame kernel, d={hc}")


def test_flat_plane_layout_makes_chunks_whole_words():
    """The draw kernel's plane for rcw_top_store_flat_kernel: agent a's pixel q sits at bit s_a + q of its region,
    s_a = (a · Ht·Wt) mod 256.  Then chunk c of the flat batch (pixels 256 c .. 256 c + 255) finds its bits in 8 whole
    words of the region of the first pixel's agent, OR the region of the last pixel's agent — checked against a direct
    flat bit array for images that are not a whole number of chunks."""
    rng = np.random.default_rng(9)
    for Ht, Wt, B in ((104, 104, 5), (44, 99, 7), (180, 140, 3), (72, 27, 11)):
        px = Ht * Wt
        PW = ((px + 255 + 255) // 256) * 8
        bits = rng.integers(0, 2, (B, px)).astype(np.uint8)
        region = np.zeros((B, PW * 32), dtype=np.uint8)
        for a in range(B):
            s = (a * px) % 256
            region[a, s:s + px] = bits[a]
        flat = bits.reshape(-1)
        total_chunks = (B * px + 255) // 256
        for c in range(total_chunks):
            a0 = (256 * c) // px
            a1 = min((256 * c + 255) // px, B - 1) if (256 * c + 255) // px < B else None
            off0 = (c - (a0 * px) // 256) * 256
            got = region[a0, off0:off0 + 256].copy()
            if a1 is not None and a1 != a0:
                got |= region[a1, 0:256]
            want = np.zeros(256, dtype=np.uint8)
            seg = flat[256 * c:256 * c + 256]
            want[:seg.size] = seg
            np.testing.assert_array_equal(got, want, err_msg=f"{Ht}x{Wt} chunk {c}")


def test_flat_store_kernel_reads_stay_inside_their_allocations():
    """rcw_top_store_flat_kernel's loads are CLAMPED into their arrays, not predicated (rcw_top_store.hip, `issue`), and two of
    them reach past the element they name: three tile_map words from any word of a map (12-byte load), eight plane words
    from a chunk's first word (two 16-byte loads).  Restated here with the allocation sizes of rcw_api.hip /
    rcw_api.hip — tile_map: B * nwords words + 16 bytes; top_plane: B * PW words + 64 bytes, PW = ((Ht Wt + 510) div
    256) * 8 — and checked for the EXTREME addresses over geometries whose images are not a whole number of chunks
    (profiles/r04_exp5_fault.txt: one of the three candidate causes of round 3's unexplained memory access fault)."""
    rng = np.random.default_rng(4)
    geometries = [(8, 8, 13), (8, 8, 10), (24, 24, 12), (9, 7, 19), (3, 3, 14), (5, 33, 9)] + \
                 [(int(rng.integers(3, 40)), int(rng.integers(3, 40)), int(rng.integers(9, 40))) for _ in range(60)]
    for H, W, pu in geometries:
        Ht, Wt = H * pu, W * pu
        if Ht % 4:
            continue                                           # (the flat kernel takes image heights that are a multiple of 4)
        px = Ht * Wt
        for B in (1, 2, 3, int(rng.integers(4, 50))):
            nwords = 2 * ((2 * H * W + 63) // 64)
            tile_alloc_words = B * nwords + 4                  # (+ 16 bytes)
            PW = ((px + 510) // 256) * 8
            plane_alloc_words = B * PW + 16                    # (+ 64 bytes)
            # tile_map: the furthest 12-byte load starts at the last word of the last agent's map
            assert (B - 1) * nwords + (nwords - 1) + 3 <= tile_alloc_words
            # plane words: chunk id of the flat batch, relative to the first chunk of its first pixel's agent
            total_chunks = (B * px + 255) // 256
            ids = np.unique(np.concatenate([np.arange(min(total_chunks, 4096)), np.arange(max(0, total_chunks - 4096), total_chunks),
                                            ((np.arange(1, B + 1) * px - 1) // 256)]))       # every agent's last chunk
            a0 = np.minimum((256 * ids) // px, B - 1)
            rel = ids - (a0 * px) // 256
            assert (rel >= 0).all()
            assert (rel * 8 + 8 <= PW).all(), (H, W, pu, B, int(rel.max()), PW)       # inside the agent's own region
            assert int((a0 * PW + rel * 8 + 8).max()) <= plane_alloc_words
            # the following agent's first eight words (clamped to the last agent)
            assert min(B - 1, int(a0.max()) + 1) * PW + 8 <= plane_alloc_words


# ---- DeviceArray / stream ownership: host logic of single_room.py that needs no GPU ----------------------------
class _HiddenRefTensor:
    """Stands in for the tensor `torch.as_tensor(exporter)` makes: it keeps the exporting object alive through a
    reference Python's collector cannot see (torch takes a Py_INCREF inside the storage's deleter)."""

    def __init__(self, export):
        import ctypes

        self._id = id(export)
        self.interface = export.__cuda_array_interface__
        ctypes.pythonapi.Py_IncRef(ctypes.py_object(export))

    def __del__(self):
        import ctypes

        ctypes.pythonapi.Py_DecRef(ctypes.cast(self._id, ctypes.py_object))


class _FakeLib:
    def __init__(self):
        self.destroyed = 0

    def rcw_destroy(self, h):
        self.destroyed += 1
        return 0


def _bare_env(rcw, lib):
    """A SingleRoom object with everything but the native calls: enough for the lifetime logic."""
    import ctypes

    SR = rcw.SingleRoomModule
    env = SR.SingleRoom.__new__(SR.SingleRoom)
    env._lib, env.batch, env.device, env.R = lib, 4, 0, np.float32
    env._handle = SR._Handle(lib)
    env._handle.h = ctypes.c_void_p(0x1000)
    env._held, env._free_events, env._torch_owned_stream, env.host_syncs = [], [], None, 0
    return env


def test_cached_device_tensors_do_not_keep_the_environment_alive(rcw, monkeypatch):
    """DeviceArray.torch() caches its tensor; torch's hidden reference to the exporter must not close a
    cycle through the environment, or no environment whose reward tensor was ever made is finalised (GiBs leaked)."""
    import gc
    import weakref

    SR = rcw.SingleRoomModule
    monkeypatch.setattr(SR.DeviceArray, "_tensor_over", staticmethod(lambda export, device: _HiddenRefTensor(export)))
    lib = _FakeLib()
    env = _bare_env(rcw, lib)
    env._reward_dev = SR.DeviceArray(0x2000, (4,), np.float32, env, lambda: None)     # what reward_device() caches
    t = env._reward_dev.torch(sync=False)
    assert env._reward_dev.torch(sync=False) is t                                    # made once
    assert t.interface["data"] == (0x2000, False) and t.interface["shape"] == (4,)
    ref = weakref.ref(env)
    del env
    gc.collect()
    assert ref() is None, "the environment is kept alive by its own cached tensor"
    assert lib.destroyed == 0, "a live tensor over engine memory keeps the engine allocated"
    del t
    gc.collect()
    assert lib.destroyed == 1                                                         # the last holder destroys it

    # without any tensor outstanding, dropping the environment alone destroys the engine; close() does so at once
    env = _bare_env(rcw, lib)
    env._reward_dev = SR.DeviceArray(0x2000, (4,), np.float32, env, lambda: None)
    env._reward_dev.torch(sync=False)
    del env
    gc.collect()
    assert lib.destroyed == 2
    env = _bare_env(rcw, lib)
    env._reward_dev = SR.DeviceArray(0x2000, (4,), np.float32, env, lambda: None)
    env.close()
    assert lib.destroyed == 3 and not env._h and "_reward_dev" not in env.__dict__
    env.close()
    assert lib.destroyed == 3


def test_only_streams_torch_owns_are_record_streamed(rcw, monkeypatch):
    """torch.cuda.ExternalStream has .cuda_stream like torch.cuda.Stream but wraps a stream its
    creator may destroy; Tensor.record_stream on it is the use-after-free of profiles/r03_record_stream_abort.txt.
    (torch's stream classes cannot be instantiated without a device: two classes with the same inheritance stand in;
    tests/test_gpu_parity.py::test_external_stream_is_not_record_streamed does it with the real ones.)"""
    import torch

    SR = rcw.SingleRoomModule

    class Lib(_FakeLib):
        def __init__(self):
            super().__init__()
            self.stream = 0

        def rcw_set_stream(self, h, s):
            self.stream = 0 if s is None else int(s.value or 0)
            return 0

        def rcw_get_stream(self, h, out):
            out._obj.value = self.stream
            return 0

        def rcw_last_error(self):
            return b""

    assert issubclass(torch.cuda.ExternalStream, torch.cuda.Stream)     # why hasattr / isinstance(Stream) is not enough

    class Owned:
        cuda_stream = 0x51

    class External(Owned):
        cuda_stream = 0x52

    monkeypatch.setattr(torch.cuda, "Stream", Owned)
    monkeypatch.setattr(torch.cuda, "ExternalStream", External)

    class Foreign:                                  # any other object with the attribute (cupy, a user's wrapper)
        cuda_stream = 0x53

    assert SR._torch_owns(Owned()) and not SR._torch_owns(External()) and not SR._torch_owns(Foreign())
    env = _bare_env(rcw, Lib())
    for stream, owned in ((Owned(), True), (External(), False), (Foreign(), False), (0x54, False), (Owned(), True), (None, False)):
        env.set_stream(stream)
        assert env.stream_ptr() == (getattr(stream, "cuda_stream", stream) or 0)
        assert env._stream_is_torch_owned() is owned, stream


def test_device_array_host_protocol(rcw):
    """truth value, iteration and `out=` of the arrays RLBase.reward / is_terminated return."""
    SR = rcw.SingleRoomModule
    copies = []

    def getter(vals):
        def g():
            copies.append(1)
            return np.array(vals)
        return g

    one = SR.DeviceArray(0x10, (1,), np.bool_, None, lambda: None, host_getter=getter([True]))
    many = SR.DeviceArray(0x10, (3,), np.bool_, None, lambda: None, host_getter=getter([False, True, False]))
    assert bool(one) is True
    with pytest.raises(ValueError):
        bool(many)                                  # as ndarray: ambiguous for more than one element
    copies.clear()
    assert list(many) == [False, True, False] and any(many) and len(copies) == 2     # one copy per iteration, not per element
    r = SR.DeviceArray(0x10, (3,), np.float32, None, lambda: None, host_getter=getter([0.0, 1.0, 0.0]))
    total = np.zeros(3)
    total += r                                      # host array on the left: fine
    assert total.tolist() == [0.0, 1.0, 0.0]
    with pytest.raises(TypeError):
        r += 1                                      # engine memory is not the host's to write
    with pytest.raises(TypeError):
        np.add(total, 1, out=r)


# ---- the `rng` keyword (SR:49,265): draws on the host in the reference's order ---------------------------------
class _ScriptedRng:
    """A generator whose draws are scripted and whose calls are logged: what `reference_reset_draws` asks for, in order."""

    def __init__(self, values):
        self.values, self.calls = list(values), []

    def integers(self, lo, hi):
        self.calls.append((lo, hi))
        v = self.values.pop(0)
        assert lo <= v < hi, (lo, v, hi)
        return v


def test_reference_reset_draws_follow_the_reference_order(rcw):
    """reset!(world) SR:110-137 + sample_empty_position UT:23-37, draw by draw: goal row, goal column, then ONE linear
    index per position trial — drawn again exactly while the tile is a wall or the new goal — then the heading."""
    draws = rcw.SingleRoomModule.reference_reset_draws
    H, W, nd = 8, 16, 128
    lin = lambda i, j: (i - 1) + H * (j - 1)
    # goal (3, 7); first trial a wall of the left column (5, 1), second the goal tile itself, third a wall of the bottom row, fourth free
    rng = _ScriptedRng([3, 7, lin(5, 1), lin(3, 7), lin(8, 4), lin(6, 12), 77])
    assert draws(rng, H, W, nd) == (3, 7, 6, 12, 77)
    assert rng.calls == [(2, H), (2, W), (0, H * W), (0, H * W), (0, H * W), (0, H * W), (0, nd)] and not rng.values
    # a free tile at once: exactly four draws; the OLD goal tile is free again (reset! clears it before drawing, SR:118)
    rng = _ScriptedRng([7, 15, lin(3, 7), 0])
    assert draws(rng, H, W, nd) == (7, 15, 3, 7, 0)
    assert rng.calls == [(2, H), (2, W), (0, H * W), (0, nd)]
    # every interior tile but the goal is reachable, none of the ring: distribution check with a real generator
    g = np.random.default_rng(1)
    seen = set()
    for _ in range(4000):
        gi, gj, ti, tj, d = draws(g, 5, 6, 4)
        assert 2 <= gi <= 4 and 2 <= gj <= 5 and 2 <= ti <= 4 and 2 <= tj <= 5 and (ti, tj) != (gi, gj) and 0 <= d < 4
        seen.add((gi, gj, ti, tj))
    assert len(seen) == 12 * 11


def test_reset_from_rng_injects_the_draws(rcw):
    """reset_(env, rng=...): one generator serves the agents in order (B reference worlds sharing it), a list serves agent
    a from rng[a]; masked-out agents draw nothing; construction draws twice per agent (SR:62-74, then SR:105 -> reset!)."""
    SR = rcw.SingleRoomModule
    captured = {}

    class Env:
        batch, T = 3, np.float32

        class cfg:
            height_tile_map_tu, width_tile_map_tu, num_directions = 8, 8, 128

        def set_state(self, goal, pos, head, mask=None):
            captured.update(goal=goal.copy(), pos=pos.copy(), head=head.copy(), mask=None if mask is None else mask.copy())

    lin = lambda i, j: (i - 1) + 8 * (j - 1)
    one = _ScriptedRng([2, 2, lin(4, 4), 1,   3, 3, lin(5, 5), 2,   4, 4, lin(6, 6), 3])
    SR._reset_from_rng(Env(), one, None)
    assert captured["goal"].tolist() == [[2, 2], [3, 3], [4, 4]] and captured["head"].tolist() == [1, 2, 3]
    assert captured["pos"].tolist() == [[3.5, 3.5], [4.5, 4.5], [5.5, 5.5]] and captured["pos"].dtype == np.float32
    per = [_ScriptedRng([2, 2, lin(4, 4), 1]), _ScriptedRng([]), _ScriptedRng([4, 4, lin(6, 6), 3])]
    SR._reset_from_rng(Env(), per, np.array([1, 0, 1], np.uint8))
    assert captured["mask"].tolist() == [1, 0, 1] and captured["goal"][[0, 2]].tolist() == [[2, 2], [4, 4]]
    assert all(not g.values for g in per) and per[1].calls == []
    twice = _ScriptedRng([7, 7, lin(2, 2), 9, 2, 2, lin(4, 4), 1] * 3)
    SR._reset_from_rng(Env(), twice, None, construction=True)
    assert not twice.values and captured["goal"].tolist() == [[2, 2]] * 3 and captured["head"].tolist() == [1, 1, 1]
    with pytest.raises(ValueError):
        SR._reset_from_rng(Env(), [one, one], None)


def test_sharded_rng_resets_do_not_depend_on_the_sharding(rcw):
    """The `rng` keyword behind ShardedSingleRoom: generators are global (one for all, or one per global agent); two shards
    of 3 + 3 agents end up with the states of one engine of 6 — at construction, at a reset, at a masked reset."""
    from raycastworlds_jl_amd.sharded import ShardedSingleRoom

    class Engine:                                          # an engine double: records what rcw_set_state would get
        T = np.float32

        class cfg:
            height_tile_map_tu, width_tile_map_tu, num_directions = 8, 16, 128

        def __init__(self, batch, agent_id_offset, device, **kw):
            self.batch, self.offset = batch, agent_id_offset
            self.goal = np.zeros((batch, 2), np.int32); self.pos = np.zeros((batch, 2), np.float32); self.head = np.zeros(batch, np.int32)

        def set_state(self, goal, pos, head, mask=None):
            m = np.ones(self.batch, bool) if mask is None else np.asarray(mask, bool)
            self.goal[m], self.pos[m], self.head[m] = goal[m], pos[m], head[m]

    def state(*envs):
        return (np.concatenate([e.goal for e in envs]), np.concatenate([e.pos for e in envs]), np.concatenate([e.head for e in envs]))

    def same(a, b):
        return all(np.array_equal(x, y) for x, y in zip(a, b))

    for make_rng in (lambda: np.random.default_rng(5), lambda: [np.random.default_rng(50 + a) for a in range(6)]):
        whole = ShardedSingleRoom(6, rank=0, world=1, env_factory=Engine, rng=make_rng())
        parts = [ShardedSingleRoom(6, rank=r, world=2, env_factory=Engine, rng=make_rng()) for r in range(2)]
        assert same(state(whole.env), state(*(p.env for p in parts))), "construction"
        whole.reset_(); [p.reset_() for p in parts]
        assert same(state(whole.env), state(*(p.env for p in parts))), "reset"
        gmask = np.array([1, 0, 0, 1, 1, 0], np.uint8)
        whole.reset_(global_mask=gmask); [p.reset_(global_mask=gmask) for p in parts]
        assert same(state(whole.env), state(*(p.env for p in parts))), "masked reset"
    one = ShardedSingleRoom(6, rank=1, world=2, env_factory=Engine, rng=np.random.default_rng(1))
    with pytest.raises(ValueError, match="GLOBAL mask"):
        one.reset_(local_mask=np.array([1, 0, 1], np.uint8))
    per = ShardedSingleRoom(6, rank=1, world=2, env_factory=Engine, rng=[np.random.default_rng(a) for a in range(6)])
    per.reset_(local_mask=np.array([1, 0, 1], np.uint8))          # per-agent generators: a local mask is enough


def test_action_keys_names_and_unit_helpers():
    """`RCW.get_action_keys` / `get_action_names` (SR:485-486) and utils.jl:5-7 as the mirror spells them."""
    import raycastworlds_jl_amd as RCW
    from raycastworlds_jl_amd import viewer

    assert RCW.get_action_keys(None) == viewer.ACTION_KEYS == ("w", "s", "a", "d")
    assert RCW.get_action_names(None) == ("MOVE_FORWARD", "MOVE_BACKWARD", "TURN_LEFT", "TURN_RIGHT")
    assert [RCW.wu_to_tu(x) for x in (0.0, 0.999, 1.0, 7.5)] == [1, 1, 2, 8]                       # floor(Int, x) + 1
    assert RCW.wu_to_pu(np.float32(0.125), 32) == 5 and RCW.wu_to_pu(np.float32(3.5), 32) == 113    # floor(Int, x * pu) + 1
    assert RCW.wu_to_pu(np.float32(0.1), 10) == 2                                                   # Float32: 0.1f0 * 10 == 1.0f0 exactly
    assert RCW.wu_to_pu(0.1 * 3, 10) == int(np.floor(0.1 * 3 * 10)) + 1
    assert [RCW.pu_to_tu(i, 32) for i in (1, 32, 33, 256)] == [1, 1, 2, 8]                          # (i - 1) ÷ pu + 1
    # Julia's ÷ truncates toward zero (Python's // floors): pixels off the image's low edge — e.g. the player circle near the border
    assert [RCW.pu_to_tu(i, 32) for i in (0, -1, -30, -31, -32, -63, -64)] == [1, 1, 1, 0, 0, -1, -1]
    assert [RCW.pu_to_tu(i, 10) for i in (0, -8, -9, -10)] == [1, 1, 0, 0]


def test_comm_init_with_a_callers_unique_id_imports_no_torch():
    """`comm_init_abi(unique_id=...)` is the path of hosts WITHOUT torch.distributed (and of several ranks in one
    process); it must not import torch.  Checked in a child interpreter in which importing torch fails, against an engine double."""
    import os
    import subprocess
    import sys

    prog = r'''
import sys, types
class _Block:
    def find_spec(self, name, path=None, target=None):
        if name == "torch" or name.startswith("torch."):
            raise ImportError("torch is not installed on this host")
sys.meta_path.insert(0, _Block())
sys.path.insert(0, %r)
import raycastworlds_jl_amd as RCW
from raycastworlds_jl_amd import _capi, sharded
calls = []
class Lib:
    def rcw_comm_init(self, h, uid, rank, world):
        calls.append((bytes(uid), rank, world)); return 0
class Env:
    def __init__(self, *a, **k): self._lib, self._h = Lib(), None
    def _check(self, rc): assert rc == 0
_capi.preload_rccl = lambda: None
s = sharded.ShardedSingleRoom(8, rank=1, world=2, env_factory=Env)
s.comm_init_abi(unique_id=bytes(range(128)))
assert calls == [(bytes(range(128)), 1, 2)], calls
assert "torch" not in sys.modules
print("ok")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    res = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and res.stdout.strip() == "ok", res.stdout + res.stderr


def test_turning_chunk_assignment_replayed_against_plain_division():
    """rcw_top_store_flat_kernel's wavefront -> chunk assignment TURNS from group to group (slot (g + k R) mod G of
    group k) and carries every lane's (agent, image column, row) from group to group with two step sets — (dq, dr) and, where the
    slot wraps past G, (dq_w, dr_w) — instead of dividing.  Replayed here statement by statement (rcw_top_store.hip, `issue` and the
    main loop) for random grids, turns, image shapes and chunk ranges, including a short last group and turns >= G: every carried
    value equals the plain division of the chunk's first pixel, every chunk of the range is taken exactly once, and everything
    stays inside 32 bits."""
    rng = np.random.default_rng(5)
    cases = [(1024, 33, 128, 128, 0, 16384 * 64), (1024, 33, 80, 80, 0, 5000), (8, 3, 44, 52, 7, 999), (4, 4, 100, 60, 0, 700), (12, 64, 48, 48, 3, 2000)]
    for _ in range(60):
        G = int(rng.integers(1, 65)) * int(rng.choice([1, 4]))
        Ht = 4 * int(rng.integers(11, 200)); Wt = int(rng.integers(40, 300))
        c0 = int(rng.integers(0, 300)); c1 = c0 + int(rng.integers(1, 64 * G * 6))
        cases.append((G, int(rng.integers(0, 3 * G + 2)), Ht, Wt, c0, c1))
    for G, rotate, Ht, Wt, chunk_begin, chunk_end in cases:
        R = rotate % G
        step_px = (G * 64 + R) * 256
        step_px_w = step_px - G * 256
        dq, dr = divmod(step_px, Ht)
        dq_w, dr_w = divmod(step_px_w, Ht)
        dqa, dqj = divmod(dq, Wt)
        dqa_w, dqj_w = divmod(dq_w, Wt)
        if G > 64 and chunk_end > 100000:
            waves, lanes = [0, 1, G // 2, G - 1], [0, 1, 31, 63]          # (the full-size case: a sample; coverage is checked on the small ones)
        else:
            waves, lanes = range(G), range(64)
        seen = {}
        for g in waves:
            for lane in lanes:
                id0 = chunk_begin + g + lane * G
                col, rem = divmod(id0 * 256, Ht)
                a_cur, j_cur = divmod(col, Wt)
                slot_i = g
                state = dict(col=col, rem=rem, a=a_cur, j=j_cur, slot=slot_i)

                def issue(base, st=state, lane=lane):
                    ident = base + lane * G
                    want_col, want_rem = divmod(ident * 256, Ht)
                    assert (st["col"], st["rem"]) == (want_col, want_rem), (G, R, Ht, Wt, g, lane, base)
                    assert (st["a"], st["j"]) == divmod(want_col, Wt)
                    assert max(st["col"], ident, st["a"]) < 2 ** 32 and st["rem"] % 4 == 0
                    if ident < chunk_end:
                        seen[ident] = seen.get(ident, 0) + 1
                    wrap = st["slot"] + R >= G
                    st["slot"] = st["slot"] + R - G if wrap else st["slot"] + R
                    st["col"] += dq_w if wrap else dq
                    st["rem"] += dr_w if wrap else dr
                    jn = st["j"] + (dqj_w if wrap else dqj)
                    if st["rem"] >= Ht:
                        st["rem"] -= Ht; st["col"] += 1; jn += 1
                    st["a"] += dqa_w if wrap else dqa
                    if jn >= Wt:
                        jn -= Wt; st["a"] += 1
                    assert jn < Wt
                    st["j"] = jn

                base, slot = chunk_begin + g, g
                if base >= chunk_end:
                    continue
                issue(base)
                delta = lambda sl: G * 64 + R - (G if sl + R >= G else 0)          # noqa: E731
                while base + delta(slot) < chunk_end:
                    issue(base + delta(slot))
                    base += delta(slot)
                    slot = slot + R - G if slot + R >= G else slot + R
                    assert (base - chunk_begin) % (G * 64) == slot                 # group k's slot, in the compact window of group k
        if waves.__class__ is range:
            assert sorted(seen) == list(range(chunk_begin, chunk_end)) and set(seen.values()) == {1}, (G, R, Ht, Wt, chunk_begin, chunk_end)


def _line_geom(key, ip, jp):
    i2, j2 = key & 0xFFFF, key >> 16
    di, dj = abs(i2 - ip), abs(j2 - jp)
    imaj = di >= dj
    return (di if imaj else dj), (dj if imaj else di), (1 if imaj else 0) | (2 if ip < i2 else 0) | (4 if jp < j2 else 0)


def _covered_prefix(key, key_lo, key_hi, ip, jp, exact=False):
    """rcw_top_draw.hip::top_covered_prefix, statement by statement (`exact`: the last division exactly instead of its low estimate)."""
    NO = 0xFFFFFFFF
    if key_lo == NO or key_hi == NO:
        return 0
    ma, mb, mo = _line_geom(key, ip, jp)
    if key == key_lo or key == key_hi:
        return ma + 1
    la, lb, lo = _line_geom(key_lo, ip, jp)
    ha, hb, ho = _line_geom(key_hi, ip, jp)
    if lo != mo or ho != mo:
        return 0
    lm, mh = lb * ma - mb * la, mb * ha - hb * ma
    if not ((lm <= 0 and mh <= 0) or (lm >= 0 and mh >= 0)):
        return 0
    P, Q = abs(lb * ha - hb * la), la * ha
    kmax = min(la, ha)
    if P > 0:
        if exact:
            kstar = (Q - 1) // P
        else:
            kstar = int(np.float32(np.float32(Q - 1) * (np.float32(1.0) / np.float32(P))) * np.float32(0.99999)) - 1
        kmax = min(kmax, kstar)
    kmax = min(kmax, ma)
    return 0 if kmax < 0 else kmax + 1


def _line_pixels(ip, jp, key, first=0):
    """SD.Line from (ip, jp) to the key's pixel as the kernels walk it: pixel k sits k steps along the major axis and
    floor((2 b k + a) / (2 a)) along the minor one (test_line_closed_form_equals_the_error_term_walk); pixels first .. a."""
    i2, j2 = key & 0xFFFF, key >> 16
    a, b, oct_ = _line_geom(key, ip, jp)
    si, sj = (1 if ip < i2 else -1), (1 if jp < j2 else -1)
    out = set()
    for k in range(first, a + 1):
        m = (2 * b * k + a) // (2 * a) if a else 0
        out.add((ip + si * k, jp + sj * m) if oct_ & 1 else (ip + si * m, jp + sj * k))
    return out


def test_draw_kernel_leaves_out_only_pixels_that_other_lines_draw(oracle):
    """Round 5: the top view's draw kernel does not walk the leading pixels of a ray's line that the rays 2^t before and behind it
    in the fan draw anyway (rcw_top_draw.hip::top_covered_prefix, and the class-by-class list in top_draw_body) — EXACTLY: the union
    of what is still walked equals the union of all lines.  Replayed here on the rays of real agents (the oracle's end points, several
    map / pixel-scale / ray-count shapes, incl. fans across an axis and rays that end in the same pixel), with the kernel's low
    Float32 estimate of the one division and with the exact quotient; and the estimate never exceeds the exact quotient."""
    rng = np.random.default_rng(11)
    P = rng.integers(1, 1 << 28, 200000); Q = rng.integers(1, 1 << 28, 200000)
    est = (np.float32(0.99999) * (np.float32(1.0) / P.astype(np.float32) * (Q - 1).astype(np.float32))).astype(np.int64) - 1
    assert (est <= (Q - 1) // P).all()
    # (the device's v_rcp_f32 is within an ulp of the division used here: 2^-23 against the 2^-16.6 the factor 0.99999 leaves)
    total_all = total_walked = 0
    for H, W, pu, N, B in ((8, 8, 32, 256, 12), (8, 16, 32, 512, 6), (8, 8, 10, 256, 12), (16, 16, 20, 64, 8), (6, 9, 13, 37, 10), (12, 12, 32, 1024, 3)):
        orc = oracle.OracleBatch(B, seed=5, height_tile_map_tu=H, width_tile_map_tu=W, num_rays=N, out_of_bounds=1)
        for _ in range(17):
            orc.step(rng.integers(1, 5, B).astype(np.uint8))
        pos = orc.position
        dist = np.asarray(orc.ray_dist).reshape(B, N); dirs = np.asarray(orc.ray_dirs).reshape(B, N, 2)
        f32 = np.float32
        ex = (pos[:, None, 0] + (dist * dirs[:, :, 0]).astype(f32)).astype(f32); ey = (pos[:, None, 1] + (dist * dirs[:, :, 1]).astype(f32)).astype(f32)
        i2 = np.floor(ex * f32(pu)).astype(np.int64) + 1; j2 = np.floor(ey * f32(pu)).astype(np.int64) + 1
        ipa = np.floor(pos[:, 0] * f32(pu)).astype(np.int64) + 1; jpa = np.floor(pos[:, 1] * f32(pu)).astype(np.int64) + 1
        for a in range(B):
            ip, jp = int(ipa[a]), int(jpa[a])
            ends = [int(i2[a, r]) | (int(j2[a, r]) << 16) for r in range(N)]
            everything = set()
            for r in range(N):
                everything |= _line_pixels(ip, jp, ends[r])
            for exact in (False, True):
                walked = set(); nwalk = 0
                for r in range(N):
                    skip = 0
                    if r > 0:
                        t = (r & -r)
                        if r + t < N:
                            skip = _covered_prefix(ends[r], ends[r - t], ends[r + t], ip, jp, exact)
                    px = _line_pixels(ip, jp, ends[r], skip)
                    nwalk += max(0, _line_geom(ends[r], ip, jp)[0] + 1 - skip)
                    walked |= px
                assert walked == everything, (H, W, pu, N, a, exact)
            total_all += sum(_line_geom(e, ip, jp)[0] + 1 for e in ends); total_walked += nwalk
        orc.close()
    assert total_walked < 0.75 * total_all          # (and it is worth it: a quarter of the pixel-steps, at least, over these shapes)


def test_the_slot_words_padding_is_column_padding():
    """rcw_cast.hip `spec_padding` (the one-launch step's slot word, four 32-bit instructions) against rcw_device.h `column_padding` (SR:433-436: 0 from
    h >= H_cam - 1 on, else (H_cam - h) / 2 in 64 bits, clamped to H_cam), both replayed here: every camera height the one-launch step takes a sample of,
    every height_line_pu around the boundaries and across the whole Int32 range."""
    def column_padding(Hc, h):
        h = h.astype(np.int64)
        pad = (Hc - h) // 2                                                  # (Hc - h > 0 wherever it is used)
        return np.where(h >= Hc - 1, 0, np.minimum(pad, Hc))

    def spec_padding(Hc, h):
        g = np.minimum(np.maximum(h.astype(np.int64), -Hc - 2), Hc)
        assert (np.abs(Hc - g) < 2 ** 31).all()                              # (the device computes in Int32)
        return np.minimum((Hc - g) >> 1, Hc)

    rng = np.random.default_rng(1)
    edge = np.array([-2 ** 31, -2 ** 31 + 1, -2 ** 30, -70000, -8194, -8193, -8192, -1, 0, 1, 2 ** 30, 2 ** 31 - 2, 2 ** 31 - 1], dtype=np.int64)
    for Hc in [1, 2, 3, 63, 64, 100, 128, 255, 256, 257, 512, 768, 1024, 4096, 8190, 8191] + [int(v) for v in rng.integers(1, 8192, 40)]:
        near = np.arange(-3 * Hc - 8, 3 * Hc + 8, dtype=np.int64)
        h = np.concatenate([edge, near, rng.integers(-2 ** 31, 2 ** 31, 20000)]).astype(np.int64)
        np.testing.assert_array_equal(spec_padding(Hc, h), column_padding(Hc, h), err_msg=f"H_cam {Hc}")
