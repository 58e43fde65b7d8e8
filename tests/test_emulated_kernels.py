"""Kernel LOGIC on the CPU: the product kernel source (csrc/bpp_kernels.hip, unmodified) compiled by g++
against the cooperative-fiber SIMT emulator under tests/emu/ and compared bit for bit with the golden
vectors recorded from the reference and with the oracle.  Same comparisons as tests/test_gpu_parity.py at
sizes a host can step in seconds; both kernel paths, both lane execution orders (a cross-lane LDS dependence
without a wave barrier differs between them), every tuning-knob shape.  This is a development aid and an
early warning -- the parity claim itself is the `-m gpu` suite on the MI355X."""
import numpy as np
import pytest

from conftest import MASK_CASES, ROLLOUT_CASES, load_golden
from test_oracle_golden import check_masks, check_rollout


@pytest.fixture(params=["tile-asc", "tile-desc", "tile4-asc", "tile2-desc", "rt-asc", "generic-asc"])
def variant(request, emu, monkeypatch):
    """tile: bpp_tile_kernel for the 10x10 / 20x20 bins (other geometries fall through to the runtime-geometry
    prefix-image kernel); rt: bpp_fast_kernel with runtime geometry for everything it supports; generic: the
    cell-scan kernel."""
    path, order = request.param.split("-")
    emu.set_knobs(force_generic=int(path == "generic"), legacy_fast=int(path == "rt"),
                  tile_groups={"tile4": 4, "tile2": 2}.get(path, 0))
    monkeypatch.setenv("BPP_EMU_ORDER", "reverse" if order == "desc" else "forward")
    yield request.param
    emu.set_knobs()


@pytest.mark.parametrize("case", ROLLOUT_CASES)
def test_emulated_rollout_matches_reference_golden(emu, variant, case):
    check_rollout(lambda pool, size, rot, E, rule: emu.EmuEnv(pool, size, rot, E, mask_rule=rule), load_golden(case))


@pytest.mark.parametrize("case", MASK_CASES)
def test_emulated_masks_match_reference_golden(emu, variant, case):
    check_masks(emu.mask_from_obs, emu.mask_from_hmap, load_golden(case))


GEOMS = [((10, 10, 10), False, 150, 11), ((10, 10, 10), True, 90, 12), ((20, 20, 20), False, 21, 13),
         ((7, 13, 8), True, 33, 14), ((5, 4, 6), False, 37, 15), ((32, 32, 40), True, 3, 16),
         ((20, 20, 10), True, 13, 17), ((20, 20, 22), True, 9, 18), ((10, 10, 11), False, 20, 20),
         ((2, 2, 5), True, 20, 21), ((1, 3, 4), False, 13, 22), ((1, 1, 3), True, 7, 24),
         ((8, 128, 10), True, 5, 25), ((4, 255, 10), False, 5, 26), ((14, 72, 12), True, 3, 27), ((12, 81, 9), False, 3, 28),
         ((10, 10, 30), True, 19, 29), ((12, 8, 46), False, 11, 30), ((10, 10, 47), False, 7, 31)]     # four histogram words (H <= 46); 47: cell scan


@pytest.mark.parametrize("size,rot,E,seed", GEOMS)
def test_emulated_vs_oracle_random_rollout(emu, oracle, variant, size, rot, E, seed):
    """Seeded rollouts with some invalid actions, a mid-run reset, bin-sized items; bins not a multiple of the
    per-wave group; elongated bins (L up to 255) that stress the candidate index decode."""
    rng = np.random.RandomState(seed)
    lo, hi = 1, max(2, min(size) // 2)
    seqs = [[tuple(rng.randint(lo, hi + 1, size=3)) for _ in range(rng.randint(3, 40))] for _ in range(17)]
    seqs[3][1] = (size[0], size[1], 1)
    seqs[5][0] = (size[0], max(1, size[1] - 1), 2)
    seqs[7][0] = (1, 1, 1)
    seqs[8][0] = (min(2, size[0]), 1, 1)
    from bpp_amd import sequences
    pool = sequences.pad_pool(seqs, size)
    steps = 24 if size[0] * size[1] <= 400 else 10
    for rule in (0, 1):
        env = emu.EmuEnv(pool, size, rot, E, env_id_base=5, env_id_total=E + 9, mask_rule=rule)
        ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=5, env_id_total=E + 9, mask_rule=rule)
        obs, mask = env.reset()
        robs, rmask = ref.reset()
        np.testing.assert_array_equal(obs, robs)
        np.testing.assert_array_equal(mask, rmask)
        M = env.M
        for t in range(steps):
            a = emu.sample_feasible(mask, seed, t, env_id_base=5)
            np.testing.assert_array_equal(a, oracle.sample_feasible(rmask, seed, t, env_id_base=5))
            bad = rng.rand(E) < 0.05
            a[bad] = rng.randint(-2, M + 3, size=int(bad.sum()))
            r, o = env.step(a), ref.step(a)
            for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
                np.testing.assert_array_equal(r[k], o[k], err_msg="%s t=%d rule=%d" % (k, t, rule))
            mask = rmask = o["mask"]
            if t == steps // 2:
                np.testing.assert_array_equal(env.reset()[0], ref.reset()[0])
                mask = rmask = ref.out["mask"].copy()
                np.testing.assert_array_equal(env.out["mask"], rmask)
        np.testing.assert_array_equal(env.hmap, ref.hmap)
        for f in ref.state.dtype.names:
            if f != "pad":
                np.testing.assert_array_equal(env.state[f], ref.state[f], err_msg=f)
        np.testing.assert_array_equal(env.ep_acc, ref.ep_acc)
        np.testing.assert_array_equal(env.episode_stats(), ref.episode_stats())
        # the many-workgroup form of the reduction (scratch buffer, last arriver runs the tree): same bits, rows cleared
        np.testing.assert_array_equal(env.episode_stats(wide=True), ref.episode_stats())
        np.testing.assert_array_equal(env.episode_stats(reset=True, wide=True), ref.episode_stats(reset=True))
        assert not env.ep_acc.any() and not env.episode_stats(wide=True).any()


@pytest.mark.parametrize("size,rot,E", [((10, 10, 10), False, 99), ((10, 10, 10), True, 70), ((20, 20, 20), False, 11),
                                         ((7, 13, 8), True, 19), ((20, 20, 20), True, 9), ((20, 20, 10), True, 7)])
def test_emulated_native_driver_and_fused_draw(emu, oracle, variant, size, rot, E):
    """bpp_rollout_uniform with the in-kernel draw of the next action == the oracle's driver."""
    from bpp_amd import sequences
    pool = sequences.cut2_pool(size, 16, seed=3, bound=(2, min(5, min(size) // 2)), native=False)
    env = emu.EmuEnv(pool, size, rot, E, env_id_base=7, env_id_total=E + 7)
    ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=7, env_id_total=E + 7)
    env.reset(), ref.reset()
    for step0, n in ((0, 9), (9, 14)):
        r, ra = emu.rollout_uniform(env, 9, step0, n)
        o, oa = oracle.rollout_uniform(ref, 9, step0, n)
        np.testing.assert_array_equal(ra, oa)
        for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(r[k], o[k], err_msg=k)
    np.testing.assert_array_equal(env.hmap, ref.hmap)


@pytest.mark.parametrize("size,rot,E", [((10, 10, 10), False, 70), ((20, 20, 20), False, 9), ((7, 13, 8), True, 19)])
def test_emulated_rollout_over_rotating_output_sets(emu, oracle, variant, size, rot, E):
    """bpp_rollout_uniform_sets (bench.py's driver: lock-step t writes output set t mod n, every lock-step draws the
    next one's actions, BPP_ROLLOUT_CONTINUE resumes without a separate draw) == the oracle's statement of it, and ==
    the plain one-set driver."""
    from bpp_amd import sequences
    pool = sequences.cut2_pool(size, 16, seed=4, bound=(2, min(5, min(size) // 2)), native=False)
    env = emu.EmuEnv(pool, size, rot, E, env_id_base=3, env_id_total=E + 3)
    ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=3, env_id_total=E + 3)
    one = oracle.OracleEnv(pool, size, rot, E, env_id_base=3, env_id_total=E + 3)
    env.reset(), ref.reset(), one.reset()
    ea = ra = None
    e_last = r_last = None
    for step0, n, nsets in ((0, 7, 3), (7, 5, 2), (12, 4, 1)):
        es, ea = emu.rollout_uniform_sets(env, 5, step0, n, nsets, resume=step0 > 0, actions=ea,
                                          first_mask=e_last["mask"] if e_last else None)
        rs, ra = oracle.rollout_uniform_sets(ref, 5, step0, n, nsets, resume=step0 > 0, actions=ra,
                                             first_mask=r_last["mask"] if r_last else None)
        np.testing.assert_array_equal(ea, ra)
        for k in range(nsets):
            for f in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
                np.testing.assert_array_equal(es[k][f], rs[k][f], err_msg="%s set %d" % (f, k))
        e_last, r_last = es[(n - 1) % nsets], rs[(n - 1) % nsets]
    o, _ = oracle.rollout_uniform(one, 5, 0, 16)
    for f in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
        np.testing.assert_array_equal(r_last[f], o[f], err_msg=f)
    np.testing.assert_array_equal(env.hmap, one.hmap)
    np.testing.assert_array_equal(env.ep_acc, one.ep_acc)


@pytest.mark.parametrize("size,rot,E", [((10, 10, 10), True, 70), ((20, 20, 20), False, 9)])
def test_emulated_epsilon_variant_of_the_rollout(emu, oracle, size, rot, E):
    """SURVEY 8d's failure-path variant (BPP_ROLLOUT_EPS in bpp_rollout_uniform_sets' flags: every draw followed by
    bpp_epsilon_override) on the product's driver == the oracle's; the override alone == its normative definition in
    numpy; epsilon = 0 changes nothing and a large epsilon really ends episodes early."""
    from bpp_amd import sequences
    pool = sequences.cut2_pool(size, 16, seed=4, native=False)
    M = size[0] * size[1] * (2 if rot else 1)
    a0 = np.arange(E, dtype=np.int64) % M
    np.testing.assert_array_equal(emu.epsilon_override(a0.copy(), M, 9, 4, 0.0, env_id_base=3), a0)
    got = emu.epsilon_override(a0.copy(), M, 9, 4, 0.3, env_id_base=3)
    np.testing.assert_array_equal(got, oracle.epsilon_override(a0.copy(), M, 9, 4, 0.3, env_id_base=3))

    def h32(seed, gid, step):      # include/bpp_abi.h: the hash of bpp_sample_feasible
        m = 0xFFFFFFFF
        h = ((seed & m) ^ (((seed >> 32) * 0x9E3779B1) & m)) ^ ((((step & m) + (step >> 32) * 0xC2B2AE3D) * 0x27D4EB2F) & m)
        h ^= (gid * 0x85EBCA77) & m
        h ^= h >> 16
        h = (h * 0x7FEB352D) & m
        h ^= h >> 15
        h = (h * 0x846CA68B) & m
        return h ^ (h >> 16)

    q = int(round(0.3 * (1 << 24)))
    want = a0.copy()
    for e in range(E):
        if (h32(9 ^ 0x5851F42D4C957F2D, 3 + e, 4) >> 8) < q:
            want[e] = (h32(9 ^ 0xDA942042E4DD58B5, 3 + e, 4) * M) >> 32
    np.testing.assert_array_equal(got, want)
    assert 0.1 * E < np.count_nonzero(got != a0) < 0.6 * E

    env = emu.EmuEnv(pool, size, rot, E, env_id_base=3, env_id_total=E + 3)
    ref = oracle.OracleEnv(pool, size, rot, E, env_id_base=3, env_id_total=E + 3)
    plain = oracle.OracleEnv(pool, size, rot, E, env_id_base=3, env_id_total=E + 3)
    env.reset(), ref.reset(), plain.reset()
    es, ea = emu.rollout_uniform_sets(env, 5, 0, 9, 2, eps=0.25)
    rs, ra = oracle.rollout_uniform_sets(ref, 5, 0, 9, 2, eps=0.25)
    np.testing.assert_array_equal(ea, ra)
    for k in range(2):
        for f in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
            np.testing.assert_array_equal(es[k][f], rs[k][f], err_msg="%s set %d" % (f, k))
    oracle.rollout_uniform_sets(plain, 5, 0, 9, 2)
    assert ref.state["episode"].sum() > plain.state["episode"].sum()      # failures ended episodes early


@pytest.mark.parametrize("epw,wpb", [(1, 1), (1, 16), (2, 4), (8, 2), (8, 8), (16, 4), (4, 16), (64, 1)])
def test_emulated_tuning_knobs_do_not_change_results(emu, epw, wpb):
    emu.set_knobs(bins_per_wave=epw, waves_per_group=wpb, xcd_remap=(epw + wpb) & 1)
    try:
        for case in ("rollout_cut2_10_rot", "rollout_wide_8x12x9_rot"):
            check_rollout(lambda pool, size, rot, E, rule: emu.EmuEnv(pool, size, rot, E, mask_rule=rule), load_golden(case))
    finally:
        emu.set_knobs()


def test_emulated_masks_property_random_geometries(emu, oracle, variant):
    rng = np.random.RandomState(99)
    for trial in range(40):
        W, L, H = rng.randint(1, 14), rng.randint(1, 14), rng.randint(1, 15)
        if trial % 8 == 0:
            W, L = rng.randint(2, 5), rng.randint(65, 256)    # elongated: L > 64 (candidate index decode)
            W = min(W, 1024 // L)
        n = rng.randint(1, 12)
        hm = rng.randint(0, H + 2, size=(n, W * L)).astype(np.int32)
        hm[rng.rand(n) < 0.4] = rng.randint(0, H + 1)
        items = np.stack([rng.randint(0, W + 2, n), rng.randint(0, L + 2, n), rng.randint(0, H + 1, n)], 1).astype(np.int32)
        size = (W, L, H)
        A = W * L
        obs = np.concatenate([hm, np.repeat(items[:, 0:1], A, 1), np.repeat(items[:, 1:2], A, 1),
                              np.repeat(items[:, 2:3], A, 1)], 1).astype(np.float32)
        for rot in (False, True):
            for rule in (0, 1):
                want = oracle.mask_from_hmap(hm, items, size, rot, rule)
                np.testing.assert_array_equal(emu.mask_from_hmap(hm, items, size, rot, rule), want,
                                              err_msg="hmap %r rot=%d rule=%d" % (size, rot, rule))
                np.testing.assert_array_equal(emu.mask_from_obs(obs, size, rot, rule), want,
                                              err_msg="obs %r rot=%d rule=%d" % (size, rot, rule))


def test_emulated_masked_act_and_stats(emu, oracle):
    """Both masked-act kernels (16 lanes per bin; wave per bin for M % 4 != 0 or M > 512, e.g. the 20x20 bin with
    rotation) against the plain PyTorch float32 reference, with the tolerances of tests/test_masked_act.py."""
    from test_masked_act import check
    for E, M in ((37, 100), (21, 200), (9, 400), (11, 36), (7, 512), (9, 800), (13, 7), (6, 1023), (10, 130)):
        check(lambda x, m, s, t, det: emu.masked_act(x, m, s, t, det), E, M, seed=E + M)
    rng = np.random.RandomState(5)
    done = (rng.rand(1000) < 0.2).astype(np.uint8)
    ret, ratio, ln = rng.rand(1000), rng.rand(1000), rng.randint(1, 50, 1000).astype(np.int32)
    for n in (1000, 5000, 1):
        done = (rng.rand(n) < 0.2).astype(np.uint8)
        ret, ratio, ln = rng.rand(n), rng.rand(n), rng.randint(1, 50, n).astype(np.int32)
        np.testing.assert_array_equal(emu.episode_stats(done, ret, ratio, ln), oracle.episode_stats(done, ret, ratio, ln))
        assert abs(oracle.episode_stats(done, ret, ratio, ln)[0] - ret[done != 0].sum()) < 1e-9


@pytest.mark.parametrize("base,total,P", [(458752, 524288, 32), (458752, 524288, 31), (65536 * 3 + 5, 65536 * 4 + 77, 10),
                                           (2 ** 31 - 700, 2 ** 31 + 12345, 17)])
def test_emulated_shard_coordinates(emu, oracle, base, total, P):
    """env_id_base / env_id_total of a multi-GPU rank (BASELINE config 5: rank 7 of 8), incl. ids beyond int32."""
    from bpp_amd import sequences
    size, E = (10, 10, 10), 70
    pool = sequences.cut2_pool(size, P, seed=1, native=False)
    env = emu.EmuEnv(pool, size, False, E, env_id_base=base, env_id_total=total)
    ref = oracle.OracleEnv(pool, size, False, E, env_id_base=base, env_id_total=total)
    env.reset(), ref.reset()
    r, ra = emu.rollout_uniform(env, 3, 0, 40)
    o, oa = oracle.rollout_uniform(ref, 3, 0, 40)
    for k in ("obs", "mask", "reward", "done", "counter", "ratio", "ep_ret", "ep_len"):
        np.testing.assert_array_equal(r[k], o[k], err_msg=k)
    for f in ("seq", "episode", "cursor", "item_cur", "item_next", "item_reset"):
        np.testing.assert_array_equal(env.state[f], ref.state[f], err_msg=f)


def test_emulated_kernel_selection(emu):
    """Default knobs: the BASELINE geometries run the tile kernel, other areas divisible by 4 the runtime-geometry
    prefix-image kernel, everything else the cell-scan kernel; the knobs reroute as documented."""
    emu.set_knobs()
    assert emu.launch_info(65536, (10, 10, 10))[:4] == [2, 1, 4, 4] and emu.launch_info(65536, (10, 10, 10))[4] == 4096
    assert emu.launch_info(65536, (10, 10, 10), True)[0] == 2 and emu.launch_info(32768, (20, 20, 20))[:3] == [2, 2, 1]
    assert emu.launch_info(100, (20, 20, 10))[:3] == [2, 1, 1] and emu.launch_info(100, (10, 10, 22))[:2] == [2, 2]
    assert emu.launch_info(100, (8, 12, 9))[0] == 1 and emu.launch_info(100, (7, 13, 8))[0] == 0
    assert emu.launch_info(100, (10, 10, 30))[0] == 0
    emu.set_knobs(legacy_fast=1)
    assert emu.launch_info(65536, (10, 10, 10))[0] == 1
    emu.set_knobs(bins_per_wave=2)
    assert emu.launch_info(65536, (10, 10, 10))[:3] == [1, 1, 2]
    emu.set_knobs(force_generic=1)
    assert emu.launch_info(65536, (10, 10, 10))[0] == 0
    emu.set_knobs()


@pytest.mark.parametrize("size,rot,E", [((10, 10, 10), True, 70), ((20, 20, 20), False, 9), ((7, 13, 8), False, 19), ((6, 6, 6), True, 37)])
def test_emulated_step_mirrors_reward_and_done_into_host_buffers(emu, oracle, variant, size, rot, E):
    """bpp_step_out.host_reward / host_done (ABI v10): the step kernels write reward and done a second time, into the
    buffers step_wait() reads without a copy -- all three kernels, NOOP bins included."""
    import ctypes
    from bpp_amd import sequences
    pool = sequences.cut2_pool(size, 8, seed=6, bound=(2, min(5, min(size) // 2)), native=False)
    for mod in (emu, oracle):
        env = mod.OracleEnv(pool, size, rot, E)
        _, mask = env.reset()
        hr, hd = np.full(E, -1.0, np.float32), np.full(E, 7, np.uint8)
        env._o.host_reward, env._o.host_done = hr.ctypes.data, hd.ctypes.data
        rng = np.random.RandomState(2)
        for t in range(12):
            a = oracle.sample_feasible(mask, 4, t)
            a[rng.rand(E) < 0.2] = -1
            a[rng.rand(E) < 0.2] = -2 ** 63                  # BPP_ACTION_NOOP
            o = env.step(a)
            np.testing.assert_array_equal(hr, o["reward"])
            np.testing.assert_array_equal(hd, o["done"])
            mask = o["mask"]
        env._o.host_done = None                              # one without the other is refused
        with pytest.raises(RuntimeError):
            env.step(a)
        env._o.host_reward = None
        env.step(a)


def test_emulated_wide_reduction_equals_the_one_workgroup_form_on_many_rows(emu, oracle):
    """bpp_episode_acc_reduce with and without a scratch buffer on 5 000 rows of awkward float64 values (sizes that are not
    a multiple of 1 024, rows past the unrolled part): identical bits, identical to the oracle's fixed order."""
    import ctypes
    rng = np.random.RandomState(5)
    for E in (1, 15, 1024, 1025, 5000, 16 * 1024 + 3):
        rows = (rng.rand(E, 4) * 10.0 ** rng.randint(-6, 6, size=(E, 4))).astype(np.float64)
        got = {}
        for name, lib, wide in (("oracle", oracle.lib(), False), ("emu_one", emu.lib(), False), ("emu_wide", emu.lib(), True)):
            acc = np.array([1.5, -2.0, 0.25, 7.0])
            buf = np.zeros(E * 4 + 4, np.float64)                      # the ABI wants the rows 32-byte aligned
            off = (-buf.ctypes.data % 32) // 8
            r = buf[off:off + 4 * E].reshape(E, 4)
            r[:] = rows
            scratch = np.zeros(1024 * 4 + 8, np.float64)
            assert lib.bpp_episode_acc_reduce(r.ctypes.data, E, acc.ctypes.data, 1, scratch.ctypes.data if wide else None, None) == 0
            assert not r.any()
            got[name] = acc
        np.testing.assert_array_equal(got["emu_one"], got["oracle"])
        np.testing.assert_array_equal(got["emu_wide"], got["oracle"])


def test_emulated_gather_finished_compacts_in_bin_order(emu, oracle):
    """bpp_gather_finished (ordered compaction by one workgroup, header count, five arrays of n entries) on the emulator ==
    the oracle's loop: sizes that are not multiples of the 16-bins-per-thread / 16 384-bins-per-round shape, none / all /
    sparse finished; a count that belongs to another step is refused."""
    import ctypes
    rng = np.random.RandomState(11)
    for E, p in ((1, 1.0), (17, 0.5), (1000, 0.1), (16384, 0.11), (16385 + 77, 0.11), (40000, 0.0), (333, 1.0)):
        done = (rng.rand(E) < p).astype(np.uint8) * rng.randint(1, 3, size=E).astype(np.uint8)
        ret, ratio = rng.rand(E) * 10, rng.rand(E)
        ln, cnt = rng.randint(1, 60, size=E).astype(np.int32), rng.randint(0, 50, size=E).astype(np.int32)
        n = int(np.count_nonzero(done))
        nb = (32 + 28 * E + 4 + 7) // 8 * 8
        got = {}
        for name, lib in (("oracle", oracle.lib()), ("emu", emu.lib())):
            lib.bpp_gather_finished.argtypes = None

            def call(count):
                dev, host = np.zeros(nb // 8, np.float64).view(np.uint8), np.full(nb // 8, -1.0).view(np.uint8)
                rc = lib.bpp_gather_finished(ctypes.c_void_p(done.ctypes.data), ctypes.c_void_p(ret.ctypes.data),
                                             ctypes.c_void_p(ratio.ctypes.data), ctypes.c_void_p(ln.ctypes.data),
                                             ctypes.c_void_p(cnt.ctypes.data), ctypes.c_int32(E), ctypes.c_void_p(dev.ctypes.data),
                                             ctypes.c_void_p(host.ctypes.data), ctypes.c_int32(count), None)
                return rc, host
            rc, host = call(n)
            assert rc == 0, lib.bpp_last_error()
            assert host[:4].view("<i4")[0] == n
            b = host[32:32 + 28 * n]
            got[name] = dict(ret=b[:8 * n].view("<f8").copy(), ratio=b[8 * n:16 * n].view("<f8").copy(), ln=b[16 * n:20 * n].view("<i4").copy(),
                             cnt=b[20 * n:24 * n].view("<i4").copy(), bins=b[24 * n:28 * n].view("<i4").copy())
            assert call(n + 1 if n < E else n - 1)[0] != 0          # the caller's `done` is not this step's
            # the EAGER form (BPP_GATHER_ENQUEUE_ONLY): arrays laid out for E entries straight in `host`, no count to agree with
            host = np.full(nb // 8, -1.0).view(np.uint8)
            rc = lib.bpp_gather_finished(ctypes.c_void_p(done.ctypes.data), ctypes.c_void_p(ret.ctypes.data), ctypes.c_void_p(ratio.ctypes.data),
                                         ctypes.c_void_p(ln.ctypes.data), ctypes.c_void_p(cnt.ctypes.data), ctypes.c_int32(E), None,
                                         ctypes.c_void_p(host.ctypes.data), ctypes.c_int32(-1), None)
            assert rc == 0 and host[:4].view("<i4")[0] == n, lib.bpp_last_error()
            b = host[32:]
            eager = dict(ret=b[:8 * n].view("<f8"), ratio=b[8 * E:8 * E + 8 * n].view("<f8"), ln=b[16 * E:16 * E + 4 * n].view("<i4"),
                         cnt=b[20 * E:20 * E + 4 * n].view("<i4"), bins=b[24 * E:24 * E + 4 * n].view("<i4"))
            for f in eager:
                np.testing.assert_array_equal(eager[f], got[name][f], err_msg="eager " + f)
            assert lib.bpp_gather_finished(ctypes.c_void_p(done.ctypes.data), ctypes.c_void_p(ret.ctypes.data), ctypes.c_void_p(ratio.ctypes.data),
                                           ctypes.c_void_p(ln.ctypes.data), ctypes.c_void_p(cnt.ctypes.data), ctypes.c_int32(E), ctypes.c_void_p(host.ctypes.data),
                                           ctypes.c_void_p(host.ctypes.data), ctypes.c_int32(-1), None) != 0      # eager + a device staging buffer: refused
        idx = np.flatnonzero(done)
        np.testing.assert_array_equal(got["oracle"]["bins"], idx)
        np.testing.assert_array_equal(got["oracle"]["ret"], ret[idx])
        np.testing.assert_array_equal(got["oracle"]["cnt"], cnt[idx])
        for f in got["oracle"]:
            np.testing.assert_array_equal(got["emu"][f], got["oracle"][f], err_msg=f)
