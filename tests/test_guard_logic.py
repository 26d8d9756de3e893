"""CPU tier: the DECISIONS of the sweep clock's guard (csrc/cw_host.cpp: cwh_guard_step) on synthetic traces -- no GPU, no events, no clock: a
model card answers every sampled sweep with a time, the state machine moves the rate.  What a live card did with it is tests/test_zz_perf_floors.py's
business; here are the properties that must hold whatever the card does: the rate stays in [floor, ceiling], a saturated rate is left within a
few samples, nothing moves UP more often than once per 32 samples, a trial that does not pay is undone and its kind backs off (doubling, capped at
2 048), a disturbance that has passed is recovered from, and traces that give no evidence (alternating, one late in ten) move nothing.
Also here: the other HIP-free helpers that moved to cw_host.cpp this round (dense view of a slot record, checkpoint section sizes), so that the
sanitizer run of tests/test_sanitizers.py covers them."""
import ctypes as C

import numpy as np
import pytest

from hostlib import host_lib

WAVES, JOBS, BESIDE_MS = 1024, 338.0, 0.010          # the headline sweep: 65 536 envs x 21 168 B in 4-KiB pieces over 1 024 waves; ~10 us beside the jobs
FLOOR, CEILING, NOTCH = 5.0, 7.7, 0.2


class Card:
    """What a sampled sweep takes at a rate: its schedule up to `capacity` TB/s, the saturated regime beyond it (12 % over what the capacity rate takes,
    profiles/r04_clock.txt), plus whatever the scenario adds."""

    def __init__(self, lib, capacity):
        self.lib, self.capacity = lib, capacity

    def scheduled(self, rate):
        p, ph, pb = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.cwh_sweep_periods(rate, WAVES, 0.4, 0.75, C.byref(p), C.byref(ph), C.byref(pb))
        return self.lib.cwh_guard_scheduled_ms(JOBS, p.value, pb.value, BESIDE_MS)

    def ms(self, rate):
        return self.scheduled(rate) if rate <= self.capacity + 1e-9 else 1.12 * self.scheduled(self.capacity)


def run(lib, L, card, rate0, n, extra=lambda i, rate: 0.0, factor=lambda i, rate: 1.0):
    """n sampled sweeps -> (guard state, list of (sample index, action, rate after)), checking the invariants on the way"""
    g = L.cwh_guard()
    lib.cwh_guard_init(C.byref(g), rate0)
    moves, last_up = [], -10 ** 9
    for i in range(n):
        before = g.rate
        a = lib.cwh_guard_step(C.byref(g), card.ms(g.rate) * factor(i, g.rate) + extra(i, g.rate), card.scheduled(g.rate))
        assert FLOOR - 1e-9 <= g.rate <= max(CEILING, rate0) + 1e-9, (i, g.rate)
        assert 16 <= g.probe_need <= 2048 and 64 <= g.recover_need <= 2048
        if a == L.CWH_GUARD_NONE or a == L.CWH_GUARD_TRIAL_KEPT:
            assert g.rate == before
        else:
            assert abs(abs(g.rate - before) - NOTCH) < 1e-9 or (a == L.CWH_GUARD_TRIAL_UP and g.rate == CEILING), (i, a, before, g.rate)
        if a == L.CWH_GUARD_TRIAL_UP:
            assert i - last_up >= 32, (i, last_up)          # a move UP at most once per 32 samples: it needs 32 samples at the rate it leaves
            last_up = i
        if a != L.CWH_GUARD_NONE:
            moves.append((i, a, round(g.rate, 3)))
    return g, moves


@pytest.fixture(scope='module')
def hl():
    return host_lib()


def test_steady_on_time_at_the_ceiling_never_moves(hl):
    L, lib = hl
    g, moves = run(lib, L, Card(lib, capacity=8.5), CEILING, 5000)
    assert moves == [] and g.rate == CEILING and g.slowdowns == 0 and g.probes == 0


def test_one_late_sample_in_ten_at_the_edge_is_not_a_reason(hl):
    """at the write path's edge one launch in ten is 7-12 % late by itself: never three in a row, so no slowdown -- and the good count still builds up"""
    L, lib = hl
    g, moves = run(lib, L, Card(lib, capacity=8.5), CEILING, 5000, factor=lambda i, r: 1.10 if i % 10 == 9 else 1.0)
    assert moves == [] and g.slowdowns == 0
    g, moves = run(lib, L, Card(lib, capacity=8.5), 7.3, 3000, factor=lambda i, r: 1.10 if i % 10 == 9 else 1.0)
    assert g.rate == CEILING and g.slowdowns == 0          # ... below the ceiling the probes climb through the noise (each one pays: shorter sweeps)
    assert [m[1] for m in moves] == [L.CWH_GUARD_TRIAL_UP, L.CWH_GUARD_TRIAL_KEPT] * 2


def test_alternating_good_and_late_gives_no_evidence_either_way(hl):
    L, lib = hl
    g, moves = run(lib, L, Card(lib, capacity=8.5), 7.3, 4000, factor=lambda i, r: 1.12 if i % 2 else 1.0)
    assert moves == [] and g.rate == 7.3 and g.late <= 1 and g.good <= 1


def test_a_saturated_start_is_left_within_a_few_samples_each_notch(hl):
    """started far above what the card takes (CW_TUNE_RATE_TBS=9.0: the GPU tier's test) every sample is ~12 % + late: three in a row, a notch down, again"""
    L, lib = hl
    g, moves = run(lib, L, Card(lib, capacity=7.4), 9.0, 200)
    downs = [m for m in moves if m[1] == L.CWH_GUARD_SLOWDOWN]
    assert [m[0] for m in downs[:4]] == [2, 5, 8, 11] and g.slowdowns == len(downs) >= 8
    # the first rate at or under the capacity keeps its schedule: it holds there (trials back UP to rates that do not pay are undone, below)
    assert min(m[2] for m in moves) >= 7.4 - 1e-9 and 7.4 - 1e-9 <= g.rate <= 7.6 + 1e-9


def test_probes_that_do_not_pay_are_undone_and_back_off(hl):
    """capacity 7.4: from 7.0 two probes pay (7.2, 7.4: rate_top follows), the third (7.6) is in the saturated regime -- on time by no measure, but
    the verdict is the MEAN: not shorter -> undone, and the next probe waits twice as long, and again, up to 2 048 samples"""
    L, lib = hl
    g, moves = run(lib, L, Card(lib, capacity=7.4), 7.0, 12000)
    kinds = [m[1] for m in moves]
    assert kinds[:4] == [L.CWH_GUARD_TRIAL_UP, L.CWH_GUARD_TRIAL_KEPT] * 2
    assert kinds[4:] == [L.CWH_GUARD_TRIAL_UP, L.CWH_GUARD_TRIAL_UNDONE] * ((len(kinds) - 4) // 2) and len(kinds) >= 14
    assert abs(g.rate_top - 7.4) < 1e-9 and abs(g.rate - 7.4) < 1e-9 and g.slowdowns == 0 and g.recovering == 0
    ups = [m[0] for m in moves if m[1] == L.CWH_GUARD_TRIAL_UP][2:]
    gaps = np.diff(ups)
    assert all(b >= a for a, b in zip(gaps, gaps[1:])) and gaps[0] < 100 and gaps[-1] >= 2048      # the waits double ...
    assert g.probe_need == 2048                                                                   # ... and stop doubling at the cap
    assert all(m[2] <= 7.6 + 1e-9 for m in moves)          # one notch beyond the best rate known, never two


def test_a_constant_delay_between_the_sweeps_does_not_walk_the_clock_down_for_good(hl):
    """a second engine's step kernel between the sweeps: every sweep a constant 16 us late (8 % at 7.7 TB/s).  Judged by the schedule alone every rate down to
    5.7 TB/s is "late"; the trials back up are judged by whether they PAY (shorter sweeps), and a rate a trial accepted is measured against what it delivered
    then -- so the clock comes back to the best rate known and stays"""
    L, lib = hl
    card = Card(lib, capacity=8.5)
    g, moves = run(lib, L, card, CEILING, 6000, extra=lambda i, r: 0.016)
    assert g.slowdowns >= 1 and min(m[2] for m in moves) >= FLOOR - 1e-9
    assert abs(g.rate - CEILING) < 1e-9, (g.rate, moves[-6:])
    tail = [m for m in moves if m[0] > 4000]
    assert tail == [], tail                                 # ... at rest: its yardstick is what the accepted trial delivered (ref_ms), not the schedule
    assert g.ref_ms > card.scheduled(CEILING)


def test_a_disturbance_that_passes_is_recovered_from(hl):
    """2 000 samples of sweeps 15 % late whatever the rate (another process on the card): the clock gives way notch by notch, down to the floor if it
    lasts; afterwards it climbs back, one paying trial per notch, to the best rate known -- and a disturbance DURING a trial ends it early (three far off)"""
    L, lib = hl
    card = Card(lib, capacity=8.5)
    g, moves = run(lib, L, card, 7.5, 7000, factor=lambda i, r: 1.15 * 1.01 if 500 <= i < 2500 else 1.0)
    before = [m for m in moves if m[0] < 500]
    during = [m for m in moves if 500 <= m[0] < 2500]
    after = [m for m in moves if m[0] >= 2500]
    assert [m[1] for m in before] == [L.CWH_GUARD_TRIAL_UP, L.CWH_GUARD_TRIAL_KEPT] and before[-1][2] == CEILING      # (it had probed up to the ceiling before)
    assert during and all(m[1] == L.CWH_GUARD_SLOWDOWN for m in during) and abs(min(m[2] for m in during) - (FLOOR + 0.1)) < 0.11
    assert all(m[1] in (L.CWH_GUARD_TRIAL_UP, L.CWH_GUARD_TRIAL_KEPT) for m in after)
    assert abs(g.rate - CEILING) < 1e-9 and abs(g.rate_top - CEILING) < 1e-9 and g.recover_need == 64
    # a burst in the middle of a trial: three samples far off end it at once
    g2, moves2 = run(lib, L, card, 7.3, 200, factor=lambda i, r: 1.40 if 70 <= i < 73 else 1.0)
    assert (63, L.CWH_GUARD_TRIAL_UP, 7.5) in moves2 and (72, L.CWH_GUARD_TRIAL_UNDONE, 7.3) in moves2 and g2.probe_need == 128


def test_periods_and_schedule_of_the_headline_sweep(hl):
    """the numbers DESIGN 4.3 quotes: 7.7 TB/s on 1 024 waves is a 545-ns period (872 sixteenths of a tick), heads 0.4 / 0.75 TB/s slower, never under the floor;
    unclocked = all zero; the schedule is jobs x period + the busy head + the launch's own cost"""
    L, lib = hl
    p, ph, pb = C.c_int32(), C.c_int32(), C.c_int32()
    lib.cwh_sweep_periods(7.7, 1024, 0.4, 0.75, C.byref(p), C.byref(ph), C.byref(pb))
    assert (p.value, ph.value, pb.value) == (872, 919, 966)
    lib.cwh_sweep_periods(5.2, 1024, 0.4, 0.75, C.byref(p), C.byref(ph), C.byref(pb))
    assert ph.value == pb.value == int(1024 * 4096 / 5.0e12 * 1e9 * 1.6 + 0.5) and p.value < ph.value
    lib.cwh_sweep_periods(0.0, 1024, 0.4, 0.75, C.byref(p), C.byref(ph), C.byref(pb))
    assert (p.value, ph.value, pb.value) == (0, 0, 0)
    ms = lib.cwh_guard_scheduled_ms(338.0, 872, 966, 0.010)
    assert abs(ms - ((338 * 545.0 + 64 * (966 - 872) / 1.6) * 1e-6 + 0.010)) < 1e-12 and 0.19 < ms < 0.21


def test_dense_view_of_a_slot_record(hl):
    L, lib = hl
    pos = np.array([3, 0xFFFF, 0xFFFE, 24, 7, 7000, 11, 0], np.uint16)      # slot 1 gone, slot 2 held, slot 5 beyond the grid (ignored)
    codes = 0x87654321
    grid = np.full(25, 99, np.uint8)
    lib.cwh_slots_to_grid(pos.ctypes.data_as(C.c_void_p), codes, 25, grid.ctypes.data_as(C.c_void_p))
    want = np.zeros(25, np.uint8)
    want[3], want[24], want[7], want[11], want[0] = 1, 4, 5, 7, 8
    assert np.array_equal(grid, want)


@pytest.mark.parametrize('n,k,la', [(1, 0, 0), (4096, 0, 1), (65536, 3, 4), (1 << 20, 64, 0)])
def test_checkpoint_section_sizes(hl, n, k, la):
    L, lib = hl
    sizes = (C.c_size_t * L.CWH_CKPT_SECTIONS)()
    total = C.c_uint64()
    assert lib.cwh_ckpt_section_bytes(n, k, la, sizes, C.byref(total)) == L.CWH_CKPT_SECTIONS == 22
    s = list(sizes)
    assert sum(s) == total.value
    per_env = 16 * 4 + 4 + 2 + 2 + 4 + 624 * 4 + 4 + k * 18 + 4 + 1 + 2 + 2 + 4 + 4
    assert total.value == n * per_env + 40 + (la and n * (48 * la + 4))                # (la records of 48 bytes per env and the ring's control word)
    assert s[8] == n * 2496 and s[10] == n * k * 18 and s[17] == 40 and (s[18:] == [0] * 4) == (not la)
    assert lib.cwh_ckpt_section_bytes(n, k, la, None, None) == 22        # (sizes and total are optional)


def test_refill_period_follows_the_slow_resets(hl):
    """cwh_la_adapt (csrc/cw_host.cpp; cw_engine.cpp feeds it the count of slow-path resets a pinned word reports): a random policy (no slow resets) keeps the
    static period; a policy whose episodes end early (slow resets every period) halves it down to 8 and holds it there; once they stop it climbs back by
    doublings, eight quiet refills apart; a few slow resets (under a 32nd of the period's steps) count as quiet, some (between a 32nd and an eighth) hold the period"""
    L, lib = hl
    q = C.c_int32(0)
    p = 64
    for _ in range(100):
        p = lib.cwh_la_adapt(p, 64, 0, C.byref(q))
    assert p == 64
    seen = []
    for _ in range(6):
        p = lib.cwh_la_adapt(p, 64, 100, C.byref(q))
        seen.append(p)
    assert seen == [32, 16, 8, 8, 8, 8] and q.value == 0
    seen = []
    for i in range(40):
        p = lib.cwh_la_adapt(p, 64, 0 if i % 9 else 1, C.byref(q))           # (1 slow reset in a period of >= 32 steps is quiet; at 8 and 16 steps it holds the period)
        seen.append(p)
    assert seen[-1] == 64 and sorted(seen) == seen and set(seen) == {8, 16, 32, 64}
    ups = [i for i in range(1, 40) if seen[i] > seen[i - 1]]
    assert all(b - a >= 8 for a, b in zip(ups, ups[1:])), ups
    q.value = 0
    assert lib.cwh_la_adapt(32, 64, 4, C.byref(q)) == 32 and lib.cwh_la_adapt(32, 64, 5, C.byref(q)) == 16        # (an eighth of the period's steps is the line)
    assert lib.cwh_la_adapt(8, 64, 10 ** 9, C.byref(q)) == 8 and lib.cwh_la_adapt(48, 48, 0, C.byref(q)) == 48
