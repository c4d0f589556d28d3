"""The static LDS-DMA schedule of the ping-pong GEMM (carel_vae_amd/csrc/gemm_pp.hip): the generator's own hazard replay
(RAW: a unit is waited for by every wave one phase before its first read; WAR: a slot is refilled two phases after its
last read) for many K lengths, the vmcnt tables against an event-by-event recount, and the committed C++ tables against
the generator (so an edit of either side alone fails here, on CPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_sched as G  # noqa: E402


def _all_configs():
    """(npn, stages, lead cap, wide, war): the fine schedules and the wide-phase ones the kernel runs by default"""
    return [(npn, st, G.LEADS[npn], False, 2) for npn, st in G.CONFIGS.items()] + [(npn, st, None, True, 1) for npn, st in G.WIDE_CONFIGS.items()]


def test_schedules_have_no_hazards_and_tables_reproduce_every_wait():
    for (npn, stages, cap, wide, war) in _all_configs():
        s, tabs, ntail = G.describe(npn, stages, cap, wide, war)
        while ntail > 1 and tabs[ntail] == tabs[0]:
            ntail -= 1
        NP = s["NP"]
        pro12 = G.prologue_of(s, 12)
        for nk in range(2, 64):
            G.check(s, nk)
            w = G.wait_counts(s, nk)
            for t in range(nk):
                R = nk - t
                tab = tabs[R] if R <= ntail else tabs[0]
                assert [w[t * NP + p] for p in range(NP)] == tab, (npn, wide, nk, t)
            assert G.prologue_of(s, nk) == pro12


def test_wide_schedule_restages_one_phase_after_the_last_read_and_keeps_a_k_tile_in_flight():
    """war = 1 is only legal because the kernel drains lgkmcnt before the barrier that ends a load segment; the payoff is a
    lead of S * NP - 1 phases, i.e. every unit is issued at least one whole K tile before the wait that retires it."""
    for npn, stages in G.WIDE_CONFIGS.items():
        s = G.make(npn, stages, None, True, 1)
        assert s["NP"] == npn and all(d == stages * npn - 1 for d in s["lead"].values())
        assert min(s["lead"].values()) - 1 >= npn            # flight (issue -> wait) of at least NP phases = one K tile
        src = open(os.path.join(ROOT, "carel_vae_amd", "csrc", "gemm_pp.hip")).read()
        assert 'if constexpr (WIDE) asm volatile("s_waitcnt lgkmcnt(0)"' in src


def test_every_unit_is_two_dma_instructions_and_lds_fits():
    for npn, stages in list(G.CONFIGS.items()) + list(G.WIDE_CONFIGS.items()):
        for bpart in (12288, 16384):
            if npn == 3 and bpart == 16384:
                continue                                   # NN form is built for npn 1, 2
            assert stages * (32768 + npn * bpart) <= 160 * 1024


def test_committed_tables_match_generator():
    inc = os.path.join(ROOT, "carel_vae_amd", "csrc", "gemm_pp_sched.inc")
    assert open(inc).read() == G.header()


def test_epilogue_input_loads_ride_behind_the_prologue_without_moving_any_retirement():
    """gemm_pp.hip (round 3) requests its epilogue inputs between the prologue's DMA units and K tile 0's and widens the prologue wait and
    tile 0's waits by their instruction count: instruction-level replay -- every unit retires at the same wait as without the loads,
    and no wait of tile 0 retires a load (they get a K tile of flight)."""
    for npn, st in G.WIDE_CONFIGS.items():
        s = G.make(npn, st, None, True, 1)
        for nk in (3, 4, 5, 6, 7, 12, 13, 36, 48):
            if nk <= 2:
                continue
            for ne in (6, 12, 24):
                assert G.first_tile_waits_target_prologue(s, nk, ne)
