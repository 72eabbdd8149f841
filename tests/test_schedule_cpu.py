"""CPU: the pure parts of the per-checkpoint precision-schedule calibration (egohmr_amd/fused.py): the ladder of candidate k, the
search over it, and the cache key's sensitivity.  The measured part (the sampling loops) is covered by tests/test_gpu_schedule.py."""
import math

from egohmr_amd.fused import FusedSampler


def test_k_ladder_is_ascending_geometric_and_ends_at_T():
    for T in (5, 10, 50, 100, 1000):
        lad = FusedSampler._k_ladder(T)
        assert lad[-1] == T and lad == sorted(set(lad)) and lad[0] >= 2
        assert all(b / a <= 1.6 for a, b in zip(lad, lad[1:-1]) if a >= 4)      # ratio ~4/3 (integer rounding at the low end)
        assert len(lad) <= 2 + math.ceil(math.log(T / 2) / math.log(4 / 3)) + 1
    assert FusedSampler._k_ladder(100, floor=19)[0] == 19                         # guided loops: never below the guided window + margin
    assert FusedSampler._k_ladder(5, floor=19) == [5]                             # nothing below T qualifies: every step f32-grade


def _search(err_a, err_b, T=100, bar=5e-6):
    lad = FusedSampler._k_ladder(T)
    calls = {"a": [], "b": []}
    idx = FusedSampler.pick_k(lad, lambda k: (calls["a"].append(k), err_a(k))[1], lambda k: (calls["b"].append(k), err_b(k))[1], bar)
    return lad[idx], calls


def test_pick_k_contracting_network_takes_a_small_k():
    e = lambda k: 3e-3 * 0.25 ** k + 1.5e-6            # errors die geometrically: the insensitive synthetic denoiser
    k, calls = _search(e, e)
    assert k == 5 and len(calls["a"]) <= 5 and calls["b"] == [5]            # 3e-3 / 4^5 + 1.5e-6 = 4.4e-6 <= 5e-6


def test_pick_k_error_carrying_network_ends_at_T():
    e = lambda k: 0.0 if k >= 100 else 9e-3 * (1 - k / 110)      # errors are carried: the trained-like denoiser
    k, calls = _search(e, e)
    assert k == 100 and calls["b"] == []               # draw B is never run when only k = T passes draw A


def test_pick_k_second_draw_can_only_raise_k():
    ea = lambda k: 0.0 if k >= 100 else 1e-4 / k       # draw A passes from k = 20 (rung 20)
    eb = lambda k: 0.0 if k >= 100 else 3e-4 / k       # draw B is three times worse: passes from k = 60 (first rung: 63)
    k, calls = _search(ea, eb)
    assert k == 63 and calls["b"] == [20, 27, 36, 47, 63]
    k2, _ = _search(ea, lambda k: 1.0)                 # a draw that never passes below T: every step f32-grade
    assert k2 == 100


def test_pick_k_non_monotone_blip_is_conservative():
    """A bisection over a non-monotone error curve may land above the true minimum - never below a k that fails draw A at that k."""
    def ea(k):
        return 0.0 if k >= 100 else (1e-5 if k in (27, 36) else 1e-6 if k >= 15 else 1e-3)
    k, _ = _search(ea, ea)
    assert ea(k) <= 5e-6 and k in (15, 20, 47)
