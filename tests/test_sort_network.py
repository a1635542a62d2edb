"""The wiring of the register sorting network of csrc/stage.hpp (sort_flip_step / sort_half_steps /
sort_merges), re-stated lane by lane in numpy: the same partner masks, register pairings and
keep-min rules must sort every input for every (lanes, values-per-lane) shape the kernels
instantiate.  (The device code itself is exercised by the GPU parity tests; this pins the design.)"""
import numpy as np
import pytest


def network_sort(values, lanes, kpl):
    """values: [lanes, kpl] array, element index = lane * kpl + r (stage.hpp blocked layout)."""
    v = np.array(values, dtype=float).reshape(lanes, kpl).copy()
    lane = np.arange(lanes)

    def flip(k):
        if k <= kpl:                                           # partners inside a lane: r ^ (k - 1)
            for r in range(kpl):
                q = r ^ (k - 1)
                if r < q:
                    a, b = v[:, r].copy(), v[:, q].copy()
                    v[:, r], v[:, q] = np.minimum(a, b), np.maximum(a, b)
        else:
            ml = k // kpl - 1                                  # lane ^ ml, register kpl - 1 - r
            keep_min = (lane & ((ml + 1) >> 1)) == 0
            p = np.stack([v[lane ^ ml, kpl - 1 - r] for r in range(kpl)], axis=1)
            lo, hi = np.minimum(v, p), np.maximum(v, p)
            v[:] = np.where(keep_min[:, None], lo, hi)

    def half(j):
        while j >= 1:
            if j < kpl:
                for r in range(kpl):
                    if (r & j) == 0:
                        a, b = v[:, r].copy(), v[:, r | j].copy()
                        v[:, r], v[:, r | j] = np.minimum(a, b), np.maximum(a, b)
            else:
                ml = j // kpl
                keep_min = (lane & ml) == 0
                p = v[lane ^ ml, :]
                lo, hi = np.minimum(v, p), np.maximum(v, p)
                v[:] = np.where(keep_min[:, None], lo, hi)
            j //= 2

    k = 2
    while k <= lanes * kpl:
        flip(k)
        half(k // 4)
        k *= 2
    return v.reshape(-1)


@pytest.mark.parametrize("lanes,kpl", [(8, 4), (8, 8), (64, 2), (64, 4), (64, 8)])
def test_network_sorts(lanes, kpl):
    rng = np.random.default_rng(lanes * 100 + kpl)
    n = lanes * kpl
    for trial in range(60):
        m = int(rng.integers(1, n + 1))
        x = np.full(n, np.inf)
        kind = trial % 4
        if kind == 0:
            x[:m] = rng.normal(0, 10, m)
        elif kind == 1:
            x[:m] = rng.integers(-3, 4, m)                      # many ties, signed zeros
        elif kind == 2:
            x[:m] = np.sort(rng.normal(0, 1, m))[::-1]          # reversed
        else:
            x[:m] = rng.choice([0.0, 1.0], m)                   # 0-1 inputs (zero-one principle samples)
        x = x[rng.permutation(n)] if kind != 2 else x
        got = network_sort(x, lanes, kpl)
        assert np.array_equal(got, np.sort(x)), (lanes, kpl, trial)


def test_xor_partner_patterns_used_by_the_device_code():
    """Every cross-lane mask the network needs is one the device fetch implements: DPP for 1, 2, 3, 7,
    8, 15, ds_swizzle below 32, ds_bpermute for 32..63 (wave.hpp lane_xor_fetch)."""
    for lanes, kpl in [(8, 4), (8, 8), (64, 2), (64, 4), (64, 8)]:
        masks = set()
        k = 2
        while k <= lanes * kpl:
            if k > kpl:
                masks.add(k // kpl - 1)
            j = k // 4
            while j >= 1:
                if j >= kpl:
                    masks.add(j // kpl)
                j //= 2
            k *= 2
        assert all(0 < m < lanes for m in masks), (lanes, kpl, masks)
