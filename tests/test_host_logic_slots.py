"""The band-proportional slots of the batched A*PA2's block-column store (csrc/sweep_logic.hpp SlotGeom; DESIGN "band-proportional
block-column store"): slot k of a pair holds the words [off, off + win) of block k's right-edge column, `off` following the main
diagonal by a fixed-point slope.  The reference keeps a block's own rows (astarpa2/src/block.rs:8-21) -- the window is this build's
layout, so its addressing is pinned here by its invariants, through the very header the kernels compile (exported by the emulator
library of oracle/apa2_emu.cpp)."""
import ctypes as C
import random

import pytest


@pytest.fixture(scope="module")
def slots(oracle):
    oracle.apa2_emu_align(b"ACGT", b"ACGT", oracle.params_simple())  # builds / loads the emulator library
    L = C.CDLL(str(oracle._DIR / "_build" / "libpa_apa2_emu.so"))
    L.pa_emu_slot_off.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_int32]
    L.pa_emu_slot_off.restype = C.c_int32
    L.pa_emu_slot_holds.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32]
    L.pa_emu_slot_holds.restype = C.c_int
    return L


def ratio_of(n, m):  # pa_hip.hip batch_create: floor(m * 2^20 / n), saturated
    return min((m << 20) // n, 0xFFFFFFFF) if n else 0


def shapes():
    rng = random.Random(20260928)
    out = [(10_000, 10_000, 48), (100_000, 100_000, 192), (100_000, 100_000, 64), (10_000_000, 10_000_000, 16_032), (1, 1, 8), (255, 70_000, 16),
           (70_000, 255, 16), (4096, 4096, 64), (4096, 4095, 8), (300, 64, 8), (64, 300, 8)]
    for _ in range(300):
        n = rng.choice([rng.randint(1, 2000), rng.randint(1, 200_000), rng.randint(1, 5_000_000)])
        m = max(1, int(n * rng.choice([1.0, 1.0, 0.9, 1.1, 0.5, 2.0, 0.01, 50.0]))) if rng.random() < 0.8 else rng.randint(1, 3_000_000)
        win = rng.choice([8, 16, 40, 64, 192, 8 * rng.randint(1, 400)])
        out.append((n, m, win))
    return out


def test_window_follows_the_diagonal_and_stays_inside_the_column(slots):
    for n, m, win in shapes():
        ratio = ratio_of(n, m)
        wtot = (m + 63) // 64
        nblk = (n + 255) // 256
        ks = sorted(set(list(range(0, min(nblk, 40) + 1)) + [nblk // 2, max(nblk - 1, 0), nblk]))
        prev = None
        for k in ks:
            off = slots.pa_emu_slot_off(n, m, win, ratio, k)
            if win >= wtot:
                assert off == 0, (n, m, win, k)  # the whole column fits: no window
                continue
            assert 0 <= off <= wtot - win, (n, m, win, k, off)
            if prev is not None:
                assert off >= prev, (n, m, win, k)  # the window only moves down with the columns
            prev = off
            # the word of the main diagonal at the block's right edge lies inside the window (the fixed-point slope may sit one word
            # above the exact one: floor(m 2^20 / n) under-estimates by < n / 2^20 rows over the whole pair)
            if (m << 20) // n > 0xFFFFFFFF:
                continue  # m > 4096 n: the 32-bit slope saturates, the window stays behind the diagonal and such a pair takes the second round
            col = min(256 * k, n)
            exact = min((col * m // n) // 64, wtot - 1)
            assert off <= exact <= off + win, (n, m, win, k, off, exact)
            fixed = min((col * ratio) >> 26, wtot - 1)
            assert exact - 1 - (n >> 26) <= fixed <= exact
            assert off <= fixed < off + win or fixed == wtot - 1


def test_slot_holds_is_containment_in_the_window(slots):
    rng = random.Random(7)
    for n, m, win in shapes()[:120]:
        ratio = ratio_of(n, m)
        wtot = (m + 63) // 64
        nblk = (n + 255) // 256
        for _ in range(20):
            k = rng.randint(0, nblk)
            off = slots.pa_emu_slot_off(n, m, win, ratio, k)
            w0 = rng.randint(0, wtot)
            w1 = rng.randint(w0, wtot)
            want = w0 >= off and w1 <= off + (win if win < wtot else max(win, wtot))
            assert bool(slots.pa_emu_slot_holds(n, m, win, ratio, k, w0, w1)) == want, (n, m, win, k, off, w0, w1)
        # the band of a well-behaved pair: a block's rows around the diagonal, half a window wide, always fit
        if win < wtot and win >= 8:
            for k in range(0, nblk + 1, max(1, nblk // 16)):
                off = slots.pa_emu_slot_off(n, m, win, ratio, k)
                mid = min((min(256 * k, n) * ratio) >> 26, wtot - 1)
                w0, w1 = max(off, mid - win // 4), min(off + win, mid + win // 4 + 1)
                assert slots.pa_emu_slot_holds(n, m, win, ratio, k, w0, w1)
