"""ORACLE (test infrastructure) — the fixed policy's counter-based RNG.

The reference has no RNG at all: every decision is an LLM call.
This 32-bit, stateless generator is a build artefact shared *by definition* (not
by code) between the policy stub that drives the reference loop, the C oracle
and the HIP kernels.  POLICY.md §RNG is the normative text.
"""
M32 = 0xFFFFFFFF
GOLDEN = 0x9E3779B9


def mix32(x: int) -> int:
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def room_key(seed: int, room: int) -> int:
    k = mix32((seed & M32) ^ 0x243F6A88)
    k = mix32(k ^ ((seed >> 32) & M32))
    k = mix32(k ^ (room & M32))
    k = mix32(k ^ ((room >> 32) & M32))
    return k


def turn_key(rkey: int, turn: int) -> int:
    return mix32(rkey ^ ((turn * GOLDEN) & M32))


def deal_key(rkey: int, game: int) -> int:
    """Key of the role deal of a room's `game`-th game (0-based; > 0 only in steady-state mode)."""
    return mix32(rkey ^ 0x44454C31 ^ ((game * GOLDEN) & M32))


def draw(tkey: int, idx: int) -> int:
    """idx = stream*16 + j ; stream 0 = per-player action draw, 1 = role picks."""
    return mix32((tkey + (idx + 1) * GOLDEN) & M32)


def pick(d: int, k: int) -> int:
    """uniform-ish index in [0,k) from a 32-bit draw (multiply-high)."""
    return (d * k) >> 32


def nth_set_bit(mask: int, n: int) -> int:
    """0-based position of the n-th (0-based) set bit, ascending."""
    for _ in range(n):
        mask &= mask - 1
    return (mask & -mask).bit_length() - 1
