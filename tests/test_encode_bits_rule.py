"""The parallel boolean encoder does not walk the bytes as encodeBits does (rle.go:97-139): it decides from a byte's
neighbours, and from whether the stretch of equal bytes before it was a run of repeats, where runs start.  This
states that rule in Python and holds it against the checker's walk on inputs of few distinct bytes, where
stretches of every kind meet one another."""
import numpy as np

from oracle import pqoracle as O


def _uvarint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def encode_bits_by_rule(s):
    n = len(s)
    if n == 0:
        return (1).to_bytes(4, "little") + b"\x00"
    starts, repeats = [], []
    stretch_repeats = False  # the stretch of equal bytes that holds byte i - 1 is a run of repeats
    for i in range(n):
        c = s[i]
        same_before = i > 0 and s[i - 1] == c
        same_after = i + 1 < n and s[i + 1] == c
        two_zeros_before = i >= 2 and s[i - 1] == 0 and s[i - 2] == 0
        first_of_stretch = not same_before
        rep = first_of_stretch and same_after and (c == 0 or (c == 0xFF and (i == 0 or two_zeros_before)))
        if n == 1 and c in (0, 0xFF):
            rep = True  # rle.go:98-104
        behind_repeats = False
        if first_of_stretch:
            behind_repeats = stretch_repeats
            stretch_repeats = rep
        in_other_stretch = c not in (0, 0xFF) and (same_before or same_after)
        if i == 0 or rep or behind_repeats or in_other_stretch:
            starts.append(i)
            repeats.append(rep)
    body = bytearray()
    for k, p in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else n
        if repeats[k]:
            body += _uvarint((8 * (e - p)) << 1) + bytes([s[p]])
        else:
            body += _uvarint(((e - p) << 1) | 1) + bytes(s[p:e])
    return len(body).to_bytes(4, "little") + bytes(body)


def test_the_rule_is_the_walk():
    rng = np.random.default_rng(0)
    for trial in range(4000):
        n = int(rng.integers(0, 40))
        alphabet = [0, 0xFF, 0x55, 1, 0xFE][: int(rng.integers(1, 6))]
        s = bytes(int(alphabet[int(x)]) for x in rng.integers(0, len(alphabet), size=n))
        assert encode_bits_by_rule(s) == bytes(O.rle_encode_boolean(np.frombuffer(s, np.uint8))), s.hex()
