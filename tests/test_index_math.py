"""Host-side check of the reciprocal division the P16 GEMM kernel uses for its index arithmetic
(matcha-tts-24k_amd/csrc/device_utils.h fdiv, gemm_p16.hip rcp32): q = (n * rcp) >> 32 with rcp = floor(2^32 / d) + 1 equals
n // d for every n with n * d < 2^32 -- the bound launch_gemm_p16 checks before it hands out a reciprocal (otherwise the kernel
divides)."""
import random


def rcp32(d):
    return 0 if d <= 1 else ((1 << 32) // d + 1) & 0xFFFFFFFF


def fdiv(n, d, rcp):
    return (n * rcp) >> 32 if rcp else (n if d <= 1 else n // d)


def test_reciprocal_division_is_exact_inside_its_bound():
    rng = random.Random(7)
    for d in list(range(1, 700)) + [rng.randrange(700, 1 << 16) for _ in range(2000)]:
        r = rcp32(d)
        top = ((1 << 32) - 1) // d            # largest n with n * d < 2^32
        probes = {0, 1, d - 1, d, d + 1, top, top - 1, max(top - d, 0)} | {rng.randrange(0, top + 1) for _ in range(40)}
        probes |= {k * d - 1 for k in (1, 2, 3, top // d) if 0 <= k * d - 1 <= top} | {k * d for k in (1, 2, top // d) if k * d <= top}
        for n in probes:
            assert fdiv(n, d, r) == n // d, (n, d)


def test_the_decoder_shapes_are_inside_the_bound():
    # config #2: 32 utterances x 322 rows, config #3 per rank the same, B = 64; serving: 128 utterances of 4000 frames
    for B, T in [(32, 322), (64, 322), (32, 640), (128, 4000)]:
        assert (B * T + 256) * T < 1 << 32
