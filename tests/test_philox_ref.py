"""tests/philox_ref.py against the known-answer vectors of Random123's kat_vectors (philox4x32, 10 rounds)"""
from philox_ref import philox4x32_10


def test_philox4x32_10_known_answers():
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        got = philox4x32_10(*ctr, *key)
        assert tuple(int(v) for v in got) == want
