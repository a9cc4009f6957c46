"""What a transmission of each protocol looks like TO ITS DECODER (rows of bits; free fields random, fixed fields as the
decoder's source documents them, integrity fields left to tools/gen_protocol_frames.py's search).  Test infrastructure."""
from tools.gen_protocol_frames import from_int, rnd_bits, template


@template("rubicson")
def _(rng):  # [id 8][battery 1][0][channel 2][temp 12][1111][crc 8], three equal rows (src/devices/rubicson.c:23-37)
    row = rnd_bits(rng, 8) + [1, 0] + from_int(int(rng.integers(0, 3)), 2) + from_int(int(rng.integers(0, 400)), 12) + [1, 1, 1, 1] + [0] * 8
    return [row] * 3


@template("nexus")
def _(rng):  # [id 8][battery][test 0][channel 2, not 3][temp 12][1111][humidity 8] (src/devices/nexus.c:17-40)
    row = rnd_bits(rng, 8) + [1, 0] + from_int(int(rng.integers(0, 3)), 2) + from_int(int(rng.integers(0, 400)), 12) + [1, 1, 1, 1] \
        + from_int(int(rng.integers(20, 96)), 8)
    return [row] * 3


@template("prologue")
def _(rng):  # [type 1001][id 8][battery][button][channel 2][temp 12][humidity 8] (src/devices/prologue.c:20-45)
    row = [1, 0, 0, 1] + rnd_bits(rng, 8) + [1, 0] + from_int(int(rng.integers(0, 3)), 2) + from_int(int(rng.integers(0, 400)), 12) \
        + from_int(int(rng.integers(20, 96)), 8)
    return [row] * 4


@template("generic_remote")
def _(rng):  # 24 bits (read inverted) and a closing 1 (src/devices/generic_remote.c:20-45)
    return [rnd_bits(rng, 24) + [1]]
