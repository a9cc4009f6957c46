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


@template("s3318p")
def _(rng):  # 42 bits: two leading, then [id 8][flags][channel][temp 12 (nibbles swapped)][humidity 8][crc 4 and flags]
    row = [0, 0] + rnd_bits(rng, 8) + [0, 0] + from_int(int(rng.integers(0, 3)), 2) + from_int(int(rng.integers(0x340, 0x7ff)), 12) \
        + from_int(int(rng.integers(0x20, 0x96)), 8) + rnd_bits(rng, 8)
    return [[], []] + [row] * 4 if False else [row] * 4


@template("tfa_pool")
def _(rng):  # 28 bits: [checksum 4][device 8][temp 12][channel 2][battery 1][0]
    return [rnd_bits(rng, 4) + rnd_bits(rng, 8) + from_int(int(rng.integers(0, 400)), 12) + rnd_bits(rng, 3) + [0]] * 7


@template("thermopro_tp11")
def _(rng):  # 32 bits: [device 12][temp 12][digest 8]
    return [rnd_bits(rng, 12) + from_int(int(rng.integers(200, 900)), 12) + [0] * 8 + [0]] * 2


@template("kerui")
def _(rng):  # 25 bits (read inverted): [id 20][command 4][1]; command one of a e 7 b 5 f
    cmd = [0xa, 0xe, 0x7, 0xb, 0x5, 0xf][int(rng.integers(0, 6))]
    inv = lambda bits: [1 - b for b in bits]
    return [inv(rnd_bits(rng, 20) + from_int(cmd, 4)) + [1]] * 9


@template("quhwa")
def _(rng):  # 18 bits (read inverted): [id 14][11][11]
    inv = lambda bits: [1 - b for b in bits]
    return [inv(rnd_bits(rng, 14) + [1, 1] + [1, 1])] * 5


@template("waveman")
def _(rng):  # 25 bits: every other bit one; state nibble e or 6
    def nib(v):
        out = []
        for k in range(4):
            out += [1, 0 if (v >> k) & 1 else 1]
        return out
    return [nib(int(rng.integers(0, 16))) + nib(int(rng.integers(0, 16))) + nib([0xe, 0x6][int(rng.integers(0, 2))]) + [0]]


@template("secplus_v1", raw=True)
def _(rng):  # two halves of 21 ternary symbols; 0 -> 0001, 1 -> 0011, 2 -> 0111; the first symbol says which half it is
    def half(first):
        sym = [first] + [int(x) for x in rng.integers(0, 3, 20)]
        out = []
        for s in sym:
            out += [[0, 0, 0, 1], [0, 0, 1, 1], [0, 1, 1, 1]][s]
        return out
    return [half(0), half(2)]


@template("ambient_f007th")
def _(rng):  # [00000001][45][id][battery, channel, temp 12][humidity][digest]
    return [from_int(0x01, 8) + from_int(0x45, 8) + rnd_bits(rng, 8) + [0] + from_int(int(rng.integers(0, 8)), 3)
            + from_int(int(rng.integers(400, 1500)), 12) + from_int(int(rng.integers(10, 99)), 8) + [0] * 8 + [0, 0]]


@template("wt450")
def _(rng):  # 36 bits: [1100][house 4][channel 2][..][battery][humidity 7][temp 8.4][seq 2][parity 2]
    return [[1, 1, 0, 0] + rnd_bits(rng, 4) + rnd_bits(rng, 2) + [1, 1, 0] + from_int(int(rng.integers(20, 90)), 7)
            + from_int(int(rng.integers(30, 110)), 8) + rnd_bits(rng, 4) + rnd_bits(rng, 4)]


@template("oregon_v1")
def _(rng):  # 32 bits, nibbles reflected: id, channel, three BCD temperature digits, flags, checksum byte
    return [rnd_bits(rng, 4) + [0, 0] + rnd_bits(rng, 2) + from_int(int(rng.integers(0, 10)), 4)[::-1] + from_int(int(rng.integers(0, 10)), 4)[::-1]
            + from_int(int(rng.integers(0, 4)), 4)[::-1] + [0, 0, 0, 0] + rnd_bits(rng, 8)]


@template("lacrosse_tx29")
def _(rng):  # aa aa aa 2d d4 | [9 = 5 nibbles follow][id 6][newbatt][0][temp 3 BCD digits][weak batt + humidity 7][crc 8]
    return [from_int(0xaaaaaa, 24) + from_int(0x2dd4, 16) + [1, 0, 0, 1] + rnd_bits(rng, 6) + [0, 0] + from_int(int(rng.integers(3, 8)), 4)
            + from_int(int(rng.integers(0, 10)), 4) + from_int(int(rng.integers(0, 10)), 4) + [0] + from_int(int(rng.integers(20, 95)), 7) + [0] * 8 + [0] * 4]


@template("steelmate")
def _(rng):  # 72 bits (read inverted and reflected): 00 00 7f preamble ...
    return [from_int(0x00007f, 24) + rnd_bits(rng, 48)]


@template("efergy_e2")
def _(rng):  # 64 bits: [0000....][address 16][flags][current 16][exponent][sum]
    return [from_int(0x0, 4) + rnd_bits(rng, 4) + rnd_bits(rng, 16) + [0, 1, 0, 0, 0, 0, 0, 0] + rnd_bits(rng, 16) + from_int(int(rng.integers(0, 8)), 8)
            + [0] * 8]


@template("acurite_606")
def _(rng):  # 32 bits: [id 8][battery][button][channel 2][temp 12][digest 8]
    return [rnd_bits(rng, 8) + [1, 0] + from_int(int(rng.integers(0, 3)), 2) + from_int(int(rng.integers(0, 400)), 12) + [0] * 8] * 3


@template("thermopro_tp12")
def _(rng):  # 41 bits: [device 8][temp1 low 8][temp1 high 4, temp2 high 4][temp2 low 8][digest 8][1]
    return [rnd_bits(rng, 8) + rnd_bits(rng, 8) + [0, 0] + rnd_bits(rng, 2) + [0, 0] + rnd_bits(rng, 2) + rnd_bits(rng, 8) + [0] * 8 + [1]] * 3


@template("gt_wt_02")
def _(rng):  # 37 bits: [id 8][battery][button][channel 2][temp 12][humidity 7][checksum 6]
    row = rnd_bits(rng, 8) + [0, 0] + from_int(int(rng.integers(0, 3)), 2) + from_int(int(rng.integers(0, 400)), 12) \
        + from_int(int(rng.integers(20, 90)), 7) + [0] * 6
    return [row] * 2


@template("bresser_3ch")
def _(rng):  # 40 bits (read inverted): [id 8][battery][0][channel 2, not 0][temp 12][humidity 8][sum 8]
    inv = lambda bits: [1 - b for b in bits]
    return [inv(rnd_bits(rng, 8) + [0, 0] + from_int(int(rng.integers(1, 4)), 2) + from_int(int(rng.integers(900, 1800)), 12)
                + from_int(int(rng.integers(20, 99)), 8) + [0] * 8)] * 3


@template("ht680")
def _(rng):  # 41 bits: sync 10101, then 18 tristate pairs (00, 10 = open, 11) with five of them always open
    tri = lambda: [[0, 0], [1, 0], [1, 1]][int(rng.integers(0, 3))]
    b = []
    for k in range(18):
        b += tri()
    bits = [1, 0, 1, 0, 1] + b
    def put(byte, mask, val):  # (b[byte] & mask) == val over the 36 payload bits
        for i in range(8):
            if mask & (0x80 >> i):
                bits[5 + byte * 8 + i] = 1 if val & (0x80 >> i) else 0
    put(1, 0xf0, 0xa0)
    put(2, 0x0c, 0x08)
    put(3, 0x30, 0x20)
    put(4, 0xf0, 0xa0)
    return [bits[:41]]
