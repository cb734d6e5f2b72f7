"""The halo rows' slot swizzle of k_conv_fwd_ws (csrc/conv3d_mfma.hip, `halo_key`): slot ^= key(column of the row in the halo box).

A ds_read_b128 of a wave is served in four groups of 16 lanes (MI355X guide, LDS table); a group is conflict-free when its 16 addresses fall
into 16 different 16-byte bank groups ((address / 16) mod 16).  This test simulates the B-fragment read of the MFMA waves - every wave,
column tile, filter row, kw, k-step and halo slot, for both halo boxes (6 x 10 x 18 of the 3-tap convolution, 5 x 9 x 17 of the parity
modes) - with the key tables read from the kernel source, and checks (1) no bank conflicts, (2) the address is `per-lane register + constant`
(what lets the kernel put everything but six registers into the instruction's immediate offset), (3) the producers' side: a DMA piece lands
where the reader looks for it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "fetal-mri-segmentation_amd", "csrc", "conv3d_mfma.hip")).read()

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def _tables():
    m = re.search(r"HKEY = HW == 18 \? (0x[0-9a-f]+)ull.*?: (0x[0-9a-f]+)ull", SRC)
    assert m, "halo_key constants not found in the kernel source"
    return {18: int(m.group(1), 16), 17: int(m.group(2), 16)}


def _key(tab, hw):
    return (tab >> (2 * hw)) & 3


def _lane_w(r, HW):
    return (((r & 15) + 16 - (HW & 15)) & 15) if (r >> 4) else (r & 15)


def _read_address(tab, HW, HH, lane, cw, j, rowoff, kw, ks, slot_base):
    """byte address of lane `lane`'s 16-byte B-fragment piece: voxel (d = cw, h = 2 j + (r >> 4), w = lane_w(r)) of the tile, shifted by the tap"""
    r, hk = lane & 31, lane >> 5
    hwc = _lane_w(r, HW) + kw
    row = (cw * HH + 2 * j + (r >> 4)) * HW + rowoff + hwc
    return slot_base + row * 64 + (((hk ^ _key(tab, hwc)) << 4) ^ (ks << 5))


def test_every_fragment_read_is_conflict_free_and_affine_in_the_tap():
    tabs = _tables()
    for HW, HH, nkw, rows in ((18, 10, 3, [kd * 10 + kh for kd in range(3) for kh in range(3)]), (17, 9, 2, [kd * 9 + kh for kd in range(2) for kh in range(2)])):
        tab = tabs[HW]
        for cw in range(4):
            for j in range(4):
                for rr in rows:
                    for kw in range(nkw):
                        for ks in range(2):
                            for slot in (0, 69 * 1024, 64 * 1024):
                                for g in GROUPS:
                                    banks = set((_read_address(tab, HW, HH, l, cw, j, rr * HW, kw, ks, slot) >> 4) & 15 for l in g)
                                    assert len(banks) == 16, (HW, cw, j, rr, kw, ks, sorted(banks))
        # affine: address(lane, cw, j, row, kw, ks, slot) = pre[ks][kw](lane, cw) + (row + 2 j) * HW * 64 + slot
        for lane in range(64):
            for cw in range(4):
                for kw in range(nkw):
                    for ks in range(2):
                        pre = _read_address(tab, HW, HH, lane, cw, 0, 0, kw, ks, 0)
                        for j in range(4):
                            for rr in rows:
                                assert _read_address(tab, HW, HH, lane, cw, j, rr * HW, kw, ks, 4096) == pre + (rr + 2 * j) * HW * 64 + 4096


def test_the_largest_immediate_fits_the_ds_offset_field():
    # (row + 2 j) * HW * 64 with row = kd * HH + kh: 3-tap box (2 * 10 + 2 + 6) * 18 * 64, parity box (1 * 9 + 1 + 6) * 17 * 64
    assert (2 * 10 + 2 + 6) * 18 * 64 < 65536 and (9 + 1 + 6) * 17 * 64 < 65536


def test_a_dma_piece_lands_where_the_reader_looks():
    """producer lane i of the halo's DMA stream writes LDS bytes [16 i, 16 i + 16): row i / 4, physical slot i % 4, and fetches channel slot
    ls = (i % 4) ^ key(column) of that row's voxel (make_pack); the reader of channel slot q = 2 ks + hk of a row looks at q ^ key(column)"""
    tabs = _tables()
    for HW, HVOX in ((18, 6 * 10 * 18), (17, 5 * 9 * 17)):
        tab = tabs[HW]
        for i in range(HVOX * 4):
            row, ps = i >> 2, i & 3
            ls = ps ^ _key(tab, row % HW)
            # the reader of logical slot ls of this row
            assert row * 64 + ((ls ^ _key(tab, row % HW)) << 4) == i * 16
