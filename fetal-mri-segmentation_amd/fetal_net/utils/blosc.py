"""Decoder for Blosc-1 frames with the BloscLZ codec - what PyTables' `Filters(complib='blosc')` puts into the chunks of the reference's
data files (reference fetal_net/data.py:12-16; HDF5 filter id 32001).  The codec lives in a dependency that is absent from this stack
(c-blosc, bundled with PyTables: 1.20.1 with PyTables 3.6.1), so this module restates its published container and stream format; it is
pinned by known-answer vectors produced by that very library (tests/golden/blosc_frames_golden.npz, generator
tests/golden/make_pytables_fixture.py) and by a whole data file written through the reference's own functions.

Frame (c-blosc README_HEADER / blosc.c `blosc_d`):
    byte 0 version, 1 codec version, 2 flags (0x1 byte shuffle, 0x2 stored uncompressed, 0x4 bit shuffle, 0x10 blocks are not split,
    bits 5-7 codec: 0 = BloscLZ), 3 typesize; uint32 LE nbytes, blocksize, cbytes; then one int32 start offset per block.
    A block is `typesize` streams (one per byte plane of the shuffled block) when splitting applies - typesize <= 16, at least 128
    elements per block, not the short last block, flag 0x10 clear - else one stream; a stream = int32 compressed size + payload, stored
    raw when that size equals the plane size.  After decoding, a shuffled block is transposed back ([typesize][n] -> [n][typesize]).
BloscLZ stream (blosclz.c `blosclz_decompress`, the FastLZ level-2 layout): the first control byte is masked to a literal run; control
    c < 32: c + 1 literal bytes follow; otherwise a match of length (c >> 5) - 1 + 3 (length field 7: extended by following bytes, 255
    continues) at distance ((c & 31) << 8 | next byte) + 1, or, for the escape 31 / 255, a 16-bit distance + 8191 + 1; matches may
    overlap their own output (distance 1 = a run).

Only host-side metadata goes through here (the chunk of VLArray heap references is a few KB); the volumes themselves sit uncompressed in
the file's global heap, so a plain Python loop is adequate.
"""
import struct

import numpy as np

_MAX_SPLITS, _MIN_BUFFERSIZE, _MAX_DISTANCE = 16, 128, 8191


class BloscError(ValueError):
    pass


def blosclz_decompress(src, maxout):
    """one BloscLZ stream -> bytes (exactly what the encoder saw; raises BloscError on a malformed stream or more than `maxout` bytes)"""
    src = bytes(src)
    n = len(src)
    if n == 0:
        return b""
    out = bytearray(maxout)
    ip, op = 1, 0
    ctrl = src[0] & 31
    while True:
        if ctrl >= 32:
            length = (ctrl >> 5) - 1
            ofs = (ctrl & 31) << 8
            if length == 6:
                while True:
                    if ip >= n:
                        raise BloscError("BloscLZ: truncated match length")
                    code = src[ip]
                    ip += 1
                    length += code
                    if code != 255:
                        break
            if ip >= n:
                raise BloscError("BloscLZ: truncated match")
            code = src[ip]
            ip += 1
            length += 3
            dist = ofs + code
            if code == 255 and ofs == (31 << 8):
                if ip + 1 >= n:
                    raise BloscError("BloscLZ: truncated far distance")
                dist = ((src[ip] << 8) | src[ip + 1]) + _MAX_DISTANCE
                ip += 2
            ref = op - dist - 1
            if ref < 0 or op + length > maxout:
                raise BloscError("BloscLZ: match outside the buffer")
            if dist + 1 >= length:
                out[op:op + length] = out[ref:ref + length]
            else:                                           # overlapping copy: the pattern of the last dist + 1 bytes repeats
                pat = bytes(out[ref:op])
                reps = -(-length // len(pat))
                out[op:op + length] = (pat * reps)[:length]
            op += length
            if ip >= n:
                break
            ctrl = src[ip]
            ip += 1
        else:
            run = ctrl + 1
            if ip + run > n or op + run > maxout:
                raise BloscError("BloscLZ: literal run outside the buffer")
            out[op:op + run] = src[ip:ip + run]
            op += run
            ip += run
            if ip >= n:
                break
            ctrl = src[ip]
            ip += 1
    return bytes(out[:op])


def _unshuffle(block, typesize):
    n = len(block) // typesize
    body = np.frombuffer(block, dtype=np.uint8, count=n * typesize).reshape(typesize, n).T
    return body.tobytes() + block[n * typesize:]


def decompress(frame):
    """one Blosc-1 frame -> bytes"""
    frame = bytes(frame)
    if len(frame) < 16:
        raise BloscError("Blosc: frame shorter than its header")
    version, _versionlz, flags, typesize = frame[0], frame[1], frame[2], frame[3]
    nbytes, blocksize, cbytes = struct.unpack_from("<III", frame, 4)
    if version != 2:
        raise BloscError("Blosc: container version %d (only 2 is known)" % version)
    if cbytes > len(frame):
        raise BloscError("Blosc: frame truncated (%d of %d bytes)" % (len(frame), cbytes))
    if nbytes == 0:
        return b""
    if flags & 0x2:                                         # stored as is
        return frame[16:16 + nbytes]
    codec = flags >> 5
    if codec != 0:
        raise BloscError("Blosc: codec %d (%s) is not supported by this reader - only BloscLZ, PyTables' default for complib='blosc'; "
                         "convert the file with tools/convert_data_file.py" % (codec, {1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}.get(codec, "?")))
    if flags & 0x4:
        raise BloscError("Blosc: bit-shuffled frames are not supported by this reader")
    if blocksize <= 0 or typesize <= 0:
        raise BloscError("Blosc: bad header")
    nblocks = -(-nbytes // blocksize)
    bstarts = struct.unpack_from("<%di" % nblocks, frame, 16)
    dont_split = bool(flags & 0x10)
    out = []
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        split = (not dont_split) and (not leftover) and typesize <= _MAX_SPLITS and blocksize // typesize >= _MIN_BUFFERSIZE
        nsplits = typesize if split else 1
        neblock = bsize // nsplits
        pos = bstarts[b]
        parts = []
        for _ in range(nsplits):
            (csize,) = struct.unpack_from("<i", frame, pos)
            pos += 4
            if csize < 0 or pos + csize > len(frame):
                raise BloscError("Blosc: stream outside the frame")
            if csize == neblock:
                parts.append(frame[pos:pos + csize])
            else:
                part = blosclz_decompress(frame[pos:pos + csize], neblock)
                if len(part) != neblock:
                    raise BloscError("Blosc: stream decoded to %d bytes, %d expected" % (len(part), neblock))
                parts.append(part)
            pos += csize
        block = b"".join(parts)
        if (flags & 0x1) and typesize > 1:
            block = _unshuffle(block, typesize)
        out.append(block)
    data = b"".join(out)
    if len(data) != nbytes:
        raise BloscError("Blosc: decoded %d bytes, header says %d" % (len(data), nbytes))
    return data
