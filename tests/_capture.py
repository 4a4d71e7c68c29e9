"""Helpers for the capture-file tests: parse pcap / pcapng written for LINKTYPE_BLUETOOTH_BREDR_BB
and blank the bytes the reference leaves undefined (stack garbage), so files can be compared."""
import struct


def normalize_pcapng(data):
    """Zero (a) the padding between an enhanced packet block's data and its options word and
    (b) the 4 bytes the reference copies beyond its BD_ADDR / clock option structs."""
    b = bytearray(data)
    pos = 0
    while pos + 12 <= len(b):
        btype, blen = struct.unpack_from("<II", b, pos)
        assert blen >= 12 and blen % 4 == 0 and pos + blen <= len(b), (pos, btype, blen)
        assert struct.unpack_from("<I", b, pos + blen - 4)[0] == blen
        if btype == 6:                                   # enhanced packet block
            caplen = struct.unpack_from("<I", b, pos + 20)[0]
            for i in range(pos + 28 + caplen, pos + blen - 8):
                b[i] = 0
        elif btype == 1:                                 # interface description block
            o = pos + 16
            while o + 4 <= pos + blen - 4:
                code, olen = struct.unpack_from("<HH", b, o)
                if code == 0:
                    break
                if code == 0xD340 and olen == 12:
                    b[o + 4 + 8:o + 4 + 12] = bytes(4)
                if code == 0xD341 and olen == 24:
                    b[o + 4 + 20:o + 4 + 24] = bytes(4)
                o += 4 + 4 * ((olen + 3) // 4)
        pos += blen
    assert pos == len(b)
    return bytes(b)


def pcapng_blocks(data):
    out, pos = [], 0
    while pos < len(data):
        btype, blen = struct.unpack_from("<II", data, pos)
        out.append((btype, data[pos:pos + blen]))
        pos += blen
    return out


def pcap_records(data):
    """[(ts_sec, ts_nsec, record bytes)] of a classic pcap file; checks the file header."""
    magic, vmaj, vmin, zone, sigfigs, snaplen, network = struct.unpack_from("<IHHiIII", data, 0)
    assert (magic, vmaj, vmin, zone, sigfigs, snaplen, network) == (0xA1B23C4D, 2, 4, 0, 0, 400, 255)
    out, pos = [], 24
    while pos < len(data):
        sec, nsec, incl, orig = struct.unpack_from("<IIII", data, pos)
        assert incl == orig
        out.append((sec, nsec, data[pos + 16:pos + 16 + incl]))
        pos += 16 + incl
    return out
