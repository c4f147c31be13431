"""A small FITS reader/writer for the DESI driver (desi/desi_fit.py).

The reference does its file I/O through astropy.io.fits (desi_fit.py:22-24,
459-492, 1302-1308), which is not part of this image.  This module implements
the subset of the FITS standard those files use: a primary HDU, IMAGE
extensions (BITPIX 8/16/32/64/-32/-64, BZERO/BSCALE), BINTABLE extensions with
fixed-width columns (L X B I J K A E D C M, repeat counts, TDIM, TZERO/TSCAL,
TUNIT), long strings through CONTINUE cards and the CHECKSUM/DATASUM keywords
(the reference writes with checksum=True).  Variable-length array columns
(P/Q) are carried as raw descriptors and not decoded.
"""
import builtins
import os

import numpy as np

BLOCK = 2880

_TFORM = {
    'L': ('i1', 1), 'X': ('u1', 1), 'B': ('u1', 1), 'I': ('>i2', 2),
    'J': ('>i4', 4), 'K': ('>i8', 8), 'A': ('S', 1), 'E': ('>f4', 4),
    'D': ('>f8', 8), 'C': ('>c8', 8), 'M': ('>c16', 16), 'P': ('>i4', 8),
    'Q': ('>i8', 16)
}
_BITPIX = {8: 'u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}
_DT2BITPIX = {'u1': 8, 'i2': 16, 'i4': 32, 'i8': 64, 'f4': -32, 'f8': -64}


# ----------------------------------------------------------------- header
class Header:
    """Ordered keyword -> (value, comment); keywords are unique except
    COMMENT/HISTORY which are kept as a list of lines."""

    def __init__(self, cards=()):
        self._k = []
        self._v = {}
        self._c = {}
        self.comments = []
        for k, v, c in cards:
            self.set(k, v, c)

    def set(self, key, value, comment=''):
        key = key.upper()
        if key not in self._v:
            self._k.append(key)
        self._v[key] = value
        self._c[key] = comment or ''

    def __setitem__(self, key, value):
        if isinstance(value, tuple):
            self.set(key, value[0], value[1])
        else:
            self.set(key, value, self._c.get(key.upper(), ''))

    def __getitem__(self, key):
        return self._v[key.upper()]

    def __contains__(self, key):
        return key.upper() in self._v

    def get(self, key, default=None):
        return self._v.get(key.upper(), default)

    def comment(self, key):
        return self._c.get(key.upper(), '')

    def keys(self):
        return list(self._k)

    def items(self):
        return [(k, self._v[k]) for k in self._k]

    def cards(self):
        return [(k, self._v[k], self._c[k]) for k in self._k]

    def copy(self):
        return Header(self.cards())

    def __delitem__(self, key):
        key = key.upper()
        if key in self._v:
            self._k.remove(key)
            del self._v[key]
            del self._c[key]


def _parse_value(s):
    """value/comment field of a card (columns 11-80)"""
    t = s.lstrip()
    if t.startswith("'"):
        # quoted string; '' is an escaped quote
        i, out = 1, []
        while i < len(t):
            if t[i] == "'":
                if i + 1 < len(t) and t[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                break
            out.append(t[i])
            i += 1
        val = ''.join(out).rstrip()
        rest = t[i + 1:]
        com = rest.split('/', 1)[1].strip() if '/' in rest else ''
        return val, com
    if '/' in t:
        v, com = t.split('/', 1)
    else:
        v, com = t, ''
    v, com = v.strip(), com.strip()
    if v == 'T':
        return True, com
    if v == 'F':
        return False, com
    if v == '':
        return None, com
    try:
        return int(v), com
    except ValueError:
        pass
    try:
        return float(v.replace('D', 'E').replace('d', 'e')), com
    except ValueError:
        return v, com


def _read_header(buf, pos):
    hdr = Header()
    last_str_key = None
    while True:
        block = buf[pos:pos + BLOCK]
        if len(block) < BLOCK:
            raise EOFError
        pos += BLOCK
        done = False
        for i in range(0, BLOCK, 80):
            card = block[i:i + 80].decode('ascii', 'replace')
            key = card[:8].rstrip()
            if key == 'END':
                done = True
                break
            if key == 'CONTINUE' and last_str_key is not None:
                v, c = _parse_value(card[8:])
                cur = hdr[last_str_key]
                if isinstance(cur, str) and cur.endswith('&'):
                    cur = cur[:-1]
                hdr.set(last_str_key, cur + (v if isinstance(v, str) else ''),
                        hdr.comment(last_str_key) or c)
                continue
            if key in ('COMMENT', 'HISTORY', ''):
                if card.strip():
                    hdr.comments.append(card.rstrip())
                continue
            if card[8:10] != '= ':
                continue
            v, c = _parse_value(card[10:])
            hdr.set(key, v, c)
            last_str_key = key if isinstance(v, str) else None
        if done:
            break
    for k, v in hdr.items():
        if isinstance(v, str):
            if v.endswith('&'):
                v = v[:-1]
            hdr.set(k, v.rstrip(), hdr.comment(k))
    return hdr, pos


def _fmt_value(v):
    if isinstance(v, (bool, np.bool_)):
        return '%20s' % ('T' if v else 'F')
    if isinstance(v, (int, np.integer)):
        return '%20d' % int(v)
    if isinstance(v, (float, np.floating)):
        s = repr(float(v)).upper()
        if 'E' not in s and '.' not in s and 'N' not in s:
            s += '.0'
        return '%20s' % s
    raise TypeError(type(v))


def _cards_for(key, value, comment):
    key = key.upper()
    if len(key) > 8:
        raise ValueError('keyword longer than 8 characters: ' + key)
    if value is None:
        return [('%-8s' % key).ljust(80)]
    if isinstance(value, str):
        esc = value.replace("'", "''")
        if len(esc) <= 68:
            body = "'%-8s'" % esc
            card = '%-8s= %-20s' % (key, body)
            if comment:
                card += ' / ' + comment
            return [card[:80].ljust(80)]
        # long string: CONTINUE convention, 67 characters + '&' per card
        parts = [esc[i:i + 67] for i in range(0, len(esc), 67)]
        out = []
        for i, p in enumerate(parts):
            amp = '&' if i < len(parts) - 1 else ''
            if i == 0:
                out.append(("%-8s= '%s%s'" % (key, p, amp)).ljust(80))
            else:
                out.append(("CONTINUE  '%s%s'" % (p, amp)).ljust(80))
        if comment:
            out[-1] = (out[-1].rstrip() + ' / ' + comment)[:80].ljust(80)
        return out
    card = '%-8s= %s' % (key, _fmt_value(value))
    if comment:
        card += ' / ' + comment
    return [card[:80].ljust(80)]


def _header_bytes(cards):
    lines = []
    for k, v, c in cards:
        lines += _cards_for(k, v, c)
    lines.append('END'.ljust(80))
    s = ''.join(lines)
    s += ' ' * ((-len(s)) % BLOCK)
    return s.encode('ascii')


# --------------------------------------------------------------- checksum
def _sum32(data, total=0):
    """ones-complement 32-bit sum of big-endian words (FITS checksum)"""
    if len(data) % 4:
        data = data + b'\0' * (4 - len(data) % 4)
    w = np.frombuffer(data, dtype='>u4')
    s = int(w.astype(np.uint64).sum()) + total
    while s >> 32:
        s = (s & 0xFFFFFFFF) + (s >> 32)
    return s


_EXCLUDE = list(range(0x3a, 0x41)) + list(range(0x5b, 0x61))


def _encode_checksum(value, complement=True):
    if complement:
        value = 0xFFFFFFFF - value
    asc = [0] * 16
    for i in range(4):
        byte = (value >> (24 - 8 * i)) & 0xFF
        quot, rem = byte // 4 + 0x30, byte % 4
        ch = [quot + rem, quot, quot, quot]
        check = True
        while check:
            check = False
            for j in (0, 2):
                if ch[j] in _EXCLUDE or ch[j + 1] in _EXCLUDE:
                    ch[j] += 1
                    ch[j + 1] -= 1
                    check = True
        for j in range(4):
            asc[4 * j + i] = ch[j]
    asc = asc[-1:] + asc[:-1]
    return ''.join(chr(_) for _ in asc)


def hdu_checksum_ok(raw_header, raw_data, header):
    """True when the HDU bytes sum to -0 and DATASUM matches (both keywords
    present); used by the tests against files written by astropy."""
    ds = _sum32(raw_data)
    ok = int(str(header.get('DATASUM', '-1'))) == ds
    return ok and _sum32(raw_header, ds) == 0xFFFFFFFF


# ------------------------------------------------------------------- HDUs
class ImageHDU:

    def __init__(self, data=None, header=None, name=None):
        self.data = None if data is None else np.asarray(data)
        self.header = header if header is not None else Header()
        self.name = name if name is not None else self.header.get('EXTNAME', '')
        self.is_primary = False

    def _struct_cards(self):
        if self.data is None:
            shape, bitpix, dat = (), 8, b''
        else:
            a = self.data
            if a.dtype == np.bool_:
                a = a.astype('u1')
            code = a.dtype.str[1:]
            if code not in _DT2BITPIX:
                raise TypeError('cannot write dtype %s as a FITS image'
                                % a.dtype)
            bitpix = _DT2BITPIX[code]
            shape = a.shape
            dat = np.ascontiguousarray(a.astype('>' + code if code != 'u1'
                                                else 'u1')).tobytes()
        if self.is_primary:
            cards = [('SIMPLE', True, 'conforms to FITS standard')]
        else:
            cards = [('XTENSION', 'IMAGE', 'Image extension')]
        cards += [('BITPIX', bitpix, 'array data type'),
                  ('NAXIS', len(shape), 'number of array dimensions')]
        for i, n in enumerate(shape[::-1]):
            cards.append(('NAXIS%d' % (i + 1), int(n), ''))
        if self.is_primary:
            cards.append(('EXTEND', True, ''))
        else:
            cards += [('PCOUNT', 0, 'number of parameters'),
                      ('GCOUNT', 1, 'number of groups')]
        return cards, dat


class PrimaryHDU(ImageHDU):

    def __init__(self, data=None, header=None):
        ImageHDU.__init__(self, data, header, name='PRIMARY')
        self.is_primary = True


class Column:

    def __init__(self, name, array, unit='', tform=None, tdim=None, null=None):
        self.name, self.array, self.unit = name, array, unit or ''
        self.tform, self.tdim, self.null = tform, tdim, null


class Columns:
    """the `.columns` attribute of a table (names / formats / units)"""

    def __init__(self, cols):
        self._cols = cols

    @property
    def names(self):
        return [c.name for c in self._cols]

    @property
    def formats(self):
        return [c.tform for c in self._cols]

    @property
    def units(self):
        return [c.unit for c in self._cols]


class FitsTable:
    """Column store with the parts of the astropy FITS_rec interface the
    driver uses: tab['COL'], tab.columns.names, len(tab), tab[mask]."""

    def __init__(self, cols=()):
        self._cols = list(cols)

    @property
    def columns(self):
        return Columns(self._cols)

    def __len__(self):
        return len(self._cols[0].array) if self._cols else 0

    def __contains__(self, name):
        return name in self.columns.names

    def column(self, name):
        for c in self._cols:
            if c.name == name:
                return c
        raise KeyError(name)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.column(key).array
        if isinstance(key, (int, np.integer)):
            return {c.name: c.array[key] for c in self._cols}
        key = np.asarray(key)
        return FitsTable([
            Column(c.name, c.array[key], c.unit, c.tform, c.tdim, c.null)
            for c in self._cols
        ])

    def add(self, name, array, unit='', tform=None):
        self._cols.append(Column(name, np.asarray(array), unit, tform))


def _tform_of(a):
    a = np.asarray(a)
    k = a.dtype.kind
    rep = int(np.prod(a.shape[1:])) if a.ndim > 1 else 1
    r = '' if rep == 1 else str(rep)
    if k == 'b':
        return r + 'L'
    if k in 'SU':
        n = a.dtype.itemsize // (4 if k == 'U' else 1)
        return '%dA' % max(n * rep, 1)
    code = {'u1': 'B', 'i2': 'I', 'i4': 'J', 'i8': 'K', 'f4': 'E', 'f8': 'D',
            'c8': 'C', 'c16': 'M'}.get(a.dtype.str[1:])
    if code is None:
        raise TypeError('cannot write dtype %s as a FITS column' % a.dtype)
    return r + code


def _parse_tform(tf):
    tf = tf.strip()
    i = 0
    while i < len(tf) and tf[i].isdigit():
        i += 1
    rep = int(tf[:i]) if i else 1
    return rep, tf[i], tf[i + 1:]


class BinTableHDU:

    def __init__(self, data=None, header=None, name=None):
        self.data = data if data is not None else FitsTable()
        self.header = header if header is not None else Header()
        self.name = name if name is not None else self.header.get('EXTNAME', '')
        self.is_primary = False

    def _struct_cards(self):
        tab = self.data
        n = len(tab)
        fields, width = [], 0
        for c in tab._cols:
            a = np.asarray(c.array)
            tf = c.tform or _tform_of(a)
            rep, code, _ = _parse_tform(tf)
            if code == 'A':
                if a.dtype.kind == 'U':
                    a = np.char.encode(a, 'ascii')
                # NUL padding (legal FITS: a NUL terminates the string)
                b = np.zeros((n, ), dtype='S%d' % rep)
                if n:
                    b[:] = a.reshape(n)
                col = np.frombuffer(b.tobytes(), dtype='u1').reshape(n, rep)
            elif code == 'L':
                v = a.reshape(n, rep).astype(bool)
                col = np.where(v, ord('T'), ord('F')).astype('u1')
            else:
                dt, size = _TFORM[code]
                col = np.ascontiguousarray(a.reshape(n, rep).astype(dt))
                col = col.view('u1').reshape(n, rep * size)
            fields.append((c, tf, col))
            width += col.shape[1]
        rows = (np.concatenate([f[2] for f in fields], axis=1)
                if fields and n else np.zeros((n, width), dtype='u1'))
        cards = [('XTENSION', 'BINTABLE', 'binary table extension'),
                 ('BITPIX', 8, 'array data type'),
                 ('NAXIS', 2, 'number of array dimensions'),
                 ('NAXIS1', int(width), 'length of dimension 1'),
                 ('NAXIS2', int(n), 'length of dimension 2'),
                 ('PCOUNT', 0, 'number of group parameters'),
                 ('GCOUNT', 1, 'number of groups'),
                 ('TFIELDS', len(fields), 'number of table fields')]
        for i, (c, tf, _) in enumerate(fields):
            cards.append(('TTYPE%d' % (i + 1), c.name, ''))
            cards.append(('TFORM%d' % (i + 1), tf, ''))
            if c.unit:
                cards.append(('TUNIT%d' % (i + 1), c.unit, ''))
            if c.null is not None:
                cards.append(('TNULL%d' % (i + 1), int(c.null), ''))
            if c.tdim:
                cards.append(('TDIM%d' % (i + 1), c.tdim, ''))
        return cards, rows.tobytes()


_STRUCT = ('SIMPLE', 'XTENSION', 'BITPIX', 'NAXIS', 'EXTEND', 'PCOUNT', 'GCOUNT',
           'TFIELDS', 'CHECKSUM', 'DATASUM', 'EXTNAME', 'BZERO', 'BSCALE')


def _is_struct(k):
    return (k in _STRUCT or k.startswith('NAXIS')
            or k[:5] in ('TTYPE', 'TFORM', 'TUNIT', 'TNULL', 'TZERO', 'TSCAL')
            or k[:4] == 'TDIM')


class HDUList(list):

    def __getitem__(self, key):
        if isinstance(key, str):
            for h in self:
                if h.name == key.upper() or h.name == key:
                    return h
            raise KeyError(key)
        return list.__getitem__(self, key)

    def __contains__(self, key):
        if isinstance(key, str):
            return any(h.name == key for h in self)
        return list.__contains__(self, key)

    def close(self):
        pass

    def writeto(self, fname, overwrite=True, checksum=True):
        if os.path.exists(fname) and not overwrite:
            raise OSError('file exists: ' + fname)
        with builtins.open(fname, 'wb') as fp:
            for i, h in enumerate(self):
                h.is_primary = (i == 0)
                fp.write(hdu_bytes(h, checksum=checksum))


def hdu_bytes(h, checksum=True):
    cards, dat = h._struct_cards()
    dat += b'\0' * ((-len(dat)) % BLOCK)
    if not h.is_primary and h.name:
        cards.append(('EXTNAME', h.name, 'extension name'))
    for k, v, c in h.header.cards():
        if not _is_struct(k):
            cards.append((k, v, c))
    if checksum:
        ds = _sum32(dat)
        cards.append(('CHECKSUM', '0' * 16, 'HDU checksum'))
        cards.append(('DATASUM', str(ds), 'data unit checksum'))
        hb = _header_bytes(cards)
        cs = _encode_checksum(_sum32(hb, ds))
        cards[-2] = ('CHECKSUM', cs, 'HDU checksum')
    return _header_bytes(cards) + dat


# ----------------------------------------------------------------- reading
def _decode_table(hdr, raw):
    n, width = hdr['NAXIS2'], hdr['NAXIS1']
    rows = np.frombuffer(raw[:n * width], dtype='u1').reshape(n, width)
    cols, off = [], 0
    for i in range(1, hdr['TFIELDS'] + 1):
        tf = hdr['TFORM%d' % i]
        rep, code, _ = _parse_tform(tf)
        dt, size = _TFORM[code]
        nb = rep * size if code not in 'X' else (rep + 7) // 8
        chunk = np.ascontiguousarray(rows[:, off:off + nb])
        off += nb
        name = hdr.get('TTYPE%d' % i, 'COL%d' % i)
        if code == 'A':
            a = chunk.view('S%d' % max(nb, 1)).reshape(n) if nb else \
                np.zeros(n, dtype='S1')
            a = np.char.rstrip(a)
            a = np.char.decode(a, 'ascii')
        elif code == 'L':
            a = (chunk == ord('T'))
            a = a.reshape(n) if rep == 1 else a
        elif code in 'PQX':
            a = chunk
        else:
            a = chunk.view(dt).reshape(n, rep)
            a = a.astype(a.dtype.newbyteorder('='))
            tz, ts = hdr.get('TZERO%d' % i), hdr.get('TSCAL%d' % i)
            if tz is not None or ts is not None:
                tz, ts = tz or 0, ts or 1
                if ts == 1 and a.dtype.kind == 'i' and \
                        tz == 2**(8 * a.dtype.itemsize - 1):
                    a = (a.astype('u%d' % a.dtype.itemsize)
                         ^ np.array(tz, dtype='u%d' % a.dtype.itemsize))
                elif ts == 1 and a.dtype == np.uint8 and tz == -128:
                    a = (a.astype(np.int16) - 128).astype(np.int8)
                else:
                    a = a * ts + tz
            td = hdr.get('TDIM%d' % i)
            if td and rep > 1:
                dims = [int(_) for _ in td.strip('() ').split(',')]
                a = a.reshape([n] + dims[::-1])
            elif rep == 1:
                a = a.reshape(n)
        cols.append(Column(name, a, hdr.get('TUNIT%d' % i, ''), tf,
                           hdr.get('TDIM%d' % i), hdr.get('TNULL%d' % i)))
    return FitsTable(cols)


def _decode_image(hdr, raw):
    nax = hdr['NAXIS']
    if nax == 0:
        return None
    shape = [hdr['NAXIS%d' % (i + 1)] for i in range(nax)][::-1]
    dt = np.dtype(_BITPIX[hdr['BITPIX']])
    cnt = int(np.prod(shape))
    a = np.frombuffer(raw[:cnt * dt.itemsize], dtype=dt).reshape(shape)
    a = a.astype(dt.newbyteorder('='))
    bz, bs = hdr.get('BZERO'), hdr.get('BSCALE')
    if bz is not None or bs is not None:
        bz, bs = bz or 0, bs or 1
        if bs == 1 and a.dtype.kind == 'i' and \
                bz == 2**(8 * a.dtype.itemsize - 1):
            a = a.astype('u%d' % a.dtype.itemsize) ^ \
                np.array(bz, dtype='u%d' % a.dtype.itemsize)
        else:
            a = a * bs + bz
    return a


def open(fname, verify_checksum=False):  # noqa: A001 (mirrors pyfits.open)
    """Read every HDU of `fname` (plain or .gz) into an HDUList."""
    if fname.endswith('.gz'):
        import gzip
        with gzip.open(fname, 'rb') as fp:
            buf = fp.read()
    else:
        with builtins.open(fname, 'rb') as fp:
            buf = fp.read()
    out = HDUList()
    pos = 0
    while pos < len(buf):
        start = pos
        try:
            hdr, pos = _read_header(buf, pos)
        except EOFError:
            break
        first = len(out) == 0
        if not first and 'XTENSION' not in hdr:
            break
        nax = hdr.get('NAXIS', 0)
        size = 0
        if nax:
            size = abs(hdr['BITPIX']) // 8 * hdr.get('GCOUNT', 1) * (
                hdr.get('PCOUNT', 0)
                + int(np.prod([hdr['NAXIS%d' % (i + 1)] for i in range(nax)])))
        padded = size + (-size) % BLOCK
        raw = buf[pos:pos + padded]
        if verify_checksum and 'CHECKSUM' in hdr:
            if not hdu_checksum_ok(buf[start:pos], raw, hdr):
                raise OSError('checksum mismatch in HDU %d of %s'
                              % (len(out), fname))
        pos += padded
        xt = hdr.get('XTENSION', 'IMAGE').strip()
        if first:
            h = PrimaryHDU(_decode_image(hdr, raw), hdr)
        elif xt == 'BINTABLE':
            h = BinTableHDU(_decode_table(hdr, raw), hdr)
        elif xt == 'IMAGE':
            h = ImageHDU(_decode_image(hdr, raw), hdr)
        else:
            h = ImageHDU(None, hdr)
        out.append(h)
    return out
