"""Host-side Page codec (numpy) for the reference's columnar format.

Mirrors, for test/bench data preparation on the host:
  * the page-fill rule of ``ColumnInserter<T>::insert/insert_null``
    (reference include/plan.h:204-221) == ``Table::to_columnar``
    (reference src/build_table.cpp:488,495,531,538,574,581), and
  * the page layout read by ``Table::from_columnar``
    (reference src/build_table.cpp:325-428).

Layout of a fixed-width page (8192 B): u16 n_rows @0, u16 n_nonnull @2, values
dense from @4 (INT32) / @8 (INT64, FP64), validity bitmap (bit i = row i is
non-NULL, LSB first) in the LAST (n_rows+7)//8 bytes.

This module is plain host logic (no GPU, no oracle); it is validated against the
oracle's restatement of the reference encoders/decoders in tests/test_pages.py.
"""
from __future__ import annotations

import numpy as np

PAGE_SIZE = 8192
INT32, INT64, FP64, VARCHAR = 0, 1, 2, 3
NP_DTYPE = {INT32: np.int32, INT64: np.int64, FP64: np.float64}
HDR = {INT32: 4, INT64: 8, FP64: 8}
WIDTH = {INT32: 4, INT64: 8, FP64: 8}


def rows_per_full_page(dtype: int) -> int:
    """Rows in a page with no NULLs: largest n with hdr + n*w + (n-1)//8 + 1 <= 8192."""
    hdr, w = HDR[dtype], WIDTH[dtype]
    n = 0
    while hdr + (n + 1) * w + (n // 8 + 1) <= PAGE_SIZE:
        n += 1
    return n  # 1984 for INT32, 1007 for INT64/FP64


def _page_boundaries(valid: np.ndarray | None, n: int, dtype: int) -> np.ndarray:
    """Start row of every page under the reference's greedy fill rule."""
    cap = rows_per_full_page(dtype)
    if valid is None or bool(valid.all()):
        return np.arange(0, n, cap, dtype=np.int64)
    hdr, w = HDR[dtype], WIDTH[dtype]
    cv = np.concatenate([[0], np.cumsum(valid.astype(np.int64))])
    starts = []
    s = 0
    # a page can hold at most (8192-hdr)*8 rows (all NULL) and n_rows is a u16
    win = 65536
    while s < n:
        starts.append(s)
        e = min(n, s + win)
        k = np.arange(0, e - s, dtype=np.int64)
        nvb = cv[s:e] - cv[s]
        cost = hdr + (nvb + valid[s:e].astype(np.int64)) * w + (k // 8 + 1)
        over = np.nonzero(cost > PAGE_SIZE)[0]
        # the reference never lets n_rows wrap its uint16_t for realistic data;
        # keep below the long-string markers 0xfffe/0xffff
        limit = min(e - s, 0xFFFD)
        nrows = int(over[0]) if over.size else limit
        nrows = min(nrows, limit)
        s += max(nrows, 1)
    return np.asarray(starts, dtype=np.int64)


def pack_fixed(values, valid=None, dtype: int = INT32) -> np.ndarray:
    """Pack a fixed-width column into pages -> uint8 array [n_pages, 8192]."""
    npdt = NP_DTYPE[dtype]
    values = np.ascontiguousarray(values, dtype=npdt)
    n = values.shape[0]
    hdr, w = HDR[dtype], WIDTH[dtype]
    if valid is not None:
        valid = np.ascontiguousarray(valid).astype(bool)
        if valid.all():
            valid = None
    if n == 0:
        return np.zeros((0, PAGE_SIZE), dtype=np.uint8)
    if valid is None:
        cap = rows_per_full_page(dtype)
        npages = (n + cap - 1) // cap
        pages = np.zeros((npages, PAGE_SIZE), dtype=np.uint8)
        full = n // cap
        body = pages[:, hdr : hdr + cap * w].view(npdt)
        if full:
            body[:full, :] = values[: full * cap].reshape(full, cap)
        rem = n - full * cap
        if rem:
            body[full, :rem] = values[full * cap :]
        counts = np.full(npages, cap, dtype=np.uint16)
        if rem:
            counts[-1] = rem
        hv = pages[:, :4].view(np.uint16)
        hv[:, 0] = counts
        hv[:, 1] = counts
        # bitmap: all ones for n_rows bits, at the page tail
        for p_cnt in np.unique(counts):
            nb = (int(p_cnt) + 7) // 8
            bm = np.full(nb, 0xFF, dtype=np.uint8)
            if p_cnt % 8:
                bm[-1] = (1 << (int(p_cnt) % 8)) - 1
            pages[counts == p_cnt, PAGE_SIZE - nb :] = bm
        return pages
    starts = _page_boundaries(valid, n, dtype)
    ends = np.concatenate([starts[1:], [n]])
    pages = np.zeros((len(starts), PAGE_SIZE), dtype=np.uint8)
    for pi, (s, e) in enumerate(zip(starts, ends)):
        v = valid[s:e]
        nr = int(e - s)
        vals = values[s:e][v]
        nv = vals.shape[0]
        hv = pages[pi, :4].view(np.uint16)
        hv[0] = nr
        hv[1] = nv
        pages[pi, hdr : hdr + nv * w] = vals.view(np.uint8)
        bm = np.packbits(v.astype(np.uint8), bitorder="little")
        pages[pi, PAGE_SIZE - bm.shape[0] :] = bm
    return pages


def unpack_fixed(pages: np.ndarray, num_rows: int, dtype: int):
    """Decode pages -> (values[num_rows], valid[num_rows]); raises on row overflow
    like the reference's ``throw std::runtime_error("row_idx")``."""
    npdt = NP_DTYPE[dtype]
    hdr, w = HDR[dtype], WIDTH[dtype]
    values = np.zeros(num_rows, dtype=npdt)
    valid = np.zeros(num_rows, dtype=bool)
    pages = np.ascontiguousarray(pages, dtype=np.uint8).reshape(-1, PAGE_SIZE)
    if pages.shape[0] == 0:
        return values, valid
    hv = pages[:, :4].copy().view(np.uint16)
    nr_all = hv[:, 0].astype(np.int64)
    cap = rows_per_full_page(dtype)
    tot = int(nr_all.sum())
    if tot <= num_rows and (nr_all[:-1] == cap).all() and 0 < nr_all[-1] <= cap and _all_valid(pages, nr_all):
        # dense fast path: full pages, every validity bit set
        body = pages[:, hdr : hdr + cap * w].copy().view(npdt)
        values[:tot] = body.reshape(-1)[:tot]
        valid[:tot] = True
        return values, valid
    # general path = the reference's loop (src/build_table.cpp:326-343): decoded from the bitmap
    # alone (the header's non-null count is never read); "row_idx" only for a NON-NULL value at
    # a row index >= num_rows, NULL rows past the end just advance the row counter
    row = 0
    for pi in range(pages.shape[0]):
        nr = int(nr_all[pi])
        nb = (nr + 7) // 8
        bits = np.unpackbits(pages[pi, PAGE_SIZE - nb :], bitorder="little")[:nr].astype(bool)
        idx = np.nonzero(bits)[0]
        if idx.shape[0] and row + int(idx[-1]) >= num_rows:
            raise RuntimeError("row_idx")
        vals = pages[pi, hdr : hdr + idx.shape[0] * w].copy().view(npdt)
        values[row + idx] = vals[: idx.shape[0]]
        valid[row + idx] = True
        row += nr
    return values, valid


def _all_valid(pages: np.ndarray, nr_all: np.ndarray) -> bool:
    """Every validity bit of every page's rows set?"""
    for cnt in np.unique(nr_all):
        cnt = int(cnt)
        nb = (cnt + 7) // 8
        want = np.full(nb, 0xFF, dtype=np.uint8)
        if cnt % 8:
            want[-1] = (1 << (cnt % 8)) - 1
        sel = pages[nr_all == cnt, PAGE_SIZE - nb :]
        if nb and not ((sel & want) == want).all():
            return False
    return True


# ------------------------------------------------------------------ VARCHAR --
def pack_varchar(strings) -> np.ndarray:
    """Pack a list of ``bytes | str | None`` (reference plan.h:230-335 /
    build_table.cpp:595-677 rules, incl. long-string pages)."""
    pages: list[np.ndarray] = []
    state = {"nr": 0, "offs": [], "chars": bytearray(), "bits": []}

    def save_page():
        p = np.zeros(PAGE_SIZE, dtype=np.uint8)
        nr, offs, chars, bits = state["nr"], state["offs"], state["chars"], state["bits"]
        hv = p[:4].view(np.uint16)
        hv[0] = nr
        hv[1] = len(offs)
        if offs:
            p[4 : 4 + 2 * len(offs)] = np.asarray(offs, dtype=np.uint16).view(np.uint8)
        p[4 + 2 * len(offs) : 4 + 2 * len(offs) + len(chars)] = np.frombuffer(bytes(chars), dtype=np.uint8)
        bm = np.packbits(np.asarray(bits, dtype=np.uint8), bitorder="little")
        p[PAGE_SIZE - bm.shape[0] :] = bm
        pages.append(p)
        state.update(nr=0, offs=[], chars=bytearray(), bits=[])

    for s in strings:
        if s is None:
            if 4 + 2 * len(state["offs"]) + len(state["chars"]) + (state["nr"] // 8 + 1) > PAGE_SIZE:
                save_page()
            state["bits"].append(0)
            state["nr"] += 1
            continue
        b = s.encode() if isinstance(s, str) else bytes(s)
        if len(b) > PAGE_SIZE - 7:
            if state["nr"] > 0:
                save_page()
            off, first = 0, True
            while off < len(b):
                p = np.zeros(PAGE_SIZE, dtype=np.uint8)
                chunk = min(len(b) - off, PAGE_SIZE - 4)
                hv = p[:4].view(np.uint16)
                hv[0] = 0xFFFF if first else 0xFFFE
                hv[1] = chunk
                p[4 : 4 + chunk] = np.frombuffer(b[off : off + chunk], dtype=np.uint8)
                pages.append(p)
                first = False
                off += chunk
            continue
        if 4 + (len(state["offs"]) + 1) * 2 + len(state["chars"]) + len(b) + (state["nr"] // 8 + 1) > PAGE_SIZE:
            save_page()
        state["chars"] += b
        state["offs"].append(len(state["chars"]))
        state["bits"].append(1)
        state["nr"] += 1
    if state["nr"]:
        save_page()
    if not pages:
        return np.zeros((0, PAGE_SIZE), dtype=np.uint8)
    return np.stack(pages)


def pack_varchar_fixed(codes, digits: int = 10, prefix: bytes = b"") -> np.ndarray:
    """Vectorised VARCHAR packer for large synthetic columns: row i is `prefix` + the
    zero-padded decimal `codes[i]` (all strings the same length, no NULLs), packed with the
    reference's fill rule (plan.h:301-320), which for equal lengths gives a constant number
    of rows per page."""
    codes = np.asarray(codes).astype(np.uint64)
    n = codes.shape[0]
    L = len(prefix) + digits
    if n == 0:
        return np.zeros((0, PAGE_SIZE), dtype=np.uint8)
    ch = np.empty((n, L), dtype=np.uint8)
    if prefix:
        ch[:, : len(prefix)] = np.frombuffer(prefix, dtype=np.uint8)
    c = codes.copy()
    for d in range(digits - 1, -1, -1):
        ch[:, len(prefix) + d] = (48 + (c % 10)).astype(np.uint8)
        c //= 10
    R = 0  # rows per full page: row k (0-based) fits iff 4 + 2(k+1) + (k+1)L + k//8 + 1 <= 8192
    while 4 + 2 * (R + 1) + (R + 1) * L + (R // 8 + 1) <= PAGE_SIZE:
        R += 1
    npages = (n + R - 1) // R
    pages = np.zeros((npages, PAGE_SIZE), dtype=np.uint8)
    full = n // R

    def fill(dst, r, chars):  # dst [k, 8192] pages each holding exactly r rows
        hv = dst[:, :4].view(np.uint16)
        hv[:, 0] = r
        hv[:, 1] = r
        dst[:, 4 : 4 + 2 * r] = ((np.arange(1, r + 1, dtype=np.uint16)) * L).astype(np.uint16).view(np.uint8)
        dst[:, 4 + 2 * r : 4 + 2 * r + r * L] = chars
        nb = (r + 7) // 8
        bm = np.full(nb, 0xFF, dtype=np.uint8)
        if r % 8:
            bm[-1] = (1 << (r % 8)) - 1
        dst[:, PAGE_SIZE - nb :] = bm

    if full:
        fill(pages[:full], R, ch[: full * R].reshape(full, R * L))
    rem = n - full * R
    if rem:
        fill(pages[full : full + 1], rem, ch[full * R :].reshape(1, rem * L))
    return pages


def unpack_varchar(pages: np.ndarray, num_rows: int) -> list:
    """Decode VARCHAR pages -> list of ``bytes | None`` of length num_rows."""
    out: list = [None] * num_rows
    pages = np.ascontiguousarray(pages, dtype=np.uint8).reshape(-1, PAGE_SIZE)
    row = 0
    for pi in range(pages.shape[0]):
        p = pages[pi]
        nr = int(p[:2].view(np.uint16)[0])
        n2 = int(p[2:4].view(np.uint16)[0])
        if nr == 0xFFFF:
            if row >= num_rows:
                raise RuntimeError("row_idx")
            out[row] = p[4 : 4 + n2].tobytes()
            row += 1
        elif nr == 0xFFFE:
            if row == 0 or out[row - 1] is None:
                raise RuntimeError("long string page 0xfffe must follows a string")
            out[row - 1] = out[row - 1] + p[4 : 4 + n2].tobytes()
        else:
            nb = (nr + 7) // 8
            bits = np.unpackbits(p[PAGE_SIZE - nb :], bitorder="little")[:nr]
            offs = p[4 : 4 + 2 * n2].copy().view(np.uint16)
            base = 4 + 2 * n2
            prev, di = 0, 0
            for i in range(nr):
                if row >= num_rows:
                    raise RuntimeError("row_idx")
                if bits[i]:
                    end = int(offs[di])
                    out[row] = p[base + prev : base + end].tobytes()
                    prev = end
                    di += 1
                row += 1
    return out
